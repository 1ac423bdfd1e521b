"""Static report on the hand-written BVH walk inside k_ff_tiles<256,false,0> from the compiler's assembly listing
(daisyriot_amd/csrc/Makefile runs it on every build -> daisyriot_amd/lib/geom_kernels.walk.txt; tests/test_abi_cpu.py reads it).
What it guards: SGPR spill reloads (v_readlane / v_writelane) placed by the register allocator right before the walk's entry or
right after its exit run once per LEAF visit -- measured +9 % kernel time when a change of the surrounding code put 16 there."""
import sys, re
src = open(sys.argv[1]).read().split("\n")
name = "_ZN2dr10k_ff_tilesILi256ELb0ELi0EEEvNS_10TileParamsE:"
try:
    a = next(i for i, l in enumerate(src) if l.startswith(name))
except StopIteration:
    print("kernel_found 0"); sys.exit(0)
b = next(i for i in range(a, len(src)) if "s_endpgm" in src[i])
body = src[a:b]
starts = [i for i, l in enumerate(body) if "#ASMSTART" in l]
ends = [i for i, l in enumerate(body) if "#ASMEND" in l]
walks = [(s, e) for s, e in zip(starts, ends) if any("s_load_dwordx8" in l for l in body[s:e])]
spill = re.compile(r"\bv_(readlane|writelane)_b32\b")
def code(lines): return [l for l in lines if l.strip() and not l.strip().startswith(";") and not l.strip().endswith(":")]
entry = sum(len([l for l in code(body[max(0, s - 40):s])[-12:] if spill.search(l)]) for s, e in walks)
exit_ = sum(len([l for l in code(body[e:e + 60])[:20] if spill.search(l)]) for s, e in walks)
print("kernel_found 1")
print("walk_blocks", len(walks))
print("walk_entry_spill_ops", entry)
print("walk_exit_spill_ops", exit_)
print("node_test_valu", min((sum(1 for l in body[s:e][i:i + 16] if l.strip().startswith("v_")) for s, e in walks for i, l in enumerate(body[s:e]) if l.strip() == "00:"), default=-1))
