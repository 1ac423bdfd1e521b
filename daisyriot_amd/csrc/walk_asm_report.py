"""Static report on the hand-written BVH walks inside k_ff_tiles from the compiler's assembly listing (the Makefile beside this
file runs it on every build -> ../lib/geom_kernels.walk.txt; tests/test_abi_cpu.py reads it).  Three kernels: <256,false,2> = path
records + the walk over the sibling-pair records (its hand-written stretches: three streams with s_load_dwordx8, the pair walk
with s_load_dwordx16), <256,false,3> = the pair walk from the root, <256,false,0> = the threaded walk (s_load_dwordx8).  What it guards: SGPR spill reloads (v_readlane / v_writelane) placed by the register allocator right
before a walk's entry or right after its exit run once per LEAF visit -- measured +9 % kernel time when a change of the
surrounding code put 16 there.  (The pair walk's own stack uses v_readlane / v_writelane INSIDE the asm statement: not counted.)"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
spill = re.compile(r"\bv_(readlane|writelane)_b32\b")


def code(lines):
    return [l for l in lines if l.strip() and not l.strip().startswith(";") and not l.strip().endswith(":")]


def report(tag, walk_id, load):
    name = "_ZN2dr10k_ff_tilesILi256ELb0ELi%dEEEvNS_10TileParamsE:" % walk_id
    try:
        a = next(i for i, l in enumerate(src) if l.startswith(name))
    except StopIteration:
        print(tag + "kernel_found 0")
        return
    b = next(i for i in range(a, len(src)) if "s_endpgm" in src[i])
    body = src[a:b]
    starts = [i for i, l in enumerate(body) if "#ASMSTART" in l]
    ends = [i for i, l in enumerate(body) if "#ASMEND" in l]
    walks = [(s, e) for s, e in zip(starts, ends) if any(load in l for l in body[s:e])]
    entry = sum(len([l for l in code(body[max(0, s - 40):s])[-12:] if spill.search(l)]) for s, e in walks)
    exit_ = sum(len([l for l in code(body[e:e + 60])[:20] if spill.search(l)]) for s, e in walks)
    print(tag + "kernel_found 1")
    print(tag + "walk_blocks", len(walks))
    print(tag + "walk_entry_spill_ops", entry)
    print(tag + "walk_exit_spill_ops", exit_)
    # vector instructions of one node test: from the first variant's load to its first branch on VCCZ
    n = -1
    for s, e in walks:
        blk = body[s:e]
        for i, l in enumerate(blk):
            if load in l:
                j = next((k for k in range(i, len(blk)) if "s_cbranch_vccz" in blk[k]), None)
                if j is not None:
                    n = sum(1 for x in blk[i:j] if x.strip().startswith("v_"))
                break
        break
    print(tag + "node_test_valu", n)


report("", 0, "s_load_dwordx8")
report("pairs_", 3, "s_load_dwordx16")
report("paths_", 2, "s_load_dwordx16")          # the pair walk behind the path records
report("paths_stream_", 2, "s_load_dwordx8")   # the three stretches of path records in front of it
