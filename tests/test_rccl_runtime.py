"""One RCCL (and one HIP runtime) per process, whatever the import order.

Round 2 ended a GPU test session with `double free or corruption` at interpreter exit (gpurun_out/r2_crash.log): the library
had bound /opt/rocm's RCCL with RTLD_GLOBAL, a later test imported torch, torch mapped the copy bundled with its wheel, the
first copy's symbols interposed the second's and the two runtimes tore each other down at exit.  Fixed at the source:
daisyriot_amd/csrc/dr_comm.cpp opens RCCL privately (RTLD_LOCAL), and daisyriot_amd/api.py hands the library the copy torch
itself will use (dr_comm_set_library) and loads the wheel's HIP runtime first, so a Python process has ONE of each.  The child
processes below make a communicator (or at least bind RCCL) BEFORE importing torch and must exit 0."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %r)
from daisyriot_amd import api
want_gpu = %d
uid = api.comm_unique_id()                 # binds RCCL (dlopen) -- before torch is imported
if want_gpu:
    c = api.Context(0)
    c.set_shard(0, 1)
    c.comm_init(uid, 0, 1)                 # a real communicator
    assert c.comm_info() == (0, 1)
assert "torch" not in sys.modules
import torch                               # maps whatever torch needs
bound, mapped = api.comm_library_info()
hip = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
print("BOUND", bound); print("MAPPED", mapped); print("HIP", hip)
assert len(mapped) == 1 and mapped[0] == bound, (bound, mapped)
assert len(hip) == 1, hip
if want_gpu:
    assert torch.cuda.is_available()
    x = torch.ones(8, device="cuda").sum().item()      # torch's runtime works beside the communicator
    assert x == 8.0
    c.close()
print("CHILD_OK")
"""


def _run(want_gpu):
    env = dict(os.environ)
    env.pop("DR_RCCL_LIB", None)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, want_gpu)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, "rc %d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-3000:])


def test_bind_rccl_then_import_torch_exits_cleanly():
    """no GPU needed: the unique id binds the library; then torch comes in; one RCCL, one HIP runtime, exit code 0"""
    pytest.importorskip("torch")
    _run(0)


@pytest.mark.gpu
def test_communicator_before_torch_exits_cleanly():
    pytest.importorskip("torch")
    _run(1)


def test_set_library_after_binding_is_refused():
    from daisyriot_amd import api
    L = api.load_library()
    api.comm_library_info()                       # binds
    assert L.dr_comm_set_library(b"/nonexistent/librccl.so") != 0
    assert b"already bound" in L.dr_last_error()
