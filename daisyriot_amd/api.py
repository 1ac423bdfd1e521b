"""ctypes binding of libdaisyriot_hip.so (include/daisyriot_hip.h).

This is the only way Python reaches the hot path: every call goes through the
C ABI.  There is no CPU fallback -- if the HIP library is missing or a call
fails, an exception is raised.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdaisyriot_hip.so")

RULE_INTEGRAND = 0       # ini cuda_on = true  (vs/OptixPrimeFunctionality.cpp:6-34)
RULE_RECIPROCITY = 1     # ini cuda_on = false (vs/OptixPrimeFunctionality.cpp:311-366)
RAYS_PER_PATCH = 50      # vs/Defines.h:25
ORIGIN_EPS = 1e-6        # vs/OptixPrimeFunctionality.cpp:194
MAX_BINS = 16
DISPLAY_BW, DISPLAY_RGB, DISPLAY_SPECTRAL = 0, 1, 2   # vs/Lightning.h:406-408, 332-334, 168-183

EXPORTS = [
    "dr_last_error", "dr_context_create", "dr_context_destroy", "dr_set_stream", "dr_set_shard",
    "dr_get_shard", "dr_scene_set_mesh", "dr_formfactors_assemble", "dr_formfactors_integrand_only",
    "dr_formfactors_read_rows", "dr_visibility_read_rows", "dr_formfactors_load_rows", "dr_solver_init",
    "dr_solver_step", "dr_solver_converge", "dr_solver_reset", "dr_solver_read", "dr_solver_residual_sums",
    "dr_comm_unique_id", "dr_comm_init", "dr_get_info", "dr_profile_enable", "dr_profile_reset",
    "dr_synchronize", "dr_debug_read_bvh", "dr_shard_rows", "dr_residual_offset",
    "dr_comm_manual", "dr_exchange_export", "dr_exchange_import", "dr_debug_read_array", "dr_debug_sah_topology",
    "dr_display_patch_colors", "dr_display_vertex_colors",
    "dr_formfactors_assemble_split", "dr_vis_exchange_bytes", "dr_vis_exchange_export", "dr_vis_exchange_import",
    "dr_formfactors_assemble_finish", "dr_vis_exchange_tracer", "dr_solver_skip_zero_blocks",
    "dr_vis_exchange_fits", "dr_residual_chunk_floats", "dr_solver_set_check_interval", "dr_comm_info",
    "dr_group_create", "dr_group_destroy", "dr_group_info", "dr_group_context", "dr_group_set_mesh",
    "dr_group_assemble", "dr_group_solver_init", "dr_group_solver_step", "dr_group_solver_converge",
    "dr_group_solver_reset", "dr_group_solver_read", "dr_group_synchronize",
    "dr_comm_set_library", "dr_comm_library_info",
    "dr_options_defaults", "dr_set_options", "dr_get_options", "dr_group_set_options", "dr_debug_sah_topology_device",
    "dr_formfactors_reserve",
]

TREE_AUTO, TREE_LBVH, TREE_SAH = 0, 1, 2
WALK_AUTO, WALK_THREADED, WALK_PAIRS, WALK_PATHS = 0, 1, 2, 3
GROUP_EXCHANGE_AUTO, GROUP_EXCHANGE_P2P, GROUP_EXCHANGE_RCCL, GROUP_EXCHANGE_INPASS = 0, 1, 2, 3


class DaisyRiotError(RuntimeError):
    pass


class Info(C.Structure):
    _fields_ = [("N", C.c_int), ("S", C.c_int), ("rank", C.c_int), ("world", C.c_int), ("row0", C.c_int),
                ("nrows", C.c_int), ("rows_per_rank", C.c_int), ("n_bvh_nodes", C.c_int),
                ("ld_F", C.c_size_t), ("bytes_F", C.c_size_t), ("last_assemble_ms", C.c_double),
                ("last_bvh_ms", C.c_double), ("pairs_traced", C.c_uint64), ("sweep_launches", C.c_uint64),
                ("sweep_ms_total", C.c_double), ("blocks_nonzero", C.c_uint64), ("blocks_total", C.c_uint64),
                ("tree_used", C.c_int32), ("tree_on_host", C.c_int32), ("tree_depth", C.c_int32), ("walk_used", C.c_int32),
                ("sweep_ksplit", C.c_int32), ("reserved_", C.c_int32)]


class Options(C.Structure):
    """dr_options (include/daisyriot_hip.h): every choice between equivalent ways of doing the same thing, per context."""
    _fields_ = [("size", C.c_int32), ("tree", C.c_int32), ("sah_on_host", C.c_int32), ("morton_key", C.c_int32),
                ("sah_bins", C.c_int32), ("sah_dilate", C.c_float), ("sah_host_threads", C.c_int32),
                ("walk", C.c_int32), ("octant_test", C.c_int32), ("vis_exchange", C.c_int32), ("tile_stats", C.c_int32),
                ("debug_pair_lo", C.c_int32), ("debug_pair_hi", C.c_int32), ("debug_ray", C.c_int32),
                ("sweep_ksplit", C.c_int32), ("sweep_taper", C.c_int32), ("sweep_rows_per_wave", C.c_int32),
                ("sweep_skew", C.c_int32), ("sweep_mfma", C.c_int32), ("sweep_fenced", C.c_int32), ("no_comm", C.c_int32),
                ("debug_converge", C.c_int32), ("group_exchange", C.c_int32), ("fault_assemble_rank", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "size"}


_lib = None


def _torch_rocm_dir():
    """torch/lib of an installed torch-ROCm wheel, found WITHOUT importing torch (None: no torch, or DR_SYSTEM_ROCM=1).

    The wheel bundles its own HIP runtime and RCCL and loads them by FILE name (NEEDED libamdhip64.so / librccl.so), which the
    loader does not match with the system copies this library would otherwise pull in by SONAME: a process that loads
    libdaisyriot_hip.so first and imports torch later ends up with two HIP runtimes and two RCCLs.  So in a Python process the
    wheel's copies are loaded first -- the runtime torch itself will use, and the one every test and bench run of this package
    has run on (they import torch first) -- whatever the import order."""
    if os.environ.get("DR_SYSTEM_ROCM"):
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None


def load_library(path=None):
    """dlopen the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise DaisyRiotError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(make -C daisyriot_amd/csrc). There is no CPU fallback." % path)
    tdir = _torch_rocm_dir()
    if tdir and "torch" not in sys.modules:
        try:
            C.CDLL(os.path.join(tdir, "libamdhip64.so"), mode=C.RTLD_GLOBAL)     # SONAME libamdhip64.so.7: what our NEEDED entry asks for
        except OSError:
            tdir = None          # a wheel whose runtime does not load on its own: the system's ROCm, as a C++ host would use
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    L.dr_comm_set_library.argtypes = [C.c_char_p]
    L.dr_comm_library_info.argtypes = [C.c_char_p, C.c_size_t]
    if tdir and not os.environ.get("DR_RCCL_LIB") and os.path.exists(os.path.join(tdir, "librccl.so")):
        L.dr_comm_set_library(os.path.join(tdir, "librccl.so").encode())   # bound (dlopen) at the first dr_comm_* call only
    vp, i = C.c_void_p, C.c_int
    L.dr_last_error.restype = C.c_char_p
    L.dr_context_create.argtypes = [i, C.POINTER(vp)]
    L.dr_context_destroy.argtypes = [vp]
    L.dr_set_stream.argtypes = [vp, vp]
    L.dr_set_shard.argtypes = [vp, i, i]
    L.dr_get_shard.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.dr_scene_set_mesh.argtypes = [vp, vp, i, vp, i, vp, vp, i]
    L.dr_formfactors_assemble.argtypes = [vp, vp, i, C.c_float, i, i]
    L.dr_formfactors_integrand_only.argtypes = [vp]
    L.dr_formfactors_reserve.argtypes = [vp]
    L.dr_formfactors_read_rows.argtypes = [vp, i, i, vp]
    L.dr_visibility_read_rows.argtypes = [vp, i, i, vp]
    L.dr_formfactors_load_rows.argtypes = [vp, i, i, vp]
    L.dr_solver_init.argtypes = [vp, i, vp, vp, i, vp]
    L.dr_solver_step.argtypes = [vp, i, C.POINTER(C.c_float)]
    L.dr_solver_converge.argtypes = [vp, C.c_float, i, i, C.POINTER(i)]
    L.dr_solver_reset.argtypes = [vp]
    L.dr_solver_read.argtypes = [vp, vp, vp]
    L.dr_solver_residual_sums.argtypes = [vp, vp]
    L.dr_comm_unique_id.argtypes = [vp]
    L.dr_comm_init.argtypes = [vp, vp, i, i]
    L.dr_get_info.argtypes = [vp, C.POINTER(Info)]
    L.dr_profile_enable.argtypes = [vp, i]
    L.dr_profile_reset.argtypes = [vp]
    L.dr_synchronize.argtypes = [vp]
    L.dr_debug_read_bvh.argtypes = [vp, vp, i]
    L.dr_debug_read_array.argtypes = [vp, i, vp, C.c_size_t]
    L.dr_debug_sah_topology.argtypes = [i, vp, vp, vp, vp, vp, vp, vp]
    L.dr_debug_sah_topology_device.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp]
    L.dr_comm_manual.argtypes = [vp]
    L.dr_exchange_export.argtypes = [vp, vp]
    L.dr_exchange_import.argtypes = [vp, i, vp, C.c_size_t]
    L.dr_shard_rows.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.dr_residual_offset.argtypes = [i, i, i, i]
    L.dr_formfactors_assemble_split.argtypes = [vp, vp, i, C.c_float, i, i]
    L.dr_vis_exchange_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.dr_vis_exchange_export.argtypes = [vp, i, vp, C.c_size_t]
    L.dr_vis_exchange_import.argtypes = [vp, i, vp, C.c_size_t]
    L.dr_vis_exchange_fits.argtypes = [i, i, i, C.c_size_t]
    L.dr_residual_chunk_floats.argtypes = [i, i]
    L.dr_solver_set_check_interval.argtypes = [vp, i]
    L.dr_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.dr_group_create.argtypes = [vp, i, C.POINTER(vp)]
    L.dr_group_destroy.argtypes = [vp]
    L.dr_group_info.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.dr_group_context.argtypes = [vp, i, C.POINTER(vp)]
    L.dr_group_set_mesh.argtypes = [vp, vp, i, vp, i, vp, vp, i]
    L.dr_group_assemble.argtypes = [vp, vp, i, C.c_float, i, i]
    L.dr_group_solver_init.argtypes = [vp, i, vp, vp, i, vp]
    L.dr_group_solver_step.argtypes = [vp, i, C.POINTER(C.c_float)]
    L.dr_group_solver_converge.argtypes = [vp, C.c_float, i, i, C.POINTER(i)]
    L.dr_group_solver_reset.argtypes = [vp]
    L.dr_group_solver_read.argtypes = [vp, vp, vp]
    L.dr_group_synchronize.argtypes = [vp]
    L.dr_formfactors_assemble_finish.argtypes = [vp]
    L.dr_vis_exchange_tracer.argtypes = [i, i, i, i]
    L.dr_solver_skip_zero_blocks.argtypes = [vp, i]
    L.dr_display_patch_colors.argtypes = [vp, i, vp, vp]
    L.dr_display_vertex_colors.argtypes = [vp, vp, vp, vp, i, vp]
    L.dr_options_defaults.argtypes = [C.POINTER(Options)]
    L.dr_set_options.argtypes = [vp, C.POINTER(Options)]
    L.dr_get_options.argtypes = [vp, C.POINTER(Options)]
    L.dr_group_set_options.argtypes = [vp, C.POINTER(Options)]
    for name in EXPORTS:
        if name not in ("dr_last_error", "dr_residual_offset", "dr_residual_chunk_floats"):
            getattr(L, name).restype = i
    L.dr_residual_offset.restype = C.c_size_t
    L.dr_residual_chunk_floats.restype = C.c_size_t
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def shard_rows(N, rank, world):
    """(row0, nrows, rows_per_rank) of rank `rank` -- pure host arithmetic of the library."""
    L = load_library()
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    if L.dr_shard_rows(int(N), int(rank), int(world), C.byref(a), C.byref(b), C.byref(c)):
        raise DaisyRiotError("dr_shard_rows: %s" % L.dr_last_error().decode())
    return a.value, b.value, c.value


def vis_exchange_tracer(N, world, patch_a, patch_b):
    """rank that traces the pair of patches (a, b) in a world-way assembly -- pure host arithmetic of the library"""
    return int(load_library().dr_vis_exchange_tracer(int(N), int(world), int(patch_a), int(patch_b)))


def residual_offset(i, s, S, rows_per_rank):
    return int(load_library().dr_residual_offset(int(i), int(s), int(S), int(rows_per_rank)))


def residual_chunk_floats(S, rows_per_rank):
    """floats of one rank's chunk of a gathered residual buffer: S*rows_per_rank values + the chunk's per-bin sums"""
    return int(load_library().dr_residual_chunk_floats(int(S), int(rows_per_rank)))


def vis_exchange_fits(N, world, keep_visibility, device_bytes):
    """does a world-way assembly take the ray-count exchange path on cards of device_bytes?  (no rank argument: every
    rank must decide alike) -- pure host arithmetic of the library"""
    return bool(load_library().dr_vis_exchange_fits(int(N), int(world), int(bool(keep_visibility)), int(device_bytes)))


def options_defaults():
    """the built-in defaults overridden by the DR_* environment variables (what a new context starts with)"""
    L = load_library()
    o = Options()
    if L.dr_options_defaults(C.byref(o)):
        raise DaisyRiotError("dr_options_defaults: %s" % L.dr_last_error().decode())
    return o


def comm_library_info():
    """(file the RCCL symbols are bound from, [every RCCL file mapped into the process])"""
    L = load_library()
    buf = C.create_string_buffer(4096)
    if L.dr_comm_library_info(buf, len(buf)):
        raise DaisyRiotError("dr_comm_library_info: %s" % L.dr_last_error().decode())
    bound, mapped = buf.value.decode().split(";mapped=")
    return bound[len("bound="):], [m for m in mapped.split(",") if m]


def comm_unique_id():
    L = load_library()
    buf = np.zeros(128, np.uint8)
    rc = L.dr_comm_unique_id(_p(buf))
    if rc:
        raise DaisyRiotError("dr_comm_unique_id: %s" % L.dr_last_error().decode())
    return buf


def sah_topology(boxes):
    """tests: the tree topology the library builds on the host for N boxes [N][6] = lo xyz, hi xyz (no device needed).
    Returns dict(order, left, right, first, last, parent) as in include/daisyriot_hip.h."""
    L = load_library()
    boxes = np.ascontiguousarray(boxes, np.float32)
    N = boxes.shape[0]
    out = {k: np.zeros(max(N - 1, 1), np.int32) for k in ("left", "right", "first", "last")}
    out["order"] = np.zeros(N, np.int32)
    out["parent"] = np.zeros(2 * N - 1, np.int32)
    rc = L.dr_debug_sah_topology(N, _p(boxes), _p(out["order"]), _p(out["left"]), _p(out["right"]), _p(out["first"]), _p(out["last"]),
                                 _p(out["parent"]))
    if rc:
        raise DaisyRiotError("dr_debug_sah_topology: %s" % L.dr_last_error().decode())
    return out


class Context:
    """One GPU's worth of the hot path.  Mirrors the call order of the reference:
    MeshS -> OptixPrimeFunctionality(mesh) -> Lightning (initMat, reset, passes)."""

    def __init__(self, device=0, _borrowed=None):
        self.L = load_library()
        self.N = 0
        self.S = 0
        self._owned = _borrowed is None
        if _borrowed is not None:        # a context that belongs to a Group
            self.h = _borrowed
            return
        h = C.c_void_p()
        self._chk(self.L.dr_context_create(int(device), C.byref(h)), "dr_context_create")
        self.h = h

    def _chk(self, rc, what):
        if rc != 0:
            raise DaisyRiotError("%s failed (%d): %s" % (what, rc, self.L.dr_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                self.L.dr_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- configuration
    def options(self):
        o = Options()
        self._chk(self.L.dr_get_options(self.h, C.byref(o)), "dr_get_options")
        return o

    def set_options(self, **kw):
        """change some of the context's options (names as in dr_options); returns the full set now in force"""
        o = self.options()
        for k, v in kw.items():
            if k not in dict(Options._fields_) or k == "size":
                raise DaisyRiotError("no option %r" % k)
            setattr(o, k, v)
        self._chk(self.L.dr_set_options(self.h, C.byref(o)), "dr_set_options")
        return o

    def set_stream(self, hip_stream_ptr):
        self._chk(self.L.dr_set_stream(self.h, C.c_void_p(hip_stream_ptr)), "dr_set_stream")

    def set_shard(self, rank, world):
        self._chk(self.L.dr_set_shard(self.h, int(rank), int(world)), "dr_set_shard")

    def shard(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.L.dr_get_shard(self.h, C.byref(a), C.byref(b), C.byref(c)), "dr_get_shard")
        return a.value, b.value, c.value

    def comm_init(self, id128, rank, world):
        id128 = np.ascontiguousarray(id128, dtype=np.uint8)
        assert id128.size == 128
        self._chk(self.L.dr_comm_init(self.h, _p(id128), int(rank), int(world)), "dr_comm_init")

    def comm_info(self):
        """(rank, world) the context's RCCL communicator itself reports; (-1, 0) without one"""
        a, b = C.c_int(), C.c_int()
        self._chk(self.L.dr_comm_info(self.h, C.byref(a), C.byref(b)), "dr_comm_info")
        return a.value, b.value

    def comm_manual(self):
        self._chk(self.L.dr_comm_manual(self.h), "dr_comm_manual")

    def exchange_export(self):
        _, _, rpr = self.shard()
        out = np.empty(residual_chunk_floats(self.S, rpr), np.float32)
        self._chk(self.L.dr_exchange_export(self.h, _p(out)), "dr_exchange_export")
        return out

    def exchange_import(self, src_rank, chunk):
        chunk = _f32(chunk)
        self._chk(self.L.dr_exchange_import(self.h, int(src_rank), _p(chunk), chunk.size), "dr_exchange_import")

    # -- scene
    def set_mesh(self, vertices, normals, tri_v, tri_n):
        v, n = _f32(vertices).reshape(-1, 3), _f32(normals).reshape(-1, 3)
        tv, tn = _i32(tri_v).reshape(-1, 3), _i32(tri_n).reshape(-1, 3)
        if tv.shape != tn.shape:
            raise DaisyRiotError("tri_v and tri_n differ in shape")
        self._chk(self.L.dr_scene_set_mesh(self.h, _p(v), v.shape[0], _p(n), n.shape[0], _p(tv), _p(tn), tv.shape[0]),
                  "dr_scene_set_mesh")
        self.N = tv.shape[0]

    # -- form factors
    def assemble(self, uv, eps=ORIGIN_EPS, rule=RULE_INTEGRAND, keep_visibility=False):
        uv = _f32(uv).reshape(-1, 2)
        self._chk(self.L.dr_formfactors_assemble(self.h, _p(uv), uv.shape[0], C.c_float(eps), int(rule),
                                                 int(bool(keep_visibility))), "dr_formfactors_assemble")

    # multi-rank assembly in steps, for a host that moves the ray-count slots itself
    def assemble_split(self, uv, eps=ORIGIN_EPS, rule=RULE_INTEGRAND, keep_visibility=False):
        uv = _f32(uv).reshape(-1, 2)
        self._chk(self.L.dr_formfactors_assemble_split(self.h, _p(uv), uv.shape[0], C.c_float(eps), int(rule),
                                                       int(bool(keep_visibility))), "dr_formfactors_assemble_split")

    def vis_exchange_export(self, dst_rank):
        """the block of ray-count slots this rank traced for dst_rank"""
        n = C.c_size_t()
        self._chk(self.L.dr_vis_exchange_bytes(self.h, C.byref(n)), "dr_vis_exchange_bytes")
        out = np.empty(n.value, np.uint8)
        self._chk(self.L.dr_vis_exchange_export(self.h, int(dst_rank), _p(out), out.size), "dr_vis_exchange_export")
        return out

    def vis_exchange_import(self, src_rank, block):
        block = np.ascontiguousarray(block, dtype=np.uint8)
        self._chk(self.L.dr_vis_exchange_import(self.h, int(src_rank), _p(block), block.size), "dr_vis_exchange_import")

    def assemble_finish(self):
        self._chk(self.L.dr_formfactors_assemble_finish(self.h), "dr_formfactors_assemble_finish")

    def reserve(self):
        """allocate this rank's shard of F now (else: on first use)"""
        self._chk(self.L.dr_formfactors_reserve(self.h), "dr_formfactors_reserve")

    def integrand_only(self):
        self._chk(self.L.dr_formfactors_integrand_only(self.h), "dr_formfactors_integrand_only")

    def read_rows(self, row0, nrows):
        out = np.empty((nrows, self.N), np.float32)
        self._chk(self.L.dr_formfactors_read_rows(self.h, int(row0), int(nrows), _p(out)), "dr_formfactors_read_rows")
        return out

    def read_visibility(self, row0, nrows):
        out = np.empty((nrows, self.N), np.uint8)
        self._chk(self.L.dr_visibility_read_rows(self.h, int(row0), int(nrows), _p(out)), "dr_visibility_read_rows")
        return out

    def load_rows(self, row0, F):
        F = _f32(F)
        assert F.ndim == 2 and F.shape[1] == self.N
        self._chk(self.L.dr_formfactors_load_rows(self.h, int(row0), F.shape[0], _p(F)), "dr_formfactors_load_rows")

    # -- solver
    def solver_init(self, E, M, mat_of_patch):
        E, M, mat = _f32(E), _f32(M), _i32(mat_of_patch)
        if E.ndim == 1:
            E = E.reshape(-1, 1)
        S = E.shape[1]
        if M.ndim == 1:
            M = M.reshape(-1, 1, 1)
        if E.shape[0] != self.N or mat.shape[0] != self.N or M.shape[1:] != (S, S):
            raise DaisyRiotError("solver input shapes do not match N=%d S=%d" % (self.N, S))
        self._chk(self.L.dr_solver_init(self.h, S, _p(E), _p(M), M.shape[0], _p(mat)), "dr_solver_init")
        self.S = S

    def step(self, n_passes=1, want_sum=False):
        if want_sum:
            out = C.c_float()
            self._chk(self.L.dr_solver_step(self.h, int(n_passes), C.byref(out)), "dr_solver_step")
            return out.value
        self._chk(self.L.dr_solver_step(self.h, int(n_passes), None), "dr_solver_step")
        return None

    def converge(self, threshold, per_bin=False, max_iters=10000):
        it = C.c_int()
        self._chk(self.L.dr_solver_converge(self.h, C.c_float(threshold), int(bool(per_bin)), int(max_iters),
                                            C.byref(it)), "dr_solver_converge")
        return it.value

    def set_check_interval(self, passes):
        """converge() looks at the device's convergence flag once per this many queued passes (default 8)"""
        self._chk(self.L.dr_solver_set_check_interval(self.h, int(passes)), "dr_solver_set_check_interval")

    def reset(self):
        self._chk(self.L.dr_solver_reset(self.h), "dr_solver_reset")

    def read(self, B=True, R=True):
        b = np.zeros((self.N, self.S), np.float32) if B else None
        r = np.zeros((self.N, self.S), np.float32) if R else None
        self._chk(self.L.dr_solver_read(self.h, _p(b) if B else None, _p(r) if R else None), "dr_solver_read")
        return b, r

    def skip_zero_blocks(self, enable=True):
        """light passes skip the 32 x 256 blocks of F that are entirely zero (off by default; results are bit-identical)"""
        self._chk(self.L.dr_solver_skip_zero_blocks(self.h, int(bool(enable))), "dr_solver_skip_zero_blocks")

    def residual_sums(self):
        s = np.zeros(MAX_BINS, np.float64)
        self._chk(self.L.dr_solver_residual_sums(self.h, _p(s)), "dr_solver_residual_sums")
        return s[:self.S].copy()

    # -- display colours
    def patch_colors(self, mode, xyz_per_bin=None):
        """Display colours of this rank's rows, (nrows, 3): DISPLAY_BW / DISPLAY_RGB / DISPLAY_SPECTRAL."""
        nrows = self.info().nrows
        out = np.zeros((nrows, 3), np.float32)
        xyz = _f32(xyz_per_bin) if xyz_per_bin is not None else None
        if xyz is not None and xyz.shape != (self.S, 3):
            raise DaisyRiotError("xyz_per_bin must be (S, 3)")
        self._chk(self.L.dr_display_patch_colors(self.h, int(mode), _p(xyz) if xyz is not None else None, _p(out)),
                  "dr_display_patch_colors")
        return out

    def vertex_colors(self, vtx_off, vtx_tri, rgb_all=None):
        """Mean colour of the patches around every vertex (CSR adjacency), (V, 3)."""
        off, adj = _i32(vtx_off), _i32(vtx_tri)
        V = off.shape[0] - 1
        out = np.zeros((V, 3), np.float32)
        rgb = _f32(rgb_all) if rgb_all is not None else None
        if rgb is not None and rgb.shape != (self.N, 3):
            raise DaisyRiotError("rgb_all must be (N, 3)")
        if adj.size == 0:
            adj = np.zeros(1, np.int32)
        self._chk(self.L.dr_display_vertex_colors(self.h, _p(rgb) if rgb is not None else None, _p(off), _p(adj), V, _p(out)),
                  "dr_display_vertex_colors")
        return out

    # -- measurement
    def info(self):
        o = Info()
        self._chk(self.L.dr_get_info(self.h, C.byref(o)), "dr_get_info")
        return o

    def profile(self, on=True):
        self._chk(self.L.dr_profile_enable(self.h, int(bool(on))), "dr_profile_enable")

    def profile_reset(self):
        self._chk(self.L.dr_profile_reset(self.h), "dr_profile_reset")

    def read_bvh(self):
        n = self.info().n_bvh_nodes
        dt = np.dtype([("lo", np.float32, 3), ("hi", np.float32, 3), ("skip", np.int32), ("tri", np.int32)])
        out = np.zeros(n, dt)
        self._chk(self.L.dr_debug_read_bvh(self.h, _p(out), n), "dr_debug_read_bvh")
        return out

    def sah_topology(self, boxes):
        """tests: the tree topology the DEVICE builder makes of N boxes [N][6] (same dict as api.sah_topology, the host's)"""
        boxes = np.ascontiguousarray(boxes, np.float32)
        N = boxes.shape[0]
        out = {k: np.zeros(max(N - 1, 1), np.int32) for k in ("left", "right", "first", "last")}
        out["order"] = np.zeros(N, np.int32)
        out["parent"] = np.zeros(2 * N - 1, np.int32)
        self._chk(self.L.dr_debug_sah_topology_device(self.h, N, _p(boxes), _p(out["order"]), _p(out["left"]), _p(out["right"]), _p(out["first"]),
                                                      _p(out["last"]), _p(out["parent"])), "dr_debug_sah_topology_device")
        return out

    def read_array(self, which, nbytes):
        out = np.zeros(nbytes, np.uint8)
        self._chk(self.L.dr_debug_read_array(self.h, int(which), _p(out), nbytes), "dr_debug_read_array")
        return out

    def synchronize(self):
        self._chk(self.L.dr_synchronize(self.h), "dr_synchronize")


class Group:
    """One process, several GPUs (dr_group): rank r on devices[r], rows of F sharded over them, every call runs on all
    devices at once.  The same device several times rehearses the group on one GPU (peer copies instead of RCCL)."""

    def __init__(self, devices):
        self.L = load_library()
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        self._chk(self.L.dr_group_create(_p(devs), devs.size, C.byref(h)), "dr_group_create")
        self.h = h
        self.n = int(devs.size)
        self.N = 0
        self.S = 0
        self.ranks = []
        for r in range(self.n):
            ch = C.c_void_p()
            self._chk(self.L.dr_group_context(self.h, r, C.byref(ch)), "dr_group_context")
            self.ranks.append(Context(_borrowed=ch))

    def _chk(self, rc, what):
        if rc != 0:
            raise DaisyRiotError("%s failed (%d): %s" % (what, rc, self.L.dr_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            for c in self.ranks:
                c.close()
            self.L.dr_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def uses_rccl(self):
        n, u = C.c_int(), C.c_int()
        self._chk(self.L.dr_group_info(self.h, C.byref(n), C.byref(u)), "dr_group_info")
        return bool(u.value)

    def set_options(self, **kw):
        """the same options on every rank (group_exchange: how the residual travels after a pass)"""
        o = self.ranks[0].options()
        for k, v in kw.items():
            if k not in dict(Options._fields_) or k == "size":
                raise DaisyRiotError("no option %r" % k)
            setattr(o, k, v)
        self._chk(self.L.dr_group_set_options(self.h, C.byref(o)), "dr_group_set_options")
        return o

    def set_mesh(self, vertices, normals, tri_v, tri_n):
        v, n = _f32(vertices).reshape(-1, 3), _f32(normals).reshape(-1, 3)
        tv, tn = _i32(tri_v).reshape(-1, 3), _i32(tri_n).reshape(-1, 3)
        self._chk(self.L.dr_group_set_mesh(self.h, _p(v), v.shape[0], _p(n), n.shape[0], _p(tv), _p(tn), tv.shape[0]),
                  "dr_group_set_mesh")
        self.N = tv.shape[0]
        for c in self.ranks:
            c.N = self.N

    def assemble(self, uv, eps=ORIGIN_EPS, rule=RULE_INTEGRAND, keep_visibility=False):
        uv = _f32(uv).reshape(-1, 2)
        self._chk(self.L.dr_group_assemble(self.h, _p(uv), uv.shape[0], C.c_float(eps), int(rule), int(bool(keep_visibility))),
                  "dr_group_assemble")

    def solver_init(self, E, M, mat_of_patch):
        E, M, mat = _f32(E), _f32(M), _i32(mat_of_patch)
        if E.ndim == 1:
            E = E.reshape(-1, 1)
        S = E.shape[1]
        if M.ndim == 1:
            M = M.reshape(-1, 1, 1)
        if E.shape[0] != self.N or mat.shape[0] != self.N or M.shape[1:] != (S, S):
            raise DaisyRiotError("solver input shapes do not match N=%d S=%d" % (self.N, S))
        self._chk(self.L.dr_group_solver_init(self.h, S, _p(E), _p(M), M.shape[0], _p(mat)), "dr_group_solver_init")
        self.S = S
        for c in self.ranks:
            c.S = S

    def step(self, n_passes=1, want_sum=False):
        out = C.c_float()
        self._chk(self.L.dr_group_solver_step(self.h, int(n_passes), C.byref(out) if want_sum else None), "dr_group_solver_step")
        return out.value if want_sum else None

    def converge(self, threshold, per_bin=False, max_iters=10000):
        it = C.c_int()
        self._chk(self.L.dr_group_solver_converge(self.h, C.c_float(threshold), int(bool(per_bin)), int(max_iters), C.byref(it)),
                  "dr_group_solver_converge")
        return it.value

    def reset(self):
        self._chk(self.L.dr_group_solver_reset(self.h), "dr_group_solver_reset")

    def read(self):
        b = np.zeros((self.N, self.S), np.float32)
        r = np.zeros((self.N, self.S), np.float32)
        self._chk(self.L.dr_group_solver_read(self.h, _p(b), _p(r)), "dr_group_solver_read")
        return b, r

    def synchronize(self):
        self._chk(self.L.dr_group_synchronize(self.h), "dr_group_synchronize")
