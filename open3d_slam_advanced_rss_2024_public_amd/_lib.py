"""ctypes binding of libo3dslam_icp_hip.so (C ABI: include/o3s_icp.h).

The library is built in-tree (``make -C open3d_slam_advanced_rss_2024_public_amd/csrc``) and loaded from the package
directory.  There is no fallback of any kind: a missing library raises ImportError-like RuntimeError here, and a
machine without a gfx950 device makes ``o3s_icp_create`` fail with O3S_ERR_HIP.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def variant_path(variant: str | None) -> str:
    return os.path.join(_HERE, f"libo3dslam_icp_hip_{variant}.so" if variant else "libo3dslam_icp_hip.so")


# The product library is libo3dslam_icp_hip.so: no environment variable reaches it (no getenv in the binary).  Builds with
# extra -D flags live beside it as libo3dslam_icp_hip_<name>.so (`make -C csrc hooks|ts|variant`):
#   hooks  -DO3S_TEST_HOOKS: the test hooks and tuning knobs (O3S_SCATTER_ORDER, O3S_FUSE, O3S_SEL_PARTIAL, O3S_NO_HINT, O3S_DBG, ...)
#          read from the environment; tests that need them load it through `with _lib.variant("hooks"):`
#   ts     in-kernel phase stamps (tools/ts.py)
# O3S_LIB_VARIANT=<name> makes <name> the default library of the process (A/B runs of tools/, the whole suite on the hooks build).
_variant = os.environ.get("O3S_LIB_VARIANT") or None
LIB_PATH = variant_path(_variant)

# o3s_status
OK = 0
ERR_EMPTY_REFERENCE = 1
ERR_EMPTY_READING = 2
ERR_BAD_SHAPE = 3
ERR_NOT_INITIALIZED = 4
ERR_NO_MATCHES = 5
ERR_NO_POINTS = 6
ERR_NAN = 7
ERR_NOT_RIGID = 8
ERR_BAD_CONFIG = 9
ERR_HIP = 10
ERR_BAD_ARGUMENT = 11


class IcpConfigC(C.Structure):
    _fields_ = [
        ("matcher", C.c_int32),
        ("max_dist", C.c_float),
        ("epsilon", C.c_float),
        ("trim_ratio", C.c_float),
        ("max_normal_angle", C.c_float),
        ("max_dist_outlier", C.c_float),
        ("use_differential", C.c_int32),
        ("min_diff_rot", C.c_float),
        ("min_diff_trans", C.c_float),
        ("smooth_length", C.c_int32),
        ("max_iters", C.c_int32),
        ("counter_first", C.c_int32),
        ("grid_cell", C.c_float),
        ("sort_queries", C.c_int32),
        ("use_graph", C.c_int32),
        ("match_stats", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


class IcpStatsC(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("max_iters_reached", C.c_int32),
        ("kept_pairs", C.c_int64),
        ("matched_pairs", C.c_int64),
        ("point_used_ratio", C.c_float),
        ("weighted_point_used_ratio", C.c_float),
        ("last_trim_limit", C.c_float),
        ("gpu_ms", C.c_float),
        ("candidates_examined", C.c_double),
        ("cells_probed", C.c_double),
    ]


# o3s_allreduce_fn(user, dev_ptr, byte_offset, count, dtype, hip_stream) -> 0 on success
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p)
XCHG_INT32, XCHG_FLOAT64 = 0, 1


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of the shared library (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "o3s_icp.h"))
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None      # the library lib() returns: the process default, or the one a `with variant(...)` block selected
_loaded = {}     # variant name ("" = product) -> CDLL
_rccl = None
RCCL_LIB_PATH = os.path.join(_HERE, "libo3dslam_icp_rccl.so")


def needs_binding(L, module: str) -> bool:
    """True the first time `module` asks about library `L`: wrapper modules declare their argtypes once per loaded library."""
    done = L.__dict__.setdefault("_o3s_bound", set())
    if module in done:
        return False
    done.add(module)
    return True


def _preload_torch_runtime(names) -> None:
    """One ROCm runtime per process.  PyTorch wheels carry their own libamdhip64 / libhsa-runtime64 / librccl (same sonames as
    the system's /opt/rocm copies, another ROCm release).  This library's NEEDED entries are the sonames, torch's are the bare
    file names: when torch is imported FIRST the loader hands this library torch's copies (one runtime, fine); when this
    library is loaded first it gets /opt/rocm's and a later `import torch` maps torch's copies BESIDE them — two HIP and two
    HSA runtimes driving the same GPU, which is what made a torch import after a long run of this library fail to
    initialise (round 2).  So: if torch is installed but not imported yet, its runtime libraries are mapped first, by path;
    whichever of the two comes next then binds to them.  O3S_SYSTEM_ROCM=1 keeps the system runtime (processes that never
    import torch); C++ hosts link one runtime anyway."""
    if "torch" in sys.modules or os.environ.get("O3S_SYSTEM_ROCM") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    mapped = [os.path.basename(p) for p in loaded_rocm_runtimes()]
    for n in names:
        path = os.path.join(libdir, n)
        # a runtime of this kind is mapped already (rocprofv3's preload, another extension, an earlier call): mapping torch's
        # beside it would CREATE the two-runtime process this function is here to prevent
        if any(m.startswith(n.split(".so")[0]) for m in mapped):
            continue
        if os.path.exists(path):
            C.CDLL(path)  # RTLD_LOCAL: the soname is what later NEEDED entries match; global symbols of librccl clash with torch at exit


def loaded_rocm_runtimes():
    """Paths of the HIP / HSA / RCCL runtime libraries mapped into this process (diagnostics, tests): one of each at most."""
    seen = set()
    with open("/proc/self/maps") as f:
        for ln in f:
            path = ln.split()[-1]
            if any(k in os.path.basename(path) for k in ("libamdhip64", "libhsa-runtime64", "librccl")):
                seen.add(os.path.realpath(path))
    return sorted(seen)


def rccl_lib() -> C.CDLL:
    """libo3dslam_icp_rccl.so (include/o3s_rccl.h): ncclAllReduce-backed exchange of the one-pair-sharded mode."""
    global _rccl
    if _rccl is not None:
        return _rccl
    if not os.path.exists(RCCL_LIB_PATH):
        raise RuntimeError(f"{RCCL_LIB_PATH} is missing: build it with `make -C {CSRC}`")
    _preload_torch_runtime(["libamdhip64.so", "librccl.so"])
    R = C.CDLL(RCCL_LIB_PATH)
    R.o3s_rccl_unique_id.argtypes = [C.c_char_p]
    R.o3s_rccl_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int, C.POINTER(C.c_void_p)]
    R.o3s_rccl_destroy.argtypes = [C.c_void_p]
    R.o3s_rccl_destroy.restype = None
    R.o3s_rccl_collectives.argtypes = [C.c_void_p]
    R.o3s_rccl_collectives.restype = C.c_int64
    R.o3s_rccl_last_error.restype = C.c_char_p
    _rccl = R
    return R


def load(variant: str | None = None) -> C.CDLL:
    """Loads (once) and binds libo3dslam_icp_hip[_<variant>].so.  Never builds implicitly on a GPU box: the .so travels with the tree."""
    key = variant or ""
    if key in _loaded:
        return _loaded[key]
    path = variant_path(variant)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `make -C {CSRC}{' ' + variant if variant else ''}` (hipcc, gfx950). "
            "There is no CPU or PyTorch fallback for the ICP path.")
    _preload_torch_runtime(["libamdhip64.so"])
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.o3s_abi_version.restype = C.c_int
    L.o3s_icp_default_config.argtypes = [C.POINTER(IcpConfigC)]
    L.o3s_icp_default_config.restype = None
    L.o3s_icp_create.argtypes = [C.POINTER(IcpConfigC), C.c_int, C.POINTER(vp)]
    L.o3s_icp_destroy.argtypes = [vp]
    L.o3s_icp_destroy.restype = None
    L.o3s_last_error.argtypes = [vp]
    L.o3s_last_error.restype = C.c_char_p
    L.o3s_icp_set_stream.argtypes = [vp, vp]
    L.o3s_icp_synchronize.argtypes = [vp]
    L.o3s_icp_init_reference.argtypes = [vp, fp, fp, C.c_int64]
    L.o3s_icp_init_reference_dev.argtypes = [vp, vp, vp, C.c_int64]
    L.o3s_icp_compute.argtypes = [vp, fp, fp, C.c_int64, fp, fp, C.POINTER(IcpStatsC)]
    L.o3s_icp_set_reading.argtypes = [vp, fp, fp, C.c_int64]
    L.o3s_icp_set_reading_dev.argtypes = [vp, vp, vp, C.c_int64]
    L.o3s_icp_compute_resident.argtypes = [vp, fp, fp, C.POINTER(IcpStatsC)]
    L.o3s_icp_compute_resident_launch.argtypes = [vp, fp]
    L.o3s_icp_compute_resident_finish.argtypes = [vp, fp, C.POINTER(IcpStatsC)]
    L.o3s_icp_compute_batch.argtypes = [C.POINTER(vp), C.c_int32, fp, fp, C.POINTER(IcpStatsC), ip]
    L.o3s_icp_get_trace.argtypes = [vp, fp, fp, C.POINTER(C.c_int64), C.c_int32]
    L.o3s_icp_reference_mean.argtypes = [vp, fp]
    L.o3s_icp_get_reading_order.argtypes = [vp, ip, C.c_int64]
    L.o3s_icp_get_reading_order.restype = C.c_int64
    L.o3s_icp_set_profiling.argtypes = [vp, C.c_int]
    L.o3s_icp_kernel_ms.argtypes = [vp, fp, ip]
    L.o3s_icp_profile_match.argtypes = [vp, fp, C.c_int32, C.c_int32, fp]
    L.o3s_icp_find_closests.argtypes = [vp, fp, C.c_int64, ip, fp]
    L.o3s_icp_outlier_weights.argtypes = [vp, fp, ip, fp, C.c_int64, fp]
    L.o3s_icp_minimize.argtypes = [vp, fp, ip, fp, fp, C.c_int64, fp, fp, fp, fp]
    L.o3s_icp_shard_configure.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, ALLREDUCE_FN, vp, vp]
    L.o3s_icp_shard_exchange_bytes.restype = C.c_int64
    L.o3s_icp_shard_set_capturable.argtypes = [vp, C.c_int]
    L.o3s_stream_copy_gbs.argtypes = [C.c_int, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
    L.o3s_matcher_init.argtypes = [vp, fp, fp, C.c_int64]
    L.o3s_icp_host_split.argtypes = [vp, C.POINTER(C.c_double)]
    L.o3s_icp_host_split_ex.argtypes = [vp, C.POINTER(C.c_double)]
    _loaded[key] = L
    return L


def lib() -> C.CDLL:
    """The library in use: the product build unless O3S_LIB_VARIANT or a `with variant(...)` block says otherwise."""
    global _lib
    if _lib is None:
        _lib = load(_variant)
    return _lib


class variant:
    """``with _lib.variant("hooks"):`` — lib() returns that build inside the block.  Handles created inside belong to it: the
    wrappers keep the library they were created with, and the block collects garbage before it hands lib() back."""

    def __init__(self, name: str | None):
        self.name = name

    def __enter__(self):
        global _lib
        self._prev = lib()
        _lib = load(self.name)
        return _lib

    def __exit__(self, *exc):
        global _lib
        import gc

        gc.collect()
        _lib = self._prev
        return False


def forked_copy(obj) -> bool:
    """True when `obj` (a handle wrapper that stored os.getpid() as _pid at creation) lives in a forked child of the process
    that created it: the device handle belongs to the parent's HIP runtime, which does not survive a fork — the child must
    drop its copy of the wrapper without calling into the library (a garbage collection in a multiprocessing worker would
    otherwise abort the worker)."""
    import os
    return getattr(obj, "_pid", None) not in (None, os.getpid())
