"""Host-side mirror of ``PointMatcher<float>::ICP`` over the C ABI (include/o3s_icp.h).

The class keeps the reference's operator surface for the scan-to-map path — ``initReference`` / ``compute`` /
``operator()`` (libpointmatcher/pointmatcher/PointMatcher.h:807-848, ICP.cpp:258-468), the YAML chain loader
(ICP.cpp:113-160) for the modules that are on the path, and the exception taxonomy (ConvergenceError,
TransformationError, runtime_error) — so tests written against the reference read the same here.  All arithmetic
happens in libo3dslam_icp_hip.so on the GPU; this file only marshals numpy arrays and maps status codes.
"""
from __future__ import annotations

import os
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib


class ConvergenceError(RuntimeError):
    """PointMatcher<T>::ConvergenceError (libpointmatcher/pointmatcher/PointMatcher.h:142-147)."""


class TransformationError(RuntimeError):
    """PointMatcherSupport::TransformationError (libpointmatcher/pointmatcher/PointMatcher.h:92-96)."""


class InvalidModuleType(RuntimeError):
    """PointMatcherSupport::InvalidModuleType: a YAML chain names a module that is not on the accelerated path."""


class HipError(RuntimeError):
    pass


_RAISE = {
    _lib.ERR_EMPTY_READING: RuntimeError,
    _lib.ERR_BAD_SHAPE: RuntimeError,
    _lib.ERR_NOT_INITIALIZED: RuntimeError,
    _lib.ERR_NO_MATCHES: ConvergenceError,
    _lib.ERR_NO_POINTS: ConvergenceError,
    _lib.ERR_NAN: ConvergenceError,
    _lib.ERR_NOT_RIGID: TransformationError,
    _lib.ERR_BAD_CONFIG: ValueError,
    _lib.ERR_BAD_ARGUMENT: ValueError,
    _lib.ERR_HIP: HipError,
}


@dataclass
class IcpConfig:
    """The ICP chain of open3d_slam_ros/param/icp.yaml (defaults) — one field per YAML parameter on the path."""
    matcher: str = "KDTreeMatcher"     # or "MirrorMatcher"
    knn: int = 1
    max_dist: float = 0.5
    epsilon: float = 0.01               # accepted, ignored: the GPU matcher is exact
    trim_ratio: float = 0.9             # TrimmedDistOutlierFilter.ratio; None => filter absent
    max_normal_angle: float = 1.57      # SurfaceNormalOutlierFilter.maxAngle; None => absent
    max_dist_outlier: float = None      # MaxDistOutlierFilter.maxDist; None => absent
    use_differential: bool = True
    min_diff_rot: float = 0.001
    min_diff_trans: float = 0.01
    smooth_length: int = 3
    max_iters: int = 15                 # CounterTransformationChecker; None/0 => absent
    counter_first: bool = False
    grid_cell: float = 0.0
    sort_queries: bool = True
    use_graph: bool = True
    match_stats: bool = False

    def to_c(self) -> _lib.IcpConfigC:
        if self.knn != 1:
            raise InvalidModuleType("KDTreeMatcher.knn != 1 is not on the accelerated path")
        c = _lib.IcpConfigC()
        c.matcher = {"KDTreeMatcher": 0, "MirrorMatcher": 1}[self.matcher]
        c.max_dist = float(self.max_dist)
        c.epsilon = float(self.epsilon)
        c.trim_ratio = -1.0 if self.trim_ratio is None else float(self.trim_ratio)
        c.max_normal_angle = -1.0 if self.max_normal_angle is None else float(self.max_normal_angle)
        c.max_dist_outlier = -1.0 if self.max_dist_outlier is None else float(self.max_dist_outlier)
        c.use_differential = int(self.use_differential)
        c.min_diff_rot = float(self.min_diff_rot)
        c.min_diff_trans = float(self.min_diff_trans)
        c.smooth_length = int(self.smooth_length)
        c.max_iters = int(self.max_iters or 0)
        c.counter_first = int(self.counter_first)
        c.grid_cell = float(self.grid_cell)
        c.sort_queries = int(self.sort_queries)
        c.use_graph = int(self.use_graph)
        c.match_stats = int(self.match_stats)
        return c

    @staticmethod
    def from_yaml(text: str) -> "IcpConfig":
        """ICPChainBase::loadFromYaml (libpointmatcher/pointmatcher/ICP.cpp:113-160) for the modules on the path.
        Unknown module names raise InvalidModuleType, as the registrar does (Registrar.h:162-176)."""
        import yaml

        doc = yaml.safe_load(text) or {}
        cfg = IcpConfig(trim_ratio=None, max_normal_angle=None, max_dist_outlier=None, use_differential=False, max_iters=None)

        def modules(node):
            if node is None:
                return []
            if isinstance(node, str):
                return [(node, {})]
            if isinstance(node, dict):
                return [(k, v or {}) for k, v in node.items()]
            out = []
            for item in node:
                out.extend(modules(item))
            return out

        for key in ("readingDataPointsFilters", "referenceDataPointsFilters", "readingStepDataPointsFilters"):
            for name, _ in modules(doc.get(key)):
                if name != "IdentityDataPointsFilter":
                    raise InvalidModuleType(f"{key}: {name} is not on the accelerated path (icp.yaml configures none)")
        for name, p in modules(doc.get("matcher")):
            if name == "KDTreeMatcher":
                cfg.matcher = name
                cfg.knn = int(p.get("knn", 1))
                cfg.max_dist = float(p.get("maxDist", math.inf))
                cfg.epsilon = float(p.get("epsilon", 0.0))
            elif name == "MirrorMatcher":
                cfg.matcher = name
                cfg.max_dist = math.inf
            else:
                raise InvalidModuleType(f"matcher {name}")
        for name, p in modules(doc.get("outlierFilters")):
            if name == "TrimmedDistOutlierFilter":
                cfg.trim_ratio = float(p.get("ratio", 0.85))
            elif name == "SurfaceNormalOutlierFilter":
                cfg.max_normal_angle = float(p.get("maxAngle", 1.57))
            elif name == "MaxDistOutlierFilter":
                cfg.max_dist_outlier = float(p.get("maxDist", 1.0))
            elif name != "NullOutlierFilter":
                raise InvalidModuleType(f"outlier filter {name}")
        for name, _ in modules(doc.get("errorMinimizer")):
            if name != "PointToPlaneErrorMinimizer":
                raise InvalidModuleType(f"error minimizer {name}")
        order = []
        for name, p in modules(doc.get("transformationCheckers")):
            order.append(name)
            if name == "CounterTransformationChecker":
                cfg.max_iters = int(p.get("maxIterationCount", 40))
            elif name == "DifferentialTransformationChecker":
                cfg.use_differential = True
                cfg.min_diff_rot = float(p.get("minDiffRotErr", 0.001))
                cfg.min_diff_trans = float(p.get("minDiffTransErr", 0.001))
                cfg.smooth_length = int(p.get("smoothLength", 3))
            else:
                raise InvalidModuleType(f"transformation checker {name}")
        if "CounterTransformationChecker" in order and "DifferentialTransformationChecker" in order:
            cfg.counter_first = order.index("CounterTransformationChecker") < order.index("DifferentialTransformationChecker")
        for key, allowed in (("inspector", "NullInspector"), ("logger", "NullLogger")):
            for name, _ in modules(doc.get(key)):
                if name != allowed:
                    raise InvalidModuleType(f"{key} {name}")
        return cfg


@dataclass
class IcpStats:
    iterations: int = 0
    max_iters_reached: bool = False
    kept_pairs: int = 0
    matched_pairs: int = 0
    point_used_ratio: float = 0.0
    weighted_point_used_ratio: float = 0.0
    last_trim_limit: float = float("nan")
    gpu_ms: float = 0.0
    candidates_examined: float = 0.0
    cells_probed: float = 0.0
    trace_T: np.ndarray = field(default_factory=lambda: np.zeros((0, 4, 4), np.float32))
    trace_limit: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    trace_kept: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def as_xyzw(points) -> np.ndarray:
    """(N,3)/(N,4) -> contiguous (N,4) fp32 = PM::DataPoints::features.data() (4 x N column-major, pad = 1)."""
    p = np.asarray(points, dtype=np.float32)
    if p.ndim != 2 or p.shape[1] not in (3, 4):
        raise ValueError("points must be (N,3) or (N,4)")
    if p.shape[1] == 3:
        p = np.concatenate([p, np.ones((p.shape[0], 1), np.float32)], axis=1)
    return np.ascontiguousarray(p, dtype=np.float32)


def _colmajor(T) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)


def _from_colmajor(t) -> np.ndarray:
    return np.asarray(t, dtype=np.float32).reshape(4, 4).T.copy()


class ICP:
    """PM::ICP for the scan-to-map chain, running on one MI355X."""

    def __init__(self, config: IcpConfig | None = None, device: int = 0):
        self.config = config or IcpConfig()
        self._L = _lib.lib()
        self._pid = os.getpid()   # _lib.forked_copy: a forked child must not destroy the handle
        self._h = C.c_void_p()
        c = self.config.to_c()
        rc = self._L.o3s_icp_create(C.byref(c), device, C.byref(self._h))
        if rc != _lib.OK:
            msg = self._L.o3s_last_error(None).decode()
            self._h = C.c_void_p()
            raise _RAISE.get(rc, RuntimeError)(f"o3s_icp_create failed ({rc}): {msg}")
        self.stats = IcpStats()
        self.device = device

    # -- lifetime ----------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not _lib.forked_copy(self):   # a forked child drops its copy of the wrapper, the handle is the parent's
                self._L.o3s_icp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == _lib.OK:
            return
        msg = self._L.o3s_last_error(self._h).decode()
        raise _RAISE.get(rc, RuntimeError)(f"[{rc}] {msg}")

    def set_stream(self, hip_stream_ptr: int | None):
        self._check(self._L.o3s_icp_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    # -- one pair sharded over the ranks of a process group (include/o3s_icp.h, o3s_icp_shard_configure) ----------
    def shard_configure(self, n_total: int, rank: int, world: int, allreduce, xbuf_ptr: int | None = None):
        """``allreduce(byte_offset, count, dtype, dev_ptr, hip_stream)`` must sum the given slice of the exchange buffer
        over all ranks in place (dtype: 0 = int32, 1 = float64).  Exceptions are reported as a failed compute."""

        def _cb(_user, dev_ptr, off, count, dtype, stream):
            try:
                allreduce(int(off), int(count), int(dtype), int(dev_ptr or 0), int(stream or 0))
                return 0
            except Exception:  # never unwind through the C frames
                import traceback

                traceback.print_exc()
                return 1

        self._shard_cb = _lib.ALLREDUCE_FN(_cb)  # keep alive as long as the handle may call it
        self._check(self._L.o3s_icp_shard_configure(self._h, rank, world, n_total, self._shard_cb, None,
                                                    C.c_void_p(xbuf_ptr or 0)))

    def shard_configure_rccl(self, n_total: int, rank: int, world: int, comm: int, capture: bool = True):
        """Native exchange: `comm` is an o3s_rccl* (include/o3s_rccl.h); the collectives never enter Python."""
        R = _lib.rccl_lib()
        fn = C.cast(R.o3s_rccl_allreduce, _lib.ALLREDUCE_FN)
        self._shard_cb = fn
        self._check(self._L.o3s_icp_shard_configure(self._h, rank, world, n_total, fn, C.c_void_p(comm), None))
        if capture:  # ncclAllReduce only enqueues on the stream it is given: kernels and collectives replay from one hipGraph
            self._check(self._L.o3s_icp_shard_set_capturable(self._h, 1))

    def shard_disable(self):
        self._check(self._L.o3s_icp_shard_configure(self._h, 0, 1, 0, _lib.ALLREDUCE_FN(0), None, None))
        self._shard_cb = None

    # -- fused path --------------------------------------------------------------------------------------------
    def init_reference(self, xyz, normals) -> bool:
        """ICP::initReference (ICP.cpp:292-328): False for an empty cloud, like the reference."""
        xyzw = as_xyzw(xyz)
        nn = None if normals is None else np.ascontiguousarray(normals, np.float32)
        rc = self._L.o3s_icp_init_reference(self._h, _fp(xyzw), _fp(nn), xyzw.shape[0])
        if rc == _lib.ERR_EMPTY_REFERENCE:
            return False
        self._check(rc)
        return True

    initReference = init_reference

    def matcher_init(self, xyz, normals=None) -> bool:
        """Matcher::init (PointMatcher.h:559-561; KDTreeMatcher::init, MatchersImpl.cpp:108-114): index the cloud AS GIVEN, no mean
        subtraction.  find_closests then takes queries in this cloud's frame; reference_mean() is (0, 0, 0)."""
        xyzw = as_xyzw(xyz)
        nn = None if normals is None else np.ascontiguousarray(normals, np.float32)
        rc = self._L.o3s_matcher_init(self._h, _fp(xyzw), _fp(nn), xyzw.shape[0])
        if rc == _lib.ERR_EMPTY_REFERENCE:
            return False
        self._check(rc)
        return True

    def init_reference_dev(self, d_xyzw_ptr: int, d_normals_ptr: int | None, M: int) -> bool:
        rc = self._L.o3s_icp_init_reference_dev(self._h, C.c_void_p(d_xyzw_ptr), C.c_void_p(d_normals_ptr or 0), M)
        if rc == _lib.ERR_EMPTY_REFERENCE:
            return False
        self._check(rc)
        return True

    def init_reference_dev_async(self, d_xyzw_ptr: int, d_normals_ptr: int | None, M: int) -> bool:
        """Same, returning once the index build is enqueued: the two device arrays must stay valid until a later call on
        this handle has waited for its stream (any compute does)."""
        rc = self._L.o3s_icp_init_reference_dev_async(self._h, C.c_void_p(d_xyzw_ptr), C.c_void_p(d_normals_ptr or 0), M)
        if rc == _lib.ERR_EMPTY_REFERENCE:
            return False
        self._check(rc)
        return True

    def wait_event(self, hip_event_ptr: int):
        """Orders this handle's stream behind a hipEvent_t recorded on another stream of the same device."""
        self._check(self._L.o3s_icp_wait_event(self._h, C.c_void_p(hip_event_ptr)))

    def _finish(self, rc, st, Tout):
        self.stats = IcpStats(st.iterations, bool(st.max_iters_reached), st.kept_pairs, st.matched_pairs, st.point_used_ratio,
                              st.weighted_point_used_ratio, st.last_trim_limit, st.gpu_ms, st.candidates_examined, st.cells_probed)
        cap = max(st.iterations, 1)
        tT = np.zeros((cap, 16), np.float32)
        tl = np.zeros(cap, np.float32)
        tk = np.zeros(cap, np.int64)
        n = self._L.o3s_icp_get_trace(self._h, _fp(tT), _fp(tl), tk.ctypes.data_as(C.POINTER(C.c_int64)), cap)
        self.stats.trace_T = np.stack([_from_colmajor(t) for t in tT[:n]]) if n else np.zeros((0, 4, 4), np.float32)
        self.stats.trace_limit = tl[:n].copy()
        self.stats.trace_kept = tk[:n].copy()
        self._check(rc)
        return _from_colmajor(Tout)

    def compute(self, reading_xyz, reading_normals, T_init) -> np.ndarray:
        """ICP::compute(reading, {}, T_init, false) (ICP.cpp:258-290 -> 332-468)."""
        xyzw = as_xyzw(reading_xyz)
        nn = None if reading_normals is None else np.ascontiguousarray(reading_normals, np.float32)
        Tin = _colmajor(T_init)
        Tout = np.zeros(16, np.float32)
        st = _lib.IcpStatsC()
        rc = self._L.o3s_icp_compute(self._h, _fp(xyzw), _fp(nn), xyzw.shape[0], _fp(Tin), _fp(Tout), C.byref(st))
        return self._finish(rc, st, Tout)

    def __call__(self, reading, reference, T_init=None):
        """ICP::operator()(readingIn, referenceIn[, T]) (ICP.cpp:232-254): initReference + compute."""
        rxyz, rn = reference
        if not self.init_reference(rxyz, rn):
            raise RuntimeError("reference cloud is empty")
        qxyz, qn = reading
        return self.compute(qxyz, qn, np.eye(4) if T_init is None else T_init)

    def set_reading(self, xyz, normals):
        xyzw = as_xyzw(xyz)
        nn = None if normals is None else np.ascontiguousarray(normals, np.float32)
        self._check(self._L.o3s_icp_set_reading(self._h, _fp(xyzw), _fp(nn), xyzw.shape[0]))

    def set_reading_dev(self, d_xyzw_ptr: int, d_normals_ptr: int | None, N: int):
        self._check(self._L.o3s_icp_set_reading_dev(self._h, C.c_void_p(d_xyzw_ptr), C.c_void_p(d_normals_ptr or 0), N))

    def compute_resident(self, T_init, with_trace: bool = True) -> np.ndarray:
        Tin = _colmajor(T_init)
        Tout = np.zeros(16, np.float32)
        st = _lib.IcpStatsC()
        rc = self._L.o3s_icp_compute_resident(self._h, _fp(Tin), _fp(Tout), C.byref(st))
        if not with_trace:
            self.stats = IcpStats(st.iterations, bool(st.max_iters_reached), st.kept_pairs, st.matched_pairs, st.point_used_ratio,
                                  st.weighted_point_used_ratio, st.last_trim_limit, st.gpu_ms, st.candidates_examined,
                                  st.cells_probed)
            self._check(rc)
            return _from_colmajor(Tout)
        return self._finish(rc, st, Tout)

    def compute_resident_launch(self, T_init) -> None:
        """First half of compute_resident (o3s_icp_compute_resident_launch): the chain is on the stream, nobody has looked at its result."""
        self._check(self._L.o3s_icp_compute_resident_launch(self._h, _fp(_colmajor(T_init))))

    def compute_resident_finish(self, with_trace: bool = True) -> np.ndarray:
        """Second half (o3s_icp_compute_resident_finish): waits, issues what an unfinished eager chain still needs, composes the pose."""
        Tout = np.zeros(16, np.float32)
        st = _lib.IcpStatsC()
        rc = self._L.o3s_icp_compute_resident_finish(self._h, _fp(Tout), C.byref(st))
        if not with_trace:
            self.stats = IcpStats(st.iterations, bool(st.max_iters_reached), st.kept_pairs, st.matched_pairs, st.point_used_ratio,
                                  st.weighted_point_used_ratio, st.last_trim_limit, st.gpu_ms, st.candidates_examined,
                                  st.cells_probed)
            self._check(rc)
            return _from_colmajor(Tout)
        return self._finish(rc, st, Tout)

    def get_max_num_iterations_reached(self) -> bool:
        """ICP::getMaxNumIterationsReached (PointMatcher.h:786)."""
        return self.stats.max_iters_reached

    getMaxNumIterationsReached = get_max_num_iterations_reached

    def reference_mean(self) -> np.ndarray:
        m = np.zeros(3, np.float32)
        self._check(self._L.o3s_icp_reference_mean(self._h, _fp(m)))
        return m

    def reading_order(self, n: int) -> np.ndarray:
        """order[s] = input index of the point the chain handles in slot s (o3s_icp_get_reading_order)."""
        order = np.zeros(n, np.int32)
        got = self._L.o3s_icp_get_reading_order(self._h, _ip(order), n)
        return order[:got]

    def set_profiling(self, on: bool):
        self._check(self._L.o3s_icp_set_profiling(self._h, int(on)))

    def host_split(self):
        """(issue_us, wait_us, queries, prepare_gpu_us) of the last compute on this handle (o3s_icp_host_split)."""
        out = (C.c_double * 4)()
        self._check(self._L.o3s_icp_host_split(self._h, out))
        return float(out[0]), float(out[1]), int(out[2]), float(out[3])

    def host_split_ex(self) -> dict:
        """o3s_icp_host_split_ex of the last compute on this handle: the split plus what ended the waits and how the chain went out."""
        out = (C.c_double * 8)()
        self._check(self._L.o3s_icp_host_split_ex(self._h, out))
        return {"host_issue_us": float(out[0]), "host_wait_us": float(out[1]), "queries": int(out[2]), "gpu_prepare_us": float(out[3]),
                "waits_ended_by_post": int(out[4]), "waits_ended_by_event": int(out[5]), "waits_ended_by_stream_guard": int(out[6]),
                "issued": ("eager", "captured", "replayed")[int(out[7])]}

    def kernel_ms(self):
        ms = np.zeros(5, np.float32)
        n = np.zeros(5, np.int32)
        self._check(self._L.o3s_icp_kernel_ms(self._h, _fp(ms), _ip(n)))
        names = ["match", "classify", "sel_finish", "normal_eq", "solve"]
        return {k: (float(m), int(c)) for k, m, c in zip(names, ms, n)}

    def profile_match(self, T_iter_refmean, reps: int = 50, flags: int = 0) -> float:
        """Average ms per launch of the matcher kernel alone (HIP events on the library stream)."""
        Tin = _colmajor(T_iter_refmean)
        ms = C.c_float()
        self._check(self._L.o3s_icp_profile_match(self._h, _fp(Tin), reps, flags, C.byref(ms)))
        return float(ms.value)

    # -- module-level path (PM::Matcher / OutlierFilters / ErrorMinimizer granularity) ------------------------------
    def find_closests(self, query_xyz):
        """Matcher::findClosests (MatchersImpl.cpp:117-132): query already in the <refMean> frame."""
        q = as_xyzw(query_xyz)
        ids = np.zeros(q.shape[0], np.int32)
        d2 = np.zeros(q.shape[0], np.float32)
        self._check(self._L.o3s_icp_find_closests(self._h, _fp(q), q.shape[0], _ip(ids), _fp(d2)))
        return ids, d2

    def outlier_weights(self, reading_normals, ids, dists2):
        nn = None if reading_normals is None else np.ascontiguousarray(reading_normals, np.float32)
        ids = np.ascontiguousarray(ids, np.int32)
        d2 = np.ascontiguousarray(dists2, np.float32)
        w = np.zeros(ids.shape[0], np.float32)
        self._check(self._L.o3s_icp_outlier_weights(self._h, _fp(nn), _ip(ids), _fp(d2), ids.shape[0], _fp(w)))
        return w

    def minimize(self, reading_xyz, ids, dists2, weights):
        """ErrorMinimizer::compute(reading, reference, weights, matches) -> (T 4x4, A 6x6, b, x)."""
        q = as_xyzw(reading_xyz)
        ids = np.ascontiguousarray(ids, np.int32)
        d2 = np.ascontiguousarray(dists2, np.float32)
        w = np.ascontiguousarray(weights, np.float32)
        T = np.zeros(16, np.float32)
        A = np.zeros(36, np.float32)
        b = np.zeros(6, np.float32)
        x = np.zeros(6, np.float32)
        self._check(self._L.o3s_icp_minimize(self._h, _fp(q), _ip(ids), _fp(d2), _fp(w), q.shape[0], _fp(T), _fp(A), _fp(b), _fp(x)))
        return _from_colmajor(T), A.reshape(6, 6).T.copy(), b, x


def compute_batch(icps, T_inits):
    """o3s_icp_compute_batch: run compute_resident on several independent (reference, reading) pairs concurrently.

    ``icps``: ICP handles, each with its reference initialised and a resident reading.  Returns (poses, statuses, stats):
    poses[k] is a 4x4 fp32 array (None when statuses[k] != 0); the call never raises for per-pair ICP failures — like
    o3d_slam::Mapper it leaves the decision to the caller (Mapper.cpp:420-422 keeps the prior on any error)."""
    n = len(icps)
    if n == 0:
        return [], [], []
    L = icps[0]._L
    hs = (C.c_void_p * n)(*[i._h for i in icps])
    Tin = np.ascontiguousarray(np.stack([_colmajor(T) for T in T_inits]), np.float32)
    Tout = np.zeros((n, 16), np.float32)
    st = (_lib.IcpStatsC * n)()
    codes = np.zeros(n, np.int32)
    rc = L.o3s_icp_compute_batch(hs, n, _fp(Tin), _fp(Tout), st, _ip(codes))
    if rc != _lib.OK:
        raise ValueError(f"o3s_icp_compute_batch: bad arguments ({rc})")
    poses, stats = [], []
    for k in range(n):
        s = st[k]
        icps[k].stats = IcpStats(s.iterations, bool(s.max_iters_reached), s.kept_pairs, s.matched_pairs, s.point_used_ratio,
                                 s.weighted_point_used_ratio, s.last_trim_limit, s.gpu_ms, s.candidates_examined, s.cells_probed)
        stats.append(icps[k].stats)
        poses.append(_from_colmajor(Tout[k]) if codes[k] == _lib.OK else None)
    return poses, codes.tolist(), stats
