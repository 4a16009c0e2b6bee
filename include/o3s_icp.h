/*
 * o3s_icp.h — C ABI of libo3dslam_icp_hip.so: MI355X (gfx950) scan-to-map ICP for open3d_slam / libpointmatcher.
 *
 * Drop-in boundary.  Paths are relative to the upstream reference checkout
 *   LPM  = libpointmatcher/pointmatcher
 *   O3S  = open3d_slam_rsl/open3d_slam/open3d_slam
 *   CONV = open3d_slam_rsl/open3d_utils/open3d_conversions
 * The two fused entry points replace the two libpointmatcher calls made by
 * o3d_slam::Mapper::addRangeMeasurement:
 *     icp_.initReference(activeSubmapPm_.dataPoints_)            O3S/src/Mapper.cpp:363   -> o3s_icp_init_reference
 *     icp_.compute(reading, empty, T_init, false)                O3S/src/Mapper.cpp:393   -> o3s_icp_compute
 * (PM::ICP::initReference LPM/ICP.cpp:292-328, PM::ICP::compute LPM/ICP.cpp:258-290,332-468).
 * The module-level entry points mirror the libpointmatcher plugin interfaces so that PM::Matcher /
 * PM::OutlierFilter / PM::ErrorMinimizer subclasses can forward to them (LPM/PointMatcher.h:547-693).
 * The open3d_slam-side helpers mirror getVoxelIdx / voxelizeWithinCroppingVolume / CroppingVolume::crop /
 * open3dToPointmatcher (O3S/include/open3d_slam/VoxelHashMap.hpp:48-51, O3S/src/helpers.cpp:117-192,
 * O3S/src/croppers.cpp:76-106, CONV/src/open3d_conversions.cpp:57-118).
 *
 * Conventions
 *  - No exceptions cross the ABI.  Every function returns an o3s_status; o3s_last_error() gives text.
 *  - All matrices are column-major (Eigen default).  "xyzw" is PM::DataPoints::features.data(): 4 x N, i.e. AoS
 *    [x y z pad] per point, pad == 1.  "normals" is the 3 x N "normals" descriptor block, AoS [nx ny nz] per point.
 *  - The caller owns every host buffer; the library copies in during the call and never keeps a host pointer.
 *    Device state (the indexed reference, the uploaded reading) lives in the handle.
 *  - One handle = one device + one HIP stream; a handle is not re-entrant (the reference serialises these calls under
 *    mapManipulationMutex_, O3S/src/Mapper.cpp:351,388).  Different handles may be used from different threads.
 *  - There is no CPU fallback: if no gfx950 device / code object is usable, o3s_icp_create fails with O3S_ERR_HIP.
 */
#ifndef O3S_ICP_H
#define O3S_ICP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O3S_ABI_VERSION 1

/* Status codes map 1:1 to the reference's exceptions so a C++ shim can rethrow them. */
typedef enum o3s_status {
  O3S_OK = 0,
  O3S_ERR_EMPTY_REFERENCE = 1, /* initReference(empty) returns false            LPM/ICP.cpp:295-298                  */
  O3S_ERR_EMPTY_READING = 2,   /* runtime_error "reading point cloud is empty"   LPM/ICP.cpp:357-359                  */
  O3S_ERR_BAD_SHAPE = 3,       /* runtime_error (matrix shapes / missing normals) LPM/ICP.cpp:340-346                 */
  O3S_ERR_NOT_INITIALIZED = 4, /* matcher not initialised (compute before initReference)                              */
  O3S_ERR_NO_MATCHES = 5,      /* ConvergenceError "No matches available..."      LPM/Matches.cpp:76-77               */
  O3S_ERR_NO_POINTS = 6,       /* ConvergenceError "no point to minimize"         LPM/ErrorMinimizer.cpp:75-77        */
  O3S_ERR_NAN = 7,             /* ConvergenceError "abs rotation norm not a number" LPM/TransformationCheckersImpl.cpp:154-157 */
  O3S_ERR_NOT_RIGID = 8,       /* TransformationError (|1 - det R| > 1e-3)        LPM/TransformationsImpl.cpp:73-74   */
  O3S_ERR_BAD_CONFIG = 9,      /* invalid o3s_icp_config                                                              */
  O3S_ERR_HIP = 10,            /* HIP runtime failure / no usable gfx950 device                                      */
  O3S_ERR_BAD_ARGUMENT = 11
} o3s_status;

/* Mirror of the ICP chain in open3d_slam_ros/param/icp.yaml (module names in comments). */
typedef struct o3s_icp_config {
  int32_t matcher;          /* 0 = KDTreeMatcher{knn 1} (exact 1-NN, epsilon ignored), 1 = MirrorMatcher             */
  float max_dist;           /* KDTreeMatcher.maxDist [m]; +inf allowed                            icp.yaml:14 (0.5)  */
  float epsilon;            /* KDTreeMatcher.epsilon — accepted and ignored: the search is exact  icp.yaml:15 (0.01) */
  float trim_ratio;         /* TrimmedDistOutlierFilter.ratio; < 0 = filter absent                icp.yaml:20 (0.90) */
  float max_normal_angle;   /* SurfaceNormalOutlierFilter.maxAngle [rad]; < 0 = absent            icp.yaml:22 (1.57) */
  float max_dist_outlier;   /* MaxDistOutlierFilter.maxDist [m]; < 0 = absent                     icp.yaml:18 (off)  */
  int32_t use_differential; /* DifferentialTransformationChecker present                          icp.yaml:30        */
  float min_diff_rot;       /* minDiffRotErr [rad]                                                icp.yaml:31 (1e-3) */
  float min_diff_trans;     /* minDiffTransErr [m]                                                icp.yaml:32 (1e-2) */
  int32_t smooth_length;    /* smoothLength (<= 15)                                               icp.yaml:33 (3)    */
  int32_t max_iters;        /* CounterTransformationChecker.maxIterationCount; <= 0 = absent      icp.yaml:35 (15)   */
  int32_t counter_first;    /* 1 if the Counter checker precedes the Differential one in the YAML list               */
  float grid_cell;          /* spatial-index cell edge [m]; 0 = choose automatically                                  */
  int32_t sort_queries;     /* 1 = process the reading in spatial (grid) order for cache locality (default 1)         */
  int32_t use_graph;        /* 1 = replay the iteration chain from a hipGraph (default 1)                             */
  int32_t match_stats;      /* 1 = count candidates / cell rows examined by the matcher (slower; default 0)           */
  int32_t reserved[4];
} o3s_icp_config;

typedef struct o3s_icp_stats {
  int32_t iterations;               /* iterations executed                                                          */
  int32_t max_iters_reached;        /* ICP::getMaxNumIterationsReached() (a flag, not an error) LPM/ICP.cpp:441-445 */
  int64_t kept_pairs;               /* pairs that entered the minimiser in the last iteration                       */
  int64_t matched_pairs;            /* finite-distance matches in the last iteration                                */
  float point_used_ratio;           /* ErrorElements::pointUsedRatio          LPM/ErrorMinimizer.cpp:139            */
  float weighted_point_used_ratio;  /* ErrorElements::weightedPointUsedRatio  LPM/ErrorMinimizer.cpp:140            */
  float last_trim_limit;            /* squared-distance trim limit of the last iteration (NaN if no Trimmed filter) */
  float gpu_ms;                     /* device time of the iteration chain: wall_clock64 stamps written by the chain's
                                       kernels (first matcher launch -> the launch that posts the final state); no HIP events */
  double candidates_examined;       /* total reference points distance-tested by the matcher over the call          */
  double cells_probed;              /* total cell rows probed by the matcher over the call                          */
} o3s_icp_stats;

typedef struct o3s_icp o3s_icp;

/* Fills cfg with the values of open3d_slam_ros/param/icp.yaml. */
void o3s_icp_default_config(o3s_icp_config* cfg);

/* Lifetime. device = HIP device ordinal. */
int o3s_icp_create(const o3s_icp_config* cfg, int device, o3s_icp** out);
void o3s_icp_destroy(o3s_icp* h);
const char* o3s_last_error(const o3s_icp* h); /* h may be NULL: error of the last failed create on this thread */
int o3s_abi_version(void);
/* Run the handle's work on a caller-supplied hipStream_t (e.g. a framework's current stream); NULL restores the
 * handle's own stream. */
int o3s_icp_set_stream(o3s_icp* h, void* hip_stream);

/* ---- fused path ------------------------------------------------------------------------------------------------ */
/* PM::ICP::initReference (LPM/ICP.cpp:292-328): copy the reference, subtract its fp32 mean, build the matcher index
 * (a dense voxel grid replaces libnabo's kd-tree).  normals may be NULL (then point-to-plane compute fails BAD_SHAPE). */
int o3s_icp_init_reference(o3s_icp* h, const float* xyzw, const float* normals, int64_t M);
/* Same, inputs already in HBM (device pointers, same layouts). */
int o3s_icp_init_reference_dev(o3s_icp* h, const void* d_xyzw, const void* d_normals, int64_t M);
/* Same, but returns as soon as the index build is enqueued on the handle's stream: the two arrays must stay valid and
 * unchanged until a later call on this handle has waited for the stream (any compute does).  Used by the resident
 * submap (o3s_submap_set_reference), whose patch buffers live until the next set_reference. */
int o3s_icp_init_reference_dev_async(o3s_icp* h, const void* d_xyzw, const void* d_normals, int64_t M);
/* Waits on the host until everything enqueued on the handle's stream has finished (after a failed compute nothing else has:
 * buffers handed over with o3s_icp_init_reference_dev_async / o3s_icp_set_reading_dev may be rewritten after this). */
int o3s_icp_synchronize(o3s_icp* h);
/* Orders everything enqueued on the handle's stream from now on behind `hip_event` (a hipEvent_t recorded on another
 * stream of the same device) — the device-side hand-over between a producer stream and this handle, no host wait. */
int o3s_icp_wait_event(o3s_icp* h, void* hip_event);

/* PM::ICP::compute(reading, {}, T_init, false) (LPM/ICP.cpp:258-290 -> 332-468).  normals may be NULL (the
 * SurfaceNormalOutlierFilter then passes everything, LPM/OutlierFiltersImpl.cpp:268-277).  stats may be NULL. */
int o3s_icp_compute(o3s_icp* h, const float* xyzw, const float* normals, int64_t N, const float T_init[16],
                    float T_out[16], o3s_icp_stats* stats);
/* Split form: upload once (host or device source), then run compute on the resident reading any number of times. */
int o3s_icp_set_reading(o3s_icp* h, const float* xyzw, const float* normals, int64_t N);
int o3s_icp_set_reading_dev(o3s_icp* h, const void* d_xyzw, const void* d_normals, int64_t N);
/* Hint for the resident reading (cleared by every set_reading): its points already come in a spatially coherent order —
 * e.g. out of a voxel down-sampler in voxel order — so the per-call counting sort that makes neighbouring lanes touch
 * neighbouring map cells is skipped.  Results do not depend on it (the matcher is exact for any order). */
int o3s_icp_reading_is_spatially_sorted(o3s_icp* h, int sorted);
int o3s_icp_compute_resident(o3s_icp* h, const float T_init[16], float T_out[16], o3s_icp_stats* stats);
/* The same call in two halves (round 5): _launch enqueues the reading's preparation and the chain on the handle's stream and returns
 * without looking at the result — a chain replayed from a graph goes out whole, one issued eagerly (a reading of a new size) goes
 * out as far as the handle's previous call needed; _finish waits, issues what is left if the chain is not done, and composes the
 * pose.  Same iterations, same bits as o3s_icp_compute_resident (where the host looks never decides what the chain computes).  A
 * host uses the gap for work that does not need the pose: MapperHip counts the map patch (Mapper.cpp:328-336) there on the scans
 * that do not renew the reference.  One call in flight per handle; nothing else may be called on the handle between the halves. */
int o3s_icp_compute_resident_launch(o3s_icp* h, const float T_init[16]);
int o3s_icp_compute_resident_finish(o3s_icp* h, float T_out[16], o3s_icp_stats* stats);
/* BASELINE config 3 (a collection of independent scan/submap pairs, e.g. loop-closure candidates, the serial loop at
 * O3S/src/PlaceRecognition.cpp:71): handles[k] is one pair — its own reference (o3s_icp_init_reference) and resident
 * reading (o3s_icp_set_reading).  All n chains are issued before any is waited for; each handle owns a stream, so the
 * chains overlap on the GPU (and handles may sit on different devices).  T_inits / T_outs: n x 16 floats.
 * statuses[k] receives pair k's o3s_status; the return value only reports argument errors. */
int o3s_icp_compute_batch(o3s_icp* const* handles, int32_t n, const float* T_inits, float* T_outs, o3s_icp_stats* stats,
                          int32_t* statuses);
/* ---- one pair sharded over several GPUs (SURVEY.md 8(e) mode 2) ---------------------------------------------------
 * Every rank holds the SAME reference (o3s_icp_init_reference) and a disjoint slice of the reading (o3s_icp_set_reading);
 * after o3s_icp_shard_configure, o3s_icp_compute / _compute_resident run the chain on the slice and form the three
 * global quantities of an iteration by all-reducing (sum) regions of one device buffer through `fn`, THREE times per
 * iteration (two without a Trimmed filter), each region reduced in place where the kernels left it:
 *   int32 x R x 2048                         : level-1 radix histogram of Matches::getDistsQuantile (LPM/Matches.cpp:61-87) in
 *                                              R = 16 >> floor(log2(world)), at least 1, replicas — 16, 8, 8, 4, 4, 4, 4, 2 for worlds
 *                                              1 .. 8: a power of two, the matcher picks a block's replica with a mask; it spreads
 *                                              its flushes over R replicas, and a rank's share of the work shrinks with the world size
 *   int32 x 8192                             : level 2 (thirteen bits in this mode, so that level 3 has seven)
 *   float64 x (128 + 34 x 128 + 34 x blocks) : level-3 counts, and the RAW (un-centred, fp64) moments of the kept pairs — per level-3
 *                                              bin for the handful of pairs the limit's last seven bits decide, per block for all
 *                                              others; blocks = ceil(n_total / world / 512), the same on every rank.  After it
 *                                              every rank knows the exact limit (the global element), |K|, the means of the kept pairs
 *                                              (LPM/ErrorMinimizers/PointToPlane.cpp:263-264) and — centring the moments
 *                                              algebraically with those means — A and b (PointToPlane.cpp:283-306)
 * o3s_icp_shard_bytes_per_iteration(world, n_total) says what that adds up to (91 KB at eight ranks for a 100 k-point reading).
 * Three, where rounds 3-4 needed four: the normal equations no longer wait for the end of the selection — raw moments need neither
 * the means nor the limit's last bits.  The price: the reference (and the unsharded chain) rounds p - mean and every product per
 * pair in fp32, the raw moments are exact products centred once — limit, |K| and the iteration count equal the unsharded chain's,
 * the pose agrees with it to the unsharded chain's own fp32 rounding noise (<= 1e-6 m on well-conditioned pairs, 1e-5 m on the
 * worst case of the test suite), well inside the 1e-4 of the north star.
 * fn must enqueue an in-place sum all-reduce of `count` elements at `dev_ptr` on `hip_stream` (or ordered after it, e.g.
 * ncclAllReduce on that stream) and return 0; every rank must receive bit-identical sums (RCCL / gloo both do).  The
 * solve and the transformation checkers run replicated, so every rank returns the same pose and iteration count.
 * byte_offset is dev_ptr's offset inside the exchange buffer: hosts that own the buffer (xbuf_dev, at least
 * o3s_icp_shard_exchange_bytes() bytes, 8-byte aligned) can address their own view of it; xbuf_dev NULL lets the
 * library allocate it.  n_total = reading points over all ranks (ErrorMinimizer.cpp:139 ratios); each slice must hold
 * at least one point.  world <= 1 with fn NULL switches the mode off.  KDTreeMatcher only; graph replay only after
 * o3s_icp_shard_set_capturable. */
#define O3S_XCHG_INT32 0
#define O3S_XCHG_FLOAT64 1
typedef int (*o3s_allreduce_fn)(void* user, void* dev_ptr, int64_t byte_offset, int64_t count, int32_t dtype,
                                void* hip_stream);
int o3s_icp_shard_configure(o3s_icp* h, int32_t rank, int32_t world, int64_t n_total, o3s_allreduce_fn fn, void* user,
                            void* xbuf_dev);
int64_t o3s_icp_shard_exchange_bytes(void);
/* Bytes the three exchanges of one iteration move per rank (the sum of their regions) for a reading of n_total points over
 * `world` ranks. */
int64_t o3s_icp_shard_bytes_per_iteration(int32_t world, int64_t n_total);
/* The caller's promise that `fn` does nothing but enqueue work on the hip_stream it is given (o3s_rccl_allreduce =
 * ncclAllReduce on that stream does; a callback that waits on the host or hops through Python does not): the sharded chain
 * — kernels AND the three collectives of every iteration — is then captured in a hipGraph the second time the same shapes
 * come back and replayed from then on, like the unsharded chain.  Every rank must make the same promise.  Cleared by
 * o3s_icp_shard_configure. */
int o3s_icp_shard_set_capturable(o3s_icp* h, int yes);

/* Per-iteration trace of the last compute: T_iter (16 floats, column-major) after each iteration, the trim limit and
 * the kept-pair count.  cap = capacity of the arrays in iterations; returns the number of iterations written. */
int o3s_icp_get_trace(const o3s_icp* h, float* T_iters, float* limits, int64_t* kept, int32_t cap);
/* Processing order of the last prepared reading: order[s] = input index of the point the chain handles in slot s (the
 * reading is counting-sorted by grid bin, STABLY: inside a bin the input order is kept, so the order — and with it the
 * order of every fp64 sum of the chain — is a function of the input alone).  Returns the number of entries written
 * (<= cap), 0 when no reading has been prepared.  Diagnostics / tests. */
int64_t o3s_icp_get_reading_order(const o3s_icp* h, int32_t* order, int64_t cap);
/* Mean subtracted from the reference at init (T_refIn_refMean translation, LPM/ICP.cpp:313-314). */
int o3s_icp_reference_mean(const o3s_icp* h, float mean3[3]);
/* Split of the last o3s_icp_compute / _compute_resident on this handle, in microseconds: out4[0] = issuing the call's work on
 * the host (uploads, launches), out4[1] = the host waiting for the chain's post, out4[2] = event / stream queries made while
 * waiting, out4[3] = device time from the call's first kernel to its first matcher launch (transform + sort of the reading).
 * Diagnostics. */
int o3s_icp_host_split(const o3s_icp* h, double out4[4]);
/* Same, plus what ended the host's waits and how the chain was issued: out8[0..3] as above, out8[4] = waits ended by the chain's
 * post (the normal case: a load from host memory saw it), out8[5] = waits ended by the event recorded behind the last launch
 * (everything issued has run and the chain is not done: the next chunk goes out), out8[6] = waits ended by the stream guard (the
 * stream queried after 2 ms without either — never in a healthy run), out8[7] = 0 the chain was issued eagerly, 1 captured into a
 * hipGraph in this call and replayed, 2 replayed from the cached graph.  Diagnostics (bench.py's per-call distribution). */
int o3s_icp_host_split_ex(const o3s_icp* h, double out8[8]);
/* Average device time (ms) per launch of each kernel of the iteration chain during the last compute() that ran with
 * profiling on (o3s_icp_set_profiling(h, 1)): [0] match, [1] select, [2] centroid, [3] normal equations, [4] solve.
 * Profiling brackets every launch with HIP events on the handle's stream and disables graph replay. */
int o3s_icp_set_profiling(o3s_icp* h, int on);
int o3s_icp_kernel_ms(const o3s_icp* h, float avg_ms[5], int32_t launches[5]);
/* Average device time (ms) of `reps` back-to-back launches of the matcher kernel alone on the resident reading, timed
 * with two HIP events on the handle's stream.  T_iter (column-major, <refMean> frame) is the pose the matcher sees —
 * pass a converged one (last entry of o3s_icp_get_trace) to time the steady state.  Used by bench.py for the roofline
 * line.  flags: 0 in the product library — any of the low eight bits is refused with O3S_ERR_BAD_ARGUMENT there (the kernel
 * switches they name exist in the test-hook build only, `make hooks`); 0x100 wipes the incumbents first and launches what the
 * first iteration of a call launches. */
int o3s_icp_profile_match(o3s_icp* h, const float T_iter[16], int32_t reps, int32_t flags, float* avg_ms);

/* Measured HBM stream-copy ceiling of the device: `reps` float4 copies of `bytes` bytes (source and
 * destination each; use a size well past the 256 MB Infinity Cache), timed with HIP events; *gbs = read + written bytes
 * per second in GB/s.  bench.py reports it beside the 8 TB/s spec peak. */
int o3s_stream_copy_gbs(int device, int64_t bytes, int32_t reps, double* gbs);

/* ---- module-level path (libpointmatcher plugin granularity) -------------------------------------------------- */
/* Matcher::init (interface LPM/PointMatcher.h:559-561; KDTreeMatcher::init, LPM/MatchersImpl.cpp:108-114): index the cloud
 * AS GIVEN — no mean is computed or subtracted; ICP::initReference has already centred what it hands to its matcher
 * (LPM/ICP.cpp:313-324) and a second subtraction of that cloud's ~1e-9 residual mean would move small coordinates by an
 * ulp.  o3s_icp_find_closests then takes its query in the frame of this very cloud, ids index it, and
 * o3s_icp_reference_mean reports (0, 0, 0).  normals (3 x M) may be NULL; the outlier / minimise entry points use them. */
int o3s_matcher_init(o3s_icp* h, const float* xyzw, const float* normals, int64_t M);
/* Matcher::findClosests (LPM/MatchersImpl.cpp:117-132): query 4 x N already in the <refMean> frame.  ids: N int32
 * (reference index, -1 = none); dists2: N floats (SQUARED distance, +inf = none). */
int o3s_icp_find_closests(o3s_icp* h, const float* query_xyzw, int64_t N, int32_t* ids, float* dists2);
/* OutlierFilters::compute (LPM/OutlierFilter.cpp:64-103) for the configured chain.  reading_normals may be NULL. */
int o3s_icp_outlier_weights(o3s_icp* h, const float* reading_normals, const int32_t* ids, const float* dists2,
                            int64_t N, float* weights);
/* ErrorMinimizer::compute(reading, reference, weights, matches) (LPM/ErrorMinimizer.cpp:218-232 ->
 * LPM/ErrorMinimizers/PointToPlane.cpp:241-368).  Outputs the 4x4 step; optionally A (6x6 col-major), b, x. */
int o3s_icp_minimize(o3s_icp* h, const float* reading_xyzw, const int32_t* ids, const float* dists2,
                     const float* weights, int64_t N, float T_out[16], float A_out[36], float b_out[6],
                     float x_out[6]);

#ifdef __cplusplus
}
#endif
#endif /* O3S_ICP_H */
