// o3s_icp.hip — host side of libo3dslam_icp_hip.so (C ABI declared in include/o3s_icp.h).
//
// Build (gfx950 only, no other targets, no CPU fallback):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -Iinclude o3s_icp.hip -o libo3dslam_icp_hip.so
#include "../../include/o3s_icp.h"

#include <hip/hip_runtime.h>
#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "icp_kernels.h"
#include "icp_shard_kernels.h"
#include "icp_types.h"

#pragma clang fp contract(off)

using namespace o3s;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  uint64_t* gen = nullptr;  // the owning handle's allocation generation: bumped whenever this buffer moves, which
                            // invalidates every captured graph that has the old address baked in
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (gen) ++*gen;
    // the first allocation is tight (a 20 M-point map is allocated once); a buffer that has to grow AGAIN belongs to a growing
    // map patch, and every move costs a device-wide synchronisation plus a re-capture of the chain's graph: double it
    const size_t had = cap;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if (had && want < 2 * had) want = 2 * had;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (gen) ++*gen;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct HostStage {  // pinned staging block for small H2D / D2H transfers
  IcpState state;
  float T0[16];
  double ref_part[1024 * 3];  // init_reference: per-block sums and bounds of k_ref_stats (a pageable landing area costs a
  float ref_bb[1024 * 6];     // staged round trip per copy)
  uint32_t n_occ;
};

constexpr int kNumKernels = 5;
#ifndef O3S_FUSE_TAIL_DEFAULT
#define O3S_FUSE_TAIL_DEFAULT false  // measured (round 4, C2): the in-launch hand-over lost to the launch boundary it replaces — see DESIGN.md
#endif
constexpr size_t kHistWords = (size_t)kHistReplicas * kHistBins + 1024;  // level-1 replicas + the level-2 histogram right behind them

}  // namespace

struct o3s_icp {
  o3s_icp_config cfg;
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::string err;

  // reference (matcher index)
  bool ref_ready = false;
  bool ref_has_normals = false;
  int64_t M = 0;
  float mean[3] = {0, 0, 0};
  GridParams grid{};
  size_t ncells = 0;
  int qf = 1, qnx = 1, qny = 1, qnz = 1;  // coarsened grid the reading is sorted on
  size_t qcells = 1;
  DevBuf d_ref_in, d_refn_in;  // staging for host-supplied references
  DevBuf d_ref, d_refn, d_cell_start, d_cell_tmp, d_qstart, d_orig_to_sorted, d_cell_of, d_scan_sums, d_ref_part, d_ref_bb;
  // the first-iteration index (dense maps only, init_reference_impl step 4): the same points sorted on a grid of 1.5 x the cell edge
  bool have_grid1 = false;
  GridParams grid1{};
  DevBuf d_ref1, d_refn1, d_cell_start1;
  bool far_rows = false;  // the matcher's far search is the row-disc search (finite maxDist), else the ring search

  // reading
  bool reading_ready = false;
  bool read_has_normals = false;
  bool reading_presorted = false;  // o3s_icp_reading_is_spatially_sorted: the per-call counting sort of the reading is skipped
  int N = 0;
  int prepared_N = 0;  // points of the last prepare_reading (d_perm holds their processing order)
  const void* ext_xyzw = nullptr;  // device pointers supplied by set_reading_dev (not owned)
  const void* ext_n = nullptr;
  DevBuf d_in_xyzw, d_in_n, d_t, d_r, d_perm, d_qcell;
  DevBuf d_qcount;  // per-bin counts of the reading's sort [qcells] + tile totals behind them: ALL ZEROS between calls
  size_t qcount_words = 0;

  // iteration chain
  DevBuf d_mn;  // matched reference normal of every query (written by k_classify, streamed by k_normal_eq)
  DevBuf d_mq;  // matched reference point of every query (k_match2: this iteration's output, the next one's pruning bound)
  DevBuf d_cand_cnt;
  DevBuf d_sel_part2, d_park;  // k_sel_partial (large readings): block partials [7][blocks]; parked records [kParkRecs] + their order keys
  DevBuf d_pos, d_d2, d_hist, d_cand, d_sel, d_cent, d_ne, d_state, d_T0, d_trace_T, d_trace_limit, d_trace_kept;
  DevBuf d_mod_a, d_mod_b, d_mod_c, d_mod_d;  // module-level scratch
  HostStage* stage = nullptr;                // pinned
  // mailbox: host-coherent pinned words a kernel writes and the host polls ([0] value, [1] sequence number, [2..10] the
  // reference statistics) — init_reference's two read-backs without a copy or a stream synchronisation
  uint32_t* mb = nullptr;
  uint32_t* mb_dev = nullptr;
  uint32_t mb_seq = 0;
  // the chain's own mailbox (icp_types.h, HostPost): the kernel that closes an iteration posts the progress word and, when the
  // chain is done, the whole state — compute() polls it; no copy command, no stream synchronisation on the per-call path
  HostPost* post = nullptr;
  HostPost* post_dev = nullptr;
  uint32_t call_seq = 0;     // sequence number of the compute() in flight (k_read_prep writes it into the state)
  int pend_issued = 0;       // iterations the call in flight has issued so far
  double wall_clock_khz = 100000.0;  // wall_clock64 rate (hipDeviceAttributeWallClockRate)
  // host-side split of the last compute(): microseconds spent issuing (compute_launch) and waiting (wait_post), stream queries made
  double host_issue_us = 0.0, host_wait_us = 0.0;
  int host_queries = 0;
  // what ended the waits of the last compute(): the chain's post, the `drained` event, the 2 ms stream guard; and how the chain was
  // issued (0 eager, 1 captured in this call and replayed, 2 replayed from the cached graph) — o3s_icp_host_split_ex
  int wait_by_post = 0, wait_by_event = 0, wait_by_guard = 0, issue_mode = 0;
  // set by o3s_icp_compute_batch while several chains share the GPU: the fused k_sel_ne trades redundant work and most of a
  // CU's LDS for one chain's latency, which costs throughput when the CUs are wanted by other chains (64 pairs of config 3:
  // 9.7 ms with the two kernels apart, 10.8 ms fused)
  bool many_in_flight = false;
  int trace_cap = 0;
  int last_iters = 0;

  // a compute() in flight between compute_launch and compute_finish
  bool pend_valid = false;
  int pend_graph_left = 0, pend_graph_chunk = 0;  // iterations the chunked graph replay has not issued yet (compute_finish)
  // o3s_icp_compute_resident_launch: an eagerly issued chain does not look at its `done` flag inside the launch — it issues as many
  // iterations as the last call needed and returns; compute_finish looks, and issues the rest two at a time if the chain is not done
  bool defer_looks = false, pend_eager = false;
  int pend_iters_cap = 0, pend_look_step = 2;
  float pend_Tc[16], pend_T0[16];
  ChainParams pend_cp{};

  // Every DevBuf of the handle reports (re)allocations here; the graph key carries the value, so a graph is never
  // replayed after ANY buffer a captured kernel points at has moved (the key's pointer list alone missed d_cand & co.).
  uint64_t alloc_gen = 0;
  std::vector<DevBuf*> all_bufs() {
    return {&d_ref_in, &d_refn_in, &d_ref, &d_refn, &d_cell_start, &d_cell_tmp, &d_qstart, &d_orig_to_sorted, &d_cell_of, &d_scan_sums,
            &d_ref_part, &d_ref_bb, &d_ref1, &d_refn1, &d_cell_start1, &d_in_xyzw, &d_in_n, &d_t, &d_r, &d_perm, &d_qcell, &d_qcount, &d_pos, &d_d2, &d_hist, &d_cand, &d_cand_cnt, &d_sel_part2, &d_park, &d_sel, &d_cent,
            &d_ne, &d_state, &d_T0, &d_mq, &d_mn, &d_trace_T, &d_trace_limit, &d_trace_kept, &d_mod_a, &d_mod_b, &d_mod_c, &d_mod_d, &shard.own};
  }

  // graph cache
  hipGraphExec_t graph_exec = nullptr;
  struct GraphKey {
    int N = -1, iters = -1, nb = -1, has_n = -1;
    uint64_t gen = 0;
    const void* ptrs[8] = {nullptr};
    ChainParams cp{};
    GridParams g{};
  } graph_key, graph_candidate;
  bool graph_candidate_valid = false;

  // one-pair-sharded mode (o3s_icp_shard_configure): this handle holds one slice of the reading
  struct Shard {
    bool active = false;
    int rank = 0, world = 1;
    int64_t n_total = 0;
    o3s_allreduce_fn fn = nullptr;
    void* user = nullptr;
    bool capturable = false;  // fn only enqueues stream work (ncclAllReduce): the sharded chain may be captured in a hipGraph
    uint8_t* xbuf = nullptr;  // exchange buffer (kXchgBytes), caller's or `own`
    DevBuf own;
  } shard;

  int eager_hint = 4;  // iterations the last call needed: where the eager (un-graphed) chain first looks at the `done` flag
  bool fuse_tail = O3S_FUSE_TAIL_DEFAULT;  // k_sel_ne closes the iteration itself (last-block ticket) instead of a k_solve launch; hooks build: O3S_TAIL
  int first_group = 4;  // lanes per query in the first iteration of a call up to 200 k points (hooks build: O3S_FIRST_GROUP)
  int match_group = 4;
  bool match_group_forced = false;  // lanes per query in k_match2: 1, 2 or 4 (tuning knob O3S_GROUP; default by reading size)
  int nb_part_cap = kMaxPartialBlocks;  // blocks of the centroid / normal-equation kernels (tuning knob O3S_NB_PART)

  // profiling
  bool profiling = false;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  std::vector<hipEvent_t> prof_events;
  float kernel_ms[kNumKernels] = {0, 0, 0, 0, 0};
  int kernel_launches[kNumKernels] = {0, 0, 0, 0, 0};
};

namespace {

#define HIP_TRY(h, expr)                                                                      \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
      return O3S_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

int fail(o3s_icp* h, int code, const char* msg) {
  h->err = msg;
  return code;
}

inline int nblocks(int64_t n, int per = kern::kBlock) { return (int)((n + per - 1) / per); }
inline int round_up8(int v) { return (v + 7) & ~7; }

// ---- fp32 4x4 helpers on the host (frame algebra of LPM/ICP.cpp:373-374, 462-465) ----------------------------
#define HM4(m, r, c) (m)[(c)*4 + (r)]
void hmul4(const float* A, const float* B, float* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float s = HM4(A, r, 0) * HM4(B, 0, c);
      s = s + HM4(A, r, 1) * HM4(B, 1, c);
      s = s + HM4(A, r, 2) * HM4(B, 2, c);
      s = s + HM4(A, r, 3) * HM4(B, 3, c);
      HM4(C, r, c) = s;
    }
}
void hidentity(float* T) {
  for (int i = 0; i < 16; ++i) T[i] = 0.f;
  T[0] = T[5] = T[10] = T[15] = 1.f;
}
bool hrigid(const float* T) {  // RigidTransformation::checkParameters (LPM/TransformationsImpl.cpp:98-113)
  const float d0 = HM4(T, 0, 0) * (HM4(T, 1, 1) * HM4(T, 2, 2) - HM4(T, 1, 2) * HM4(T, 2, 1));
  const float d1 = HM4(T, 0, 1) * (HM4(T, 1, 0) * HM4(T, 2, 2) - HM4(T, 1, 2) * HM4(T, 2, 0));
  const float d2 = HM4(T, 0, 2) * (HM4(T, 1, 0) * HM4(T, 2, 1) - HM4(T, 1, 1) * HM4(T, 2, 0));
  const float det = d0 - d1 + d2;
  return !(std::fabs(1.f - det) > 0.001f);
}

ChainParams make_chain(const o3s_icp* h, bool reading_normals) {
  const o3s_icp_config& c = h->cfg;
  ChainParams cp{};
  cp.has_trim = c.trim_ratio >= 0.f;
  cp.trim_ratio = c.trim_ratio;
  cp.has_normal_gate = (c.max_normal_angle >= 0.f) && reading_normals && h->ref_has_normals;
  cp.cos_max_angle = std::cos(c.max_normal_angle);  // fp32 cos, as eps(cos(maxAngle)) at LPM/OutlierFiltersImpl.cpp:229
  cp.max_out_r2 = c.max_dist_outlier >= 0.f ? (float)std::pow((double)c.max_dist_outlier, 2)
                                            : std::numeric_limits<float>::infinity();
  cp.use_differential = c.use_differential;
  cp.min_diff_rot = c.min_diff_rot;
  cp.min_diff_trans = c.min_diff_trans;
  cp.smooth_length = c.smooth_length;
  cp.max_iters = c.max_iters;
  cp.counter_first = c.counter_first;
  cp.mirror = c.matcher == 1;
#ifdef O3S_TEST_HOOKS
  if (const char* e = O3S_HOOK_ENV("O3S_DBG")) cp.dbg = std::atoi(e);
#endif
  return cp;
}

int validate_config(const o3s_icp_config& c, std::string& why) {
  if (c.matcher != 0 && c.matcher != 1) return why = "matcher must be 0 (KDTreeMatcher) or 1 (MirrorMatcher)", O3S_ERR_BAD_CONFIG;
  if (!(c.max_dist > 0.f)) return why = "max_dist must be > 0", O3S_ERR_BAD_CONFIG;
  if (c.trim_ratio > 1.0f) return why = "trim_ratio must be <= 1", O3S_ERR_BAD_CONFIG;
  if (c.use_differential && (c.smooth_length < 0 || c.smooth_length > kMaxSmooth))
    return why = "smooth_length must be in [0, 15]", O3S_ERR_BAD_CONFIG;
  if (c.max_iters <= 0 && !c.use_differential) return why = "no transformation checker configured", O3S_ERR_BAD_CONFIG;
  if (c.grid_cell < 0.f) return why = "grid_cell must be >= 0", O3S_ERR_BAD_CONFIG;
  return O3S_OK;
}

// exclusive scan of n uint32 counts into out[n+1] on the handle's stream
// zero_in: the input array is cleared once read (it becomes the per-cell cursor array of the scatter that follows).
// post_nonzero: the number of non-zero inputs is posted to the mailbox; *seq_out is the sequence number to wait for.
int device_scan(o3s_icp* h, uint32_t* in, int64_t n, uint32_t* out, bool zero_in = false, bool post_nonzero = false, uint32_t* seq_out = nullptr) {
  const int64_t nb = (n + kern::kScanTile - 1) / kern::kScanTile;
  HIP_TRY(h, h->d_scan_sums.ensure((size_t)nb * 8));
  uint32_t* sums = h->d_scan_sums.as<uint32_t>();
  uint32_t* nonzero = post_nonzero ? sums + nb : nullptr;
  uint32_t seq = 0;
  if (post_nonzero) {
    if (++h->mb_seq == 0) ++h->mb_seq;
    seq = h->mb_seq;
    if (seq_out) *seq_out = seq;
  }
  hipLaunchKernelGGL(kern::k_scan_block_sums, dim3((unsigned)nb), dim3(kern::kBlock), 0, h->stream, in, n, sums, nonzero);
  hipLaunchKernelGGL(kern::k_scan_sums, dim3(1), dim3(1024), 0, h->stream, sums, nb, nonzero, h->mb_dev, seq);
  hipLaunchKernelGGL(kern::k_scan_apply, dim3((unsigned)nb), dim3(kern::kBlock), 0, h->stream, in, n, sums, out, zero_in ? in : nullptr);
  HIP_TRY(h, hipGetLastError());
  return O3S_OK;
}

// polls the mailbox until a kernel has posted `seq`: 1 = posted, 0 = the stream drained without it (not expected), < 0 = a
// HIP error.  The stream is queried every few thousand polls so that a fault upstream cannot leave the host spinning.
inline double now_us();
int mailbox_wait(o3s_icp* h, uint32_t seq) {
  // hipStreamQuery is not free for the GPU side (the runtime may put a marker packet into the queue for every call): it is
  // only the guard against a fault upstream, looked at every 200 us of waiting, never part of the polling itself
  double t_guard = now_us();
  for (;;) {
    for (int spin = 0; spin < 4096; ++spin)
      if (__atomic_load_n(h->mb + 1, __ATOMIC_ACQUIRE) == seq) return 1;
    const double t = now_us();
    if (t - t_guard < 200.0) continue;
    t_guard = t;
    const hipError_t q = hipStreamQuery(h->stream);
    if (q == hipSuccess) return __atomic_load_n(h->mb + 1, __ATOMIC_ACQUIRE) == seq ? 1 : 0;
    if (q != hipErrorNotReady) return -1;
  }
}

inline double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

int wait_post_impl(o3s_icp* h, uint32_t seq, hipEvent_t drained, bool* done);
int wait_post(o3s_icp* h, uint32_t seq, hipEvent_t drained, bool* done) {
  const double t0 = now_us();
  const int rc = wait_post_impl(h, seq, drained, done);
  h->host_wait_us += now_us() - t0;
  return rc;
}
// `drained` (nullable): an event recorded behind the last launch issued so far — how the host learns that everything issued has
// run WITHOUT ending the chain (kernels post once, when the chain is done).  Chains that are certain to end inside what was
// issued (a Counter checker and all of max_iters issued) pass none.  Neither the event nor the stream is queried more than every
// few microseconds: the polling itself is a load from host memory.
int wait_post_impl(o3s_icp* h, uint32_t seq, hipEvent_t drained, bool* done) {
  auto look = [&]() -> bool {  // the final post of THIS call
    const unsigned long long w = __atomic_load_n(&h->post->word, __ATOMIC_ACQUIRE);
    if ((uint32_t)(w >> 32) != seq || !(w & 1ull)) return false;
    *done = true;
    return true;
  };
  *done = false;
  double t_guard = now_us(), t_event = t_guard;
  for (;;) {
    for (int spin = 0; spin < 256; ++spin)
      if (look()) {
        h->wait_by_post += 1;
        return 1;
      }
    const double tn = now_us();
    if (drained && tn - t_event >= 4.0) {  // every 4 us at most: the query is a call into the runtime, not a load
      t_event = tn;
      const hipError_t q = hipEventQuery(drained);
      h->host_queries += 1;
      if (q == hipSuccess) {  // everything issued has run: either the final post is there by now, or the chain is not done yet
        (void)look();
        h->wait_by_event += 1;
        return 1;
      }
      if (q != hipErrorNotReady) return -1;
    }
    const double t = now_us();
    if (t - t_guard < 2000.0) continue;
    t_guard = t;  // a fault upstream must not leave the host spinning: the stream itself, every 2 ms
    const hipError_t q = hipStreamQuery(h->stream);
    h->host_queries += 1;
    if (q == hipSuccess) {
      (void)look();
      h->wait_by_guard += 1;
      return *done ? 1 : (drained ? 1 : 0);
    }
    if (q != hipErrorNotReady) return -1;
  }
}

// ---- initReference on device-resident input --------------------------------------------------------------------
// center = false: Matcher::init semantics (LPM/MatchersImpl.cpp:108-114) — the cloud is indexed as given; x - 0.0f is exact,
// so the index kernels run unchanged with a zero mean
// d_count (nullable): the number of points is a word on the device (o3s_icp_init_reference_dev_counted_async): M is then an upper
// bound that sizes the statistics launch, and the count arrives with the statistics — one hand-over for both; *M_out receives it
int init_reference_impl(o3s_icp* h, const float4* d_xyzw, const float* d_normals, int64_t M, bool wait_end = true, bool center = true,
                        const uint32_t* d_count = nullptr, int64_t* M_out = nullptr) {
  h->ref_ready = false;
  if (M_out) *M_out = 0;
  if (M <= 0) return fail(h, O3S_ERR_EMPTY_REFERENCE, "reference cloud is empty");
  if (M > (int64_t)0x7fffffff) return fail(h, O3S_ERR_BAD_ARGUMENT, "reference larger than 2^31-1 points");
  HIP_TRY(h, hipSetDevice(h->device));
  // 1. mean (fp64 accumulate, rounded once: rowwise().mean() at LPM/ICP.cpp:313) and bounds: per-block partials, folded
  //    by one block that posts the nine numbers to the mailbox
  const int G = std::min(1024, nblocks(M));
  HIP_TRY(h, h->d_ref_part.ensure((size_t)G * 3 * sizeof(double)));
  HIP_TRY(h, h->d_ref_bb.ensure((size_t)G * 6 * sizeof(float)));
  hipLaunchKernelGGL(kern::k_ref_stats, dim3(G), dim3(kern::kBlock), 0, h->stream, d_xyzw, M, d_count, h->d_ref_part.as<double>(),
                     h->d_ref_bb.as<float>());
  if (++h->mb_seq == 0) ++h->mb_seq;
  hipLaunchKernelGGL(kern::k_ref_stats_post, dim3(1), dim3(kern::kBlock), 0, h->stream, h->d_ref_part.as<double>(), h->d_ref_bb.as<float>(), G, M,
                     d_count, h->mb_dev, h->mb_seq);
  HIP_TRY(h, hipGetLastError());
  {
    const int w = mailbox_wait(h, h->mb_seq);
    if (w < 0) return fail(h, O3S_ERR_HIP, "init_reference: the statistics kernels failed");
    if (w == 0) return fail(h, O3S_ERR_HIP, "init_reference: the statistics were not posted");
  }
  if (d_count) {  // the count the device formed
    M = (int64_t)__atomic_load_n(h->mb + 11, __ATOMIC_RELAXED);
    if (M_out) *M_out = M;
    if (M <= 0) return fail(h, O3S_ERR_EMPTY_REFERENCE, "reference cloud is empty");
  }
  float lo[3], hi[3];
  for (int c = 0; c < 3; ++c) {
    const uint32_t um = __atomic_load_n(h->mb + 2 + c, __ATOMIC_RELAXED), ul = __atomic_load_n(h->mb + 5 + c, __ATOMIC_RELAXED),
                   uh = __atomic_load_n(h->mb + 8 + c, __ATOMIC_RELAXED);
    std::memcpy(&h->mean[c], &um, 4);
    std::memcpy(&lo[c], &ul, 4);
    std::memcpy(&hi[c], &uh, 4);
    if (!center) h->mean[c] = 0.f;
  }
  for (int c = 0; c < 3; ++c) {
    lo[c] = lo[c] - h->mean[c];  // x - mean is monotone in x, so the centred bounds are the bounds of the centred cloud
    hi[c] = hi[c] - h->mean[c];
    if (!(std::isfinite(lo[c]) && std::isfinite(hi[c]))) return fail(h, O3S_ERR_BAD_ARGUMENT, "reference contains non-finite coordinates");
  }
  // 2. grid geometry.  Start from maxDist/3 (the 3x3x3 block then covers maxDist/3 around any query and rings 2-3 the
  //    rest up to maxDist; measured on C2: k_match 18.9 us vs 23.0 us at maxDist/2, 28.2 us at 0.6 maxDist); if the map
  //    is much denser than that (mean points per occupied cell > 8) shrink the cell so that it holds ~4 points — the
  //    ring expansion keeps the search exact for any cell size.  A user-supplied grid_cell is taken as is.
  const float ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
  float cell = h->cfg.grid_cell;
  const bool adaptive = !(cell > 0.f);
  if (adaptive) {
    if (std::isfinite(h->cfg.max_dist)) {
      cell = h->cfg.max_dist * (1.0f / 3.0f);
    } else {
      const double vol = std::max((double)ext[0], 1e-3) * std::max((double)ext[1], 1e-3) * std::max((double)ext[2], 1e-3);
      cell = (float)(2.0 * std::cbrt(vol / (double)M));
    }
  }
  const float maxext = std::max(ext[0], std::max(ext[1], ext[2]));
  const float min_cell = std::max(maxext * 1e-6f, 1e-6f);
  cell = std::max(cell, min_cell);
  const double kMaxCells = (double)(1u << 28);
  float maxabs = 0.f;
  for (int c = 0; c < 3; ++c) maxabs = std::max(maxabs, std::max(std::fabs(lo[c]), std::fabs(hi[c])));
  HIP_TRY(h, h->d_cell_of.ensure((size_t)M * 4));
  HIP_TRY(h, h->d_ref.ensure((size_t)M * sizeof(float4)));
  HIP_TRY(h, h->d_refn.ensure((size_t)M * sizeof(float4)));
  HIP_TRY(h, h->d_orig_to_sorted.ensure((size_t)M * 4));
  const int gb = nblocks(M);
  GridParams g{};
  int64_t dims[3];
  // grid geometry for a cell edge (enlarged until the grid has at most 2^28 cells); returns the number of cells
  auto geometry = [&](float& edge, GridParams& gp, int64_t d[3]) -> size_t {
    for (;;) {
      double total = 1;
      for (int c = 0; c < 3; ++c) {
        d[c] = (int64_t)std::floor((double)ext[c] / (double)edge) + 1;
        total *= (double)d[c];
      }
      if (total <= kMaxCells) break;
      edge *= 1.26f;
    }
    gp.ox = lo[0];
    gp.oy = lo[1];
    gp.oz = lo[2];
    gp.cell = edge;
    gp.inv_cell = 1.0f / edge;
    gp.nx = (int)d[0];
    gp.ny = (int)d[1];
    gp.nz = (int)d[2];
    gp.margin = std::max(edge * 1e-3f, 16.f * maxabs * 1.1920929e-7f);
    gp.max_r2 = h->cfg.max_dist * h->cfg.max_dist;  // libnabo: maxRadius2 = maxRadius * maxRadius
    return (size_t)d[0] * (size_t)d[1] * (size_t)d[2];
  };
  for (int attempt = 0;; ++attempt) {
    h->ncells = geometry(cell, g, dims);
    // 3. counting sort of the reference into cell order: per-cell populations, scan (which also counts the occupied
    //    cells and clears the populations: they become the scatter's cursors), scatter.  The density probe is
    //    speculative: the sort is enqueued for this cell size right away and only repeated, with a smaller cell, when
    //    the occupied-cell count — posted by the scan, long there by the time it is looked at — says the map is dense.
    HIP_TRY(h, h->d_cell_tmp.ensure(h->ncells * 4));
    HIP_TRY(h, h->d_cell_start.ensure((h->ncells + 1 + 4) * 4));  // +4: headers are fetched as 4-word groups
    HIP_TRY(h, hipMemsetAsync(h->d_cell_tmp.p, 0, h->ncells * 4, h->stream));
    hipLaunchKernelGGL(kern::k_ref_assign, dim3(gb), dim3(kern::kBlock), 0, h->stream, d_xyzw, M, h->mean[0], h->mean[1], h->mean[2], g,
                       h->d_cell_of.as<uint32_t>(), h->d_cell_tmp.as<uint32_t>());
    HIP_TRY(h, hipGetLastError());
    const bool probe = adaptive && attempt < 2;
    uint32_t seq = 0;
    int rc = device_scan(h, h->d_cell_tmp.as<uint32_t>(), (int64_t)h->ncells, h->d_cell_start.as<uint32_t>(), /*zero_in=*/true, probe, &seq);
    if (rc != O3S_OK) return rc;
    hipLaunchKernelGGL(kern::k_ref_scatter, dim3(gb), dim3(kern::kBlock), 0, h->stream, d_xyzw, d_normals, M, h->mean[0], h->mean[1], h->mean[2],
                       h->d_cell_of.as<uint32_t>(), h->d_cell_start.as<uint32_t>(), h->d_cell_tmp.as<uint32_t>(), h->d_ref.as<float4>(),
                       h->d_refn.as<float4>(), h->d_orig_to_sorted.as<int32_t>());
    HIP_TRY(h, hipGetLastError());
    if (!probe) break;
    const int w = mailbox_wait(h, seq);
    if (w < 0) return fail(h, O3S_ERR_HIP, "init_reference: the index kernels failed");
    if (w == 0) return fail(h, O3S_ERR_HIP, "init_reference: the occupied-cell count was not posted");
    const uint32_t n_occ = __atomic_load_n(h->mb, __ATOMIC_RELAXED);
    const double rho = (double)M / (double)std::max(1u, n_occ);
    if (rho <= 8.0) break;
    const float next = std::max(min_cell, cell * (float)std::sqrt(4.0 / rho));  // points per cell ~ cell^2 on surfaces
    if (next > 0.9f * cell) break;
    cell = next;
  }
  h->grid = g;
  // 4. dense maps only: a second index of the same points for the FIRST iteration of a call.  The cell above was shrunk to ~4 points
  //    per occupied cell, which is what a converged iteration wants (few candidates; the incumbent bounds the search to a cell or two).
  //    A first iteration has no incumbents: under the initial guess the neighbour is several such cells away and the far search has
  //    to verify every row of the (y, z) disc of that radius — at C4 (cell 0.0445 m) 52 row windows per query, 0.93 ms.  How far away
  //    the neighbour is depends on the misalignment, not on the density, so the edge that suits this search is an absolute one:
  //    maxDist / 6 — the largest that still has the seed probe along the normal on (k_match2: reach >= 5 cells) — measured best or
  //    near best on maps of 0.02 and 0.03 m voxels (0.93 -> 0.64 .. 0.70 ms, 0.37 -> 0.30 ms), while every edge beyond maxDist / 5
  //    LOSES against the fine grid (tools/r05_first_grid.py, profiles/LAB_NOTES_r05.md 6).  Built when the main cell is at most 0.85
  //    of it.  Results do not depend on the grid (exact search, ties by original index).  Costs a second counting sort in
  //    init_reference and 32 bytes per reference point; maps at the nominal cell (maxDist / 3: C2, the per-scan loop) do not build it.
  h->have_grid1 = false;
  {
    const char* fg = O3S_HOOK_ENV("O3S_FIRST_GRID");  // hooks build: 0 = off, else the cell edge in metres
    const float nominal = std::isfinite(h->cfg.max_dist) ? h->cfg.max_dist * (1.0f / 3.0f) : 0.f;
    const float want = fg ? (float)std::atof(fg) : 0.5f * nominal;
    if (adaptive && nominal > 0.f && want > 0.f && g.cell <= 0.85f * want) {
      float cell1 = std::min(want, nominal);
      GridParams g1{};
      int64_t d1[3];
      const size_t nc1 = geometry(cell1, g1, d1);
      HIP_TRY(h, h->d_ref1.ensure((size_t)M * sizeof(float4)));
      if (d_normals) HIP_TRY(h, h->d_refn1.ensure((size_t)M * sizeof(float4)));
      HIP_TRY(h, h->d_cell_tmp.ensure(nc1 * 4));
      HIP_TRY(h, h->d_cell_start1.ensure((nc1 + 1 + 4) * 4));
      HIP_TRY(h, hipMemsetAsync(h->d_cell_tmp.p, 0, nc1 * 4, h->stream));
      hipLaunchKernelGGL(kern::k_ref_assign, dim3(gb), dim3(kern::kBlock), 0, h->stream, d_xyzw, M, h->mean[0], h->mean[1], h->mean[2], g1,
                         h->d_cell_of.as<uint32_t>(), h->d_cell_tmp.as<uint32_t>());
      HIP_TRY(h, hipGetLastError());
      const int rc1 = device_scan(h, h->d_cell_tmp.as<uint32_t>(), (int64_t)nc1, h->d_cell_start1.as<uint32_t>(), /*zero_in=*/true);
      if (rc1 != O3S_OK) return rc1;
      hipLaunchKernelGGL(kern::k_ref_scatter, dim3(gb), dim3(kern::kBlock), 0, h->stream, d_xyzw, d_normals, M, h->mean[0], h->mean[1], h->mean[2],
                         h->d_cell_of.as<uint32_t>(), h->d_cell_start1.as<uint32_t>(), h->d_cell_tmp.as<uint32_t>(), h->d_ref1.as<float4>(),
                         h->d_refn1.as<float4>(), (int32_t*)nullptr);
      HIP_TRY(h, hipGetLastError());
      h->grid1 = g1;
      h->have_grid1 = true;
    }
  }
  if (O3S_HOOK_ENV("O3S_PRINT_GRID"))  // hooks build: the cell edges the two indexes ended up with (tools/r05_first_grid.py)
    std::fprintf(stderr, "o3s grid: M %lld cell %.4f (%d x %d x %d) first-iteration cell %.4f\n", (long long)M, (double)g.cell, g.nx, g.ny, g.nz,
                 h->have_grid1 ? (double)h->grid1.cell : 0.0);
  {
    // the row-disc far search needs a finite bound to end; an unbounded maxDist (or one that reaches across more cells than an
    // int comfortably indexes) keeps the ring search, which expands until something is found.  O3S_FAR=0 forces it (A/B runs).
    const char* fe = O3S_HOOK_ENV("O3S_FAR");
    const double reach = std::isfinite(h->cfg.max_dist) ? (double)h->cfg.max_dist / (double)g.cell : 1e30;
    h->far_rows = reach <= (double)kern::kFarMaxCells && !(fe && std::atoi(fe) == 0);
  }
  // the reading is sorted on a coarsened grid of at most 2^22 bins
  {
    int qf = 1;
    while ((double)((dims[0] + qf - 1) / qf) * (double)((dims[1] + qf - 1) / qf) * (double)((dims[2] + qf - 1) / qf) > (double)(1 << 22)) ++qf;
    h->qf = qf;
    h->qnx = (int)((dims[0] + qf - 1) / qf);
    h->qny = (int)((dims[1] + qf - 1) / qf);
    h->qnz = (int)((dims[2] + qf - 1) / qf);
    h->qcells = (size_t)h->qnx * (size_t)h->qny * (size_t)h->qnz;
  }
  HIP_TRY(h, h->d_qstart.ensure((h->qcells + 1) * 4));
  // wait_end = false (o3s_icp_init_reference_dev_async): the scatter may still be reading the caller's arrays when this
  // returns; everything later on this handle is ordered behind it on the stream
  if (wait_end) HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->M = M;
  h->ref_has_normals = d_normals != nullptr;
  h->ref_ready = true;
  h->reading_ready = false;  // a resident reading was prepared against the previous grid
  if (h->graph_exec) {
    (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
  }
  return O3S_OK;
}

int ensure_iteration_buffers(o3s_icp* h, int N) {
  HIP_TRY(h, h->d_t.ensure((size_t)N * 6 * 4));
  HIP_TRY(h, h->d_r.ensure((size_t)N * 6 * 4));
  HIP_TRY(h, h->d_perm.ensure((size_t)N * 4));
  HIP_TRY(h, h->d_qcell.ensure((size_t)N * 4));
  HIP_TRY(h, h->d_pos.ensure((size_t)N * 4));
  HIP_TRY(h, h->d_d2.ensure((size_t)N * 4));
  HIP_TRY(h, h->d_mq.ensure((size_t)N * sizeof(float4)));
  HIP_TRY(h, h->d_mn.ensure((size_t)N * sizeof(float4)));
  HIP_TRY(h, h->d_hist.ensure(kHistWords * 4));
  HIP_TRY(h, h->d_cand.ensure((size_t)nblocks(N, kern::kClsBlock) * kern::kClsBlock * sizeof(CandRec)));  // one region per classify block
  HIP_TRY(h, h->d_cand_cnt.ensure(((size_t)nblocks(N, kern::kClsBlock) * 2 + 2) * 4));  // counts [nb] + bases [nb + 1]
  HIP_TRY(h, h->d_sel.ensure(sizeof(SelScratch)));
  HIP_TRY(h, h->d_sel_part2.ensure((size_t)kCentComps * kern::kSelPartMaxBlocks * sizeof(double)));
  HIP_TRY(h, h->d_park.ensure((size_t)kern::kParkRecs * (sizeof(CandRec) + 4)));
  HIP_TRY(h, h->d_cent.ensure((size_t)nblocks(N) * kCentComps * sizeof(double)));
  HIP_TRY(h, h->d_ne.ensure((size_t)kMaxPartialBlocks * kNeComps * sizeof(double)));
  HIP_TRY(h, h->d_state.ensure(sizeof(IcpState)));
  HIP_TRY(h, h->d_T0.ensure(16 * 4));
  return O3S_OK;
}

// the reading sort's count arrays: zeroed when (re)allocated or when the grid changes the split between bins and tiles; the
// kernels of prepare_reading leave them zeroed
int ensure_qcount(o3s_icp* h) {
  // whole tiles (the counts of a tile are kept transposed inside it: kern::qcount_slot), then the tile totals
  const size_t words = ((h->qcells + kern::kScanTile - 1) / kern::kScanTile) * kern::kScanTile + (size_t)kern::kMaxQTiles * kern::kTileReplicas;
  const size_t cap_before = h->d_qcount.cap;
  HIP_TRY(h, h->d_qcount.ensure(words * 4));
  if (h->d_qcount.cap != cap_before) HIP_TRY(h, hipMemsetAsync(h->d_qcount.p, 0, h->d_qcount.cap, h->stream));
  return O3S_OK;
}

int ensure_trace(o3s_icp* h, int cap) {
  cap = std::max(cap, 1);
  HIP_TRY(h, h->d_trace_T.ensure((size_t)cap * 16 * 4));
  HIP_TRY(h, h->d_trace_limit.ensure((size_t)cap * 4));
  HIP_TRY(h, h->d_trace_kept.ensure((size_t)cap * 8));
  h->trace_cap = cap;
  return O3S_OK;
}

struct ChainArgs {
  int N;
  int match_g;  // lanes per query of k_match2
  int nb_match, nb_part, nb_cls;
  int nb_fused;  // blocks of the fused selection + normal-equation kernel (0: the two kernels are launched separately)
  bool has_n;
  float *rx, *ry, *rz, *rnx, *rny, *rnz;
  ChainParams cp;
  GridParams g;
};

ChainArgs chain_args(o3s_icp* h, const ChainParams& cp) {
  ChainArgs a{};
  a.N = h->N;
  // k_match2: lanes per query.  O3S_GROUP forces 1 / 2 / 4; otherwise by reading size (kernels)
  //   measured (converged pose, us): C2 100k: G=4 10.6, G=2 8.9, G=1 9.4;  C4 500k: 37.2 / 27.8 / 33.1.  Two lanes halve the
  //   per-query set-up every lane of a group repeats; below ~32k queries four lanes are needed to fill 1024 SIMDs.  Between 32k and
  //   64k the two are equal on the synthetic pairs (50k: 30.6 k it/s either way) and four lanes win on ray-cast sweeps against a
  //   voxel map (47k queries, 17 candidates per query: registration stage 0.267 -> 0.239 ms); from 65k up two lanes win (-1..3 %).
  a.match_g = h->match_group_forced ? h->match_group : (h->N < 65536 ? 4 : 2);
  a.nb_match = round_up8(nblocks(h->N, kern::kBlock / a.match_g));  // one tile per block (steady state; the launch sizes its own grid)
  a.nb_cls = nblocks(h->N, kern::kClsBlock);
  a.nb_part = std::min(h->nb_part_cap, nblocks(h->N, kern::kBlock * kern::kNePPT));
  if (h->shard.active) {
    // sharded: the block partials of the normal equations are all-reduced AS THEY ARE ([27][blocks]), so the number of blocks must be
    // the same on every rank — derived from the whole reading and the world size, not from this rank's slice (slices differ by
    // one point: 2 * 512 * k + 1 points over two ranks gave 4 blocks here and 3 there, i.e. collectives of different lengths)
    const int64_t per_rank = (h->shard.n_total + h->shard.world - 1) / h->shard.world;
    a.nb_part = std::min(h->nb_part_cap, nblocks(per_rank, kern::kBlock * kern::kNePPT));
  }
  {  // fused selection + normal equations while the blocks fit one generation (O3S_FUSE=0 keeps the two kernels apart)
    const char* fe = O3S_HOOK_ENV("O3S_FUSE");  // read per call: the tests run both chains in one process
    const bool fuse = !(fe && std::atoi(fe) == 0);
    const int nbf = nblocks(h->N, kern::kBlock * kern::kNePPT);  // the blocks k_normal_eq would use: same partials, same bits
    a.nb_fused = (fuse && !h->shard.active && !h->many_in_flight && nbf <= kern::kFusedMaxBlocks && nbf == a.nb_part) ? nbf : 0;
  }
  a.has_n = h->read_has_normals;
  float* r = h->d_r.as<float>();
  a.rx = r;
  a.ry = r + (size_t)h->N;
  a.rz = r + 2 * (size_t)h->N;
  a.rnx = r + 3 * (size_t)h->N;
  a.rny = r + 4 * (size_t)h->N;
  a.rnz = r + 5 * (size_t)h->N;
  a.cp = cp;
  a.g = h->grid;
  return a;
}

// the level-1 replicas (+ level-2 histogram right behind the 16-replica area) the chain works on: in the sharded mode they live
// inside the exchange buffer, so that the all-reduces act on them in place
uint32_t* chain_hist(o3s_icp* h) {
  return h->shard.active ? reinterpret_cast<uint32_t*>(h->shard.xbuf + kXchgI32Off) : h->d_hist.as<uint32_t>();
}
// level-1 replicas the matcher spreads its flushes over: 16; in the sharded mode max(1, 16 / world) — they travel
int chain_replicas(const o3s_icp* h) { return h->shard.active ? shard_replicas(h->shard.world) : kHistReplicas; }

// RCB = candidates per round trip of the far search: 4 for the row-disc search (C2 first iteration 50.6 -> 43.2 us), 2 for the
// ring search (the kernel stays at <= 72 VGPRs; 8 was measured there in round 2 and bought nothing)
// From 200 k queries up k_match2 also fetches the matched normal (one more gather at the end of a launch with thousands of waves
// to hide it) and k_classify streams it instead of gathering it as an exposed round trip: C4 k_classify 17.9 -> 14.8 us,
// k_match2 54.3 -> 56.9 us, 9.41 -> 9.69 k it/s.  Below, the two cancel (C2: 26.7 k either way) and k_classify keeps the gather.
inline bool normals_from_matcher(const ChainArgs& a) { return a.N >= 200000 && !a.cp.mirror; }
// the index an iteration of the chain searches: the first-iteration index (init_reference_impl step 4) for iteration 0 of a chain when
// the map has one, else the main one.  Slots (d_pos) are positions in THAT index's order: the kernels of the same iteration that gather
// by slot (k_match2's own normal fetch, k_classify) get the same index; nothing carries a slot from one iteration to the next (the
// incumbent is the matched POINT, d_mq).  The module entry points (find_closests & co.) always use the main index.
struct RefIndex {
  const float4* ref;
  const float4* refn;
  const uint32_t* cell_start;
  GridParams g;
};
inline RefIndex main_index(o3s_icp* h) { return RefIndex{h->d_ref.as<float4>(), h->d_refn.as<float4>(), h->d_cell_start.as<uint32_t>(), h->grid}; }
inline RefIndex chain_index(o3s_icp* h, int it) {
  if (it == 0 && h->have_grid1 && h->far_rows)
    return RefIndex{h->d_ref1.as<float4>(), h->ref_has_normals ? h->d_refn1.as<float4>() : h->d_refn.as<float4>(), h->d_cell_start1.as<uint32_t>(), h->grid1};
  return main_index(h);
}
template <bool STATS, int G>
void launch_match2(o3s_icp* h, const ChainArgs& a, const ChainParams& cp, const RefIndex& ix, hipStream_t s) {
  const int nb = round_up8(nblocks(a.N, kern::kBlock / G));  // one tile of kBlock / G queries per block
  if (h->far_rows)
    hipLaunchKernelGGL((kern::k_match2<STATS, G, 2, 4, true>), dim3(nb), dim3(kern::kBlock), 0, s, a.rx, a.ry, a.rz, a.N, ix.ref, ix.cell_start, ix.g,
                       h->d_state.as<IcpState>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), h->d_mq.as<float4>(), chain_hist(h), ix.refn,
                       normals_from_matcher(a) ? h->d_mn.as<float4>() : (float4*)nullptr, a.has_n ? a.rnx : (const float*)nullptr, a.rny, a.rnz,
                       chain_replicas(h) - 1 O3S_DBG_ARG(cp.dbg));
  else
    hipLaunchKernelGGL((kern::k_match2<STATS, G, 2, 2, false>), dim3(nb), dim3(kern::kBlock), 0, s, a.rx, a.ry, a.rz, a.N, ix.ref, ix.cell_start, ix.g,
                       h->d_state.as<IcpState>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), h->d_mq.as<float4>(), chain_hist(h), ix.refn,
                       normals_from_matcher(a) ? h->d_mn.as<float4>() : (float4*)nullptr, a.has_n ? a.rnx : (const float*)nullptr, a.rny, a.rnz,
                       chain_replicas(h) - 1 O3S_DBG_ARG(cp.dbg));
}
// `first`: the first iteration of a call — no incumbents yet, half the queries go through the far search.  Up to 200 k points
// it runs with FOUR lanes per query whatever the steady-state choice: the far search is a chain of dependent round trips per lane,
// and twice the lanes halve the rows and candidates each has to walk (C2: 50 -> 39.5 us; at C4 the launch is candidate-bound and
// gains nothing).  Results do not depend on the lanes per query (exact search, integer histogram).
void launch_match2_any(o3s_icp* h, const ChainArgs& a, const ChainParams& cp, bool stats, bool first, const RefIndex& ix, hipStream_t s) {
  if (cp.mirror) {
    hipLaunchKernelGGL(kern::k_match_mirror, dim3(nblocks(a.N)), dim3(kern::kBlock), 0, s, a.N, h->d_ref.as<float4>(), h->d_orig_to_sorted.as<int32_t>(),
                       h->d_perm.as<int32_t>(), h->d_state.as<IcpState>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), h->d_mq.as<float4>(),
                       chain_hist(h));
    return;
  }
  const int G = (first && h->far_rows && !h->match_group_forced && a.N < 200000) ? h->first_group : a.match_g;
  if (stats) {
    if (G == 1) launch_match2<true, 1>(h, a, cp, ix, s);
    else if (G == 2) launch_match2<true, 2>(h, a, cp, ix, s);
    else launch_match2<true, 4>(h, a, cp, ix, s);
  } else {
    if (G == 1) launch_match2<false, 1>(h, a, cp, ix, s);
    else if (G == 2) launch_match2<false, 2>(h, a, cp, ix, s);
    else launch_match2<false, 4>(h, a, cp, ix, s);
  }
}
void launch_match_any(o3s_icp* h, const ChainArgs& a, const ChainParams& cp, bool stats, hipStream_t s, bool first = false) {
  launch_match2_any(h, a, cp, stats, first, main_index(h), s);
}
// the matcher launch of iteration `it` of a chain, on the index chain_index() picks for it
void launch_match_chain(o3s_icp* h, const ChainArgs& a, bool stats, hipStream_t s, int it, const RefIndex& ix) {
  launch_match2_any(h, a, a.cp, stats, it == 0, ix, s);
}

void launch_iteration(o3s_icp* h, const ChainArgs& a, bool stats, hipEvent_t* ev /*6 events or null*/, int it) {
  IcpState* st = h->d_state.as<IcpState>();
  hipStream_t s = h->stream;
  const int mode = kern::kModeCentroid | kern::kModeGate | (normals_from_matcher(a) ? kern::kModeNormalReady : 0);
  if (ev) (void)hipEventRecord(ev[0], s);
  const RefIndex ix = chain_index(h, it);
  launch_match_chain(h, a, stats, s, it, ix);
  if (ev) (void)hipEventRecord(ev[1], s);
  hipLaunchKernelGGL(kern::k_classify, dim3(a.nb_cls), dim3(kern::kClsBlock), 0, s, a.rx, a.ry, a.rz, a.rnx, a.rny, a.rnz, a.N, ix.ref,
                     ix.refn, h->d_pos.as<int32_t>(), h->d_d2.as<float>(), h->d_hist.as<uint32_t>(), a.cp, st,
                     h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(), h->d_cand_cnt.as<uint32_t>(), h->d_hist.as<uint32_t>() + (size_t)kHistReplicas * kHistBins, h->d_mq.as<float4>(), h->d_mn.as<float4>(), h->d_cent.as<double>(), mode, kHistReplicas);
  if (ev) (void)hipEventRecord(ev[2], s);
  uint32_t* hist2 = h->d_hist.as<uint32_t>() + (size_t)kHistReplicas * kHistBins;
  if (a.nb_fused > 0) {  // selection + normal equations in one launch (kern::k_sel_ne); timed under "sel_finish"
#define O3S_SEL_NE_ARGS                                                                                                                            \
  dim3(a.nb_fused), dim3(kern::kFinThreads), kern::kSelCap * 4, s, a.cp, st, h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(),                   \
      h->d_cand_cnt.as<uint32_t>(), hist2, h->d_cand_cnt.as<uint32_t>() + a.nb_cls, h->d_cent.as<double>(), a.nb_cls, mode, a.rx, a.ry, a.rz, a.N, \
      h->d_mq.as<float4>(), h->d_mn.as<float4>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), h->d_ne.as<double>(), h->d_hist.as<uint32_t>(),     \
      h->d_trace_T.as<float>(), h->d_trace_limit.as<float>(), h->d_trace_kept.as<int64_t>(), h->trace_cap, h->post_dev
#ifdef O3S_TEST_HOOKS  // the fused-tail instantiation exists in the hooks build only (round 4's measured-and-left-off experiment, O3S_TAIL)
    if (h->fuse_tail) hipLaunchKernelGGL(kern::k_sel_ne<true>, O3S_SEL_NE_ARGS);
    else
#endif
      hipLaunchKernelGGL(kern::k_sel_ne<false>, O3S_SEL_NE_ARGS);
#undef O3S_SEL_NE_ARGS
    if (ev) (void)hipEventRecord(ev[3], s);
    if (ev) (void)hipEventRecord(ev[4], s);
    // fuse_tail: the closing step (solve, checkers, post) is the tail of the block that stored its partials last — no k_solve
    if (!h->fuse_tail)
      hipLaunchKernelGGL(kern::k_solve, dim3(1), dim3(kern::kBlock), 0, s, h->d_ne.as<double>(), a.nb_fused, a.N, a.cp, st, h->d_trace_T.as<float>(),
                         h->d_trace_limit.as<float>(), h->d_trace_kept.as<int64_t>(), h->trace_cap, 1, h->post_dev);
    if (ev) (void)hipEventRecord(ev[5], s);
    return;
  }
  {
    // large readings (more classify blocks than the finishing block has threads): the candidate sweep runs on many blocks first
    const char* pe = O3S_HOOK_ENV("O3S_SEL_PARTIAL");  // read per call (A/B runs, tests of both paths): 0 keeps the single-block sweep
    const bool partial = a.nb_cls > kern::kFinThreads && !(pe && std::atoi(pe) == 0);
    const int nbp = partial ? std::min(kern::kSelPartMaxBlocks, nblocks(a.nb_cls, 4)) : 0;
    CandRec* park_rec = h->d_park.as<CandRec>();
    uint32_t* park_key = reinterpret_cast<uint32_t*>(h->d_park.as<char>() + (size_t)kern::kParkRecs * sizeof(CandRec));
    if (partial)
      hipLaunchKernelGGL(kern::k_sel_partial, dim3(nbp), dim3(kern::kFinThreads), 0, s, st, h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(),
                         h->d_cand_cnt.as<uint32_t>(), hist2, a.nb_cls, mode, h->d_sel_part2.as<double>(), park_rec, park_key);
    hipLaunchKernelGGL(kern::k_sel_finish, dim3(1), dim3(kern::kFinThreads), kern::kSelCap * 4, s, (uint32_t*)nullptr, a.cp, st,
                       h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(), h->d_cand_cnt.as<uint32_t>(), hist2,
                       h->d_cand_cnt.as<uint32_t>() + a.nb_cls, h->d_cent.as<double>(), a.nb_cls, mode,
                       partial ? h->d_sel_part2.as<double>() : (const double*)nullptr, nbp, park_rec, park_key);
  }
  if (ev) (void)hipEventRecord(ev[3], s);
  hipLaunchKernelGGL(kern::k_normal_eq, dim3(a.nb_part), dim3(kern::kBlock), 0, s, a.rx, a.ry, a.rz, a.N, h->d_mq.as<float4>(),
                     h->d_mn.as<float4>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), a.cp, st, h->d_ne.as<double>(), h->d_hist.as<uint32_t>());
  if (ev) (void)hipEventRecord(ev[4], s);
  hipLaunchKernelGGL(kern::k_solve, dim3(1), dim3(kern::kBlock), 0, s, h->d_ne.as<double>(), a.nb_part, a.N, a.cp, st, h->d_trace_T.as<float>(),
                     h->d_trace_limit.as<float>(), h->d_trace_kept.as<int64_t>(), h->trace_cap, 1, h->post_dev);
  if (ev) (void)hipEventRecord(ev[5], s);
}

// One iteration of the one-pair-sharded mode: four kernels with the three global quantities formed by three all-reduces of
// regions of the exchange buffer between them (csrc/icp_shard_kernels.h).  Everything is enqueued on the handle's
// stream; the callback enqueues the collective on (or ordered after) the same stream.
int launch_iteration_sharded(o3s_icp* h, const ChainArgs& a, bool stats, int it) {
  IcpState* st = h->d_state.as<IcpState>();
  hipStream_t s = h->stream;
  // no centroid partials from k_classify here: the raw moments of k_shard_moments carry the sums of p and q themselves
  const int mode = kern::kModeGate | (normals_from_matcher(a) ? kern::kModeNormalReady : 0);
  uint8_t* xb = h->shard.xbuf;
  double* xm = reinterpret_cast<double*>(xb + kXchgMOff);
  uint32_t* l1 = reinterpret_cast<uint32_t*>(xb + kXchgI32Off);
  uint32_t* l2 = l1 + kXchgL1Words;
  auto exchange = [&](int64_t byte_off, int64_t count, int32_t dtype) -> int {
    const int rc = h->shard.fn(h->shard.user, xb + byte_off, byte_off, count, dtype, (void*)s);
    if (rc != 0) {
      h->err = "shard exchange callback failed (rc " + std::to_string(rc) + ")";
      return O3S_ERR_HIP;
    }
    return O3S_OK;
  };
  int rc;
  const int R = chain_replicas(h);
  const RefIndex ix = chain_index(h, it);  // (every rank holds the same reference, so every rank picks the same index)
  launch_match_chain(h, a, stats, s, it, ix);  // level-1 replicas = region I of the exchange buffer (chain_hist), R of them
  if ((rc = exchange(kXchgI32Off, (int64_t)R * kHistBins, O3S_XCHG_INT32)) != O3S_OK) return rc;
  hipLaunchKernelGGL(kern::k_classify, dim3(a.nb_cls), dim3(kern::kClsBlock), 0, s, a.rx, a.ry, a.rz, a.rnx, a.rny, a.rnz, a.N, ix.ref,
                     ix.refn, h->d_pos.as<int32_t>(), h->d_d2.as<float>(), l1, a.cp, st, h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(),
                     h->d_cand_cnt.as<uint32_t>(), l2, h->d_mq.as<float4>(), h->d_mn.as<float4>(), h->d_cent.as<double>(), mode, R, 20 - kShardL2Bits,
                     (uint32_t)(kShardL2Bins - 1));
  if (a.cp.has_trim && (rc = exchange(kXchgI32Off + (int64_t)kXchgL1Words * 4, kShardL2Bins, O3S_XCHG_INT32)) != O3S_OK) return rc;
  hipLaunchKernelGGL(kern::k_shard_moments, dim3(a.nb_part + 1), dim3(kern::kBlock), 0, s, a.cp, st, h->d_sel.as<SelScratch>(), l2, a.rx, a.ry, a.rz, a.N,
                     h->d_mq.as<float4>(), h->d_mn.as<float4>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), xm, a.nb_part, l1, R,
                     h->d_cand.as<CandRec>(), h->d_cand_cnt.as<uint32_t>(), a.nb_cls, h->d_cand_cnt.as<uint32_t>() + a.nb_cls);
  if ((rc = exchange(kXchgMOff, shard_moment_doubles(a.nb_part), O3S_XCHG_FLOAT64)) != O3S_OK) return rc;
  hipLaunchKernelGGL(kern::k_solve_shard, dim3(1), dim3(kern::kBlock), 0, s, a.cp, st, h->d_sel.as<SelScratch>(), l2, xm, a.nb_part,
                     (int)std::min<int64_t>(h->shard.n_total, 0x7fffffff), h->d_trace_T.as<float>(), h->d_trace_limit.as<float>(),
                     h->d_trace_kept.as<int64_t>(), h->trace_cap, h->post_dev);
  return O3S_OK;
}

void init_state(IcpState& st) {
  std::memset(&st, 0, sizeof(st));
  hidentity(st.T_iter);
  st.limit = std::numeric_limits<float>::infinity();
}

bool graph_key_equal(const o3s_icp::GraphKey& a, const o3s_icp::GraphKey& b) {
  return a.N == b.N && a.iters == b.iters && a.nb == b.nb && a.has_n == b.has_n && a.gen == b.gen && std::memcmp(a.ptrs, b.ptrs, sizeof(a.ptrs)) == 0 &&
         std::memcmp(&a.cp, &b.cp, sizeof(ChainParams)) == 0 && std::memcmp(&a.g, &b.g, sizeof(GridParams)) == 0;
}

// transform + spatial sort of the reading; with reset_chain the first kernel also resets what a chain starts from
// (histograms, selection hand-off, incumbents, state — kern::PrepInit): no separate fill / copy commands
int prepare_reading(o3s_icp* h, const float* T0, bool sort, bool reset_chain, bool seed_differential) {
  const int N = h->N;
  const float4* in = reinterpret_cast<const float4*>(h->ext_xyzw ? h->ext_xyzw : h->d_in_xyzw.p);
  const float* in_n = h->read_has_normals ? reinterpret_cast<const float*>(h->ext_n ? h->ext_n : h->d_in_n.p) : nullptr;
  kern::Mat16 T0v;
  std::memcpy(T0v.v, T0, 16 * sizeof(float));
  kern::PrepInit init{};
  if (reset_chain) {
    init.hist = chain_hist(h);
    init.hist_words = h->shard.active ? (int)(kXchgL1Words + kShardL2Bins) : (int)kHistWords;  // level-1 replicas + the level-2 histogram behind them
    init.sel = h->d_sel.as<uint32_t>();
    init.sel_words = (int)(sizeof(SelScratch) / 4);
    init.mq = h->d_mq.as<float4>();
    init.state = h->d_state.as<IcpState>();
    init.seed_differential = seed_differential ? 1 : 0;
    init.seq = h->call_seq;
  }
  float* t = h->d_t.as<float>();
  float* r = h->d_r.as<float>();
  const size_t n = (size_t)N;
  const int nb = nblocks(N);
  if (sort) {
    // 4 launches: transform + bin counts + arrival ranks | bin starts | who sits where | stable placement.  The count arrays are
    // all zeros between calls (zeroed when allocated, cleared by the kernels that read them): no per-call memset.
    const int n_tiles = (int)((h->qcells + kern::kScanTile - 1) / kern::kScanTile);
    int rc = ensure_qcount(h);
    if (rc != O3S_OK) return rc;
    uint32_t* counts = h->d_qcount.as<uint32_t>();
    uint32_t* tile_cnt = counts + (size_t)n_tiles * kern::kScanTile;
    uint32_t* ticket = h->d_d2.as<uint32_t>();  // free until the first matcher launch, like d_pos below
    hipLaunchKernelGGL(kern::k_read_prep, dim3(nb), dim3(kern::kBlock), 0, h->stream, in, in_n, N, T0v, h->grid, t, t + n,
                       t + 2 * n, t + 3 * n, t + 4 * n, t + 5 * n, h->d_qcell.as<uint32_t>(), counts, h->qf, h->qnx,
                       h->qny, init, ticket, tile_cnt, (int32_t*)nullptr);
    hipLaunchKernelGGL(kern::k_read_starts, dim3(n_tiles), dim3(kern::kBlock), 0, h->stream, counts, (int64_t)h->qcells, tile_cnt,
                       h->d_qstart.as<uint32_t>());
    // stable counting sort: every point is placed by the rank of its input index inside its bin.
    // d_pos is free until the first matcher launch and holds the slot -> index table in between.
    const char* so = O3S_HOOK_ENV("O3S_SCATTER_ORDER");  // test hook: reversed arrival order, same placed reading
    int32_t* who = h->d_pos.as<int32_t>();
    hipLaunchKernelGGL(kern::k_read_scatter, dim3(nb), dim3(kern::kBlock), 0, h->stream, N, h->d_qcell.as<uint32_t>(), h->d_qstart.as<uint32_t>(),
                       ticket, who, tile_cnt, n_tiles, (so && std::atoi(so) == 1) ? 1 : 0);
    hipLaunchKernelGGL(kern::k_read_place, dim3(nb), dim3(kern::kBlock), 0, h->stream, N, h->d_qcell.as<uint32_t>(), h->d_qstart.as<uint32_t>(), who, t,
                       t + n, t + 2 * n, t + 3 * n, t + 4 * n, t + 5 * n, h->read_has_normals ? 1 : 0, r, r + n, r + 2 * n, r + 3 * n, r + 4 * n,
                       r + 5 * n, h->d_perm.as<int32_t>());
  } else {
    hipLaunchKernelGGL(kern::k_read_prep, dim3(nb), dim3(kern::kBlock), 0, h->stream, in, in_n, N, T0v, h->grid, r, r + n,
                       r + 2 * n, r + 3 * n, r + 4 * n, r + 5 * n, (uint32_t*)nullptr, (uint32_t*)nullptr, 1, 1, 1, init, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, h->d_perm.as<int32_t>());
  }
  HIP_TRY(h, hipGetLastError());
  h->prepared_N = N;
  return O3S_OK;
}

int push_state(o3s_icp* h, const IcpState& st) {
  h->stage->state = st;
  HIP_TRY(h, hipMemcpyAsync(h->d_state.p, &h->stage->state, sizeof(IcpState), hipMemcpyHostToDevice, h->stream));
  return O3S_OK;
}
int pull_state(o3s_icp* h) {
  HIP_TRY(h, hipMemcpyAsync(&h->stage->state, h->d_state.p, sizeof(IcpState), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return O3S_OK;
}

// Enqueues one whole compute() on the handle's stream (prepare, iteration chain, read-back of the state into pinned
// memory).  Returns without waiting when the chain could be issued in one go (Counter checker present, no profiling):
// compute_finish() then waits and composes the result, so several handles can be in flight at once.
int compute_launch(o3s_icp* h, const float* T_init) {
  h->pend_valid = false;
  h->pend_graph_left = 0;
  h->pend_eager = false;
  if (!h->ref_ready) return fail(h, O3S_ERR_NOT_INITIALIZED, "compute before a successful init_reference");
  if (!h->reading_ready || h->N <= 0) return fail(h, O3S_ERR_EMPTY_READING, "the reading point cloud is empty");
  if (!h->ref_has_normals) return fail(h, O3S_ERR_BAD_SHAPE, "point-to-plane needs reference normals");
  if (h->cfg.matcher == 1 && (int64_t)h->N > h->M) return fail(h, O3S_ERR_BAD_SHAPE, "MirrorMatcher needs reading size <= reference size");
  HIP_TRY(h, hipSetDevice(h->device));
  const int N = h->N;
  int rc = ensure_iteration_buffers(h, N);
  if (rc != O3S_OK) return rc;
  const ChainParams cp = make_chain(h, h->read_has_normals);
  const int iters_cap = cp.max_iters > 0 ? cp.max_iters : 4096;
  rc = ensure_trace(h, std::min(iters_cap, 4096));
  if (rc != O3S_OK) return rc;

  // T_refMean_readMean = T_refIn_refMean^-1 * T_refIn_readIn  (LPM/ICP.cpp:373-374; reading mean forced to 0 at :364)
  float* Tc = h->pend_Tc;
  float* T0 = h->pend_T0;
  float TcInv[16];
  hidentity(Tc);
  hidentity(TcInv);
  for (int d = 0; d < 3; ++d) {
    HM4(Tc, d, 3) = h->mean[d];
    HM4(TcInv, d, 3) = -h->mean[d];
  }
  hmul4(TcInv, T_init, T0);
  if (!hrigid(T0)) return fail(h, O3S_ERR_NOT_RIGID, "RigidTransformation: rotation matrix is not orthogonal (initial guess)");

  if (++h->call_seq == 0) ++h->call_seq;  // what this call's posts carry (k_read_prep writes it into the state)
  h->pend_issued = 0;
  rc = prepare_reading(h, T0, h->cfg.sort_queries != 0 && h->cfg.matcher == 0 && !h->reading_presorted, /*reset_chain=*/true, cp.use_differential != 0);
  if (rc != O3S_OK) return rc;

  const ChainArgs a = chain_args(h, cp);
  const bool want_stats = h->cfg.match_stats != 0;
  h->pend_cp = cp;
  // looks at the chain's post after `upto` iterations have been issued: true when the chain is done
  auto chain_done_after = [&](int /*upto*/, bool* done) -> int {
    HIP_TRY(h, hipEventRecord(h->ev_end, h->stream));  // behind everything issued so far
    const int w = wait_post(h, h->call_seq, h->ev_end, done);
    if (w < 0) return fail(h, O3S_ERR_HIP, "compute: a kernel of the iteration chain failed");
    if (w == 0) return fail(h, O3S_ERR_HIP, "compute: the iteration chain ended without posting its state");
    return O3S_OK;
  };
  if (h->shard.active && h->cfg.matcher != 0) return fail(h, O3S_ERR_BAD_CONFIG, "the sharded mode supports KDTreeMatcher only");
  const bool shard_graph = h->shard.active && h->shard.capturable && h->cfg.use_graph && cp.max_iters > 0 && !h->profiling;
  if (h->shard.active && !shard_graph) {
    // every rank issues the same iterations: the state is bit-identical across ranks, so the chunked `done` test below
    // breaks out on the same iteration everywhere and the collectives stay matched
    // A chain that can only end at max_iters (no Differential checker) is issued in one go: no host round trip at all.
    // One that may stop by itself is looked at every kChunk iterations (the same chunk as the graph replay of the
    // unsharded chain); the flag every rank reads is bit-identical, so all ranks leave the loop together.
    constexpr int kChunk = 5;
    const bool may_stop_early = cp.use_differential != 0 || cp.max_iters <= 0;
    for (int it = 0; it < iters_cap; ++it) {
      rc = launch_iteration_sharded(h, a, want_stats, it);
      if (rc != O3S_OK) return rc;
      h->pend_issued = it + 1;
      if (may_stop_early && (it % kChunk) == kChunk - 1 && it + 1 < iters_cap) {
        bool done = false;
        rc = chain_done_after(it + 1, &done);
        if (rc != O3S_OK) return rc;
        if (done) break;
      }
    }
    HIP_TRY(h, hipGetLastError());
  } else if (h->profiling) {
    for (int k = 0; k < kNumKernels; ++k) {
      h->kernel_ms[k] = 0.f;
      h->kernel_launches[k] = 0;
    }
    const size_t need = (size_t)iters_cap * 6;
    while (h->prof_events.size() < need) {
      hipEvent_t e;
      HIP_TRY(h, hipEventCreate(&e));
      h->prof_events.push_back(e);
    }
    int launched = 0;
    for (int it = 0; it < iters_cap; ++it) {
      launch_iteration(h, a, want_stats, &h->prof_events[(size_t)it * 6], it);
      ++launched;
      h->pend_issued = launched;
      if (cp.max_iters <= 0 && (it % 16) == 15) {
        bool done = false;
        rc = chain_done_after(launched, &done);
        if (rc != O3S_OK) return rc;
        if (done) break;
      }
    }
    HIP_TRY(h, hipGetLastError());
    rc = pull_state(h);  // the events have to be complete before they are read: this path waits for the stream
    if (rc != O3S_OK) return rc;
    const int ran = std::min(launched, h->stage->state.iter + (h->stage->state.status ? 1 : 0));
    for (int it = 0; it < ran; ++it)
      for (int k = 0; k < kNumKernels; ++k) {
        if ((k == 3 || (k == 4 && h->fuse_tail)) && a.nb_fused > 0) continue;  // fused chain: k_sel_ne (timed as k = 2) holds the selection and the normal equations (and, with fuse_tail, the closing step)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->prof_events[(size_t)it * 6 + k], h->prof_events[(size_t)it * 6 + k + 1]) == hipSuccess) {
          h->kernel_ms[k] += ms;
          h->kernel_launches[k] += 1;
        }
      }
  } else {
    // Replay policy.  A hipGraph of the whole chain (5 launches x max_iters) pays off when the same shapes come back:
    // it is captured the SECOND time a key is seen in a row and replayed from then on.  A call with new shapes (live
    // scans change size every time) is issued eagerly in chunks; between chunks the host looks at the `done` flag, so
    // a chain that converges after a few iterations does not pay for the rest of max_iters.
    // With a Differential checker the chain usually stops long before max_iters, and a graph of all max_iters iterations
    // would still issue 5 no-op launches (~1.8 us each) for every iteration after convergence.  The graph therefore holds
    // kGraphChunk iterations and is replayed until the `done` flag comes back set (compute_finish); the first replay's
    // read-back is the one every call needs anyway, so a chain that converges inside the first chunk pays nothing for it.
    // Fixed-length chains (no Differential checker: the bench configurations) keep one graph of max_iters iterations.
    constexpr int kGraphChunk = 5;
    const int chunk = (cp.use_differential && cp.max_iters > kGraphChunk + 2) ? kGraphChunk : cp.max_iters;
    o3s_icp::GraphKey key;
    key.N = N;
    key.iters = chunk;
    key.nb = a.nb_fused > 0 ? -a.nb_fused : a.nb_part;  // the fused and the two-kernel chain are different graphs
    key.has_n = a.has_n ? 1 : 0;
    key.gen = h->alloc_gen;  // every ensure() of this call has already run (ensure_iteration_buffers / ensure_trace / prepare)
    key.ptrs[0] = h->d_r.p;
    key.ptrs[1] = h->d_pos.p;
    key.ptrs[2] = h->d_d2.p;
    key.ptrs[3] = h->d_ref.p;
    key.ptrs[4] = h->have_grid1 ? h->d_cell_start1.p : h->d_cell_start.p;  // (any re-allocation moves key.gen as well; this tells the two kinds of chain apart)
    key.ptrs[5] = h->d_trace_T.p;
    key.ptrs[6] = (const void*)(uintptr_t)((want_stats ? 1 : 0) | (h->shard.active ? 2 : 0) | (h->fuse_tail ? 4 : 0) | ((uintptr_t)(h->shard.active ? h->shard.world : 0) << 8));
    key.ptrs[7] = h->d_perm.p;
    key.cp = cp;
    key.g = h->grid;
    const bool graph_ok = h->cfg.use_graph && cp.max_iters > 0 && (!h->shard.active || shard_graph);
    const bool have = graph_ok && h->graph_exec && graph_key_equal(key, h->graph_key);
    const bool seen_before = graph_ok && h->graph_candidate_valid && graph_key_equal(key, h->graph_candidate);
    bool capture_failed = false;
    if (graph_ok && !have && seen_before) {
      if (h->graph_exec) {
        (void)hipGraphExecDestroy(h->graph_exec);
        h->graph_exec = nullptr;
      }
      // Capture window: whatever fails inside it, the stream (possibly the caller's, o3s_icp_set_stream) must leave
      // capture mode again and the partial graph must go; the call then falls back to the eager chunked path below.
      hipGraph_t graph = nullptr;
      hipError_t ge = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
      if (ge == hipSuccess) {
        int src = O3S_OK;
        for (int it = 0; it < chunk && src == O3S_OK; ++it) {
          // sharded + capturable exchange (ncclAllReduce on this stream): the four collectives of an iteration are graph nodes too
          if (h->shard.active) src = launch_iteration_sharded(h, a, want_stats, it);
          else launch_iteration(h, a, want_stats, nullptr, it);
        }
        hipError_t le = hipGetLastError();
        if (le == hipSuccess && src != O3S_OK) le = hipErrorUnknown;
        ge = hipStreamEndCapture(h->stream, &graph);  // always: ends the capture even after a failed launch
        if (ge == hipSuccess && le != hipSuccess) ge = le;
      }
      if (ge == hipSuccess) ge = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (ge != hipSuccess) {
        (void)hipGetLastError();  // clear the sticky error: the eager path below reports its own
        h->graph_exec = nullptr;
        capture_failed = true;
      } else {
        h->graph_key = key;
        h->issue_mode = 1;
      }
    }
    h->graph_candidate = key;
    h->graph_candidate_valid = !capture_failed;
    h->pend_graph_left = 0;
    if (graph_ok && h->graph_exec && graph_key_equal(key, h->graph_key)) {
      HIP_TRY(h, hipGraphLaunch(h->graph_exec, h->stream));
      if (h->issue_mode != 1) h->issue_mode = 2;
      h->pend_graph_left = cp.max_iters - chunk;
      h->pend_graph_chunk = chunk;
      h->pend_issued = chunk;
      if (h->pend_graph_left > 0) HIP_TRY(h, hipEventRecord(h->ev_end, h->stream));  // the chain may go on beyond this chunk
    } else {
      // Where the host looks at the `done` flag: first after as many iterations as the LAST call on this handle needed (a
      // mapping loop's registrations take the same three or four iterations sweep after sweep, and every iteration issued
      // beyond the last one is four launches that return at once: ~20 us of launch time per sweep), then every second one.
      // The iterations themselves, and so the result, do not depend on where the host looks.
      // Sharded chains: every rank must issue the SAME iterations (each carries collectives its peers have to match), so the
      // schedule may depend on the input only — never on this handle's history (eager_hint) nor on whether THIS rank's graph
      // capture succeeded: the host looks exactly where the chunked graph replay would, after every `chunk` iterations.
      const int first_look = h->shard.active ? chunk : std::min(std::max(h->eager_hint, 2), 8);
      const int look_step = h->shard.active ? chunk : 2;
      int next_look = first_look;
      if (h->defer_looks && !h->shard.active) {  // the split entry points: the first look (and everything behind it) is compute_finish's
        const int n0 = std::min(first_look, iters_cap);
        for (int it = 0; it < n0; ++it) launch_iteration(h, a, want_stats, nullptr, it);
        h->pend_issued = n0;
        h->pend_eager = true;
        h->pend_iters_cap = iters_cap;
        h->pend_look_step = look_step;
        next_look = iters_cap + 1;  // (skips the loop below)
      }
      for (int it = h->pend_eager ? iters_cap : 0; it < iters_cap; ++it) {
        if (h->shard.active) {
          rc = launch_iteration_sharded(h, a, want_stats, it);
          if (rc != O3S_OK) return rc;
        } else {
          launch_iteration(h, a, want_stats, nullptr, it);
        }
        h->pend_issued = it + 1;
        if (it + 1 == next_look && it + 1 < iters_cap) {
          bool done = false;
          rc = chain_done_after(it + 1, &done);  // the closing kernel's post: no copy, no stream synchronisation
          if (rc != O3S_OK) return rc;
          if (done) break;
          next_look += look_step;
        }
      }
      HIP_TRY(h, hipGetLastError());
    }
  }
  h->pend_valid = true;
  return O3S_OK;
}

int compute_finish(o3s_icp* h, float* T_out, o3s_icp_stats* stats) {
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (!h->pend_valid) return fail(h, O3S_ERR_BAD_ARGUMENT, "compute_finish without a successful compute_launch");
  h->pend_valid = false;
  HIP_TRY(h, hipSetDevice(h->device));  // compute_batch finishes handles in turn: the current device is the last launch's
  bool done = false;
  for (;;) {
    // the chain is certain to end inside what was issued when a Counter checker is there and all of max_iters went out
    const bool certain = h->pend_cp.max_iters > 0 && h->pend_issued >= h->pend_cp.max_iters && h->pend_graph_left <= 0;
    if (!certain && h->pend_graph_left <= 0) HIP_TRY(h, hipEventRecord(h->ev_end, h->stream));
    const int w = wait_post(h, h->call_seq, certain ? (hipEvent_t) nullptr : h->ev_end, &done);
    if (w < 0) return fail(h, O3S_ERR_HIP, "compute: a kernel of the iteration chain failed");
    if (w == 0) return fail(h, O3S_ERR_HIP, "compute: the iteration chain ended without posting its state");
    if (done) break;
    if (h->pend_graph_left > 0) {
      HIP_TRY(h, hipGraphLaunch(h->graph_exec, h->stream));  // chunked graph replay: not converged yet
      h->pend_graph_left -= h->pend_graph_chunk;
      h->pend_issued += h->pend_graph_chunk;
      if (h->pend_graph_left > 0) HIP_TRY(h, hipEventRecord(h->ev_end, h->stream));
      continue;
    }
    if (h->pend_eager && h->pend_issued < h->pend_iters_cap) {  // a deferred eager chain that is not done yet: two more iterations, look again
      const ChainArgs a = chain_args(h, h->pend_cp);
      const int upto = std::min(h->pend_issued + h->pend_look_step, h->pend_iters_cap);
      for (int it = h->pend_issued; it < upto; ++it) launch_iteration(h, a, h->cfg.match_stats != 0, nullptr, it);
      HIP_TRY(h, hipGetLastError());
      h->pend_issued = upto;
      continue;
    }
    break;
  }
  h->pend_graph_left = 0;
  h->pend_eager = false;
  const ChainParams& cp = h->pend_cp;
  const float* Tc = h->pend_Tc;
  const float* T0 = h->pend_T0;
  if (!done) {  // every issued iteration ran and the chain is not done (no Counter checker and the cap reached): fetch the state the slow way
    const int rc = pull_state(h);
    if (rc != O3S_OK) return rc;
  } else {
    std::memcpy(&h->stage->state, &h->post->state, sizeof(IcpState));  // posted in front of the progress word (system-scope release)
  }
  const IcpState& st = h->stage->state;
  h->last_iters = std::min(st.iter, h->trace_cap);  // the trace stays on the device until o3s_icp_get_trace asks for it
  h->eager_hint = st.iter;
  if (stats) {
    stats->iterations = st.iter;
    stats->max_iters_reached = st.max_iters_reached;
    stats->kept_pairs = st.kept;
    stats->matched_pairs = st.n_finite;
    stats->point_used_ratio = st.point_used_ratio;
    stats->weighted_point_used_ratio = st.weighted_ratio;
    stats->last_trim_limit = cp.has_trim ? st.limit : std::numeric_limits<float>::quiet_NaN();
    // the chain's own clock: first matcher launch -> the launch that posted the final state (wall_clock64 stamps in the state)
    if (st.t_end > st.t_begin) stats->gpu_ms = (float)((double)(st.t_end - st.t_begin) / h->wall_clock_khz);
    stats->candidates_examined = (double)st.cand_count;
    stats->cells_probed = (double)st.row_count;
  }
  if (st.status != 0) {
    static const char* msgs[] = {"", "", "", "", "", "No matches available for computing distance quantiles",
                                 "ErrorMinimizer: no point to minimize", "abs rotation/translation norm not a number",
                                 "RigidTransformation: rotation matrix is not orthogonal"};
    h->err = (st.status >= 5 && st.status <= 8) ? msgs[st.status] : "device-side failure";
    return st.status;
  }
  if (!st.done && cp.max_iters <= 0) return fail(h, O3S_ERR_BAD_CONFIG, "iteration cap reached without a Counter checker");
  // icpCorrected_T_refIn_readIn = T_refIn_refMean * (T_iter * T_refMean_readMean)   (LPM/ICP.cpp:462-465)
  float tmp[16], out[16];
  hmul4(st.T_iter, T0, tmp);
  hmul4(Tc, tmp, out);
  std::memcpy(T_out, out, sizeof(out));
  return O3S_OK;
}

int compute_impl(o3s_icp* h, const float* T_init, float* T_out, o3s_icp_stats* stats) {
  if (stats) std::memset(stats, 0, sizeof(*stats));
  h->host_wait_us = 0.0;
  h->host_queries = 0;
  h->wait_by_post = h->wait_by_event = h->wait_by_guard = h->issue_mode = 0;
  const double t0 = now_us();
  const int rc = compute_launch(h, T_init);
  h->host_issue_us = now_us() - t0 - h->host_wait_us;
  if (rc != O3S_OK) return rc;
  return compute_finish(h, T_out, stats);
}

int upload_reading(o3s_icp* h, const float* xyzw, const float* normals, int64_t N) {
  h->reading_ready = false;
  h->ext_xyzw = h->ext_n = nullptr;
  if (N <= 0) {
    h->N = 0;
    return fail(h, O3S_ERR_EMPTY_READING, "the reading point cloud is empty");
  }
  if (N > (int64_t)(1 << 30)) return fail(h, O3S_ERR_BAD_ARGUMENT, "reading larger than 2^30 points");
  if (!xyzw) return fail(h, O3S_ERR_BAD_ARGUMENT, "xyzw is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, h->d_in_xyzw.ensure((size_t)N * 16));
  // compute() returns when the chain has POSTED its result, a moment before its last kernel retires: a copy from pageable memory
  // issued into a stream that is still busy takes the runtime's slow path (measured: 9.5 ms instead of 2.5 ms per call with host
  // buffers) — drain the stream first (a spin of a few microseconds), as the synchronising compute of round 3 did implicitly
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_in_xyzw.p, xyzw, (size_t)N * 16, hipMemcpyHostToDevice, h->stream));
  if (normals) {
    HIP_TRY(h, h->d_in_n.ensure((size_t)N * 12));
    HIP_TRY(h, hipMemcpyAsync(h->d_in_n.p, normals, (size_t)N * 12, hipMemcpyHostToDevice, h->stream));
  }
  h->N = (int)N;
  h->read_has_normals = normals != nullptr;
  h->reading_ready = true;
  h->reading_presorted = false;
  return O3S_OK;
}

}  // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

int o3s_abi_version(void) { return O3S_ABI_VERSION; }

#ifdef O3S_TS
// tuning builds only (-DO3S_TS): phase timestamps of block 0 of the small kernels, see O3S_TSTAMP in icp_kernels.h
int o3s_debug_ts(unsigned long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(kern::g_ts), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

void o3s_icp_default_config(o3s_icp_config* c) {
  if (!c) return;
  std::memset(c, 0, sizeof(*c));
  c->matcher = 0;
  c->max_dist = 0.5f;
  c->epsilon = 0.01f;
  c->trim_ratio = 0.90f;
  c->max_normal_angle = 1.57f;
  c->max_dist_outlier = -1.f;
  c->use_differential = 1;
  c->min_diff_rot = 0.001f;
  c->min_diff_trans = 0.01f;
  c->smooth_length = 3;
  c->max_iters = 15;
  c->counter_first = 0;
  c->grid_cell = 0.f;
  c->sort_queries = 1;
  c->use_graph = 1;
}

int o3s_icp_create(const o3s_icp_config* cfg, int device, o3s_icp** out) {
  if (!cfg || !out) {
    g_create_error = "NULL argument";
    return O3S_ERR_BAD_ARGUMENT;
  }
  *out = nullptr;
  std::string why;
  const int vc = validate_config(*cfg, why);
  if (vc != O3S_OK) {
    g_create_error = why;
    return vc;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
    g_create_error = "no usable HIP device (this library has no CPU fallback)";
    return O3S_ERR_HIP;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    g_create_error = "hipGetDeviceProperties failed";
    return O3S_ERR_HIP;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code objects only";
    return O3S_ERR_HIP;
  }
  o3s_icp* h = new o3s_icp();
  for (DevBuf* b : h->all_bufs()) b->gen = &h->alloc_gen;
  h->cfg = *cfg;
  h->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) {
    const char* px = O3S_HOOK_ENV("O3S_X_PRIO");  // experiment switch of the hooks build (cloud_dev.h make_stream): 2, 3 raise the mapping side's streams
    int least = 0, greatest = 0;
    if (px && (atoi(px) == 2 || atoi(px) == 3) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
      e = hipStreamCreateWithPriority(&h->own_stream, hipStreamNonBlocking, greatest);
    else
      e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
  }
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->stage, sizeof(HostStage), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->mb, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) {
    std::memset(h->mb, 0, 64);
    e = hipHostGetDevicePointer((void**)&h->mb_dev, h->mb, 0);
  }
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->post, sizeof(HostPost), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) {
    std::memset(h->post, 0, sizeof(HostPost));
    e = hipHostGetDevicePointer((void**)&h->post_dev, h->post, 0);
  }
  if (e == hipSuccess) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) h->wall_clock_khz = (double)khz;
  }
  if (e == hipSuccess) e = hipEventCreate(&h->ev_begin);
  if (e == hipSuccess) e = hipEventCreate(&h->ev_end);
  if (e == hipSuccess)
    e = hipFuncSetAttribute((const void*)kern::k_sel_finish, hipFuncAttributeMaxDynamicSharedMemorySize, kern::kSelCap * 4);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kern::k_sel_ne<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kern::kSelCap * 4);
#ifdef O3S_TEST_HOOKS
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kern::k_sel_ne<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kern::kSelCap * 4);
#endif
  if (e != hipSuccess) {
    g_create_error = std::string("HIP initialisation failed: ") + hipGetErrorString(e);
    o3s_icp_destroy(h);
    return O3S_ERR_HIP;
  }
  h->stream = h->own_stream;
  if (const char* e = O3S_HOOK_ENV("O3S_GROUP")) {
    const int g = std::atoi(e);
    h->match_group = (g == 2 || g == 1) ? g : 4;
    h->match_group_forced = true;
  }
  if (const char* e = O3S_HOOK_ENV("O3S_TAIL")) h->fuse_tail = std::atoi(e) != 0;
  if (const char* e = O3S_HOOK_ENV("O3S_FIRST_GROUP")) h->first_group = std::atoi(e) == 2 ? 2 : (std::atoi(e) == 1 ? 1 : 4);
  if (const char* e = O3S_HOOK_ENV("O3S_NB_PART")) h->nb_part_cap = std::max(1, std::min(kMaxPartialBlocks, std::atoi(e)));
  *out = h;
  return O3S_OK;
}

void o3s_icp_destroy(o3s_icp* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  for (DevBuf* b : h->all_bufs()) b->release();
  for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
  if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
  if (h->ev_end) (void)hipEventDestroy(h->ev_end);
  if (h->stage) (void)hipHostFree(h->stage);
  if (h->mb) (void)hipHostFree(h->mb);
  if (h->post) (void)hipHostFree(h->post);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
}

const char* o3s_last_error(const o3s_icp* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int o3s_icp_set_stream(o3s_icp* h, void* hip_stream) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  (void)hipStreamSynchronize(h->stream);
  h->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->own_stream;
  if (h->graph_exec) {
    (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
  }
  return O3S_OK;
}

namespace {
typedef float v4f __attribute__((ext_vector_type(4)));
// One 16-byte element per lane and no loop: 6.26 TB/s on the box for a 2 GiB copy (tools/native/copy_bench.hip), the
// figure MI355X_MICROARCH.md quotes for a float4 copy (6.29).  The grid-stride forms with 4-8 loads in flight per lane
// that this entry used in round 1 stop at 4.4-4.8 TB/s (as does hipMemcpyDtoD, 4.8): a block that walks eight strides
// apart keeps eight DRAM pages open per channel instead of one.
__global__ void __launch_bounds__(256) k_stream_copy(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
}  // namespace

int o3s_stream_copy_gbs(int device, int64_t bytes, int32_t reps, double* gbs) {
  if (!gbs || bytes < 16 || reps < 1) return O3S_ERR_BAD_ARGUMENT;
  *gbs = 0.0;
  if (hipSetDevice(device) != hipSuccess) return O3S_ERR_HIP;
  const size_t n = (size_t)bytes / 16;
  if (n > (size_t)0x7fffffff * 256) return O3S_ERR_BAD_ARGUMENT;
  void *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = O3S_ERR_HIP;
  if (hipMalloc(&a, n * 16) == hipSuccess && hipMalloc(&b, n * 16) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
      hipEventCreate(&e1) == hipSuccess && hipMemset(a, 1, n * 16) == hipSuccess && hipMemset(b, 0, n * 16) == hipSuccess) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, nullptr, (const v4f*)a, (v4f*)b, n);  // warm-up
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, nullptr, (const v4f*)a, (v4f*)b, n);
    (void)hipEventRecord(e1, nullptr);
    float ms = 0.f;
    if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) {
      *gbs = 2.0 * (double)(n * 16) * reps / (ms * 1e-3) / 1e9;
      rc = O3S_OK;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  return rc;
}

int o3s_icp_shard_configure(o3s_icp* h, int32_t rank, int32_t world, int64_t n_total, o3s_allreduce_fn fn, void* user, void* xbuf_dev) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  if (world <= 1 && !fn) {  // back to the single-GPU chain
    h->shard.active = false;
    h->shard.capturable = false;
    if (h->graph_exec) {
      (void)hipGraphExecDestroy(h->graph_exec);
      h->graph_exec = nullptr;
    }
    h->graph_candidate_valid = false;
    return O3S_OK;
  }
  if (!fn || world < 1 || rank < 0 || rank >= world || n_total <= 0) return fail(h, O3S_ERR_BAD_ARGUMENT, "shard_configure: bad rank / world / n_total / callback");
  HIP_TRY(h, hipSetDevice(h->device));
  if (xbuf_dev) {
    h->shard.xbuf = reinterpret_cast<uint8_t*>(xbuf_dev);
  } else {
    HIP_TRY(h, h->shard.own.ensure((size_t)kXchgBytes));
    h->shard.xbuf = h->shard.own.as<uint8_t>();
  }
  HIP_TRY(h, hipMemsetAsync(h->shard.xbuf, 0, (size_t)kXchgBytes, h->stream));
  h->shard.rank = rank;
  h->shard.world = world;
  h->shard.n_total = n_total;
  h->shard.fn = fn;
  h->shard.user = user;
  h->shard.active = true;
  h->shard.capturable = false;
  if (h->graph_exec) {  // a graph captured for another exchange must not be replayed
    (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
  }
  h->graph_candidate_valid = false;
  return O3S_OK;
}

int o3s_icp_shard_set_capturable(o3s_icp* h, int yes) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  if (!h->shard.active) return fail(h, O3S_ERR_BAD_ARGUMENT, "shard_set_capturable: configure the sharded mode first");
  h->shard.capturable = yes != 0;
  return O3S_OK;
}

int64_t o3s_icp_shard_exchange_bytes(void) { return (int64_t)kXchgBytes; }
int64_t o3s_icp_shard_bytes_per_iteration(int32_t world, int64_t n_total) {
  if (world < 1 || n_total < 1) return 0;
  const int64_t per_rank = (n_total + world - 1) / world;
  return shard_bytes_per_iteration(world, std::min(kMaxPartialBlocks, nblocks(per_rank, kern::kBlock * kern::kNePPT)));
}

int o3s_icp_init_reference(o3s_icp* h, const float* xyzw, const float* normals, int64_t M) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  if (M <= 0) {
    h->ref_ready = false;
    return fail(h, O3S_ERR_EMPTY_REFERENCE, "reference cloud is empty");
  }
  if (!xyzw) return fail(h, O3S_ERR_BAD_ARGUMENT, "xyzw is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, h->d_ref_in.ensure((size_t)M * 16));
  HIP_TRY(h, hipMemcpyAsync(h->d_ref_in.p, xyzw, (size_t)M * 16, hipMemcpyHostToDevice, h->stream));
  if (normals) {
    HIP_TRY(h, h->d_refn_in.ensure((size_t)M * 12));
    HIP_TRY(h, hipMemcpyAsync(h->d_refn_in.p, normals, (size_t)M * 12, hipMemcpyHostToDevice, h->stream));
  }
  return init_reference_impl(h, h->d_ref_in.as<float4>(), normals ? h->d_refn_in.as<float>() : nullptr, M);
}

int o3s_matcher_init(o3s_icp* h, const float* xyzw, const float* normals, int64_t M) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  if (M <= 0) {
    h->ref_ready = false;
    return fail(h, O3S_ERR_EMPTY_REFERENCE, "reference cloud is empty");
  }
  if (!xyzw) return fail(h, O3S_ERR_BAD_ARGUMENT, "xyzw is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, h->d_ref_in.ensure((size_t)M * 16));
  HIP_TRY(h, hipMemcpyAsync(h->d_ref_in.p, xyzw, (size_t)M * 16, hipMemcpyHostToDevice, h->stream));
  if (normals) {
    HIP_TRY(h, h->d_refn_in.ensure((size_t)M * 12));
    HIP_TRY(h, hipMemcpyAsync(h->d_refn_in.p, normals, (size_t)M * 12, hipMemcpyHostToDevice, h->stream));
  }
  return init_reference_impl(h, h->d_ref_in.as<float4>(), normals ? h->d_refn_in.as<float>() : nullptr, M, /*wait_end=*/true, /*center=*/false);
}

int o3s_icp_init_reference_dev(o3s_icp* h, const void* d_xyzw, const void* d_normals, int64_t M) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  if (M > 0 && !d_xyzw) return fail(h, O3S_ERR_BAD_ARGUMENT, "d_xyzw is NULL");
  return init_reference_impl(h, reinterpret_cast<const float4*>(d_xyzw), reinterpret_cast<const float*>(d_normals), M);
}

int o3s_icp_init_reference_dev_async(o3s_icp* h, const void* d_xyzw, const void* d_normals, int64_t M) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  if (M > 0 && !d_xyzw) return fail(h, O3S_ERR_BAD_ARGUMENT, "d_xyzw is NULL");
  return init_reference_impl(h, reinterpret_cast<const float4*>(d_xyzw), reinterpret_cast<const float*>(d_normals), M, /*wait_end=*/false);
}

// Internal to the library (the resident submap's o3s_submap_set_reference; declared in csrc/cloud_ops.hip, not part of the C ABI):
// the reference's size is a word on the device — the patch has just been compacted there — and comes back with the statistics'
// post instead of a hand-over of its own.  max_M sizes the first launches; *M_out receives the count (0: O3S_ERR_EMPTY_REFERENCE).
int o3s_icp_init_reference_dev_counted_async(o3s_icp* h, const void* d_xyzw, const void* d_normals, const uint32_t* d_count, int64_t max_M,
                                             int64_t* M_out) {
  if (!h || !d_count) return O3S_ERR_BAD_ARGUMENT;
  if (max_M > 0 && !d_xyzw) return fail(h, O3S_ERR_BAD_ARGUMENT, "d_xyzw is NULL");
  return init_reference_impl(h, reinterpret_cast<const float4*>(d_xyzw), reinterpret_cast<const float*>(d_normals), max_M, /*wait_end=*/false,
                             /*center=*/true, d_count, M_out);
}

int o3s_icp_synchronize(o3s_icp* h) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return O3S_OK;
}

int o3s_icp_wait_event(o3s_icp* h, void* hip_event) {
  if (!h || !hip_event) return O3S_ERR_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamWaitEvent(h->stream, reinterpret_cast<hipEvent_t>(hip_event), 0));
  return O3S_OK;
}

int o3s_icp_set_reading(o3s_icp* h, const float* xyzw, const float* normals, int64_t N) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  return upload_reading(h, xyzw, normals, N);
}

int o3s_icp_set_reading_dev(o3s_icp* h, const void* d_xyzw, const void* d_normals, int64_t N) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  h->reading_ready = false;
  if (N <= 0) {
    h->N = 0;
    return fail(h, O3S_ERR_EMPTY_READING, "the reading point cloud is empty");
  }
  if (N > (int64_t)(1 << 30)) return fail(h, O3S_ERR_BAD_ARGUMENT, "reading larger than 2^30 points");
  if (!d_xyzw) return fail(h, O3S_ERR_BAD_ARGUMENT, "d_xyzw is NULL");
  h->ext_xyzw = d_xyzw;
  h->ext_n = d_normals;
  h->N = (int)N;
  h->read_has_normals = d_normals != nullptr;
  h->reading_ready = true;
  h->reading_presorted = false;
  return O3S_OK;
}

int o3s_icp_reading_is_spatially_sorted(o3s_icp* h, int sorted) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  if (!h->reading_ready) return fail(h, O3S_ERR_EMPTY_READING, "reading_is_spatially_sorted: set a reading first");
  h->reading_presorted = sorted != 0;
  return O3S_OK;
}

int o3s_icp_compute_resident(o3s_icp* h, const float T_init[16], float T_out[16], o3s_icp_stats* stats) {
  if (!h || !T_init || !T_out) return O3S_ERR_BAD_ARGUMENT;
  return compute_impl(h, T_init, T_out, stats);
}

int o3s_icp_compute_resident_launch(o3s_icp* h, const float T_init[16]) {
  if (!h || !T_init) return O3S_ERR_BAD_ARGUMENT;
  h->host_wait_us = 0.0;
  h->host_queries = 0;
  h->wait_by_post = h->wait_by_event = h->wait_by_guard = h->issue_mode = 0;
  const double t0 = now_us();
  h->defer_looks = true;
  const int rc = compute_launch(h, T_init);
  h->defer_looks = false;
  h->host_issue_us = now_us() - t0 - h->host_wait_us;
  return rc;
}

int o3s_icp_compute_resident_finish(o3s_icp* h, float T_out[16], o3s_icp_stats* stats) {
  if (!h || !T_out) return O3S_ERR_BAD_ARGUMENT;
  return compute_finish(h, T_out, stats);
}

int o3s_icp_compute(o3s_icp* h, const float* xyzw, const float* normals, int64_t N, const float T_init[16], float T_out[16],
                    o3s_icp_stats* stats) {
  if (!h || !T_init || !T_out) return O3S_ERR_BAD_ARGUMENT;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (!h->ref_ready) return fail(h, O3S_ERR_NOT_INITIALIZED, "compute before a successful init_reference");
  const int rc = upload_reading(h, xyzw, normals, N);
  if (rc != O3S_OK) return rc;
  return compute_impl(h, T_init, T_out, stats);
}

int o3s_icp_compute_batch(o3s_icp* const* handles, int32_t n, const float* T_inits, float* T_outs, o3s_icp_stats* stats, int32_t* statuses) {
  if (!handles || n < 0 || !T_inits || !T_outs || !statuses) return O3S_ERR_BAD_ARGUMENT;
  for (int32_t k = 0; k < n; ++k)  // a handle holds ONE call in flight (pend_*, the pinned stage): the same handle twice is a caller error
    for (int32_t j = 0; j < k; ++j)
      if (handles[k] && handles[k] == handles[j]) return O3S_ERR_BAD_ARGUMENT;
  for (int32_t k = 0; k < n; ++k) {  // issue every chain first (one stream per handle: the chains overlap on the GPU) ...
    o3s_icp* h = handles[k];
    if (h) h->many_in_flight = n > 1;
    statuses[k] = h ? compute_launch(h, T_inits + 16 * (size_t)k) : (int32_t)O3S_ERR_BAD_ARGUMENT;
    if (h) h->many_in_flight = false;
  }
  for (int32_t k = 0; k < n; ++k) {  // ... then collect
    if (statuses[k] != O3S_OK) {
      if (stats) std::memset(&stats[k], 0, sizeof(o3s_icp_stats));
      continue;
    }
    statuses[k] = compute_finish(handles[k], T_outs + 16 * (size_t)k, stats ? &stats[k] : nullptr);
  }
  return O3S_OK;
}

int o3s_icp_get_trace(const o3s_icp* h, float* T_iters, float* limits, int64_t* kept, int32_t cap) {
  if (!h) return 0;
  const int n = std::min((int)cap, h->last_iters);
  if (n <= 0) return 0;
  if (hipSetDevice(h->device) != hipSuccess) return 0;
  if (T_iters && hipMemcpy(T_iters, h->d_trace_T.p, (size_t)n * 16 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (limits && hipMemcpy(limits, h->d_trace_limit.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (kept && hipMemcpy(kept, h->d_trace_kept.p, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
}

int64_t o3s_icp_get_reading_order(const o3s_icp* h, int32_t* order, int64_t cap) {
  if (!h || !order || cap <= 0 || h->prepared_N <= 0) return 0;
  const int64_t n = std::min<int64_t>(cap, h->prepared_N);
  if (hipSetDevice(h->device) != hipSuccess) return 0;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return 0;
  if (hipMemcpy(order, h->d_perm.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
}

int o3s_icp_reference_mean(const o3s_icp* h, float mean3[3]) {
  if (!h || !mean3) return O3S_ERR_BAD_ARGUMENT;
  if (!h->ref_ready) return O3S_ERR_NOT_INITIALIZED;
  for (int d = 0; d < 3; ++d) mean3[d] = h->mean[d];
  return O3S_OK;
}

int o3s_icp_host_split(const o3s_icp* h, double out4[4]) {
  if (!h || !out4) return O3S_ERR_BAD_ARGUMENT;
  out4[0] = h->host_issue_us;
  out4[1] = h->host_wait_us;
  out4[2] = (double)h->host_queries;
  out4[3] = h->stage->state.t_begin > h->stage->state.t_prep ? 1e3 * (double)(h->stage->state.t_begin - h->stage->state.t_prep) / h->wall_clock_khz : 0.0;
  return O3S_OK;
}

int o3s_icp_host_split_ex(const o3s_icp* h, double out8[8]) {
  if (!h || !out8) return O3S_ERR_BAD_ARGUMENT;
  const int rc = o3s_icp_host_split(h, out8);
  if (rc != O3S_OK) return rc;
  out8[4] = (double)h->wait_by_post;
  out8[5] = (double)h->wait_by_event;
  out8[6] = (double)h->wait_by_guard;
  out8[7] = (double)h->issue_mode;
  return O3S_OK;
}

int o3s_icp_set_profiling(o3s_icp* h, int on) {
  if (!h) return O3S_ERR_BAD_ARGUMENT;
  h->profiling = on != 0;
  return O3S_OK;
}

int o3s_icp_kernel_ms(const o3s_icp* h, float avg_ms[5], int32_t launches[5]) {
  if (!h || !avg_ms) return O3S_ERR_BAD_ARGUMENT;
  for (int k = 0; k < kNumKernels; ++k) {
    avg_ms[k] = h->kernel_launches[k] ? h->kernel_ms[k] / (float)h->kernel_launches[k] : 0.f;
    if (launches) launches[k] = h->kernel_launches[k];
  }
  return O3S_OK;
}

int o3s_icp_profile_match(o3s_icp* h, const float T_iter[16], int32_t reps, int32_t flags, float* avg_ms) {
  if (!h || !T_iter || !avg_ms || reps <= 0) return O3S_ERR_BAD_ARGUMENT;
  if (!h->ref_ready) return fail(h, O3S_ERR_NOT_INITIALIZED, "profile_match before a successful init_reference");
  if (!h->reading_ready || h->N <= 0) return fail(h, O3S_ERR_EMPTY_READING, "profile_match needs a resident reading (set_reading + compute)");
  HIP_TRY(h, hipSetDevice(h->device));
  int rc = ensure_iteration_buffers(h, h->N);
  if (rc != O3S_OK) return rc;
  ChainParams cp = make_chain(h, h->read_has_normals);
#ifndef O3S_TEST_HOOKS
  if (flags & 0xff) return fail(h, O3S_ERR_BAD_ARGUMENT, "profile_match: the kernel switches exist in the hooks build only (make hooks)");
#endif
  IcpState st0;
  init_state(st0);
  std::memcpy(st0.T_iter, T_iter, 16 * sizeof(float));
  rc = push_state(h, st0);
  if (rc != O3S_OK) return rc;
  const ChainArgs a = chain_args(h, cp);
#ifdef O3S_TEST_HOOKS
  cp.dbg = flags & 0xff;
#endif
  // 0x100: wipe the incumbents first (with flag 8 — no outputs — every launch then runs like the first iteration of a call)
  if (flags & 0x100) HIP_TRY(h, hipMemsetAsync(h->d_mq.p, 0, (size_t)h->N * sizeof(float4), h->stream));
  auto launch = [&]() { launch_match_any(h, a, cp, false, h->stream, /*first=*/(flags & 0x100) != 0); };  // 0x100: the launch a first iteration makes (lanes per query)
  for (int k = 0; k < 3; ++k) launch();  // warm-up
  HIP_TRY(h, hipEventRecord(h->ev_begin, h->stream));
  for (int k = 0; k < reps; ++k) launch();
  HIP_TRY(h, hipEventRecord(h->ev_end, h->stream));
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_begin, h->ev_end));
  *avg_ms = ms / (float)reps;
  HIP_TRY(h, hipMemsetAsync(h->d_hist.p, 0, kHistWords * 4, h->stream));  // the launches left counts behind
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return O3S_OK;
}

// ---- module-level path ----------------------------------------------------------------------------------------

int o3s_icp_find_closests(o3s_icp* h, const float* query_xyzw, int64_t N, int32_t* ids, float* dists2) {
  if (!h || !ids || !dists2) return O3S_ERR_BAD_ARGUMENT;
  if (!h->ref_ready) return fail(h, O3S_ERR_NOT_INITIALIZED, "find_closests before a successful init_reference");
  if (h->cfg.matcher == 1 && N > h->M) return fail(h, O3S_ERR_BAD_SHAPE, "MirrorMatcher needs reading size <= reference size");
  int rc = upload_reading(h, query_xyzw, nullptr, N);
  if (rc != O3S_OK) return rc;
  rc = ensure_iteration_buffers(h, (int)N);
  if (rc != O3S_OK) return rc;
  rc = ensure_trace(h, 1);
  if (rc != O3S_OK) return rc;
  float I[16];
  hidentity(I);
  rc = prepare_reading(h, I, h->cfg.sort_queries != 0 && h->cfg.matcher == 0, /*reset_chain=*/true, false);
  if (rc != O3S_OK) return rc;
  ChainParams cp = make_chain(h, false);
  const ChainArgs a = chain_args(h, cp);
  launch_match_any(h, a, a.cp, false, h->stream, /*first=*/true);  // no incumbents: the far search does the work
  HIP_TRY(h, h->d_mod_a.ensure((size_t)N * 4));
  HIP_TRY(h, h->d_mod_b.ensure((size_t)N * 4));
  hipLaunchKernelGGL(kern::k_export_matches, dim3(nblocks(N)), dim3(kern::kBlock), 0, h->stream, (int)N, h->d_pos.as<int32_t>(),
                     h->d_d2.as<float>(), h->d_ref.as<float4>(), h->d_perm.as<int32_t>(), h->d_mod_a.as<int32_t>(), h->d_mod_b.as<float>());
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(ids, h->d_mod_a.p, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(dists2, h->d_mod_b.p, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->reading_ready = false;  // the resident reading was replaced by the query
  return O3S_OK;
}

static int import_matches(o3s_icp* h, const int32_t* ids, const float* dists2, const float* weights, int64_t N) {
  HIP_TRY(h, h->d_mod_a.ensure((size_t)N * 4));
  HIP_TRY(h, h->d_mod_b.ensure((size_t)N * 4));
  HIP_TRY(h, hipMemcpyAsync(h->d_mod_a.p, ids, (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_mod_b.p, dists2, (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
  if (weights) {
    HIP_TRY(h, h->d_mod_c.ensure((size_t)N * 4));
    HIP_TRY(h, hipMemcpyAsync(h->d_mod_c.p, weights, (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
  }
  hipLaunchKernelGGL(kern::k_import_matches, dim3(nblocks(N)), dim3(kern::kBlock), 0, h->stream, (int)N, h->d_mod_a.as<int32_t>(),
                     h->d_mod_b.as<float>(), weights ? h->d_mod_c.as<float>() : (const float*)nullptr, h->d_orig_to_sorted.as<int32_t>(), h->M,
                     h->d_ref.as<float4>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), h->d_mq.as<float4>());
  HIP_TRY(h, hipGetLastError());
  return O3S_OK;
}

int o3s_icp_outlier_weights(o3s_icp* h, const float* reading_normals, const int32_t* ids, const float* dists2, int64_t N, float* weights) {
  if (!h || !ids || !dists2 || !weights) return O3S_ERR_BAD_ARGUMENT;
  if (!h->ref_ready) return fail(h, O3S_ERR_NOT_INITIALIZED, "outlier_weights before a successful init_reference");
  if (N <= 0) return fail(h, O3S_ERR_EMPTY_READING, "empty matches");
  HIP_TRY(h, hipSetDevice(h->device));
  int rc = ensure_iteration_buffers(h, (int)N);
  if (rc != O3S_OK) return rc;
  h->reading_ready = false;
  rc = import_matches(h, ids, dists2, nullptr, N);
  if (rc != O3S_OK) return rc;
  ChainParams cp = make_chain(h, reading_normals != nullptr);
  const bool any = h->cfg.trim_ratio >= 0.f || h->cfg.max_normal_angle >= 0.f || h->cfg.max_dist_outlier >= 0.f;
  IcpState st0;
  init_state(st0);
  rc = push_state(h, st0);
  if (rc != O3S_OK) return rc;
  HIP_TRY(h, hipMemsetAsync(h->d_hist.p, 0, kHistWords * 4, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->d_sel.p, 0, sizeof(SelScratch), h->stream));
  hipLaunchKernelGGL(kern::k_hist, dim3(nblocks(N)), dim3(kern::kBlock), 0, h->stream, h->d_d2.as<float>(), (int)N, h->d_hist.as<uint32_t>());
  {
    float* r = h->d_r.as<float>();
    const size_t n = (size_t)N;
    const int nbc = nblocks(N, kern::kClsBlock);
    hipLaunchKernelGGL(kern::k_classify, dim3(nbc), dim3(kern::kClsBlock), 0, h->stream, r, r + n, r + 2 * n, r + 3 * n, r + 4 * n, r + 5 * n,
                       (int)N, h->d_ref.as<float4>(), h->d_refn.as<float4>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(),
                       h->d_hist.as<uint32_t>(), cp, h->d_state.as<IcpState>(), h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(),
                       h->d_cand_cnt.as<uint32_t>(), h->d_hist.as<uint32_t>() + (size_t)kHistReplicas * kHistBins, h->d_mq.as<float4>(), h->d_mn.as<float4>(), h->d_cent.as<double>(), 0, kHistReplicas);
    hipLaunchKernelGGL(kern::k_sel_finish, dim3(1), dim3(kern::kFinThreads), kern::kSelCap * 4, h->stream, h->d_hist.as<uint32_t>(), cp,
                       h->d_state.as<IcpState>(), h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(), h->d_cand_cnt.as<uint32_t>(), h->d_hist.as<uint32_t>() + (size_t)kHistReplicas * kHistBins,
                       h->d_cand_cnt.as<uint32_t>() + nbc, h->d_cent.as<double>(), nbc, 0, (const double*)nullptr, 0, (const CandRec*)nullptr,
                       (const uint32_t*)nullptr);
  }
  const float* d_rn = nullptr;
  if (reading_normals) {
    HIP_TRY(h, h->d_in_n.ensure((size_t)N * 12));
    HIP_TRY(h, hipMemcpyAsync(h->d_in_n.p, reading_normals, (size_t)N * 12, hipMemcpyHostToDevice, h->stream));
    d_rn = h->d_in_n.as<float>();
  }
  HIP_TRY(h, h->d_mod_d.ensure((size_t)N * 4));
  hipLaunchKernelGGL(kern::k_weights, dim3(nblocks(N)), dim3(kern::kBlock), 0, h->stream, (int)N, h->d_pos.as<int32_t>(), h->d_d2.as<float>(), d_rn,
                     h->d_refn.as<float4>(), cp, h->d_state.as<IcpState>(), any ? 1 : 0, h->d_mod_d.as<float>());
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(weights, h->d_mod_d.p, (size_t)N * 4, hipMemcpyDeviceToHost, h->stream));
  rc = pull_state(h);
  if (rc != O3S_OK) return rc;
  if (h->stage->state.status != 0) return fail(h, h->stage->state.status, "No matches available for computing distance quantiles");
  return O3S_OK;
}

int o3s_icp_minimize(o3s_icp* h, const float* reading_xyzw, const int32_t* ids, const float* dists2, const float* weights, int64_t N,
                     float T_out[16], float A_out[36], float b_out[6], float x_out[6]) {
  if (!h || !reading_xyzw || !ids || !dists2 || !weights || !T_out) return O3S_ERR_BAD_ARGUMENT;
  if (!h->ref_ready) return fail(h, O3S_ERR_NOT_INITIALIZED, "minimize before a successful init_reference");
  if (!h->ref_has_normals) return fail(h, O3S_ERR_BAD_SHAPE, "point-to-plane needs reference normals");
  int rc = upload_reading(h, reading_xyzw, nullptr, N);
  if (rc != O3S_OK) return rc;
  h->reading_ready = false;
  rc = ensure_iteration_buffers(h, (int)N);
  if (rc != O3S_OK) return rc;
  rc = ensure_trace(h, 1);
  if (rc != O3S_OK) return rc;
  float* r = h->d_r.as<float>();
  hipLaunchKernelGGL(kern::k_aos_to_soa, dim3(nblocks(N)), dim3(kern::kBlock), 0, h->stream, h->d_in_xyzw.as<float4>(), (int)N, r, r + (size_t)N,
                     r + 2 * (size_t)N);
  rc = import_matches(h, ids, dists2, weights, N);
  if (rc != O3S_OK) return rc;
  ChainParams cp = make_chain(h, false);
  cp.max_out_r2 = std::numeric_limits<float>::infinity();  // the caller's weights already carry every filter
  cp.has_trim = 0;
  cp.has_normal_gate = 0;
  IcpState st0;
  init_state(st0);
  rc = push_state(h, st0);
  if (rc != O3S_OK) return rc;
  HIP_TRY(h, hipMemsetAsync(h->d_hist.p, 0, kHistWords * 4, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->d_sel.p, 0, sizeof(SelScratch), h->stream));
  const ChainArgs a = chain_args(h, cp);
  IcpState* st = h->d_state.as<IcpState>();
  hipLaunchKernelGGL(kern::k_classify, dim3(a.nb_cls), dim3(kern::kClsBlock), 0, h->stream, a.rx, a.ry, a.rz, a.rnx, a.rny, a.rnz, a.N,
                     h->d_ref.as<float4>(), h->d_refn.as<float4>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), h->d_hist.as<uint32_t>(), cp, st,
                     h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(), h->d_cand_cnt.as<uint32_t>(), h->d_hist.as<uint32_t>() + (size_t)kHistReplicas * kHistBins, h->d_mq.as<float4>(), h->d_mn.as<float4>(),
                     h->d_cent.as<double>(), kern::kModeCentroid, kHistReplicas);
  hipLaunchKernelGGL(kern::k_sel_finish, dim3(1), dim3(kern::kFinThreads), kern::kSelCap * 4, h->stream, h->d_hist.as<uint32_t>(), cp, st,
                     h->d_sel.as<SelScratch>(), h->d_cand.as<CandRec>(), h->d_cand_cnt.as<uint32_t>(), h->d_hist.as<uint32_t>() + (size_t)kHistReplicas * kHistBins,
                     h->d_cand_cnt.as<uint32_t>() + a.nb_cls, h->d_cent.as<double>(), a.nb_cls, kern::kModeCentroid, (const double*)nullptr, 0,
                     (const CandRec*)nullptr, (const uint32_t*)nullptr);
  hipLaunchKernelGGL(kern::k_normal_eq, dim3(a.nb_part), dim3(kern::kBlock), 0, h->stream, a.rx, a.ry, a.rz, a.N, h->d_mq.as<float4>(),
                     h->d_mn.as<float4>(), h->d_pos.as<int32_t>(), h->d_d2.as<float>(), cp, st, h->d_ne.as<double>(), (uint32_t*)nullptr);
  hipLaunchKernelGGL(kern::k_solve, dim3(1), dim3(kern::kBlock), 0, h->stream, h->d_ne.as<double>(), a.nb_part, a.N, cp, st,
                     h->d_trace_T.as<float>(), h->d_trace_limit.as<float>(), h->d_trace_kept.as<int64_t>(), h->trace_cap, 0, (HostPost*)nullptr);
  HIP_TRY(h, hipGetLastError());
  rc = pull_state(h);
  if (rc != O3S_OK) return rc;
  const IcpState& s = h->stage->state;
  if (s.status != 0) return fail(h, s.status, "ErrorMinimizer: no point to minimize");
  std::memcpy(T_out, s.dT, sizeof(s.dT));
  if (A_out) std::memcpy(A_out, s.A, sizeof(s.A));
  if (b_out) std::memcpy(b_out, s.b, sizeof(s.b));
  if (x_out) std::memcpy(x_out, s.x, sizeof(s.x));
  return O3S_OK;
}

}  // extern "C"
