// icp_types.h — device-visible plain structs shared by the kernels and the host side of libo3dslam_icp_hip.so.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// Test hooks and tuning knobs are compiled into the `hooks` build only (make hooks -> libo3dslam_icp_hip_hooks.so,
// -DO3S_TEST_HOOKS): the product library never reads the environment and its kernels carry no work-skipping switch.
//   O3S_HOOK_ENV(name)   getenv in the hooks build, NULL in the product
//   O3S_DBG(bits)        `dbg & bits` inside a kernel that received O3S_DBG_PARAM; constant false in the product
#ifdef O3S_TEST_HOOKS
#include <stdlib.h>
#define O3S_HOOK_ENV(name) getenv(name)
#define O3S_DBG_PARAM , int dbg
#define O3S_DBG_ARG(x) , (x)
#define O3S_DBG(bits) ((dbg & (bits)) != 0)
#define O3S_CP_DBG(cp, bits) (((cp).dbg & (bits)) != 0)
#else
#define O3S_HOOK_ENV(name) ((const char*)0)
#define O3S_DBG_PARAM
#define O3S_DBG_ARG(x)
#define O3S_DBG(bits) false
#define O3S_CP_DBG(cp, bits) false
#endif

namespace o3s {

constexpr int kHistBins = 2048;       // top 11 bits below the sign of a non-negative fp32 squared distance
constexpr int kHistReplicas = 16;     // level-1 histogram replicas (blockIdx % 16, two per XCD): bounds same-address atomics (measured: 4 -> 27.9 us, 8 -> 18.2 us, 16 -> 16.1 us, 32 -> 16.1 us for k_match)
constexpr int kMaxSmooth = 15;        // DifferentialTransformationChecker.smoothLength upper bound
constexpr int kHistRing = 16;         // quaternion / translation ring (smooth_length + 1 <= 16)
constexpr int kMaxPartialBlocks = 512;  // upper bound on blocks of the classify / normal-equation kernels
constexpr int kCentComps = 7;         // sum p(3), sum q(3), count
constexpr int kNeComps = 27;          // upper triangle of A (21) + b (6)
constexpr int kSegs = 4;              // candidate segments of the trim selection (block b appends to segment b % 4); more segments lengthen the slot arithmetic of k_sel_finish (16: +3.7 us), fewer did not slow k_classify

// Uniform grid over the mean-centred reference (the matcher index that replaces libnabo's kd-tree).
struct GridParams {
  float ox, oy, oz;   // grid origin in the <refMean> frame
  float cell;         // cell edge
  float inv_cell;
  int32_t nx, ny, nz;
  float margin;       // slack subtracted from every pruning bound (absorbs fp32 rounding of cell assignment)
  float max_r2;       // KDTreeMatcher.maxDist^2 (may be +inf)
};

// Parameters of the outlier chain and the checkers (kernel argument, by value).
struct ChainParams {
  int32_t has_trim;
  float trim_ratio;
  int32_t has_normal_gate;   // SurfaceNormalOutlierFilter present AND both clouds carry normals
  float cos_max_angle;       // eps = cos(maxAngle), evaluated in fp32 on the host
  float max_out_r2;          // MaxDistOutlierFilter limit (squared); +inf when absent
  int32_t use_differential;
  float min_diff_rot, min_diff_trans;
  int32_t smooth_length;
  int32_t max_iters;         // <= 0: no Counter checker
  int32_t counter_first;
  int32_t mirror;            // MirrorMatcher
#ifdef O3S_TEST_HOOKS
  int32_t dbg;               // hooks build only: timing experiments (env O3S_DBG, o3s_icp_profile_match flags); non-zero invalidates results
#endif
};

// Device-resident state of one compute() call.  One per handle; read back once at the end of the call.
// The first 32 words are the "header" every kernel of the chain needs: one coalesced 128-byte load + lane broadcasts.
struct IcpState {
  float T_iter[16];          // [0..15]  column-major; T_iter(i+1) = dT * T_iter(i)   (LPM/ICP.cpp:433-434)
  int32_t iter;              // [16] iterations completed
  int32_t done;              // [17] 1 => every later kernel of the chain returns immediately
  int32_t status;            // [18] o3s_status
  float limit;               // [19] trim limit (squared distance); +inf when no Trimmed filter
  uint32_t n_finite;         // [20] matches with finite distance
  float mp[3];               // [21..23] mean of the kept (transformed) reading points
  float mq[3];               // [24..26] mean of their reference points
  int32_t kept;              // [27] |K|
  int32_t max_iters_reached; // [28]
  int32_t counter;           // [29] CounterTransformationChecker::conditionVariables(0)
  int32_t hist_total;        // [30] DifferentialTransformationChecker: rotations.size()
  int32_t solve_branch;      // [31] 0 LLT, 1 min-norm QR, 2 fp64 fallback
  float quat_ring[kHistRing][4];   // x y z w
  float trans_ring[kHistRing][3];
  float point_used_ratio, weighted_ratio;
  float A[36];               // last normal equations (column-major), b, x — exposed by the module-level API
  float b[6];
  float x[6];
  float dT[16];              // last step
  float ang_ring[kHistRing];      // DifferentialTransformationChecker: angularDistance(entry i, entry i - 1) of ring entry i, computed once
  float tnorm_ring[kHistRing];    // ... and |t_i - t_(i-1)|: the smoothing window re-reads them instead of re-deriving three atan2 per iteration
  uint32_t call_seq;              // sequence number of the compute() this state belongs to (k_read_prep); echoed in every post
  uint32_t posted;                // 1: the final state of this call has been posted to the host (later launches of the chain are silent)
  unsigned long long t_begin;     // wall_clock64 stamps: first matcher launch of the call, and the launch that posted the final state
  unsigned long long t_end;
  unsigned long long t_prep;      // ... and the start of the call's first kernel (k_read_prep)
  unsigned long long cand_count;  // matcher statistics (sum over the call)
  unsigned long long row_count;
};
constexpr int kStateTailWords = 4;  // cand_count / row_count: only ever touched by the matcher's atomics, never by a state write-back
static_assert(sizeof(IcpState) % 4 == 0, "IcpState is copied word-wise");

// One in-bin candidate of the trim selection: everything the finishing kernel needs, so it never chases an index.
struct CandRec {
  float px, py, pz;   // transformed reading point
  uint32_t bits;      // fp32 bit pattern of its squared match distance
  float qx, qy, qz;   // matched (mean-centred) reference point
  int32_t keep;       // bit 0: every non-Trimmed weight of the chain is 1 (normal gate, MaxDist); bits 1..31: the query's slot index
                      // (the sharded chain fetches the matched normal of the few pairs its last block takes)
};

// Hand-off between the two selection kernels (device-resident).
struct SelScratch {
  uint32_t seg_count[kSegs];  // candidates appended per segment (zeroed by k_sel_finish for the next iteration)
  uint32_t bin;               // level-1 bin holding rank k
  uint32_t kk;                // rank inside that bin
  uint32_t bin_count;
  uint32_t skip;              // 1: nothing to select (no Trimmed filter, or no finite match)
  uint32_t ne_ticket;         // k_sel_ne: blocks that have stored their 27 partial sums (the last one closes the iteration)
  uint32_t pad[3];            // sharded chain: [0] the level-2 digit, [1] the rank inside it (k_shard_moments -> k_solve_shard)
};

// What the host polls instead of copying the state back (host-coherent pinned memory, one per handle): the kernel that closes
// an iteration stores the progress word — sequence number of the call, iterations completed, done — with system-scope release;
// when the chain is done it has stored the whole IcpState in front of it.
struct HostPost {
  IcpState state;
  unsigned long long word;  // (call_seq << 32) | (iterations completed << 8) | done
};
__host__ __device__ inline unsigned long long post_word(uint32_t seq, int iter, int done) {
  return ((unsigned long long)seq << 32) | ((unsigned long long)(uint32_t)iter << 8) | (done ? 1ull : 0ull);
}

}  // namespace o3s
