// icp_types.h — device-visible plain structs shared by the kernels and the host side of libo3dslam_icp_hip.so.
#pragma once
#include <stdint.h>

namespace o3s {

constexpr int kHistBins = 2048;       // top 11 bits below the sign of a non-negative fp32 squared distance
constexpr int kMaxSmooth = 15;        // DifferentialTransformationChecker.smoothLength upper bound
constexpr int kHistRing = 16;         // quaternion / translation ring (smooth_length + 1 <= 16)
constexpr int kMaxPartialBlocks = 512;  // upper bound on blocks of the centroid / normal-equation kernels
constexpr int kCentComps = 7;         // sum p(3), sum q(3), count
constexpr int kNeComps = 27;          // upper triangle of A (21) + b (6)

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
  int32_t dbg;               // timing experiments only (env O3S_DBG); any non-zero value invalidates results
};

// Device-resident state of one compute() call.  One per handle; read back once at the end of the call.
struct IcpState {
  float T_iter[16];          // column-major; T_iter(i+1) = dT * T_iter(i)   (LPM/ICP.cpp:433-434)
  int32_t iter;              // iterations completed
  int32_t done;              // 1 => every later kernel of the chain returns immediately
  int32_t status;            // o3s_status
  int32_t max_iters_reached;
  int32_t counter;           // CounterTransformationChecker::conditionVariables(0)
  int32_t hist_total;        // DifferentialTransformationChecker: rotations.size()
  float quat_ring[kHistRing][4];   // x y z w
  float trans_ring[kHistRing][3];
  // per-iteration scalars
  float limit;               // trim limit (squared distance); +inf when no Trimmed filter
  uint32_t n_finite;         // matches with finite distance
  int64_t kept;              // |K|
  float mp[3], mq[3];        // means of kept reading / reference points
  float point_used_ratio, weighted_ratio;
  int32_t solve_branch;      // 0 LLT, 1 min-norm QR, 2 fp64 fallback
  int32_t pad0;
  unsigned long long cand_count;  // matcher statistics (sum over the call)
  unsigned long long row_count;
  float A[36];               // last normal equations (column-major), b, x — exposed by the module-level API
  float b[6];
  float x[6];
  float dT[16];              // last step
};

}  // namespace o3s
