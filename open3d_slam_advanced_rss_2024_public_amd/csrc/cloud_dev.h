// cloud_dev.h — device side of the open3d_slam point-cloud operators, shared by cloud_ops.hip (host-pointer C ABI,
// include/o3s_cloud_ops.h) and submap.hip (device-resident submap, include/o3s_submap.h).  gfx950 only.
//
// Domain kernels are hand-written; rocPRIM (the native AMD primitive library shipped with ROCm) supplies the stable
// radix sort and the scans that order voxels.  fp64 throughout, no FMA contraction (-ffp-contract=off), so every
// decision (voxel index, inside/outside) and every per-voxel mean is bit-identical to the reference's sequential loops.
#pragma once
#include "../../include/o3s_cloud_ops.h"
#include "../../include/o3s_icp.h"
#include "icp_types.h"  // O3S_HOOK_ENV

#include <string.h>
#include <time.h>

#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstdint>
#include <limits>
#include <atomic>
#include <mutex>
#include <vector>

#pragma clang fp contract(off)

namespace {  // internal linkage
namespace o3s_cloud {


constexpr int kB = 256;
inline unsigned nblk(int64_t n) { return (unsigned)((n + kB - 1) / kB); }

// Global extrema live in kExtSlots replicas (block b uses replica b % kExtSlots) that the host folds after the read-back:
// together with the wave reduction and the look-before-atomic below this keeps same-address atomics off the critical path.
constexpr int kExtSlots = 64;

// Wave-wide min / max before touching a global extremum: one atomic per wave instead of one per point (same-address
// atomics serialise at ~11 ns each — per-point atomics made the bound kernels the largest item of the mapping loop).
__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const unsigned long long o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const unsigned long long o = __shfl_xor(v, m, 64);
    v = o > v ? o : v;
  }
  return v;
}

struct Buf {  // grow-only: a second alloc() that fits re-uses the memory (contents are not kept when it has to grow)
  void* p = nullptr;
  size_t cap = 0;
  hipError_t alloc(size_t bytes) {
    if (p && bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e == hipSuccess) cap = bytes ? bytes : 16;
    return e;
  }
  ~Buf() {
    if (p) (void)hipFree(p);
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

#define CK(expr)                               \
  do {                                         \
    if ((expr) != hipSuccess) return O3S_ERR_HIP; \
  } while (0)

int pick_device(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return O3S_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return O3S_ERR_HIP;
  return O3S_OK;
}

// Streams by role: `side` = the receiving side (staging a raw sweep, pre-processing a scan), everything else is the mapping
// thread's critical path.  Experiment switch (hooks build): O3S_X_PRIO=1 side streams at the lowest priority, 2: main streams at the highest
// as well, 3: only the main streams raised.
inline hipError_t make_stream(hipStream_t* s, bool side) {
  const char* e = O3S_HOOK_ENV("O3S_X_PRIO");
  const int mode = e ? atoi(e) : 0;
  int least = 0, greatest = 0;
  if (mode && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
    if (side && (mode == 1 || mode == 2)) return hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
    if (!side && (mode == 2 || mode == 3)) return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
  }
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// ---- kernels ---------------------------------------------------------------------------------------------------
// getVoxelIdx(p, InverseVoxelSize): int(std::floor(p * inv))  (VoxelHashMap.hpp:48-51)
__global__ void __launch_bounds__(kB) k_voxel_idx(const double* __restrict__ pts, int64_t n3, double inv, int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i < n3) idx[i] = (int32_t)floor(pts[i] * inv);
}

// EigenVec3iHash (VoxelHashMap.hpp:25-35)
__global__ void __launch_bounds__(kB) k_voxel_hash(const int32_t* __restrict__ idx, int64_t N, uint64_t* __restrict__ hash) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const uint64_t sl = 17191ull, sl2 = sl * sl;
  const uint64_t v = (uint64_t)(int64_t)idx[3 * i] + (uint64_t)(int64_t)idx[3 * i + 1] * sl + (uint64_t)(int64_t)idx[3 * i + 2] * sl2;
  hash[i] = (uint64_t)(uint32_t)v;
}

// open3dToPointmatcher (open3d_conversions.cpp:57-118)
__global__ void __launch_bounds__(kB) k_o3d_to_pm(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t N,
                                                  float4* __restrict__ xyzw, float* __restrict__ out_n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  xyzw[i] = make_float4((float)pts[3 * i], (float)pts[3 * i + 1], (float)pts[3 * i + 2], 1.0f);
  if (nrm) {
    out_n[3 * i] = (float)nrm[3 * i];
    out_n[3 * i + 1] = (float)nrm[3 * i + 1];
    out_n[3 * i + 2] = (float)nrm[3 * i + 2];
  }
}

// CroppingVolume::isWithinVolume (croppers.cpp:57-59, 121-167); (p - c).norm() = sqrt((dx^2 + dy^2) + dz^2)
__device__ __forceinline__ bool within(const o3s_cropper& c, double x, double y, double z) {
  const double dx = x - c.centre[0], dy = y - c.centre[1], dz = z - c.centre[2];
  bool in;
  switch (c.kind) {
    case 1: in = sqrt(dx * dx + dy * dy + dz * dz) <= c.p0; break;
    case 2: in = sqrt(dx * dx + dy * dy + dz * dz) >= c.p0; break;
    case 3: {
      const double d = sqrt(dx * dx + dy * dy + dz * dz);
      in = d <= c.p1 && d >= c.p0;
      break;
    }
    case 4: in = z >= c.p1 && z <= c.p2 && sqrt(dx * dx + dy * dy) <= c.p0; break;
    default: in = true;
  }
  return c.invert ? !in : in;
}

// flag[i] = 1 if the point is KEPT IN PLACE (crop: inside; voxelise: outside = pass-through)
// Flag producers (k_mask, k_mask_cnt, k_heads) can leave the number of set flags of every block beside the flags (blk_cnt[blockIdx.x]):
// the exclusive scan of the flags is then ONE launch in which every block adds up the counts of the blocks before it
// (k_scan_flags_blk) instead of rocPRIM's two (look-back state + scan) — five scans per sweep of the per-scan loop.
__device__ __forceinline__ void put_block_count(uint32_t f, uint32_t* __restrict__ blk_cnt) {  // reached by every thread of the block
  if (blk_cnt == nullptr) return;                                                              // (uniform)
  const int c = __syncthreads_count(f != 0u);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = (uint32_t)c;
}
__global__ void __launch_bounds__(kB) k_mask(o3s_cropper c, const double* __restrict__ pts, int64_t N, int keep_inside,
                                             uint32_t* __restrict__ flag, uint32_t* __restrict__ zero16 = nullptr,
                                             uint32_t* __restrict__ blk_cnt = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (zero16 && blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0u;  // the pipeline's status / count words
  uint32_t f = 0u;
  if (i < N) {
    const bool in = within(c, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    f = (in == (keep_inside != 0)) ? 1u : 0u;
    flag[i] = f;
  }
  put_block_count(f, blk_cnt);
}

__global__ void __launch_bounds__(kB) k_compact(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t N,
                                                const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off,
                                                double* __restrict__ out_pts, double* __restrict__ out_n, int32_t* __restrict__ out_idx) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N || !flag[i]) return;
  const uint32_t o = off[i];
  for (int a = 0; a < 3; ++a) {
    out_pts[3 * (int64_t)o + a] = pts[3 * i + a];
    if (nrm) out_n[3 * (int64_t)o + a] = nrm[3 * i + a];
    if (out_idx) out_idx[3 * (int64_t)o + a] = INT32_MIN;
  }
}

// k_compact and k_o3d_to_pm in one pass: the points inside the volume, in order, straight into the PM::DataPoints layout
// (open3d_conversions.cpp:57-118: float casts of the doubles, pad = 1)
// total (nullable): also receives the number of kept points (off[N - 1] + flag[N - 1]) — the word the next consumer on the device
// reads instead of a count handed over through the host
__global__ void __launch_bounds__(kB) k_compact_pm(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t N,
                                                   const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off,
                                                   float4* __restrict__ xyzw, float* __restrict__ out_n, uint32_t* __restrict__ total = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (total && i == N - 1) *total = off[i] + flag[i];
  if (i >= N || !flag[i]) return;
  const int64_t o = (int64_t)off[i];
  xyzw[o] = make_float4((float)pts[3 * i], (float)pts[3 * i + 1], (float)pts[3 * i + 2], 1.0f);
  if (nrm) {
    out_n[3 * o] = (float)nrm[3 * i];
    out_n[3 * o + 1] = (float)nrm[3 * i + 1];
    out_n[3 * o + 2] = (float)nrm[3 * i + 2];
  }
}

// k_compact that also writes the kept points in the PM::DataPoints layout (the reading the ICP will be handed)
__global__ void __launch_bounds__(kB) k_compact_dual(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t N,
                                                     const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off,
                                                     double* __restrict__ out_pts, double* __restrict__ out_n, float4* __restrict__ xyzw,
                                                     float* __restrict__ n32) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N || !flag[i]) return;
  const int64_t o = (int64_t)off[i];
  const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  out_pts[3 * o] = x;
  out_pts[3 * o + 1] = y;
  out_pts[3 * o + 2] = z;
  xyzw[o] = make_float4((float)x, (float)y, (float)z, 1.0f);
  if (nrm) {
    const double a = nrm[3 * i], b = nrm[3 * i + 1], c = nrm[3 * i + 2];
    out_n[3 * o] = a;
    out_n[3 * o + 1] = b;
    out_n[3 * o + 2] = c;
    n32[3 * o] = (float)a;
    n32[3 * o + 1] = (float)b;
    n32[3 * o + 2] = (float)c;
  }
}

// voxel index of every voxelised point; mode 0: absolute grid, reciprocal form (helpers.cpp:156); mode 1: Open3D
// (p - anchor) / voxel.  Points that are not voxelised (flag == 1 = pass-through) get no index.
__global__ void __launch_bounds__(kB) k_vox_keys_idx(const double* __restrict__ pts, int64_t N, const uint32_t* __restrict__ passflag,
                                                     int mode, double inv, double voxel, double ax, double ay, double az,
                                                     int32_t* __restrict__ vidx, int32_t* __restrict__ mm_slots /*[kExtSlots][min[3], max[3]]*/) {
  int32_t* mm = mm_slots + 6 * (blockIdx.x & (kExtSlots - 1));
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  const bool live = i < N && !(passflag && passflag[i]);
  int32_t v[3] = {0, 0, 0};
  if (live) {
    if (mode == 0) {
      v[0] = (int32_t)floor(pts[3 * i] * inv);
      v[1] = (int32_t)floor(pts[3 * i + 1] * inv);
      v[2] = (int32_t)floor(pts[3 * i + 2] * inv);
    } else {
      v[0] = (int32_t)floor((pts[3 * i] - ax) / voxel);
      v[1] = (int32_t)floor((pts[3 * i + 1] - ay) / voxel);
      v[2] = (int32_t)floor((pts[3 * i + 2] - az) / voxel);
    }
    for (int a = 0; a < 3; ++a) vidx[3 * i + a] = v[a];
  }
  for (int a = 0; a < 3; ++a) {  // all lanes take part in the wave reduction; dead lanes carry the neutral element
    const int32_t lo = wave_min_i32(live ? v[a] : INT32_MAX), hi = wave_max_i32(live ? v[a] : INT32_MIN);
    if ((threadIdx.x & 63) == 0 && lo <= hi) {  // a (possibly stale) look first: extrema are monotone, so skipping is safe
      if (lo < __atomic_load_n(&mm[a], __ATOMIC_RELAXED)) atomicMin(&mm[a], lo);
      if (hi > __atomic_load_n(&mm[3 + a], __ATOMIC_RELAXED)) atomicMax(&mm[3 + a], hi);
    }
  }
}

// pass_key: the key of points that are not voxelised — above every packed key, so they sort last
__global__ void __launch_bounds__(kB) k_vox_pack(int64_t N, const uint32_t* __restrict__ passflag, const int32_t* __restrict__ vidx,
                                                 int32_t x0, int32_t y0, int32_t z0, uint64_t ex, uint64_t ey, uint64_t pass_key,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  vals[i] = (uint32_t)i;
  if (passflag && passflag[i]) {
    keys[i] = pass_key;
    return;
  }
  const uint64_t x = (uint64_t)((int64_t)vidx[3 * i] - x0), y = (uint64_t)((int64_t)vidx[3 * i + 1] - y0),
                 z = (uint64_t)((int64_t)vidx[3 * i + 2] - z0);
  keys[i] = (z * ey + y) * ex + x;
}

// key_shift: low bits of the key that only order the members of a voxel (the merge insert's source rank), not part of the voxel
__global__ void __launch_bounds__(kB) k_heads(const uint64_t* __restrict__ keys, int64_t N, uint64_t pass_key, uint32_t* __restrict__ head,
                                              int key_shift = 0, uint32_t* __restrict__ blk_cnt = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  uint32_t f = 0u;
  if (i < N) {
    const uint64_t k = keys[i];
    f = (k != pass_key && (i == 0 || (keys[i - 1] >> key_shift) != (k >> key_shift))) ? 1u : 0u;
    head[i] = f;
  }
  put_block_count(f, blk_cnt);
}

// one lane per voxel: sums run over the voxel's points in ascending input index (stable sort), exactly the order of the
// reference's sequential accumulation (helpers.cpp:30-44, 153-161), so the fp64 means are bit-identical.
__global__ void __launch_bounds__(kB) k_vox_reduce(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                   const uint32_t* __restrict__ head, const uint32_t* __restrict__ ord, int64_t N,
                                                   const double* __restrict__ pts, const double* __restrict__ nrm, const int32_t* __restrict__ vidx,
                                                   int skip_nan_normals, int normalise, int64_t out_base, double* __restrict__ out_pts,
                                                   double* __restrict__ out_n, int32_t* __restrict__ out_idx,
                                                   const uint32_t* __restrict__ base_flag = nullptr, const uint32_t* __restrict__ base_off = nullptr,
                                                   int64_t base_n = 0, int key_shift = 0) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N || !head[i]) return;
  if (base_off) out_base = (int64_t)base_off[base_n - 1] + (int64_t)base_flag[base_n - 1];  // the pass-through count, still on the device
  const uint64_t k = keys[i] >> key_shift;
  double sp[3] = {0, 0, 0}, sn[3] = {0, 0, 0};
  int cnt = 0;
  int64_t j = i;
  for (; j < N && (keys[j] >> key_shift) == k; ++j) {
    const int64_t p = vals[j];
    sp[0] += pts[3 * p];
    sp[1] += pts[3 * p + 1];
    sp[2] += pts[3 * p + 2];
    if (nrm) {
      const double a = nrm[3 * p], b = nrm[3 * p + 1], c = nrm[3 * p + 2];
      if (!skip_nan_normals || (!isnan(a) && !isnan(b) && !isnan(c))) {
        sn[0] += a;
        sn[1] += b;
        sn[2] += c;
      }
    }
    ++cnt;
  }
  const int64_t o = out_base + (int64_t)ord[i];
  const double dn = (double)cnt;
  double an[3] = {sn[0] / dn, sn[1] / dn, sn[2] / dn};
  for (int a = 0; a < 3; ++a) out_pts[3 * o + a] = sp[a] / dn;
  if (nrm) {
    if (normalise) {  // Eigen normalized(): divide by the norm when squaredNorm() > 0
      const double z = an[0] * an[0] + an[1] * an[1] + an[2] * an[2];
      if (z > 0.0) {
        const double r = sqrt(z);
        an[0] = an[0] / r;
        an[1] = an[1] / r;
        an[2] = an[2] / r;
      }
    }
    for (int a = 0; a < 3; ++a) out_n[3 * o + a] = an[a];
  }
  if (out_idx) {
    const int64_t p0 = vals[i];
    for (int a = 0; a < 3; ++a) out_idx[3 * o + a] = vidx[3 * p0 + a];
  }
}

// Optional per-point attributes that ride along with the points: colours (3 doubles) and covariances (9 doubles, the
// Eigen::Matrix3d of open3d::geometry::PointCloud::covariances_ in memory order).  All pointers nullable.
struct Attrs {
  const double* col = nullptr;
  const double* cov = nullptr;
  double* out_col = nullptr;
  double* out_cov = nullptr;
  bool any() const { return col || cov; }
};

// order-preserving compaction of the attributes with the flags / offsets of k_compact
__global__ void __launch_bounds__(kB) k_compact_attr(const double* __restrict__ col, const double* __restrict__ cov, int64_t N,
                                                     const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off,
                                                     double* __restrict__ out_col, double* __restrict__ out_cov) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N || !flag[i]) return;
  const int64_t o = (int64_t)off[i];
  if (col)
    for (int a = 0; a < 3; ++a) out_col[3 * o + a] = col[3 * i + a];
  if (cov)
    for (int a = 0; a < 9; ++a) out_cov[9 * o + a] = cov[9 * i + a];
}

// one lane per voxel, same run structure as k_vox_reduce.  Colours: voxelizeWithinCroppingVolume keeps the LAST colour of the
// voxel in input order (AccumulatedPoint::AddPoint assigns, helpers.cpp:40-42; its isValidColor test compares a bool with
// 0.0 / 1.0 and is always true) and GetAverageColor returns it undivided (:57-59); Open3D's VoxelDownSample (mean_colour)
// averages.  Covariances: sum in input order / count (helpers.cpp:44-46, 61-64).
__global__ void __launch_bounds__(kB) k_vox_reduce_attr(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                        const uint32_t* __restrict__ head, const uint32_t* __restrict__ ord, int64_t N,
                                                        const double* __restrict__ col, const double* __restrict__ cov, int mean_colour,
                                                        int64_t out_base, double* __restrict__ out_col, double* __restrict__ out_cov,
                                                        const uint32_t* __restrict__ base_flag = nullptr, const uint32_t* __restrict__ base_off = nullptr,
                                                        int64_t base_n = 0) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N || !head[i]) return;
  if (base_off) out_base = (int64_t)base_off[base_n - 1] + (int64_t)base_flag[base_n - 1];
  const uint64_t k = keys[i];
  double sc[3] = {0, 0, 0}, sv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int64_t last = vals[i];
  int cnt = 0;
  for (int64_t j = i; j < N && keys[j] == k; ++j) {
    const int64_t p = vals[j];
    last = p;
    if (col && mean_colour)
      for (int a = 0; a < 3; ++a) sc[a] += col[3 * p + a];
    if (cov)
      for (int a = 0; a < 9; ++a) sv[a] += cov[9 * p + a];
    ++cnt;
  }
  const int64_t o = out_base + (int64_t)ord[i];
  const double dn = (double)cnt;
  if (col)
    for (int a = 0; a < 3; ++a) out_col[3 * o + a] = mean_colour ? sc[a] / dn : col[3 * last + a];
  if (cov)
    for (int a = 0; a < 9; ++a) out_cov[9 * o + a] = sv[a] / dn;
}

__global__ void __launch_bounds__(kB) k_min_bound(const double* __restrict__ pts, int64_t N, unsigned long long* __restrict__ mn_slots /*[kExtSlots][3], ordered bits*/) {
  unsigned long long* mn = mn_slots + 3 * (blockIdx.x & (kExtSlots - 1));
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  for (int a = 0; a < 3; ++a) {
    // order-preserving map of a double to u64 so that atomicMin works on negatives too
    unsigned long long u = ~0ull;
    if (i < N) {
      u = (unsigned long long)__double_as_longlong(pts[3 * i + a]);
      u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
    }
    u = wave_min_u64(u);
    if ((threadIdx.x & 63) == 0 && u < __atomic_load_n(&mn[a], __ATOMIC_RELAXED)) atomicMin(&mn[a], u);
  }
}

// ---- the same pipelines without host round trips in the middle ("hinted"): when the cropping volume bounds the voxel
// index range, keys are packed against that range straight away (no extrema, no read-back of them), every count stays on
// the device for the kernels that need it, and ONE post at the end hands the counts and a status word to the host.
struct VoxHint {
  int32_t x0 = 0, y0 = 0, z0 = 0;  // mode 0: lowest index of the range; mode 1: 0 (indices are relative to the anchor)
  uint64_t ex = 0, ey = 0, ez = 0;
  int bits = 0;                    // key bits of the packed range
};

// per-block minima of the points inside `c` (use_crop) as order-preserving u64 bit patterns: part[block][3]
__global__ void __launch_bounds__(kB) k_min_part(o3s_cropper c, int use_crop, const double* __restrict__ pts, int64_t N,
                                                 unsigned long long* __restrict__ part, uint32_t* __restrict__ status) {
  __shared__ unsigned long long sh[kB / 64][3];
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 16) status[threadIdx.x] = 0u;
  double x = 0, y = 0, z = 0;
  bool live = i < N;
  if (live) {
    x = pts[3 * i];
    y = pts[3 * i + 1];
    z = pts[3 * i + 2];
    if (use_crop) live = within(c, x, y, z);
  }
  const double v[3] = {x, y, z};
  for (int a = 0; a < 3; ++a) {
    unsigned long long u = ~0ull;
    if (live) {
      u = (unsigned long long)__double_as_longlong(v[a]);
      u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
    }
    u = wave_min_u64(u);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][a] = u;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned long long u = sh[0][threadIdx.x];
    for (int w = 1; w < kB / 64; ++w) u = sh[w][threadIdx.x] < u ? sh[w][threadIdx.x] : u;
    part[(size_t)blockIdx.x * 3 + threadIdx.x] = u;
  }
}

// voxel index and packed sort key in one pass.  mode 0: floor(p * inv) (helpers.cpp:156) against the hinted range;
// mode 1: Open3D's floor((p - anchor) / voxel) with anchor = min_bound - voxel / 2 folded here from k_min_part's minima.
// Points that are not voxelised — pass-through (passflag) or dropped by the fused crop (mode 1, use_crop) — get pass_key.
__global__ void __launch_bounds__(kB) k_vox_key_direct(const double* __restrict__ pts, int64_t N, const uint32_t* __restrict__ passflag,
                                                       o3s_cropper c, int use_crop, int mode, double inv, double voxel,
                                                       const unsigned long long* __restrict__ part, int n_part, VoxHint h, uint64_t pass_key,
                                                       int32_t* __restrict__ vidx /*nullable*/, uint64_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals, uint32_t* __restrict__ status) {
  __shared__ unsigned long long sh[kB / 64][3];
  __shared__ double s_anchor[3];
  if (mode == 1) {
    unsigned long long m[3] = {~0ull, ~0ull, ~0ull};
    for (int b = threadIdx.x; b < n_part; b += kB)
      for (int a = 0; a < 3; ++a) {
        const unsigned long long u = part[(size_t)b * 3 + a];
        m[a] = u < m[a] ? u : m[a];
      }
    for (int a = 0; a < 3; ++a) {
      const unsigned long long u = wave_min_u64(m[a]);
      if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][a] = u;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      unsigned long long u = sh[0][threadIdx.x];
      for (int w = 1; w < kB / 64; ++w) u = sh[w][threadIdx.x] < u ? sh[w][threadIdx.x] : u;
      u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
      s_anchor[threadIdx.x] = __longlong_as_double((long long)u) - voxel * 0.5;
    }
    __syncthreads();
  }
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  vals[i] = (uint32_t)i;
  const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  bool skip = passflag && passflag[i];
  if (!skip && use_crop) skip = !within(c, x, y, z);
  if (skip) {
    keys[i] = pass_key;
    return;
  }
  int32_t v[3];
  if (mode == 0) {
    v[0] = (int32_t)floor(x * inv);
    v[1] = (int32_t)floor(y * inv);
    v[2] = (int32_t)floor(z * inv);
  } else {
    v[0] = (int32_t)floor((x - s_anchor[0]) / voxel);
    v[1] = (int32_t)floor((y - s_anchor[1]) / voxel);
    v[2] = (int32_t)floor((z - s_anchor[2]) / voxel);
  }
  if (vidx)
    for (int a = 0; a < 3; ++a) vidx[3 * i + a] = v[a];
  const int64_t rx = (int64_t)v[0] - h.x0, ry = (int64_t)v[1] - h.y0, rz = (int64_t)v[2] - h.z0;
  if (rx < 0 || ry < 0 || rz < 0 || (uint64_t)rx >= h.ex || (uint64_t)ry >= h.ey || (uint64_t)rz >= h.ez) {
    atomicOr(status, 1u);  // outside the hinted range: the host repeats the call on the path that measures the range
    keys[i] = 0;
    return;
  }
  keys[i] = ((uint64_t)rz * h.ey + (uint64_t)ry) * h.ex + (uint64_t)rx;
}

// k_mask over a cloud whose size is still on the device: n = cnt_off[cnt_n - 1] + cnt_flag[cnt_n - 1]
__global__ void __launch_bounds__(kB) k_mask_cnt(o3s_cropper c, const double* __restrict__ pts, const uint32_t* __restrict__ cnt_flag,
                                                 const uint32_t* __restrict__ cnt_off, int64_t cnt_n, int64_t N_upper, int keep_inside,
                                                 uint32_t* __restrict__ flag, uint32_t* __restrict__ blk_cnt = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  uint32_t f = 0u;
  if (i < N_upper) {
    const int64_t n = (int64_t)cnt_off[cnt_n - 1] + (int64_t)cnt_flag[cnt_n - 1];
    if (i < n) {
      const bool in = within(c, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
      f = (in == (keep_inside != 0)) ? 1u : 0u;
    }
    flag[i] = f;
  }
  put_block_count(f, blk_cnt);
}

// the one read-back of a hinted pipeline: status and up to three counts (each the total of a flag / offset pair, or 0)
__global__ void k_post_counts(const uint32_t* __restrict__ status, const uint32_t* fa, const uint32_t* oa, int64_t na, const uint32_t* fb,
                              const uint32_t* ob, int64_t nb, const uint32_t* fc, const uint32_t* oc, int64_t nc, uint32_t* __restrict__ dev_out,
                              uint32_t* __restrict__ mailbox, uint32_t seq) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint32_t w[4] = {status[0], fa ? oa[na - 1] + fa[na - 1] : 0u, fb ? ob[nb - 1] + fb[nb - 1] : 0u, fc ? oc[nc - 1] + fc[nc - 1] : 0u};
  for (int k = 0; k < 4; ++k) dev_out[k] = w[k];
  if (mailbox) {
    for (int k = 0; k < 4; ++k) __hip_atomic_store(mailbox + 2 + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// small pinned landing area for the counts the host reads back between kernels (one per host thread): a device-to-host
// copy into pageable memory is staged and costs a full round trip of its own (~20 us in the per-scan loop's trace)
struct PinnedArea {
  uint32_t* p = nullptr;        // 4 KB landing area for small device-to-host copies
  uint32_t* mb = nullptr;       // mailbox a kernel writes directly: [0] value, [1] sequence number (host-coherent memory)
  uint32_t* mb_dev = nullptr;   // the same mailbox as the device addresses it
  uint32_t seq = 0;
  PinnedArea() {
    if (hipHostMalloc(reinterpret_cast<void**>(&p), 4096, hipHostMallocPortable) != hipSuccess) p = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&mb), 128, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) mb = nullptr;  // 32 words
    if (mb) {
      mb[0] = mb[1] = 0;
      if (hipHostGetDevicePointer(reinterpret_cast<void**>(&mb_dev), mb, 0) != hipSuccess) mb_dev = nullptr;
    }
  }
  // no destructor on purpose: areas live in a process-wide pool that is never torn down.  A hipHostFree from a
  // thread_local / static destructor can run after the HIP runtime has been unloaded (exit-time crashes), and the worker
  // threads of o3s_o3d_registration_icp_batch would otherwise allocate and free two pinned buffers per call.
};
struct PinnedPool {
  std::mutex mu;
  std::vector<PinnedArea*> idle;
};
inline PinnedPool& pinned_pool() {
  static PinnedPool* pool = new PinnedPool();  // leaked deliberately (see PinnedArea)
  return *pool;
}
struct PinnedLease {  // one per host thread; hands the area back to the pool when the thread ends
  PinnedArea* a = nullptr;
  PinnedLease() {
    PinnedPool& pool = pinned_pool();
    {
      std::lock_guard<std::mutex> lk(pool.mu);
      if (!pool.idle.empty()) {
        a = pool.idle.back();
        pool.idle.pop_back();
      }
    }
    if (!a) a = new PinnedArea();
  }
  ~PinnedLease() {
    PinnedPool& pool = pinned_pool();
    std::lock_guard<std::mutex> lk(pool.mu);
    pool.idle.push_back(a);
  }
};
inline PinnedArea& pinned_area() {
  static thread_local PinnedLease lease;
  return *lease.a;
}
inline uint32_t* pinned_words() { return pinned_area().p; }
inline bool mailbox_enabled(const PinnedArea& pa) {
  static const bool on = O3S_HOOK_ENV("O3S_NO_MAILBOX") == nullptr;
  return on && pa.mb && pa.mb_dev;
}
inline uint32_t mailbox_next(PinnedArea& pa) {
  if (++pa.seq == 0) ++pa.seq;  // never 0
  return pa.seq;
}
// polls the mailbox until a kernel has posted `seq`; 1 = posted, 0 = the stream drained without the write becoming
// visible (the caller reads the value the slow way), < 0 = error.  A fault upstream must not leave the host spinning:
// the stream is queried every few thousand polls.
inline double poll_now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}
// hipStreamQuery is a call into the runtime (its locks, possibly a marker packet in the queue): it is only the guard against a fault
// upstream, looked at every 200 us of waiting, never part of the polling itself.  (Round 5: polled every ~10 us, the receiving thread's
// waits slowed the MAPPING thread's launches down whenever the two overlapped — the reference re-init took 0.34 ms instead of 0.11
// with page-locked sweeps, where the receiving thread reaches its wait early.)
constexpr double kPollGuardUs = 200.0;
inline int mailbox_wait(PinnedArea& pa, uint32_t seq, hipStream_t s) {
  double t_guard = poll_now_us();
  for (;;) {
    for (int spin = 0; spin < 4096; ++spin)
      if (__atomic_load_n(pa.mb + 1, __ATOMIC_ACQUIRE) == seq) return 1;
    const double t = poll_now_us();
    if (t - t_guard < kPollGuardUs) continue;
    t_guard = t;
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return __atomic_load_n(pa.mb + 1, __ATOMIC_ACQUIRE) == seq ? 1 : 0;
    if (q != hipErrorNotReady) return -1;
  }
}
// A post that is looked at LATER (o3s_submap_insert_processed: the counts of a merge insert, fetched by the next call that needs the
// map).  It lands in the second half of the issuing thread's mailbox (words 16..31: sequence number at 17, values from 18), which no
// other hand-over uses, and in `dev_out` (device memory the owner keeps) for a reader that finds the slot taken by a later post.
struct LazyPost {
  uint32_t* mb_host = nullptr;  // the slot as the host reads it (pinned areas are never freed: any thread may poll it)
  uint32_t* mb_dev = nullptr;
  uint32_t* dev_out = nullptr;  // 4 words of device memory
  uint32_t seq = 0;
  hipStream_t stream = nullptr;
};
// One post per slot at a time: a thread that opens a second lazy post while its first has not landed yet (pending inserts on two submaps,
// never the per-scan loop) waits for the first one here, so that a reader never sees the values of one post under the sequence number of
// another.  The wait ends with the post or with its stream drained (or gone).
struct LazySlotGuard {
  bool busy = false;
  uint32_t seq = 0;
  hipStream_t stream = nullptr;
};
inline LazySlotGuard& lazy_slot_guard() {
  static thread_local LazySlotGuard g;
  return g;
}
inline bool lazy_post_open(PinnedArea& pa, uint32_t* dev_out, hipStream_t s, LazyPost* lp) {
  if (!mailbox_enabled(pa) || !dev_out) return false;
  LazySlotGuard& g = lazy_slot_guard();
  while (g.busy) {
    bool landed = false;
    for (int spin = 0; spin < 4096 && !landed; ++spin) landed = __atomic_load_n(pa.mb + 16 + 1, __ATOMIC_ACQUIRE) == g.seq;
    if (landed || hipStreamQuery(g.stream) != hipErrorNotReady) g.busy = false;
  }
  lp->mb_host = pa.mb + 16;
  lp->mb_dev = pa.mb_dev + 16;
  lp->dev_out = dev_out;
  lp->seq = mailbox_next(pa);
  lp->stream = s;
  g.busy = true;
  g.seq = lp->seq;
  g.stream = s;
  return true;
}
// the four words of a lazy post; waits for them if they are not there yet
inline int lazy_post_fetch(const LazyPost& lp, uint32_t r[4]) {
  double t_guard = poll_now_us() - kPollGuardUs;  // (the first look at the stream may come at once: the post is usually long there, or never will be)
  for (;;) {
    bool posted = false;
    for (int spin = 0; spin < 4096 && !posted; ++spin) posted = __atomic_load_n(lp.mb_host + 1, __ATOMIC_ACQUIRE) == lp.seq;
    if (posted) {
      for (int k = 0; k < 4; ++k) r[k] = __atomic_load_n(lp.mb_host + 2 + k, __ATOMIC_RELAXED);
      if (__atomic_load_n(lp.mb_host + 1, __ATOMIC_ACQUIRE) == lp.seq) return O3S_OK;  // (not overwritten by a later post meanwhile)
    }
    const double t = poll_now_us();
    if (t - t_guard < kPollGuardUs) continue;
    t_guard = t;
    const hipError_t q = hipStreamQuery(lp.stream);
    if (q == hipErrorNotReady) continue;
    if (q != hipSuccess) return O3S_ERR_HIP;
    if (__atomic_load_n(lp.mb_host + 1, __ATOMIC_ACQUIRE) == lp.seq) continue;  // drained and posted: read it above
    // drained, and the slot holds another post (the issuing thread went on to another submap): the device copy
    uint32_t local[4];
    CK(hipMemcpy(local, lp.dev_out, 16, hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) r[k] = local[k];
    return O3S_OK;
  }
}

// folds the kExtSlots replicas of the int32 extrema and posts the six results (mailbox words 2..7)
__global__ void k_ext_post(const int32_t* __restrict__ slots, uint32_t* __restrict__ mailbox, uint32_t seq) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int a = 0; a < 6; ++a) {
    int32_t v = a < 3 ? INT32_MAX : INT32_MIN;
    for (int k = 0; k < kExtSlots; ++k) v = a < 3 ? min(v, slots[k * 6 + a]) : max(v, slots[k * 6 + a]);
    __hip_atomic_store(mailbox + 2 + a, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// host side of the replicated extrema: initialise all replicas, read them back and fold
inline int ext_i32_init(int32_t* d, hipStream_t s) {
  int32_t init[kExtSlots * 6];  // pageable source: hipMemcpyAsync stages it before returning
  for (int k = 0; k < kExtSlots; ++k)
    for (int a = 0; a < 6; ++a) init[k * 6 + a] = a < 3 ? INT32_MAX : INT32_MIN;
  CK(hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, s));
  return O3S_OK;
}
// the same for the three u64 (order-preserving double bits) minima of k_min_bound: words 2..7 = lo/hi halves
__global__ void k_mn_post(const unsigned long long* __restrict__ slots, uint32_t* __restrict__ mailbox, uint32_t seq) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int a = 0; a < 3; ++a) {
    unsigned long long v = ~0ull;
    for (int k = 0; k < kExtSlots; ++k) v = slots[k * 3 + a] < v ? slots[k * 3 + a] : v;
    __hip_atomic_store(mailbox + 2 + 2 * a, (uint32_t)(v & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox + 3 + 2 * a, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
inline int ext_i32_fetch(const int32_t* d, int32_t out[6], hipStream_t s) {
  PinnedArea& pa = pinned_area();
  if (mailbox_enabled(pa)) {
    const uint32_t seq = mailbox_next(pa);
    hipLaunchKernelGGL(k_ext_post, dim3(1), dim3(64), 0, s, d, pa.mb_dev, seq);
    CK(hipGetLastError());
    const int w = mailbox_wait(pa, seq, s);
    if (w < 0) return O3S_ERR_HIP;
    if (w == 1) {
      for (int a = 0; a < 6; ++a) out[a] = (int32_t)__atomic_load_n(pa.mb + 2 + a, __ATOMIC_RELAXED);
      return O3S_OK;
    }
  }
  int32_t local[kExtSlots * 6];
  int32_t* h = pinned_words() ? reinterpret_cast<int32_t*>(pinned_words()) : local;
  CK(hipMemcpyAsync(h, d, sizeof(local), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  for (int a = 0; a < 6; ++a) out[a] = a < 3 ? INT32_MAX : INT32_MIN;
  for (int k = 0; k < kExtSlots; ++k)
    for (int a = 0; a < 6; ++a) out[a] = a < 3 ? std::min(out[a], h[k * 6 + a]) : std::max(out[a], h[k * 6 + a]);
  return O3S_OK;
}

// ---- grow-only device arena: one allocation per pipeline call at most, none once it has seen the largest input -----
struct Arena {
  void* base = nullptr;
  size_t cap = 0, used = 0;
  ~Arena() {
    if (base) (void)hipFree(base);
  }
  hipError_t reserve(size_t bytes) {
    used = 0;
    if (bytes <= cap) return hipSuccess;
    const size_t had = cap;
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    if (had && want < 2 * had) want = 2 * had;  // growing again: the cloud behind it grows (a map); hipFree / hipMalloc stall the device for milliseconds
    const hipError_t e = hipMalloc(&base, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  static size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(reinterpret_cast<char*>(base) + used);
    used += pad(count * sizeof(T));
    return p;
  }
};

// grow-only work area of the overlap selection (overlap_impl.h); owned by the target submap or by the host-buffer entry
struct OverlapWork {
  Arena arena;
};

inline size_t scan_temp_bytes(int64_t n) {  // rocPRIM's scan, or the block counts of the one-launch scan (k_scan_flags_blk): either lives there
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(), nullptr);
  return std::max(bytes, (size_t)(nblk(std::max<int64_t>(n, 1)) + 1) * 4 + 256);
}
// radix-sort bits that cover keys in [0, n_keys): every pass of 8 bits the sort does not have to make is a pass over the data saved
inline int key_bits(uint64_t n_keys) {
  int b = 1;
  while (b < 64 && (n_keys - 1) >> b) ++b;
  return b;
}
// rocPRIM sorts up to a million pairs by merge sort (a block sort, then one pass per doubling) and switches to Onesweep above.
// For the 50 k - 130 k-point clouds of the per-scan loop that is the right choice (Onesweep forced: pre-process 0.12 -> 0.18 ms;
// 4 096-item sort tiles: no change).  For the 0.4 M - 0.8 M-point clouds of a loop-closure refinement (target grid, source order)
// and of a large map sorted as a whole, with keys whose range is known (<= 32 bits: four Onesweep passes against ten merge
// passes) the switch is taken earlier.  O3S_ONESWEEP_FROM=<n> moves it (0: rocPRIM's default).
using SortConfigOnesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;
inline int64_t onesweep_from() {
  static const int64_t v = [] {
    const char* e = O3S_HOOK_ENV("O3S_ONESWEEP_FROM");
    return e ? (int64_t)atoll(e) : (int64_t)262144;
  }();
  return v;
}
// (Rounds 3 - 4 sorted up to 16 384 pairs with a sort of their own — tiles of 2 048 sorted by a bitonic network in LDS, ranks by binary search
// in the other tiles.  Round 5 measured it kernel by kernel: the tile sort alone is a 26 us launch (66 steps, a barrier after each; a version
// with eight pairs per thread and wave shuffles: 34 us), rocPRIM's whole sort of 16 k pairs 28 us, and the per-scan loop on 12 k-point scans ran
// 8 % faster without it — removed; profiles/LAB_NOTES_r05.md 12.)
inline hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, const uint64_t* keys, uint64_t* keys_out, const uint32_t* vals, uint32_t* vals_out, size_t n,
                             int end_bit, hipStream_t s) {
  const int64_t from = onesweep_from();
  if (from > 0 && (int64_t)n >= from && end_bit <= 40)
    return rocprim::radix_sort_pairs<SortConfigOnesweep>(tmp, tmp_bytes, keys, keys_out, vals, vals_out, n, 0, (unsigned)end_bit, s);
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_out, vals, vals_out, n, 0, (unsigned)end_bit, s);
}
inline size_t sort_temp_bytes(int64_t n) {  // enough for either algorithm and any bit range
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n,
                                  0, 64, nullptr);
  (void)rocprim::radix_sort_pairs<SortConfigOnesweep>(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                                      (uint32_t*)nullptr, (size_t)n, 0, 64, nullptr);
  return std::max(a, b);
}

// off[n] = number of set flags; with a mailbox the count and then a sequence number also go straight into host-coherent
// pinned memory, where the host is already polling: no copy kernel, no completion interrupt — the wait between two
// dependent launches drops from ~20 us to the PCIe write
__global__ void k_scan_total(const uint32_t* __restrict__ flag, uint32_t* __restrict__ off, int64_t n, uint32_t* __restrict__ mailbox, uint32_t seq) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const uint32_t total = off[n - 1] + flag[n - 1];
  off[n] = total;
  if (mailbox) {
    __hip_atomic_store(mailbox, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// flag -> exclusive offsets (off holds n + 1 words: off[n] = the number of set flags, which is also returned)
inline int scan_flags_dev(const uint32_t* flag, uint32_t* off, int64_t n, void* tmp, size_t tmp_bytes, hipStream_t s, const uint32_t* blk_cnt = nullptr);
inline int scan_flags(const uint32_t* flag, uint32_t* off, int64_t n, void* tmp, size_t tmp_bytes, int64_t* count, hipStream_t s,
                      const uint32_t* blk_cnt = nullptr /*the producer's per-block counts (put_block_count): the scan is then one launch*/) {
  {
    const int rc = scan_flags_dev(flag, off, n, tmp, tmp_bytes, s, blk_cnt);
    if (rc != O3S_OK) return rc;
  }
  PinnedArea& pa = pinned_area();
  if (mailbox_enabled(pa)) {
    const uint32_t seq = mailbox_next(pa);
    hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(64), 0, s, flag, off, n, pa.mb_dev, seq);
    CK(hipGetLastError());
    const int w = mailbox_wait(pa, seq, s);
    if (w < 0) return O3S_ERR_HIP;
    if (w == 1) {
      *count = (int64_t)__atomic_load_n(pa.mb, __ATOMIC_RELAXED);
      return O3S_OK;
    }
  } else {
    hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(64), 0, s, flag, off, n, (uint32_t*)nullptr, 0u);
  }
  uint32_t local = 0;
  uint32_t* dst = pa.p ? pa.p : &local;
  CK(hipMemcpyAsync(dst, off + n, 4, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  *count = (int64_t)*dst;
  return O3S_OK;
}

inline size_t crop_arena_bytes(int64_t N) { return Arena::pad((size_t)N * 4) + Arena::pad((size_t)(N + 1) * 4) + Arena::pad(scan_temp_bytes(N)) + 1024; }

// CroppingVolume::crop (croppers.cpp:76-106) on device arrays: order-preserving compaction of the points inside
inline int crop_dev(Arena& ar, const o3s_cropper& c, const double* d_pts, const double* d_nrm, int64_t N, double* d_opts, double* d_on,
                    int64_t* kept, hipStream_t s, const Attrs* at = nullptr) {
  *kept = 0;
  if (N == 0) return O3S_OK;
  CK(ar.reserve(crop_arena_bytes(N)));
  uint32_t* flag = ar.take<uint32_t>((size_t)N);
  uint32_t* off = ar.take<uint32_t>((size_t)N + 1);
  const size_t tb = scan_temp_bytes(N);
  void* tmp = ar.take<char>(tb);
  hipLaunchKernelGGL(k_mask, dim3(nblk(N)), dim3(kB), 0, s, c, d_pts, N, 1, flag);
  const int rc = scan_flags(flag, off, N, tmp, tb, kept, s);
  if (rc != O3S_OK) return rc;
  hipLaunchKernelGGL(k_compact, dim3(nblk(N)), dim3(kB), 0, s, d_pts, d_nrm, N, flag, off, d_opts, d_on, (int32_t*)nullptr);
  if (at && at->any()) hipLaunchKernelGGL(k_compact_attr, dim3(nblk(N)), dim3(kB), 0, s, at->col, at->cov, N, flag, off, at->out_col, at->out_cov);
  CK(hipGetLastError());
  return O3S_OK;
}

inline size_t voxel_arena_bytes(int64_t N) {
  const size_t n = (size_t)N;
  return Arena::pad(n * 4) + Arena::pad((n + 1) * 4)                 // flag, off
         + Arena::pad(n * 12) + Arena::pad(kExtSlots * 6 * 4) + Arena::pad(kExtSlots * 3 * 8)  // vidx, mm, mn
         + 2 * Arena::pad(n * 8) + 2 * Arena::pad(n * 4)             // keys x2, vals x2
         + Arena::pad(n * 4) + Arena::pad((n + 1) * 4)               // head, ord
         + Arena::pad(std::max(scan_temp_bytes(N), sort_temp_bytes(N))) + 4096;
}

// Shared body of the two voxelisers on device arrays: pass-through compaction, sort of the voxelised points by packed
// voxel key, per-voxel reduction.  d_opts / d_on / d_oidx hold up to N points.
inline int voxel_pipeline_dev(Arena& ar, int mode, const o3s_cropper* crop, double voxel, const double* d_pts, const double* d_nrm, int64_t N,
                              double* d_opts, double* d_on, int32_t* d_oidx, int64_t* n_out, hipStream_t s, const Attrs* at = nullptr) {
  *n_out = 0;
  if (N == 0) return O3S_OK;
  if (N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  CK(ar.reserve(voxel_arena_bytes(N)));
  uint32_t* flag = ar.take<uint32_t>((size_t)N);
  uint32_t* off = ar.take<uint32_t>((size_t)N + 1);
  int32_t* vidx = ar.take<int32_t>((size_t)N * 3);
  int32_t* d_mm = ar.take<int32_t>(kExtSlots * 6);
  unsigned long long* d_mn = ar.take<unsigned long long>(kExtSlots * 3);
  uint64_t* keys = ar.take<uint64_t>((size_t)N);
  uint64_t* keys2 = ar.take<uint64_t>((size_t)N);
  uint32_t* vals = ar.take<uint32_t>((size_t)N);
  uint32_t* vals2 = ar.take<uint32_t>((size_t)N);
  uint32_t* head = ar.take<uint32_t>((size_t)N);
  uint32_t* ord = ar.take<uint32_t>((size_t)N + 1);
  const size_t tb_scan = scan_temp_bytes(N), tb_sort = sort_temp_bytes(N);
  void* tmp = ar.take<char>(std::max(tb_scan, tb_sort));
  int64_t n_pass = 0;
  const uint32_t* passflag = nullptr;
  if (crop) {  // pass-through points: outside the volume, emitted first in input order (helpers.cpp:162-176)
    hipLaunchKernelGGL(k_mask, dim3(nblk(N)), dim3(kB), 0, s, *crop, d_pts, N, 0, flag);
    const int rc = scan_flags(flag, off, N, tmp, tb_scan, &n_pass, s);
    if (rc != O3S_OK) return rc;
    hipLaunchKernelGGL(k_compact, dim3(nblk(N)), dim3(kB), 0, s, d_pts, d_nrm, N, flag, off, d_opts, d_on, d_oidx);
    if (at && at->any()) hipLaunchKernelGGL(k_compact_attr, dim3(nblk(N)), dim3(kB), 0, s, at->col, at->cov, N, flag, off, at->out_col, at->out_cov);
    passflag = flag;
  }
  double ax = 0, ay = 0, az = 0;
  if (mode == 1) {  // Open3D: anchor = min_bound - voxel/2
    CK(hipMemsetAsync(d_mn, 0xff, kExtSlots * 24, s));
    hipLaunchKernelGGL(k_min_bound, dim3(nblk(N)), dim3(kB), 0, s, d_pts, N, d_mn);
    unsigned long long mn[3] = {~0ull, ~0ull, ~0ull};
    PinnedArea& pa = pinned_area();
    int posted = 0;
    if (mailbox_enabled(pa)) {
      const uint32_t seq = mailbox_next(pa);
      hipLaunchKernelGGL(k_mn_post, dim3(1), dim3(64), 0, s, d_mn, pa.mb_dev, seq);
      CK(hipGetLastError());
      posted = mailbox_wait(pa, seq, s);
      if (posted < 0) return O3S_ERR_HIP;
      if (posted == 1)
        for (int a = 0; a < 3; ++a)
          mn[a] = (unsigned long long)__atomic_load_n(pa.mb + 2 + 2 * a, __ATOMIC_RELAXED) |
                  ((unsigned long long)__atomic_load_n(pa.mb + 3 + 2 * a, __ATOMIC_RELAXED) << 32);
    }
    if (posted != 1) {
      unsigned long long mn_local[kExtSlots * 3];
      unsigned long long* mn_all = pa.p ? reinterpret_cast<unsigned long long*>(pa.p) : mn_local;
      CK(hipMemcpyAsync(mn_all, d_mn, sizeof(mn_local), hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      for (int k = 0; k < kExtSlots; ++k)
        for (int a = 0; a < 3; ++a) mn[a] = std::min(mn[a], mn_all[k * 3 + a]);
    }
    double m[3];
    for (int a = 0; a < 3; ++a) {
      unsigned long long u = mn[a];
      u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
      std::memcpy(&m[a], &u, 8);
    }
    ax = m[0] - voxel * 0.5;
    ay = m[1] - voxel * 0.5;
    az = m[2] - voxel * 0.5;
  }
  {
    const int rc0 = ext_i32_init(d_mm, s);
    if (rc0 != O3S_OK) return rc0;
  }
  hipLaunchKernelGGL(k_vox_keys_idx, dim3(nblk(N)), dim3(kB), 0, s, d_pts, N, passflag, mode, 1.0 / voxel, voxel, ax, ay, az, vidx, d_mm);
  int32_t mm[6];
  {
    const int rc0 = ext_i32_fetch(d_mm, mm, s);
    if (rc0 != O3S_OK) return rc0;
  }
  int64_t n_vox = 0;
  if (n_pass < N) {
    const uint64_t ex = (uint64_t)((int64_t)mm[3] - mm[0] + 1), ey = (uint64_t)((int64_t)mm[4] - mm[1] + 1),
                   ez = (uint64_t)((int64_t)mm[5] - mm[2] + 1);
    const long double prod = (long double)ex * (long double)ey * (long double)ez;
    if (prod >= 9.0e18L) return O3S_ERR_BAD_ARGUMENT;  // voxel index range does not pack into 63 bits
    // only as many key bits as the packed range needs; the pass-through key is the next power of two (one more bit) —
    // sorting all 64 bits took nine radix passes over the whole map where four do
    int bits = 1;
    while (bits < 63 && ((long double)(1ull << bits)) <= prod) ++bits;
    const uint64_t pass_key = bits < 63 ? (1ull << bits) : ~0ull;
    const int end_bit = passflag ? (bits < 63 ? bits + 1 : 64) : bits;
    hipLaunchKernelGGL(k_vox_pack, dim3(nblk(N)), dim3(kB), 0, s, N, passflag, vidx, mm[0], mm[1], mm[2], ex, ey, pass_key, keys, vals);
    size_t tb = tb_sort;
    CK(sort_pairs(tmp, tb, keys, keys2, vals, vals2, (size_t)N, end_bit, s));
    hipLaunchKernelGGL(k_heads, dim3(nblk(N)), dim3(kB), 0, s, keys2, N, pass_key, head);
    const int rc = scan_flags(head, ord, N, tmp, tb_scan, &n_vox, s);
    if (rc != O3S_OK) return rc;
    hipLaunchKernelGGL(k_vox_reduce, dim3(nblk(N)), dim3(kB), 0, s, keys2, vals2, head, ord, N, d_pts, d_nrm, vidx, mode == 0 ? 1 : 0,
                       mode == 0 ? 1 : 0, n_pass, d_opts, d_on, d_oidx);
    if (at && at->any())
      hipLaunchKernelGGL(k_vox_reduce_attr, dim3(nblk(N)), dim3(kB), 0, s, keys2, vals2, head, ord, N, at->col, at->cov, mode == 1 ? 1 : 0, n_pass,
                         at->out_col, at->out_cov);
  }
  CK(hipGetLastError());
  *n_out = n_pass + n_vox;
  return O3S_OK;
}

inline bool hints_enabled() { return O3S_HOOK_ENV("O3S_NO_HINT") == nullptr; }  // read per call: the tests run both paths in one process
// axis-aligned box that certainly contains the volume; false for unbounded volumes (croppers.cpp:121-167)
inline bool cropper_aabb(const o3s_cropper& c, double lo[3], double hi[3]) {
  if (c.invert) return false;
  double r;
  switch (c.kind) {
    case 1: r = c.p0; break;
    case 3: r = c.p1; break;
    case 4: r = c.p0; break;
    default: return false;
  }
  if (!(r >= 0.0) || !std::isfinite(r)) return false;
  for (int a = 0; a < 3; ++a) {
    lo[a] = c.centre[a] - r;
    hi[a] = c.centre[a] + r;
  }
  if (c.kind == 4) {
    lo[2] = c.p1;
    hi[2] = c.p2;
  }
  for (int a = 0; a < 3; ++a)
    if (!std::isfinite(lo[a]) || !std::isfinite(hi[a]) || !(hi[a] >= lo[a])) return false;
  return true;
}
// voxel index range of everything inside [lo, hi] (two cells of slack per side); false when the packed range would need
// more key bits than a measured range typically does (the caller then measures)
inline bool vox_hint(int mode, const double lo[3], const double hi[3], double voxel, VoxHint* h) {
  if (!(voxel > 0.0)) return false;
  const double inv = 1.0 / voxel;
  int32_t o[3];
  uint64_t e[3];
  for (int a = 0; a < 3; ++a) {
    double a0, a1;
    if (mode == 0) {
      a0 = std::floor(lo[a] * inv) - 2.0;
      a1 = std::floor(hi[a] * inv) + 2.0;
    } else {
      a0 = 0.0;
      a1 = std::floor((hi[a] - lo[a]) / voxel) + 3.0;
    }
    if (!(std::fabs(a0) < 1.0e9) || !(std::fabs(a1) < 1.0e9)) return false;
    o[a] = (int32_t)a0;
    e[a] = (uint64_t)(a1 - a0 + 1.0);
  }
  if (O3S_HOOK_ENV("O3S_HINT_MISS")) e[0] = e[1] = e[2] = 1;  // test hook: a range nothing fits in, so that the status word trips and
                                                        // the caller has to repeat on the measuring path
  const long double prod = (long double)e[0] * (long double)e[1] * (long double)e[2];
  int bits = 1;
  while (bits < 63 && ((long double)(1ull << bits)) < prod) ++bits;
  if (bits > 36) return false;
  h->x0 = o[0];
  h->y0 = o[1];
  h->z0 = o[2];
  h->ex = e[0];
  h->ey = e[1];
  h->ez = e[2];
  h->bits = bits;
  return true;
}

// Exclusive scan of 0 / 1 flags whose producer left its per-block counts (put_block_count; blocks of kB flags): block b adds up the counts
// of the blocks before it (<= n / 256 words, from L2), ranks its own flags with a ballot per wave and writes off[i]; the last flag's
// thread also writes off[n] = the total.  One launch, no look-back state, no spinning.
__global__ void __launch_bounds__(kB) k_scan_flags_blk(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ blk_cnt, int64_t n,
                                                       uint32_t* __restrict__ off) {
  __shared__ uint32_t s_part[kB / 64], s_wave[kB / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t acc = 0u;
  for (int k = threadIdx.x; k < (int)blockIdx.x; k += kB) acc += blk_cnt[k];
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  const uint32_t f = i < n ? (flag[i] != 0u ? 1u : 0u) : 0u;
  const unsigned long long m = __ballot(f != 0u);
  if (lane == 0) {
    s_part[w] = acc;
    s_wave[w] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  uint32_t base = 0u;
#pragma unroll
  for (int k = 0; k < kB / 64; ++k) base += s_part[k] + (k < w ? s_wave[k] : 0u);
  const uint32_t excl = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (i < n) off[i] = excl;
  if (i == n - 1) off[n] = excl + f;
}
inline int scan_flags_dev(const uint32_t* flag, uint32_t* off, int64_t n, void* tmp, size_t tmp_bytes, hipStream_t s, const uint32_t* blk_cnt) {
  if (blk_cnt && n > 0 && O3S_HOOK_ENV("O3S_SCAN_ROCPRIM") == nullptr) {
    hipLaunchKernelGGL(k_scan_flags_blk, dim3(nblk(n)), dim3(kB), 0, s, flag, blk_cnt, n, off);
    return hipGetLastError() == hipSuccess ? O3S_OK : O3S_ERR_HIP;
  }
  CK(rocprim::exclusive_scan(tmp, tmp_bytes, flag, off, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
  return O3S_OK;
}

// The voxelisers with a hinted index range.  mode 0 (voxelizeWithinCroppingVolume): points outside `crop` pass through
// first.  mode 1 (Open3D VoxelDownSample): points outside `crop` (nullable) are DROPPED — the wide crop of
// ScanToMapIcp::processForScanMatchingAndMerging fused into the down-sampler; per-voxel sums still run in input order.
// post_crop (nullable): CroppingVolume::crop of the voxelised output into d_ppts / d_pn (the narrow crop).
// counts[0..2] = pass-through points, voxels, points kept by post_crop.  *ok = false: an index fell outside the hint (or
// the mailbox is off) and nothing was produced — the caller repeats on the measuring path.
inline int voxel_pipeline_hint_dev(Arena& ar, int mode, const o3s_cropper* crop, const VoxHint& h, double voxel, const double* d_pts,
                                   const double* d_nrm, int64_t N, double* d_opts, double* d_on, int32_t* d_oidx, const Attrs* at,
                                   const o3s_cropper* post_crop, double* d_ppts, double* d_pn, int64_t counts[3], bool* ok, hipStream_t s,
                                   float4* pm_xyzw = nullptr, float* pm_n = nullptr /*post_crop output also in the PM layout*/) {
  counts[0] = counts[1] = counts[2] = 0;
  *ok = false;
  PinnedArea& pa = pinned_area();
  if (N <= 0 || N > (int64_t)0x7fffffff || !mailbox_enabled(pa)) return O3S_OK;
  CK(ar.reserve(voxel_arena_bytes(N)));
  uint32_t* flag = ar.take<uint32_t>((size_t)N);
  uint32_t* off = ar.take<uint32_t>((size_t)N + 1);
  int32_t* vidx = ar.take<int32_t>((size_t)N * 3);
  uint32_t* status = reinterpret_cast<uint32_t*>(ar.take<int32_t>(kExtSlots * 6));  // 16 words used
  unsigned long long* part = ar.take<unsigned long long>(kExtSlots * 3);           // mode 1 needs nblk(N) * 3: taken from vidx when it is free
  uint64_t* keys = ar.take<uint64_t>((size_t)N);
  uint64_t* keys2 = ar.take<uint64_t>((size_t)N);
  uint32_t* vals = ar.take<uint32_t>((size_t)N);
  uint32_t* vals2 = ar.take<uint32_t>((size_t)N);
  uint32_t* head = ar.take<uint32_t>((size_t)N);
  uint32_t* ord = ar.take<uint32_t>((size_t)N + 1);
  const size_t tb_scan = scan_temp_bytes(N), tb_sort = sort_temp_bytes(N);
  void* tmp = ar.take<char>(std::max(tb_scan, tb_sort));
  uint32_t* blk = reinterpret_cast<uint32_t*>(tmp);  // the flag producers' per-block counts (scan_temp_bytes has room for them): written and
                                                     // consumed between two uses of tmp by the sort
  const unsigned nb = nblk(N);
  const bool pass = mode == 0 && crop != nullptr;
  const o3s_cropper none{};
  if (mode == 1) {
    // nblk(N) * 24 bytes of minima: `head` (4 N bytes) is free until k_heads and large enough for N >= 2 blocks of input
    part = (size_t)nb * 24 <= (size_t)N * 4 ? reinterpret_cast<unsigned long long*>(head) : part;
    if ((size_t)nb * 24 > (size_t)N * 4 && nb > (unsigned)kExtSlots) return O3S_OK;
    hipLaunchKernelGGL(k_min_part, dim3(nb), dim3(kB), 0, s, crop ? *crop : none, crop ? 1 : 0, d_pts, N, part, status);
  } else {
    if (!pass) CK(hipMemsetAsync(status, 0, 64, s));
    if (pass) {
      hipLaunchKernelGGL(k_mask, dim3(nb), dim3(kB), 0, s, *crop, d_pts, N, 0, flag, status, blk);
      const int rc = scan_flags_dev(flag, off, N, tmp, tb_scan, s, blk);
      if (rc != O3S_OK) return rc;
      hipLaunchKernelGGL(k_compact, dim3(nb), dim3(kB), 0, s, d_pts, d_nrm, N, flag, off, d_opts, d_on, d_oidx);
      if (at && at->any()) hipLaunchKernelGGL(k_compact_attr, dim3(nb), dim3(kB), 0, s, at->col, at->cov, N, flag, off, at->out_col, at->out_cov);
    }
  }
  const uint64_t pass_key = 1ull << h.bits;
  hipLaunchKernelGGL(k_vox_key_direct, dim3(nb), dim3(kB), 0, s, d_pts, N, pass ? flag : nullptr, (mode == 1 && crop) ? *crop : none,
                     (mode == 1 && crop) ? 1 : 0, mode, 1.0 / voxel, voxel, part, (int)nb, h, pass_key, d_oidx ? vidx : nullptr, keys, vals, status);
  size_t tb = tb_sort;
  CK(sort_pairs(tmp, tb, keys, keys2, vals, vals2, (size_t)N, h.bits + 1, s));
  hipLaunchKernelGGL(k_heads, dim3(nb), dim3(kB), 0, s, keys2, N, pass_key, head, 0, blk);
  {
    const int rc = scan_flags_dev(head, ord, N, tmp, tb_scan, s, blk);
    if (rc != O3S_OK) return rc;
  }
  hipLaunchKernelGGL(k_vox_reduce, dim3(nb), dim3(kB), 0, s, keys2, vals2, head, ord, N, d_pts, d_nrm, vidx, mode == 0 ? 1 : 0, mode == 0 ? 1 : 0,
                     (int64_t)0, d_opts, d_on, d_oidx, pass ? flag : nullptr, pass ? off : nullptr, N);
  if (at && at->any())
    hipLaunchKernelGGL(k_vox_reduce_attr, dim3(nb), dim3(kB), 0, s, keys2, vals2, head, ord, N, at->col, at->cov, mode == 1 ? 1 : 0, (int64_t)0,
                       at->out_col, at->out_cov, pass ? flag : nullptr, pass ? off : nullptr, N);
  const bool post = post_crop != nullptr && !pass;  // flag / off are free when nothing passes through
  if (post) {
    hipLaunchKernelGGL(k_mask_cnt, dim3(nb), dim3(kB), 0, s, *post_crop, d_opts, head, ord, N, N, 1, flag, blk);
    const int rc = scan_flags_dev(flag, off, N, tmp, tb_scan, s, blk);
    if (rc != O3S_OK) return rc;
    if (pm_xyzw) hipLaunchKernelGGL(k_compact_dual, dim3(nb), dim3(kB), 0, s, d_opts, d_on, N, flag, off, d_ppts, d_pn, pm_xyzw, pm_n);
    else hipLaunchKernelGGL(k_compact, dim3(nb), dim3(kB), 0, s, d_opts, d_on, N, flag, off, d_ppts, d_pn, (int32_t*)nullptr);
  } else if (post_crop) {
    return O3S_ERR_BAD_ARGUMENT;
  }
  const uint32_t seq = mailbox_next(pa);
  hipLaunchKernelGGL(k_post_counts, dim3(1), dim3(64), 0, s, status, pass ? flag : nullptr, pass ? off : nullptr, N, head, ord, N,
                     post ? flag : nullptr, post ? off : nullptr, N, status + 4, pa.mb_dev, seq);
  CK(hipGetLastError());
  const int w = mailbox_wait(pa, seq, s);
  if (w < 0) return O3S_ERR_HIP;
  uint32_t r[4];
  if (w == 1) {
    for (int k = 0; k < 4; ++k) r[k] = __atomic_load_n(pa.mb + 2 + k, __ATOMIC_RELAXED);
  } else {
    uint32_t local[4];
    uint32_t* dst = pa.p ? pa.p : local;
    CK(hipMemcpyAsync(dst, status + 4, 16, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    for (int k = 0; k < 4; ++k) r[k] = dst[k];
  }
  if (r[0] != 0u) return O3S_OK;  // *ok stays false
  counts[0] = (int64_t)r[1];
  counts[1] = (int64_t)r[2];
  counts[2] = (int64_t)r[3];
  *ok = true;
  return O3S_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Submap::insertScan without re-sorting the map (the reference's own TODO, Submap.cpp:89-92).
// After a voxelisation the map array is [pass-through points | one point per voxel, in (z, y, x) key order]; the next insert
// appends a scan and voxelises [PT_old | V_old | S] again.  The stable sort of that whole array by voxel key is, whenever
//   (a) no PT_old point lies inside the new volume (they were left behind; a revisit brings them back) and
//   (b) the V_old points inside the new volume still have strictly increasing keys (they do: a voxel's mean stays in its voxel),
// the MERGE of V_old-inside (already in order, compacted in place order) with the scan sorted by key: a key's members come out
// as [the old voxel point, then the scan's points in scan order] — the input order the sort-based path sums them in — and the
// voxels come out in key order.  Both conditions are checked on the device (status bits 4 and 2); when one fails nothing is
// kept and the caller runs the sort-based pipeline, so the two paths can never disagree.  Only the scan (~50 k points) is
// sorted, the map (0.6 M and growing) is streamed: keys, compaction, one merge, heads, per-voxel sums.
// The source rank rides in the two low bits of the key (1 = old voxel point, 2 = scan), so ties need no stability argument.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kB) k_insert_split(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t n_pt, int64_t n_old,
                                                     int64_t n_tmp, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off, double inv, VoxHint h,
                                                     uint64_t sentinel, uint64_t* __restrict__ keysA, uint32_t* __restrict__ valsA,
                                                     uint64_t* __restrict__ keysU, uint32_t* __restrict__ valsU, uint32_t* __restrict__ status,
                                                     double* __restrict__ out_pts, double* __restrict__ out_n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= n_tmp) return;
  const bool pass = flag[i] != 0u;
  if (pass) {  // the pass-through part of the output, in input order (what k_compact does in the sort-based pipeline)
    const int64_t o = (int64_t)off[i];
    for (int a = 0; a < 3; ++a) {
      out_pts[3 * o + a] = pts[3 * i + a];
      if (nrm) out_n[3 * o + a] = nrm[3 * i + a];
    }
  }
  uint64_t key = 0;
  if (!pass) {
    const int32_t v0 = (int32_t)floor(pts[3 * i] * inv), v1 = (int32_t)floor(pts[3 * i + 1] * inv), v2 = (int32_t)floor(pts[3 * i + 2] * inv);
    const int64_t rx = (int64_t)v0 - h.x0, ry = (int64_t)v1 - h.y0, rz = (int64_t)v2 - h.z0;
    if (rx < 0 || ry < 0 || rz < 0 || (uint64_t)rx >= h.ex || (uint64_t)ry >= h.ey || (uint64_t)rz >= h.ez) atomicOr(status, 1u);  // outside the hinted range
    else key = ((uint64_t)rz * h.ey + (uint64_t)ry) * h.ex + (uint64_t)rx;
  }
  if (i < n_pt) {
    if (!pass) atomicOr(status, 4u);  // an old pass-through point is back inside the volume: the sort-based path handles it
    return;
  }
  if (i < n_old) {
    // inside points before i in [n_pt, i): (i - off[i]) counts the inside points before i over the whole array
    const int64_t base = n_pt - (int64_t)off[n_pt];
    const int64_t n_a = (n_old - (int64_t)off[n_old]) - base;  // n_old < n_tmp: off[n_old] is inside the scanned range
    if (!pass) {
      const int64_t a = (i - (int64_t)off[i]) - base;
      keysA[a] = (key << 2) | 1ull;
      valsA[a] = (uint32_t)i;
    }
    const int64_t r = i - n_pt;
    if (r >= n_a) keysA[r] = sentinel;  // the slots the compaction leaves free: they merge to the very end and head nothing
    return;
  }
  const int64_t u = i - n_old;
  keysU[u] = pass ? sentinel : ((key << 2) | 2ull);
  valsU[u] = (uint32_t)i;
}

__global__ void __launch_bounds__(kB) k_check_increasing(const uint64_t* __restrict__ keys, int64_t n, uint64_t sentinel, uint32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i + 1 >= n) return;
  const uint64_t a = keys[i], b = keys[i + 1];
  if (b != sentinel && !(a < b)) atomicOr(status, 2u);
}

inline size_t merge_temp_bytes(int64_t n1, int64_t n2) {
  size_t bytes = 0;
  (void)rocprim::merge(nullptr, bytes, (const uint64_t*)nullptr, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n1, (size_t)n2, rocprim::less<uint64_t>(), nullptr);
  return bytes;
}
inline size_t insert_merge_arena_bytes(int64_t n_tmp, int64_t n_v, int64_t n_s) {
  const size_t nt = (size_t)n_tmp, nv = (size_t)std::max<int64_t>(n_v, 1), ns = (size_t)std::max<int64_t>(n_s, 1), nm = nv + ns;
  return Arena::pad(nt * 4) + Arena::pad((nt + 1) * 4) + Arena::pad(256)                      // flag, off, status
         + Arena::pad(nv * 8) + Arena::pad(nv * 4) + 2 * Arena::pad(ns * 8) + 2 * Arena::pad(ns * 4)  // A, U, U sorted
         + Arena::pad(nm * 8) + Arena::pad(nm * 4) + Arena::pad(nm * 4) + Arena::pad((nm + 1) * 4)   // merged keys / vals, head, ord
         + Arena::pad(std::max(std::max(scan_temp_bytes(n_tmp), sort_temp_bytes(n_s)), merge_temp_bytes(n_v, n_s))) + 4096;
}

// d_pts / d_nrm: [PT_old (n_pt) | V_old (n_old - n_pt) | appended scan (n_tmp - n_old)].  counts[0..1] = pass-through points,
// voxels.  *ok = false: a condition above failed (or an index fell outside the hint, or the mailbox is off) — nothing usable
// was produced.
// lazy (nullable, opened by lazy_post_open): the counts are posted there and NOT waited for — the call returns with everything enqueued,
// *ok = false and *issued = true; voxel_insert_merge_result() reads them later.
inline int voxel_insert_merge_result(const LazyPost& lp, int64_t counts[2], bool* ok) {
  counts[0] = counts[1] = 0;
  *ok = false;
  uint32_t r[4];
  const int rc = lazy_post_fetch(lp, r);
  if (rc != O3S_OK) return rc;
  if (r[0] != 0u) return O3S_OK;
  counts[0] = (int64_t)r[1];
  counts[1] = (int64_t)r[2];
  *ok = true;
  return O3S_OK;
}
inline int voxel_insert_merge_dev(Arena& ar, const o3s_cropper& crop, const VoxHint& h, double voxel, const double* d_pts, const double* d_nrm,
                                  int64_t n_pt, int64_t n_old, int64_t n_tmp, double* d_opts, double* d_on, int64_t counts[2], bool* ok,
                                  hipStream_t s, const LazyPost* lazy = nullptr, bool* issued = nullptr) {
  counts[0] = counts[1] = 0;
  *ok = false;
  PinnedArea& pa = pinned_area();
  const int64_t n_v = n_old - n_pt, n_s = n_tmp - n_old;
  if (n_v <= 0 || n_s <= 0 || n_pt < 0 || n_tmp > (int64_t)0x7fffffff || h.bits > 58 || !mailbox_enabled(pa)) return O3S_OK;
  CK(ar.reserve(insert_merge_arena_bytes(n_tmp, n_v, n_s)));
  const int64_t n_m = n_v + n_s;
  uint32_t* flag = ar.take<uint32_t>((size_t)n_tmp);
  uint32_t* off = ar.take<uint32_t>((size_t)n_tmp + 1);
  uint32_t* status = ar.take<uint32_t>(64);
  uint64_t* keysA = ar.take<uint64_t>((size_t)n_v);
  uint32_t* valsA = ar.take<uint32_t>((size_t)n_v);
  uint64_t* keysU = ar.take<uint64_t>((size_t)n_s);
  uint64_t* keysU2 = ar.take<uint64_t>((size_t)n_s);
  uint32_t* valsU = ar.take<uint32_t>((size_t)n_s);
  uint32_t* valsU2 = ar.take<uint32_t>((size_t)n_s);
  uint64_t* keysM = ar.take<uint64_t>((size_t)n_m);
  uint32_t* valsM = ar.take<uint32_t>((size_t)n_m);
  uint32_t* head = ar.take<uint32_t>((size_t)n_m);
  uint32_t* ord = ar.take<uint32_t>((size_t)n_m + 1);
  const size_t tb_scan = scan_temp_bytes(n_tmp), tb_sort = sort_temp_bytes(n_s), tb_merge = merge_temp_bytes(n_v, n_s);
  void* tmp = ar.take<char>(std::max(std::max(tb_scan, tb_sort), tb_merge));
  const unsigned nb = nblk(n_tmp);
  const uint64_t sentinel = 1ull << (h.bits + 2);
  // pass-through points: outside the volume, emitted first in input order (helpers.cpp:162-176)
  uint32_t* blk = reinterpret_cast<uint32_t*>(tmp);  // per-block counts of the flag producers (room: scan_temp_bytes), consumed by the scan right behind
  hipLaunchKernelGGL(k_mask, dim3(nb), dim3(kB), 0, s, crop, d_pts, n_tmp, 0, flag, status, blk);
  {
    const int rc = scan_flags_dev(flag, off, n_tmp, tmp, tb_scan, s, blk);
    if (rc != O3S_OK) return rc;
  }
  hipLaunchKernelGGL(k_insert_split, dim3(nb), dim3(kB), 0, s, d_pts, d_nrm, n_pt, n_old, n_tmp, flag, off, 1.0 / voxel, h, sentinel, keysA, valsA, keysU,
                     valsU, status, d_opts, d_on);
  hipLaunchKernelGGL(k_check_increasing, dim3(nblk(n_v)), dim3(kB), 0, s, keysA, n_v, sentinel, status);
  {
    size_t tb = tb_sort;
    CK(sort_pairs(tmp, tb, keysU, keysU2, valsU, valsU2, (size_t)n_s, h.bits + 3, s));
    size_t tm = tb_merge;
    CK(rocprim::merge(tmp, tm, keysA, keysU2, keysM, valsA, valsU2, valsM, (size_t)n_v, (size_t)n_s, rocprim::less<uint64_t>(), s));
  }
  hipLaunchKernelGGL(k_heads, dim3(nblk(n_m)), dim3(kB), 0, s, keysM, n_m, sentinel, head, 2, blk);
  {
    const int rc = scan_flags_dev(head, ord, n_m, tmp, tb_scan, s, blk);
    if (rc != O3S_OK) return rc;
  }
  hipLaunchKernelGGL(k_vox_reduce, dim3(nblk(n_m)), dim3(kB), 0, s, keysM, valsM, head, ord, n_m, d_pts, d_nrm, (const int32_t*)nullptr, 1, 1, (int64_t)0,
                     d_opts, d_on, (int32_t*)nullptr, flag, off, n_tmp, 2);
  if (lazy) {
    hipLaunchKernelGGL(k_post_counts, dim3(1), dim3(64), 0, s, status, flag, off, n_tmp, head, ord, n_m, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, (int64_t)0, lazy->dev_out, lazy->mb_dev, lazy->seq);
    CK(hipGetLastError());
    if (issued) *issued = true;
    return O3S_OK;
  }
  const uint32_t seq = mailbox_next(pa);
  hipLaunchKernelGGL(k_post_counts, dim3(1), dim3(64), 0, s, status, flag, off, n_tmp, head, ord, n_m, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                     (int64_t)0, status + 4, pa.mb_dev, seq);
  CK(hipGetLastError());
  const int w = mailbox_wait(pa, seq, s);
  if (w < 0) return O3S_ERR_HIP;
  uint32_t r[4];
  if (w == 1) {
    for (int k = 0; k < 4; ++k) r[k] = __atomic_load_n(pa.mb + 2 + k, __ATOMIC_RELAXED);
  } else {
    uint32_t local[4];
    uint32_t* dst = pa.p ? pa.p : local;
    CK(hipMemcpyAsync(dst, status + 4, 16, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    for (int k = 0; k < 4; ++k) r[k] = dst[k];
  }
  if (r[0] != 0u) return O3S_OK;  // *ok stays false
  counts[0] = (int64_t)r[1];
  counts[1] = (int64_t)r[2];
  *ok = true;
  return O3S_OK;
}

}  // namespace o3s_cloud
}  // namespace
