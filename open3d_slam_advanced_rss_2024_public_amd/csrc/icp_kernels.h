// icp_kernels.h — hand-written gfx950 (CDNA4, wave64) kernels of the scan-to-map ICP iteration chain.
//
// One ICP iteration = 5 launches on one stream (no host round trip; a `done` flag in IcpState turns the remaining
// launches of a pre-recorded chain into no-ops).  At 100k points every kernel is bound by the LATENCY of its chain of
// dependent memory round trips (~1 us each), not by bandwidth, so each kernel is organised as a few wide batches of
// independent loads and nothing funnels through a single hot address:
//   k_match       transform by T_iter, exact 1-NN in the voxel grid (8 lanes per point), level-1 d2 histogram
//   k_classify    trim bin, normal-angle gate, decided-kept centroid sums, undecided pairs -> candidate segments
//   k_sel_finish  exact k-th smallest finite d2 (TrimmedDistOutlierFilter limit), means of the kept pairs
//   k_normal_eq   27 fp64 partial sums of G G^T / G h per block (centred in fp32 exactly like the reference)
//   k_solve       reduce partials, 6x6 solve, SE(3) step, T_iter update, stop rules
// Data layout in HBM: the reading is SoA fp32 (x[], y[], z[], nx[], ny[], nz[]) and is streamed with fully coalesced
// 4-byte loads; the reference is stored cell-sorted as 16-byte records {x,y,z,orig index} (+ a parallel {nx,ny,nz,0}
// array) so that every candidate / winner gather is one 16-byte load from one cache-line sector.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "icp_types.h"
#include "solve_device.h"

#pragma clang fp contract(off)

namespace o3s {
namespace kern {

constexpr int kBlock = 256;

// Phase timestamps for tuning (build with -DO3S_TS; read with o3s_debug_ts).  Not part of the product build.
#ifdef O3S_TS
__device__ unsigned long long g_ts[64];
#define O3S_TSTAMP(k)                                                     \
  do {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                            \
      __builtin_amdgcn_s_waitcnt(0);                                      \
      g_ts[k] = __builtin_amdgcn_s_memtime();                             \
    }                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
  } while (0)
#else
#define O3S_TSTAMP(k)
#endif
constexpr float kInfF = __builtin_huge_valf();

// ------------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------------
// Wave-wide sums.  The in-row steps use DPP lane permutes (VALU speed, no LDS crossbar round trip: a ds_bpermute-based
// __shfl tree of a double costs ~12 dependent LDS operations, which dominated the small kernels); the four 16-lane row
// sums are then combined through v_readlane.  Every lane returns the total.
//   quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = dpp_i32<CTRL>(__double2loint(v)), hi = dpp_i32<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // pairs
  v += dpp_f64<0x4E>(v);   // quads
  v += dpp_f64<0x141>(v);  // 8 lanes
  v += dpp_f64<0x140>(v);  // 16-lane rows
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  v += (uint32_t)dpp_i32<0xB1>((int)v);
  v += (uint32_t)dpp_i32<0x4E>((int)v);
  v += (uint32_t)dpp_i32<0x141>((int)v);
  v += (uint32_t)dpp_i32<0x140>((int)v);
  return ((uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16)) +
         ((uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48));
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Remap so that each XCD walks a contiguous
// slice of the (spatially sorted) reading: its L2 then sees one compact region of the reference grid.
__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
  const int per = (nblocks + 7) >> 3;
  const int t = (b & 7) * per + (b >> 3);
  return t;  // may be >= nblocks for the ragged tail: callers bounds-check the derived point index
}

// T (rows 0..2 of a column-major 4x4) applied to a point: ((T0 x + T1 y) + T2 z) + T3, fp32, no contraction
__device__ __forceinline__ float xf_row(const float* T, int r, float x, float y, float z) {
  float s = T[0 * 4 + r] * x;
  s = s + T[1 * 4 + r] * y;
  s = s + T[2 * 4 + r] * z;
  s = s + T[3 * 4 + r];
  return s;
}
__device__ __forceinline__ float rot_row(const float* T, int r, float x, float y, float z) {
  float s = T[0 * 4 + r] * x;
  s = s + T[1 * 4 + r] * y;
  s = s + T[2 * 4 + r] * z;
  return s;
}
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  float d = dx * dx;
  d = d + dy * dy;
  d = d + dz * dz;
  return d;
}

// ------------------------------------------------------------------------------------------------------------------
// reference preparation (initReference)
// ------------------------------------------------------------------------------------------------------------------
// pass 1: per-block fp64 sums and fp32 min/max of the raw reference (for the mean and the grid bounds)
// d_M (nullable): the number of points lives on the device (the patch a resident submap has just compacted: its count is the last
// word of the compaction's offsets) — the launch is sized for an upper bound and the host learns M from k_ref_stats_post's post
__global__ void __launch_bounds__(kBlock) k_ref_stats(const float4* __restrict__ xyzw, int64_t M, const uint32_t* __restrict__ d_M, double* __restrict__ part /*[grid][3]*/,
                                                      float* __restrict__ bb /*[grid][6]*/) {
  // a device-counted reference is summed by exactly the blocks, in exactly the partition, that the host would have launched for
  // its M points (min(1024, ceil(M / 256))): the mean's bits are a function of the cloud, not of the launch's upper bound
  int64_t G = (int64_t)gridDim.x;
  if (d_M) {
    M = (int64_t)*d_M;
    G = (M + kBlock - 1) / kBlock;
    G = G < 1024 ? G : 1024;
    if ((int64_t)blockIdx.x >= G) return;
  }
  double s0 = 0, s1 = 0, s2 = 0;
  float lo0 = kInfF, lo1 = kInfF, lo2 = kInfF, hi0 = -kInfF, hi1 = -kInfF, hi2 = -kInfF;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += G * kBlock) {
    const float4 p = xyzw[i];
    s0 += p.x;
    s1 += p.y;
    s2 += p.z;
    lo0 = fminf(lo0, p.x);
    lo1 = fminf(lo1, p.y);
    lo2 = fminf(lo2, p.z);
    hi0 = fmaxf(hi0, p.x);
    hi1 = fmaxf(hi1, p.y);
    hi2 = fmaxf(hi2, p.z);
  }
  __shared__ double sh[4][3];
  __shared__ float shb[4][6];
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo0 = fminf(lo0, __shfl_down(lo0, off, 64));
    lo1 = fminf(lo1, __shfl_down(lo1, off, 64));
    lo2 = fminf(lo2, __shfl_down(lo2, off, 64));
    hi0 = fmaxf(hi0, __shfl_down(hi0, off, 64));
    hi1 = fmaxf(hi1, __shfl_down(hi1, off, 64));
    hi2 = fmaxf(hi2, __shfl_down(hi2, off, 64));
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) {
    sh[w][0] = s0;
    sh[w][1] = s1;
    sh[w][2] = s2;
    shb[w][0] = lo0;
    shb[w][1] = lo1;
    shb[w][2] = lo2;
    shb[w][3] = hi0;
    shb[w][4] = hi1;
    shb[w][5] = hi2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int c = 0; c < 3; ++c) part[blockIdx.x * 3 + c] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
    for (int c = 0; c < 3; ++c) {
      bb[blockIdx.x * 6 + c] = fminf(fminf(shb[0][c], shb[1][c]), fminf(shb[2][c], shb[3][c]));
      bb[blockIdx.x * 6 + 3 + c] = fmaxf(fmaxf(shb[0][3 + c], shb[1][3 + c]), fmaxf(shb[2][3 + c], shb[3][3 + c]));
    }
  }
}

// single block: folds the per-block results of k_ref_stats in a fixed order — wave c sums component c: lane l adds the
// partials l, l + 64, ... in order, then the wave's DPP tree — and the bounds in any order, and posts mean / lo / hi (9
// words, mailbox[2..10]) and then the sequence number into host-coherent pinned memory.  (A first version summed the
// partials sequentially in one lane per component, as the host loop had: 11.8 us for 1 024 partials.)
__global__ void __launch_bounds__(kBlock) k_ref_stats_post(const double* __restrict__ part, const float* __restrict__ bb, int G, int64_t M,
                                                           const uint32_t* __restrict__ d_M, uint32_t* __restrict__ mailbox, uint32_t seq) {
  if (d_M) {
    M = (int64_t)*d_M;
    const int64_t Ge = (M + kBlock - 1) / kBlock;
    G = (int)(Ge < 1024 ? Ge : 1024);  // the blocks of k_ref_stats that took part
  }
  __shared__ float s_b[kBlock / 64][6];
  __shared__ float s_out[9];
  float lo[3] = {kInfF, kInfF, kInfF}, hi[3] = {-kInfF, -kInfF, -kInfF};
  for (int b = threadIdx.x; b < G; b += kBlock)
    for (int c = 0; c < 3; ++c) {
      lo[c] = fminf(lo[c], bb[(size_t)b * 6 + c]);
      hi[c] = fmaxf(hi[c], bb[(size_t)b * 6 + 3 + c]);
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    for (int c = 0; c < 3; ++c) {
      lo[c] = fminf(lo[c], __shfl_down(lo[c], off, 64));
      hi[c] = fmaxf(hi[c], __shfl_down(hi[c], off, 64));
    }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0)
    for (int c = 0; c < 3; ++c) {
      s_b[w][c] = lo[c];
      s_b[w][3 + c] = hi[c];
    }
  double sum = 0.0;
  if (w < 3) {
    for (int b = l; b < G; b += 64) sum += part[(size_t)b * 3 + w];
    sum = wave_sum(sum);
  }
  __syncthreads();
  if (w < 3 && l == 0) {
    float lo_c = s_b[0][w], hi_c = s_b[0][3 + w];
    for (int k = 1; k < kBlock / 64; ++k) {
      lo_c = fminf(lo_c, s_b[k][w]);
      hi_c = fmaxf(hi_c, s_b[k][3 + w]);
    }
    s_out[w] = (float)(sum / (double)M);
    s_out[3 + w] = lo_c;
    s_out[6 + w] = hi_c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 9; ++k) __hip_atomic_store(mailbox + 2 + k, __float_as_uint(s_out[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox + 11, (uint32_t)M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // the point count (device-counted references)
    __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv, int n) {
  int c = (int)floorf((v - o) * inv);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// pass 2: subtract the fp32 mean (LPM/ICP.cpp:320), assign a grid cell, count cell populations
__global__ void __launch_bounds__(kBlock) k_ref_assign(const float4* __restrict__ xyzw, int64_t M, float mx, float my, float mz,
                                                       GridParams g, uint32_t* __restrict__ cell_of, uint32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= M) return;
  const float4 p = xyzw[i];
  const float x = p.x - mx, y = p.y - my, z = p.z - mz;
  const int cx = cell_coord(x, g.ox, g.inv_cell, g.nx);
  const int cy = cell_coord(y, g.oy, g.inv_cell, g.ny);
  const int cz = cell_coord(z, g.oz, g.inv_cell, g.nz);
  const uint32_t lin = ((uint32_t)cz * (uint32_t)g.ny + (uint32_t)cy) * (uint32_t)g.nx + (uint32_t)cx;
  cell_of[i] = lin;
  atomicAdd(&counts[lin], 1u);
}

// pass 4 (after the scan): scatter into cell order.  Order inside a cell is arbitrary; the matcher's (d2, index)
// tie-break makes results independent of it.
__global__ void __launch_bounds__(kBlock) k_ref_scatter(const float4* __restrict__ xyzw, const float* __restrict__ normals /*3xM AoS or null*/,
                                                        int64_t M, float mx, float my, float mz, const uint32_t* __restrict__ cell_of,
                                                        const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ fill,
                                                        float4* __restrict__ ref_sorted, float4* __restrict__ refn_sorted,
                                                        int32_t* __restrict__ orig_to_sorted) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= M) return;
  const uint32_t c = cell_of[i];
  const uint32_t pos = cell_start[c] + atomicAdd(&fill[c], 1u);
  const float4 p = xyzw[i];
  ref_sorted[pos] = make_float4(p.x - mx, p.y - my, p.z - mz, __int_as_float((int)i));
  if (normals) refn_sorted[pos] = make_float4(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2], 0.f);
  if (orig_to_sorted) orig_to_sorted[i] = (int32_t)pos;  // null: the first-iteration index (no inverse map needed)
}

// ------------------------------------------------------------------------------------------------------------------
// exclusive scan of uint32 counts (3 launches: block sums, scan of block sums, add back).  out has n + 1 entries.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kScanItems = 8;                       // items per thread
constexpr int kScanTile = kBlock * kScanItems;      // 2048 items per block

// inclusive prefix sum across the wave at VALU speed: row_shr 1/2/4/8 inside the 16-lane rows (zeros shifted in), then
// row_bcast15 / row_bcast31 carry the row totals into the following rows.  (A __shfl_up tree is six dependent
// ds_bpermute round trips through the LDS crossbar: ~3 us of k_classify + k_sel_finish went there.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_zero_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += dpp_zero_u32<0x111, 0xF>(v);  // row_shr:1
  v += dpp_zero_u32<0x112, 0xF>(v);  // row_shr:2
  v += dpp_zero_u32<0x114, 0xF>(v);  // row_shr:4
  v += dpp_zero_u32<0x118, 0xF>(v);  // row_shr:8
  v += dpp_zero_u32<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_zero_u32<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* sh /*>= 17 words*/) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint32_t inc = wave_incl_scan_u32(v);
  __syncthreads();
  if (l == 63) sh[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int k = 0; k < nw; ++k) {
    const uint32_t s = sh[k];
    if (k < w) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

// two independent exclusive scans that share their barriers (the selection block scans candidate counts and level-2 bins)
__device__ __forceinline__ void block_excl_scan2(uint32_t va, uint32_t vb, uint32_t* total_a, uint32_t* total_b, uint32_t* ex_a, uint32_t* ex_b,
                                                 uint32_t* sh /*>= 34 words*/) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint32_t ia = wave_incl_scan_u32(va), ib = wave_incl_scan_u32(vb);
  __syncthreads();
  if (l == 63) {
    sh[w] = ia;
    sh[17 + w] = ib;
  }
  __syncthreads();
  uint32_t base_a = 0, tot_a = 0, base_b = 0, tot_b = 0;
  for (int k = 0; k < nw; ++k) {
    const uint32_t sa = sh[k], sb = sh[17 + k];
    if (k < w) {
      base_a += sa;
      base_b += sb;
    }
    tot_a += sa;
    tot_b += sb;
  }
  *total_a = tot_a;
  *total_b = tot_b;
  *ex_a = base_a + ia - va;
  *ex_b = base_b + ib - vb;
}

// ------------------------------------------------------------------------------------------------------------------
// Block-wide sums of K fp64 components per thread through LDS, in a FIXED order (thread order inside a 32-thread
// segment, then segment order): ~K stores + 32 loads + 31 adds per thread instead of K wave_sum trees (27 of those were
// 6 000 of k_normal_eq's 10 400 cycles).  s_a holds K x (NT/32) segments of 33 doubles (the pad keeps the 32-lane
// groups of ds_read_b64 conflict-free), s_b the K x (NT/32) segment sums; total(c) = sum over seg of s_b[c * SEG + seg].
// ------------------------------------------------------------------------------------------------------------------
template <int K, int NT>
struct BlockSum {
  static constexpr int SEG = NT / 32;
  static constexpr int kWordsA = K * SEG * 33;
  static constexpr int kWordsB = K * SEG;
  __device__ static __forceinline__ void run(const double (&v)[K], double* s_a, double* s_b) {
    const int t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < K; ++c) s_a[c * (SEG * 33) + (t >> 5) * 33 + (t & 31)] = v[c];
    __syncthreads();
    for (int u = t; u < K * SEG; u += NT) {
      const double* src = s_a + (u / SEG) * (SEG * 33) + (u % SEG) * 33;
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 32; ++j) s += src[j];
      s_b[u] = s;
    }
    __syncthreads();
  }
  // the same sums, in the same order, inside a block of MORE than NT threads: the first NT take part, every thread passes
  // the two barriers
  __device__ static __forceinline__ void run_first(const double (&v)[K], double* s_a, double* s_b) {
    const int t = threadIdx.x;
    if (t < NT) {
#pragma unroll
      for (int c = 0; c < K; ++c) s_a[c * (SEG * 33) + (t >> 5) * 33 + (t & 31)] = v[c];
    }
    __syncthreads();
    if (t < NT)
      for (int u = t; u < K * SEG; u += NT) {
        const double* src = s_a + (u / SEG) * (SEG * 33) + (u % SEG) * 33;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 32; ++j) s += src[j];
        s_b[u] = s;
      }
    __syncthreads();
  }
  __device__ static __forceinline__ double total(const double* s_b, int c) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < SEG; ++g) s += s_b[c * SEG + g];
    return s;
  }
};

// occupied (nullable): also counts the non-zero inputs (the occupied cells of the grid: the density probe of
// init_reference rides on the scan instead of a pass of its own); nonzero[block] is folded by k_scan_sums
__global__ void __launch_bounds__(kBlock) k_scan_block_sums(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ sums,
                                                            uint32_t* __restrict__ nonzero /*[grid] or null*/) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  uint32_t s = 0, z = 0;
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < n) {
      const uint32_t v = in[i];
      s += v;
      z += v ? 1u : 0u;
    }
  }
  s = wave_sum_u32(s);
  __shared__ uint32_t sh[4], shz[4];
  if (nonzero) z = wave_sum_u32(z);
  if ((threadIdx.x & 63) == 0) {
    sh[threadIdx.x >> 6] = s;
    shz[threadIdx.x >> 6] = z;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    if (nonzero) nonzero[blockIdx.x] = shz[0] + shz[1] + shz[2] + shz[3];
  }
}

// single block: exclusive scan of the block sums in place (serial over tiles of 1024).  With `nonzero` the per-block
// counts of k_scan_block_sums are summed too and posted — value, then sequence number — into host-coherent pinned memory
// the host is polling (mailbox; no copy, no stream synchronisation).
__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t* __restrict__ sums, int64_t nb, const uint32_t* __restrict__ nonzero,
                                                    uint32_t* __restrict__ mailbox, uint32_t seq) {
  __shared__ uint32_t sh[32];
  uint32_t carry = 0, zc = 0;
  for (int64_t base = 0; base < nb; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? sums[i] : 0u;
    if (nonzero && i < nb) zc += nonzero[i];
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, &tot, sh);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (nonzero) {
    zc = wave_sum_u32(zc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = zc;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int w = 0; w < 16; ++w) t += sh[w];
      __hip_atomic_store(mailbox, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// zero (nullable, = in): the inputs are cleared once read — the counting sorts re-use the count array as per-cell cursors
__global__ void __launch_bounds__(kBlock) k_scan_apply(const uint32_t* __restrict__ in, int64_t n, const uint32_t* __restrict__ sums,
                                                       uint32_t* __restrict__ out /* n + 1 */, uint32_t* __restrict__ zero) {
  __shared__ uint32_t sh[32];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  // A tile without a single count (the dense grid of a 150 k-point patch has ~5 M cells, 97 % of them empty): the scanned block sums
  // of this tile and the next are equal, every output is that value, and there is nothing to read or to clear.  (The last tile has no
  // successor to compare with and takes the general path.)
  if (blockIdx.x + 1 < gridDim.x) {
    const uint32_t here = sums[blockIdx.x];
    if (sums[blockIdx.x + 1] == here) {  // uniform over the block
#pragma unroll
      for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) out[base + k] = here;
      return;
    }
  }
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0u;
    s += v[k];
  }
  if (zero) {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
      if (base + k < n) zero[base + k] = 0u;
  }
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, &tot, sh) + sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
  if (base <= n - 1 && n - 1 < base + kScanItems) out[n] = ex;  // the thread owning the last item writes the total
}

// ------------------------------------------------------------------------------------------------------------------
// reading preparation (per compute call)
// ------------------------------------------------------------------------------------------------------------------
// transform by T0 = T_refIn_refMean^-1 * T_init once (LPM/ICP.cpp:373-375), count query cells for the spatial sort
// What a compute() has to reset before its first iteration, done by the first kernel of the call instead of one fill /
// copy command each: the histograms, the selection hand-off, the incumbents (no previous match: new reading / pose /
// reference) and the chain state (zeros, T_iter = I, limit = +inf — what init_state writes on the host for the module-level
// entry points — and the identity DifferentialTransformationChecker::init pushes, TransformationCheckersImpl.cpp:85-100:
// Quaternion(I) = (0, 0, 0, 1), zero translation, one entry in the ring).
constexpr int kMaxQTiles = 2048;  // the reading is sorted on at most 2^22 bins = 2 048 tiles of kScanTile
// The tile totals are kept in kTileReplicas copies, one per XCD (a block adds to copy blockIdx & 7: blocks b and b + 8 share an L2).
// With one copy every block that holds a point of a tile adds to the SAME word, and atomics onto one address from all eight XCDs
// are served one after the other at ~80 ns each (tools/native/atomic_bench.hip: 100 k points into 2 000 addresses take 4 us longer than
// into 50 000) — a floor's points fall into a few dozen tiles, hundreds of blocks hold some of each: that, not the per-point arrival
// rank, was 12 of k_read_prep's 21 us at C2.  Per XCD the adds stay in its own L2.  k_read_starts sums the copies.
constexpr int kTileReplicas = 8;
// Where a bin's count lives inside its tile of 2 048 words: bins that are neighbours along x — a floor's or a wall's points fall into
// runs of them — are kept 32 words (one 128-byte line) apart.  Device-scope atomics are served per cache LINE: with the counts in bin
// order the ~64 points of 32 adjacent bins queued on one line, and the one returning atomic per point was 7.6 us of k_read_prep's
// block (tools/r05_ts_prep.py); the transposed layout spreads them over 32 lines.  k_read_starts reads a tile through the same map.
__device__ __forceinline__ uint32_t qcount_slot(uint32_t lin) { return (lin & ~2047u) | ((lin & 63u) << 5) | ((lin >> 6) & 31u); }
static_assert(kScanTile == 2048, "qcount_slot transposes a 64 x 32 tile");
struct Mat16 {
  float v[16];
};
struct PrepInit {
  uint32_t* hist;   // null: nothing to reset (module-level callers that manage the state themselves)
  int hist_words;
  uint32_t* sel;
  int sel_words;
  float4* mq;
  IcpState* state;
  int seed_differential;
  uint32_t seq;  // sequence number of this compute(): echoed by every post of the chain (HostPost)
};

__global__ void __launch_bounds__(kBlock) k_read_prep(const float4* __restrict__ in_xyzw, const float* __restrict__ in_n /*3xN AoS or null*/,
                                                      int N, Mat16 T0, GridParams g,
                                                      float* __restrict__ tx, float* __restrict__ ty, float* __restrict__ tz,
                                                      float* __restrict__ tnx, float* __restrict__ tny, float* __restrict__ tnz,
                                                      uint32_t* __restrict__ cell_of, uint32_t* __restrict__ counts /*null: no sort*/,
                                                      int qf, int qnx, int qny, PrepInit init,
                                                      uint32_t* __restrict__ ticket /*[N]: arrival rank of the point inside its bin*/,
                                                      uint32_t* __restrict__ tile_cnt /*points per tile of kScanTile bins*/,
                                                      int32_t* __restrict__ perm /*no sort: slot -> original index = identity*/) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  O3S_TSTAMP(56);
  if (init.hist) {  // uniform
    const int stride = gridDim.x * kBlock;
#ifndef O3S_X_NOZERO
    for (int k = i; k < init.hist_words; k += stride) init.hist[k] = 0u;
    for (int k = i; k < init.sel_words; k += stride) init.sel[k] = 0u;
#endif
    if (blockIdx.x == 0) {
      constexpr int kWords = (int)(sizeof(IcpState) / 4);
      uint32_t* w = reinterpret_cast<uint32_t*>(init.state);
      for (int k = threadIdx.x; k < kWords; k += kBlock) w[k] = 0u;
      __syncthreads();
      if (threadIdx.x == 0) {
        IcpState* S = init.state;
        S->T_iter[0] = S->T_iter[5] = S->T_iter[10] = S->T_iter[15] = 1.f;
        S->limit = kInfF;
        S->call_seq = init.seq;
        S->t_prep = wall_clock64();
        if (init.seed_differential) {
          S->quat_ring[0][3] = 1.f;
          S->hist_total = 1;
        }
      }
    }
  }
  // tile totals: counted per block in LDS first and flushed by the first toucher of every tile — the points of a floor or a wall
  // fall into a handful of tiles, and one global atomic per point onto those few addresses serialised (45 us at C2)
  __shared__ uint32_t s_tile[kMaxQTiles];
  if (counts) {  // uniform
    for (int k = threadIdx.x; k < kMaxQTiles; k += kBlock) s_tile[k] = 0u;
    __syncthreads();
  }
  int my_tile = -1;
  O3S_TSTAMP(57);
  if (i < N) {
#ifndef O3S_X_NOMQ
    if (init.mq) init.mq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    const float4 p = in_xyzw[i];
    float T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = T0.v[k];
    const float x = xf_row(T, 0, p.x, p.y, p.z), y = xf_row(T, 1, p.x, p.y, p.z), z = xf_row(T, 2, p.x, p.y, p.z);
    tx[i] = x;
    ty[i] = y;
    tz[i] = z;
    if (in_n) {
      const float a = in_n[3 * i], b = in_n[3 * i + 1], c = in_n[3 * i + 2];
      tnx[i] = rot_row(T, 0, a, b, c);
      tny[i] = rot_row(T, 1, a, b, c);
      tnz[i] = rot_row(T, 2, a, b, c);
    }
    O3S_TSTAMP(58);
    if (counts) {
      const int cx = cell_coord(x, g.ox, g.inv_cell, g.nx);
      const int cy = cell_coord(y, g.oy, g.inv_cell, g.ny);
      const int cz = cell_coord(z, g.oz, g.inv_cell, g.nz);
      // the reading is ordered on a coarsened copy of the grid (qf x qf x qf cells per bin): enough for locality, and the
      // per-call histogram stays small however fine the matcher grid is
      const uint32_t lin = ((uint32_t)(cz / qf) * (uint32_t)qny + (uint32_t)(cy / qf)) * (uint32_t)qnx + (uint32_t)(cx / qf);
      cell_of[i] = lin;
      // the arrival rank is only a slot inside the bin (k_read_scatter); k_read_place turns it into the input rank.  The tile
      // totals spare the scan its block-sum launches: k_read_starts scans the <= 2 048 of them for itself
      ticket[i] = atomicAdd(&counts[qcount_slot(lin)], 1u);
      const int tile = (int)(lin / (uint32_t)kScanTile);
      if (atomicAdd(&s_tile[tile], 1u) == 0u) my_tile = tile;
    } else if (perm) {
      perm[i] = i;
    }
  }
  O3S_TSTAMP(59);
  if (counts) {  // uniform
    __syncthreads();
    if (my_tile >= 0) atomicAdd(&tile_cnt[(blockIdx.x & (kTileReplicas - 1)) * kMaxQTiles + my_tile], s_tile[my_tile]);
  }
  O3S_TSTAMP(60);
}

// Starts of the reading's bins (exclusive scan of the per-bin counts) in ONE launch: block b owns tile b (kScanTile bins); its
// base is the sum of the tile totals before it, which every block forms for itself from the <= kMaxQTiles totals k_read_prep
// counted.  The counts are cleared as they are read — the array is all zeros between calls (allocated zeroed, left zeroed: no
// per-call memset of up to 16 MB) — and the tile totals are cleared by k_read_scatter, once every block here has read them.
__global__ void __launch_bounds__(kBlock) k_read_starts(uint32_t* __restrict__ counts, int64_t n, const uint32_t* __restrict__ tile_cnt,
                                                        uint32_t* __restrict__ out /* n + 1 */) {
  __shared__ uint32_t sh[32];
  __shared__ uint32_t s_base[4];
  uint32_t before = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += kBlock) {
    uint32_t v[kTileReplicas];
#pragma unroll
    for (int r = 0; r < kTileReplicas; ++r) v[r] = tile_cnt[r * kMaxQTiles + t];
#pragma unroll
    for (int r = 0; r < kTileReplicas; ++r) before += v[r];
  }
  before = wave_sum_u32(before);
  if ((threadIdx.x & 63) == 0) s_base[threadIdx.x >> 6] = before;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? counts[qcount_slot((uint32_t)(base + k))] : 0u;
    s += v[k];
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < n && v[k]) counts[qcount_slot((uint32_t)(base + k))] = 0u;
  uint32_t tot;
  uint32_t ex = block_excl_scan(s, &tot, sh);  // its two barriers also publish s_base
  ex += (s_base[0] + s_base[1]) + (s_base[2] + s_base[3]);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
  if (base <= n - 1 && n - 1 < base + kScanItems) out[n] = ex;  // the thread owning the last bin writes the total
}

// The counting sort of the reading is STABLE: inside a bin the points keep their input order, so the order in which every
// later fp64 sum of the chain runs over them is a function of the input alone — not of the arrival order of an atomic.
//   k_read_prep     draws every point's arrival rank inside its bin with an integer atomic while it counts the bins;
//   k_read_scatter  records WHO sits in which slot of the bin (arrival order);
//   k_read_place    one lane per slot: the rank of its point's input index among the indices of its bin is its place
//                   inside the bin.  A bin holds a handful of points (the loop is over the bin's own segment, broadcast
//                   loads); a reading that piles up in one bin (a scan far outside the grid is clamped to a border cell)
//                   costs O(len^2) there and is still ordered.
// `reverse` (a test hook of the hooks build, O3S_SCATTER_ORDER=1) turns the arrival order inside every bin around: the placed
// reading must not change (tests/test_gpu_parity.py).
__global__ void __launch_bounds__(kBlock) k_read_scatter(int N, const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ start,
                                                         const uint32_t* __restrict__ ticket, int32_t* __restrict__ who /* slot -> original index, bin order arbitrary */,
                                                         uint32_t* __restrict__ tile_cnt, int n_tiles, int reverse) {
  if (blockIdx.x < kTileReplicas)  // k_read_starts is done with the tile totals: all zeros again for the next call
    for (int t = threadIdx.x; t < n_tiles; t += kBlock) tile_cnt[blockIdx.x * kMaxQTiles + t] = 0u;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const uint32_t c = cell_of[i];
  const uint32_t b = start[c];
  // `reverse` (hooks build): the bin's slots dealt back to front — another arrival order, the same placed reading
  const uint32_t tk = reverse ? (start[c + 1] - b) - 1u - ticket[i] : ticket[i];
  who[b + tk] = i;
}

__global__ void __launch_bounds__(kBlock) k_read_place(int N, const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ start,
                                                       const int32_t* __restrict__ who, const float* __restrict__ tx, const float* __restrict__ ty,
                                                       const float* __restrict__ tz, const float* __restrict__ tnx, const float* __restrict__ tny,
                                                       const float* __restrict__ tnz, int has_n, float* __restrict__ rx, float* __restrict__ ry,
                                                       float* __restrict__ rz, float* __restrict__ rnx, float* __restrict__ rny,
                                                       float* __restrict__ rnz, int32_t* __restrict__ perm /* sorted slot -> original index */) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= N) return;
  const int i = who[s];
  const uint32_t c = cell_of[i];
  const uint32_t b = start[c], e = start[c + 1];
  uint32_t rank = 0;
  for (uint32_t j = b; j < e; ++j) rank += who[j] < i ? 1u : 0u;
  const uint32_t pos = b + rank;
  rx[pos] = tx[i];
  ry[pos] = ty[i];
  rz[pos] = tz[i];
  if (has_n) {
    rnx[pos] = tnx[i];
    rny[pos] = tny[i];
    rnz[pos] = tnz[i];
  }
  perm[pos] = i;
}

__global__ void __launch_bounds__(kBlock) k_iota(int N, int32_t* __restrict__ perm) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < N) perm[i] = i;
}

// ------------------------------------------------------------------------------------------------------------------
// Header broadcast: the first 32 words of IcpState (T_iter, done, status, limit, means, |K|) arrive with ONE coalesced
// 128-byte load per wave and are handed out with v_readlane — one memory round trip instead of a chain of scalar loads.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hdr_load(const IcpState* __restrict__ st) {
  return reinterpret_cast<const float*>(st)[threadIdx.x & 31];
}
__device__ __forceinline__ float hdr_f(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
__device__ __forceinline__ int hdr_i(float v, int k) { return __builtin_amdgcn_readlane(__float_as_int(v), k); }
enum { H_ITER = 16, H_DONE = 17, H_STATUS = 18, H_LIMIT = 19, H_NFIN = 20, H_MP = 21, H_MQ = 24, H_KEPT = 27 };

// ------------------------------------------------------------------------------------------------------------------
// Matcher::findClosests fused with the step transform (LPM/ICP.cpp:401-413, LPM/MatchersImpl.cpp:117-132): helpers shared
// by the matcher kernels.  (Round 1's k_match — 4 / 8 lanes per query over nine masked row slots — was removed once
// k_match2 had replaced it in every path; DESIGN.md section 6 keeps its history.)
// ------------------------------------------------------------------------------------------------------------------
struct __attribute__((aligned(4))) U32Pair {  // two adjacent uint32 words at any 4-byte aligned address: one 8-byte load
  uint32_t x, y;
};

__device__ __forceinline__ float cell_gap(int d, float l, float cell, float margin) {
  const float up = (float)d * cell - l, down = l + (float)(-d - 1) * cell;
  const float gap = (d > 0 ? up : (d < 0 ? down : 0.f)) - margin;
  return fmaxf(gap, 0.f);
}

// value of lane (group base + S) for every lane of a 4-lane group: a quad_perm DPP broadcast (VALU, no LDS round trip)
template <int S>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, S * 0x55, 0xF, 0xF, false);
}

// ------------------------------------------------------------------------------------------------------------------
// k_match2 — the matcher: exact 1-NN (ids and squared distances bit-identical to the brute force), with the work per
// query cut by two things round 1's kernel did not have.
//   incumbent   the reference point this query matched in the PREVIOUS iteration arrives with the query itself (a
//               coalesced 16-byte record, `mq`, written by the previous launch), so its distance under the new pose is
//               known before anything is searched.  It is only ever used as a pruning BOUND: a row of cells (and,
//               inside a row, its left / right cell) whose conservative lower bound exceeds it is not fetched at all.
//               The search stays exact — every point at most as far as an actual reference point lies in an unpruned
//               cell, the incumbent included — and ties still go to the lowest original index.
//   compaction  the candidates that survive (c-bar ~ 5 instead of 21 at C2) are dealt to the 4 lanes of the query as ONE
//               flat list: 9 row lengths -> prefix sums broadcast inside the quad by DPP -> lane `sub` takes entries
//               sub, sub + 4, ...  A round costs one 16-byte load and ~35 VALU per lane and a wave runs as many rounds as
//               its longest query needs, instead of 18 masked slots per lane for every query.
// Three dependent round trips as before (points + incumbent | row headers | candidates), ~half the instructions and a
// register footprint that lets every wave of a 100k-point reading be resident at once.
// ------------------------------------------------------------------------------------------------------------------
struct Own {  // a lane's best among ITS candidates, with the matched point itself (handed to the next iteration)
  float d;
  int idx;
  int pos;
  float qx, qy, qz;
};

__device__ __forceinline__ void own_take(Own& b, float d, const float4& q, int j, float lim, bool enable) {
  const int qi = __float_as_int(q.w);
  const bool c = enable & (d <= lim) & ((d < b.d) | ((d == b.d) & (qi < b.idx)));
  b.d = c ? d : b.d;
  b.idx = c ? qi : b.idx;
  b.pos = c ? j : b.pos;
  b.qx = c ? q.x : b.qx;
  b.qy = c ? q.y : b.qy;
  b.qz = c ? q.z : b.qz;
}

// cell of a (transformed) query and its offset inside the cell; far-away queries are clamped so that the int conversion
// cannot overflow (the bounds derived from the clamped values stay lower bounds)
struct CellGeom {
  int cx, cy, cz;
  float lx, ly, lz;
};
__device__ __forceinline__ CellGeom cell_geom(float sx, float sy, float sz, const GridParams& g) {
  CellGeom c;
  const float big = 1.0e9f;
  c.cx = (int)floorf(fminf(fmaxf((sx - g.ox) * g.inv_cell, -big), big));
  c.cy = (int)floorf(fminf(fmaxf((sy - g.oy) * g.inv_cell, -big), big));
  c.cz = (int)floorf(fminf(fmaxf((sz - g.oz) * g.inv_cell, -big), big));
  c.lx = fminf(fmaxf((sx - g.ox) - (float)c.cx * g.cell, 0.f), g.cell);
  c.ly = fminf(fmaxf((sy - g.oy) - (float)c.cy * g.cell, 0.f), g.cell);
  c.lz = fminf(fmaxf((sz - g.oz) - (float)c.cz * g.cell, 0.f), g.cell);
  return c;
}
// For a query outside the grid's box every reference point is at least gap_b away along each axis b.  A point of ring r is
// (r - 1) cells away along the ring's axis a (which already contains gap_a) AND gap_b away along the other two, so the
// ring's squared lower bound may be raised by the two smaller squared gaps: sum(gap^2) - max(gap^2).  Without it a query
// far outside the map with an unbounded maxDist walks every ring that intersects the grid — the whole grid — because
// (r - 1) * cell alone never exceeds its true distance.
__device__ __forceinline__ float outside_extra2(float sx, float sy, float sz, const GridParams& g) {
  const float px = sx - g.ox, py = sy - g.oy, pz = sz - g.oz;
  const float gx = fmaxf(fmaxf(-px, px - (float)g.nx * g.cell) - g.margin, 0.f);
  const float gy = fmaxf(fmaxf(-py, py - (float)g.ny * g.cell) - g.margin, 0.f);
  const float gz = fmaxf(fmaxf(-pz, pz - (float)g.nz * g.cell) - g.margin, 0.f);
  const float mx = fmaxf(gx, fmaxf(gy, gz));
  const float e = (gx * gx + gy * gy + gz * gz) - mx * mx;
  return e > 0.f ? e * 0.999f : 0.f;  // a hair below: the three squares are rounded
}
// first ring (>= 2) that can hold reference points for this query, the last one, and the query's distance to its cell walls
__device__ __forceinline__ void ring_range(const CellGeom& c, const GridParams& g, int& r_first, int& r_last, float& m) {
  m = fminf(fminf(fminf(c.lx, g.cell - c.lx), fminf(c.ly, g.cell - c.ly)), fminf(c.lz, g.cell - c.lz));
  int r0 = 0;
  r0 = max(r0, max(-c.cx, c.cx - (g.nx - 1)));
  r0 = max(r0, max(-c.cy, c.cy - (g.ny - 1)));
  r0 = max(r0, max(-c.cz, c.cz - (g.nz - 1)));
  r_last = max(max(c.cx, g.nx - 1 - c.cx), max(max(c.cy, g.ny - 1 - c.cy), max(c.cz, g.nz - 1 - c.cz)));
  r_first = max(2, r0);
}
// squared lower bound of ring r for a query at wall distance m (0 when the ring may touch the query's own cell)
__device__ __forceinline__ float ring_lb2(int r, float m, const GridParams& g) {
  const float lb = (float)(r - 1) * g.cell + m - g.margin;
  return lb > 0.f ? lb * lb : 0.f;
}

// value of lane (group base + S) for every lane of a G-lane group (G = 1, 2, 4): quad_perm DPP, no LDS round trip
template <int G, int S>
__device__ __forceinline__ uint32_t group_bcast(uint32_t v) {
  if (G == 1) return v;
  if (G == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, S == 0 ? 0xA0 : 0xF5, 0xF, 0xF, false);  // [0,0,2,2] / [1,1,3,3]
  return quad_bcast<S>(v);
}
template <int G>
__device__ __forceinline__ uint32_t group_bcast_dyn(uint32_t v, int s) {  // s is a compile-time constant after unrolling
  switch (s & (G - 1)) {
    case 0: return group_bcast<G, 0>(v);
    case 1: return group_bcast<G, 1 % G>(v);
    case 2: return group_bcast<G, 2 % G>(v);
    default: return group_bcast<G, 3 % G>(v);
  }
}
// (d, idx) minimum over the G lanes of a group; every lane ends with the group's winner
template <int G>
__device__ __forceinline__ void group_min_di(float d, int idx, float& gd, int& gi) {
  if (G >= 2) {
    const float od = __int_as_float(dpp_i32<0xB1>(__float_as_int(d)));
    const int oi = dpp_i32<0xB1>(idx);
    const bool c = (od < d) | ((od == d) & (oi < idx));
    d = c ? od : d;
    idx = c ? oi : idx;
  }
  if (G >= 4) {
    const float od = __int_as_float(dpp_i32<0x4E>(__float_as_int(d)));
    const int oi = dpp_i32<0x4E>(idx);
    const bool c = (od < d) | ((od == d) & (oi < idx));
    d = c ? od : d;
    idx = c ? oi : idx;
  }
  gd = d;
  gi = idx;
}

// ------------------------------------------------------------------------------------------------------------------
// Far search for a finite maxDist ("row-disc" search) — queries whose neighbour lies beyond the 3 x 3 x 3 cells around them:
// half the queries of a first iteration (no incumbents, pose off by more than a cell), every unmatched point.
//   The ring search walks 3-D shells of cells: (2r+1)^2 rows per shell r, two header loads per row, three rows per round
//   trip and their candidates range after range — the first iteration of a call cost 6x (C2) to 78x (C4) a converged one.
//   A row of cells (fixed y, z) is CONTIGUOUS in the cell-sorted reference, so everything a query may need from a row is
//   one range.  The far search therefore walks 2-D rings in (y, z) only — ring rho = the rows at Chebyshev distance rho from
//   the query's row, 8 rho of them — and takes from every row the x window the bound still leaves:
//       g2 = gap_y^2 + gap_z^2 > bound          -> the row is closed (no load at all: open rows are compacted first)
//       else  cells with gap_x^2 <= bound - g2  -> ONE range [start(x_lo), start(x_hi + 1))
//   Open ranges are queued and fetched in batches: 2 Q header words in one round trip, then the points of all Q ranges as
//   one flat list.  A disc of radius R cells has ~ pi R^2 rows where the shells have ~ 4 R^3 / 3 * 3 rows, and the bound
//   — tightened by every batch — closes most of them before they are loaded.
//   Exactness: a row or a cell is skipped only when its conservative lower bound (cell_gap: margin taken off) is strictly
//   above a bound that is never below the squared distance of an existing reference point within maxDist (or maxDist^2
//   itself); rings end when (rho - 1) cell + (distance to the own cell's y / z walls) exceeds the bound.  The three cells
//   around the query in the nine central rows are left out: the stage before examined every one of them that the bound
//   left open.  ids and squared distances stay bit-identical to the brute force (test_find_closests_bit_exact).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kFarMaxCells = 4096;  // the host selects the ring search when maxDist reaches beyond this many cells (or is unbounded)

// Bound sharing between neighbouring queries of a wave.  The reading is sorted by grid cell, so the queries a wave holds lie
// next to each other, and ANY reference point is an upper bound of a query's nearest-neighbour distance: a lane takes the best
// point its neighbours (the next two queries on either side) have found so far, measures its own distance to it and tightens
// its pruning bound — never its match: the bound only decides which rows, cells and records are skipped, and it is never below
// the distance of an existing reference point within maxDist, which is all the exactness argument asks for.  In a first
// iteration (no incumbents) this is what spares a query the full-radius walk its neighbour has just made.
template <int G>
__device__ __forceinline__ float neighbour_bound(const Own& b, float sx, float sy, float sz, float lim, float bound) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int off = (k & 1 ? -1 : 1) * (k < 2 ? G : 2 * G);
    const int src = min(max(lane + off, 0), 63);
    const float nd = __shfl(b.d, src, 64), nx = __shfl(b.qx, src, 64), ny = __shfl(b.qy, src, 64), nz = __shfl(b.qz, src, 64);
    const float dn = dist2(sx, sy, sz, nx, ny, nz);
    bound = (nd < kInfF && dn <= lim) ? fminf(bound, dn) : bound;  // NaN / no find: unchanged
  }
  return bound;
}

// G lanes per query (4 at C2: 6 k waves fill the chip; 1 for large readings, where the per-query set-up that every lane
// of a group repeats is the larger part of the work); UN candidate rounds per batch of loads.
// RCB: ring candidates per round trip (2: the kernel stays at <= 72 VGPRs; 8 was measured and bought nothing).
// FAR: queries that the 3 x 3 x 3 cells leave open go through the occupancy words (finite maxDist) instead of the ring search.
template <bool STATS, int G, int UN, int RCB, bool FAR>
#ifndef O3S_FAR_WAVES
#define O3S_FAR_WAVES 5  // waves per SIMD the far variant is compiled for: 6 spills (80 VGPRs + 8-12 B of scratch), 5 does not and is as fast
#endif
#ifndef O3S_SHARE_BOUNDS
#define O3S_SHARE_BOUNDS 0  // 1: neighbouring queries of a wave share what they have found as pruning bounds (neighbour_bound) — measured in round 4: slower (C2 first iteration 40.7 -> 45.1 us, C4 unchanged)
#endif
#ifndef O3S_FAR_Q
#define O3S_FAR_Q 6  // ranges per batch of the row-disc search (4: 53.9 us, 6: 50.6 us, 8: no better, for the first iteration at C2)
#endif
__global__ void __launch_bounds__(kBlock, FAR ? O3S_FAR_WAVES : 7) k_match2(const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
                                                      int N, const float4* __restrict__ ref, const uint32_t* __restrict__ cell_start,
                                                      GridParams g, IcpState* __restrict__ st,
                                                      int32_t* __restrict__ pos_out, float* __restrict__ d2_out, float4* __restrict__ mq,
                                                      uint32_t* __restrict__ hist_rep,
                                                      const float4* __restrict__ refn /*reference normals in slot order (nullable)*/,
                                                      float4* __restrict__ mn /*out (nullable): the matched normal of every query*/,
                                                      const float* __restrict__ rnx, const float* __restrict__ rny,
                                                      const float* __restrict__ rnz /*FAR: the reading's normals (nullable): the seed probe's direction*/,
                                                      int rep_mask /*level-1 replicas - 1 (15; fewer in the sharded mode, where they travel)*/
                                                      O3S_DBG_PARAM /*hooks build only: timing experiments (o3s_icp_profile_match)*/) {
  __shared__ uint32_t s_hist[kHistBins];
  constexpr int TQ = kBlock / G;        // queries per block: ONE tile per block (straight-line code, nothing kept alive across tiles)
  constexpr int NK = (9 + G - 1) / G;   // rows of the 3x3x3 block a lane owns: t = sub, sub + G, ...
  const int sub = threadIdx.x & (G - 1);
  const int qib = threadIdx.x / G;
  // XCD-aware: gridDim.x is a multiple of 8; blocks b, b + 8, ... (one XCD) walk a contiguous range of the sorted reading
  const int tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int i = tile * TQ + qib;
  const bool valid = i < N;
  // the state header, the query and its incumbent travel in the same round trip
  const float hv = hdr_load(st);
  float px = 0.f, py = 0.f, pz = 0.f;
  float4 inc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) {
    px = rx[i];
    py = ry[i];
    pz = rz[i];
    inc = mq[i];
  }
  for (int k = threadIdx.x; k < kHistBins; k += kBlock) s_hist[k] = 0u;
  if (hdr_i(hv, H_DONE)) return;
  // the level-2 histogram (right behind the level-1 replicas) is filled by k_classify after this kernel and read by the
  // selection after that; block 0 clears it here because the fused k_sel_ne cannot (its other blocks may still be reading)
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < 1024; k += kBlock) hist_rep[(size_t)kHistReplicas * kHistBins + k] = 0u;
    if (threadIdx.x == 0 && hdr_i(hv, H_ITER) == 0) st->t_begin = wall_clock64();  // the chain's own clock (stats.gpu_ms): no HIP events on the call path
  }
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = hdr_f(hv, k);
  __syncthreads();
  const float lim = g.max_r2;
  unsigned long long n_cand = 0, n_rows = 0;

  const float sx = xf_row(T, 0, px, py, pz), sy = xf_row(T, 1, px, py, pz), sz = xf_row(T, 2, px, py, pz);
  Own b{kInfF, 0x7fffffff, -1, 0.f, 0.f, 0.f};
  float gd = kInfF;  // the group's best so far
  int gi = 0x7fffffff;
  // a NaN or infinite query (a NaN in the reading, an overflow under the pose) has no neighbour — every distance test fails, in
  // libnabo's walk as in the brute force — and is settled here: its cell and offsets are not numbers, and with an unbounded
  // maxDist the ring search would walk every ring of the grid for it
  bool active = valid && ((fabsf(sx) + fabsf(sy)) + fabsf(sz) < kInfF);
  float bound = lim;
  if (active) {
    // ---- pruning bound: the previous correspondence under the new pose (any reference point is an upper bound) ----
    const float di = dist2(sx, sy, sz, inc.x, inc.y, inc.z);
    bound = ((inc.w != 0.f) & (di <= lim)) ? di : lim;  // NaN -> lim
  }
  // ---- seed probe (first iteration of a call, unmatched points: no incumbent).  Without a bound the far search opens every row
  //      and every cell window at full radius until it finds something — at C4 that is 430 distance tests per query where 60
  //      would do.  A query that carries a normal looks along it: the cells the line  s +- k cell n  passes through, k = 1 ..
  //      maxDist / cell (their headers travel in ONE round trip, dealt over the lanes of the group), the nearest occupied one's
  //      first points in a second.  Whatever it finds is an existing reference point, i.e. a valid upper bound of the
  //      neighbour distance — a BOUND only, like the incumbent: the search that follows stays exact whatever the probe hits or
  //      misses (a query 0.3 m above a floor hits the cell under it; a bad normal just finds nothing).
  //      Only where the far search has many rings to walk (maxDist >= 5 cells: dense maps, C4): at C2's three cells the probe
  //      costs what it saves (first iteration 40.7 us without, 42.1 us with).
  if (FAR && rnx != nullptr && lim * g.inv_cell * g.inv_cell >= 25.f && __any(valid && inc.w == 0.f)) {  // uniform
    constexpr int kProbeSteps = 12;  // per direction; more cells than that to maxDist: the probe stops short (still a valid bound)
    constexpr int NP = (2 * kProbeSteps + G - 1) / G;  // probe cells per lane
    const bool want = valid && inc.w == 0.f;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (want) {
      const float a = rnx[i], b_ = rny[i], c_ = rnz[i];
      nx = rot_row(T, 0, a, b_, c_);
      ny = rot_row(T, 1, a, b_, c_);
      nz = rot_row(T, 2, a, b_, c_);
    }
    const int n_steps = min(kProbeSteps, (int)(__builtin_sqrtf(lim) * g.inv_cell) + 1);  // lim = +inf never comes here (FAR)
    U32Pair hd[NP];
    int kk_[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int e = sub + u * G;              // 0 .. 2 kProbeSteps - 1: step e / 2 + 1, direction by parity
      const int st_ = (e >> 1) + 1;
      const float t_ = (float)st_ * g.cell * ((e & 1) ? -1.f : 1.f);
      const float qx_ = sx + t_ * nx, qy_ = sy + t_ * ny, qz_ = sz + t_ * nz;
      const int cx_ = (int)floorf((qx_ - g.ox) * g.inv_cell), cy_ = (int)floorf((qy_ - g.oy) * g.inv_cell), cz_ = (int)floorf((qz_ - g.oz) * g.inv_cell);
      const bool in = want & (st_ <= n_steps) & ((unsigned)cx_ < (unsigned)g.nx) & ((unsigned)cy_ < (unsigned)g.ny) & ((unsigned)cz_ < (unsigned)g.nz);
      const uint32_t off = in ? (((uint32_t)cz_ * (uint32_t)g.ny + (uint32_t)cy_) * (uint32_t)g.nx + (uint32_t)cx_) : 0u;
      hd[u] = *reinterpret_cast<const U32Pair*>(cell_start + off);
      kk_[u] = in ? st_ : 0x7fffffff;
    }
    uint32_t pj = 0u, pn = 0u;
    int pk = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < NP; ++u) {  // this lane's nearest occupied probe cell
      const bool better = (hd[u].y > hd[u].x) & (kk_[u] < pk);
      pj = better ? hd[u].x : pj;
      pn = better ? hd[u].y - hd[u].x : pn;
      pk = better ? kk_[u] : pk;
    }
    float ds = kInfF;
    {
      float4 sp[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) sp[v] = ref[pj + ((uint32_t)v < pn ? (uint32_t)v : 0u)];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float d_ = dist2(sx, sy, sz, sp[v].x, sp[v].y, sp[v].z);
        ds = ((uint32_t)v < pn && d_ <= lim) ? fminf(ds, d_) : ds;
      }
    }
    if (G >= 2) ds = fminf(ds, __int_as_float(dpp_i32<0xB1>(__float_as_int(ds))));
    if (G >= 4) ds = fminf(ds, __int_as_float(dpp_i32<0x4E>(__float_as_int(ds))));
    bound = fminf(bound, ds);
  }
  if (active) {
    const CellGeom c = cell_geom(sx, sy, sz, g);
    // squared gaps to the neighbour cells on each axis (the query's own cell: 0), margin already taken off
    const float gxn = fmaxf(c.lx - g.margin, 0.f), gxp = fmaxf((g.cell - c.lx) - g.margin, 0.f);
    const float gyn = fmaxf(c.ly - g.margin, 0.f), gyp = fmaxf((g.cell - c.ly) - g.margin, 0.f);
    const float gzn = fmaxf(c.lz - g.margin, 0.f), gzp = fmaxf((g.cell - c.lz) - g.margin, 0.f);
    const float gxn2 = gxn * gxn, gxp2 = gxp * gxp, gyn2 = gyn * gyn, gyp2 = gyp * gyp, gzn2 = gzn * gzn, gzp2 = gzp * gzp;
    // ---- round trip 2: headers of the rows of the 3x3x3 block that the bound leaves open.  A row's header is one
    //      16-byte load starting at its first open cell `lo` (cell_start[lo .. lo + 3]: the row's begin is word 0, its end
    //      word 1..3 for 1..3 open cells).  Branch-free: a closed row reads offset 0. ----
    uint32_t jb[NK], len[NK];
    {
      uint4 w[NK];
      int span[NK];
      uint32_t inm[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int t = sub + k * G, tt = t < 9 ? t : 4;
        const int dz = tt / 3 - 1, dy = tt % 3 - 1;
        const int z = c.cz + dz, y = c.cy + dy;
        const float gz2 = dz < 0 ? gzn2 : (dz > 0 ? gzp2 : 0.f);
        const float gy2 = dy < 0 ? gyn2 : (dy > 0 ? gyp2 : 0.f);
        const float g2 = gz2 + gy2;
        const float rem = bound - g2;  // what the x direction may still spend (+inf stays +inf)
        const int lo = max(c.cx - (int)!(gxn2 > rem), 0);
        const int hi = min(c.cx + (int)!(gxp2 > rem), g.nx - 1);
        const bool in = (t < 9) & (lo <= hi) & ((unsigned)y < (unsigned)g.ny) & ((unsigned)z < (unsigned)g.nz) & !(g2 > bound) & !O3S_DBG(4);
        inm[k] = (uint32_t) - (int)in;
        span[k] = hi - lo;  // 0..2 open cells beyond the first
        const uint32_t off = (((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx + (uint32_t)lo) & inm[k];
        w[k] = *reinterpret_cast<const uint4*>(cell_start + off);  // global_load_dwordx4 needs 4-byte alignment only
      }
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const uint32_t ve = span[k] == 0 ? w[k].y : (span[k] == 1 ? w[k].z : w[k].w);
        jb[k] = w[k].x & inm[k];
        len[k] = (ve - w[k].x) & inm[k];
        if (STATS) {
          n_rows += len[k] ? 1 : 0;
          n_cand += (unsigned long long)len[k];
        }
      }
    }
    // ---- the query's flat candidate list: P[t] = candidates in rows before t, D[t] = jb[t] - P[t], every lane of the
    //      group holds all nine (quad_perm broadcasts from the owner lanes) ----
    uint32_t P[9], D[9];
    uint32_t total = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const uint32_t l_t = group_bcast_dyn<G>(len[t / G], t % G);
      const uint32_t j_t = group_bcast_dyn<G>(jb[t / G], t % G);
      P[t] = total;
      D[t] = j_t - total;
      total += l_t;
    }
    if (O3S_DBG(2)) total = 0;
    // ---- round trip 3: rounds of G candidates per query (one per lane), UN rounds per batch of loads ----
    for (uint32_t f0 = (uint32_t)sub; __any(f0 < total); f0 += (uint32_t)(G * UN)) {
      float4 qv[UN];
      uint32_t jj[UN];
      bool ok[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const uint32_t f = f0 + (uint32_t)(u * G);
        ok[u] = f < total;
        uint32_t dsel = D[0];
#pragma unroll
        for (int t = 1; t < 9; ++t) dsel = (P[t] <= f) ? D[t] : dsel;
        jj[u] = ok[u] ? f + dsel : 0u;
        qv[u] = ref[jj[u]];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) own_take(b, dist2(sx, sy, sz, qv[u].x, qv[u].y, qv[u].z), qv[u], (int)jj[u], lim, ok[u]);
    }
  }
  group_min_di<G>(b.d, b.idx, gd, gi);
  // what the queries next door found in their 27 cells — only in waves that hold a query without an incumbent (a first iteration,
  // unmatched points): with incumbents everywhere the bound is tight already and the converged path stays as short as it was
  if (FAR && O3S_SHARE_BOUNDS && __any(valid && inc.w == 0.f)) {
    bound = neighbour_bound<G>(b, sx, sy, sz, lim, bound);
    // every lane's value is a valid bound of the group's query: all lanes of a group go on with the smallest, so that whether the
    // query enters the far search is one decision per query
    if (G >= 2) bound = fminf(bound, __int_as_float(dpp_i32<0xB1>(__float_as_int(bound))));
    if (G >= 4) bound = fminf(bound, __int_as_float(dpp_i32<0x4E>(__float_as_int(bound))));
  }
  // ---- rings r >= 2: only queries whose bound / best reaches beyond the 3x3x3 block (far prior, no incumbent).  Ring 2
  //      lies at least cell - margin away, which settles almost every query without looking at its geometry again; the
  //      exact ring bounds are only worked out (from the query, not kept alive across the common path) when some lane of
  //      the wave is still open. ----
  {
    const float q = g.cell - g.margin;
    active = active && !(q * q > fminf(gd, bound)) && !O3S_DBG(16);
  }
  if (FAR) {
    if (__any(active)) {
      constexpr int Q = O3S_FAR_Q;  // ranges fetched per batch of loads
      const CellGeom c = cell_geom(sx, sy, sz, g);
      const float m_yz = fminf(fminf(c.ly, g.cell - c.ly), fminf(c.lz, g.cell - c.lz));
      // rings that hold rows of the grid at all (a query outside the grid in y or z starts further out)
      const int rho0 = max(max(0, max(-c.cy, c.cy - (g.ny - 1))), max(-c.cz, c.cz - (g.nz - 1)));
      const int rho_max = max(max(c.cy, g.ny - 1 - c.cy), max(c.cz, g.nz - 1 - c.cz));
      int rho = rho0;
      float best = fminf(gd, bound);  // the pruning bound: never below the squared distance of an existing point within maxDist
      auto ring_lb2_yz = [&](int r) {
        const float lb = (float)(r - 1) * g.cell + m_yz - g.margin;
        return (r >= 2 && lb > 0.f) ? lb * lb : 0.f;
      };
      if (rho > rho_max || ring_lb2_yz(rho) > best) active = false;
      uint32_t qa[Q], qb[Q];  // queued cell ranges [qa, qb) of the cell-sorted reference
#pragma unroll
      for (int k = 0; k < Q; ++k) qa[k] = qb[k] = 0u;
      while (__any(active)) {
        if (active) {
          const int n_rows_r = rho == 0 ? 1 : 8 * rho;
          int t = sub;  // this lane's next row of the ring: t, t + G, ...
          for (;;) {
            // ---- gather open rows until the queue is (nearly) full: closed rows cost no load
            int nq = 0;
            while (nq <= Q - 2 && t < n_rows_r) {
              int dy, dz;
              {
                const int s1 = 2 * rho + 1;
                if (t < s1) {
                  dz = -rho;
                  dy = t - rho;
                } else if (t < 2 * s1) {
                  dz = rho;
                  dy = t - s1 - rho;
                } else {
                  const int v = t - 2 * s1;
                  dy = (v & 1) ? rho : -rho;
                  dz = (v >> 1) - (rho - 1);
                }
              }
              t += G;
              const int y = c.cy + dy, z = c.cz + dz;
              if (((unsigned)y >= (unsigned)g.ny) | ((unsigned)z >= (unsigned)g.nz)) continue;
              const float gy = cell_gap(dy, c.ly, g.cell, g.margin), gz = cell_gap(dz, c.lz, g.cell, g.margin);
              const float g2 = gy * gy + gz * gz;
              const float bl = fminf(best, b.d);
              if (g2 > bl) continue;
              if (O3S_DBG(64)) continue;  // hooks build, timing only: the enumeration of the rings' rows and their gap test alone
              // x window: gap(dx) <= s  <=>  dx <= (s + lx) / cell  and  -dx <= (s - lx) / cell + 1, s = sqrt(rest) + margin; the float
              // estimate may fall one cell short, so the next cell out is tested with the bound's own comparison
              const float rest = bl - g2;
              const float s_ = __builtin_sqrtf(rest) + g.margin;
              int kr = min(max((int)floorf((s_ + c.lx) * g.inv_cell), -(1 << 28)), 1 << 28), kl = min(max((int)floorf((s_ - c.lx) * g.inv_cell) + 1, -(1 << 28)), 1 << 28);
              {
                const float gr = cell_gap(kr + 1, c.lx, g.cell, g.margin), gl = cell_gap(-(kl + 1), c.lx, g.cell, g.margin);
                kr += !(gr * gr > rest) ? 1 : 0;
                kl += !(gl * gl > rest) ? 1 : 0;
              }
              const int x_lo = max(c.cx - kl, 0), x_hi = min(c.cx + kr, g.nx - 1);
              if (x_lo > x_hi) continue;
              const uint32_t rowbase = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
              const bool central = (rho <= 1);  // rows of rings 0 and 1: cells cx - 1 .. cx + 1 were examined by the stage before
              const int a1 = central ? min(x_hi, c.cx - 2) : x_hi;
              const int b0 = max(x_lo, c.cx + 2);
              if (x_lo <= a1) {
                const uint32_t c0 = rowbase + (uint32_t)x_lo, c1 = rowbase + (uint32_t)a1 + 1u;
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                  qa[k] = nq == k ? c0 : qa[k];
                  qb[k] = nq == k ? c1 : qb[k];
                }
                nq += 1;
              }
              if (central && b0 <= x_hi) {
                const uint32_t c0 = rowbase + (uint32_t)b0, c1 = rowbase + (uint32_t)x_hi + 1u;
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                  qa[k] = nq == k ? c0 : qa[k];
                  qb[k] = nq == k ? c1 : qb[k];
                }
                nq += 1;
              }
            }
            if (nq > 0) {
              // ---- one batch: the 2 Q header words in one round trip, then the points of all ranges as ONE flat list
              uint32_t P[Q], D[Q], total = 0;
              {
                uint32_t ha[Q], hb[Q];
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                  ha[k] = cell_start[k < nq ? qa[k] : 0u];
                  hb[k] = cell_start[k < nq ? qb[k] : 0u];
                }
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                  const uint32_t len = k < nq ? hb[k] - ha[k] : 0u;
                  P[k] = total;
                  D[k] = ha[k] - total;
                  total += len;
                }
              }
              if (STATS) {
                n_rows += (unsigned long long)nq;
                n_cand += (unsigned long long)total;
              }
              if (O3S_DBG(32)) total = 0;  // hooks build, timing only: the row walk without its candidates
              for (uint32_t f0 = 0; f0 < total; f0 += (uint32_t)RCB) {
                float4 qv[RCB];
                uint32_t jj[RCB];
                bool ok[RCB];
#pragma unroll
                for (int v = 0; v < RCB; ++v) {
                  const uint32_t f = f0 + (uint32_t)v;
                  ok[v] = f < total;
                  uint32_t dsel = D[0];
#pragma unroll
                  for (int k = 1; k < Q; ++k) dsel = (P[k] <= f) ? D[k] : dsel;
                  jj[v] = ok[v] ? f + dsel : 0u;
                  qv[v] = ref[jj[v]];
                }
#pragma unroll
                for (int v = 0; v < RCB; ++v) own_take(b, dist2(sx, sy, sz, qv[v].x, qv[v].y, qv[v].z), qv[v], (int)jj[v], lim, ok[v]);
              }
            }
            if (t >= n_rows_r) break;
          }
        }
        group_min_di<G>(b.d, b.idx, gd, gi);
        best = fminf(best, gd);
        if (O3S_SHARE_BOUNDS) best = neighbour_bound<G>(b, sx, sy, sz, lim, best);  // ... and in the ring they have just walked
        if (active) {
          rho += 1;
          if (rho > rho_max || ring_lb2_yz(rho) > best) active = false;
        }
      }
    }
  } else
  if (__any(active)) {
    const CellGeom c = cell_geom(sx, sy, sz, g);
    int r = 2, rmax = 0;
    float m = 0.f;
    ring_range(c, g, r, rmax, m);
    const float extra2 = outside_extra2(sx, sy, sz, g);
    if (r > rmax || ring_lb2(r, m, g) + extra2 > fminf(gd, bound)) active = false;
    while (__any(active)) {
      if (active) {
        // Ring r = the shell of cells at Chebyshev distance r.  A lane takes the (dz, dy) rows t = sub, sub + G, ... in
        // chunks of RCH: the headers of a whole chunk travel in ONE round trip (a face row is one range [cx-r, cx+r], an
        // interior row its two end cells), then the candidates of each range go RCB at a time.  (One row and one
        // candidate per round trip, as in round 1, made the first iteration of a call — no incumbents, pose 0.1 m / 2 deg
        // off — five times as long as a converged one.)
        constexpr int RCH = 3;
        // only the rows of the shell that lie inside the grid: a query far outside would otherwise step through
        // (2 r + 1)^2 row slots per ring, nearly all of them beyond the grid
        const int dz_lo = max(-r, -c.cz), dz_hi = min(r, g.nz - 1 - c.cz);
        const int dy_lo = max(-r, -c.cy), dy_hi = min(r, g.ny - 1 - c.cy);
        const int ny_r = dy_hi - dy_lo + 1;
        const int n_rows_r = (dz_hi >= dz_lo && ny_r > 0) ? (dz_hi - dz_lo + 1) * ny_r : 0;
        for (int t0 = sub; t0 < n_rows_r; t0 += G * RCH) {
          uint32_t ja[RCH], jb2[RCH], jc[RCH], jd[RCH];
#pragma unroll
          for (int u = 0; u < RCH; ++u) {
            const int t = t0 + u * G;
            const int tz = t / ny_r;
            const int dz = dz_lo + tz, dy = dy_lo + (t - tz * ny_r);
            const int z = c.cz + dz, y = c.cy + dy;
            const float gz = cell_gap(dz, c.lz, g.cell, g.margin), gy = cell_gap(dy, c.ly, g.cell, g.margin);
            const bool open = (t < n_rows_r) & ((unsigned)z < (unsigned)g.nz) & ((unsigned)y < (unsigned)g.ny) &
                              !(gz * gz + gy * gy > fminf(fminf(gd, b.d), bound));
            const bool full = (dz == r) | (dz == -r) | (dy == r) | (dy == -r);
            const uint32_t rowbase = open ? ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx : 0u;
            // range A: the whole row of a face, the left end cell of an interior row; range B: the right end cell
            const int xa = full ? max(c.cx - r, 0) : c.cx - r;
            const int xb = full ? min(c.cx + r, g.nx - 1) : c.cx - r;
            const int xc = c.cx + r;
            const bool a_ok = open & (xa <= xb) & (xa >= 0) & (xb < g.nx);
            const bool b_ok = open & !full & (xc >= 0) & (xc < g.nx);
            const uint32_t oa = a_ok ? rowbase + (uint32_t)xa : 0u, ob = a_ok ? rowbase + (uint32_t)xb + 1u : 0u;
            const uint32_t oc = b_ok ? rowbase + (uint32_t)xc : 0u;
            // two 8-byte loads per row instead of four 4-byte ones: {start, next start} of the left end cell, and of the right
            // end cell of an interior row / the end of a face row's range (4-byte aligned pairs: U32Pair)
            const U32Pair pa = *reinterpret_cast<const U32Pair*>(cell_start + oa);
            const U32Pair pq = *reinterpret_cast<const U32Pair*>(cell_start + (full ? ob : oc));
            ja[u] = pa.x;
            jb2[u] = full ? pq.x : pa.y;
            jc[u] = pq.x;
            jd[u] = pq.y;
            if (!a_ok) jb2[u] = ja[u];
            if (!b_ok) jd[u] = jc[u];
          }
#pragma unroll
          for (int u = 0; u < RCH; ++u) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const uint32_t j0 = half == 0 ? ja[u] : jc[u], j1 = half == 0 ? jb2[u] : jd[u];
              for (uint32_t j = j0; j < j1; j += RCB) {
                float4 q[RCB];
#pragma unroll
                for (int v = 0; v < RCB; ++v) q[v] = ref[j + v < j1 ? j + v : j0];
#pragma unroll
                for (int v = 0; v < RCB; ++v) own_take(b, dist2(sx, sy, sz, q[v].x, q[v].y, q[v].z), q[v], (int)(j + v), lim, j + v < j1);
              }
              if (STATS) {
                n_rows += j1 > j0 ? 1 : 0;
                n_cand += (unsigned long long)(j1 - j0);
              }
            }
          }
        }
      }
      group_min_di<G>(b.d, b.idx, gd, gi);
      if (active) {
        r += 1;
        if (r > rmax || ring_lb2(r, m, g) + extra2 > fminf(gd, bound)) active = false;
      }
    }
  }
  // ---- outputs: the lane that examined the winner writes it (slot, d2, the matched point for the next iteration);
  //      lane 0 of the group writes the "no match" record.  Level-1 histogram as in k_match. ----
  int mybin = -1;
  if (valid && !O3S_DBG(8)) {
    const bool found = gi != 0x7fffffff;
    // two lanes of a group never examine the same reference point (disjoint rows; the central cells are left to the first
    // stage), but should they ever hold the same winner exactly one may write it and count it: the lowest lane that holds it
    bool mine = found && b.idx == gi && b.pos >= 0;
    if (FAR && G >= 2) {
      const uint32_t m = mine ? 1u : 0u;
      // the broadcasts run on every lane of the group (a DPP read from a lane that is switched off returns the reader's own value)
      const uint32_t m0 = group_bcast<G, 0>(m), m1 = group_bcast<G, 1 % G>(m), m2 = group_bcast<G, 2 % G>(m);
      const bool lower = ((sub > 0) & (m0 != 0u)) | ((G >= 4) & (sub > 1) & (m1 != 0u)) | ((G >= 4) & (sub > 2) & (m2 != 0u));
      mine = mine && !lower;
    }
    if (mine) {
      // the matched normal: one more gather at the very end of the launch, where thousands of other waves hide it — in
      // k_classify, which used to fetch it, the same gather was an exposed round trip of a one-wave-per-SIMD kernel
      const float4 nq = (mn && refn) ? refn[b.pos] : make_float4(0.f, 0.f, 0.f, 0.f);
      pos_out[i] = b.pos;
      d2_out[i] = b.d;
      mq[i] = make_float4(b.qx, b.qy, b.qz, 1.f);
      if (mn) mn[i] = nq;
      const int bin = (int)((__float_as_uint(b.d) >> 20) & (kHistBins - 1));
      if (!O3S_DBG(1) && atomicAdd(&s_hist[bin], 1u) == 0u) mybin = bin;
    } else if (!found && sub == 0) {
      pos_out[i] = -1;
      d2_out[i] = kInfF;
      mq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mn) mn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  if (mybin >= 0) atomicAdd(&hist_rep[(size_t)(blockIdx.x & (unsigned)rep_mask) * kHistBins + mybin], s_hist[mybin]);
  if (STATS) {
    n_cand = wave_sum_u64(n_cand);
    n_rows = wave_sum_u64(n_rows);
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&st->cand_count, n_cand);
      atomicAdd(&st->row_count, n_rows);
    }
  }
}

// MirrorMatcher (LPM/MatchersImpl.cpp:58-85): id = i, dist = 0 for every reading point; same outputs as the matcher
__global__ void __launch_bounds__(kBlock) k_match_mirror(int N, const float4* __restrict__ ref, const int32_t* __restrict__ orig_to_sorted,
                                                         const int32_t* __restrict__ perm, IcpState* __restrict__ st,
                                                         int32_t* __restrict__ pos_out, float* __restrict__ d2_out, float4* __restrict__ mq,
                                                         uint32_t* __restrict__ hist_rep) {
  const float hv = hdr_load(st);
  if (hdr_i(hv, H_DONE)) return;
  if (blockIdx.x == 0) {  // the level-2 histogram and the chain's start stamp, as in k_match2
    for (int k = threadIdx.x; k < 1024; k += kBlock) hist_rep[(size_t)kHistReplicas * kHistBins + k] = 0u;
    if (threadIdx.x == 0 && hdr_i(hv, H_ITER) == 0) st->t_begin = wall_clock64();
  }
  const int i = blockIdx.x * kBlock + threadIdx.x;
  uint32_t mine = 0;
  if (i < N) {
    const int slot = orig_to_sorted[perm[i]];
    const float4 q = ref[slot];
    pos_out[i] = slot;
    d2_out[i] = 0.f;
    mq[i] = make_float4(q.x, q.y, q.z, 1.f);
    mine = 1;
  }
  mine = wave_sum_u32(mine);  // every distance is +0.0f: bin 0 of the level-1 histogram
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&hist_rep[(size_t)(blockIdx.x & (kHistReplicas - 1)) * kHistBins], mine);
}

// histogram of externally supplied distances (module-level outlier API) into replica 0
__global__ void __launch_bounds__(kBlock) k_hist(const float* __restrict__ d2, int N, uint32_t* __restrict__ hist) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < N) {
    const float d = d2[i];
    if (d != kInfF) atomicAdd(&hist[(__float_as_uint(d) >> 20) & (kHistBins - 1)], 1u);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Outlier chain + trim limit + kept-pair centroids in two launches.
//   Matches::getDistsQuantile (LPM/Matches.cpp:61-87) must return the EXACT element nth_element would: a 3-level radix
//   selection on the fp32 bit pattern (non-negative floats order like their bits).  Level 1 (bits 30..20) was counted by
//   k_match.  Knowing the level-1 bin B that holds rank k already decides most pairs:
//       bin <  B  -> d2 <= limit for sure  (Trimmed weight 1)      bin > B -> weight 0      bin == B -> undecided
//   k_classify  (all CUs)  finds B (every block repeats the same integer scan of the summed histogram), applies the
//               SurfaceNormalOutlierFilter (LPM/OutlierFiltersImpl.cpp:236-281; a rejected pair is re-encoded as
//               pos = -2 - slot), sums p / q / count of the decided-kept pairs in fp64 (PointToPlane.cpp:263-264) and
//               appends the undecided pairs — with everything needed to finish them — to a candidate segment.
//   k_sel_finish (one 1024-lane block) resolves levels 2 and 3 over the candidates (in LDS when they fit), adds the
//               candidates with d2 <= limit to the sums and publishes limit, means and |K| in the state header.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSelThreads = 1024;   // the selection kernels of the sharded mode: one radix bin per thread
constexpr int kFinThreads = 512;    // k_sel_finish: fewer waves per barrier, two radix bins per thread
constexpr int kFinPerSmall = 4, kFinPerBig = 8;  // candidate records in flight per thread and batch of the selection sweep (<= 2 048 candidates: one batch of 4)
constexpr int kSelCap = 32768;   // candidates resolved in LDS; larger bins are resolved in global memory

enum { kModeCentroid = 1, kModeGate = 2, kModeNormalReady = 4 /*k_match2 has already written the matched normals (mn)*/ };
constexpr int kClsBlock = 512;  // threads (= points) per k_classify block: halves the blocks that each re-sum the histogram replicas

__global__ void __launch_bounds__(kClsBlock) k_classify(const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
                                                     const float* __restrict__ rnx, const float* __restrict__ rny, const float* __restrict__ rnz,
                                                     int N, const float4* __restrict__ ref, const float4* __restrict__ refn,
                                                     int32_t* __restrict__ pos, const float* __restrict__ d2,
                                                     const uint32_t* __restrict__ hist_rep, ChainParams cp, IcpState* __restrict__ st,
                                                     SelScratch* __restrict__ ss, CandRec* __restrict__ cand /*[grid][kClsBlock]*/,
                                                     uint32_t* __restrict__ cand_cnt /*[grid]*/, uint32_t* __restrict__ hist2 /*[1024]*/,
                                                     const float4* __restrict__ mq /*matched point of every query (k_match2 / k_import_matches)*/,
                                                     float4* __restrict__ mn /*out: matched normal, streamed by k_normal_eq*/,
                                                     double* __restrict__ part /*[7][grid]*/, int mode,
                                                     int n_rep /*level-1 replicas to sum: kHistReplicas; fewer in the sharded mode, where they travel*/,
                                                     int l2_shift = 10, uint32_t l2_mask = 1023u /*the level-2 digit: bits 19..10; the sharded chain takes
                                                     thirteen bits (shift 7, mask 8191: csrc/icp_shard_kernels.h)*/) {
  __shared__ uint32_t s_sc[32];
  __shared__ uint32_t s_res[4];
  __shared__ uint32_t s_wcnt[kClsBlock / 64];
  using Sum = BlockSum<kCentComps, kClsBlock>;
  __shared__ double s_a[Sum::kWordsA];
  __shared__ double s_b[Sum::kWordsB];
  O3S_TSTAMP(40);
  const float hv = hdr_load(st);
  // this thread's point and the level-1 histogram (8 replicas) are fetched in the same round trip as the header
  const int i = blockIdx.x * kClsBlock + threadIdx.x;
  const bool inb = i < N;
  const int pe0 = inb ? pos[i] : -1;
  const float d = inb ? d2[i] : kInfF;
  const float x0 = inb ? rx[i] : 0.f, y0 = inb ? ry[i] : 0.f, z0 = inb ? rz[i] : 0.f;
  const float4 q = inb ? mq[i] : make_float4(0.f, 0.f, 0.f, 0.f);  // the matched point arrives with the query: no gather
  const bool nready = (mode & kModeNormalReady) != 0;               // ... and so does its normal when the matcher wrote it
  const float4 rn_in = (nready && inb) ? mn[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const bool gate = (mode & kModeGate) && cp.has_normal_gate;
  float a0 = 0.f, b0 = 0.f, c0 = 0.f;
  if (gate && inb) {
    a0 = rnx[i];
    b0 = rny[i];
    c0 = rnz[i];
  }
  constexpr int kBpt = kHistBins / kClsBlock;  // level-1 bins owned by a thread (4)
  static_assert(kBpt == 4, "one uint4 per replica and thread");
  uint32_t c[kBpt] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < kHistReplicas; ++r) {  // branch-free: a replica beyond n_rep re-reads replica 0 and counts nothing
    const bool on = r < n_rep;
    const uint4 u0 = *reinterpret_cast<const uint4*>(hist_rep + (size_t)(on ? r : 0) * kHistBins + threadIdx.x * kBpt);
    c[0] += on ? u0.x : 0u;
    c[1] += on ? u0.y : 0u;
    c[2] += on ? u0.z : 0u;
    c[3] += on ? u0.w : 0u;
  }
  O3S_TSTAMP(41);
  if (hdr_i(hv, H_DONE)) return;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = hdr_f(hv, k);
  // dependent gather of the matched reference normal; it is handed on (coalesced) to k_normal_eq
  const int slot0 = pe0 >= 0 ? pe0 : (pe0 <= -2 ? -2 - pe0 : -1);
  const bool matched = pe0 >= 0 || pe0 <= -2;
  float4 rn = rn_in;
  if (!nready) {  // uniform
    if (matched && refn) rn = refn[slot0];
    if (inb) mn[i] = rn;
  }
  O3S_TSTAMP(42);
  // ---- rank-k bin: every block repeats the same integer arithmetic on the same summed histogram ----
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < kBpt; ++k) mine += c[k];
  uint32_t n_fin;
  const uint32_t ex = block_excl_scan(mine, &n_fin, s_sc);
  uint32_t bin = kHistBins;  // no Trimmed filter: every finite distance is "below"
  bool skip = false;
  if (cp.has_trim) {
    if (n_fin == 0) {  // "No matches available for computing distance quantiles" (Matches.cpp:76-77)
      skip = true;
      if (blockIdx.x == 0 && threadIdx.x == 0) st->status = 5;
    } else {
      // index: values.size() * quantile evaluated in fp32, truncated (Matches.cpp:85-86); ratio == 1 -> max element
      uint32_t k;
      if (cp.trim_ratio == 1.0f) {
        k = n_fin - 1;
      } else {
        k = (uint32_t)((float)n_fin * cp.trim_ratio);
        if (k >= n_fin) k = n_fin - 1;
      }
      if (mine > 0 && ex <= k && k < ex + mine) {
        uint32_t acc = ex;
#pragma unroll
        for (int qd = 0; qd < kBpt; ++qd) {
          if (c[qd] > 0 && acc <= k && k < acc + c[qd]) {
            s_res[0] = threadIdx.x * kBpt + qd;
            s_res[1] = k - acc;
            s_res[2] = c[qd];
          }
          acc += c[qd];
        }
      }
      __syncthreads();
      bin = s_res[0];
    }
  } else {
    skip = true;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->n_finite = n_fin;
    ss->skip = skip ? 1u : 0u;
    if (!skip) {
      ss->bin = bin;
      ss->kk = s_res[1];
      ss->bin_count = s_res[2];
    }
  }
  if (cp.has_trim && n_fin == 0) return;
  O3S_TSTAMP(43);
  // ---- per-pair weights ----
  bool keep = matched && pe0 >= 0;  // caller-supplied zero weights arrive as pos <= -2 (module-level minimise)
  if (gate && matched) {  // w = (n_read . n_ref < cos(maxAngle)) ? 0 : 1 on the ROTATED reading normal
    const float nx = rot_row(T, 0, a0, b0, c0), ny = rot_row(T, 1, a0, b0, c0), nz = rot_row(T, 2, a0, b0, c0);
    float v = nx * rn.x;
    v = v + ny * rn.y;
    v = v + nz * rn.z;
    if (v < cp.cos_max_angle) {
      keep = false;
      pos[i] = -2 - slot0;
    }
  }
  if (!(d <= cp.max_out_r2)) keep = false;
  const uint32_t u = __float_as_uint(d);
  const bool finite = matched && d != kInfF;
  const uint32_t pbin = (u >> 20) & (kHistBins - 1);
  const float sx = xf_row(T, 0, x0, y0, z0), sy = xf_row(T, 1, x0, y0, z0), sz = xf_row(T, 2, x0, y0, z0);
  const bool decided_kept = finite && keep && pbin < bin;
  const bool undecided = finite && pbin == bin;
  // ---- undecided pairs -> THIS block's candidate region, in thread order: no reservation atomic, and the order in which
  //      k_sel_finish later adds them up is the same on every run.  Their level-2 digits (bits 19..10) are counted here
  //      (~2 000 atomics spread over 1 024 addresses), so the single finishing block starts from a ready histogram. ----
  const unsigned long long umask = __ballot(undecided);
  if ((threadIdx.x & 63) == 0) s_wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(umask);
  O3S_TSTAMP(44);
  // ---- fp64 sums of the decided-kept pairs ----
  if (mode & kModeCentroid) {  // uniform
    double a[kCentComps];
    a[0] = decided_kept ? (double)sx : 0.0;
    a[1] = decided_kept ? (double)sy : 0.0;
    a[2] = decided_kept ? (double)sz : 0.0;
    a[3] = decided_kept ? (double)q.x : 0.0;
    a[4] = decided_kept ? (double)q.y : 0.0;
    a[5] = decided_kept ? (double)q.z : 0.0;
    a[6] = decided_kept ? 1.0 : 0.0;
    Sum::run(a, s_a, s_b);
    if (threadIdx.x < kCentComps) part[threadIdx.x * gridDim.x + blockIdx.x] = Sum::total(s_b, threadIdx.x);
  } else {
    __syncthreads();
  }
  {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t base = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < kClsBlock / 64; ++k) {
      const uint32_t ck = s_wcnt[k];
      base += k < w ? ck : 0u;
      cnt += ck;
    }
    if (undecided) {
      CandRec rec;
      rec.px = sx;
      rec.py = sy;
      rec.pz = sz;
      rec.bits = u;
      rec.qx = q.x;
      rec.qy = q.y;
      rec.qz = q.z;
      rec.keep = (int32_t)(((uint32_t)i << 1) | (keep ? 1u : 0u));  // bit 0: every other weight of the chain is 1; above it: the query's slot
      cand[(size_t)blockIdx.x * kClsBlock + base + (uint32_t)__popcll(umask & ((1ull << lane) - 1ull))] = rec;
      atomicAdd(&hist2[(u >> l2_shift) & l2_mask], 1u);
    }
    if (threadIdx.x == 0) cand_cnt[blockIdx.x] = cnt;
  }
  O3S_TSTAMP(45);
}

// ------------------------------------------------------------------------------------------------------------------
// k_sel_finish (one block) — exact k-th smallest finite d2 + means of the kept pairs.
//   in   classify partials [7][nb], candidate counts [nb] and records [nb][kClsBlock], level-2 histogram [1024]
//   1.   exclusive scan of the counts -> base of every classify block's candidates in ONE flat, run-independent order
//   2.   scan of the level-2 histogram -> digit d1 holding rank kk, rank kk2 inside it
//   3.   every thread loads its chunk of the flat list (the block of its first slot by binary search in LDS); the few
//        candidates that carry the 21-bit prefix (bin, d1) go to an LDS list; <= 64 of them: exact rank by counting in one
//        wave, more: a 10-bit LDS histogram (heavy ties)
//   4.   candidates with d2 <= limit join the sums (thread order, then the fixed-order block sum); publish
// ------------------------------------------------------------------------------------------------------------------
constexpr int kBaseCap = 2048;  // classify blocks whose bases live in LDS (readings up to 1 M points); beyond: global scratch

// block of flat slot f: largest b with base[b] <= f (f < base[nb]); base is non-decreasing
__device__ __forceinline__ int flat_block(const uint32_t* base, int nb, uint32_t f) {
  int lo = 0, hi = nb;  // invariant: base[lo] <= f < base[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (base[mid] <= f) lo = mid;
    else hi = mid;
  }
  return lo;
}

// one 10-bit radix level over an LDS list of bit patterns that all share the higher bits: digit holding rank kk
__device__ __forceinline__ void select_level_list(const uint32_t* vals, uint32_t n_vals, int shift, uint32_t* s_bins /*1024*/, uint32_t* s_tmp,
                                                  uint32_t& kk, uint32_t& digit) {
  constexpr int BPT = 1024 / kFinThreads;  // radix bins owned by a thread
#pragma unroll
  for (int q = 0; q < BPT; ++q) s_bins[threadIdx.x * BPT + q] = 0u;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_vals; i += kFinThreads) atomicAdd(&s_bins[(vals[i] >> shift) & 1023u], 1u);
  __syncthreads();
  uint32_t cb[BPT], c = 0;
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    cb[q] = s_bins[threadIdx.x * BPT + q];
    c += cb[q];
  }
  uint32_t tot;
  const uint32_t ex = block_excl_scan(c, &tot, s_tmp);
  __syncthreads();
  if (c > 0 && ex <= kk && kk < ex + c) {
    uint32_t acc = ex;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      if (cb[q] > 0 && acc <= kk && kk < acc + cb[q]) {
        s_tmp[40] = threadIdx.x * BPT + q;
        s_tmp[41] = kk - acc;
      }
      acc += cb[q];
    }
  }
  __syncthreads();
  digit = s_tmp[40];
  kk = s_tmp[41];
  __syncthreads();
}

// exact element of rank kk2 among the m (<= kSelCap) bit patterns of the LDS list `vals`; block-wide call
__device__ __forceinline__ uint32_t rank_in_list(const uint32_t* vals, uint32_t m, uint32_t kk2, uint32_t* s_bins, uint32_t* s_tmp) {
  if (m <= 64u) {  // uniform: one wave counts, for every element, how many are smaller / not larger
    if (threadIdx.x < 64) {
      const uint32_t v = threadIdx.x < m ? vals[threadIdx.x] : 0xffffffffu;
      uint32_t lt = 0, le = 0;
      for (uint32_t j = 0; j < m; ++j) {
        const uint32_t x = vals[j];
        lt += x < v ? 1u : 0u;
        le += x <= v ? 1u : 0u;
      }
      if (threadIdx.x < m && lt <= kk2 && kk2 < le) s_tmp[42] = v;  // every hit holds the same value
    }
    __syncthreads();
    return s_tmp[42];
  }
  uint32_t d0, kk = kk2;
  select_level_list(vals, m, 0, s_bins, s_tmp, kk, d0);
  return (vals[0] & ~1023u) | d0;
}

// LDS plan of the selection (the 128 KB dynamic area): bit patterns of the parked candidates | their records | their flat
// indices.  A candidate is PARKED when its 21 leading bits equal (bin, d1): only those few are still undecided.
constexpr int kParkBits = 16384;   // parked bit patterns kept for the rank (more: counted from memory)
constexpr int kParkRecs = 1024;    // parked records kept for the sums (more: a second sweep adds them)
static_assert(kParkBits + kParkRecs * 8 + kParkRecs <= kSelCap, "selection LDS plan");

// One sweep over the flat candidate list in batches of PER slots per thread (slot = t + 512 * (PER * batch + k): the PER
// binary searches of a batch run interleaved, its PER record loads go out together).  Decided candidates — 21-bit prefix
// below (bin, d1): kept, above: dropped — are added to the thread's sums at once, in a run-independent order; the
// undecided ones are parked in LDS.  Returns nothing; the caller ranks the parked bit patterns.
template <int PER>
__device__ __forceinline__ void sel_sweep(const CandRec* __restrict__ cand, const uint32_t* s_base, int nb, uint32_t total, uint32_t prefix21,
                                          uint32_t* s_dyn, uint32_t* s_tmp, int mode, double* a) {
  uint32_t* s_list = s_dyn;
  CandRec* s_rec = reinterpret_cast<CandRec*>(s_dyn + kParkBits);
  uint32_t* s_flat = s_dyn + kParkBits + kParkRecs * 8;
  for (uint32_t base = 0; base < total; base += (uint32_t)(kFinThreads * PER)) {  // uniform trip count
    int lo[PER], hi[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      lo[k] = 0;
      hi[k] = nb;
    }
    for (int step = nb; step > 1; step = (step + 1) >> 1) {  // ceil(log2(nb)) rounds; a settled search idles
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint32_t f = base + threadIdx.x + (uint32_t)k * kFinThreads;
        const int mid = (lo[k] + hi[k]) >> 1;
        const bool go = hi[k] - lo[k] > 1;
        const bool up = s_base[mid] <= (f < total ? f : 0u);
        lo[k] = (go && up) ? mid : lo[k];
        hi[k] = (go && !up) ? mid : hi[k];
      }
    }
    CandRec rec[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t f = base + threadIdx.x + (uint32_t)k * kFinThreads;
      rec[k] = cand[f < total ? (size_t)lo[k] * kClsBlock + (f - s_base[lo[k]]) : (size_t)0];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t f = base + threadIdx.x + (uint32_t)k * kFinThreads;
      if (f >= total) continue;
      const uint32_t p21 = rec[k].bits >> 10;
      if (p21 == prefix21) {
        const uint32_t slot = atomicAdd(&s_tmp[43], 1u);
        if (slot < (uint32_t)kParkBits) s_list[slot] = rec[k].bits;
        if (slot < (uint32_t)kParkRecs) {
          s_rec[slot] = rec[k];
          s_flat[slot] = f;
        }
      } else if ((mode & kModeCentroid) && p21 < prefix21 && (rec[k].keep & 1)) {
        a[0] += (double)rec[k].px;
        a[1] += (double)rec[k].py;
        a[2] += (double)rec[k].pz;
        a[3] += (double)rec[k].qx;
        a[4] += (double)rec[k].qy;
        a[5] += (double)rec[k].qz;
        a[6] += 1.0;
      }
    }
  }
}

// The selection sweep of large readings, without the flat index: every thread walks the candidate regions of the classify
// blocks it already holds the counts of (consecutive blocks, slots in order — a run-independent order), kOwnUn records per
// round trip.  No prefix sum over the counts, no binary search from a flat index back to (block, slot).  Used when no
// region holds more than kOwnMax candidates (else one thread would serialise a long region) and the undecided candidates
// fit the parking area; the flat sweep above remains for everything else.  A parked record's order key is
// (block << 9 | slot): the same order as its flat index.
constexpr int kOwnMax = 32, kOwnUn = 8;
static_assert(kClsBlock <= 512, "order key: 9 bits of slot");
// one record of the sweep: parked when it carries the 21-bit prefix, summed when it lies below it and is kept
__device__ __forceinline__ void sel_take(const CandRec& r, uint32_t key, uint32_t prefix21, uint32_t* s_dyn, uint32_t* s_tmp, int mode, double* a) {
  uint32_t* s_list = s_dyn;
  CandRec* s_rec = reinterpret_cast<CandRec*>(s_dyn + kParkBits);
  uint32_t* s_flat = s_dyn + kParkBits + kParkRecs * 8;
  const uint32_t p21 = r.bits >> 10;
  if (p21 == prefix21) {
    const uint32_t slot = atomicAdd(&s_tmp[43], 1u);
    if (slot < (uint32_t)kParkBits) s_list[slot] = r.bits;
    if (slot < (uint32_t)kParkRecs) {
      s_rec[slot] = r;
      s_flat[slot] = key;
    }
  } else if ((mode & kModeCentroid) && p21 < prefix21 && (r.keep & 1)) {
    a[0] += (double)r.px;
    a[1] += (double)r.py;
    a[2] += (double)r.pz;
    a[3] += (double)r.qx;
    a[4] += (double)r.qy;
    a[5] += (double)r.qz;
    a[6] += 1.0;
  }
}
template <int NREG>
__device__ __forceinline__ void sel_sweep_own(const CandRec* __restrict__ cand, int b0, int b1, const uint32_t (&cnts)[NREG], uint32_t prefix21,
                                              uint32_t* s_dyn, uint32_t* s_tmp, int mode, double* a) {
#pragma unroll
  for (int j = 0; j < NREG; ++j) {
    const int b = b0 + j;
    const uint32_t n = b < b1 ? cnts[j] : 0u;
    const CandRec* region = cand + (size_t)(b < b1 ? b : 0) * kClsBlock;
    for (uint32_t s0 = 0; s0 < n; s0 += kOwnUn) {
      CandRec rec[kOwnUn];
#pragma unroll
      for (int k = 0; k < kOwnUn; ++k) rec[k] = region[s0 + k < n ? s0 + k : 0u];
#pragma unroll
      for (int k = 0; k < kOwnUn; ++k)
        if (s0 + k < n) sel_take(rec[k], ((uint32_t)b << 9) | (s0 + k), prefix21, s_dyn, s_tmp, mode, a);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// k_sel_partial — the selection sweep of LARGE readings (more classify blocks than the finishing block has threads: C4's 977)
// spread over many blocks, in front of k_sel_finish.  The finishing block alone walked ~10 k candidate records (320 KB) and
// took 19 us at C4; here every block repeats the (cheap) level-2 digit search on the ready histogram, sweeps the candidate
// regions of ITS classify blocks — 32 lanes per region, slots in order — sums the decided candidates in a fixed order
// (thread partials, then the block sum) and appends the few that carry the 21-bit prefix to one global list together with
// their order key (block << 9 | slot): k_sel_finish ranks that list, adds its kept members in key order and folds the block
// partials in block order, so repeated runs give the same bits.  The list holds kParkRecs records; with more (heavy ties)
// k_sel_finish falls back to its own sweep and ignores everything written here.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSelPartRegions = kFinThreads / 32;  // classify regions a block sweeps per round: one per 32 lanes
constexpr int kSelPartMaxBlocks = 256;
__global__ void __launch_bounds__(kFinThreads) k_sel_partial(const IcpState* __restrict__ st, SelScratch* __restrict__ ss, const CandRec* __restrict__ cand,
                                                             const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ hist2, int nb, int mode,
                                                             double* __restrict__ part2 /*[7][grid]*/, CandRec* __restrict__ park_rec /*[kParkRecs]*/,
                                                             uint32_t* __restrict__ park_key /*[kParkRecs]*/) {
  __shared__ uint32_t s_tmp[64];
  using Sum = BlockSum<kCentComps, kFinThreads>;
  __shared__ double s_a[Sum::kWordsA];
  __shared__ double s_b[Sum::kWordsB];
  const float hv = hdr_load(st);
  const uint32_t ssw = reinterpret_cast<const uint32_t*>(ss)[threadIdx.x & 7];
  const uint2 h2 = *reinterpret_cast<const uint2*>(hist2 + 2 * threadIdx.x);
  if (hdr_i(hv, H_DONE)) return;
  const uint32_t bin = (uint32_t)__builtin_amdgcn_readlane((int)ssw, kSegs), kk = (uint32_t)__builtin_amdgcn_readlane((int)ssw, kSegs + 1),
                 skip = (uint32_t)__builtin_amdgcn_readlane((int)ssw, kSegs + 3);
  double a[kCentComps] = {0, 0, 0, 0, 0, 0, 0};
  if (!skip) {  // uniform
    const uint32_t c2 = h2.x + h2.y;
    uint32_t tot2;
    const uint32_t ex2 = block_excl_scan(c2, &tot2, s_tmp);
    if (c2 > 0 && ex2 <= kk && kk < ex2 + c2) s_tmp[40] = 2 * threadIdx.x + (kk < ex2 + h2.x ? 0 : 1);
    __syncthreads();
    const uint32_t prefix21 = (bin << 10) | s_tmp[40];
    const int per_block = (nb + gridDim.x - 1) / gridDim.x;  // consecutive classify regions of this block
    const int r0 = blockIdx.x * per_block, r1 = min(r0 + per_block, nb);
    const int grp = threadIdx.x >> 5, lane = threadIdx.x & 31;
    for (int rb = r0 + grp; rb < r1; rb += kSelPartRegions) {
      const uint32_t n = cand_cnt[rb];
      const CandRec* region = cand + (size_t)rb * kClsBlock;
      for (uint32_t sl = (uint32_t)lane; sl < n; sl += 32u) {
        const CandRec r = region[sl];
        const uint32_t p21 = r.bits >> 10;
        if (p21 == prefix21) {
          const uint32_t slot = atomicAdd(&ss->seg_count[0], 1u);
          if (slot < (uint32_t)kParkRecs) {
            park_rec[slot] = r;
            park_key[slot] = ((uint32_t)rb << 9) | sl;
          }
        } else if ((mode & kModeCentroid) && p21 < prefix21 && (r.keep & 1)) {
          a[0] += (double)r.px;
          a[1] += (double)r.py;
          a[2] += (double)r.pz;
          a[3] += (double)r.qx;
          a[4] += (double)r.qy;
          a[5] += (double)r.qz;
          a[6] += 1.0;
        }
      }
    }
  }
  Sum::run(a, s_a, s_b);
  if (threadIdx.x < kCentComps) part2[threadIdx.x * gridDim.x + blockIdx.x] = Sum::total(s_b, threadIdx.x);
}

// The body of k_sel_finish as a block-wide device function, so that k_sel_ne can run it in EVERY block in front of the
// normal equations (FUSED): all blocks then hold the same limit and means — the same integers, the same fixed-order fp64
// sums — without a kernel boundary in between; only block 0 publishes them to the state.  Returns false when the iteration
// ends here (chain done, an earlier error, no pair kept); otherwise s_out = {limit, mean of the reading points (3), mean of
// the matched points (3)} is valid after the caller's next barrier.
// What the fused kernel's blocks form for themselves instead of publishing it field by field (see solve_body)
struct SolveOverride {
  float limit, mp[3], mq[3];
  int32_t kept;
  int32_t status;     // != 0: this launch found that the iteration cannot go on (no pair kept): 6
  int32_t has_limit;  // the limit above supersedes the state's
};

template <bool FUSED>
__device__ __forceinline__ bool sel_finish_body(uint32_t* __restrict__ hist_rep, const ChainParams& cp, IcpState* __restrict__ st,
                                                const SelScratch* __restrict__ ss, const CandRec* __restrict__ cand,
                                                const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ hist2,
                                                uint32_t* __restrict__ base_scratch /*[nb + 1], used when nb > kBaseCap*/,
                                                const double* __restrict__ part /*[7][nb]*/, int nb, int mode, float hv, float* s_out /*[8], LDS*/,
                                                const double* __restrict__ part2 = nullptr /*[7][nbp]: k_sel_partial ran in front (large readings)*/,
                                                int nbp = 0, const CandRec* __restrict__ park_rec = nullptr, const uint32_t* __restrict__ park_key = nullptr,
                                                uint32_t* __restrict__ park_cnt = nullptr, SolveOverride* s_ov = nullptr /*LDS, FUSED only*/,
                                                bool tail = false /*FUSED: the launch closes the iteration itself*/) {
  // with the closing tail the fused kernel's last block writes limit / means / |K| with the rest of the state (solve_body);
  // without it block 0 publishes them for k_solve, as the single-block k_sel_finish does
  const bool publish = !FUSED || (!tail && blockIdx.x == 0);
  extern __shared__ __align__(16) uint32_t s_dyn[];  // kSelCap words: the level-3 list, then the final block sum
  __shared__ uint32_t s_bins[1024];
  __shared__ uint32_t s_tmp[64];
  __shared__ uint32_t s_ssw[8];
  __shared__ uint32_t s_base_lds[kBaseCap + 1];
  using Sum = BlockSum<kCentComps, kFinThreads>;
  static_assert((Sum::kWordsA + Sum::kWordsB) * 8 <= kSelCap * 4, "the block sum borrows the selection buffer");
  double* s_a = reinterpret_cast<double*>(s_dyn);
  double* s_b = s_a + Sum::kWordsA;
  O3S_TSTAMP(0);
  // first round trip: header, hand-off words, this thread's share of the classify partials, candidate counts, level 2
  const uint32_t ssw = reinterpret_cast<const uint32_t*>(ss)[threadIdx.x & 7];
  double a[kCentComps] = {0, 0, 0, 0, 0, 0, 0};
  if (mode & kModeCentroid) {
    for (int b = threadIdx.x; b < nb; b += kFinThreads) {
#pragma unroll
      for (int k = 0; k < kCentComps; ++k) a[k] += part[k * nb + b];
    }
  }
  const int per_thread = (nb + kFinThreads - 1) / kFinThreads;  // consecutive classify blocks owned by a thread
  const int b0 = min(threadIdx.x * per_thread, nb), b1 = min(b0 + per_thread, nb);
  constexpr int kCntRegs = kBaseCap / kFinThreads;  // counts a thread keeps in registers (readings up to 1 M points)
  uint32_t cnts[kCntRegs];
  uint32_t my_cnt = 0;
#pragma unroll
  for (int j = 0; j < kCntRegs; ++j) {
    cnts[j] = b0 + j < b1 ? cand_cnt[b0 + j] : 0u;
    my_cnt += cnts[j];
  }
  for (int b = b0 + kCntRegs; b < b1; ++b) my_cnt += cand_cnt[b];  // larger readings: re-read below
  const uint2 h2 = *reinterpret_cast<const uint2*>(hist2 + 2 * threadIdx.x);
  const uint32_t parked_before = park_cnt ? *park_cnt : 0u;  // what k_sel_partial appended (uniform)
  O3S_TSTAMP(1);
  if (hdr_i(hv, H_DONE)) return false;
  if (hist_rep) {  // NULL when k_normal_eq clears the replicas (the fused chain); uniform
    __syncthreads();  // every thread holds its level-2 words before anyone clears them
    for (int k = threadIdx.x; k < kHistReplicas * kHistBins + 1024; k += kFinThreads) hist_rep[k] = 0u;  // + level 2, ready for the next iteration
  }
  if (threadIdx.x < 8) s_ssw[threadIdx.x] = ssw;  // seg_count[4] (unused), bin, kk, bin_count, skip
  if (threadIdx.x == 0) {
    s_tmp[42] = 0x7f800000u;
    s_tmp[43] = 0u;
    s_tmp[45] = 0u;
  }
  __syncthreads();
  const uint32_t bin = s_ssw[kSegs], skip = s_ssw[kSegs + 3];
  uint32_t kk = s_ssw[kSegs + 1];
  float limit = kInfF;
  O3S_TSTAMP(2);
  if (!skip) {  // uniform
    uint32_t* s_base = nb <= kBaseCap ? s_base_lds : base_scratch;
    uint32_t total = 0, lbits, d1, kk2, prefix21;
    bool own = false;
    if (part2 && parked_before <= (uint32_t)kParkRecs) {  // uniform: k_sel_partial swept the candidates; fold what it left
      {
        const uint32_t c2 = h2.x + h2.y;
        uint32_t tot2;
        const uint32_t ex2 = block_excl_scan(c2, &tot2, s_tmp);
        if (c2 > 0 && ex2 <= kk && kk < ex2 + c2) {
          const bool first = kk < ex2 + h2.x;
          s_tmp[40] = 2 * threadIdx.x + (first ? 0 : 1);
          s_tmp[41] = first ? kk - ex2 : kk - ex2 - h2.x;
        }
      }
      for (int b = threadIdx.x; b < nbp; b += kFinThreads) {
#pragma unroll
        for (int k = 0; k < kCentComps; ++k) a[k] += part2[k * nbp + b];
      }
      {
        uint32_t* s_list = s_dyn;
        CandRec* s_rec = reinterpret_cast<CandRec*>(s_dyn + kParkBits);
        uint32_t* s_flat = s_dyn + kParkBits + kParkRecs * 8;
        for (uint32_t j = threadIdx.x; j < parked_before; j += kFinThreads) {
          const CandRec r = park_rec[j];
          s_list[j] = r.bits;
          s_rec[j] = r;
          s_flat[j] = park_key[j];
        }
      }
      if (threadIdx.x == 0) s_tmp[43] = parked_before;
      __syncthreads();
      d1 = s_tmp[40];
      kk2 = s_tmp[41];
      prefix21 = (bin << 10) | d1;
      own = true;  // no sweep of this block's own
    } else if (nb > kFinThreads && nb <= kBaseCap) {  // uniform: large readings (more classify blocks than threads)
      // 1. the level-2 digit (bits 19..10) that holds rank kk, the rank inside it, and how many candidates carry it (= the
      //    number that will have to be parked: known before the sweep)
      {
        const uint32_t c2 = h2.x + h2.y;
        uint32_t tot2;
        const uint32_t ex2 = block_excl_scan(c2, &tot2, s_tmp);
        if (c2 > 0 && ex2 <= kk && kk < ex2 + c2) {
          const bool first = kk < ex2 + h2.x;
          s_tmp[40] = 2 * threadIdx.x + (first ? 0 : 1);
          s_tmp[41] = first ? kk - ex2 : kk - ex2 - h2.x;
          s_tmp[45] = first ? h2.x : h2.y;
        }
      }
      uint32_t long_region = 0;
#pragma unroll
      for (int j = 0; j < kCntRegs; ++j) long_region |= cnts[j] > (uint32_t)kOwnMax ? 1u : 0u;
      const bool any_long = __syncthreads_or((int)long_region) != 0;  // also publishes s_tmp[40..45]
      d1 = s_tmp[40];
      kk2 = s_tmp[41];
      prefix21 = (bin << 10) | d1;
      // own-region sweep: at C4 19.2 us for the kernel against 24.9 us with the flat sweep's three batches of ten-step
      // searches.  (At C2 — 196 regions of ~10 — the flat sweep's single batch is as fast: 10.9 us vs 11.1 us with the
      // regions shared between threads and 12.0 us with one thread per region; small readings keep it, below.)
      own = !any_long && s_tmp[45] <= (uint32_t)kParkRecs;
      if (own) {
        sel_sweep_own<kCntRegs>(cand, b0, b1, cnts, prefix21, s_dyn, s_tmp, mode, a);
      } else {
        uint32_t run = block_excl_scan(my_cnt, &total, s_tmp);
#pragma unroll
        for (int j = 0; j < kCntRegs; ++j)
          if (b0 + j < b1) {
            s_base[b0 + j] = run;
            run += cnts[j];
          }
        if (threadIdx.x == 0) s_base[nb] = total;
        __syncthreads();
      }
    } else {
      // 1. bases of the classify blocks' candidate runs and 2. the level-2 digit (bits 19..10) that holds rank kk: the two
      //    scans share their barriers
      const uint32_t c2 = h2.x + h2.y;
      uint32_t tot2, run, ex2;
      block_excl_scan2(my_cnt, c2, &total, &tot2, &run, &ex2, s_tmp);
#pragma unroll
      for (int j = 0; j < kCntRegs; ++j)
        if (b0 + j < b1) {
          s_base[b0 + j] = run;
          run += cnts[j];
        }
      for (int b = b0 + kCntRegs; b < b1; ++b) {
        s_base[b] = run;
        run += cand_cnt[b];
      }
      if (threadIdx.x == 0) s_base[nb] = total;
      if (c2 > 0 && ex2 <= kk && kk < ex2 + c2) {
        const bool first = kk < ex2 + h2.x;
        s_tmp[40] = 2 * threadIdx.x + (first ? 0 : 1);
        s_tmp[41] = first ? kk - ex2 : kk - ex2 - h2.x;
      }
      if (nb > kBaseCap) __threadfence_block();
      __syncthreads();
      d1 = s_tmp[40];
      kk2 = s_tmp[41];
      prefix21 = (bin << 10) | d1;
    }
    if (!own) {
      if (total <= (uint32_t)(kFinThreads * kFinPerSmall)) sel_sweep<kFinPerSmall>(cand, s_base, nb, total, prefix21, s_dyn, s_tmp, mode, a);
      else sel_sweep<kFinPerBig>(cand, s_base, nb, total, prefix21, s_dyn, s_tmp, mode, a);
    }
    O3S_TSTAMP(3);
    __syncthreads();
    const uint32_t m = s_tmp[43];  // parked = undecided candidates (typically a handful)
    if (m <= (uint32_t)kParkBits) {
      lbits = rank_in_list(s_dyn, m, kk2, s_bins, s_tmp);
    } else {
      // more than 16 384 candidates share 21 leading bits (pathological ties): count the last 10 bits straight from memory
#pragma unroll
      for (int q = 0; q < 1024 / kFinThreads; ++q) s_bins[threadIdx.x * (1024 / kFinThreads) + q] = 0u;
      __syncthreads();
      for (uint32_t f = threadIdx.x; f < total; f += kFinThreads) {
        const int b = flat_block(s_base, nb, f);
        const uint32_t bits = cand[(size_t)b * kClsBlock + (f - s_base[b])].bits;
        if ((bits >> 10) == prefix21) atomicAdd(&s_bins[bits & 1023u], 1u);
      }
      __syncthreads();
      const uint32_t c0 = s_bins[2 * threadIdx.x], c1 = s_bins[2 * threadIdx.x + 1];
      uint32_t tot3;
      const uint32_t ex3 = block_excl_scan(c0 + c1, &tot3, s_tmp);
      __syncthreads();
      if (c0 + c1 > 0 && ex3 <= kk2 && kk2 < ex3 + c0 + c1) s_tmp[44] = 2 * threadIdx.x + (kk2 < ex3 + c0 ? 0 : 1);
      __syncthreads();
      lbits = (prefix21 << 10) | s_tmp[44];
    }
    O3S_TSTAMP(5);
    if (mode & kModeCentroid) {  // the parked candidates with d2 <= limit join the sums (ties at the limit are all kept)
      if (m <= (uint32_t)kParkRecs) {
        // They were parked in arrival order; they are added in FLAT order so that repeated runs form the same sums:
        // thread r takes the parked record whose flat index has rank r among the m (m is a handful, at most 1 024).
        const CandRec* s_rec = reinterpret_cast<const CandRec*>(s_dyn + kParkBits);
        const uint32_t* s_flat = s_dyn + kParkBits + kParkRecs * 8;
        __syncthreads();  // s_bins is free again
        for (uint32_t j = threadIdx.x; j < m; j += kFinThreads) {
          const uint32_t fj = s_flat[j];
          uint32_t rank = 0;
          for (uint32_t i = 0; i < m; ++i) rank += s_flat[i] < fj ? 1u : 0u;
          s_bins[rank] = j;
        }
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < m; r += kFinThreads) {
          const CandRec rc = s_rec[s_bins[r]];
          if ((rc.keep & 1) && rc.bits <= lbits) {
            a[0] += (double)rc.px;
            a[1] += (double)rc.py;
            a[2] += (double)rc.pz;
            a[3] += (double)rc.qx;
            a[4] += (double)rc.qy;
            a[5] += (double)rc.qz;
            a[6] += 1.0;
          }
        }
      } else {  // heavy ties: a second sweep in flat order
        for (uint32_t f = threadIdx.x; f < total; f += kFinThreads) {
          const int b = flat_block(s_base, nb, f);
          const CandRec rc = cand[(size_t)b * kClsBlock + (f - s_base[b])];
          if ((rc.bits >> 10) == prefix21 && (rc.keep & 1) && rc.bits <= lbits) {
            a[0] += (double)rc.px;
            a[1] += (double)rc.py;
            a[2] += (double)rc.pz;
            a[3] += (double)rc.qx;
            a[4] += (double)rc.qy;
            a[5] += (double)rc.qz;
            a[6] += 1.0;
          }
        }
      }
    }
    limit = __uint_as_float(lbits);
  }
  O3S_TSTAMP(6);
  if (mode & kModeCentroid) {  // uniform
    __syncthreads();  // the selection is done with s_dyn
    Sum::run(a, s_a, s_b);
  }
  O3S_TSTAMP(7);
  // publish: lanes 0..5 each own one mean (fixed-order block sum of their component and of the count, one division);
  // lane 0 also owns limit / |K| / status.
  const int status = hdr_i(hv, H_STATUS);
  bool go_on = status == 0;
  if (status == 0 && (mode & kModeCentroid)) go_on = Sum::total(s_b, 6) != 0.0;  // uniform: every thread reads the same sum
  if (threadIdx.x < 6) {
    if (threadIdx.x == 0) {
      const float lim_out = (!cp.has_trim || !skip) ? limit : hdr_f(hv, H_LIMIT);
      if (publish && (!cp.has_trim || !skip)) st->limit = limit;
      s_out[0] = lim_out;
      if (s_ov) {
        s_ov->limit = limit;
        s_ov->has_limit = (!cp.has_trim || !skip) ? 1 : 0;
        s_ov->status = 0;
        s_ov->kept = hdr_i(hv, H_KEPT);
      }
    }
    if (s_ov) {  // means of an iteration that ends here: the state's stay
      if (threadIdx.x < 3) s_ov->mp[threadIdx.x] = hdr_f(hv, H_MP + threadIdx.x);
      else s_ov->mq[threadIdx.x - 3] = hdr_f(hv, H_MQ + threadIdx.x - 3);
    }
    if (status != 0) {
      if (publish && threadIdx.x == 0) st->done = 1;
    } else if (mode & kModeCentroid) {
      const double sk = Sum::total(s_b, threadIdx.x), K = Sum::total(s_b, 6);
      if (publish && threadIdx.x == 0) st->kept = (int32_t)K;
      if (s_ov && threadIdx.x == 0) s_ov->kept = (int32_t)K;
      if (K == 0.0) {  // "no point to minimize" (ErrorMinimizer.cpp:75-77)
        if (publish && threadIdx.x == 0) {
          st->status = 6;
          st->done = 1;
        }
        if (s_ov && threadIdx.x == 0) s_ov->status = 6;
      } else {  // rowwise().mean(): fp64 sums rounded once to fp32
        const float mean = (float)(sk / K);
        s_out[1 + threadIdx.x] = mean;
        if (publish) {
          if (threadIdx.x < 3) st->mp[threadIdx.x] = mean;
          else st->mq[threadIdx.x - 3] = mean;
        }
        if (s_ov) {
          if (threadIdx.x < 3) s_ov->mp[threadIdx.x] = mean;
          else s_ov->mq[threadIdx.x - 3] = mean;
        }
      }
    }
  }
  O3S_TSTAMP(8);
  return go_on;
}

__global__ void __launch_bounds__(kFinThreads) k_sel_finish(uint32_t* __restrict__ hist_rep, ChainParams cp, IcpState* __restrict__ st,
                                                            SelScratch* __restrict__ ss, const CandRec* __restrict__ cand,
                                                            const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ hist2,
                                                            uint32_t* __restrict__ base_scratch /*[nb + 1], used when nb > kBaseCap*/,
                                                            const double* __restrict__ part /*[7][nb]*/, int nb, int mode,
                                                            const double* __restrict__ part2 /*nullable: [7][nbp] of k_sel_partial*/, int nbp,
                                                            const CandRec* __restrict__ park_rec, const uint32_t* __restrict__ park_key) {
  __shared__ float s_out[8];
  const float hv = hdr_load(st);
  (void)sel_finish_body<false>(hist_rep, cp, st, ss, cand, cand_cnt, hist2, base_scratch, part, nb, mode, hv, s_out, part2, nbp, park_rec, park_key,
                               part2 ? &ss->seg_count[0] : nullptr);
  if (part2 && threadIdx.x == 0) ss->seg_count[0] = 0u;  // the parked-list counter of k_sel_partial, ready for the next iteration
}

// ------------------------------------------------------------------------------------------------------------------
// kept-pair predicate of k_normal_eq: product of the chain's binary weights
//   Trimmed: d2 <= limit   MaxDist: d2 <= max^2   SurfaceNormal / zero weight: encoded in pos   no match: pos == -1
// (LPM/OutlierFilter.cpp:64-103, LPM/ErrorMinimizer.cpp:98-108)
// ------------------------------------------------------------------------------------------------------------------
constexpr int kNePPT = 2;  // points per lane per trip of k_normal_eq

__device__ __forceinline__ bool kept_pair(int pe, float d, float limit, float max_out_r2) {
  return pe >= 0 && d <= limit && d <= max_out_r2;
}

// k_normal_eq — formulatePointMatchingConstraints (PointToPlane.cpp:108-156): G = [(p-mp) x n ; n], h = n.((p-mp)-(q-mq)),
// A = G G^T, b = -(G h^T).  Per-pair arithmetic is fp32 in the reference's order; the K-long sums are fp64.
__global__ void __launch_bounds__(kBlock) k_normal_eq(const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz, int N,
                                                      const float4* __restrict__ mq /*matched points*/, const float4* __restrict__ mn /*matched normals*/,
                                                      const int32_t* __restrict__ pos, const float* __restrict__ d2, ChainParams cp,
                                                      const IcpState* __restrict__ st, double* __restrict__ part /*[27][grid]*/,
                                                      uint32_t* __restrict__ hist_zero /*nullable: level-1 replicas to clear*/) {
  using Sum = BlockSum<kNeComps, kBlock>;
  __shared__ double s_a[Sum::kWordsA];
  __shared__ double s_b[Sum::kWordsB];
  O3S_TSTAMP(32);
  const float hv = hdr_load(st);
  if (hdr_i(hv, H_DONE)) return;
  // the level-1 histogram was consumed by k_classify; clearing it for the next k_match here spreads the stores over all
  // blocks of this kernel instead of loading them onto the one block of k_sel_finish
  if (hist_zero)
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < kHistReplicas * kHistBins + 1024; k += gridDim.x * kBlock) hist_zero[k] = 0u;  // + level 2
  O3S_TSTAMP(33);
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = hdr_f(hv, k);
  const float limit = hdr_f(hv, H_LIMIT);
  const float mpx = hdr_f(hv, H_MP), mpy = hdr_f(hv, H_MP + 1), mpz = hdr_f(hv, H_MP + 2);
  const float mqx = hdr_f(hv, H_MQ), mqy = hdr_f(hv, H_MQ + 1), mqz = hdr_f(hv, H_MQ + 2);
  double acc[kNeComps];
#pragma unroll
  for (int c = 0; c < kNeComps; ++c) acc[c] = 0.0;
  // two points per lane per trip; everything is a coalesced stream (the matched point was written by the matcher, the
  // matched normal by k_classify), so a trip is ONE memory round trip
  if (!O3S_CP_DBG(cp, 4))
  for (int base = blockIdx.x * (kBlock * kNePPT) + threadIdx.x; base < N; base += gridDim.x * (kBlock * kNePPT)) {
    int pe[kNePPT];
    float d[kNePPT], x0[kNePPT], y0[kNePPT], z0[kNePPT];
    bool keep[kNePPT];
#pragma unroll
    for (int u = 0; u < kNePPT; ++u) {
      const int i = base + u * kBlock;
      const bool in = i < N;
      pe[u] = in ? pos[i] : -1;
      d[u] = in ? d2[i] : kInfF;
      x0[u] = in ? rx[i] : 0.f;
      y0[u] = in ? ry[i] : 0.f;
      z0[u] = in ? rz[i] : 0.f;
    }
    float4 q[kNePPT], n[kNePPT];
#pragma unroll
    for (int u = 0; u < kNePPT; ++u) {
      const int i = min(base + u * kBlock, N - 1);
      q[u] = mq[i];
      n[u] = mn[i];
    }
#pragma unroll
    for (int u = 0; u < kNePPT; ++u) keep[u] = kept_pair(pe[u], d[u], limit, cp.max_out_r2);
#pragma unroll
    for (int u = 0; u < kNePPT; ++u) {
      if (!keep[u]) continue;
      const float px = xf_row(T, 0, x0[u], y0[u], z0[u]) - mpx, py = xf_row(T, 1, x0[u], y0[u], z0[u]) - mpy,
                  pz = xf_row(T, 2, x0[u], y0[u], z0[u]) - mpz;
      const float qx = q[u].x - mqx, qy = q[u].y - mqy, qz = q[u].z - mqz;
      float gv[6];
      gv[0] = py * n[u].z - pz * n[u].y;
      gv[1] = pz * n[u].x - px * n[u].z;
      gv[2] = px * n[u].y - py * n[u].x;
      gv[3] = n[u].x;
      gv[4] = n[u].y;
      gv[5] = n[u].z;
      const float ex = px - qx, ey = py - qy, ez = pz - qz;
      float h = 0.f;
      h = h + ex * n[u].x;
      h = h + ey * n[u].y;
      h = h + ez * n[u].z;
      int t = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = a; c < 6; ++c) acc[t++] += (double)(gv[a] * gv[c]);
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[21 + a] += (double)(gv[a] * h);
    }
  }
  O3S_TSTAMP(34);
  Sum::run(acc, s_a, s_b);
  O3S_TSTAMP(35);
  if (threadIdx.x < kNeComps) part[threadIdx.x * gridDim.x + blockIdx.x] = Sum::total(s_b, threadIdx.x);
  O3S_TSTAMP(36);
}

// ------------------------------------------------------------------------------------------------------------------
// The end of an iteration: reduce the 27 x nb block partials in block order, solve, build the step, update T_iter, run the
// checkers, write the state back and tell the host.  A block-wide device function (NT threads, all of them call it) so that
// it can run as a kernel of its own (k_solve) or as the tail of k_sel_ne in the block that stored its partials last.
//   LDS   SolveLds (BlockSum scratch, the 6x6 work area, the state staged through LDS: lane 0 then works on LDS only)
//   ov    (nullable) values this block formed itself in this launch and that supersede the state's: the fused kernel's blocks
//         do not publish limit / means / |K| one by one — the closing block writes them with the rest of the state
//   post  (nullable) host-coherent HostPost: when the chain is done, the whole state and then the word the host polls
//         (system-scope release) — instead of a copy command and a stream synchronisation
//   SC1   the partials were handed over INSIDE this launch (stored write-through by the other blocks): every load of them
//         bypasses this CU's L1 (relaxed agent-scope loads), which takes the place of an acquire fence
// ------------------------------------------------------------------------------------------------------------------
struct SolveLds {
  double s_a[BlockSum<kNeComps, kBlock>::kWordsA];
  double s_b[BlockSum<kNeComps, kBlock>::kWordsB];
  double s_sum[kNeComps];
  dev::SolveWork work;
  IcpState st;
  float Tn[16];
};
__device__ __forceinline__ void post_state(const IcpState* S /*LDS*/, const IcpState* __restrict__ st_global, HostPost* __restrict__ post) {
  // the FINAL state of a call, once: one wave's lanes store the state words, the wave's release fence covers them all, lane 0
  // stores the word the host polls.  Unfinished iterations post nothing (a store to host memory on every iteration's critical
  // path cost more than it saved): the host learns "not done yet" from the drained stream.
  constexpr int kWords = (int)(sizeof(IcpState) / 4);
  if (threadIdx.x < 64) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(&post->state);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(S);
    for (int k = threadIdx.x; k < kWords - kStateTailWords; k += 64) __hip_atomic_store(dst + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x < kStateTailWords)  // the matcher's statistics: complete since the launch boundary behind k_match2
      __hip_atomic_store(dst + kWords - kStateTailWords + threadIdx.x,
                         reinterpret_cast<const uint32_t*>(st_global)[kWords - kStateTailWords + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (threadIdx.x == 0) __hip_atomic_store(&post->word, post_word(S->call_seq, S->iter, 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int NT, bool SC1>
__device__ __forceinline__ void solve_body(const double* __restrict__ part, int nb, int N, const ChainParams& cp, IcpState* __restrict__ st,
                                           float* __restrict__ trace_T, float* __restrict__ trace_limit, int64_t* __restrict__ trace_kept,
                                           int trace_cap, int update_pose, HostPost* __restrict__ post, SolveLds& L, const SolveOverride* ov /*LDS or null*/) {
  using Sum = BlockSum<kNeComps, kBlock>;
  static_assert(NT >= kBlock, "the partial sums are folded by the first kBlock threads");
  constexpr int kWords = (int)(sizeof(IcpState) / 4);
  IcpState& s_st = L.st;
  O3S_TSTAMP(16);
  for (int k = threadIdx.x; k < kWords; k += NT) reinterpret_cast<uint32_t*>(&s_st)[k] = reinterpret_cast<const uint32_t*>(st)[k];
  // partials: thread t takes the 27 sums of block t (and of block t + 256 for readings beyond 131 k points).  The loads are
  // branch-free (clamped address, value masked afterwards: a predicated load compiles to an exec-mask region with its own
  // s_waitcnt, i.e. one memory round trip per component) and coalesced along the block index; the 27 x 256 values are
  // then summed through LDS in a fixed order.
  const bool sums_formed = part == nullptr;  // the sharded chain's front has put the 27 sums into L.s_sum (and says so through `ov`)
  if (!sums_formed) {
    static_assert(kMaxPartialBlocks <= 2 * kBlock, "two partial blocks per thread at most");
    const int nbm1 = nb > 0 ? nb - 1 : 0;
    const int t = threadIdx.x;
    double v[kNeComps];
#pragma unroll
    for (int c = 0; c < kNeComps; ++c) v[c] = 0.0;
    if (t < kBlock) {
      auto ld = [&](int idx) -> double {
        if (SC1) return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(part) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        return part[idx];
      };
#pragma unroll
      for (int c = 0; c < kNeComps; ++c) v[c] = ld(c * nb + min(t, nbm1));
#pragma unroll
      for (int c = 0; c < kNeComps; ++c) v[c] = t < nb ? v[c] : 0.0;
      if (nb > kBlock) {  // uniform
        double u[kNeComps];
#pragma unroll
        for (int c = 0; c < kNeComps; ++c) u[c] = ld(c * nb + min(t + kBlock, nbm1));
#pragma unroll
        for (int c = 0; c < kNeComps; ++c) v[c] += (t + kBlock < nb) ? u[c] : 0.0;
      }
    }
    O3S_TSTAMP(24);
    O3S_TSTAMP(25);
    if (NT == kBlock) Sum::run(v, L.s_a, L.s_b);
    else Sum::run_first(v, L.s_a, L.s_b);
    if (t < kNeComps) L.s_sum[t] = Sum::total(L.s_b, t);
  }
  if (ov && threadIdx.x == 0) {  // what this launch formed: the state's copy is a launch old (barrier above: the staging is complete)
    if (ov->has_limit) s_st.limit = ov->limit;
    s_st.kept = ov->kept;
    for (int d = 0; d < 3; ++d) {
      s_st.mp[d] = ov->mp[d];
      s_st.mq[d] = ov->mq[d];
    }
    if (ov->status != 0) s_st.status = ov->status;
  }
  __syncthreads();
  // the 6x6 system in fp32 (sums rounded once), for the solver and for the state: 42 lanes in parallel instead of one
  if (sums_formed && !ov) {
    // nothing was formed in this launch (the chain ended earlier): the state's A and b stay as they are
  } else if (threadIdx.x < 36) {
    const int a = threadIdx.x / 6, c = threadIdx.x % 6;
    const int lo = a < c ? a : c, hi = a < c ? c : a;
    const float v = (float)L.s_sum[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];  // index in the row-major upper triangle
    L.work.S.A[a][c] = v;
    s_st.A[c * 6 + a] = v;
  } else if (threadIdx.x < 42) {
    const int a = threadIdx.x - 36;
    const float v = -(float)L.s_sum[21 + a];
    L.work.S.b[a] = v;
    s_st.b[a] = v;
  }
  __syncthreads();
  O3S_TSTAMP(17);
  if (s_st.done) {  // uniform.  Set by an EARLIER kernel of this iteration (an error found there) and not told yet: tell the host
    if (post && !s_st.posted) {
      if (threadIdx.x == 0) {
        st->posted = 1u;
        st->t_end = wall_clock64();
        s_st.posted = 1u;
        s_st.t_end = st->t_end;
      }
      __syncthreads();
      post_state(&s_st, st, post);
    }
    return;
  }
  const bool failed = s_st.status != 0;  // uniform: the state sits in LDS
  O3S_TSTAMP(18);
  // the fast path's two halves side by side: lane 0 factors and solves, lane 64 (the next wave) factors again and bounds the
  // condition number — x is used only if both say yes, else lane 0 runs the reference's general sequence
  if (!failed && (threadIdx.x == 0 || threadIdx.x == 64)) {
    dev::SolveWork& W = L.work;
    float Ar[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) Ar[r][c] = W.S.A[r][c];
    if (threadIdx.x == 0) {
      float br[6], xr[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) br[r] = W.S.b[r];
      W.fast_solved = dev::llt_fast_solve(Ar, br, xr) ? 1 : 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) W.xfast[r] = xr[r];
    } else {
      W.fast_bounded = dev::llt_fast_bound(Ar) ? 1 : 0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    IcpState* S = &s_st;
    if (failed) {
      S->done = 1;
    } else {
      dev::SolveWork& W = L.work;
      int branch = 0;
      if (O3S_CP_DBG(cp, 8)) {
        branch = 0;
      } else if (W.fast_solved && W.fast_bounded) {
#pragma unroll
        for (int r = 0; r < 6; ++r) W.x[r] = W.xfast[r];
      } else {
        branch = dev::solve_sys6_general(W);
      }
      O3S_TSTAMP(19);
      const float* x = W.x;
      float* dT = S->dT;
      if (O3S_CP_DBG(cp, 16)) { for (int k = 0; k < 16; ++k) dT[k] = (k % 5 == 0) ? 1.f : 0.f; } else dev::build_step(x, S->mp, S->mq, dT);
      for (int a = 0; a < 6; ++a) S->x[a] = x[a];
      S->solve_branch = branch;
      S->point_used_ratio = (float)S->kept / (float)N;   // ErrorMinimizer.cpp:139
      S->weighted_ratio = (float)S->kept / (float)N;     // binary weights: sum w == |K| (ErrorMinimizer.cpp:140)
      if (!update_pose) {
        S->iter += 1;
        S->done = 1;
      }
    }
  }
  __syncthreads();
  if (!failed && update_pose) {  // uniform
    // T_iter <- dT * T_iter (LPM/ICP.cpp:433-434): one lane per entry, each with mul4's operation order
    const int it = s_st.iter;
    if (threadIdx.x < 16) {
      const int r = threadIdx.x & 3, c = threadIdx.x >> 2;
      const float* A = s_st.dT;
      const float* B = s_st.T_iter;
      float v = A[0 * 4 + r] * B[c * 4 + 0];
      v = v + A[1 * 4 + r] * B[c * 4 + 1];
      v = v + A[2 * 4 + r] * B[c * 4 + 2];
      v = v + A[3 * 4 + r] * B[c * 4 + 3];
      L.Tn[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
      s_st.T_iter[threadIdx.x] = L.Tn[threadIdx.x];
      if (it < trace_cap) trace_T[it * 16 + threadIdx.x] = L.Tn[threadIdx.x];
    } else if (threadIdx.x == 16 && it < trace_cap) {
      trace_limit[it] = cp.has_trim ? s_st.limit : __builtin_nanf("");
      trace_kept[it] = (int64_t)s_st.kept;
    }
    if (threadIdx.x == 0) {
      IcpState* S = &s_st;
      bool iterate = true;
      O3S_TSTAMP(20);
      int status = dev::run_checkers(S, cp, L.Tn, &iterate);
      O3S_TSTAMP(21);
      S->iter = it + 1;
      // the next iteration starts with transformations.apply(stepReading, T_iter) -> checkParameters (TransformationsImpl.cpp:73-74)
      if (status == 0 && iterate && !dev::rigid_ok(L.Tn)) status = 8;
      if (status != 0) {
        S->status = status;
        S->done = 1;
      } else if (!iterate) {
        S->done = 1;
      }
    }
  }
  if (threadIdx.x == 0 && s_st.done && post) {  // lane 0 wrote `done` itself
    s_st.posted = 1u;
    s_st.t_end = wall_clock64();
  }
  __syncthreads();
  O3S_TSTAMP(22);
  // cand_count / row_count (the last words) are only ever touched by k_match's atomics: leave them alone
  for (int k = threadIdx.x; k < kWords - kStateTailWords; k += NT) reinterpret_cast<uint32_t*>(st)[k] = reinterpret_cast<const uint32_t*>(&s_st)[k];
  if (post && s_st.done) post_state(&s_st, st, post);
  O3S_TSTAMP(23);
}

// k_solve — the closing step as a launch of its own (the two-kernel chain of large readings and of batches, the sharded chain,
// the module-level minimiser)
__global__ void __launch_bounds__(kBlock) k_solve(const double* __restrict__ part, int nb, int N, ChainParams cp, IcpState* __restrict__ st,
                                                  float* __restrict__ trace_T, float* __restrict__ trace_limit, int64_t* __restrict__ trace_kept,
                                                  int trace_cap, int update_pose, HostPost* __restrict__ post) {
  __shared__ SolveLds lds;
  solve_body<kBlock, false>(part, nb, N, cp, st, trace_T, trace_limit, trace_kept, trace_cap, update_pose, post, lds, nullptr);
}

// k_sel_ne = k_sel_finish + k_normal_eq in one launch, for readings whose normal equations fit ONE generation of blocks
// (<= kFusedMaxBlocks): every block first repeats the (small) exact selection for itself — sel_finish_body<true>; costs
// ~0.5 us more than one block doing it alone, measured with 196 redundant blocks — and then accumulates its share of the
// 27 sums with the limit and the means it has just formed.  Saves the second kernel's start-up round trip and the hand-over
// through the state header.  This block's points are requested before the selection starts, so their round trip hides
// behind it.  The level-1 replicas are cleared here as k_normal_eq does; the level-2 histogram — still being read by other
// blocks of this launch — is cleared by block 0 of the next k_match2.
constexpr int kFusedMaxBlocks = 256;   // one block per CU (the selection's LDS plan fills most of a CU's LDS)
// The normal-equation half uses the first kBlock (256) threads with kNePPT points each and the same fixed-order block sum
// as k_normal_eq: with the same number of blocks the 27 x blocks partials — and therefore the pose — are bit-identical to the
// two-kernel chain's (o3s_icp_compute_batch runs that one; a pair must not depend on how it was issued).
// TAIL: the block that stores its partials last closes the iteration itself (solve_body; measured slower, LAB_NOTES_r04.md 2: only
// the hooks build launches it); without it k_solve follows, and the instantiation carries none of the closing step's registers.
template <bool TAIL>
__global__ void __launch_bounds__(kFinThreads) k_sel_ne(ChainParams cp, IcpState* __restrict__ st, SelScratch* __restrict__ ss,
                                                        const CandRec* __restrict__ cand, const uint32_t* __restrict__ cand_cnt,
                                                        const uint32_t* __restrict__ hist2, uint32_t* __restrict__ base_scratch,
                                                        const double* __restrict__ part_cent /*[7][nb_cls]*/, int nb_cls, int mode,
                                                        const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz, int N,
                                                        const float4* __restrict__ mq, const float4* __restrict__ mn, const int32_t* __restrict__ pos,
                                                        const float* __restrict__ d2, double* __restrict__ part_ne /*[27][grid]*/,
                                                        uint32_t* __restrict__ hist_zero /*level-1 replicas*/,
                                                        float* __restrict__ trace_T, float* __restrict__ trace_limit, int64_t* __restrict__ trace_kept,
                                                        int trace_cap, HostPost* __restrict__ post) {
  constexpr bool tail = TAIL;
  extern __shared__ __align__(16) uint32_t s_dyn[];
  __shared__ float s_out[8];
  __shared__ SolveOverride s_ov;
  __shared__ uint32_t s_ticket;
  static_assert(sizeof(SolveLds) <= kSelCap * 4, "the closing step borrows the selection buffer");
  using Sum = BlockSum<kNeComps, kBlock>;
  static_assert((Sum::kWordsA + Sum::kWordsB) * 8 <= kSelCap * 4, "the 27-component block sum borrows the selection buffer");
  static_assert(kFinThreads >= kBlock, "the normal-equation half runs on the first kBlock threads");
  const float hv = hdr_load(st);
  const bool worker = threadIdx.x < kBlock;
  int pe[kNePPT];
  float d[kNePPT], x0[kNePPT], y0[kNePPT], z0[kNePPT];
  float4 q[kNePPT], n[kNePPT];
#pragma unroll
  for (int u = 0; u < kNePPT; ++u) {  // this block's points: same assignment as k_normal_eq (one trip: gridDim covers N)
    const int i = blockIdx.x * (kBlock * kNePPT) + u * kBlock + (worker ? threadIdx.x : 0);
    const bool in = worker && i < N;
    const int ic = i < N ? i : N - 1;
    pe[u] = in ? pos[ic] : -1;
    d[u] = in ? d2[ic] : kInfF;
    x0[u] = rx[ic];
    y0[u] = ry[ic];
    z0[u] = rz[ic];
    q[u] = mq[ic];
    n[u] = mn[ic];
  }
  if (hdr_i(hv, H_DONE)) return;  // the chain has ended in an earlier iteration: nothing to close
  const bool go_on = sel_finish_body<true>(nullptr, cp, st, ss, cand, cand_cnt, hist2, base_scratch, part_cent, nb_cls, mode, hv, s_out, nullptr, 0, nullptr,
                                           nullptr, nullptr, &s_ov, tail != 0);
  __syncthreads();     // s_out / s_ov are complete, the selection is done with s_dyn
  if (!go_on) {        // uniform, and the same in every block: an earlier error, or no pair kept — block 0 closes the iteration
    if constexpr (TAIL)
      if (blockIdx.x == 0)
        solve_body<kFinThreads, false>(part_ne, (int)gridDim.x, N, cp, st, trace_T, trace_limit, trace_kept, trace_cap, 1, post,
                                     *reinterpret_cast<SolveLds*>(s_dyn), &s_ov);
    return;
  }
  if (hist_zero)
    for (int k = blockIdx.x * kFinThreads + threadIdx.x; k < kHistReplicas * kHistBins; k += gridDim.x * kFinThreads) hist_zero[k] = 0u;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = hdr_f(hv, k);
  const float limit = s_out[0];
  const float mpx = s_out[1], mpy = s_out[2], mpz = s_out[3], mqx = s_out[4], mqy = s_out[5], mqz = s_out[6];
  double acc[kNeComps];
#pragma unroll
  for (int c = 0; c < kNeComps; ++c) acc[c] = 0.0;
#pragma unroll
  for (int u = 0; u < kNePPT; ++u) {
    if (!kept_pair(pe[u], d[u], limit, cp.max_out_r2)) continue;  // pe = -1 on the threads beyond kBlock: nothing kept
    const float px = xf_row(T, 0, x0[u], y0[u], z0[u]) - mpx, py = xf_row(T, 1, x0[u], y0[u], z0[u]) - mpy,
                pz = xf_row(T, 2, x0[u], y0[u], z0[u]) - mpz;
    const float qx = q[u].x - mqx, qy = q[u].y - mqy, qz = q[u].z - mqz;
    float gv[6];
    gv[0] = py * n[u].z - pz * n[u].y;
    gv[1] = pz * n[u].x - px * n[u].z;
    gv[2] = px * n[u].y - py * n[u].x;
    gv[3] = n[u].x;
    gv[4] = n[u].y;
    gv[5] = n[u].z;
    const float ex = px - qx, ey = py - qy, ez = pz - qz;
    float h = 0.f;
    h = h + ex * n[u].x;
    h = h + ey * n[u].y;
    h = h + ez * n[u].z;
    int t = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = a; c < 6; ++c) acc[t++] += (double)(gv[a] * gv[c]);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += (double)(gv[a] * h);
  }
  double* s_a = reinterpret_cast<double*>(s_dyn);
  double* s_b = s_a + Sum::kWordsA;
  Sum::run_first(acc, s_a, s_b);
  // ---- hand-over inside the launch (cdna_hip_programming.md, Guideline 16, counter form): the 27 partials go out write-through
  //      (agent-scope stores), every wave drains its stores, the block's barrier, ONE relaxed agent-scope add to the ticket; the
  //      block that draws the last ticket reads all partials with L1-bypassing loads (solve_body<.., true>), folds them in block
  //      order — the order k_solve folds them in, same bits — and closes the iteration: one launch boundary less per iteration.
  if constexpr (!TAIL) {  // k_solve folds the partials behind the launch boundary
    if (threadIdx.x < kNeComps) part_ne[threadIdx.x * gridDim.x + blockIdx.x] = Sum::total(s_b, threadIdx.x);
    return;
  } else {
  if (threadIdx.x < kNeComps)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(part_ne) + threadIdx.x * gridDim.x + blockIdx.x,
                       (unsigned long long)__double_as_longlong(Sum::total(s_b, threadIdx.x)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&ss->ne_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_ticket != gridDim.x - 1u) return;  // uniform
  if (threadIdx.x == 0) __hip_atomic_store(&ss->ne_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next iteration (k_read_prep zeroes it per call)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the compiler from moving the loads below above the ticket
  solve_body<kFinThreads, true>(part_ne, (int)gridDim.x, N, cp, st, trace_T, trace_limit, trace_kept, trace_cap, 1, post,
                                *reinterpret_cast<SolveLds*>(s_dyn), &s_ov);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// module-level helpers
// ------------------------------------------------------------------------------------------------------------------
// internal (pos, d2) in sorted-query order -> API (ids, dists) in caller order  (Matches, LPM/PointMatcher.h:444-464)
__global__ void __launch_bounds__(kBlock) k_export_matches(int N, const int32_t* __restrict__ pos, const float* __restrict__ d2,
                                                           const float4* __restrict__ ref, const int32_t* __restrict__ perm,
                                                           int32_t* __restrict__ ids, float* __restrict__ dists) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int pe = pos[i];
  const int o = perm[i];
  const int slot = pe >= 0 ? pe : (pe <= -2 ? -2 - pe : -1);
  ids[o] = slot >= 0 ? __float_as_int(ref[slot].w) : -1;
  dists[o] = d2[i];
}

// caller-supplied matches (+ optional weights) -> internal encoding, identity query order
__global__ void __launch_bounds__(kBlock) k_import_matches(int N, const int32_t* __restrict__ ids, const float* __restrict__ dists,
                                                           const float* __restrict__ weights /*nullable*/, const int32_t* __restrict__ orig_to_sorted,
                                                           int64_t M, const float4* __restrict__ ref, int32_t* __restrict__ pos, float* __restrict__ d2,
                                                           float4* __restrict__ mq) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int id = ids[i];
  const float d = dists[i];
  int pe = -1;
  if (id >= 0 && (int64_t)id < M) {
    pe = orig_to_sorted[id];
    if (weights && weights[i] == 0.0f) pe = -2 - pe;
  }
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (id >= 0 && (int64_t)id < M) {
    const float4 r = ref[orig_to_sorted[id]];
    q = make_float4(r.x, r.y, r.z, 1.f);
  }
  mq[i] = q;  // the chain's kernels stream the matched point instead of gathering it
  if (d == kInfF) pe = pe >= 0 ? -2 - pe : pe;  // ErrorElements skips infinite distances (ErrorMinimizer.cpp:103-105)
  pos[i] = pe;
  d2[i] = d;
}

// OutlierFilters::compute for the configured chain on caller-supplied matches; reading normals already rotated
__global__ void __launch_bounds__(kBlock) k_weights(int N, const int32_t* __restrict__ pos, const float* __restrict__ d2,
                                                    const float* __restrict__ read_n /*3xN AoS or null*/, const float4* __restrict__ refn,
                                                    ChainParams cp, const IcpState* __restrict__ st, int any_filter, float* __restrict__ w) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int pe = pos[i];
  const float d = d2[i];
  if (!any_filter) {
    w[i] = (d == kInfF) ? 0.f : 1.f;
    return;
  }
  float wi = 1.f;
  if (cp.max_out_r2 != kInfF) wi = wi * ((d <= cp.max_out_r2) ? 1.f : 0.f);
  if (cp.has_trim) wi = wi * ((d <= st->limit) ? 1.f : 0.f);
  if (cp.has_normal_gate) {
    float g = 0.f;
    const int slot = pe >= 0 ? pe : (pe <= -2 ? -2 - pe : -1);
    if (slot >= 0) {
      const float4 rn = refn[slot];
      float v = read_n[3 * i] * rn.x;
      v = v + read_n[3 * i + 1] * rn.y;
      v = v + read_n[3 * i + 2] * rn.z;
      g = (v < cp.cos_max_angle) ? 0.f : 1.f;
    }
    wi = wi * g;
  }
  w[i] = wi;
}

// AoS 4xN -> SoA without transform (module-level entry points take data already in the <refMean> frame)
__global__ void __launch_bounds__(kBlock) k_aos_to_soa(const float4* __restrict__ in, int N, float* __restrict__ x, float* __restrict__ y,
                                                       float* __restrict__ z) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const float4 p = in[i];
  x[i] = p.x;
  y[i] = p.y;
  z[i] = p.z;
}

}  // namespace kern
}  // namespace o3s
