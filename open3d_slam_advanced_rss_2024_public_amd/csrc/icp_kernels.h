// icp_kernels.h — hand-written gfx950 (CDNA4, wave64) kernels of the scan-to-map ICP iteration chain.
//
// One ICP iteration = 6 launches on one stream (no host round trip; a `done` flag in IcpState turns the remaining
// launches of a pre-recorded chain into no-ops):
//   k_match      transform reading point by T_iter, exact 1-NN in the voxel grid, normal-angle gate, d2 histogram
//   k_sel_*      exact k-th smallest finite d2 (TrimmedDistOutlierFilter limit) by radix selection (2 launches)
//   k_centroid   sum p, sum q, |K| over kept pairs (fp64 partials per block)
//   k_normal_eq  27 fp64 partial sums of G G^T / G h per block (centred in fp32 exactly like the reference)
//   k_solve      reduce partials, 6x6 solve, SE(3) step, T_iter update, stop rules
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
constexpr float kInfF = __builtin_huge_valf();

// ------------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
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
__global__ void __launch_bounds__(kBlock) k_ref_stats(const float4* __restrict__ xyzw, int64_t M, double* __restrict__ part /*[grid][3]*/,
                                                      float* __restrict__ bb /*[grid][6]*/) {
  double s0 = 0, s1 = 0, s2 = 0;
  float lo0 = kInfF, lo1 = kInfF, lo2 = kInfF, hi0 = -kInfF, hi1 = -kInfF, hi2 = -kInfF;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += (int64_t)gridDim.x * kBlock) {
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
  orig_to_sorted[i] = (int32_t)pos;
}

// ------------------------------------------------------------------------------------------------------------------
// exclusive scan of uint32 counts (3 launches: block sums, scan of block sums, add back).  out has n + 1 entries.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kScanItems = 8;                       // items per thread
constexpr int kScanTile = kBlock * kScanItems;      // 2048 items per block

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* sh /*>= 17 words*/) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (l >= off) inc += t;
  }
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

__global__ void __launch_bounds__(kBlock) k_scan_block_sums(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ sums) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  uint32_t s = 0;
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < n) s += in[i];
  }
  s = wave_sum_u32(s);
  __shared__ uint32_t sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// single block: exclusive scan of the block sums in place (serial over tiles of 1024)
__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t* __restrict__ sums, int64_t nb) {
  __shared__ uint32_t sh[32];
  uint32_t carry = 0;
  for (int64_t base = 0; base < nb; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const uint32_t v = i < nb ? sums[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, &tot, sh);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kBlock) k_scan_apply(const uint32_t* __restrict__ in, int64_t n, const uint32_t* __restrict__ sums,
                                                       uint32_t* __restrict__ out /* n + 1 */) {
  __shared__ uint32_t sh[32];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0u;
    s += v[k];
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
__global__ void __launch_bounds__(kBlock) k_read_prep(const float4* __restrict__ in_xyzw, const float* __restrict__ in_n /*3xN AoS or null*/,
                                                      int N, const float* __restrict__ T0 /*16, device*/, GridParams g,
                                                      float* __restrict__ tx, float* __restrict__ ty, float* __restrict__ tz,
                                                      float* __restrict__ tnx, float* __restrict__ tny, float* __restrict__ tnz,
                                                      uint32_t* __restrict__ cell_of, uint32_t* __restrict__ counts /*null: no sort*/) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const float4 p = in_xyzw[i];
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = T0[k];
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
  if (counts) {
    const int cx = cell_coord(x, g.ox, g.inv_cell, g.nx);
    const int cy = cell_coord(y, g.oy, g.inv_cell, g.ny);
    const int cz = cell_coord(z, g.oz, g.inv_cell, g.nz);
    const uint32_t lin = ((uint32_t)cz * (uint32_t)g.ny + (uint32_t)cy) * (uint32_t)g.nx + (uint32_t)cx;
    cell_of[i] = lin;
    atomicAdd(&counts[lin], 1u);
  }
}

__global__ void __launch_bounds__(kBlock) k_read_scatter(int N, const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ start,
                                                         uint32_t* __restrict__ fill, const float* __restrict__ tx, const float* __restrict__ ty,
                                                         const float* __restrict__ tz, const float* __restrict__ tnx, const float* __restrict__ tny,
                                                         const float* __restrict__ tnz, int has_n, float* __restrict__ rx, float* __restrict__ ry,
                                                         float* __restrict__ rz, float* __restrict__ rnx, float* __restrict__ rny,
                                                         float* __restrict__ rnz, int32_t* __restrict__ perm /* sorted slot -> original index */) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const uint32_t c = cell_of[i];
  const uint32_t pos = start[c] + atomicAdd(&fill[c], 1u);
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
// k_match — Matcher::findClosests fused with the step transform and the SurfaceNormalOutlierFilter.
//   LPM/ICP.cpp:401-413 (copy + transform + match), LPM/MatchersImpl.cpp:117-132, LPM/OutlierFiltersImpl.cpp:236-281.
// EIGHT lanes cooperate on one reading point (8 points per wave64): the work of one query is a handful of dependent
// gathers, so spreading it over lanes shortens the latency chain 8x and fills the chip (100k points -> 12.5k waves).
//   phase 1  the centre row of the 3x3x3 cell block: cells (cx-1..cx+1, cy, cz) are ONE contiguous [begin,end) range of
//            the cell-sorted reference; its candidates are dealt round-robin to the 8 lanes (coalesced 16-byte loads);
//   phase 2  the 8 remaining (dz,dy) rows of the block, one row per lane, skipped when the row's lower-bound distance
//            already exceeds the best of phase 1;
//   phase 3  Chebyshev rings r >= 2 (rows dealt round-robin) until the ring's lower bound exceeds min(best, maxDist^2)
//            or the ring leaves the grid — only far / unmatched points get here.
// After each phase the group's (d2, original index, slot) minimum is combined with 3 xor-shuffles; ties keep the lowest
// original index, so the result is independent of lane assignment and of the order inside a cell.
// Output per point: d2 (squared fp32 distance, +inf = none) and pos = slot in the sorted reference, -1 = none,
// -2 - slot = matched but rejected by the normal gate (its distance still takes part in the trim quantile).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kGroup = 8;                       // lanes per query
constexpr int kTileQ = kBlock / kGroup;         // queries per block pass (32)
constexpr int kMatchMaxBlocks = 2048;           // 8 resident blocks per CU

__device__ __forceinline__ float cell_gap(int d, float l, float cell, float margin) {
  float gap = 0.f;
  if (d > 0) gap = (float)d * cell - l;
  else if (d < 0) gap = l + (float)(-d - 1) * cell;
  gap -= margin;
  return gap > 0.f ? gap : 0.f;
}

struct Best {
  float d;
  int idx;
  int pos;
};

__device__ __forceinline__ void best_take(Best& b, float d, int qi, int j, float lim) {
  if (d <= lim && (d < b.d || (d == b.d && qi < b.idx))) {
    b.d = d;
    b.idx = qi;
    b.pos = j;
  }
}

// minimum over the 8 lanes of a group, every lane ends with the group's winner
__device__ __forceinline__ void group_min(Best& b) {
#pragma unroll
  for (int m = 1; m < kGroup; m <<= 1) {
    const float od = __shfl_xor(b.d, m, 64);
    const int oi = __shfl_xor(b.idx, m, 64);
    const int op = __shfl_xor(b.pos, m, 64);
    if (od < b.d || (od == b.d && oi < b.idx)) {
      b.d = od;
      b.idx = oi;
      b.pos = op;
    }
  }
}

template <bool STATS>
__global__ void __launch_bounds__(kBlock) k_match(const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
                                                  const float* __restrict__ rnx, const float* __restrict__ rny, const float* __restrict__ rnz,
                                                  int N, const float4* __restrict__ ref, const float4* __restrict__ refn,
                                                  const uint32_t* __restrict__ cell_start, const int32_t* __restrict__ orig_to_sorted,
                                                  const int32_t* __restrict__ perm, GridParams g, ChainParams cp, IcpState* __restrict__ st,
                                                  int32_t* __restrict__ pos_out, float* __restrict__ d2_out, uint32_t* __restrict__ hist) {
  if (st->done) return;
  __shared__ uint32_t s_hist[kHistBins];
  for (int k = threadIdx.x; k < kHistBins; k += kBlock) s_hist[k] = 0u;
  __syncthreads();
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T_iter[k];

  const int sub = threadIdx.x & (kGroup - 1);
  const int qib = threadIdx.x >> 3;  // query within the tile
  const int ntiles = (N + kTileQ - 1) / kTileQ;
  // XCD-aware: gridDim.x is a multiple of 8; logical block = (b % 8) * (grid / 8) + b / 8 walks a contiguous tile range,
  // so the blocks that share an XCD (and its L2) cover one compact part of the spatially sorted reading.
  const int lb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tpb = (ntiles + gridDim.x - 1) / gridDim.x;
  const float lim = g.max_r2;
  unsigned long long n_cand = 0, n_rows = 0;

  for (int tile = lb * tpb; tile < min((lb + 1) * tpb, ntiles); ++tile) {
    const int i = tile * kTileQ + qib;
    const bool valid = i < N;
    const float px = valid ? rx[i] : 0.f, py = valid ? ry[i] : 0.f, pz = valid ? rz[i] : 0.f;
    const float sx = xf_row(T, 0, px, py, pz), sy = xf_row(T, 1, px, py, pz), sz = xf_row(T, 2, px, py, pz);
    Best b{kInfF, 0x7fffffff, -1};
    bool active = valid && !cp.mirror;
    int cx = 0, cy = 0, cz = 0, r = 2, rmax = 0;
    float lx = 0.f, ly = 0.f, lz = 0.f, m = 0.f;
    if (valid && cp.mirror) {  // MirrorMatcher (LPM/MatchersImpl.cpp:65-85): id = i, dist = 0
      b.pos = orig_to_sorted[perm[i]];
      b.idx = 0;
      b.d = 0.f;
    }
    if (active) {
      const float big = 1.0e9f;  // clamp far-away queries: the int conversion cannot overflow, bounds stay lower bounds
      cx = (int)floorf(fminf(fmaxf((sx - g.ox) * g.inv_cell, -big), big));
      cy = (int)floorf(fminf(fmaxf((sy - g.oy) * g.inv_cell, -big), big));
      cz = (int)floorf(fminf(fmaxf((sz - g.oz) * g.inv_cell, -big), big));
      lx = fminf(fmaxf((sx - g.ox) - (float)cx * g.cell, 0.f), g.cell);
      ly = fminf(fmaxf((sy - g.oy) - (float)cy * g.cell, 0.f), g.cell);
      lz = fminf(fmaxf((sz - g.oz) - (float)cz * g.cell, 0.f), g.cell);
      m = fminf(fminf(fminf(lx, g.cell - lx), fminf(ly, g.cell - ly)), fminf(lz, g.cell - lz));
      int r0 = 0;
      r0 = max(r0, max(-cx, cx - (g.nx - 1)));
      r0 = max(r0, max(-cy, cy - (g.ny - 1)));
      r0 = max(r0, max(-cz, cz - (g.nz - 1)));
      rmax = max(max(cx, g.nx - 1 - cx), max(max(cy, g.ny - 1 - cy), max(cz, g.nz - 1 - cz)));
      r = max(2, r0);
      const int xa = max(cx - 1, 0), xb = min(cx + 1, g.nx - 1);
      // ---- phase 1: centre row, candidates dealt to the 8 lanes ----
      if (xa <= xb && cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz) {
        const uint32_t rowbase = ((uint32_t)cz * (uint32_t)g.ny + (uint32_t)cy) * (uint32_t)g.nx;
        const uint32_t jb = cell_start[rowbase + (uint32_t)xa], je = cell_start[rowbase + (uint32_t)xb + 1u];
        for (uint32_t j = jb + (uint32_t)sub; j < je; j += kGroup) {
          const float4 q = ref[j];
          best_take(b, dist2(sx, sy, sz, q.x, q.y, q.z), __float_as_int(q.w), (int)j, lim);
        }
        if (STATS && sub == 0) {
          n_rows += 1;
          n_cand += (unsigned long long)(je - jb);
        }
      }
    }
    group_min(b);
    if (cp.dbg & 2) active = false;
    if (active) {
      // ---- phase 2: the 8 neighbour rows of the 3x3x3 block, one per lane ----
      const int t = sub < 4 ? sub : sub + 1;
      const int dz = t / 3 - 1, dy = t % 3 - 1;
      const int z = cz + dz, y = cy + dy;
      const int xa = max(cx - 1, 0), xb = min(cx + 1, g.nx - 1);
      if (xa <= xb && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
        const float gz = cell_gap(dz, lz, g.cell, g.margin), gy = cell_gap(dy, ly, g.cell, g.margin);
        if (!(gz * gz + gy * gy > fminf(b.d, lim))) {
          const uint32_t rowbase = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
          const uint32_t jb = cell_start[rowbase + (uint32_t)xa], je = cell_start[rowbase + (uint32_t)xb + 1u];
          for (uint32_t j = jb; j < je; j += 2) {
            const uint32_t j1 = min(j + 1u, je - 1u);  // pairs of independent loads; a duplicate test is harmless
            const float4 q0 = ref[j];
            const float4 q1 = ref[j1];
            best_take(b, dist2(sx, sy, sz, q0.x, q0.y, q0.z), __float_as_int(q0.w), (int)j, lim);
            best_take(b, dist2(sx, sy, sz, q1.x, q1.y, q1.z), __float_as_int(q1.w), (int)j1, lim);
          }
          if (STATS) {
            n_rows += 1;
            n_cand += (unsigned long long)(je - jb);
          }
        }
      }
    }
    group_min(b);
    // ---- phase 3: rings r >= 2 ----
    if (active) {
      const float lb2 = (float)(r - 1) * g.cell + m - g.margin;
      if (r > rmax || (lb2 > 0.f && lb2 * lb2 > fminf(b.d, lim))) active = false;
    }
    while (__any(active)) {
      if (active) {
        const int side = 2 * r + 1;
        for (int t = sub; t < side * side; t += kGroup) {
          const int dz = t / side - r, dy = t % side - r;
          const int z = cz + dz, y = cy + dy;
          if (z < 0 || z >= g.nz || y < 0 || y >= g.ny) continue;
          const float gz = cell_gap(dz, lz, g.cell, g.margin), gy = cell_gap(dy, ly, g.cell, g.margin);
          if (gz * gz + gy * gy > fminf(b.d, lim)) continue;
          const bool full = (dz == r) || (dz == -r) || (dy == r) || (dy == -r);
          const uint32_t rowbase = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
          const int nseg = full ? 1 : 2;  // face row: one range [cx-r, cx+r]; interior row: the two end cells only
          for (int sgi = 0; sgi < nseg; ++sgi) {
            int xa, xb;
            if (full) {
              xa = max(cx - r, 0);
              xb = min(cx + r, g.nx - 1);
            } else {
              xa = xb = (sgi == 0) ? cx - r : cx + r;
              if (xa < 0 || xa >= g.nx) continue;
            }
            if (xa > xb) continue;
            const uint32_t jb = cell_start[rowbase + (uint32_t)xa], je = cell_start[rowbase + (uint32_t)xb + 1u];
            for (uint32_t j = jb; j < je; ++j) {
              const float4 q = ref[j];
              best_take(b, dist2(sx, sy, sz, q.x, q.y, q.z), __float_as_int(q.w), (int)j, lim);
            }
            if (STATS) {
              n_rows += 1;
              n_cand += (unsigned long long)(je - jb);
            }
          }
        }
      }
      group_min(b);
      if (active) {
        r += 1;
        const float lb2 = (float)(r - 1) * g.cell + m - g.margin;
        if (r > rmax || (lb2 > 0.f && lb2 * lb2 > fminf(b.d, lim))) active = false;
      }
    }
    // ---- gate, outputs, histogram: lane 0 of the group ----
    if (valid && sub == 0) {
      int penc = -1;
      float dout = kInfF;
      if (b.pos >= 0) {
        penc = b.pos;
        dout = b.d;
        if (cp.has_normal_gate) {  // w = (n_read . n_ref < cos(maxAngle)) ? 0 : 1, on the ROTATED reading normal
          const float a = rnx[i], bb = rny[i], c = rnz[i];
          const float nx = rot_row(T, 0, a, bb, c), ny = rot_row(T, 1, a, bb, c), nz = rot_row(T, 2, a, bb, c);
          const float4 rn = refn[b.pos];
          float v = nx * rn.x;
          v = v + ny * rn.y;
          v = v + nz * rn.z;
          if (v < cp.cos_max_angle) penc = -2 - b.pos;
        }
        atomicAdd(&s_hist[(__float_as_uint(dout) >> 20) & (kHistBins - 1)], 1u);
      }
      pos_out[i] = penc;
      d2_out[i] = dout;
    }
  }
  __syncthreads();
  if (!(cp.dbg & 1))
  for (int k = threadIdx.x; k < kHistBins; k += kBlock) {
    const uint32_t v = s_hist[k];
    if (v) atomicAdd(&hist[k], v);
  }
  if (STATS) {
    n_cand = wave_sum_u64(n_cand);
    n_rows = wave_sum_u64(n_rows);
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&st->cand_count, n_cand);
      atomicAdd(&st->row_count, n_rows);
    }
  }
}

// histogram of externally supplied distances (module-level outlier API)
__global__ void __launch_bounds__(kBlock) k_hist(const float* __restrict__ d2, int N, uint32_t* __restrict__ hist) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < N) {
    const float d = d2[i];
    if (d != kInfF) atomicAdd(&hist[(__float_as_uint(d) >> 20) & (kHistBins - 1)], 1u);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Trim limit — Matches::getDistsQuantile (LPM/Matches.cpp:61-87): the EXACT element nth_element would return, by a
// 3-level radix selection on the fp32 bit pattern (non-negative floats order like their bits).
//   level 1 (bits 30..20, 2048 bins)  accumulated by k_match;
//   k_sel_compact (all CUs)           every block finds the bin that holds rank k, then the members of that bin in its
//                                     slice of d2 are appended to a candidate list (wave-aggregated atomics);
//   k_sel_final (one 1024-lane block) levels 2 and 3 (10 bits each) over the candidates, in LDS when they fit.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSelThreads = 1024;
constexpr int kSelCap = 32768;   // candidates resolved in LDS; larger bins are resolved in global memory

struct SelScratch {   // device-resident, between the two select kernels
  uint32_t count;     // candidates appended (zeroed by k_sel_final for the next iteration)
  uint32_t bin;       // level-1 bin holding rank k
  uint32_t kk;        // rank inside that bin
  uint32_t bin_count;
  uint32_t skip;      // 1: nothing to select (no Trimmed filter / error)
};

__global__ void __launch_bounds__(kBlock) k_sel_compact(const float* __restrict__ d2, int N, const uint32_t* __restrict__ hist, ChainParams cp,
                                                        IcpState* __restrict__ st, SelScratch* __restrict__ ss, uint32_t* __restrict__ cand) {
  if (st->done) return;
  __shared__ uint32_t s_sc[32];
  __shared__ uint32_t s_res[4];
  // rank-k bin: every block repeats the same integer arithmetic on the same histogram
  uint32_t c[8];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c[k] = hist[threadIdx.x * 8 + k];
    mine += c[k];
  }
  uint32_t n_fin;
  const uint32_t ex = block_excl_scan(mine, &n_fin, s_sc);
  if (!cp.has_trim || n_fin == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->n_finite = n_fin;
      ss->skip = 1;
      if (!cp.has_trim) {
        st->limit = kInfF;
      } else {  // "No matches available for computing distance quantiles" (Matches.cpp:76-77)
        st->status = 5;
      }
    }
    return;
  }
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
    for (int q = 0; q < 8; ++q) {
      if (c[q] > 0 && acc <= k && k < acc + c[q]) {
        s_res[0] = threadIdx.x * 8 + q;
        s_res[1] = k - acc;
        s_res[2] = c[q];
      }
      acc += c[q];
    }
  }
  __syncthreads();
  const uint32_t bin = s_res[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->n_finite = n_fin;
    ss->bin = bin;
    ss->kk = s_res[1];
    ss->bin_count = s_res[2];
    ss->skip = 0;
  }
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < ((N + 63) & ~63); i += gridDim.x * kBlock) {
    const float d = i < N ? d2[i] : kInfF;
    const uint32_t u = __float_as_uint(d);
    const bool in = (d != kInfF) && ((u >> 20) == bin);
    const unsigned long long mask = __ballot(in);
    if (mask) {
      const int lane = threadIdx.x & 63;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&ss->count, (uint32_t)__popcll(mask));
      base = __shfl(base, 0, 64);
      if (in) cand[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = u;
    }
  }
}

__device__ __forceinline__ void select_level(const uint32_t* vals, int n_vals, uint32_t prefix, int prefix_shift, int shift,
                                             uint32_t* s_bins /*1024*/, uint32_t* s_tmp, uint32_t& kk, uint32_t& digit) {
  // histogram of the 10 bits at `shift` among the values whose bits above prefix_shift equal prefix
  s_bins[threadIdx.x] = 0u;
  __syncthreads();
  for (int i = threadIdx.x; i < n_vals; i += kSelThreads) {
    const uint32_t u = vals[i];
    if ((u >> prefix_shift) == prefix) atomicAdd(&s_bins[(u >> shift) & 1023u], 1u);
  }
  __syncthreads();
  const uint32_t c = s_bins[threadIdx.x];
  uint32_t tot;
  const uint32_t ex = block_excl_scan(c, &tot, s_tmp);
  __syncthreads();
  if (c > 0 && ex <= kk && kk < ex + c) {
    s_tmp[40] = threadIdx.x;
    s_tmp[41] = kk - ex;
  }
  __syncthreads();
  digit = s_tmp[40];
  kk = s_tmp[41];
  __syncthreads();
}

__global__ void __launch_bounds__(kSelThreads) k_sel_final(uint32_t* __restrict__ hist, IcpState* __restrict__ st, SelScratch* __restrict__ ss,
                                                           const uint32_t* __restrict__ cand) {
  if (st->done) return;
  extern __shared__ uint32_t s_dyn[];  // kSelCap values
  __shared__ uint32_t s_bins[1024];
  __shared__ uint32_t s_tmp[64];
  for (int k = threadIdx.x; k < kHistBins; k += kSelThreads) hist[k] = 0u;  // ready for the next k_match
  const uint32_t cnt = ss->count, bin = ss->bin, skip = ss->skip;
  uint32_t kk = ss->kk;
  __syncthreads();
  if (threadIdx.x == 0) ss->count = 0u;
  if (skip) {
    if (threadIdx.x == 0 && st->status != 0) st->done = 1;
    return;
  }
  const uint32_t* vals = cand;
  if (cnt <= (uint32_t)kSelCap) {
    for (uint32_t i = threadIdx.x; i < cnt; i += kSelThreads) s_dyn[i] = cand[i];
    vals = s_dyn;
    __syncthreads();
  }
  uint32_t d1, d0;
  select_level(vals, (int)cnt, bin, 20, 10, s_bins, s_tmp, kk, d1);
  select_level(vals, (int)cnt, (bin << 10) | d1, 10, 0, s_bins, s_tmp, kk, d0);
  if (threadIdx.x == 0) st->limit = __uint_as_float((bin << 20) | (d1 << 10) | d0);
}

// ------------------------------------------------------------------------------------------------------------------
// kept-pair predicate shared by k_centroid / k_normal_eq: product of the chain's binary weights
//   Trimmed: d2 <= limit   MaxDist: d2 <= max^2   SurfaceNormal: encoded in pos   no match: pos == -1
// (LPM/OutlierFilter.cpp:64-103, LPM/ErrorMinimizer.cpp:98-108)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool kept_pair(int pe, float d, float limit, float max_out_r2) {
  return pe >= 0 && d <= limit && d <= max_out_r2;
}

// k_centroid — means of the kept reading / associated reference points (PointToPlane.cpp:263-264), fp64 partials
__global__ void __launch_bounds__(kBlock) k_centroid(const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz, int N,
                                                     const float4* __restrict__ ref, const int32_t* __restrict__ pos, const float* __restrict__ d2,
                                                     ChainParams cp, const IcpState* __restrict__ st, double* __restrict__ part /*[7][grid]*/) {
  if (st->done) return;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T_iter[k];
  const float limit = st->limit;
  double a[kCentComps] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
    const int pe = pos[i];
    const float d = d2[i];
    if (kept_pair(pe, d, limit, cp.max_out_r2)) {
      const float px = rx[i], py = ry[i], pz = rz[i];
      const float4 q = ref[pe];
      a[0] += (double)xf_row(T, 0, px, py, pz);
      a[1] += (double)xf_row(T, 1, px, py, pz);
      a[2] += (double)xf_row(T, 2, px, py, pz);
      a[3] += (double)q.x;
      a[4] += (double)q.y;
      a[5] += (double)q.z;
      a[6] += 1.0;
    }
  }
  __shared__ double sh[4][kCentComps];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < kCentComps; ++c) {
    const double v = wave_sum(a[c]);
    if (l == 0) sh[w][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < kCentComps)
    part[threadIdx.x * gridDim.x + blockIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// every block reduces the centroid partials in the same fixed order -> bit-identical means in all blocks
__device__ __forceinline__ void reduce_centroid(const double* __restrict__ part, int nb, double* out7, double* sh /*[4][7]*/) {
  double a[kCentComps];
#pragma unroll
  for (int c = 0; c < kCentComps; ++c) {
    double s = 0;
    for (int b = threadIdx.x; b < nb; b += kBlock) s += part[c * nb + b];
    a[c] = wave_sum(s);
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0)
    for (int c = 0; c < kCentComps; ++c) sh[w * kCentComps + c] = a[c];
  __syncthreads();
  for (int c = 0; c < kCentComps; ++c)
    out7[c] = (sh[0 * kCentComps + c] + sh[1 * kCentComps + c]) + (sh[2 * kCentComps + c] + sh[3 * kCentComps + c]);
}

// k_normal_eq — formulatePointMatchingConstraints (PointToPlane.cpp:108-156): G = [(p-mp) x n ; n], h = n.((p-mp)-(q-mq)),
// A = G G^T, b = -(G h^T).  Per-pair arithmetic is fp32 in the reference's order; the K-long sums are fp64.
__global__ void __launch_bounds__(kBlock) k_normal_eq(const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz, int N,
                                                      const float4* __restrict__ ref, const float4* __restrict__ refn,
                                                      const int32_t* __restrict__ pos, const float* __restrict__ d2, ChainParams cp,
                                                      IcpState* __restrict__ st, const double* __restrict__ cent_part, int cent_nb,
                                                      double* __restrict__ part /*[27][grid]*/) {
  if (st->done) return;
  __shared__ double sh[4 * kNeComps];
  double c7[kCentComps];
  reduce_centroid(cent_part, cent_nb, c7, sh);
  const double K = c7[6];
  if (K == 0.0) {  // "no point to minimize" (ErrorMinimizer.cpp:75-77); every block takes the same branch
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->kept = 0;
      st->status = 6;
    }
    return;  // k_solve sees status != 0 and raises done
  }
  const float mpx = (float)(c7[0] / K), mpy = (float)(c7[1] / K), mpz = (float)(c7[2] / K);
  const float mqx = (float)(c7[3] / K), mqy = (float)(c7[4] / K), mqz = (float)(c7[5] / K);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->kept = (int64_t)K;
    st->mp[0] = mpx;
    st->mp[1] = mpy;
    st->mp[2] = mpz;
    st->mq[0] = mqx;
    st->mq[1] = mqy;
    st->mq[2] = mqz;
  }
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = st->T_iter[k];
  const float limit = st->limit;
  double acc[kNeComps];
#pragma unroll
  for (int c = 0; c < kNeComps; ++c) acc[c] = 0.0;
  if (!(cp.dbg & 4))
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
    const int pe = pos[i];
    const float d = d2[i];
    if (!kept_pair(pe, d, limit, cp.max_out_r2)) continue;
    const float x0 = rx[i], y0 = ry[i], z0 = rz[i];
    const float4 q = ref[pe];
    const float4 n = refn[pe];
    const float px = xf_row(T, 0, x0, y0, z0) - mpx, py = xf_row(T, 1, x0, y0, z0) - mpy, pz = xf_row(T, 2, x0, y0, z0) - mpz;
    const float qx = q.x - mqx, qy = q.y - mqy, qz = q.z - mqz;
    float gv[6];
    gv[0] = py * n.z - pz * n.y;
    gv[1] = pz * n.x - px * n.z;
    gv[2] = px * n.y - py * n.x;
    gv[3] = n.x;
    gv[4] = n.y;
    gv[5] = n.z;
    const float ex = px - qx, ey = py - qy, ez = pz - qz;
    float h = 0.f;
    h = h + ex * n.x;
    h = h + ey * n.y;
    h = h + ez * n.z;
    int t = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int c = a; c < 6; ++c) acc[t++] += (double)(gv[a] * gv[c]);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += (double)(gv[a] * h);
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kNeComps; ++c) {
    const double v = wave_sum(acc[c]);
    if (l == 0) sh[w * kNeComps + c] = v;
  }
  __syncthreads();
  if (threadIdx.x < kNeComps) {
    const int c = threadIdx.x;
    part[c * gridDim.x + blockIdx.x] = (sh[0 * kNeComps + c] + sh[1 * kNeComps + c]) + (sh[2 * kNeComps + c] + sh[3 * kNeComps + c]);
  }
}

// k_solve — closes the iteration: reduce the partials, solve, build the step, update T_iter, run the checkers.
__global__ void __launch_bounds__(kBlock) k_solve(const double* __restrict__ part, int nb, int N, ChainParams cp, IcpState* __restrict__ st,
                                                  float* __restrict__ trace_T, float* __restrict__ trace_limit, int64_t* __restrict__ trace_kept,
                                                  int trace_cap, int update_pose) {
  if (st->done) return;
  __shared__ double s_sum[kNeComps];
  __shared__ dev::SolveWork s_work;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (st->status == 0) {
    for (int c = w; c < kNeComps; c += 4) {
      double s = 0;
      for (int b = l; b < nb; b += 64) s += part[c * nb + b];
      s = wave_sum(s);
      if (l == 0) s_sum[c] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (st->status != 0) {
    st->done = 1;
    return;
  }
  dev::SolveWork& W = s_work;
  int t = 0;
  for (int a = 0; a < 6; ++a)
    for (int c = a; c < 6; ++c) {
      const float v = (float)s_sum[t++];
      W.S.A[a][c] = v;
      W.S.A[c][a] = v;
    }
  for (int a = 0; a < 6; ++a) W.S.b[a] = -(float)s_sum[21 + a];
  const int branch = (cp.dbg & 8) ? 0 : dev::solve_sys6(W);
  const float* x = W.x;
  const dev::Sys6& S = W.S;
  float dT[16], Tn[16];
  dev::build_step(x, st->mp, st->mq, dT);
  for (int a = 0; a < 6; ++a) {
    for (int c = 0; c < 6; ++c) st->A[c * 6 + a] = S.A[a][c];
    st->b[a] = S.b[a];
    st->x[a] = x[a];
  }
  for (int k = 0; k < 16; ++k) st->dT[k] = dT[k];
  st->solve_branch = branch;
  st->point_used_ratio = (float)st->kept / (float)N;       // ErrorMinimizer.cpp:139
  st->weighted_ratio = (float)st->kept / (float)N;         // binary weights: sum w == |K| (ErrorMinimizer.cpp:140)
  if (!update_pose) {
    st->iter += 1;
    st->done = 1;
    return;
  }
  dev::mul4(dT, st->T_iter, Tn);
  for (int k = 0; k < 16; ++k) st->T_iter[k] = Tn[k];
  const int it = st->iter;
  if (it < trace_cap) {
    for (int k = 0; k < 16; ++k) trace_T[it * 16 + k] = Tn[k];
    trace_limit[it] = cp.has_trim ? st->limit : __builtin_nanf("");
    trace_kept[it] = st->kept;
  }
  bool iterate = true;
  int status = dev::run_checkers(st, cp, Tn, &iterate);
  st->iter = it + 1;
  // the next iteration starts with transformations.apply(stepReading, T_iter) -> checkParameters (TransformationsImpl.cpp:73-74)
  if (status == 0 && iterate && !dev::rigid_ok(Tn)) status = 8;
  if (status != 0) {
    st->status = status;
    st->done = 1;
  } else if (!iterate) {
    st->done = 1;
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
                                                           int64_t M, int32_t* __restrict__ pos, float* __restrict__ d2) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int id = ids[i];
  const float d = dists[i];
  int pe = -1;
  if (id >= 0 && (int64_t)id < M) {
    pe = orig_to_sorted[id];
    if (weights && weights[i] == 0.0f) pe = -2 - pe;
  }
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
