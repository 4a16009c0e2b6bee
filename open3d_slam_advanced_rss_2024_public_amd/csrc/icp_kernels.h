// icp_kernels.h — hand-written gfx950 (CDNA4, wave64) kernels of the scan-to-map ICP iteration chain.
//
// One ICP iteration = 5 launches on one stream (no host round trip; a `done` flag in IcpState turns the remaining
// launches of a pre-recorded chain into no-ops):
//   k_match      transform reading point by T_iter, exact 1-NN in the voxel grid, normal-angle gate, d2 histogram
//   k_select     exact k-th smallest finite d2 (TrimmedDistOutlierFilter limit) by radix selection
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
// One lane per reading point.  Exact 1-NN by ring expansion over the dense grid: cells of one (z,y) row are contiguous
// in the cell-sorted reference, so a row of the (2r+1)^3 neighbourhood is ONE [begin,end) range read from cell_start.
// A row/ring is skipped when its lower-bound distance exceeds min(best, maxDist^2); ties keep the lowest original index.
// Output per point: d2 (squared fp32 distance, +inf = none) and pos = slot in the sorted reference, -1 = none,
// -2 - slot = matched but rejected by the normal gate (the distance still takes part in the trim quantile).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cell_gap(int d, float l, float cell, float margin) {
  float gap = 0.f;
  if (d > 0) gap = (float)d * cell - l;
  else if (d < 0) gap = l + (float)(-d - 1) * cell;
  gap -= margin;
  return gap > 0.f ? gap : 0.f;
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

  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int i = blk * kBlock + threadIdx.x;
  unsigned long long n_cand = 0, n_rows = 0;
  if (i < N) {
    const float px = rx[i], py = ry[i], pz = rz[i];
    const float sx = xf_row(T, 0, px, py, pz), sy = xf_row(T, 1, px, py, pz), sz = xf_row(T, 2, px, py, pz);
    float best = kInfF;
    int bidx = 0x7fffffff, bpos = -1;
    if (cp.mirror) {  // MirrorMatcher (LPM/MatchersImpl.cpp:65-85): id = i, dist = 0
      bpos = orig_to_sorted[perm[i]];
      best = 0.f;
    } else {
      const float lim = g.max_r2;
      const float ux = (sx - g.ox) * g.inv_cell, uy = (sy - g.oy) * g.inv_cell, uz = (sz - g.oz) * g.inv_cell;
      // clamp far-away queries so the int conversion cannot overflow; the ring bounds stay valid lower bounds
      const float big = 1.0e9f;
      const int cx = (int)floorf(fminf(fmaxf(ux, -big), big));
      const int cy = (int)floorf(fminf(fmaxf(uy, -big), big));
      const int cz = (int)floorf(fminf(fmaxf(uz, -big), big));
      const float lx = fminf(fmaxf((sx - g.ox) - (float)cx * g.cell, 0.f), g.cell);
      const float ly = fminf(fmaxf((sy - g.oy) - (float)cy * g.cell, 0.f), g.cell);
      const float lz = fminf(fmaxf((sz - g.oz) - (float)cz * g.cell, 0.f), g.cell);
      const float m = fminf(fminf(fminf(lx, g.cell - lx), fminf(ly, g.cell - ly)), fminf(lz, g.cell - lz));
      // first ring that touches the grid box, last ring that still does
      int r0 = 0;
      r0 = max(r0, max(-cx, cx - (g.nx - 1)));
      r0 = max(r0, max(-cy, cy - (g.ny - 1)));
      r0 = max(r0, max(-cz, cz - (g.nz - 1)));
      int rmax = max(max(cx, g.nx - 1 - cx), max(max(cy, g.ny - 1 - cy), max(cz, g.nz - 1 - cz)));
      for (int r = r0; r <= rmax; ++r) {
        if (r > 0) {
          const float lb = (float)(r - 1) * g.cell + m - g.margin;
          if (lb > 0.f && lb * lb > fminf(best, lim)) break;
        }
        const int z0 = max(cz - r, 0), z1 = min(cz + r, g.nz - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, g.ny - 1);
        for (int z = z0; z <= z1; ++z) {
          const int dz = z - cz;
          const float gz = cell_gap(dz, lz, g.cell, g.margin);
          const bool zface = (dz == r) || (dz == -r);
          for (int y = y0; y <= y1; ++y) {
            const int dy = y - cy;
            const float gy = cell_gap(dy, ly, g.cell, g.margin);
            if (gz * gz + gy * gy > fminf(best, lim)) continue;
            const bool full = zface || (dy == r) || (dy == -r);
            const uint32_t rowbase = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx;
            // full row: one range [cx-r, cx+r]; interior row: the two end cells only
            const int nseg = full ? 1 : 2;
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
              const uint32_t jb = cell_start[rowbase + (uint32_t)xa];
              const uint32_t je = cell_start[rowbase + (uint32_t)xb + 1u];
              if (STATS) n_rows += 1;
              for (uint32_t j = jb; j < je; ++j) {
                const float4 q = ref[j];
                const float d = dist2(sx, sy, sz, q.x, q.y, q.z);
                const int qi = __float_as_int(q.w);
                if (d <= lim && (d < best || (d == best && qi < bidx))) {
                  best = d;
                  bidx = qi;
                  bpos = (int)j;
                }
              }
              if (STATS) n_cand += (unsigned long long)(je - jb);
            }
          }
        }
      }
    }
    int penc = -1;
    float dout = kInfF;
    if (bpos >= 0) {
      penc = bpos;
      dout = best;
      if (cp.has_normal_gate) {  // w = (n_read . n_ref < cos(maxAngle)) ? 0 : 1, on the ROTATED reading normal
        const float a = rnx[i], b = rny[i], c = rnz[i];
        const float nx = rot_row(T, 0, a, b, c), ny = rot_row(T, 1, a, b, c), nz = rot_row(T, 2, a, b, c);
        const float4 rn = refn[bpos];
        float v = nx * rn.x;
        v = v + ny * rn.y;
        v = v + nz * rn.z;
        if (v < cp.cos_max_angle) penc = -2 - bpos;
      }
      atomicAdd(&s_hist[(__float_as_uint(dout) >> 20) & (kHistBins - 1)], 1u);
    }
    pos_out[i] = penc;
    d2_out[i] = dout;
  }
  __syncthreads();
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
// k_select — Matches::getDistsQuantile (LPM/Matches.cpp:61-87): the EXACT element nth_element would return.
// Single 1024-lane workgroup.  Level 1 (top 11 bits) comes from the histogram k_match accumulated; the winning bin's
// members are compacted into LDS (<= 32768 values) and resolved there with two 10-bit radix passes; if the bin is
// larger the two passes run over global memory instead.  Also clears the histogram for the next iteration.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSelThreads = 1024;
constexpr int kSelCap = 32768;

__device__ __forceinline__ void select_level(const uint32_t* vals, int n_vals, bool from_global, const float* __restrict__ d2, int N,
                                             uint32_t prefix, int prefix_shift, int shift, uint32_t* s_bins /*1024*/, uint32_t* s_tmp,
                                             uint32_t& kk, uint32_t& digit) {
  // histogram of 10 bits at `shift` among values whose bits above prefix_shift equal prefix
  s_bins[threadIdx.x] = 0u;
  __syncthreads();
  if (from_global) {
    for (int i = threadIdx.x; i < N; i += kSelThreads) {
      const float d = d2[i];
      const uint32_t u = __float_as_uint(d);
      if (d != kInfF && (u >> prefix_shift) == prefix) atomicAdd(&s_bins[(u >> shift) & 1023u], 1u);
    }
  } else {
    for (int i = threadIdx.x; i < n_vals; i += kSelThreads) {
      const uint32_t u = vals[i];
      if ((u >> prefix_shift) == prefix) atomicAdd(&s_bins[(u >> shift) & 1023u], 1u);
    }
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

__global__ void __launch_bounds__(kSelThreads) k_select(const float* __restrict__ d2, int N, uint32_t* __restrict__ hist, ChainParams cp,
                                                        IcpState* __restrict__ st) {
  if (st->done) return;
  extern __shared__ uint32_t s_dyn[];          // kSelCap values
  __shared__ uint32_t s_h[kHistBins];
  __shared__ uint32_t s_bins[1024];
  __shared__ uint32_t s_tmp[64];
  for (int k = threadIdx.x; k < kHistBins; k += kSelThreads) {
    s_h[k] = hist[k];
    hist[k] = 0u;
  }
  __syncthreads();
  const uint32_t c0 = s_h[2 * threadIdx.x], c1 = s_h[2 * threadIdx.x + 1];
  uint32_t n_fin;
  const uint32_t ex = block_excl_scan(c0 + c1, &n_fin, s_tmp);
  __syncthreads();
  if (!cp.has_trim) {
    if (threadIdx.x == 0) {
      st->limit = kInfF;
      st->n_finite = n_fin;
    }
    return;
  }
  if (n_fin == 0) {  // "No matches available for computing distance quantiles" (Matches.cpp:76-77)
    if (threadIdx.x == 0) {
      st->n_finite = 0;
      st->status = 5;
      st->done = 1;
    }
    return;
  }
  // index: values.size() * quantile evaluated in fp32, truncated (Matches.cpp:85-86); ratio == 1 -> max element
  uint32_t k;
  if (cp.trim_ratio == 1.0f) {
    k = n_fin - 1;
  } else {
    const float fk = (float)n_fin * cp.trim_ratio;
    k = (uint32_t)fk;
    if (k >= n_fin) k = n_fin - 1;
  }
  if (c0 + c1 > 0 && ex <= k && k < ex + c0 + c1) {
    if (k < ex + c0) {
      s_tmp[40] = 2 * threadIdx.x;
      s_tmp[41] = k - ex;
      s_tmp[42] = c0;
    } else {
      s_tmp[40] = 2 * threadIdx.x + 1;
      s_tmp[41] = k - ex - c0;
      s_tmp[42] = c1;
    }
  }
  __syncthreads();
  const uint32_t bin = s_tmp[40];
  uint32_t kk = s_tmp[41];
  const uint32_t bin_count = s_tmp[42];
  __syncthreads();
  const bool in_lds = bin_count <= (uint32_t)kSelCap;
  if (in_lds) {
    if (threadIdx.x == 0) s_tmp[43] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += kSelThreads) {
      const float d = d2[i];
      const uint32_t u = __float_as_uint(d);
      if (d != kInfF && (u >> 20) == bin) {
        const uint32_t slot = atomicAdd(&s_tmp[43], 1u);
        s_dyn[slot] = u;
      }
    }
    __syncthreads();
  }
  uint32_t d1, d0;
  select_level(s_dyn, (int)bin_count, !in_lds, d2, N, bin, 20, 10, s_bins, s_tmp, kk, d1);
  select_level(s_dyn, (int)bin_count, !in_lds, d2, N, (bin << 10) | d1, 10, 0, s_bins, s_tmp, kk, d0);
  if (threadIdx.x == 0) {
    st->limit = __uint_as_float((bin << 20) | (d1 << 10) | d0);
    st->n_finite = n_fin;
  }
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
  dev::Sys6 S;
  int t = 0;
  for (int a = 0; a < 6; ++a)
    for (int c = a; c < 6; ++c) {
      const float v = (float)s_sum[t++];
      S.A[a][c] = v;
      S.A[c][a] = v;
    }
  for (int a = 0; a < 6; ++a) S.b[a] = -(float)s_sum[21 + a];
  float x[6];
  const int branch = dev::solve_sys6(S, x);
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
