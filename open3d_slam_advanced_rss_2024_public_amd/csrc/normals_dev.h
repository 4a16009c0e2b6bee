// normals_dev.h — Open3D-semantics normal estimation on device arrays (part of the cloud_ops.hip translation unit).
//
// What every call site of the reference runs on a cloud without normals (O3S/src/CloudRegistration.cpp:71-74,
// O3S/src/Submap.cpp:269-271, ROS/src/RosbagRangeDataProcessorRos.cpp:168-171):
//     cloud.EstimateNormals(KDTreeSearchParamHybrid(radius, max_nn));   // fast_normal_computation = true
//     cloud.NormalizeNormals();
//     cloud.OrientNormalsTowardsCameraLocation();                        // camera = (0, 0, 0)
// Open3D v0.15.1 is not part of the reference tree; the arithmetic follows its published source (EstimateNormals.cpp,
// utility/Eigen.cpp FastEigen3x3, KDTreeFlann::SearchHybrid) exactly as oracle/icp_oracle.cpp restates it:
//   neighbours = the max_nn nearest points of the SAME cloud (the query included), ascending (d2, index), cut at
//   d2 < radius^2; covariance from the nine cumulants summed in neighbour order; eigenvector of the smallest eigenvalue
//   by the closed-form symmetric 3x3 solver; fewer than 3 neighbours -> (0, 0, 1).
// The kd-tree is replaced by a uniform grid over the cloud (cell-sorted copy + dense begin/end arrays) searched ring by
// ring with conservative lower bounds, so the neighbour lists are exact.  fp64 throughout, no FMA contraction.
#pragma once
#include "cloud_dev.h"

namespace {
namespace o3s_cloud {

constexpr int kNnMax = 32;  // largest max_nn served (the reference's parameter files use 5 .. 20)

// Bounds of a cloud as order-preserving u64 bit patterns.  Every slot is a MINIMUM — the maxima are kept as the minimum of the
// complemented pattern — so that one byte fill (0xFF) initialises all replicas.
__global__ void __launch_bounds__(kB) k_bounds(const double* __restrict__ pts, int64_t N, unsigned long long* __restrict__ slots /*[kExtSlots][min[3], ~max[3]], ordered bits*/) {
  unsigned long long* mnmx = slots + 6 * (blockIdx.x & (kExtSlots - 1));
  // a thread folds its points first (blocks stride over the cloud): one wave reduction per 8+ points instead of one per point — the
  // reduction's 72 lane permutes were the kernel (35 us at 0.76 M points)
  unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < N; i += (int64_t)gridDim.x * kB)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      unsigned long long u = (unsigned long long)__double_as_longlong(pts[3 * i + a]);
      u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
      lo[a] = u < lo[a] ? u : lo[a];
      hi[a] = u > hi[a] ? u : hi[a];
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const unsigned long long l = wave_min_u64(lo[a]), h = wave_max_u64(hi[a]);
    if ((threadIdx.x & 63) == 0 && l <= h) {
      if (l < __atomic_load_n(&mnmx[a], __ATOMIC_RELAXED)) atomicMin(&mnmx[a], l);
      if (~h < __atomic_load_n(&mnmx[3 + a], __ATOMIC_RELAXED)) atomicMin(&mnmx[3 + a], ~h);
    }
  }
}
// folds the replicas (one per lane) and posts the six bounds (mailbox words 2..13: lo / hi halves; the maxima un-complemented), then
// the sequence number
__global__ void __launch_bounds__(64) k_bounds_post(const unsigned long long* __restrict__ slots, uint32_t* __restrict__ mailbox, uint32_t seq) {
  static_assert(kExtSlots == 64, "one replica per lane");
  if (blockIdx.x != 0) return;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    unsigned long long v = wave_min_u64(slots[threadIdx.x * 6 + a]);
    if (a >= 3) v = ~v;
    if (threadIdx.x == 0) {
      __hip_atomic_store(mailbox + 2 + 2 * a, (uint32_t)(v & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(mailbox + 3 + 2 * a, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
inline double ordered_to_double(unsigned long long u) {
  u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
  double d;
  std::memcpy(&d, &u, 8);
  return d;
}

// begin / end of every occupied cell in the key-sorted order (empty cells keep begin = end = 0)
__global__ void __launch_bounds__(kB) k_cell_ranges(const uint64_t* __restrict__ keys, int64_t N, uint32_t* __restrict__ cbeg, uint32_t* __restrict__ cend) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const uint64_t k = keys[i];
  if (i == 0 || keys[i - 1] != k) cbeg[k] = (uint32_t)i;
  if (i == N - 1 || keys[i + 1] != k) cend[k] = (uint32_t)i + 1u;
}

__global__ void __launch_bounds__(kB) k_gather_sorted(const double* __restrict__ pts, const uint32_t* __restrict__ vals, int64_t N, double* __restrict__ sp) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const int64_t j = vals[i];
  sp[3 * i] = pts[3 * j];
  sp[3 * i + 1] = pts[3 * j + 1];
  sp[3 * i + 2] = pts[3 * j + 2];
}

struct NGrid {
  double ox, oy, oz, cell;
  int32_t nx, ny, nz;
};

struct D3 {
  double x, y, z;
};
__device__ __forceinline__ D3 cross3(const D3& a, const D3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot3(const D3& a, const D3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// utility/Eigen.cpp ComputeEigenvector0 / ComputeEigenvector1 / FastEigen3x3 — same statements as oracle/icp_oracle.cpp
__device__ inline D3 eigenvector0(const double (*A)[3], double eval0) {
  const D3 row0{A[0][0] - eval0, A[0][1], A[0][2]};
  const D3 row1{A[0][1], A[1][1] - eval0, A[1][2]};
  const D3 row2{A[0][2], A[1][2], A[2][2] - eval0};
  const D3 r0xr1 = cross3(row0, row1), r0xr2 = cross3(row0, row2), r1xr2 = cross3(row1, row2);
  const double d0 = dot3(r0xr1, r0xr1), d1 = dot3(r0xr2, r0xr2), d2 = dot3(r1xr2, r1xr2);
  double dmax = d0;
  int imax = 0;
  if (d1 > dmax) {
    dmax = d1;
    imax = 1;
  }
  if (d2 > dmax) imax = 2;
  if (imax == 0) {
    const double s = sqrt(d0);
    return {r0xr1.x / s, r0xr1.y / s, r0xr1.z / s};
  } else if (imax == 1) {
    const double s = sqrt(d1);
    return {r0xr2.x / s, r0xr2.y / s, r0xr2.z / s};
  }
  const double s = sqrt(d2);
  return {r1xr2.x / s, r1xr2.y / s, r1xr2.z / s};
}

__device__ inline D3 eigenvector1(const double (*A)[3], const D3& e0, double eval1) {
  D3 U, V;
  if (fabs(e0.x) > fabs(e0.y)) {
    const double inv = 1.0 / sqrt(e0.x * e0.x + e0.z * e0.z);
    U = {-e0.z * inv, 0.0, e0.x * inv};
  } else {
    const double inv = 1.0 / sqrt(e0.y * e0.y + e0.z * e0.z);
    U = {0.0, e0.z * inv, -e0.y * inv};
  }
  V = cross3(e0, U);
  const D3 AU{(A[0][0] * U.x + A[0][1] * U.y) + A[0][2] * U.z, (A[0][1] * U.x + A[1][1] * U.y) + A[1][2] * U.z,
              (A[0][2] * U.x + A[1][2] * U.y) + A[2][2] * U.z};
  const D3 AV{(A[0][0] * V.x + A[0][1] * V.y) + A[0][2] * V.z, (A[0][1] * V.x + A[1][1] * V.y) + A[1][2] * V.z,
              (A[0][2] * V.x + A[1][2] * V.y) + A[2][2] * V.z};
  double m00 = dot3(U, AU) - eval1, m01 = dot3(U, AV), m11 = dot3(V, AV) - eval1;
  const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
  if (a00 >= a11) {
    const double mx = fmax(a00, a01);
    if (mx > 0) {
      if (a00 >= a01) {
        m01 /= m00;
        m00 = 1 / sqrt(1 + m01 * m01);
        m01 *= m00;
      } else {
        m00 /= m01;
        m01 = 1 / sqrt(1 + m00 * m00);
        m00 *= m01;
      }
      return {m01 * U.x - m00 * V.x, m01 * U.y - m00 * V.y, m01 * U.z - m00 * V.z};
    }
    return U;
  }
  const double mx = fmax(a11, a01);
  if (mx > 0) {
    if (a11 >= a01) {
      m01 /= m11;
      m11 = 1 / sqrt(1 + m01 * m01);
      m01 *= m11;
    } else {
      m11 /= m01;
      m01 = 1 / sqrt(1 + m11 * m11);
      m11 *= m01;
    }
    return {m11 * U.x - m01 * V.x, m11 * U.y - m01 * V.y, m11 * U.z - m01 * V.z};
  }
  return U;
}

__device__ inline D3 fast_eigen3x3(double (*A)[3]) {
  double mc = A[0][0];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) mc = fmax(mc, A[r][c]);
  if (mc == 0) return {0, 0, 0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) A[r][c] /= mc;
  const double norm = (A[0][1] * A[0][1] + A[0][2] * A[0][2]) + A[1][2] * A[1][2];
  if (norm > 0) {
    const double q = ((A[0][0] + A[1][1]) + A[2][2]) / 3;
    const double b00 = A[0][0] - q, b11 = A[1][1] - q, b22 = A[2][2] - q;
    const double p = sqrt((((b00 * b00 + b11 * b11) + b22 * b22) + norm * 2) / 6);
    const double c00 = b11 * b22 - A[1][2] * A[1][2];
    const double c01 = A[0][1] * b22 - A[1][2] * A[0][2];
    const double c02 = A[0][1] * A[1][2] - b11 * A[0][2];
    const double det = ((b00 * c00 - A[0][1] * c01) + A[0][2] * c02) / ((p * p) * p);
    double half_det = det * 0.5;
    half_det = fmin(fmax(half_det, -1.0), 1.0);
    const double angle = acos(half_det) / 3.0;
    const double two_thirds_pi = 2.09439510239319549;
    const double beta2 = cos(angle) * 2;
    const double beta0 = cos(angle + two_thirds_pi) * 2;
    const double beta1 = -(beta0 + beta2);
    const double ev0 = q + p * beta0, ev1 = q + p * beta1, ev2 = q + p * beta2;
    if (half_det >= 0) {
      const D3 e2 = eigenvector0(A, ev2);
      if (ev2 < ev0 && ev2 < ev1) return e2;
      const D3 e1 = eigenvector1(A, e2, ev1);
      if (ev1 < ev0 && ev1 < ev2) return e1;
      return cross3(e1, e2);
    }
    const D3 e0 = eigenvector0(A, ev0);
    if (ev0 < ev1 && ev0 < ev2) return e0;
    const D3 e1 = eigenvector1(A, e0, ev1);
    if (ev1 < ev0 && ev1 < ev2) return e1;
    return cross3(e0, e1);
  }
  if (A[0][0] < A[1][1] && A[0][0] < A[2][2]) return {1, 0, 0};
  if (A[1][1] < A[0][0] && A[1][1] < A[2][2]) return {0, 1, 0};
  return {0, 0, 1};
}

// One lane per point (in cell-sorted order, so neighbouring lanes walk neighbouring cells).  The K nearest candidates
// live in registers as a sorted list; an insertion is an unrolled compare-exchange chain (no dynamic indexing).
template <int K>
__global__ void __launch_bounds__(kB) k_normals(const double* __restrict__ sp /*cell-sorted points*/, const uint32_t* __restrict__ vals /*sorted -> original*/,
                                                const double* __restrict__ pts /*original order*/, int64_t N, NGrid g,
                                                const uint32_t* __restrict__ cbeg, const uint32_t* __restrict__ cend, int max_nn, double r2,
                                                double* __restrict__ out_n, int32_t* __restrict__ out_idx /*nullable, N x max_nn*/) {
  const int64_t t = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (t >= N) return;
  const double qx = sp[3 * t], qy = sp[3 * t + 1], qz = sp[3 * t + 2];
  const int64_t self = vals[t];
  double D[K];
  int32_t J[K];
#pragma unroll
  for (int s = 0; s < K; ++s) {
    D[s] = __builtin_huge_val();
    J[s] = 0x7fffffff;
  }
  // the query's cell: the same expression that built the keys (k_vox_keys_idx, mode 1)
  const int cx = (int)floor((qx - g.ox) / g.cell), cy = (int)floor((qy - g.oy) / g.cell), cz = (int)floor((qz - g.oz) / g.cell);
  const double lx = (qx - g.ox) - (double)cx * g.cell, ly = (qy - g.oy) - (double)cy * g.cell, lz = (qz - g.oz) - (double)cz * g.cell;
  double m = fmin(fmin(fmin(lx, g.cell - lx), fmin(ly, g.cell - ly)), fmin(lz, g.cell - lz));
  m = fmax(m, 0.0);
  const double margin = g.cell * 1e-9;
  const int rmax = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
  // The K nearest under the total order (d2, index) do not depend on the order in which candidates are looked at, and every
  // bound below is conservative, so the walk is free to fetch in batches: a cell-after-cell, candidate-after-candidate walk is a
  // chain of dependent misses (the sparse points of a sweep's far field need several rings: those lanes set the kernel's time).
  auto kth_d2 = [&]() {
    double kth = D[0];
#pragma unroll
    for (int s = 1; s < K; ++s) kth = (s == max_nn - 1) ? D[s] : kth;
    return kth;
  };
  // Candidates kNb at a time: their loads go out together.  A sweep's point has 100-400 candidates (10th neighbour at 0.33 m in
  // the median, 0.74 m at the 99th percentile), and what costs is keeping the K best: the sorted insertion is a chain of K
  // compare-exchange steps, and a wave runs it whenever ANY of its 64 lanes wants a candidate — with 64 different queries that
  // is nearly every candidate (measured: the search is 0.16 / 0.23 / 0.36 / 1.0 ms for K = 8 / 8 / 16 / 32 at knn 4 / 8 / 10 / 20).
  // So a batch is first reduced, per lane, to the candidates that beat the current K-th best (a bit mask; distances and indices
  // parked in LDS), and the chain then runs popcount(mask) times — the wave pays the LARGEST count among its lanes, a few per
  // batch, instead of all kNb — and K is the smallest instantiated size that holds max_nn.
  constexpr int kNb = 16;
  __shared__ double s_cd[kNb][kB];
  __shared__ int32_t s_cj[kNb][kB];
  auto scan_run = [&](uint32_t jb, uint32_t je) {
    for (uint32_t j0 = jb; j0 < je; j0 += kNb) {
      double px[kNb], py[kNb], pz[kNb];
      int32_t pid[kNb];
#pragma unroll
      for (int u = 0; u < kNb; ++u) {
        const size_t j = (size_t)min(j0 + (uint32_t)u, je - 1u);
        px[u] = sp[3 * j];
        py[u] = sp[3 * j + 1];
        pz[u] = sp[3 * j + 2];
        pid[u] = (int32_t)vals[j];
      }
      uint32_t mask = 0u;
#pragma unroll
      for (int u = 0; u < kNb; ++u) {
        const double ddx = qx - px[u], ddy = qy - py[u], ddz = qz - pz[u];
        double d = ddx * ddx;
        d = d + ddy * ddy;
        d = d + ddz * ddz;
        s_cd[u][threadIdx.x] = d;
        s_cj[u][threadIdx.x] = pid[u];
        if (j0 + (uint32_t)u < je && ((d < D[K - 1]) || (d == D[K - 1] && pid[u] < J[K - 1]))) mask |= 1u << u;
      }
      while (mask) {  // own slots of LDS only: no barrier needed
        const int u = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        double cd = s_cd[u][threadIdx.x];
        int32_t cj = s_cj[u][threadIdx.x];
        if ((cd < D[K - 1]) || (cd == D[K - 1] && cj < J[K - 1])) {  // the K-th best may have moved since the mask was formed
#pragma unroll
          for (int s = 0; s < K; ++s) {
            const bool sw = (cd < D[s]) || (cd == D[s] && cj < J[s]);
            const double td = sw ? D[s] : cd;
            const int32_t tj = sw ? J[s] : cj;
            D[s] = sw ? cd : D[s];
            J[s] = sw ? cj : J[s];
            cd = td;
            cj = tj;
          }
        }
      }
    }
  };
  auto done_after = [&](int r) {  // everything outside shells 0..r is at least lb away
    const double lb = (double)r * g.cell + m - margin;
    if (!(lb > 0.0)) return false;
    const double lb2 = lb * lb;
    return kth_d2() < lb2 || lb2 >= r2;  // the max_nn nearest are final, or nothing closer than the radius is left
  };
  bool done = false;
  // distance from the query to the slab of cells at offset dc along one axis (l = the query's offset inside its own cell)
  auto axis_gap = [&](int dc, double l) { return dc == 0 ? 0.0 : (dc < 0 ? l + (double)(-dc - 1) * g.cell : (g.cell - l) + (double)(dc - 1) * g.cell); };
  {  // shells 0 and 1 = the 3 x 3 x 3 block: 27 cell ranges as ONE batch of independent loads.  The three cells of a row are
     // consecutive in memory, so their points form one run.  The runs are visited nearest first — own cell, its two x neighbours,
     // the four rows that share a face with the own row, the four corner rows — and a run whose nearest possible point is
     // already beyond the K-th best is skipped: the K best settle early and most later candidates cost a distance and nothing
     // else.  Runs and their bounds go through LDS (one slot per thread) so that ONE copy of the scan loop serves them all.
    __shared__ uint32_t s_lo[11][kB], s_hi[11][kB];
    __shared__ double s_lb[11][kB];
    {
      uint32_t cb[27], ce[27];
#pragma unroll
      for (int u = 0; u < 27; ++u) {
        const int z = cz + u / 9 - 1, y = cy + (u / 3) % 3 - 1, x = cx + u % 3 - 1;
        const bool in = z >= 0 && z < g.nz && y >= 0 && y < g.ny && x >= 0 && x < g.nx;
        const size_t c = in ? ((size_t)z * (size_t)g.ny + (size_t)y) * (size_t)g.nx + (size_t)x : 0;
        const uint32_t b0 = cbeg[c], e0 = cend[c];
        cb[u] = in ? b0 : 0u;
        ce[u] = in ? e0 : 0u;
      }
      s_lo[0][threadIdx.x] = cb[13];
      s_hi[0][threadIdx.x] = ce[13];
      s_lb[0][threadIdx.x] = 0.0;
      s_lo[1][threadIdx.x] = cb[12];
      s_hi[1][threadIdx.x] = ce[12];
      s_lb[1][threadIdx.x] = lx * lx;
      s_lo[2][threadIdx.x] = cb[14];
      s_hi[2][threadIdx.x] = ce[14];
      s_lb[2][threadIdx.x] = (g.cell - lx) * (g.cell - lx);
      constexpr int kSlotOfRow[9] = {7, 5, 8, 3, -1, 4, 9, 6, 10};  // row = (dz + 1) * 3 + (dy + 1); face neighbours first, corners last
#pragma unroll
      for (int row = 0; row < 9; ++row) {
        if (row == 4) continue;
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int q = row * 3 + u;
          if (ce[q] > cb[q]) {
            lo = min(lo, cb[q]);
            hi = max(hi, ce[q]);
          }
        }
        const double gy = axis_gap(row % 3 - 1, ly), gz = axis_gap(row / 3 - 1, lz);
        s_lo[kSlotOfRow[row]][threadIdx.x] = hi > lo ? lo : 0u;
        s_hi[kSlotOfRow[row]][threadIdx.x] = hi > lo ? hi : 0u;
        s_lb[kSlotOfRow[row]][threadIdx.x] = gy * gy + gz * gz;
      }
    }
    for (int slot = 0; slot < 11; ++slot) {
      const double lb2 = s_lb[slot][threadIdx.x] * (1.0 - 1e-9) - margin;
      if (lb2 > fmin(D[K - 1], r2)) continue;  // D[K - 1] >= the max_nn-th best: conservative; a tie is not "beyond"
      scan_run(s_lo[slot][threadIdx.x], s_hi[slot][threadIdx.x]);
    }
    done = done_after(1) || rmax <= 1;
  }
  for (int r = 2; r <= rmax && !done; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      const int z = cz + dz;
      if (z < 0 || z >= g.nz) continue;
      const double gz = axis_gap(dz, lz);
      for (int dy = -r; dy <= r; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= g.ny) continue;
        const bool face = (dz == r) || (dz == -r) || (dy == r) || (dy == -r);
        const double gy = axis_gap(dy, ly), gx = face ? 0.0 : fmin(axis_gap(-r, lx), axis_gap(r, lx));
        const double row_lb = (gz * gz + gy * gy + gx * gx) * (1.0 - 1e-9) - margin;  // every cell of the row is at least this far (squared)
        if (row_lb > fmin(kth_d2(), r2)) continue;  // beyond the radius or the k-th best so far (a tie is not "beyond")
        const size_t row0 = ((size_t)z * (size_t)g.ny + (size_t)y) * (size_t)g.nx;
        // a face row: its cells in batches of eight headers, each batch one contiguous run; an inner row: its two end cells
        const int xa = face ? max(cx - r, 0) : cx - r, xb = face ? min(cx + r, g.nx - 1) : cx + r;
        for (int x0 = xa; x0 <= xb; x0 += face ? 8 : 2 * r) {
          uint32_t lo = 0xffffffffu, hi = 0u;
          if (face) {
            uint32_t b8[8], e8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int x = min(x0 + u, xb);
              b8[u] = cbeg[row0 + (size_t)x];
              e8[u] = cend[row0 + (size_t)x];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (e8[u] > b8[u]) {  // empty cells carry begin = end = 0
                lo = min(lo, b8[u]);
                hi = max(hi, e8[u]);
              }
          } else if (x0 >= 0 && x0 < g.nx) {
            lo = cbeg[row0 + (size_t)x0];
            hi = cend[row0 + (size_t)x0];
          }
          if (hi > lo && lo != 0xffffffffu) scan_run(lo, hi);
        }
      }
    }
    done = done_after(r);
  }
  // KDTreeFlann::SearchHybrid: the max_nn nearest, cut at d2 < radius^2
  int k = 0;
#pragma unroll
  for (int s = 0; s < K; ++s)
    if (s < max_nn && J[s] != 0x7fffffff && D[s] < r2) k = s + 1;
  if (out_idx) {
#pragma unroll
    for (int s = 0; s < K; ++s)
      if (s < max_nn) out_idx[self * max_nn + s] = s < k ? J[s] : -1;
  }
  double C[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (k >= 3) {  // utility::ComputeCovariance: nine cumulants in neighbour order
    double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < K; ++s) {
      if (s < k) {
        const double* p = pts + 3 * (size_t)J[s];
        const double a = p[0], b = p[1], c = p[2];
        cu[0] += a;
        cu[1] += b;
        cu[2] += c;
        cu[3] += a * a;
        cu[4] += a * b;
        cu[5] += a * c;
        cu[6] += b * b;
        cu[7] += b * c;
        cu[8] += c * c;
      }
    }
    for (int s = 0; s < 9; ++s) cu[s] /= (double)k;
    C[0][0] = cu[3] - cu[0] * cu[0];
    C[1][1] = cu[6] - cu[1] * cu[1];
    C[2][2] = cu[8] - cu[2] * cu[2];
    C[0][1] = C[1][0] = cu[4] - cu[0] * cu[1];
    C[0][2] = C[2][0] = cu[5] - cu[0] * cu[2];
    C[1][2] = C[2][1] = cu[7] - cu[1] * cu[2];
  }
  D3 n = fast_eigen3x3(C);
  if (sqrt(dot3(n, n)) == 0.0) n = {0.0, 0.0, 1.0};
  {  // NormalizeNormals()
    const double z = dot3(n, n);
    if (z > 0) {
      const double s = sqrt(z);
      n = {n.x / s, n.y / s, n.z / s};
    }
  }
  {  // OrientNormalsTowardsCameraLocation(camera = 0)
    const D3 ref{0.0 - qx, 0.0 - qy, 0.0 - qz};
    if (sqrt(dot3(n, n)) == 0.0) {
      n = ref;
      const double l = sqrt(dot3(n, n));
      if (l == 0.0) n = {0.0, 0.0, 1.0};
      else n = {n.x / l, n.y / l, n.z / l};
    } else if (dot3(n, ref) < 0.0) {
      n = {n.x * -1.0, n.y * -1.0, n.z * -1.0};
    }
  }
  out_n[3 * self] = n.x;
  out_n[3 * self + 1] = n.y;
  out_n[3 * self + 2] = n.z;
}

struct NormalsWork {  // grow-only buffers of the estimator (owned by the caller's object)
  Arena arena;
  void* cells = nullptr;  // begin[ncells] | end[ncells]
  size_t cells_cap = 0;
  ~NormalsWork() {
    if (cells) (void)hipFree(cells);
  }
};

struct GridIndex {  // uniform grid over a cloud: cell-sorted copy + dense per-cell ranges (valid until the next build on the same work area)
  NGrid g;
  const uint32_t* cbeg;
  const uint32_t* cend;
  const double* sp;      // points in cell order
  const uint32_t* vals;  // cell order -> original index
};

// cell keys of the grid index in one pass: what k_vox_keys_idx (mode 1) + k_vox_pack make of a point — floor((p - a) / cell) per
// axis, (z ey + y) ex + x — without the index array in between and without the extrema nobody reads here (27 us -> 7 at 0.5 M points)
__global__ void __launch_bounds__(kB) k_grid_keys(const double* __restrict__ pts, int64_t N, double cell, double ax, double ay, double az, uint64_t ex,
                                                  uint64_t ey, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const int32_t vx = (int32_t)floor((pts[3 * i] - ax) / cell), vy = (int32_t)floor((pts[3 * i + 1] - ay) / cell), vz = (int32_t)floor((pts[3 * i + 2] - az) / cell);
  const uint64_t x = (uint64_t)(int64_t)vx, y = (uint64_t)(int64_t)vy, z = (uint64_t)(int64_t)vz;
  keys[i] = (z * ey + y) * ex + x;
  vals[i] = (uint32_t)i;
}

inline size_t grid_index_arena_bytes(int64_t N) {
  const size_t n = (size_t)N;
  return Arena::pad(kExtSlots * 6 * 8) + 2 * Arena::pad(n * 8) + 2 * Arena::pad(n * 4) + Arena::pad(n * 4) + Arena::pad((n + 1) * 4) +
         Arena::pad(n * 24) + Arena::pad(std::max(scan_temp_bytes(N), sort_temp_bytes(N))) + 8192;
}
constexpr size_t kGridMaxCells = (size_t)1 << 24;

// cell0: first guess of the cell edge; the cell is then re-sized once so that an occupied cell holds ~target_rho points
// (surface-like data: density ~ cell^2), never above cell_max.  Any cell size keeps the searches exact.
// known_bb (nullable): the cloud's bounds as the six order-preserving bit patterns k_bounds_post makes (minima, maxima) — a caller that has
// just written the cloud has them for free, and the build then starts without its first pass over the cloud and its first wait
inline int build_grid_index(NormalsWork& w, const double* d_pts, int64_t N, double cell0, double target_rho, double cell_max, GridIndex* out,
                            hipStream_t s, const unsigned long long* known_bb = nullptr) {
  if (N <= 0 || N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  const size_t n = (size_t)N;
  CK(w.arena.reserve(grid_index_arena_bytes(N)));
  Arena& ar = w.arena;
  unsigned long long* d_bb = ar.take<unsigned long long>(kExtSlots * 6);
  uint64_t* keys = ar.take<uint64_t>(n);
  uint64_t* keys2 = ar.take<uint64_t>(n);
  uint32_t* vals = ar.take<uint32_t>(n);
  uint32_t* vals2 = ar.take<uint32_t>(n);
  uint32_t* head = ar.take<uint32_t>(n);
  uint32_t* ord = ar.take<uint32_t>(n + 1);
  double* sp = ar.take<double>(n * 3);
  const size_t tb_scan = scan_temp_bytes(N), tb_sort = sort_temp_bytes(N);
  void* tmp = ar.take<char>(std::max(tb_scan, tb_sort));
  // bounds: replicas initialised by one byte fill, folded on the device and posted into the mailbox the host polls (the copy of
  // the replicas into pageable memory plus a stream synchronisation was 25-40 us of every build)
  if (!known_bb) {
  CK(hipMemsetAsync(d_bb, 0xFF, (size_t)kExtSlots * 6 * 8, s));
  hipLaunchKernelGGL(k_bounds, dim3(std::min(nblk(N), 1024u)), dim3(kB), 0, s, d_pts, N, d_bb);
  }
  unsigned long long bb[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull};
  if (known_bb) {
    for (int a = 0; a < 6; ++a) bb[a] = known_bb[a];
  } else {
    PinnedArea& pa = pinned_area();
    int posted = 0;
    if (mailbox_enabled(pa)) {
      const uint32_t seq = mailbox_next(pa);
      hipLaunchKernelGGL(k_bounds_post, dim3(1), dim3(64), 0, s, (const unsigned long long*)d_bb, pa.mb_dev, seq);
      CK(hipGetLastError());
      posted = mailbox_wait(pa, seq, s);
      if (posted < 0) return O3S_ERR_HIP;
      if (posted == 1)
        for (int a = 0; a < 6; ++a)
          bb[a] = (unsigned long long)__atomic_load_n(pa.mb + 2 + 2 * a, __ATOMIC_RELAXED) |
                  ((unsigned long long)__atomic_load_n(pa.mb + 3 + 2 * a, __ATOMIC_RELAXED) << 32);
    }
    if (posted != 1) {
      unsigned long long bb_all[kExtSlots * 6];
      CK(hipMemcpyAsync(bb_all, d_bb, sizeof(bb_all), hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      for (int k = 0; k < kExtSlots; ++k)
        for (int a = 0; a < 6; ++a) bb[a] = a < 3 ? std::min(bb[a], bb_all[k * 6 + a]) : std::max(bb[a], ~bb_all[k * 6 + a]);
    }
  }
  double lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    lo[a] = ordered_to_double(bb[a]);
    hi[a] = ordered_to_double(bb[3 + a]);
    if (!(std::isfinite(lo[a]) && std::isfinite(hi[a]))) return O3S_ERR_BAD_ARGUMENT;
  }
  const double ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
  double cell = std::max(std::max(cell0, ext / 512.0), 1e-9);
  const double kMaxCells = (double)kGridMaxCells;
  int64_t dims[3];
  auto size_grid = [&]() {
    for (;;) {
      double total = 1;
      for (int a = 0; a < 3; ++a) {
        dims[a] = (int64_t)std::floor((hi[a] - lo[a]) / cell) + 2;  // +1 for the floor, +1 of slack for the division's rounding
        total *= (double)dims[a];
      }
      if (total <= kMaxCells) break;
      cell *= 1.26;
    }
  };
  auto resize_for = [&](int64_t n_occ) {  // true: the cell edge changed
    const double rho = (double)N / (double)std::max<int64_t>(n_occ, 1);
    if (rho > 2.0 * target_rho || rho < 0.5 * target_rho) {
      const double c2 = std::min(std::max(cell * std::sqrt(target_rho / rho), ext / 1024.0), std::max(cell_max, ext / 512.0));
      if (std::fabs(c2 - cell) > 0.05 * cell) {
        cell = std::max(c2, 1e-9);
        return true;
      }
    }
    return false;
  };
  // (An occupancy bitmap filled with atomics instead of the first sort was measured: 80-260 us at 0.15-0.55 M points against the
  // ~130 us of the key + sort + count pass it replaces — bits of one word set from eight L2s.)
  for (int attempt = 0; attempt < 2; ++attempt) {
    size_grid();
    hipLaunchKernelGGL(k_grid_keys, dim3(nblk(N)), dim3(kB), 0, s, d_pts, N, cell, lo[0], lo[1], lo[2], (uint64_t)dims[0], (uint64_t)dims[1], keys, vals);
    size_t tb = tb_sort;
    CK(sort_pairs(tmp, tb, keys, keys2, vals, vals2, n, key_bits((uint64_t)dims[0] * (uint64_t)dims[1] * (uint64_t)dims[2]), s));  // keys are cell indices of the grid just sized
    if (attempt == 0) {
      hipLaunchKernelGGL(k_heads, dim3(nblk(N)), dim3(kB), 0, s, keys2, N, ~0ull, head);
      int64_t n_occ = 0;
      const int rc = scan_flags(head, ord, N, tmp, tb_scan, &n_occ, s);
      if (rc != O3S_OK) return rc;
      if (resize_for(n_occ)) continue;
    }
    break;
  }
  NGrid g{};
  g.ox = lo[0];
  g.oy = lo[1];
  g.oz = lo[2];
  g.cell = cell;
  g.nx = (int32_t)dims[0];
  g.ny = (int32_t)dims[1];
  g.nz = (int32_t)dims[2];
  const size_t ncells = (size_t)dims[0] * (size_t)dims[1] * (size_t)dims[2];
  if (ncells * 8 > w.cells_cap) {
    if (w.cells) (void)hipFree(w.cells);
    w.cells = nullptr;
    w.cells_cap = 0;
    CK(hipMalloc(&w.cells, ncells * 8 + 4096));
    w.cells_cap = ncells * 8 + 4096;
  }
  uint32_t* cbeg = reinterpret_cast<uint32_t*>(w.cells);
  uint32_t* cend = cbeg + ncells;
  CK(hipMemsetAsync(w.cells, 0, ncells * 8, s));
  hipLaunchKernelGGL(k_cell_ranges, dim3(nblk(N)), dim3(kB), 0, s, keys2, N, cbeg, cend);
  hipLaunchKernelGGL(k_gather_sorted, dim3(nblk(N)), dim3(kB), 0, s, d_pts, vals2, N, sp);
  CK(hipGetLastError());
  out->g = g;
  out->cbeg = cbeg;
  out->cend = cend;
  out->sp = sp;
  out->vals = vals2;
  return O3S_OK;
}

// d_pts (3 x N doubles, device) -> d_out_n (3 x N doubles); d_out_idx nullable (N x max_nn int32)
inline int estimate_normals_dev(NormalsWork& w, const double* d_pts, int64_t N, double radius, int max_nn, double* d_out_n, int32_t* d_out_idx,
                                hipStream_t s) {
  if (N == 0) return O3S_OK;
  if (N > (int64_t)0x7fffffff || max_nn < 1 || max_nn > kNnMax || !(radius > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  GridIndex gi;
  // ~0.8 max_nn points per occupied cell (between 4 and 12): the 3 x 3 x 3 block then holds the max_nn nearest for nearly every point
  // and, with the nearest-first order of the runs, most of its candidates are turned away by one comparison.  (Round 2's
  // max_nn / 3 was tuned for a search that paid per candidate kept; ray-cast sweep, knn 10: search 0.22 ms at 3.3 points per cell,
  // 0.13 ms at 4 .. 12, 0.22 ms at 16.)  O3S_NRM_RHO overrides.  Any cell size keeps the lists exact.
  const double rho = O3S_HOOK_ENV("O3S_NRM_RHO") ? atof(O3S_HOOK_ENV("O3S_NRM_RHO")) : std::min(12.0, std::max(4.0, 0.8 * (double)max_nn));
  const int rc = build_grid_index(w, d_pts, N, radius * 0.5, rho, radius, &gi, s);
  if (rc != O3S_OK) return rc;
  const double r2 = radius * radius;
#define O3S_NORMALS_LAUNCH(KK) \
  hipLaunchKernelGGL(k_normals<KK>, dim3(nblk(N)), dim3(kB), 0, s, gi.sp, gi.vals, d_pts, N, gi.g, gi.cbeg, gi.cend, max_nn, r2, d_out_n, d_out_idx)
  if (max_nn <= 6) O3S_NORMALS_LAUNCH(6);
  else if (max_nn <= 8) O3S_NORMALS_LAUNCH(8);
  else if (max_nn <= 10) O3S_NORMALS_LAUNCH(10);
  else if (max_nn <= 12) O3S_NORMALS_LAUNCH(12);
  else if (max_nn <= 16) O3S_NORMALS_LAUNCH(16);
  else if (max_nn <= 24) O3S_NORMALS_LAUNCH(24);
  else O3S_NORMALS_LAUNCH(32);
#undef O3S_NORMALS_LAUNCH
  CK(hipGetLastError());
  return O3S_OK;
}

}  // namespace o3s_cloud
}  // namespace
