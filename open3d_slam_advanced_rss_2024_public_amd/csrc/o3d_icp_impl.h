// o3d_icp_impl.h — Open3D-semantics point-to-plane ICP + information matrix (C ABI: include/o3s_registration.h), gfx950
// only.  Included at the end of cloud_ops.hip (shares the grid index of normals_dev.h).  fp64, no FMA contraction.
#pragma once
#include "../../include/o3s_registration.h"

#include "normals_dev.h"

#include <atomic>
#include <thread>

namespace {
namespace o3s_cloud {

constexpr int kAccComps = 30;  // [0..20] upper triangle of J^T J (or G^T G), [21..26] J^T r, [27] sum r^2, [28] sum d2, [29] count

// wave-wide fp64 sum through DPP lane permutes + readlane (see csrc/icp_kernels.h wave_sum)
template <int CTRL>
__device__ __forceinline__ double dpp_f64c(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64c(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64c<0xB1>(v);
  v += dpp_f64c<0x4E>(v);
  v += dpp_f64c<0x141>(v);
  v += dpp_f64c<0x140>(v);
  return (readlane_f64c(v, 0) + readlane_f64c(v, 16)) + (readlane_f64c(v, 32) + readlane_f64c(v, 48));
}

// PointCloud::Transform (TransformPoints): p = (T [p 1]).head<3>() / w, in place
__global__ void __launch_bounds__(kB) k_o3d_transform(double* __restrict__ p, int64_t N, const double* __restrict__ Tm) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const double x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
  double v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s = Tm[r] * x;
    s = s + Tm[4 + r] * y;
    s = s + Tm[8 + r] * z;
    s = s + Tm[12 + r] * 1.0;
    v[r] = s;
  }
  p[3 * i] = v[0] / v[3];
  p[3 * i + 1] = v[1] / v[3];
  p[3 * i + 2] = v[2] / v[3];
}

// Source points are visited in the order of the TARGET grid's cells (sorted once, under the initial guess): lanes of a
// wave then walk the same few cells, so the cell ranges and target points they read are shared cache lines instead of
// one DRAM miss per lane.  The correspondences themselves never leave the device, so their order is free.
__global__ void __launch_bounds__(kB) k_src_cell_keys(const double* __restrict__ p, int64_t N, NGrid g, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const double big = 1.0e9;
  const double fx = fmin(fmax(floor((p[3 * i] - g.ox) / g.cell), -big), big), fy = fmin(fmax(floor((p[3 * i + 1] - g.oy) / g.cell), -big), big),
               fz = fmin(fmax(floor((p[3 * i + 2] - g.oz) / g.cell), -big), big);
  const uint64_t x = (uint64_t)min(max((long long)fx, 0ll), (long long)g.nx - 1), y = (uint64_t)min(max((long long)fy, 0ll), (long long)g.ny - 1),
                 z = (uint64_t)min(max((long long)fz, 0ll), (long long)g.nz - 1);
  keys[i] = (z * (uint64_t)g.ny + y) * (uint64_t)g.nx + x;
  vals[i] = (uint32_t)i;
}

// GetRegistrationResultAndCorrespondences + the sums of the NEXT ComputeTransformation (mode 0) or of the information
// matrix (mode 1), one lane per source point: exact nearest target point by ring search, kept iff d2 < r2.
// Two launches of the same body.  PHASE 0, the search: writes corr[i] and nothing else — without the 30 running sums it needs
// half the registers (193 -> ~100 VGPRs), so twice as many waves hide its dependent loads.  PHASE 1, the sums: streams the
// correspondences back in and accumulates, with the SAME query-to-lane assignment and the same reduction as the one-kernel
// version had, so the 30 sums are the same bits (the distance is formed again from the same coordinates in the same order).
template <int PHASE>
__global__ void __launch_bounds__(kB) k_o3d_corr(const double* __restrict__ pcd, int64_t Ns, GridIndex gi, const double* __restrict__ tgt,
                                                 const double* __restrict__ tn, double r2, int mode, int32_t* __restrict__ corr,
                                                 double* __restrict__ part /*[kAccComps][gridDim.x]*/) {
  __shared__ double sh[PHASE == 1 ? 4 : 1][kAccComps];
  double acc[PHASE == 1 ? kAccComps : 1];
#pragma unroll
  for (int c = 0; c < (PHASE == 1 ? kAccComps : 1); ++c) acc[c] = 0.0;
  const NGrid g = gi.g;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * kB) {
    const double qx = pcd[3 * i], qy = pcd[3 * i + 1], qz = pcd[3 * i + 2];
    double best = __builtin_huge_val();
    int32_t bj = -1;
    if (PHASE == 0) {
    const double big = 1.0e9;
    const double fx = fmin(fmax(floor((qx - g.ox) / g.cell), -big), big), fy = fmin(fmax(floor((qy - g.oy) / g.cell), -big), big),
                 fz = fmin(fmax(floor((qz - g.oz) / g.cell), -big), big);
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    const double lx = (qx - g.ox) - fx * g.cell, ly = (qy - g.oy) - fy * g.cell, lz = (qz - g.oz) - fz * g.cell;
    double m = fmin(fmin(fmin(lx, g.cell - lx), fmin(ly, g.cell - ly)), fmin(lz, g.cell - lz));
    m = fmin(fmax(m, 0.0), g.cell);
    const double margin = g.cell * 1e-9 + (fabs(qx) + fabs(qy) + fabs(qz)) * 1e-15;
    // rings that can still hold a point closer than the radius; rings entirely outside the grid are skipped by the bounds
    int r0 = 0;
    r0 = max(r0, max(-cx, cx - (g.nx - 1)));
    r0 = max(r0, max(-cy, cy - (g.ny - 1)));
    r0 = max(r0, max(-cz, cz - (g.nz - 1)));
    const long long rmax = max(max(max((long long)cx, (long long)g.nx - 1 - cx), max((long long)cy, (long long)g.ny - 1 - cy)),
                               max((long long)cz, (long long)g.nz - 1 - cz));
    // candidates four at a time: their 16 loads go out together (a loop that fetches one candidate per trip pays a full memory
    // round trip per candidate — a cell holds ~12), the comparisons then run in index order as before
    auto scan_cell = [&](uint32_t jb, uint32_t je) {
      for (uint32_t j0 = jb; j0 < je; j0 += 4) {
        double px[4], py[4], pz[4];
        int32_t pid[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const size_t j = (size_t)min(j0 + (uint32_t)t, je - 1u);
          px[t] = gi.sp[3 * j];
          py[t] = gi.sp[3 * j + 1];
          pz[t] = gi.sp[3 * j + 2];
          pid[t] = (int32_t)gi.vals[j];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double ddx = qx - px[t], ddy = qy - py[t], ddz = qz - pz[t];
          double d = ddx * ddx;
          d = d + ddy * ddy;
          d = d + ddz * ddz;
          const bool take = j0 + (uint32_t)t < je && ((d < best) || (d == best && pid[t] < bj));
          best = take ? d : best;
          bj = take ? pid[t] : bj;
        }
      }
    };
    long long rr = r0;
    if (r0 <= 1) {
      // Rings 0 and 1 = the 3x3x3 block: its 27 cell ranges are fetched as ONE batch of independent loads (the dense
      // begin / end arrays are large and sparse — every access is a DRAM miss, and 27 dependent misses in a row were
      // most of this kernel's time); the own cell is scanned first so that the common case stops before the other 26.
      uint32_t cb[27], ce[27];
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const int z = cz + t / 9 - 1, y = cy + (t / 3) % 3 - 1, x = cx + t % 3 - 1;
        const bool in = z >= 0 && z < g.nz && y >= 0 && y < g.ny && x >= 0 && x < g.nx;
        const size_t c = in ? ((size_t)z * (size_t)g.ny + (size_t)y) * (size_t)g.nx + (size_t)x : 0;
        const uint32_t b0 = gi.cbeg[c], e0 = gi.cend[c];
        cb[t] = in ? b0 : 0u;
        ce[t] = in ? e0 : 0u;
      }
      scan_cell(cb[13], ce[13]);
      const double lb0 = m - margin;  // everything outside the own cell
      if (!(lb0 > 0.0 && (lb0 * lb0 >= r2 || best < lb0 * lb0))) {
        // the 26 neighbours, each against its own exact lower bound (the query's distance to that cell's box): with a match a
        // few centimetres away all but the one or two cells across the nearest wall are skipped — without the test a query
        // within `best` of any wall (two out of three at 12 points per cell) paid for all 26 cells, ~300 candidates
#pragma unroll
        for (int t = 0; t < 27; ++t) {
          if (t == 13) continue;
          const int ddz = t / 9 - 1, ddy = (t / 3) % 3 - 1, ddx = t % 3 - 1;
          const double gx = ddx == 0 ? 0.0 : (ddx < 0 ? lx : g.cell - lx), gy = ddy == 0 ? 0.0 : (ddy < 0 ? ly : g.cell - ly),
                       gz = ddz == 0 ? 0.0 : (ddz < 0 ? lz : g.cell - lz);
          const double cell_lb = (gx * gx + gy * gy + gz * gz) * (1.0 - 1e-9) - margin;
          if (cell_lb > fmin(best, r2)) continue;  // a tie at `best` is not "beyond": it stays in
          scan_cell(cb[t], ce[t]);
        }
        rr = 2;
      } else {
        rr = rmax + 1;  // done
      }
    }
    for (; rr <= rmax; ++rr) {
      const int r = (int)rr;
      {  // everything in rings >= r is at least lb away
        const double lb = (double)(r - 1) * g.cell + m - margin;
        if (lb > 0.0 && (lb * lb >= r2 || best < lb * lb)) break;
      }
      // A query without a neighbour inside the radius walks every ring up to radius / cell, and a walk that asks one cell after
      // the other is a chain of dependent misses (the dense begin / end arrays are large and sparse): one such lane used to set the
      // kernel's duration (1 ms for 0.5 M queries of which a few per cent are unmatched).  Per ROW instead: rows whose own lower
      // bound already exceeds what can still matter are skipped (the cube's corners), a face row's cells are consecutive in
      // memory — their headers are fetched as one batch of independent loads and, the points being stored in cell order, the
      // non-empty ones form ONE contiguous run of candidates; an inner row contributes its two end cells.
      for (int dz = -r; dz <= r; ++dz) {
        const int z = cz + dz;
        if (z < 0 || z >= g.nz) continue;
        const double gz = (double)max(abs(dz) - 1, 0) * g.cell;
        for (int dy = -r; dy <= r; ++dy) {
          const int y = cy + dy;
          if (y < 0 || y >= g.ny) continue;
          const bool face = (dz == r) || (dz == -r) || (dy == r) || (dy == -r);
          const double gy = (double)max(abs(dy) - 1, 0) * g.cell, gx = face ? 0.0 : (double)(r - 1) * g.cell;
          const double row_lb = (gz * gz + gy * gy + gx * gx) * (1.0 - 1e-9) - margin;  // every cell of the row is at least this far (squared)
          if (row_lb > fmin(best, r2)) continue;  // beyond the radius or the best so far (a tie at `best` is not "beyond": it stays in)
          const size_t row0 = ((size_t)z * (size_t)g.ny + (size_t)y) * (size_t)g.nx;
          if (face) {
            const int xa = max(cx - r, 0), xb = min(cx + r, g.nx - 1);
            for (int x0 = xa; x0 <= xb; x0 += 8) {
              uint32_t b8[8], e8[8];
#pragma unroll
              for (int t = 0; t < 8; ++t) {
                const int x = min(x0 + t, xb);
                b8[t] = gi.cbeg[row0 + (size_t)x];
                e8[t] = gi.cend[row0 + (size_t)x];
              }
              uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
              for (int t = 0; t < 8; ++t)
                if (e8[t] > b8[t]) {  // empty cells carry begin = end = 0
                  lo = min(lo, b8[t]);
                  hi = max(hi, e8[t]);
                }
              if (hi > lo) scan_cell(lo, hi);
            }
          } else {
            const int x1 = cx - r, x2 = cx + r;
            const bool in1 = x1 >= 0 && x1 < g.nx, in2 = x2 >= 0 && x2 < g.nx;
            const uint32_t b1 = in1 ? gi.cbeg[row0 + (size_t)(in1 ? x1 : 0)] : 0u, e1 = in1 ? gi.cend[row0 + (size_t)(in1 ? x1 : 0)] : 0u;
            const uint32_t b2 = in2 ? gi.cbeg[row0 + (size_t)(in2 ? x2 : 0)] : 0u, e2 = in2 ? gi.cend[row0 + (size_t)(in2 ? x2 : 0)] : 0u;
            scan_cell(b1, e1);
            scan_cell(b2, e2);
          }
        }
      }
    }
    corr[i] = (bj >= 0 && best < r2) ? bj : -1;
    continue;
    }  // PHASE 0
    bj = corr[i];
    const bool hit = bj >= 0;
    if (hit) {
      const double tx = tgt[3 * (size_t)bj], ty = tgt[3 * (size_t)bj + 1], tz = tgt[3 * (size_t)bj + 2];
      {  // the squared distance, formed as the search formed it
        const double ddx = qx - tx, ddy = qy - ty, ddz = qz - tz;
        double d = ddx * ddx;
        d = d + ddy * ddy;
        d = d + ddz * ddz;
        best = d;
      }
      double J[6], rres = 0.0;
      double rows[3][6];
      if (mode == 0) {  // TransformationEstimationPointToPlane: r = (vs - vt) . nt, J = [vs x nt ; nt]
        const double nx = tn[3 * (size_t)bj], ny = tn[3 * (size_t)bj + 1], nz = tn[3 * (size_t)bj + 2];
        const double ex = qx - tx, ey = qy - ty, ez = qz - tz;
        rres = (ex * nx + ey * ny) + ez * nz;
        J[0] = qy * nz - qz * ny;
        J[1] = qz * nx - qx * nz;
        J[2] = qx * ny - qy * nx;
        J[3] = nx;
        J[4] = ny;
        J[5] = nz;
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) acc[t++] += J[a] * J[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * rres;
        acc[27] += rres * rres;
      } else {  // GetInformationMatrixFromPointClouds: three rows per correspondence, built from the TARGET point
        const double r0v[6] = {0.0, tz, -ty, 1.0, 0.0, 0.0}, r1v[6] = {-tz, 0.0, tx, 0.0, 1.0, 0.0}, r2v[6] = {ty, -tx, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          rows[0][a] = r0v[a];
          rows[1][a] = r1v[a];
          rows[2][a] = r2v[a];
        }
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) {
            double s = acc[t] + rows[0][a] * rows[0][b];
            s = s + rows[1][a] * rows[1][b];
            s = s + rows[2][a] * rows[2][b];
            acc[t++] = s;
          }
      }
      acc[28] += best;
      acc[29] += 1.0;
    }
  }
  if (PHASE == 0) return;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < kAccComps; ++c) {
    const double v = wave_sum_f64(acc[PHASE == 1 ? c : 0]);
    if (l == 0) sh[PHASE == 1 ? w : 0][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAccComps)
    part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (sh[0][threadIdx.x] + sh[PHASE == 1 ? 1 : 0][threadIdx.x]) + (sh[PHASE == 1 ? 2 : 0][threadIdx.x] + sh[PHASE == 1 ? 3 : 0][threadIdx.x]);
}

// one wave per component (grid = kAccComps): lane l adds the partials l, l + 64, ... in that order, eight loads in flight at a
// time, then the wave's fixed tree.  (One block walking all 30 components wave by wave took 65 us per pass: 256 dependent
// round trips; the order of the additions — and so the result — is the same.)
__global__ void __launch_bounds__(64) k_o3d_fold(const double* __restrict__ part, int nb, double* __restrict__ out /*kAccComps*/) {
  const int c = blockIdx.x, l = threadIdx.x;
  const double* p = part + (size_t)c * nb;
  double s = 0;
  int b = l;
  for (; b + 7 * 64 < nb; b += 8 * 64) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[b + k * 64];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; b < nb; b += 64) s += p[b];
  s = wave_sum_f64(s);
  if (l == 0) out[c] = s;
}

// ---- host side of the loop (Eigen pieces restated sequentially in fp64) ------------------------------------------
inline void h_mul4(const double* A, const double* B, double* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = A[0 * 4 + r] * B[c * 4 + 0];
      s = s + A[1 * 4 + r] * B[c * 4 + 1];
      s = s + A[2 * 4 + r] * B[c * 4 + 2];
      s = s + A[3 * 4 + r] * B[c * 4 + 3];
      C[c * 4 + r] = s;
    }
}
inline bool h_is_identity(const double* T) {  // Eigen isIdentity(prec = 1e-12)
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      const double v = T[c * 4 + r];
      if (r == c) {
        if (!(std::fabs(v - 1.0) <= 1e-12 * std::min(std::fabs(v), 1.0))) return false;
      } else if (!(std::fabs(v) <= 1e-12)) {
        return false;
      }
    }
  return true;
}
// Eigen LDLT<Matrix6d>::compute + solve: lower, in place, largest-diagonal pivoting, D^-1 as a pseudo-inverse
inline void h_ldlt_solve6(const double Ain[6][6], const double* b, double* x) {
  double A[6][6];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) A[r][c] = Ain[r][c];
  int tr[6];
  for (int k = 0; k < 6; ++k) {
    int big = k;
    double bv = std::fabs(A[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(A[i][i]) > bv) {
        bv = std::fabs(A[i][i]);
        big = i;
      }
    tr[k] = big;
    if (k != big) {
      for (int c = 0; c < k; ++c) std::swap(A[k][c], A[big][c]);
      for (int r = big + 1; r < 6; ++r) std::swap(A[r][k], A[r][big]);
      std::swap(A[k][k], A[big][big]);
      for (int i = k + 1; i < big; ++i) std::swap(A[i][k], A[big][i]);
    }
    if (k > 0) {
      double temp[6];
      for (int c = 0; c < k; ++c) temp[c] = A[c][c] * A[k][c];
      double s = 0;
      for (int c = 0; c < k; ++c) s += A[k][c] * temp[c];
      A[k][k] -= s;
      for (int r = k + 1; r < 6; ++r) {
        double t = 0;
        for (int c = 0; c < k; ++c) t += A[r][c] * temp[c];
        A[r][k] -= t;
      }
    }
    const double akk = A[k][k];
    if (std::fabs(akk) > 0)
      for (int r = k + 1; r < 6; ++r) A[r][k] /= akk;
  }
  double y[6];
  for (int i = 0; i < 6; ++i) y[i] = b[i];
  for (int k = 0; k < 6; ++k) std::swap(y[k], y[tr[k]]);
  for (int i = 0; i < 6; ++i)
    for (int c = 0; c < i; ++c) y[i] -= A[i][c] * y[c];
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < 6; ++i) y[i] = std::fabs(A[i][i]) > tol ? y[i] / A[i][i] : 0.0;
  for (int i = 5; i >= 0; --i)
    for (int r = i + 1; r < 6; ++r) y[i] -= A[r][i] * y[r];
  for (int k = 5; k >= 0; --k) std::swap(y[k], y[tr[k]]);
  for (int i = 0; i < 6; ++i) x[i] = y[i];
}
// utility::TransformVector6dToMatrix4d: (AngleAxis(z) * AngleAxis(y) * AngleAxis(x)).matrix() through quaternions
inline void h_vec6_to_T(const double* v, double* T) {
  struct Q {
    double w, x, y, z;
  };
  auto mul = [](const Q& a, const Q& b) {
    return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
  };
  const Q qx{std::cos(0.5 * v[0]), std::sin(0.5 * v[0]), 0, 0};
  const Q qy{std::cos(0.5 * v[1]), 0, std::sin(0.5 * v[1]), 0};
  const Q qz{std::cos(0.5 * v[2]), 0, 0, std::sin(0.5 * v[2])};
  const Q q = mul(mul(qz, qy), qx);
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  for (int i = 0; i < 16; ++i) T[i] = 0;
  T[15] = 1;
  T[0 * 4 + 0] = 1 - (tyy + tzz);
  T[1 * 4 + 0] = txy - twz;
  T[2 * 4 + 0] = txz + twy;
  T[0 * 4 + 1] = txy + twz;
  T[1 * 4 + 1] = 1 - (txx + tzz);
  T[2 * 4 + 1] = tyz - twx;
  T[0 * 4 + 2] = txz - twy;
  T[1 * 4 + 2] = tyz + twx;
  T[2 * 4 + 2] = 1 - (txx + tyy);
  T[3 * 4 + 0] = v[3];
  T[3 * 4 + 1] = v[4];
  T[3 * 4 + 2] = v[5];
}

struct O3dIcpWork {
  NormalsWork grid;  // index over the target
  Buf d_src, d_src_in, d_tgt, d_tn, d_corr, d_part, d_sum, d_T;
  const double* tgt = nullptr;  // the target cloud the kernels read: d_tgt / d_tn, or arrays that already live in HBM
  const double* tn = nullptr;
  Arena sort_arena;
  int nb = 0;
};

// src_on_device / tgt_on_device: the pointers are device arrays (a resident submap): the source is copied inside HBM (it
// is transformed in place), the target is read where it lies
inline int o3d_prepare(O3dIcpWork& w, const double* source, int64_t Ns, const double* target, const double* tn, int64_t Nt, double max_dist,
                       GridIndex* gi, hipStream_t s, bool on_device = false) {
  if (Ns > (int64_t)0x7fffffff || Nt > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  CK(w.d_src.alloc((size_t)Ns * 24));
  CK(w.d_corr.alloc((size_t)Ns * 4));
  CK(w.d_T.alloc(128));
  CK(hipMemcpyAsync(w.d_src.p, source, (size_t)Ns * 24, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  if (on_device) {
    w.tgt = target;
    w.tn = tn;
  } else {
    CK(w.d_tgt.alloc((size_t)Nt * 24));
    CK(hipMemcpyAsync(w.d_tgt.p, target, (size_t)Nt * 24, hipMemcpyHostToDevice, s));
    w.tgt = w.d_tgt.as<double>();
    w.tn = nullptr;
    if (tn) {
      CK(w.d_tn.alloc((size_t)Nt * 24));
      CK(hipMemcpyAsync(w.d_tn.p, tn, (size_t)Nt * 24, hipMemcpyHostToDevice, s));
      w.tn = w.d_tn.as<double>();
    }
  }
  w.nb = (int)std::min<int64_t>((Ns + kB - 1) / kB, 2048);
  CK(w.d_part.alloc((size_t)w.nb * kAccComps * 8));
  CK(w.d_sum.alloc(kAccComps * 8));
  // ~3 points per occupied cell was right for the searches that end in the query's own cell; a loop-closure refinement also has
  // a few per cent of queries WITHOUT a neighbour inside max_dist, whose ring walk grows with (max_dist / cell)^3: 12 points per
  // cell (cell ~ 0.35 m on a 0.1 m-voxel map, max_dist 1 m) halves the refinement (6.2 / 15.9 / 8.4 / 11.0 -> 3.7 / 7.0 / 6.1 / 7.9 ms
  // on the closed-loop run's four closures; 8..16 are equal, 32 and 64 slower again).  Any cell size keeps the search exact.
  double rho = 12.0;
  if (const char* e = O3S_HOOK_ENV("O3S_O3D_RHO")) rho = atof(e);
  return build_grid_index(w.grid, w.tgt, Nt, max_dist * 0.5, rho, max_dist, gi, s);
}

inline int o3d_transform(O3dIcpWork& w, int64_t Ns, const double* T, hipStream_t s) {
  CK(hipMemcpyAsync(w.d_T.p, T, 128, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_o3d_transform, dim3(nblk(Ns)), dim3(kB), 0, s, w.d_src.as<double>(), Ns, w.d_T.as<double>());
  CK(hipGetLastError());
  return O3S_OK;
}

// reorders d_src by target-grid cell (d_src_in is the scratch copy)
inline int o3d_sort_source(O3dIcpWork& w, int64_t Ns, const GridIndex& gi, hipStream_t s) {
  const size_t n = (size_t)Ns;
  const size_t tb = sort_temp_bytes(Ns);
  CK(w.sort_arena.reserve(2 * Arena::pad(n * 8) + 2 * Arena::pad(n * 4) + Arena::pad(tb) + 4096));
  uint64_t* keys = w.sort_arena.take<uint64_t>(n);
  uint64_t* keys2 = w.sort_arena.take<uint64_t>(n);
  uint32_t* vals = w.sort_arena.take<uint32_t>(n);
  uint32_t* vals2 = w.sort_arena.take<uint32_t>(n);
  void* tmp = w.sort_arena.take<char>(tb);
  hipLaunchKernelGGL(k_src_cell_keys, dim3(nblk(Ns)), dim3(kB), 0, s, w.d_src.as<double>(), Ns, gi.g, keys, vals);
  size_t tbb = tb;
  CK(sort_pairs(tmp, tbb, keys, keys2, vals, vals2, n, key_bits((uint64_t)gi.g.nx * (uint64_t)gi.g.ny * (uint64_t)gi.g.nz), s));  // cell indices of the target grid
  CK(w.d_src_in.alloc(n * 24));
  CK(hipMemcpyAsync(w.d_src_in.p, w.d_src.p, n * 24, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(k_gather_sorted, dim3(nblk(Ns)), dim3(kB), 0, s, w.d_src_in.as<double>(), vals2, Ns, w.d_src.as<double>());
  CK(hipGetLastError());
  return O3S_OK;
}

inline int o3d_corr_pass(O3dIcpWork& w, int64_t Ns, const GridIndex& gi, double r2, int mode, double* sums /*kAccComps*/, hipStream_t s) {
  hipLaunchKernelGGL(k_o3d_corr<0>, dim3(w.nb), dim3(kB), 0, s, w.d_src.as<double>(), Ns, gi, w.tgt, w.tn, r2, mode,
                     w.d_corr.as<int32_t>(), w.d_part.as<double>());
  hipLaunchKernelGGL(k_o3d_corr<1>, dim3(w.nb), dim3(kB), 0, s, w.d_src.as<double>(), Ns, gi, w.tgt, w.tn, r2, mode,
                     w.d_corr.as<int32_t>(), w.d_part.as<double>());
  hipLaunchKernelGGL(k_o3d_fold, dim3(kAccComps), dim3(64), 0, s, w.d_part.as<double>(), w.nb, w.d_sum.as<double>());
  CK(hipGetLastError());
  CK(hipMemcpyAsync(sums, w.d_sum.p, kAccComps * 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  return O3S_OK;
}

}  // namespace o3s_cloud
}  // namespace

extern "C" {

void o3s_o3d_icp_default_criteria(o3s_o3d_icp_criteria* c) {
  if (!c) return;
  c->relative_fitness = 1e-6;
  c->relative_rmse = 1e-6;
  c->max_iteration = 30;
}

}  // extern "C"

namespace {
using namespace o3s_cloud;

// RegistrationICP for one pair on stream s (the device is already current on the calling thread); w: grow-only work area
int o3d_icp_run(O3dIcpWork& w, const double* source, int64_t Ns, const double* target, const double* target_normals, int64_t Nt, double max_dist,
                const double init[16], const o3s_o3d_icp_criteria* criteria, o3s_o3d_icp_result* result, hipStream_t s, bool on_device = false) {
  if (!source || !target || !init || !result || Ns <= 0 || Nt <= 0 || !(max_dist > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  if (!target_normals) return O3S_ERR_BAD_SHAPE;  // "requires target pointcloud to have normals"
  o3s_o3d_icp_criteria cr;
  o3s_o3d_icp_default_criteria(&cr);
  if (criteria) cr = *criteria;
  int rc = O3S_OK;
  GridIndex gi;
  rc = o3d_prepare(w, source, Ns, target, target_normals, Nt, max_dist, &gi, s, on_device);
  if (rc != O3S_OK) return rc;
  const double r2 = max_dist * max_dist;
  double T[16];
  std::memcpy(T, init, sizeof(T));
  if (!h_is_identity(init)) {
    rc = o3d_transform(w, Ns, init, s);
    if (rc != O3S_OK) return rc;
  }
  rc = o3d_sort_source(w, Ns, gi, s);
  if (rc != O3S_OK) return rc;
  double sums[kAccComps];
  rc = o3d_corr_pass(w, Ns, gi, r2, 0, sums, s);
  if (rc != O3S_OK) return rc;
  auto fitness = [&](const double* v) { return v[29] > 0 ? v[29] / (double)Ns : 0.0; };
  auto rmse = [&](const double* v) { return v[29] > 0 ? std::sqrt(v[28] / v[29]) : 0.0; };
  int it = 0;
  for (int i = 0; i < cr.max_iteration; ++i) {
    double update[16];
    for (int k = 0; k < 16; ++k) update[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (sums[29] > 0) {  // ComputeTransformation: empty correspondence set -> identity
      double JTJ[6][6], nb[6], x[6];
      int t = 0;
      for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
          JTJ[a][b] = sums[t];
          JTJ[b][a] = sums[t];
          ++t;
        }
      for (int a = 0; a < 6; ++a) nb[a] = -sums[21 + a];
      h_ldlt_solve6(JTJ, nb, x);
      h_vec6_to_T(x, update);
    }
    double Tn[16];
    h_mul4(update, T, Tn);
    std::memcpy(T, Tn, sizeof(T));
    rc = o3d_transform(w, Ns, update, s);
    if (rc != O3S_OK) return rc;
    const double f0 = fitness(sums), e0 = rmse(sums);
    rc = o3d_corr_pass(w, Ns, gi, r2, 0, sums, s);
    if (rc != O3S_OK) return rc;
    ++it;
    if (std::fabs(f0 - fitness(sums)) < cr.relative_fitness && std::fabs(e0 - rmse(sums)) < cr.relative_rmse) break;
  }
  std::memcpy(result->transformation, T, sizeof(T));
  result->fitness = fitness(sums);
  result->inlier_rmse = rmse(sums);
  result->correspondences = (int64_t)sums[29];
  result->iterations = it;
  return O3S_OK;
}

// GetInformationMatrixFromPointClouds for one pair on stream s
int o3d_info_run(O3dIcpWork& w, const double* source, int64_t Ns, const double* target, int64_t Nt, double max_dist, const double T[16],
                 double info[36], hipStream_t s, bool on_device = false) {
  if (!source || !target || !T || !info || Ns <= 0 || Nt <= 0 || !(max_dist > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  int rc = O3S_OK;
  GridIndex gi;
  rc = o3d_prepare(w, source, Ns, target, nullptr, Nt, max_dist, &gi, s, on_device);
  if (rc != O3S_OK) return rc;
  if (!h_is_identity(T)) {
    rc = o3d_transform(w, Ns, T, s);
    if (rc != O3S_OK) return rc;
  }
  rc = o3d_sort_source(w, Ns, gi, s);
  if (rc != O3S_OK) return rc;
  double sums[kAccComps];
  rc = o3d_corr_pass(w, Ns, gi, max_dist * max_dist, 1, sums, s);
  if (rc != O3S_OK) return rc;
  int t = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) {
      info[b * 6 + a] = sums[t];
      info[a * 6 + b] = sums[t];
      ++t;
    }
  return O3S_OK;
}

}  // namespace

extern "C" {

int o3s_o3d_registration_icp(int device, const double* source, int64_t Ns, const double* target, const double* target_normals, int64_t Nt,
                             double max_dist, const double init[16], const o3s_o3d_icp_criteria* criteria, o3s_o3d_icp_result* result) {
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  O3dIcpWork w;
  return o3d_icp_run(w, source, Ns, target, target_normals, Nt, max_dist, init, criteria, result, nullptr);
}

int o3s_o3d_information_matrix(int device, const double* source, int64_t Ns, const double* target, int64_t Nt, double max_dist, const double T[16],
                               double info[36]) {
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  O3dIcpWork w;
  return o3d_info_run(w, source, Ns, target, Nt, max_dist, T, info, nullptr);
}

// Candidate pairs are independent (the reference walks them in a serial loop, PlaceRecognition.cpp:70-71, with the
// `omp parallel for` commented out): kO3dBatchLanes host threads take pairs from a shared counter, each with its
// own HIP stream, so the uploads, index builds, kernels and the small per-iteration read-backs of different pairs overlap.
int o3s_o3d_registration_icp_batch(int device, int32_t n_pairs, const o3s_o3d_pair* pairs, double max_dist, const o3s_o3d_icp_criteria* criteria,
                                   o3s_o3d_icp_result* results, double* infos, int32_t* status) {
  if (n_pairs < 0 || (n_pairs > 0 && (!pairs || !results || !status))) return O3S_ERR_BAD_ARGUMENT;
  if (n_pairs == 0) return O3S_OK;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  int kO3dBatchLanes = 2;  // measured: 16 pairs of 200 k vs 400 k points take 42 / 34 / 48 / 83 ms with 1 / 2 / 4 / 8 lanes (pageable H2D contends)
  if (const char* e = O3S_HOOK_ENV("O3S_O3D_LANES")) kO3dBatchLanes = std::max(1, atoi(e));
  const int lanes = std::min<int>(kO3dBatchLanes, n_pairs);
  std::atomic<int32_t> next{0};
  auto worker = [&]() {
    hipStream_t s = nullptr;
    const bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
    O3dIcpWork w, wi;  // grow-only work areas of this lane: no allocation once they have seen the lane's largest pair
    for (;;) {
      const int32_t k = next.fetch_add(1);
      if (k >= n_pairs) break;
      if (!ok) {
        status[k] = O3S_ERR_HIP;
        continue;
      }
      const o3s_o3d_pair& p = pairs[k];
      int r = o3d_icp_run(w, p.source, p.n_source, p.target, p.target_normals, p.n_target, max_dist, p.init, criteria, &results[k], s);
      if (r == O3S_OK && infos)
        r = o3d_info_run(wi, p.source, p.n_source, p.target, p.n_target, max_dist, results[k].transformation, infos + 36 * (size_t)k, s);
      status[k] = r;
    }
    if (s) {
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < lanes; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  for (int32_t k = 0; k < n_pairs; ++k)
    if (status[k] != O3S_OK) return status[k];
  return O3S_OK;
}

}  // extern "C"
