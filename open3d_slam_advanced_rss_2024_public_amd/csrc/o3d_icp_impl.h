// o3d_icp_impl.h — Open3D-semantics point-to-plane ICP + information matrix (C ABI: include/o3s_registration.h), gfx950
// only.  Included at the end of cloud_ops.hip (shares the grid index of normals_dev.h).  fp64, no FMA contraction.
#pragma once
#include "../../include/o3s_registration.h"

#include "normals_dev.h"

#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {
namespace o3s_cloud {

constexpr int kO3dFarBlocks = 2048;  // k_o3d_search_far: 8192 waves stride over the work list
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

// PointCloud::Transform (TransformPoints) of one point: p = (T [p 1]).head<3>() / w
struct O3dPose {  // a 4x4 (column-major) as a kernel argument: no 128-byte host-to-device copy in front of the launch
  double m[16];
};
__device__ __forceinline__ void o3d_apply(const double* Tm, double& x, double& y, double& z) {
  double v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s = Tm[r] * x;
    s = s + Tm[4 + r] * y;
    s = s + Tm[8 + r] * z;
    s = s + Tm[12 + r] * 1.0;
    v[r] = s;
  }
  x = v[0] / v[3];
  y = v[1] / v[3];
  z = v[2] / v[3];
}

// Source points are visited in the order of the TARGET grid's cells (sorted once, under the initial guess): lanes of a
// wave then walk the same few cells, so the cell ranges and target points they read are shared cache lines instead of
// one DRAM miss per lane.  The correspondences themselves never leave the device, so their order is free.
// Keys of the points as the transformation Tm places them (apply = 0: as they are — Open3D skips an identity); the source itself
// is left where it lies (a resident submap's array) and is only ever read.
__global__ void __launch_bounds__(kB) k_src_cell_keys(const double* __restrict__ p, int64_t N, O3dPose Tm, int apply, NGrid g,
                                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  double px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
  if (apply) o3d_apply(Tm.m, px, py, pz);
  const double big = 1.0e9;
  const double fx = fmin(fmax(floor((px - g.ox) / g.cell), -big), big), fy = fmin(fmax(floor((py - g.oy) / g.cell), -big), big),
               fz = fmin(fmax(floor((pz - g.oz) / g.cell), -big), big);
  const uint64_t x = (uint64_t)min(max((long long)fx, 0ll), (long long)g.nx - 1), y = (uint64_t)min(max((long long)fy, 0ll), (long long)g.ny - 1),
                 z = (uint64_t)min(max((long long)fz, 0ll), (long long)g.nz - 1);
  keys[i] = (z * (uint64_t)g.ny + y) * (uint64_t)g.nx + x;
  vals[i] = (uint32_t)i;
}
// out[i] = Tm . p[order[i]]: the working copy of the source, placed and in search order, in one pass
__global__ void __launch_bounds__(kB) k_o3d_place(const double* __restrict__ p, const uint32_t* __restrict__ order, int64_t N,
                                                  O3dPose Tm, int apply, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const size_t j = order[i];
  double px = p[3 * j], py = p[3 * j + 1], pz = p[3 * j + 2];
  if (apply) o3d_apply(Tm.m, px, py, pz);
  out[3 * i] = px;
  out[3 * i + 1] = py;
  out[3 * i + 2] = pz;
}

// ---- the correspondence search ----------------------------------------------------------------------------------------
// One 32-byte record per target point in cell order: two 16-byte loads fetch a candidate and its original index (the cell-sorted
// copy of the index build keeps coordinates and indices in two arrays: four loads per candidate).
struct __attribute__((aligned(32))) O3dRec {
  double x, y, z;
  long long id;
};
__global__ void __launch_bounds__(kB) k_o3d_records(const double* __restrict__ sp, const uint32_t* __restrict__ vals, int64_t N, O3dRec* __restrict__ rec) {
  const int64_t j = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (j >= N) return;
  O3dRec r;
  r.x = sp[3 * j];
  r.y = sp[3 * j + 1];
  r.z = sp[3 * j + 2];
  r.id = (long long)vals[j];
  rec[j] = r;
}

// ---- nearest neighbour with a certificate ----------------------------------------------------------------------------
// What a search knows about a point when it ends: the nearest target point (d, j) and `others`, a lower bound of the squared
// distance of EVERY OTHER target point — the smallest of: every candidate examined that is not the nearest, the lower bound of every
// cell turned away unopened, the lower bound of the shells never entered.
struct O3dBest {
  double d, others;
  int32_t j;
};
// the smaller of two (d2, index) pairs, lexicographic: the nearest point, the lower index on a tie; the loser is one of the "others"
__device__ __forceinline__ void o3d_take(O3dBest& b, double d, int32_t id, bool ok) {
  ok = ok && id != b.j;  // the neighbour of the last pass comes by again in its cell
  const bool t = ok && ((d < b.d) || (d == b.d && id < b.j));
  b.others = ok ? fmin(b.others, t ? b.d : d) : b.others;  // (NaN distances are never taken and never counted)
  b.d = t ? d : b.d;
  b.j = t ? id : b.j;
}
__device__ __forceinline__ void o3d_exclude(O3dBest& b, double lb2) { b.others = fmin(b.others, lb2); }  // a region not looked at
template <int G>
__device__ __forceinline__ void o3d_group_min(O3dBest& b) {
#pragma unroll
  for (int o = 1; o < G; o <<= 1) {
    const double ob = __shfl_xor(b.d, o), oo = __shfl_xor(b.others, o);
    const int32_t oj = __shfl_xor(b.j, o);
    b.others = fmin(b.others, oo);
    o3d_take(b, ob, oj, oj >= 0);
  }
}

// GetRegistrationResultAndCorrespondences, the search: corr[i] = the exact nearest target point of source point i (lower index on a
// tie) if closer than the radius, else -1.  Per pass:
//  k_o3d_keep (every pass but the first), one lane per point: the point has moved by delta since its last search, whose certificate
//   says every target point but its neighbour is at least L away: if L - delta still exceeds the neighbour's new distance, the
//   neighbour is still THE nearest (strictly: no tie can arise) and nothing is searched; likewise a point without a neighbour stays
//   without one while L - delta exceeds the radius.  After the second update of an ICP that settles nearly every point.  The others
//   go onto the search list.  The kernel also applies the update to the point (PointCloud::Transform), which it reads anyway.
//  k_o3d_search<G>, G lanes per listed point (a wave holds 64 / G points; the points arrive in the order of the target grid's
//   cells, so the lanes of a wave read the same few cells): the own cell and the shell of 26 cells around it.
//   * The correspondence of the PREVIOUS pass (`use_inc`) is a candidate like any other and is looked at first: its distance bounds
//     the search.
//   * Own cell: its candidates are dealt to the G lanes.  Shell 1: see o3d_shell1.
//   * A point whose search is not settled by then — no neighbour yet, or one further away than the next shell — goes onto the far list.
//  k_o3d_search_far: one WAVE per point of the far list, all remaining shells in one walk (o3d_rows_wave).  These are the points
//   without a neighbour inside the radius (1-8 % of a loop-closure refinement's source; they lie together beyond the edge of the
//   overlap, so they fill whole waves): with G lanes each, one such wave walked a chain of ~50 dependent round trips while the rest
//   of the GPU had finished — the launch lasted as long as that wave.  As a list they spread over all CUs, a cube's cells over 64 lanes.
// Round 3's search (one lane per point, the cells scanned one after the other) took 299 us per pass at 0.45 M points; any order of
// looking gives the same nearest neighbour.
struct O3dQuery {
  double qx, qy, qz, lx, ly, lz, m, margin;
  int cx, cy, cz, r0, rmax;
};
__device__ __forceinline__ O3dQuery o3d_query(const double* __restrict__ pcd, int64_t i, const NGrid& g) {
  O3dQuery q;
  q.qx = pcd[3 * i];
  q.qy = pcd[3 * i + 1];
  q.qz = pcd[3 * i + 2];
  // the cell the walk is centred on: any cell near the point will do (the bounds below are formed from the point's offsets lx, ly, lz
  // to THAT cell's corner, whatever they are), so one reciprocal serves the three axes
  const double big = 1.0e9, inv = 1.0 / g.cell;
  const double fx = fmin(fmax(floor((q.qx - g.ox) * inv), -big), big), fy = fmin(fmax(floor((q.qy - g.oy) * inv), -big), big),
               fz = fmin(fmax(floor((q.qz - g.oz) * inv), -big), big);
  q.cx = (int)fx;
  q.cy = (int)fy;
  q.cz = (int)fz;
  q.lx = (q.qx - g.ox) - fx * g.cell;
  q.ly = (q.qy - g.oy) - fy * g.cell;
  q.lz = (q.qz - g.oz) - fz * g.cell;
  const double m = fmin(fmin(fmin(q.lx, g.cell - q.lx), fmin(q.ly, g.cell - q.ly)), fmin(q.lz, g.cell - q.lz));
  q.m = fmin(fmax(m, 0.0), g.cell);
  q.margin = g.cell * 1e-9 + (fabs(q.qx) + fabs(q.qy) + fabs(q.qz)) * 1e-15;
  // shells that can hold cells of the grid at all
  int r0 = 0;
  r0 = max(r0, max(-q.cx, q.cx - (g.nx - 1)));
  r0 = max(r0, max(-q.cy, q.cy - (g.ny - 1)));
  r0 = max(r0, max(-q.cz, q.cz - (g.nz - 1)));
  q.r0 = r0;
  q.rmax = max(max(max(q.cx, g.nx - 1 - q.cx), max(q.cy, g.ny - 1 - q.cy)), max(q.cz, g.nz - 1 - q.cz));  // |c| <= 1e9, n <= 2^24: no overflow
  return q;
}
__device__ __forceinline__ bool o3d_finite(const O3dQuery& q) {
  return (fabs(q.qx) + fabs(q.qy)) + fabs(q.qz) < __builtin_huge_val();  // false for NaN and for +-inf in any coordinate
}
__device__ __forceinline__ double o3d_dist2(double qx, double qy, double qz, double px, double py, double pz) {
  const double ddx = qx - px, ddy = qy - py, ddz = qz - pz;
  double d = ddx * ddx;
  d = d + ddy * ddy;
  d = d + ddz * ddz;
  return d;
}
__device__ __forceinline__ void o3d_cand(const O3dQuery& q, const O3dRec* __restrict__ p, bool ok, O3dBest& b) {
  const double2 a = reinterpret_cast<const double2*>(p)[0];
  const double2 c = reinterpret_cast<const double2*>(p)[1];
  o3d_take(b, o3d_dist2(q.qx, q.qy, q.qz, a.x, a.y, c.x), (int32_t)__double_as_longlong(c.y), ok);
}
// Everything in shells >= rr is at least sqrt(o3d_shell_lb2) away (0: no bound).
__device__ __forceinline__ double o3d_shell_lb2(const O3dQuery& q, const NGrid& g, int rr) {
  const double lb = (double)(rr - 1) * g.cell + q.m - q.margin;
  return lb > 0.0 ? lb * lb : 0.0;
}
// How far a search looks.  Exactness needs every cell that can hold a point inside the radius that beats (or ties) the best so far;
// the search opens a little more — up to `pad` beyond the best so far, up to sqrt(r2o) > radius without one — so that what it does
// NOT open lies at least that much further out and the certificate it leaves (O3dBest::others) survives the next updates of the pose
// (k_o3d_keep).  Without the pad the turned-away cells sit right behind the bound, most of them empty, and every point is searched
// again after every update.
struct O3dReach {
  double r2o, pad;
  int r_cap;  // no shell beyond ceil(sqrt(r2o) / cell) + 1 can hold a point within the reach: a second limit of every walk over shells,
              // beside the bounds formed from the query (INT_MAX for an unbounded radius)
};
__device__ __forceinline__ double o3d_bound(const O3dBest& b, const O3dReach& rc) {
  const double e = sqrt(b.d) + rc.pad;  // inf stays inf
  return fmin(rc.r2o, e * e);
}
// can shell rr hold a point within `bound`?
__device__ __forceinline__ bool o3d_shell_open(const O3dQuery& q, const NGrid& g, int rr, double bound) {
  const double lb2 = o3d_shell_lb2(q, g, rr);
  return !(rr > q.rmax || (lb2 > 0.0 && lb2 > bound));
}
// one batch of up to Q cells of a lane: their 2 Q header words in one round trip, then their points as ONE flat list, kCand per round trip
template <int Q, int kCand>
__device__ __forceinline__ void o3d_batch(const O3dQuery& q, const GridIndex& gi, const O3dRec* __restrict__ rec, const uint32_t (&c)[Q],
                                          const bool (&want)[Q], O3dBest& b) {
  uint32_t P[Q], D[Q], total = 0;
  {
    uint32_t hb[Q], he[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      hb[k] = gi.cbeg[c[k]];
      he[k] = gi.cend[c[k]];
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      P[k] = total;
      D[k] = hb[k] - total;
      total += want[k] ? he[k] - hb[k] : 0u;
    }
  }
  for (uint32_t f0 = 0; f0 < total; f0 += (uint32_t)kCand) {
    const O3dRec* pp[kCand];
    bool ok[kCand];
#pragma unroll
    for (int t = 0; t < kCand; ++t) {
      const uint32_t f = f0 + (uint32_t)t;
      uint32_t dsel = D[0];
#pragma unroll
      for (int k = 1; k < Q; ++k) dsel = (P[k] <= f) ? D[k] : dsel;
      ok[t] = f < total;
      pp[t] = rec + (ok[t] ? f + dsel : 0u);
    }
    double2 a[kCand], cc[kCand];
#pragma unroll
    for (int t = 0; t < kCand; ++t) {
      a[t] = reinterpret_cast<const double2*>(pp[t])[0];
      cc[t] = reinterpret_cast<const double2*>(pp[t])[1];
    }
#pragma unroll
    for (int t = 0; t < kCand; ++t)
      o3d_take(b, o3d_dist2(q.qx, q.qy, q.qz, a[t].x, a[t].y, cc[t].x), (int32_t)__double_as_longlong(cc[t].y), ok[t]);
  }
}

// Shells r_lo .. r with the whole wave on ONE query (k_o3d_search_far), as one walk over the cube (2 r + 1)^3 without its core.  (A
// point without a neighbour has nothing to gain from looking shell by shell — every cell within the radius has to be opened — and
// one walk makes half as many, fuller round trips as three.)  The squared gap of a cell is the sum of three per-axis terms that only
// depend on the offset along that axis: lane l works out the three terms of offset l - r once and a row fetches its two from the
// lanes.  Needs 2 r + 1 <= 64.  By ROWS: lane l takes the (y, z) row l of
// the cube's (2 r + 1)^2; a row whose own bound (the two gap terms it shares) is beyond the reach is turned away as a whole — the
// cube's corners, a third of it; in an open row the cells within reach form ONE interval of x (the gap along x is V-shaped), found
// by a wave-uniform walk over x whose gap term is a scalar; the interval's headers are fetched together and, the points being
// stored in cell order, its occupied cells are ONE run of candidates.  The runs of the 64 rows form a flat list dealt evenly to the
// lanes.  (A cell-by-cell walk — 702 cells per point, each with two multiplies-high and three fp64 shuffles — was 30 us per heavy pass
// slower; profiles/LAB_NOTES_r04.md 6.)  The interval may span the core that was looked at
// before (rows through the centre): its candidates are seen twice, which changes nothing.
// W: lanes on one query (64: the whole wave; 32, 16: two, four queries per wave — as many times the queries in flight per wave slot,
// each with fewer rows per trip).  `lane` is the lane within the query's group; shuffles stay inside the group.
template <int kCand, int W>
__device__ __forceinline__ void o3d_rows_wave(const O3dQuery& q, const GridIndex& gi, const O3dRec* __restrict__ rec, int r_lo, int r, int lane,
                                              const O3dReach& rc, O3dBest& b) {
  constexpr int kW = 9;  // headers fetched per round trip, lane and row (the interval of a row of four shells)
  constexpr int kR = 1;  // rows per lane and trip (2 — the 81 rows of four shells in ONE trip — measured no faster: 189 / 228 us against 159 / 225)
  const NGrid& g = gi.g;
  const int side = 2 * r + 1, n_rows = side * side;
  const uint32_t M = (uint32_t)(0x100000000ull / (unsigned)side) + 1u;  // t / side = umulhi(t, M) for t < 2^26
  double tx, ty, tz;
  {
    const int d = lane - r;
    const double base = (double)(abs(d) - 1) * g.cell;
    const double ax = d == 0 ? 0.0 : base + (d < 0 ? q.lx : g.cell - q.lx), ay = d == 0 ? 0.0 : base + (d < 0 ? q.ly : g.cell - q.ly),
                 az = d == 0 ? 0.0 : base + (d < 0 ? q.lz : g.cell - q.lz);
    tx = ax * ax;
    ty = ay * ay;
    tz = az * az;
  }
  for (int t0 = 0; t0 < n_rows; t0 += W * kR) {  // uniform within the group
    const double bound = o3d_bound(b, rc);
    double gyz[kR];
    bool row_open[kR], row_core[kR];
    uint32_t rowbase[kR];
    int ia[kR], ib[kR];
#pragma unroll
    for (int j = 0; j < kR; ++j) {
      const uint32_t t = (uint32_t)(t0 + j * W + lane);
      const bool in_sq = t < (uint32_t)n_rows;
      const uint32_t iz = in_sq ? __umulhi(t, M) : 0u, iy = in_sq ? t - iz * (uint32_t)side : 0u;
      gyz[j] = __shfl(ty, (int)iy, W) + __shfl(tz, (int)iz, W);
      const int dy = (int)iy - r, dz = (int)iz - r;
      const int y = q.cy + dy, z = q.cz + dz;
      const bool row = in_sq & ((unsigned)y < (unsigned)g.ny) & ((unsigned)z < (unsigned)g.nz);
      const double row_lb = gyz[j] * (1.0 - 1e-9) - q.margin;
      row_open[j] = row & !(row_lb > bound);
      if (row & !row_open[j]) o3d_exclude(b, row_lb);
      row_core[j] = max(abs(dy), abs(dz)) < r_lo;  // rows through the core: the cells near the centre were looked at before
      rowbase[j] = row_open[j] ? ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx + (uint32_t)(q.cx - r) : 0u;
      ia[j] = side;
      ib[j] = -1;
    }
    for (int ix = 0; ix < side; ++ix) {  // uniform within the group; the gap term along x is the same for every lane of it
      const double gx = W == 64 ? readlane_f64c(tx, ix) : __shfl(tx, ix, W);
      const int dx = ix - r, x = q.cx + dx;
      const bool xin = (unsigned)x < (unsigned)g.nx, xcore = abs(dx) < r_lo;
#pragma unroll
      for (int j = 0; j < kR; ++j) {
        const bool cell = row_open[j] & xin & !(row_core[j] & xcore);
        const double cell_lb = (gx + gyz[j]) * (1.0 - 1e-9) - q.margin;
        const bool w = cell & !(cell_lb > bound);  // a tie at the bound is not "beyond": it stays in
        if (cell & !w) o3d_exclude(b, cell_lb);
        ia[j] = w ? min(ia[j], ix) : ia[j];
        ib[j] = w ? ix : ib[j];
      }
    }
    // headers of the intervals, kW cells per row and round trip: the run of the occupied ones
    uint32_t run_b[kR], run_e[kR];
#pragma unroll
    for (int j = 0; j < kR; ++j) {
      run_b[j] = 0xffffffffu;
      run_e[j] = 0u;
    }
    bool more = false;
#pragma unroll
    for (int j = 0; j < kR; ++j) more = more | (ia[j] <= ib[j]);
    for (int i0 = 0; __any(more); i0 += kW) {
      uint32_t hb[kR][kW], he[kR][kW];
#pragma unroll
      for (int j = 0; j < kR; ++j)
#pragma unroll
        for (int k = 0; k < kW; ++k) {
          const int ix = ia[j] + i0 + k;
          const uint32_t c = ix <= ib[j] ? rowbase[j] + (uint32_t)ix : 0u;
          hb[j][k] = gi.cbeg[c];
          he[j][k] = gi.cend[c];
        }
      more = false;
#pragma unroll
      for (int j = 0; j < kR; ++j) {
#pragma unroll
        for (int k = 0; k < kW; ++k) {
          const bool occ = (ia[j] + i0 + k <= ib[j]) && he[j][k] > hb[j][k];  // empty cells carry begin = end = 0
          run_b[j] = occ ? min(run_b[j], hb[j][k]) : run_b[j];
          run_e[j] = occ ? max(run_e[j], he[j][k]) : run_e[j];
        }
        more = more | (ia[j] + i0 + kW <= ib[j]);
      }
    }
    // the runs of the rows as one flat list, candidate f to lane f mod 64; a lane's runs are adjacent in the list
    uint32_t len[kR], mine = 0;
#pragma unroll
    for (int j = 0; j < kR; ++j) {
      len[j] = run_e[j] > run_b[j] ? run_e[j] - run_b[j] : 0u;
      mine += len[j];
    }
    uint32_t S = mine;
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
      const uint32_t up = __shfl_up(S, o, W);
      S += lane >= o ? up : 0u;
    }
    const uint32_t total = __shfl(S, W - 1, W);
    S -= mine;
    for (uint32_t f0 = 0; f0 < total; f0 += (uint32_t)(W * kCand)) {  // uniform within the group
      const O3dRec* pp[kCand];
      bool ok[kCand];
#pragma unroll
      for (int k = 0; k < kCand; ++k) {
        const uint32_t f_raw = f0 + (uint32_t)(k * W + lane);
        ok[k] = f_raw < total;
        const uint32_t f = ok[k] ? f_raw : total - 1u;
        int own = 0;  // the last lane whose offset is <= f
#pragma unroll
        for (int step = W / 2; step > 0; step >>= 1) {
          const int cand = own + step;
          const uint32_t sc = __shfl(S, cand & (W - 1), W);
          own = (cand < W && sc <= f) ? cand : own;
        }
        uint32_t loc = f - __shfl(S, own, W), start = __shfl(run_b[0], own, W);
#pragma unroll
        for (int j = 1; j < kR; ++j) {  // which of the owner's runs
          const uint32_t lprev = __shfl(len[j - 1], own, W), bj = __shfl(run_b[j], own, W);
          const bool next = loc >= lprev;
          loc = next ? loc - lprev : loc;
          start = next ? bj : start;
        }
        pp[k] = rec + (start + loc);
      }
      double2 a[kCand], cc[kCand];
#pragma unroll
      for (int k = 0; k < kCand; ++k) {
        a[k] = reinterpret_cast<const double2*>(pp[k])[0];
        cc[k] = reinterpret_cast<const double2*>(pp[k])[1];
      }
#pragma unroll
      for (int k = 0; k < kCand; ++k)
        o3d_take(b, o3d_dist2(q.qx, q.qy, q.qz, a[k].x, a[k].y, cc[k].x), (int32_t)__double_as_longlong(cc[k].y), ok[k]);
    }
  }
}

// shell r of the query the plain way, its cells dealt to G lanes (lane `sub` takes the cube indices sub, sub + G, ... of the cube
// (2 r + 1)^3, (dx, dy, dz) kept as counters): only for shells wider than a wave (r > 31)
template <int G, int Q, int kCand>
__device__ __forceinline__ void o3d_shell(const O3dQuery& q, const GridIndex& gi, const O3dRec* __restrict__ rec, int r, int sub, const O3dReach& rc,
                                          O3dBest& b) {
  const NGrid& g = gi.g;
  const double bound = o3d_bound(b, rc);
  const int side = 2 * r + 1;
  int dx = -r + sub, dy = -r, dz = -r;
  auto wrap = [&]() {
    while (dx > r) {
      dx -= side;
      if (++dy > r) {
        dy = -r;
        ++dz;
      }
    }
  };
  wrap();
  while (dz <= r) {
    uint32_t c[Q];
    bool want[Q], any = false;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      bool w = dz <= r;
      const bool shell = (dx == r) | (dx == -r) | (dy == r) | (dy == -r) | (dz == r) | (dz == -r);
      const int x = q.cx + dx, y = q.cy + dy, z = q.cz + dz;
      w = w & shell & ((unsigned)x < (unsigned)g.nx) & ((unsigned)y < (unsigned)g.ny) & ((unsigned)z < (unsigned)g.nz);
      if (w) {
        const double gx = dx == 0 ? 0.0 : (double)(abs(dx) - 1) * g.cell + (dx < 0 ? q.lx : g.cell - q.lx);
        const double gy = dy == 0 ? 0.0 : (double)(abs(dy) - 1) * g.cell + (dy < 0 ? q.ly : g.cell - q.ly);
        const double gz = dz == 0 ? 0.0 : (double)(abs(dz) - 1) * g.cell + (dz < 0 ? q.lz : g.cell - q.lz);
        const double cell_lb = (gx * gx + gy * gy + gz * gz) * (1.0 - 1e-9) - q.margin;
        w = !(cell_lb > bound);  // a tie at the bound is not "beyond": it stays in
        if (!w) o3d_exclude(b, cell_lb);
      }
      want[k] = w;
      any = any | w;
      c[k] = w ? ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx + (uint32_t)x : 0u;
      dx += G;
      wrap();
    }
    if (!any) continue;  // every cell of the batch fails its bound
    o3d_batch<Q, kCand>(q, gi, rec, c, want, b);
  }
}

// Shell 1 in k_o3d_search.  The best so far is nearly always much smaller than a cell (the own cell has been scanned, and after the
// first pass the last neighbour bounds the search), so only the cells across the nearest walls can matter: per axis, can the slab
// below / above the own cell hold a point within the bound at all (six comparisons) — the cells of THAT box, typically 1 or 3
// instead of 26, are dealt to the lanes and tested exactly.  (Testing all 26 cells of every query, ~30 fp64 instructions each, made
// the launch issue-bound: 80 of its 108 us at 0.45 M points.)
template <int G>
__device__ __forceinline__ void o3d_shell1(const O3dQuery& q, const GridIndex& gi, const O3dRec* __restrict__ rec, int sub, double bound, O3dBest& b) {
  constexpr int Q = 2, kCand = 2;
  const NGrid& g = gi.g;
  const double hx = g.cell - q.lx, hy = g.cell - q.ly, hz = g.cell - q.lz;
  const double g2x[2] = {q.lx * q.lx, hx * hx}, g2y[2] = {q.ly * q.ly, hy * hy}, g2z[2] = {q.lz * q.lz, hz * hz};
  // necessary for any cell beyond that wall; a slab that cannot be reached is turned away as a whole
  auto reach = [&](double g2) {
    const double lb = g2 * (1.0 - 1e-9) - q.margin;
    const bool in = !(lb > bound);
    if (!in) o3d_exclude(b, lb);
    return in;
  };
  const int x0 = reach(g2x[0]) ? -1 : 0, sx = (reach(g2x[1]) ? 1 : 0) - x0 + 1;
  const int y0 = reach(g2y[0]) ? -1 : 0, sy = (reach(g2y[1]) ? 1 : 0) - y0 + 1;
  const int z0 = reach(g2z[0]) ? -1 : 0, sz = (reach(g2z[1]) ? 1 : 0) - z0 + 1;
  const int n = sx * sy * sz;
  auto div_small = [](int t, int d) { return d == 1 ? t : (d == 2 ? t >> 1 : (t * 11) >> 5); };  // t / d for d in 1..3, t < 32
  for (int t0 = sub; t0 < n; t0 += G * Q) {
    uint32_t c[Q];
    bool want[Q], any = false;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int t = t0 + k * G;
      const int q1 = div_small(t, sx), q2 = div_small(q1, sy);
      const int dx = x0 + (t - q1 * sx), dy = y0 + (q1 - q2 * sy), dz = z0 + q2;
      const int x = q.cx + dx, y = q.cy + dy, z = q.cz + dz;
      bool w = (t < n) & ((dx | dy | dz) != 0) & ((unsigned)x < (unsigned)g.nx) & ((unsigned)y < (unsigned)g.ny) & ((unsigned)z < (unsigned)g.nz);
      if (w) {
        const double gx = dx == 0 ? 0.0 : g2x[dx > 0], gy = dy == 0 ? 0.0 : g2y[dy > 0], gz = dz == 0 ? 0.0 : g2z[dz > 0];
        const double cell_lb = (gx + gy + gz) * (1.0 - 1e-9) - q.margin;
        w = !(cell_lb > bound);  // a tie at the bound is not "beyond": it stays in
        if (!w) o3d_exclude(b, cell_lb);
      }
      want[k] = w;
      any = any | w;
      c[k] = w ? ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx + (uint32_t)x : 0u;
    }
    if (!any) continue;
    o3d_batch<Q, kCand>(q, gi, rec, c, want, b);
  }
}

struct __attribute__((aligned(8))) O3dFarItem {  // a point whose search goes on in k_o3d_search_far, with what it has found so far
  double best, others;
  int32_t i, bj;
};
struct __attribute__((aligned(32))) O3dCert {  // where a point was when it was last searched, and how far every target point but its neighbour is
  double x, y, z, others;                      // others: squared; 0 = no certificate
  double nx, ny, nz, pad_;                     // the neighbour's coordinates (k_o3d_keep streams them instead of gathering from the target)
};
// the end of a search: the correspondence and the certificate
__device__ __forceinline__ void o3d_finish(const O3dQuery& q, O3dBest b, double r2, int64_t i, const double* __restrict__ tgt, int32_t* __restrict__ corr,
                                           O3dCert* __restrict__ cert) {
  const bool hit = b.j >= 0 && b.d < r2;
  if (!hit) b.others = fmin(b.others, b.d);  // a nearest point beyond the radius is not kept: it is one of the others
  corr[i] = hit ? b.j : -1;
  O3dCert c;
  c.x = q.qx;
  c.y = q.qy;
  c.z = q.qz;
  c.others = b.others;
  const size_t j = hit ? (size_t)b.j : 0;
  c.nx = tgt[3 * j];
  c.ny = tgt[3 * j + 1];
  c.nz = tgt[3 * j + 2];
  c.pad_ = 0.0;
  cert[i] = c;
}
// appends the flagged lanes of the BLOCK to a list: one atomic per block (thousands of waves adding to one counter take tens of
// microseconds: same-address atomics are served one after the other).  Every thread of the block must call it.
template <int NT = kB>
__device__ __forceinline__ uint32_t o3d_block_slot(bool flag, uint32_t* __restrict__ count) {
  constexpr int NW = NT / 64;
  __shared__ uint32_t s_n[NW + 1];
  __syncthreads();  // the last call's readers are done with s_n
  const unsigned long long mask = __ballot(flag);
  const int lane = (int)(threadIdx.x & 63), w = (int)(threadIdx.x >> 6);
  if (lane == 0) s_n[w] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const uint32_t c = s_n[k];
      s_n[k] = total;
      total += c;
    }
    s_n[NW] = total ? atomicAdd(count, total) : 0u;
  }
  __syncthreads();
  return s_n[NW] + s_n[w] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// counts[0] = points on the search list, counts[1] = points on the far list (both cleared behind the pass by k_o3d_fold)
constexpr int kKeepThreads = 1024;  // one atomic per block onto the list counter: 0.6 k of them at 0.6 M points instead of 2.3 k
__global__ void __launch_bounds__(kKeepThreads) k_o3d_keep(double* __restrict__ pcd, int64_t Ns, O3dPose Tm, int apply,
                                                 const double* __restrict__ tgt, double r2, int32_t* __restrict__ corr, O3dCert* __restrict__ cert,
                                                 uint32_t* __restrict__ list, uint32_t* __restrict__ counts O3S_DBG_PARAM) {
  const int64_t i = (int64_t)blockIdx.x * kKeepThreads + threadIdx.x;
  bool search = false;
  if (i < Ns) {
    double px = pcd[3 * i], py = pcd[3 * i + 1], pz = pcd[3 * i + 2];
    if (apply) {  // PointCloud::Transform(update)
      o3d_apply(Tm.m, px, py, pz);
      pcd[3 * i] = px;
      pcd[3 * i + 1] = py;
      pcd[3 * i + 2] = pz;
    }
    const O3dCert c = cert[i];
    const int32_t inc = corr[i];
    search = true;
    if (c.others > 0.0 && !O3S_DBG(64)) {  // (hooks build, 64: no certificate is honoured — every point is searched again)
      // distances rounded against the decision: L down, delta and the neighbour's distance up
      const double L = sqrt(c.others) * (1.0 - 1e-12);
      const double delta = sqrt(o3d_dist2(px, py, pz, c.x, c.y, c.z)) * (1.0 + 1e-12);
      if (inc >= 0) {
        const double d0 = o3d_dist2(px, py, pz, c.nx, c.ny, c.nz);  // the neighbour's coordinates, as the target holds them
        if (L - delta > sqrt(d0) * (1.0 + 1e-12)) {  // every other target point is further away than the neighbour: it is still the nearest
          search = false;
          if (!(d0 < r2)) {  // ... but has left the radius: no correspondence, and the certificate no longer names its exception
            corr[i] = -1;
            cert[i].others = 0.0;
          }
        }
      } else if (L - delta > sqrt(r2) * (1.0 + 1e-12)) {
        search = false;  // still nothing inside the radius
      }
    }
  }
  const uint32_t slot = o3d_block_slot<kKeepThreads>(search, counts);
  if (search) list[slot] = (uint32_t)i;
}

template <int G>
__global__ void __launch_bounds__(kB, 6) k_o3d_search(const double* __restrict__ pcd, int64_t Ns, GridIndex gi, const O3dRec* __restrict__ rec,
                                                   const double* __restrict__ tgt, double r2, O3dReach rc, int32_t* __restrict__ corr,
                                                   O3dCert* __restrict__ cert, int use_inc, const uint32_t* __restrict__ list /*nullptr: every point*/,
                                                   O3dFarItem* __restrict__ far, uint32_t* __restrict__ counts O3S_DBG_PARAM) {
  constexpr int kCand = 4;
  const int sub = (int)(threadIdx.x & (G - 1));
  const int64_t n = list ? (int64_t)counts[0] : Ns;
  const NGrid g = gi.g;
  // blocks stride over the points: the launch of a later pass does not know how short its list is (block-uniform trip count: the
  // list append below synchronises the block)
  for (int64_t blk = blockIdx.x; blk * (kB / G) < n; blk += gridDim.x) {
    const int64_t k_raw = (blk * kB + threadIdx.x) / G;
    const bool valid = k_raw < n;
    const int64_t k = valid ? k_raw : n - 1;
    const int64_t i = list ? (int64_t)list[k] : k;
    const O3dQuery q = o3d_query(pcd, i, g);
    // a NaN or infinite source point has no neighbour (Open3D's KD-tree returns none: every distance test fails); it is settled here,
    // before any shell logic — its cell, offsets and margin are not numbers, and a walk sized from them would never end
    const bool finite = o3d_finite(q);
    O3dBest b;
    b.d = __builtin_huge_val();
    b.others = __builtin_huge_val();
    b.j = -1;
    if (finite && use_inc && !O3S_DBG(4)) {
      const int32_t inc = corr[i];
      if (inc >= 0) {
        b.d = o3d_dist2(q.qx, q.qy, q.qz, tgt[3 * (size_t)inc], tgt[3 * (size_t)inc + 1], tgt[3 * (size_t)inc + 2]);
        b.j = inc;
      }
    }
    if (finite && q.r0 == 0 && !O3S_DBG(1)) {  // the own cell, its candidates dealt to the lanes
      const uint32_t c = ((uint32_t)q.cz * (uint32_t)g.ny + (uint32_t)q.cy) * (uint32_t)g.nx + (uint32_t)q.cx;
      const uint32_t jb = gi.cbeg[c], je = gi.cend[c];
      for (uint32_t j0 = jb + (uint32_t)sub; __any(j0 < je); j0 += (uint32_t)(G * kCand)) {
#pragma unroll
        for (int t = 0; t < kCand; ++t) {
          const uint32_t j = j0 + (uint32_t)(t * G);
          o3d_cand(q, rec + (j < je ? j : jb), j < je, b);
        }
      }
      o3d_group_min<G>(b);
    }
    int rr = max(1, q.r0);
    double bound = o3d_bound(b, rc);
    if (finite && rr == 1 && !O3S_DBG(2)) {
      if (o3d_shell_open(q, g, 1, bound)) {
        o3d_shell1<G>(q, gi, rec, sub, bound, b);
        o3d_group_min<G>(b);
        bound = o3d_bound(b, rc);
        rr = 2;
      } else {
        rr = q.rmax + 1;  // settled in the own cell: everything else is beyond the walls
        o3d_exclude(b, o3d_shell_lb2(q, g, 1));
      }
    }
    const bool more = finite && rr <= rc.r_cap && o3d_shell_open(q, g, rr, bound);
    if (!finite) b.others = 0.0;  // no certificate: the point is looked at (and settled) again in every pass
    if (finite && !more && rr <= q.rmax) o3d_exclude(b, o3d_shell_lb2(q, g, rr));  // the shells never entered
    const bool to_far = valid && more && sub == 0;
    const uint32_t slot = o3d_block_slot(to_far, counts + 1);
    if (to_far) {
      O3dFarItem it;
      it.best = b.d;
      it.others = b.others;
      it.i = (int32_t)i;
      it.bj = b.j;
      far[slot] = it;
    }
    if (valid && !more && sub == 0) o3d_finish(q, b, r2, i, tgt, corr, cert);
  }
}

// W lanes per listed point (groups stride over the list).  The walk of a point is a chain of dependent round trips, so for a long
// list what counts is how many points are in flight: first pass of the closed-loop run's four refinements with 64 / 32 / 16 lanes per
// point: 160 / 113 / 82, 226 / 157 / 132, 161 / 116 / 82, 64 / 47 / 45 us (8 lanes cannot hold the 2 r + 1 = 9 gap terms of four
// shells).  A short list — the passes after the second, whose certificates settle most points — is done when its slowest point is,
// and a point is fastest with the whole wave on it (10 us against 22): the host launches the 16-lane instantiation for the first three
// passes after a placement and the 64-lane one afterwards (either is exact for any list).
template <int W>
__global__ void __launch_bounds__(kB, W == 64 ? 1 : 5) k_o3d_search_far(const double* __restrict__ pcd, GridIndex gi, const O3dRec* __restrict__ rec,
                                                       const double* __restrict__ tgt, double r2, O3dReach rc, int32_t* __restrict__ corr, O3dCert* __restrict__ cert,
                                                       const O3dFarItem* __restrict__ far, const uint32_t* __restrict__ counts) {
  const uint32_t n = counts[1];
  const int lane = (int)(threadIdx.x & (W - 1));
  const uint32_t n_groups = gridDim.x * (kB / W);
  const NGrid g = gi.g;
  for (uint32_t w = blockIdx.x * (kB / W) + (threadIdx.x / W); w < n; w += n_groups) {
    const O3dFarItem it = far[w];
    const O3dQuery q = o3d_query(pcd, (int64_t)it.i, g);
    O3dBest b;
    b.d = it.best;
    b.others = it.others;
    b.j = it.bj;
    // the shells the search reaches into (uniform within the group: one query)
    const double bound = o3d_bound(b, rc);
    const int r_lo = max(2, q.r0);
    int r_hi = r_lo - 1;
    while (r_hi + 1 <= rc.r_cap && o3d_shell_open(q, g, r_hi + 1, bound)) ++r_hi;
    if (r_hi >= r_lo && 2 * r_hi + 1 <= W) {
      o3d_rows_wave<4, W>(q, gi, rec, r_lo, r_hi, lane, rc, b);
      o3d_group_min<W>(b);
      if (r_hi + 1 <= q.rmax) o3d_exclude(b, o3d_shell_lb2(q, g, r_hi + 1));  // the shells beyond the cube
    } else {
      for (int rr = r_lo; rr <= rc.r_cap && o3d_shell_open(q, g, rr, o3d_bound(b, rc)); ++rr) {
        o3d_shell<W, 2, 2>(q, gi, rec, rr, lane, rc, b);
        o3d_group_min<W>(b);
      }
      b.others = 0.0;  // no certificate from this path
    }
    if (lane == 0) o3d_finish(q, b, r2, (int64_t)it.i, tgt, corr, cert);
  }
}

// The sums of the NEXT ComputeTransformation (mode 0) or of the information matrix (mode 1) over the correspondences the search
// left in corr[], one lane per source point; the distance is formed again from the same coordinates in the same order.
template <int PHASE>
__global__ void __launch_bounds__(kB) k_o3d_corr(const double* __restrict__ pcd, int64_t Ns, GridIndex gi, const double* __restrict__ tgt,
                                                 const double* __restrict__ tn, double r2, int mode, int32_t* __restrict__ corr,
                                                 double* __restrict__ part /*[kAccComps][gridDim.x]*/) {
  static_assert(PHASE == 1, "the search is k_o3d_search");
  __shared__ double sh[4][kAccComps];
  double acc[kAccComps];
#pragma unroll
  for (int c = 0; c < kAccComps; ++c) acc[c] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < Ns; i += (int64_t)gridDim.x * kB) {
    const double qx = pcd[3 * i], qy = pcd[3 * i + 1], qz = pcd[3 * i + 2];
    double best = __builtin_huge_val();
    int32_t bj = -1;
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
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < kAccComps; ++c) {
    const double v = wave_sum_f64(acc[c]);
    if (l == 0) sh[w][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAccComps)
    part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// one wave per component (grid = kAccComps): lane l adds the partials l, l + 64, ... in that order, eight loads in flight at a
// time, then the wave's fixed tree.  (One block walking all 30 components wave by wave took 65 us per pass: 256 dependent
// round trips; the order of the additions — and so the result — is the same.)
// counts[2]: the ticket of the blocks.  post (nullable): 30 doubles + a sequence word in host-coherent pinned memory — the block that
// draws the last ticket hands the sums to the host, which polls the word (a copy + stream synchronisation per pass was ~20 us of a
// 100 us pass).  Hand-over between blocks: result stored, fence, ticket; the last block reads the results with agent-scope loads.
__global__ void __launch_bounds__(64) k_o3d_fold(const double* __restrict__ part, int nb, double* __restrict__ out /*kAccComps*/, uint32_t* __restrict__ counts,
                                                 double* __restrict__ post, uint32_t seq) {
  const int c = blockIdx.x, l = threadIdx.x;
  if (c == 0 && l < 2) counts[l] = 0u;  // the search's two work lists are empty again for the next pass
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
  if (!post) {
    if (l == 0) out[c] = s;
    return;
  }
  uint32_t ticket = 0;
  if (l == 0) {
    __hip_atomic_store(&out[c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    ticket = atomicAdd(&counts[2], 1u);
  }
  ticket = (uint32_t)__shfl((int)ticket, 0);
  if (ticket != (uint32_t)gridDim.x - 1u) return;
  __threadfence();
  if (l < (int)gridDim.x) {
    const double v = __hip_atomic_load(&out[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&post[l], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  if (l == 0) {
    counts[2] = 0u;
    __hip_atomic_store(reinterpret_cast<uint32_t*>(post + 32), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
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
  Buf d_src, d_tgt, d_tn, d_corr, d_part, d_sum, d_rec, d_far, d_far_count, d_cert, d_list;
  const double* tgt = nullptr;  // the target cloud the kernels read: d_tgt / d_tn, or arrays that already live in HBM
  const double* tn = nullptr;
  Arena sort_arena;
  int nb = 0;
  double* h_post = nullptr;      // 30 sums + sequence word (at double 32) in host-coherent pinned memory, written by k_o3d_fold
  double* h_post_dev = nullptr;  // the same as the device addresses it
  uint32_t post_seq = 0;
  int pass_no = 0;               // passes since the source was placed (o3d_place_source): the first two search nearly every point
  bool corr_valid = false;  // d_corr holds the correspondences of an earlier pass over the same source order: bounds for the next search
  // what a registration leaves behind for the information matrix of the same pair (o3d_info_after_icp)
  Buf d_orig;                    // the source as given, when it came from the host
  const double* orig = nullptr;  // the source as given, on the device (d_orig or the caller's resident array)
  const uint32_t* order = nullptr;  // search order -> index into the source (lives in sort_arena)
  GridIndex gi{};
  int64_t n_src = 0;
  bool pair_ready = false;
  O3dIcpWork() = default;
  O3dIcpWork(const O3dIcpWork&) = delete;
  O3dIcpWork& operator=(const O3dIcpWork&) = delete;
  ~O3dIcpWork() {  // only ever runs through o3s_o3d_registration_release (the pool itself is never torn down)
    if (h_post) (void)hipHostFree(h_post);
  }
};

// Work areas of the registrations of a device, handed out per call and taken back: hipMalloc / hipFree stall the whole device for
// milliseconds (a refinement is 1-2 ms), so nothing is allocated per call and nothing per pair of submaps — a loop closure against
// another target finds the buffers of the last one.  o3s_o3d_registration_reserve sizes one area ahead of time.
size_t reg_overlap_arena_bytes(int64_t Ns, int64_t Nt);  // overlap_impl.h
void reg_warm_lane_streams(int device);                   // overlap_impl.h

struct RegArea {
  int device = -1;
  OverlapWork ov;                // overlap selection (overlap_impl.h)
  Buf ov_src, ov_tgt, ov_tgtn;   // the two selected clouds of o3s_o3d_registration_icp_submaps_overlap
  O3dIcpWork reg;
};
struct RegPool {
  std::mutex m;
  std::vector<std::unique_ptr<RegArea>> idle;
};
inline RegPool& reg_pool() {
  static RegPool* p = new RegPool;  // never destroyed: its buffers must not be freed behind the runtime's own teardown
  return *p;
}
struct RegLease {  // the calling thread's area for the duration of a call (the device is current)
  std::unique_ptr<RegArea> a;
  hipStream_t s = nullptr;  // the stream the call enqueues on
  bool ok = false;          // the call ended through end(O3S_OK): it has waited for its stream itself
  // Every early error return (a failed launch, O3S_ERR_HIP from a post wait, an empty selection behind enqueued compactions) can
  // leave kernels in flight that still read or write the area's buffers: the area goes back to the pool — where a registration
  // on another stream or thread may lease it at once — only behind a drained stream.  Successful calls have drained it already.
  int end(int rc) {
    ok = rc == O3S_OK;
    return rc;
  }
  explicit RegLease(int device, hipStream_t stream = nullptr) : s(stream) {
    RegPool& p = reg_pool();
    {
      std::lock_guard<std::mutex> g(p.m);
      for (size_t k = 0; k < p.idle.size(); ++k)
        if (p.idle[k]->device == device) {
          a = std::move(p.idle[k]);
          p.idle.erase(p.idle.begin() + (long)k);
          break;
        }
    }
    if (!a) {
      a.reset(new RegArea);
      a->device = device;
    }
  }
  ~RegLease() {
    if (!ok) (void)hipStreamSynchronize(s);
    RegPool& p = reg_pool();
    std::lock_guard<std::mutex> g(p.m);
    p.idle.push_back(std::move(a));
  }
  RegArea* operator->() { return a.get(); }
};

// the pinned post of the sums: allocated once per work area (a pinned allocation takes milliseconds: o3s_o3d_registration_reserve
// makes it ahead of time) and never freed (the areas live in a pool that is never torn down); without it the sums are copied back
inline void o3d_ensure_post(O3dIcpWork& w) {
  if (w.h_post) return;
  if (hipHostMalloc(reinterpret_cast<void**>(&w.h_post), 512, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
    std::memset(w.h_post, 0, 512);
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&w.h_post_dev), w.h_post, 0) != hipSuccess) w.h_post_dev = nullptr;
  } else {
    w.h_post = nullptr;
  }
}

// on_device: the pointers are device arrays (a resident submap): both clouds are read where they lie; the working copy of the
// source (placed by the current pose, in search order) is made by o3d_place_source
inline int o3d_prepare(O3dIcpWork& w, const double* source, int64_t Ns, const double* target, const double* tn, int64_t Nt, double max_dist,
                       GridIndex* gi, hipStream_t s, bool on_device = false, const unsigned long long* target_bounds = nullptr) {
  if (Ns > (int64_t)0x7fffffff || Nt > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  CK(w.d_src.alloc((size_t)Ns * 24));
  CK(w.d_corr.alloc((size_t)Ns * 4));
  w.pair_ready = false;
  if (on_device) {
    w.orig = source;
  } else {
    CK(w.d_orig.alloc((size_t)Ns * 24));
    CK(hipMemcpyAsync(w.d_orig.p, source, (size_t)Ns * 24, hipMemcpyHostToDevice, s));
    w.orig = w.d_orig.as<double>();
  }
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
  const int rc = build_grid_index(w.grid, w.tgt, Nt, max_dist * 0.5, rho, max_dist, gi, s, target_bounds);
  if (rc != O3S_OK) return rc;
  CK(w.d_rec.alloc((size_t)Nt * sizeof(O3dRec)));
  hipLaunchKernelGGL(k_o3d_records, dim3(nblk(Nt)), dim3(kB), 0, s, gi->sp, gi->vals, Nt, w.d_rec.as<O3dRec>());
  CK(hipGetLastError());
  w.corr_valid = false;
  CK(w.d_far.alloc((size_t)Ns * sizeof(O3dFarItem)));
  CK(w.d_cert.alloc((size_t)Ns * sizeof(O3dCert)));
  CK(w.d_list.alloc((size_t)Ns * 4));
  // the two list counters and the fold's ticket: every pass leaves them at zero (k_o3d_fold), but a pass that was cut short by an
  // error upstream may not have — a registration starts from zeros
  CK(w.d_far_count.alloc(256));
  CK(hipMemsetAsync(w.d_far_count.p, 0, 256, s));
  o3d_ensure_post(w);
  return O3S_OK;
}

// d_src = T . source in the order of the target grid's cells under T (T nullptr / identity: the points as they are).  One key
// pass over the source where it lies, the sort, one gather that applies T on the way (round 3: copy, transform in place, keys,
// sort, copy, gather).
inline int o3d_place_source(O3dIcpWork& w, int64_t Ns, const GridIndex& gi, const double* T, hipStream_t s) {
  const size_t n = (size_t)Ns;
  const size_t tb = sort_temp_bytes(Ns);
  CK(w.sort_arena.reserve(2 * Arena::pad(n * 8) + 2 * Arena::pad(n * 4) + Arena::pad(tb) + 4096));
  uint64_t* keys = w.sort_arena.take<uint64_t>(n);
  uint64_t* keys2 = w.sort_arena.take<uint64_t>(n);
  uint32_t* vals = w.sort_arena.take<uint32_t>(n);
  uint32_t* vals2 = w.sort_arena.take<uint32_t>(n);
  void* tmp = w.sort_arena.take<char>(tb);
  const int apply = (T && !h_is_identity(T)) ? 1 : 0;
  O3dPose Tp{};
  if (apply) std::memcpy(Tp.m, T, sizeof(Tp.m));
  hipLaunchKernelGGL(k_src_cell_keys, dim3(nblk(Ns)), dim3(kB), 0, s, w.orig, Ns, Tp, apply, gi.g, keys, vals);
  size_t tbb = tb;
  CK(sort_pairs(tmp, tbb, keys, keys2, vals, vals2, n, key_bits((uint64_t)gi.g.nx * (uint64_t)gi.g.ny * (uint64_t)gi.g.nz), s));  // cell indices of the target grid
  hipLaunchKernelGGL(k_o3d_place, dim3(nblk(Ns)), dim3(kB), 0, s, w.orig, vals2, Ns, Tp, apply, w.d_src.as<double>());
  CK(hipGetLastError());
  w.order = vals2;
  w.gi = gi;
  w.n_src = Ns;
  w.corr_valid = false;
  w.pass_no = 0;
  return O3S_OK;
}

// One pass of GetRegistrationResultAndCorrespondences: `update` (nullable) is applied to the working copy of the source first
// (PointCloud::Transform(update), Registration.cpp RegistrationICP), then the correspondences and the sums.
inline int o3d_corr_pass(O3dIcpWork& w, int64_t Ns, const GridIndex& gi, double r2, int mode, double* sums /*kAccComps*/, hipStream_t s,
                         const double* update = nullptr) {
  int G = 4;  // lanes per source point in the search
  if (const char* e = O3S_HOOK_ENV("O3S_O3D_G")) G = atoi(e);
  const unsigned nbs = (unsigned)((Ns * G + kB - 1) / kB);
  int kdbg = 0;  // hooks build: timing only: 1 = no own cell, 2 = no shell 1, 4 = no incumbent; 64 = certificates ignored (results stay valid)
  if (const char* e = O3S_HOOK_ENV("O3S_O3D_KDBG")) kdbg = atoi(e);
  (void)kdbg;
  uint32_t* counts = w.d_far_count.as<uint32_t>();
  const uint32_t* list = nullptr;
  // how far beyond what exactness needs a search looks (O3dReach): 10 % of the radius without a neighbour, 2 % of it beyond one
  const double r = std::sqrt(r2);
  O3dReach rc;
  double beyond = 1.1;
  if (const char* e = O3S_HOOK_ENV("O3S_O3D_BEYOND")) beyond = atof(e);  // hooks build: A/B
  rc.r2o = (beyond * r) * (beyond * r);
  {
    const double shells = std::ceil(beyond * r / gi.g.cell) + 1.0;
    rc.r_cap = (std::isfinite(shells) && shells < 1.0e9) ? (int)shells : 0x7fffffff;
  }
  rc.pad = 0.02 * r;  // closed-loop run, ms per refinement with pads of 0 / 1.5 / 3 / 6 / 10 %: 1.72 / 1.58 / 1.61 / 1.70 / 1.70 (the 8-pass one)
  if (const char* e = O3S_HOOK_ENV("O3S_O3D_PAD")) rc.pad = atof(e) * r;  // hooks build: A/B of the pad
  if (w.corr_valid) {  // every pass but the first: most points keep their neighbour without a search
    O3dPose Tp{};
    if (update) std::memcpy(Tp.m, update, sizeof(Tp.m));
    hipLaunchKernelGGL(k_o3d_keep, dim3((unsigned)((Ns + kKeepThreads - 1) / kKeepThreads)), dim3(kKeepThreads), 0, s, w.d_src.as<double>(), Ns, Tp, update ? 1 : 0, w.tgt, r2,
                       w.d_corr.as<int32_t>(), w.d_cert.as<O3dCert>(), w.d_list.as<uint32_t>(), counts O3S_DBG_ARG(kdbg));
    list = w.d_list.as<uint32_t>();
  } else if (update) {
    return O3S_ERR_BAD_ARGUMENT;  // the first pass runs on the source as placed
  }
#define O3S_O3D_SEARCH(GG) \
  hipLaunchKernelGGL(k_o3d_search<GG>, dim3(list ? std::min(nbs, 2048u) : nbs), dim3(kB), 0, s, w.d_src.as<double>(), Ns, gi, w.d_rec.as<O3dRec>(), w.tgt, r2, rc, w.d_corr.as<int32_t>(), \
                     w.d_cert.as<O3dCert>(), w.corr_valid ? 1 : 0, list, w.d_far.as<O3dFarItem>(), counts O3S_DBG_ARG(kdbg))
  if (G == 1) O3S_O3D_SEARCH(1);
  else if (G == 2) O3S_O3D_SEARCH(2);
  else if (G == 8) O3S_O3D_SEARCH(8);
  else O3S_O3D_SEARCH(4);
#undef O3S_O3D_SEARCH
  if (O3S_HOOK_ENV("O3S_O3D_DBG")) {  // hooks build: how many points were searched / went onto the far list
    uint32_t n[2] = {0, 0};
    (void)hipMemcpyAsync(n, counts, 8, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    std::fprintf(stderr, "o3d pass: Ns=%lld searched=%u far=%u cell=%.3f grid=%dx%dx%d later_pass=%d\n", (long long)Ns, list ? n[0] : (uint32_t)Ns, n[1],
                 gi.g.cell, gi.g.nx, gi.g.ny, gi.g.nz, (int)w.corr_valid);
  }
  if (w.pass_no < 3)
    hipLaunchKernelGGL(k_o3d_search_far<16>, dim3(kO3dFarBlocks), dim3(kB), 0, s, w.d_src.as<double>(), gi, w.d_rec.as<O3dRec>(), w.tgt, r2, rc,
                       w.d_corr.as<int32_t>(), w.d_cert.as<O3dCert>(), w.d_far.as<O3dFarItem>(), counts);
  else
    hipLaunchKernelGGL(k_o3d_search_far<64>, dim3(kO3dFarBlocks / 4), dim3(kB), 0, s,  // short lists: 2 048 waves (an empty block costs too)
                       w.d_src.as<double>(), gi, w.d_rec.as<O3dRec>(), w.tgt, r2, rc,
                       w.d_corr.as<int32_t>(), w.d_cert.as<O3dCert>(), w.d_far.as<O3dFarItem>(), counts);
  ++w.pass_no;
  w.corr_valid = true;
  hipLaunchKernelGGL(k_o3d_corr<1>, dim3(w.nb), dim3(kB), 0, s, w.d_src.as<double>(), Ns, gi, w.tgt, w.tn, r2, mode,
                     w.d_corr.as<int32_t>(), w.d_part.as<double>());
  const bool post = w.h_post && w.h_post_dev;
  if (post && ++w.post_seq == 0) ++w.post_seq;
  hipLaunchKernelGGL(k_o3d_fold, dim3(kAccComps), dim3(64), 0, s, w.d_part.as<double>(), w.nb, w.d_sum.as<double>(), counts, post ? w.h_post_dev : nullptr,
                     w.post_seq);
  CK(hipGetLastError());
  if (post) {
    const uint32_t* word = reinterpret_cast<const uint32_t*>(w.h_post + 32);
    double t_guard = o3s_cloud::poll_now_us();
    for (;;) {
      bool seen = false;
      for (int spin = 0; spin < 4096 && !seen; ++spin) seen = __atomic_load_n(word, __ATOMIC_ACQUIRE) == w.post_seq;
      if (seen) break;
      const double t = o3s_cloud::poll_now_us();
      if (t - t_guard < o3s_cloud::kPollGuardUs) continue;  // the stream is only looked at every 200 us of waiting (cloud_dev.h)
      t_guard = t;
      const hipError_t q = hipStreamQuery(s);  // a fault upstream must not leave the host spinning
      if (q == hipSuccess) {
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != w.post_seq) return O3S_ERR_HIP;
        break;
      }
      if (q != hipErrorNotReady) return O3S_ERR_HIP;
    }
    for (int c = 0; c < kAccComps; ++c) sums[c] = w.h_post[c];
    return O3S_OK;
  }
  CK(hipMemcpyAsync(sums, w.d_sum.p, kAccComps * 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  return O3S_OK;
}

}  // namespace o3s_cloud
}  // namespace

extern "C" {

}  // extern "C"

namespace {
namespace o3s_cloud {
// sizes one leased work area for clouds of up to ns / nt points
inline int reg_reserve_area(RegLease& area, int64_t max_source_points, int64_t max_target_points) {
  const size_t ns = (size_t)max_source_points, nt = (size_t)max_target_points;
  O3dIcpWork& w = area->reg;
  CK(w.d_src.alloc(ns * 24));
  CK(w.d_orig.alloc(ns * 24));
  CK(w.d_corr.alloc(ns * 4));
  CK(w.d_far.alloc(ns * sizeof(O3dFarItem)));
  CK(w.d_cert.alloc(ns * sizeof(O3dCert)));
  CK(w.d_list.alloc(ns * 4));
  CK(w.d_tgt.alloc(nt * 24));
  CK(w.d_tn.alloc(nt * 24));
  CK(w.d_rec.alloc(nt * sizeof(O3dRec)));
  CK(w.d_part.alloc((size_t)2048 * kAccComps * 8));
  CK(w.d_sum.alloc(kAccComps * 8));
  o3d_ensure_post(w);
  CK(w.d_far_count.alloc(256));
  CK(w.grid.arena.reserve(grid_index_arena_bytes(max_target_points)));
  if (w.grid.cells_cap < kGridMaxCells * 8 + 4096) {
    if (w.grid.cells) (void)hipFree(w.grid.cells);
    w.grid.cells = nullptr;
    w.grid.cells_cap = 0;
    CK(hipMalloc(&w.grid.cells, kGridMaxCells * 8 + 4096));
    w.grid.cells_cap = kGridMaxCells * 8 + 4096;
  }
  CK(w.sort_arena.reserve(2 * Arena::pad(ns * 8) + 2 * Arena::pad(ns * 4) + Arena::pad(sort_temp_bytes(max_source_points)) + 4096));
  CK(area->ov_src.alloc(ns * 24));
  CK(area->ov_tgt.alloc(nt * 24));
  CK(area->ov_tgtn.alloc(nt * 24));
  CK(area->ov.arena.reserve(reg_overlap_arena_bytes(max_source_points, max_target_points)));
  w.pair_ready = false;
  return area.end(O3S_OK);
}
}  // namespace o3s_cloud
}  // namespace

extern "C" {

int o3s_o3d_registration_reserve_n(int device, int64_t max_source_points, int64_t max_target_points, int32_t count) {
  using namespace o3s_cloud;
  if (max_source_points <= 0 || max_target_points <= 0 || max_source_points > (int64_t)0x7fffffff || max_target_points > (int64_t)0x7fffffff || count < 1 ||
      count > 16)
    return O3S_ERR_BAD_ARGUMENT;
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  if (count > 1) reg_warm_lane_streams(device);  // the lanes' streams too (made once per device; ~3 ms each)
  std::vector<std::unique_ptr<RegLease>> held;  // all leased at once: `count` DIFFERENT areas are sized, then go back to the pool together
  for (int32_t k = 0; k < count; ++k) {
    held.emplace_back(new RegLease(device));
    const int r = reg_reserve_area(*held.back(), max_source_points, max_target_points);
    if (r != O3S_OK) return r;
  }
  return O3S_OK;
}

int o3s_o3d_registration_reserve(int device, int64_t max_source_points, int64_t max_target_points) {
  return o3s_o3d_registration_reserve_n(device, max_source_points, max_target_points, 1);
}

int o3s_o3d_registration_release(int device) {
  using namespace o3s_cloud;
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  if (hipDeviceSynchronize() != hipSuccess) return O3S_ERR_HIP;
  RegPool& p = reg_pool();
  std::lock_guard<std::mutex> g(p.m);
  for (size_t k = 0; k < p.idle.size();)
    if (p.idle[k]->device == device) p.idle.erase(p.idle.begin() + (long)k);
    else ++k;
  return O3S_OK;
}

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
                const double init[16], const o3s_o3d_icp_criteria* criteria, o3s_o3d_icp_result* result, hipStream_t s, bool on_device = false,
                const unsigned long long* target_bounds = nullptr /*of the target, when the caller has them (build_grid_index)*/) {
  if (!source || !target || !init || !result || Ns <= 0 || Nt <= 0 || !(max_dist > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  if (!target_normals) return O3S_ERR_BAD_SHAPE;  // "requires target pointcloud to have normals"
  o3s_o3d_icp_criteria cr;
  o3s_o3d_icp_default_criteria(&cr);
  if (criteria) cr = *criteria;
  int rc = O3S_OK;
  GridIndex gi;
  rc = o3d_prepare(w, source, Ns, target, target_normals, Nt, max_dist, &gi, s, on_device, target_bounds);
  if (rc != O3S_OK) return rc;
  const double r2 = max_dist * max_dist;
  double T[16];
  std::memcpy(T, init, sizeof(T));
  rc = o3d_place_source(w, Ns, gi, init, s);
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
    const double f0 = fitness(sums), e0 = rmse(sums);
    rc = o3d_corr_pass(w, Ns, gi, r2, 0, sums, s, update);  // pcd.Transform(update), then the correspondences at the new pose
    if (rc != O3S_OK) return rc;
    ++it;
    if (std::fabs(f0 - fitness(sums)) < cr.relative_fitness && std::fabs(e0 - rmse(sums)) < cr.relative_rmse) break;
  }
  std::memcpy(result->transformation, T, sizeof(T));
  result->fitness = fitness(sums);
  result->inlier_rmse = rmse(sums);
  result->correspondences = (int64_t)sums[29];
  result->iterations = it;
  w.pair_ready = true;
  CK(hipStreamSynchronize(s));  // the passes return at their post, a moment before the last kernel retires: the call ends on a drained stream
  return O3S_OK;
}

// GetInformationMatrixFromPointClouds(source, target, max_dist, T) right after o3d_icp_run on the SAME work area and pair: the index
// over the target, its records and the search order of the source are still there, and the correspondences of the last pass bound
// the search.  The source is placed afresh from the cloud as given (T . p, not the chain of the ICP's updates), so the
// correspondences are those of a stand-alone call; only the order in which the 21 + 1 sums are added differs (search order under the
// registration's initial guess instead of under T).
int o3d_info_after_icp(O3dIcpWork& w, double max_dist, const double T[16], double info[36], hipStream_t s) {
  if (!w.pair_ready || !T || !info || !(max_dist > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  const int64_t Ns = w.n_src;
  const int apply = h_is_identity(T) ? 0 : 1;
  O3dPose Tp{};
  if (apply) std::memcpy(Tp.m, T, sizeof(Tp.m));
  hipLaunchKernelGGL(k_o3d_place, dim3(nblk(Ns)), dim3(kB), 0, s, w.orig, w.order, Ns, Tp, apply, w.d_src.as<double>());
  CK(hipGetLastError());
  double sums[kAccComps];
  const int rc = o3d_corr_pass(w, Ns, w.gi, max_dist * max_dist, 1, sums, s);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(s));
  int t = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) {
      info[b * 6 + a] = sums[t];
      info[a * 6 + b] = sums[t];
      ++t;
    }
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
  rc = o3d_place_source(w, Ns, gi, T, s);
  if (rc != O3S_OK) return rc;
  double sums[kAccComps];
  rc = o3d_corr_pass(w, Ns, gi, max_dist * max_dist, 1, sums, s);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(s));
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
  RegLease area(device, nullptr);
  return area.end(o3d_icp_run(area->reg, source, Ns, target, target_normals, Nt, max_dist, init, criteria, result, nullptr));
}

int o3s_o3d_information_matrix(int device, const double* source, int64_t Ns, const double* target, int64_t Nt, double max_dist, const double T[16],
                               double info[36]) {
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  RegLease area(device, nullptr);
  return area.end(o3d_info_run(area->reg, source, Ns, target, Nt, max_dist, T, info, nullptr));
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
    RegLease area(device, s);  // this lane's work area: no allocation once it has seen the lane's largest pair
    area.ok = true;            // (the lane drains its stream after every failed pair, below, and again before it leaves)
    O3dIcpWork& w = area->reg;
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
        r = o3d_info_after_icp(w, max_dist, results[k].transformation, infos + 36 * (size_t)k, s);
      if (r != O3S_OK) (void)hipStreamSynchronize(s);  // nothing of the failed pair may still be running when the next one re-uses the area
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
