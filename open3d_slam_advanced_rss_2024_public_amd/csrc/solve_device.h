// solve_device.h — single-lane device code that closes one ICP iteration on the GPU:
//   6x6 normal equations -> x          solvePossiblyUnderdeterminedLinearSystem  LPM/ErrorMinimizers/PointToPlane.cpp:185-238
//   x -> 4x4 step                      PointToPlaneErrorMinimizer::compute        LPM/ErrorMinimizers/PointToPlane.cpp:276-332
//   T_iter update + stop rules         LPM/ICP.cpp:433-445, LPM/TransformationCheckersImpl.cpp:57-76,102-158
// Everything is fp32 in the reference's operation order (this TU is compiled with -ffp-contract=off); the only fp64
// pieces are the last-resort pseudo-inverse (the reference's double JacobiSVD) and sin/cos/atan2 evaluated in fp64 and
// rounded once so that they agree with a correctly rounded host libm (sincos_cr below; atan2 through the math library).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "icp_types.h"

#pragma clang fp contract(off)

namespace o3s {
namespace dev {

#define O3S_EPS_F 1.1920928955078125e-07f
#define O3S_FLT_MIN 1.17549435e-38f

// All scratch of the solve lives in ONE struct that k_solve places in LDS: the algorithms index small matrices with
// run-time subscripts, which in private memory would become scratch (HBM) traffic on a single lane.
struct Sys6 {
  float A[6][6];  // A[r][c]
  float b[6];
};

struct SolveWork {
  Sys6 S;
  float qr[6][6];
  float h[6];
  int rowT[6], colT[6], perm[6];
  int nonzero;
  float maxpivot;
  float Q[6][6], R1[6][6], G[6][6], L[6][6];
  float rhs[6], y[6], xt[6], x[6], ax[6], df[6], qa[6];
  float xfast[6];             // k_solve: what lane 0 solved while lane 64 decided whether it may be used (llt_fast_solve / llt_fast_bound)
  int fast_solved, fast_bounded;
  double pA[6][6], pV[6][6];  // fp64 fallback (pinv_solve_f64)
};

__device__ inline float atan2f_cr(float y, float x) { return (float)atan2((double)y, (double)x); }

// sin and cos of an fp32 angle, each evaluated in fp64 and rounded once — what a correctly rounded host sinf / cosf returns
// (Eigen's AngleAxis::toRotationMatrix calls std::sin / std::cos on the float angle).  Written out instead of calling the math
// library's sincos(double): that call was 1.5 us of k_solve's single-lane critical path (3 650 cycles: the library carries a
// Payne-Hanek reduction for arguments no ICP step has, with its words in private memory — the kernel's only scratch).  The
// algorithm is fdlibm's: k = nearest integer to x * 2/pi, y = x - k * pi/2 by the three-part Cody-Waite subtraction of
// __ieee754_rem_pio2 (k * pio2_1 is exact for |k| < 2^20: 33-bit head), then __kernel_sin / __kernel_cos with the tail of y on
// [-pi/4, pi/4] (|error| < 2^-57) and the quadrant.  Checked on the host against glibc's sin / cos rounded to float: 4e8 angles from
// 2^-27 to 2^21 rad, both signs, no difference (the build has no FMA contraction, like the device).  Beyond ~2^21 rad the
// reduction is no longer exact (a rotation step of millions of radians is a diverged solve whatever its last bit).
__device__ inline void sincos_cr(float a, float* s, float* c) {
  const double x = (double)a;
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21;
  const double fn = rint(x * invpio2);
  const double t = x - fn * pio2_1;
  double w = fn * pio2_2;
  const double r = t - w;
  w = fn * pio2_2t - ((t - r) - w);
  const double y0 = r - w;
  const double y1 = (r - y0) - w;
  const double z = y0 * y0;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double v = z * y0;
  const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  const double sn = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
  const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const double hz = 0.5 * z;
  const double ww = 1.0 - hz;
  const double cs = ww + (((1.0 - ww) - hz) + (z * rc - y0 * y1));
  const int q = (int)((long long)fn & 3ll);
  const double sd = (q == 0) ? sn : (q == 1) ? cs : (q == 2) ? -sn : -cs;
  const double cd = (q == 0) ? cs : (q == 1) ? -sn : (q == 2) ? -cs : sn;
  *s = (float)sd;
  *c = (float)cd;
}

// ---- Cholesky (Eigen LLT, lower, unblocked) + solve, n <= 6 ----------------------------------------------------
__device__ inline void llt_solve(float (*L)[6], int n, const float* rhs, float* y, float* x) {
  for (int k = 0; k < n; ++k) {
    float d = L[k][k];
    if (k > 0) {
      float s = 0.f;
      for (int j = 0; j < k; ++j) s = s + L[k][j] * L[k][j];
      d = d - s;
    }
    if (d <= 0.f) break;  // Eigen reports NumericalIssue and solve() still runs on the partial factor
    d = sqrtf(d);
    L[k][k] = d;
    for (int r = k + 1; r < n; ++r) {
      float s = 0.f;
      for (int j = 0; j < k; ++j) s = s + L[r][j] * L[k][j];
      L[r][k] = (L[r][k] - s) / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    float s = rhs[i];
    for (int j = 0; j < i; ++j) s = s - L[i][j] * y[j];
    y[i] = s / L[i][i];
  }
  for (int i = n - 1; i >= 0; --i) {
    float s = y[i];
    for (int j = i + 1; j < n; ++j) s = s - L[j][i] * x[j];
    x[i] = s / L[i][i];
  }
}

// H = I - tau v v^T, v = [1, w.qr[k+1..5][k]], applied on the left of the (6-k) x nc block of M at (k, c0)
__device__ inline void house_left(const SolveWork& w, float (*M)[6], int k, int c0, int nc, float tau) {
  const int nr = 6 - k;
  if (nc <= 0) return;
  if (nr == 1) {
    for (int c = 0; c < nc; ++c) M[k][c0 + c] = M[k][c0 + c] * (1.f - tau);
    return;
  }
  if (tau == 0.f) return;
  for (int c = 0; c < nc; ++c) {
    float t = 0.f;
    for (int r = 1; r < nr; ++r) t = t + w.qr[k + r][k] * M[k + r][c0 + c];
    t = t + M[k][c0 + c];
    M[k][c0 + c] = M[k][c0 + c] - tau * t;
    for (int r = 1; r < nr; ++r) M[k + r][c0 + c] = M[k + r][c0 + c] - tau * w.qr[k + r][k] * t;
  }
}

// Eigen FullPivHouseholderQR of S.A, kept as (qr, hCoeffs, transpositions)
__device__ inline void fpqr_compute(SolveWork& w) {
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) w.qr[r][c] = w.S.A[r][c];
  const float precision = O3S_EPS_F * 6.f;
  w.nonzero = 6;
  w.maxpivot = 0.f;
  float biggest = 0.f;
  for (int k = 0; k < 6; ++k) {
    int rb = k, cb = k;
    float best = -1.f;
    for (int c = k; c < 6; ++c)      // column-major visit order, first maximum wins (Eigen's maxCoeff visitor)
      for (int r = k; r < 6; ++r) {
        const float v = fabsf(w.qr[r][c]);
        if (v > best) {
          best = v;
          rb = r;
          cb = c;
        }
      }
    if (k == 0) biggest = best;
    if (fabsf(best) <= fabsf(biggest) * precision) {
      w.nonzero = k;
      for (int i = k; i < 6; ++i) {
        w.rowT[i] = i;
        w.colT[i] = i;
        w.h[i] = 0.f;
      }
      break;
    }
    w.rowT[k] = rb;
    w.colT[k] = cb;
    if (k != rb)
      for (int c = k; c < 6; ++c) {
        const float t = w.qr[k][c];
        w.qr[k][c] = w.qr[rb][c];
        w.qr[rb][c] = t;
      }
    if (k != cb)
      for (int r = 0; r < 6; ++r) {
        const float t = w.qr[r][k];
        w.qr[r][k] = w.qr[r][cb];
        w.qr[r][cb] = t;
      }
    float tail = 0.f;
    for (int r = k + 1; r < 6; ++r) tail = tail + w.qr[r][k] * w.qr[r][k];
    const float c0 = w.qr[k][k];
    float tau, beta;
    if (tail <= O3S_FLT_MIN) {
      tau = 0.f;
      beta = c0;
      for (int r = k + 1; r < 6; ++r) w.qr[r][k] = 0.f;
    } else {
      beta = sqrtf(c0 * c0 + tail);
      if (c0 >= 0.f) beta = -beta;
      for (int r = k + 1; r < 6; ++r) w.qr[r][k] = w.qr[r][k] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    w.h[k] = tau;
    w.qr[k][k] = beta;
    if (fabsf(beta) > w.maxpivot) w.maxpivot = fabsf(beta);
    house_left(w, w.qr, k, k + 1, 6 - k - 1, tau);
  }
  for (int i = 0; i < 6; ++i) w.perm[i] = i;
  for (int k = 0; k < 6; ++k) {
    const int t = w.perm[k];
    w.perm[k] = w.perm[w.colT[k]];
    w.perm[w.colT[k]] = t;
  }
}

__device__ inline int fpqr_rank(const SolveWork& w) {
  const float pre = fabsf(w.maxpivot) * (O3S_EPS_F * 6.f);
  int r = 0;
  for (int i = 0; i < w.nonzero; ++i) r += (fabsf(w.qr[i][i]) > pre) ? 1 : 0;
  return r;
}

__device__ inline void fpqr_Q(SolveWork& w) {
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) w.Q[r][c] = (r == c) ? 1.f : 0.f;
  for (int k = 5; k >= 0; --k) {
    house_left(w, w.Q, k, k, 6 - k, w.h[k]);
    if (w.rowT[k] != k)
      for (int c = 0; c < 6; ++c) {
        const float t = w.Q[k][c];
        w.Q[k][c] = w.Q[w.rowT[k]][c];
        w.Q[w.rowT[k]][c] = t;
      }
  }
}

// fp64 cyclic-Jacobi pseudo-inverse solve of the symmetric system (the double JacobiSVD least-squares fallback)
// A and V live in the caller's LDS work area (as private arrays they were the kernel's only scratch memory)
__device__ __noinline__ void pinv_solve_f64(const float (*Af)[6], const float* bf, float* x, double (*A)[6], double (*V)[6]) {
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) {
      A[r][c] = 0.5 * ((double)Af[r][c] + (double)Af[c][r]);
      V[r][c] = (r == c) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r + 1; c < 6; ++c) off += A[r][c] * A[r][c];
    if (off < 1e-300) break;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) {
          const double a = A[k][p], b = A[k][q];
          A[k][p] = c * a - s * b;
          A[k][q] = s * a + c * b;
        }
        for (int k = 0; k < 6; ++k) {
          const double a = A[p][k], b = A[q][k];
          A[p][k] = c * a - s * b;
          A[q][k] = s * a + c * b;
        }
        for (int k = 0; k < 6; ++k) {
          const double a = V[k][p], b = V[k][q];
          V[k][p] = c * a - s * b;
          V[k][q] = s * a + c * b;
        }
      }
  }
  double smax = 0;
  for (int i = 0; i < 6; ++i) smax = fmax(smax, fabs(A[i][i]));
  const double thr = fmax(smax * 6.0 * 2.220446049250313e-16, 2.2250738585072014e-308);
  double xd[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) {
    const double lam = A[i][i];
    if (fabs(lam) > thr) {
      double vb = 0;
      for (int k = 0; k < 6; ++k) vb += V[k][i] * (double)bf[k];
      const double coef = vb / lam;
      for (int k = 0; k < 6; ++k) xd[k] += V[k][i] * coef;
    }
  }
  for (int k = 0; k < 6; ++k) x[k] = (float)xd[k];
}

__device__ inline float nrm6(const float* v) {
  float s = 0.f;
  for (int i = 0; i < 6; ++i) s = s + v[i] * v[i];
  return sqrtf(s);
}

// ---- fast path of the rank decision -------------------------------------------------------------------------------
// The reference asks FullPivHouseholderQR whether A is invertible (all pivots > 6 eps * max pivot, eps = 2^-23) and
// then solves with LLT.  Every pivot of a (column-pivoted) QR is >= sigma_min(A) and none exceeds ||A||_2, so
// cond_2(A) <= cond_F(A) = ||A||_F ||A^-1||_F <= ||A||_F ||L^-1||_F^2 < 1e4 proves "invertible" with three orders of
// margin (the QR's own early-exit test included) — no QR needed.  The Cholesky factor is the one LLT::solve uses, in the same operation
// order, so x is bit-identical to the slow path.  All indices are compile-time constants: the 6x6 lives in registers.
// Returns false when the bound is not met (near-singular systems): the caller then runs the full QR path.
// The fast path comes in two halves that share nothing but their input, so that two lanes of two waves run them side by side
// (k_solve: the single-lane solve was 6 570 of the kernel's 22 800 cycles):
//   llt_fast_solve  the factor and the two substitutions: x, and whether every pivot was positive
//   llt_fast_bound  the factor again, its inverse and the two Frobenius norms: whether cond_F(A) < 1e4, i.e. whether x may be used
// Each forms the factor with the operations, in the order, of Eigen's LLT (llt_solve above).
__device__ inline void chol6_regs(float (&L)[6][6], bool& ok) {
  ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float d = L[k][k];
    if (k > 0) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < k; ++j) s = s + L[k][j] * L[k][j];
      d = d - s;
    }
    if (!(d > 0.f)) ok = false;
    d = sqrtf(d);
    L[k][k] = d;
#pragma unroll
    for (int r = k + 1; r < 6; ++r) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < k; ++j) s = s + L[r][j] * L[k][j];
      L[r][k] = (L[r][k] - s) / d;
    }
  }
}
__device__ inline bool llt_fast_solve(const float (*A)[6], const float* b, float* x) {
  float L[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) L[r][c] = A[r][c];
  bool ok;
  chol6_regs(L, ok);
  float y[6], xr[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = b[i];
#pragma unroll
    for (int j = 0; j < i; ++j) s = s - L[i][j] * y[j];
    y[i] = s / L[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = y[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) s = s - L[j][i] * xr[j];
    xr[i] = s / L[i][i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = xr[i];
  return ok;
}
__device__ inline bool llt_fast_bound(const float (*A)[6]) {
  float L[6][6];
  float an = 0.f;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      L[r][c] = A[r][c];
      an = an + A[r][c] * A[r][c];
    }
  bool ok;
  chol6_regs(L, ok);
  if (!ok) return false;
  // inverse of the factor; ||A^-1||_F = ||Linv^T Linv||_F <= ||Linv||_F^2, so an upper bound of cond_F^2 needs only the
  // 21 squares of Linv.  This part decides a branch, it never touches x: reciprocals are multiplied instead of divided
  // (15 IEEE divisions less on the single-lane critical path) and the bound is inflated by 1 % for their rounding.
  float Li[6][6], ri[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) ri[i] = 1.f / L[i][i];
  float li2 = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) Li[i][j] = 0.f;
    Li[i][i] = ri[i];
    li2 = li2 + ri[i] * ri[i];
#pragma unroll
    for (int j = 0; j < i; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = j; k < i; ++k) s = s + L[i][k] * Li[k][j];
      Li[i][j] = -s * ri[i];
      li2 = li2 + Li[i][j] * Li[i][j];
    }
  }
  const float in2 = 1.01f * (li2 * li2);
  return an * in2 < 1.0e8f;  // cond_F^2 < (1e4)^2 (false for NaN)
}
__device__ inline bool llt_fast_path(const float (*A)[6], const float* b, float* x) {  // both halves on one lane
  if (!llt_fast_bound(A)) return false;
  return llt_fast_solve(A, b, x);
}

// solves w.S into w.x; returns the branch taken: 0 LLT, 1 min-norm QR, 2 fp64 fallback
// the general path: the reference's own sequence (full-pivot QR rank decision, then LLT / minimum norm / fp64 pseudo-inverse)
__device__ __forceinline__ int solve_sys6_general(SolveWork& w) {
  fpqr_compute(w);
  const int rank = fpqr_rank(w);
  if (rank == 6) {
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) w.L[r][c] = w.S.A[r][c];
    llt_solve(w.L, 6, w.S.b, w.y, w.x);
    return 0;
  }
  if (rank == 0) {
    for (int i = 0; i < 6; ++i) w.x[i] = 0.f;
    return 1;
  }
  fpqr_Q(w);
  for (int i = 0; i < rank; ++i) {
    for (int j = 0; j < 6; ++j) {  // (Q1t * A)(i, j), Q1t(i,k) = Q(k,i)
      float s = 0.f;
      for (int k = 0; k < 6; ++k) s = s + w.Q[k][i] * w.S.A[k][j];
      w.qa[j] = s;
    }
    for (int j = 0; j < 6; ++j) w.R1[i][j] = w.qa[w.perm[j]];
    float s = 0.f;
    for (int k = 0; k < 6; ++k) s = s + w.Q[k][i] * w.S.b[k];
    w.rhs[i] = s;
  }
  for (int i = 0; i < rank; ++i)
    for (int j = 0; j < rank; ++j) {
      float t = 0.f;
      for (int k = 0; k < 6; ++k) t = t + w.R1[i][k] * w.R1[j][k];
      w.G[i][j] = t;
    }
  llt_solve(w.G, rank, w.rhs, w.y, w.xt);   // w.xt[0..rank) = (R1 R1^T)^-1 Q1t b
  for (int i = 0; i < rank; ++i) w.y[i] = w.xt[i];
  for (int j = 0; j < 6; ++j) {
    float s = 0.f;
    for (int i = 0; i < rank; ++i)
      if (j >= i) s = s + w.R1[i][j] * w.y[i];
    w.xt[j] = s;
  }
  for (int i = 0; i < 6; ++i) w.x[w.perm[i]] = w.xt[i];
  for (int i = 0; i < 6; ++i) {
    float s = 0.f;
    for (int k = 0; k < 6; ++k) s = s + w.S.A[i][k] * w.x[k];
    w.ax[i] = s;
    w.df[i] = w.S.b[i] - s;
  }
  const float nb = nrm6(w.S.b), nax = nrm6(w.ax), nd = nrm6(w.df);
  const float lo = fminf(nb * nb, nax * nax);
  if (!((nd * nd) <= 1e-5f * 1e-5f * lo)) {
    pinv_solve_f64(w.S.A, w.S.b, w.x, w.pA, w.pV);
    return 2;
  }
  return 1;
}
__device__ inline int solve_sys6(SolveWork& w) {  // one lane does it all (kept for callers without a second wave)
  float Ar[6][6], br[6], xr[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    br[r] = w.S.b[r];
#pragma unroll
    for (int c = 0; c < 6; ++c) Ar[r][c] = w.S.A[r][c];
  }
  if (llt_fast_path(Ar, br, xr)) {
#pragma unroll
    for (int r = 0; r < 6; ++r) w.x[r] = xr[r];
    return 0;
  }
  return solve_sys6_general(w);
}

// ---- 4x4 column-major helpers ------------------------------------------------------------------------------------
#define M4(m, r, c) (m)[(c)*4 + (r)]

__device__ inline void mul4(const float* A, const float* B, float* C) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float s = M4(A, r, 0) * M4(B, 0, c);
      s = s + M4(A, r, 1) * M4(B, 1, c);
      s = s + M4(A, r, 2) * M4(B, 2, c);
      s = s + M4(A, r, 3) * M4(B, 3, c);
      M4(C, r, c) = s;
    }
}

__device__ inline bool rigid_ok(const float* T) {
  const float d0 = M4(T, 0, 0) * (M4(T, 1, 1) * M4(T, 2, 2) - M4(T, 1, 2) * M4(T, 2, 1));
  const float d1 = M4(T, 0, 1) * (M4(T, 1, 0) * M4(T, 2, 2) - M4(T, 1, 2) * M4(T, 2, 0));
  const float d2 = M4(T, 0, 2) * (M4(T, 1, 0) * M4(T, 2, 1) - M4(T, 1, 1) * M4(T, 2, 0));
  const float det = d0 - d1 + d2;
  return !(fabsf(1.f - det) > 0.001f);
}

// x (6) + centroids -> step matrix  (PointToPlane.cpp:276-332)
__device__ inline void build_step(const float* x, const float* mp, const float* mq, float* T) {
  float n2 = x[0] * x[0];
  n2 = n2 + x[1] * x[1];
  n2 = n2 + x[2] * x[2];
  const float ang = sqrtf(n2);
  float ax[3] = {x[0], x[1], x[2]};
  const float wmax = fmaxf(fabsf(x[0]), fmaxf(fabsf(x[1]), fabsf(x[2])));
  const float a0 = x[0] / wmax, a1 = x[1] / wmax, a2 = x[2] / wmax;
  float z = a0 * a0;
  z = z + a1 * a1;
  z = z + a2 * a2;
  if (z > 0.f) {  // stableNormalized(); NaN compares false and leaves the vector as is
    const float den = sqrtf(z) * wmax;
    ax[0] = x[0] / den;
    ax[1] = x[1] / den;
    ax[2] = x[2] / den;
  }
  float s, c;
  sincos_cr(ang, &s, &c);  // fp64, rounded once each: std::sin / std::cos of the float angle on a correctly rounding host
  const float sx = s * ax[0], sy = s * ax[1], sz = s * ax[2];
  const float cx = (1.f - c) * ax[0], cy = (1.f - c) * ax[1], cz = (1.f - c) * ax[2];
  float R[3][3];
  float t = cx * ax[1];
  R[0][1] = t - sz;
  R[1][0] = t + sz;
  t = cx * ax[2];
  R[0][2] = t + sy;
  R[2][0] = t - sy;
  t = cy * ax[2];
  R[1][2] = t - sx;
  R[2][1] = t + sx;
  R[0][0] = cx * ax[0] + c;
  R[1][1] = cy * ax[1] + c;
  R[2][2] = cz * ax[2] + c;
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = 0.f;
  M4(T, 3, 3) = 1.f;
  bool nan = false;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      M4(T, r, cc) = R[r][cc];
      nan = nan || (R[r][cc] != R[r][cc]);
    }
    float v = R[r][0] * (-mp[0]);
    v = v + R[r][1] * (-mp[1]);
    v = v + R[r][2] * (-mp[2]);
    v = v + (x[3 + r] + mq[r]);
    M4(T, r, 3) = v;
    nan = nan || (v != v);
  }
  if (nan) {  // degenerate solve: rotation := I (PointToPlane.cpp:326-332)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) M4(T, r, cc) = (r == cc) ? 1.f : 0.f;
  }
}

// Eigen Quaternion(Matrix3) (quaternionbase_assign_impl<Other,3,3>), q = {x,y,z,w}
__device__ inline void quat_from_T(const float* T, float* q) {
  float t = M4(T, 0, 0) + M4(T, 1, 1) + M4(T, 2, 2);
  if (t > 0.f) {
    t = sqrtf(t + 1.0f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (M4(T, 2, 1) - M4(T, 1, 2)) * t;
    q[1] = (M4(T, 0, 2) - M4(T, 2, 0)) * t;
    q[2] = (M4(T, 1, 0) - M4(T, 0, 1)) * t;
  } else {
    int i = 0;
    if (M4(T, 1, 1) > M4(T, 0, 0)) i = 1;
    if (M4(T, 2, 2) > M4(T, i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(M4(T, i, i) - M4(T, j, j) - M4(T, k, k) + 1.0f);
    q[i] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (M4(T, k, j) - M4(T, j, k)) * t;
    q[j] = (M4(T, j, i) + M4(T, i, j)) * t;
    q[k] = (M4(T, k, i) + M4(T, i, k)) * t;
  }
}

// Quaternion::angularDistance (Eigen >= 3.3): d = a * conj(b); 2 atan2(|d.vec|, |d.w|)
__device__ inline float quat_angdist(const float* a, const float* b) {
  const float bx = -b[0], by = -b[1], bz = -b[2], bw = b[3];
  const float w = a[3] * bw - a[0] * bx - a[1] * by - a[2] * bz;
  const float x = a[3] * bx + a[0] * bw + a[1] * bz - a[2] * by;
  const float y = a[3] * by + a[1] * bw + a[2] * bx - a[0] * bz;
  const float z = a[3] * bz + a[2] * bw + a[0] * by - a[1] * bx;
  const float vn = sqrtf(x * x + y * y + z * z);
  return 2.f * atan2f_cr(vn, fabsf(w));
}

// push of the checker's history; the distances to the previous entry are formed here, once — the smoothing window of the next
// smooth_length checks re-reads them (same inputs, same operations: the bits the reference recomputes every time)
__device__ inline void diff_push(IcpState* st, const float* T) {
  const int i = st->hist_total, slot = i % kHistRing;
  quat_from_T(T, st->quat_ring[slot]);
  st->trans_ring[slot][0] = M4(T, 0, 3);
  st->trans_ring[slot][1] = M4(T, 1, 3);
  st->trans_ring[slot][2] = M4(T, 2, 3);
  if (i >= 1) {
    const int prev = (i - 1) % kHistRing;
    st->ang_ring[slot] = fabsf(quat_angdist(st->quat_ring[slot], st->quat_ring[prev]));
    const float* ta = st->trans_ring[slot];
    const float* tb = st->trans_ring[prev];
    const float ex = ta[0] - tb[0], ey = ta[1] - tb[1], ez = ta[2] - tb[2];
    float nn = ex * ex;
    nn = nn + ey * ey;
    nn = nn + ez * ez;
    st->tnorm_ring[slot] = fabsf(sqrtf(nn));
  }
  st->hist_total += 1;
}

// transformationCheckers.check(T_iter, iterate) in YAML order; returns status, clears *iterate when a rule fires
__device__ inline int run_checkers(IcpState* st, const ChainParams& cp, const float* T, bool* iterate) {
  int status = 0;
  bool threw = false;
  for (int pass = 0; pass < 2 && !threw && status == 0; ++pass) {
    const bool counter_turn = (pass == 0) == (cp.counter_first != 0);
    if (counter_turn) {
      if (cp.max_iters > 0) {
        st->counter += 1;
        if (st->counter >= cp.max_iters) {  // throws MaxNumIterationsReached, caught at ICP.cpp:441-445
          *iterate = false;
          st->max_iters_reached = 1;
          threw = true;
        }
      }
    } else if (cp.use_differential) {
      diff_push(st, T);
      float cv0 = 0.f, cv1 = 0.f;
      const int sz = st->hist_total, sl = cp.smooth_length;
      if (sz > sl) {
        for (int i = sz - 1; i >= sz - sl && i >= 1; --i) {  // TransformationCheckersImpl.cpp:128-140, newest pair first
          cv0 = cv0 + st->ang_ring[i % kHistRing];
          cv1 = cv1 + st->tnorm_ring[i % kHistRing];
        }
        cv0 = cv0 / (float)sl;
        cv1 = cv1 / (float)sl;
        if (cv0 < cp.min_diff_rot && cv1 < cp.min_diff_trans) *iterate = false;
      }
      if (cv0 != cv0 || cv1 != cv1) status = 7;  // O3S_ERR_NAN
    }
  }
  return status;
}

}  // namespace dev
}  // namespace o3s
