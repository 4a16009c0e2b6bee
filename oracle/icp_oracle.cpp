/*
 * icp_oracle.cpp — CPU restatement of the scan-to-map ICP path.  TEST INFRASTRUCTURE ONLY (see icp_oracle.h).
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).  -ffp-contract=off
 * matters: every per-element fp32 operation below is meant to round exactly once, as x86-64 SSE2 code
 * generated from the reference's Eigen expressions does in a default (no -march=native) build.
 *
 * Abbreviations in citations: LPM = libpointmatcher/pointmatcher, O3S = open3d_slam_rsl/open3d_slam/open3d_slam,
 * CONV = open3d_slam_rsl/open3d_utils/open3d_conversions.
 */
#include "icp_oracle.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <map>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

const float kInf = std::numeric_limits<float>::infinity();

// ------------------------------------------------------------------------------------------------
// small fp32 4x4 helpers, column-major (Eigen default).  M(r,c) = m[c*4+r].
// ------------------------------------------------------------------------------------------------
struct Mat4 {
  float m[16];
  float& operator()(int r, int c) { return m[c * 4 + r]; }
  float operator()(int r, int c) const { return m[c * 4 + r]; }
};

Mat4 identity4() {
  Mat4 I;
  for (int i = 0; i < 16; ++i) I.m[i] = 0.f;
  I(0, 0) = I(1, 1) = I(2, 2) = I(3, 3) = 1.f;
  return I;
}

// dense product, inner index summed sequentially k = 0..3 in fp32 (Eigen's coefficient-based product order)
Mat4 mul4(const Mat4& A, const Mat4& B) {
  Mat4 C;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float s = A(r, 0) * B(0, c);
      s = s + A(r, 1) * B(1, c);
      s = s + A(r, 2) * B(2, c);
      s = s + A(r, 3) * B(3, c);
      C(r, c) = s;
    }
  return C;
}

bool hasNaN4(const Mat4& A) {
  for (int i = 0; i < 16; ++i)
    if (A.m[i] != A.m[i]) return true;
  return false;
}

// Eigen 3x3 determinant (bruteforce_det3_helper), fp32
float det3(const Mat4& T) {
  auto h = [&](int a, int b, int c) { return T(0, a) * (T(1, b) * T(2, c) - T(1, c) * T(2, b)); };
  return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
}

// RigidTransformation::checkParameters  (LPM/TransformationsImpl.cpp:98-113)
bool isRigid(const Mat4& T) {
  const float epsilon = 0.001f;
  return !(std::fabs(1.f - det3(T)) > epsilon);
}

// RigidTransformation::inPlaceCompute  (LPM/TransformationsImpl.cpp:61-96): features = T * features (4x4 * 4xN),
// normals = R * normals.  Per output coefficient the inner sum runs k = 0..3 (resp. 0..2) in fp32.
void applyRigid(const Mat4& T, float* xyzw, float* normals, int64_t N) {
  for (int64_t i = 0; i < N; ++i) {
    float* p = xyzw + 4 * i;
    const float x = p[0], y = p[1], z = p[2], w = p[3];
    for (int r = 0; r < 4; ++r) {
      float s = T(r, 0) * x;
      s = s + T(r, 1) * y;
      s = s + T(r, 2) * z;
      s = s + T(r, 3) * w;
      p[r] = s;
    }
  }
  if (normals) {
    for (int64_t i = 0; i < N; ++i) {
      float* n = normals + 3 * i;
      const float x = n[0], y = n[1], z = n[2];
      for (int r = 0; r < 3; ++r) {
        float s = T(r, 0) * x;
        s = s + T(r, 1) * y;
        s = s + T(r, 2) * z;
        n[r] = s;
      }
    }
  }
}

// libnabo's dist2: sum over dims of diff*diff, accumulated x,y,z in fp32
inline float dist2f(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  float d = dx * dx;
  d = d + dy * dy;
  d = d + dz * dz;
  return d;
}

// ------------------------------------------------------------------------------------------------
// Exact 1-NN kd-tree (restates the libnabo contract visible at LPM/MatchersImpl.cpp:109-132:
// squared fp32 distances, radius-limited (inclusive), InvalidIndex = -1, InvalidValue = +inf).
// Deviation (documented): epsilon = 0 (exact) and ties resolved towards the lowest reference index.
// ------------------------------------------------------------------------------------------------
struct KdNode {
  int32_t left = -1, right = -1;  // children; leaf if left < 0
  int32_t begin = 0, end = 0;     // leaf point range in perm
  int32_t dim = 0;
  float split = 0.f;
};

struct KdTree {
  std::vector<float> pts;  // xyz interleaved, original order
  std::vector<int32_t> perm;
  std::vector<KdNode> nodes;
  float bbmin[3], bbmax[3];
  static constexpr int kLeaf = 12;

  void build(const float* xyz3, int64_t M) {
    pts.assign(xyz3, xyz3 + 3 * M);
    perm.resize(M);
    for (int64_t i = 0; i < M; ++i) perm[i] = (int32_t)i;
    nodes.clear();
    nodes.reserve(2 * (M / kLeaf + 1));
    for (int d = 0; d < 3; ++d) {
      bbmin[d] = kInf;
      bbmax[d] = -kInf;
    }
    for (int64_t i = 0; i < M; ++i)
      for (int d = 0; d < 3; ++d) {
        bbmin[d] = std::min(bbmin[d], pts[3 * i + d]);
        bbmax[d] = std::max(bbmax[d], pts[3 * i + d]);
      }
    if (M > 0) buildRec(0, (int32_t)M);
  }

  int32_t buildRec(int32_t b, int32_t e) {
    const int32_t id = (int32_t)nodes.size();
    nodes.emplace_back();
    if (e - b <= kLeaf) {
      nodes[id].begin = b;
      nodes[id].end = e;
      std::sort(perm.begin() + b, perm.begin() + e);  // ascending index inside a leaf
      return id;
    }
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    for (int32_t i = b; i < e; ++i)
      for (int d = 0; d < 3; ++d) {
        const float v = pts[3 * perm[i] + d];
        lo[d] = std::min(lo[d], v);
        hi[d] = std::max(hi[d], v);
      }
    int dim = 0;
    if (hi[1] - lo[1] > hi[dim] - lo[dim]) dim = 1;
    if (hi[2] - lo[2] > hi[dim] - lo[dim]) dim = 2;
    const int32_t mid = b + (e - b) / 2;
    std::nth_element(perm.begin() + b, perm.begin() + mid, perm.begin() + e, [&](int32_t a, int32_t c) {
      const float va = pts[3 * a + dim], vc = pts[3 * c + dim];
      return va < vc || (va == vc && a < c);
    });
    const float split = pts[3 * perm[mid] + dim];
    const int32_t l = buildRec(b, mid);
    const int32_t r = buildRec(mid, e);
    nodes[id].left = l;
    nodes[id].right = r;
    nodes[id].dim = dim;
    nodes[id].split = split;
    return id;
  }

  // best = (d2, idx) lexicographic minimum among points with d2 <= maxR2.
  void search(int32_t nid, const float q[3], double off[3], double rd, float maxR2, float& bestD, int32_t& bestI) const {
    const KdNode& n = nodes[nid];
    if (n.left < 0) {
      for (int32_t k = n.begin; k < n.end; ++k) {
        const int32_t pi = perm[k];
        const float d = dist2f(q[0], q[1], q[2], pts[3 * pi], pts[3 * pi + 1], pts[3 * pi + 2]);
        if (d <= maxR2 && (d < bestD || (d == bestD && pi < bestI))) {
          bestD = d;
          bestI = pi;
        }
      }
      return;
    }
    const double diff = (double)q[n.dim] - (double)n.split;
    const int32_t nearC = diff < 0 ? n.left : n.right;
    const int32_t farC = diff < 0 ? n.right : n.left;
    search(nearC, q, off, rd, maxR2, bestD, bestI);
    const double oldOff = off[n.dim];
    const double newRd = rd - oldOff * oldOff + diff * diff;
    // conservative prune: fp32 distances carry <= ~3e-7 relative rounding error; keep a 1e-6 margin so a point whose
    // rounded distance ties or beats the incumbent is never skipped.
    const double lim = (double)std::min(bestD, maxR2);
    if (newRd * (1.0 - 1e-6) <= lim) {
      off[n.dim] = diff;
      search(farC, q, off, newRd, maxR2, bestD, bestI);
      off[n.dim] = oldOff;
    }
  }

  void nearest(const float q[3], float maxR2, int32_t& id, float& d2) const {
    id = -1;
    d2 = kInf;
    if (nodes.empty()) return;
    // a query with a NaN or infinite coordinate is at no finite distance from anything: libnabo replaces its incumbent only when
    // d2 < best (strict, best starts at +inf: nabo/kdtree_cpu.cpp), so such a query comes back with InvalidIndex / InvalidValue —
    // which is also what Matches means by "no match" (dist == inf <=> id == -1, LPM/PointMatcher.h:450-451)
    if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]))) return;
    double off[3] = {0, 0, 0};
    double rd = 0;
    for (int d = 0; d < 3; ++d) {
      if (q[d] < bbmin[d]) off[d] = (double)q[d] - bbmin[d];
      else if (q[d] > bbmax[d]) off[d] = (double)q[d] - bbmax[d];
      rd += off[d] * off[d];
    }
    if (rd * (1.0 - 1e-6) > (double)maxR2) return;
    float bestD = kInf;
    int32_t bestI = std::numeric_limits<int32_t>::max();
    search(0, q, off, rd, maxR2, bestD, bestI);
    if (bestI != std::numeric_limits<int32_t>::max()) {
      id = bestI;
      d2 = bestD;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// libnabo's configured search, restated: KDTREE_LINEAR_HEAP = KDTreeUnbalancedPtInLeavesImplicitBoundsStackOpt with a
// brute-force-vector heap (LPM/MatchersImpl.cpp:113 passes searchType 1, dim = features.rows() - 1 = 3, default bucket
// size 8), queried with knn 1, epsilon, ALLOW_SELF_MATCH and maxRadius (LPM/MatchersImpl.cpp:129; icp.yaml:11-15:
// epsilon 0.01).  libnabo is NOT in the reference tree (ANYbotics/libnabo >= 1.0.7, unpinned); this follows its published
// algorithm (nabo/kdtree_cpu.cpp of 1.0.7):
//   build   split the box's LARGEST dimension (first maximum) at the coordinate of the element nth_element puts at
//           position leftCount = count - count / 2; children get the box cut at that value ("implicit bounds"); a node
//           with <= 8 points is a bucket; nodes are stored in pre-order (left child = n + 1)
//   search  descend towards the query's side first; the other side is visited only while
//           rd <= maxRadius2  and  rd * (1 + epsilon)^2 < current best d2,
//           rd = squared distance from the query to the other side's box, updated incrementally per dimension (and
//           starting from 0 at the root even for queries outside the cloud's box); a bucket point replaces the incumbent
//           when  d2 <= maxRadius2 and d2 < best  (strict: among equals the FIRST one visited wins — not the lowest index)
// With epsilon > 0 the returned neighbour may be up to (1 + epsilon) times farther than the true nearest one.  The exact
// tree above is what the GPU path is compared with; this one MEASURES what the configured approximation does to ids, the
// trim limit, the kept set and the pose (tests/test_oracle_known_answers.py, tools/epsilon_effect.py): parity unpinned at
// this boundary, because neither libnabo nor its std::nth_element partition order can be pinned from the tree.
// ------------------------------------------------------------------------------------------------
struct NaboTree {
  static constexpr int kBucket = 8;
  struct Node {
    int32_t dim = 3;         // 0..2: split node; 3: bucket
    int32_t right_or_size = 0;  // right child of a split node / number of points of a bucket
    float cut = 0.f;
    int32_t bucket_begin = 0;
  };
  std::vector<float> pts;  // xyz interleaved, original order
  std::vector<int32_t> bucket_idx;
  std::vector<Node> nodes;

  void build(const float* xyz3, int64_t M) {
    pts.assign(xyz3, xyz3 + 3 * M);
    nodes.clear();
    bucket_idx.clear();
    bucket_idx.reserve(M);
    if (M <= 0) return;
    std::vector<int32_t> order(M);
    for (int64_t i = 0; i < M; ++i) order[i] = (int32_t)i;
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    for (int64_t i = 0; i < M; ++i)
      for (int d = 0; d < 3; ++d) {
        lo[d] = std::min(lo[d], pts[3 * i + d]);
        hi[d] = std::max(hi[d], pts[3 * i + d]);
      }
    buildNodes(order.data(), order.data() + M, lo, hi);
  }

  int32_t buildNodes(int32_t* first, int32_t* last, const float* minV, const float* maxV) {
    const int count = (int)(last - first);
    const int32_t pos = (int32_t)nodes.size();
    if (count <= kBucket) {
      Node n;
      n.dim = 3;
      n.right_or_size = count;
      n.bucket_begin = (int32_t)bucket_idx.size();
      for (int i = 0; i < count; ++i) bucket_idx.push_back(first[i]);
      nodes.push_back(n);
      return pos;
    }
    int cutDim = 0;  // argMax: first strictly larger extent wins
    {
      float maxVal = 0.f;
      for (int d = 0; d < 3; ++d)
        if (maxV[d] - minV[d] > maxVal) {
          maxVal = maxV[d] - minV[d];
          cutDim = d;
        }
    }
    const int rightCount = count / 2, leftCount = count - rightCount;
    std::nth_element(first, first + leftCount, last, [&](int32_t a, int32_t b) { return pts[3 * a + cutDim] < pts[3 * b + cutDim]; });
    const float cutVal = pts[3 * first[leftCount] + cutDim];
    float leftMax[3] = {maxV[0], maxV[1], maxV[2]}, rightMin[3] = {minV[0], minV[1], minV[2]};
    leftMax[cutDim] = cutVal;
    rightMin[cutDim] = cutVal;
    nodes.push_back(Node());
    nodes[pos].cut = cutVal;
    buildNodes(first, first + leftCount, minV, leftMax);  // lands at pos + 1
    const int32_t rightChild = buildNodes(first + leftCount, last, rightMin, maxV);
    nodes[pos].dim = cutDim;
    nodes[pos].right_or_size = rightChild;
    return pos;
  }

  void recurse(const float* q, int32_t n, float rd, float* off, float maxError2, float maxR2, float& bestD, int32_t& bestI) const {
    const Node& node = nodes[n];
    if (node.dim == 3) {
      for (int i = 0; i < node.right_or_size; ++i) {
        const int32_t pi = bucket_idx[node.bucket_begin + i];
        const float d = dist2f(q[0], q[1], q[2], pts[3 * pi], pts[3 * pi + 1], pts[3 * pi + 2]);
        if (d <= maxR2 && d < bestD) {
          bestD = d;
          bestI = pi;
        }
      }
      return;
    }
    const int cd = node.dim;
    const float old_off = off[cd];
    const float new_off = q[cd] - node.cut;
    const int32_t nearC = new_off > 0 ? node.right_or_size : n + 1;
    const int32_t farC = new_off > 0 ? n + 1 : node.right_or_size;
    recurse(q, nearC, rd, off, maxError2, maxR2, bestD, bestI);
    rd += -old_off * old_off + new_off * new_off;
    if (rd <= maxR2 && rd * maxError2 < bestD) {
      off[cd] = new_off;
      recurse(q, farC, rd, off, maxError2, maxR2, bestD, bestI);
      off[cd] = old_off;
    }
  }

  void nearest(const float q[3], float epsilon, float maxR2, int32_t& id, float& d2) const {
    id = -1;
    d2 = kInf;
    if (nodes.empty()) return;
    float off[3] = {0.f, 0.f, 0.f};
    const float maxError2 = (1.f + epsilon) * (1.f + epsilon);
    recurse(q, 0, 0.f, off, maxError2, maxR2, d2, id);
  }
};

// ------------------------------------------------------------------------------------------------
// 6x6 solver: restates solvePossiblyUnderdeterminedLinearSystem (LPM/ErrorMinimizers/PointToPlane.cpp:185-238)
// with Eigen's FullPivHouseholderQR / LLT algorithms written out for n = 6 in fp32.
// ------------------------------------------------------------------------------------------------
constexpr int N6 = 6;
struct Mat6 {
  float a[36];
  float& operator()(int r, int c) { return a[c * 6 + r]; }
  float operator()(int r, int c) const { return a[c * 6 + r]; }
};

struct FullPivQR {
  Mat6 qr;
  float hCoeffs[6];
  int rowsT[6], colsT[6];
  int nonzeroPivots;
  float maxpivot;
  int colPerm[6];  // PermutationMatrix indices

  void compute(const Mat6& A) {
    qr = A;
    const float eps = std::numeric_limits<float>::epsilon();
    const float precision = eps * 6.f;
    nonzeroPivots = 6;
    maxpivot = 0.f;
    float biggest = 0.f;
    for (int k = 0; k < 6; ++k) {
      // biggest |coeff| in the bottom-right corner; Eigen's maxCoeff visitor scans column-major and keeps the first maximum
      int rb = k, cb = k;
      float best = -1.f;
      for (int c = k; c < 6; ++c)
        for (int r = k; r < 6; ++r) {
          const float v = std::fabs(qr(r, c));
          if (v > best) {
            best = v;
            rb = r;
            cb = c;
          }
        }
      const float biggestInCorner = best;
      if (k == 0) biggest = biggestInCorner;
      if (std::fabs(biggestInCorner) <= std::fabs(biggest) * precision) {  // isMuchSmallerThan
        nonzeroPivots = k;
        for (int i = k; i < 6; ++i) {
          rowsT[i] = i;
          colsT[i] = i;
          hCoeffs[i] = 0.f;
        }
        break;
      }
      rowsT[k] = rb;
      colsT[k] = cb;
      if (k != rb)
        for (int c = k; c < 6; ++c) std::swap(qr(k, c), qr(rb, c));
      if (k != cb)
        for (int r = 0; r < 6; ++r) std::swap(qr(r, k), qr(r, cb));
      // makeHouseholderInPlace on qr.col(k).tail(6-k)
      float tailSq = 0.f;
      for (int r = k + 1; r < 6; ++r) tailSq = tailSq + qr(r, k) * qr(r, k);
      const float c0 = qr(k, k);
      float tau, beta;
      if (tailSq <= std::numeric_limits<float>::min()) {
        tau = 0.f;
        beta = c0;
        for (int r = k + 1; r < 6; ++r) qr(r, k) = 0.f;
      } else {
        beta = std::sqrt(c0 * c0 + tailSq);
        if (c0 >= 0.f) beta = -beta;
        for (int r = k + 1; r < 6; ++r) qr(r, k) = qr(r, k) / (c0 - beta);
        tau = (beta - c0) / beta;
      }
      hCoeffs[k] = tau;
      qr(k, k) = beta;
      if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
      // apply H_k on the left of bottomRightCorner(6-k, 6-k-1)
      applyHouseholderLeft(qr, k, k + 1, 6 - k, 6 - k - 1, k, tau);
    }
    for (int i = 0; i < 6; ++i) colPerm[i] = i;
    for (int k = 0; k < 6; ++k) std::swap(colPerm[k], colPerm[colsT[k]]);
  }

  // Block (r0,c0,nr,nc) of M: apply H = I - tau v v^T with v = [1, qr(kcol+1.., kcol)] on the left
  void applyHouseholderLeft(Mat6& M, int r0, int c0, int nr, int nc, int kcol, float tau) const {
    if (nc <= 0) return;
    if (nr == 1) {
      for (int c = 0; c < nc; ++c) M(r0, c0 + c) = M(r0, c0 + c) * (1.f - tau);
      return;
    }
    if (tau == 0.f) return;
    for (int c = 0; c < nc; ++c) {
      float tmp = 0.f;
      for (int r = 1; r < nr; ++r) tmp = tmp + qr(kcol + r, kcol) * M(r0 + r, c0 + c);
      tmp = tmp + M(r0, c0 + c);
      M(r0, c0 + c) = M(r0, c0 + c) - tau * tmp;
      for (int r = 1; r < nr; ++r) M(r0 + r, c0 + c) = M(r0 + r, c0 + c) - tau * qr(kcol + r, kcol) * tmp;
    }
  }

  int rank() const {
    const float thr = std::numeric_limits<float>::epsilon() * 6.f;
    const float pre = std::fabs(maxpivot) * thr;
    int r = 0;
    for (int i = 0; i < nonzeroPivots; ++i) r += (std::fabs(qr(i, i)) > pre) ? 1 : 0;
    return r;
  }

  // matrixQ(): H_0 H_1 ... H_5 with row transpositions (FullPivHouseholderQRMatrixQReturnType::evalTo)
  Mat6 matrixQ() const {
    Mat6 Q;
    for (int i = 0; i < 36; ++i) Q.a[i] = 0.f;
    for (int i = 0; i < 6; ++i) Q(i, i) = 1.f;
    for (int k = 5; k >= 0; --k) {
      applyHouseholderLeft(Q, k, k, 6 - k, 6 - k, k, hCoeffs[k]);
      if (rowsT[k] != k)
        for (int c = 0; c < 6; ++c) std::swap(Q(k, c), Q(rowsT[k], c));
    }
    return Q;
  }
};

// Eigen LLT (unblocked, lower) + solve.  Returns false if a non-positive pivot was met (Eigen then leaves the
// factor partially computed and solve() proceeds regardless; restated as such).
void lltSolve(const float* A, int n, int lda, const float* b, float* x) {
  std::vector<float> L(A, A + (size_t)lda * n);
  auto at = [&](int r, int c) -> float& { return L[(size_t)c * lda + r]; };
  for (int k = 0; k < n; ++k) {
    float xk = at(k, k);
    if (k > 0) {
      float s = 0.f;
      for (int j = 0; j < k; ++j) s = s + at(k, j) * at(k, j);
      xk = xk - s;
    }
    if (xk <= 0.f) break;
    xk = std::sqrt(xk);
    at(k, k) = xk;
    for (int r = k + 1; r < n; ++r) {
      float s = 0.f;
      for (int j = 0; j < k; ++j) s = s + at(r, j) * at(k, j);
      at(r, k) = (at(r, k) - s) / xk;
    }
  }
  // forward L y = b
  std::vector<float> y(n);
  for (int i = 0; i < n; ++i) {
    float s = b[i];
    for (int j = 0; j < i; ++j) s = s - at(i, j) * y[j];
    y[i] = s / at(i, i);
  }
  // backward L^T x = y
  for (int i = n - 1; i >= 0; --i) {
    float s = y[i];
    for (int j = i + 1; j < n; ++j) s = s - at(j, i) * x[j];
    x[i] = s / at(i, i);
  }
}

// fp64 symmetric Jacobi eigen-solver based pseudo-inverse solve: mathematically the JacobiSVD least-squares
// solution used as the last-resort fallback (PointToPlane.cpp:219-231) for a symmetric A = G G^T.
void svdSolveSym6(const Mat6& Af, const float* bf, float* x) {
  double A[6][6], V[6][6], b[6];
  for (int r = 0; r < 6; ++r) {
    b[r] = bf[r];
    for (int c = 0; c < 6; ++c) {
      A[r][c] = 0.5 * ((double)Af(r, c) + (double)Af(c, r));
      V[r][c] = (r == c) ? 1.0 : 0.0;
    }
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = r + 1; c < 6; ++c) off += A[r][c] * A[r][c];
    if (off < 1e-300) break;
    for (int p = 0; p < 6; ++p)
      for (int q = p + 1; q < 6; ++q) {
        if (std::fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 6; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 6; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  double smax = 0;
  for (int i = 0; i < 6; ++i) smax = std::max(smax, std::fabs(A[i][i]));
  const double thr = std::max(smax * 6.0 * std::numeric_limits<double>::epsilon(), std::numeric_limits<double>::min());
  double xd[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) {
    const double lam = A[i][i];
    if (std::fabs(lam) > thr) {
      double vb = 0;
      for (int k = 0; k < 6; ++k) vb += V[k][i] * b[k];
      const double coef = vb / lam;
      for (int k = 0; k < 6; ++k) xd[k] += V[k][i] * coef;
    }
  }
  for (int k = 0; k < 6; ++k) x[k] = (float)xd[k];
}

float norm6(const float* v) {
  float s = 0.f;
  for (int i = 0; i < 6; ++i) s = s + v[i] * v[i];
  return std::sqrt(s);
}

// returns branch: 0 LLT, 1 min-norm QR, 2 SVD fallback
int solve6(const Mat6& A, const float* b, float* x) {
  FullPivQR qr;
  qr.compute(A);
  const int rank = qr.rank();
  if (rank == 6) {  // isInvertible()
    lltSolve(A.a, 6, 6, b, x);
    return 0;
  }
  if (rank == 0) {  // Eigen would build empty blocks; the min-norm solution of a rank-0 system is x = 0
    for (int i = 0; i < 6; ++i) x[i] = 0.f;
    return 1;
  }
  const Mat6 Q = qr.matrixQ();
  // Q1t = Q^T.block(0,0,rank,6);  R1 = (Q1t * A * P).block(0,0,rank,6)
  float Q1t[6][6], QA[6][6], R1[6][6];
  for (int i = 0; i < rank; ++i)
    for (int j = 0; j < 6; ++j) Q1t[i][j] = Q(j, i);
  for (int i = 0; i < rank; ++i)
    for (int j = 0; j < 6; ++j) {
      float s = 0.f;
      for (int k = 0; k < 6; ++k) s = s + Q1t[i][k] * A(k, j);
      QA[i][j] = s;
    }
  for (int i = 0; i < rank; ++i)
    for (int j = 0; j < 6; ++j) R1[i][j] = QA[i][qr.colPerm[j]];  // (M*P)(:,j) = M(:,p(j))
  // y = (R1 R1^T).llt().solve(Q1t b)
  float RRt[36], rhs[6], y[6];
  for (int i = 0; i < rank; ++i) {
    float s = 0.f;
    for (int k = 0; k < 6; ++k) s = s + Q1t[i][k] * b[k];
    rhs[i] = s;
    for (int j = 0; j < rank; ++j) {
      float t = 0.f;
      for (int k = 0; k < 6; ++k) t = t + R1[i][k] * R1[j][k];
      RRt[j * rank + i] = t;
    }
  }
  lltSolve(RRt, rank, rank, rhs, y);
  // x = R1.triangularView<Upper>().transpose() * y ; then x = P * x
  float xt[6];
  for (int j = 0; j < 6; ++j) {
    float s = 0.f;
    for (int i = 0; i < rank; ++i)
      if (j >= i) s = s + R1[i][j] * y[i];
    xt[j] = s;
  }
  for (int i = 0; i < 6; ++i) x[qr.colPerm[i]] = xt[i];  // (P v)(p(i)) = v(i)
  // if (!b.isApprox(A x, 1e-5)) -> fp64 SVD
  float ax[6], diff[6];
  for (int i = 0; i < 6; ++i) {
    float s = 0.f;
    for (int k = 0; k < 6; ++k) s = s + A(i, k) * x[k];
    ax[i] = s;
    diff[i] = b[i] - s;
  }
  const float nb = norm6(b), nax = norm6(ax), nd = norm6(diff);
  const float prec = 1e-5f;
  const bool approx = (nd * nd) <= prec * prec * std::min(nb * nb, nax * nax);
  if (!approx) {
    svdSolveSym6(A, b, x);
    return 2;
  }
  return 1;
}

// ------------------------------------------------------------------------------------------------
// Eigen geometry helpers (fp32)
// ------------------------------------------------------------------------------------------------
struct Quat {
  float x, y, z, w;
};

// QuaternionBase::operator=(Matrix3)  (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>)
Quat quatFromRot(const Mat4& T) {
  Quat q;
  float c[4];
  float t = T(0, 0) + T(1, 1) + T(2, 2);
  if (t > 0.f) {
    t = std::sqrt(t + 1.0f);
    q.w = 0.5f * t;
    t = 0.5f / t;
    q.x = (T(2, 1) - T(1, 2)) * t;
    q.y = (T(0, 2) - T(2, 0)) * t;
    q.z = (T(1, 0) - T(0, 1)) * t;
  } else {
    int i = 0;
    if (T(1, 1) > T(0, 0)) i = 1;
    if (T(2, 2) > T(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(T(i, i) - T(j, j) - T(k, k) + 1.0f);
    c[i] = 0.5f * t;
    t = 0.5f / t;
    c[3] = (T(k, j) - T(j, k)) * t;
    c[j] = (T(j, i) + T(i, j)) * t;
    c[k] = (T(k, i) + T(i, k)) * t;
    q.x = c[0];
    q.y = c[1];
    q.z = c[2];
    q.w = c[3];
  }
  return q;
}

// Quaternion::angularDistance (Eigen >= 3.3): d = a * conj(b); 2*atan2(|d.vec|, |d.w|)
float angularDistance(const Quat& a, const Quat& b) {
  const Quat c{-b.x, -b.y, -b.z, b.w};
  const float w = a.w * c.w - a.x * c.x - a.y * c.y - a.z * c.z;
  const float x = a.w * c.x + a.x * c.w + a.y * c.z - a.z * c.y;
  const float y = a.w * c.y + a.y * c.w + a.z * c.x - a.x * c.z;
  const float z = a.w * c.z + a.z * c.w + a.x * c.y - a.y * c.x;
  const float vn = std::sqrt(x * x + y * y + z * z);
  return 2.f * std::atan2(vn, std::fabs(w));
}

// ------------------------------------------------------------------------------------------------
// the ICP handle
// ------------------------------------------------------------------------------------------------
}  // namespace

struct orc_icp {
  orc_config cfg;
  int threads = 1;
  bool initialized = false;
  int64_t M = 0;
  std::vector<float> refXyzw;     // 4xM, mean-centred
  std::vector<float> refXyz;      // 3xM, mean-centred (kd-tree input)
  std::vector<float> refNormals;  // 3xM or empty
  float mean[3] = {0, 0, 0};
  KdTree tree;
  float nabo_epsilon = -1.f;  // >= 0: findClosests runs libnabo's epsilon-approximate search (orc_set_nabo_epsilon)
  NaboTree nabo;
};

namespace {

void findClosests(const orc_icp* h, const float* q4, int64_t N, int32_t* ids, float* d2, bool brute) {
  if (h->cfg.matcher == 1) {  // MirrorMatcher (LPM/MatchersImpl.cpp:65-85)
    for (int64_t i = 0; i < N; ++i) {
      ids[i] = (int32_t)i;
      d2[i] = 0.f;
    }
    return;
  }
  const float maxR2 = h->cfg.max_dist * h->cfg.max_dist;  // libnabo: maxRadius2 = maxRadius*maxRadius
  if (brute) {
#pragma omp parallel for num_threads(h->threads) schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
      float best = kInf;
      int32_t bi = -1;
      const float* q = q4 + 4 * i;
      const bool finite = std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]);  // see KdTree::nearest
      for (int64_t j = 0; finite && j < h->M; ++j) {
        const float* p = &h->refXyz[3 * j];
        const float d = dist2f(q[0], q[1], q[2], p[0], p[1], p[2]);
        if (d <= maxR2 && (bi < 0 || d < best)) {  // ascending j => first (lowest index) minimum kept
          best = d;
          bi = (int32_t)j;
        }
      }
      ids[i] = bi;
      d2[i] = bi < 0 ? kInf : best;
    }
    return;
  }
  if (h->nabo_epsilon >= 0.f) {
#pragma omp parallel for num_threads(h->threads) schedule(dynamic, 256)
    for (int64_t i = 0; i < N; ++i) {
      const float q[3] = {q4[4 * i], q4[4 * i + 1], q4[4 * i + 2]};
      h->nabo.nearest(q, h->nabo_epsilon, maxR2, ids[i], d2[i]);
    }
    return;
  }
#pragma omp parallel for num_threads(h->threads) schedule(dynamic, 256)
  for (int64_t i = 0; i < N; ++i) {
    const float q[3] = {q4[4 * i], q4[4 * i + 1], q4[4 * i + 2]};
    h->tree.nearest(q, maxR2, ids[i], d2[i]);
  }
}

// Matches::getDistsQuantile (LPM/Matches.cpp:61-87)
int distsQuantile(const float* d2, int64_t n, float quantile, float* out) {
  std::vector<float> values;
  values.reserve(n);
  for (int64_t i = 0; i < n; ++i)
    if (d2[i] != kInf) values.push_back(d2[i]);
  if (values.empty()) return ORC_ERR_NO_MATCHES;
  if (quantile < 0.0f || quantile > 1.0f) return ORC_ERR_NO_MATCHES;
  if (quantile == 1.0f) {
    *out = *std::max_element(values.begin(), values.end());
    return ORC_OK;
  }
  // `values.size() * quantile`: size_t -> float, fp32 multiply, truncated when used as an index
  const float fidx = (float)values.size() * quantile;
  size_t idx = (size_t)fidx;
  if (idx >= values.size()) idx = values.size() - 1;  // UB in the reference; clamped here
  std::nth_element(values.begin(), values.begin() + idx, values.end());
  *out = values[idx];
  return ORC_OK;
}

// OutlierFilters::compute (LPM/OutlierFilter.cpp:64-103) for the configured chain
int outlierWeights(const orc_icp* h, const float* readNormals, const int32_t* ids, const float* d2, int64_t N,
                   float* w, float* limitOut) {
  const orc_config& c = h->cfg;
  const bool hasTrim = c.trim_ratio >= 0.f, hasNormal = c.max_normal_angle >= 0.f, hasMaxDist = c.max_dist_outlier >= 0.f;
  if (limitOut) *limitOut = std::numeric_limits<float>::quiet_NaN();
  if (!hasTrim && !hasNormal && !hasMaxDist) {
    for (int64_t i = 0; i < N; ++i) w[i] = (d2[i] == kInf) ? 0.f : 1.f;
    return ORC_OK;
  }
  for (int64_t i = 0; i < N; ++i) w[i] = 1.f;
  if (hasMaxDist) {  // MaxDistOutlierFilter (LPM/OutlierFiltersImpl.cpp:67-81): maxDist stored squared via pow(.,2)
    const float lim = (float)std::pow((double)c.max_dist_outlier, 2);  // pow(T, int) promotes to double, stored as T
    for (int64_t i = 0; i < N; ++i) w[i] = w[i] * ((d2[i] <= lim) ? 1.f : 0.f);
  }
  if (hasTrim) {  // TrimmedDistOutlierFilter (LPM/OutlierFiltersImpl.cpp:140-147)
    float limit;
    const int st = distsQuantile(d2, N, c.trim_ratio, &limit);
    if (st != ORC_OK) return st;
    if (limitOut) *limitOut = limit;
    for (int64_t i = 0; i < N; ++i) w[i] = w[i] * ((d2[i] <= limit) ? 1.f : 0.f);
  }
  if (hasNormal) {  // SurfaceNormalOutlierFilter (LPM/OutlierFiltersImpl.cpp:227-281)
    if (readNormals && !h->refNormals.empty()) {
      const float eps = std::cos(c.max_normal_angle);  // float overload: cos evaluated in fp32
      for (int64_t i = 0; i < N; ++i) {
        float wi;
        if (ids[i] == -1) {
          wi = 0.f;
        } else {
          const float* a = readNormals + 3 * i;
          const float* b = &h->refNormals[3 * (int64_t)ids[i]];
          float v = a[0] * b[0];
          v = v + a[1] * b[1];
          v = v + a[2] * b[2];
          wi = (v < eps) ? 0.f : 1.f;
        }
        w[i] = w[i] * wi;
      }
    }
  }
  return ORC_OK;
}

struct StepOut {
  Mat4 T;
  Mat6 A;
  float b[6], x[6];
  int64_t kept = 0;
  float pointUsedRatio = 0, weightedRatio = 0;
  int branch = 0;
};

// ErrorMinimizer::compute -> ErrorElements (LPM/ErrorMinimizer.cpp:59-193) -> PointToPlaneErrorMinimizer::compute
// (LPM/ErrorMinimizers/PointToPlane.cpp:241-368) for dim == 4, force2D = force4DOF = false.
int p2planeStep(const orc_icp* h, const float* read4, const int32_t* ids, const float* d2, const float* w, int64_t N,
                StepOut& out) {
  std::vector<int64_t> kept;
  kept.reserve(N);
  float wsum = 0.f;
  int64_t nonzero = 0;
  for (int64_t i = 0; i < N; ++i) nonzero += (w[i] != 0.0f) ? 1 : 0;
  if (nonzero == 0) return ORC_ERR_NO_POINTS;
  for (int64_t i = 0; i < N; ++i) {
    if (d2[i] == kInf) continue;
    if (w[i] != 0.0f) {
      kept.push_back(i);
      wsum = wsum + w[i];
    }
  }
  const int64_t K = (int64_t)kept.size();
  out.kept = K;
  out.pointUsedRatio = (float)K / (float)N;
  out.weightedRatio = wsum / (float)N;
  if (K == 0) return ORC_ERR_NO_POINTS;  // the reference would index empty matrices; treated as "no point to minimize"

  // means of the kept reading / associated reference (rowwise().mean(); fp64 accumulate, rounded once)
  double sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
  for (int64_t j = 0; j < K; ++j) {
    const float* p = read4 + 4 * kept[j];
    const float* q = &h->refXyzw[4 * (int64_t)ids[kept[j]]];
    for (int d = 0; d < 3; ++d) {
      sp[d] += p[d];
      sq[d] += q[d];
    }
  }
  float mp[3], mq[3];
  for (int d = 0; d < 3; ++d) {
    mp[d] = (float)(sp[d] / (double)K);
    mq[d] = (float)(sq[d] / (double)K);
  }

  // G = [ (p - mp) x n ; n ],  h = n . ((p - mp) - (q - mq)),  A = G G^T, b = -(G h^T)
  double Ad[6][6] = {{0}}, bd[6] = {0};
  for (int64_t j = 0; j < K; ++j) {
    const float* p = read4 + 4 * kept[j];
    const int64_t id = ids[kept[j]];
    const float* q = &h->refXyzw[4 * id];
    const float* n = &h->refNormals[3 * id];
    const float px = p[0] - mp[0], py = p[1] - mp[1], pz = p[2] - mp[2];
    const float qx = q[0] - mq[0], qy = q[1] - mq[1], qz = q[2] - mq[2];
    float g[6];
    g[0] = py * n[2] - pz * n[1];  // crossProduct (LPM/ErrorMinimizer.cpp:304-306)
    g[1] = pz * n[0] - px * n[2];
    g[2] = px * n[1] - py * n[0];
    g[3] = n[0];
    g[4] = n[1];
    g[5] = n[2];
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    float hh = 0.f;
    hh = hh + dx * n[0];
    hh = hh + dy * n[1];
    hh = hh + dz * n[2];
    for (int a = 0; a < 6; ++a) {
      for (int c = a; c < 6; ++c) Ad[a][c] += (double)(g[a] * g[c]);
      bd[a] += (double)(g[a] * hh);
    }
  }
  for (int a = 0; a < 6; ++a) {
    for (int c = a; c < 6; ++c) {
      out.A(a, c) = (float)Ad[a][c];
      out.A(c, a) = (float)Ad[a][c];
    }
    out.b[a] = -(float)bd[a];
  }
  out.branch = solve6(out.A, out.b, out.x);
  const float* x = out.x;

  // AngleAxis(|x[0:3]|, x[0:3].stableNormalized()).toRotationMatrix()
  float n2 = x[0] * x[0];
  n2 = n2 + x[1] * x[1];
  n2 = n2 + x[2] * x[2];
  const float angle = std::sqrt(n2);
  float axis[3] = {x[0], x[1], x[2]};
  {
    const float wmax = std::max(std::fabs(x[0]), std::max(std::fabs(x[1]), std::fabs(x[2])));
    const float a0 = x[0] / wmax, a1 = x[1] / wmax, a2 = x[2] / wmax;
    float z = a0 * a0;
    z = z + a1 * a1;
    z = z + a2 * a2;
    if (z > 0.f) {
      const float den = std::sqrt(z) * wmax;
      axis[0] = x[0] / den;
      axis[1] = x[1] / den;
      axis[2] = x[2] / den;
    }
  }
  const float sA = std::sin(angle), cA = std::cos(angle);
  const float sx = sA * axis[0], sy = sA * axis[1], sz = sA * axis[2];
  const float c1x = (1.f - cA) * axis[0], c1y = (1.f - cA) * axis[1], c1z = (1.f - cA) * axis[2];
  float R[3][3];
  float tmp = c1x * axis[1];
  R[0][1] = tmp - sz;
  R[1][0] = tmp + sz;
  tmp = c1x * axis[2];
  R[0][2] = tmp + sy;
  R[2][0] = tmp - sy;
  tmp = c1y * axis[2];
  R[1][2] = tmp - sx;
  R[2][1] = tmp + sx;
  R[0][0] = c1x * axis[0] + cA;
  R[1][1] = c1y * axis[1] + cA;
  R[2][2] = c1z * axis[2] + cA;

  // T = Trans(mq) * [R, t] * Trans(mp)^-1 :  linear = R, translation = R*(-mp) + (t + mq)
  Mat4 T = identity4();
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T(r, c) = R[r][c];
    float s = R[r][0] * (-mp[0]);
    s = s + R[r][1] * (-mp[1]);
    s = s + R[r][2] * (-mp[2]);
    T(r, 3) = s + (x[3 + r] + mq[r]);
  }
  if (hasNaN4(T)) {  // PointToPlane.cpp:326-332
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) T(r, c) = (r == c) ? 1.f : 0.f;
  }
  out.T = T;
  return ORC_OK;
}

struct DiffChecker {
  std::vector<Quat> rot;
  std::vector<float> tr;  // 3 per entry
  void init(const Mat4& T) {
    rot.clear();
    tr.clear();
    push(T);
  }
  void push(const Mat4& T) {
    rot.push_back(quatFromRot(T));
    tr.push_back(T(0, 3));
    tr.push_back(T(1, 3));
    tr.push_back(T(2, 3));
  }
};

double nowMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// =================================================================================================
// C API
// =================================================================================================
extern "C" {

orc_icp* orc_create(const orc_config* cfg) {
  if (!cfg) return nullptr;
  orc_icp* h = new orc_icp();
  h->cfg = *cfg;
  return h;
}

void orc_destroy(orc_icp* h) { delete h; }

void orc_set_threads(orc_icp* h, int n) { h->threads = n < 1 ? 1 : n; }

void orc_set_nabo_epsilon(orc_icp* h, float epsilon) {
  h->nabo_epsilon = epsilon;
  if (epsilon >= 0.f && h->initialized && h->cfg.matcher == 0) h->nabo.build(h->refXyz.data(), h->M);
}

// ICP::initReference (LPM/ICP.cpp:292-328); center = false: Matcher::init alone (KDTreeMatcher::init,
// LPM/MatchersImpl.cpp:108-114 — the search structure is built over the features exactly as they are handed in)
static int initReferenceImpl(orc_icp* h, const float* xyzw, const float* normals, int64_t M, bool center) {
  if (M <= 0) {
    h->initialized = false;
    return ORC_ERR_EMPTY_REFERENCE;
  }
  h->M = M;
  h->refXyzw.assign(xyzw, xyzw + 4 * M);
  if (normals) h->refNormals.assign(normals, normals + 3 * M);
  else h->refNormals.clear();
  double s[3] = {0, 0, 0};
  for (int64_t i = 0; i < M; ++i)
    for (int d = 0; d < 3; ++d) s[d] += xyzw[4 * i + d];
  for (int d = 0; d < 3; ++d) h->mean[d] = center ? (float)(s[d] / (double)M) : 0.f;
  h->refXyz.resize(3 * M);
  for (int64_t i = 0; i < M; ++i)
    for (int d = 0; d < 3; ++d) {
      const float v = xyzw[4 * i + d] - h->mean[d];
      h->refXyzw[4 * i + d] = v;
      h->refXyz[3 * i + d] = v;
    }
  if (h->cfg.matcher == 0) h->tree.build(h->refXyz.data(), M);
  if (h->cfg.matcher == 0 && h->nabo_epsilon >= 0.f) h->nabo.build(h->refXyz.data(), M);
  h->initialized = true;
  return ORC_OK;
}

int orc_init_reference(orc_icp* h, const float* xyzw, const float* normals, int64_t M) { return initReferenceImpl(h, xyzw, normals, M, true); }
int orc_matcher_init(orc_icp* h, const float* xyzw, const float* normals, int64_t M) { return initReferenceImpl(h, xyzw, normals, M, false); }

void orc_reference_mean(const orc_icp* h, float* mean3) {
  for (int d = 0; d < 3; ++d) mean3[d] = h->mean[d];
}

// ICP::compute -> computeWithTransformedReference (LPM/ICP.cpp:258-290, 332-468)
int orc_compute(orc_icp* h, const float* xyzw, const float* normals, int64_t N, const float* T_init, float* T_out,
                orc_stats* stats, float* trace_T, float* trace_limit, int64_t* trace_kept, int32_t trace_cap) {
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (!h->initialized) return ORC_ERR_NOT_INITIALIZED;
  if (N <= 0) return ORC_ERR_EMPTY_READING;
  const orc_config& c = h->cfg;
  if (c.max_iters <= 0 && !c.use_differential) return ORC_ERR_BAD_CONFIG;
  if (c.matcher == 1 && N > h->M) return ORC_ERR_BAD_SHAPE;
  if (h->refNormals.empty()) return ORC_ERR_BAD_SHAPE;  // point-to-plane needs reference normals (PointToPlane.cpp:117-120)

  Mat4 Tinit;
  std::memcpy(Tinit.m, T_init, sizeof(Tinit.m));
  Mat4 Tc = identity4(), TcInv = identity4();
  for (int d = 0; d < 3; ++d) {
    Tc(d, 3) = h->mean[d];
    TcInv(d, 3) = -h->mean[d];
  }
  const Mat4 T0 = mul4(TcInv, Tinit);  // T_refMean_readMean (reading mean forced to 0, ICP.cpp:364-374)

  std::vector<float> reading(xyzw, xyzw + 4 * N);
  std::vector<float> readNormals;
  if (normals) readNormals.assign(normals, normals + 3 * N);
  if (!isRigid(T0)) return ORC_ERR_NOT_RIGID;
  applyRigid(T0, reading.data(), normals ? readNormals.data() : nullptr, N);

  Mat4 Titer = identity4();
  bool iterate = true;
  bool maxIterReached = false;
  int counter = 0;
  DiffChecker diff;
  if (c.use_differential) diff.init(Titer);

  std::vector<float> step(4 * N), stepNormals(normals ? 3 * N : 0), d2(N), w(N);
  std::vector<int32_t> ids(N);
  int iterationCount = 0;
  StepOut so;
  float lastLimit = std::numeric_limits<float>::quiet_NaN();
  int64_t lastMatched = 0;
  double tMatch = 0, tOut = 0, tMin = 0;
  const double tStart = nowMs();
  int status = ORC_OK;

  while (iterate) {
    step = reading;
    if (normals) stepNormals = readNormals;
    if (!isRigid(Titer)) {
      status = ORC_ERR_NOT_RIGID;
      break;
    }
    applyRigid(Titer, step.data(), normals ? stepNormals.data() : nullptr, N);

    double t0 = nowMs();
    findClosests(h, step.data(), N, ids.data(), d2.data(), false);
    double t1 = nowMs();
    tMatch += t1 - t0;

    status = outlierWeights(h, normals ? stepNormals.data() : nullptr, ids.data(), d2.data(), N, w.data(), &lastLimit);
    double t2 = nowMs();
    tOut += t2 - t1;
    if (status != ORC_OK) break;
    lastMatched = 0;
    for (int64_t i = 0; i < N; ++i) lastMatched += (d2[i] != kInf) ? 1 : 0;

    status = p2planeStep(h, step.data(), ids.data(), d2.data(), w.data(), N, so);
    if (status != ORC_OK) break;
    Titer = mul4(so.T, Titer);
    tMin += nowMs() - t2;

    if (iterationCount < trace_cap) {
      if (trace_T) std::memcpy(trace_T + 16 * iterationCount, Titer.m, sizeof(Titer.m));
      if (trace_limit) trace_limit[iterationCount] = lastLimit;
      if (trace_kept) trace_kept[iterationCount] = so.kept;
    }

    // transformationCheckers.check (YAML order), MaxNumIterationsReached caught at ICP.cpp:441-445
    auto checkCounter = [&]() -> bool {  // returns true if it "throws"
      if (c.max_iters <= 0) return false;
      ++counter;
      if (counter >= c.max_iters) {
        iterate = false;
        maxIterReached = true;
        return true;
      }
      return false;
    };
    auto checkDiff = [&]() -> int {
      if (!c.use_differential) return ORC_OK;
      diff.push(Titer);
      float cv0 = 0.f, cv1 = 0.f;
      const size_t sz = diff.rot.size();
      const size_t sl = (size_t)std::max(c.smooth_length, 0);
      if (sz > sl) {
        for (size_t i = sz - 1; i >= sz - sl && sl > 0; --i) {
          cv0 = cv0 + std::fabs(angularDistance(diff.rot[i], diff.rot[i - 1]));
          const float ex = diff.tr[3 * i] - diff.tr[3 * (i - 1)];
          const float ey = diff.tr[3 * i + 1] - diff.tr[3 * (i - 1) + 1];
          const float ez = diff.tr[3 * i + 2] - diff.tr[3 * (i - 1) + 2];
          float nn = ex * ex;
          nn = nn + ey * ey;
          nn = nn + ez * ez;
          cv1 = cv1 + std::fabs(std::sqrt(nn));
          if (i == 0) break;
        }
        cv0 = cv0 / (float)sl;
        cv1 = cv1 / (float)sl;
        if (cv0 < c.min_diff_rot && cv1 < c.min_diff_trans) iterate = false;
      }
      if (cv0 != cv0 || cv1 != cv1) return ORC_ERR_NAN;
      return ORC_OK;
    };
    if (c.counter_first) {
      if (!checkCounter()) status = checkDiff();
    } else {
      status = checkDiff();
      if (status == ORC_OK) checkCounter();
    }
    ++iterationCount;
    if (status != ORC_OK) break;
  }

  if (stats) {
    stats->iterations = iterationCount;
    stats->max_iters_reached = maxIterReached ? 1 : 0;
    stats->kept_pairs = so.kept;
    stats->matched_pairs = lastMatched;
    stats->point_used_ratio = so.pointUsedRatio;
    stats->weighted_point_used_ratio = so.weightedRatio;
    stats->last_trim_limit = lastLimit;
    stats->match_ms = tMatch;
    stats->outlier_ms = tOut;
    stats->minimize_ms = tMin;
    stats->total_ms = nowMs() - tStart;
  }
  if (status != ORC_OK) return status;
  const Mat4 out = mul4(Tc, mul4(Titer, T0));  // ICP.cpp:462-465
  std::memcpy(T_out, out.m, sizeof(out.m));
  return ORC_OK;
}

int orc_find_closests(orc_icp* h, const float* q, int64_t N, int32_t* ids, float* d2, int brute) {
  if (!h->initialized) return ORC_ERR_NOT_INITIALIZED;
  findClosests(h, q, N, ids, d2, brute != 0);
  return ORC_OK;
}

int orc_dists_quantile(const float* d2, int64_t n, float ratio, float* out) { return distsQuantile(d2, n, ratio, out); }

int orc_outlier_weights(orc_icp* h, const float* readNormals, const int32_t* ids, const float* d2, int64_t N,
                        float* weights) {
  return outlierWeights(h, readNormals, ids, d2, N, weights, nullptr);
}

int orc_p2plane_step(orc_icp* h, const float* read4, const int32_t* ids, const float* d2, const float* w, int64_t N,
                     float* T_out, float* A_out, float* b_out, float* x_out) {
  if (!h->initialized) return ORC_ERR_NOT_INITIALIZED;
  if (h->refNormals.empty()) return ORC_ERR_BAD_SHAPE;
  StepOut so;
  const int st = p2planeStep(h, read4, ids, d2, w, N, so);
  if (st != ORC_OK) return st;
  std::memcpy(T_out, so.T.m, sizeof(so.T.m));
  if (A_out) std::memcpy(A_out, so.A.a, sizeof(so.A.a));
  if (b_out) std::memcpy(b_out, so.b, sizeof(so.b));
  if (x_out) std::memcpy(x_out, so.x, sizeof(so.x));
  return ORC_OK;
}

void orc_solve6(const float* A, const float* b, float* x, int32_t* branch_out) {
  Mat6 Am;
  std::memcpy(Am.a, A, sizeof(Am.a));
  const int br = solve6(Am, b, x);
  if (branch_out) *branch_out = br;
}

int orc_rigid_transform(const float* T, float* xyzw, float* normals, int64_t N) {
  Mat4 Tm;
  std::memcpy(Tm.m, T, sizeof(Tm.m));
  if (!isRigid(Tm)) return ORC_ERR_NOT_RIGID;
  applyRigid(Tm, xyzw, normals, N);
  return ORC_OK;
}

// ---- open3d_slam side ---------------------------------------------------------------------------

// getVoxelIdx(p, InverseVoxelSize)  (O3S/include/open3d_slam/VoxelHashMap.hpp:37-51)
void orc_voxel_idx(const double* pts, int64_t N, double voxel_size, int32_t* idx) {
  const double inv = 1.0 / voxel_size;
  for (int64_t i = 0; i < 3 * N; ++i) idx[i] = (int32_t)std::floor(pts[i] * inv);
}

// the dividing overloads (VoxelHashMap.hpp:53-61)
void orc_voxel_idx_div(const double* pts, int64_t N, double voxel_size, const double* min_bound, int32_t* idx) {
  for (int64_t i = 0; i < N; ++i)
    for (int d = 0; d < 3; ++d) {
      const double p = min_bound ? (pts[3 * i + d] - min_bound[d]) : pts[3 * i + d];
      idx[3 * i + d] = (int32_t)std::floor(p / voxel_size);
    }
}

// EigenVec3iHash (VoxelHashMap.hpp:25-35): int -> size_t conversions wrap mod 2^64, result truncated to 32 bits
void orc_voxel_hash(const int32_t* idx, int64_t N, uint64_t* hash) {
  const size_t sl = 17191, sl2 = sl * sl;
  for (int64_t i = 0; i < N; ++i) {
    const size_t v = (size_t)(int64_t)idx[3 * i] + (size_t)(int64_t)idx[3 * i + 1] * sl + (size_t)(int64_t)idx[3 * i + 2] * sl2;
    hash[i] = (uint64_t) static_cast<unsigned int>(v);
  }
}

static bool withinImpl(const orc_cropper* c, const double* p) {
  const double dx = p[0] - c->centre[0], dy = p[1] - c->centre[1], dz = p[2] - c->centre[2];
  switch (c->kind) {
    case 1: return std::sqrt(dx * dx + dy * dy + dz * dz) <= c->p0;                       // MaxRadius   (croppers.cpp:136-138)
    case 2: return std::sqrt(dx * dx + dy * dy + dz * dz) >= c->p0;                       // MinRadius   (:150-152)
    case 3: {                                                                             // MinMaxRadius (:121-124)
      const double d = std::sqrt(dx * dx + dy * dy + dz * dz);
      return d <= c->p1 && d >= c->p0;
    }
    case 4: return p[2] >= c->p1 && p[2] <= c->p2 && std::sqrt(dx * dx + dy * dy) <= c->p0;  // Cylinder (:164-166)
    default: return true;                                                                 // base (:53-55)
  }
}

// CroppingVolume::isWithinVolume (O3S/src/croppers.cpp:57-59)
void orc_crop_mask(const orc_cropper* c, const double* pts, int64_t N, uint8_t* mask) {
  for (int64_t i = 0; i < N; ++i) {
    const bool in = withinImpl(c, pts + 3 * i);
    mask[i] = (c->invert ? !in : in) ? 1 : 0;
  }
}

namespace {
struct KeyHash {
  size_t operator()(const std::array<int32_t, 3>& k) const {
    const size_t sl = 17191, sl2 = sl * sl;
    return static_cast<unsigned int>((size_t)(int64_t)k[0] + (size_t)(int64_t)k[1] * sl + (size_t)(int64_t)k[2] * sl2);
  }
};
struct Acc {
  int num = 0;
  double p[3] = {0, 0, 0}, n[3] = {0, 0, 0};
};
}  // namespace

// voxelizeWithinCroppingVolume (O3S/src/helpers.cpp:117-192); colours/covariances not carried (not on the ICP path).
int64_t orc_voxelize_within_crop(const orc_cropper* c, double voxel_size, const double* pts, const double* normals,
                                 int64_t N, double* out_pts, double* out_normals, int32_t* out_voxel_idx) {
  int64_t n_out = 0;
  if (voxel_size <= 0.0) {
    for (int64_t i = 0; i < N; ++i) {
      for (int d = 0; d < 3; ++d) {
        out_pts[3 * n_out + d] = pts[3 * i + d];
        if (normals) out_normals[3 * n_out + d] = normals[3 * i + d];
        if (out_voxel_idx) out_voxel_idx[3 * n_out + d] = INT32_MIN;
      }
      ++n_out;
    }
    return n_out;
  }
  const double inv = 1.0 / voxel_size;
  std::unordered_map<std::array<int32_t, 3>, Acc, KeyHash> vox;
  std::vector<std::array<int32_t, 3>> order;  // first-touch order (the reference's hash-map order is unspecified)
  vox.reserve(N);
  for (int64_t i = 0; i < N; ++i) {
    const double* p = pts + 3 * i;
    const bool in0 = withinImpl(c, p);
    const bool in = c->invert ? !in0 : in0;
    if (in) {
      const std::array<int32_t, 3> key = {(int32_t)std::floor(p[0] * inv), (int32_t)std::floor(p[1] * inv),
                                          (int32_t)std::floor(p[2] * inv)};
      auto it = vox.find(key);
      if (it == vox.end()) {
        it = vox.emplace(key, Acc()).first;
        order.push_back(key);
      }
      Acc& a = it->second;
      for (int d = 0; d < 3; ++d) a.p[d] += p[d];
      if (normals) {
        const double* nn = normals + 3 * i;
        if (!std::isnan(nn[0]) && !std::isnan(nn[1]) && !std::isnan(nn[2]))
          for (int d = 0; d < 3; ++d) a.n[d] += nn[d];
      }
      a.num++;
    } else {
      for (int d = 0; d < 3; ++d) {
        out_pts[3 * n_out + d] = p[d];
        if (normals) out_normals[3 * n_out + d] = normals[3 * i + d];
        if (out_voxel_idx) out_voxel_idx[3 * n_out + d] = INT32_MIN;
      }
      ++n_out;
    }
  }
  for (const auto& key : order) {
    const Acc& a = vox[key];
    double avgN[3];
    for (int d = 0; d < 3; ++d) {
      out_pts[3 * n_out + d] = a.p[d] / double(a.num);
      avgN[d] = a.n[d] / double(a.num);
      if (out_voxel_idx) out_voxel_idx[3 * n_out + d] = key[d];
    }
    if (normals) {  // GetAverageNormal().normalized(): Eigen's normalized() divides by the norm when it is > 0
      const double nrm2 = avgN[0] * avgN[0] + avgN[1] * avgN[1] + avgN[2] * avgN[2];
      const double nrm = std::sqrt(nrm2);
      for (int d = 0; d < 3; ++d) out_normals[3 * n_out + d] = nrm2 > 0.0 ? avgN[d] / nrm : avgN[d];
    }
    ++n_out;
  }
  return n_out;
}

// computeIndicesOfOverlappingPoints (O3S/src/helpers.cpp:319-345): VoxelMap(voxelSize) with a "target" layer (the target
// cloud) and a "source" layer (the source moved by Open3D's PointCloud::Transform: p' = (T [p 1]).head<3>() / w); a voxel
// contributes the indices of both layers when each holds >= minNumPointsPerVoxel points.  The reference walks its
// unordered_map (order unspecified): the indices are returned in ascending order here.  Keys: getVoxelIdx(p, 1 / voxelSize)
// (VoxelMap::getKey, O3S/include/open3d_slam/VoxelHashMap.hpp:43-51,127).
void orc_overlap_indices(const double* source, int64_t Ns, const double* target, int64_t Nt, const double* T, double voxel_size,
                         int64_t min_pts, int64_t* idx_source, int64_t* n_source, int64_t* idx_target, int64_t* n_target) {
  auto M = [&](int r, int c) { return T[c * 4 + r]; };
  const double inv = 1.0 / voxel_size;
  struct Key {
    int32_t x, y, z;
    bool operator<(const Key& o) const { return z != o.z ? z < o.z : (y != o.y ? y < o.y : x < o.x); }
  };
  std::map<Key, std::pair<int64_t, int64_t>> counts;  // voxel -> (source points, target points)
  std::vector<Key> ks((size_t)Ns), kt((size_t)Nt);
  for (int64_t i = 0; i < Nt; ++i) {
    kt[i] = Key{(int32_t)std::floor(target[3 * i] * inv), (int32_t)std::floor(target[3 * i + 1] * inv), (int32_t)std::floor(target[3 * i + 2] * inv)};
    counts[kt[i]].second += 1;
  }
  for (int64_t i = 0; i < Ns; ++i) {
    const double x = source[3 * i], y = source[3 * i + 1], z = source[3 * i + 2];
    double v[4];
    for (int r = 0; r < 4; ++r) {
      double s = M(r, 0) * x;
      s = s + M(r, 1) * y;
      s = s + M(r, 2) * z;
      s = s + M(r, 3) * 1.0;
      v[r] = s;
    }
    ks[i] = Key{(int32_t)std::floor((v[0] / v[3]) * inv), (int32_t)std::floor((v[1] / v[3]) * inv), (int32_t)std::floor((v[2] / v[3]) * inv)};
    counts[ks[i]].first += 1;
  }
  int64_t a = 0, b = 0;
  for (int64_t i = 0; i < Ns; ++i) {
    const auto& c = counts[ks[i]];
    if (c.first >= min_pts && c.second >= min_pts) idx_source[a++] = i;
  }
  for (int64_t i = 0; i < Nt; ++i) {
    const auto& c = counts[kt[i]];
    if (c.first >= min_pts && c.second >= min_pts) idx_target[b++] = i;
  }
  *n_source = a;
  *n_target = b;
}

// o3d_slam::transform (O3S/src/helpers.cpp:283-318).  Eigen's fixed 4x4 * 4x1 product is restated as the k = 0..3
// accumulation ((T_r0 x + T_r1 y) + T_r2 z) + T_r3 w in fp64 without contraction (Eigen is not in the tree: unpinned).
int64_t orc_transform_cloud(const double* T, const double* pts, const double* normals, int64_t N, double* out_pts,
                            double* out_normals) {
  auto M = [&](int r, int c) { return T[c * 4 + r]; };
  double dev = 0.0;  // (T - Identity).array().abs().maxCoeff()
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) dev = std::max(dev, std::fabs(M(r, c) - (r == c ? 1.0 : 0.0)));
  int64_t n = 0;
  if (dev < 1e-4) {  // "*out = cloud" — and the loop below still appends (helpers.cpp:285-288, 300-304)
    for (int64_t i = 0; i < N; ++i, ++n)
      for (int d = 0; d < 3; ++d) {
        out_pts[3 * n + d] = pts[3 * i + d];
        if (normals) out_normals[3 * n + d] = normals[3 * i + d];
      }
  }
  for (int64_t i = 0; i < N; ++i, ++n) {
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    double v[4];
    for (int r = 0; r < 4; ++r) {
      double s = M(r, 0) * x;
      s = s + M(r, 1) * y;
      s = s + M(r, 2) * z;
      s = s + M(r, 3) * 1.0;
      v[r] = s;
    }
    for (int d = 0; d < 3; ++d) out_pts[3 * n + d] = v[d] / v[3];
    if (normals) {
      const double a = normals[3 * i], b = normals[3 * i + 1], c = normals[3 * i + 2];
      for (int r = 0; r < 3; ++r) {
        double s = M(r, 0) * a;
        s = s + M(r, 1) * b;
        s = s + M(r, 2) * c;
        s = s + M(r, 3) * 0.0;
        out_normals[3 * n + r] = s;
      }
    }
  }
  return n;
}

// ----------------------------------------------------------------------------------------------------------------
// Open3D v0.15.1 EstimateNormals / NormalizeNormals / OrientNormalsTowardsCameraLocation (external to the reference
// tree; restated from the published source).  fp64, reference operation order, no contraction.
// ----------------------------------------------------------------------------------------------------------------
namespace {
struct V3 {
  double x, y, z;
};
inline V3 cross3(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot3(const V3& a, const V3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// utility/Eigen.cpp ComputeEigenvector0 (Geometric Tools "A Robust Eigensolver for 3x3 Symmetric Matrices")
V3 eigenvector0(const double A[3][3], double eval0) {
  const V3 row0{A[0][0] - eval0, A[0][1], A[0][2]};
  const V3 row1{A[0][1], A[1][1] - eval0, A[1][2]};
  const V3 row2{A[0][2], A[1][2], A[2][2] - eval0};
  const V3 r0xr1 = cross3(row0, row1), r0xr2 = cross3(row0, row2), r1xr2 = cross3(row1, row2);
  const double d0 = dot3(r0xr1, r0xr1), d1 = dot3(r0xr2, r0xr2), d2 = dot3(r1xr2, r1xr2);
  double dmax = d0;
  int imax = 0;
  if (d1 > dmax) {
    dmax = d1;
    imax = 1;
  }
  if (d2 > dmax) imax = 2;
  if (imax == 0) {
    const double s = std::sqrt(d0);
    return {r0xr1.x / s, r0xr1.y / s, r0xr1.z / s};
  } else if (imax == 1) {
    const double s = std::sqrt(d1);
    return {r0xr2.x / s, r0xr2.y / s, r0xr2.z / s};
  }
  const double s = std::sqrt(d2);
  return {r1xr2.x / s, r1xr2.y / s, r1xr2.z / s};
}

V3 eigenvector1(const double A[3][3], const V3& e0, double eval1) {
  V3 U, V;
  if (std::fabs(e0.x) > std::fabs(e0.y)) {
    const double inv = 1.0 / std::sqrt(e0.x * e0.x + e0.z * e0.z);
    U = {-e0.z * inv, 0.0, e0.x * inv};
  } else {
    const double inv = 1.0 / std::sqrt(e0.y * e0.y + e0.z * e0.z);
    U = {0.0, e0.z * inv, -e0.y * inv};
  }
  V = cross3(e0, U);
  const V3 AU{(A[0][0] * U.x + A[0][1] * U.y) + A[0][2] * U.z, (A[0][1] * U.x + A[1][1] * U.y) + A[1][2] * U.z,
              (A[0][2] * U.x + A[1][2] * U.y) + A[2][2] * U.z};
  const V3 AV{(A[0][0] * V.x + A[0][1] * V.y) + A[0][2] * V.z, (A[0][1] * V.x + A[1][1] * V.y) + A[1][2] * V.z,
              (A[0][2] * V.x + A[1][2] * V.y) + A[2][2] * V.z};
  double m00 = dot3(U, AU) - eval1, m01 = dot3(U, AV), m11 = dot3(V, AV) - eval1;
  const double a00 = std::fabs(m00), a01 = std::fabs(m01), a11 = std::fabs(m11);
  if (a00 >= a11) {
    const double mx = std::max(a00, a01);
    if (mx > 0) {
      if (a00 >= a01) {
        m01 /= m00;
        m00 = 1 / std::sqrt(1 + m01 * m01);
        m01 *= m00;
      } else {
        m00 /= m01;
        m01 = 1 / std::sqrt(1 + m00 * m00);
        m00 *= m01;
      }
      return {m01 * U.x - m00 * V.x, m01 * U.y - m00 * V.y, m01 * U.z - m00 * V.z};
    }
    return U;
  }
  const double mx = std::max(a11, a01);
  if (mx > 0) {
    if (a11 >= a01) {
      m01 /= m11;
      m11 = 1 / std::sqrt(1 + m01 * m01);
      m01 *= m11;
    } else {
      m11 /= m01;
      m01 = 1 / std::sqrt(1 + m11 * m11);
      m11 *= m01;
    }
    return {m11 * U.x - m01 * V.x, m11 * U.y - m01 * V.y, m11 * U.z - m01 * V.z};
  }
  return U;
}

// utility/Eigen.cpp FastEigen3x3: eigenvector of the smallest eigenvalue of a symmetric 3x3
V3 fast_eigen3x3(double A[3][3]) {
  double mc = A[0][0];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) mc = std::max(mc, A[r][c]);
  if (mc == 0) return {0, 0, 0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) A[r][c] /= mc;
  const double norm = (A[0][1] * A[0][1] + A[0][2] * A[0][2]) + A[1][2] * A[1][2];
  if (norm > 0) {
    const double q = ((A[0][0] + A[1][1]) + A[2][2]) / 3;
    const double b00 = A[0][0] - q, b11 = A[1][1] - q, b22 = A[2][2] - q;
    const double p = std::sqrt((((b00 * b00 + b11 * b11) + b22 * b22) + norm * 2) / 6);
    const double c00 = b11 * b22 - A[1][2] * A[1][2];
    const double c01 = A[0][1] * b22 - A[1][2] * A[0][2];
    const double c02 = A[0][1] * A[1][2] - b11 * A[0][2];
    const double det = ((b00 * c00 - A[0][1] * c01) + A[0][2] * c02) / ((p * p) * p);
    double half_det = det * 0.5;
    half_det = std::min(std::max(half_det, -1.0), 1.0);
    const double angle = std::acos(half_det) / 3.0;
    const double two_thirds_pi = 2.09439510239319549;
    const double beta2 = std::cos(angle) * 2;
    const double beta0 = std::cos(angle + two_thirds_pi) * 2;
    const double beta1 = -(beta0 + beta2);
    const double ev0 = q + p * beta0, ev1 = q + p * beta1, ev2 = q + p * beta2;
    if (half_det >= 0) {
      const V3 e2 = eigenvector0(A, ev2);
      if (ev2 < ev0 && ev2 < ev1) return e2;
      const V3 e1 = eigenvector1(A, e2, ev1);
      if (ev1 < ev0 && ev1 < ev2) return e1;
      return cross3(e1, e2);
    }
    const V3 e0 = eigenvector0(A, ev0);
    if (ev0 < ev1 && ev0 < ev2) return e0;
    const V3 e1 = eigenvector1(A, e0, ev1);
    if (ev1 < ev0 && ev1 < ev2) return e1;
    return cross3(e0, e1);
  }
  // diagonal matrix (A was scaled by a positive number: comparisons are unchanged)
  if (A[0][0] < A[1][1] && A[0][0] < A[2][2]) return {1, 0, 0};
  if (A[1][1] < A[0][0] && A[1][1] < A[2][2]) return {0, 1, 0};
  return {0, 0, 1};
}
}  // namespace

int orc_estimate_normals(const double* pts, int64_t N, double radius, int32_t max_nn, double* out_normals, int32_t* nn_idx) {
  const double r2 = radius * radius;
#pragma omp parallel
  {
    std::vector<std::pair<double, int32_t>> cand((size_t)N);
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
      const double qx = pts[3 * i], qy = pts[3 * i + 1], qz = pts[3 * i + 2];
      for (int64_t j = 0; j < N; ++j) {  // nanoflann L2_Simple: result accumulated over x, y, z
        const double dx = qx - pts[3 * j], dy = qy - pts[3 * j + 1], dz = qz - pts[3 * j + 2];
        double d = dx * dx;
        d = d + dy * dy;
        d = d + dz * dz;
        cand[(size_t)j] = {d, (int32_t)j};
      }
      const int64_t k0 = std::min<int64_t>(max_nn, N);
      std::partial_sort(cand.begin(), cand.begin() + k0, cand.end());  // ascending (d2, index)
      int64_t k = 0;  // KDTreeFlann::SearchHybrid: knnSearch(max_nn), then cut at lower_bound(radius^2)
      while (k < k0 && cand[(size_t)k].first < r2) ++k;
      if (nn_idx)
        for (int32_t t = 0; t < max_nn; ++t) nn_idx[i * max_nn + t] = t < k ? cand[(size_t)t].second : -1;
      double C[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};  // fewer than 3 neighbours: identity covariance
      if (k >= 3) {  // utility::ComputeCovariance: cumulants in neighbour order
        double cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t t = 0; t < k; ++t) {
          const double* p = pts + 3 * (int64_t)cand[(size_t)t].second;
          cu[0] += p[0];
          cu[1] += p[1];
          cu[2] += p[2];
          cu[3] += p[0] * p[0];
          cu[4] += p[0] * p[1];
          cu[5] += p[0] * p[2];
          cu[6] += p[1] * p[1];
          cu[7] += p[1] * p[2];
          cu[8] += p[2] * p[2];
        }
        for (int t = 0; t < 9; ++t) cu[t] /= (double)k;
        C[0][0] = cu[3] - cu[0] * cu[0];
        C[1][1] = cu[6] - cu[1] * cu[1];
        C[2][2] = cu[8] - cu[2] * cu[2];
        C[0][1] = C[1][0] = cu[4] - cu[0] * cu[1];
        C[0][2] = C[2][0] = cu[5] - cu[0] * cu[2];
        C[1][2] = C[2][1] = cu[7] - cu[1] * cu[2];
      }
      V3 n = fast_eigen3x3(C);
      if (std::sqrt(dot3(n, n)) == 0.0) n = {0.0, 0.0, 1.0};  // EstimateNormals: zero normal -> (0, 0, 1)
      {  // NormalizeNormals(): Eigen normalize()
        const double z = dot3(n, n);
        if (z > 0) {
          const double s = std::sqrt(z);
          n = {n.x / s, n.y / s, n.z / s};
        }
      }
      {  // OrientNormalsTowardsCameraLocation(camera = 0)
        const V3 ref{0.0 - qx, 0.0 - qy, 0.0 - qz};
        if (std::sqrt(dot3(n, n)) == 0.0) {
          n = ref;
          const double l = std::sqrt(dot3(n, n));
          if (l == 0.0) n = {0.0, 0.0, 1.0};
          else n = {n.x / l, n.y / l, n.z / l};
        } else if (dot3(n, ref) < 0.0) {
          n = {n.x * -1.0, n.y * -1.0, n.z * -1.0};
        }
      }
      out_normals[3 * i] = n.x;
      out_normals[3 * i + 1] = n.y;
      out_normals[3 * i + 2] = n.z;
    }
  }
  return 0;
}

// ----------------------------------------------------------------------------------------------------------------
// Open3D v0.15.1 RegistrationICP (point-to-plane, L2) + GetInformationMatrixFromPointClouds, restated.
// ----------------------------------------------------------------------------------------------------------------
namespace {
inline void o3d_mul4(const double* A, const double* B, double* C) {  // column-major 4x4, k = 0..3 accumulation
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = A[0 * 4 + r] * B[c * 4 + 0];
      s = s + A[1 * 4 + r] * B[c * 4 + 1];
      s = s + A[2 * 4 + r] * B[c * 4 + 2];
      s = s + A[3 * 4 + r] * B[c * 4 + 3];
      C[c * 4 + r] = s;
    }
}
inline bool o3d_is_identity(const double* T) {  // Eigen isIdentity(): |off-diag| <= prec * 1 and |diag - 1| <= prec, prec = 1e-12
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
// PointCloud::Transform -> TransformPoints: p = (T [p 1]).head<3>() / w
void o3d_transform_points(const double* T, std::vector<double>& p) {
  const int64_t n = (int64_t)p.size() / 3;
  for (int64_t i = 0; i < n; ++i) {
    const double x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
    double v[4];
    for (int r = 0; r < 4; ++r) {
      double s = T[0 * 4 + r] * x;
      s = s + T[1 * 4 + r] * y;
      s = s + T[2 * 4 + r] * z;
      s = s + T[3 * 4 + r] * 1.0;
      v[r] = s;
    }
    p[3 * i] = v[0] / v[3];
    p[3 * i + 1] = v[1] / v[3];
    p[3 * i + 2] = v[2] / v[3];
  }
}
struct O3dCorr {
  std::vector<int32_t> tgt;  // -1 = no correspondence
  double fitness = 0, rmse = 0;
  int64_t count = 0;
};
// GetRegistrationResultAndCorrespondences: SearchHybrid(point, max_dist, 1) = nearest neighbour, kept iff d2 < max_dist^2
O3dCorr o3d_correspondences(const std::vector<double>& pcd, const double* tgt, int64_t Nt, double max_dist) {
  const int64_t Ns = (int64_t)pcd.size() / 3;
  O3dCorr c;
  c.tgt.assign((size_t)Ns, -1);
  std::vector<double> d2((size_t)Ns, 0.0);
  if (max_dist <= 0.0) return c;
  const double r2 = max_dist * max_dist;
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < Ns; ++i) {
    const double qx = pcd[3 * i], qy = pcd[3 * i + 1], qz = pcd[3 * i + 2];
    double best = std::numeric_limits<double>::infinity();
    int32_t bj = -1;
    for (int64_t j = 0; j < Nt; ++j) {
      const double dx = qx - tgt[3 * j], dy = qy - tgt[3 * j + 1], dz = qz - tgt[3 * j + 2];
      double d = dx * dx;
      d = d + dy * dy;
      d = d + dz * dz;
      if (d < best) {
        best = d;
        bj = (int32_t)j;
      }
    }
    if (bj >= 0 && best < r2) {
      c.tgt[(size_t)i] = bj;
      d2[(size_t)i] = best;
    }
  }
  double err2 = 0;
  for (int64_t i = 0; i < Ns; ++i)
    if (c.tgt[(size_t)i] >= 0) {
      err2 += d2[(size_t)i];
      c.count++;
    }
  if (c.count > 0) {
    c.fitness = (double)c.count / (double)Ns;
    c.rmse = std::sqrt(err2 / (double)c.count);
  }
  return c;
}
// Eigen LDLT<Matrix6d> (lower, in place, diagonal pivoting) + solve, restated sequentially
void o3d_ldlt_solve6(const double Ain[6][6], const double* b, double* x) {
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
  for (int k = 0; k < 6; ++k) std::swap(y[k], y[tr[k]]);   // P b
  for (int i = 0; i < 6; ++i)                               // L^-1
    for (int c = 0; c < i; ++c) y[i] -= A[i][c] * y[c];
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < 6; ++i) y[i] = std::fabs(A[i][i]) > tol ? y[i] / A[i][i] : 0.0;   // D^-1 (pseudo-inverse)
  for (int i = 5; i >= 0; --i)                              // L^-T
    for (int r = i + 1; r < 6; ++r) y[i] -= A[r][i] * y[r];
  for (int k = 5; k >= 0; --k) std::swap(y[k], y[tr[k]]);  // P^T
  for (int i = 0; i < 6; ++i) x[i] = y[i];
}
// utility::TransformVector6dToMatrix4d: R = (AngleAxis(z) * AngleAxis(y) * AngleAxis(x)).matrix() via quaternions
void o3d_vec6_to_T(const double* v, double* T) {
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
// TransformationEstimationPointToPlane::ComputeTransformation (L2 loss: w = 1)
void o3d_p2plane_update(const std::vector<double>& pcd, const double* tgt, const double* tn, const O3dCorr& c, double* update) {
  for (int i = 0; i < 16; ++i) update[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (c.count == 0 || !tn) return;
  double JTJ[6][6] = {}, JTr[6] = {};
  const int64_t Ns = (int64_t)pcd.size() / 3;
  for (int64_t i = 0; i < Ns; ++i) {
    const int32_t j = c.tgt[(size_t)i];
    if (j < 0) continue;
    const double sx = pcd[3 * i], sy = pcd[3 * i + 1], sz = pcd[3 * i + 2];
    const double nx = tn[3 * j], ny = tn[3 * j + 1], nz = tn[3 * j + 2];
    const double ex = sx - tgt[3 * j], ey = sy - tgt[3 * j + 1], ez = sz - tgt[3 * j + 2];
    const double r = (ex * nx + ey * ny) + ez * nz;
    const double J[6] = {sy * nz - sz * ny, sz * nx - sx * nz, sx * ny - sy * nx, nx, ny, nz};
    for (int a = 0; a < 6; ++a) {
      for (int b = 0; b < 6; ++b) JTJ[a][b] += J[a] * J[b];
      JTr[a] += J[a] * r;
    }
  }
  double nb[6], x[6];
  for (int a = 0; a < 6; ++a) nb[a] = -JTr[a];
  o3d_ldlt_solve6(JTJ, nb, x);
  o3d_vec6_to_T(x, update);
}
}  // namespace

int orc_o3d_registration_icp(const double* src, int64_t Ns, const double* tgt, const double* tgt_normals, int64_t Nt, double max_dist,
                             const double* init, double relative_fitness, double relative_rmse, int32_t max_iteration,
                             orc_o3d_icp_result* out) {
  if (!out || max_dist <= 0.0) return ORC_ERR_BAD_CONFIG;
  double T[16];
  for (int i = 0; i < 16; ++i) T[i] = init[i];
  std::vector<double> pcd(src, src + 3 * Ns);
  if (!o3d_is_identity(init)) o3d_transform_points(init, pcd);
  O3dCorr res = o3d_correspondences(pcd, tgt, Nt, max_dist);
  int it = 0;
  for (int i = 0; i < max_iteration; ++i) {
    double update[16], Tn[16];
    o3d_p2plane_update(pcd, tgt, tgt_normals, res, update);
    o3d_mul4(update, T, Tn);
    for (int k = 0; k < 16; ++k) T[k] = Tn[k];
    o3d_transform_points(update, pcd);
    const O3dCorr backup = res;
    res = o3d_correspondences(pcd, tgt, Nt, max_dist);
    ++it;
    if (std::fabs(backup.fitness - res.fitness) < relative_fitness && std::fabs(backup.rmse - res.rmse) < relative_rmse) break;
  }
  for (int i = 0; i < 16; ++i) out->transformation[i] = T[i];
  out->fitness = res.fitness;
  out->inlier_rmse = res.rmse;
  out->correspondences = res.count;
  out->iterations = it;
  return 0;
}

int orc_o3d_information_matrix(const double* src, int64_t Ns, const double* tgt, int64_t Nt, double max_dist, const double* T, double* info36) {
  std::vector<double> pcd(src, src + 3 * Ns);
  if (!o3d_is_identity(T)) o3d_transform_points(T, pcd);
  const O3dCorr c = o3d_correspondences(pcd, tgt, Nt, max_dist);
  double G[6][6] = {};
  for (int64_t i = 0; i < Ns; ++i) {
    const int32_t j = c.tgt[(size_t)i];
    if (j < 0) continue;
    const double x = tgt[3 * j], y = tgt[3 * j + 1], z = tgt[3 * j + 2];
    const double rows[3][6] = {{0.0, z, -y, 1.0, 0.0, 0.0}, {-z, 0.0, x, 0.0, 1.0, 0.0}, {y, -x, 0.0, 0.0, 0.0, 1.0}};
    for (int k = 0; k < 3; ++k)
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) G[a][b] += rows[k][a] * rows[k][b];
  }
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) info36[b * 6 + a] = G[a][b];
  return 0;
}

// getIdxsOfCarvedPoints (O3S/src/helpers.cpp:245-281) over a VoxelMap of the subset (O3S/src/Voxel.cpp:123-149)
void orc_carve(const double* scan, int64_t Ns, const double* map, const double* map_normals, int64_t Nm, const uint8_t* subset,
               const double* sensor, double voxel, double max_length, double truncation, double min_dot, uint8_t* remove) {
  const double inv = 1.0 / voxel;
  std::unordered_map<std::array<int32_t, 3>, std::vector<int64_t>, KeyHash> vox;
  for (int64_t i = 0; i < Nm; ++i) {
    remove[i] = 0;
    if (subset && !subset[i]) continue;
    const double* p = map + 3 * i;
    vox[{(int32_t)std::floor(p[0] * inv), (int32_t)std::floor(p[1] * inv), (int32_t)std::floor(p[2] * inv)}].push_back(i);
  }
  for (int64_t i = 0; i < Ns; ++i) {
    const double dx = scan[3 * i] - sensor[0], dy = scan[3 * i + 1] - sensor[1], dz = scan[3 * i + 2] - sensor[2];
    const double length = std::sqrt((dx * dx + dy * dy) + dz * dz);
    const double ux = dx / length, uy = dy / length, uz = dz / length;
    double distance = 0.0;
    const double maxPath = std::max(voxel, std::min(length - truncation, max_length));
    while (distance < maxPath) {
      const double cx = distance * ux + sensor[0], cy = distance * uy + sensor[1], cz = distance * uz + sensor[2];
      const auto it = vox.find({(int32_t)std::floor(cx * inv), (int32_t)std::floor(cy * inv), (int32_t)std::floor(cz * inv)});
      if (it != vox.end()) {
        for (const int64_t id : it->second) {
          bool rm = true;
          if (map_normals) {
            const double* n = map_normals + 3 * id;
            double nx = n[0], ny = n[1], nz = n[2];
            const double z = (nx * nx + ny * ny) + nz * nz;  // Eigen normalized()
            if (z > 0) {
              const double s = std::sqrt(z);
              nx /= s;
              ny /= s;
              nz /= s;
            }
            rm = std::fabs((ux * nx + uy * ny) + uz * nz) > min_dot;
          }
          if (rm) remove[id] = 1;
        }
      }
      distance += voxel;
    }
  }
}

// Open3D v0.15.1 geometry::PointCloud::VoxelDownSample (published algorithm; external to the reference tree):
// voxel_min_bound = min_bound - voxel/2; ref_coord = (p - voxel_min_bound)/voxel; idx = floor(ref_coord);
// average point / normal per voxel (normals are averaged, NOT renormalised).
int64_t orc_voxel_downsample_o3d(double voxel_size, const double* pts, const double* normals, int64_t N, double* out_pts,
                                 double* out_normals, int32_t* out_voxel_idx) {
  if (N == 0) return 0;
  double mn[3] = {pts[0], pts[1], pts[2]};
  for (int64_t i = 1; i < N; ++i)
    for (int d = 0; d < 3; ++d) mn[d] = std::min(mn[d], pts[3 * i + d]);
  for (int d = 0; d < 3; ++d) mn[d] -= voxel_size * 0.5;
  std::unordered_map<std::array<int32_t, 3>, Acc, KeyHash> vox;
  std::vector<std::array<int32_t, 3>> order;
  for (int64_t i = 0; i < N; ++i) {
    std::array<int32_t, 3> key;
    for (int d = 0; d < 3; ++d) key[d] = (int32_t)std::floor((pts[3 * i + d] - mn[d]) / voxel_size);
    auto it = vox.find(key);
    if (it == vox.end()) {
      it = vox.emplace(key, Acc()).first;
      order.push_back(key);
    }
    Acc& a = it->second;
    for (int d = 0; d < 3; ++d) {
      a.p[d] += pts[3 * i + d];
      if (normals) a.n[d] += normals[3 * i + d];
    }
    a.num++;
  }
  int64_t n_out = 0;
  for (const auto& key : order) {
    const Acc& a = vox[key];
    for (int d = 0; d < 3; ++d) {
      out_pts[3 * n_out + d] = a.p[d] / double(a.num);
      if (normals) out_normals[3 * n_out + d] = a.n[d] / double(a.num);
      if (out_voxel_idx) out_voxel_idx[3 * n_out + d] = key[d];
    }
    ++n_out;
  }
  return n_out;
}

// Colours / covariances of the two voxelisers, emitted in the SAME order as the points of orc_voxelize_within_crop
// (mode 0: pass-through points first, then voxels in first-touch order) / orc_voxel_downsample_o3d (mode 1).
//   mode 0  AccumulatedPoint::AddPoint (O3S/src/helpers.cpp:30-64): color_ = cloud.colors_[index] whenever isValidColor — which
//           compares `c.array().all()` (a bool) with 0.0 and 1.0 and is therefore always true (helpers.cpp:83-85) — so the
//           voxel keeps the LAST colour in input order and GetAverageColor returns it undivided; covariance_ += ..., / n.
//   mode 1  Open3D v0.15.1 VoxelDownSample: colours and covariances are averaged.
int64_t orc_voxelize_attrs(int mode, const orc_cropper* c, double voxel_size, const double* pts, const double* colors, const double* covs,
                           int64_t N, double* out_colors, double* out_covs) {
  struct AccA {
    double col[3] = {0, 0, 0};
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int num = 0;
  };
  int64_t n_out = 0;
  auto copy_point = [&](int64_t i) {
    if (colors)
      for (int d = 0; d < 3; ++d) out_colors[3 * n_out + d] = colors[3 * i + d];
    if (covs)
      for (int d = 0; d < 9; ++d) out_covs[9 * n_out + d] = covs[9 * i + d];
    ++n_out;
  };
  if (voxel_size <= 0.0) {
    for (int64_t i = 0; i < N; ++i) copy_point(i);
    return n_out;
  }
  double mn[3] = {0, 0, 0};
  if (mode == 1 && N > 0) {
    for (int d = 0; d < 3; ++d) mn[d] = pts[d];
    for (int64_t i = 1; i < N; ++i)
      for (int d = 0; d < 3; ++d) mn[d] = std::min(mn[d], pts[3 * i + d]);
    for (int d = 0; d < 3; ++d) mn[d] -= voxel_size * 0.5;
  }
  const double inv = 1.0 / voxel_size;
  std::unordered_map<std::array<int32_t, 3>, AccA, KeyHash> vox;
  std::vector<std::array<int32_t, 3>> order;
  for (int64_t i = 0; i < N; ++i) {
    const double* p = pts + 3 * i;
    bool in = true;
    if (mode == 0) {
      const bool in0 = withinImpl(c, p);
      in = c->invert ? !in0 : in0;
    }
    if (!in) {
      copy_point(i);
      continue;
    }
    std::array<int32_t, 3> key;
    for (int d = 0; d < 3; ++d)
      key[d] = mode == 0 ? (int32_t)std::floor(p[d] * inv) : (int32_t)std::floor((p[d] - mn[d]) / voxel_size);
    auto it = vox.find(key);
    if (it == vox.end()) {
      it = vox.emplace(key, AccA()).first;
      order.push_back(key);
    }
    AccA& a = it->second;
    if (colors)
      for (int d = 0; d < 3; ++d) a.col[d] = mode == 0 ? colors[3 * i + d] : a.col[d] + colors[3 * i + d];
    if (covs)
      for (int d = 0; d < 9; ++d) a.cov[d] += covs[9 * i + d];
    a.num++;
  }
  for (const auto& key : order) {
    const AccA& a = vox[key];
    if (colors)
      for (int d = 0; d < 3; ++d) out_colors[3 * n_out + d] = mode == 0 ? a.col[d] : a.col[d] / double(a.num);
    if (covs)
      for (int d = 0; d < 9; ++d) out_covs[9 * n_out + d] = a.cov[d] / double(a.num);
    ++n_out;
  }
  return n_out;
}

// covariances under o3d_slam::transform (O3S/src/helpers.cpp:310-314): R * cov * R^T, restated as two plain k = 0..2 products;
// an (almost-)identity T returns the input covariances first (the cloud copy of :285-288), then the transformed ones
int64_t orc_transform_cov(const double* T, const double* covs, int64_t N, double* out) {
  auto M = [&](int r, int c) { return T[c * 4 + r]; };
  double dev = 0.0;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) dev = std::max(dev, std::fabs(M(r, c) - (r == c ? 1.0 : 0.0)));
  int64_t n = 0;
  if (dev < 1e-4)
    for (int64_t i = 0; i < N; ++i, ++n)
      for (int d = 0; d < 9; ++d) out[9 * n + d] = covs[9 * i + d];
  for (int64_t i = 0; i < N; ++i, ++n) {
    double C[3][3], RC[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) C[r][c] = covs[9 * i + c * 3 + r];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = M(r, 0) * C[0][c];
        s = s + M(r, 1) * C[1][c];
        s = s + M(r, 2) * C[2][c];
        RC[r][c] = s;
      }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = RC[r][0] * M(c, 0);
        s = s + RC[r][1] * M(c, 1);
        s = s + RC[r][2] * M(c, 2);
        out[9 * n + c * 3 + r] = s;
      }
  }
  return n;
}

// open3dToPointmatcher (CONV/src/open3d_conversions.cpp:57-118): double -> float assignment (round to nearest), pad = 1
void orc_o3d_to_pm(const double* pts, const double* normals, int64_t N, float* xyzw, float* out_normals) {
  for (int64_t i = 0; i < N; ++i) {
    xyzw[4 * i + 0] = (float)pts[3 * i + 0];
    xyzw[4 * i + 1] = (float)pts[3 * i + 1];
    xyzw[4 * i + 2] = (float)pts[3 * i + 2];
    xyzw[4 * i + 3] = 1.0f;
    if (normals && out_normals) {
      out_normals[3 * i + 0] = (float)normals[3 * i + 0];
      out_normals[3 * i + 1] = (float)normals[3 * i + 1];
      out_normals[3 * i + 2] = (float)normals[3 * i + 2];
    }
  }
}

}  // extern "C"
