// cloud_ops.hip — open3d_slam-side point-cloud operators (C ABI: include/o3s_cloud_ops.h), gfx950 only.
//
// Domain kernels are hand-written; rocPRIM (the native AMD primitive library shipped with ROCm) supplies the stable
// radix sort and the scans that order voxels.  fp64 throughout, no FMA contraction (-ffp-contract=off), so every
// decision (voxel index, inside/outside) and every per-voxel mean is bit-identical to the reference's sequential loops.
#include "../../include/o3s_cloud_ops.h"
#include "../../include/o3s_icp.h"

#include <string.h>

#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr int kB = 256;
inline unsigned nblk(int64_t n) { return (unsigned)((n + kB - 1) / kB); }

struct Buf {
  void* p = nullptr;
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
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
__global__ void __launch_bounds__(kB) k_mask(o3s_cropper c, const double* __restrict__ pts, int64_t N, int keep_inside,
                                             uint32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const bool in = within(c, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  flag[i] = (in == (keep_inside != 0)) ? 1u : 0u;
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

// voxel index of every voxelised point; mode 0: absolute grid, reciprocal form (helpers.cpp:156); mode 1: Open3D
// (p - anchor) / voxel.  Points that are not voxelised (flag == 1 = pass-through) get no index.
__global__ void __launch_bounds__(kB) k_vox_keys_idx(const double* __restrict__ pts, int64_t N, const uint32_t* __restrict__ passflag,
                                                     int mode, double inv, double voxel, double ax, double ay, double az,
                                                     int32_t* __restrict__ vidx, int32_t* __restrict__ mm /*min[3], max[3]*/) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  if (passflag && passflag[i]) return;
  int32_t v[3];
  if (mode == 0) {
    v[0] = (int32_t)floor(pts[3 * i] * inv);
    v[1] = (int32_t)floor(pts[3 * i + 1] * inv);
    v[2] = (int32_t)floor(pts[3 * i + 2] * inv);
  } else {
    v[0] = (int32_t)floor((pts[3 * i] - ax) / voxel);
    v[1] = (int32_t)floor((pts[3 * i + 1] - ay) / voxel);
    v[2] = (int32_t)floor((pts[3 * i + 2] - az) / voxel);
  }
  for (int a = 0; a < 3; ++a) {
    vidx[3 * i + a] = v[a];
    atomicMin(&mm[a], v[a]);
    atomicMax(&mm[3 + a], v[a]);
  }
}

__global__ void __launch_bounds__(kB) k_vox_pack(int64_t N, const uint32_t* __restrict__ passflag, const int32_t* __restrict__ vidx,
                                                 int32_t x0, int32_t y0, int32_t z0, uint64_t ex, uint64_t ey,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  vals[i] = (uint32_t)i;
  if (passflag && passflag[i]) {
    keys[i] = ~0ull;
    return;
  }
  const uint64_t x = (uint64_t)((int64_t)vidx[3 * i] - x0), y = (uint64_t)((int64_t)vidx[3 * i + 1] - y0),
                 z = (uint64_t)((int64_t)vidx[3 * i + 2] - z0);
  keys[i] = (z * ey + y) * ex + x;
}

__global__ void __launch_bounds__(kB) k_heads(const uint64_t* __restrict__ keys, int64_t N, uint32_t* __restrict__ head) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const uint64_t k = keys[i];
  head[i] = (k != ~0ull && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

// one lane per voxel: sums run over the voxel's points in ascending input index (stable sort), exactly the order of the
// reference's sequential accumulation (helpers.cpp:30-44, 153-161), so the fp64 means are bit-identical.
__global__ void __launch_bounds__(kB) k_vox_reduce(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                   const uint32_t* __restrict__ head, const uint32_t* __restrict__ ord, int64_t N,
                                                   const double* __restrict__ pts, const double* __restrict__ nrm, const int32_t* __restrict__ vidx,
                                                   int skip_nan_normals, int normalise, int64_t out_base, double* __restrict__ out_pts,
                                                   double* __restrict__ out_n, int32_t* __restrict__ out_idx) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N || !head[i]) return;
  const uint64_t k = keys[i];
  double sp[3] = {0, 0, 0}, sn[3] = {0, 0, 0};
  int cnt = 0;
  int64_t j = i;
  for (; j < N && keys[j] == k; ++j) {
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

__global__ void __launch_bounds__(kB) k_min_bound(const double* __restrict__ pts, int64_t N, unsigned long long* __restrict__ mn /*3, ordered bits*/) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  for (int a = 0; a < 3; ++a) {
    // order-preserving map of a double to u64 so that atomicMin works on negatives too
    unsigned long long u = (unsigned long long)__double_as_longlong(pts[3 * i + a]);
    u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
    atomicMin(&mn[a], u);
  }
}

int exclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, hipStream_t s) {
  size_t bytes = 0;
  CK(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
  Buf tmp;
  CK(tmp.alloc(bytes));
  CK(rocprim::exclusive_scan(tmp.p, bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
  return O3S_OK;
}

// shared tail of the two voxelisers: sort voxelised points by packed voxel key, reduce per voxel, return counts
int voxel_pipeline(int mode, const o3s_cropper* crop, double voxel, const double* pts, const double* normals, int64_t N,
                   double* out_pts, double* out_normals, int32_t* out_voxel_idx, int64_t* n_out) {
  *n_out = 0;
  if (N == 0) return O3S_OK;
  if (N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  hipStream_t s = nullptr;
  Buf d_pts, d_nrm, d_flag, d_off, d_vidx, d_mm, d_keys, d_vals, d_keys2, d_vals2, d_head, d_ord, d_opts, d_on, d_oidx, d_mn;
  CK(d_pts.alloc((size_t)N * 24));
  CK(hipMemcpyAsync(d_pts.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(d_nrm.alloc((size_t)N * 24));
    CK(hipMemcpyAsync(d_nrm.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  CK(d_opts.alloc((size_t)N * 24));
  CK(d_on.alloc((size_t)N * 24));
  CK(d_oidx.alloc((size_t)N * 12));
  int64_t n_pass = 0;
  const uint32_t* passflag = nullptr;
  if (crop) {  // pass-through points: outside the volume, emitted first in input order (helpers.cpp:162-176)
    CK(d_flag.alloc((size_t)N * 4));
    CK(d_off.alloc((size_t)(N + 1) * 4));
    hipLaunchKernelGGL(k_mask, dim3(nblk(N)), dim3(kB), 0, s, *crop, d_pts.as<double>(), N, 0, d_flag.as<uint32_t>());
    int rc = exclusive_scan_u32(d_flag.as<uint32_t>(), d_off.as<uint32_t>(), N, s);
    if (rc != O3S_OK) return rc;
    uint32_t last_off = 0, last_flag = 0;
    CK(hipMemcpyAsync(&last_off, d_off.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(&last_flag, d_flag.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    n_pass = (int64_t)last_off + last_flag;
    hipLaunchKernelGGL(k_compact, dim3(nblk(N)), dim3(kB), 0, s, d_pts.as<double>(), normals ? d_nrm.as<double>() : nullptr, N,
                       d_flag.as<uint32_t>(), d_off.as<uint32_t>(), d_opts.as<double>(), d_on.as<double>(), d_oidx.as<int32_t>());
    passflag = d_flag.as<uint32_t>();
  }
  double ax = 0, ay = 0, az = 0;
  if (mode == 1) {  // Open3D: anchor = min_bound - voxel/2
    CK(d_mn.alloc(24));
    CK(hipMemsetAsync(d_mn.p, 0xff, 24, s));
    hipLaunchKernelGGL(k_min_bound, dim3(nblk(N)), dim3(kB), 0, s, d_pts.as<double>(), N, d_mn.as<unsigned long long>());
    unsigned long long mn[3];
    CK(hipMemcpyAsync(mn, d_mn.p, 24, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
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
  CK(d_vidx.alloc((size_t)N * 12));
  CK(d_mm.alloc(24));
  const int32_t mm_init[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
  CK(hipMemcpyAsync(d_mm.p, mm_init, 24, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_vox_keys_idx, dim3(nblk(N)), dim3(kB), 0, s, d_pts.as<double>(), N, passflag, mode, 1.0 / voxel, voxel, ax, ay, az,
                     d_vidx.as<int32_t>(), d_mm.as<int32_t>());
  int32_t mm[6];
  CK(hipMemcpyAsync(mm, d_mm.p, 24, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  int64_t n_vox = 0;
  if (n_pass < N) {
    const uint64_t ex = (uint64_t)((int64_t)mm[3] - mm[0] + 1), ey = (uint64_t)((int64_t)mm[4] - mm[1] + 1),
                   ez = (uint64_t)((int64_t)mm[5] - mm[2] + 1);
    const long double prod = (long double)ex * (long double)ey * (long double)ez;
    if (prod >= 9.0e18L) return O3S_ERR_BAD_ARGUMENT;  // voxel index range does not pack into 63 bits
    CK(d_keys.alloc((size_t)N * 8));
    CK(d_vals.alloc((size_t)N * 4));
    CK(d_keys2.alloc((size_t)N * 8));
    CK(d_vals2.alloc((size_t)N * 4));
    hipLaunchKernelGGL(k_vox_pack, dim3(nblk(N)), dim3(kB), 0, s, N, passflag, d_vidx.as<int32_t>(), mm[0], mm[1], mm[2], ex, ey,
                       d_keys.as<uint64_t>(), d_vals.as<uint32_t>());
    size_t bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_vals.as<uint32_t>(), d_vals2.as<uint32_t>(),
                                 (size_t)N, 0, 64, s));
    Buf tmp;
    CK(tmp.alloc(bytes));
    CK(rocprim::radix_sort_pairs(tmp.p, bytes, d_keys.as<uint64_t>(), d_keys2.as<uint64_t>(), d_vals.as<uint32_t>(), d_vals2.as<uint32_t>(),
                                 (size_t)N, 0, 64, s));
    CK(d_head.alloc((size_t)N * 4));
    CK(d_ord.alloc((size_t)(N + 1) * 4));
    hipLaunchKernelGGL(k_heads, dim3(nblk(N)), dim3(kB), 0, s, d_keys2.as<uint64_t>(), N, d_head.as<uint32_t>());
    int rc = exclusive_scan_u32(d_head.as<uint32_t>(), d_ord.as<uint32_t>(), N, s);
    if (rc != O3S_OK) return rc;
    uint32_t lo = 0, lh = 0;
    CK(hipMemcpyAsync(&lo, d_ord.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(&lh, d_head.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    n_vox = (int64_t)lo + lh;
    hipLaunchKernelGGL(k_vox_reduce, dim3(nblk(N)), dim3(kB), 0, s, d_keys2.as<uint64_t>(), d_vals2.as<uint32_t>(), d_head.as<uint32_t>(),
                       d_ord.as<uint32_t>(), N, d_pts.as<double>(), normals ? d_nrm.as<double>() : nullptr, d_vidx.as<int32_t>(),
                       mode == 0 ? 1 : 0, mode == 0 ? 1 : 0, n_pass, d_opts.as<double>(), d_on.as<double>(), d_oidx.as<int32_t>());
  }
  CK(hipGetLastError());
  const int64_t total = n_pass + n_vox;
  CK(hipMemcpyAsync(out_pts, d_opts.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
  if (normals && out_normals) CK(hipMemcpyAsync(out_normals, d_on.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
  if (out_voxel_idx) CK(hipMemcpyAsync(out_voxel_idx, d_oidx.p, (size_t)total * 12, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  *n_out = total;
  return O3S_OK;
}

}  // namespace

extern "C" {

int o3s_voxel_idx(int device, const double* pts, int64_t N, double voxel_size, int32_t* idx) {
  if (!pts || !idx || N < 0 || !(voxel_size > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  if (N == 0) return O3S_OK;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  Buf a, b;
  CK(a.alloc((size_t)N * 24));
  CK(b.alloc((size_t)N * 12));
  CK(hipMemcpy(a.p, pts, (size_t)N * 24, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_voxel_idx, dim3(nblk(3 * N)), dim3(kB), 0, nullptr, a.as<double>(), 3 * N, 1.0 / voxel_size, b.as<int32_t>());
  CK(hipGetLastError());
  CK(hipMemcpy(idx, b.p, (size_t)N * 12, hipMemcpyDeviceToHost));
  return O3S_OK;
}

int o3s_voxel_hash(int device, const int32_t* idx, int64_t N, uint64_t* hash) {
  if (!idx || !hash || N < 0) return O3S_ERR_BAD_ARGUMENT;
  if (N == 0) return O3S_OK;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  Buf a, b;
  CK(a.alloc((size_t)N * 12));
  CK(b.alloc((size_t)N * 8));
  CK(hipMemcpy(a.p, idx, (size_t)N * 12, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_voxel_hash, dim3(nblk(N)), dim3(kB), 0, nullptr, a.as<int32_t>(), N, b.as<uint64_t>());
  CK(hipGetLastError());
  CK(hipMemcpy(hash, b.p, (size_t)N * 8, hipMemcpyDeviceToHost));
  return O3S_OK;
}

int o3s_o3d_to_pm(int device, const double* pts, const double* normals, int64_t N, float* xyzw, float* out_normals) {
  if (!pts || !xyzw || N < 0 || (normals && !out_normals)) return O3S_ERR_BAD_ARGUMENT;
  if (N == 0) return O3S_OK;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  Buf a, b, c, d;
  CK(a.alloc((size_t)N * 24));
  CK(c.alloc((size_t)N * 16));
  CK(hipMemcpy(a.p, pts, (size_t)N * 24, hipMemcpyHostToDevice));
  if (normals) {
    CK(b.alloc((size_t)N * 24));
    CK(d.alloc((size_t)N * 12));
    CK(hipMemcpy(b.p, normals, (size_t)N * 24, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(k_o3d_to_pm, dim3(nblk(N)), dim3(kB), 0, nullptr, a.as<double>(), normals ? b.as<double>() : nullptr, N, c.as<float4>(),
                     d.as<float>());
  CK(hipGetLastError());
  CK(hipMemcpy(xyzw, c.p, (size_t)N * 16, hipMemcpyDeviceToHost));
  if (normals) CK(hipMemcpy(out_normals, d.p, (size_t)N * 12, hipMemcpyDeviceToHost));
  return O3S_OK;
}

int o3s_crop(int device, const o3s_cropper* c, const double* pts, const double* normals, int64_t N, double* out_pts, double* out_normals,
             int64_t* n_out) {
  if (!c || !pts || !out_pts || !n_out || N < 0 || (normals && !out_normals)) return O3S_ERR_BAD_ARGUMENT;
  *n_out = 0;
  if (N == 0) return O3S_OK;
  if (N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  hipStream_t s = nullptr;
  Buf d_pts, d_nrm, d_flag, d_off, d_opts, d_on;
  CK(d_pts.alloc((size_t)N * 24));
  CK(hipMemcpyAsync(d_pts.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(d_nrm.alloc((size_t)N * 24));
    CK(hipMemcpyAsync(d_nrm.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  CK(d_flag.alloc((size_t)N * 4));
  CK(d_off.alloc((size_t)(N + 1) * 4));
  CK(d_opts.alloc((size_t)N * 24));
  CK(d_on.alloc((size_t)N * 24));
  hipLaunchKernelGGL(k_mask, dim3(nblk(N)), dim3(kB), 0, s, *c, d_pts.as<double>(), N, 1, d_flag.as<uint32_t>());
  rc = exclusive_scan_u32(d_flag.as<uint32_t>(), d_off.as<uint32_t>(), N, s);
  if (rc != O3S_OK) return rc;
  uint32_t lo = 0, lf = 0;
  CK(hipMemcpyAsync(&lo, d_off.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
  CK(hipMemcpyAsync(&lf, d_flag.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
  hipLaunchKernelGGL(k_compact, dim3(nblk(N)), dim3(kB), 0, s, d_pts.as<double>(), normals ? d_nrm.as<double>() : nullptr, N, d_flag.as<uint32_t>(),
                     d_off.as<uint32_t>(), d_opts.as<double>(), d_on.as<double>(), (int32_t*)nullptr);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(s));
  const int64_t kept = (int64_t)lo + lf;
  CK(hipMemcpy(out_pts, d_opts.p, (size_t)kept * 24, hipMemcpyDeviceToHost));
  if (normals) CK(hipMemcpy(out_normals, d_on.p, (size_t)kept * 24, hipMemcpyDeviceToHost));
  *n_out = kept;
  return O3S_OK;
}

int o3s_voxelize_within_crop(int device, const o3s_cropper* c, double voxel_size, const double* pts, const double* normals, int64_t N,
                             double* out_pts, double* out_normals, int32_t* out_voxel_idx, int64_t* n_out) {
  if (!c || !pts || !out_pts || !n_out || N < 0 || (normals && !out_normals)) return O3S_ERR_BAD_ARGUMENT;
  if (voxel_size <= 0.0) {  // helpers.cpp:122-125: the cloud is returned unchanged
    std::memcpy(out_pts, pts, (size_t)N * 24);
    if (normals) std::memcpy(out_normals, normals, (size_t)N * 24);
    if (out_voxel_idx)
      for (int64_t i = 0; i < 3 * N; ++i) out_voxel_idx[i] = INT32_MIN;
    *n_out = N;
    return O3S_OK;
  }
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  return voxel_pipeline(0, c, voxel_size, pts, normals, N, out_pts, out_normals, out_voxel_idx, n_out);
}

int o3s_voxel_downsample(int device, double voxel_size, const double* pts, const double* normals, int64_t N, double* out_pts,
                         double* out_normals, int32_t* out_voxel_idx, int64_t* n_out) {
  if (!pts || !out_pts || !n_out || N < 0 || !(voxel_size > 0.0) || (normals && !out_normals)) return O3S_ERR_BAD_ARGUMENT;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  return voxel_pipeline(1, nullptr, voxel_size, pts, normals, N, out_pts, out_normals, out_voxel_idx, n_out);
}

}  // extern "C"
