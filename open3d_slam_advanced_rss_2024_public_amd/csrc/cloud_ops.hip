// cloud_ops.hip — open3d_slam-side point-cloud operators on HOST buffers (C ABI: include/o3s_cloud_ops.h), gfx950 only.
// The kernels and the device-level pipelines live in cloud_dev.h (shared with the device-resident submap, submap.hip);
// this file stages the caller's buffers through HBM.
#include "cloud_dev.h"

using namespace o3s_cloud;

namespace {

// host-buffer wrapper of voxel_pipeline_dev (colours / covariances optional)
int voxel_pipeline(int mode, const o3s_cropper* crop, double voxel, const double* pts, const double* normals, const double* colors,
                   const double* covariances, int64_t N, double* out_pts, double* out_normals, double* out_colors, double* out_covariances,
                   int32_t* out_voxel_idx, int64_t* n_out) {
  *n_out = 0;
  if (N == 0) return O3S_OK;
  if (N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  hipStream_t s = nullptr;
  Buf d_pts, d_nrm, d_opts, d_on, d_oidx, d_col, d_cov, d_ocol, d_ocov;
  Arena ar;
  CK(d_pts.alloc((size_t)N * 24));
  CK(hipMemcpyAsync(d_pts.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(d_nrm.alloc((size_t)N * 24));
    CK(hipMemcpyAsync(d_nrm.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  Attrs at;
  if (colors) {
    CK(d_col.alloc((size_t)N * 24));
    CK(d_ocol.alloc((size_t)N * 24));
    CK(hipMemcpyAsync(d_col.p, colors, (size_t)N * 24, hipMemcpyHostToDevice, s));
    at.col = d_col.as<double>();
    at.out_col = d_ocol.as<double>();
  }
  if (covariances) {
    CK(d_cov.alloc((size_t)N * 72));
    CK(d_ocov.alloc((size_t)N * 72));
    CK(hipMemcpyAsync(d_cov.p, covariances, (size_t)N * 72, hipMemcpyHostToDevice, s));
    at.cov = d_cov.as<double>();
    at.out_cov = d_ocov.as<double>();
  }
  CK(d_opts.alloc((size_t)N * 24));
  CK(d_on.alloc((size_t)N * 24));
  CK(d_oidx.alloc((size_t)N * 12));
  int64_t total = 0;
  const int rc = voxel_pipeline_dev(ar, mode, crop, voxel, d_pts.as<double>(), normals ? d_nrm.as<double>() : nullptr, N, d_opts.as<double>(),
                                    d_on.as<double>(), d_oidx.as<int32_t>(), &total, s, &at);
  if (rc != O3S_OK) return rc;
  CK(hipMemcpyAsync(out_pts, d_opts.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
  if (normals && out_normals) CK(hipMemcpyAsync(out_normals, d_on.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
  if (colors) CK(hipMemcpyAsync(out_colors, d_ocol.p, (size_t)total * 24, hipMemcpyDeviceToHost, s));
  if (covariances) CK(hipMemcpyAsync(out_covariances, d_ocov.p, (size_t)total * 72, hipMemcpyDeviceToHost, s));
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

int o3s_crop_attr(int device, const o3s_cropper* c, const double* pts, const double* normals, const double* colors, const double* covariances,
                  int64_t N, double* out_pts, double* out_normals, double* out_colors, double* out_covariances, int64_t* n_out) {
  if (!c || !pts || !out_pts || !n_out || N < 0 || (normals && !out_normals) || (colors && !out_colors) || (covariances && !out_covariances))
    return O3S_ERR_BAD_ARGUMENT;
  *n_out = 0;
  if (N == 0) return O3S_OK;
  if (N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  hipStream_t s = nullptr;
  Buf d_pts, d_nrm, d_opts, d_on, d_col, d_cov, d_ocol, d_ocov;
  Arena ar;
  CK(d_pts.alloc((size_t)N * 24));
  CK(hipMemcpyAsync(d_pts.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(d_nrm.alloc((size_t)N * 24));
    CK(hipMemcpyAsync(d_nrm.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  Attrs at;
  if (colors) {
    CK(d_col.alloc((size_t)N * 24));
    CK(d_ocol.alloc((size_t)N * 24));
    CK(hipMemcpyAsync(d_col.p, colors, (size_t)N * 24, hipMemcpyHostToDevice, s));
    at.col = d_col.as<double>();
    at.out_col = d_ocol.as<double>();
  }
  if (covariances) {
    CK(d_cov.alloc((size_t)N * 72));
    CK(d_ocov.alloc((size_t)N * 72));
    CK(hipMemcpyAsync(d_cov.p, covariances, (size_t)N * 72, hipMemcpyHostToDevice, s));
    at.cov = d_cov.as<double>();
    at.out_cov = d_ocov.as<double>();
  }
  CK(d_opts.alloc((size_t)N * 24));
  CK(d_on.alloc((size_t)N * 24));
  int64_t kept = 0;
  rc = crop_dev(ar, *c, d_pts.as<double>(), normals ? d_nrm.as<double>() : nullptr, N, d_opts.as<double>(), d_on.as<double>(), &kept, s, &at);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(out_pts, d_opts.p, (size_t)kept * 24, hipMemcpyDeviceToHost));
  if (normals) CK(hipMemcpy(out_normals, d_on.p, (size_t)kept * 24, hipMemcpyDeviceToHost));
  if (colors) CK(hipMemcpy(out_colors, d_ocol.p, (size_t)kept * 24, hipMemcpyDeviceToHost));
  if (covariances) CK(hipMemcpy(out_covariances, d_ocov.p, (size_t)kept * 72, hipMemcpyDeviceToHost));
  *n_out = kept;
  return O3S_OK;
}

int o3s_crop(int device, const o3s_cropper* c, const double* pts, const double* normals, int64_t N, double* out_pts, double* out_normals,
             int64_t* n_out) {
  return o3s_crop_attr(device, c, pts, normals, nullptr, nullptr, N, out_pts, out_normals, nullptr, nullptr, n_out);
}

int o3s_voxelize_within_crop_attr(int device, const o3s_cropper* c, double voxel_size, const double* pts, const double* normals,
                                  const double* colors, const double* covariances, int64_t N, double* out_pts, double* out_normals,
                                  double* out_colors, double* out_covariances, int32_t* out_voxel_idx, int64_t* n_out) {
  if (!c || !pts || !out_pts || !n_out || N < 0 || (normals && !out_normals) || (colors && !out_colors) || (covariances && !out_covariances))
    return O3S_ERR_BAD_ARGUMENT;
  if (voxel_size <= 0.0) {  // helpers.cpp:122-125: the cloud is returned unchanged
    std::memcpy(out_pts, pts, (size_t)N * 24);
    if (normals) std::memcpy(out_normals, normals, (size_t)N * 24);
    if (colors) std::memcpy(out_colors, colors, (size_t)N * 24);
    if (covariances) std::memcpy(out_covariances, covariances, (size_t)N * 72);
    if (out_voxel_idx)
      for (int64_t i = 0; i < 3 * N; ++i) out_voxel_idx[i] = INT32_MIN;
    *n_out = N;
    return O3S_OK;
  }
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  return voxel_pipeline(0, c, voxel_size, pts, normals, colors, covariances, N, out_pts, out_normals, out_colors, out_covariances, out_voxel_idx, n_out);
}

int o3s_voxelize_within_crop(int device, const o3s_cropper* c, double voxel_size, const double* pts, const double* normals, int64_t N,
                             double* out_pts, double* out_normals, int32_t* out_voxel_idx, int64_t* n_out) {
  return o3s_voxelize_within_crop_attr(device, c, voxel_size, pts, normals, nullptr, nullptr, N, out_pts, out_normals, nullptr, nullptr,
                                       out_voxel_idx, n_out);
}

int o3s_voxel_downsample_attr(int device, double voxel_size, const double* pts, const double* normals, const double* colors,
                              const double* covariances, int64_t N, double* out_pts, double* out_normals, double* out_colors,
                              double* out_covariances, int32_t* out_voxel_idx, int64_t* n_out) {
  if (!pts || !out_pts || !n_out || N < 0 || !(voxel_size > 0.0) || (normals && !out_normals) || (colors && !out_colors) ||
      (covariances && !out_covariances))
    return O3S_ERR_BAD_ARGUMENT;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  return voxel_pipeline(1, nullptr, voxel_size, pts, normals, colors, covariances, N, out_pts, out_normals, out_colors, out_covariances, out_voxel_idx,
                        n_out);
}

int o3s_voxel_downsample(int device, double voxel_size, const double* pts, const double* normals, int64_t N, double* out_pts,
                         double* out_normals, int32_t* out_voxel_idx, int64_t* n_out) {
  return o3s_voxel_downsample_attr(device, voxel_size, pts, normals, nullptr, nullptr, N, out_pts, out_normals, nullptr, nullptr, out_voxel_idx, n_out);
}

}  // extern "C"

#include "o3d_icp_impl.h"
#include "submap_impl.h"
#include "dense_map_impl.h"
#include "overlap_impl.h"

#ifdef O3S_TEST_HOOKS
// hooks build only (tests/test_gpu_cloud_ops.py): the pair sort of the work areas (cloud_dev.h sort_pairs) on host arrays
extern "C" int o3s_test_sort_pairs(int device, const uint64_t* keys, const uint32_t* vals, int64_t n, int end_bit, uint64_t* keys_out, uint32_t* vals_out) {
  using namespace o3s_cloud;
  if (n <= 0 || !keys || !vals || !keys_out || !vals_out) return O3S_ERR_BAD_ARGUMENT;
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  Buf k1, k2, v1, v2, tmp;
  size_t tb = sort_temp_bytes(n);
  CK(k1.alloc((size_t)n * 8));
  CK(k2.alloc((size_t)n * 8));
  CK(v1.alloc((size_t)n * 4));
  CK(v2.alloc((size_t)n * 4));
  CK(tmp.alloc(tb));
  CK(hipMemcpy(k1.p, keys, (size_t)n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(v1.p, vals, (size_t)n * 4, hipMemcpyHostToDevice));
  CK(sort_pairs(tmp.p, tb, k1.as<uint64_t>(), k2.as<uint64_t>(), v1.as<uint32_t>(), v2.as<uint32_t>(), (size_t)n, end_bit, nullptr));
  CK(hipMemcpy(keys_out, k2.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(vals_out, v2.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return O3S_OK;
}
#endif
