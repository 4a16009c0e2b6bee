// overlap_impl.h — computeIndicesOfOverlappingPoints (O3S/src/helpers.cpp:319-345) and the loop-closure refinement that
// uses it (O3S/src/PlaceRecognition.cpp:97-150) on the device; gfx950 only.  Included at the end of cloud_ops.hip after
// dense_map_impl.h (one TU: it shares the voxel-key packing of the dense map, the rocPRIM sort / scan instantiations and
// the Open3D-semantics ICP of o3d_icp_impl.h).
//
// The reference fills a VoxelMap (unordered_map voxel key -> per-layer index lists) with the target cloud and with the
// source cloud moved by sourceToTarget, then walks the map and keeps the indices of every voxel that holds at least
// minNumPointsPerVoxel points of BOTH layers.  Here: one packed voxel key per point (getVoxelIdx, reciprocal form,
// VoxelHashMap.hpp:43-51), both key arrays radix-sorted, and per point two binary searches per sorted array (how many
// points of its own layer / of the other layer share its voxel).  The reference's output order is its hash map's
// iteration order (unspecified); ascending index order is used here, on the device and in the oracle.
#pragma once
#include "dense_map_impl.h"

namespace {
namespace o3s_cloud {

// Open3D PointCloud::Transform on points only: p' = (T [p 1]).head<3>() / w (same arithmetic as k_transform_append)
__global__ void __launch_bounds__(kB) k_ov_keys(const double* __restrict__ pts, int64_t N, const double* __restrict__ Tm /*nullable: identity*/,
                                                double inv, uint64_t* __restrict__ keys, uint32_t* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  if (Tm) {
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
  const double fx = floor(x * inv), fy = floor(y * inv), fz = floor(z * inv);
  if (dm_in_range(fx) && dm_in_range(fy) && dm_in_range(fz)) {
    keys[i] = dm_pack((int32_t)fx, (int32_t)fy, (int32_t)fz);
  } else {
    keys[i] = kDmEmpty;
    *err = 1u;
  }
}

__device__ __forceinline__ int64_t ov_lower(const uint64_t* __restrict__ a, int64_t n, uint64_t k) {  // first i with a[i] >= k
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < k) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int64_t ov_upper(const uint64_t* __restrict__ a, int64_t n, uint64_t k) {  // first i with a[i] > k
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] <= k) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// flag[i] = 1 iff the voxel of point i holds >= min_pts points of its own layer and of the other layer
__global__ void __launch_bounds__(kB) k_ov_flag(const uint64_t* __restrict__ keys, int64_t N, const uint64_t* __restrict__ own_sorted, int64_t n_own,
                                                const uint64_t* __restrict__ other_sorted, int64_t n_other, int64_t min_pts,
                                                uint32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const uint64_t k = keys[i];
  const int64_t c_other = ov_upper(other_sorted, n_other, k) - ov_lower(other_sorted, n_other, k);
  bool in = c_other >= min_pts;
  if (in && min_pts > 1) in = ov_upper(own_sorted, n_own, k) - ov_lower(own_sorted, n_own, k) >= min_pts;
  flag[i] = in ? 1u : 0u;
}

__global__ void __launch_bounds__(kB) k_ov_indices(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off, int64_t N,
                                                   int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i < N && flag[i]) out[off[i]] = i;
}

inline size_t overlap_arena_bytes(int64_t Ns, int64_t Nt) {
  const size_t ns = (size_t)Ns, nt = (size_t)Nt, nmax = std::max(ns, nt);
  size_t sort_keys_bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, sort_keys_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, nmax, 0, 64, nullptr);
  return 2 * Arena::pad(ns * 8) + 2 * Arena::pad(nt * 8) + Arena::pad(ns * 4) + Arena::pad(nt * 4) + Arena::pad((ns + 1) * 4) +
         Arena::pad((nt + 1) * 4) + Arena::pad(std::max(sort_keys_bytes, scan_temp_bytes((int64_t)nmax))) + Arena::pad(64) + Arena::pad(128) + 4096;
}

// flags + exclusive offsets of both layers on the device; counts on the host.  d_T: 16 doubles on the device.
inline int overlap_dev(OverlapWork& w, const double* d_src, int64_t Ns, const double* d_tgt, int64_t Nt, const double T[16], double voxel,
                       int64_t min_pts, uint32_t** flag_s, uint32_t** off_s, int64_t* n_s, uint32_t** flag_t, uint32_t** off_t, int64_t* n_t,
                       hipStream_t s) {
  *n_s = *n_t = 0;
  if (Ns > (int64_t)0x7fffffff || Nt > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  CK(w.arena.reserve(overlap_arena_bytes(Ns, Nt)));
  Arena& ar = w.arena;
  uint64_t* ks = ar.take<uint64_t>((size_t)Ns);
  uint64_t* ks2 = ar.take<uint64_t>((size_t)Ns);
  uint64_t* kt = ar.take<uint64_t>((size_t)Nt);
  uint64_t* kt2 = ar.take<uint64_t>((size_t)Nt);
  uint32_t* fs = ar.take<uint32_t>((size_t)Ns);
  uint32_t* ft = ar.take<uint32_t>((size_t)Nt);
  uint32_t* os = ar.take<uint32_t>((size_t)Ns + 1);
  uint32_t* ot = ar.take<uint32_t>((size_t)Nt + 1);
  size_t sort_keys_bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, sort_keys_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)std::max(Ns, Nt), 0, 64, nullptr);
  const size_t tb_scan = scan_temp_bytes(std::max(Ns, Nt));
  const size_t tb = std::max(sort_keys_bytes, tb_scan);
  void* tmp = ar.take<char>(tb);
  uint32_t* err = ar.take<uint32_t>(16);
  double* d_T = ar.take<double>(16);
  CK(hipMemsetAsync(err, 0, 4, s));
  CK(hipMemcpyAsync(d_T, T, 16 * sizeof(double), hipMemcpyHostToDevice, s));  // pageable source: staged before the call returns
  const double inv = 1.0 / voxel;
  hipLaunchKernelGGL(k_ov_keys, dim3(nblk(Ns)), dim3(kB), 0, s, d_src, Ns, (const double*)d_T, inv, ks, err);
  hipLaunchKernelGGL(k_ov_keys, dim3(nblk(Nt)), dim3(kB), 0, s, d_tgt, Nt, (const double*)nullptr, inv, kt, err);
  CK(hipGetLastError());
  size_t t1 = tb;
  CK(rocprim::radix_sort_keys(tmp, t1, ks, ks2, (size_t)Ns, 0, 64, s));
  t1 = tb;
  CK(rocprim::radix_sort_keys(tmp, t1, kt, kt2, (size_t)Nt, 0, 64, s));
  hipLaunchKernelGGL(k_ov_flag, dim3(nblk(Ns)), dim3(kB), 0, s, ks, Ns, ks2, Ns, kt2, Nt, min_pts, fs);
  hipLaunchKernelGGL(k_ov_flag, dim3(nblk(Nt)), dim3(kB), 0, s, kt, Nt, kt2, Nt, ks2, Ns, min_pts, ft);
  CK(hipGetLastError());
  int rc = scan_flags(fs, os, Ns, tmp, tb_scan, n_s, s);
  if (rc != O3S_OK) return rc;
  rc = scan_flags(ft, ot, Nt, tmp, tb_scan, n_t, s);
  if (rc != O3S_OK) return rc;
  uint32_t herr = 0;
  CK(hipMemcpyAsync(&herr, err, 4, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  if (herr) return O3S_ERR_BAD_ARGUMENT;  // NaN / voxel index beyond +-2^20: int(floor(.)) is undefined behaviour in the reference
  *flag_s = fs;
  *off_s = os;
  *flag_t = ft;
  *off_t = ot;
  return O3S_OK;
}

}  // namespace o3s_cloud
}  // namespace

extern "C" {

int o3s_overlap_indices(int device, const double* source, int64_t Ns, const double* target, int64_t Nt, const double source_to_target[16],
                        double voxel_size, int64_t min_points_per_voxel, int64_t* idx_source, int64_t* n_source, int64_t* idx_target,
                        int64_t* n_target) {
  using namespace o3s_cloud;
  if (!n_source || !n_target || !source_to_target || Ns < 0 || Nt < 0 || !(voxel_size > 0.0) || min_points_per_voxel < 1)
    return O3S_ERR_BAD_ARGUMENT;
  *n_source = *n_target = 0;
  if (Ns == 0 || Nt == 0) return O3S_OK;
  if (!source || !target || !idx_source || !idx_target) return O3S_ERR_BAD_ARGUMENT;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  hipStream_t s = nullptr;
  Buf d_s, d_t, d_is, d_it;
  CK(d_s.alloc((size_t)Ns * 24));
  CK(d_t.alloc((size_t)Nt * 24));
  CK(hipMemcpyAsync(d_s.p, source, (size_t)Ns * 24, hipMemcpyHostToDevice, s));
  CK(hipMemcpyAsync(d_t.p, target, (size_t)Nt * 24, hipMemcpyHostToDevice, s));
  OverlapWork w;
  uint32_t *fs, *os, *ft, *ot;
  int64_t ns = 0, nt = 0;
  rc = overlap_dev(w, d_s.as<double>(), Ns, d_t.as<double>(), Nt, source_to_target, voxel_size, min_points_per_voxel, &fs, &os, &ns, &ft, &ot, &nt, s);
  if (rc != O3S_OK) return rc;
  CK(d_is.alloc((size_t)std::max<int64_t>(ns, 1) * 8));
  CK(d_it.alloc((size_t)std::max<int64_t>(nt, 1) * 8));
  hipLaunchKernelGGL(k_ov_indices, dim3(nblk(Ns)), dim3(kB), 0, s, fs, os, Ns, d_is.as<int64_t>());
  hipLaunchKernelGGL(k_ov_indices, dim3(nblk(Nt)), dim3(kB), 0, s, ft, ot, Nt, d_it.as<int64_t>());
  CK(hipGetLastError());
  if (ns) CK(hipMemcpyAsync(idx_source, d_is.p, (size_t)ns * 8, hipMemcpyDeviceToHost, s));
  if (nt) CK(hipMemcpyAsync(idx_target, d_it.p, (size_t)nt * 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  *n_source = ns;
  *n_target = nt;
  return O3S_OK;
}

int o3s_o3d_registration_icp_submaps_overlap(const o3s_submap* source, const o3s_submap* target, double max_dist, const double init[16],
                                             const o3s_o3d_icp_criteria* criteria, double overlap_voxel_size, int64_t min_points_per_voxel,
                                             o3s_o3d_icp_result* result, double* info36, int64_t* n_overlap) {
  using namespace o3s_cloud;
  if (!source || !target || !init || !result || !(max_dist > 0.0) || !(overlap_voxel_size > 0.0) || min_points_per_voxel < 1)
    return O3S_ERR_BAD_ARGUMENT;
  if (source->device != target->device) return O3S_ERR_BAD_ARGUMENT;
  if (n_overlap) n_overlap[0] = n_overlap[1] = 0;
  if (source->n == 0 || target->n == 0) return O3S_ERR_EMPTY_REFERENCE;
  if (target->has_normals != 1) return O3S_ERR_BAD_SHAPE;  // "requires target pointcloud to have normals"
  int rc = set_dev(target);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(source->stream));
  CK(hipStreamSynchronize(target->stream));
  hipStream_t s = target->stream;
  const double* sp = source->pts[source->cur].d();
  const double* tp = target->pts[target->cur].d();
  const double* tn = target->nrm[target->cur].d();
  uint32_t *fs, *os, *ft, *ot;
  int64_t ns = 0, nt = 0;
  rc = overlap_dev(target->ov_work, sp, source->n, tp, target->n, init, overlap_voxel_size, min_points_per_voxel, &fs, &os, &ns, &ft, &ot, &nt, s);
  if (rc != O3S_OK) return rc;
  if (n_overlap) {
    n_overlap[0] = ns;
    n_overlap[1] = nt;
  }
  if (ns == 0 || nt == 0) return O3S_ERR_EMPTY_REFERENCE;
  // source.SelectByIndex(sourceIdxs) / target.SelectByIndex(targetIdxs) in HBM (ascending index order)
  CK(target->ov_src.ensure((size_t)ns * 24, 0, s));
  CK(target->ov_tgt.ensure((size_t)nt * 24, 0, s));
  CK(target->ov_tgtn.ensure((size_t)nt * 24, 0, s));
  hipLaunchKernelGGL(k_compact, dim3(nblk(source->n)), dim3(kB), 0, s, sp, (const double*)nullptr, source->n, fs, os, target->ov_src.d(),
                     (double*)nullptr, (int32_t*)nullptr);
  hipLaunchKernelGGL(k_compact, dim3(nblk(target->n)), dim3(kB), 0, s, tp, tn, target->n, ft, ot, target->ov_tgt.d(), target->ov_tgtn.d(),
                     (int32_t*)nullptr);
  CK(hipGetLastError());
  rc = o3d_icp_run(target->reg_work, target->ov_src.d(), ns, target->ov_tgt.d(), target->ov_tgtn.d(), nt, max_dist, init, criteria, result, s,
                   /*on_device=*/true);
  if (rc == O3S_OK && info36)
    rc = o3d_info_run(target->reg_work_info, target->ov_src.d(), ns, target->ov_tgt.d(), nt, max_dist, result->transformation, info36, s,
                      /*on_device=*/true);
  return rc;
}

}  // extern "C"
