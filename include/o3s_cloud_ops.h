/*
 * o3s_cloud_ops.h — C ABI of the open3d_slam-side point-cloud operators around the ICP path (same shared library,
 * libo3dslam_icp_hip.so).  Paths are relative to the upstream reference checkout:
 *   O3S  = open3d_slam_rsl/open3d_slam/open3d_slam,  CONV = open3d_slam_rsl/open3d_utils/open3d_conversions
 *
 *   o3s_voxel_idx              getVoxelIdx(p, InverseVoxelSize)        O3S/include/open3d_slam/VoxelHashMap.hpp:43-51
 *   o3s_voxel_hash             EigenVec3iHash                          O3S/include/open3d_slam/VoxelHashMap.hpp:25-35
 *   o3s_crop                   CroppingVolume::crop                    O3S/src/croppers.cpp:76-106 (+ predicates :121-167)
 *   o3s_voxelize_within_crop   voxelizeWithinCroppingVolume            O3S/src/helpers.cpp:117-192
 *   o3s_voxel_downsample       o3d_slam::voxelize -> Open3D v0.15.1 PointCloud::VoxelDownSample   O3S/src/helpers.cpp:108-115
 *   o3s_o3d_to_pm              open3dToPointmatcher                    CONV/src/open3d_conversions.cpp:57-118
 *   o3s_estimate_normals       EstimateNormals(Hybrid(radius, max_nn)) + NormalizeNormals + OrientNormalsTowardsCameraLocation
 *                              (Open3D v0.15.1)                        O3S/src/CloudRegistration.cpp:71-74, O3S/src/Submap.cpp:269-271
 *
 * Conventions: stateless; `device` is the HIP device ordinal; points / normals are 3 x N column-major doubles (the
 * memory of std::vector<Eigen::Vector3d>); the caller owns all buffers, which are HOST pointers (the library stages
 * them through HBM).  Return value: o3s_status (include/o3s_icp.h).  There is no CPU fallback.
 * Arithmetic is fp64 in the reference's operation order, so voxel indices and crop decisions are bit-exact and voxel
 * means are bit-exact too (per-voxel sums run in input order, like the reference's sequential loop).
 */
#ifndef O3S_CLOUD_OPS_H
#define O3S_CLOUD_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* idx[3*i + a] = int(floor(pts[3*i + a] * (1.0 / voxel_size))) */
int o3s_voxel_idx(int device, const double* pts, int64_t N, double voxel_size, int32_t* idx);
/* hash[i] = static_cast<unsigned int>(x + y*17191 + z*17191^2) evaluated in size_t (negatives wrap mod 2^64) */
int o3s_voxel_hash(int device, const int32_t* idx, int64_t N, uint64_t* hash);

typedef struct o3s_cropper {
  int32_t kind;   /* 0 = CroppingVolume (everything inside), 1 = MaxRadius(p0), 2 = MinRadius(p0),
                     3 = MinMaxRadius(p0 = min, p1 = max), 4 = Cylinder(p0 = radius, p1 = minZ, p2 = maxZ) */
  int32_t invert; /* CroppingVolume::setIsInvertVolume */
  double p0, p1, p2;
  double centre[3]; /* pose_.translation() */
} o3s_cropper;

/* Order-preserving compaction of the points inside the volume.  out_* hold up to N points; *n_out = kept count.
 * normals / out_normals may be NULL. */
int o3s_crop(int device, const o3s_cropper* c, const double* pts, const double* normals, int64_t N, double* out_pts,
             double* out_normals, int64_t* n_out);

/* Points outside the cropper pass through first (input order); then one mean point (+ normalised mean of the non-NaN
 * normals) per occupied voxel of the absolute grid, in ascending (z, y, x) voxel-index order (the reference's hash-map
 * order is unspecified — compare as a set keyed by voxel index).  out_voxel_idx (nullable): 3 x n_out int32, INT32_MIN
 * for pass-through points.  voxel_size <= 0 copies the cloud. */
int o3s_voxelize_within_crop(int device, const o3s_cropper* c, double voxel_size, const double* pts,
                             const double* normals, int64_t N, double* out_pts, double* out_normals,
                             int32_t* out_voxel_idx, int64_t* n_out);

/* Open3D v0.15.1 VoxelDownSample: grid anchored at min_bound - voxel/2, idx = floor((p - anchor) / voxel); mean point
 * and mean (not renormalised) normal per voxel, ascending (z, y, x) voxel-index order. */
int o3s_voxel_downsample(int device, double voxel_size, const double* pts, const double* normals, int64_t N,
                         double* out_pts, double* out_normals, int32_t* out_voxel_idx, int64_t* n_out);

/* The same operators with the optional per-point attributes of open3d::geometry::PointCloud riding along (all nullable):
 * colors 3 x N, covariances 9 x N (Eigen::Matrix3d in memory order).  O3S/src/croppers.cpp:76-106 copies both;
 * voxelizeWithinCroppingVolume keeps, per voxel, the LAST colour in input order (AccumulatedPoint::AddPoint assigns the
 * colour, its isValidColor test is always true, GetAverageColor returns it undivided: O3S/src/helpers.cpp:30-64, 83-85)
 * and the mean covariance; Open3D's VoxelDownSample averages colours and covariances. */
int o3s_crop_attr(int device, const o3s_cropper* c, const double* pts, const double* normals, const double* colors,
                  const double* covariances, int64_t N, double* out_pts, double* out_normals, double* out_colors,
                  double* out_covariances, int64_t* n_out);
int o3s_voxelize_within_crop_attr(int device, const o3s_cropper* c, double voxel_size, const double* pts,
                                  const double* normals, const double* colors, const double* covariances, int64_t N,
                                  double* out_pts, double* out_normals, double* out_colors, double* out_covariances,
                                  int32_t* out_voxel_idx, int64_t* n_out);
int o3s_voxel_downsample_attr(int device, double voxel_size, const double* pts, const double* normals,
                              const double* colors, const double* covariances, int64_t N, double* out_pts,
                              double* out_normals, double* out_colors, double* out_covariances, int32_t* out_voxel_idx,
                              int64_t* n_out);
/* o3d_slam::transform (O3S/src/helpers.cpp:283-318): p' = (T [p 1]).head<3>() / w, n' = (T [n 0]).head<3>(),
 * C' = R C R^T; colours are copied by the caller (out->colors_ = cloud.colors_).  For an (almost-)identity T
 * (max |T - I| < 1e-4) the reference returns the cloud TWICE — the copy of :285-288 followed by the loop's appends — and so
 * does this: the out buffers hold up to 2 N points, *n_out says how many. */
int o3s_transform_cloud(int device, const double T[16], const double* pts, const double* normals,
                        const double* covariances, int64_t N, double* out_pts, double* out_normals,
                        double* out_covariances, int64_t* n_out);

/* fp64 xyz (+ normals) -> fp32 PM::DataPoints layout: xyzw 4 x N (pad = 1) and normals 3 x N. */
int o3s_o3d_to_pm(int device, const double* pts, const double* normals, int64_t N, float* xyzw, float* out_normals);

/* Normals of a cloud without normals, as every call site of the reference computes them: for each point the max_nn
 * nearest points of the same cloud (itself included) with squared distance < radius^2, covariance from the nine
 * cumulants in neighbour order, eigenvector of the smallest eigenvalue (closed-form symmetric 3x3 solver), unit length,
 * flipped towards the sensor origin (0, 0, 0); fewer than 3 neighbours -> (0, 0, 1) before orientation.  Exact
 * neighbour lists (uniform grid, ring search); ties in distance go to the lower index.  1 <= max_nn <= 32.
 * out_nn_idx (nullable): N x max_nn int32, ascending distance, -1 padded.  Open3D itself is not part of the reference
 * tree: parity is against the oracle's restatement of its published source (tolerance 1e-9 on the components; the
 * neighbour lists are bit-exact). */
int o3s_estimate_normals(int device, const double* pts, int64_t N, double radius, int32_t max_nn, double* out_normals,
                         int32_t* out_nn_idx);

#ifdef __cplusplus
}
#endif
#endif /* O3S_CLOUD_OPS_H */
