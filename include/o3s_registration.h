/*
 * o3s_registration.h — C ABI of the Open3D-semantics ICP the reference uses OUTSIDE the scan-to-map path: loop-closure
 * refinement and odometry constraints between submaps (same shared library, libo3dslam_icp_hip.so).
 * Paths: O3S = open3d_slam_rsl/open3d_slam/open3d_slam.
 *
 *   o3s_o3d_registration_icp     open3d::pipelines::registration::RegistrationICP(source, target, max_dist, init,
 *                                TransformationEstimationPointToPlane(), criteria)
 *                                  O3S/src/CloudRegistration.cpp:57-61 (registerClouds), O3S/src/PlaceRecognition.cpp:111,
 *                                  O3S/src/constraint_builders.cpp:60-68
 *   o3s_o3d_information_matrix   open3d::pipelines::registration::GetInformationMatrixFromPointClouds
 *                                  O3S/src/PlaceRecognition.cpp:144-145, O3S/src/constraint_builders.cpp:71-74
 *
 * fp64 like Open3D (this is a different arithmetic from the fp32 libpointmatcher chain of o3s_icp.h): the source cloud
 * is transformed incrementally by every update, correspondences are the nearest target point with squared distance
 * < max_dist^2 (exact: uniform grid + ring search; ties to the lower index), the 6x6 system J^T J x = -J^T r is solved
 * with Eigen's pivoted LDLT restated on the host, the update is Rz * Ry * Rx through quaternions, and the loop stops when
 * both |d fitness| < relative_fitness and |d rmse| < relative_rmse, or after max_iteration updates.
 * Open3D v0.15.1 is not part of the reference tree: parity is against the oracle's restatement of its published source;
 * correspondences, fitness and iteration counts agree exactly, poses to 1e-9 (the sums run in a different order).
 * Points / normals: 3 x N column-major doubles (host); poses: Eigen::Matrix4d::data() order.  Return: o3s_status.
 */
#ifndef O3S_REGISTRATION_H
#define O3S_REGISTRATION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct o3s_o3d_icp_criteria { /* open3d ICPConvergenceCriteria; defaults 1e-6, 1e-6, 30 */
  double relative_fitness;
  double relative_rmse;
  int32_t max_iteration;
} o3s_o3d_icp_criteria;

typedef struct o3s_o3d_icp_result { /* open3d RegistrationResult */
  double transformation[16];
  double fitness;          /* correspondences / source points */
  double inlier_rmse;      /* sqrt(sum d2 / correspondences) */
  int64_t correspondences; /* correspondence_set_.size() */
  int32_t iterations;      /* updates applied */
} o3s_o3d_icp_result;

void o3s_o3d_icp_default_criteria(o3s_o3d_icp_criteria* c);
int o3s_o3d_registration_icp(int device, const double* source, int64_t Ns, const double* target,
                             const double* target_normals, int64_t Nt, double max_correspondence_distance,
                             const double init[16], const o3s_o3d_icp_criteria* criteria, o3s_o3d_icp_result* result);
/* info: 6 x 6, column-major (symmetric). */
int o3s_o3d_information_matrix(int device, const double* source, int64_t Ns, const double* target, int64_t Nt,
                               double max_correspondence_distance, const double T[16], double info[36]);


/* One candidate pair of a batch (host pointers; target_normals must not be NULL; init: 4x4 column-major). */
typedef struct o3s_o3d_pair {
  const double* source;
  int64_t n_source;
  const double* target;
  const double* target_normals;
  int64_t n_target;
  double init[16];
} o3s_o3d_pair;
/* RegistrationICP over independent candidate pairs — the loop-closure candidates of
 * PlaceRecognition::buildLoopClosureConstraints (O3S/src/PlaceRecognition.cpp:70-150, a serial loop in the reference)
 * or the odometry constraints between adjacent submaps (O3S/src/constraint_builders.cpp:55-75) — run concurrently on one
 * device, each pair on its own HIP stream.  Every pair gives exactly the result of o3s_o3d_registration_icp on it.
 * infos (nullable): n_pairs x 36 doubles, GetInformationMatrixFromPointClouds at each pair's final transformation
 * (PlaceRecognition.cpp:144-145), computed on the pair's registration index (the correspondences of o3s_o3d_information_matrix,
 * its sums in another order: equal to 1e-12 relative).  status: n_pairs o3s_status values; the return value is the first one that is not OK. */
int o3s_o3d_registration_icp_batch(int device, int32_t n_pairs, const o3s_o3d_pair* pairs,
                                   double max_correspondence_distance, const o3s_o3d_icp_criteria* criteria,
                                   o3s_o3d_icp_result* results, double* infos, int32_t* status);

/* computeIndicesOfOverlappingPoints (O3S/src/helpers.cpp:319-345, called at O3S/src/PlaceRecognition.cpp:103 in front of
 * the loop-closure ICP): the points of `source` (moved by source_to_target, Open3D PointCloud::Transform) and of `target`
 * that fall into voxels of edge voxel_size (getVoxelIdx, reciprocal form, VoxelHashMap.hpp:43-51) holding at least
 * min_points_per_voxel points of BOTH clouds.  The reference emits them in its unordered_map's iteration order
 * (unspecified); here: ascending index order.  idx_source / idx_target hold up to Ns / Nt entries (size_t in the
 * reference).  Points whose voxel index leaves +-2^20 (or NaN) are refused with O3S_ERR_BAD_ARGUMENT. */
int o3s_overlap_indices(int device, const double* source, int64_t Ns, const double* target, int64_t Nt,
                        const double source_to_target[16], double voxel_size, int64_t min_points_per_voxel,
                        int64_t* idx_source, int64_t* n_source, int64_t* idx_target, int64_t* n_target);

/* Work memory of the registrations above and of o3s_o3d_registration_icp_submaps[_overlap] (o3s_submap.h).  Open3D allocates its
 * KD-tree and correspondence sets per call (Registration.cpp RegistrationICP); on the device an allocation stalls every stream for
 * milliseconds, so the library keeps its work areas per device and hands them out per call.  o3s_o3d_registration_reserve sizes
 * one area for clouds of up to max_source_points / max_target_points ahead of time (a mapper calls it once with its submaps'
 * maxNumPoints, O3S param MapBuilderParameters): no registration up to those sizes allocates afterwards.  Without it the areas
 * grow on demand.  o3s_o3d_registration_release returns the idle areas of the device to the allocator. */
int o3s_o3d_registration_reserve(int device, int64_t max_source_points, int64_t max_target_points);
/* `count` areas of that size (the lanes of o3s_o3d_registration_icp_submaps_overlap_batch / o3s_o3d_registration_icp_batch: up to four). */
int o3s_o3d_registration_reserve_n(int device, int64_t max_source_points, int64_t max_target_points, int32_t count);
int o3s_o3d_registration_release(int device);

#ifdef __cplusplus
}
#endif
#endif /* O3S_REGISTRATION_H */
