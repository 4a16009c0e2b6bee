/*
 * dense_map_oracle.h — CPU restatement of the reference's dense map (o3d_slam::VoxelizedPointCloud) and of its space
 * carving.  TEST INFRASTRUCTURE ONLY (part of oracle/liboracle.so): used by tests/ to check the HIP dense map
 * (include/o3s_dense_map.h); the product never links or calls it.
 *
 * Parity pinning: the reference holds NO test for VoxelizedPointCloud, getVoxelsWithinPointNeighborhood,
 * removeDuplicatePointsWithinSameVoxels or getKeysOfCarvedPoints, and cannot be compiled here (Eigen / Open3D absent),
 * so this file is a line-by-line restatement pinned only by hand-computed cases in tests/test_oracle_dense_map.py:
 * "parity unpinned" at the Eigen boundary (operation order inside Transform * Vector3d and Vector3d::norm()).
 * Paths: O3S = open3d_slam_rsl/open3d_slam/open3d_slam.
 */
#ifndef DENSE_MAP_ORACLE_H
#define DENSE_MAP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_dense_map orc_dense_map;

/* VoxelizedPointCloud(voxelSize) (O3S/src/Voxel.cpp:40; key = getVoxelIdx(p, 1/voxel), VoxelHashMap.hpp:48-51,127) */
orc_dense_map* orc_dense_create(double voxel_size);
void orc_dense_destroy(orc_dense_map* m);
/* number of voxels (VoxelHashMap::size) */
int64_t orc_dense_size(const orc_dense_map* m);
/* VoxelizedPointCloud::insert (O3S/src/Voxel.cpp:66-88): per point, in order: sum position, ++count, sum normal */
void orc_dense_insert(orc_dense_map* m, const double* pts, const double* normals /*nullable*/, int64_t N);
/* VoxelizedPointCloud::toPointCloud (O3S/src/Voxel.cpp:90-114): mean position / mean (NOT normalised) normal of every
 * voxel with count > 0, here in ascending (z, y, x) key order (the reference's hash-map order is unspecified).
 * normals / keys / counts are nullable.  Returns the number of voxels written. */
int64_t orc_dense_to_point_cloud(const orc_dense_map* m, double* pts, double* normals, int32_t* keys, int32_t* counts);
/* VoxelizedPointCloud::transform (O3S/src/Voxel.cpp:49-64): the SUMS of every voxel are mapped as points
 * (R s + t, also the normal sum) and stay under their OLD key — kept exactly as the reference does it. */
void orc_dense_transform(orc_dense_map* m, const double* T16 /*column-major*/);
/* removeDuplicatePointsWithinSameVoxels (O3S/src/Voxel.cpp:162-192): keep[i] = 1 for the first point of every voxel */
int64_t orc_remove_duplicate_points(const double* pts, int64_t N, double voxel_size, uint8_t* keep);
/* getVoxelsWithinPointNeighborhood (O3S/src/VoxelHashMap.cpp:13-46), duplicates included, in the reference's order.
 * Returns the number of keys (which may exceed cap; only the first cap are written). */
int64_t orc_voxels_within_neighborhood(const double* p3, double radius, double voxel_size, int32_t* keys, int64_t cap);
/* Submap::carve(scan, sensorPosition, param, VoxelizedPointCloud*) without the every-N-scans gate
 * (O3S/src/Submap.cpp:146-157) = removeDuplicatePointsWithinSameVoxels(scan, voxel of the map)
 * + getKeysOfCarvedPoints (O3S/src/helpers.cpp:360-390) + removeKey.  Returns the number of voxels removed. */
int64_t orc_dense_carve(orc_dense_map* m, const double* scan, int64_t N, const double* sensor3, double neighborhood_radius,
                        double max_raytracing_length, double truncation_distance);

#ifdef __cplusplus
}
#endif
#endif
