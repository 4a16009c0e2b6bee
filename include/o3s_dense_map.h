/*
 * o3s_dense_map.h — C ABI of the device-resident DENSE map (same shared library, libo3dslam_icp_hip.so): the second
 * half of SURVEY.md 8(f) rank 4 — the reference's o3d_slam::VoxelizedPointCloud and the space carving that runs on it.
 * Paths: O3S = open3d_slam_rsl/open3d_slam/open3d_slam.
 *
 *   o3s_dense_map_insert          VoxelizedPointCloud::insert                   O3S/src/Voxel.cpp:66-88
 *   o3s_dense_map_to_point_cloud  VoxelizedPointCloud::toPointCloud             O3S/src/Voxel.cpp:90-114
 *   o3s_dense_map_transform       VoxelizedPointCloud::transform                O3S/src/Voxel.cpp:49-64
 *   o3s_dense_map_carve           Submap::carve(scan, sensorPosition, param, VoxelizedPointCloud*) without its cadence
 *                                 gate                                          O3S/src/Submap.cpp:146-157
 *                                 = removeDuplicatePointsWithinSameVoxels       O3S/src/Voxel.cpp:162-192
 *                                 + getKeysOfCarvedPoints                       O3S/src/helpers.cpp:360-390
 *                                   (getVoxelsWithinPointNeighborhood           O3S/src/VoxelHashMap.cpp:13-46)
 *                                 + removeKey                                   O3S/include/open3d_slam/VoxelHashMap.hpp:128
 *   o3s_dense_map_insert_scan     Submap::insertScanDenseMap                    O3S/src/Submap.cpp:97-113
 *
 * The map is an open-addressing hash table in HBM: one slot per voxel = key (three 21-bit biased indices packed in 63
 * bits), point count, fp64 sums of positions and normals.  Voxel keys are getVoxelIdx(p, 1 / voxel) in fp64
 * (VoxelHashMap.hpp:48-51,127); the sums of a voxel are accumulated in the order the points were inserted, so means are
 * bit-identical to the reference's sequential loop.  Voxel indices outside [-2^20, 2^20) per axis are refused with
 * O3S_ERR_BAD_ARGUMENT (at 5 cm voxels: beyond +-52 km).  Colours are not carried (nothing on this path has them).
 *
 * Conventions as in o3s_submap.h: points / normals are 3 x N doubles (point-major), poses 4x4 column-major doubles.
 * One handle = one device + one stream; calls on a handle are serialised by the caller (the reference holds
 * denseMapMutex_ around every one of them).  Return: o3s_status.
 */
#ifndef O3S_DENSE_MAP_H
#define O3S_DENSE_MAP_H

#include <stdint.h>

#include "o3s_cloud_ops.h"
#include "o3s_icp.h"
#include "o3s_scan.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct o3s_dense_map o3s_dense_map;

/* The dense-map part of SpaceCarvingParameters (O3S/include/open3d_slam/Parameters.hpp:88-95). */
typedef struct o3s_dense_carving_params {
  double neighborhood_radius_dense_map; /* 0.1: rays are marched in steps of twice this radius; must be > 0 */
  double max_raytracing_length;         /* 20.0 */
  double truncation_distance;           /* 0.1  */
  int32_t carve_space_every_n_scans;    /* 10: used by o3s_dense_map_insert_scan only */
  int32_t reserved;
} o3s_dense_carving_params;

/* voxel_size = denseMapBuilder_.mapVoxelSize_ (> 0). */
int o3s_dense_map_create(int device, double voxel_size, o3s_dense_map** out);
void o3s_dense_map_destroy(o3s_dense_map* m);
/* Number of voxels (VoxelHashMap::size). */
int64_t o3s_dense_map_size(const o3s_dense_map* m);
/* 1 once a cloud with normals has been inserted (isHasNormals_). */
int o3s_dense_map_has_normals(const o3s_dense_map* m);
void o3s_dense_map_clear(o3s_dense_map* m);

/* Host cloud already in the map frame.  normals: nullable. */
int o3s_dense_map_insert(o3s_dense_map* m, const double* pts, const double* normals, int64_t N);

/* Raw scan (sensor frame, host): crop with the dense-map cropper at the identity pose, move into the map frame
 * (o3d_slam::transform, including its quirk of emitting the cloud twice for a near-identity pose, helpers.cpp:285-304),
 * insert; then, if `carving` is not NULL (isPerformCarving) and the number of scans inserted so far leaves remainder 1
 * modulo carve_space_every_n_scans, carve with the RAW scan and the sensor position exactly as the reference does
 * (Submap.cpp:108-110).  n_removed: nullable, voxels carved away. */
int o3s_dense_map_insert_scan(o3s_dense_map* m, const o3s_cropper* dense_map_cropper, const double* raw_pts,
                              const double* raw_normals, int64_t N, const double T_map_sensor[16],
                              const o3s_dense_carving_params* carving, int64_t* n_removed);

/* The same with the raw scan that o3s_scan_preprocess left in HBM (include/o3s_scan.h): the scan that was uploaded once
 * for the scan-to-map registration also feeds the dense map, no second copy.  An o3s_scan that has not pre-processed a
 * scan yet counts as an empty scan. */
int o3s_dense_map_insert_resident_scan(o3s_dense_map* m, const o3s_cropper* dense_map_cropper, const o3s_scan* scan,
                                       const double T_map_sensor[16], const o3s_dense_carving_params* carving,
                                       int64_t* n_removed);

/* Carves along the rays sensor_position -> scan point (both in the frame the caller chooses; the reference passes the
 * raw scan): the scan is first reduced to one point per map voxel, every ray is marched up to
 * max(step, min(length - truncation, max_length)) and at each stop all existing voxels of the point neighbourhood are
 * removed.  n_removed: nullable. */
int o3s_dense_map_carve(o3s_dense_map* m, const o3s_dense_carving_params* p, const double* scan_pts, int64_t N,
                        const double sensor_position[3], int64_t* n_removed);

/* Mean position / mean (not normalised) normal of every voxel, in ascending (z, y, x) key order (the reference's
 * hash-map order is unspecified).  Buffers hold o3s_dense_map_size() entries; normals / keys (3 x V int32) / counts
 * are nullable.  *n_out (nullable) = voxels written. */
int o3s_dense_map_to_point_cloud(const o3s_dense_map* m, double* pts, double* normals, int32_t* keys, int32_t* counts,
                                 int64_t* n_out);

/* Maps the position sum and the normal sum of every voxel as points (R s + t); keys are left as they are — the
 * reference's behaviour, kept. */
int o3s_dense_map_transform(o3s_dense_map* m, const double T[16]);

#ifdef __cplusplus
}
#endif
#endif /* O3S_DENSE_MAP_H */
