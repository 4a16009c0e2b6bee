/*
 * o3s_scan.h — C ABI of the device-resident pre-processed scan (same shared library, libo3dslam_icp_hip.so): the part
 * of SURVEY.md 8(f) rank 2 that does not need normal estimation.  Together with o3s_submap.h it keeps the whole
 * per-scan loop of Mapper::addRangeMeasurement in HBM: raw scan in, pose out, map updated.
 * Paths: O3S = open3d_slam_rsl/open3d_slam/open3d_slam.
 *
 *   o3s_scan_preprocess         ScanToMapIcp::processForScanMatchingAndMerging   O3S/src/ScanToMapRegistration.cpp:36-69
 *                               = mapBuilderCropper_->crop(in)                    croppers.cpp:76-106
 *                               + o3d_slam::voxelize(scanProcessing_.voxelSize_)  helpers.cpp:108-115 (Open3D VoxelDownSample)
 *                               + scanMatcherCropper_(identity pose)->crop(wide)  ScanToMapRegistration.cpp:62-64
 *   o3s_scan_set_reading        open3dToPointmatcher(*processed.match_) -> reading of icp_.compute   O3S/src/Mapper.cpp:307-309, 393
 *   o3s_submap_insert_processed submaps_->insertScan(rawScan, *processed.merge_, mapToRangeSensor_)   O3S/src/Mapper.cpp:487
 *
 * A cloud that carries normals keeps them (RegistrationIcpPointToPlane::estimateNormalsOrCovariancesIfNeeded returns
 * early, O3S/src/CloudRegistration.cpp:63-67); for a cloud without normals they are estimated on the voxelised wide cloud
 * (CloudRegistration.cpp:69-74 = o3s_estimate_normals) once o3s_scan_set_normal_estimation has supplied knn / radius,
 * otherwise the scan is rejected with O3S_ERR_BAD_SHAPE.  RandomDownSample is taken at ratio 1.0 ("For reproducability, random
 * rownsampling must be disabled", ScanToMapRegistration.cpp:43).  Conventions as in o3s_submap.h.
 */
#ifndef O3S_SCAN_H
#define O3S_SCAN_H

#include <stddef.h>
#include <stdint.h>

#include "o3s_cloud_ops.h"
#include "o3s_icp.h"
#include "o3s_submap.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct o3s_scan o3s_scan;

int o3s_scan_create(int device, o3s_scan** out);
void o3s_scan_destroy(o3s_scan* s);
/* CloudRegistrationParameters: maxRadiusNormalEstimation_ (icp.max_distance_knn) and knnNormalEstimation_ (icp.knn),
 * 1 <= knn <= 32.  knn <= 0 switches estimation off again. */
int o3s_scan_set_normal_estimation(o3s_scan* s, double max_radius, int32_t knn);
/* Raw scan (sensor frame, host) -> resident "merge" (wide crop, voxelised) and "match" (narrow crop of it) clouds.
 * The croppers' centres are used as given (the reference leaves both at the identity pose here).  voxel_size <= 0
 * skips the voxelisation (helpers.cpp:109-111).  n_merge / n_match (nullable) receive the sizes. */
int o3s_scan_preprocess(o3s_scan* s, const o3s_cropper* map_builder_cropper, double voxel_size,
                        const o3s_cropper* scan_matcher_cropper, const double* pts, const double* normals, int64_t N,
                        int64_t* n_merge, int64_t* n_match);
/* A raw scan staged in HBM ahead of the mapping call.  The reference feeds its mapping worker from a buffer another thread
 * fills (SlamWrapper.cpp:217-253, 660-709); here that other thread calls o3s_raw_scan_upload — a blocking host-to-device
 * copy of the sweep (6 MB for 64 x 2048 returns, ~0.2 ms of the 0.9 ms a sweep takes end to end) on the object's own stream —
 * while the mapping thread still works on the previous sweep, and o3s_scan_preprocess_staged then starts from the staged
 * copy (one device-to-device copy: the scan keeps its own raw cloud for the dense map).  One object = one sweep in flight:
 * do not upload into an object a preprocess is still reading (two objects, used alternately, are enough). */
typedef struct o3s_raw_scan o3s_raw_scan;
int o3s_raw_scan_create(int device, o3s_raw_scan** out);
void o3s_raw_scan_destroy(o3s_raw_scan* r);
int o3s_raw_scan_upload(o3s_raw_scan* r, const double* pts, const double* normals, int64_t N);
int64_t o3s_raw_scan_size(const o3s_raw_scan* r);
int o3s_scan_preprocess_staged(o3s_scan* s, const o3s_cropper* map_builder_cropper, double voxel_size,
                               const o3s_cropper* scan_matcher_cropper, const o3s_raw_scan* raw, int64_t* n_merge,
                               int64_t* n_match);
/* Page-locked host memory for the sweeps a receiving thread hands to o3s_raw_scan_upload / o3s_scan_preprocess: the
 * host-to-device copy of a 64 x 2048 sweep (6 MB) then runs at the link's rate (~0.12 ms) instead of being staged through the
 * runtime's bounce buffers (~0.4 ms from pageable memory).  Plain host memory as far as the caller is concerned; free it with
 * o3s_host_free_pinned before the process ends. */
int o3s_host_alloc_pinned(size_t bytes, void** out);
void o3s_host_free_pinned(void* p);
/* which: 0 = merge cloud, 1 = match cloud.  Returns the size; with pts != NULL also copies the cloud to the host. */
int64_t o3s_scan_get(const o3s_scan* s, int which, double* pts, double* normals);
/* The match cloud becomes the ICP handle's resident reading (o3s_icp_set_reading_dev); run o3s_icp_compute_resident
 * afterwards.  The scan object must stay alive (and unchanged) until that compute has returned. */
int o3s_scan_set_reading(o3s_scan* s, o3s_icp* icp);
/* Submap::insertScan with the resident merge cloud (no host copy).
 * Completion may be PENDING when this returns (round 5): the whole insert is enqueued on the submap's stream and its counts are on
 * their way to the host, but nobody has waited for them — the mapping thread goes on (hands the pose out, takes the next sweep) while
 * the GPU finishes.  Every later call that takes the submap (size, set_reference, download, carve, another insert, a registration
 * between submaps, hand_over, trim, clone, destroy ...) completes it first — waiting only if the GPU has not got there yet — and is the
 * call that reports an error of the completion (a failed allocation in the sort-based path the merge may have to give way to).
 * Results are those of the insert that waits (tests/test_gpu_submap.py).  The scan object may be refilled at once: its stream is
 * ordered behind the kernels that read it.  o3s_submap_size_bounds answers "is the map empty / can it have outgrown N" without waiting. */
int o3s_submap_insert_processed(o3s_submap* m, const o3s_scan* s, const double T_map_sensor[16]);

#ifdef __cplusplus
}
#endif
#endif /* O3S_SCAN_H */
