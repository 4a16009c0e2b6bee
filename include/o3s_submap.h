/*
 * o3s_submap.h — C ABI of the device-resident active submap (same shared library, libo3dslam_icp_hip.so): SURVEY.md
 * 8(f) rank 1.  The map cloud lives in HBM between scans; inserting a scan, re-voxelising the map inside the map-builder
 * cropping volume, cropping the scan-matcher patch and handing it to the ICP as its reference all run on the device, so
 * the reference's per-scan whole-map host loops and its double -> float copy ("This is time consuming",
 * O3S/src/Mapper.cpp:356) disappear and no copy of the map leaves HBM.
 * Paths: O3S = open3d_slam_rsl/open3d_slam/open3d_slam.
 *
 *   o3s_submap_insert_scan     Submap::insertScan without carving        O3S/src/Submap.cpp:39-96
 *                              = o3d_slam::transform                      O3S/src/helpers.cpp:283-318
 *                              + mapCloud_ += transformed; cropper pose   O3S/src/Submap.cpp:84-88
 *                              + voxelizeInsideCroppingVolume             O3S/src/Submap.cpp:159-167 -> helpers.cpp:117-192
 *   o3s_submap_set_reference   ScanToMapIcp::cropSubmap                   O3S/src/ScanToMapRegistration.cpp:90-96
 *                              + open3dToPointmatcher + icp_.initReference   O3S/src/Mapper.cpp:349-366
 *
 * Conventions: points / normals are 3 x N column-major doubles (std::vector<Eigen::Vector3d>), poses are 4x4 doubles
 * in column-major order (Eigen::Matrix4d::data()); cropper poses only use the translation (croppers.cpp:57-59, 121-167).
 * Arithmetic is fp64 in the reference's operation order without FMA contraction; the voxel part of the map is kept in
 * ascending (z, y, x) voxel-index order (the reference's unordered_map order is unspecified).  Return: o3s_status.
 *   o3s_submap_carve           Submap::carve (sparse map)                O3S/src/Submap.cpp:116-130
 *                              = getIdxsOfCarvedPoints + removeByIds      O3S/src/helpers.cpp:245-281, 225-232
 * Colours ride along (o3s_submap_insert_scan_colored); covariances are not kept in the resident map; the dense map lives in
 * o3s_dense_map.h; the isUseInitialMap_ branch is in cpp/o3s_mapper.hpp (o3s_submap_upload holds the initial map).
 */
#ifndef O3S_SUBMAP_H
#define O3S_SUBMAP_H

#include <stdint.h>

#include "o3s_cloud_ops.h"
#include "o3s_icp.h"
#include "o3s_registration.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct o3s_submap o3s_submap;

/* map_voxel_size = MapBuilderParameters::mapVoxelSize_ (<= 0: the map is never voxelised, Submap.cpp:164-166);
 * map_builder_cropper = the volume inside which the map is re-voxelised on every insert (its centre follows the sensor).
 * The submaps of a device share a few HIP streams (made on first use, kept for the life of the process, dealt round-robin):
 * creating a stream costs ~3 ms on the calling thread, which is the mapping thread whenever SubmapCollection::createNewSubmap
 * runs.  Work on two submaps that share a stream is merely ordered; the snapshots of o3s_submap_clone own their stream. */
int o3s_submap_create(int device, double map_voxel_size, const o3s_cropper* map_builder_cropper, o3s_submap** out);
void o3s_submap_destroy(o3s_submap* m);
/* Host scan (sensor frame, pre-processed) + mapToRangeSensor.  normals may be NULL only if every scan comes without. */
int o3s_submap_insert_scan(o3s_submap* m, const double* pts, const double* normals, int64_t N,
                           const double T_map_sensor[16]);
/* The same with the scan's colours (3 x N, nullable = o3s_submap_insert_scan).  The map keeps one colour per point the way
 * the reference's containers do: o3d_slam::transform copies the colours (O3S/src/helpers.cpp:291), `mapCloud_ += scan` keeps
 * them only while both sides have colours (Open3D PointCloud::operator+=; a scan without colours — or one doubled by the
 * almost-identity quirk, whose colour count no longer matches its point count — clears them for good), the re-voxelisation
 * gives every voxel the LAST colour of its points in input order (helpers.cpp:40-42, 57-59), carving drops the colours of
 * the carved points.  Not covered: covariances in the resident map (only the GICP variants read them). */
int o3s_submap_insert_scan_colored(o3s_submap* m, const double* pts, const double* normals, const double* colors,
                                   int64_t N, const double T_map_sensor[16]);
int o3s_submap_has_colors(const o3s_submap* m);
/* 3 x size doubles; O3S_ERR_BAD_SHAPE when the map carries no colours. */
int o3s_submap_download_colors(const o3s_submap* m, double* colors);
/* SpaceCarvingParameters (O3S/include/open3d_slam/Parameters.hpp:88-95). */
typedef struct o3s_carving_params {
  double voxel_size;              /* 0.1  */
  double max_raytracing_length;   /* 20.0 */
  double truncation_distance;     /* 0.1  */
  double min_dot_product_with_normal; /* 0.5 */
} o3s_carving_params;
/* Space carving with a RAW scan (sensor frame, host; no normals needed): the scan is moved into the map frame, every
 * ray is marched from the sensor position in steps of voxel_size, and each map point inside the map-builder cropper (at
 * the pose of the PREVIOUS insert, as in the reference: setPose follows the carve, Submap.cpp:66-86) that lies in a
 * visited voxel is removed when |ray . normal| > min_dot (always, for a map without normals).  The survivors keep their
 * order.  Call it before o3s_submap_insert_* on the scans the host's cadence selects (carveSpaceEveryNscans_).
 * n_removed: nullable. */
int o3s_submap_carve(o3s_submap* m, const o3s_carving_params* p, const double* raw_pts, int64_t N,
                     const double T_map_sensor[16], int64_t* n_removed);
int64_t o3s_submap_size(const o3s_submap* m);
/* The size without waiting for an insert whose completion is pending (o3s_submap_insert_processed, o3s_scan.h): exact
 * (at_least == at_most) when nothing is pending, else at_least = 1 — voxelising never empties a cloud — and at_most = the map before
 * the insert + the scan.  What Mapper.cpp:179 ("is this the first scan") and SubmapCollection.cpp:118-120 ("has the submap outgrown
 * maxNumPoints_") ask on every sweep is answered by the bounds almost always. */
int o3s_submap_size_bounds(const o3s_submap* m, int64_t* at_least, int64_t* at_most);
/* Room for n_points map points (and the work area their re-voxelisation needs) up front — SubmapParameters::maxNumPoints_ plus
 * one scan is what a submap can reach (SubmapCollection.cpp:118-120).  Without it the arrays double whenever the map outgrows
 * them, and every move stalls the device for a few milliseconds (hipFree / hipMalloc); the map's contents are kept either way. */
int o3s_submap_reserve(o3s_submap* m, int64_t n_points);
/* The opposite, for a submap that is no longer inserted into (SubmapCollection::createNewSubmap / a switch of the active submap,
 * SubmapCollection.cpp:94-162): everything but the map cloud goes back to the allocator — the spare ping-pong arrays, the sort /
 * scan work area, the scan staging — and the map arrays shrink to what the map holds.  The map, its layout and every
 * later call stay valid (buffers come back on demand: reserve again when the submap is re-activated).  Waits for the submap's
 * OWN stream only, and the shrink gives the map arrays new addresses: no other call that reads this submap — a registration with
 * it as source or target on another stream or thread, a clone in flight — may run concurrently with the trim; a worker that
 * refines while the mapper goes on must work on an o3s_submap_clone snapshot.  Arrays the map does not use (normals / colours
 * of a map without them) are given back whole.  o3s_submap_device_bytes reports what the object holds (tests, memory
 * accounting). */
int o3s_submap_trim(o3s_submap* m);
/* A submap is closed and a NEW one takes over (SubmapCollection::createNewSubmap, SubmapCollection.cpp:150-162, after the closing
 * scan has gone into the previous submap, :216-239): `closed` keeps its map cloud in arrays of just that size, and every other
 * device buffer it holds — the reserved map arrays, the spare ping-pong arrays, the sort / scan work area, the scan staging, the
 * patch buffers — MOVES to `fresh`.  Pointers change hands: no hipFree (each one waits for the whole device) and no allocation
 * of the large arrays, where o3s_submap_trim + o3s_submap_reserve freed and made ~0.3 GB per switch (5 - 6 ms on the mapping
 * thread, and a millisecond again every time the new submap's patch buffers doubled).  `fresh` must hold no points and live on
 * the same device; the map bits of `closed`, its layout and every later call on either object stay valid.  Waits for both
 * submaps' streams; like trim it may not run while anything else reads `closed` (an ICP handle still indexing its last patch
 * included: any compute on that handle since the last o3s_submap_set_reference has waited for the index). */
int o3s_submap_hand_over(o3s_submap* closed, o3s_submap* fresh);
int64_t o3s_submap_device_bytes(const o3s_submap* m);
/* A second submap object with the same parameters and a COPY of the map cloud, on `device` — the same GPU or another one (a peer
 * copy, over xGMI where the devices are peers).  What a loop-closure worker needs: the reference refines loop closures on a
 * thread of its own (SlamWrapper.cpp:1061-1103) while the mapper keeps inserting; with a snapshot of the two submaps taken at
 * the moment the candidate is found (0.5 M points: 24 MB, ~10 us inside one GPU's HBM) the refinement
 * (o3s_o3d_registration_icp_submaps_overlap, 3-7 ms) runs on its own stream or device and the mapping thread never waits.
 * Call it from the thread that inserts into `src` (no insert may be in flight).  Destroy the copy with o3s_submap_destroy. */
int o3s_submap_clone(const o3s_submap* src, int device, o3s_submap** out);
/* How the voxelising inserts of this submap ran so far (all pointers nullable): `merged` = the scan was sorted and merged into the
 * map, which is kept in voxel order between inserts (the reference's TODO at Submap.cpp:89-92); `sorted` = the whole map was
 * sorted again (first insert, coloured maps, after a carve or an upload, unbounded volumes); `fell_back` = a merge was started
 * and abandoned for the sort because points left behind earlier were back inside the volume (a revisit).  Same map either way. */
int o3s_submap_insert_stats(const o3s_submap* m, int64_t* merged, int64_t* sorted, int64_t* fell_back);
/* Submap::computeSubmapCenter (O3S/src/Submap.cpp:282-286) = open3d PointCloud::GetCenter(): the mean of the map points
 * (zero for an empty map).  fp64 sums in a fixed, run-independent order — not Open3D's sequential one, so the last bits may
 * differ; the value only feeds the distance tests of SubmapCollection::updateActiveSubmap. */
int o3s_submap_center(const o3s_submap* m, double center[3]);
/* Copies the resident map to the host (3 x size doubles each; normals may be NULL). */
int o3s_submap_download(const o3s_submap* m, double* pts, double* normals);
/* Replaces the resident map (e.g. a map loaded from disk). */
int o3s_submap_upload(o3s_submap* m, const double* pts, const double* normals, int64_t N);
/* Crops the map around T_map_sensor with the scan-matcher cropper, converts the patch to PM::DataPoints precision and
 * makes it the ICP handle's reference (o3s_icp_init_reference_dev) — all in HBM.  *n_patch (nullable) = patch size.
 * An empty patch returns O3S_ERR_EMPTY_REFERENCE ("Map patch is empty", Mapper.cpp:330-336). */
int o3s_submap_set_reference(o3s_submap* m, const o3s_cropper* scan_matcher_cropper, const double T_map_sensor[16],
                             o3s_icp* icp, int64_t* n_patch);

/* Size of the patch ScanToMapIcp::cropSubmap would return at T_map_sensor (O3S/src/ScanToMapRegistration.cpp:90-96) without
 * building it: the reference crops the active submap on EVERY scan and gives the scan up when the patch is empty
 * (Mapper.cpp:328-336), also between two renewals of the ICP reference.  One mask + count on the device, one read-back. */
int o3s_submap_patch_count(o3s_submap* m, const o3s_cropper* scan_matcher_cropper, const double T_map_sensor[16], int64_t* n_patch);

/* RegistrationICP(source map, target map, max_dist, init, PointToPlane, criteria) between two RESIDENT submaps of the same
 * device — the odometry constraint between adjacent submaps (O3S/src/constraint_builders.cpp:55-75) and the loop-closure
 * refinement (O3S/src/PlaceRecognition.cpp:111) without moving either cloud: the result is exactly what
 * o3s_o3d_registration_icp returns on the downloaded clouds.  info36 (nullable): GetInformationMatrixFromPointClouds
 * at the final transformation, 6 x 6 column-major, computed on the registration's index: the correspondences of a stand-alone
 * o3s_o3d_information_matrix call, its sums added in another order (equal to 1e-12 relative).  The target must carry normals (O3S_ERR_BAD_SHAPE otherwise); an
 * empty submap gives O3S_ERR_EMPTY_REFERENCE. */
int o3s_o3d_registration_icp_submaps(const o3s_submap* source, const o3s_submap* target,
                                     double max_correspondence_distance, const double init[16],
                                     const o3s_o3d_icp_criteria* criteria, o3s_o3d_icp_result* result, double* info36);

/* The loop-closure refinement of PlaceRecognition::buildLoopClosureConstraints (O3S/src/PlaceRecognition.cpp:97-150)
 * between two RESIDENT submaps: overlap selection at `init` (the RANSAC pose; overlap_voxel_size =
 * magic::voxelExpansionFactorOverlapComputation (20) x map voxel size, min_points_per_voxel = 1 in the reference),
 * RegistrationICP(source.SelectByIndex, target.SelectByIndex, max_dist, init, PointToPlane, criteria), and (info36
 * nullable) GetInformationMatrixFromPointClouds on the two selections at the refined pose — nothing leaves HBM.
 * Equals o3s_submap_download x 2 + o3s_overlap_indices + o3s_o3d_registration_icp + o3s_o3d_information_matrix on the
 * selected clouds (the information matrix up to the order of its sums, as above).  n_overlap (nullable, 2 entries): sizes of the source / target selection; an empty one returns
 * O3S_ERR_EMPTY_REFERENCE. */
int o3s_o3d_registration_icp_submaps_overlap(const o3s_submap* source, const o3s_submap* target,
                                             double max_correspondence_distance, const double init[16],
                                             const o3s_o3d_icp_criteria* criteria, double overlap_voxel_size,
                                             int64_t min_points_per_voxel, o3s_o3d_icp_result* result, double* info36,
                                             int64_t* n_overlap);

/* The same for n independent pairs at once — the candidates a finished submap is refined against (the serial loop at
 * O3S/src/PlaceRecognition.cpp:70-71, whose `omp parallel for` is commented out): up to four pairs are in flight together, each on
 * a host thread, a stream and a work area of its own (the later passes of a refinement are chains of small dependent launches that
 * leave most of the GPU idle; several chains fill it).  Every pair returns exactly what the single call returns.  inits: n x 16,
 * infos (nullable): n x 36, n_overlaps (nullable): n x 2, statuses: n (an empty overlap is that pair's O3S_ERR_EMPTY_REFERENCE, not
 * an error of the call).  All submaps on one device.  o3s_o3d_registration_reserve_n sizes the work areas of the lanes ahead. */
int o3s_o3d_registration_icp_submaps_overlap_batch(int32_t n, const o3s_submap* const* sources, const o3s_submap* const* targets,
                                                   double max_correspondence_distance, const double* inits,
                                                   const o3s_o3d_icp_criteria* criteria, double overlap_voxel_size,
                                                   int64_t min_points_per_voxel, o3s_o3d_icp_result* results, double* infos,
                                                   int64_t* n_overlaps, int32_t* statuses);

#ifdef __cplusplus
}
#endif
#endif /* O3S_SUBMAP_H */
