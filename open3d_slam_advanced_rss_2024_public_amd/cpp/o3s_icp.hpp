// o3s_icp.hpp — header-only C++17 shim over the C ABI (include/o3s_icp.h) for host code that stays C++/catkin.
//
// It gives open3d_slam the same two calls it makes on its public `PM::ICP icp_` member
// (open3d_slam/include/open3d_slam/Mapper.hpp:70-72):
//     icp_.initReference(referenceDataPoints)                     Mapper.cpp:363
//     icp_.compute(reading, {}, T_init, false)                    Mapper.cpp:393
// and rethrows the library's status codes as the exceptions libpointmatcher would have thrown, so the Mapper's existing
// `catch (const std::runtime_error&)` (Mapper.cpp:420-422) keeps working unchanged.
// No Eigen / libpointmatcher headers are needed: matrices are passed as raw column-major float pointers, which is what
// `PM::Matrix::data()` returns.  See INTEGRATION.md for the exact patch.
#pragma once

#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>

#include "o3s_icp.h"
#include "o3s_dense_map.h"
#include "o3s_submap.h"

namespace o3s {

// PointMatcher<T>::ConvergenceError derives from std::runtime_error (PointMatcher.h:142-147); so does TransformationError.
struct ConvergenceError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct TransformationError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

class IcpHip {
 public:
  // cfg mirrors param/icp.yaml; o3s_icp_default_config gives the shipped values
  explicit IcpHip(int device = 0) {
    o3s_icp_config cfg;
    o3s_icp_default_config(&cfg);
    create(cfg, device);
  }
  IcpHip(const o3s_icp_config& cfg, int device) { create(cfg, device); }
  ~IcpHip() { o3s_icp_destroy(h_); }
  IcpHip(const IcpHip&) = delete;
  IcpHip& operator=(const IcpHip&) = delete;

  // ICP::initReference(referenceIn): features = referenceIn.features.data() (4 x M), normals =
  // referenceIn.getDescriptorViewByName("normals").data() (3 x M).  Returns false for an empty cloud (ICP.cpp:295-298).
  bool initReference(const float* features4xM, const float* normals3xM, std::int64_t M) {
    const int rc = o3s_icp_init_reference(h_, features4xM, normals3xM, M);
    if (rc == O3S_ERR_EMPTY_REFERENCE) return false;
    raise(rc);
    return true;
  }

  // ICP::compute(readingIn, {}, T_refIn_readIn, false): T_init / T_out are 4x4 column-major (PM::TransformationParameters::data()).
  void compute(const float* features4xN, const float* normals3xN, std::int64_t N, const float* T_init, float* T_out) {
    raise(o3s_icp_compute(h_, features4xN, normals3xN, N, T_init, T_out, &stats_));
  }

  bool getMaxNumIterationsReached() const { return stats_.max_iters_reached != 0; }  // PointMatcher.h:786
  const o3s_icp_stats& stats() const { return stats_; }
  o3s_icp* handle() { return h_; }

 private:
  void create(const o3s_icp_config& cfg, int device) {
    const int rc = o3s_icp_create(&cfg, device, &h_);
    if (rc != O3S_OK) throw std::runtime_error(std::string("o3s_icp_create: ") + o3s_last_error(nullptr));
  }
  void raise(int rc) const {
    if (rc == O3S_OK) return;
    const std::string msg = o3s_last_error(h_);
    switch (rc) {
      case O3S_ERR_NO_MATCHES:
      case O3S_ERR_NO_POINTS:
      case O3S_ERR_NAN: throw ConvergenceError(msg);
      case O3S_ERR_NOT_RIGID: throw TransformationError(msg);
      default: throw std::runtime_error(msg);
    }
  }
  o3s_icp* h_ = nullptr;
  o3s_icp_stats stats_{};
};

// Device-resident map cloud of the active submap (include/o3s_submap.h).  Mirrors the two places the reference walks
// the whole map on the host for every scan:
//     Submap::insertScan(raw, preProcessed, mapToRangeSensor, time, carve)        Submap.cpp:39-96
//     ScanToMapIcp::cropSubmap + open3dToPointmatcher + icp_.initReference        Mapper.cpp:328-366
// points / normals: std::vector<Eigen::Vector3d>::data() cast to double*; poses: Eigen::Matrix4d::data().
class SubmapHip {
 public:
  SubmapHip(double mapVoxelSize, const o3s_cropper& mapBuilderCropper, int device = 0) {
    const int rc = o3s_submap_create(device, mapVoxelSize, &mapBuilderCropper, &m_);
    if (rc != O3S_OK) throw std::runtime_error("o3s_submap_create failed (status " + std::to_string(rc) + ")");
  }
  ~SubmapHip() { o3s_submap_destroy(m_); }
  SubmapHip(const SubmapHip&) = delete;
  SubmapHip& operator=(const SubmapHip&) = delete;

  bool insertScan(const double* points3xN, const double* normals3xN, std::int64_t N, const double* mapToRangeSensor4x4) {
    const int rc = o3s_submap_insert_scan(m_, points3xN, normals3xN, N, mapToRangeSensor4x4);
    if (rc != O3S_OK) throw std::runtime_error("o3s_submap_insert_scan failed (status " + std::to_string(rc) + ")");
    return true;
  }
  // Submap::carve (Submap.cpp:116-130) on the raw scan; call before insertScan on the scans the cadence selects
  std::int64_t carve(const o3s_carving_params& p, const double* rawPoints3xN, std::int64_t N, const double* mapToRangeSensor4x4) {
    std::int64_t removed = 0;
    if (o3s_submap_carve(m_, &p, rawPoints3xN, N, mapToRangeSensor4x4, &removed) != O3S_OK) throw std::runtime_error("o3s_submap_carve failed");
    return removed;
  }
  std::int64_t size() const { return o3s_submap_size(m_); }
  // the two questions the per-scan loop asks about the size, answered without waiting for an insert whose completion is pending
  // (o3s_submap_size_bounds) whenever the bounds decide them
  bool empty() const {
    std::int64_t lo = 0, hi = 0;
    if (o3s_submap_size_bounds(m_, &lo, &hi) != O3S_OK) throw std::runtime_error("o3s_submap_size_bounds failed");
    return hi == 0 || (lo == 0 && size() == 0);
  }
  bool largerThan(std::int64_t n) const {
    std::int64_t lo = 0, hi = 0;
    if (o3s_submap_size_bounds(m_, &lo, &hi) != O3S_OK) throw std::runtime_error("o3s_submap_size_bounds failed");
    if (hi <= n) return false;
    if (lo > n) return true;
    return size() > n;
  }
  // a copy of the map on `device` (o3s_submap_clone): the snapshot a loop-closure worker refines while this submap is inserted into
  o3s_submap* cloneHandle(int device) const {
    o3s_submap* c = nullptr;
    if (o3s_submap_clone(m_, device, &c) != O3S_OK) throw std::runtime_error("o3s_submap_clone failed");
    return c;
  }
  // room for nPoints up front (SubmapParameters::maxNumPoints_ + one scan): no re-allocation stalls while the map grows
  void reserve(std::int64_t nPoints) {
    if (o3s_submap_reserve(m_, nPoints) != O3S_OK) throw std::runtime_error("o3s_submap_reserve failed");
  }
  // this submap is closed, `fresh` (empty) takes over: the map stays here in arrays of its own size, every other device buffer
  // moves to `fresh` (o3s_submap_hand_over: no hipFree / hipMalloc of the large arrays)
  void handOverTo(SubmapHip& fresh) {
    if (o3s_submap_hand_over(m_, fresh.m_) != O3S_OK) throw std::runtime_error("o3s_submap_hand_over failed");
  }
  // a submap that is no longer inserted into gives everything but its map cloud back to the allocator (o3s_submap_trim)
  void trim() { (void)o3s_submap_trim(m_); }
  // false = "Map patch is empty" (Mapper.cpp:330-336) or an empty reference (ICP.cpp:295-298)
  bool setReference(const o3s_cropper& scanMatcherCropper, const double* mapToRangeSensor4x4, IcpHip& icp, std::int64_t* nPatch = nullptr) {
    const int rc = o3s_submap_set_reference(m_, &scanMatcherCropper, mapToRangeSensor4x4, icp.handle(), nPatch);
    if (rc == O3S_ERR_EMPTY_REFERENCE) return false;
    if (rc != O3S_OK) throw std::runtime_error(std::string("o3s_submap_set_reference: ") + o3s_last_error(icp.handle()));
    return true;
  }
  // size of the patch cropSubmap would return at this pose (Mapper.cpp:328): the reference looks at it on EVERY scan
  std::int64_t patchCount(const o3s_cropper& scanMatcherCropper, const double* mapToRangeSensor4x4) {
    std::int64_t n = 0;
    if (o3s_submap_patch_count(m_, &scanMatcherCropper, mapToRangeSensor4x4, &n) != O3S_OK) throw std::runtime_error("o3s_submap_patch_count failed");
    return n;
  }
  void download(double* points3xN, double* normals3xN) const {
    if (o3s_submap_download(m_, points3xN, normals3xN) != O3S_OK) throw std::runtime_error("o3s_submap_download failed");
  }
  o3s_submap* handle() { return m_; }

 private:
  o3s_submap* m_ = nullptr;
};

// Device-resident stand-in for Submap::denseMap_ (o3d_slam::VoxelizedPointCloud) and the calls Submap makes on it:
//     Submap::insertScanDenseMap(rawScan, mapToRangeSensor, time, isPerformCarving)   Submap.cpp:97-113
//     Submap::carve(scan, sensorPosition, param, &denseMap_)                          Submap.cpp:146-157
//     VoxelizedPointCloud::insert / toPointCloud / transform                          Voxel.cpp:49-114
class DenseMapHip {
 public:
  explicit DenseMapHip(double denseMapVoxelSize, int device = 0) {
    const int rc = o3s_dense_map_create(device, denseMapVoxelSize, &m_);
    if (rc != O3S_OK) throw std::runtime_error("o3s_dense_map_create failed (status " + std::to_string(rc) + ")");
  }
  ~DenseMapHip() { o3s_dense_map_destroy(m_); }
  DenseMapHip(const DenseMapHip&) = delete;
  DenseMapHip& operator=(const DenseMapHip&) = delete;

  void insert(const double* points3xN, const double* normals3xN, std::int64_t N) {
    check(o3s_dense_map_insert(m_, points3xN, normals3xN, N), "o3s_dense_map_insert");
  }
  // carving == nullptr <=> isPerformCarving == false; returns the number of voxels carved away
  std::int64_t insertScanDenseMap(const o3s_cropper& denseMapCropper, const double* rawPoints3xN, const double* rawNormals3xN, std::int64_t N,
                                  const double* mapToRangeSensor4x4, const o3s_dense_carving_params* carving = nullptr) {
    std::int64_t removed = 0;
    check(o3s_dense_map_insert_scan(m_, &denseMapCropper, rawPoints3xN, rawNormals3xN, N, mapToRangeSensor4x4, carving, &removed),
          "o3s_dense_map_insert_scan");
    return removed;
  }
  std::int64_t carve(const o3s_dense_carving_params& p, const double* scanPoints3xN, std::int64_t N, const double* sensorPosition3) {
    std::int64_t removed = 0;
    check(o3s_dense_map_carve(m_, &p, scanPoints3xN, N, sensorPosition3, &removed), "o3s_dense_map_carve");
    return removed;
  }
  void transform(const double* T4x4) { check(o3s_dense_map_transform(m_, T4x4), "o3s_dense_map_transform"); }
  std::int64_t size() const { return o3s_dense_map_size(m_); }
  bool empty() const { return size() == 0; }
  bool hasNormals() const { return o3s_dense_map_has_normals(m_) != 0; }
  // buffers sized by size(); returns the number of voxels written
  std::int64_t toPointCloud(double* points3xV, double* normals3xV) const {
    std::int64_t n = 0;
    check(o3s_dense_map_to_point_cloud(m_, points3xV, normals3xV, nullptr, nullptr, &n), "o3s_dense_map_to_point_cloud");
    return n;
  }
  o3s_dense_map* handle() { return m_; }

 private:
  static void check(int rc, const char* what) {
    if (rc != O3S_OK) throw std::runtime_error(std::string(what) + " failed (status " + std::to_string(rc) + ")");
  }
  o3s_dense_map* m_ = nullptr;
};

}  // namespace o3s
