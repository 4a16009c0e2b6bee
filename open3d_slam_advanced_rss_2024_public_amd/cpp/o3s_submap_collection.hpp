// o3s_submap_collection.hpp — header-only C++17 restatement of the submap bookkeeping that sits between the Mapper and the
// map clouds: o3d_slam::SubmapCollection (open3d_slam/src/SubmapCollection.cpp), with every submap resident on the device
// (o3s_submap over the C ABI).  Line numbers below are SubmapCollection.cpp's.
//
//   :28-32     constructor: one empty submap, scan buffer of 5 (setParameters: numScansOverlap_, :255-256)
//   :86-92     insertBufferedScans: the buffered (pre-processed) scans go into the new active submap at their own poses
//   :94-148    updateActiveSubmap: forced creation; minNumRangeData_ gate; localisation mode never switches; closest submap by
//              centre; maxNumPoints_ forces a new submap at the NEXT scan; another submap within radius_: stay / switch when
//              adjacent (isSwitchingSubmapsConsistant returns true, :392-407) / create when the active one is left behind;
//              nobody within radius_: create
//   :150-162   createNewSubmap (id, parent, origin)
//   :164-174   findClosestSubmap (first minimum of the centre distances)
//   :193-247   insertScan: buffer the scan; on a switch the scan
//              still goes into the PREVIOUS submap, whose centre is then computed (Submap::computeSubmapCenter, Submap.cpp:
//              282-286), it is queued as finished, an adjacency edge is added, the buffer is replayed into the new one
// Not here: feature computation / place recognition / pose-graph transforms of finished submaps (host work, out of scope).
// The scans the buffer keeps are resident o3s_scan objects: the caller hands over the scan it has just pre-processed and
// gets another one to fill next (a ring of numScansOverlap_ + 1 handles, nothing is copied).
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>

#include "o3s_icp.hpp"
#include "o3s_scan.h"

namespace o3s {

struct SubmapParams {            // o3d_slam::SubmapParameters (Parameters.hpp:103-109)
  double radius = 20.0;
  int minNumRangeData = 5;
  std::int64_t maxNumPoints = 400000;
  int numScansOverlap = 3;
};

// o3d_slam::AdjacencyMatrix (AdjacencyMatrix.cpp:16-21, 61-71), the part updateActiveSubmap asks
class AdjacencyHip {
 public:
  void addEdge(std::size_t a, std::size_t b) {
    adj_[a].insert(b);
    adj_[b].insert(a);
  }
  bool isAdjacent(std::size_t a, std::size_t b) const {
    if (a == b) return true;
    const auto it = adj_.find(a);
    return it != adj_.end() && it->second.count(b) != 0;
  }
  const std::map<std::size_t, std::set<std::size_t>>& edges() const { return adj_; }

 private:
  std::map<std::size_t, std::set<std::size_t>> adj_;
};

class SubmapCollectionHip {
 public:
  struct Entry {
    std::unique_ptr<SubmapHip> map;
    std::size_t id = 0, parentId = 0;
    double origin[3] = {0, 0, 0};   // mapToSubmap_.translation()
    double center[3] = {0, 0, 0};   // submapCenter_ once computed
    bool isCenterComputed = false;
    const double* mapToSubmapCenter() const { return isCenterComputed ? center : origin; }  // Submap.cpp:203-205
  };

  SubmapCollectionHip(const SubmapParams& sp, double mapVoxelSize, const o3s_cropper& mapBuilderCropper, bool isUseInitialMap, int device = 0)
      : params_(sp), voxel_(mapVoxelSize), cropper_(mapBuilderCropper), isUseInitialMap_(isUseInitialMap), device_(device) {
    if (sp.numScansOverlap <= 0) throw std::invalid_argument("Num scan overlap has to be > 0");  // :255
    const double eye[3] = {0, 0, 0};
    createNewSubmap(eye);  // :30, at the default-constructed (identity) pose
    for (int k = 0; k < sp.numScansOverlap + 1; ++k) {
      o3s_scan* sc = nullptr;
      if (o3s_scan_create(device, &sc) != O3S_OK) throw std::runtime_error("o3s_scan_create failed");
      free_.push_back(sc);
    }
  }
  ~SubmapCollectionHip() {
    for (auto& b : buffer_) o3s_scan_destroy(b.scan);
    for (o3s_scan* sc : free_) o3s_scan_destroy(sc);
  }
  SubmapCollectionHip(const SubmapCollectionHip&) = delete;
  SubmapCollectionHip& operator=(const SubmapCollectionHip&) = delete;

  // a scan object to pre-process the next scan into (handed back through insertScan)
  o3s_scan* scanForNextMeasurement() {
    if (free_.empty()) throw std::logic_error("the previous scan was not handed back");
    return free_.back();
  }
  // The same when the caller has pre-processed the next scan into an object of its OWN (another thread, the object's own
  // stream): `filled` takes the place of the ring's free object, which is handed out in exchange — nothing is copied, the ring
  // keeps its size, and the caller owns (and eventually destroys, or fills next) what it gets back.
  o3s_scan* exchangeScanForNextMeasurement(o3s_scan* filled) {
    if (free_.empty()) throw std::logic_error("the previous scan was not handed back");
    if (!filled) throw std::invalid_argument("exchangeScanForNextMeasurement: no scan");
    o3s_scan* spare = free_.back();
    free_.back() = filled;
    return spare;
  }
  // CloudRegistrationParameters::maxRadiusNormalEstimation_ / knnNormalEstimation_ for sweeps that arrive without normals: set on
  // every scan object of the ring (objects handed in through exchangeScanForNextMeasurement are the caller's to configure)
  void setScanNormalEstimation(double maxRadius, int knn) {
    for (auto& b : buffer_)
      if (o3s_scan_set_normal_estimation(b.scan, maxRadius, knn) != O3S_OK) throw std::invalid_argument("o3s_scan_set_normal_estimation");
    for (o3s_scan* sc : free_)
      if (o3s_scan_set_normal_estimation(sc, maxRadius, knn) != O3S_OK) throw std::invalid_argument("o3s_scan_set_normal_estimation");
  }
  std::size_t numSubmaps() const { return submaps_.size(); }
  std::size_t activeSubmapIdx() const { return activeIdx_; }
  SubmapHip& activeSubmap() { return *submaps_[activeIdx_].map; }
  const Entry& submap(std::size_t i) const { return submaps_.at(i); }
  SubmapHip& submapMap(std::size_t i) { return *submaps_.at(i).map; }
  const AdjacencyHip& adjacency() const { return adjacency_; }
  void forceNewSubmapCreationAtNextScan() { isForceNewSubmapCreation_ = true; }
  // SubmapCollection::updateAdjacencyMatrix (:72-78): a loop-closure constraint makes its two submaps adjacent
  void addLoopClosureEdge(std::size_t idA, std::size_t idB) { adjacency_.addEdge(idA, idB); }
  // SubmapCollection::popFinishedSubmapIds (:53-55)
  std::vector<std::pair<std::size_t, double>> popFinishedSubmapIds() {
    std::vector<std::pair<std::size_t, double>> out(finished_.begin(), finished_.end());
    finished_.clear();
    return out;
  }
  bool lastInsertSwitchedSubmaps() const { return lastSwitched_; }
  // where the last switch of submaps spent its time, ms: creating the new object | the closing scan into the previous submap | its
  // centre | retiring it (hand-over / trim) | the buffered scans into the new one
  const double* lastSwitchMs() const { return lastSwitchMs_; }

  // SubmapCollection::insertScan (:193-247).  `scan` is the object returned by scanForNextMeasurement(), pre-processed
  // (its merge cloud is what Submap::insertScan receives as preProcessedScan).  The initial map of the localisation mode
  // (Mapper.cpp:180-183) goes straight into activeSubmap(): that mode never switches submaps, so the buffer is never replayed.
  bool insertScan(o3s_scan* scan, const double mapToRangeSensor[16], double timestamp) {
    lastSwitched_ = false;
    for (int k = 0; k < 16; ++k) mapToRangeSensor_[k] = mapToRangeSensor[k];
    const std::size_t prevActive = activeIdx_;
    // ":201 if (submaps_.empty())" never holds — the constructor has created submap 0 — so the first scan takes the general
    // path like every other: buffered, no switch before minNumRangeData_ scans, inserted into the active submap
    addScanToBuffer(scan, mapToRangeSensor, timestamp);  // :210
    updateActiveSubmap();                                // :213
    if (prevActive != activeIdx_) {                      // :216-239
      lastSwitched_ = true;
      const auto t0 = std::chrono::steady_clock::now();
      insertInto(prevActive, scan, mapToRangeSensor);
      const auto t1 = std::chrono::steady_clock::now();
      computeSubmapCenter(prevActive);
      const auto t2 = std::chrono::steady_clock::now();
      retire(prevActive);  // only now: the closing scan has gone in, the previous submap is no longer inserted into
      const auto t3 = std::chrono::steady_clock::now();
      finished_.emplace_back(prevActive, timestamp);
      numScansMergedInActiveSubmap_ = 0;
      adjacency_.addEdge(submaps_[prevActive].id, submaps_[activeIdx_].id);
      insertBufferedScans(activeIdx_);
      const auto t4 = std::chrono::steady_clock::now();
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      lastSwitchMs_[0] = lastCreateMs_;
      lastSwitchMs_[1] = ms(t0, t1);
      lastSwitchMs_[2] = ms(t1, t2);
      lastSwitchMs_[3] = ms(t2, t3);
      lastSwitchMs_[4] = ms(t3, t4);
      if (submaps_[activeIdx_].map->size() == 0) throw std::logic_error("submap should not be empty after switching");
    } else {
      insertInto(activeIdx_, scan, mapToRangeSensor);  // :243
    }
    ++numScansMergedInActiveSubmap_;
    return true;
  }

 private:
  struct Buffered {
    o3s_scan* scan;
    double T[16];
    double time;
  };

  void insertInto(std::size_t idx, o3s_scan* scan, const double T[16]) {
    const int rc = o3s_submap_insert_processed(submaps_[idx].map->handle(), scan, T);
    if (rc != O3S_OK) throw std::runtime_error("o3s_submap_insert_processed failed (status " + std::to_string(rc) + ")");
  }
  // CircularBuffer::push with a size limit (:82-84): the oldest entry falls out and its scan object becomes free again
  void addScanToBuffer(o3s_scan* scan, const double T[16], double time) {
    if (free_.empty() || free_.back() != scan) throw std::logic_error("insertScan expects the object of scanForNextMeasurement()");
    free_.pop_back();
    Buffered b{scan, {}, time};
    for (int k = 0; k < 16; ++k) b.T[k] = T[k];
    buffer_.push_back(b);
    while ((int)buffer_.size() > params_.numScansOverlap) {
      free_.push_back(buffer_.front().scan);
      buffer_.pop_front();
    }
    if (free_.empty()) throw std::logic_error("scan ring exhausted");
  }
  void insertBufferedScans(std::size_t idx) {  // :86-92 (pops everything, oldest first)
    while (!buffer_.empty()) {
      insertInto(idx, buffer_.front().scan, buffer_.front().T);
      free_.push_back(buffer_.front().scan);
      buffer_.pop_front();
    }
  }
  void computeSubmapCenter(std::size_t idx) {
    Entry& e = submaps_[idx];
    if (o3s_submap_center(e.map->handle(), e.center) != O3S_OK) throw std::runtime_error("o3s_submap_center failed");
    e.isCenterComputed = true;
  }
  static double dist3(const double* a, const double* b) {
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return std::sqrt((dx * dx + dy * dy) + dz * dz);
  }
  // A submap that stops being the active one keeps its map cloud and nothing else: the closed submaps of a long run do not hold a
  // spare array and a work area each (~120 MB per submap at the default limits).  What it held goes to its successor — AFTER the
  // closing scan, which insertScan still puts into it (:216-239; round 4 trimmed it before that insert, which allocated most of it
  // again).  A brand-new successor takes the buffers over as they are (o3s_submap_hand_over: pointers change hands, no hipFree /
  // hipMalloc of the large arrays — 5 - 6 ms on the mapping thread per created submap before); an older submap that is
  // re-activated holds a map already: the previous one is trimmed and the re-activated one sized again (rare: a revisit).
  void retire(std::size_t prev) {
    if (prev >= submaps_.size() || prev == activeIdx_) return;
    SubmapHip& closed = *submaps_[prev].map;
    SubmapHip& next = *submaps_[activeIdx_].map;
    if (next.size() == 0) {
      closed.handOverTo(next);
    } else {
      closed.trim();
      reserveFor(next);
    }
  }
  void reserveFor(SubmapHip& m) {
    // a submap is closed at the scan after it passes maxNumPoints_ (:118-120): with a finite limit its arrays are sized once
    if (params_.maxNumPoints > 0 && params_.maxNumPoints <= kReserveLimit) m.reserve(params_.maxNumPoints + kReserveScanPoints);
  }
  void createNewSubmap(const double origin[3]) {  // :150-162
    const auto c0 = std::chrono::steady_clock::now();
    Entry e;
    e.map = std::make_unique<SubmapHip>(voxel_, cropper_, device_);
    if (submaps_.empty()) reserveFor(*e.map);  // the first submap; every later one inherits its predecessor's buffers (retire)
    e.id = submapId_++;
    e.parentId = activeIdx_;
    for (int a = 0; a < 3; ++a) e.origin[a] = origin[a];
    submaps_.push_back(std::move(e));
    activeIdx_ = submaps_.size() - 1;
    numScansMergedInActiveSubmap_ = 0;
    lastCreateMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
  }
  std::size_t findClosestSubmap(const double p0[3]) const {  // :164-174, std::min_element: the first minimum
    std::size_t best = 0;
    for (std::size_t i = 1; i < submaps_.size(); ++i)
      if (dist3(p0, submaps_[i].mapToSubmapCenter()) < dist3(p0, submaps_[best].mapToSubmapCenter())) best = i;
    return best;
  }
  void updateActiveSubmap() {  // :94-148
    const double* p0 = mapToRangeSensor_ + 12;
    if (isForceNewSubmapCreation_) {
      createNewSubmap(p0);
      isForceNewSubmapCreation_ = false;
      return;
    }
    if (numScansMergedInActiveSubmap_ < params_.minNumRangeData) return;
    if (isUseInitialMap_) return;
    const std::size_t closest = findClosestSubmap(p0);
    const std::size_t active = activeIdx_;
    if (submaps_[active].map->largerThan((std::int64_t)params_.maxNumPoints)) isForceNewSubmapCreation_ = true;  // (size() > maxNumPoints_, :118-120)
    const bool isAnotherSubmapWithinRange = dist3(p0, submaps_[closest].mapToSubmapCenter()) < params_.radius;
    if (isAnotherSubmapWithinRange) {
      if (closest == active) return;
      if (adjacency_.isAdjacent(submaps_[closest].id, submaps_[active].id)) {  // && isSwitchingSubmapsConsistant(...) == true
        activeIdx_ = closest;  // (insertScan retires the previous one once the closing scan is in)
      } else {
        const bool isTraveledSufficientDistance = dist3(p0, submaps_[active].mapToSubmapCenter()) > params_.radius;
        if (isTraveledSufficientDistance) createNewSubmap(p0);
      }
    } else {
      createNewSubmap(p0);
    }
  }

  static constexpr std::int64_t kReserveLimit = 8000000;      // larger limits mean "no limit": the arrays then double as the map grows
  static constexpr std::int64_t kReserveScanPoints = 262144;  // the scan that takes the map over the limit (2 x a 64 x 2048 sweep)
  SubmapParams params_;
  double voxel_;
  o3s_cropper cropper_;
  bool isUseInitialMap_;
  int device_;
  std::vector<Entry> submaps_;
  std::size_t activeIdx_ = 0, submapId_ = 0;
  int numScansMergedInActiveSubmap_ = 0;
  bool isForceNewSubmapCreation_ = false, lastSwitched_ = false;
  double lastSwitchMs_[5] = {0, 0, 0, 0, 0}, lastCreateMs_ = 0.0;
  double mapToRangeSensor_[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::deque<Buffered> buffer_;
  std::vector<o3s_scan*> free_;
  std::deque<std::pair<std::size_t, double>> finished_;
  AdjacencyHip adjacency_;
};

}  // namespace o3s
