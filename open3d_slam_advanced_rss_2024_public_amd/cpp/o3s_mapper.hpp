// o3s_mapper.hpp — header-only C++17 restatement of the caller glue around the two ICP calls:
// o3d_slam::Mapper::addRangeMeasurement (open3d_slam/src/Mapper.cpp:168-504), with every cloud operation on the device
// (o3s_scan / o3s_submap / o3s_icp over the C ABI).  It is what a catkin package would compile instead of the reference's
// Mapper.cpp body; no Eigen / Open3D / libpointmatcher headers are needed.  Line numbers below are Mapper.cpp's.
//
//   :168-174      no calibration set (and no initial map): return false
//   :176          submaps_->setMapToRangeSensor(mapToRangeSensor_)
//   :179-195      first scan: pre-process, insert at the given pose, push the pose buffers
//   :197-235      out-of-order timestamp: propagate the previous pose by the odometry motion, no registration
//   :237-262      odometry availability (a pose within 100 ms of the buffer's latest counts as available)
//   :265-281      prior = mapToRangeSensorPrev_ * ((odomPrev * C^-1)^-1 * (odomNow * C^-1)), C = calibration_ (:221-222, :270-273),
//                 unless isNewValueSetMapper_ / isIgnoreOdometryPrediction_
//   :305-318, :359-376, :382-411, :481-501   the four stage stopwatches (o3d_slam::Timer): lastTimings() / meanTimings()
//   :307-309      processForScanMatchingAndMerging + open3dToPointmatcher        -> o3s_scan_preprocess + o3s_scan_set_reading
//   :323          prior cast to float (PmTfParameters)
//   :328-336      cropSubmap(activeSubmap, mapToRangeSensor_); empty patch -> return false
//   :346-366      every referenceCloudSettingPeriod_ seconds (or after a pose reset): open3dToPointmatcher(patch) +
//                 icp_.initReference                                             -> o3s_submap_set_reference
//   :393          icp_.compute(reading, {}, prior, false)                         -> o3s_icp_compute_resident
//   :420-422      catch (std::runtime_error): keep the prior
//   :435          result cast back to double
//   :440-455      isNewValueSetMapper_: adopt the GIVEN pose, skip this scan's result and the insert, ignore odometry next time
//   :465-479      initial-map mode: no merging (or not before mapMergeDelayInSeconds_)
//   :483-489      insert the merge cloud if the sensor moved at least minMovementBetweenMappingSteps_
// 4x4 matrices are column-major doubles (Eigen::Matrix4d::data()).  Isometry products / inverses are restated as plain
// k = 0..3 accumulations (Eigen is not part of the tree: its evaluation order is not pinned).
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>

#include "o3s_icp.hpp"
#include "o3s_scan.h"
#include "o3s_submap_collection.hpp"

namespace o3s {

struct Mat4 {
  double m[16];
  static Mat4 identity() {
    Mat4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
  }
  double& operator()(int r, int c) { return m[c * 4 + r]; }
  double operator()(int r, int c) const { return m[c * 4 + r]; }
};
inline Mat4 mul(const Mat4& A, const Mat4& B) {
  Mat4 C{};
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = A(r, 0) * B(0, c);
      s = s + A(r, 1) * B(1, c);
      s = s + A(r, 2) * B(2, c);
      s = s + A(r, 3) * B(3, c);
      C(r, c) = s;
    }
  return C;
}
// Eigen::Isometry3d::inverse(): [R^T, -R^T t]
inline Mat4 inverse_isometry(const Mat4& T) {
  Mat4 R = Mat4::identity();
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R(r, c) = T(c, r);
  for (int r = 0; r < 3; ++r) {
    double s = R(r, 0) * T(0, 3);
    s = s + R(r, 1) * T(1, 3);
    s = s + R(r, 2) * T(2, 3);
    R(r, 3) = -s;
  }
  return R;
}

// o3d_slam::TransformInterpolationBuffer restricted to what the Mapper asks of it when the odometry is sampled at the
// scan stamps: has(t), latest_time(), lookup(t) (exact stamp; the interpolation itself is a host utility, out of scope)
class PoseBuffer {
 public:
  void push(double t, const Mat4& T) { poses_[t] = T; }
  bool has(double t) const { return poses_.count(t) != 0; }
  bool empty() const { return poses_.empty(); }
  double latest_time() const { return poses_.rbegin()->first; }
  const Mat4& lookup(double t) const {
    auto it = poses_.find(t);
    if (it == poses_.end()) throw std::runtime_error("PoseBuffer: no pose at the requested stamp");
    return it->second;
  }

 private:
  std::map<double, Mat4> poses_;
};

// The Mapper's four stopwatches (Mapper.cpp:305-318 "Auxilary time", :359-376 "Reference Cloud Re-init time", :382-411
// "Scan2Map Registration", :481-501 "Scan Insertion"), wall-clock milliseconds like o3d_slam::Timer; 0 for a stage a scan skipped
struct MapperTimings {
  double auxiliaryMs = 0.0, referenceInitMs = 0.0, registrationMs = 0.0, insertionMs = 0.0;
};

struct MapperParams {
  double scanVoxelSize = 0.1;                        // scanProcessing_.voxelSize_
  double mapVoxelSize = 0.1;                         // mapBuilder_.mapVoxelSize_
  o3s_cropper mapBuilderCropper{};                   // wide crop of the raw scan / volume the map is re-voxelised in
  o3s_cropper scanMatcherCropper{};                  // narrow crop of the scan / of the map patch
  double referenceCloudSettingPeriod = 1.0;          // scanMatcher_.icp_.referenceCloudSettingPeriod_ (Parameters.hpp:71)
  double minMovementBetweenMappingSteps = 0.0;       // Parameters.hpp
  bool isUseInitialMap = false;
  bool isMergeScansIntoMap = true;
  double mapMergeDelayInSeconds = 0.0;
  SubmapParams submaps;                              // submaps_ (Parameters.hpp:103-109): radius_, minNumRangeData_, ...
};

class MapperHip {
 public:
  MapperHip(const MapperParams& p, const o3s_icp_config& icpCfg, int device = 0)
      : params_(p), icp_(icpCfg, device), submaps_(p.submaps, p.mapVoxelSize, p.mapBuilderCropper, p.isUseInitialMap, device) {}
  MapperHip(const MapperHip&) = delete;
  MapperHip& operator=(const MapperHip&) = delete;

  // Mapper::setMapToRangeSensorInitial / setMapToRangeSensor with a new value (Mapper.cpp:96-118): the next scan adopts it
  void setMapToRangeSensorInitial(const Mat4& T) {
    mapToRangeSensor_ = T;
    mapToRangeSensorPrev_ = T;
    isNewValueSetMapper_ = true;
  }
  void setMapToRangeSensor(const Mat4& T) { mapToRangeSensor_ = T; }  // first-scan pose (no flag: Mapper.cpp:179-195)
  // Mapper::loopClosureUpdate (Mapper.cpp:93-96)
  void loopClosureUpdate(const Mat4& loopClosureCorrection) {
    mapToRangeSensor_ = mul(loopClosureCorrection, mapToRangeSensor_);
    mapToRangeSensorPrev_ = mul(loopClosureCorrection, mapToRangeSensorPrev_);
  }
  void addOdometryPose(double t, const Mat4& odomToRangeSensor) { odomToRangeSensorBuffer_.push(t, odomToRangeSensor); }
  // Mapper::setExternalOdometryFrameToCloudFrameCalibration (Mapper.cpp:66-85): the odometry poses are those of ANOTHER frame
  // (the tracking camera / IMU); every lookup is multiplied by calibration^-1 before the motion is formed (:221-222, :270-273)
  void setCalibration(const Mat4& odometryFrameToCloudFrame) {
    calibrationInv_ = inverse_isometry(odometryFrameToCloudFrame);
    isCalibrationSet_ = true;
  }
  bool isCalibrationSet() const { return isCalibrationSet_; }
  const MapperTimings& lastTimings() const { return lastTimings_; }
  MapperTimings meanTimings() const {  // Timer::getAvgMeasurementMsec over the scans that ran the stage
    MapperTimings m;
    m.auxiliaryMs = nTimed_[0] ? sumTimings_.auxiliaryMs / nTimed_[0] : 0.0;
    m.referenceInitMs = nTimed_[1] ? sumTimings_.referenceInitMs / nTimed_[1] : 0.0;
    m.registrationMs = nTimed_[2] ? sumTimings_.registrationMs / nTimed_[2] : 0.0;
    m.insertionMs = nTimed_[3] ? sumTimings_.insertionMs / nTimed_[3] : 0.0;
    return m;
  }
  // initial map for the localisation mode (isUseInitialMap_): Mapper.cpp:180-183 inserts it as the first "scan"
  SubmapHip& activeSubmap() { return submaps_.activeSubmap(); }
  SubmapCollectionHip& submaps() { return submaps_; }
  IcpHip& icp() { return icp_; }
  const Mat4& mapToRangeSensor() const { return mapToRangeSensor_; }
  const Mat4& lastPrior() const { return lastPrior_; }
  bool lastScanInserted() const { return lastInserted_; }
  bool lastReferenceReset() const { return lastReferenceReset_; }
  bool lastIcpThrew() const { return lastIcpThrew_; }
  int lastIterations() const { return lastIterations_; }

  // rawScan in the sensor frame (3 x N doubles, normals nullable when normal estimation is configured on the scan object)
  bool addRangeMeasurement(const double* rawPts, const double* rawNormals, std::int64_t N, double timestamp) {
    return add(Raw{rawPts, rawNormals, N, nullptr, nullptr}, timestamp);
  }
  // The same with the raw sweep staged in HBM beforehand (o3s_raw_scan_upload, typically from the thread that receives the
  // sweeps — the reference's mapping worker is fed from a buffer too, SlamWrapper.cpp:660-709): the host-to-device copy of
  // sweep k + 1 then runs while this thread still registers sweep k.
  bool addRangeMeasurement(const o3s_raw_scan* staged, double timestamp) {
    if (!staged) throw std::runtime_error("addRangeMeasurement: no staged scan");
    return add(Raw{nullptr, nullptr, o3s_raw_scan_size(staged), staged, nullptr}, timestamp);
  }
  // The same with the sweep already PRE-PROCESSED by the receiving thread: o3s_scan_preprocess with THIS mapper's croppers and
  // scan voxel size (params()) into a scan object of the caller's, on that object's own stream, while this thread still
  // registers and inserts sweep k.  The pre-processing does not depend on the pose (both croppers sit at the sensor,
  // ScanToMapRegistration.cpp:36-69), so the result is the same bits; the "Auxilary time" stage of this call shrinks to a
  // pointer exchange.  On return `preprocessed` is a spare object to fill next (the filled one has joined the ring of
  // resident scans the submap collection keeps for its overlap buffer); when the call returns before the pre-processing
  // stage — no calibration yet, an out-of-order stamp — it is left as it was.
  bool addRangeMeasurement(o3s_scan*& preprocessed, double timestamp) {
    if (!preprocessed) throw std::runtime_error("addRangeMeasurement: no pre-processed scan");
    return add(Raw{nullptr, nullptr, 0, nullptr, &preprocessed}, timestamp);
  }
  const MapperParams& params() const { return params_; }

 private:
  struct Raw {
    const double* pts;
    const double* normals;
    std::int64_t N;
    const o3s_raw_scan* staged;
    o3s_scan** ready;  // a scan object the caller has already pre-processed this sweep into
  };
  bool add(const Raw& raw, double timestamp) {
    lastInserted_ = lastReferenceReset_ = lastIcpThrew_ = false;
    lastTimings_ = MapperTimings{};
    if (!params_.isUseInitialMap && !isCalibrationSet_) return false;  // "Calibration is not set. Returning from mapping." (:169-174)
    scan_ = submaps_.scanForNextMeasurement();  // stays ours until submaps_.insertScan takes it into its overlap buffer
    // ---- first scan (:179-195) ----
    if (submaps_.activeSubmap().empty()) {
      if (params_.isUseInitialMap) {  // the raw "scan" IS the map: inserted as is (:181-183)
        if (raw.staged || raw.ready) throw std::runtime_error("the initial map is handed over as host arrays");
        submaps_.activeSubmap().insertScan(raw.pts, raw.normals, raw.N, mapToRangeSensor_.m);
      } else {
        mapToRangeSensorPrev_ = mapToRangeSensor_;
        preprocess(raw);
        submaps_.insertScan(scan_, mapToRangeSensor_.m, timestamp);
        lastInserted_ = true;
      }
      return true;
    }
    // ---- out-of-order stamp (:197-235): propagate by the odometry motion, no registration ----
    if (haveLast_ && timestamp <= lastMeasurementTimestamp_) {
      const double latest = odomToRangeSensorBuffer_.latest_time();
      const Mat4 motion = mul(inverse_isometry(odomInCloudFrame(lastMeasurementTimestamp_)), odomInCloudFrame(latest));
      mapToRangeSensor_ = mul(mapToRangeSensorPrev_, motion);
      mapToRangeSensorPrev_ = mapToRangeSensor_;
      return true;
    }
    // ---- odometry prior (:237-281) ----
    bool isOdomOkay = odomToRangeSensorBuffer_.has(timestamp);
    if (!odomToRangeSensorBuffer_.empty() && (timestamp - odomToRangeSensorBuffer_.latest_time()) * 1e3 < 100.0) isOdomOkay = true;
    Mat4 estimate = mapToRangeSensorPrev_;
    if (isOdomOkay && haveLast_ && !isNewValueSetMapper_ && !isIgnoreOdometryPrediction_ && odomToRangeSensorBuffer_.has(timestamp) &&
        odomToRangeSensorBuffer_.has(lastMeasurementTimestamp_)) {
      const Mat4 motion = mul(inverse_isometry(odomInCloudFrame(lastMeasurementTimestamp_)), odomInCloudFrame(timestamp));
      estimate = mul(mapToRangeSensorPrev_, motion);
    }
    isIgnoreOdometryPrediction_ = false;
    lastPrior_ = estimate;
    // ---- pre-processing on the device (:307-309), under the "Auxilary time" stopwatch (:305-312) ----
    auto t0 = Clock::now();
    preprocess(raw);
    stamp(t0, lastTimings_.auxiliaryMs, sumTimings_.auxiliaryMs, 0);
    float prior32[16], corrected32[16];
    for (int k = 0; k < 16; ++k) corrected32[k] = prior32[k] = (float)estimate.m[k];  // :323, :338
    // ---- map patch + reference (:328-366) ----
    o3s_cropper patch = params_.scanMatcherCropper;
    for (int a = 0; a < 3; ++a) patch.centre[a] = mapToRangeSensor_(a, 3);  // cropSubmap: setPose(mapToRangeSensor_)
    const bool resetRef = isNewValueSetMapper_ || !haveRef_ || (timestamp - lastReferenceInitializationTimestamp_) >= params_.referenceCloudSettingPeriod;
    // the reference crops the submap on EVERY scan and gives the scan up when the patch is empty (:328-336) — also between two
    // renewals of the ICP reference (a patch emptied by carving or a submap switch must not be registered against a stale index).
    // On those scans the count needs the map as the previous insert left it, the registration does not (it runs against the index
    // of an earlier renewal): the chain is put on the GPU first (o3s_icp_compute_resident_launch), the patch is counted while it
    // runs — completing the pending insert —, and an empty patch gives the scan up exactly as before, the finished chain unused.
    std::int64_t nPatchNow = -1;
    try {
      if (resetRef) {  // "Reference Cloud Re-init time" (:359-370)
        t0 = Clock::now();
        std::int64_t nPatch = 0;
        if (!submaps_.activeSubmap().setReference(patch, mapToRangeSensor_.m, icp_, &nPatch)) return false;  // "Map patch is empty" / initReference failed
        lastReferenceInitializationTimestamp_ = timestamp;
        haveRef_ = true;
        lastReferenceReset_ = true;
        stamp(t0, lastTimings_.referenceInitMs, sumTimings_.referenceInitMs, 1);
      }
      t0 = Clock::now();  // "Scan2Map Registration" (:382-405)
      check(o3s_scan_set_reading(scan_, icp_.handle()), "o3s_scan_set_reading");
      o3s_icp_stats st{};
      int rc;
      if (resetRef) {
        rc = o3s_icp_compute_resident(icp_.handle(), prior32, corrected32, &st);
      } else {
        const int rl = o3s_icp_compute_resident_launch(icp_.handle(), prior32);
        try {
          nPatchNow = submaps_.activeSubmap().patchCount(patch, mapToRangeSensor_.m);
        } catch (...) {
          if (rl == O3S_OK) (void)o3s_icp_compute_resident_finish(icp_.handle(), corrected32, &st);
          throw;
        }
        rc = rl == O3S_OK ? o3s_icp_compute_resident_finish(icp_.handle(), corrected32, &st) : rl;
        if (nPatchNow == 0) {  // "Map patch is empty": the scan is given up (:333-336); nothing of the registration is kept
          (void)o3s_icp_synchronize(icp_.handle());
          return false;
        }
      }
      lastIterations_ = st.iterations;
      if (rc != O3S_OK) throw std::runtime_error(o3s_last_error(icp_.handle()));  // every libpointmatcher exception derives from it
      stamp(t0, lastTimings_.registrationMs, sumTimings_.registrationMs, 2);
    } catch (const std::runtime_error&) {
      // (a scan whose reading could not even be handed over: the reference would have looked at the patch first)
      if (!resetRef && nPatchNow < 0 && submaps_.activeSubmap().patchCount(patch, mapToRangeSensor_.m) == 0) return false;
      lastIcpThrew_ = true;  // :420-422: the prior stays (corrected32 must not hold a half-written result)
      for (int k = 0; k < 16; ++k) corrected32[k] = prior32[k];
      // a compute that failed before it waited for its stream may leave the asynchronous index build / the reading's hand-over in
      // flight: nothing below may rewrite the buffers they read until the handle's stream has drained
      (void)o3s_icp_synchronize(icp_.handle());
    }
    Mat4 corrected{};
    for (int k = 0; k < 16; ++k) corrected.m[k] = (double)corrected32[k];  // :435
    // ---- pose reset (:440-455) ----
    if (isNewValueSetMapper_) {  // the GIVEN pose is adopted; lastMeasurementTimestamp_ is left as it was (:440-455)
      initTime_ = timestamp;
      mapToRangeSensorPrev_ = mapToRangeSensor_;
      isNewValueSetMapper_ = false;
      isIgnoreOdometryPrediction_ = true;
      return true;
    }
    mapToRangeSensor_ = corrected;
    // ---- localisation mode: no merging (:465-479) ----
    const double timeSinceInit = timestamp - initTime_;
    if ((params_.isUseInitialMap && !params_.isMergeScansIntoMap) ||
        (timeSinceInit < params_.mapMergeDelayInSeconds && params_.isUseInitialMap && params_.isMergeScansIntoMap)) {
      lastMeasurementTimestamp_ = timestamp;
      haveLast_ = true;
      mapToRangeSensorPrev_ = mapToRangeSensor_;
      return true;
    }
    // ---- insert (:483-489), under the "Scan Insertion" stopwatch (:481-495) ----
    t0 = Clock::now();
    const Mat4 motion = mul(inverse_isometry(mapToRangeSensorLastScanInsertion_), mapToRangeSensor_);
    const double moved = std::sqrt(motion(0, 3) * motion(0, 3) + motion(1, 3) * motion(1, 3) + motion(2, 3) * motion(2, 3));
    if (!(moved < params_.minMovementBetweenMappingSteps)) {
      submaps_.insertScan(scan_, mapToRangeSensor_.m, timestamp);  // :487 submaps_->insertScan(rawScan, *mergeScan, mapToRangeSensor_, timestamp)
      mapToRangeSensorLastScanInsertion_ = mapToRangeSensor_;
      lastInserted_ = true;
    }
    lastMeasurementTimestamp_ = timestamp;
    haveLast_ = true;
    mapToRangeSensorPrev_ = mapToRangeSensor_;
    stamp(t0, lastTimings_.insertionMs, sumTimings_.insertionMs, 3);
    return true;
  }

  using Clock = std::chrono::steady_clock;
  void stamp(const Clock::time_point& t0, double& last, double& sum, int which) {
    last = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
    sum += last;
    nTimed_[which] += 1;
  }
  // getTransform(t, odomToRangeSensorBuffer_) * calibration_.inverse()   (:221-222, :270-273)
  Mat4 odomInCloudFrame(double t) const { return mul(odomToRangeSensorBuffer_.lookup(t), calibrationInv_); }
  void preprocess(const Raw& raw) {
    std::int64_t nMerge = 0, nMatch = 0;
    if (raw.ready) {
      o3s_scan* filled = *raw.ready;
      *raw.ready = submaps_.exchangeScanForNextMeasurement(filled);
      scan_ = filled;
    } else if (raw.staged)
      check(o3s_scan_preprocess_staged(scan_, &params_.mapBuilderCropper, params_.scanVoxelSize, &params_.scanMatcherCropper, raw.staged, &nMerge, &nMatch),
            "o3s_scan_preprocess_staged");
    else
      check(o3s_scan_preprocess(scan_, &params_.mapBuilderCropper, params_.scanVoxelSize, &params_.scanMatcherCropper, raw.pts, raw.normals, raw.N, &nMerge,
                                &nMatch),
            "o3s_scan_preprocess");
  }
  static void check(int rc, const char* what) {
    if (rc != O3S_OK) throw std::runtime_error(std::string(what) + " failed (status " + std::to_string(rc) + ")");
  }

  MapperParams params_;
  IcpHip icp_;
  SubmapCollectionHip submaps_;
  o3s_scan* scan_ = nullptr;  // owned by submaps_ (its ring of resident scans)
  PoseBuffer odomToRangeSensorBuffer_;
  Mat4 mapToRangeSensor_ = Mat4::identity(), mapToRangeSensorPrev_ = Mat4::identity(), lastPrior_ = Mat4::identity();
  Mat4 mapToRangeSensorLastScanInsertion_ = Mat4::identity();
  double lastMeasurementTimestamp_ = 0.0, lastReferenceInitializationTimestamp_ = 0.0, initTime_ = 0.0;
  bool haveLast_ = false, haveRef_ = false;  // haveLast_: lastMeasurementTimestamp_ holds a stamp (the reference leaves it default-constructed after the first scan, whose lookup would throw: no odometry prior is formed then)
  bool isNewValueSetMapper_ = false, isIgnoreOdometryPrediction_ = false;
  bool lastInserted_ = false, lastReferenceReset_ = false, lastIcpThrew_ = false;
  int lastIterations_ = 0;
  Mat4 calibrationInv_ = Mat4::identity();
  bool isCalibrationSet_ = false;
  MapperTimings lastTimings_, sumTimings_;
  long long nTimed_[4] = {0, 0, 0, 0};
};

}  // namespace o3s
