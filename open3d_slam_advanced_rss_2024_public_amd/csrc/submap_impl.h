// submap_impl.h — device-resident active submap (C ABI: include/o3s_submap.h), gfx950 only.  Included at the end of
// cloud_ops.hip so that both share one instantiation of the kernels and of the rocPRIM sort / scan in cloud_dev.h.
#pragma once
#include "../../include/o3s_scan.h"
#include "../../include/o3s_submap.h"

#include "cloud_dev.h"
#include "normals_dev.h"

namespace {

// grow-only device array
struct DArr {
  void* p = nullptr;
  size_t cap = 0;
  ~DArr() {
    if (p) (void)hipFree(p);
  }
  // keeps the first `keep` bytes when it has to move
  hipError_t ensure(size_t bytes, size_t keep, hipStream_t s) {
    if (bytes <= cap) return hipSuccess;
    size_t want = bytes + bytes / 2 + 4096;
    if (cap && want < 2 * cap) want = 2 * cap;  // a map that grows: each move is a device-wide stall of several milliseconds
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) return e;
    if (p && keep) {
      e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (p) (void)hipFree(p);
    p = q;
    cap = want;
    return e;
  }
  double* d() const { return reinterpret_cast<double*>(p); }
};

// o3d_slam::transform (helpers.cpp:283-318): p' = (T [p 1]).head<3>() / w, n' = (T [n 0]).head<3>(); the 4-term
// products are accumulated k = 0..3 in fp64 without contraction
struct Mat4d {  // a pose as a kernel argument: no 128-byte host-to-device copy in front of the launch
  double m[16];
};
__device__ __forceinline__ void transform_append_one(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t i, const double* T,
                                                     double* __restrict__ out_pts, double* __restrict__ out_n);
__global__ void __launch_bounds__(kB) k_transform_append_v(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t N, Mat4d Tm,
                                                           double* __restrict__ out_pts, double* __restrict__ out_n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  transform_append_one(pts, nrm, i, Tm.m, out_pts, out_n);
}
__global__ void __launch_bounds__(kB) k_transform_append(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t N,
                                                         const double* __restrict__ Tm /*16, column-major*/, double* __restrict__ out_pts,
                                                         double* __restrict__ out_n) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  double T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = Tm[k];
  transform_append_one(pts, nrm, i, T, out_pts, out_n);
}
__device__ __forceinline__ void transform_append_one(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t i, const double* T,
                                                     double* __restrict__ out_pts, double* __restrict__ out_n) {
  const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  double v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s = T[r] * x;
    s = s + T[4 + r] * y;
    s = s + T[8 + r] * z;
    s = s + T[12 + r] * 1.0;
    v[r] = s;
  }
  out_pts[3 * i] = v[0] / v[3];
  out_pts[3 * i + 1] = v[1] / v[3];
  out_pts[3 * i + 2] = v[2] / v[3];
  if (nrm) {
    const double a = nrm[3 * i], b = nrm[3 * i + 1], c = nrm[3 * i + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double s = T[r] * a;
      s = s + T[4 + r] * b;
      s = s + T[8 + r] * c;
      s = s + T[12 + r] * 0.0;
      out_n[3 * i + r] = s;
    }
  }
}

// o3d_slam::transform (helpers.cpp:283-318) on host buffers: points (/ w), normals, covariances R C R^T; an (almost-)identity
// T returns the cloud TWICE (the copy of :285-288 followed by the loop's appends), which is what the reference does.
__global__ void __launch_bounds__(kB) k_transform_cov(const double* __restrict__ cov, int64_t N, const double* __restrict__ Tm, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  double R[3][3], C[3][3], RC[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      R[r][c] = Tm[c * 4 + r];
      C[r][c] = cov[9 * i + c * 3 + r];  // Eigen::Matrix3d is column-major
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = R[r][0] * C[0][c];
      s = s + R[r][1] * C[1][c];
      s = s + R[r][2] * C[2][c];
      RC[r][c] = s;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = RC[r][0] * R[c][0];
      s = s + RC[r][1] * R[c][1];
      s = s + RC[r][2] * R[c][2];
      out[9 * i + c * 3 + r] = s;
    }
}

// voxel key of the carving VoxelMap: getVoxelIdx(p, 1 / voxel) (VoxelHashMap.hpp:48-51), packed relative to the subset's
// index box; points outside the subset get `out_key` — the next power of two above every packed key — and sort last
__global__ void __launch_bounds__(kB) k_carve_keys(const double* __restrict__ pts, int64_t N, const uint32_t* __restrict__ inflag, double inv,
                                                   int32_t x0, int32_t y0, int32_t z0, uint64_t ex, uint64_t ey, uint64_t out_key,
                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  vals[i] = (uint32_t)i;
  if (!inflag[i]) {
    keys[i] = out_key;
    return;
  }
  const int64_t x = (int64_t)(int32_t)floor(pts[3 * i] * inv) - x0, y = (int64_t)(int32_t)floor(pts[3 * i + 1] * inv) - y0,
                z = (int64_t)(int32_t)floor(pts[3 * i + 2] * inv) - z0;
  keys[i] = ((uint64_t)z * ey + (uint64_t)y) * ex + (uint64_t)x;
}
__global__ void __launch_bounds__(kB) k_carve_box(const double* __restrict__ pts, int64_t N, const uint32_t* __restrict__ inflag, double inv,
                                                  int32_t* __restrict__ mm_slots /*[kExtSlots][min[3], max[3]]*/) {
  int32_t* mm = mm_slots + 6 * (blockIdx.x & (kExtSlots - 1));
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  const bool live = i < N && inflag[i];
  for (int a = 0; a < 3; ++a) {
    const int32_t v = live ? (int32_t)floor(pts[3 * i + a] * inv) : 0;
    const int32_t lo = wave_min_i32(live ? v : INT32_MAX), hi = wave_max_i32(live ? v : INT32_MIN);
    if ((threadIdx.x & 63) == 0 && lo <= hi) {  // a (possibly stale) look first: extrema are monotone, so skipping is safe
      if (lo < __atomic_load_n(&mm[a], __ATOMIC_RELAXED)) atomicMin(&mm[a], lo);
      if (hi > __atomic_load_n(&mm[3 + a], __ATOMIC_RELAXED)) atomicMax(&mm[3 + a], hi);
    }
  }
}
// getIdxsOfCarvedPoints (helpers.cpp:252-281): one lane per ray, the same sequential march as the reference
__global__ void __launch_bounds__(kB) k_carve_rays(const double* __restrict__ scan /*map frame*/, int64_t Ns, double sx, double sy, double sz,
                                                   double voxel, double inv, double max_len, double trunc, double min_dot,
                                                   const uint64_t* __restrict__ keys /*sorted*/, const uint32_t* __restrict__ vals, int64_t n_in,
                                                   int32_t x0, int32_t y0, int32_t z0, int64_t ex, int64_t ey, int64_t ez,
                                                   const double* __restrict__ map_n /*nullable*/, uint32_t* __restrict__ remove) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= Ns) return;
  const double dx = scan[3 * i] - sx, dy = scan[3 * i + 1] - sy, dz = scan[3 * i + 2] - sz;
  const double length = sqrt((dx * dx + dy * dy) + dz * dz);
  if (!(length > 0.0)) return;  // a return at the sensor origin: the reference's NaN positions never find a voxel
  const double ux = dx / length, uy = dy / length, uz = dz / length;
  double distance = 0.0;
  const double max_path = fmax(voxel, fmin(length - trunc, max_len));
  while (distance < max_path) {  // NaN lengths (a return at the sensor origin) make this false, like the reference
    const double cx = distance * ux + sx, cy = distance * uy + sy, cz = distance * uz + sz;
    const int64_t x = (int64_t)(int32_t)floor(cx * inv) - x0, y = (int64_t)(int32_t)floor(cy * inv) - y0, z = (int64_t)(int32_t)floor(cz * inv) - z0;
    if (x >= 0 && x < ex && y >= 0 && y < ey && z >= 0 && z < ez) {
      const uint64_t key = ((uint64_t)z * (uint64_t)ey + (uint64_t)y) * (uint64_t)ex + (uint64_t)x;
      int64_t lo = 0, hi = n_in;  // lower_bound in the sorted subset keys
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
      }
      for (int64_t j = lo; j < n_in && keys[j] == key; ++j) {
        const uint32_t id = vals[j];
        bool rm = true;
        if (map_n) {
          double nx = map_n[3 * (size_t)id], ny = map_n[3 * (size_t)id + 1], nz = map_n[3 * (size_t)id + 2];
          const double zz = (nx * nx + ny * ny) + nz * nz;  // Eigen normalized()
          if (zz > 0) {
            const double s = sqrt(zz);
            nx /= s;
            ny /= s;
            nz /= s;
          }
          rm = fabs((ux * nx + uy * ny) + uz * nz) > min_dot;
        }
        if (rm) remove[id] = 1u;
      }
    }
    distance += voxel;
  }
}
// Submap::computeSubmapCenter = open3d PointCloud::GetCenter(): the mean of the map points.  Per-block fp64 partial sums in a
// fixed order (thread-strided, wave reduction, then the four waves), folded by block 0's caller on the host in block order.
__global__ void __launch_bounds__(kB) k_center_part(const double* __restrict__ pts, int64_t N, double* __restrict__ part /*[grid][3]*/) {
  double s[3] = {0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < N; i += (int64_t)gridDim.x * kB)
    for (int a = 0; a < 3; ++a) s[a] += pts[3 * i + a];
  __shared__ double sh[kB / 64][3];
  for (int a = 0; a < 3; ++a) {
    double v = s[a];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_down(v, m, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][a] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double v = sh[0][threadIdx.x];
    for (int w = 1; w < kB / 64; ++w) v += sh[w][threadIdx.x];
    part[(size_t)blockIdx.x * 3 + threadIdx.x] = v;
  }
}
__global__ void __launch_bounds__(kB) k_invert_flags(const uint32_t* __restrict__ remove, int64_t N, uint32_t* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i < N) keep[i] = remove[i] ? 0u : 1u;
}

}  // namespace

// csrc/o3s_icp.hip (same library, not part of the C ABI): index build over a reference whose point count is a word on the device
extern "C" int o3s_icp_init_reference_dev_counted_async(o3s_icp* h, const void* d_xyzw, const void* d_normals, const uint32_t* d_count, int64_t max_M,
                                                        int64_t* M_out);

struct o3s_submap {
  int device = 0;
  double voxel = 0.0;
  o3s_cropper cropper{};
  hipStream_t stream = nullptr;
  bool owns_stream = true;        // false: `stream` is one of the device's shared submap streams (submap_stream_pool)
  hipEvent_t handover = nullptr;  // recorded on `stream`, waited for by the ICP handle's stream (o3s_icp_wait_event)
  DArr pts[2], nrm[2];  // ping-pong: voxelisation reads [cur] and writes [1 - cur]
  int cur = 0;
  int64_t n = 0;
  // layout of the map array after a voxelisation: [n_pt pass-through points | one point per voxel in key order]; valid only while
  // nothing else has rewritten the array (voxel_insert_merge_dev relies on it, and checks it on the device)
  int64_t n_pt = 0;
  bool layout_valid = false;
  int merge_backoff = 0;              // inserts left before the merge path is tried again after it had to give way to the sort
  int64_t n_merged = 0, n_sorted = 0, n_fell_back = 0;  // how the voxelising inserts ran (o3s_submap_insert_stats)
  int has_normals = -1;  // -1: undecided (empty map)
  DArr col[2];           // colours of the map cloud (open3d PointCloud::colors_), ping-pong like the points
  int has_colors = 0;    // 1 while the map carries one colour per point (PointCloud::HasColors())
  DArr scan_c;
  DArr scan_p, scan_n, carve_scan, d_T, patch_xyzw, patch_n32;
  Arena arena;
  // An insert whose completion is looked at later (o3s_submap_insert_processed): everything is enqueued, the merge path's counts and
  // verdict are on their way to a mailbox slot, and `n`, `cur`, `n_pt` still describe the map BEFORE the insert.  submap_settle() —
  // the first thing every entry point does — fetches them (waiting only if the GPU has not got there yet), runs the sort-based
  // pipeline if the merge had to give way, and commits the new state.
  struct PendingInsert {
    bool active = false;
    LazyPost post;
    int c = 0;
    int64_t n_tmp = 0;
    bool hn = false;
    VoxHint vh{};
  } pend;
  DArr d_post;                      // 4 words the pending insert's counts also go to (lazy_post_fetch's fallback)
  hipEvent_t scan_read = nullptr;   // recorded when an insert has read its device scan: the scan's stream waits for it before the object is rewritten
};

namespace {
int set_dev(const o3s_submap* m) { return hipSetDevice(m->device) == hipSuccess ? O3S_OK : O3S_ERR_HIP; }
int submap_settle(o3s_submap* m);  // completes a pending insert (defined with insert_dev below)
inline int submap_settle(const o3s_submap* m) { return submap_settle(const_cast<o3s_submap*>(m)); }

// Creating a HIP stream makes a hardware queue: 2.6 - 3.5 ms on the calling thread — the mapping thread, every time
// SubmapCollection::createNewSubmap runs (it was the largest part of a switch of submaps once the buffers changed hands instead of
// being freed and made again).  The submaps of a device therefore share a few streams, made on first use and kept for the life of
// the process, dealt round-robin: two submaps on one stream only ever order their work behind each other (the mapper works on one
// submap at a time), which no call's semantics depends on.  Snapshots made for a worker thread (o3s_submap_clone) get a stream of
// their own: they are there to run beside the mapper.
struct SubmapStreamPool {
  static constexpr int kPerDevice = 4;
  std::mutex m;
  std::vector<std::vector<hipStream_t>> streams;  // [device][k]
  std::vector<unsigned> next;
  hipStream_t get(int device) {  // the device is current
    std::lock_guard<std::mutex> g(m);
    if ((size_t)device >= streams.size()) {
      streams.resize((size_t)device + 1);
      next.resize((size_t)device + 1, 0u);
    }
    std::vector<hipStream_t>& v = streams[(size_t)device];
    const unsigned k = next[(size_t)device]++ % (unsigned)kPerDevice;
    // all of a device's streams are made with its first submap (a collection's constructor), not one per switch of submaps later
    while (v.size() < (size_t)kPerDevice) {
      hipStream_t s = nullptr;
      if (make_stream(&s, false) != hipSuccess) break;
      v.push_back(s);
    }
    if (v.empty()) return nullptr;
    return v[k % (unsigned)v.size()];
  }
};
inline SubmapStreamPool& submap_stream_pool() {
  static SubmapStreamPool* p = new SubmapStreamPool;  // never destroyed: its streams must not be torn down behind the runtime's own exit
  return *p;
}
}  // namespace

extern "C" {

static int submap_create_impl(int device, double map_voxel_size, const o3s_cropper* map_builder_cropper, bool own_stream, o3s_submap** out) {
  if (!out || !map_builder_cropper) return O3S_ERR_BAD_ARGUMENT;
  *out = nullptr;
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  o3s_submap* m = new o3s_submap();
  m->device = device;
  m->voxel = map_voxel_size;
  m->cropper = *map_builder_cropper;
  m->owns_stream = own_stream;
  if (own_stream) {
    if (make_stream(&m->stream, false) != hipSuccess) m->stream = nullptr;
  } else {
    m->stream = submap_stream_pool().get(device);
  }
  if (!m->stream || hipEventCreateWithFlags(&m->handover, hipEventDisableTiming) != hipSuccess) {
    if (m->stream && m->owns_stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return O3S_ERR_HIP;
  }
  *out = m;
  return O3S_OK;
}

int o3s_submap_create(int device, double map_voxel_size, const o3s_cropper* map_builder_cropper, o3s_submap** out) {
  return submap_create_impl(device, map_voxel_size, map_builder_cropper, /*own_stream=*/false, out);
}

void o3s_submap_destroy(o3s_submap* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  m->pend.active = false;  // nothing will look at the map again; the stream is drained below
  if (m->stream) {
    (void)hipStreamSynchronize(m->stream);
    if (m->owns_stream) (void)hipStreamDestroy(m->stream);
  }
  if (m->handover) (void)hipEventDestroy(m->handover);
  if (m->scan_read) (void)hipEventDestroy(m->scan_read);
  delete m;
}

int64_t o3s_submap_size(const o3s_submap* m) {
  (void)submap_settle(m);  // (an insert that fails while completing leaves the appended cloud: its size is what is reported)
  return m ? m->n : 0;
}
// the size without waiting for a pending insert: exact when none is pending, else [1, map before + scan] (voxelising never empties a cloud)
int o3s_submap_size_bounds(const o3s_submap* m, int64_t* at_least, int64_t* at_most) {
  if (!m || !at_least || !at_most) return O3S_ERR_BAD_ARGUMENT;
  if (m->pend.active) {
    *at_least = m->pend.n_tmp > 0 ? 1 : 0;
    *at_most = m->pend.n_tmp;
  } else {
    *at_least = *at_most = m->n;
  }
  return O3S_OK;
}

int o3s_submap_clone(const o3s_submap* src, int device, o3s_submap** out) {
  if (const int rs_ = submap_settle(src); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!src || !out) return O3S_ERR_BAD_ARGUMENT;
  *out = nullptr;
  int rc = submap_create_impl(device, src->voxel, &src->cropper, /*own_stream=*/true, out);  // a snapshot runs beside the mapper
  if (rc != O3S_OK) return rc;
  o3s_submap* m = *out;
  auto fail = [&](int code) {
    o3s_submap_destroy(m);
    *out = nullptr;
    return code;
  };
  if (hipSetDevice(src->device) != hipSuccess || hipStreamSynchronize(src->stream) != hipSuccess) return fail(O3S_ERR_HIP);  // the source is complete
  if (hipSetDevice(m->device) != hipSuccess) return fail(O3S_ERR_HIP);
  const size_t bytes = (size_t)src->n * 24;
  hipStream_t s = m->stream;
  auto copy = [&](DArr& dst, const DArr& from) -> bool {
    if (dst.ensure(bytes, 0, s) != hipSuccess) return false;
    if (m->device == src->device) return hipMemcpyAsync(dst.p, from.p, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess;
    return hipMemcpyPeerAsync(dst.p, m->device, from.p, src->device, bytes, s) == hipSuccess;  // over xGMI when the devices are peers
  };
  if (src->n > 0) {
    if (!copy(m->pts[0], src->pts[src->cur])) return fail(O3S_ERR_HIP);
    if (src->has_normals == 1 && !copy(m->nrm[0], src->nrm[src->cur])) return fail(O3S_ERR_HIP);
    if (src->has_colors == 1 && !copy(m->col[0], src->col[src->cur])) return fail(O3S_ERR_HIP);
    if (hipStreamSynchronize(s) != hipSuccess) return fail(O3S_ERR_HIP);
  }
  m->cur = 0;
  m->n = src->n;
  m->has_normals = src->has_normals;
  m->has_colors = src->has_colors;
  m->n_pt = src->n_pt;
  m->layout_valid = src->layout_valid;
  return O3S_OK;
}

int o3s_submap_insert_stats(const o3s_submap* m, int64_t* merged, int64_t* sorted, int64_t* fell_back) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m) return O3S_ERR_BAD_ARGUMENT;
  if (merged) *merged = m->n_merged;
  if (sorted) *sorted = m->n_sorted;
  if (fell_back) *fell_back = m->n_fell_back;
  return O3S_OK;
}

int o3s_submap_reserve(o3s_submap* m, int64_t n_points) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m || n_points < 0 || n_points > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  const int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  const size_t bytes = (size_t)n_points * 24;
  for (int k = 0; k < 2; ++k) {
    CK(m->pts[k].ensure(bytes, k == m->cur ? (size_t)m->n * 24 : 0, s));
    CK(m->nrm[k].ensure(bytes, (k == m->cur && m->has_normals == 1) ? (size_t)m->n * 24 : 0, s));
  }
  // the work area of either insert pipeline at that size (the merge takes the map plus a scan of up to 2 x 64 x 2048 points)
  const size_t work = std::max(voxel_arena_bytes(n_points), insert_merge_arena_bytes(n_points, n_points, std::min<int64_t>(n_points, 262144)));
  if (m->arena.cap < work) CK(m->arena.reserve(work));
  return O3S_OK;
}

int o3s_submap_trim(o3s_submap* m) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m) return O3S_ERR_BAD_ARGUMENT;
  const int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(m->stream));
  // a submap that is no longer inserted into keeps its map cloud ([cur]) and nothing else: the spare ping-pong arrays, the
  // sort / scan work area and the scan staging go back to the allocator (they come back on the next reserve / insert); the patch
  // buffers stay: an ICP handle may still be indexing the last patch on its own stream
  auto drop = [](DArr& a) {
    if (a.p) (void)hipFree(a.p);
    a.p = nullptr;
    a.cap = 0;
  };
  const int spare = 1 - m->cur;
  drop(m->pts[spare]);
  drop(m->nrm[spare]);
  drop(m->col[spare]);
  drop(m->scan_p);
  drop(m->scan_n);
  drop(m->scan_c);
  drop(m->carve_scan);
  if (m->arena.base) (void)hipFree(m->arena.base);
  m->arena.base = nullptr;
  m->arena.cap = m->arena.used = 0;
  // shrink the map arrays themselves to what the map holds (a submap closed by radius at a fraction of the reserved size)
  const size_t need = (size_t)m->n * 24;
  auto shrink = [&](DArr& a, bool used) -> hipError_t {
    if (!a.p || !used || a.cap <= need + need / 8 + 4096) return hipSuccess;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, need + 4096);
    if (e != hipSuccess) return hipSuccess;  // no room for the copy: keep the large array
    e = hipMemcpyAsync(q, a.p, need, hipMemcpyDeviceToDevice, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) {
      (void)hipFree(q);
      return e;
    }
    (void)hipFree(a.p);
    a.p = q;
    a.cap = need + 4096;
    return hipSuccess;
  };
  CK(shrink(m->pts[m->cur], true));
  // a map without normals / colours keeps none of the reserved room for them (the arrays come back with the next reserve / insert)
  if (m->has_normals == 1) CK(shrink(m->nrm[m->cur], true));
  else drop(m->nrm[m->cur]);
  if (m->has_colors == 1) CK(shrink(m->col[m->cur], true));
  else drop(m->col[m->cur]);
  return O3S_OK;
}

int o3s_submap_hand_over(o3s_submap* from, o3s_submap* to) {
  if (const int rs_ = submap_settle(from); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (const int rt_ = submap_settle(to); rt_ != O3S_OK) return rt_;  // a pending insert is completed first
  if (!from || !to || from == to || from->device != to->device || to->n != 0) return O3S_ERR_BAD_ARGUMENT;
  const int rc = set_dev(from);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(from->stream));
  CK(hipStreamSynchronize(to->stream));
  hipStream_t s = from->stream;
  const int c = from->cur;
  const size_t need = (size_t)from->n * 24;
  DArr* ff[3] = {from->pts, from->nrm, from->col};
  DArr* tf[3] = {to->pts, to->nrm, to->col};
  const bool used[3] = {true, from->has_normals == 1, from->has_colors == 1};
  // 1. the closed submap's map into arrays of its own size (the only allocations: a few MB each; nothing has moved if one fails)
  void* tight[3] = {nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int f = 0; f < 3 && e == hipSuccess; ++f)
    if (used[f] && need > 0 && ff[f][c].p) {
      e = hipMalloc(&tight[f], need + 4096);
      if (e == hipSuccess) e = hipMemcpyAsync(tight[f], ff[f][c].p, need, hipMemcpyDeviceToDevice, s);
    }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    for (void* q : tight)
      if (q) (void)hipFree(q);
    return O3S_ERR_HIP;
  }
  // 2. everything else changes hands
  auto give = [](DArr& dst, DArr& src) {
    if (dst.p) (void)hipFree(dst.p);  // a fresh submap holds nothing; one that was reserved gives that back
    dst.p = src.p;
    dst.cap = src.cap;
    src.p = nullptr;
    src.cap = 0;
  };
  for (int f = 0; f < 3; ++f) {
    give(tf[f][0], ff[f][c]);
    give(tf[f][1], ff[f][1 - c]);
    if (tight[f]) {
      ff[f][c].p = tight[f];
      ff[f][c].cap = need + 4096;
    }
  }
  to->cur = 0;
  give(to->scan_p, from->scan_p);
  give(to->scan_n, from->scan_n);
  give(to->scan_c, from->scan_c);
  give(to->carve_scan, from->carve_scan);
  give(to->patch_xyzw, from->patch_xyzw);
  give(to->patch_n32, from->patch_n32);
  give(to->d_post, from->d_post);
  if (to->arena.base) (void)hipFree(to->arena.base);
  to->arena.base = from->arena.base;
  to->arena.cap = from->arena.cap;
  to->arena.used = 0;
  from->arena.base = nullptr;
  from->arena.cap = from->arena.used = 0;
  return O3S_OK;
}

int64_t o3s_submap_device_bytes(const o3s_submap* m) {
  (void)submap_settle(m);
  if (!m) return 0;
  size_t b = m->arena.cap;
  for (int k = 0; k < 2; ++k) b += m->pts[k].cap + m->nrm[k].cap + m->col[k].cap;
  b += m->scan_p.cap + m->scan_n.cap + m->scan_c.cap + m->carve_scan.cap + m->d_T.cap + m->patch_xyzw.cap + m->patch_n32.cap;
  return (int64_t)b;
}

int o3s_submap_center(const o3s_submap* m, double center[3]) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m || !center) return O3S_ERR_BAD_ARGUMENT;
  center[0] = center[1] = center[2] = 0.0;
  if (m->n == 0) return O3S_OK;  // open3d ComputeCenter: zero for an empty cloud
  int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  const int G = (int)std::min<int64_t>(256, (m->n + kB - 1) / kB);
  Buf part;
  CK(part.alloc((size_t)G * 24));
  hipLaunchKernelGGL(k_center_part, dim3(G), dim3(kB), 0, s, (const double*)m->pts[m->cur].d(), m->n, part.as<double>());
  CK(hipGetLastError());
  double h[256 * 3];
  CK(hipMemcpyAsync(h, part.p, (size_t)G * 24, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  for (int a = 0; a < 3; ++a) {
    double t = 0.0;
    for (int b = 0; b < G; ++b) t += h[b * 3 + a];
    center[a] = t / (double)m->n;
  }
  return O3S_OK;
}

int o3s_transform_cloud(int device, const double T[16], const double* pts, const double* normals, const double* covariances, int64_t N,
                        double* out_pts, double* out_normals, double* out_covariances, int64_t* n_out) {
  if (!T || !n_out || N < 0 || (N > 0 && (!pts || !out_pts)) || (normals && !out_normals) || (covariances && !out_covariances)) return O3S_ERR_BAD_ARGUMENT;
  *n_out = 0;
  if (N == 0) return O3S_OK;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  double dev = 0.0;  // (T - Identity).array().abs().maxCoeff()
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) dev = std::max(dev, std::fabs(T[c * 4 + r] - (r == c ? 1.0 : 0.0)));
  const int64_t lead = dev < 1e-4 ? N : 0;  // "*out = cloud" first (helpers.cpp:285-288), the loop below appends regardless
  hipStream_t s = nullptr;
  Buf a, b, c, d, e, f, t;
  CK(a.alloc((size_t)N * 24));
  CK(d.alloc((size_t)N * 24));
  CK(t.alloc(128));
  CK(hipMemcpyAsync(a.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  CK(hipMemcpyAsync(t.p, T, 128, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(b.alloc((size_t)N * 24));
    CK(e.alloc((size_t)N * 24));
    CK(hipMemcpyAsync(b.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(k_transform_append, dim3(nblk(N)), dim3(kB), 0, s, a.as<double>(), normals ? b.as<double>() : nullptr, N, t.as<double>(),
                     d.as<double>(), e.as<double>());
  if (covariances) {
    CK(c.alloc((size_t)N * 72));
    CK(f.alloc((size_t)N * 72));
    CK(hipMemcpyAsync(c.p, covariances, (size_t)N * 72, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_transform_cov, dim3(nblk(N)), dim3(kB), 0, s, c.as<double>(), N, t.as<double>(), f.as<double>());
  }
  CK(hipGetLastError());
  if (lead) {
    std::memcpy(out_pts, pts, (size_t)N * 24);
    if (normals) std::memcpy(out_normals, normals, (size_t)N * 24);
    if (covariances) std::memcpy(out_covariances, covariances, (size_t)N * 72);
  }
  CK(hipMemcpyAsync(out_pts + 3 * lead, d.p, (size_t)N * 24, hipMemcpyDeviceToHost, s));
  if (normals) CK(hipMemcpyAsync(out_normals + 3 * lead, e.p, (size_t)N * 24, hipMemcpyDeviceToHost, s));
  if (covariances) CK(hipMemcpyAsync(out_covariances + 9 * lead, f.p, (size_t)N * 72, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  *n_out = lead + N;
  return O3S_OK;
}

int o3s_submap_upload(o3s_submap* m, const double* pts, const double* normals, int64_t N) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m || N < 0 || (N > 0 && !pts)) return O3S_ERR_BAD_ARGUMENT;
  int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  CK(m->pts[m->cur].ensure((size_t)N * 24, 0, s));
  if (N) CK(hipMemcpyAsync(m->pts[m->cur].p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(m->nrm[m->cur].ensure((size_t)N * 24, 0, s));
    if (N) CK(hipMemcpyAsync(m->nrm[m->cur].p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  CK(hipStreamSynchronize(s));
  m->n = N;
  m->layout_valid = false;
  m->has_normals = N == 0 ? -1 : (normals ? 1 : 0);
  m->has_colors = 0;  // an uploaded map comes without colours
  return O3S_OK;
}

int o3s_submap_download(const o3s_submap* m, double* pts, double* normals) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m || (m->n > 0 && !pts)) return O3S_ERR_BAD_ARGUMENT;
  if (m->n == 0) return O3S_OK;
  if (hipSetDevice(m->device) != hipSuccess) return O3S_ERR_HIP;
  CK(hipStreamSynchronize(m->stream));
  CK(hipMemcpy(pts, m->pts[m->cur].p, (size_t)m->n * 24, hipMemcpyDeviceToHost));
  if (normals) {
    if (m->has_normals != 1) return O3S_ERR_BAD_SHAPE;
    CK(hipMemcpy(normals, m->nrm[m->cur].p, (size_t)m->n * 24, hipMemcpyDeviceToHost));
  }
  return O3S_OK;
}

namespace {
// What is left of an insert once the merge path has answered (merged: its counts) or was not tried: the sort-based pipelines where
// needed, then the submap's new state.  The appended cloud [old map | scan] is in pts[c], the voxelised map goes to pts[1 - c].
int insert_finish(o3s_submap* m, int c, int64_t n_tmp, bool hn, bool have_vh, const VoxHint& vh, bool merged, const int64_t cnt_m[2]) {
  hipStream_t s = m->stream;
  Attrs at;
  if (m->has_colors == 1) {
    at.col = m->col[c].d();
    at.out_col = m->col[1 - c].d();
  }
  int rc = O3S_OK;
  bool hinted = merged;
  int64_t n_out = merged ? cnt_m[0] + cnt_m[1] : 0, n_pt_new = merged ? cnt_m[0] : 0;
  if (!hinted && have_vh) {
    int64_t cnt[3];
    rc = voxel_pipeline_hint_dev(m->arena, 0, &m->cropper, vh, m->voxel, m->pts[c].d(), hn ? m->nrm[c].d() : nullptr, n_tmp, m->pts[1 - c].d(),
                                 m->nrm[1 - c].d(), nullptr, &at, nullptr, nullptr, nullptr, cnt, &hinted, s);
    if (rc == O3S_OK && hinted) {
      n_out = cnt[0] + cnt[1];
      n_pt_new = cnt[0];
      ++m->n_sorted;
    }
  }
  m->layout_valid = false;
  if (rc == O3S_OK && !hinted)
    rc = voxel_pipeline_dev(m->arena, 0, &m->cropper, m->voxel, m->pts[c].d(), hn ? m->nrm[c].d() : nullptr, n_tmp, m->pts[1 - c].d(),
                            m->nrm[1 - c].d(), nullptr, &n_out, s, &at);
  if (rc != O3S_OK) {
    m->n = n_tmp;  // the appended cloud is still a valid map
    return rc;
  }
  CK(hipStreamSynchronize(s));
  m->cur = 1 - c;
  m->n = n_out;
  if (hinted) {  // the hinted pipelines emit [pass-through | voxels in key order] and say how many of each
    m->n_pt = n_pt_new;
    m->layout_valid = true;
  }
  return O3S_OK;
}

// Completes a pending insert (o3s_submap::PendingInsert); a no-op otherwise.  Every entry point that looks at or changes the map
// calls it first, so nothing outside ever sees the state in between.
int submap_settle(o3s_submap* m) {
  if (!m || !m->pend.active) return O3S_OK;
  const o3s_submap::PendingInsert p = m->pend;
  m->pend.active = false;
  if (hipSetDevice(m->device) != hipSuccess) return O3S_ERR_HIP;
  int64_t cnt[2];
  bool merged = false;
  const int rc = voxel_insert_merge_result(p.post, cnt, &merged);
  if (rc != O3S_OK) {
    m->n = p.n_tmp;
    return rc;
  }
  if (merged) ++m->n_merged;
  else {  // old pass-through points are back inside the volume (a revisit): that lasts for a while
    ++m->n_fell_back;
    m->merge_backoff = 4;
  }
  return insert_finish(m, p.c, p.n_tmp, p.hn, true, p.vh, merged, cnt);
}
// Submap::insertScan on a scan that already lives in HBM (d_pts / d_nrm: 3 x N doubles)
// lazy: the call may return with the insert enqueued and its completion pending (submap_settle)
// read_done (nullable): recorded on the submap's stream as soon as nothing enqueued here reads d_pts / d_nrm / d_col any more
int insert_dev(o3s_submap* m, const double* d_pts, const double* d_nrm, int64_t N, const double T_map_sensor[16], const double* d_col = nullptr,
               bool lazy = false, hipEvent_t read_done = nullptr) {
  hipStream_t s = m->stream;
  const bool hn = d_nrm != nullptr;
  // (T - Identity).array().abs().maxCoeff() < 1e-4: the reference copies the input cloud into the output and then
  // STILL appends the transformed points (helpers.cpp:285-288, 300-304) — the scan enters the map twice.  Kept as is.
  double dev = 0.0;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) dev = std::max(dev, std::fabs(T_map_sensor[c * 4 + r] - (r == c ? 1.0 : 0.0)));
  const bool doubled = dev < 1e-4;
  const int64_t add = doubled ? 2 * N : N;
  const int64_t n_tmp = m->n + add;
  if (n_tmp > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  Mat4d Tv;
  for (int k = 0; k < 16; ++k) Tv.m[k] = T_map_sensor[k];
  const int c = m->cur;
  CK(m->pts[c].ensure((size_t)n_tmp * 24, (size_t)m->n * 24, s));
  if (hn) CK(m->nrm[c].ensure((size_t)n_tmp * 24, (size_t)m->n * 24, s));
  // mapCloud_ += *transformedCloud (Submap.cpp:85)
  double* dst_p = m->pts[c].d() + 3 * m->n;
  double* dst_n = hn ? m->nrm[c].d() + 3 * m->n : nullptr;
  if (doubled) {
    CK(hipMemcpyAsync(dst_p, d_pts, (size_t)N * 24, hipMemcpyDeviceToDevice, s));
    if (hn) CK(hipMemcpyAsync(dst_n, d_nrm, (size_t)N * 24, hipMemcpyDeviceToDevice, s));
    dst_p += 3 * N;
    if (hn) dst_n += 3 * N;
  }
  hipLaunchKernelGGL(k_transform_append_v, dim3(nblk(N)), dim3(kB), 0, s, d_pts, d_nrm, N, Tv, dst_p, dst_n);
  CK(hipGetLastError());
  m->has_normals = hn ? 1 : 0;
  // Colours.  o3d_slam::transform copies them (out->colors_ = cloud.colors_, helpers.cpp:291) — N colours also when the
  // almost-identity quirk has doubled the points, so that cloud no longer "HasColors()" — and Open3D's
  // PointCloud::operator+= keeps the map's colours only if (map empty or map has colours) and the added cloud has them;
  // otherwise it clears them for good.
  {
    const bool scan_has = d_col != nullptr && !doubled;
    const bool keep_colors = (m->n == 0 || m->has_colors == 1) && scan_has;
    if (keep_colors) {
      CK(m->col[c].ensure((size_t)n_tmp * 24, (size_t)m->n * 24, s));
      CK(hipMemcpyAsync(m->col[c].d() + 3 * m->n, d_col, (size_t)N * 24, hipMemcpyDeviceToDevice, s));
    }
    m->has_colors = keep_colors ? 1 : 0;
  }
  if (read_done) CK(hipEventRecord(read_done, s));  // the scan has been taken in: from here on only the map's own arrays are read
  // mapBuilderCropper_->setPose(mapToRangeSensor) (Submap.cpp:86)
  for (int d = 0; d < 3; ++d) m->cropper.centre[d] = T_map_sensor[12 + d];
  if (!(m->voxel > 0.0)) {  // "Map voxel size is zero. Not voxelizing the map." (Submap.cpp:164-166)
    m->n = n_tmp;
    m->layout_valid = false;
    CK(hipStreamSynchronize(s));
    return O3S_OK;
  }
  // voxelizeInsideCroppingVolume: *map = *voxelizeWithinCroppingVolume(voxel, cropper, *map) (Submap.cpp:159-163)
  CK(m->pts[1 - c].ensure((size_t)n_tmp * 24, 0, s));
  CK(m->nrm[1 - c].ensure((size_t)n_tmp * 24, 0, s));
  if (m->has_colors == 1) CK(m->col[1 - c].ensure((size_t)n_tmp * 24, 0, s));
  int rc = O3S_OK;
  bool merged = false;
  int64_t cnt_m[2] = {0, 0};
  const int64_t n_old = m->n;
  VoxHint vh{};
  bool have_vh = false;
  {  // a bounded map-builder volume bounds the voxel indices: no extrema, one read-back (cloud_dev.h, "hinted")
    double lo[3], hi[3];
    have_vh = hints_enabled() && cropper_aabb(m->cropper, lo, hi) && vox_hint(0, lo, hi, m->voxel, &vh);
    if (have_vh) {
      // the map is already in voxel order: merge the (sorted) scan into it instead of sorting everything again
      if (m->merge_backoff > 0) --m->merge_backoff;
      else if (m->layout_valid && m->has_colors != 1 && n_old > m->n_pt && O3S_HOOK_ENV("O3S_INSERT_SORT") == nullptr) {
        LazyPost lp;
        bool issued = false;
        bool can_lazy = lazy && O3S_HOOK_ENV("O3S_INSERT_EAGER") == nullptr;
        if (can_lazy) {
          CK(m->d_post.ensure(64, 0, s));
          can_lazy = lazy_post_open(pinned_area(), reinterpret_cast<uint32_t*>(m->d_post.d()), s, &lp);
        }
        rc = voxel_insert_merge_dev(m->arena, m->cropper, vh, m->voxel, m->pts[c].d(), hn ? m->nrm[c].d() : nullptr, m->n_pt, n_old, n_tmp,
                                    m->pts[1 - c].d(), m->nrm[1 - c].d(), cnt_m, &merged, s, can_lazy ? &lp : nullptr, &issued);
        if (rc == O3S_OK && issued) {  // the answer is on its way: submap_settle() takes it from here
          m->layout_valid = false;
          m->pend.active = true;
          m->pend.post = lp;
          m->pend.c = c;
          m->pend.n_tmp = n_tmp;
          m->pend.hn = hn;
          m->pend.vh = vh;
          return O3S_OK;
        }
        if (rc == O3S_OK && merged) ++m->n_merged;
        else if (rc == O3S_OK) {  // old pass-through points are back inside the volume (a revisit): that lasts for a while
          ++m->n_fell_back;
          m->merge_backoff = 4;
        }
      }
    }
  }
  if (rc != O3S_OK) {
    m->layout_valid = false;
    m->n = n_tmp;  // the appended cloud is still a valid map
    return rc;
  }
  return insert_finish(m, c, n_tmp, hn, have_vh, vh, merged, cnt_m);
}
}  // namespace

int o3s_submap_insert_scan(o3s_submap* m, const double* pts, const double* normals, int64_t N, const double T_map_sensor[16]) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m || N < 0 || !T_map_sensor || (N > 0 && !pts)) return O3S_ERR_BAD_ARGUMENT;
  if (N == 0) return O3S_OK;  // "if (preProcessedScan.IsEmpty()) return true" (Submap.cpp:41-43)
  if (m->has_normals >= 0 && m->has_normals != (normals ? 1 : 0)) return O3S_ERR_BAD_SHAPE;
  const int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  CK(m->scan_p.ensure((size_t)N * 24, 0, s));
  CK(hipMemcpyAsync(m->scan_p.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(m->scan_n.ensure((size_t)N * 24, 0, s));
    CK(hipMemcpyAsync(m->scan_n.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  return insert_dev(m, m->scan_p.d(), normals ? m->scan_n.d() : nullptr, N, T_map_sensor);
}

int o3s_submap_insert_scan_colored(o3s_submap* m, const double* pts, const double* normals, const double* colors, int64_t N,
                                   const double T_map_sensor[16]) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!colors) return o3s_submap_insert_scan(m, pts, normals, N, T_map_sensor);
  if (!m || N < 0 || !T_map_sensor || (N > 0 && !pts)) return O3S_ERR_BAD_ARGUMENT;
  if (N == 0) return O3S_OK;
  if (m->has_normals >= 0 && m->has_normals != (normals ? 1 : 0)) return O3S_ERR_BAD_SHAPE;
  const int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  CK(m->scan_p.ensure((size_t)N * 24, 0, s));
  CK(hipMemcpyAsync(m->scan_p.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(m->scan_n.ensure((size_t)N * 24, 0, s));
    CK(hipMemcpyAsync(m->scan_n.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  CK(m->scan_c.ensure((size_t)N * 24, 0, s));
  CK(hipMemcpyAsync(m->scan_c.p, colors, (size_t)N * 24, hipMemcpyHostToDevice, s));
  return insert_dev(m, m->scan_p.d(), normals ? m->scan_n.d() : nullptr, N, T_map_sensor, m->scan_c.d());
}

int o3s_submap_has_colors(const o3s_submap* m) {
  (void)submap_settle(m);
  return m && m->n > 0 && m->has_colors == 1 ? 1 : 0;
}

int o3s_submap_download_colors(const o3s_submap* m, double* colors) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m || !colors) return O3S_ERR_BAD_ARGUMENT;
  if (m->n == 0) return O3S_OK;
  if (m->has_colors != 1) return O3S_ERR_BAD_SHAPE;
  if (hipSetDevice(m->device) != hipSuccess) return O3S_ERR_HIP;
  CK(hipStreamSynchronize(m->stream));
  CK(hipMemcpy(colors, m->col[m->cur].p, (size_t)m->n * 24, hipMemcpyDeviceToHost));
  return O3S_OK;
}

int o3s_submap_carve(o3s_submap* m, const o3s_carving_params* cp, const double* raw_pts, int64_t N, const double T_map_sensor[16], int64_t* n_removed) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (n_removed) *n_removed = 0;
  if (!m || !cp || !T_map_sensor || N < 0 || (N > 0 && !raw_pts) || !(cp->voxel_size > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  if (m->n == 0 || N == 0) return O3S_OK;  // "if (map->points_.empty() ...) return" (Submap.cpp:118-120)
  int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  const int64_t Nm = m->n;
  const bool hn = m->has_normals == 1;
  const int c = m->cur;
  // scan = transform(mapToRangeSensor, rawScan) (Submap.cpp:122)
  CK(m->scan_p.ensure((size_t)N * 24, 0, s));
  CK(m->carve_scan.ensure((size_t)N * 24, 0, s));
  CK(m->d_T.ensure(128, 0, s));
  CK(hipMemcpyAsync(m->scan_p.p, raw_pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  CK(hipMemcpyAsync(m->d_T.p, T_map_sensor, 128, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_transform_append, dim3(nblk(N)), dim3(kB), 0, s, m->scan_p.d(), (const double*)nullptr, N, m->d_T.d(), m->carve_scan.d(),
                     (double*)nullptr);
  // wideCroppedIdxs = cropper.getIndicesWithinVolume(*map): the map-builder cropper at the pose of the previous insert
  const size_t nm = (size_t)Nm;
  const size_t need = 3 * Arena::pad(nm * 4) + Arena::pad((nm + 1) * 4) + 2 * Arena::pad(nm * 8) + 2 * Arena::pad(nm * 4) + Arena::pad(kExtSlots * 6 * 4) +
                      Arena::pad(std::max(scan_temp_bytes(Nm), sort_temp_bytes(Nm))) + 8192;
  CK(m->arena.reserve(need));
  Arena& ar = m->arena;
  uint32_t* inflag = ar.take<uint32_t>(nm);
  uint32_t* remove = ar.take<uint32_t>(nm);
  uint32_t* keep = ar.take<uint32_t>(nm);
  uint32_t* off = ar.take<uint32_t>(nm + 1);
  uint64_t* keys = ar.take<uint64_t>(nm);
  uint64_t* keys2 = ar.take<uint64_t>(nm);
  uint32_t* vals = ar.take<uint32_t>(nm);
  uint32_t* vals2 = ar.take<uint32_t>(nm);
  int32_t* d_mm = ar.take<int32_t>(kExtSlots * 6);
  const size_t tb_scan = scan_temp_bytes(Nm), tb_sort = sort_temp_bytes(Nm);
  void* tmp = ar.take<char>(std::max(tb_scan, tb_sort));
  hipLaunchKernelGGL(k_mask, dim3(nblk(Nm)), dim3(kB), 0, s, m->cropper, m->pts[c].d(), Nm, 1, inflag);
  int64_t n_in = 0;
  rc = scan_flags(inflag, off, Nm, tmp, tb_scan, &n_in, s);
  if (rc != O3S_OK) return rc;
  if (n_in == 0) return O3S_OK;
  const double inv = 1.0 / cp->voxel_size;
  rc = ext_i32_init(d_mm, s);
  if (rc != O3S_OK) return rc;
  hipLaunchKernelGGL(k_carve_box, dim3(nblk(Nm)), dim3(kB), 0, s, m->pts[c].d(), Nm, inflag, inv, d_mm);
  int32_t mm[6];
  rc = ext_i32_fetch(d_mm, mm, s);
  if (rc != O3S_OK) return rc;
  const int64_t ex = (int64_t)mm[3] - mm[0] + 1, ey = (int64_t)mm[4] - mm[1] + 1, ez = (int64_t)mm[5] - mm[2] + 1;
  if ((long double)ex * (long double)ey * (long double)ez >= 9.0e18L) return O3S_ERR_BAD_ARGUMENT;
  // sort bits: the packed range plus one bit for the key of the points outside the volume (the whole map is sorted: nine passes
  // with all 64 bits, three or four with the range that is known here)
  const int kb = key_bits((uint64_t)ex * (uint64_t)ey * (uint64_t)ez);
  const uint64_t out_key = kb < 63 ? (1ull << kb) : ~0ull;
  hipLaunchKernelGGL(k_carve_keys, dim3(nblk(Nm)), dim3(kB), 0, s, m->pts[c].d(), Nm, inflag, inv, mm[0], mm[1], mm[2], (uint64_t)ex, (uint64_t)ey, out_key,
                     keys, vals);
  size_t tb = tb_sort;
  CK(sort_pairs(tmp, tb, keys, keys2, vals, vals2, nm, kb < 63 ? kb + 1 : 64, s));
  CK(hipMemsetAsync(remove, 0, nm * 4, s));
  hipLaunchKernelGGL(k_carve_rays, dim3(nblk(N)), dim3(kB), 0, s, m->carve_scan.d(), N, T_map_sensor[12], T_map_sensor[13], T_map_sensor[14],
                     cp->voxel_size, inv, cp->max_raytracing_length, cp->truncation_distance, cp->min_dot_product_with_normal, keys2, vals2, n_in, mm[0],
                     mm[1], mm[2], ex, ey, ez, hn ? m->nrm[c].d() : nullptr, remove);
  // removeByIds: SelectByIndex(ids, invert) keeps the survivors in their order
  hipLaunchKernelGGL(k_invert_flags, dim3(nblk(Nm)), dim3(kB), 0, s, remove, Nm, keep);
  int64_t n_keep = 0;
  rc = scan_flags(keep, off, Nm, tmp, tb_scan, &n_keep, s);
  if (rc != O3S_OK) return rc;
  if (n_keep < Nm) {
    CK(m->pts[1 - c].ensure((size_t)Nm * 24, 0, s));
    CK(m->nrm[1 - c].ensure((size_t)Nm * 24, 0, s));
    hipLaunchKernelGGL(k_compact, dim3(nblk(Nm)), dim3(kB), 0, s, m->pts[c].d(), hn ? m->nrm[c].d() : nullptr, Nm, keep, off, m->pts[1 - c].d(),
                       m->nrm[1 - c].d(), (int32_t*)nullptr);
    if (m->has_colors == 1) {  // SelectByIndex carries the colours along
      CK(m->col[1 - c].ensure((size_t)Nm * 24, 0, s));
      hipLaunchKernelGGL(k_compact_attr, dim3(nblk(Nm)), dim3(kB), 0, s, (const double*)m->col[c].d(), (const double*)nullptr, Nm, keep, off,
                         m->col[1 - c].d(), (double*)nullptr);
    }
    CK(hipGetLastError());
    CK(hipStreamSynchronize(s));
    m->cur = 1 - c;
    m->n = n_keep;
    m->layout_valid = false;  // order is kept, but how many pass-through points survived is not known here
  }
  if (n_removed) *n_removed = Nm - n_keep;
  return O3S_OK;
}

int o3s_submap_patch_count(o3s_submap* m, const o3s_cropper* scan_matcher_cropper, const double T_map_sensor[16], int64_t* n_patch) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m || !scan_matcher_cropper || !T_map_sensor || !n_patch) return O3S_ERR_BAD_ARGUMENT;
  *n_patch = 0;
  if (m->n == 0) return O3S_OK;
  int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  o3s_cropper c = *scan_matcher_cropper;  // scanMatcherCropper_->setPose(mapToRangeSensor)
  for (int d = 0; d < 3; ++d) c.centre[d] = T_map_sensor[12 + d];
  const int64_t N = m->n;
  CK(m->arena.reserve(crop_arena_bytes(N)));
  uint32_t* flag = m->arena.take<uint32_t>((size_t)N);
  uint32_t* off = m->arena.take<uint32_t>((size_t)N + 1);
  const size_t tb = scan_temp_bytes(N);
  void* tmp = m->arena.take<char>(tb);
  uint32_t* blk = reinterpret_cast<uint32_t*>(tmp);
  hipLaunchKernelGGL(k_mask, dim3(nblk(N)), dim3(kB), 0, s, c, (const double*)m->pts[m->cur].d(), N, 1, flag, (uint32_t*)nullptr, blk);
  return scan_flags(flag, off, N, tmp, tb, n_patch, s, blk);
}

int o3s_submap_set_reference(o3s_submap* m, const o3s_cropper* scan_matcher_cropper, const double T_map_sensor[16], o3s_icp* icp,
                             int64_t* n_patch) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (n_patch) *n_patch = 0;
  if (!m || !scan_matcher_cropper || !T_map_sensor || !icp) return O3S_ERR_BAD_ARGUMENT;
  if (m->n == 0) return O3S_ERR_EMPTY_REFERENCE;
  int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  // The previous patch may still be read by the handle's asynchronous index build when a compute failed before anything
  // waited for its stream (ERR_NOT_RIGID, an empty reading: the Mapper keeps the prior and goes on): the patch buffers are
  // rewritten below, so the handle's stream is drained first.  It is idle in the normal course of a scan.
  rc = o3s_icp_synchronize(icp);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  o3s_cropper c = *scan_matcher_cropper;  // scanMatcherCropper_->setPose(mapToRangeSensor)
  for (int d = 0; d < 3; ++d) c.centre[d] = T_map_sensor[12 + d];
  const bool hn = m->has_normals == 1;
  // cropSubmap + open3dToPointmatcher in one compaction: mask, scan, then the kept points straight into fp32.  The COUNT stays on
  // the device: the compaction is launched for all N points into buffers that hold N, its last thread leaves the count behind the
  // offsets, and the index build reads it there — the host learns it from the post of the reference's statistics, which it waits
  // for anyway (round 4: a hand-over of the count first, then one of the statistics: two waits, ~15 us of every re-init)
  int64_t kept = 0;
  uint32_t* d_count = nullptr;
  const int64_t N = m->n;
  {
    CK(m->arena.reserve(crop_arena_bytes(N)));
    uint32_t* flag = m->arena.take<uint32_t>((size_t)N);
    uint32_t* off = m->arena.take<uint32_t>((size_t)N + 1);
    const size_t tb = scan_temp_bytes(N);
    void* tmp = m->arena.take<char>(tb);
    uint32_t* blk = reinterpret_cast<uint32_t*>(tmp);  // the mask's per-block counts: the scan is one launch (cloud_dev.h k_scan_flags_blk)
    hipLaunchKernelGGL(k_mask, dim3(nblk(N)), dim3(kB), 0, s, c, (const double*)m->pts[m->cur].d(), N, 1, flag, (uint32_t*)nullptr, blk);
    rc = scan_flags_dev(flag, off, N, tmp, tb, s, blk);
    if (rc != O3S_OK) return rc;
    CK(m->patch_xyzw.ensure((size_t)N * 16, 0, s));
    CK(m->patch_n32.ensure((size_t)N * 12, 0, s));
    d_count = off + N;
    hipLaunchKernelGGL(k_compact_pm, dim3(nblk(N)), dim3(kB), 0, s, (const double*)m->pts[m->cur].d(), hn ? (const double*)m->nrm[m->cur].d() : nullptr, N,
                       flag, off, reinterpret_cast<float4*>(m->patch_xyzw.p), reinterpret_cast<float*>(m->patch_n32.p), d_count);
    CK(hipGetLastError());
  }
  // the ICP handle works on its own stream: it waits for the patch on the device, and reads it asynchronously — the patch
  // buffers (and the count word in this submap's work area) are not touched again before the next call on this submap, which the
  // host only reaches after a compute has waited
  CK(hipEventRecord(m->handover, s));
  rc = o3s_icp_wait_event(icp, m->handover);
  if (rc != O3S_OK) return rc;
  rc = o3s_icp_init_reference_dev_counted_async(icp, m->patch_xyzw.p, hn ? m->patch_n32.p : nullptr, d_count, N, &kept);
  if (n_patch) *n_patch = kept;
  return rc;
}

// RegistrationICP between two resident submaps: nothing is uploaded; the source cloud is copied inside HBM (the
// iteration transforms it in place), the target and its normals are read where they lie
int o3s_o3d_registration_icp_submaps(const o3s_submap* source, const o3s_submap* target, double max_dist, const double init[16],
                                     const o3s_o3d_icp_criteria* criteria, o3s_o3d_icp_result* result, double* info36) {
  if (const int rs_ = submap_settle(source); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (const int rt_ = submap_settle(target); rt_ != O3S_OK) return rt_;  // a pending insert is completed first
  if (!source || !target || !init || !result || !(max_dist > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  if (source->device != target->device) return O3S_ERR_BAD_ARGUMENT;
  if (source->n == 0 || target->n == 0) return O3S_ERR_EMPTY_REFERENCE;
  if (target->has_normals != 1) return O3S_ERR_BAD_SHAPE;  // "requires target pointcloud to have normals"
  int rc = set_dev(target);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(source->stream));
  CK(hipStreamSynchronize(target->stream));
  hipStream_t s = target->stream;
  o3s_cloud::RegLease area(target->device, s);
  rc = o3d_icp_run(area->reg, source->pts[source->cur].d(), source->n, target->pts[target->cur].d(), target->nrm[target->cur].d(), target->n,
                   max_dist, init, criteria, result, s, /*on_device=*/true);
  if (rc == O3S_OK && info36) rc = o3d_info_after_icp(area->reg, max_dist, result->transformation, info36, s);
  return area.end(rc);
}

// ---- device-resident pre-processed scan (include/o3s_scan.h) ---------------------------------------------------------
}  // extern "C"

struct o3s_scan {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t handover = nullptr;  // recorded on `stream`, waited for by the ICP handle's stream
  DArr raw_p, raw_n, tmp_p, tmp_n, wide_p, wide_n, narrow_p, narrow_n, xyzw, n32;
  int64_t n_wide = 0, n_narrow = 0, n_raw = 0;  // n_raw: the raw scan of the last preprocess, still in raw_p (/ raw_n)
  int raw_has_normals = 0;
  double normal_radius = 0.0;
  int32_t normal_knn = 0;
  bool voxel_ordered = false;  // the last preprocess down-sampled: merge / match clouds are in (z, y, x) voxel order
  bool pm_ready = false;       // ... and already wrote the match cloud in the PM layout (xyzw / n32)
  Arena arena;
  NormalsWork nwork;
};

// A raw scan staged in HBM ahead of the mapping call (include/o3s_scan.h: o3s_raw_scan)
struct o3s_raw_scan {
  int device = 0;
  hipStream_t stream = nullptr;
  DArr p, n;
  int64_t N = 0;
  int has_normals = 0;
};

extern "C" {

int o3s_raw_scan_create(int device, o3s_raw_scan** out) {
  if (!out) return O3S_ERR_BAD_ARGUMENT;
  *out = nullptr;
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  o3s_raw_scan* r = new o3s_raw_scan();
  r->device = device;
  if (make_stream(&r->stream, true) != hipSuccess) {
    delete r;
    return O3S_ERR_HIP;
  }
  *out = r;
  return O3S_OK;
}

void o3s_raw_scan_destroy(o3s_raw_scan* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  if (r->stream) {
    (void)hipStreamSynchronize(r->stream);
    (void)hipStreamDestroy(r->stream);
  }
  delete r;
}

int o3s_raw_scan_upload(o3s_raw_scan* r, const double* pts, const double* normals, int64_t N) {
  if (!r || N < 0 || (N > 0 && !pts) || N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  r->N = 0;
  if (N == 0) return O3S_OK;
  if (hipSetDevice(r->device) != hipSuccess) return O3S_ERR_HIP;
  hipStream_t s = r->stream;
  CK(r->p.ensure((size_t)N * 24, 0, s));
  CK(hipMemcpyAsync(r->p.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(r->n.ensure((size_t)N * 24, 0, s));
    CK(hipMemcpyAsync(r->n.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  CK(hipStreamSynchronize(s));  // the caller's arrays are free again, the staged copy is complete
  r->N = N;
  r->has_normals = normals ? 1 : 0;
  return O3S_OK;
}

int64_t o3s_raw_scan_size(const o3s_raw_scan* r) { return r ? r->N : -1; }

int o3s_host_alloc_pinned(size_t bytes, void** out) {
  if (!out || bytes == 0) return O3S_ERR_BAD_ARGUMENT;
  *out = nullptr;
  return hipHostMalloc(out, bytes, hipHostMallocPortable) == hipSuccess ? O3S_OK : O3S_ERR_HIP;
}

void o3s_host_free_pinned(void* p) {
  if (p) (void)hipHostFree(p);
}

int o3s_scan_create(int device, o3s_scan** out) {
  if (!out) return O3S_ERR_BAD_ARGUMENT;
  *out = nullptr;
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  o3s_scan* sc = new o3s_scan();
  sc->device = device;
  if (make_stream(&sc->stream, true) != hipSuccess ||
      hipEventCreateWithFlags(&sc->handover, hipEventDisableTiming) != hipSuccess) {
    if (sc->stream) (void)hipStreamDestroy(sc->stream);
    delete sc;
    return O3S_ERR_HIP;
  }
  *out = sc;
  return O3S_OK;
}

void o3s_scan_destroy(o3s_scan* sc) {
  if (!sc) return;
  (void)hipSetDevice(sc->device);
  if (sc->stream) {
    (void)hipStreamSynchronize(sc->stream);
    (void)hipStreamDestroy(sc->stream);
  }
  if (sc->handover) (void)hipEventDestroy(sc->handover);
  delete sc;
}

int o3s_scan_set_normal_estimation(o3s_scan* sc, double max_radius, int32_t knn) {
  if (!sc) return O3S_ERR_BAD_ARGUMENT;
  if (knn <= 0) {
    sc->normal_knn = 0;
    return O3S_OK;
  }
  if (knn > kNnMax || !(max_radius > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  sc->normal_radius = max_radius;
  sc->normal_knn = knn;
  return O3S_OK;
}

}  // extern "C"

// pts / normals: host arrays, or (on_device) arrays already in HBM on the scan's device
static int scan_preprocess_impl(o3s_scan* sc, const o3s_cropper* map_builder_cropper, double voxel_size, const o3s_cropper* scan_matcher_cropper,
                                const double* pts, const double* normals, int64_t N, int64_t* n_merge, int64_t* n_match, bool on_device) {
  if (n_merge) *n_merge = 0;
  if (n_match) *n_match = 0;
  if (!sc || !map_builder_cropper || !scan_matcher_cropper || N < 0 || (N > 0 && !pts)) return O3S_ERR_BAD_ARGUMENT;
  const bool estimate = normals == nullptr;
  if (N > 0 && estimate && sc->normal_knn <= 0) return O3S_ERR_BAD_SHAPE;  // no normals and no estimation parameters
  if (N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  sc->n_wide = sc->n_narrow = sc->n_raw = 0;
  sc->pm_ready = false;
  if (N == 0) return O3S_OK;
  if (hipSetDevice(sc->device) != hipSuccess) return O3S_ERR_HIP;
  hipStream_t s = sc->stream;
  CK(sc->raw_p.ensure((size_t)N * 24, 0, s));
  CK(sc->raw_n.ensure((size_t)N * 24, 0, s));
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  CK(hipMemcpyAsync(sc->raw_p.p, pts, (size_t)N * 24, kind, s));
  if (!estimate) CK(hipMemcpyAsync(sc->raw_n.p, normals, (size_t)N * 24, kind, s));
  for (DArr* a : {&sc->tmp_p, &sc->tmp_n, &sc->wide_p, &sc->wide_n, &sc->narrow_p, &sc->narrow_n}) CK(a->ensure((size_t)N * 24, 0, s));
  int rc = O3S_OK;
  {  // bounded wide volume: crop, down-sample and narrow crop as one pipeline with a single read-back (cloud_dev.h, "hinted")
    double lo[3], hi[3];
    VoxHint vh;
    if (hints_enabled() && voxel_size > 0.0 && cropper_aabb(*map_builder_cropper, lo, hi) && vox_hint(1, lo, hi, voxel_size, &vh)) {
      int64_t cnt[3];
      bool ok = false;
      if (!estimate) {  // the narrow crop writes the reading for the ICP as well: no conversion pass in o3s_scan_set_reading
        CK(sc->xyzw.ensure((size_t)N * 16, 0, s));
        CK(sc->n32.ensure((size_t)N * 12, 0, s));
      }
      rc = voxel_pipeline_hint_dev(sc->arena, 1, map_builder_cropper, vh, voxel_size, sc->raw_p.d(), estimate ? nullptr : sc->raw_n.d(), N,
                                   sc->wide_p.d(), sc->wide_n.d(), nullptr, nullptr, estimate ? nullptr : scan_matcher_cropper, sc->narrow_p.d(),
                                   sc->narrow_n.d(), cnt, &ok, s, estimate ? nullptr : reinterpret_cast<float4*>(sc->xyzw.p),
                                   estimate ? nullptr : reinterpret_cast<float*>(sc->n32.p));
      if (rc != O3S_OK) return rc;
      if (ok) {
        int64_t n_wide = cnt[1], n_narrow = cnt[2];
        if (estimate) {
          if (n_wide > 0) {
            rc = estimate_normals_dev(sc->nwork, sc->wide_p.d(), n_wide, sc->normal_radius, sc->normal_knn, sc->wide_n.d(), nullptr, s);
            if (rc != O3S_OK) return rc;
          }
          rc = crop_dev(sc->arena, *scan_matcher_cropper, sc->wide_p.d(), sc->wide_n.d(), n_wide, sc->narrow_p.d(), sc->narrow_n.d(), &n_narrow, s);
          if (rc != O3S_OK) return rc;
        }
        CK(hipStreamSynchronize(s));
        sc->n_wide = n_wide;
        sc->n_narrow = n_narrow;
        sc->n_raw = N;
        sc->raw_has_normals = estimate ? 0 : 1;
        sc->voxel_ordered = true;
        sc->pm_ready = !estimate;
        if (n_merge) *n_merge = n_wide;
        if (n_match) *n_match = n_narrow;
        return O3S_OK;
      }
    }
  }
  // preprocess(): croppedCloud = mapBuilderCropper_->crop(in)
  int64_t n_crop = 0;
  rc = crop_dev(sc->arena, *map_builder_cropper, sc->raw_p.d(), estimate ? nullptr : sc->raw_n.d(), N, sc->tmp_p.d(), sc->tmp_n.d(), &n_crop, s);
  if (rc != O3S_OK) return rc;
  // o3d_slam::voxelize(voxelSize, croppedCloud): Open3D VoxelDownSample, or nothing for voxelSize <= 0
  int64_t n_wide = 0;
  if (voxel_size > 0.0 && n_crop > 0) {
    rc = voxel_pipeline_dev(sc->arena, 1, nullptr, voxel_size, sc->tmp_p.d(), estimate ? nullptr : sc->tmp_n.d(), n_crop, sc->wide_p.d(), sc->wide_n.d(),
                            nullptr, &n_wide, s);
    if (rc != O3S_OK) return rc;
  } else {
    n_wide = n_crop;
    if (n_crop) {
      CK(hipMemcpyAsync(sc->wide_p.p, sc->tmp_p.p, (size_t)n_crop * 24, hipMemcpyDeviceToDevice, s));
      if (!estimate) CK(hipMemcpyAsync(sc->wide_n.p, sc->tmp_n.p, (size_t)n_crop * 24, hipMemcpyDeviceToDevice, s));
    }
  }
  // cloudRegistration->estimateNormalsOrCovariancesIfNeeded(croppedCloud): only for clouds that came without normals
  if (estimate && n_wide > 0) {
    rc = estimate_normals_dev(sc->nwork, sc->wide_p.d(), n_wide, sc->normal_radius, sc->normal_knn, sc->wide_n.d(), nullptr, s);
    if (rc != O3S_OK) return rc;
  }
  // narrowCropped = scanMatcherCropper_->crop(*wideCropped)
  int64_t n_narrow = 0;
  rc = crop_dev(sc->arena, *scan_matcher_cropper, sc->wide_p.d(), sc->wide_n.d(), n_wide, sc->narrow_p.d(), sc->narrow_n.d(), &n_narrow, s);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(s));
  sc->n_wide = n_wide;
  sc->n_narrow = n_narrow;
  sc->n_raw = N;
  sc->raw_has_normals = estimate ? 0 : 1;
  sc->voxel_ordered = voxel_size > 0.0;
  sc->pm_ready = false;
  if (n_merge) *n_merge = n_wide;
  if (n_match) *n_match = n_narrow;
  return O3S_OK;
}

extern "C" {

int o3s_scan_preprocess(o3s_scan* sc, const o3s_cropper* map_builder_cropper, double voxel_size, const o3s_cropper* scan_matcher_cropper,
                        const double* pts, const double* normals, int64_t N, int64_t* n_merge, int64_t* n_match) {
  return scan_preprocess_impl(sc, map_builder_cropper, voxel_size, scan_matcher_cropper, pts, normals, N, n_merge, n_match, false);
}

int o3s_scan_preprocess_staged(o3s_scan* sc, const o3s_cropper* map_builder_cropper, double voxel_size, const o3s_cropper* scan_matcher_cropper,
                               const o3s_raw_scan* raw, int64_t* n_merge, int64_t* n_match) {
  if (!raw || !sc || raw->device != sc->device) return O3S_ERR_BAD_ARGUMENT;
  return scan_preprocess_impl(sc, map_builder_cropper, voxel_size, scan_matcher_cropper, raw->p.d(), raw->has_normals ? raw->n.d() : nullptr, raw->N,
                              n_merge, n_match, true);
}

int64_t o3s_scan_get(const o3s_scan* sc, int which, double* pts, double* normals) {
  if (!sc || (which != 0 && which != 1)) return -1;
  const int64_t n = which == 0 ? sc->n_wide : sc->n_narrow;
  if (!pts || n == 0) return n;
  if (hipSetDevice(sc->device) != hipSuccess) return -1;
  const DArr& p = which == 0 ? sc->wide_p : sc->narrow_p;
  const DArr& q = which == 0 ? sc->wide_n : sc->narrow_n;
  if (hipStreamSynchronize(sc->stream) != hipSuccess) return -1;
  if (hipMemcpy(pts, p.p, (size_t)n * 24, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (normals && hipMemcpy(normals, q.p, (size_t)n * 24, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}

int o3s_scan_set_reading(o3s_scan* sc, o3s_icp* icp) {
  if (!sc || !icp) return O3S_ERR_BAD_ARGUMENT;
  if (sc->n_narrow == 0) return O3S_ERR_EMPTY_READING;  // "narrow cropped size is zero" (ScanToMapRegistration.cpp:66)
  if (hipSetDevice(sc->device) != hipSuccess) return O3S_ERR_HIP;
  hipStream_t s = sc->stream;
  const int64_t n = sc->n_narrow;
  if (!sc->pm_ready) {
    CK(sc->xyzw.ensure((size_t)n * 16, 0, s));
    CK(sc->n32.ensure((size_t)n * 12, 0, s));
    hipLaunchKernelGGL(k_o3d_to_pm, dim3(nblk(n)), dim3(kB), 0, s, sc->narrow_p.d(), sc->narrow_n.d(), n, reinterpret_cast<float4*>(sc->xyzw.p),
                       reinterpret_cast<float*>(sc->n32.p));
    CK(hipGetLastError());
  }
  CK(hipEventRecord(sc->handover, s));  // the ICP handle's stream waits for the conversion on the device
  const int rc = o3s_icp_wait_event(icp, sc->handover);
  if (rc != O3S_OK) return rc;
  const int rc2 = o3s_icp_set_reading_dev(icp, sc->xyzw.p, sc->n32.p, n);
  if (rc2 != O3S_OK) return rc2;
  return o3s_icp_reading_is_spatially_sorted(icp, sc->voxel_ordered ? 1 : 0);  // voxel order is spatial order: no re-sort per compute
}

int o3s_submap_insert_processed(o3s_submap* m, const o3s_scan* sc, const double T_map_sensor[16]) {
  if (const int rs_ = submap_settle(m); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (!m || !sc || !T_map_sensor) return O3S_ERR_BAD_ARGUMENT;
  if (sc->n_wide == 0) return O3S_OK;
  if (m->device != sc->device) return O3S_ERR_BAD_ARGUMENT;
  if (m->has_normals == 0) return O3S_ERR_BAD_SHAPE;
  const int rc = set_dev(m);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(sc->stream));
  // the insert may come back with its completion pending (submap_settle): every later call on the submap completes it first.  The
  // scan object may be refilled as soon as this returns, so its stream is ordered behind the kernels that read it here.
  if (!m->scan_read) CK(hipEventCreateWithFlags(&m->scan_read, hipEventDisableTiming));
  const int ri = insert_dev(m, sc->wide_p.d(), sc->wide_n.d(), sc->n_wide, T_map_sensor, nullptr, /*lazy=*/true, m->scan_read);
  if (m->pend.active) CK(hipStreamWaitEvent(sc->stream, m->scan_read, 0));
  return ri;
}

int o3s_estimate_normals(int device, const double* pts, int64_t N, double radius, int32_t max_nn, double* out_normals, int32_t* out_nn_idx) {
  if (!pts || !out_normals || N < 0 || max_nn < 1 || max_nn > kNnMax || !(radius > 0.0)) return O3S_ERR_BAD_ARGUMENT;
  if (N == 0) return O3S_OK;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  hipStream_t s = nullptr;
  Buf d_p, d_n, d_i;
  NormalsWork w;
  CK(d_p.alloc((size_t)N * 24));
  CK(d_n.alloc((size_t)N * 24));
  if (out_nn_idx) CK(d_i.alloc((size_t)N * (size_t)max_nn * 4));
  CK(hipMemcpyAsync(d_p.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  rc = estimate_normals_dev(w, d_p.as<double>(), N, radius, max_nn, d_n.as<double>(), out_nn_idx ? d_i.as<int32_t>() : nullptr, s);
  if (rc != O3S_OK) return rc;
  CK(hipMemcpyAsync(out_normals, d_n.p, (size_t)N * 24, hipMemcpyDeviceToHost, s));
  if (out_nn_idx) CK(hipMemcpyAsync(out_nn_idx, d_i.p, (size_t)N * (size_t)max_nn * 4, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  return O3S_OK;
}

}  // extern "C"
