// overlap_impl.h — computeIndicesOfOverlappingPoints (O3S/src/helpers.cpp:319-345) and the loop-closure refinement that
// uses it (O3S/src/PlaceRecognition.cpp:97-150) on the device; gfx950 only.  Included at the end of cloud_ops.hip after
// dense_map_impl.h (one TU: it shares the voxel-key packing of the dense map, the rocPRIM sort / scan instantiations and
// the Open3D-semantics ICP of o3d_icp_impl.h).
//
// The reference fills a VoxelMap (unordered_map voxel key -> per-layer index lists) with the target cloud and with the
// source cloud moved by sourceToTarget, then walks the map and keeps the indices of every voxel that holds at least
// minNumPointsPerVoxel points of BOTH layers.  Here: one packed voxel key per point (getVoxelIdx, reciprocal form,
// VoxelHashMap.hpp:43-51) and ONE open-addressing table keyed by it with a count per layer: the lanes of a wave that fall
// into the same voxel (clouds arrive scan by scan, so most do) add their count with one atomic; a second pass looks every
// point's voxel up.  The overlap voxels are large (2 m: a few thousand of them under 0.5 M points), so the table starts at
// 2^16 slots; a table that fills up is reported and the pass repeated with room for one voxel per point.  (Round 3 sorted both
// 64-bit key arrays — rocPRIM takes its merge sort for them — and searched them per point: 0.46 ms of a 3.1 ms refinement.)
// The reference's output order is its hash map's iteration order (unspecified); ascending index order is used here, on the
// device and in the oracle.
#pragma once
#include "dense_map_impl.h"

namespace {
namespace o3s_cloud {

// Open3D PointCloud::Transform on points only: p' = (T [p 1]).head<3>() / w (same arithmetic as k_transform_append)
__global__ void __launch_bounds__(kB) k_ov_keys(const double* __restrict__ pts, int64_t N, const double* __restrict__ Tm /*nullable: identity*/,
                                                double inv, uint64_t* __restrict__ keys, uint32_t* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  if (Tm) {
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double s = Tm[r] * x;
      s = s + Tm[4 + r] * y;
      s = s + Tm[8 + r] * z;
      s = s + Tm[12 + r] * 1.0;
      v[r] = s;
    }
    x = v[0] / v[3];
    y = v[1] / v[3];
    z = v[2] / v[3];
  }
  const double fx = floor(x * inv), fy = floor(y * inv), fz = floor(z * inv);
  if (dm_in_range(fx) && dm_in_range(fy) && dm_in_range(fz)) {
    keys[i] = dm_pack((int32_t)fx, (int32_t)fy, (int32_t)fz);
  } else {
    keys[i] = kDmEmpty;
    *err = 1u;
  }
}

struct __attribute__((aligned(16))) OvSlot {
  unsigned long long key1;  // packed voxel key + 1; 0 = empty
  uint32_t n[2];            // points of the source layer, of the target layer
};
__device__ __forceinline__ uint32_t ov_hash(uint64_t k) {  // splitmix64 finaliser
  k ^= k >> 30;
  k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27;
  k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return (uint32_t)k;
}

// counts the points of `layer` per voxel and notes every point's slot.  keys[i] must hold the voxel key of point i (k_ov_keys);
// err[1] = the table is full.
__global__ void __launch_bounds__(kB) k_ov_count(const uint64_t* __restrict__ keys, int64_t N, OvSlot* __restrict__ tab, uint32_t mask, int layer,
                                                 uint32_t* __restrict__ slot_of, uint32_t* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63);
  const uint64_t k = i < N ? keys[i] : kDmEmpty;
  bool pending = k != kDmEmpty;  // out-of-range points carry kDmEmpty (and have raised err[0])
  uint32_t my_slot = 0xffffffffu;
  for (;;) {
    const unsigned long long open = __ballot(pending);
    if (!open) break;
    const int leader = __ffsll((long long)open) - 1;
    const uint64_t lk = ((uint64_t)(uint32_t)__shfl((int)(k >> 32), leader) << 32) | (uint64_t)(uint32_t)__shfl((int)(k & 0xffffffffu), leader);
    const bool same = pending && k == lk;
    const unsigned long long grp = __ballot(same);
    uint32_t h = 0xffffffffu;
    if (lane == leader) {
      const unsigned long long k1 = lk + 1ull;
      h = ov_hash(k1) & mask;
      bool done = false;
      for (uint32_t probe = 0; probe <= mask; ++probe) {
        const unsigned long long prev = atomicCAS(&tab[h].key1, 0ull, k1);
        if (prev == 0ull || prev == k1) {
          atomicAdd(&tab[h].n[layer], (uint32_t)__popcll(grp));
          done = true;
          break;
        }
        h = (h + 1u) & mask;
      }
      if (!done) {
        err[1] = 1u;
        h = 0xffffffffu;
      }
    }
    h = (uint32_t)__shfl((int)h, leader);
    my_slot = same ? h : my_slot;
    pending = pending && !same;
  }
  if (i < N) slot_of[i] = my_slot;
}

// flag[i] = 1 iff the voxel of point i holds >= min_pts points of its own layer and of the other layer (slot_of: where k_ov_count
// found or made the voxel's entry; 0xffffffff: no entry — a point beyond the index range, or a table that overflowed: the pass is repeated)
__global__ void __launch_bounds__(kB) k_ov_flag(const uint32_t* __restrict__ slot_of, int64_t N, const OvSlot* __restrict__ tab, int layer, int64_t min_pts,
                                                uint32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const uint32_t h = slot_of[i];
  bool in = false;
  if (h != 0xffffffffu) {
    const OvSlot sl = tab[h];
    in = (int64_t)sl.n[1 - layer] >= min_pts && (min_pts <= 1 || (int64_t)sl.n[layer] >= min_pts);
  }
  flag[i] = in ? 1u : 0u;
}

// SelectByIndex of the target with its normals (k_compact) that also keeps the bounds of what it writes, in the replicas k_bounds
// uses (every slot a minimum, the maxima complemented; filled with 0xFF before): the index over the selection is built next, and
// the pass over it that would find its bounds again, with a wait of its own, is saved.  A thread folds its points first.
__global__ void __launch_bounds__(kB) k_compact_bounds(const double* __restrict__ pts, const double* __restrict__ nrm, int64_t N,
                                                       const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off, double* __restrict__ out_pts,
                                                       double* __restrict__ out_n, unsigned long long* __restrict__ slots) {
  unsigned long long* mnmx = slots + 6 * (blockIdx.x & (kExtSlots - 1));
  unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
  for (int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x; i < N; i += (int64_t)gridDim.x * kB) {
    if (!flag[i]) continue;
    const int64_t o = (int64_t)off[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = pts[3 * i + a];
      out_pts[3 * o + a] = v;
      out_n[3 * o + a] = nrm[3 * i + a];
      unsigned long long u = (unsigned long long)__double_as_longlong(v);
      u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
      lo[a] = u < lo[a] ? u : lo[a];
      hi[a] = u > hi[a] ? u : hi[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const unsigned long long l = wave_min_u64(lo[a]), h = wave_max_u64(hi[a]);
    if ((threadIdx.x & 63) == 0 && l <= h) {
      if (l < __atomic_load_n(&mnmx[a], __ATOMIC_RELAXED)) atomicMin(&mnmx[a], l);
      if (~h < __atomic_load_n(&mnmx[3 + a], __ATOMIC_RELAXED)) atomicMin(&mnmx[3 + a], ~h);
    }
  }
}

// the selection's one hand-over to the host: both counts (off[n] = the number of set flags) and the two error words (mailbox words
// 2..5), with `slots` also the folded bounds of the selected target (words 6..17: lo / hi halves, maxima un-complemented), then the
// sequence number.  One wave: a replica per lane.
__global__ void __launch_bounds__(64) k_ov_post(const uint32_t* __restrict__ fs, uint32_t* __restrict__ os, int64_t Ns, const uint32_t* __restrict__ ft,
                                                uint32_t* __restrict__ ot, int64_t Nt, const uint32_t* __restrict__ err,
                                                const unsigned long long* __restrict__ slots /*nullable*/, uint32_t* __restrict__ mailbox, uint32_t seq) {
  static_assert(kExtSlots == 64, "one replica per lane");
  if (blockIdx.x != 0) return;
  if (slots && mailbox) {
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      unsigned long long v = wave_min_u64(slots[threadIdx.x * 6 + a]);
      if (a >= 3) v = ~v;
      if (threadIdx.x == 0) {
        __hip_atomic_store(mailbox + 6 + 2 * a, (uint32_t)(v & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(mailbox + 7 + 2 * a, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (threadIdx.x != 0) return;
  const uint32_t ns = os[Ns - 1] + fs[Ns - 1], nt = ot[Nt - 1] + ft[Nt - 1];
  os[Ns] = ns;
  ot[Nt] = nt;
  if (mailbox) {
    __hip_atomic_store(mailbox + 2, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox + 3, nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox + 4, err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox + 5, err[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// What o3s_o3d_registration_icp_submaps_overlap lets the selection do on the way, when its output buffers can hold either cloud
// whole (o3s_o3d_registration_reserve): the two SelectByIndex copies are enqueued BEFORE the counts are known and the bounds of the
// selected target travel with the counts.
struct OvSelect {
  const double* tgt_normals = nullptr;
  double *out_src = nullptr, *out_tgt = nullptr, *out_tgt_n = nullptr;
  unsigned long long bounds[6] = {0, 0, 0, 0, 0, 0};  // of the selected target: minima, maxima (ordered bit patterns)
  bool have_bounds = false;  // ... and the two copies have been made
};

__global__ void __launch_bounds__(kB) k_ov_indices(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off, int64_t N,
                                                   int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i < N && flag[i]) out[off[i]] = i;
}

constexpr uint32_t kOvFirstSlots = 1u << 16;
inline uint32_t ov_slots_for(int64_t n_points) {  // room for one voxel per point at half load
  uint32_t c = 1u << 10;
  while ((int64_t)c < 2 * n_points && c < (1u << 31)) c <<= 1;
  return c;
}
inline size_t overlap_arena_bytes(int64_t Ns, int64_t Nt, uint32_t slots) {
  const size_t ns = (size_t)Ns, nt = (size_t)Nt, nmax = std::max(ns, nt);
  return Arena::pad(ns * 8) + Arena::pad(nt * 8) + Arena::pad(ns * 4) + Arena::pad(nt * 4) + Arena::pad((ns + 1) * 4) + Arena::pad((nt + 1) * 4) +
         Arena::pad(scan_temp_bytes((int64_t)nmax)) + Arena::pad((size_t)slots * sizeof(OvSlot)) + Arena::pad(64) + Arena::pad(128) +
         Arena::pad(kExtSlots * 6 * 8) + 4096;
}

size_t reg_overlap_arena_bytes(int64_t Ns, int64_t Nt) { return overlap_arena_bytes(Ns, Nt, ov_slots_for(Ns + Nt)); }

// flags + exclusive offsets of both layers on the device; counts on the host.  d_T: 16 doubles on the device.
inline int overlap_dev(OverlapWork& w, const double* d_src, int64_t Ns, const double* d_tgt, int64_t Nt, const double T[16], double voxel,
                       int64_t min_pts, uint32_t** flag_s, uint32_t** off_s, int64_t* n_s, uint32_t** flag_t, uint32_t** off_t, int64_t* n_t,
                       hipStream_t s, OvSelect* sel = nullptr) {
  *n_s = *n_t = 0;
  if (sel) sel->have_bounds = false;
  if (Ns > (int64_t)0x7fffffff || Nt > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  uint32_t slots = std::min(kOvFirstSlots, ov_slots_for(Ns + Nt));
  uint32_t *fs = nullptr, *ft = nullptr, *os = nullptr, *ot = nullptr, *err = nullptr;
  void* tmp = nullptr;
  const size_t tb_scan = scan_temp_bytes(std::max(Ns, Nt));
  for (int attempt = 0;; ++attempt) {
    if (sel) sel->have_bounds = false;  // an attempt whose table filled up has posted the bounds of an INCOMPLETE selection: never carried over
    CK(w.arena.reserve(overlap_arena_bytes(Ns, Nt, slots)));
    Arena& ar = w.arena;
    uint64_t* ks = ar.take<uint64_t>((size_t)Ns);
    uint64_t* kt = ar.take<uint64_t>((size_t)Nt);
    fs = ar.take<uint32_t>((size_t)Ns);
    ft = ar.take<uint32_t>((size_t)Nt);
    os = ar.take<uint32_t>((size_t)Ns + 1);
    ot = ar.take<uint32_t>((size_t)Nt + 1);
    tmp = ar.take<char>(tb_scan);
    OvSlot* tab = ar.take<OvSlot>((size_t)slots);
    err = ar.take<uint32_t>(16);
    double* d_T = ar.take<double>(16);
    unsigned long long* bb = ar.take<unsigned long long>(kExtSlots * 6);
    CK(hipMemsetAsync(err, 0, 8, s));
    CK(hipMemsetAsync(tab, 0, (size_t)slots * sizeof(OvSlot), s));
    CK(hipMemcpyAsync(d_T, T, 16 * sizeof(double), hipMemcpyHostToDevice, s));  // pageable source: staged before the call returns
    const double inv = 1.0 / voxel;
    hipLaunchKernelGGL(k_ov_keys, dim3(nblk(Ns)), dim3(kB), 0, s, d_src, Ns, (const double*)d_T, inv, ks, err);
    hipLaunchKernelGGL(k_ov_keys, dim3(nblk(Nt)), dim3(kB), 0, s, d_tgt, Nt, (const double*)nullptr, inv, kt, err);
    // the slots land in the offset arrays, which the scans below overwrite once the flags are made
    hipLaunchKernelGGL(k_ov_count, dim3(nblk(Ns)), dim3(kB), 0, s, ks, Ns, tab, slots - 1u, 0, os, err);
    hipLaunchKernelGGL(k_ov_count, dim3(nblk(Nt)), dim3(kB), 0, s, kt, Nt, tab, slots - 1u, 1, ot, err);
    hipLaunchKernelGGL(k_ov_flag, dim3(nblk(Ns)), dim3(kB), 0, s, (const uint32_t*)os, Ns, tab, 0, min_pts, fs);
    hipLaunchKernelGGL(k_ov_flag, dim3(nblk(Nt)), dim3(kB), 0, s, (const uint32_t*)ot, Nt, tab, 1, min_pts, ft);
    CK(hipGetLastError());
    // both scans, then ONE hand-over of the two counts and the error words (three separate waits — two counts and a copied-back error
    // word — were 40-50 us of a refinement)
    int rc = scan_flags_dev(fs, os, Ns, tmp, tb_scan, s);
    if (rc != O3S_OK) return rc;
    rc = scan_flags_dev(ft, ot, Nt, tmp, tb_scan, s);
    if (rc != O3S_OK) return rc;
    uint32_t herr[2] = {0, 0};
    {
      PinnedArea& pa = pinned_area();
      int posted = 0;
      const bool with_sel = sel && mailbox_enabled(pa);
      if (with_sel) {  // the two selections, enqueued before their sizes are known (the buffers hold either cloud whole)
        CK(hipMemsetAsync(bb, 0xFF, (size_t)kExtSlots * 6 * 8, s));
        hipLaunchKernelGGL(k_compact, dim3(nblk(Ns)), dim3(kB), 0, s, d_src, (const double*)nullptr, Ns, (const uint32_t*)fs, (const uint32_t*)os, sel->out_src,
                           (double*)nullptr, (int32_t*)nullptr);
        hipLaunchKernelGGL(k_compact_bounds, dim3(std::min(nblk(Nt), 1024u)), dim3(kB), 0, s, d_tgt, sel->tgt_normals, Nt, (const uint32_t*)ft,
                           (const uint32_t*)ot, sel->out_tgt, sel->out_tgt_n, bb);
      }
      if (mailbox_enabled(pa)) {
        const uint32_t seq = mailbox_next(pa);
        hipLaunchKernelGGL(k_ov_post, dim3(1), dim3(64), 0, s, (const uint32_t*)fs, os, Ns, (const uint32_t*)ft, ot, Nt, (const uint32_t*)err,
                           with_sel ? (const unsigned long long*)bb : (const unsigned long long*)nullptr, pa.mb_dev, seq);
        CK(hipGetLastError());
        posted = mailbox_wait(pa, seq, s);
        if (posted < 0) return O3S_ERR_HIP;
        if (posted == 1) {
          *n_s = (int64_t)__atomic_load_n(pa.mb + 2, __ATOMIC_RELAXED);
          *n_t = (int64_t)__atomic_load_n(pa.mb + 3, __ATOMIC_RELAXED);
          herr[0] = __atomic_load_n(pa.mb + 4, __ATOMIC_RELAXED);
          herr[1] = __atomic_load_n(pa.mb + 5, __ATOMIC_RELAXED);
          if (with_sel) {
            for (int a = 0; a < 6; ++a)
              sel->bounds[a] = (unsigned long long)__atomic_load_n(pa.mb + 6 + 2 * a, __ATOMIC_RELAXED) |
                               ((unsigned long long)__atomic_load_n(pa.mb + 7 + 2 * a, __ATOMIC_RELAXED) << 32);
            sel->have_bounds = *n_t > 0;
          }
        }
      } else {
        hipLaunchKernelGGL(k_ov_post, dim3(1), dim3(64), 0, s, (const uint32_t*)fs, os, Ns, (const uint32_t*)ft, ot, Nt, (const uint32_t*)err,
                           (const unsigned long long*)nullptr, (uint32_t*)nullptr, 0u);
        CK(hipGetLastError());
      }
      if (posted != 1) {
        if (sel) sel->have_bounds = false;  // the caller falls back to its own compaction and bounds pass
        uint32_t cnt[2] = {0, 0};
        CK(hipMemcpyAsync(&cnt[0], os + Ns, 4, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(&cnt[1], ot + Nt, 4, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(herr, err, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        *n_s = (int64_t)cnt[0];
        *n_t = (int64_t)cnt[1];
      }
    }
    if (herr[0]) return O3S_ERR_BAD_ARGUMENT;  // NaN / voxel index beyond +-2^20: int(floor(.)) is undefined behaviour in the reference
    if (!herr[1]) break;
    // the first table filled up (more than 2^16 overlap voxels: rare): once more with room for one voxel per point
    if (attempt > 0 || slots >= ov_slots_for(Ns + Nt)) return O3S_ERR_HIP;
    slots = ov_slots_for(Ns + Nt);
    *n_s = *n_t = 0;
  }
  *flag_s = fs;
  *off_s = os;
  *flag_t = ft;
  *off_t = ot;
  return O3S_OK;
}

}  // namespace o3s_cloud
}  // namespace

extern "C" {

int o3s_overlap_indices(int device, const double* source, int64_t Ns, const double* target, int64_t Nt, const double source_to_target[16],
                        double voxel_size, int64_t min_points_per_voxel, int64_t* idx_source, int64_t* n_source, int64_t* idx_target,
                        int64_t* n_target) {
  using namespace o3s_cloud;
  if (!n_source || !n_target || !source_to_target || Ns < 0 || Nt < 0 || !(voxel_size > 0.0) || min_points_per_voxel < 1)
    return O3S_ERR_BAD_ARGUMENT;
  *n_source = *n_target = 0;
  if (Ns == 0 || Nt == 0) return O3S_OK;
  if (!source || !target || !idx_source || !idx_target) return O3S_ERR_BAD_ARGUMENT;
  int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  hipStream_t s = nullptr;
  Buf d_s, d_t, d_is, d_it;
  CK(d_s.alloc((size_t)Ns * 24));
  CK(d_t.alloc((size_t)Nt * 24));
  CK(hipMemcpyAsync(d_s.p, source, (size_t)Ns * 24, hipMemcpyHostToDevice, s));
  CK(hipMemcpyAsync(d_t.p, target, (size_t)Nt * 24, hipMemcpyHostToDevice, s));
  RegLease area(device, s);
  OverlapWork& w = area->ov;
  uint32_t *fs, *os, *ft, *ot;
  int64_t ns = 0, nt = 0;
  rc = overlap_dev(w, d_s.as<double>(), Ns, d_t.as<double>(), Nt, source_to_target, voxel_size, min_points_per_voxel, &fs, &os, &ns, &ft, &ot, &nt, s);
  if (rc != O3S_OK) return rc;
  CK(d_is.alloc((size_t)std::max<int64_t>(ns, 1) * 8));
  CK(d_it.alloc((size_t)std::max<int64_t>(nt, 1) * 8));
  hipLaunchKernelGGL(k_ov_indices, dim3(nblk(Ns)), dim3(kB), 0, s, fs, os, Ns, d_is.as<int64_t>());
  hipLaunchKernelGGL(k_ov_indices, dim3(nblk(Nt)), dim3(kB), 0, s, ft, ot, Nt, d_it.as<int64_t>());
  CK(hipGetLastError());
  if (ns) CK(hipMemcpyAsync(idx_source, d_is.p, (size_t)ns * 8, hipMemcpyDeviceToHost, s));
  if (nt) CK(hipMemcpyAsync(idx_target, d_it.p, (size_t)nt * 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  *n_source = ns;
  *n_target = nt;
  return area.end(O3S_OK);
}

}  // extern "C"

namespace {
// the refinement of ONE pair on stream s (both submaps' own streams drained by the caller, the device current)
int refine_overlap_on(const o3s_submap* source, const o3s_submap* target, double max_dist, const double init[16], const o3s_o3d_icp_criteria* criteria,
                      double overlap_voxel_size, int64_t min_points_per_voxel, o3s_o3d_icp_result* result, double* info36, int64_t* n_overlap,
                      hipStream_t s) {
  using namespace o3s_cloud;
  int rc = O3S_OK;
  const double* sp = source->pts[source->cur].d();
  const double* tp = target->pts[target->cur].d();
  const double* tn = target->nrm[target->cur].d();
  uint32_t *fs, *os, *ft, *ot;
  int64_t ns = 0, nt = 0;
  RegLease area(target->device, s);
  // with room for either cloud whole (o3s_o3d_registration_reserve) the two SelectByIndex copies and the bounds of the selected target
  // ride on the selection's own hand-over: no wait for the counts in front of the copies, none for the bounds in front of the index
  OvSelect sel;
  const bool roomy = area->ov_src.cap >= (size_t)source->n * 24 && area->ov_tgt.cap >= (size_t)target->n * 24 && area->ov_tgtn.cap >= (size_t)target->n * 24;
  if (roomy) {
    sel.tgt_normals = tn;
    sel.out_src = area->ov_src.as<double>();
    sel.out_tgt = area->ov_tgt.as<double>();
    sel.out_tgt_n = area->ov_tgtn.as<double>();
  }
  rc = overlap_dev(area->ov, sp, source->n, tp, target->n, init, overlap_voxel_size, min_points_per_voxel, &fs, &os, &ns, &ft, &ot, &nt, s, roomy ? &sel : nullptr);
  if (rc != O3S_OK) return rc;
  if (n_overlap) {
    n_overlap[0] = ns;
    n_overlap[1] = nt;
  }
  if (ns == 0 || nt == 0) return O3S_ERR_EMPTY_REFERENCE;
  if (!sel.have_bounds) {
    // source.SelectByIndex(sourceIdxs) / target.SelectByIndex(targetIdxs) in HBM (ascending index order)
    CK(area->ov_src.alloc((size_t)ns * 24));
    CK(area->ov_tgt.alloc((size_t)nt * 24));
    CK(area->ov_tgtn.alloc((size_t)nt * 24));
    hipLaunchKernelGGL(k_compact, dim3(nblk(source->n)), dim3(kB), 0, s, sp, (const double*)nullptr, source->n, fs, os, area->ov_src.as<double>(),
                       (double*)nullptr, (int32_t*)nullptr);
    hipLaunchKernelGGL(k_compact, dim3(nblk(target->n)), dim3(kB), 0, s, tp, tn, target->n, ft, ot, area->ov_tgt.as<double>(), area->ov_tgtn.as<double>(),
                       (int32_t*)nullptr);
    CK(hipGetLastError());
  }
  rc = o3d_icp_run(area->reg, area->ov_src.as<double>(), ns, area->ov_tgt.as<double>(), area->ov_tgtn.as<double>(), nt, max_dist, init, criteria, result, s,
                   /*on_device=*/true, sel.have_bounds ? sel.bounds : nullptr);
  if (rc == O3S_OK && info36) rc = o3d_info_after_icp(area->reg, max_dist, result->transformation, info36, s);
  return area.end(rc);
}

// streams of the batch entry's lanes: made once per device, kept for the life of the process (a stream costs ~3 ms to create)
struct RefineStreams {
  static constexpr int kLanes = 4;
  std::mutex m;
  std::vector<std::vector<hipStream_t>> per_device;
  hipStream_t get(int device, int lane) {  // the device is current
    std::lock_guard<std::mutex> g(m);
    if ((size_t)device >= per_device.size()) per_device.resize((size_t)device + 1);
    std::vector<hipStream_t>& v = per_device[(size_t)device];
    while (v.size() < (size_t)kLanes) {
      hipStream_t s = nullptr;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
      v.push_back(s);
    }
    return v[(size_t)lane % v.size()];
  }
};
inline RefineStreams& refine_streams() {
  static RefineStreams* p = new RefineStreams;  // never destroyed (see RegPool)
  return *p;
}
namespace o3s_cloud {
void reg_warm_lane_streams(int device) { (void)refine_streams().get(device, 0); }  // o3s_o3d_registration_reserve_n with count > 1
}  // namespace o3s_cloud
}  // namespace

extern "C" {

int o3s_o3d_registration_icp_submaps_overlap(const o3s_submap* source, const o3s_submap* target, double max_dist, const double init[16],
                                             const o3s_o3d_icp_criteria* criteria, double overlap_voxel_size, int64_t min_points_per_voxel,
                                             o3s_o3d_icp_result* result, double* info36, int64_t* n_overlap) {
  if (!source || !target || !init || !result || !(max_dist > 0.0) || !(overlap_voxel_size > 0.0) || min_points_per_voxel < 1)
    return O3S_ERR_BAD_ARGUMENT;
  if (source->device != target->device) return O3S_ERR_BAD_ARGUMENT;
  if (n_overlap) n_overlap[0] = n_overlap[1] = 0;
  if (const int rs_ = submap_settle(source); rs_ != O3S_OK) return rs_;  // a pending insert is completed first
  if (const int rt_ = submap_settle(target); rt_ != O3S_OK) return rt_;
  if (source->n == 0 || target->n == 0) return O3S_ERR_EMPTY_REFERENCE;
  if (target->has_normals != 1) return O3S_ERR_BAD_SHAPE;  // "requires target pointcloud to have normals"
  int rc = set_dev(target);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(source->stream));
  CK(hipStreamSynchronize(target->stream));
  return refine_overlap_on(source, target, max_dist, init, criteria, overlap_voxel_size, min_points_per_voxel, result, info36, n_overlap, target->stream);
}

int o3s_o3d_registration_icp_submaps_overlap_batch(int32_t n, const o3s_submap* const* sources, const o3s_submap* const* targets, double max_dist,
                                                   const double* inits, const o3s_o3d_icp_criteria* criteria, double overlap_voxel_size,
                                                   int64_t min_points_per_voxel, o3s_o3d_icp_result* results, double* infos, int64_t* n_overlaps,
                                                   int32_t* statuses) {
  if (n < 0 || (n > 0 && (!sources || !targets || !inits || !results || !statuses)) || !(max_dist > 0.0) || !(overlap_voxel_size > 0.0) ||
      min_points_per_voxel < 1)
    return O3S_ERR_BAD_ARGUMENT;
  if (n == 0) return O3S_OK;
  int device = -1;
  for (int32_t k = 0; k < n; ++k) {
    if (!sources[k] || !targets[k] || sources[k]->device != targets[k]->device) return O3S_ERR_BAD_ARGUMENT;
    if (device < 0) device = targets[k]->device;
    if (targets[k]->device != device) return O3S_ERR_BAD_ARGUMENT;  // one device per call (the work areas and the lanes' streams are per device)
  }
  if (hipSetDevice(device) != hipSuccess) return O3S_ERR_HIP;
  for (int32_t k = 0; k < n; ++k) {  // every submap involved is complete before any lane reads it
    if (const int rs_ = submap_settle(sources[k]); rs_ != O3S_OK) return rs_;
    if (const int rt_ = submap_settle(targets[k]); rt_ != O3S_OK) return rt_;
    CK(hipStreamSynchronize(sources[k]->stream));
    CK(hipStreamSynchronize(targets[k]->stream));
  }
  const int lanes = std::min<int>(RefineStreams::kLanes, n);
  std::vector<hipStream_t> streams((size_t)lanes);
  for (int l = 0; l < lanes; ++l) {
    // one pair: the target's own stream, like the single call (no lane stream is needed — or made: the first use of the lanes'
    // streams creates them, ~3 ms each, unless o3s_o3d_registration_reserve_n has done so ahead of time)
    streams[(size_t)l] = lanes == 1 ? targets[0]->stream : refine_streams().get(device, l);
    if (!streams[(size_t)l]) return O3S_ERR_HIP;
  }
  auto one = [&](int32_t k, hipStream_t s) {
    if (n_overlaps) n_overlaps[2 * k] = n_overlaps[2 * k + 1] = 0;
    if (sources[k]->n == 0 || targets[k]->n == 0) return (int)O3S_ERR_EMPTY_REFERENCE;
    if (targets[k]->has_normals != 1) return (int)O3S_ERR_BAD_SHAPE;
    return refine_overlap_on(sources[k], targets[k], max_dist, inits + 16 * (size_t)k, criteria, overlap_voxel_size, min_points_per_voxel, &results[k],
                             infos ? infos + 36 * (size_t)k : nullptr, n_overlaps ? n_overlaps + 2 * (size_t)k : nullptr, s);
  };
  auto lane = [&](int l) {
    (void)hipSetDevice(device);
    for (int32_t k = l; k < n; k += lanes) statuses[k] = one(k, streams[(size_t)l]);
    (void)hipStreamSynchronize(streams[(size_t)l]);
  };
  std::vector<std::thread> pool;
  for (int l = 1; l < lanes; ++l) pool.emplace_back(lane, l);
  lane(0);
  for (auto& th : pool) th.join();
  for (int32_t k = 0; k < n; ++k)
    if (statuses[k] != O3S_OK && statuses[k] != O3S_ERR_EMPTY_REFERENCE) return statuses[k];
  return O3S_OK;
}

}  // extern "C"
