// dense_map_impl.h — device-resident dense map (C ABI: include/o3s_dense_map.h), gfx950 only.  Included at the end of
// cloud_ops.hip after submap_impl.h (one TU: shared kernels, one instantiation of the rocPRIM sort / scan).
//
// The reference's VoxelizedPointCloud is an unordered_map<Vector3i, AggregatedVoxel> filled point by point
// (O3S/src/Voxel.cpp:66-88).  Here it is an open-addressing table in HBM (linear probing, load <= 1/2, tombstones):
//   keys[cap] u64 | cnt[cap] i32 | sum[cap][6] f64 (position xyz, normal xyz)
// An insert sorts the new points by packed voxel key (stable, so a voxel's points stay in input order), and ONE lane per
// distinct key claims / finds the slot (64-bit CAS) and adds its points one after the other — the same additions in the
// same order as the reference's loop, hence bit-identical sums, and no two lanes ever touch the same slot.
// Carving (O3S/src/helpers.cpp:360-390) is a lookup-only ray march: one lane per ray, existence tests against the table,
// hits flagged per slot, and a second kernel turns the flagged slots into tombstones.
#pragma once
#include "../../include/o3s_dense_map.h"

#include "cloud_dev.h"

namespace {

constexpr uint64_t kDmEmpty = ~0ull, kDmTomb = ~0ull - 1ull;
constexpr int32_t kDmBias = 1 << 20;
constexpr int kDmMaxOff = 16;  // neighbourhood offsets per axis: radius / voxel <= 7.5

__device__ __forceinline__ uint64_t dm_pack(int32_t x, int32_t y, int32_t z) {
  return ((uint64_t)(uint32_t)(z + kDmBias) << 42) | ((uint64_t)(uint32_t)(y + kDmBias) << 21) | (uint64_t)(uint32_t)(x + kDmBias);
}
__device__ __forceinline__ bool dm_in_range(double f) { return f >= -(double)kDmBias && f < (double)kDmBias; }  // false for NaN
__device__ __forceinline__ uint64_t dm_hash(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}
__device__ __forceinline__ uint64_t dm_home(uint64_t key, uint64_t mask) { return dm_hash(key) & mask; }
// Occupancy filter for the carving march, rebuilt from the table before every carve: one 64-bit word per 4 x 4 x 4 block
// of voxels (word = hash of the block, bit = position inside), so a stop's whole neighbourhood is answered by the few
// words its blocks hash to and the table is probed only where a bit is set.  Colliding blocks OR their bits: a set bit
// may be a false alarm (the table decides), a clear bit is always right.  A second array under an independent hash is
// consulted only when the first says "maybe": measured on the bench, one array alone sent 5 % of all free-space
// questions on to the table (65 M probes per carve for 0.1 M voxels actually found).
__device__ __forceinline__ uint64_t dm_block(uint64_t key) {
  return ((key >> 2) & 0x7ffffull) | (((key >> 23) & 0x7ffffull) << 19) | (((key >> 44) & 0x7ffffull) << 38);
}
__device__ __forceinline__ uint64_t dm_hash2(uint64_t b) { return dm_hash(b ^ 0x9e3779b97f4a7c15ull); }
__device__ __forceinline__ unsigned dm_bit(uint64_t key) {
  return (unsigned)((key & 3ull) | (((key >> 21) & 3ull) << 2) | (((key >> 42) & 3ull) << 4));
}
__device__ __forceinline__ uint64_t dm_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// slot of `key`, or -1; the table always holds an empty slot (load <= 1/2), so the probe ends
__device__ __forceinline__ int64_t dm_find(const uint64_t* __restrict__ keys, uint64_t mask, uint64_t key) {
  uint64_t h = dm_home(key, mask);
  for (;;) {
    const uint64_t k = dm_load(keys + h);
    if (k == key) return (int64_t)h;
    if (k == kDmEmpty) return -1;
    h = (h + 1) & mask;
  }
}

// one lane of a wave adds the wave's count to a global counter (same-address atomics serialise at ~11 ns each)
__device__ __forceinline__ void wave_count(bool flag, uint32_t* counter) {
  const uint64_t b = __ballot(flag);
  if (b && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(b)) atomicAdd(counter, (uint32_t)__builtin_popcountll(b));
}

// getVoxelIdx(p, InverseVoxelSize) (VoxelHashMap.hpp:48-51) -> packed key; out-of-range points raise err and sort last
__global__ void __launch_bounds__(kB) k_dm_keys(const double* __restrict__ pts, int64_t N, double inv, uint64_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals, uint32_t* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const double fx = floor(pts[3 * i] * inv), fy = floor(pts[3 * i + 1] * inv), fz = floor(pts[3 * i + 2] * inv);
  vals[i] = (uint32_t)i;
  if (dm_in_range(fx) && dm_in_range(fy) && dm_in_range(fz)) {
    keys[i] = dm_pack((int32_t)fx, (int32_t)fy, (int32_t)fz);
  } else {
    keys[i] = kDmEmpty;
    *err = 1u;
  }
}

// VoxelizedPointCloud::insert (Voxel.cpp:66-88) on key-sorted points: one lane per distinct key.
// counters: [0] slots newly occupied, [1] of those, how many re-used a tombstone
__global__ void __launch_bounds__(kB) k_dm_insert(const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ svals, int64_t N,
                                                  const double* __restrict__ pts, const double* __restrict__ nrm, uint64_t* __restrict__ keys,
                                                  int32_t* __restrict__ cnt, double* __restrict__ sum, uint64_t mask,
                                                  uint32_t* __restrict__ counters) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  bool is_new = false, from_tomb = false;
  if (i < N) {
    const uint64_t key = skeys[i];
    if (key != kDmEmpty && (i == 0 || skeys[i - 1] != key)) {
      // find the key, else claim the first tombstone seen on the probe path, else the empty slot that ended the probe
      const uint64_t h0 = dm_home(key, mask);
      uint64_t h = h0;
      int64_t tomb = -1, slot = -1;
      for (;;) {
        const uint64_t k = dm_load(keys + h);
        if (k == key) {
          slot = (int64_t)h;
          break;
        }
        if (k == kDmTomb && tomb < 0) tomb = (int64_t)h;
        if (k == kDmEmpty) {
          const uint64_t target = tomb >= 0 ? (uint64_t)tomb : h;
          const uint64_t expect = tomb >= 0 ? kDmTomb : kDmEmpty;
          const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(keys + target), (unsigned long long)expect, (unsigned long long)key);
          if (old == expect) {
            slot = (int64_t)target;
            is_new = true;
            from_tomb = tomb >= 0;
            break;
          }
          h = h0;  // another key took that slot: probe again (this key is still ours alone to insert)
          tomb = -1;
          continue;
        }
        h = (h + 1) & mask;
      }
      double s[6] = {0, 0, 0, 0, 0, 0};
      int32_t c = 0;
      if (!is_new) {
        c = cnt[slot];
#pragma unroll
        for (int a = 0; a < 6; ++a) s[a] = sum[6 * slot + a];
      }
      for (int64_t j = i; j < N && skeys[j] == key; ++j) {  // aggregatePoint, aggregateNormal (Voxel.cpp:27-33), input order
        const int64_t p = svals[j];
        s[0] += pts[3 * p];
        s[1] += pts[3 * p + 1];
        s[2] += pts[3 * p + 2];
        ++c;
        if (nrm) {
          s[3] += nrm[3 * p];
          s[4] += nrm[3 * p + 1];
          s[5] += nrm[3 * p + 2];
        }
      }
      cnt[slot] = c;
#pragma unroll
      for (int a = 0; a < 6; ++a) sum[6 * slot + a] = s[a];
    }
  }
  wave_count(is_new, counters);
  wave_count(from_tomb, counters + 1);
}

// moves every live slot into a fresh (all-empty) table
__global__ void __launch_bounds__(kB) k_dm_rehash(const uint64_t* __restrict__ okeys, const int32_t* __restrict__ ocnt, const double* __restrict__ osum,
                                                  int64_t ocap, uint64_t* __restrict__ keys, int32_t* __restrict__ cnt, double* __restrict__ sum,
                                                  uint64_t mask) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= ocap) return;
  const uint64_t key = okeys[i];
  if (key == kDmEmpty || key == kDmTomb) return;
  uint64_t h = dm_home(key, mask);
  for (;;) {
    if (dm_load(keys + h) == kDmEmpty &&
        atomicCAS(reinterpret_cast<unsigned long long*>(keys + h), (unsigned long long)kDmEmpty, (unsigned long long)key) == kDmEmpty)
      break;
    h = (h + 1) & mask;
  }
  cnt[h] = ocnt[i];
#pragma unroll
  for (int a = 0; a < 6; ++a) sum[6 * h + a] = osum[6 * i + a];
}

// toPointCloud (Voxel.cpp:90-114): voxels with numAggregatedPoints_ > 0
__global__ void __launch_bounds__(kB) k_dm_live(const uint64_t* __restrict__ keys, const int32_t* __restrict__ cnt, int64_t cap,
                                                uint32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= cap) return;
  const uint64_t k = keys[i];
  flag[i] = (k != kDmEmpty && k != kDmTomb && cnt[i] > 0) ? 1u : 0u;
}
__global__ void __launch_bounds__(kB) k_dm_collect(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ off,
                                                   int64_t cap, uint64_t* __restrict__ okeys, uint32_t* __restrict__ oslots) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= cap || !flag[i]) return;
  okeys[off[i]] = keys[i];
  oslots[off[i]] = (uint32_t)i;
}
__global__ void __launch_bounds__(kB) k_dm_emit(const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ sslots, int64_t V,
                                                const int32_t* __restrict__ cnt, const double* __restrict__ sum, double* __restrict__ out_p,
                                                double* __restrict__ out_n, int32_t* __restrict__ out_k, int32_t* __restrict__ out_c) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= V) return;
  const uint64_t key = skeys[i];
  const int64_t s = sslots[i];
  const int32_t c = cnt[s];
  const double d = (double)c;  // aggregatedPosition_ / numAggregatedPoints_ (Voxel.cpp:18-23)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    out_p[3 * i + a] = sum[6 * s + a] / d;
    out_n[3 * i + a] = sum[6 * s + 3 + a] / d;
  }
  out_k[3 * i] = (int32_t)(key & 0x1fffffull) - kDmBias;
  out_k[3 * i + 1] = (int32_t)((key >> 21) & 0x1fffffull) - kDmBias;
  out_k[3 * i + 2] = (int32_t)((key >> 42) & 0x1fffffull) - kDmBias;
  out_c[i] = c;
}

// VoxelizedPointCloud::transform (Voxel.cpp:49-64): Isometry3d * Vector3d = translation + linear * v on both sums
__global__ void __launch_bounds__(kB) k_dm_transform(const uint64_t* __restrict__ keys, const int32_t* __restrict__ cnt, double* __restrict__ sum,
                                                     int64_t cap, const double* __restrict__ Tm) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= cap) return;
  const uint64_t k = keys[i];
  if (k == kDmEmpty || k == kDmTomb || cnt[i] <= 0) return;
  double T[16];
#pragma unroll
  for (int a = 0; a < 16; ++a) T[a] = Tm[a];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const double x = sum[6 * i + 3 * g], y = sum[6 * i + 3 * g + 1], z = sum[6 * i + 3 * g + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) sum[6 * i + 3 * g + r] = T[12 + r] + ((T[r] * x + T[4 + r] * y) + T[8 + r] * z);
  }
}

// removeDuplicatePointsWithinSameVoxels (Voxel.cpp:162-192) on key-sorted points: the first point of every voxel
__global__ void __launch_bounds__(kB) k_dm_heads(const uint64_t* __restrict__ skeys, int64_t N, uint32_t* __restrict__ head) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= N) return;
  const uint64_t k = skeys[i];
  head[i] = (k != kDmEmpty && (i == 0 || skeys[i - 1] != k)) ? 1u : 0u;
}
__global__ void __launch_bounds__(kB) k_dm_first(const uint32_t* __restrict__ svals, const uint32_t* __restrict__ head, const uint32_t* __restrict__ off,
                                                 int64_t N, uint32_t* __restrict__ first) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i < N && head[i]) first[off[i]] = svals[i];
}

__global__ void __launch_bounds__(kB) k_dm_build_filter(const uint64_t* __restrict__ keys, int64_t cap, unsigned long long* __restrict__ words,
                                                        uint64_t wmask) {
  const int64_t i = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (i >= cap) return;
  const uint64_t k = keys[i];
  if (k == kDmEmpty || k == kDmTomb) return;
  const uint64_t b = dm_block(k);
  atomicOr(words + (dm_hash(b) & wmask), 1ull << dm_bit(k));
  atomicOr(words + (wmask + 1) + (dm_hash2(b) & wmask), 1ull << dm_bit(k));  // the second array follows the first
}

struct DmOffsets {
  int n;
  double d[kDmMaxOff];  // the values the reference's `for (dx = -r; dx <= r; dx += step)` loop takes (VoxelHashMap.cpp:24)
};

// getKeysOfCarvedPoints (helpers.cpp:360-390) with getVoxelsWithinPointNeighborhood (VoxelHashMap.cpp:13-46) inlined:
// one lane per ray, the reference's sequential march; every existing voxel it names gets rm[slot] = 1.
// A packed key is the OR of three per-axis parts, so each stop forms per axis (n values each): the shifted key part
// (bit 63 set when the index is out of range) and the squared offset from the voxel centre — 3 n divisions instead of
// n^3 — and the n^3 combinations cost an OR, an add and a compare each.  NX > 0: n == NX is known at compile time, the
// x tables stay in registers and the innermost loop is unrolled; NX == 0: any n <= kDmMaxOff.
// kDmRayLanes lanes share a ray: lane j takes stops j, j + kDmRayLanes, ... (every lane forms `distance` by the same
// repeated addition as the reference, so the stops are the same numbers).  120 k rays alone leave the chip at 2 waves per
// SIMD waiting on dependent look-ups; eight lanes per ray fill it.
constexpr int kDmRayLanes = 8;
constexpr uint64_t kDmBad = 1ull << 63;
#ifdef O3S_DM_STATS
__device__ unsigned long long g_dm_stats[4];  // stops, in-radius combinations, filter hits, voxels found
#endif
template <int NX>
__global__ void __launch_bounds__(kB) k_dm_carve_rays(const double* __restrict__ scan, const uint32_t* __restrict__ first, int64_t n_first, double sx,
                                                      double sy, double sz, double voxel, double radius, double r2lo, double r2hi, double step,
                                                      double max_len, double trunc, DmOffsets off, const uint64_t* __restrict__ keys, uint64_t mask,
                                                      const uint64_t* __restrict__ words, uint64_t wmask, uint8_t* __restrict__ rm) {
  constexpr int CAPX = NX > 0 ? NX : kDmMaxOff;
  const int64_t t = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (t >= n_first * kDmRayLanes) return;
  const int sub = (int)(t % kDmRayLanes);
  const int64_t i = first[t / kDmRayLanes];
  const double dx = scan[3 * i] - sx, dy = scan[3 * i + 1] - sy, dz = scan[3 * i + 2] - sz;
  const double length = sqrt((dx * dx + dy * dy) + dz * dz);
  if (!(length > 0.0)) return;  // NaN direction (a return at the sensor, or a NaN point): no voxel can be addressed
  const double ux = dx / length, uy = dy / length, uz = dz / length;
  const double max_path = fmax(step, fmin(length - trunc, max_len));
  const double half = voxel * 0.5;
  const int n = NX > 0 ? NX : off.n;
  uint64_t blk[2] = {~0ull, ~0ull}, wrd[2] = {0, 0};  // the two filter words used last (x runs straddle two blocks)
  auto maybe_there = [&](uint64_t key) -> bool {
    const uint64_t b = dm_block(key);
    if (b != blk[0]) {
      if (b == blk[1]) {
        const uint64_t tb = blk[0], tw = wrd[0];
        blk[0] = blk[1];
        wrd[0] = wrd[1];
        blk[1] = tb;
        wrd[1] = tw;
      } else {
        blk[1] = blk[0];
        wrd[1] = wrd[0];
        blk[0] = b;
        wrd[0] = words[dm_hash(b) & wmask];
      }
    }
    return (wrd[0] >> dm_bit(key)) & 1ull;
  };
#ifdef O3S_DM_STATS
  unsigned long long st_stops = 0, st_in = 0, st_fhit = 0, st_found = 0;
#endif
  auto probe = [&](uint64_t key) {
#ifdef O3S_DM_STATS
    ++st_in;
#endif
    if (!maybe_there(key)) return;
    if (!((words[(wmask + 1) + (dm_hash2(dm_block(key)) & wmask)] >> dm_bit(key)) & 1ull)) return;
    const int64_t s = dm_find(keys, mask, key);
#ifdef O3S_DM_STATS
    ++st_fhit;
    if (s >= 0) ++st_found;
#endif
    if (s >= 0) rm[s] = 1;
  };
  // key part and squared centre offset of one test coordinate
  auto axis = [&](double tq, int shift, uint64_t& part, double& e2) {
    const double f = floor(tq / voxel);                          // getVoxelIdx(p, voxelSize): the dividing form
    const double e = tq - ((double)(int32_t)f * voxel + half);   // getVoxelCenter = key * voxel + voxel * 0.5
    e2 = e * e;
    part = dm_in_range(f) ? (uint64_t)(uint32_t)((int32_t)f + kDmBias) << shift : kDmBad;
  };
  double distance = 0.0;
  for (int k = 0; k < sub && distance < max_path; ++k) distance += step;  // this lane's first stop
  while (distance < max_path) {
    const double cx = distance * ux + sx, cy = distance * uy + sy, cz = distance * uz + sz;
    uint64_t px[CAPX], py[kDmMaxOff], pz[kDmMaxOff];
    double ex2[CAPX], ey2[kDmMaxOff], ez2[kDmMaxOff];
    if constexpr (NX > 0) {
#pragma unroll
      for (int q = 0; q < NX; ++q) axis(cx + off.d[q], 0, px[q], ex2[q]);
    } else {
      for (int q = 0; q < n; ++q) axis(cx + off.d[q], 0, px[q], ex2[q]);
    }
    for (int q = 0; q < n; ++q) {
      axis(cy + off.d[q], 21, py[q], ey2[q]);
      axis(cz + off.d[q], 42, pz[q], ez2[q]);
    }
    uint64_t ckey;
    {
      uint64_t a, b, c;
      double unused;
      axis(cx, 0, a, unused);
      axis(cy, 21, b, unused);
      axis(cz, 42, c, unused);
      ckey = a | b | c;
    }
    bool centre_added = false;
    for (int qz = 0; qz < n; ++qz)
      for (int qy = 0; qy < n; ++qy) {
        const uint64_t pyz = py[qy] | pz[qz];
        const double y2 = ey2[qy], z2 = ez2[qz];
        auto combo = [&](int qx) {
          // (testPoint - center).norm() <= radius: the square root is only taken inside a 1e-14 band around radius^2
          const double v = (ex2[qx] + y2) + z2;
          if (!(v <= r2lo || (v <= r2hi && sqrt(v) <= radius))) return;
          const uint64_t key = px[qx] | pyz;
          if (key == ckey) centre_added = true;
          if (key & kDmBad) return;  // a voxel outside the index range cannot be in the map
          probe(key);
        };
        if constexpr (NX > 0) {
#pragma unroll
          for (int qx = 0; qx < NX; ++qx) combo(qx);
        } else {
          for (int qx = 0; qx < n; ++qx) combo(qx);
        }
      }
    if (!centre_added && !(ckey & kDmBad)) probe(ckey);
    for (int k = 0; k < kDmRayLanes && distance < max_path; ++k) distance += step;  // the lane's next stop
#ifdef O3S_DM_STATS
    ++st_stops;
#endif
  }
#ifdef O3S_DM_STATS
  atomicAdd(g_dm_stats + 0, st_stops);
  atomicAdd(g_dm_stats + 1, st_in);
  atomicAdd(g_dm_stats + 2, st_fhit);
  atomicAdd(g_dm_stats + 3, st_found);
#endif
}

// removeKey for every flagged slot; one lane looks at eight flags at a time (cap is a power of two >= 2^16)
__global__ void __launch_bounds__(kB) k_dm_apply_remove(const uint8_t* __restrict__ rm, uint64_t* __restrict__ keys, int64_t cap,
                                                        uint32_t* __restrict__ counter /*16 replicas, summed by the host*/) {
  const int64_t g = (int64_t)blockIdx.x * kB + threadIdx.x;
  if (g * 8 >= cap) return;
  const uint64_t w = reinterpret_cast<const uint64_t*>(rm)[g];
  if (!w) return;
  uint32_t n = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b)
    if ((w >> (8 * b)) & 0xffull) {
      keys[g * 8 + b] = kDmTomb;
      ++n;
    }
  atomicAdd(counter + (blockIdx.x & 15), n);
}

}  // namespace

struct o3s_dense_map {
  int device = 0;
  double voxel = 0.0, inv = 0.0;
  hipStream_t stream = nullptr;
  void *keys = nullptr, *cnt = nullptr, *sum = nullptr;  // the table
  int64_t cap = 0, live = 0, tomb = 0;
  int has_normals = 0;
  int64_t n_scans_inserted = 0;  // nScansInsertedDenseMap_
  mutable DArr in_p, in_n, crop_p, crop_n, tf_p, tf_n, d_T, d_ctr, out;
  mutable Arena arena;
  uint64_t* K() const { return reinterpret_cast<uint64_t*>(keys); }
  int32_t* C() const { return reinterpret_cast<int32_t*>(cnt); }
  double* S() const { return reinterpret_cast<double*>(sum); }
};

namespace {

void dm_free_table(o3s_dense_map* m) {
  if (m->keys) (void)hipFree(m->keys);
  if (m->cnt) (void)hipFree(m->cnt);
  if (m->sum) (void)hipFree(m->sum);
  m->keys = m->cnt = m->sum = nullptr;
  m->cap = m->live = m->tomb = 0;
}

// makes room for `add` more voxels at load <= 1/2, re-hashing into a larger (or tombstone-free) table when needed
int dm_reserve(o3s_dense_map* m, int64_t add) {
  hipStream_t s = m->stream;
  if (m->cap > 0 && 2 * (m->live + m->tomb + add) <= m->cap) return O3S_OK;
  int64_t ncap = 1 << 16;
  while (ncap < 4 * (m->live + add)) ncap <<= 1;
  Buf nk, nc, ns;  // released on every early return; handed to the map at the end
  CK(nk.alloc((size_t)ncap * 8));
  CK(nc.alloc((size_t)ncap * 4));
  CK(ns.alloc((size_t)ncap * 48));
  CK(hipMemsetAsync(nk.p, 0xff, (size_t)ncap * 8, s));
  if (m->cap > 0 && m->live > 0) {
    hipLaunchKernelGGL(k_dm_rehash, dim3(nblk(m->cap)), dim3(kB), 0, s, m->K(), m->C(), m->S(), m->cap, nk.as<uint64_t>(), nc.as<int32_t>(),
                       ns.as<double>(), (uint64_t)(ncap - 1));
    CK(hipGetLastError());
  }
  CK(hipStreamSynchronize(s));
  const int64_t live = m->live;
  dm_free_table(m);
  m->keys = nk.p;
  m->cnt = nc.p;
  m->sum = ns.p;
  nk.p = nc.p = ns.p = nullptr;
  m->cap = ncap;
  m->live = live;
  return O3S_OK;
}

// keys of a device cloud, stably sorted; returns the sorted keys / values inside the arena
int dm_sorted_keys(o3s_dense_map* m, const double* d_pts, int64_t N, size_t extra_bytes, uint64_t** skeys, uint32_t** svals, void** tmp,
                   size_t* tmp_bytes) {
  hipStream_t s = m->stream;
  const size_t n = (size_t)N;
  const size_t tb = std::max(sort_temp_bytes(N), scan_temp_bytes(N));
  CK(m->arena.reserve(2 * Arena::pad(n * 8) + 2 * Arena::pad(n * 4) + Arena::pad(tb) + extra_bytes + 4096));
  uint64_t* k1 = m->arena.take<uint64_t>(n);
  uint64_t* k2 = m->arena.take<uint64_t>(n);
  uint32_t* v1 = m->arena.take<uint32_t>(n);
  uint32_t* v2 = m->arena.take<uint32_t>(n);
  *tmp = m->arena.take<char>(tb);
  *tmp_bytes = tb;
  CK(m->d_ctr.ensure(64, 0, s));
  CK(hipMemsetAsync(m->d_ctr.p, 0, 64, s));
  uint32_t* ctr = reinterpret_cast<uint32_t*>(m->d_ctr.p);
  hipLaunchKernelGGL(k_dm_keys, dim3(nblk(N)), dim3(kB), 0, s, d_pts, N, m->inv, k1, v1, ctr + 8);
  size_t stb = tb;
  CK(rocprim::radix_sort_pairs(*tmp, stb, k1, k2, v1, v2, n, 0, 64, s));
  *skeys = k2;
  *svals = v2;
  return O3S_OK;
}

int dm_read_counters(o3s_dense_map* m, uint32_t out[16]) {
  CK(hipMemcpyAsync(out, m->d_ctr.p, 64, hipMemcpyDeviceToHost, m->stream));
  CK(hipStreamSynchronize(m->stream));
  return O3S_OK;
}

// VoxelizedPointCloud::insert on a device cloud
int dm_insert_dev(o3s_dense_map* m, const double* d_pts, const double* d_nrm, int64_t N) {
  if (N == 0) return O3S_OK;
  if (N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  hipStream_t s = m->stream;
  int rc = dm_reserve(m, N);
  if (rc != O3S_OK) return rc;
  uint64_t* sk = nullptr;
  uint32_t* sv = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  rc = dm_sorted_keys(m, d_pts, N, 0, &sk, &sv, &tmp, &tb);
  if (rc != O3S_OK) return rc;
  uint32_t ctr[16];
  rc = dm_read_counters(m, ctr);  // the range check has to be known before the table is touched
  if (rc != O3S_OK) return rc;
  if (ctr[8]) return O3S_ERR_BAD_ARGUMENT;
  hipLaunchKernelGGL(k_dm_insert, dim3(nblk(N)), dim3(kB), 0, s, sk, sv, N, d_pts, d_nrm, m->K(), m->C(), m->S(), (uint64_t)(m->cap - 1),
                     reinterpret_cast<uint32_t*>(m->d_ctr.p));
  CK(hipGetLastError());
  rc = dm_read_counters(m, ctr);
  if (rc != O3S_OK) return rc;
  m->live += ctr[0];
  m->tomb -= ctr[1];
  if (d_nrm) m->has_normals = 1;  // isHasNormals_ (Voxel.cpp:80-82)
  return O3S_OK;
}

int dm_upload(o3s_dense_map* m, const double* pts, const double* normals, int64_t N) {
  hipStream_t s = m->stream;
  CK(m->in_p.ensure((size_t)N * 24, 0, s));
  CK(hipMemcpyAsync(m->in_p.p, pts, (size_t)N * 24, hipMemcpyHostToDevice, s));
  if (normals) {
    CK(m->in_n.ensure((size_t)N * 24, 0, s));
    CK(hipMemcpyAsync(m->in_n.p, normals, (size_t)N * 24, hipMemcpyHostToDevice, s));
  }
  return O3S_OK;
}

// Submap::carve on the dense map, scan already on the device
int dm_carve_dev(o3s_dense_map* m, const o3s_dense_carving_params* p, const double* d_scan, int64_t N, const double sensor[3], int64_t* n_removed) {
  if (n_removed) *n_removed = 0;
  if (m->live == 0 || N == 0) return O3S_OK;  // "if (cloud->empty() ...) return" (Submap.cpp:148-150)
  if (N > (int64_t)0x7fffffff) return O3S_ERR_BAD_ARGUMENT;
  const double radius = p->neighborhood_radius_dense_map, step = 2.0 * radius;
  // the reference's march never ends for a zero step; refuse what would not terminate or not fit the offset table
  if (!(step > 0.0) || !std::isfinite(step) || !std::isfinite(p->max_raytracing_length) || !std::isfinite(p->truncation_distance) ||
      p->max_raytracing_length / step > 1.0e6)
    return O3S_ERR_BAD_ARGUMENT;
  DmOffsets off{};
  for (double d = -radius; d <= radius; d += m->voxel) {
    if (off.n == kDmMaxOff) return O3S_ERR_BAD_ARGUMENT;
    off.d[off.n++] = d;
  }
  hipStream_t s = m->stream;
  const size_t n = (size_t)N, cap = (size_t)m->cap;
  uint64_t* sk = nullptr;
  uint32_t* sv = nullptr;
  void* tmp = nullptr;
  size_t tb = 0;
  int64_t nwords = 4096;  // about one word per four voxels: a few MB, L2 / Infinity-Cache resident
  while (nwords * 4 < m->live) nwords <<= 1;
  int rc = dm_sorted_keys(m, d_scan, N, 2 * Arena::pad(n * 4) + Arena::pad((n + 1) * 4) + Arena::pad(cap) + Arena::pad((size_t)nwords * 16), &sk, &sv,
                          &tmp, &tb);
  if (rc != O3S_OK) return rc;
  uint32_t* head = m->arena.take<uint32_t>(n);
  uint32_t* ord = m->arena.take<uint32_t>(n + 1);
  uint32_t* first = m->arena.take<uint32_t>(n);
  uint8_t* rm = m->arena.take<uint8_t>(cap);
  unsigned long long* words = m->arena.take<unsigned long long>((size_t)nwords * 2);  // two arrays of nwords
  CK(hipMemsetAsync(words, 0, (size_t)nwords * 16, s));
  hipLaunchKernelGGL(k_dm_build_filter, dim3(nblk(m->cap)), dim3(kB), 0, s, m->K(), m->cap, words, (uint64_t)(nwords - 1));
  hipLaunchKernelGGL(k_dm_heads, dim3(nblk(N)), dim3(kB), 0, s, sk, N, head);
  int64_t n_first = 0;
  rc = scan_flags(head, ord, N, tmp, tb, &n_first, s);
  if (rc != O3S_OK) return rc;
  uint32_t ctr[16];
  rc = dm_read_counters(m, ctr);
  if (rc != O3S_OK) return rc;
  if (ctr[8]) return O3S_ERR_BAD_ARGUMENT;  // a NaN / out-of-range scan point has no voxel key
  if (n_first == 0) return O3S_OK;
  hipLaunchKernelGGL(k_dm_first, dim3(nblk(N)), dim3(kB), 0, s, sv, head, ord, N, first);
  CK(hipMemsetAsync(rm, 0, cap, s));
  {
    // sqrt(v) <= radius is certain below r2lo and impossible above r2hi (sqrt is monotone and correctly rounded)
    const double r2 = radius * radius, r2lo = r2 * (1.0 - 1.0e-14), r2hi = r2 * (1.0 + 1.0e-14);
    const dim3 grid(nblk(n_first * kDmRayLanes)), block(kB);
#define O3S_DM_CARVE(NX)                                                                                                                     \
  hipLaunchKernelGGL(k_dm_carve_rays<NX>, grid, block, 0, s, d_scan, first, n_first, sensor[0], sensor[1], sensor[2], m->voxel, radius, r2lo, r2hi, \
                     step, p->max_raytracing_length, p->truncation_distance, off, m->K(), (uint64_t)(m->cap - 1),                            \
                     reinterpret_cast<const uint64_t*>(words), (uint64_t)(nwords - 1), rm)
    switch (off.n) {
      case 1: O3S_DM_CARVE(1); break;
      case 2: O3S_DM_CARVE(2); break;
      case 3: O3S_DM_CARVE(3); break;
      case 4: O3S_DM_CARVE(4); break;
      case 5: O3S_DM_CARVE(5); break;
      case 6: O3S_DM_CARVE(6); break;
      case 7: O3S_DM_CARVE(7); break;
      default: O3S_DM_CARVE(0); break;
    }
#undef O3S_DM_CARVE
#ifdef O3S_DM_STATS
    unsigned long long st[4];
    CK(hipStreamSynchronize(s));
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_dm_stats), sizeof(st)));
    fprintf(stderr, "[dm stats, cumulative] rays %lld stops %llu in-radius %llu filter-hits %llu found %llu\n", (long long)n_first, st[0], st[1], st[2], st[3]);
#endif
  }
  CK(hipMemsetAsync(m->d_ctr.p, 0, 64, s));
  hipLaunchKernelGGL(k_dm_apply_remove, dim3(nblk(m->cap / 8)), dim3(kB), 0, s, rm, m->K(), m->cap, reinterpret_cast<uint32_t*>(m->d_ctr.p));
  CK(hipGetLastError());
  rc = dm_read_counters(m, ctr);
  if (rc != O3S_OK) return rc;
  int64_t removed = 0;
  for (int k = 0; k < 16; ++k) removed += ctr[k];
  m->live -= removed;
  m->tomb += removed;
  if (n_removed) *n_removed = removed;
  return O3S_OK;
}

int dm_set_dev(const o3s_dense_map* m) { return hipSetDevice(m->device) == hipSuccess ? O3S_OK : O3S_ERR_HIP; }

}  // namespace

extern "C" {

int o3s_dense_map_create(int device, double voxel_size, o3s_dense_map** out) {
  if (!out || !(voxel_size > 0.0) || !std::isfinite(voxel_size)) return O3S_ERR_BAD_ARGUMENT;
  *out = nullptr;
  const int rc = pick_device(device);
  if (rc != O3S_OK) return rc;
  o3s_dense_map* m = new o3s_dense_map();
  m->device = device;
  m->voxel = voxel_size;
  m->inv = 1.0 / voxel_size;
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
    delete m;
    return O3S_ERR_HIP;
  }
  *out = m;
  return O3S_OK;
}

void o3s_dense_map_destroy(o3s_dense_map* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->stream) {
    (void)hipStreamSynchronize(m->stream);
    (void)hipStreamDestroy(m->stream);
  }
  dm_free_table(m);
  delete m;
}

int64_t o3s_dense_map_size(const o3s_dense_map* m) { return m ? m->live : 0; }
int o3s_dense_map_has_normals(const o3s_dense_map* m) { return m ? m->has_normals : 0; }

void o3s_dense_map_clear(o3s_dense_map* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipStreamSynchronize(m->stream);
  dm_free_table(m);
}

int o3s_dense_map_insert(o3s_dense_map* m, const double* pts, const double* normals, int64_t N) {
  if (!m || N < 0 || (N > 0 && !pts)) return O3S_ERR_BAD_ARGUMENT;
  if (N == 0) return O3S_OK;
  int rc = dm_set_dev(m);
  if (rc != O3S_OK) return rc;
  rc = dm_upload(m, pts, normals, N);
  if (rc != O3S_OK) return rc;
  return dm_insert_dev(m, m->in_p.d(), normals ? m->in_n.d() : nullptr, N);
}

int o3s_dense_map_carve(o3s_dense_map* m, const o3s_dense_carving_params* p, const double* scan_pts, int64_t N, const double sensor_position[3],
                        int64_t* n_removed) {
  if (n_removed) *n_removed = 0;
  if (!m || !p || !sensor_position || N < 0 || (N > 0 && !scan_pts)) return O3S_ERR_BAD_ARGUMENT;
  if (N == 0 || m->live == 0) return O3S_OK;
  int rc = dm_set_dev(m);
  if (rc != O3S_OK) return rc;
  rc = dm_upload(m, scan_pts, nullptr, N);
  if (rc != O3S_OK) return rc;
  return dm_carve_dev(m, p, m->in_p.d(), N, sensor_position, n_removed);
}

}  // extern "C"

namespace {
// Submap::insertScanDenseMap (Submap.cpp:97-113) on a raw scan that is already in HBM
int dm_insert_scan_dev(o3s_dense_map* m, const o3s_cropper* dense_map_cropper, const double* d_raw_p, const double* d_raw_n, int64_t N,
                       const double T[16], const o3s_dense_carving_params* carving, int64_t* n_removed) {
  hipStream_t s = m->stream;
  const bool hn = d_raw_n != nullptr;
  int rc = O3S_OK;
  if (N > 0) {
    // denseMapCropper_->setPose(Identity); crop(rawScan) (Submap.cpp:99-100); no colours: colorCropper_ passes all
    o3s_cropper c = *dense_map_cropper;
    c.centre[0] = c.centre[1] = c.centre[2] = 0.0;
    CK(m->crop_p.ensure((size_t)N * 24, 0, s));
    CK(m->crop_n.ensure((size_t)N * 24, 0, s));
    int64_t kept = 0;
    rc = crop_dev(m->arena, c, d_raw_p, d_raw_n, N, m->crop_p.d(), m->crop_n.d(), &kept, s);
    if (rc != O3S_OK) return rc;
    if (kept > 0) {
      // o3d_slam::transform (helpers.cpp:283-318): a near-identity pose emits the cloud twice (copy + transformed)
      double dev = 0.0;
      for (int cc = 0; cc < 4; ++cc)
        for (int r = 0; r < 4; ++r) dev = std::max(dev, std::fabs(T[cc * 4 + r] - (r == cc ? 1.0 : 0.0)));
      const bool doubled = dev < 1e-4;
      const int64_t n_tf = doubled ? 2 * kept : kept;
      CK(m->tf_p.ensure((size_t)n_tf * 24, 0, s));
      CK(m->tf_n.ensure((size_t)n_tf * 24, 0, s));
      CK(m->d_T.ensure(128, 0, s));
      CK(hipMemcpyAsync(m->d_T.p, T, 128, hipMemcpyHostToDevice, s));
      double* dp = m->tf_p.d();
      double* dn = m->tf_n.d();
      if (doubled) {
        CK(hipMemcpyAsync(dp, m->crop_p.p, (size_t)kept * 24, hipMemcpyDeviceToDevice, s));
        if (hn) CK(hipMemcpyAsync(dn, m->crop_n.p, (size_t)kept * 24, hipMemcpyDeviceToDevice, s));
        dp += 3 * kept;
        dn += 3 * kept;
      }
      hipLaunchKernelGGL(k_transform_append, dim3(nblk(kept)), dim3(kB), 0, s, m->crop_p.d(), hn ? m->crop_n.d() : nullptr, kept, m->d_T.d(), dp,
                         hn ? dn : nullptr);
      CK(hipGetLastError());
      rc = dm_insert_dev(m, m->tf_p.d(), hn ? m->tf_n.d() : nullptr, n_tf);
      if (rc != O3S_OK) return rc;
    }
  }
  // carve(rawScan, mapToRangeSensor.translation(), carving_, &denseMap_) (Submap.cpp:108-110, 146-157)
  if (carving && N > 0 && m->live > 0 && (m->n_scans_inserted % carving->carve_space_every_n_scans) == 1) {
    const double sensor[3] = {T[12], T[13], T[14]};
    rc = dm_carve_dev(m, carving, d_raw_p, N, sensor, n_removed);
    if (rc != O3S_OK) return rc;
  }
  ++m->n_scans_inserted;
  return O3S_OK;
}
}  // namespace

extern "C" {

int o3s_dense_map_insert_scan(o3s_dense_map* m, const o3s_cropper* dense_map_cropper, const double* raw_pts, const double* raw_normals, int64_t N,
                              const double T[16], const o3s_dense_carving_params* carving, int64_t* n_removed) {
  if (n_removed) *n_removed = 0;
  if (!m || !dense_map_cropper || !T || N < 0 || (N > 0 && !raw_pts)) return O3S_ERR_BAD_ARGUMENT;
  if (carving && carving->carve_space_every_n_scans <= 0) return O3S_ERR_BAD_ARGUMENT;  // the reference divides by it
  int rc = dm_set_dev(m);
  if (rc != O3S_OK) return rc;
  if (N > 0) {
    rc = dm_upload(m, raw_pts, raw_normals, N);
    if (rc != O3S_OK) return rc;
  }
  return dm_insert_scan_dev(m, dense_map_cropper, m->in_p.d(), raw_normals ? m->in_n.d() : nullptr, N, T, carving, n_removed);
}

int o3s_dense_map_insert_resident_scan(o3s_dense_map* m, const o3s_cropper* dense_map_cropper, const o3s_scan* sc, const double T[16],
                                       const o3s_dense_carving_params* carving, int64_t* n_removed) {
  if (n_removed) *n_removed = 0;
  if (!m || !dense_map_cropper || !sc || !T) return O3S_ERR_BAD_ARGUMENT;
  if (carving && carving->carve_space_every_n_scans <= 0) return O3S_ERR_BAD_ARGUMENT;
  if (sc->device != m->device) return O3S_ERR_BAD_ARGUMENT;
  const int rc = dm_set_dev(m);
  if (rc != O3S_OK) return rc;
  CK(hipStreamSynchronize(sc->stream));  // the scan's own work is complete (its calls end synchronised); this map's stream takes over
  return dm_insert_scan_dev(m, dense_map_cropper, sc->raw_p.d(), sc->raw_has_normals ? sc->raw_n.d() : nullptr, sc->n_raw, T, carving, n_removed);
}

int o3s_dense_map_to_point_cloud(const o3s_dense_map* m, double* pts, double* normals, int32_t* keys, int32_t* counts, int64_t* n_out) {
  if (n_out) *n_out = 0;
  if (!m) return O3S_ERR_BAD_ARGUMENT;
  if (m->live == 0) return O3S_OK;
  if (!pts) return O3S_ERR_BAD_ARGUMENT;
  int rc = dm_set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  const size_t cap = (size_t)m->cap, v = (size_t)m->live;
  const size_t tb = std::max(scan_temp_bytes(m->cap), sort_temp_bytes(m->live));
  CK(m->arena.reserve(Arena::pad(cap * 4) + Arena::pad((cap + 1) * 4) + 2 * Arena::pad(v * 8) + 2 * Arena::pad(v * 4) + Arena::pad(tb) +
                      2 * Arena::pad(v * 24) + Arena::pad(v * 12) + Arena::pad(v * 4) + 4096));
  uint32_t* flag = m->arena.take<uint32_t>(cap);
  uint32_t* off = m->arena.take<uint32_t>(cap + 1);
  uint64_t* k1 = m->arena.take<uint64_t>(v);
  uint64_t* k2 = m->arena.take<uint64_t>(v);
  uint32_t* s1 = m->arena.take<uint32_t>(v);
  uint32_t* s2 = m->arena.take<uint32_t>(v);
  void* tmp = m->arena.take<char>(tb);
  double* op = m->arena.take<double>(v * 3);
  double* on = m->arena.take<double>(v * 3);
  int32_t* ok = m->arena.take<int32_t>(v * 3);
  int32_t* oc = m->arena.take<int32_t>(v);
  hipLaunchKernelGGL(k_dm_live, dim3(nblk(m->cap)), dim3(kB), 0, s, m->K(), m->C(), m->cap, flag);
  int64_t V = 0;
  rc = scan_flags(flag, off, m->cap, tmp, tb, &V, s);
  if (rc != O3S_OK) return rc;
  if (V > m->live) return O3S_ERR_HIP;  // cannot happen: the host count tracks every claim and removal
  if (V == 0) return O3S_OK;
  hipLaunchKernelGGL(k_dm_collect, dim3(nblk(m->cap)), dim3(kB), 0, s, m->K(), flag, off, m->cap, k1, s1);
  size_t stb = tb;
  CK(rocprim::radix_sort_pairs(tmp, stb, k1, k2, s1, s2, (size_t)V, 0, 64, s));
  hipLaunchKernelGGL(k_dm_emit, dim3(nblk(V)), dim3(kB), 0, s, k2, s2, V, m->C(), m->S(), op, on, ok, oc);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(pts, op, (size_t)V * 24, hipMemcpyDeviceToHost, s));
  if (normals) CK(hipMemcpyAsync(normals, on, (size_t)V * 24, hipMemcpyDeviceToHost, s));
  if (keys) CK(hipMemcpyAsync(keys, ok, (size_t)V * 12, hipMemcpyDeviceToHost, s));
  if (counts) CK(hipMemcpyAsync(counts, oc, (size_t)V * 4, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  if (n_out) *n_out = V;
  return O3S_OK;
}

int o3s_dense_map_transform(o3s_dense_map* m, const double T[16]) {
  if (!m || !T) return O3S_ERR_BAD_ARGUMENT;
  if (m->live == 0) return O3S_OK;  // "if (empty()) return" (Voxel.cpp:51-53)
  const int rc = dm_set_dev(m);
  if (rc != O3S_OK) return rc;
  hipStream_t s = m->stream;
  CK(m->d_T.ensure(128, 0, s));
  CK(hipMemcpyAsync(m->d_T.p, T, 128, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_dm_transform, dim3(nblk(m->cap)), dim3(kB), 0, s, m->K(), m->C(), m->S(), m->cap, m->d_T.d());
  CK(hipGetLastError());
  CK(hipStreamSynchronize(s));
  return O3S_OK;
}

}  // extern "C"
