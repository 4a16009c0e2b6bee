// icp_shard_kernels.h — the extra kernels of the ONE-PAIR-SHARDED mode (SURVEY.md §8(e) mode 2): the reading is split
// over the ranks, the reference index is replicated, and the three global quantities of an iteration — the trim limit
// (LPM/Matches.cpp:61-87), the kept-pair means (LPM/ErrorMinimizers/PointToPlane.cpp:263-264) and the 21 + 6 sums of the
// normal equations (:283-306) — are formed by FOUR in-place sum all-reduces of regions of one exchange buffer, between five
// kernels (the unsharded chain has four):
//
//   k_match2 (local; its R = max(1, 16 / world) level-1 replicas ARE the exchange buffer's L1 region)  -> [AR int32 x R x 2048]
//   k_classify (local, unchanged: it sums the replicas itself, and every replica now holds the sum over the ranks, so bin /
//               rank / n_finite are global; it counts this rank's level-2 digits straight into the L2 region)
//                                                                               -> [AR int32 x 1024]
//   k_shard_l3_sums   level-2 digit from the reduced histogram; this rank's candidates: the decided ones join the rank's base
//                     sums, the ones that carry the 21-bit prefix are counted AND summed per level-3 bin (in flat order)
//                                                                               -> [AR f64 x (8 + 1024 + 7 x 1024)]
//   k_shard_sel_ne    every block: level-3 digit from the reduced counts -> the exact limit; kept-pair sums = base + the
//                     bins up to that digit, in a fixed order -> means; block 0 publishes; then this block's share of the
//                     27 normal-equation sums, written as block partials into the NE region
//                                                                               -> [AR f64 x 27 x blocks]
//   k_solve (replicated; reduces the summed block partials exactly as in the unsharded chain)
//
// Round 1's schedule had five collectives and eleven kernels (fold / copy / publish kernels between them); what changed: the
// replicas and the block partials are reduced as they are (no fold kernels), the kept-pair sums ride on the level-3
// exchange as per-bin sums (no fourth exchange for them), and the kernels that only published reduced values are gone —
// every block that needs them forms them itself from the reduced buffer.
// Every rank ends an iteration with bit-identical state (the all-reduce hands every rank the same sums and the rest is
// deterministic integer / fp32 / fp64 arithmetic), so the `done` decision is identical and no broadcast is needed.
// Integer sums are exact; the fp64 sums are rounded to fp32 once, exactly as in the unsharded chain.
#pragma once
#include "icp_kernels.h"

namespace o3s {

// exchange buffer layout (bytes).  Region A (doubles): [0..7] this rank's base sums (7 used) | [8 .. 8+1024) level-3 counts
// (as doubles: exact) | per-bin kept sums [7][1024].  Region NE (doubles): [27][blocks] block partials of the normal
// equations (blocks = what this rank's slice needs: N / world / 512).  Region I (int32): the level-1 replicas
// [R][2048], R = max(1, 16 / world) — the matcher spreads its histogram flushes over R replicas to bound same-address atomics, and
// a rank's share of the blocks shrinks with the world size, so the replicas that travel shrink with it (16 KB at eight ranks) —
// and the level-2 histogram [1024] right behind the 16-replica area.
constexpr int kXaBase = 0, kXaCnt = 8, kXaSum = 8 + 1024;
constexpr int kXaDoubles = 8 + 1024 + kCentComps * 1024;                         // 8200
constexpr int kXchgAOff = 0;
constexpr int kXchgNeOff = kXaDoubles * 8;                                       // byte offset of region NE
constexpr int kXchgNeDoubles = kNeComps * kMaxPartialBlocks;
constexpr int kXchgI32Off = kXchgNeOff + kXchgNeDoubles * 8;                      // byte offset of region I
constexpr int kXchgL1Words = kHistReplicas * kHistBins;                           // room for 16 replicas; R of them are used and travel
constexpr int kXchgBytes = kXchgI32Off + (kXchgL1Words + 1024) * 4;
inline int shard_replicas(int world) { return world >= kHistReplicas ? 1 : (world <= 1 ? kHistReplicas : kHistReplicas / (1 << (31 - __builtin_clz((unsigned)world)))); }
// bytes that cross the links per iteration and rank: R x 2048 x 4 + 1024 x 4 + 8200 x 8 + 27 x blocks x 8 — at eight ranks and C2
// (12.5 k points per rank: 25 blocks) 16 + 4 + 65.6 + 5.4 = 91 KB (round 3: 312 KB whatever the world size).  What is left is exchange
// 3: the level-3 counts AND the kept sums of every level-3 bin travel together because the kept sums depend on the limit's last ten
// bits, which only the summed counts give — sending them apart would be one more (latency-bound) collective.
inline int64_t shard_bytes_per_iteration(int world, int nb_part) {
  return (int64_t)shard_replicas(world) * kHistBins * 4 + 1024 * 4 + (int64_t)kXaDoubles * 8 + (int64_t)kNeComps * nb_part * 8;
}

namespace kern {

// digit of a 1024-bin histogram that holds rank kk (block-wide, kSelThreads == 1024 lanes); kk becomes the rank inside it
__device__ __forceinline__ void shard_pick_digit(const uint32_t* __restrict__ hist, uint32_t* s_tmp, uint32_t& kk, uint32_t& digit) {
  const uint32_t c = hist[threadIdx.x];
  uint32_t tot;
  const uint32_t ex = block_excl_scan(c, &tot, s_tmp);
  __syncthreads();
  if (threadIdx.x == 0) {
    s_tmp[40] = 0u;
    s_tmp[41] = 0u;
  }
  __syncthreads();
  if (c > 0 && ex <= kk && kk < ex + c) {
    s_tmp[40] = threadIdx.x;
    s_tmp[41] = kk - ex;
  }
  __syncthreads();
  digit = s_tmp[40];
  kk = s_tmp[41];
  __syncthreads();
}

// the same with 256 lanes owning four consecutive bins each (the blocks of k_shard_sel_ne); c4 = this lane's four counts
__device__ __forceinline__ void shard_pick_digit4(const uint32_t c4[4], uint32_t* s_tmp, uint32_t& kk, uint32_t& digit) {
  const uint32_t c = (c4[0] + c4[1]) + (c4[2] + c4[3]);
  uint32_t tot;
  const uint32_t ex = block_excl_scan(c, &tot, s_tmp);
  __syncthreads();
  if (threadIdx.x == 0) {
    s_tmp[40] = 0u;
    s_tmp[41] = 0u;
  }
  __syncthreads();
  if (c > 0 && ex <= kk && kk < ex + c) {
    uint32_t acc = ex;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (c4[q] > 0 && acc <= kk && kk < acc + c4[q]) {
        s_tmp[40] = threadIdx.x * 4 + q;
        s_tmp[41] = kk - acc;
      }
      acc += c4[q];
    }
  }
  __syncthreads();
  digit = s_tmp[40];
  kk = s_tmp[41];
  __syncthreads();
}

// exclusive bases of the classify blocks' candidate runs (block-wide, kSelThreads lanes); returns the total
__device__ __forceinline__ uint32_t shard_bases(const uint32_t* __restrict__ cand_cnt, int nb, uint32_t* base /*[nb + 1]*/, uint32_t* s_tmp) {
  const int per_thread = (nb + kSelThreads - 1) / kSelThreads;
  const int b0 = min((int)threadIdx.x * per_thread, nb), b1 = min(b0 + per_thread, nb);
  uint32_t mine = 0;
  for (int b = b0; b < b1; ++b) mine += cand_cnt[b];
  uint32_t total;
  uint32_t run = block_excl_scan(mine, &total, s_tmp);
  for (int b = b0; b < b1; ++b) {
    base[b] = run;
    run += cand_cnt[b];
  }
  if (threadIdx.x == 0) base[nb] = total;
  __threadfence_block();
  __syncthreads();
  return total;
}

constexpr int kShardPark = 1024;  // candidates with the 21-bit prefix kept in LDS (more: the bins are summed from memory)

// This rank's share of the selection, ready to be summed over the ranks (region A of the exchange buffer):
//   base[7]        fp64 sums of the pairs that are kept whatever the last ten bits of the limit turn out to be: the classify
//                  partials + this rank's candidates below the prefix (bin, d1)
//   cnt[1024]      this rank's candidates with that prefix, per value of their last ten bits (every finite one: the rank
//                  statistic counts them all)
//   sum[7][1024]   the kept ones among them, summed per bin in flat candidate order (run-independent)
__global__ void __launch_bounds__(kSelThreads) k_shard_l3_sums(ChainParams cp, const IcpState* __restrict__ st, const SelScratch* __restrict__ ss,
                                                               const CandRec* __restrict__ cand, const uint32_t* __restrict__ cand_cnt, int nb,
                                                               uint32_t* __restrict__ base_scratch /*[nb + 1]*/, const double* __restrict__ part /*[7][nb]*/,
                                                               const uint32_t* __restrict__ l2 /*reduced level-2 histogram*/, double* __restrict__ xa) {
  extern __shared__ __align__(16) uint32_t s_dyn[];  // parked records | flat indices | order; later the block sum
  __shared__ uint32_t s_tmp[64];
  __shared__ uint32_t s_bins[1024];
  __shared__ uint32_t s_base[kBaseCap + 1];
  using Sum = BlockSum<kCentComps, kSelThreads>;
  CandRec* s_rec = reinterpret_cast<CandRec*>(s_dyn);
  uint32_t* s_flat = s_dyn + kShardPark * 8;
  uint32_t* s_ord = s_flat + kShardPark;
  const float hv = hdr_load(st);
  double a[kCentComps] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nb; b += kSelThreads) {
#pragma unroll
    for (int k = 0; k < kCentComps; ++k) a[k] += part[k * nb + b];
  }
  const bool idle = hdr_i(hv, H_DONE) != 0, failed = hdr_i(hv, H_STATUS) != 0;  // uniform; an idle rank still takes part, with zeros
  const uint32_t skip = ss->skip;
  double cnt_out = 0.0, sum_out[kCentComps] = {0, 0, 0, 0, 0, 0, 0};
  if (!idle && !failed && !skip) {  // uniform
    uint32_t kk = ss->kk, d1;
    shard_pick_digit(l2, s_tmp, kk, d1);
    const uint32_t prefix21 = (ss->bin << 10) | d1;
    s_bins[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_tmp[43] = 0u;
    __syncthreads();
    // flat sweep with the bases of the classify blocks' candidate runs in LDS: a record's block is found by a binary search in
    // LDS (the bases used to live in global memory: ten DEPENDENT global loads per record), and a thread's records are
    // independent loads.  (A region-wise sweep — 32 lanes per classify block, no bases at all — was tried: the 31 regions a
    // group walks one after the other cost two dependent round trips each, +28 us per iteration at C4 either way.)
    const uint32_t* base = nb <= kBaseCap ? s_base : base_scratch;
    const uint32_t total = shard_bases(cand_cnt, nb, nb <= kBaseCap ? s_base : base_scratch, s_tmp);
    for (uint32_t f = threadIdx.x; f < total; f += kSelThreads) {
      const int b = flat_block(base, nb, f);
      const CandRec r = cand[(size_t)b * kClsBlock + (f - base[b])];
      const uint32_t p21 = r.bits >> 10;
      if (p21 == prefix21) {
        atomicAdd(&s_bins[r.bits & 1023u], 1u);
        const uint32_t slot = atomicAdd(&s_tmp[43], 1u);
        if (slot < (uint32_t)kShardPark) {
          s_rec[slot] = r;
          s_flat[slot] = f;
        }
      } else if (p21 < prefix21 && r.keep) {
        a[0] += (double)r.px;
        a[1] += (double)r.py;
        a[2] += (double)r.pz;
        a[3] += (double)r.qx;
        a[4] += (double)r.qy;
        a[5] += (double)r.qz;
        a[6] += 1.0;
      }
    }
    __syncthreads();
    const uint32_t m = s_tmp[43];
    cnt_out = (double)s_bins[threadIdx.x];
    if (m <= (uint32_t)kShardPark) {
      // parked in arrival order; summed in FLAT order: rank of every parked record among the m flat indices, then lane t
      // walks the ordered list and adds the kept records of bin t
      for (uint32_t j = threadIdx.x; j < m; j += kSelThreads) {
        const uint32_t fj = s_flat[j];
        uint32_t rank = 0;
        for (uint32_t i = 0; i < m; ++i) rank += s_flat[i] < fj ? 1u : 0u;
        s_ord[rank] = j;
      }
      __syncthreads();
      for (uint32_t r = 0; r < m; ++r) {
        const CandRec& rc = s_rec[s_ord[r]];
        if ((rc.bits & 1023u) == threadIdx.x && rc.keep) {
          sum_out[0] += (double)rc.px;
          sum_out[1] += (double)rc.py;
          sum_out[2] += (double)rc.pz;
          sum_out[3] += (double)rc.qx;
          sum_out[4] += (double)rc.qy;
          sum_out[5] += (double)rc.qz;
          sum_out[6] += 1.0;
        }
      }
    } else {  // heavy ties: every lane walks the flat list for its own bin
      for (uint32_t f = 0; f < total; ++f) {
        const int b = flat_block(base, nb, f);
        const CandRec rc = cand[(size_t)b * kClsBlock + (f - base[b])];
        if ((rc.bits >> 10) == prefix21 && (rc.bits & 1023u) == threadIdx.x && rc.keep) {
          sum_out[0] += (double)rc.px;
          sum_out[1] += (double)rc.py;
          sum_out[2] += (double)rc.pz;
          sum_out[3] += (double)rc.qx;
          sum_out[4] += (double)rc.qy;
          sum_out[5] += (double)rc.qz;
          sum_out[6] += 1.0;
        }
      }
    }
  }
  xa[kXaCnt + threadIdx.x] = cnt_out;
#pragma unroll
  for (int c = 0; c < kCentComps; ++c) xa[kXaSum + c * 1024 + threadIdx.x] = sum_out[c];
  __syncthreads();  // the parked records are done with: the block sum borrows their memory
  double* s_a = reinterpret_cast<double*>(s_dyn);
  double* s_b = s_a + Sum::kWordsA;
  Sum::run(a, s_a, s_b);
  if (threadIdx.x < 8) xa[kXaBase + threadIdx.x] = (threadIdx.x < kCentComps && !idle && !failed) ? Sum::total(s_b, threadIdx.x) : 0.0;
}
constexpr size_t kShardL3DynBytes =
    (size_t)((BlockSum<kCentComps, kSelThreads>::kWordsA + BlockSum<kCentComps, kSelThreads>::kWordsB) * 8) > (size_t)(kShardPark * 40)
        ? (size_t)((BlockSum<kCentComps, kSelThreads>::kWordsA + BlockSum<kCentComps, kSelThreads>::kWordsB) * 8)
        : (size_t)(kShardPark * 40);

// Every block: the exact limit and the kept-pair means from the REDUCED region A (the same integers and the same fixed-order
// fp64 sums in every block and on every rank), block 0 publishes them; then this block's share of the normal equations
// (k_normal_eq's loop) with block partials into the NE region, and the level-1 replicas cleared for the next k_match2.
__global__ void __launch_bounds__(kBlock) k_shard_sel_ne(ChainParams cp, IcpState* __restrict__ st, const SelScratch* __restrict__ ss,
                                                         const uint32_t* __restrict__ l2, const double* __restrict__ xa,
                                                         const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz, int N,
                                                         const float4* __restrict__ mq, const float4* __restrict__ mn, const int32_t* __restrict__ pos,
                                                         const float* __restrict__ d2, double* __restrict__ xne /*[27][grid]*/,
                                                         uint32_t* __restrict__ hist_zero /*level-1 replicas*/, int n_rep) {
  __shared__ uint32_t s_tmp[64];
  __shared__ float s_out[8];
  using SumC = BlockSum<kCentComps, kBlock>;
  using SumN = BlockSum<kNeComps, kBlock>;
  __shared__ double s_a[SumN::kWordsA];  // the 27-component sum is the larger one; the 7-component sum borrows it first
  __shared__ double s_b[SumN::kWordsB];
  static_assert(SumC::kWordsA <= SumN::kWordsA && SumC::kWordsB <= SumN::kWordsB, "block sum buffers");
  const float hv = hdr_load(st);
  if (hdr_i(hv, H_DONE)) return;
  const int status = hdr_i(hv, H_STATUS);
  const uint32_t skip = ss->skip;
  float limit = kInfF;
  double a[kCentComps];
#pragma unroll
  for (int c = 0; c < kCentComps; ++c) a[c] = threadIdx.x == 0 ? xa[kXaBase + c] : 0.0;
  if (status == 0 && !skip) {  // uniform
    uint32_t kk = ss->kk, d1, d0;
    const uint4 u2 = *reinterpret_cast<const uint4*>(l2 + 4 * threadIdx.x);
    const uint32_t c2[4] = {u2.x, u2.y, u2.z, u2.w};
    shard_pick_digit4(c2, s_tmp, kk, d1);
    uint32_t c3[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c3[q] = (uint32_t)xa[kXaCnt + 4 * threadIdx.x + q];  // counts summed as doubles: exact
    shard_pick_digit4(c3, s_tmp, kk, d0);
    const uint32_t lbits = (ss->bin << 20) | (d1 << 10) | d0;
    limit = __uint_as_float(lbits);
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // bins in ascending order inside the lane, lanes in the block sum's fixed order
      const uint32_t b = 4 * threadIdx.x + q;
      if (b <= d0) {
#pragma unroll
        for (int c = 0; c < kCentComps; ++c) a[c] += xa[kXaSum + c * 1024 + b];
      }
    }
  }
  SumC::run(a, s_a, s_b);
  const double K = SumC::total(s_b, 6);
  if (threadIdx.x < 6) {
    const bool publish = blockIdx.x == 0;
    if (threadIdx.x == 0) {
      if (publish && (!cp.has_trim || !skip)) st->limit = limit;
      s_out[0] = limit;
    }
    if (status != 0) {
      if (publish && threadIdx.x == 0) st->done = 1;
    } else {
      if (publish && threadIdx.x == 0) st->kept = (int32_t)K;
      if (K == 0.0) {  // "no point to minimize" (ErrorMinimizer.cpp:75-77)
        if (publish && threadIdx.x == 0) {
          st->status = 6;
          st->done = 1;
        }
      } else {
        const float mean = (float)(SumC::total(s_b, threadIdx.x) / K);
        s_out[1 + threadIdx.x] = mean;
        if (publish) {
          if (threadIdx.x < 3) st->mp[threadIdx.x] = mean;
          else st->mq[threadIdx.x - 3] = mean;
        }
      }
    }
  }
  if (status != 0 || K == 0.0) return;  // uniform
  __syncthreads();
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < n_rep * kHistBins; k += gridDim.x * kBlock) hist_zero[k] = 0u;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = hdr_f(hv, k);
  limit = s_out[0];
  const float mpx = s_out[1], mpy = s_out[2], mpz = s_out[3], mqx = s_out[4], mqy = s_out[5], mqz = s_out[6];
  double acc[kNeComps];
#pragma unroll
  for (int c = 0; c < kNeComps; ++c) acc[c] = 0.0;
  for (int base = blockIdx.x * (kBlock * kNePPT) + threadIdx.x; base < N; base += gridDim.x * (kBlock * kNePPT)) {
#pragma unroll
    for (int u = 0; u < kNePPT; ++u) {
      const int i = base + u * kBlock;
      if (i >= N) continue;
      if (!kept_pair(pos[i], d2[i], limit, cp.max_out_r2)) continue;
      const float x0 = rx[i], y0 = ry[i], z0 = rz[i];
      const float4 q = mq[i], n = mn[i];
      const float px = xf_row(T, 0, x0, y0, z0) - mpx, py = xf_row(T, 1, x0, y0, z0) - mpy, pz = xf_row(T, 2, x0, y0, z0) - mpz;
      const float qx = q.x - mqx, qy = q.y - mqy, qz = q.z - mqz;
      float gv[6];
      gv[0] = py * n.z - pz * n.y;
      gv[1] = pz * n.x - px * n.z;
      gv[2] = px * n.y - py * n.x;
      gv[3] = n.x;
      gv[4] = n.y;
      gv[5] = n.z;
      const float ex = px - qx, ey = py - qy, ez = pz - qz;
      float h = 0.f;
      h = h + ex * n.x;
      h = h + ey * n.y;
      h = h + ez * n.z;
      int t = 0;
#pragma unroll
      for (int aa = 0; aa < 6; ++aa) {
#pragma unroll
        for (int c = aa; c < 6; ++c) acc[t++] += (double)(gv[aa] * gv[c]);
      }
#pragma unroll
      for (int aa = 0; aa < 6; ++aa) acc[21 + aa] += (double)(gv[aa] * h);
    }
  }
  __syncthreads();  // the 7-component totals have been read
  SumN::run(acc, s_a, s_b);
  if (threadIdx.x < kNeComps) xne[threadIdx.x * gridDim.x + blockIdx.x] = SumN::total(s_b, threadIdx.x);
}

}  // namespace kern
}  // namespace o3s
