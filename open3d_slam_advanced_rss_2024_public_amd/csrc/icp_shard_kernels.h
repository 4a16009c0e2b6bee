// icp_shard_kernels.h — the extra kernels of the ONE-PAIR-SHARDED mode (SURVEY.md §8(e) mode 2): the reading is split
// over the ranks, the reference index is replicated, and the three global quantities of an iteration — the trim limit
// (LPM/Matches.cpp:61-87), the kept-pair means (LPM/ErrorMinimizers/PointToPlane.cpp:263-264) and the 21 + 6 sums of the
// normal equations (:283-306) — are formed by all-reducing fixed-size buffers between the local kernels:
//
//   k_match2 (local) -> k_shard_fold_hist  -> [AR int32 x 2048: level-1 histogram]          -> copied into replica 0
//   k_classify (local, unchanged: it now sees the GLOBAL histogram, so bin / rank / n_finite are global; it also counts
//               the level-2 digits of this rank's candidates)
//   k_shard_l2_out -> [AR int32 x 1024]   k_shard_l3_hist -> [AR int32 x 1024]                (exact radix selection)
//   k_shard_sel_apply   -> [AR f64 x 8: sum p, sum q, |K|]  -> k_shard_publish (limit, means, |K| into the header)
//   k_normal_eq (local, unchanged) -> k_shard_fold_ne -> [AR f64 x 27] -> k_solve (replicated, nb = 1)
//
// Every rank ends an iteration with bit-identical state (the all-reduce hands every rank the same sums and the rest is
// deterministic integer / fp32 / fp64 arithmetic), so the `done` decision is identical and no broadcast is needed.
// Integer sums are exact; the fp64 sums are rounded to fp32 once, exactly as in the unsharded chain.
#pragma once
#include "icp_kernels.h"

namespace o3s {

// exchange buffer layout (bytes): f64[40] | int32 level-1[2048] | int32 level-2[1024] | int32 level-3[1024]
constexpr int kXchgF64 = 40;                 // [0..7] centroid sums, [8..34] normal-equation sums
constexpr int kXchgCentOff = 0;
constexpr int kXchgNeOff = 8;
constexpr int kXchgI32Off = kXchgF64 * 8;     // byte offset of the int32 region
constexpr int kXchgL1 = 0, kXchgL2 = kHistBins, kXchgL3 = kHistBins + 1024;  // int32 word offsets inside that region
constexpr int kXchgBytes = kXchgI32Off + (kHistBins + 2048) * 4;            // 16704

namespace kern {

// level-1 replicas -> one histogram in the exchange buffer; the replicas are cleared (replica 0 is refilled with the
// reduced histogram by a device copy, so k_classify's 8-replica sum yields the global counts)
__global__ void __launch_bounds__(kBlock) k_shard_fold_hist(uint32_t* __restrict__ hist_rep, uint32_t* __restrict__ out) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= kHistBins) return;
  uint32_t s = 0;
#pragma unroll
  for (int r = 0; r < kHistReplicas; ++r) {
    s += hist_rep[(size_t)r * kHistBins + b];
    hist_rep[(size_t)r * kHistBins + b] = 0u;
  }
  out[b] = s;
}

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

// level 2 -> exchange buffer: k_classify counted this rank's level-2 digits (bits 19..10 of its candidates) already
__global__ void __launch_bounds__(kSelThreads) k_shard_l2_out(const IcpState* __restrict__ st, const SelScratch* __restrict__ ss,
                                                              uint32_t* __restrict__ hist2, uint32_t* __restrict__ xi) {
  const bool idle = st->done != 0 || ss->skip != 0;  // uniform; the exchange still runs on every rank, on zeros
  xi[kXchgL2 + threadIdx.x] = idle ? 0u : hist2[threadIdx.x];
  hist2[threadIdx.x] = 0u;  // ready for the next iteration
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

// local histogram of the last 10 bits over this rank's candidates that carry the 21-bit prefix (bin, d1)
__global__ void __launch_bounds__(kSelThreads) k_shard_l3_hist(const IcpState* __restrict__ st, const SelScratch* __restrict__ ss,
                                                               const CandRec* __restrict__ cand, const uint32_t* __restrict__ cand_cnt, int nb,
                                                               uint32_t* __restrict__ base_scratch /*[nb + 1]*/, uint32_t* __restrict__ xi) {
  __shared__ uint32_t s_bins[1024];
  __shared__ uint32_t s_tmp[64];
  s_bins[threadIdx.x] = 0u;
  __syncthreads();
  const bool idle = st->done != 0 || ss->skip != 0;
  if (idle) {  // uniform
    xi[kXchgL3 + threadIdx.x] = 0u;
    return;
  }
  uint32_t kk = ss->kk, d1;
  shard_pick_digit(xi + kXchgL2, s_tmp, kk, d1);
  const uint32_t prefix21 = (ss->bin << 10) | d1;
  const uint32_t total = shard_bases(cand_cnt, nb, base_scratch, s_tmp);
  for (uint32_t f = threadIdx.x; f < total; f += kSelThreads) {
    const int b = flat_block(base_scratch, nb, f);
    const uint32_t u = cand[(size_t)b * kClsBlock + (f - base_scratch[b])].bits;
    if ((u >> 10) == prefix21) atomicAdd(&s_bins[u & 1023u], 1u);
  }
  __syncthreads();
  xi[kXchgL3 + threadIdx.x] = s_bins[threadIdx.x];
}

// limit from the two reduced histograms; this rank's share of the kept-pair sums (classify partials + its candidates
// with d2 <= limit, in the run-independent flat order) -> exchange buffer
__global__ void __launch_bounds__(kSelThreads) k_shard_sel_apply(uint32_t* __restrict__ hist_rep, ChainParams cp, IcpState* __restrict__ st,
                                                                 const SelScratch* __restrict__ ss, const CandRec* __restrict__ cand,
                                                                 const uint32_t* __restrict__ cand_cnt, const uint32_t* __restrict__ base_scratch,
                                                                 const double* __restrict__ part /*[7][nb]*/, int nb,
                                                                 const uint32_t* __restrict__ xi, double* __restrict__ xd) {
  __shared__ uint32_t s_tmp[64];
  using Sum = BlockSum<kCentComps, kSelThreads>;
  __shared__ double s_a[Sum::kWordsA];
  __shared__ double s_b[Sum::kWordsB];
  const float hv = hdr_load(st);
  double a[kCentComps] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nb; b += kSelThreads) {
#pragma unroll
    for (int k = 0; k < kCentComps; ++k) a[k] += part[k * nb + b];
  }
  if (hdr_i(hv, H_DONE)) {  // uniform
    if (threadIdx.x < 8) xd[kXchgCentOff + threadIdx.x] = 0.0;
    return;
  }
  for (int k = threadIdx.x; k < kHistReplicas * kHistBins; k += kSelThreads) hist_rep[k] = 0u;
  const uint32_t bin = ss->bin, skip = ss->skip;
  uint32_t kk = ss->kk;
  float limit = kInfF;
  const bool failed = hdr_i(hv, H_STATUS) != 0;  // e.g. no finite match: the partials were never written
  if (!skip) {  // uniform
    uint32_t d1, d0;
    shard_pick_digit(xi + kXchgL2, s_tmp, kk, d1);
    shard_pick_digit(xi + kXchgL3, s_tmp, kk, d0);
    const uint32_t lbits = (bin << 20) | (d1 << 10) | d0;
    limit = __uint_as_float(lbits);
    const uint32_t total = base_scratch[nb];  // written by k_shard_l3_hist of this iteration
    for (uint32_t f = threadIdx.x; f < total; f += kSelThreads) {
      const int b = flat_block(base_scratch, nb, f);
      const CandRec r = cand[(size_t)b * kClsBlock + (f - base_scratch[b])];
      if (r.keep && r.bits <= lbits) {
        a[0] += (double)r.px;
        a[1] += (double)r.py;
        a[2] += (double)r.pz;
        a[3] += (double)r.qx;
        a[4] += (double)r.qy;
        a[5] += (double)r.qz;
        a[6] += 1.0;
      }
    }
  }
  Sum::run(a, s_a, s_b);
  if (threadIdx.x < 8) xd[kXchgCentOff + threadIdx.x] = (threadIdx.x < kCentComps && !failed) ? Sum::total(s_b, threadIdx.x) : 0.0;
  if (threadIdx.x == 0 && (!cp.has_trim || !skip)) st->limit = limit;
}

// reduced sums -> |K| and the two means in the state header (the tail of k_sel_finish; ErrorMinimizer.cpp:75-77)
__global__ void k_shard_publish(IcpState* __restrict__ st, const double* __restrict__ xd) {
  if (threadIdx.x != 0 || st->done) return;
  if (st->status != 0) {
    st->done = 1;
    return;
  }
  const double* t = xd + kXchgCentOff;
  const double K = t[6];
  st->kept = (int32_t)K;
  if (K == 0.0) {
    st->status = 6;
    st->done = 1;
  } else {
    st->mp[0] = (float)(t[0] / K);
    st->mp[1] = (float)(t[1] / K);
    st->mp[2] = (float)(t[2] / K);
    st->mq[0] = (float)(t[3] / K);
    st->mq[1] = (float)(t[4] / K);
    st->mq[2] = (float)(t[5] / K);
  }
}

// this rank's 27 normal-equation sums: [27][nb] block partials -> exchange buffer
__global__ void __launch_bounds__(kBlock) k_shard_fold_ne(const double* __restrict__ part, int nb, const IcpState* __restrict__ st,
                                                          double* __restrict__ xd) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const bool idle = st->done != 0;  // k_normal_eq did not run: contribute zeros
  for (int c = w; c < kNeComps; c += kBlock / 64) {
    double s = 0;
    if (!idle)
      for (int b = l; b < nb; b += 64) s += part[c * nb + b];
    s = wave_sum(s);
    if (l == 0) xd[kXchgNeOff + c] = s;
  }
}

}  // namespace kern
}  // namespace o3s
