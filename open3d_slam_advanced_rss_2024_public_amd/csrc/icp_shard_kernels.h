// icp_shard_kernels.h — the extra kernels of the ONE-PAIR-SHARDED mode (SURVEY.md §8(e) mode 2): the reading is split
// over the ranks, the reference index is replicated, and the three global quantities of an iteration — the trim limit
// (LPM/Matches.cpp:61-87), the kept-pair means (LPM/ErrorMinimizers/PointToPlane.cpp:263-264) and the 21 + 6 sums of the
// normal equations (:283-306) — are formed by THREE in-place sum all-reduces of regions of one exchange buffer, between
// four kernels (round 5; rounds 3-4: four all-reduces between five kernels):
//
//   k_match2 (local; its R level-1 replicas ARE the exchange buffer's L1 region)                -> [AR int32 x R x 2048]
//   k_classify (local, unchanged but for the width of its level-2 digit: it sums the replicas itself, and every replica now
//               holds the sum over the ranks, so bin / rank / n_finite are global; it counts this rank's level-2 digits —
//               THIRTEEN bits here — straight into the L2 region)                               -> [AR int32 x 8192]
//   k_shard_moments   every block: level-2 digit from the reduced histogram -> the 24-bit prefix (bin, d1) of the limit.
//                     Blocks 0 .. nb - 1 stream this rank's pairs: every pair that is kept WHATEVER the limit's last seven bits
//                     are (its 24 leading bits below the prefix, all other weights 1) adds its RAW moments — un-centred, fp64:
//                     sum cc^T, sum cn^T, sum nn^T, sum c r, sum n r, sum p, sum q, 1, with c = p x n, r = n.(p - q) — to the
//                     block's partial.  The last block takes the pairs that carry the prefix (a handful): counted per level-3
//                     bin, the kept ones' raw moments summed per bin, in index order.           -> [AR f64 x (128 + 34 x 128 + 34 x nb)]
//   k_solve_shard (replicated)  level-3 digit from the summed counts -> the exact limit; moments = block partials (block order)
//                     + the bins up to that digit; means = sum p / K, sum q / K rounded to fp32 as the reference's are; the normal
//                     equations follow from the raw moments by CENTRING ALGEBRAICALLY with those means; then the closing step
//                     of the unsharded chain (solve_body: solve, step, T_iter, checkers, post).
//
// Why three suffice now.  The selection is still a chain of dependent sums (bin -> digit -> last bits), but the normal equations
// no longer wait for its end: raw moments need neither the means (algebraic centring) nor the last bits of the limit (the pairs
// those bits decide travel apart, per bin).  What this costs is the one promise the sharded mode never made: the reference
// centres every pair in fp32 BEFORE it multiplies (p - mean is rounded per pair), the raw moments are exact products centred
// once in fp64 — the 6 x 6 system differs from the unsharded chain's in its last bits, as it already did through the order of the
// fp64 sums (pose within 1e-6; the integers — limit, kept count, iterations — stay equal, tests/test_gpu_sharded.py).
// The level-2 digit has thirteen bits in this mode (8 192 bins, 32 KB) so that level 3 has seven: 128 bins x 34 moments is
// 35 KB where 1 024 bins would be 280 KB.  Bytes per iteration at eight ranks and C2: 16 + 32 + 43 = 91 KB, as before.
// Every rank ends an iteration with bit-identical state (the all-reduce hands every rank the same sums and the rest is
// deterministic), so the `done` decision is identical and no broadcast is needed.
#pragma once
#include "icp_kernels.h"

namespace o3s {

constexpr int kMom = 34;  // raw moments of a kept pair: cc^T (6) | cn^T (9) | nn^T (6) | c r (3) | n r (3) | p (3) | q (3) | count
constexpr int kShardL2Bits = 13, kShardL2Bins = 1 << kShardL2Bits;  // level-2 digit of the sharded chain: bits 19..7
constexpr int kShardL3Bits = 20 - kShardL2Bits, kShardL3Bins = 1 << kShardL3Bits;  // 7 bits, 128 bins
// exchange buffer layout.  Region M (doubles): level-3 counts [128] (as doubles: exact) | per-bin raw moments [34][128] | block
// partials of the raw moments [34][blocks] (blocks = what a rank's share of the reading needs: N / world / 512, the same on every
// rank).  Region I (int32): the level-1 replicas [R][2048], R = 16 >> floor(log2(world)) (the matcher spreads its flushes over R
// replicas to bound same-address atomics; a rank's share of the blocks shrinks with the world size, so the replicas that travel
// shrink with it) and, right behind the 16-replica area, the level-2 histogram [8192].
constexpr int kXmCnt = 0, kXmBin = kShardL3Bins, kXmPart = kShardL3Bins + kMom * kShardL3Bins;
constexpr int kXchgMOff = 0;
constexpr int kXchgMDoubles = kXmPart + kMom * kMaxPartialBlocks;
constexpr int kXchgI32Off = kXchgMDoubles * 8;                                    // byte offset of region I
constexpr int kXchgL1Words = kHistReplicas * kHistBins;                           // room for 16 replicas; R of them are used and travel
constexpr int kXchgBytes = kXchgI32Off + (kXchgL1Words + kShardL2Bins) * 4;
inline int shard_replicas(int world) { return world >= kHistReplicas ? 1 : (world <= 1 ? kHistReplicas : kHistReplicas / (1 << (31 - __builtin_clz((unsigned)world)))); }
inline int64_t shard_moment_doubles(int nb_part) { return (int64_t)kXmPart + (int64_t)kMom * nb_part; }
// bytes that cross the links per iteration and rank: R x 2048 x 4 + 8192 x 4 + (128 + 34 x 128 + 34 x blocks) x 8 — at eight ranks
// and C2 (12.5 k points per rank: 25 blocks) 16 + 32 + 42.6 = 91 KB in three collectives (rounds 3-4: 91 KB in four)
inline int64_t shard_bytes_per_iteration(int world, int nb_part) {
  return (int64_t)shard_replicas(world) * kHistBins * 4 + (int64_t)kShardL2Bins * 4 + shard_moment_doubles(nb_part) * 8;
}

namespace kern {

// digit of the reduced level-2 histogram (8 192 bins) that holds rank kk; kk becomes the rank inside it.  Block-wide, kBlock
// (256) lanes owning 32 consecutive bins each; s_tmp: >= 64 words.
__device__ __forceinline__ void shard_pick_digit_l2(const uint32_t* __restrict__ l2, uint32_t* s_tmp, uint32_t& kk, uint32_t& digit) {
  constexpr int BPT = kShardL2Bins / kBlock;  // 32
  uint32_t cb[BPT], c = 0;
#pragma unroll
  for (int q = 0; q < BPT / 4; ++q) {
    const uint4 u = *reinterpret_cast<const uint4*>(l2 + (size_t)threadIdx.x * BPT + 4 * q);
    cb[4 * q] = u.x;
    cb[4 * q + 1] = u.y;
    cb[4 * q + 2] = u.z;
    cb[4 * q + 3] = u.w;
    c += (u.x + u.y) + (u.z + u.w);
  }
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
    for (int q = 0; q < BPT; ++q) {
      if (cb[q] > 0 && acc <= kk && kk < acc + cb[q]) {
        s_tmp[40] = threadIdx.x * BPT + q;
        s_tmp[41] = kk - acc;
      }
      acc += cb[q];
    }
  }
  __syncthreads();
  digit = s_tmp[40];
  kk = s_tmp[41];
  __syncthreads();
}

// raw moments of one pair: p the transformed reading point, q its reference point, n the reference normal (fp32 values, products
// in fp64: exact up to the last sum)
__device__ __forceinline__ void shard_add_moments(double (&acc)[kMom], float pxf, float pyf, float pzf, const float4& qf, const float4& nf) {
  const double P[3] = {(double)pxf, (double)pyf, (double)pzf}, Q[3] = {(double)qf.x, (double)qf.y, (double)qf.z},
               Nn[3] = {(double)nf.x, (double)nf.y, (double)nf.z};
  const double c[3] = {P[1] * Nn[2] - P[2] * Nn[1], P[2] * Nn[0] - P[0] * Nn[2], P[0] * Nn[1] - P[1] * Nn[0]};
  const double r = (Nn[0] * (P[0] - Q[0]) + Nn[1] * (P[1] - Q[1])) + Nn[2] * (P[2] - Q[2]);
  int t = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a; b < 3; ++b) acc[t++] += c[a] * c[b];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[t++] += c[a] * Nn[b];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a; b < 3; ++b) acc[t++] += Nn[a] * Nn[b];
#pragma unroll
  for (int a = 0; a < 3; ++a) acc[t++] += c[a] * r;
#pragma unroll
  for (int a = 0; a < 3; ++a) acc[t++] += Nn[a] * r;
#pragma unroll
  for (int a = 0; a < 3; ++a) acc[t++] += P[a];
#pragma unroll
  for (int a = 0; a < 3; ++a) acc[t++] += Q[a];
  acc[t] += 1.0;
}

// block-wide fixed-order sums of the 34 moments in two halves of 17 (one BlockSum<34> would need 72 KB of LDS): tot[c] on every thread
// that asks through `total17`; s_a / s_b: BlockSum<17, kBlock> buffers
using MomSum = BlockSum<17, kBlock>;

constexpr int kShardPark = 1024;  // pairs with the 24-bit prefix kept in LDS (more: every bin walks the candidate records itself)

// grid = nb_part + 1 blocks of kBlock threads.  xm = region M of the exchange buffer.
__global__ void __launch_bounds__(kBlock) k_shard_moments(ChainParams cp, const IcpState* __restrict__ st, SelScratch* __restrict__ ss,
                                                          const uint32_t* __restrict__ l2 /*reduced level-2 histogram [8192]*/,
                                                          const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz, int N,
                                                          const float4* __restrict__ mq, const float4* __restrict__ mn, const int32_t* __restrict__ pos,
                                                          const float* __restrict__ d2, double* __restrict__ xm, int nb_part,
                                                          uint32_t* __restrict__ hist_zero /*level-1 replicas*/, int n_rep,
                                                          const CandRec* __restrict__ cand, const uint32_t* __restrict__ cand_cnt, int nb_cls,
                                                          uint32_t* __restrict__ base_scratch /*[nb_cls + 1], used when nb_cls > kBaseCap*/) {
  __shared__ uint32_t s_tmp[64];
  __shared__ double s_a[MomSum::kWordsA];
  __shared__ double s_b[MomSum::kWordsB];
  __shared__ uint32_t s_idx[kShardPark], s_ord[kShardPark], s_bins[kShardL3Bins], s_cnt;
  __shared__ uint32_t s_base[kBaseCap + 1];
  const float hv = hdr_load(st);
  if (hdr_i(hv, H_DONE)) return;
  const bool failed = hdr_i(hv, H_STATUS) != 0;  // uniform; a failed rank still takes part in the exchange, with zeros
  const uint32_t skip = ss->skip;
  const bool sel = !failed && !skip;               // a Trimmed filter with something to select
  uint32_t prefix24 = 0xffffffffu;                 // no selection: every finite pair is "below"
  if (sel) {
    uint32_t kk = ss->kk, d1;
    shard_pick_digit_l2(l2, s_tmp, kk, d1);
    prefix24 = (ss->bin << kShardL2Bits) | d1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the closing kernel starts from the digit and the rank inside it (it does not scan level 2 again)
      ss->pad[0] = d1;
      ss->pad[1] = kk;
    }
  }
  // the level-1 replicas were consumed by k_classify: cleared here for the next k_match2, spread over all blocks
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < n_rep * kHistBins; k += gridDim.x * kBlock) hist_zero[k] = 0u;
  float T[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) T[k] = hdr_f(hv, k);
  double acc[kMom];
#pragma unroll
  for (int c = 0; c < kMom; ++c) acc[c] = 0.0;
  if ((int)blockIdx.x < nb_part) {
    // ---- this block's share of the pairs that are kept whatever the last seven bits of the limit turn out to be ----
    if (!failed)
      for (int base = blockIdx.x * (kBlock * kNePPT) + threadIdx.x; base < N; base += nb_part * (kBlock * kNePPT)) {
        // everything a trip needs travels in ONE round trip (coalesced streams, clamped addresses); the decisions come afterwards
        int pe[kNePPT];
        float d[kNePPT], x0[kNePPT], y0[kNePPT], z0[kNePPT];
        float4 q[kNePPT], n[kNePPT];
#pragma unroll
        for (int u = 0; u < kNePPT; ++u) {
          const int i = min(base + u * kBlock, N - 1);
          pe[u] = pos[i];
          d[u] = d2[i];
          x0[u] = rx[i];
          y0[u] = ry[i];
          z0[u] = rz[i];
          q[u] = mq[i];
          n[u] = mn[i];
        }
#pragma unroll
        for (int u = 0; u < kNePPT; ++u) {
          const bool in = base + u * kBlock < N;
          bool keep = in && pe[u] >= 0 && d[u] <= cp.max_out_r2 && d[u] != kInfF;
          keep = keep && !(sel && (__float_as_uint(d[u]) >> kShardL3Bits) >= prefix24);  // beyond the limit, or decided by its last bits
          if (keep) shard_add_moments(acc, xf_row(T, 0, x0[u], y0[u], z0[u]), xf_row(T, 1, x0[u], y0[u], z0[u]), xf_row(T, 2, x0[u], y0[u], z0[u]), q[u], n[u]);
        }
      }
    double lo[17], hi[17];
#pragma unroll
    for (int c = 0; c < 17; ++c) {
      lo[c] = acc[c];
      hi[c] = acc[17 + c];
    }
    MomSum::run(lo, s_a, s_b);
    if (threadIdx.x < 17) xm[kXmPart + (size_t)threadIdx.x * nb_part + blockIdx.x] = MomSum::total(s_b, threadIdx.x);
    __syncthreads();
    MomSum::run(hi, s_a, s_b);
    if (threadIdx.x < 17) xm[kXmPart + (size_t)(17 + threadIdx.x) * nb_part + blockIdx.x] = MomSum::total(s_b, threadIdx.x);
    return;
  }
  // ---- the last block: the pairs whose 24 leading bits ARE the prefix — counted per value of their last seven bits (every
  //      finite one: the rank statistic counts them all), the kept ones' raw moments summed per bin in INDEX order ----
  if (threadIdx.x < kShardL3Bins) s_bins[threadIdx.x] = 0u;
  if (threadIdx.x == 0) s_cnt = 0u;
  __syncthreads();
  uint32_t total = 0;
  const uint32_t* base = nb_cls <= kBaseCap ? s_base : base_scratch;
  if (sel) {  // uniform
    // k_classify left the trim bin's pairs as records in one region per classify block: a flat sweep over them (a few per cent of the
    // reading) instead of over the reading.  Bases of the regions' runs first; a record's region by binary search in LDS.
    uint32_t* wbase = nb_cls <= kBaseCap ? s_base : base_scratch;
    const int per_thread = (nb_cls + kBlock - 1) / kBlock;
    const int b0 = min((int)threadIdx.x * per_thread, nb_cls), b1 = min(b0 + per_thread, nb_cls);
    uint32_t mine = 0;
    for (int b = b0; b < b1; ++b) mine += cand_cnt[b];
    uint32_t run = block_excl_scan(mine, &total, s_tmp);
    for (int b = b0; b < b1; ++b) {
      wbase[b] = run;
      run += cand_cnt[b];
    }
    if (threadIdx.x == 0) wbase[nb_cls] = total;
    __threadfence_block();
    __syncthreads();
    constexpr int PER = 8;  // records in flight per thread: their searches run interleaved, their loads go out together (one round trip per batch)
    for (uint32_t f0 = 0; f0 < total; f0 += (uint32_t)(kBlock * PER)) {  // uniform trip count
      uint32_t bits[PER], keepw[PER];
      bool ok[PER];
      const CandRec* src[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint32_t f = f0 + threadIdx.x + (uint32_t)k * kBlock;
        ok[k] = f < total;
        const uint32_t fc = ok[k] ? f : 0u;
        const int b = flat_block(base, nb_cls, fc);
        src[k] = cand + ((size_t)b * kClsBlock + (fc - base[b]));
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const uint4 w = reinterpret_cast<const uint4*>(src[k])[0];  // px py pz bits
        bits[k] = w.w;
        keepw[k] = reinterpret_cast<const uint32_t*>(src[k])[7];    // keep | slot << 1
      }
#pragma unroll
      for (int k = 0; k < PER; ++k)
        if (ok[k] && (bits[k] >> kShardL3Bits) == prefix24) {
          atomicAdd(&s_bins[bits[k] & (uint32_t)(kShardL3Bins - 1)], 1u);
          const uint32_t slot = atomicAdd(&s_cnt, 1u);
          if (slot < (uint32_t)kShardPark) s_idx[slot] = keepw[k] >> 1;  // the pair's slot in the reading
        }
    }
  }
  __syncthreads();
  const uint32_t m = s_cnt;
  if (threadIdx.x < kShardL3Bins) xm[kXmCnt + threadIdx.x] = (double)s_bins[threadIdx.x];
  auto take = [&](int i, uint32_t bin) {
    const int pe = pos[i];
    const float d = d2[i];
    const uint32_t u = __float_as_uint(d);
    if (!((pe >= 0 || pe <= -2) && d != kInfF && (u >> kShardL3Bits) == prefix24 && (u & (uint32_t)(kShardL3Bins - 1)) == bin)) return;
    if (!(pe >= 0 && d <= cp.max_out_r2)) return;  // counted above, but another weight of the chain is 0
    const float x0 = rx[i], y0 = ry[i], z0 = rz[i];
    shard_add_moments(acc, xf_row(T, 0, x0, y0, z0), xf_row(T, 1, x0, y0, z0), xf_row(T, 2, x0, y0, z0), mq[i], mn[i]);
  };
  if (m <= (uint32_t)kShardPark) {
    // parked in arrival order, summed in index order: the rank of every parked index among the m, then lane t walks the ordered
    // list and adds the kept pairs of bin t
    for (uint32_t j = threadIdx.x; j < m; j += kBlock) {
      const uint32_t ij = s_idx[j];
      uint32_t rank = 0;
      for (uint32_t k = 0; k < m; ++k) rank += s_idx[k] < ij ? 1u : 0u;
      s_ord[rank] = ij;
    }
    __syncthreads();
    if (threadIdx.x < kShardL3Bins)
      for (uint32_t r = 0; r < m; ++r) take((int)s_ord[r], threadIdx.x);
  } else if (threadIdx.x < kShardL3Bins) {  // heavy ties (more than 1 024 pairs share 24 leading bits): every bin walks the records itself, in flat (= index) order
    for (uint32_t f = 0; f < total; ++f) {
      const int b = flat_block(base, nb_cls, f);
      const CandRec r = cand[(size_t)b * kClsBlock + (f - base[b])];
      if ((r.bits >> kShardL3Bits) == prefix24 && (r.bits & (uint32_t)(kShardL3Bins - 1)) == threadIdx.x) take((int)((uint32_t)r.keep >> 1), threadIdx.x);
    }
  }
  if (threadIdx.x < kShardL3Bins) {
#pragma unroll
    for (int c = 0; c < kMom; ++c) xm[kXmBin + c * kShardL3Bins + threadIdx.x] = acc[c];
  }
}

// The front of the sharded chain's closing step (one block of kBlock threads): from the REDUCED region M and level-2 histogram to
// what solve_body expects — the 21 + 6 sums of the centred normal equations in L.s_sum and limit / means / |K| / status as a
// SolveOverride.  Returns false when the chain has ended before (nothing to form).
__device__ __forceinline__ bool shard_solve_front(const ChainParams& cp, const IcpState* __restrict__ st, const SelScratch* __restrict__ ss,
                                                  uint32_t* __restrict__ l2, const double* __restrict__ xm, int nb_part, float hv, SolveLds& L,
                                                  SolveOverride& ov, uint32_t* s_tmp, double* s_tot /*[kMom]*/) {
  if (hdr_i(hv, H_DONE)) return false;
  O3S_TSTAMP(48);
  const int status = hdr_i(hv, H_STATUS);
  const uint32_t skip = ss->skip;
  const bool sel = status == 0 && !skip;
  uint32_t d0 = (uint32_t)(kShardL3Bins - 1);
  float limit = kInfF;
  if (sel) {  // uniform
    const uint32_t d1 = ss->pad[0];  // level-2 digit and the rank inside it: block 0 of k_shard_moments left them
    uint32_t kk = ss->pad[1];
    O3S_TSTAMP(49);
    // level 3: 128 bins, one per lane of the first two waves; the counts were summed as doubles (exact)
    const uint32_t c3 = threadIdx.x < kShardL3Bins ? (uint32_t)xm[kXmCnt + threadIdx.x] : 0u;
    uint32_t tot3;
    const uint32_t ex3 = block_excl_scan(c3, &tot3, s_tmp);
    __syncthreads();
    if (threadIdx.x == 0) s_tmp[44] = 0u;
    __syncthreads();
    if (c3 > 0 && ex3 <= kk && kk < ex3 + c3) s_tmp[44] = threadIdx.x;
    __syncthreads();
    d0 = s_tmp[44];
    limit = __uint_as_float((ss->bin << 20) | (d1 << kShardL3Bits) | d0);
  }
  for (int k = threadIdx.x; k < kShardL2Bins; k += kBlock) l2[k] = 0u;  // k_shard_moments was its last reader: ready for the next iteration's k_classify
  O3S_TSTAMP(50);
  // moments: block partials in block order, then the bins up to the digit in bin order — one fixed-order block sum per half
  const int nbm1 = nb_part > 0 ? nb_part - 1 : 0;
  const int t = threadIdx.x;
  const bool two = nb_part > kBlock;              // uniform: two block partials per lane (readings beyond 131 k points per rank)
  const bool binned = sel && t < kShardL3Bins && (uint32_t)t <= d0;
  // every load first, branch-free (clamped addresses), ONE round trip for both halves; the masks afterwards: a load under a
  // condition compiles to an exec-mask region with its own wait — 34 dependent round trips per half, 17 us when it was written so
  double pa[kMom], pc[kMom];
#pragma unroll
  for (int c = 0; c < kMom; ++c) {
    pa[c] = xm[kXmPart + (size_t)c * nb_part + min(t, nbm1)];
    pc[c] = xm[kXmBin + c * kShardL3Bins + min(t, kShardL3Bins - 1)];
  }
#pragma unroll
  for (int c = 0; c < kMom; ++c) {
    pa[c] = t < nb_part ? pa[c] : 0.0;
    pa[c] += binned ? pc[c] : 0.0;
  }
  if (two) {  // uniform
#pragma unroll
    for (int c = 0; c < kMom; ++c) pc[c] = xm[kXmPart + (size_t)c * nb_part + min(t + kBlock, nbm1)];
#pragma unroll
    for (int c = 0; c < kMom; ++c) pa[c] += (t + kBlock < nb_part) ? pc[c] : 0.0;
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    double v[17];
#pragma unroll
    for (int c = 0; c < 17; ++c) v[c] = pa[half * 17 + c];
    __syncthreads();
    MomSum::run(v, L.s_a, L.s_b);
    if (t < 17) s_tot[half * 17 + t] = MomSum::total(L.s_b, t);
  }
  __syncthreads();
  O3S_TSTAMP(51);
  if (t == 0) {
    ov.limit = limit;
    ov.has_limit = (!cp.has_trim || !skip) ? 1 : 0;
    ov.status = 0;
    ov.kept = hdr_i(hv, H_KEPT);
    for (int d = 0; d < 3; ++d) {
      ov.mp[d] = hdr_f(hv, H_MP + d);
      ov.mq[d] = hdr_f(hv, H_MQ + d);
    }
    if (status == 0) {
      const double* M = s_tot;
      const double K = M[33];
      ov.kept = (int32_t)K;
      if (K == 0.0) {  // "no point to minimize" (ErrorMinimizer.cpp:75-77)
        ov.status = 6;
        for (int c = 0; c < kNeComps; ++c) L.s_sum[c] = 0.0;
      } else {
        // rowwise().mean(): fp64 sums rounded once to fp32 — the means the step is built with; the centring uses the same values
        float mpf[3], mqf[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          mpf[d] = (float)(M[27 + d] / K);
          mqf[d] = (float)(M[30 + d] / K);
          ov.mp[d] = mpf[d];
          ov.mq[d] = mqf[d];
        }
        const double mp[3] = {(double)mpf[0], (double)mpf[1], (double)mpf[2]};
        const double dd[3] = {mp[0] - (double)mqf[0], mp[1] - (double)mqf[1], mp[2] - (double)mqf[2]};
        // every index below is a compile-time constant (fully unrolled): the small matrices live in registers — with run-time
        // subscripts they were private-memory arrays and this lane's ~300 fp64 operations took 20 us
        double Scc[3][3], Scn[3][3], Snn[3][3];
        {
          int k = 0;
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) {
              Scc[a][b] = M[k];
              Scc[b][a] = M[k];
              ++k;
            }
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Scn[a][b] = M[k++];
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) {
              Snn[a][b] = M[k];
              Snn[b][a] = M[k];
              ++k;
            }
        }
        const double Scr[3] = {M[21], M[22], M[23]}, Snr[3] = {M[24], M[25], M[26]};
        // X = [mp]x : (mp x v) = X v
        const double X[3][3] = {{0.0, -mp[2], mp[1]}, {mp[2], 0.0, -mp[0]}, {-mp[1], mp[0], 0.0}};
        double XS[3][3], XSX[3][3], CX[3][3];  // X Snn | X Snn X^T | Scn X^T
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            double sx = 0.0, u = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              sx += X[a][k] * Snn[k][b];
              u += Scn[a][k] * X[b][k];
            }
            XS[a][b] = sx;
            CX[a][b] = u;
          }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            double sx = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) sx += XS[a][k] * X[b][k];
            XSX[a][b] = sx;
          }
        double A[6][6];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            A[a][b] = ((Scc[a][b] - CX[a][b]) - CX[b][a]) + XSX[a][b];      // sum (c - mp x n)(c - mp x n)^T
            A[a][3 + b] = Scn[a][b] - XS[a][b];                              // sum (c - mp x n) n^T
            A[3 + a][3 + b] = Snn[a][b];
            A[3 + a][b] = 0.0;                                               // (lower left: not used)
          }
        double gh[6];  // sum g h, h = r - n.d
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          double scnd = 0.0, xsnr = 0.0, xsd = 0.0, snnd = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            scnd += Scn[a][k] * dd[k];
            xsnr += X[a][k] * Snr[k];
            xsd += XS[a][k] * dd[k];
            snnd += Snn[a][k] * dd[k];
          }
          gh[a] = ((Scr[a] - scnd) - xsnr) + xsd;
          gh[3 + a] = Snr[a] - snnd;
        }
        {
          int k = 0;
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) L.s_sum[k++] = A[a][b];
#pragma unroll
          for (int a = 0; a < 6; ++a) L.s_sum[21 + a] = gh[a];
        }
      }
    }
  }
  __syncthreads();
  O3S_TSTAMP(52);
  return true;
}

// k_solve_shard — the closing step of the sharded chain: shard_solve_front, then the unsharded chain's solve_body on the sums it formed
__global__ void __launch_bounds__(kBlock) k_solve_shard(ChainParams cp, IcpState* __restrict__ st, const SelScratch* __restrict__ ss, uint32_t* __restrict__ l2,
                                                        const double* __restrict__ xm, int nb_part, int N_total, float* __restrict__ trace_T,
                                                        float* __restrict__ trace_limit, int64_t* __restrict__ trace_kept, int trace_cap,
                                                        HostPost* __restrict__ post) {
  __shared__ SolveLds lds;
  __shared__ SolveOverride s_ov;
  __shared__ uint32_t s_tmp[64];
  __shared__ double s_tot[kMom];
  const float hv = hdr_load(st);
  const bool formed = shard_solve_front(cp, st, ss, l2, xm, nb_part, hv, lds, s_ov, s_tmp, s_tot);
  solve_body<kBlock, false>(nullptr, 0, N_total, cp, st, trace_T, trace_limit, trace_kept, trace_cap, 1, post, lds, formed ? &s_ov : nullptr);
  O3S_TSTAMP(53);
}

}  // namespace kern
}  // namespace o3s
