// sort_bench.hip — the mid-size pair sorts of the per-scan loop (round 5): 64-bit voxel keys with ~30 significant bits + 32-bit values,
// 20 k .. 130 k pairs (a sweep's voxel keys in the receiving thread's voxel grid and in the merge insert).  rocPRIM's default for these
// sizes is a block sort + one odd-even merge launch per doubling (13 merge launches per sweep in the loop's kernel trace); timed here
// against Onesweep forced early and merge configurations with larger sorted blocks, back to back on one stream (launch gaps included:
// that is what the mapping thread waits for).
// hipcc -O3 --offload-arch=gfx950 sort_bench.hip -o sort_bench && ./sort_bench
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using Default = rocprim::default_config;
using Onesweep = rocprim::radix_sort_config<Default, Default, Default, 4096>;
template <unsigned OE, unsigned SB, unsigned IPT>
using Merge = rocprim::radix_sort_config<Default, rocprim::merge_sort_config<OE, SB, IPT>, Default, 1024 * 1024>;

template <class Config>
int run(const char* name, size_t n, int end_bit, const uint64_t* dk, const uint32_t* dv, uint64_t* ok, uint32_t* ov, const std::vector<uint64_t>& ref, hipStream_t s) {
  size_t tb = 0;
  CK(rocprim::radix_sort_pairs<Config>(nullptr, tb, dk, ok, dv, ov, n, 0, (unsigned)end_bit, s));
  void* tmp = nullptr;
  CK(hipMalloc(&tmp, tb + 256));
  for (int w = 0; w < 5; ++w) CK(rocprim::radix_sort_pairs<Config>(tmp, tb, dk, ok, dv, ov, n, 0, (unsigned)end_bit, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int reps = 200;
  CK(hipEventRecord(a, s));
  for (int r = 0; r < reps; ++r) CK(rocprim::radix_sort_pairs<Config>(tmp, tb, dk, ok, dv, ov, n, 0, (unsigned)end_bit, s));
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<uint64_t> got(n);
  CK(hipMemcpy(got.data(), ok, n * 8, hipMemcpyDeviceToHost));
  const bool same = got == ref;
  std::printf("  %-34s %8.2f us per sort   %s\n", name, 1e3 * ms / reps, same ? "ok" : "WRONG");
  CK(hipFree(tmp));
  return 0;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int end_bit = 31;
  for (size_t n : {20000ul, 45000ul, 70000ul, 100000ul, 130000ul, 260000ul}) {
    std::mt19937_64 rng(7 + n);
    std::vector<uint64_t> k(n);
    std::vector<uint32_t> v(n);
    for (size_t i = 0; i < n; ++i) {
      k[i] = (rng() >> 36) & ((1ull << end_bit) - 1);  // ~28 bits, many ties with the low range
      v[i] = (uint32_t)i;
    }
    std::vector<uint64_t> ref = k;
    std::stable_sort(ref.begin(), ref.end());
    uint64_t *dk, *ok;
    uint32_t *dv, *ov;
    CK(hipMalloc(&dk, n * 8));
    CK(hipMalloc(&ok, n * 8));
    CK(hipMalloc(&dv, n * 4));
    CK(hipMalloc(&ov, n * 4));
    CK(hipMemcpy(dk, k.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice));
    std::printf("n = %zu pairs (u64 key, %d bits; u32 value)\n", n, end_bit);
    if (run<Default>("rocPRIM default", n, end_bit, dk, dv, ok, ov, ref, s)) return 1;
    if (run<Onesweep>("Onesweep (merge limit 4096)", n, end_bit, dk, dv, ok, ov, ref, s)) return 1;
    if (run<Merge<512, 256, 4>>("merge, blocks of 256 x 4", n, end_bit, dk, dv, ok, ov, ref, s)) return 1;
    if (run<Merge<512, 256, 8>>("merge, blocks of 256 x 8", n, end_bit, dk, dv, ok, ov, ref, s)) return 1;
    if (run<Merge<512, 256, 16>>("merge, blocks of 256 x 16", n, end_bit, dk, dv, ok, ov, ref, s)) return 1;
    if (run<Merge<512, 512, 8>>("merge, blocks of 512 x 8", n, end_bit, dk, dv, ok, ov, ref, s)) return 1;
    if (run<Merge<256, 512, 8>>("merge, blocks of 512 x 8, odd-even 256", n, end_bit, dk, dv, ok, ov, ref, s)) return 1;
    if (run<Merge<1024, 1024, 4>>("merge, blocks of 1024 x 4", n, end_bit, dk, dv, ok, ov, ref, s)) return 1;
    CK(hipFree(dk));
    CK(hipFree(ok));
    CK(hipFree(dv));
    CK(hipFree(ov));
  }
  return 0;
}
