// atomic_bench.hip — what one returning global atomic per point costs on MI355X (round 5: k_read_prep's arrival ranks).
// N points fall into `bins` addresses of a larger array (uniformly, or piled: a tenth of the bins take most points);
//   ret      : v[i] = atomicAdd(&c[b], 1)          (the arrival rank k_read_prep draws)
//   noret    : atomicAdd(&c[b], 1)                 (counting only)
//   lds_agg  : ranks inside the block through an LDS hash of the block's bins, one returning atomic per (block, distinct bin)
// hipcc -O3 --offload-arch=gfx950 atomic_bench.hip -o atomic_bench && ./atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_ret(const uint32_t* __restrict__ bin, int n, uint32_t* __restrict__ c, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = atomicAdd(&c[bin[i]], 1u);
}
__global__ void __launch_bounds__(256) k_noret(const uint32_t* __restrict__ bin, int n, uint32_t* __restrict__ c, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    atomicAdd(&c[bin[i]], 1u);
    out[i] = bin[i];
  }
}
// block-local ranks: an open-addressing table of the block's distinct bins in LDS (512 slots for 256 points)
__global__ void __launch_bounds__(256) k_lds(const uint32_t* __restrict__ bin, int n, uint32_t* __restrict__ c, uint32_t* __restrict__ out) {
  __shared__ uint32_t s_key[512], s_cnt[512], s_base[512];
  for (int k = threadIdx.x; k < 512; k += 256) { s_key[k] = 0xffffffffu; s_cnt[k] = 0u; }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint32_t b = 0, slot = 0, r = 0;
  const bool in = i < n;
  if (in) {
    b = bin[i];
    slot = (b * 2654435761u) >> 23;  // 9 bits
    for (;;) {
      const uint32_t prev = atomicCAS(&s_key[slot], 0xffffffffu, b);
      if (prev == 0xffffffffu || prev == b) break;
      slot = (slot + 1) & 511;
    }
    r = atomicAdd(&s_cnt[slot], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 512; k += 256)
    if (s_cnt[k]) s_base[k] = atomicAdd(&c[s_key[k]], s_cnt[k]);
  __syncthreads();
  if (in) out[i] = s_base[slot] + r;
}

int main() {
  const int n = 100000, range = 1 << 20;
  uint32_t *d_bin, *d_c, *d_out;
  CK(hipMalloc(&d_bin, n * 4)); CK(hipMalloc(&d_c, range * 4)); CK(hipMalloc(&d_out, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::mt19937 rng(1);
  for (int mode = 0; mode < 3; ++mode) {
    const int bins = mode == 0 ? 50000 : (mode == 1 ? 10000 : 2000);
    std::vector<uint32_t> addr(bins), h(n);
    for (auto& a : addr) a = rng() % range;
    for (auto& v : h) v = addr[rng() % bins];
    CK(hipMemcpy(d_bin, h.data(), n * 4, hipMemcpyHostToDevice));
    for (int which = 0; which < 3; ++which) {
      float best = 1e9f;
      for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemset(d_c, 0, range * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL(k_ret, dim3((n + 255) / 256), dim3(256), 0, 0, d_bin, n, d_c, d_out);
        else if (which == 1) hipLaunchKernelGGL(k_noret, dim3((n + 255) / 256), dim3(256), 0, 0, d_bin, n, d_c, d_out);
        else hipLaunchKernelGGL(k_lds, dim3((n + 255) / 256), dim3(256), 0, 0, d_bin, n, d_c, d_out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      std::printf("bins %6d  %-8s best %.2f us (event pair, includes ~6 us of launch path)\n", bins, which == 0 ? "ret" : which == 1 ? "noret" : "lds_agg", best * 1e3f);
    }
  }
  return 0;
}
