// copy_bench — which form of a device-to-device stream copy reaches the measured HBM ceiling on this box
// (MI355X_MICROARCH.md: 6.29 TB/s for a float4 copy).  Tuning tool only; the winner lives in o3s_stream_copy_gbs.
//   hipcc -O3 --offload-arch=gfx950 tools/native/copy_bench.hip -o tools/native/copy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy_strided(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    v4f v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (NT) __builtin_nontemporal_store(v[k], &dst[i + k * stride]);
      else dst[i + k * stride] = v[k];
    }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}
// one contiguous chunk of U*256 vectors per block, no grid-stride loop
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy_tile(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
  const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  v4f v[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i < n) v[k] = NT ? __builtin_nontemporal_load(&src[i]) : src[i];
  }
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i < n) {
      if (NT) __builtin_nontemporal_store(v[k], &dst[i]);
      else dst[i] = v[k];
    }
  }
}
template <int U>
__global__ void __launch_bounds__(256) k_read_only(const v4f* __restrict__ src, float* __restrict__ sink, size_t n) {
  const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  v4f acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const size_t i = base + (size_t)k * 256;
    if (i < n) acc += src[i];
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 30);
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const size_t n = bytes / 16;
  void *a, *b;
  hipMalloc(&a, bytes);
  hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes);
  hipMemset(b, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch, double factor) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.1f GB/s\n", name, factor * (double)bytes * reps / (ms * 1e-3) / 1e9);
  };
  const v4f* s = (const v4f*)a;
  v4f* d = (v4f*)b;
  for (int grid : {2048, 4096, 8192, 16384}) {
    char nm[64];
    snprintf(nm, 64, "strided U8 nt   grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_copy_strided<8, true>), dim3(grid), dim3(256), 0, 0, s, d, n); }, 2.0);
    snprintf(nm, 64, "strided U8 plain grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_copy_strided<8, false>), dim3(grid), dim3(256), 0, 0, s, d, n); }, 2.0);
    snprintf(nm, 64, "strided U4 plain grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_copy_strided<4, false>), dim3(grid), dim3(256), 0, 0, s, d, n); }, 2.0);
  }
  timeit("tile U1 plain", [&] { hipLaunchKernelGGL((k_copy_tile<1, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, s, d, n); }, 2.0);
  timeit("tile U4 plain", [&] { hipLaunchKernelGGL((k_copy_tile<4, false>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, s, d, n); }, 2.0);
  timeit("tile U8 plain", [&] { hipLaunchKernelGGL((k_copy_tile<8, false>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, s, d, n); }, 2.0);
  timeit("tile U4 nt", [&] { hipLaunchKernelGGL((k_copy_tile<4, true>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, s, d, n); }, 2.0);
  timeit("tile U8 nt", [&] { hipLaunchKernelGGL((k_copy_tile<8, true>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, s, d, n); }, 2.0);
  timeit("read only U8", [&] { hipLaunchKernelGGL((k_read_only<8>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, s, (float*)b, n); }, 1.0);
  timeit("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 2.0);
  timeit("hipMemsetAsync (write only)", [&] { hipMemsetAsync(b, 0, bytes, 0); }, 1.0);
  return 0;
}
