// Micro-benchmark (round 5): how fast can waves pull cache lines towards L2 through the SCALAR memory path (s_load_dword touches, one per
// 128-byte line), compared with vector loads (global_load_dword, one lane per line)?  Question behind it: the fused GRU step's cell update
// reads its gate pre-activations with HBM-latency vector loads that share the CU's vector-memory path with the other workgroup's LDS-DMA
// stream; if scalar touches can warm L2 for them at >= ~15 GB/s per CU of lines, a prefetch through the scalar path is worth building.
//   hipcc --offload-arch=gfx950 -O3 -o build/smem_touch tools/micro/smem_touch.hip && build/smem_touch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)

// every wave touches `per_wave` consecutive 128-byte lines of its own region; U touches in flight before each wait
template <int U, int MODE>   // MODE 0: s_load_dword, 1: global_load_dword (lane 0..U-1 one line each), 2: s_load_dwordx16 (64 B), two per line
__global__ void __launch_bounds__(256) touch_kernel(const char* base, long per_wave, unsigned* sink) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + threadIdx.x / 64));
  const char* p = base + (long)wave * per_wave * 128;
  unsigned acc = 0;
  for (long i = 0; i < per_wave; i += U) {
    if (MODE == 0) {
      unsigned v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const char* q = p + (i + u) * 128;
        const unsigned long long qa = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)qa), hi = __builtin_amdgcn_readfirstlane((unsigned)(qa >> 32));
        const unsigned long long qs = ((unsigned long long)hi << 32) | lo;
        asm volatile("s_load_dword %0, %1, 0x0" : "=&s"(v[u]) : "s"(qs) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // SMEM returns asynchronously: every destination must stay allocated until the wait (an asm output that looks dead is handed to
      // another value at once -- the first version of this file faulted on a base pointer overwritten by a late s_load_dwordx16)
#pragma unroll
      for (int u = 0; u < U; ++u) asm volatile("" : "+s"(v[u]));
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    } else if (MODE == 1) {
      const int lane = threadIdx.x & 63;
      if (lane < U) acc += *(const volatile unsigned*)(p + (i + lane) * 128);
    } else {
      typedef unsigned u16x __attribute__((ext_vector_type(16)));
      u16x v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const char* q = p + (i + u) * 128;
        const unsigned long long qa = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)qa), hi = __builtin_amdgcn_readfirstlane((unsigned)(qa >> 32));
        const unsigned long long qs = ((unsigned long long)hi << 32) | lo;
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(v[u]) : "s"(qs) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < U; ++u) asm volatile("" : "+s"(v[u]));
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][15];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int U, int MODE>
static void run(const char* name, const char* buf, size_t bytes, unsigned* sink, int wgs) {
  const long waves = (long)wgs * 4, per_wave = (long)(bytes / 128 / waves) / U * U;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((touch_kernel<U, MODE>), dim3(wgs), dim3(256), 0, 0, buf, per_wave, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double lines = (double)per_wave * waves;
    if (rep == 2) printf("%-44s U=%2d wgs=%4d: %8.3f ms  %7.1f M lines/s  = %7.1f GB/s of 128-B lines (%.2f GB/s per CU)\n", name, U, wgs, ms,
                         lines / ms / 1e3, lines * 128 / ms / 1e6, lines * 128 / ms / 1e6 / 256);
  }
}

int main() {
  const size_t bytes = (size_t)2 << 30;          // 2 GiB: beyond the 256 MiB Infinity Cache
  char* buf; unsigned* sink;
  CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&sink, 256));
  CK(hipMemset(buf, 1, bytes));
  for (int wgs : {256, 512, 2048}) {
    run<8, 0>("s_load_dword, one per 128-B line", buf, bytes, sink, wgs);
    run<16, 0>("s_load_dword, one per 128-B line", buf, bytes, sink, wgs);
    run<4, 2>("s_load_dwordx16 (64 B of each line)", buf, bytes, sink, wgs);
    run<16, 1>("global_load_dword, one lane per line", buf, bytes, sink, wgs);
    run<64, 1>("global_load_dword, one lane per line", buf, bytes, sink, wgs);
  }
  // L2-resident region (2 MiB per XCD-ish: 16 MiB total): the hit rate of the scalar path itself
  run<16, 0>("s_load_dword, 16 MiB region (L2 / MALL hits)", buf, (size_t)16 << 20, sink, 512);
  run<64, 1>("global_load_dword, 16 MiB region", buf, (size_t)16 << 20, sink, 512);
  return 0;
}
