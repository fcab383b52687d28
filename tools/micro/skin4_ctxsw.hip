// Stand-alone reproducer attempt for the shared-GPU anomaly of DESIGN.md section 10: the ACTUAL skinning kernel of
// csrc/smpl.hip (smpl_skin4_kernel: thread = vertex, 4 (joint, weight) pairs in registers, the person's 24 joint transforms
// staged in LDS, persons looped) with its launch geometry, on synthetic operands, repeated; every launch is compared bit for bit
// with the first.  No library, no torch: if two concurrent copies of THIS program show mismatches, the defect does not depend
// on anything else in libtepose_hip.so.
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/skin4_ctxsw.hip -o /tmp/skin4            (packed fp32 VALU code allowed: the failing build)
//   hipcc ... -Xclang -target-feature -Xclang -packed-fp32-ops ... -o /tmp/skin4_nopk   (the shipped code generation)
//   (/tmp/skin4 4000 & /tmp/skin4 4000; wait)
// Variants (-DVAR=n), to bisect what the anomaly needs:
//   0  the kernel as shipped
//   1  transforms read from global memory (L2) instead of the LDS copy: no LDS, no barrier
//   2  LDS copy over-allocated (4 KB) and every element of it written each person
//   3  __launch_bounds__(256, 1): one workgroup per CU (other save-area footprint)
//   5  s_icache_inv at the top of every person iteration (every wave keeps invalidating the instruction cache, so the packed
//      block below is fetched with misses at varying points: instruction-fetch bubbles like the ones a second process causes)
//   6  variant 1 plus an in-kernel check: the first two rows again with scalar FMAs (asm) from the SAME operand registers; lanes whose
//      packed accumulators differ are counted per element in chk[] (argv: nothing to add; printed at the end)
//   4  the row accumulation as explicit scalar fmaf in asm-protected order (no packed op can be formed for it) while the rest of
//      the kernel keeps whatever the compiler chooses
// argv: launches [persons N = 512] [streams = 1] [heavy neighbour: MFMA iterations, 0 = none]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#ifndef VAR
#define VAR 0
#endif

constexpr int kNV = 6890, kNJ = 24, kVertLd = 20672, kSkinPG = 64;

#if VAR == 3
#define LB __launch_bounds__(256, 1)
#else
#define LB __launch_bounds__(256)
#endif

#if VAR == 6
__device__ unsigned chk[32];
#endif
__global__ void LB skin4(const int* __restrict__ cidx, const float* __restrict__ cval, const float* __restrict__ vposed,
                         const float* __restrict__ Amat, int N, float* __restrict__ verts, int pg) {
#if VAR == 2
  __shared__ __attribute__((aligned(16))) float As[1024];
#elif VAR != 1 && VAR != 6
  __shared__ __attribute__((aligned(16))) float As[kNJ * 12];
#endif
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int v = blockIdx.x * 256 + threadIdx.x;
  const bool ok = v < kNV;
  int jx[4];
  float wv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { jx[k] = ok ? cidx[v * 4 + k] * 12 : 0; wv[k] = ok ? cval[v * 4 + k] : 0.f; }
  const int p0 = blockIdx.y * pg;
  const int p1 = min(p0 + pg, N);
  for (int p = p0; p < p1; ++p) {
#if VAR == 5
    asm volatile("s_icache_inv\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" ::: "memory");
#endif
#if VAR == 1 || VAR == 6
    const float* As = Amat + (long)p * kNJ * 12;
#else
    __syncthreads();
#if VAR == 2
    for (int i = threadIdx.x; i < 1024; i += 256) As[i] = i < kNJ * 12 ? Amat[(long)p * kNJ * 12 + i] : 0.f;
#else
    for (int i = threadIdx.x; i < kNJ * 12; i += 256) As[i] = Amat[(long)p * kNJ * 12 + i];
#endif
    __syncthreads();
#endif
    float t[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) t[e] = 0.f;
#if VAR == 6
    float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f4 r0 = *(const f4*)(As + jx[k]), r1 = *(const f4*)(As + jx[k] + 4), r2 = *(const f4*)(As + jx[k] + 8);
#if VAR == 6
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(u[e]) : "v"(wv[k]), "v"(r0[e]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(u[4 + e]) : "v"(wv[k]), "v"(r1[e]));
      }
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#if VAR == 4
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[e]) : "v"(wv[k]), "v"(r0[e]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[4 + e]) : "v"(wv[k]), "v"(r1[e]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[8 + e]) : "v"(wv[k]), "v"(r2[e]));
#else
        t[e] += wv[k] * r0[e]; t[4 + e] += wv[k] * r1[e]; t[8 + e] += wv[k] * r2[e];
#endif
      }
    }
#if VAR == 6
    if (ok) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (__float_as_uint(u[e]) != __float_as_uint(t[e])) { atomicAdd(&chk[e], 1u); atomicAdd(&chk[8 + ((threadIdx.x & 63) >> 4)], 1u); }
    }
#endif
    if (ok) {
      const float* vp = vposed + (long)p * kVertLd + 3 * v;
      const float x = vp[0], y = vp[1], z = vp[2];
      float* o = verts + ((long)p * kNV + v) * 3;
      o[0] = t[0] * x + t[1] * y + t[2] * z + t[3];
      o[1] = t[4] * x + t[5] * y + t[6] * z + t[7];
      o[2] = t[8] * x + t[9] * y + t[10] * z + t[11];
    }
  }
}

// A neighbour in the launch sequence that looks like the blend-shape GEMM of the library's SMPL path to the context-switch
// machinery: 512 threads, 144 KB of LDS, MFMA accumulators (AGPR / unified register file), a few hundred microseconds long.
typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8h __attribute__((ext_vector_type(8)));
template <int MODE>   // 0: LDS reads + MFMA; 1: MFMA on constant register operands; 2: LDS reads only; 3: scalar-fp32 VALU FMAs only;
                      // 4: MFMA on register operands that change every iteration (no LDS)
__global__ void __launch_bounds__(512) heavy(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char big[144 * 1024];
  // random fp16 operands in [-2, 2): MFMA power depends on operand toggling (a loop on near-constant data draws far less)
  for (int i = threadIdx.x; i < 144 * 1024 / 2; i += 512) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    ((_Float16*)big)[i] = (_Float16)(((float)(h & 0xffff) - 32768.f) / 16384.f);
  }
  __syncthreads();
  f32x16h acc[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  h16x8h a = *(const h16x8h*)(big + threadIdx.x * 16), b = *(const h16x8h*)(big + 65536 + threadIdx.x * 16);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
      a = *(const h16x8h*)(big + ((threadIdx.x * 16 + it * 8192) % (144 * 1024 - 16) & ~15));
      b = *(const h16x8h*)(big + ((threadIdx.x * 16 + it * 4096 + 512) % (144 * 1024 - 16) & ~15));
    }
    if (MODE == 4) {           // MFMA operands that change every iteration WITHOUT touching LDS: two register-resident vectors swapped and
      const h16x8h t2 = a; a = b; b = t2;     // sign-flipped (same operand toggling as MODE 0, no LDS instruction in the loop)
      a = -a;
    }
    if (MODE == 0 || MODE == 1 || MODE == 4) {
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    } else if (MODE == 2) {
      acc[0][it & 15] += (float)a[0] + (float)b[1];
    } else {
      for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][e]) : "v"((float)a[e & 7]), "v"((float)b[j]));
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// The same neighbour with what the library's GEMMs add: LDS-DMA (global_load_lds_dwordx4: global memory straight into LDS, M0 =
// destination), a ring of stages, counted vmcnt waits
__global__ void __launch_bounds__(512) heavy_dma(const char* __restrict__ src, size_t src_bytes, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char big[144 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16h acc[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const char* g = src + ((size_t)blockIdx.x * 65536 + wave * 1024 + lane * 16) % (src_bytes - (1 << 20));
  for (int it = 0; it < iters; ++it) {
    // 8 waves x 4 instructions x 1 KB = 32 KB per iteration into slot it % 4
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + q * 8192),
                                       (__attribute__((address_space(3))) void*)(big + (it & 3) * 32768 + (wave * 4 + q) * 1024), 16, 0, 0);
    g += 32768;
    if (g > src + src_bytes - (1 << 20)) g = src + lane * 16;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const h16x8h a = *(const h16x8h*)(big + ((it + 3) & 3) * 32768 + (threadIdx.x & 255) * 16);
    const h16x8h b = *(const h16x8h*)(big + ((it + 3) & 3) * 32768 + 8192 + (threadIdx.x & 255) * 16);
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// bitwise comparison with the reference result; mism: this launch's own flag (reset by the first block is not needed: the
// flag word is per launch slot, cnt[32 + launch % 32] is only used to count a launch once)
__global__ void __launch_bounds__(256) compare(const float* __restrict__ a, const float* __restrict__ b, size_t n, unsigned* cnt, int launch) {
  unsigned local = 0, firstidx = 0xffffffffu, c3[3] = {0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    if (__float_as_uint(a[i]) != __float_as_uint(b[i])) { ++local; ++c3[i % 3]; if (firstidx == 0xffffffffu) firstidx = (unsigned)i; }
  if (local) {
    atomicAdd(&cnt[0], local);
    for (int c = 0; c < 3; ++c) if (c3[c]) atomicAdd(&cnt[1 + c], c3[c]);
    if (atomicMax(&cnt[6], (unsigned)launch + 1u) < (unsigned)launch + 1u) {      // first thread of this launch to report
      atomicAdd(&cnt[4], 1u);
      const unsigned k = atomicAdd(&cnt[5], 1u);
      if (k < 12) { cnt[8 + 2 * k] = (unsigned)launch; cnt[8 + 2 * k + 1] = firstidx; }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float urand() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((rng_state >> 40) & 0xFFFFFF) / 16777216.f;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 4000;
  const int N = argc > 2 ? atoi(argv[2]) : 512;
  const int nstreams = argc > 3 ? atoi(argv[3]) : 1;
  const int with_heavy = argc > 4 ? atoi(argv[4]) : 0;
  const int role = argc > 5 ? atoi(argv[5]) : 0;           // 0: skinning (+ neighbour in front when asked); 1: ONLY the MFMA neighbour; 2: ONLY the
                                                           // MFMA + LDS-DMA neighbour; 3: only LDS-DMA neighbour with no MFMA work to speak of     // > 0: the 144 KB-LDS MFMA kernel in front of every skinning launch, that many MFMA iterations
  std::vector<int> cidx(kNV * 4);
  std::vector<float> cval(kNV * 4), vposed((size_t)N * kVertLd), A((size_t)N * kNJ * 12);
  for (int v = 0; v < kNV; ++v) {
    float s = 0.f;
    for (int k = 0; k < 4; ++k) { cidx[v * 4 + k] = (int)(urand() * kNJ) % kNJ; cval[v * 4 + k] = urand() * urand() + 1e-3f; s += cval[v * 4 + k]; }
    for (int k = 0; k < 4; ++k) cval[v * 4 + k] /= s;
  }
  for (auto& x : vposed) x = urand() - 0.5f;
  for (auto& x : A) x = urand() * 2.f - 1.f;
  int* d_cidx; float *d_cval, *d_vp, *d_A;
  CK(hipMalloc(&d_cidx, cidx.size() * 4)); CK(hipMalloc(&d_cval, cval.size() * 4));
  CK(hipMalloc(&d_vp, vposed.size() * 4)); CK(hipMalloc(&d_A, A.size() * 4));
  CK(hipMemcpy(d_cidx, cidx.data(), cidx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cval, cval.data(), cval.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_vp, vposed.data(), vposed.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_A, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  const size_t on = (size_t)N * kNV * 3;
  std::vector<hipStream_t> st(nstreams);
  std::vector<float*> d_out(nstreams);
  for (int i = 0; i < nstreams; ++i) { CK(hipStreamCreate(&st[i])); CK(hipMalloc(&d_out[i], on * 4)); }
  float* d_ref; unsigned* d_cnt;
  CK(hipMalloc(&d_ref, on * 4)); CK(hipMalloc(&d_cnt, 64 * sizeof(unsigned)));
  CK(hipMemset(d_cnt, 0, 64 * sizeof(unsigned)));
  const int vb = (kNV + 255) / 256;
  int pg = (int)((long)N * vb / 512);
  pg = pg < 1 ? 1 : (pg > kSkinPG ? kSkinPG : pg);
  dim3 grid(vb, (N + pg - 1) / pg);
  hipLaunchKernelGGL(skin4, grid, dim3(256), 0, st[0], d_cidx, d_cval, d_vp, d_A, N, d_ref, pg);
  CK(hipStreamSynchronize(st[0]));
  if (role) {
    char* d_src; const size_t sb = (size_t)512 << 20;
    CK(hipMalloc(&d_src, sb)); CK(hipMemset(d_src, 0x3c, sb));
    for (int it = 0; it < launches; ++it) {
      if (role == 1) hipLaunchKernelGGL(heavy<0>, dim3(256), dim3(512), 0, st[0], d_out[0], with_heavy ? with_heavy : 2000);
      else if (role == 11) hipLaunchKernelGGL(heavy<1>, dim3(256), dim3(512), 0, st[0], d_out[0], with_heavy ? with_heavy : 2000);
      else if (role == 12) hipLaunchKernelGGL(heavy<2>, dim3(256), dim3(512), 0, st[0], d_out[0], with_heavy ? with_heavy : 2000);
      else if (role == 14) hipLaunchKernelGGL(heavy<4>, dim3(with_heavy < 0 ? 64 : 256), dim3(512), 0, st[0], d_out[0], 4000);
      else if (role == 15) hipLaunchKernelGGL(heavy<0>, dim3(64), dim3(512), 0, st[0], d_out[0], 4000);      // only 64 workgroups: a quarter of the CUs
      else if (role == 13) hipLaunchKernelGGL(heavy<3>, dim3(256), dim3(512), 0, st[0], d_out[0], with_heavy ? with_heavy : 500);
      else hipLaunchKernelGGL(heavy_dma, dim3(256), dim3(512), 0, st[0], d_src, sb, d_out[0], with_heavy ? with_heavy : 500);
      if (it % 64 == 63) CK(hipStreamSynchronize(st[0]));
    }
    CK(hipDeviceSynchronize());
    printf("neighbour role %d done: %d launches\n", role, launches);
    return 0;
  }
  hipStream_t nb_stream = nullptr;
  float* d_nb = nullptr;
  if (role == 0 && with_heavy < 0) {                     // negative: the MFMA neighbour runs on a SECOND STREAM of this process
    CK(hipStreamCreate(&nb_stream)); CK(hipMalloc(&d_nb, 4096));
  }
  // launches are queued back to back (the GPU never idles, so a time slice of the other process always interrupts a skinning
  // kernel); each is compared ON THE DEVICE with the first result: cnt[0] wrong floats, cnt[1..3] by coordinate, cnt[4] launches
  // with a mismatch, cnt[8..] (launch, first wrong index) of the first few
  long bad_launches = 0, bad_floats = 0;
  for (int it = 0; it < launches; it += nstreams) {
    if (nb_stream) hipLaunchKernelGGL(heavy<0>, dim3(256), dim3(512), 0, nb_stream, d_nb, -with_heavy);
    for (int i = 0; i < nstreams; ++i) {
      if (with_heavy > 0) hipLaunchKernelGGL(heavy<0>, dim3(160), dim3(512), 0, st[i], d_out[i], with_heavy);
      hipLaunchKernelGGL(skin4, grid, dim3(256), 0, st[i], d_cidx, d_cval, d_vp, d_A, N, d_out[i], pg);
      hipLaunchKernelGGL(compare, dim3(1024), dim3(256), 0, st[i], d_out[i], d_ref, on, d_cnt, it + i);
    }
    if ((it / nstreams) % 64 == 63)
      for (int i = 0; i < nstreams; ++i) CK(hipStreamSynchronize(st[i]));
  }
  CK(hipDeviceSynchronize());
  unsigned cnt[64];
  CK(hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
  bad_floats = cnt[0]; bad_launches = cnt[4];
  for (unsigned k = 0; k < cnt[5] && k < 12; ++k) {
    const unsigned idx = cnt[8 + 2 * k + 1];
    printf("launch %u: first wrong float: person %u vertex %u coordinate %u\n", cnt[8 + 2 * k], idx / (kNV * 3), idx % (kNV * 3) / 3, idx % 3);
  }
  printf("wrong floats by coordinate: x %u y %u z %u\n", cnt[1], cnt[2], cnt[3]);
#if VAR == 6
  unsigned hc[32];
  CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(chk), sizeof(hc)));
  printf("in-kernel check, lanes whose packed accumulator differs from the scalar one: row 0 elements %u %u %u %u | row 1 elements %u %u %u %u | by quarter wave %u %u %u %u\n",
         hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7], hc[8], hc[9], hc[10], hc[11]);
#endif
  printf("VAR %d: %d launches, %ld with mismatches, %ld wrong floats\n", VAR, launches, bad_launches, bad_floats);
  return 0;
}
