// Minimal probe for the packed-fp32 erratum of DESIGN.md section 10 (found in round 4 with tools/micro/skin4_ctxsw.hip):
// a wave that runs a DEPENDENT CHAIN of v_pk_*_f32 instructions with scalar VALU instructions in between -- the tail of the
// skinning kernel, instruction for instruction -- while other workgroups on the same CU run ds_read_b128 + v_mfma loops, gets a
// wrong LOW half of the packed result in lanes 48..63.  One process, two streams, no library:
//     hipcc --offload-arch=gfx950 -O2 tools/micro/pk_chain_mfma.hip -o /tmp/pkc && /tmp/pkc 20000
// Victim: every lane runs the 8-instruction sequence (explicit registers v100..v115, inline asm) on lane-dependent inputs and, in
// separate asm statements, the same arithmetic with scalar v_fma_f32 / v_mul_f32 / v_add_f32 only; any bitwise difference is counted
// per quarter wave.  Neighbour: 512-thread workgroups, 144 KB of LDS, a loop of two ds_read_b128 + four v_mfma_f32_32x32x16_f16 on
// random fp16 data (MFMA on register operands, LDS reads alone, or scalar VALU alone do NOT trigger it: skin4_ctxsw.hip roles 11-14).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#ifndef CHAIN
#define CHAIN 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512) neighbour(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char big[144 * 1024];
  for (int i = threadIdx.x; i < 144 * 1024 / 2; i += 512) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    ((_Float16*)big)[i] = (_Float16)(((float)(h & 0xffff) - 32768.f) / 16384.f);
  }
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const h16x8 a = *(const h16x8*)(big + ((threadIdx.x * 16 + it * 8192) % (144 * 1024 - 16) & ~15));
    const h16x8 b = *(const h16x8*)(big + ((threadIdx.x * 16 + it * 4096 + 512) % (144 * 1024 - 16) & ~15));
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// cnt[0..3]: lanes with a wrong LOW half by quarter wave; cnt[4..7]: wrong HIGH half; cnt[8]: lanes checked (x reps)
__global__ void __launch_bounds__(256) victim(unsigned* cnt, int reps, float seed) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  unsigned bad_lo = 0, bad_hi = 0;
  for (int r = 0; r < reps; ++r) {
    // the skinning tail: o0 = t0 x + t1 y + t2 z + t3 (low halves), o1 = t4 x + t5 y + t6 z + t7 (high halves), o2 scalar
    const float x = seed + 1e-3f * (gid & 1023) + r * 0.01f, y = 0.5f - 7e-4f * (gid & 511), z = -0.25f + 3e-4f * (gid & 255);
    const float t0 = 0.9f + 1e-4f * gid, t1 = -0.1f + 2e-5f * r, t2 = 0.2f, t3 = 0.01f * (gid & 7);
    const float t4 = 0.05f, t5 = 0.8f - 1e-5f * gid, t6 = -0.3f, t7 = 0.02f;
    const float t8 = 0.11f, t9 = 0.22f, t10 = 0.33f;
    float o0, o1, o2;
    asm volatile(
        "v_mov_b32 v100, %3\n\tv_mov_b32 v101, %4\n\t"          // v[100:101] = (x, y)
        "v_mov_b32 v102, %6\n\tv_mov_b32 v103, %11\n\t"         // v[102:103] = (t0, t5)  -> accumulator pair
        "v_mov_b32 v104, %7\n\tv_mov_b32 v105, %10\n\t"         // v[104:105] = (t1, t4)
        "v_mov_b32 v106, %8\n\tv_mov_b32 v107, %12\n\t"         // v[106:107] = (t2, t6)
        "v_mov_b32 v108, %9\n\tv_mov_b32 v109, %13\n\t"         // v[108:109] = (t3, t7)
        "v_mov_b32 v110, %5\n\t"                                 // z
        "v_mov_b32 v112, %14\n\tv_mov_b32 v113, %15\n\tv_mov_b32 v114, %16\n\t"   // t8, t9, t10
        "s_nop 4\n\t"
#if CHAIN == 0      // the skinning tail as hipcc emitted it
        "v_pk_mul_f32 v[102:103], v[102:103], v[100:101]\n\t"
        "v_mul_f32_e32 v117, v113, v101\n\t"                    // the scalar row, written into the HIGH register of pair v[116:117]
        "v_mov_b32_e32 v116, v110\n\t"
        "v_pk_fma_f32 v[102:103], v[104:105], v[100:101], v[102:103] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_fmac_f32_e32 v117, v112, v100\n\t"
        "v_pk_fma_f32 v[102:103], v[106:107], v[116:117], v[102:103] op_sel_hi:[1,0,1]\n\t"
        "v_fmac_f32_e32 v117, v114, v110\n\t"
        "v_pk_add_f32 v[102:103], v[108:109], v[102:103]\n\t"
#elif CHAIN == 1    // the four packed instructions back to back, the scalar row after them
        "v_mov_b32_e32 v116, v110\n\tv_mov_b32_e32 v117, v110\n\t"
        "v_pk_mul_f32 v[102:103], v[102:103], v[100:101]\n\t"
        "v_pk_fma_f32 v[102:103], v[104:105], v[100:101], v[102:103] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 v[102:103], v[106:107], v[116:117], v[102:103] op_sel_hi:[1,0,1]\n\t"
        "v_pk_add_f32 v[102:103], v[108:109], v[102:103]\n\t"
        "v_mul_f32_e32 v117, v113, v101\n\tv_fmac_f32_e32 v117, v112, v100\n\tv_fmac_f32_e32 v117, v114, v110\n\t"
#elif CHAIN == 2    // as 0, but the scalar row lives in a register that is NOT the high half of a packed source pair (v118)
        "v_pk_mul_f32 v[102:103], v[102:103], v[100:101]\n\t"
        "v_mul_f32_e32 v118, v113, v101\n\t"
        "v_mov_b32_e32 v116, v110\n\tv_mov_b32_e32 v117, v110\n\t"
        "v_pk_fma_f32 v[102:103], v[104:105], v[100:101], v[102:103] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_fmac_f32_e32 v118, v112, v100\n\t"
        "v_pk_fma_f32 v[102:103], v[106:107], v[116:117], v[102:103] op_sel_hi:[1,0,1]\n\t"
        "v_fmac_f32_e32 v118, v114, v110\n\t"
        "v_pk_add_f32 v[102:103], v[108:109], v[102:103]\n\t"
        "v_mov_b32_e32 v117, v118\n\t"
#elif CHAIN == 3    // as 0 with two wait states after every packed instruction
        "v_pk_mul_f32 v[102:103], v[102:103], v[100:101]\n\ts_nop 1\n\t"
        "v_mul_f32_e32 v117, v113, v101\n\t"
        "v_mov_b32_e32 v116, v110\n\t"
        "v_pk_fma_f32 v[102:103], v[104:105], v[100:101], v[102:103] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\ts_nop 1\n\t"
        "v_fmac_f32_e32 v117, v112, v100\n\t"
        "v_pk_fma_f32 v[102:103], v[106:107], v[116:117], v[102:103] op_sel_hi:[1,0,1]\n\ts_nop 1\n\t"
        "v_fmac_f32_e32 v117, v114, v110\n\t"
        "v_pk_add_f32 v[102:103], v[108:109], v[102:103]\n\t"
#elif CHAIN == 4    // only v_pk_mul_f32 and v_pk_add_f32 packed (no op_sel anywhere); the two FMAs as scalar instructions on the halves
        "v_mov_b32_e32 v116, v110\n\tv_mul_f32_e32 v117, v113, v101\n\tv_fmac_f32_e32 v117, v112, v100\n\tv_fmac_f32_e32 v117, v114, v110\n\t"
        "v_pk_mul_f32 v[102:103], v[102:103], v[100:101]\n\t"
        "v_fmac_f32_e32 v102, v104, v101\n\tv_fmac_f32_e32 v103, v105, v100\n\t"
        "v_fmac_f32_e32 v102, v106, v116\n\tv_fmac_f32_e32 v103, v107, v116\n\t"
        "v_pk_add_f32 v[102:103], v[108:109], v[102:103]\n\t"
#elif CHAIN == 5    // only the two v_pk_fma_f32 with op_sel modifiers packed; mul and add scalar
        "v_mov_b32_e32 v116, v110\n\tv_mul_f32_e32 v117, v113, v101\n\tv_fmac_f32_e32 v117, v112, v100\n\tv_fmac_f32_e32 v117, v114, v110\n\t"
        "v_mul_f32_e32 v102, v102, v100\n\tv_mul_f32_e32 v103, v103, v101\n\t"
        "v_pk_fma_f32 v[102:103], v[104:105], v[100:101], v[102:103] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 v[102:103], v[106:107], v[116:117], v[102:103] op_sel_hi:[1,0,1]\n\t"
        "v_add_f32_e32 v102, v108, v102\n\tv_add_f32_e32 v103, v109, v103\n\t"
#elif CHAIN == 6    // ONE packed instruction: v_pk_mul_f32; everything else scalar
        "v_mov_b32_e32 v116, v110\n\tv_mul_f32_e32 v117, v113, v101\n\tv_fmac_f32_e32 v117, v112, v100\n\tv_fmac_f32_e32 v117, v114, v110\n\t"
        "v_pk_mul_f32 v[102:103], v[102:103], v[100:101]\n\t"
        "v_fmac_f32_e32 v102, v104, v101\n\tv_fmac_f32_e32 v103, v105, v100\n\t"
        "v_fmac_f32_e32 v102, v106, v116\n\tv_fmac_f32_e32 v103, v107, v116\n\t"
        "v_add_f32_e32 v102, v108, v102\n\tv_add_f32_e32 v103, v109, v103\n\t"
#elif CHAIN == 7    // ONE packed instruction: v_pk_add_f32
        "v_mov_b32_e32 v116, v110\n\tv_mul_f32_e32 v117, v113, v101\n\tv_fmac_f32_e32 v117, v112, v100\n\tv_fmac_f32_e32 v117, v114, v110\n\t"
        "v_mul_f32_e32 v102, v102, v100\n\tv_mul_f32_e32 v103, v103, v101\n\t"
        "v_fmac_f32_e32 v102, v104, v101\n\tv_fmac_f32_e32 v103, v105, v100\n\t"
        "v_fmac_f32_e32 v102, v106, v116\n\tv_fmac_f32_e32 v103, v107, v116\n\t"
        "v_pk_add_f32 v[102:103], v[108:109], v[102:103]\n\t"
#elif CHAIN == 8    // ONE packed instruction: the v_pk_fma_f32 whose LOW result takes the HIGH half of src1 (op_sel:[0,1,0])
        "v_mov_b32_e32 v116, v110\n\tv_mul_f32_e32 v117, v113, v101\n\tv_fmac_f32_e32 v117, v112, v100\n\tv_fmac_f32_e32 v117, v114, v110\n\t"
        "v_mul_f32_e32 v102, v102, v100\n\tv_mul_f32_e32 v103, v103, v101\n\t"
        "v_pk_fma_f32 v[102:103], v[104:105], v[100:101], v[102:103] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_fmac_f32_e32 v102, v106, v116\n\tv_fmac_f32_e32 v103, v107, v116\n\t"
        "v_add_f32_e32 v102, v108, v102\n\tv_add_f32_e32 v103, v109, v103\n\t"
#elif CHAIN == 9    // ONE packed instruction: the v_pk_fma_f32 whose HIGH result takes the LOW half of src1 (op_sel_hi:[1,0,1])
        "v_mov_b32_e32 v116, v110\n\tv_mul_f32_e32 v117, v113, v101\n\tv_fmac_f32_e32 v117, v112, v100\n\tv_fmac_f32_e32 v117, v114, v110\n\t"
        "v_mul_f32_e32 v102, v102, v100\n\tv_mul_f32_e32 v103, v103, v101\n\t"
        "v_fmac_f32_e32 v102, v104, v101\n\tv_fmac_f32_e32 v103, v105, v100\n\t"
        "v_pk_fma_f32 v[102:103], v[106:107], v[116:117], v[102:103] op_sel_hi:[1,0,1]\n\t"
        "v_add_f32_e32 v102, v108, v102\n\tv_add_f32_e32 v103, v109, v103\n\t"
#endif
        "s_nop 4\n\t"
        "v_mov_b32 %0, v102\n\tv_mov_b32 %1, v103\n\tv_mov_b32 %2, v117\n\t"
        : "=v"(o0), "=v"(o1), "=v"(o2)
        : "v"(x), "v"(y), "v"(z), "v"(t0), "v"(t1), "v"(t2), "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7), "v"(t8), "v"(t9), "v"(t10)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v112", "v113", "v114", "v116", "v117", "v118");
    // the same arithmetic, one scalar instruction at a time (same operation order: mul, fma, fma, add)
    float e0, e1;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(t0), "v"(x));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e0) : "v"(t1), "v"(y));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e0) : "v"(t2), "v"(z));
    asm volatile("v_add_f32 %0, %1, %0" : "+v"(e0) : "v"(t3));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(t5), "v"(y));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e1) : "v"(t4), "v"(x));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(e1) : "v"(t6), "v"(z));
    asm volatile("v_add_f32 %0, %1, %0" : "+v"(e1) : "v"(t7));
    bad_lo += __float_as_uint(o0) != __float_as_uint(e0);
    bad_hi += __float_as_uint(o1) != __float_as_uint(e1);
    if (o2 == 123456.f) bad_lo += 1000;                          // (keeps the scalar row alive)
  }
  const int q = (threadIdx.x & 63) >> 4;
  if (bad_lo) atomicAdd(&cnt[q], bad_lo);
  if (bad_hi) atomicAdd(&cnt[4 + q], bad_hi);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20000;
  const int with_neighbour = argc > 2 ? atoi(argv[2]) : 1;
  unsigned* cnt; float* nb_out;
  (void)hipMalloc(&cnt, 64); (void)hipMemset(cnt, 0, 64); (void)hipMalloc(&nb_out, 4096);
  hipStream_t sv, sn;
  (void)hipStreamCreate(&sv); (void)hipStreamCreate(&sn);
  for (int l = 0; l < launches; ++l) {
    if (with_neighbour && l % 2 == 0) hipLaunchKernelGGL(neighbour, dim3(256), dim3(512), 0, sn, nb_out, 4000);
    hipLaunchKernelGGL(victim, dim3(540), dim3(256), 0, sv, cnt, 8, 0.1f + l * 1e-4f);
    if (l % 64 == 63) (void)hipStreamSynchronize(sv);
  }
  (void)hipDeviceSynchronize();
  unsigned h[16];
  (void)hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost);
  printf("CHAIN %d, neighbour %d, %d launches of 540 x 256 lanes x 8 chains: wrong LOW halves by quarter wave %u %u %u %u | wrong HIGH halves %u %u %u %u\n",
         CHAIN, with_neighbour, launches, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  return 0;
}
