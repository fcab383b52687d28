// Split-precision GEMM: fp32 operands carried as two fp16 halves (a = hi + lo, hi = fp16(a),
// lo = fp16(a - hi): 22 significant bits), three fp16 MFMAs per k-step (hi*hi + hi*lo + lo*hi,
// the lo*lo term is below 2^-22 relative), fp32 accumulation inside the matrix core.
//
// Why: gfx950's fp16 MFMA (v_mfma_f32_32x32x16_f16) runs at 16x the rate of the exact-fp32 MFMA,
// so three of them are still 5.3x faster, and fp16 x fp16 products are exact in the fp32
// accumulator.  Emulated through the whole L=2 / H=1024 / T=16 pipeline the vertices stay within
// 1.3e-6 of an fp64 run (exact-fp32 path: 4.8e-7), against a parity budget of 1e-4.
//
// Operands live in HBM as separate hi / lo planes of fp16 (same bytes as fp32), K-contiguous.
// Block tile 256 x 128, 8 waves as 4(M) x 2(N), wave tile 64 x 64 (2 x 2 MFMA tiles of 32x32),
// K-tile 32; LDS stage = [A_hi | A_lo | W_hi | W_lo] = 48 KB, double-buffered, filled by LDS-DMA.
// A tile row is 64 B = 4 slots of 16 B (8 halfs); slot index XOR ((row>>2)&3) on the DMA source and
// on the read keeps ds_read_b128 conflict-free with a lane-linear LDS image.
#include "common.h"

#include <stdlib.h>

namespace tepose {

#ifndef TEPOSE_H3_ABL
#define TEPOSE_H3_ABL 0   // timing-only diagnostic builds: 1 no DMA in the loop, 2 no barrier, 3 no fragment reads
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ void glds16b(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// fp32 [rows][ld] -> hi / lo fp16 planes [rows][Kp] (columns >= K zero)
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ src, long ld, long rows, int K,
                                                           int Kp, _Float16* __restrict__ hi,
                                                           _Float16* __restrict__ lo) {
  const long total = rows * Kp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / Kp;
    const int k = (int)(idx - r * Kp);
    const float a = k < K ? src[r * ld + k] : 0.f;
    split_hi_lo(a, hi[idx], lo[idx]);
  }
}

hipError_t launch_split_planes(const float* src, long ld, long rows, int K, int Kp, void* hi, void* lo,
                               hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const long total = rows * Kp;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, s, src, ld, rows, K, Kp, (_Float16*)hi,
                     (_Float16*)lo);
  return hipGetLastError();
}


__device__ __forceinline__ void h3_tile_of_block(int bid, int nwg, int tilesM, int tilesN, int& tm, int& tn) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  constexpr int GM = 4;                 // 4 x 256 rows share every W panel on an XCD
  const int gsz = GM * tilesN;
  const int g = lin / gsz, rem = lin - g * gsz;
  const int first_m = g * GM;
  const int gm = min(GM, tilesM - first_m);
  tm = first_m + rem % gm;
  tn = rem / gm;
}

template <int N>
__device__ __forceinline__ void wait_vm() {          // s_waitcnt vmcnt(N) with a literal count
  static_assert(N >= 0 && N <= 24, "vmcnt immediate");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the literal");
}

__device__ __forceinline__ float g_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float g_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

// WMF: 32-row MFMA fragments per wave along M (block rows = 128 * WMF); WNT: 32-column tiles per wave along
// N (block columns = 64 * WNT); HK: K-tile (32 | 16); NST: stages in the LDS ring (NST - 1 K-tiles of DMA in
// flight); GRU: the wave's 3 column tiles are the r, z, n gates of the same 32 hidden units and the epilogue is
// the GRU cell update (fp32 state + hi/lo planes out) instead of a plain store.
template <int WMF, int WNT, int HK, int NST, bool GRU>
__global__ void __launch_bounds__(512, 2) gemm_h3_kernel(H3Batch batch, int tilesM, int tilesN) {
  constexpr int HM = 128 * WMF;
  constexpr int HN = 64 * WNT;
  static_assert(!GRU || WNT == 3, "GRU epilogue needs the three gate tiles in one wave");
  constexpr int RB = HK * 2;                          // bytes per tile row of one plane
  constexpr int SL = RB / 16;                         // 16-byte slots per row
  constexpr int RPB = 256 / RB;                       // rows per 256-byte LDS bank row
  constexpr int RPI = 1024 / RB;                      // rows moved by one wave-wide DMA instruction
  constexpr int STAGE = (2 * HM + 2 * HN) * RB;
  constexpr int NDMA = STAGE / 1024 / 8;              // DMA instructions per wave per stage
  constexpr int KS = HK / 16;                         // 16-deep MFMA steps per stage
  static_assert(STAGE % 8192 == 0 && NST * STAGE <= 160 * 1024, "stage geometry");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const H3Args& a = batch.p[blockIdx.y];
  int tm, tn;
  h3_tile_of_block(blockIdx.x, gridDim.x, tilesM, tilesN, tm, tn);
  const int m0 = tm * HM, n0 = tn * HN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // ---- DMA: stage image = [A_hi | A_lo | W_hi | W_lo], instruction i moves RPI consecutive rows of it ----
  const int lrow = lane / SL, lslot = lane % SL;
  const char* gsrc[NDMA];
#pragma unroll
  for (int q = 0; q < NDMA; ++q) {
    const int i = wave * NDMA + q;
    int ri = i * RPI + lrow;                          // row index inside the stage image
    const char* base;
    long ldb;
    int row, grow;
    if (ri < 2 * HM) {
      const bool lo = ri >= HM;
      row = lo ? ri - HM : ri;
      grow = min(m0 + row, a.M - 1);
      base = (const char*)(lo ? a.Al : a.Ah);
      ldb = a.lda * 2;
    } else {
      ri -= 2 * HM;
      const bool lo = ri >= HN;
      row = lo ? ri - HN : ri;
      grow = n0 + row;
      base = (const char*)(lo ? a.Wl : a.Wh);
      ldb = (long)a.Kp * 2;
    }
    gsrc[q] = base + (long)grow * ldb + 16 * (lslot ^ ((row / RPB) % SL));
  }
  auto issue = [&](int kt, int buf) {
    char* st = lds + buf * STAGE;
#pragma unroll
    for (int q = 0; q < NDMA; ++q) glds16b(gsrc[q] + (long)kt * RB, st + (wave * NDMA + q) * 1024);
  };

  // ---- fragment offsets (bytes inside a stage) ---------------------------------------------------------
  const int sw = (r / RPB) % SL;
  int aoff[WMF], boff[WNT];
#pragma unroll
  for (int i = 0; i < WMF; ++i) aoff[i] = (wm * 32 * WMF + i * 32 + r) * RB;
#pragma unroll
  for (int j = 0; j < WNT; ++j) boff[j] = 2 * HM * RB + (wn * 32 * WNT + j * 32 + r) * RB;
  constexpr int A_LO = HM * RB, W_LO = HN * RB;

  f32x16 acc[WMF][WNT], accx[WMF][WNT];     // hi*hi sums, and the cross terms (scaled by kLoScale)
#pragma unroll
  for (int i = 0; i < WMF; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accx[i][j][e] = 0.f; }

  struct Frags { h16x8 ah[WMF], al[WMF], bh[WNT], bl[WNT]; };
  auto load_frags = [&](const char* st, int s, Frags& f) {
    const int sx = 16 * (((2 * s + h) % SL) ^ sw);
#pragma unroll
    for (int i = 0; i < WMF; ++i) {
      f.ah[i] = *(const h16x8*)(st + aoff[i] + sx);
      f.al[i] = *(const h16x8*)(st + A_LO + aoff[i] + sx);
    }
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      f.bh[j] = *(const h16x8*)(st + boff[j] + sx);
      f.bl[j] = *(const h16x8*)(st + W_LO + boff[j] + sx);
    }
  };
  // Ring of NST stages, NST-1 K-tiles of LDS-DMA in flight.  Per K-tile: every wave waits until its own DMA
  // instructions of stage kt have landed (counted vmcnt: the newer stages' instructions may stay
  // outstanding), the raw barrier then makes the whole stage visible and also proves that every wave is
  // done reading stage kt-1, whose slot the next DMA overwrites.  __syncthreads() would drain vmcnt(0).
  const int KT = a.Kp / HK;
#if TEPOSE_H3_ABL == 3
  Frags f[KS];
#endif
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < KT) issue(p, p);
  for (int kt = 0; kt < KT; ++kt) {
    const int newer = min(NST - 2, KT - 1 - kt);     // stages issued after kt that may still be in flight
    if (newer >= 2) wait_vm<(NST >= 4 ? 2 : 0) * NDMA>();
    else if (newer == 1) wait_vm<NDMA>();
    else wait_vm<0>();
#if TEPOSE_H3_ABL != 2
    __builtin_amdgcn_s_barrier();
#endif
    const char* st = lds + (kt % NST) * STAGE;
    // Fragments first, then the MFMAs with the next stage's DMA instructions spread between them: all 8
    // waves leave the barrier together, so DMA issued up front would keep every matrix pipe idle meanwhile.
    const bool more = kt + NST - 1 < KT;
    char* dst = lds + ((kt + NST - 1) % NST) * STAGE;
    const long koff = (long)(kt + NST - 1) * RB;
#if TEPOSE_H3_ABL != 3
    Frags f[KS];
#endif
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#if TEPOSE_H3_ABL == 3
      if (kt == 0)
#endif
      load_frags(st, ks, f[ks]);
    }
    int q = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[ks].ah[i], f[ks].bh[j], acc[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[ks].ah[i], f[ks].bl[j], accx[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[ks].al[i], f[ks].bh[j], accx[i][j], 0, 0, 0);
          if (q < NDMA) {
#if TEPOSE_H3_ABL != 1
            if (more) glds16b(gsrc[q] + koff, dst + (wave * NDMA + q) * 1024);
#endif
            ++q;
          }
        }
    static_assert(NDMA <= KS * WMF * WNT, "one DMA per MFMA triple");
  }
  wait_vm<0>();

  if constexpr (GRU) {
    // columns of this wave: gates r, z, n of hidden units j = tn*64 + wn*32 + r (ROW_GATES_TILED order)
    const GateDir& d = batch.gate[blockIdx.y];
    const int Hp = batch.Hp;
    const int j = tn * 64 + wn * 32 + r;
    const float br = d.bhh[j], bz = d.bhh[Hp + j], bn = d.bhh[2 * Hp + j];
#pragma unroll
    for (int i = 0; i < WMF; ++i) {
      const int rbase = m0 + wm * 32 * WMF + i * 32 + 4 * h;
      float gr[16], gz[16], gn[16], hp[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = min(rbase + (e & 3) + 8 * (e >> 2), a.M - 1);
        const float* gi = d.gi + (long)row * d.ldgi + j;
        gr[e] = gi[0]; gz[e] = gi[Hp]; gn[e] = gi[2 * Hp];
        hp[e] = d.hprev[(long)row * d.ldh + j];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        const float hr = acc[i][0][e] + accx[i][0][e] * (1.f / kLoScale);
        const float hz = acc[i][1][e] + accx[i][1][e] * (1.f / kLoScale);
        const float hn = acc[i][2][e] + accx[i][2][e] * (1.f / kLoScale);
        const float rg = g_sigmoid(gr[e] + (hr + br));
        const float zg = g_sigmoid(gz[e] + (hz + bz));
        const float ng = g_tanh(gn[e] + rg * (hn + bn));
        const float hv = (1.f - zg) * ng + zg * hp[e];
        if (row < a.M) {
          const long o = (long)row * d.ldo + j;
          d.hout[o] = hv;
          split_hi_lo(hv, d.hout_hi[o], d.hout_lo[o]);
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = n0 + wn * 32 * WNT + j * 32 + r;
      if (col >= a.N) continue;
      const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < WMF; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 32 * WMF + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < a.M) a.C[(long)row * a.ldc + col] = (acc[i][j][e] + accx[i][j][e] * (1.f / kLoScale)) + bv;
        }
      }
    }
  }
}

// test / bench entry: fp32 A[M,K], W[N,K] -> planes in `ws` -> C (K multiple of 32)
hipError_t launch_gemm_h3_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C,
                              long ldc, int M, int N, int K, void* ws, hipStream_t s) {
  const int Np = round_up(N, 256);     // the widest tile variant reads 256-row W panels
  char* p = (char*)ws;
  _Float16* Ah = (_Float16*)p; p += align_up((size_t)M * K * 2, 256);
  _Float16* Al = (_Float16*)p; p += align_up((size_t)M * K * 2, 256);
  _Float16* Wh = (_Float16*)p; p += align_up((size_t)Np * K * 2, 256);
  _Float16* Wl = (_Float16*)p;
  hipError_t e = hipMemsetAsync(Wh, 0, 2 * align_up((size_t)Np * K * 2, 256), s);
  if (e != hipSuccess) return e;
  if ((e = launch_split_planes(A, lda, M, K, K, Ah, Al, s)) != hipSuccess) return e;
  if ((e = launch_split_planes(W, ldw, N, K, K, Wh, Wl, s)) != hipSuccess) return e;
  H3Batch b{};
  b.p[0] = H3Args{Ah, Al, (long)K, Wh, Wl, K, C, ldc, bias, M, N};
  b.n = 1;
  return launch_gemm_h3(b, s);
}

// up to 3 independent products of the same M, N, Kp in one launch (the directions of a GRU step)
hipError_t launch_gemm_h3(const H3Batch& b, hipStream_t s) {
  if (b.n <= 0 || b.p[0].M <= 0) return hipSuccess;
  static const int variant = [] {
    const char* e = getenv("TEPOSE_H3_VARIANT");
    return e ? atoi(e) : 0;
  }();
  const int tilesM = (b.p[0].M + 255) / 256;
  if (variant == 2) {            // 256 x 128 tile, K-tile 16, 4-stage ring
    const int tilesN = (b.p[0].N + 127) / 128;
    hipLaunchKernelGGL((gemm_h3_kernel<2, 2, 16, 4, false>), dim3(tilesM * tilesN, b.n), dim3(512), 0, s, b, tilesM,
                       tilesN);
  } else {                       // 256 x 128 tile, K-tile 32, 3-stage ring
    const int tilesN = (b.p[0].N + 127) / 128;
    hipLaunchKernelGGL((gemm_h3_kernel<2, 2, 32, 3, false>), dim3(tilesM * tilesN, b.n), dim3(512), 0, s, b, tilesM,
                       tilesN);
  }
  return hipGetLastError();
}

hipError_t launch_gru_h3(const H3Batch& b, hipStream_t s) {
  if (b.n <= 0 || b.p[0].M <= 0) return hipSuccess;
  const int tilesM = (b.p[0].M + 127) / 128, tilesJ = b.Hp / 64;   // block = 128 rows x (64 hidden units x 3 gates)
  hipLaunchKernelGGL((gemm_h3_kernel<1, 3, 32, 3, true>), dim3(tilesM * tilesJ, b.n), dim3(512), 0, s, b, tilesM,
                     tilesJ);
  return hipGetLastError();
}

// x[rows][2133] fp32 -> hi / lo planes [rows][2144] (pad columns zero): the split kernel's A operand
__global__ void __launch_bounds__(256) pad_input_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ hi,
                                                               _Float16* __restrict__ lo, long rows) {
  const long total = rows * kInputP;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / kInputP;
    const int k = (int)(idx - row * kInputP);
    const float a = k < kInput ? x[row * kInput + k] : 0.f;
    split_hi_lo(a, hi[idx], lo[idx]);
  }
}

hipError_t launch_pad_input_planes(const float* x, void* hi, void* lo, long rows, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const long total = rows * kInputP;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(pad_input_planes_kernel, dim3(blocks), dim3(256), 0, s, x, (_Float16*)hi, (_Float16*)lo, rows);
  return hipGetLastError();
}

// GRU gate update after the recurrent product gh = h W_hh^T (natural gate order [r | z | n], no bias):
// thread = (row, hidden unit); writes the new state as fp32 and as hi / lo planes for the next product.
__global__ void __launch_bounds__(256) gru_gates_kernel(GateBatch gb, int M, int Hp, int first) {
  const GateDir& d = gb.d[blockIdx.y];
  const long total = (long)M * Hp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / Hp;
    const int j = (int)(idx - row * Hp);
    const float* gi = d.gi + row * d.ldgi + j;
    float hr = d.bhh[j], hz = d.bhh[Hp + j], hn = d.bhh[2 * Hp + j], hp = 0.f;
    if (!first) {
      const float* gh = d.gh + row * 3 * Hp + j;
      hr += gh[0]; hz += gh[Hp]; hn += gh[2 * Hp];
      hp = d.hprev[row * d.ldh + j];
    }
    const float rg = g_sigmoid(gi[0] + hr), zg = g_sigmoid(gi[Hp] + hz);
    const float ng = g_tanh(gi[2 * Hp] + rg * hn);
    const float hv = (1.f - zg) * ng + zg * hp;
    d.hout[row * d.ldo + j] = hv;
    split_hi_lo(hv, d.hout_hi[row * d.ldo + j], d.hout_lo[row * d.ldo + j]);
  }
}

hipError_t launch_gru_gates(const GateBatch& gb, int ndir, int M, int Hp, int first, hipStream_t s) {
  if (M <= 0 || ndir <= 0) return hipSuccess;
  const long total = (long)M * Hp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gru_gates_kernel, dim3(blocks, ndir), dim3(256), 0, s, gb, M, Hp, first);
  return hipGetLastError();
}

size_t gemm_h3_ws_bytes(int M, int N, int K) {
  return 2 * align_up((size_t)M * K * 2, 256) + 2 * align_up((size_t)round_up(N, 256) * K * 2, 256) + 256;
}

}  // namespace tepose
