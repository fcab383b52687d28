// Scaled-plane helpers (fp32 -> [K/16][R][16] hi / lo planes, per-matrix power-of-two scale) and the 128 x 288-tile product of mid-size batches.
//
// Scaled planes (since round 1; every large-batch kernel reads them): an operand matrix is stored as hi = fp16(v * p), lo = fp16(v * p - hi) with one
// power-of-two scale p per matrix (so that max |v * p| ~ 2^9..2^14 and the low halves stay normal fp16 numbers without the 2^11 factor of gemm_h3.hip);
// hi*hi + hi*lo + lo*hi then share ONE fp32 accumulator, C = acc / (pA * pW).  K-tile 16 (32-byte plane rows, slot swizzle (row >> 3) & 1).
// The kernels on this format: gemm_h3s_persist16c_kernel (gemm_h3s16c.hip: every plain product of large batches), gru_step16_kernel (gru_step16.hip: the
// fused GRU step), and gemm_h3s_kernel below (v_mfma_f32_32x32x16_f16, 12 waves of 32 x 96 = 128 x 288 tiles: the layer-0 projection of 512 <= B * T < 8192
// rows, whose 9 Hp columns are always a multiple of 288 -- cfg-B's 1024 rows x 9216 columns are exactly one round of 256 tiles).
// Removed in round 5 (measured slower than their successors; DESIGN_history.md): the 256 x 256 one-workgroup-per-tile form, the persistent 32x32x16 form
// (gemm_h3s_persist_kernel), the 32x32x16 fused GRU step, the 6-slot ring of the mid tile.
#include <type_traits>

#include "common.h"


namespace tepose {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16s(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vms() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// fp32 [rows][ld] (K valid columns) * p -> scaled planes of [R x Kp].  A wave converts 8 rows of one K-tile per unit
// (a lane two consecutive k of one row): it writes 256 contiguous bytes per plane; consecutive waves take
// consecutive K-tiles of the same rows, so a block's reads stay contiguous; 4 units in flight per wave.
__global__ void __launch_bounds__(256) split_planes16_kernel(const float* __restrict__ src, long ld, long rows, int K,
                                                             int Kp, long R, float p, _Float16* __restrict__ hi,
                                                             _Float16* __restrict__ lo) {
  typedef _Float16 h16x2s __attribute__((ext_vector_type(2)));
  const int KT = Kp / 16;
  const long units = ((rows + 7) / 8) * KT;
  const int lane = threadIdx.x & 63;
  constexpr int U = 4;
  for (long u0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * U; u0 < units; u0 += (long)gridDim.x * 4 * U) {
    float v0[U], v1[U];
    long o[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const long u = u0 + i;
      const long grp = u / KT;
      const int kt = (int)(u - grp * KT);
      const long row = 8 * grp + (lane >> 3);
      const int k = kt * 16 + 2 * (lane & 7);
      const bool ok = u < units && row < rows;
      v0[i] = (ok && k < K) ? src[row * ld + k] : 0.f;
      v1[i] = (ok && k + 1 < K) ? src[row * ld + k + 1] : 0.f;
      o[i] = ok ? plane16_index(row, k, R) : -1;
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if (o[i] < 0) continue;
      const float a0 = v0[i] * p, a1 = v1[i] * p;
      const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
      *(h16x2s*)(hi + o[i]) = h16x2s{h0, h1};
      *(h16x2s*)(lo + o[i]) = h16x2s{(_Float16)(a0 - (float)h0), (_Float16)(a1 - (float)h1)};
    }
  }
}

hipError_t launch_split_planes16(const float* src, long ld, long rows, int K, int Kp, long R, float p, void* hi,
                                 void* lo, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const long units = ((rows + 7) / 8) * (Kp / 16);
  const long want = (units + 15) / 16;
  const int blocks = (int)(want < 16384 ? want : 16384);
  hipLaunchKernelGGL(split_planes16_kernel, dim3(blocks), dim3(256), 0, s, src, ld, rows, K, Kp, R, p, (_Float16*)hi,
                     (_Float16*)lo);
  return hipGetLastError();
}

// max |v| of a buffer into *out (one float, zeroed by the caller): bit pattern of |v| as an unsigned atomic max
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ src, size_t n, unsigned* out) {
  float m = 0.f;
  // a NaN counts as infinite (fmaxf alone would drop it and a NaN weight would pass the fp16 range guard)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = fabsf(src[i]);
    m = fmaxf(m, v <= 3.402823466e38f ? v : __builtin_inff());
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

hipError_t launch_absmax(const float* src, size_t n, float* out, hipStream_t s) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float), s);
  if (e != hipSuccess || n == 0) return e;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, src, n, (unsigned*)out);
  return hipGetLastError();
}

__device__ __forceinline__ void h3s_tile_of_block(int bid, int nwg, int tilesM, int tilesN, int& tm, int& tn) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  constexpr int GM = 4;
  const int group = lin / (GM * tilesN), rem = lin - group * GM * tilesN;
  const int gm = min(GM, tilesM - group * GM);
  tm = group * GM + rem % gm;
  tn = rem / gm;
}

// 12 waves of 32 x 96 (WMF 1, WNT 3, NWM 4, NWN 3) = 128 x 288 block tile; K-tile 16, NST ring slots (NST - 1 stages requested ahead).
template <int WMF, int WNT, int NWM, int NWN, int NST = 4>
__global__ void __launch_bounds__(64 * NWM * NWN) gemm_h3s_kernel(H3SBatch batch, int tilesM, int tilesN) {
  constexpr int NW = NWM * NWN;                           // waves per block: 8, or 12 for the 128 x 288 tile of mid-size batches
  constexpr int HM = 32 * WMF * NWM, HN = 32 * WNT * NWN, HK = 16;
  static_assert(NW == 8 || NW == 12, "8 or 12 waves");
  const H3SArgs& a = batch.p[blockIdx.y];
  constexpr int RB = HK * 2;                             // 32 bytes per plane row of a stage
  constexpr int RPI = 1024 / RB;                         // 32 rows per DMA instruction
  constexpr int STAGE = (2 * HM + 2 * HN) * RB;          // 32 KB
  constexpr int TOT = STAGE / 1024;                      // DMA instructions per stage, dealt to the 8 waves:
  constexpr int Q = TOT / NW, REM = TOT % NW;            // waves < REM issue Q + 1 of them, the others Q
  constexpr int NDMA = Q + (REM ? 1 : 0);
  static_assert(STAGE % 1024 == 0 && NST * STAGE <= 160 * 1024, "ring fits the LDS");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  int tm, tn;
  h3s_tile_of_block(blockIdx.x, gridDim.x, tilesM, tilesN, tm, tn);
  const int m0 = tm * HM, n0 = tn * HN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int r = lane & 31, h = lane >> 5;

  const int nd = Q + (wave < REM ? 1 : 0);               // this wave's share
  const int i0 = wave * Q + min(wave, REM);              // its first instruction of a stage
  // wait until at most n whole stages of this wave's DMA are still in flight (counted vmcnt, literal per share)
  static_assert((NST - 2) * (Q + 1) <= 63, "vmcnt is 6 bits");
  auto wait_stages = [&](auto n) __attribute__((always_inline)) {
    constexpr int N = decltype(n)::value;
    if (REM && wave < REM) wait_vms<N * (Q + 1)>(); else wait_vms<N * Q>();
  };
  const char* gsrc[NDMA];
  long kst[NDMA];
#pragma unroll
  for (int q = 0; q < NDMA; ++q) {
    const int i = min(i0 + q, TOT - 1);
    int ri = i * RPI + lane / 2;                         // row of the stage image [A_hi | A_lo | W_hi | W_lo]
    const char* base;
    long grow, ks;
    if (ri < 2 * HM) {
      const bool lo = ri >= HM;
      base = (const char*)(lo ? a.Al : a.Ah);
      grow = min(m0 + (lo ? ri - HM : ri), a.M - 1);
      ks = a.a_kst * 2;
    } else {
      ri -= 2 * HM;
      const bool lo = ri >= HN;
      base = (const char*)(lo ? a.Wl : a.Wh);
      grow = n0 + (lo ? ri - HN : ri);
      ks = a.w_kst * 2;
    }
    gsrc[q] = base + grow * RB + 16 * (lane & 1);
    kst[q] = ((long)__builtin_amdgcn_readfirstlane((int)(ks >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)ks);
  }
  auto dma_part = [&](int stage, int q) {
    if (REM == 0 || q < nd) {
      glds16s(gsrc[q], lds + (stage % NST) * STAGE + (i0 + q) * 1024);
      gsrc[q] += kst[q];
    }
  };

  // fragment byte offsets inside a stage: row * 32 + 16 * (h ^ swz(row)); tile rows are multiples of 32, so the
  // swizzle bit is that of the lane's row
  const int sx = 16 * (h ^ ((r >> 3) & 1));
  int aoff[WMF], boff[WNT];
#pragma unroll
  for (int i = 0; i < WMF; ++i) aoff[i] = (wm * 32 * WMF + i * 32 + r) * RB + sx;
#pragma unroll
  for (int j = 0; j < WNT; ++j) boff[j] = 2 * HM * RB + (wn * 32 * WNT + j * 32 + r) * RB + sx;
  constexpr int A_LO = HM * RB, W_LO = HN * RB;

  f32x16 acc[WMF][WNT];
#pragma unroll
  for (int i = 0; i < WMF; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int KT = a.Kp / HK;
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < KT) {
#pragma unroll
      for (int q = 0; q < NDMA; ++q) dma_part(p, q);
    }
  auto ktile = [&](int kt, auto dma) __attribute__((always_inline)) {
    constexpr bool DMA = decltype(dma)::value;
    static_assert(NST == 4 || NST == 6, "waits written for a 4- or 6-slot ring");
    if constexpr (DMA) {
      wait_stages(std::integral_constant<int, NST - 2>{});
    } else {
      const int newer = min(NST - 2, KT - 1 - kt);           // tail: stages younger than this K-tile's still in flight
      if (NST == 6 && newer >= 4) wait_stages(std::integral_constant<int, NST == 6 ? 4 : 2>{});
      else if (NST == 6 && newer == 3) wait_stages(std::integral_constant<int, NST == 6 ? 3 : 2>{});
      else if (newer >= 2) wait_stages(std::integral_constant<int, 2>{});
      else if (newer == 1) wait_stages(std::integral_constant<int, 1>{});
      else wait_vms<0>();
    }
    __builtin_amdgcn_s_barrier();
    const char* st = lds + (kt % NST) * STAGE;
    h16x8 ah[WMF], al[WMF], bh[WNT], bl[WNT];
#pragma unroll
    for (int i = 0; i < WMF; ++i) {
      ah[i] = *(const h16x8*)(st + aoff[i]);
      al[i] = *(const h16x8*)(st + A_LO + aoff[i]);
    }
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      bh[j] = *(const h16x8*)(st + boff[j]);
      bl[j] = *(const h16x8*)(st + W_LO + boff[j]);
    }
    int q = 0;
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        const int t = i * WNT + j;
#pragma unroll
        for (; q < (t + 1) * NDMA / (WMF * WNT); ++q)
          if constexpr (DMA) dma_part(kt + NST - 1, q);
      }
    // the cross terms after all hi*hi products: consecutive MFMAs never share an accumulator
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
  };
  int kt = 0;
  for (; kt + NST - 1 < KT; ++kt) ktile(kt, std::true_type{});
  for (; kt < KT; ++kt) ktile(kt, std::false_type{});
  wait_vms<0>();

  {
    float bv[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      const int col = n0 + wn * 32 * WNT + j * 32 + r;
      bv[j] = (a.bias && col < a.N) ? a.bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 * WMF + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= a.M) continue;
        const float rs = a.row_scale ? a.row_scale[row] * a.inv_scale : a.inv_scale;   // per-row input scale (launch_split_rows)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          const int col = n0 + wn * 32 * WNT + j * 32 + r;
          if (col < a.N) a.C[(long)row * a.ldc + col] = acc[i][j][e] * rs + bv[j];
        }
      }
  }
}

hipError_t launch_gemm_h3s(const H3SArgs& a, hipStream_t s, const Options& o, int tag) {   // every plain scaled-plane product of large batches: the barrier-free persistent kernel
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  return launch_gemm_h3s16c(a, s, tag, o.s16_gm);
}

// Mid-size products (a few hundred to a few thousand rows) whose N is a multiple of 288 -- the stacked layer-0 block, 9 Hp
// columns: 128 x 288 tiles, 12 waves of 32 x 96 (three per SIMD).  B * T = 1024 rows x 9216 columns are exactly 256 tiles
// = one round of the chip, where 128 x 128 tiles of the two-accumulator kernel make 576 = 2.25 rounds.
bool gemm_h3s_mid_ok(const H3SArgs& a) { return a.N % 288 == 0 && a.Kp % 16 == 0 && a.M > 0; }
hipError_t launch_gemm_h3s_mid(const H3SArgs& a, hipStream_t s) {
  if (!gemm_h3s_mid_ok(a)) return hipErrorInvalidValue;
  const int tilesM = (a.M + 127) / 128, tilesN = a.N / 288;
  H3SBatch b{};
  b.p[0] = a; b.n = 1;
  hipLaunchKernelGGL((gemm_h3s_kernel<1, 3, 4, 3>), dim3(tilesM * tilesN, 1), dim3(768), 0, s, b, tilesM, tilesN);
  return hipGetLastError();
}

size_t gemm_h3s_ws_bytes(int M, int N, int K) {
  const size_t Kp = (size_t)round_up(K, 16);
  return 2 * align_up((size_t)M * Kp * 2, 256) + 2 * align_up((size_t)round_up(N, 256) * Kp * 2, 256) + 256;
}

// test / bench entry: fp32 A[M,K], W[N,K] (no bias) -> scaled planes in ws -> C; pA / pW: power-of-two operand scales
hipError_t launch_gemm_h3s_f32(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N,
                               int K, float pA, float pW, void* ws, hipStream_t s, const Options& o, const float* bias, int mid) {
  const int Kp = round_up(K, 16), Np = round_up(N, 256);
  char* p = (char*)ws;
  _Float16* Ah = (_Float16*)p; p += align_up((size_t)M * Kp * 2, 256);
  _Float16* Al = (_Float16*)p; p += align_up((size_t)M * Kp * 2, 256);
  _Float16* Wh = (_Float16*)p; p += align_up((size_t)Np * Kp * 2, 256);
  _Float16* Wl = (_Float16*)p;
  hipError_t e = hipMemsetAsync(Wh, 0, 2 * align_up((size_t)Np * Kp * 2, 256), s);
  if (e != hipSuccess) return e;
  if ((e = launch_split_planes16(A, lda, M, K, Kp, M, pA, Ah, Al, s)) != hipSuccess) return e;
  if ((e = launch_split_planes16(W, ldw, N, K, Kp, Np, pW, Wh, Wl, s)) != hipSuccess) return e;
  H3SArgs a{Ah, Al, (long)M * 16, Wh, Wl, (long)Np * 16, Kp, C, ldc, bias, 1.f / (pA * pW), M, N};
  if (mid) return launch_gemm_h3s_mid(a, s);
  return launch_gemm_h3s(a, s, o);
}

}  // namespace tepose
