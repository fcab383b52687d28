// The split-precision GEMM with ONE accumulator per tile and a 256 x 256 block tile.  Used for the layer-0 input
// projection of large batches (the one product whose operands -- input planes and the stacked layer-0 W_ih -- no
// other kernel shares), and through the test / bench entry tepose_gemm_h3_f32 with TEPOSE_H3S=1.
//
// Differences from gemm_h3.hip:
//  * scaled planes: an operand matrix is stored as hi = fp16(v * p), lo = fp16(v * p - hi) with one power-of-two
//    scale p per matrix (so that max |v * p| ~ 2^9..2^14 and the low halves stay normal fp16 numbers without the
//    2^11 factor of gemm_h3.hip); hi*hi + hi*lo + lo*hi then share one fp32 accumulator, C = acc / (pA * pW);
//  * that halves the accumulator registers, so a wave owns 64 x 128 (2 x 4 MFMA tiles, 128 VGPRs) and the block
//    256 x 256: 33 % fewer LDS-DMA bytes and 25 % fewer fragment reads per MFMA;
//  * K-tile 16 (32-byte plane rows, [K/16][R][16] blocked, slot swizzle (row >> 3) & 1), 32 KB stages, 4-slot ring
//    (3 stages in flight): 24 MFMAs per wave and barrier, as in gemm_h3.hip.
#include <type_traits>

#include "common.h"

#ifndef TEPOSE_H3S_ABL
#define TEPOSE_H3S_ABL 0   // timing-only ablations of gemm_h3s_persist_kernel (wrong results): 1 no LDS-DMA after a tile's first three
#endif                     // stages (the ring keeps real data: operand toggling, hence power, stays realistic), 4 fragment reads
                           // only in a tile's first K-tile, 8 no C stores
#ifndef TEPOSE_GRU_ABL
#define TEPOSE_GRU_ABL 0   // timing-only ablations of the fused GRU step (wrong results): 1 no LDS-DMA, 2 no epilogue loads,
#endif                     // 4 no epilogue stores, 8 no LDS turn, 16 no MFMA, 64 no epilogue at all, 256 no LDS-DMA after a tile's first
                           // three stages.  CAUTION (round 3): every one but 256
                           // of them also changes the DATA the matrix pipes see (stale LDS, constant states, planes never
                           // written), and these kernels are power-limited: a build whose operands stop toggling clocks 10-25 %
                           // higher, so "no X" overstates the cost of X.

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
                                                             _Float16* __restrict__ lo, int tiled_hp) {
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
      // tiled_hp: destination row q = jt * 192 + wn * 96 + gate * 32 + i of the gate-interleaved tile order reads the natural row gate * Hp + jt * 64 + wn * 32 + i
      const long srow = tiled_hp ? (long)((row % 96) / 32) * tiled_hp + (row / 192) * 64 + ((row % 192) / 96) * 32 + row % 32 : row;
      v0[i] = (ok && k < K) ? src[srow * ld + k] : 0.f;
      v1[i] = (ok && k + 1 < K) ? src[srow * ld + k + 1] : 0.f;
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
                                 void* lo, hipStream_t s, int tiled_hp) {
  if (rows <= 0) return hipSuccess;
  if (tiled_hp && (tiled_hp % 64 != 0 || rows != 3 * (long)tiled_hp)) return hipErrorInvalidValue;
  const long units = ((rows + 7) / 8) * (Kp / 16);
  const long want = (units + 15) / 16;
  const int blocks = (int)(want < 16384 ? want : 16384);
  hipLaunchKernelGGL(split_planes16_kernel, dim3(blocks), dim3(256), 0, s, src, ld, rows, K, Kp, R, p, (_Float16*)hi,
                     (_Float16*)lo, tiled_hp);
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
  constexpr int GM = 4;   // (2 / 8 / 16 row tiles per group measured on the fused GRU step in round 3: all within 1 %)
  const int group = lin / (GM * tilesN), rem = lin - group * GM * tilesN;
  const int gm = min(GM, tilesM - group * GM);
  tm = group * GM + rem % gm;
  tn = rem / gm;
}

__device__ __forceinline__ float gs_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float gs_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

// Plain product: 4 x 2 waves of 64 x 128 (WMF 2, WNT 4) = 256 x 256 block.  GRU step: 4 x 2 waves of 32 x 96 (the
// r, z, n tiles of 32 hidden units) = 128 rows x 64 hidden units x 3 gates, W_hh rows in the gate-interleaved tile
// order, cell update in the epilogue, new state out as fp32 and as scaled planes.
// NST: ring slots (NST - 1 stages requested ahead).  4 where a K-tile carries enough MFMA work to cover a stage's latency; the
// 128 x 288 tile of mid-size batches (12 waves x 9 MFMAs per K-tile = 0.36 us of matrix work against ~2.5 us from request to
// landing) looked bound by 3 stages / latency = 0.89 us per K-tile; with 6 slots (160 KB of LDS exactly, 5 stages ahead;
// TEPOSE_MID_RING6=1) it measured 1 % SLOWER (round 3), so request depth is not what limits that shape.
template <int WMF, int WNT, int NWM, int NWN, bool GRU, int NST = 4>
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
  auto dma_part = [&](int stage, int q, bool steady = false) {
    if (REM == 0 || q < nd) {
      // (ablation 1: no LDS-DMA; 256: none after the first NST - 1 stages, so that the ring keeps real operand data)
      if (!(GRU && ((TEPOSE_GRU_ABL & 1) || ((TEPOSE_GRU_ABL & 256) && steady)))) glds16s(gsrc[q], lds + (stage % NST) * STAGE + (i0 + q) * 1024);
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
        if constexpr (GRU && (TEPOSE_GRU_ABL & 16)) acc[i][j][0] += (float)ah[i][0] * (float)bh[j][0] + (float)al[i][0] * (float)bl[j][0];
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        const int t = i * WNT + j;
#pragma unroll
        for (; q < (t + 1) * NDMA / (WMF * WNT); ++q)
          if constexpr (DMA) dma_part(kt + NST - 1, q, true);
      }
    // the cross terms after all hi*hi products: consecutive MFMAs never share an accumulator
    if constexpr (!(GRU && (TEPOSE_GRU_ABL & 16))) {
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };
  int kt = 0;
  for (; kt + NST - 1 < KT; ++kt) ktile(kt, std::true_type{});
  for (; kt < KT; ++kt) ktile(kt, std::false_type{});
  wait_vms<0>();

  if constexpr (GRU) {
    static_assert(!GRU || WNT == 3, "r, z, n tiles");
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    typedef _Float16 h16x4s __attribute__((ext_vector_type(4)));
    const GateDir& d = batch.gate[blockIdx.y];
    const int Hp = batch.Hp;
    const int jb = tn * (32 * NWN) + wn * 32;              // first hidden unit of this wave's 32
    // The three products of a 32-row fragment go through the idle ring, so that a lane ends up with 4 consecutive
    // hidden units of one row: the cell operands come in as 16-byte loads, the new state leaves as one 16-byte
    // store and two 8-byte plane stores (a quarter of the memory instructions of the one-column-per-lane layout).
    __syncthreads();
    float* tile = (float*)lds + wave * 32 * 32;            // one gate's [row][32] block at a time (4 KB per wave)
    static_assert(NW * 32 * 32 * 4 <= NST * STAGE, "epilogue staging fits the ring");
    const bool vec = (((size_t)d.hout | (size_t)d.gi | (size_t)d.hprev) & 15) == 0 && (d.ldo & 3) == 0 &&
                     (d.ldgi & 3) == 0 && (d.ldh & 3) == 0;
    if constexpr ((TEPOSE_GRU_ABL & 64) != 0) {
      float sacc = 0.f;
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc += acc[0][g][e];
      if (sacc == 123.456f) d.hout[tid] = sacc;
      return;
    }
#pragma unroll
    for (int i = 0; i < WMF; ++i) {
      f32x4s hg[3][4];                                     // [gate][task]: 4 consecutive hidden units of one row
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        if constexpr ((TEPOSE_GRU_ABL & 8) != 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t) hg[g][t] = f32x4s{acc[i][g][4 * t], acc[i][g][4 * t + 1], acc[i][g][4 * t + 2], acc[i][g][4 * t + 3]} * a.inv_scale;
        } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          tile[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[i][g][e] * a.inv_scale;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int idx = t * 64 + lane;
          hg[g][t] = *(const f32x4s*)(tile + (idx >> 3) * 32 + (idx & 7) * 4);
        }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int idx = t * 64 + lane, rl = idx >> 3, c4 = idx & 7;
        const int row = m0 + wm * 32 * WMF + i * 32 + rl, j = jb + c4 * 4;
        if (row < a.M && j < Hp) {
          const f32x4s hr = hg[0][t], hz = hg[1][t], hn = hg[2][t];
          const float* gi = d.gi + (long)row * d.ldgi + j;
          const float* hq = d.hprev + (long)row * d.ldh + j;
          f32x4s gr, gz, gn, hp, br, bz, bn;
          if constexpr ((TEPOSE_GRU_ABL & 2) != 0) {
            const float cc = (float)(row + j) * 1e-6f;
            gr = f32x4s{cc, cc, cc, cc}; gz = gr * 0.5f; gn = gr * 0.25f; hp = gr * 2.f; br = gr; bz = gr; bn = gr;
          } else {
          if (vec) {
            gr = *(const f32x4s*)gi; gz = *(const f32x4s*)(gi + Hp); gn = *(const f32x4s*)(gi + 2 * Hp);
            hp = *(const f32x4s*)hq;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) { gr[c] = gi[c]; gz[c] = gi[Hp + c]; gn[c] = gi[2 * Hp + c]; hp[c] = hq[c]; }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) { br[c] = d.bhh[j + c]; bz[c] = d.bhh[Hp + j + c]; bn[c] = d.bhh[2 * Hp + j + c]; }
          }
          f32x4s v;
          _Float16 hh[4], ll[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float rg = gs_sigmoid(gr[c] + (hr[c] + br[c]));
            const float zg = gs_sigmoid(gz[c] + (hz[c] + bz[c]));
            const float ng = gs_tanh(gn[c] + rg * (hn[c] + bn[c]));
            v[c] = (1.f - zg) * ng + zg * hp[c];
            const float sv = v[c] * batch.state_scale;
            hh[c] = (_Float16)sv;
            ll[c] = (_Float16)(sv - (float)hh[c]);
          }
          if ((TEPOSE_GRU_ABL & 4) && v[0] + v[1] + v[2] + v[3] + (float)hh[0] + (float)ll[1] != 1234.5678f) continue;
          float* ho = d.hout + (long)row * d.ldo + j;
          const long o = (long)(j >> 4) * d.okst + plane16_index(row, j & 15, 0);
          if (vec) {
            *(f32x4s*)ho = v;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) ho[c] = v[c];
          }
          *(h16x4s*)((_Float16*)d.hout_hi + o) = h16x4s{hh[0], hh[1], hh[2], hh[3]};
          *(h16x4s*)((_Float16*)d.hout_lo + o) = h16x4s{ll[0], ll[1], ll[2], ll[3]};
        }
      }
    }
  } else {
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

// The plain product as a PERSISTENT kernel: 256 workgroups (one per CU) walk the 256 x 256 tiles b, b + 256, ...
// What it buys over one workgroup per tile: the C stores of a tile (256 KB) used to drain at the end of its workgroup,
// with the CU's matrix pipes idle until the next workgroup had refilled the ring (ablation: ~10 % of the projection).
// Here the next tile's first three stages are requested BEFORE the stores of the finished tile are issued, and the
// first three K-tiles of the next tile only wait for their own stage (vmcnt allows the 32 younger stores to stay in
// flight), so the stores drain under the next tile's MFMAs.  To get a countable, small number of stores the product is
// formed transposed (W fragments as the MFMA's row operand): a lane then owns 4 consecutive columns of one row of C,
// i.e. 32 16-byte stores per wave instead of 128 4-byte ones.  Same fragments, same K order, same accumulators.
// TAG only names the instantiation: 0 = the layer-0 input projection (the kernel bench.py's roofline is about), 1 = every other
// plain product, so that a rocprofv3 --stats table lists the dominant launches under a symbol of their own
template <int TAG>
__global__ void __launch_bounds__(512) gemm_h3s_persist_kernel(H3SArgs a, int tilesM, int tilesN) {
  constexpr int WMF = 2, WNT = 4, NWN = 2, NST = 4;
  constexpr int HM = 256, HN = 256, HK = 16, RB = HK * 2, RPI = 1024 / RB;
  constexpr int STAGE = (2 * HM + 2 * HN) * RB, TOT = STAGE / 1024, Q = TOT / 8;
  constexpr int NSTORE = WMF * WNT * 4;                   // 16-byte stores per wave and tile
  static_assert(TOT % 8 == 0 && NST * STAGE + HN * 4 <= 160 * 1024 && 2 * Q + NSTORE <= 63, "ring / vmcnt budget");
  typedef float f32x4p __attribute__((ext_vector_type(4)));
  // ONE __shared__ object (ring + the tile's bias row): with two, hipcc tags LDS accesses with alias scopes, starts tracking
  // the LDS-DMA requests per object and answers every fragment read with s_waitcnt vmcnt(0) -- no DMA stays in flight
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE + HN * 4];
  float* sbias = (float*)(lds + NST * STAGE);
  const int ntiles = tilesM * tilesN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int r = lane & 31, h = lane >> 5;
  const int i0 = wave * Q;

  // this lane's rows of the stage image [A_hi | A_lo | W_hi | W_lo] (the same for every tile)
  bool isA[Q], isLo[Q];
  int lrow[Q];
  long kst[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    int ri = (i0 + q) * RPI + lane / 2;
    isA[q] = ri < 2 * HM;
    if (!isA[q]) ri -= 2 * HM;
    isLo[q] = ri >= (isA[q] ? HM : HN);
    lrow[q] = isLo[q] ? ri - (isA[q] ? HM : HN) : ri;
    const long ks = (isA[q] ? a.a_kst : a.w_kst) * 2;
    kst[q] = ((long)__builtin_amdgcn_readfirstlane((int)(ks >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ks);
  }
  const char* gbase[Q];                                   // plane base of each row role (+ this lane's 16-byte half), once
#pragma unroll
  for (int q = 0; q < Q; ++q)
    gbase[q] = (const char*)(isA[q] ? (isLo[q] ? a.Al : a.Ah) : (isLo[q] ? a.Wl : a.Wh)) + 16 * (lane & 1);
  const char* gsrc[Q];
  int m0 = 0, n0 = 0;
  auto setup = [&](int tile) {
    int tm, tn;
    h3s_tile_of_block(tile, ntiles, tilesM, tilesN, tm, tn);
    m0 = tm * HM; n0 = tn * HN;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long grow = isA[q] ? min((long)m0 + lrow[q], (long)a.M - 1) : (long)n0 + lrow[q];
      gsrc[q] = gbase[q] + grow * RB;
    }
  };
  auto dma_part = [&](int stage, int q, bool steady = false) {
    if (!((TEPOSE_H3S_ABL & 1) && steady)) glds16s(gsrc[q], lds + (stage % NST) * STAGE + (i0 + q) * 1024);
    gsrc[q] += kst[q];
  };
  const int sx = 16 * (h ^ ((r >> 3) & 1));
  int aoff[WMF], boff[WNT];
#pragma unroll
  for (int i = 0; i < WMF; ++i) aoff[i] = (wm * 32 * WMF + i * 32 + r) * RB + sx;
#pragma unroll
  for (int j = 0; j < WNT; ++j) boff[j] = 2 * HM * RB + (wn * 32 * WNT + j * 32 + r) * RB + sx;
  constexpr int A_LO = HM * RB, W_LO = HN * RB;
  const int KT = a.Kp / HK;
  const bool overlap = KT >= 2 * NST;                     // the store-drain accounting below needs a few K-tiles
  h16x8 ah[WMF], al[WMF], bh[WNT], bl[WNT];
  const bool vec = (((size_t)a.C | (size_t)a.bias) & 15) == 0 && (a.ldc & 3) == 0;

  f32x16 acc[WMF][WNT];
  auto ktile = [&](int kt, auto dma, auto extra) __attribute__((always_inline)) {
    constexpr bool DMA = decltype(dma)::value;
    constexpr int EXTRA = decltype(extra)::value;         // younger stores of the previous tile that may stay in flight
    if constexpr (DMA) {
      wait_vms<2 * Q + EXTRA>();
    } else {
      const int newer = min(NST - 2, KT - 1 - kt);
      if (newer >= 2) wait_vms<2 * Q>();
      else if (newer == 1) wait_vms<Q>();
      else wait_vms<0>();
    }
    __builtin_amdgcn_s_barrier();
    const char* st = lds + (kt % NST) * STAGE;
    if (!(TEPOSE_H3S_ABL & 4) || kt == 0) {
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
    }
    int q = 0;
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        const int t = i * WNT + j;
#pragma unroll
        for (; q < (t + 1) * Q / (WMF * WNT); ++q)
          if constexpr (DMA) dma_part(kt + NST - 1, q, true);
      }
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  setup(tile);
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < KT) {
#pragma unroll
      for (int q = 0; q < Q; ++q) dma_part(p, q);
    }
  bool pending = false;                                   // NSTORE stores of the previous tile are younger than this tile's first stages
  for (;;) {
#pragma unroll
    for (int i = 0; i < WMF; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int kt = 0;
    if (pending) {
      for (; kt < NST - 1; ++kt) ktile(kt, std::true_type{}, std::integral_constant<int, NSTORE>{});
    }
    for (; kt + NST - 1 < KT; ++kt) ktile(kt, std::true_type{}, std::integral_constant<int, 0>{});
    for (; kt < KT; ++kt) ktile(kt, std::false_type{}, std::integral_constant<int, 0>{});
    wait_vms<0>();

    const int tm0 = m0, tn0 = n0;                          // the finished tile
    const int next = tile + (int)gridDim.x;
    const bool full = tm0 + HM <= a.M && tn0 + HN <= a.N && vec;
    // epilogue operands first: nothing may be loaded from global memory between the next tile's requests and the stores
    // (the tile's 256 bias values wait in LDS, outside the ring)
    float rs[WMF];
    if (full) {
      if (tid < 64)
        *(f32x4p*)(sbias + 4 * tid) = a.bias ? *(const f32x4p*)(a.bias + tn0 + 4 * tid) : f32x4p{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < WMF; ++i) {
        const int row = tm0 + wm * 32 * WMF + i * 32 + r;
        rs[i] = a.row_scale ? a.row_scale[row] * a.inv_scale : a.inv_scale;
      }
      wait_vms<0>();
    }
    const bool ov = full && overlap && next < ntiles;
    __syncthreads();                                       // every wave has read the last stages: the ring is free; sbias is written
    if (next < ntiles) {
      setup(next);
      if (ov) {
#pragma unroll
        for (int p = 0; p < NST - 1; ++p)
#pragma unroll
          for (int q = 0; q < Q; ++q) dma_part(p, q);
      }
    }
    if (full) {
      // the bias reads are asm: hipcc's wait-count pass answers a C++ LDS read behind LDS-DMA requests with
      // s_waitcnt vmcnt(0) in the next K-tiles, i.e. exactly the store drain this loop is built to avoid
      const unsigned sb = (unsigned)(size_t)sbias + (unsigned)(wn * 32 * WNT + 4 * h) * 4u;
      float* c0 = a.C + (long)(tm0 + wm * 32 * WMF + r) * a.ldc + tn0 + wn * 32 * WNT + 4 * h;
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        f32x4p bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[g]) : "v"(sb), "n"((j * 32 + 8 * g) * 4));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#pragma unroll
        for (int i = 0; i < WMF; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4p v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = acc[i][j][4 * g + c] * rs[i] + bq[g][c];
            if ((TEPOSE_H3S_ABL & 8) && v[0] + v[1] != 1234.5678f) continue;
            *(f32x4p*)(c0 + (long)i * 32 * a.ldc + j * 32 + 8 * g) = v;
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < WMF; ++i) {
        const int row = tm0 + wm * 32 * WMF + i * 32 + r;
        if (row >= a.M) continue;
        const float rsv = a.row_scale ? a.row_scale[row] * a.inv_scale : a.inv_scale;
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int col = tn0 + wn * 32 * WNT + j * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (col < a.N) a.C[(long)row * a.ldc + col] = acc[i][j][e] * rsv + (a.bias ? a.bias[col] : 0.f);
          }
      }
    }
    if (next >= ntiles) break;
    tile = next;
    if (!ov) {                                            // partial tile / short K: drain, then fill the ring as a first tile does
      wait_vms<0>();
#pragma unroll
      for (int p = 0; p < NST - 1; ++p)
        if (p < KT) {
#pragma unroll
          for (int q = 0; q < Q; ++q) dma_part(p, q);
        }
    }
    pending = ov;
  }
}

// (Round 3 had a persistent 256-row form of the fused GRU step here -- gru_h3s_persist_kernel, opt-in, bit-identical, 4 % slower at
// B = 8192 and up to 22 % slower at B = 2048 because one workgroup per CU keeps all CUs in lock step.  Removed in round 4: the
// default step is gru_h3s16_kernel<0, 2> of gemm_h3s16.hip.  What was learnt from it is in DESIGN.md section 12b.)
static bool h3s_persist() {
  static const bool v = [] { const char* e = getenv("TEPOSE_H3S_PERSIST"); return e ? atoi(e) != 0 : true; }();
  return v;
}

hipError_t launch_gemm_h3s(const H3SArgs& a, hipStream_t s, int tag) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  const int tilesM = (a.M + 255) / 256, tilesN = (a.N + 255) / 256;
  H3SBatch b{};
  b.p[0] = a; b.n = 1;
  if (h3s_persist() && a.shape16 == 3 && gemm_h3s16_ok(a)) return launch_gemm_h3s16c(a, s, tag);   // TEPOSE_MFMA16 bit 8: barrier-free (default)
  if (a.c_blk_hp) return hipErrorInvalidValue;              // only that kernel writes the blocked gate pre-activation layout
  if (h3s_persist() && a.shape16 && gemm_h3s16_ok(a)) return launch_gemm_h3s16(a, s, tag);
  if (h3s_persist()) {
    const int nt = tilesM * tilesN;
    if (tag == 0) hipLaunchKernelGGL(gemm_h3s_persist_kernel<0>, dim3(nt < 256 ? nt : 256), dim3(512), 0, s, a, tilesM, tilesN);
    else hipLaunchKernelGGL(gemm_h3s_persist_kernel<1>, dim3(nt < 256 ? nt : 256), dim3(512), 0, s, a, tilesM, tilesN);
  } else {
    hipLaunchKernelGGL((gemm_h3s_kernel<2, 4, 4, 2, false>), dim3(tilesM * tilesN, 1), dim3(512), 0, s, b, tilesM, tilesN);
  }
  return hipGetLastError();
}

// Mid-size products (a few hundred to a few thousand rows) whose N is a multiple of 288 -- the stacked layer-0 block, 9 Hp
// columns: 128 x 288 tiles, 12 waves of 32 x 96 (three per SIMD).  B * T = 1024 rows x 9216 columns are exactly 256 tiles
// = one round of the chip, where 128 x 128 tiles of the two-accumulator kernel make 576 = 2.25 rounds.
bool gemm_h3s_blocked_ok() { return h3s_persist(); }     // the blocked gate pre-activation layout needs the persistent 16x16x32 kernels
bool gemm_h3s_mid_ok(const H3SArgs& a) { return a.N % 288 == 0 && a.Kp % 16 == 0 && a.M > 0; }
hipError_t launch_gemm_h3s_mid(const H3SArgs& a, hipStream_t s) {
  if (!gemm_h3s_mid_ok(a)) return hipErrorInvalidValue;
  const int tilesM = (a.M + 127) / 128, tilesN = a.N / 288;
  H3SBatch b{};
  b.p[0] = a; b.n = 1;
  static const bool deep = [] { const char* e = getenv("TEPOSE_MID_RING6"); return e ? atoi(e) != 0 : false; }();   // A/B (round 3): 0.577 vs 0.570 ms per B = 64 forward -- the 4-slot ring stays
  if (deep) hipLaunchKernelGGL((gemm_h3s_kernel<1, 3, 4, 3, false, 6>), dim3(tilesM * tilesN, 1), dim3(768), 0, s, b, tilesM, tilesN);
  else hipLaunchKernelGGL((gemm_h3s_kernel<1, 3, 4, 3, false>), dim3(tilesM * tilesN, 1), dim3(768), 0, s, b, tilesM, tilesN);
  return hipGetLastError();
}

// GRU step of up to 3 directions: p[d] = {hprev planes, W_hh planes (gate-tiled rows, padded to 384), Kp = Hp,
// inv_scale}, gate[d] = cell operands / outputs (okst = halfs between 16-column groups of the output planes)
hipError_t launch_gru_h3s(const H3SBatch& b, hipStream_t s) {
  if (b.n <= 0 || b.p[0].M <= 0) return hipSuccess;
  // block = 128 rows x 64 hidden units x 3 gates (4 x 2 waves of 32 x 96): 100 VGPRs and an 80 KB ring, so two blocks
  // share a CU and cover each other's pipeline fill and store drain.  Measured against 128 x 128 units (148 VGPRs,
  // one block per CU): -1 % at B = 8192, and it keeps winning down to B ~ 2048; 256 rows x 64 units: +2.5 %.
  bool blk = false;
  for (int d = 0; d < b.n; ++d) blk = blk || b.gate[d].gi_blk != 0 || b.gate[d].hp_blk != 0 || b.gate[d].ho_blk != 0;
  if (b.p[0].shape16 == 4 && gru_h3s16c_ok(b) && !blk) return launch_gru_h3s16c(b, s);        // TEPOSE_MFMA16 bit 16
  if (b.p[0].shape16 && gru_h3s16_ok(b)) return launch_gru_h3s16(b, s);
  if (blk) return hipErrorInvalidValue;                     // only gru_h3s16_kernel reads the blocked gate pre-activation / state layouts
  const int tm = (b.p[0].M + 127) / 128, tj = (b.Hp + 63) / 64;
  hipLaunchKernelGGL((gemm_h3s_kernel<1, 3, 4, 2, true>), dim3(tm * tj, b.n), dim3(512), 0, s, b, tm, tj);
  return hipGetLastError();
}

size_t gemm_h3s_ws_bytes(int M, int N, int K) {
  const size_t Kp = (size_t)round_up(K, 16);
  return 2 * align_up((size_t)M * Kp * 2, 256) + 2 * align_up((size_t)round_up(N, 256) * Kp * 2, 256) + 256;
}

// test / bench entry: fp32 A[M,K], W[N,K] (no bias) -> scaled planes in ws -> C; pA / pW: power-of-two operand scales
hipError_t launch_gemm_h3s_f32(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N,
                               int K, float pA, float pW, void* ws, hipStream_t s, const float* bias, int shape16) {
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
  a.shape16 = shape16;
  return launch_gemm_h3s(a, s);
}

}  // namespace tepose
