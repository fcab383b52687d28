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
#include <type_traits>

#include "common.h"

#include <stdlib.h>

namespace tepose {

#ifndef TEPOSE_GRU_PF
#define TEPOSE_GRU_PF 1    // 0: fetch the GRU cell operands in the epilogue instead of during the last K-tiles (A/B)
#endif
#ifndef TEPOSE_H3_ABL
#define TEPOSE_H3_ABL 0   // timing-only diagnostic builds, bit mask: 1 no DMA in the loop, 2 no barrier, 4 no fragment reads, 8 no GRU epilogue stores, 16 no plain epilogue stores, 32 every stage re-reads K-tile 0 (L2 hits)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4v __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));


__device__ __forceinline__ void glds16b(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// fp32 [rows][ld] -> blocked hi / lo fp16 planes of [R x Kp] (columns >= K zero).  A wave converts four rows of
// one K-tile per unit, a lane two consecutive k of one row: it reads 2 floats and writes one 4-byte pair per plane
// (2-byte stores cost as much per instruction and move half as much); consecutive waves take consecutive K-tiles of
// the same rows, so a block's reads stay contiguous.
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ src, long ld, long rows, int K,
                                                           int Kp, long R, _Float16* __restrict__ hi,
                                                           _Float16* __restrict__ lo, int relu) {
  typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
  const int KT = Kp / kPlaneK;
  const long units = ((rows + 3) / 4) * KT;                 // (4-row group, K-tile)
  const int lane = threadIdx.x & 63;
  constexpr int U = 4;                                      // units per wave and trip: 2U independent loads in flight
  for (long u0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * U; u0 < units; u0 += (long)gridDim.x * 4 * U) {
    float v0[U], v1[U];
    long o[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const long u = u0 + i;
      const long grp = u / KT;
      const int kt = (int)(u - grp * KT);
      const long row = 4 * grp + (lane >> 4);
      const int k = kt * kPlaneK + 2 * (lane & 15);
      const bool ok = u < units && row < rows;
      v0[i] = (ok && k < K) ? src[row * ld + k] : 0.f;
      v1[i] = (ok && k + 1 < K) ? src[row * ld + k + 1] : 0.f;
      o[i] = ok ? plane_index(row, k, R) : -1;
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if (o[i] < 0) continue;
      const float a0 = relu ? fmaxf(v0[i], 0.f) : v0[i], a1 = relu ? fmaxf(v1[i], 0.f) : v1[i];
      half_t h0, l0, h1, l1;
      split_hi_lo(a0, h0, l0);
      split_hi_lo(a1, h1, l1);
      *(h16x2v*)(hi + o[i]) = h16x2v{h0, h1};
      *(h16x2v*)(lo + o[i]) = h16x2v{l0, l1};
    }
  }
}

hipError_t launch_split_planes(const float* src, long ld, long rows, int K, int Kp, long R, void* hi, void* lo,
                               hipStream_t s, int relu) {
  if (rows <= 0) return hipSuccess;
  const long units = ((rows + 3) / 4) * (Kp / kPlaneK);
  const long want = (units + 15) / 16;                      // 4 waves x 4 units per block and trip
  const int blocks = (int)(want < 16384 ? want : 16384);
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, s, src, ld, rows, K, Kp, R, (_Float16*)hi,
                     (_Float16*)lo, relu);
  return hipGetLastError();
}

// A block converts 8 consecutive rows.  Phase 1: wave w reads rows 2w, 2w+1 whole (NP pairs per lane and row, coalesced),
// takes each row's largest magnitude, picks its power-of-two scale and parks the scaled row in LDS.  Phase 2: the block
// walks the K-tiles; one wave instruction writes the 8 rows x 16 k (scaled format) or 4 rows x 32 k (blocked format) of a
// K-tile = 256 contiguous bytes per plane (a row-at-a-time conversion writes 32-byte pieces: 1.0 vs 0.4 ms at cfg-C).
constexpr int kRowsLd = 2148;        // LDS row stride (floats): 2144 + 4 keeps the 8 rows of a tile off each other's banks
template <int NP, bool F16>
__global__ void __launch_bounds__(256) split_rows_kernel(const float* __restrict__ src, long ld, long rows, int K, int Kp,
                                                         long R, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                         float* __restrict__ row_scale, uint4* __restrict__ zero,
                                                         long zero_n, int permT, RowPairSrc pr) {
  typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float buf[8 * kRowsLd];
  // the forward's arrival counters / granules (first kernel of a forward: later kernels see them cleared)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long)gridDim.x * 256) zero[i] = uint4{0u, 0u, 0u, 0u};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = (long)blockIdx.x * 8;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int lr = 2 * wave + rr;
    const long row = row0 + lr;
    if (row >= rows) continue;                                // wave-uniform
    // permT = T: the source is [B][T] windows x frames, the planes (and row_scale) are FRAME-major: plane row t * B + b = source row b * T + t
    const long srow = permT ? (row % (rows / permT)) * permT + row / (rows / permT) : row;
    // (pr.f0: the rows are gathered -- features | theta of one frame for rows < pr.B, features | zeros of another frame from there on: RowPairSrc)
    const bool second = pr.f0 && row >= pr.B;
    const float* x = pr.f0 ? (second ? pr.f1 + (row - pr.B) * pr.fld : pr.f0 + row * pr.fld) : src + srow * ld;
    const float* xt = pr.f0 && !second ? pr.th0 + row * pr.thld - kFeat : nullptr;       // indexed by k >= kFeat
    const int kf = pr.f0 ? kFeat : K;
    float v0[NP], v1[NP];
    float m = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int k = 2 * (lane + 64 * i);
      v0[i] = k < kf ? x[k] : (xt && k < K ? xt[k] : 0.f);
      v1[i] = k + 1 < kf ? x[k + 1] : (xt && k + 1 < K ? xt[k + 1] : 0.f);
      m = fmaxf(m, fmaxf(fabsf(v0[i]), fabsf(v1[i])));        // fmaxf drops NaN: tracked separately
      bad |= !(fabsf(v0[i]) <= 3.0e38f) || !(fabsf(v1[i]) <= 3.0e38f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    bad = __any(bad);
    float sc = 1.f, inv = 1.f;
    if (m > 0.f && !bad) {
      int ex;
      (void)frexpf(m, &ex);                                   // m = f * 2^ex, f in [0.5, 1)
      int e = 14 - ex;                                        // m * 2^e in [2^13, 2^14)
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
      sc = ldexpf(1.f, e);
      inv = ldexpf(1.f, -e);
    }
    if (lane == 0) row_scale[row] = inv;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int k = 2 * (lane + 64 * i);
      if (k < Kp) *(float2*)(buf + lr * kRowsLd + k) = float2{v0[i] * sc, v1[i] * sc};
    }
  }
  __syncthreads();
  if (F16) {
    const int lr = lane >> 3, kp = lane & 7;                  // 8 rows x 8 pairs = one 16-wide K-tile
    const long row = row0 + lr;
    for (int kt = wave; kt < Kp / 16; kt += 4) {
      if (row >= rows) continue;
      const int k = kt * 16 + 2 * kp;
      const float2 a = *(const float2*)(buf + lr * kRowsLd + k);
      const _Float16 h0 = (_Float16)a.x, h1 = (_Float16)a.y;
      const long o = plane16_index(row, k, R);
      *(h16x2v*)(hi + o) = h16x2v{h0, h1};
      *(h16x2v*)(lo + o) = h16x2v{(_Float16)(a.x - (float)h0), (_Float16)(a.y - (float)h1)};
    }
  } else {
    const int lr4 = lane >> 4, kp = lane & 15;                // 4 rows x 16 pairs = half of a 32-wide K-tile's 8 rows
    for (int u = wave; u < 2 * (Kp / 32); u += 4) {
      const int kt = u >> 1, lr = (u & 1) * 4 + lr4;
      const long row = row0 + lr;
      if (row >= rows) continue;
      const int k = kt * 32 + 2 * kp;
      const float2 a = *(const float2*)(buf + lr * kRowsLd + k);
      half_t h0, l0, h1, l1;
      split_hi_lo(a.x, h0, l0);
      split_hi_lo(a.y, h1, l1);
      const long o = plane_index(row, k, R);
      *(h16x2v*)(hi + o) = h16x2v{h0, h1};
      *(h16x2v*)(lo + o) = h16x2v{l0, l1};
    }
  }
}

// The same conversion for a handful of rows (one window, a few clips): one workgroup per row, a thread per 8 consecutive k
// (= one 16-byte run of either layout), the row's maximum through one LDS round.  The 8-rows-per-block kernel above is
// built for bandwidth; on 16 rows it is two workgroups walking 67 K-tiles each (11.6 us of a 0.2 ms forward).
template <bool F16>
__global__ void __launch_bounds__(256) split_rows_few_kernel(const float* __restrict__ src, long ld, long rows, int K, int Kp,
                                                             long R, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                             float* __restrict__ row_scale, uint4* __restrict__ zero,
                                                             long zero_n, RowPairSrc pr) {
  typedef _Float16 h16x8v __attribute__((ext_vector_type(8)));
  __shared__ float wmax[4];
  __shared__ int wbad[4];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long)gridDim.x * 256) zero[i] = uint4{0u, 0u, 0u, 0u};
  const long row = blockIdx.x;
  const bool second = pr.f0 && row >= pr.B;
  const float* x = pr.f0 ? (second ? pr.f1 + (row - pr.B) * pr.fld : pr.f0 + row * pr.fld) : src + row * ld;
  const float* xt = pr.f0 && !second ? pr.th0 + row * pr.thld - kFeat : nullptr;
  const int kf = pr.f0 ? kFeat : K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NO = 2;                                        // octets per thread: Kp <= 4096
  float v[NO][8];
  float m = 0.f;
  bool bad = false;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const int k0 = 8 * ((int)threadIdx.x + 256 * o);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[o][i] = (k0 + i < kf) ? x[k0 + i] : (xt && k0 + i < K ? xt[k0 + i] : 0.f);
      m = fmaxf(m, fabsf(v[o][i]));
      bad |= !(fabsf(v[o][i]) <= 3.0e38f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  bad = __any(bad);
  if (lane == 0) { wmax[wave] = m; wbad[wave] = bad ? 1 : 0; }
  __syncthreads();
  m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  bad = (wbad[0] | wbad[1] | wbad[2] | wbad[3]) != 0;
  float sc = 1.f, inv = 1.f;
  if (m > 0.f && !bad) {
    int ex;
    (void)frexpf(m, &ex);
    int e = 14 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    sc = ldexpf(1.f, e);
    inv = ldexpf(1.f, -e);
  }
  if (threadIdx.x == 0) row_scale[row] = inv;
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const int k0 = 8 * ((int)threadIdx.x + 256 * o);
    if (k0 >= Kp) continue;
    h16x8v h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a = v[o][i] * sc;
      if (F16) {
        const _Float16 hh = (_Float16)a;
        h[i] = hh; l[i] = (_Float16)(a - (float)hh);
      } else {
        half_t hh, ll;
        split_hi_lo(a, hh, ll);
        h[i] = hh; l[i] = ll;
      }
    }
    const long off = F16 ? plane16_index(row, k0, R) : plane_index(row, k0, R);
    *(h16x8v*)(hi + off) = h;
    *(h16x8v*)(lo + off) = l;
  }
}

hipError_t launch_split_rows(const float* src, long ld, long rows, int K, int Kp, long R, int fmt16, void* hi, void* lo,
                             float* row_scale, hipStream_t s, const Options& o, void* zero, size_t zero_bytes, int permT, const RowPairSrc* pair) {
  // (one workgroup per row up to Options::split_few_max_rows rows; the 8-rows-per-block kernel is for bandwidth, from a few thousand rows)
  if (rows <= 0) return hipSuccess;
  const RowPairSrc pr = pair ? *pair : RowPairSrc{};
  if (pair && (permT || !pr.f0 || !pr.f1 || !pr.th0 || K <= kFeat)) return hipErrorInvalidValue;
  if (permT && rows % permT != 0) return hipErrorInvalidValue;
  if (zero_bytes % 16 != 0) return hipErrorInvalidValue;
  if (Kp > 64 * 2 * 17 || Kp > kRowsLd || (Kp & 31)) return hipErrorInvalidValue;        // the [., 2144] input rows
  if (split_rows_few_ok(rows, Kp, permT, o)) {      // (the frame-major permutation lives in the 8-rows-per-block kernel: whatever the threshold says)
    if (fmt16)
      hipLaunchKernelGGL((split_rows_few_kernel<true>), dim3((unsigned)rows), dim3(256), 0, s, src, ld, rows, K, Kp, R,
                         (_Float16*)hi, (_Float16*)lo, row_scale, (uint4*)zero, (long)(zero_bytes / 16), pr);
    else
      hipLaunchKernelGGL((split_rows_few_kernel<false>), dim3((unsigned)rows), dim3(256), 0, s, src, ld, rows, K, Kp, R,
                         (_Float16*)hi, (_Float16*)lo, row_scale, (uint4*)zero, (long)(zero_bytes / 16), pr);
    return hipGetLastError();
  }
  const dim3 grid((unsigned)((rows + 7) / 8));
  if (fmt16)
    hipLaunchKernelGGL((split_rows_kernel<17, true>), grid, dim3(256), 0, s, src, ld, rows, K, Kp, R, (_Float16*)hi,
                       (_Float16*)lo, row_scale, (uint4*)zero, (long)(zero_bytes / 16), permT, pr);
  else
    hipLaunchKernelGGL((split_rows_kernel<17, false>), grid, dim3(256), 0, s, src, ld, rows, K, Kp, R, (_Float16*)hi,
                       (_Float16*)lo, row_scale, (uint4*)zero, (long)(zero_bytes / 16), permT, pr);
  return hipGetLastError();
}

hipError_t launch_pad_input_planes(const float* x, void* hi, void* lo, long rows, hipStream_t s) {
  return launch_split_planes(x, kInput, rows, kInput, kInputP, rows, hi, lo, s);
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
  static_assert(N >= 0 && N <= 63, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `newer` whole stages (NDMA instructions each, per wave) are still in flight
template <int NDMA, int MAXNEWER>
__device__ __forceinline__ void wait_stages(int newer) {
  static_assert(MAXNEWER * NDMA <= 63 && MAXNEWER <= 5, "ring depth");
  if (MAXNEWER >= 5 && newer >= 5) wait_vm<(MAXNEWER >= 5 ? 5 : 0) * NDMA>();
  else if (MAXNEWER >= 4 && newer == 4) wait_vm<(MAXNEWER >= 4 ? 4 : 0) * NDMA>();
  else if (MAXNEWER >= 3 && newer == 3) wait_vm<(MAXNEWER >= 3 ? 3 : 0) * NDMA>();
  else if (MAXNEWER >= 2 && newer == 2) wait_vm<(MAXNEWER >= 2 ? 2 : 0) * NDMA>();
  else if (newer == 1) wait_vm<NDMA>();
  else wait_vm<0>();
}

__device__ __forceinline__ float g_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float g_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

// WMF: 32-row MFMA fragments per wave along M, NWM: waves along M (block rows = 32 * WMF * NWM; 2 waves along N); WNT: 32-column tiles per wave along
// N (block columns = 64 * WNT); NST: stages in the LDS ring (NST - 1 K-tiles of DMA in flight); ROT: barrier in the middle of the K-tile (see the main loop); GRU: the wave's 3 column tiles are the r, z, n gates of the same 32 hidden units and the epilogue is
// the GRU cell update (fp32 state + hi/lo planes out) instead of a plain store.
template <int WMF, int WNT, int NST, bool GRU, bool ROT, int NWM = 4>
__global__ void __launch_bounds__(128 * NWM) gemm_h3_kernel(H3Batch batch, int tilesM, int tilesN) {
  constexpr int HK = kPlaneK;                         // K-tile = the planes' block width
  constexpr int NW = 2 * NWM;                         // waves: NWM along M x 2 along N
  constexpr int HM = 32 * WMF * NWM;
  constexpr int HN = 64 * WNT;
  static_assert(!GRU || WNT == 3, "GRU epilogue needs the three gate tiles in one wave");
  constexpr int RB = HK * 2;                          // bytes per tile row of one plane
  constexpr int SL = RB / 16;                         // 16-byte slots per row
  constexpr int RPB = 256 / RB;                       // rows per 256-byte LDS bank row
  constexpr int RPI = 1024 / RB;                      // rows moved by one wave-wide DMA instruction
  constexpr int STAGE = (2 * HM + 2 * HN) * RB;
  constexpr int NDMA = STAGE / 1024 / NW;             // DMA instructions per wave per stage
  constexpr int KS = HK / 16;                         // 16-deep MFMA steps per stage
  static_assert(STAGE % (1024 * NW) == 0 && NST * STAGE <= 160 * 1024, "stage geometry");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const H3Args& a = batch.p[blockIdx.y];
  int tm, tn;
  h3_tile_of_block(blockIdx.x, gridDim.x, tilesM, tilesN, tm, tn);
  const int m0 = tm * HM, n0 = tn * HN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // ---- DMA: stage image = [A_hi | A_lo | W_hi | W_lo], instruction i moves RPI = 16 consecutive rows of it: one
  // contiguous KiB of a blocked plane, lane l taking bytes [16 l, 16 l + 16).  The slot swizzle the fragment
  // reads rely on is in the stored data (plane_index), not in these addresses.
  const char* gsrc[NDMA];
  long kst[NDMA];                                     // bytes between K-tiles of that source
#pragma unroll
  for (int q = 0; q < NDMA; ++q) {
    const int i = wave * NDMA + q;
    int ri = i * RPI + lane / SL;                     // row index inside the stage image
    const char* base;
    long grow;
    if (ri < 2 * HM) {
      const bool lo = ri >= HM;
      base = (const char*)(lo ? a.Al : a.Ah);
      grow = min(m0 + (lo ? ri - HM : ri), a.M - 1);  // rows past M repeat the last row; never stored
      kst[q] = a.a_kst * 2;
    } else {
      ri -= 2 * HM;
      const bool lo = ri >= HN;
      base = (const char*)(lo ? a.Wl : a.Wh);
      grow = n0 + (lo ? ri - HN : ri);
      kst[q] = a.w_kst * 2;
    }
    gsrc[q] = base + grow * RB + 16 * (lane % SL);
    // wave-uniform (an instruction's 16 rows belong to one plane): keep the stride in scalar registers
    kst[q] = ((long)__builtin_amdgcn_readfirstlane((int)(kst[q] >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)kst[q]);
#if TEPOSE_H3_ABL & 32
    kst[q] = 0;
#endif
  }
  // every DMA instruction q is issued once per stage, in stage order: gsrc[q] walks along K by itself
  auto dma_part = [&](int stage, int q) {
#if !(TEPOSE_H3_ABL & 1)
    glds16b(gsrc[q], lds + (stage % NST) * STAGE + (wave * NDMA + q) * 1024);
#endif
    gsrc[q] += kst[q];
  };
  auto issue = [&](int stage) {
#pragma unroll
    for (int q = 0; q < NDMA; ++q) dma_part(stage, q);
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
  const int KT = a.Kp / HK;
  // GRU: cell operands of this wave's 32 rows x 32 hidden units, fetched during the last K-tiles so that their
  // HBM latency hides under MFMAs instead of standing between the last product and the cell update
  float pf_gr[16], pf_gz[16], pf_gn[16], pf_hp[16], pf_b[3];
  const int gj = tn * 64 + wn * 32 + r;       // GRU: hidden unit of this lane's columns (ROW_GATES_TILED order)
  if constexpr (ROT) {
    // ---- main loop ---------------------------------------------------------------------------------------
    // Ring of NST = 3 stages; stage s lives in slot s % 3.  The stage barrier sits in the MIDDLE of a K-tile:
    //
    //   K-tile kt, first half : read the ks=1 fragments of stage kt; MFMAs of ks=0 (fragments loaded during the
    //                           previous K-tile); the last NB DMA instructions of stage kt+2
    //   lgkmcnt(0); vmcnt     : own reads of stage kt done; own DMA of stage kt+1 landed (stage kt+2 may fly)
    //   s_barrier             : => stage kt+1 complete and visible, and nobody reads stage kt any more
    //   K-tile kt, second half: read the ks=0 fragments of stage kt+1; MFMAs of ks=1; the first NA DMA
    //                           instructions of stage kt+3 (into the slot stage kt just left)
    //
    // so the fragment reads of a half always run under the other half's MFMAs (with the barrier at the top of the
    // K-tile, all 8 waves waited for their first ds_read_b128s together after every barrier), two K-tiles of
    // DMA stay in flight, and the steady state is one basic block (tail K-tiles are a second instantiation).
    static_assert(!ROT || (NST == 3 && KS == 2), "the rotated pipeline is written for a 3-slot ring and two k-steps per stage");
    constexpr int TPH = WMF * WNT;                      // MFMA triples per half K-tile
    constexpr int NB = NDMA / 2, NA = NDMA - NB;        // DMA instructions issued in the first / second half
    static_assert(NA <= 2 * TPH && NB <= 2 * TPH, "at most two DMA instructions per MFMA triple");

    auto half = [&](const Frags& f, int stage, int q0, int nq, bool dma) {
#pragma unroll
      for (int i = 0; i < WMF; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], accx[i][j], 0, 0, 0);
          accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], accx[i][j], 0, 0, 0);
          const int t = i * WNT + j;                         // DMA instructions [t*nq/TPH, (t+1)*nq/TPH) go here
          if (dma) {
#pragma unroll
            for (int q = t * nq / TPH; q < (t + 1) * nq / TPH; ++q) dma_part(stage, q0 + q);
          }
        }
    };

    // prologue: stages 0, 1 and the first part of stage 2; then stage 0 visible, its ks=0 fragments in flight
    issue(0);
    if (KT > 1) issue(1);
    if (KT > 2) {
#pragma unroll
      for (int q = 0; q < NA; ++q) dma_part(2, q);
    }
    if (KT > 2) wait_vm<NDMA + NA>();
    else if (KT > 1) wait_vm<NDMA>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    Frags f0, f1;
    load_frags(lds, 0, f0);
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0) on the loop's entry edge too (see the end of ktile)

    auto ktile = [&](int kt, auto steady_t) {
      constexpr bool STEADY = decltype(steady_t)::value;        // stage kt+3 exists
      const char* st = lds + (kt % NST) * STAGE;
      const bool has1 = STEADY || kt + 1 < KT, has2 = STEADY || kt + 2 < KT;
#if !(TEPOSE_H3_ABL & 4)
      load_frags(st, 1, f1);
#endif
      __builtin_amdgcn_sched_barrier(0);
      half(f0, kt + 2, NA, NB, has2);
      __builtin_amdgcn_sched_barrier(0);
      if (has1) {
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): this wave's reads of stage kt are done
        if (has2) wait_vm<NDMA>();
        else wait_vm<0>();
#if !(TEPOSE_H3_ABL & 2)
        __builtin_amdgcn_s_barrier();
#endif
#if !(TEPOSE_H3_ABL & 4)
        load_frags(lds + ((kt + 1) % NST) * STAGE, 0, f0);
#endif
      }
      if constexpr (GRU && !STEADY) {
        if (TEPOSE_GRU_PF && kt == (KT >= 2 ? KT - 2 : 0)) {     // youngest memory operations from here on: no DMA follows
          const GateDir& d = batch.gate[blockIdx.y];
          const int Hp = batch.Hp;
          pf_b[0] = d.bhh[gj]; pf_b[1] = d.bhh[Hp + gj]; pf_b[2] = d.bhh[2 * Hp + gj];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = min(m0 + wm * 32 + 4 * h + (e & 3) + 8 * (e >> 2), a.M - 1);
            const float* gi = d.gi + (long)row * d.ldgi + gj;
            pf_gr[e] = gi[0]; pf_gz[e] = gi[Hp]; pf_gn[e] = gi[2 * Hp];
            pf_hp[e] = d.hprev[(long)row * d.ldh + gj];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      half(f1, kt + 3, 0, NA, STEADY);
      __builtin_amdgcn_sched_barrier(0);
      // the ks=0 fragments of the next stage landed long ago; saying so keeps the compiler from putting a
      // conservative lgkmcnt(0) between the next half's reads and its first MFMA (loop-header merge)
      __builtin_amdgcn_s_waitcnt(0xC07F);
    };
    int kt = 0;
    for (; kt + 3 < KT; ++kt) ktile(kt, std::true_type{});
    for (; kt < KT; ++kt) ktile(kt, std::false_type{});
  } else {
    // Ring of NST stages, NST-1 K-tiles of LDS-DMA in flight.  Per K-tile: every wave waits until its own DMA
    // instructions of stage kt have landed (counted vmcnt: the newer stages' instructions may stay
    // outstanding), the raw barrier then makes the whole stage visible and also proves that every wave is
    // done reading stage kt-1, whose slot the next DMA overwrites.  __syncthreads() would drain vmcnt(0).
#if TEPOSE_H3_ABL & 4
    Frags f[KS];
#endif
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
      if (p < KT) issue(p);
    // GRU: the cell operands of this wave's 32 rows x 32 hidden units (gate pre-activations, previous state,
    // b_hh) are fetched two K-tiles before the end of the loop, so that their HBM latency hides under MFMAs
    // instead of standing between the last product and the cell update.
    const int pf_kt = KT >= NST - 1 ? KT - (NST - 1) : 0;      // first K-tile of the tail (no DMA issued after it)
    // One K-tile.  DMA = std::true_type in the steady state (the stage NST-1 ahead exists: its DMA instructions
    // are issued unconditionally, one per MFMA triple, and exactly NST-2 newer stages may stay in flight at the
    // wait), std::false_type in the last NST-1 K-tiles.  Two instantiations instead of a per-instruction
    // `if (more)` keep the K-tile one basic block, so the compiler can schedule reads, MFMAs and DMA freely.
    auto ktile = [&](int kt, auto dma) {
      constexpr bool DMA = decltype(dma)::value;
      if constexpr (DMA) wait_vm<(NST - 2) * NDMA>();
      else wait_stages<NDMA, NST - 2>(min(NST - 2, KT - 1 - kt));
#if !(TEPOSE_H3_ABL & 2)
      __builtin_amdgcn_s_barrier();
#endif
      const char* st = lds + (kt % NST) * STAGE;
      if constexpr (GRU && !DMA) {
        if (kt == pf_kt) {
          const GateDir& d = batch.gate[blockIdx.y];
          const int Hp = batch.Hp;
          pf_b[0] = d.bhh[gj]; pf_b[1] = d.bhh[Hp + gj]; pf_b[2] = d.bhh[2 * Hp + gj];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = min(m0 + wm * 32 + 4 * h + (e & 3) + 8 * (e >> 2), a.M - 1);
            const float* gi = d.gi + (long)row * d.ldgi + gj;
            pf_gr[e] = gi[0]; pf_gz[e] = gi[Hp]; pf_gn[e] = gi[2 * Hp];
            pf_hp[e] = d.hprev[(long)row * d.ldh + gj];
          }
        }
      }
      // Fragments first, then the MFMAs with the next stage's DMA instructions spread between them: all 8
      // waves leave the barrier together, so DMA issued up front would keep every matrix pipe idle meanwhile.
#if !(TEPOSE_H3_ABL & 4)
      Frags f[KS];
#endif
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#if TEPOSE_H3_ABL & 4
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
            constexpr int NT = KS * WMF * WNT;       // MFMA triples per K-tile; DMA instructions spread over them
            const int t = (ks * WMF + i) * WNT + j;
#pragma unroll
            for (; q < (t + 1) * NDMA / NT; ++q)
              if constexpr (DMA) dma_part(kt + NST - 1, q);
          }
    };
    int kt = 0;
    for (; kt + NST - 1 < KT; ++kt) ktile(kt, std::true_type{});
    for (; kt < KT; ++kt) ktile(kt, std::false_type{});
  }
  wait_vm<0>();

  if constexpr (GRU) {
    static_assert(!GRU || WMF == 1, "GRU epilogue prefetch holds one 32-row fragment of cell operands");
    const GateDir& d = batch.gate[blockIdx.y];
    const int rbase = m0 + wm * 32 + 4 * h;
    if (!TEPOSE_GRU_PF) {
      const int Hp = batch.Hp;
      pf_b[0] = d.bhh[gj]; pf_b[1] = d.bhh[Hp + gj]; pf_b[2] = d.bhh[2 * Hp + gj];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = min(rbase + (e & 3) + 8 * (e >> 2), a.M - 1);
        const float* gi = d.gi + (long)row * d.ldgi + gj;
        pf_gr[e] = gi[0]; pf_gz[e] = gi[Hp]; pf_gn[e] = gi[2 * Hp];
        pf_hp[e] = d.hprev[(long)row * d.ldh + gj];
      }
    }
    __syncthreads();                                       // every wave is done with the last stage: LDS is free
    float* gtile = (float*)lds + wave * 32 * 32;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = rbase + (e & 3) + 8 * (e >> 2);
      const float hr = acc[0][0][e] + accx[0][0][e] * (1.f / kLoScale);
      const float hz = acc[0][1][e] + accx[0][1][e] * (1.f / kLoScale);
      const float hn = acc[0][2][e] + accx[0][2][e] * (1.f / kLoScale);
      const float rg = g_sigmoid(pf_gr[e] + (hr + pf_b[0]));
      const float zg = g_sigmoid(pf_gz[e] + (hz + pf_b[1]));
      const float ng = g_tanh(pf_gn[e] + rg * (hn + pf_b[2]));
      const float hv = (1.f - zg) * ng + zg * pf_hp[e];
      gtile[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = hv;
      (void)row;
    }
    // the wave's 32 x 32 block of new states, turned through LDS so that a lane stores 4 consecutive hidden
    // units: 16-byte fp32 stores and 8-byte plane stores instead of 4- and 2-byte ones (12 store instructions per
    // wave instead of 48; see the plain epilogue below)
    const bool vec = ((size_t)d.hout & 15) == 0 && (d.ldo & 3) == 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * 64 + lane, rl = idx >> 3, c4 = idx & 7;
      const int row = m0 + wm * 32 + rl, j = (gj & ~31) + c4 * 4;
#if TEPOSE_H3_ABL & 8
      if (acc[0][0][0] == 12345.678f)
#endif
      if (row < a.M) {
        const f32x4v v = *(const f32x4v*)(gtile + rl * 32 + c4 * 4);
        float* hp = d.hout + (long)row * d.ldo + j;
        if (vec) {
          *(f32x4v*)hp = v;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) hp[c] = v[c];
        }
        const long o = (long)(j >> 5) * d.okst + plane_index(row, j & 31, 0);
        half_t hh[4], ll[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) split_hi_lo(v[c], hh[c], ll[c]);
        *(h16x4v*)(d.hout_hi + o) = h16x4v{hh[0], hh[1], hh[2], hh[3]};
        *(h16x4v*)(d.hout_lo + o) = h16x4v{ll[0], ll[1], ll[2], ll[3]};
      }
    }
  } else {
    // Plain epilogue.  The accumulator layout gives a lane one column and 16 rows, i.e. 4-byte stores; the tile is
    // turned through the (now idle) LDS ring instead so that a lane owns 4 consecutive columns and the C / plane
    // stores are 16 / 8 bytes wide: a quarter of the store instructions, and the wave's implicit wait for its
    // stores at s_endpgm (nothing else can run on the CU meanwhile: one workgroup fills the LDS) shrinks with them.
    constexpr int TC = 32 * WNT;                           // this wave's sub-tile: 32*WMF rows x TC columns,
    constexpr int EP = (NW * 32 * WMF * TC * 4 + NST * STAGE - 1) / (NST * STAGE);   // staged in EP passes of
    constexpr int FP = WMF / EP, TR = 32 * FP;             // FP 32-row fragments each
    static_assert(WMF % EP == 0 && NW * TR * TC * 4 <= NST * STAGE, "epilogue staging fits the ring");
    const float sc = a.scale != 0.f ? a.scale : 1.f;
    __syncthreads();                                       // every wave is done with the last stage
    float* tile = (float*)lds + wave * TR * TC;
    const bool vec = (((size_t)a.C | (size_t)(a.addend ? a.addend : a.C)) & 15) == 0 && (a.ldc & 3) == 0 &&
                     (!a.addend || (a.ldadd & 3) == 0);
    constexpr int C4 = TC / 4;                             // float4 groups per sub-tile row
#pragma unroll
    for (int pass = 0; pass < EP; ++pass) {
#pragma unroll
      for (int i = 0; i < FP; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            tile[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * TC + j * 32 + r] =
                acc[pass * FP + i][j][e] + accx[pass * FP + i][j][e] * (1.f / kLoScale);
#pragma unroll
      for (int t = 0; t < TR * C4 / 64; ++t) {
        const int idx = t * 64 + lane, rl = idx / C4, c4 = idx % C4;
        const int row = m0 + wm * 32 * WMF + pass * TR + rl, col = n0 + wn * TC + c4 * 4;
#if TEPOSE_H3_ABL & 16
        if (acc[0][0][0] == 12345.678f)
#endif
        if (row < a.M && col < a.N) {
          f32x4v v = *(const f32x4v*)(tile + rl * TC + c4 * 4);
          if (a.row_scale) v *= a.row_scale[row];
          const int nv = min(4, a.N - col);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (a.bias && c < nv) v[c] += a.bias[col + c];
          float* cp = a.C + (long)row * a.ldc + col;
          if (vec && nv == 4) {
            if (a.addend) {
              const f32x4v ad = *(const f32x4v*)(a.addend + (long)row * a.ldadd + col);
              v += ad;
            }
            v *= sc;
            *(f32x4v*)cp = v;
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < nv) {
                if (a.addend) v[c] += a.addend[(long)row * a.ldadd + col + c];
                v[c] *= sc;
                cp[c] = v[c];
              }
          }
          if (a.Chi) {                                     // 4 consecutive columns share an 8-column slot of the plane
            const long o = (long)(col >> 5) * a.c_kst + plane_index(row, col & 31, 0);
            half_t hh[4], ll[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) split_hi_lo(c < nv ? v[c] : 0.f, hh[c], ll[c]);
            if (nv == 4) {
              *(h16x4v*)(a.Chi + o) = h16x4v{hh[0], hh[1], hh[2], hh[3]};
              *(h16x4v*)(a.Clo + o) = h16x4v{ll[0], ll[1], ll[2], ll[3]};
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (c < nv) { a.Chi[o + c] = hh[c]; a.Clo[o + c] = ll[c]; }
            }
          }
        }
      }
    }
  }
}

// test / bench entry: fp32 A[M,K], W[N,K] -> planes in `ws` -> C (K multiple of 32)
hipError_t launch_gemm_h3_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C,
                              long ldc, int M, int N, int K, void* ws, hipStream_t s, const Options& o, int kind) {
  const int Np = round_up(N, 128);
  const long Ra = M, Rw = Np;
  char* p = (char*)ws;
  _Float16* Ah = (_Float16*)p; p += align_up((size_t)Ra * K * 2, 256);
  _Float16* Al = (_Float16*)p; p += align_up((size_t)Ra * K * 2, 256);
  _Float16* Wh = (_Float16*)p; p += align_up((size_t)Rw * K * 2, 256);
  _Float16* Wl = (_Float16*)p;
  hipError_t e = hipMemsetAsync(Wh, 0, 2 * align_up((size_t)Rw * K * 2, 256), s);
  if (e != hipSuccess) return e;
  if ((e = launch_split_planes(A, lda, M, K, K, Ra, Ah, Al, s)) != hipSuccess) return e;
  if ((e = launch_split_planes(W, ldw, N, K, K, Rw, Wh, Wl, s)) != hipSuccess) return e;
  H3Batch b{};
  b.p[0] = H3Args{(const half_t*)Ah, (const half_t*)Al, Ra * 32, (const half_t*)Wh, (const half_t*)Wl, Rw * 32, K,
                  C, ldc, bias, M, N};
  b.n = 1;
  if (kind) return launch_skinny_gemm_h3(b.p[0], s, o);      // test / bench entry: the width-first kernel at any M
  return launch_gemm_h3(b, s, o);
}

// 256 x 128 block tile, 3-stage ring of 48 KB stages
hipError_t launch_gemm_h3(const H3Batch& b, hipStream_t s, const Options& o) {
  if (b.n <= 0 || b.p[0].M <= 0) return hipSuccess;
  const int tilesM = (b.p[0].M + 255) / 256, tilesN = (b.p[0].N + 127) / 128;
  const int tile_dbg = o.h3_tile;   // A/B: force a tile shape
  if (tile_dbg == 64) {
    const int tm = (b.p[0].M + 63) / 64;
    hipLaunchKernelGGL((gemm_h3_kernel<1, 2, 3, false, false, 2>), dim3(tm * tilesN, b.n), dim3(256), 0, s, b, tm, tilesN);
    return hipGetLastError();
  }
  if (tile_dbg == 192) {
    const int tm = (b.p[0].M + 127) / 128, tn = (b.p[0].N + 191) / 192;
    hipLaunchKernelGGL((gemm_h3_kernel<1, 3, 3, false, false>), dim3(tm * tn, b.n), dim3(512), 0, s, b, tm, tn);
    return hipGetLastError();
  }
  if (tile_dbg == 256) {
    hipLaunchKernelGGL((gemm_h3_kernel<2, 2, 3, false, false>), dim3(tilesM * tilesN, b.n), dim3(512), 0, s, b, tilesM, tilesN);
    return hipGetLastError();
  }
  if (b.p[0].M <= 2048 && tilesM * tilesN * b.n < 1024) {     // few rows and < 4 rounds of 256-row tiles: 128-row
    // tiles quantise better (B=64: layer-0 projection 288 -> 576 tiles; B=64 forward 0.95 -> 0.89 ms)
    const int tm = (b.p[0].M + 127) / 128;
    // ... and 64-row tiles (4 waves, 72 KB ring: two workgroups per CU) where they need fewer rounds of the chip.  Measured per round at K = 2144
    // (profiles/r05_mid_rows_gemm.txt): a 128-row tile 47 us, one workgroup per CU; a 64-row tile 33 us, and 33 us x workgroups / 256 once they share
    // CUs.  444 x 9216: 97.8 -> 62.8 us; 128 x 9216: 44.9 -> 34.4; 288 x 3072: 44.0 -> 33.2; 1024 x 3072 stays (49.2 against 53.0).
    const int t64 = o.h3_tile64;     // 0: never (A/B)
    const int tm64 = (b.p[0].M + 63) / 64;
    const double w128 = (double)tm * tilesN * b.n, w64 = (double)tm64 * tilesN * b.n;
    // (more than one 64-row workgroup per CU: the first sharing costs 1.55 rounds whatever the count -- 288 workgroups 51.8 us, 384: 53.0, 504: 62.8)
    const double c128 = 47. * (double)(long)((w128 + 255.) / 256.), c64 = 33. * (w64 <= 256. ? 1. : (w64 / 256. > 1.55 ? w64 / 256. : 1.55));
    // ... and 128 x 192 tiles (round 6) where they turn "a round and a bit" of 128 x 128 tiles into exactly one round: a tile's time follows its operand
    // bytes, (128 + 192) against (128 + 128) rows per K-tile, measured 70 us per round at K = 2048 -- the layer-1 projections of cfg-B (two products of
    // 1024 x 3072: 384 tiles of 128 x 128 = 1.5 rounds, 256 tiles of 128 x 192 = one round): 81.8 -> 72.4 us (profiles/r06_README.md).  Column counts
    // that are whole 192-column tiles only (3 Hp always is): the W planes are padded to 128 rows, not to 192.
    if (o.h3_tile192 && b.p[0].N % 192 == 0) {
      const double w192 = (double)tm * (b.p[0].N / 192) * b.n, c192 = 70. * (double)(long)((w192 + 255.) / 256.);
      if (c192 < 0.9 * c128 && c192 <= c64) {
        hipLaunchKernelGGL((gemm_h3_kernel<1, 3, 3, false, false>), dim3(tm * (b.p[0].N / 192), b.n), dim3(512), 0, s, b, tm, b.p[0].N / 192);
        return hipGetLastError();
      }
    }
    if (t64 && c64 < 0.9 * c128) {
      hipLaunchKernelGGL((gemm_h3_kernel<1, 2, 3, false, false, 2>), dim3(tm64 * tilesN, b.n), dim3(256), 0, s, b, tm64, tilesN);
      return hipGetLastError();
    }
    hipLaunchKernelGGL((gemm_h3_kernel<1, 2, 3, false, false>), dim3(tm * tilesN, b.n), dim3(512), 0, s, b, tm, tilesN);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((gemm_h3_kernel<2, 2, 3, false, false>), dim3(tilesM * tilesN, b.n), dim3(512), 0, s, b, tilesM,
                     tilesN);
  return hipGetLastError();
}

hipError_t launch_gru_h3(const H3Batch& b, hipStream_t s) {
  if (b.n <= 0 || b.p[0].M <= 0) return hipSuccess;
  int tilesM = (b.p[0].M + 127) / 128;
  const int tilesJ = b.Hp / 64;                      // block = 128 rows x (64 hidden units x 3 gates)
  if (tilesM * tilesJ * b.n <= 160) {                // mid-size batches: 64-row blocks (4 waves) fill more CUs
    tilesM = (b.p[0].M + 63) / 64;
    hipLaunchKernelGGL((gemm_h3_kernel<1, 3, 3, true, true, 2>), dim3(tilesM * tilesJ, b.n), dim3(256), 0, s, b, tilesM,
                       tilesJ);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((gemm_h3_kernel<1, 3, 3, true, true>), dim3(tilesM * tilesJ, b.n), dim3(512), 0, s, b, tilesM,
                     tilesJ);
  return hipGetLastError();
}

// First cell step of a direction: h_prev = 0, so h W_hh^T vanishes and the step is element-wise.  Writes the
// new state as fp32 and as blocked hi / lo planes for the next step's product.
__global__ void __launch_bounds__(256) gru_first_kernel(GateBatch gb, int M, int Hp, int scaled16) {
  typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
  const GateDir& d = gb.d[blockIdx.y];
  const int Hh = Hp / 2;                                     // a thread owns two consecutive hidden units
  const long total = (long)M * Hh;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / Hh;
    const int j = 2 * (int)(idx - row * Hh);
    const float* gi = d.gi + row * d.ldgi + j;
    float hv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float xr, xz, xn;
      if (d.gi_blk) {        // blocked gate pre-activations (common.h gi_blk_offset)
        xr = d.gi[gi_blk_offset(row, 0, j + c, d.gi_blk)]; xz = d.gi[gi_blk_offset(row, 1, j + c, d.gi_blk)]; xn = d.gi[gi_blk_offset(row, 2, j + c, d.gi_blk)];
      } else { xr = gi[c]; xz = gi[Hp + c]; xn = gi[2 * Hp + c]; }
      const float rg = g_sigmoid(xr + d.bhh[j + c]), zg = g_sigmoid(xz + d.bhh[Hp + j + c]);
      const float ng = g_tanh(xn + rg * d.bhh[2 * Hp + j + c]);
      hv[c] = (1.f - zg) * ng;
    }
    if (d.ho_blk) { d.hout_b[st_blk_offset(row, j, d.ho_blk)] = hv[0]; d.hout_b[st_blk_offset(row, j + 1, d.ho_blk)] = hv[1]; }
    else { d.hout[row * d.ldo + j] = hv[0]; d.hout[row * d.ldo + j + 1] = hv[1]; }
    half_t h0, l0, h1, l1;
    long o;
    if (scaled16) {       // planes of gemm_h3s.hip: value * kStateScale, [K/16][R][16]
      o = (long)(j >> 4) * d.okst + plane16_index(row, j & 15, 0);
      const float s0 = hv[0] * kStateScale, s1 = hv[1] * kStateScale;
      h0 = (half_t)s0; l0 = (half_t)(s0 - (float)h0);
      h1 = (half_t)s1; l1 = (half_t)(s1 - (float)h1);
    } else {
      o = (long)(j >> 5) * d.okst + plane_index(row, j & 31, 0);
      split_hi_lo(hv[0], h0, l0);
      split_hi_lo(hv[1], h1, l1);
    }
    *(h16x2v*)(d.hout_hi + o) = h16x2v{h0, h1};
    *(h16x2v*)(d.hout_lo + o) = h16x2v{l0, l1};
  }
}

// The same step for the scaled [K/16][R][16] planes of large batches, laid out for whole cache lines: a wave takes 16 rows x 32 hidden units,
// lane (t = lane & 15, g = lane >> 4) row t and the units 4 g .. 4 g + 3 of two 16-unit tiles -- 16-byte loads / state stores whose two tiles
// are the halves of one 128-byte line per row, and 8-byte plane stores that are 512 contiguous bytes per wave instruction (the element-wise
// kernel above moves 8 / 4 bytes per lane and scatters a wave's plane stores over 16 K-tiles).  Same arithmetic, bit-identical.
__global__ void __launch_bounds__(256) gru_first16_kernel(GateBatch gb, int M, int Hp) {
  typedef _Float16 h16x4v __attribute__((ext_vector_type(4)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const GateDir& d = gb.d[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = lane & 15, g = lane >> 4;
  const int tilesJ = Hp / 128;
  const int tm = blockIdx.x / tilesJ, tj = blockIdx.x - tm * tilesJ;
  const long row = (long)tm * 16 + t;
  if (row >= M) return;
  f32x4v gr[2], gz[2], gn[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (d.gi_blk) {          // blocked gate pre-activations (common.h gi_blk_offset): this wave's 16 x 16 block of a gate is one contiguous KB
      const float* gq = d.gi + (long)tm * d.gi_blk + gi_blk_block(0, tj * 128 + wave * 32 + u * 16) + lane * 4;
      gr[u] = *(const f32x4v*)gq; gz[u] = *(const f32x4v*)(gq + 256); gn[u] = *(const f32x4v*)(gq + 512);
    } else {
      const float* gi = d.gi + row * d.ldgi + tj * 128 + wave * 32 + u * 16 + 4 * g;
      gr[u] = *(const f32x4v*)gi; gz[u] = *(const f32x4v*)(gi + Hp); gn[u] = *(const f32x4v*)(gi + 2 * Hp);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = tj * 128 + wave * 32 + u * 16 + 4 * g;
    const f32x4v br = *(const f32x4v*)(d.bhh + j), bz = *(const f32x4v*)(d.bhh + Hp + j), bn = *(const f32x4v*)(d.bhh + 2 * Hp + j);
    f32x4v hv;
    half_t hh[4], ll[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float rg = g_sigmoid(gr[u][c] + br[c]), zg = g_sigmoid(gz[u][c] + bz[c]);
      const float ng = g_tanh(gn[u][c] + rg * bn[c]);
      hv[c] = (1.f - zg) * ng;
      const float sv = hv[c] * kStateScale;
      hh[c] = (half_t)sv;
      ll[c] = (half_t)(sv - (float)hh[c]);
    }
    if (d.ho_blk) *(f32x4v*)(d.hout_b + (long)tm * d.ho_blk + (j >> 4) * 256 + lane * 4) = hv;       // blocked state (common.h st_blk_offset)
    else *(f32x4v*)(d.hout + row * d.ldo + j) = hv;
    const long o = (long)(j >> 4) * d.okst + plane16_index(row, j & 15, 0);
    *(h16x4v*)(d.hout_hi + o) = h16x4v{hh[0], hh[1], hh[2], hh[3]};
    *(h16x4v*)(d.hout_lo + o) = h16x4v{ll[0], ll[1], ll[2], ll[3]};
  }
}

hipError_t launch_gru_first(const GateBatch& gb, int ndir, int M, int Hp, hipStream_t s, int scaled16) {
  if (M <= 0 || ndir <= 0) return hipSuccess;
  if (scaled16 && gru_first16_shape_ok(Hp)) {
    bool vec = true;
    for (int i = 0; i < ndir; ++i) {
      const GateDir& d = gb.d[i];
      vec = vec && (((size_t)d.gi | (size_t)d.bhh | (size_t)d.hout) & 15) == 0 && (d.ldgi & 3) == 0 && (d.ldo & 3) == 0 &&
            (((size_t)d.hout_hi | (size_t)d.hout_lo) & 7) == 0 && (d.okst & 3) == 0;
    }
    if (vec) {
      hipLaunchKernelGGL(gru_first16_kernel, dim3((unsigned)(((long)M + 15) / 16 * (Hp / 128)), ndir), dim3(256), 0, s, gb, M, Hp);
      return hipGetLastError();
    }
  }
  const long total = (long)M * Hp / 2;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gru_first_kernel, dim3(blocks, ndir), dim3(256), 0, s, gb, M, Hp, scaled16);
  return hipGetLastError();
}

size_t gemm_h3_ws_bytes(int M, int N, int K) {
  return 2 * align_up((size_t)M * K * 2, 256) + 2 * align_up((size_t)round_up(N, 128) * K * 2, 256) + 256;
}

}  // namespace tepose
