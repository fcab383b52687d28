// The persistent single-accumulator split GEMM of gemm_h3s.hip on the OTHER fp16 MFMA shape: v_mfma_f32_16x16x32_f16.
//
// Why a second shape: these kernels are power-limited (the chip holds 1.65-1.8 GHz of 2.4 under them), and the clock the chip
// holds depends on the MFMA shape -- MI355X_MICROARCH.md "DVFS give-back" item 7 measured 1.12-1.15 x the FLOP/s for 16x16x32
// over 32x32x16 at equal cycles per FLOP, on random data, operands re-read from LDS.  Same output tile per wave (64 x 128),
// same planes ([K/16][R][16] scaled hi / lo, gemm_h3s.hip), same ring bytes, same three products per k (hi*hi, lo*hi, hi*lo).
//
// What the shape changes: one MFMA spans 32 k, i.e. TWO K-tiles of the 16-wide plane format.  Lane group g = lane >> 4 of an
// operand holds 8 consecutive k: g = 0, 1 are the two 16-byte slots of a row in stage s, g = 2, 3 those of the same row in
// stage s + 1 (the same assignment on both operands, so the products pair up).  The K loop therefore walks PAIRS of stages:
//   wait (pair p landed) | barrier | fragment reads and 72 MFMAs (W side in quarters) | barrier B' | 24 MFMAs with the requests of pair p + 2 between
// The second barrier is what keeps the request depth: once every wave holds pair p's fragments in registers its two ring slots
// are free, so pair p + 2 is requested DURING interval p and has until barrier p + 2 to land -- up to 4 stages (128 KB) in
// flight against the 3 of the 32x32x16 kernel (whose "two K-tiles per barrier" variant lost 3.5 % by requesting only one pair
// ahead).  Barriers per k: the same (2 per 32).
// C layout of the transposed product (W fragment = the MFMA's row operand): lane (t = lane & 15, g) owns 4 consecutive columns
// 4 g .. 4 g + 3 of row t of a 16 x 16 tile = one 16-byte store per tile, 32 per wave as before (vmcnt accounting unchanged).
#include <type_traits>

#include "common.h"

#ifndef TEPOSE_G16_RT
#define TEPOSE_G16_RT 1     // row tiles of the cell update whose loads are issued together (A/B builds: 1, 2, 4)
#endif
#ifndef TEPOSE_G16_PRIO
#define TEPOSE_G16_PRIO 0   // A/B builds: bits 0-1 wave priority in the K loop, bits 2-3 in the cell update (0 = never touched)
#endif
#ifndef TEPOSE_G16_ABL
#define TEPOSE_G16_ABL 0    // timing-only ablations of gru_h3s16_kernel (WRONG results; tools/g16_ablate.sh): 1 no LDS-DMA, 2 no cell-update loads, 4 no fp32 state
#endif                     // store, 8 no plane stores, 16 no MFMA, 64 no cell update at all
#if TEPOSE_G16_ABL & 16
#define G16_MFMA(b, a, c) (c)
#else
#define G16_MFMA(b, a, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0)
#endif

#ifndef TEPOSE_S16_STAMPS
#define TEPOSE_S16_STAMPS 0   // diagnostic builds only (tools/s16_stamps.py): wave 0 of workgroup 0 stamps s_memtime at the phases
#endif                        // of its pair steps into a buffer of its own; the shipped kernel executes no stamp

namespace tepose {

#if TEPOSE_S16_STAMPS
__device__ unsigned long long tepose_s16_stamp_buf[8192];
__device__ unsigned long long tepose_g16_stamp_buf[16 * 16];     // fused GRU step: wave 0 of 16 sampled workgroups, 16 stamps each (tools/g16_stamps.py)
#define G16_STAMP(k) do { if (stamp_slot >= 0) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0) tepose_g16_stamp_buf[stamp_slot * 16 + (k)] = tm_; } } while (0)
#else
#define G16_STAMP(k) do { } while (0)
#endif

typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8q __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16q(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmq() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void h3s16_tile_of_block(int bid, int nwg, int tilesM, int tilesN, int& tm, int& tn, int GM = 4) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int group = lin / (GM * tilesN), rem = lin - group * GM * tilesN;
  const int gm = min(GM, tilesM - group * GM);
  tm = group * GM + rem % gm;
  tn = rem / gm;
}

__device__ __forceinline__ float s16_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float s16_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

template <int TAG>
__global__ void __launch_bounds__(512) gemm_h3s_persist16_kernel(H3SArgs a, int tilesM, int tilesN, int GM) {
  constexpr int NWN = 2, NST = 4, MT = 4, NT = 8;         // 4 x 2 waves of 64 x 128 = 4 x 8 MFMA tiles of 16 x 16
  constexpr int HM = 256, HN = 256, HK = 16, RB = HK * 2, RPI = 1024 / RB;
  constexpr int STAGE = (2 * HM + 2 * HN) * RB, TOT = STAGE / 1024, Q = TOT / 8;
  constexpr int PQ = 2 * Q;                               // LDS-DMA instructions per wave and pair of stages
  constexpr int NSTORE = MT * NT;                         // 16-byte stores per wave and tile
  static_assert(TOT % 8 == 0 && NST * STAGE + HN * 4 <= 160 * 1024 && 2 * PQ + NSTORE <= 63, "ring / vmcnt budget");
  // ONE __shared__ object (ring + bias row), as in gemm_h3s_persist_kernel
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE + HN * 4];
  float* sbias = (float*)(lds + NST * STAGE);
  const int ntiles = tilesM * tilesN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int t = lane & 15, g = lane >> 4;
  // Register budget: 128 accumulators + 64 fragment registers leave ~60 for everything else, and hipcc hoists every
  // lane-derived address of the tile set-up and of the epilogue out of the K loop, where they stay live.  Those two places
  // derive what they need from an opaque copy of the lane number instead (one asm statement: nothing to hoist).
  auto fresh_lane = [&]() __attribute__((always_inline)) {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  // A wave's LDS-DMA instructions each move 32 whole rows of one plane, so the plane (and its K stride) is wave-uniform:
  // base pointers and K positions live in SGPRs (advanced by scalar adds), a lane keeps one 32-bit byte offset per instruction
  const int i0 = wave * Q;
  bool isA[Q];
  int lrow0[Q];
  long kst[Q];
  const char* pbase[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    int ri = (i0 + q) * RPI;                              // first row of the stage image [A_hi | A_lo | W_hi | W_lo]
    isA[q] = ri < 2 * HM;
    if (!isA[q]) ri -= 2 * HM;
    const bool lo = ri >= (isA[q] ? HM : HN);
    lrow0[q] = lo ? ri - (isA[q] ? HM : HN) : ri;
    pbase[q] = (const char*)(isA[q] ? (lo ? a.Al : a.Ah) : (lo ? a.Wl : a.Wh));
    kst[q] = (isA[q] ? a.a_kst : a.w_kst) * 2;
  }
  const char* sbase[Q];                                   // SGPR: plane base + K position
  unsigned voff[Q];                                       // VGPR: this lane's row * 32 + its 16-byte half
  int m0 = 0, n0 = 0;
  auto setup = [&](int tile) {
    int tm, tn;
    h3s16_tile_of_block(tile, ntiles, tilesM, tilesN, tm, tn, GM);
    m0 = tm * HM; n0 = tn * HN;
    const int l = fresh_lane();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int grow = isA[q] ? min(m0 + lrow0[q] + l / 2, a.M - 1) : n0 + lrow0[q] + l / 2;
      voff[q] = (unsigned)grow * RB + 16u * (l & 1);
      sbase[q] = pbase[q];
    }
  };
  auto dma_part = [&](int stage, int q) {
    // (the K position passes through an opaque SGPR pair: otherwise loop strength reduction folds base + offset back into one
    // 64-bit VGPR pointer per instruction that it advances with VALU adds -- 16 more VGPRs, and the kernel spills)
    unsigned long long sb = (unsigned long long)sbase[q];
    asm volatile("" : "+s"(sb));
    glds16q((const char*)sb + voff[q], lds + (stage % NST) * STAGE + (i0 + q) * 1024);
    sbase[q] = (const char*)(sb + (unsigned long long)kst[q]);
  };
  auto request_pair = [&](int p) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < Q; ++q) dma_part(2 * p + s, q);
  };
  // fragment addresses: row (tile row t) * 32 + 16 * (slot ^ swz(row)), slot = g & 1, stage of the pair = g >> 1; the tile
  // origins are multiples of 16 rows, so the swizzle bit is that of t
  const unsigned lds0 = (unsigned)(size_t)lds;
  const unsigned sx = 16u * ((g & 1) ^ ((t >> 3) & 1)) + (unsigned)(g >> 1) * STAGE;
  const unsigned abase = lds0 + (unsigned)(wm * 16 * MT + t) * RB + sx;
  const unsigned bbase = lds0 + 2 * HM * RB + (unsigned)(wn * 16 * NT + t) * RB + sx;
  constexpr int A_LO = HM * RB, W_LO = HN * RB;
  const int KT = a.Kp / HK, NP = KT / 2;                  // the launcher guarantees an even KT >= 4
  const bool overlap = NP >= 6;
  const bool vec = (((size_t)a.C | (size_t)a.bias) & 15) == 0 && (a.ldc & 3) == 0;

  f32x4q acc[MT][NT];
#if TEPOSE_S16_STAMPS
  int stamp_n = 0;
  auto stamp = [&](int id) __attribute__((always_inline)) {
    if (blockIdx.x == 0 && wave == 0 && stamp_n < 8190) {
      const unsigned long long tm = __builtin_amdgcn_s_memtime();
      if (lane == 0) tepose_s16_stamp_buf[stamp_n] = (tm << 8) | (unsigned)id;
      ++stamp_n;
    }
  };
#define STAMP(id) stamp(id)
#else
#define STAMP(id)
#endif
  // NEWER: LDS-DMA instructions younger than pair p's that may stay in flight (PQ: pair p + 1's; 0 at the last pair);
  // EXTRA: the previous tile's stores, younger than the first two pairs of this tile
  auto pairstep = [&](int p, auto dma, auto newer, auto extra) __attribute__((always_inline)) {
    constexpr bool DMA = decltype(dma)::value;
    constexpr int NEWER = decltype(newer)::value, EXTRA = decltype(extra)::value;
    STAMP(1);
    wait_vmq<NEWER + EXTRA>();
    STAMP(2);
    __builtin_amdgcn_s_barrier();
    STAMP(3);
    const unsigned par = (unsigned)(p & 1) * 2u * STAGE;
    const unsigned ab = abase + par, bb = bbase + par;
    // Register budget (128 accumulators): the W-side fragments come in four quarters of 2 tiles through two alternating
    // buffers, quarter n + 1 requested before quarter n's MFMAs; the A-side fragments (all 4 row tiles) stay for the pair.
    // asm reads and counted lgkmcnt waits: the order below is the order executed.
    h16x8q ah[MT], al[MT], bh[2][2], bl[2][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[i]) : "v"(ab), "n"(i * 16 * RB));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[i]) : "v"(ab), "n"(i * 16 * RB + A_LO));
    }
#define TEPOSE_READ_B(QD)                                                                                                     \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                             \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[(QD) & 1][u]) : "v"(bb), "n"((2 * (QD) + u) * 16 * RB));          \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[(QD) & 1][u]) : "v"(bb), "n"((2 * (QD) + u) * 16 * RB + W_LO));   \
  }
    TEPOSE_READ_B(0)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int X = qd & 1;
      if (qd < 3) {
        TEPOSE_READ_B(qd + 1)
        asm volatile("s_waitcnt lgkmcnt(4)"
                     : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]),
                       "+v"(bh[X][0]), "+v"(bh[X][1]), "+v"(bl[X][0]), "+v"(bl[X][1])
                     :
                     : "memory");
        STAMP(4 + qd);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[X][0]), "+v"(bh[X][1]), "+v"(bl[X][0]), "+v"(bl[X][1]) : : "memory");
        STAMP(7);
        __builtin_amdgcn_s_barrier();
        STAMP(8);                      // B': every wave holds what it needs of pair p -> its two slots are free
      }
      int q = 0;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          acc[i][2 * qd + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[X][u], ah[i], acc[i][2 * qd + u], 0, 0, 0);
          const int n = i * 2 + u;
#pragma unroll
          for (; q < (n + 1) * PQ / (MT * 2); ++q)
            if (DMA && qd == 3) dma_part(2 * p + 4 + q / Q, q % Q);
        }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          acc[i][2 * qd + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[X][u], ah[i], acc[i][2 * qd + u], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          acc[i][2 * qd + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[X][u], al[i], acc[i][2 * qd + u], 0, 0, 0);
    }
#undef TEPOSE_READ_B
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using IP = std::integral_constant<int, PQ>;
  using IS = std::integral_constant<int, NSTORE>;

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  setup(tile);
  request_pair(0);
  request_pair(1);
  bool pending = false;
  for (;;) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4q{0.f, 0.f, 0.f, 0.f};
    int p = 0;
    if (pending) {
      pairstep(0, T_{}, IP{}, IS{});
      pairstep(1, T_{}, IP{}, IS{});
      p = 2;
    }
    for (; p + 2 < NP; ++p) pairstep(p, T_{}, IP{}, I0{});
    pairstep(NP - 2, F_{}, IP{}, I0{});
    pairstep(NP - 1, F_{}, I0{}, I0{});

    const int tm0 = m0, tn0 = n0;
    const int next = tile + (int)gridDim.x;
    const bool full = tm0 + HM <= a.M && tn0 + HN <= a.N && vec;
    float rs[MT];
    const int le = fresh_lane(), t = le & 15, g = le >> 4;   // (shadow the kernel-scope t, g on purpose)
    if (full) {
      if (wave == 0)
        *(f32x4q*)(sbias + 4 * le) = a.bias ? *(const f32x4q*)(a.bias + tn0 + 4 * le) : f32x4q{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = tm0 + wm * 16 * MT + i * 16 + t;
        rs[i] = a.row_scale ? a.row_scale[row] * a.inv_scale : a.inv_scale;
      }
      wait_vmq<0>();
    }
    const bool ov = full && overlap && next < ntiles;
    __syncthreads();                                       // sbias is written; (the ring has been free since the last B')
    if (next < ntiles) {
      setup(next);
      if (ov) { request_pair(0); request_pair(1); }
    }
    if (full) {
      const unsigned sb = (unsigned)(size_t)sbias + (unsigned)(wn * 16 * NT + 4 * g) * 4u;
      float* c0 = a.C + (long)(tm0 + wm * 16 * MT + t) * a.ldc + tn0 + wn * 16 * NT + 4 * g;
#pragma unroll
      for (int jj = 0; jj < NT; jj += 4) {
        f32x4q bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[u]) : "v"(sb), "n"((jj + u) * 16 * 4));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            f32x4q v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = acc[i][jj + u][c] * rs[i] + bq[u][c];
            *(f32x4q*)(c0 + (long)i * 16 * a.ldc + (jj + u) * 16) = v;
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = tm0 + wm * 16 * MT + i * 16 + t;
        if (row >= a.M) continue;
        const float rsv = a.row_scale ? a.row_scale[row] * a.inv_scale : a.inv_scale;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int col = tn0 + wn * 16 * NT + j * 16 + 4 * g + c;
            if (col < a.N) a.C[(long)row * a.ldc + col] = acc[i][j][c] * rsv + (a.bias ? a.bias[col] : 0.f);
          }
      }
    }
    if (next >= ntiles) break;
    tile = next;
    if (!ov) {                                            // partial tile / short K: drain, then fill the ring as a first tile does
      wait_vmq<0>();
      request_pair(0);
      request_pair(1);
    }
    pending = ov;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The fused GRU cell step of large batches (gemm_h3s_kernel<1, 3, 4, 2, true> in gemm_h3s.hip) on the 16x16x32 shape.
// Same block: 128 rows x 64 hidden units x 3 gates, 4 x 2 waves of 32 rows x (r, z, n of 32 units), W_hh rows in the
// gate-interleaved tile order, 20 KB stages, 4-slot ring (80 KB: two workgroups per CU cover each other's stalls).
// Differences: the K loop walks pairs of stages (barrier | 16 fragment reads | barrier B' | 36 MFMAs with the requests of pair
// p + 2 between, as the first form of the projection kernel above), and the product is formed TRANSPOSED (W fragment = the
// MFMA's row operand), so that lane (t = lane & 15, g = lane >> 4) of a 16 x 16 tile owns row t and the 4 consecutive hidden
// units 4 g .. 4 g + 3 -- for r, z and n alike.  That is the layout the cell update wants: 16-byte loads of the gate
// pre-activations / previous state, one 16-byte state store, two 8-byte plane stores, with NO turn through LDS (the 32x32x16
// form stages every accumulator through the idle ring to get there: 3 x 16 ds_write_b32 + 12 ds_read_b128 per 32-row fragment
// and a workgroup barrier).
// NWM = 4: eight waves of 32 rows x 96 columns (122 VGPRs, four waves per SIMD with two workgroups per CU).  NWM = 2 (round 4): FOUR
// waves of 64 x 96 -- 20 fragment reads per 72 MFMAs instead of 16 per 36, i.e. 160 + 80 KB of LDS traffic per pair of stages and CU
// instead of 256 + 80 (the eight-wave form is LDS-bound: 2688 LDS cycles against 2304 MFMA cycles per pair and CU), two waves per SIMD.
// GIDMA (round 5; opt-in A/B: TEPOSE_MFMA16 bit 32): the cell update's gate pre-activations / previous state travel through the SAME
// in-order LDS-DMA request stream as the K panels (into ring slots the last pair steps have freed) and are read from LDS; the launcher selects this
// instantiation only for full tiles on the blocked layouts (no run-time path choice inside: with both paths in one kernel hipcc threaded the other
// path's loads in front of the LDS reads and answered them with vmcnt(0)).  Bit-identical results.
// HPL (with GIDMA; opt-in A/B: TEPOSE_MFMA16 bit 64): the previous state of the cell update is rebuilt from the hi / lo PLANES the K loop
// streams anyway (22 significant bits: exactly the value the matrix product consumed) instead of a separate fp32 copy, and the fp32 copy of a state that only
// the next step reads (blocked layout) is no longer written: 8 of 24 bytes per element less beyond L2.  NOT bit-identical to the fp32-state form.
template <int TAG, int NWM, bool GIDMA = false, bool HPL = false>
__global__ void __launch_bounds__(128 * NWM, NWM == 2 ? 2 : 1) gru_h3s16_kernel(H3SBatch batch, int tilesM, int tilesN, int GM) {
  static_assert(!HPL || GIDMA, "the plane-fed previous state rides on the LDS-DMA-fed cell update");
  constexpr int NWN = 2, NW = NWM * NWN, NST = 4, MT = 8 / NWM, NT = 6;  // wave = MT row tiles x (3 gates x 2 unit tiles) of 16 x 16
  constexpr int HM = 128, HN = 192, HK = 16, RB = HK * 2, RPI = 1024 / RB;
  constexpr int STAGE = (2 * HM + 2 * HN) * RB;            // 20 KB
  constexpr int TOT = STAGE / 1024, Q = TOT / NW, REM = TOT % NW;   // 20 instructions per stage: waves < 4 issue 3, the others 2
  static_assert(STAGE % 1024 == 0 && NST * STAGE <= 80 * 1024 && 4 * (Q + (REM ? 1 : 0)) <= 63, "ring / vmcnt budget");
  typedef _Float16 h16x4q __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const H3SArgs& a = batch.p[blockIdx.y];
  int tm, tn;
  h3s16_tile_of_block(blockIdx.x, gridDim.x, tilesM, tilesN, tm, tn, GM);
  const int m0 = tm * HM, n0 = tn * HN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int t = lane & 15, g = lane >> 4;
  static_assert(NWM == 4 || NWM == 2, "eight or four waves");
  const int nd = Q + (wave < REM ? 1 : 0);
  const int i0 = wave * Q + min(wave, REM);

  // LDS-DMA requests: plane base + K position in SGPRs, one 32-bit lane offset per instruction (see dma_part above)
  constexpr int ND = Q + (REM ? 1 : 0);
  const char* sbase[ND];
  long kst[ND];
  unsigned voff[ND];
#pragma unroll
  for (int q = 0; q < ND; ++q) {
    int ri = min(i0 + q, TOT - 1) * RPI;                   // first row of the stage image [A_hi | A_lo | W_hi | W_lo]
    const bool isA = ri < 2 * HM;
    if (!isA) ri -= 2 * HM;
    const bool lo = ri >= (isA ? HM : HN);
    const int lrow0 = lo ? ri - (isA ? HM : HN) : ri;
    sbase[q] = (const char*)(isA ? (lo ? a.Al : a.Ah) : (lo ? a.Wl : a.Wh));
    kst[q] = (isA ? a.a_kst : a.w_kst) * 2;
    const int grow = isA ? min(m0 + lrow0 + lane / 2, a.M - 1) : n0 + lrow0 + lane / 2;
    voff[q] = (unsigned)grow * RB + 16u * (lane & 1);
  }
  auto dma_part = [&](int stage, int q) {
    if (REM == 0 || q < nd) {
      const unsigned dst = (unsigned)(size_t)lds + (unsigned)((stage % NST) * STAGE + (i0 + q) * 1024);
#if !(TEPOSE_G16_ABL & 1)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :
                   : "v"(voff[q]), "s"(sbase[q]), "s"(dst)
                   : "m0", "memory");
#else
      asm volatile("" : : "v"(voff[q]), "s"(sbase[q]), "s"(dst) : "memory");
#endif
      sbase[q] += kst[q];
    }
  };
  auto request_pair = [&](int p) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < ND; ++q) dma_part(2 * p + s, q);
  };
  const unsigned sx = 16u * ((g & 1) ^ ((t >> 3) & 1)) + (unsigned)(g >> 1) * STAGE;
  const unsigned abase = (unsigned)(size_t)lds + (unsigned)(wm * 16 * MT + t) * RB + sx;
  const unsigned bbase = (unsigned)(size_t)lds + 2 * HM * RB + (unsigned)(wn * 16 * NT + t) * RB + sx;
  constexpr int A_LO = HM * RB, W_LO = HN * RB;
  const int NP = a.Kp / (2 * HK);                          // the launcher guarantees Kp % 32 == 0, NP >= 2

  f32x4q acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4q{0.f, 0.f, 0.f, 0.f};
#if TEPOSE_S16_STAMPS
  const int stamp_slot = (blockIdx.y == 0 && wave == 0 && blockIdx.x % 64 == 5 && blockIdx.x / 64 < 16) ? (int)(blockIdx.x / 64) : -1;
#endif
  G16_STAMP(0);
#if TEPOSE_G16_PRIO
  __builtin_amdgcn_s_setprio(TEPOSE_G16_PRIO & 3);         // K loop: this wave's MFMAs / fragment reads before the cell-update VALU work of the CU's other workgroup
#endif
  request_pair(0);
  request_pair(1);
  G16_STAMP(1);
  // NEWER: pairs younger than pair p whose requests may stay in flight (1, or 0 at the last pair)
  // EXTRA: other vector-memory instructions younger than the pairs' requests that may stay in flight (the L2 touches below)
  auto pairstep = [&](int p, auto dma, auto newer, auto extra) __attribute__((always_inline)) {
    constexpr bool DMA = decltype(dma)::value;
    constexpr int NEWER = decltype(newer)::value, EXTRA = decltype(extra)::value;
    if (REM && wave < REM) wait_vmq<NEWER * 2 * (Q + 1) + EXTRA>(); else wait_vmq<NEWER * 2 * Q + EXTRA>();
    __builtin_amdgcn_s_barrier();
    const unsigned par = (unsigned)(p & 1) * 2u * STAGE;
    const unsigned ab = abase + par, bb = bbase + par;
    if constexpr (MT == 4) {
      // four waves of 64 x 96: the W-side fragments stream through two 2-tile buffers (chunk c = the unit tiles of gate c), the next
      // chunk requested before this chunk's 24 MFMAs; only the last chunk runs behind B' (with the LDS-DMA requests)
      h16x8q ah[MT], al[MT], bh[2][2], bl[2][2];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[i]) : "v"(ab), "n"(i * 16 * RB));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[i]) : "v"(ab), "n"(i * 16 * RB + A_LO));
      }
#define TEPOSE_GRU_READ_B(C)                                                                                                   \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[(C) & 1][u]) : "v"(bb), "n"((2 * (C) + u) * 16 * RB));             \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[(C) & 1][u]) : "v"(bb), "n"((2 * (C) + u) * 16 * RB + W_LO));      \
  }
      TEPOSE_GRU_READ_B(0)
      TEPOSE_GRU_READ_B(1)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int X = c & 1;
        if (c == 0) {
          asm volatile("s_waitcnt lgkmcnt(4)"
                       : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]),
                         "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bl[0][0]), "+v"(bl[0][1])
                       :
                       : "memory");
        } else if (c == 1) {
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bh[1][0]), "+v"(bh[1][1]), "+v"(bl[1][0]), "+v"(bl[1][1]) : : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bl[0][0]), "+v"(bl[0][1]) : : "memory");
          __builtin_amdgcn_s_barrier();                    // B': every wave holds what it needs of pair p -> its two slots are free
        }
        __builtin_amdgcn_sched_barrier(0);
        int q = 0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            acc[i][2 * c + u] = G16_MFMA(bh[X][u], ah[i], acc[i][2 * c + u]);
            const int n = i * 2 + u;
#pragma unroll
            for (; q < (n + 1) * 2 * ND / (MT * 2); ++q)
              if (DMA && c == 2) dma_part(2 * p + 4 + q / ND, q % ND);
          }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            acc[i][2 * c + u] = G16_MFMA(bl[X][u], ah[i], acc[i][2 * c + u]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int u = 0; u < 2; ++u)
            acc[i][2 * c + u] = G16_MFMA(bh[X][u], al[i], acc[i][2 * c + u]);
        __builtin_amdgcn_sched_barrier(0);
        if (c == 0) { TEPOSE_GRU_READ_B(2) }
      }
#undef TEPOSE_GRU_READ_B
    } else {
    h16x8q ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[i]) : "v"(ab), "n"(i * 16 * RB));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[i]) : "v"(ab), "n"(i * 16 * RB + A_LO));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[j]) : "v"(bb), "n"(j * 16 * RB));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[j]) : "v"(bb), "n"(j * 16 * RB + W_LO));
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(ah[0]), "+v"(ah[1]), "+v"(al[0]), "+v"(al[1]), "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]),
                   "+v"(bh[4]), "+v"(bh[5]), "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3]), "+v"(bl[4]), "+v"(bl[5])
                 :
                 : "memory");
    __builtin_amdgcn_s_barrier();                          // B': every wave holds pair p in registers -> its two slots are free
    int q = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        const int n = i * NT + j;
#pragma unroll
        for (; q < (n + 1) * 2 * ND / (MT * NT); ++q)
          if constexpr (DMA) dma_part(2 * p + 4 + q / ND, q % ND);
      }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  const GateDir& d = batch.gate[blockIdx.y];
  const int Hp = batch.Hp;
  const int jb = tn * (32 * NWN) + wn * 32;
  int p = 0;
  for (; p + 2 < NP; ++p) pairstep(p, T_{}, I1{}, I0{});
  // (Pulling the cell operands towards L2 two pair steps ahead -- one global_load_dword per 128-byte line, counted in the last two
  // waits -- was measured: 11.50 against 11.31 ms for the recurrent part of a forward, same baseline.  Not kept.)
  // GIDMA: cell operands through the LDS-DMA stream: a wave's operands of one 16-row tile are 8 contiguous KB-blocks (6 of gate pre-activations: (u, gate),
  // 2 of fp32 previous state) = 8 requests into a wave-private 8 KB of the ring slots whose pair every wave has read (behind that pair's B').
  // Row tile 0 is requested behind pair NP - 2, row tile 1 behind pair NP - 1 (~1.5 pair steps before the K loop ends); later row tiles reuse the
  // two buffers.  Full tiles on the blocked layouts only (the launcher checks): the vmcnt waits of the update count the stores of every row tile.
  using I6 = std::integral_constant<int, 6>;
  using I8 = std::integral_constant<int, 8>;
  const unsigned lane16 = (unsigned)lane * 16u;
  // HPL: lanes >= 32 fetch from the lo plane (the launcher checks that the planes are less than 4 GB apart)
  const unsigned hpl_off = (unsigned)(lane & 31) * 16u + (lane >= 32 ? (unsigned)((const char*)a.Al - (const char*)a.Ah) : 0u);
  const unsigned gslot[2] = {(unsigned)(size_t)lds + (unsigned)(((NP - 2) & 1) * 2 * STAGE) + (unsigned)wave * 8192u,
                             (unsigned)(size_t)lds + (unsigned)(((NP - 1) & 1) * 2 * STAGE) + (unsigned)wave * 8192u};
  auto issue_rt = [&](int i, unsigned slot) __attribute__((always_inline)) {
    const int rt = (m0 + wm * 16 * MT + i * 16) >> 4;
    const char* gb = (const char*)(d.gi + (long)rt * d.gi_blk + (long)(jb >> 5) * 1536);
    const char* hb = HPL ? nullptr : (const char*)(d.hprev_b + (long)rt * d.hp_blk + (long)(jb >> 4) * 256);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned dst = slot + (unsigned)k * 1024u;
      if (HPL && k >= 6) {
        // rows rt * 16 .. + 15 of the K-tile (jb + u * 16) / 16 of the state planes: 512 contiguous bytes of the hi plane (lanes 0..31) and of the lo plane (lanes 32..63)
        const char* src = (const char*)(a.Ah + (long)((jb >> 4) + (k - 6)) * a.a_kst + (long)rt * 256);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(hpl_off), "s"(src), "s"(dst) : "m0", "memory");
      } else {
        const char* src = k < 6 ? gb + k * 1024 : hb + (k - 6) * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane16), "s"(src), "s"(dst) : "m0", "memory");
      }
    }
  };
  f32x4q pre_b[6];
  if constexpr (GIDMA) {
    static_assert(!GIDMA || (MT == 4 && TEPOSE_G16_RT == 1), "four waves, one row tile per round");
    // the recurrent biases of this lane's 2 x 4 units (r, z, n): six 16-byte loads issued HERE as asm, in front of the cell operands' requests --
    // compiler-visible loads at the start of the cell update would be answered with vmcnt(0) at their first use (hipcc does not see the asm
    // requests), draining every row tile in flight.  They are older than row tile 0's requests, so the wait of pair NP - 1 covers them.
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float* bp = d.bhh + (long)(k % 3) * Hp + jb + (k / 3) * 16 + 4 * g;            // k = u * 3 + gate
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre_b[k]) : "v"(bp) : "memory");
    }
    pairstep(NP - 2, F_{}, I1{}, I6{});                    // (the six bias loads are younger than pair NP - 1's requests)
    issue_rt(0, gslot[0]);
    pairstep(NP - 1, F_{}, I0{}, I8{});                    // pair NP - 1 AND the bias loads have landed; the 8 younger requests (row tile 0) may still be in flight
    issue_rt(1, gslot[1]);
    asm volatile("" : "+v"(pre_b[0]), "+v"(pre_b[1]), "+v"(pre_b[2]), "+v"(pre_b[3]), "+v"(pre_b[4]), "+v"(pre_b[5]));   // defined from here on (no use may move above the wait)
  } else {
    pairstep(NP - 2, F_{}, I1{}, I0{});
    pairstep(NP - 1, F_{}, I0{}, I0{});
  }

  G16_STAMP(2);
#if TEPOSE_G16_PRIO
  __builtin_amdgcn_s_setprio((TEPOSE_G16_PRIO >> 2) & 3);
#endif
  // ---- cell update: lane (t, g) holds, for row tile i and unit tile u, rows m0 + wm * 32 + i * 16 + t and the hidden units
  // jb + u * 16 + 4 g .. + 3 of the three gates (W_hh tile j = gate * 2 + u of this wave's 96 rows)
  const bool vec = (((size_t)d.hout | (size_t)d.gi | (size_t)d.hprev | (size_t)d.bhh) & 15) == 0 && (d.ldo & 3) == 0 &&
                   (d.ldgi & 3) == 0 && (d.ldh & 3) == 0;
#if TEPOSE_G16_ABL & 64
  if (acc[0][0][0] == 12345.678f) d.hout[0] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3] + acc[0][5][0] + acc[3][4][1];
  if (true) return;
#endif
  // row tile outermost, the two unit tiles of a row together: lanes g = 0..3 of a row cover 64 bytes per unit tile, and the two unit tiles are the
  // two halves of ONE 128-byte line of every operand -- requested back to back instead of one whole gate-math pass apart
  // (a generic lambda: the LDS-DMA-fed form (TEPOSE_G16_GIDMA) is its own instantiation, chosen by ONE branch -- a flag tested per row tile let hipcc
  // issue the other path's loads first and put vmcnt(0) in front of the LDS reads)
  auto cell_update = [&](auto lds_tag) __attribute__((always_inline)) {
    constexpr bool FROM_LDS = decltype(lds_tag)::value;
    f32x4q br[2], bz[2], bn[2];
    int jj[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      jj[u] = min(jb + u * 16 + 4 * g, Hp - 4);
      if constexpr (FROM_LDS) { br[u] = pre_b[u * 3]; bz[u] = pre_b[u * 3 + 1]; bn[u] = pre_b[u * 3 + 2]; continue; }
#pragma unroll
      for (int c = 0; c < 4; ++c) { br[u][c] = d.bhh[jj[u] + c]; bz[u][c] = d.bhh[Hp + jj[u] + c]; bn[u][c] = d.bhh[2 * Hp + jj[u] + c]; }
    }
    // RT row tiles per round: their 8 RT loads are issued together (one exposed round trip per round; the loads of a wave cannot be issued
    // before its K loop ends -- vmcnt completes in order, an HBM-latency load ahead of the LDS-DMA requests would stall every pair step)
    constexpr int RT = TEPOSE_G16_RT < MT ? TEPOSE_G16_RT : MT;
#pragma unroll
    for (int i0 = 0; i0 < MT; i0 += RT) {
      f32x4q gr[RT][2], gz[RT][2], gn[RT][2], hp[RT][2];
      if constexpr (FROM_LDS) {
        // vector-memory operations of this wave in issue order: rt0 rt1 | rt2 st0 | rt3 st1 | st2 | st3   (rt = 8 requests, st = 6 stores)
        if (HPL && d.ho_blk) {      // (st = 4: the fp32 copy of a state that only the next step reads is not written)
          if (i0 == 0) wait_vmq<8>(); else if (i0 == 1) wait_vmq<12>(); else if (i0 == 2) wait_vmq<16>(); else wait_vmq<8>();
        } else {
          if (i0 == 0) wait_vmq<8>(); else if (i0 == 1) wait_vmq<14>(); else if (i0 == 2) wait_vmq<20>(); else wait_vmq<12>();
        }
        const unsigned sl = gslot[i0 & 1] + lane16;
        asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(gr[0][0]) : "v"(sl));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(gz[0][0]) : "v"(sl));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(gn[0][0]) : "v"(sl));
        asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(gr[0][1]) : "v"(sl));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(gz[0][1]) : "v"(sl));
        asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(gn[0][1]) : "v"(sl));
        if constexpr (HPL) {
          // this lane's 4 units of row t: 8 bytes of the hi block and 8 of the lo block (plane16_index: 32 bytes per row, the two 16-byte slots swizzled by row bit 3)
          typedef _Float16 h16x4r __attribute__((ext_vector_type(4)));
          const unsigned sp = gslot[i0 & 1] + (unsigned)t * 32u + (unsigned)((((g >> 1) ^ (t >> 3)) & 1) * 16 + (g & 1) * 8);
          h16x4r qh[2], ql[2];
          asm volatile("ds_read_b64 %0, %1 offset:6144" : "=v"(qh[0]) : "v"(sp));
          asm volatile("ds_read_b64 %0, %1 offset:6656" : "=v"(ql[0]) : "v"(sp));
          asm volatile("ds_read_b64 %0, %1 offset:7168" : "=v"(qh[1]) : "v"(sp));
          asm volatile("ds_read_b64 %0, %1 offset:7680" : "=v"(ql[1]) : "v"(sp));
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(gr[0][0]), "+v"(gz[0][0]), "+v"(gn[0][0]), "+v"(gr[0][1]), "+v"(gz[0][1]), "+v"(gn[0][1]), "+v"(qh[0]), "+v"(ql[0]), "+v"(qh[1]), "+v"(ql[1])
                       :
                       : "memory");
          const float inv_ss = 1.f / batch.state_scale;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) hp[0][u][c] = ((float)qh[u][c] + (float)ql[u][c]) * inv_ss;     // exact: 11 + 11 bits, power-of-two scale
        } else {
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(hp[0][0]) : "v"(sl));
        asm volatile("ds_read_b128 %0, %1 offset:7168" : "=v"(hp[0][1]) : "v"(sl));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(gr[0][0]), "+v"(gz[0][0]), "+v"(gn[0][0]), "+v"(gr[0][1]), "+v"(gz[0][1]), "+v"(gn[0][1]), "+v"(hp[0][0]), "+v"(hp[0][1])
                     :
                     : "memory");
        }
        if (i0 + 2 < MT) issue_rt(i0 + 2, gslot[i0 & 1]);  // this buffer has been read: the row tile after next goes into it
      }
#pragma unroll
      for (int ii = 0; ii < (FROM_LDS ? 0 : RT); ++ii) {
        const int row = min(m0 + wm * 16 * MT + (i0 + ii) * 16 + t, a.M - 1);
        const int rtl = min(m0 + wm * 16 * MT + (i0 + ii) * 16, a.M - 1) >> 4;      // this wave's row tile (clamped like the rows)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float* gi = d.gi + (long)row * d.ldgi + jj[u];
          const float* hq = d.hprev + (long)row * d.ldh + jj[u];
#if TEPOSE_G16_ABL & 2
          if (vec) { gr[ii][u] = f32x4q{0.1f, 0.2f, 0.3f, 0.4f}; gz[ii][u] = gr[ii][u]; gn[ii][u] = gr[ii][u]; hp[ii][u] = gr[ii][u]; asm volatile("" : : "v"(gi), "v"(hq)); } else {
#else
#if TEPOSE_G16_ABL & 128
          if (vec) {     // timing only: the same bytes as ONE contiguous 1 KB per wave instruction, 24 KB per wave and row tile... (wrong data)
            const float* gq = d.gi + ((long)((blockIdx.x * 4 + wave) * MT + (i0 + ii)) * 6 + u * 3) * 256 + lane * 4;
            gr[ii][u] = *(const f32x4q*)gq; gz[ii][u] = *(const f32x4q*)(gq + 256); gn[ii][u] = *(const f32x4q*)(gq + 512);
#if TEPOSE_G16_ABL & 256
            hp[ii][u] = *(const f32x4q*)(d.hprev + ((long)((blockIdx.x * 4 + wave) * MT + (i0 + ii)) * 2 + u) * 256 + lane * 4);
#else
            hp[ii][u] = d.hp_blk ? *(const f32x4q*)(d.hprev_b + (long)rtl * d.hp_blk + ((jb + u * 16) >> 4) * 256 + lane * 4) : *(const f32x4q*)hq;
#endif
          } else {
#else
          if (vec && d.gi_blk) {
            // blocked gate pre-activations (common.h gi_blk_offset): 1 KB per wave instruction, this lane's 16 bytes at lane * 16
            const int rt = min(m0 + wm * 16 * MT + (i0 + ii) * 16, a.M - 1) >> 4;
            const float* gq = d.gi + (long)rt * d.gi_blk + gi_blk_block(0, jb + u * 16) + lane * 4;
            gr[ii][u] = *(const f32x4q*)gq; gz[ii][u] = *(const f32x4q*)(gq + 256); gn[ii][u] = *(const f32x4q*)(gq + 512);
            hp[ii][u] = d.hp_blk ? *(const f32x4q*)(d.hprev_b + (long)rtl * d.hp_blk + ((jb + u * 16) >> 4) * 256 + lane * 4) : *(const f32x4q*)hq;
          } else if (vec) {
            gr[ii][u] = *(const f32x4q*)gi; gz[ii][u] = *(const f32x4q*)(gi + Hp); gn[ii][u] = *(const f32x4q*)(gi + 2 * Hp);
            hp[ii][u] = d.hp_blk ? *(const f32x4q*)(d.hprev_b + (long)rtl * d.hp_blk + ((jb + u * 16) >> 4) * 256 + lane * 4) : *(const f32x4q*)hq;
          } else {
#endif
#endif
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (d.gi_blk) {
                gr[ii][u][c] = d.gi[gi_blk_offset(row, 0, jj[u] + c, d.gi_blk)]; gz[ii][u][c] = d.gi[gi_blk_offset(row, 1, jj[u] + c, d.gi_blk)];
                gn[ii][u][c] = d.gi[gi_blk_offset(row, 2, jj[u] + c, d.gi_blk)];
              } else { gr[ii][u][c] = gi[c]; gz[ii][u][c] = gi[Hp + c]; gn[ii][u][c] = gi[2 * Hp + c]; }
              hp[ii][u][c] = d.hp_blk ? d.hprev_b[st_blk_offset(row, jj[u] + c, d.hp_blk)] : hq[c];
            }
          }
        }
      }
      G16_STAMP(3 + 3 * (i0 / RT));                                // loads issued
#if TEPOSE_S16_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      G16_STAMP(4 + 3 * (i0 / RT));                                // loads landed
#endif
#pragma unroll
      for (int ii = 0; ii < RT; ++ii) {
        const int i = i0 + ii;
        const int row = m0 + wm * 16 * MT + i * 16 + t;
        if (row >= a.M) continue;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = jb + u * 16 + 4 * g;
          if (j >= Hp) continue;
          f32x4q v;
          _Float16 hh[4], ll[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float hr = acc[i][0 + u][c] * a.inv_scale, hz = acc[i][2 + u][c] * a.inv_scale, hn = acc[i][4 + u][c] * a.inv_scale;
            const float rg = s16_sigmoid(gr[ii][u][c] + (hr + br[u][c]));
            const float zg = s16_sigmoid(gz[ii][u][c] + (hz + bz[u][c]));
            const float ng = s16_tanh(gn[ii][u][c] + rg * (hn + bn[u][c]));
            v[c] = (1.f - zg) * ng + zg * hp[ii][u][c];
            const float sv = v[c] * batch.state_scale;
            hh[c] = (_Float16)sv;
            ll[c] = (_Float16)(sv - (float)hh[c]);
          }
          float* ho = d.hout + (long)row * d.ldo + j;
          const long o = (long)(j >> 4) * d.okst + plane16_index(row, j & 15, 0);
#if TEPOSE_G16_ABL & 4
          asm volatile("" : : "v"(v), "v"(ho));
#else
          if (vec) {
#if TEPOSE_G16_ABL & 512
            *(f32x4q*)(d.hout + ((long)((blockIdx.x * 4 + wave) * MT + i) * 2 + u) * 256 + lane * 4) = v;     // timing only: contiguous 1 KB per instruction
#else
            if (d.ho_blk) { if constexpr (!HPL) *(f32x4q*)(d.hout_b + (long)(row >> 4) * d.ho_blk + ((jb + u * 16) >> 4) * 256 + lane * 4) = v; }     // blocked state: one contiguous KB per instruction
            else *(f32x4q*)ho = v;
#endif
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (d.ho_blk) d.hout_b[st_blk_offset(row, j + c, d.ho_blk)] = v[c];
              else ho[c] = v[c];
            }
          }
#endif
#if TEPOSE_G16_ABL & 8
          asm volatile("" : : "v"(hh[0]), "v"(hh[1]), "v"(hh[2]), "v"(hh[3]), "v"(ll[0]), "v"(ll[1]), "v"(ll[2]), "v"(ll[3]), "v"(o));
#else
          *(h16x4q*)((_Float16*)d.hout_hi + o) = h16x4q{hh[0], hh[1], hh[2], hh[3]};
          *(h16x4q*)((_Float16*)d.hout_lo + o) = h16x4q{ll[0], ll[1], ll[2], ll[3]};
#endif
        }
      }
      G16_STAMP(5 + 3 * (i0 / RT));                                // math done, stores issued
    }
  };
  cell_update(std::integral_constant<bool, GIDMA>{});
  (void)hpl_off;
#if TEPOSE_S16_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G16_STAMP(15);
#endif
}

bool gru_h3s16_ok(const H3SBatch& b) {
  if (b.n < 1 || b.Hp % 64 != 0) return false;
  for (int d = 0; d < b.n; ++d)
    if (b.p[d].Kp % 32 != 0 || b.p[d].Kp < 64 || b.p[d].M != b.p[0].M) return false;
  return true;
}

hipError_t launch_gru_h3s16(const H3SBatch& b, hipStream_t s) {
  if (b.n <= 0 || b.p[0].M <= 0) return hipSuccess;
  if (!gru_h3s16_ok(b)) return hipErrorInvalidValue;
  const int tm = (b.p[0].M + 127) / 128, tj = (b.Hp + 63) / 64;
  // tile rows per XCD group (TEPOSE_GRU_GM): the 64 workgroups resident on an XCD cover GM row tiles x 64 / GM unit tiles of one direction
  static const int gm = [] { const char* e = getenv("TEPOSE_GRU_GM"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 64 ? v : 4; }();
  if (b.p[0].shape16 == 5 || b.p[0].shape16 == 6) {        // TEPOSE_MFMA16 bit 32 / 64 (A/B): cell operands through the LDS-DMA stream, where every tile qualifies
    bool ok = b.p[0].M % 128 == 0 && b.Hp % 64 == 0;
    for (int d = 0; d < b.n; ++d) {
      const GateDir& g = b.gate[d];
      ok = ok && g.gi_blk != 0 && g.hp_blk != 0 && g.hprev_b != nullptr &&
           (((size_t)g.hout | (size_t)g.gi | (size_t)g.hprev | (size_t)g.hprev_b | (size_t)g.bhh) & 15) == 0 && (g.ldo & 3) == 0 && (g.ldgi & 3) == 0 && (g.ldh & 3) == 0;
    }
    if (ok && b.p[0].shape16 == 6) {
      for (int d = 0; d < b.n; ++d) {
        const long dist = (const char*)b.p[d].Al - (const char*)b.p[d].Ah;
        ok = ok && dist > 0 && dist < (1l << 31);
      }
      if (ok) {
        hipLaunchKernelGGL((gru_h3s16_kernel<0, 2, true, true>), dim3(tm * tj, b.n), dim3(256), 0, s, b, tm, tj, gm);
        return hipGetLastError();
      }
      // (falling back is consistent: the predicate depends on the batch size, the hidden size and the layouts only, so EVERY step launch of a forward
      // takes the same decision -- the plain kernel below reads and writes the fp32 state copies)
    } else if (ok) {
      hipLaunchKernelGGL((gru_h3s16_kernel<0, 2, true>), dim3(tm * tj, b.n), dim3(256), 0, s, b, tm, tj, gm);
      return hipGetLastError();
    }
  }
  if (b.p[0].shape16 == 2 || b.p[0].shape16 == 5 || b.p[0].shape16 == 6)          // TEPOSE_MFMA16 bit 4: four waves of 64 x 96
    hipLaunchKernelGGL((gru_h3s16_kernel<0, 2>), dim3(tm * tj, b.n), dim3(256), 0, s, b, tm, tj, gm);
  else hipLaunchKernelGGL((gru_h3s16_kernel<0, 4>), dim3(tm * tj, b.n), dim3(512), 0, s, b, tm, tj, gm);
  return hipGetLastError();
}

bool gemm_h3s16_ok(const H3SArgs& a) { return a.Kp % 32 == 0 && a.Kp >= 64; }

#if TEPOSE_S16_STAMPS
extern "C" int tepose_debug_g16_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tepose_g16_stamp_buf), (size_t)n * 8);
}
extern "C" int tepose_debug_s16_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tepose_s16_stamp_buf), (size_t)n * 8);
}
#endif

hipError_t launch_gemm_h3s16(const H3SArgs& a, hipStream_t s, int tag) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (!gemm_h3s16_ok(a)) return hipErrorInvalidValue;
  const int tilesM = (a.M + 255) / 256, tilesN = (a.N + 255) / 256;
  const int nt = tilesM * tilesN;
  // row tiles per XCD group of the walk (the 32 workgroups of an XCD take GM x 32 / GM tiles at a time).  Measured on the layer-0
  // projection, two rounds, same box (TEPOSE_S16_GM): GM 1: 11.95, 2: 11.71, 4: 11.72, 8: 11.59, 16: 12.50 ms -> 8
  static const int gm = [] { const char* e = getenv("TEPOSE_S16_GM"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= 32 ? v : 8; }();
  // (Round-4 experiment, removed: a rendezvous of the 32 workgroups of an XCD at every tile start -- one agent-scope atomic + spin -- cut the
  // beyond-L2 fetches by 10 % (8.28 -> 7.42 GB of FETCH_SIZE) and made the kernel 0.5 % SLOWER under the profiler, 9 % slower without.)
  if (tag == 0) hipLaunchKernelGGL(gemm_h3s_persist16_kernel<0>, dim3(nt < 256 ? nt : 256), dim3(512), 0, s, a, tilesM, tilesN, gm);
  else hipLaunchKernelGGL(gemm_h3s_persist16_kernel<1>, dim3(nt < 256 ? nt : 256), dim3(512), 0, s, a, tilesM, tilesN, gm);
  return hipGetLastError();
}

}  // namespace tepose
