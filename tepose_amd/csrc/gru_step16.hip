// The fused GRU cell step of large batches (B >= 640 by default) on v_mfma_f32_16x16x32_f16.
//
// Reference math (lib/models/tepose.py:53-64,73-76 = torch.nn.GRU, gates r, z, n):  with gi = x_t W_ih^T + b_ih already in memory,
//   gh = h_{t-1} W_hh^T + b_hh,  r = sig(gi_r + gh_r),  z = sig(gi_z + gh_z),  n = tanh(gi_n + r * gh_n),  h_t = (1 - z) n + z h_{t-1}
// One launch = one step of up to three directions (blockIdx.y).  Workgroup = 128 rows x 64 hidden units x 3 gates: FOUR waves of 64 x 96 (the r, z, n
// tiles of 32 hidden units), W_hh rows in the gate-interleaved tile order, scaled hi / lo planes ([K/16][R][16], common.h plane16_index), 20 KB stages,
// 4-slot ring (80 KB: two workgroups per CU cover each other's stalls).  The K loop walks PAIRS of stages (one 16x16x32 MFMA spans two 16-wide K-tiles):
//   wait (pair p landed) | barrier | fragment reads, W side streamed in three chunks through two 2-tile buffers | barrier B' | last chunk + the requests of pair p + 2
// The product is formed TRANSPOSED (W fragment = the MFMA's row operand): lane (t = lane & 15, g = lane >> 4) of a 16 x 16 tile owns row t and the 4
// consecutive hidden units 4 g .. 4 g + 3 -- for r, z and n alike -- so the cell update needs no turn through LDS.
// Round-4 history of this kernel (MFMA shape A/B, eight-wave form, update order, blocked layouts): DESIGN_history.md; profiles/r04_shape_ab.txt.
//
// Two instantiations:
//   PLANES = false  the general form: gate pre-activations / fp32 previous state loaded by the update (blocked or row-major, any tile raggedness),
//                   fp32 state written next to the planes.
//   PLANES = true   (round 5, default wherever every tile is full and the operands are in the blocked layouts)
//     * the update's operands travel through the SAME in-order LDS-DMA request stream as the K panels, into ring slots the last pair steps have freed
//       (a wave's operands of one 16-row tile are 8 contiguous KB: 6 blocks of gate pre-activations, 2 of previous state), two row tiles ahead of their
//       use, at no VGPR cost: the update issues no vector load of its own (profiles/r05_gidma_hpl_ab.txt: -2 %, bit-identical);
//     * the previous state is rebuilt from the hi / lo PLANES the K loop streams anyway -- 22 significant bits, exactly the value the matrix product
//       consumed -- and the fp32 copy of a state that only the next step reads is no longer written: 8 of 24 bytes per element less beyond L2
//       (-6 % on the recurrent part; features 9e-8 from the fp32-state form, both ~1e-6 from the fp64 oracle).
//     hipcc note: the two forms are separate instantiations chosen by the launcher.  With both in one kernel behind a run-time flag hipcc threaded the
//     other path's loads in front of the LDS reads and answered them with s_waitcnt vmcnt(0), draining every request in flight; compiler-visible bias
//     loads inside the update did the same, so the PLANES form loads its biases with asm in front of the operand requests.
#include <type_traits>

#include "common.h"

namespace tepose {

typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8q __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4q __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmq() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XCD-aware block -> tile map: the workgroups resident on an XCD (equal bid % 8) cover GM row tiles x (resident / GM) unit tiles of one direction
__device__ __forceinline__ void step16_tile_of_block(int bid, int nwg, int tilesM, int tilesN, int& tm, int& tn, int GM) {
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

template <bool PLANES>
__global__ void __launch_bounds__(256, 2) gru_step16_kernel(H3SBatch batch, int tilesM, int tilesN, int GM) {
  constexpr int NWN = 2, NW = 4, NST = 4, MT = 4, NT = 6;  // wave = MT row tiles x (3 gates x 2 unit tiles) of 16 x 16
  constexpr int HM = 128, HN = 192, HK = 16, RB = HK * 2, RPI = 1024 / RB;
  constexpr int STAGE = (2 * HM + 2 * HN) * RB;            // 20 KB: [A_hi | A_lo | W_hi | W_lo] rows of 32 bytes
  constexpr int TOT = STAGE / 1024, Q = TOT / NW;          // 20 KB-instructions per stage, 5 per wave
  static_assert(TOT % NW == 0 && NST * STAGE <= 80 * 1024 && 4 * Q <= 63, "ring / vmcnt budget");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const H3SArgs& a = batch.p[blockIdx.y];
  const GateDir& d = batch.gate[blockIdx.y];
  int tm, tn;
  step16_tile_of_block(blockIdx.x, gridDim.x, tilesM, tilesN, tm, tn, GM);
  const int m0 = tm * HM, n0 = tn * HN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int t = lane & 15, g = lane >> 4;
  const int i0 = wave * Q;
  const int Hp = batch.Hp;
  const int jb = tn * (32 * NWN) + wn * 32;

  // LDS-DMA requests: plane base + K position in SGPRs, one 32-bit lane offset per instruction
  const char* sbase[Q];
  long kst[Q];
  unsigned voff[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    int ri = (i0 + q) * RPI;                               // first row of the stage image [A_hi | A_lo | W_hi | W_lo]
    const bool isA = ri < 2 * HM;
    if (!isA) ri -= 2 * HM;
    const bool lo = ri >= (isA ? HM : HN);
    const int lrow0 = lo ? ri - (isA ? HM : HN) : ri;
    sbase[q] = (const char*)(isA ? (lo ? a.Al : a.Ah) : (lo ? a.Wl : a.Wh));
    kst[q] = (isA ? a.a_kst : a.w_kst) * 2;
    const int grow = isA ? min(m0 + lrow0 + lane / 2, a.M - 1) : n0 + lrow0 + lane / 2;
    voff[q] = (unsigned)grow * RB + 16u * (lane & 1);
  }
  auto dma_part = [&](int stage, int q) __attribute__((always_inline)) {
    const unsigned dst = (unsigned)(size_t)lds + (unsigned)((stage % NST) * STAGE + (i0 + q) * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[q]), "s"(sbase[q]), "s"(dst) : "m0", "memory");
    sbase[q] += kst[q];
  };
  auto request_pair = [&](int p) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < Q; ++q) dma_part(2 * p + s, q);
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
  request_pair(0);
  request_pair(1);
  // NEWER: pairs younger than pair p whose requests may stay in flight (1, or 0 at the last pair)
  // EXTRA: other vector-memory instructions younger than the pairs' requests that may stay in flight
  auto pairstep = [&](int p, auto dma, auto newer, auto extra) __attribute__((always_inline)) {
    constexpr bool DMA = decltype(dma)::value;
    constexpr int NEWER = decltype(newer)::value, EXTRA = decltype(extra)::value;
    wait_vmq<NEWER * 2 * Q + EXTRA>();
    __builtin_amdgcn_s_barrier();
    const unsigned par = (unsigned)(p & 1) * 2u * STAGE;
    const unsigned ab = abase + par, bb = bbase + par;
    // the W-side fragments stream through two 2-tile buffers (chunk c = the unit tiles of gate c), the next chunk requested before this chunk's
    // 24 MFMAs; only the last chunk runs behind B' (with the LDS-DMA requests)
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
        __builtin_amdgcn_s_barrier();                      // B': every wave holds what it needs of pair p -> its two slots are free
      }
      __builtin_amdgcn_sched_barrier(0);
      int q = 0;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          acc[i][2 * c + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[X][u], ah[i], acc[i][2 * c + u], 0, 0, 0);
          const int n = i * 2 + u;
#pragma unroll
          for (; q < (n + 1) * 2 * Q / (MT * 2); ++q)
            if (DMA && c == 2) dma_part(2 * p + 4 + q / Q, q % Q);
        }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[i][2 * c + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[X][u], ah[i], acc[i][2 * c + u], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[i][2 * c + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[X][u], al[i], acc[i][2 * c + u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (c == 0) { TEPOSE_GRU_READ_B(2) }
    }
#undef TEPOSE_GRU_READ_B
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I6 = std::integral_constant<int, 6>;
  using I8 = std::integral_constant<int, 8>;
  int p = 0;
  for (; p + 2 < NP; ++p) pairstep(p, T_{}, I1{}, I0{});

  // ---- PLANES: cell operands through the LDS-DMA stream.  A wave's operands of one 16-row tile are 8 contiguous KB-blocks (6 of gate pre-activations:
  // (u, gate); 2 of the previous state's planes: per unit tile 512 bytes of the hi plane by lanes 0 .. 31 and of the lo plane by lanes 32 .. 63) =
  // 8 requests into a wave-private 8 KB of the ring slots whose pair every wave has read (behind that pair's B').  Row tile 0 is requested behind pair
  // NP - 2, row tile 1 behind pair NP - 1 (~1.5 pair steps before the K loop ends); row tiles 2, 3 reuse the two buffers once they have been read.
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned hpl_off = (unsigned)(lane & 31) * 16u + (lane >= 32 ? (unsigned)((const char*)a.Al - (const char*)a.Ah) : 0u);
  const unsigned gslot[2] = {(unsigned)(size_t)lds + (unsigned)(((NP - 2) & 1) * 2 * STAGE) + (unsigned)wave * 8192u,
                             (unsigned)(size_t)lds + (unsigned)(((NP - 1) & 1) * 2 * STAGE) + (unsigned)wave * 8192u};
  auto issue_rt = [&](int i, unsigned slot) __attribute__((always_inline)) {
    const int rt = (m0 + wm * 16 * MT + i * 16) >> 4;
    const char* gb = (const char*)(d.gi + (long)rt * d.gi_blk + (long)(jb >> 5) * 1536);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned dst = slot + (unsigned)k * 1024u;
      if (k < 6) {
        const char* src = gb + k * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane16), "s"(src), "s"(dst) : "m0", "memory");
      } else {
        // rows rt * 16 .. + 15 of the K-tile (jb + u * 16) / 16 of the state planes
        const char* src = (const char*)(a.Ah + (long)((jb >> 4) + (k - 6)) * a.a_kst + (long)rt * 256);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(hpl_off), "s"(src), "s"(dst) : "m0", "memory");
      }
    }
  };
  f32x4q pre_b[6];
  if constexpr (PLANES) {
    // the recurrent biases of this lane's 2 x 4 units (r, z, n): six 16-byte loads issued HERE as asm, in front of the cell operands' requests (see the
    // hipcc note in the header).  They are older than row tile 0's requests, so the wait of pair NP - 1 covers them.
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

  // ---- cell update: lane (t, g) holds, for row tile i and unit tile u, rows m0 + wm * 64 + i * 16 + t and the hidden units
  // jb + u * 16 + 4 g .. + 3 of the three gates (W_hh tile j = gate * 2 + u of this wave's 96 rows).
  // Row tile outermost, the two unit tiles of a row together: lanes g = 0..3 of a row cover 64 bytes per unit tile, and the two unit tiles are the
  // two halves of ONE 128-byte line of every operand -- requested back to back instead of one whole gate-math pass apart.
  const bool vec = (((size_t)d.hout | (size_t)d.gi | (size_t)d.hprev | (size_t)d.bhh) & 15) == 0 && (d.ldo & 3) == 0 &&
                   (d.ldgi & 3) == 0 && (d.ldh & 3) == 0;
  f32x4q br[2], bz[2], bn[2];
  int jj[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    jj[u] = min(jb + u * 16 + 4 * g, Hp - 4);
    if constexpr (PLANES) {
      br[u] = pre_b[u * 3]; bz[u] = pre_b[u * 3 + 1]; bn[u] = pre_b[u * 3 + 2];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) { br[u][c] = d.bhh[jj[u] + c]; bz[u][c] = d.bhh[Hp + jj[u] + c]; bn[u][c] = d.bhh[2 * Hp + jj[u] + c]; }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    f32x4q gr[2], gz[2], gn[2], hp[2];
    if constexpr (PLANES) {
      // vector-memory operations of this wave in issue order: rt0 rt1 | rt2 st0 | rt3 st1 | st2 | st3   (rt = 8 requests; st = 4 stores per row tile, or 6
      // where the fp32 state is written row-major for a reader outside this kernel)
      if (d.ho_blk) { if (i == 0) wait_vmq<8>(); else if (i == 1) wait_vmq<12>(); else if (i == 2) wait_vmq<16>(); else wait_vmq<8>(); }
      else { if (i == 0) wait_vmq<8>(); else if (i == 1) wait_vmq<14>(); else if (i == 2) wait_vmq<20>(); else wait_vmq<12>(); }
      const unsigned sl = gslot[i & 1] + lane16;
      // this lane's 4 units of row t: 8 bytes of the hi block and 8 of the lo block (plane16_index: 32 bytes per row, the two 16-byte slots swizzled by row bit 3)
      const unsigned sp = gslot[i & 1] + (unsigned)t * 32u + (unsigned)((((g >> 1) ^ (t >> 3)) & 1) * 16 + (g & 1) * 8);
      h16x4q qh[2], ql[2];
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(gr[0]) : "v"(sl));
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(gz[0]) : "v"(sl));
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(gn[0]) : "v"(sl));
      asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(gr[1]) : "v"(sl));
      asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(gz[1]) : "v"(sl));
      asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(gn[1]) : "v"(sl));
      asm volatile("ds_read_b64 %0, %1 offset:6144" : "=v"(qh[0]) : "v"(sp));
      asm volatile("ds_read_b64 %0, %1 offset:6656" : "=v"(ql[0]) : "v"(sp));
      asm volatile("ds_read_b64 %0, %1 offset:7168" : "=v"(qh[1]) : "v"(sp));
      asm volatile("ds_read_b64 %0, %1 offset:7680" : "=v"(ql[1]) : "v"(sp));
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(gr[0]), "+v"(gz[0]), "+v"(gn[0]), "+v"(gr[1]), "+v"(gz[1]), "+v"(gn[1]), "+v"(qh[0]), "+v"(ql[0]), "+v"(qh[1]), "+v"(ql[1])
                   :
                   : "memory");
      if (i + 2 < MT) issue_rt(i + 2, gslot[i & 1]);       // this buffer has been read: the row tile after next goes into it
      const float inv_ss = 1.f / batch.state_scale;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) hp[u][c] = ((float)qh[u][c] + (float)ql[u][c]) * inv_ss;     // exact: 11 + 11 bits, power-of-two scale
    } else {
      const int row = min(m0 + wm * 16 * MT + i * 16 + t, a.M - 1);
      const int rtl = min(m0 + wm * 16 * MT + i * 16, a.M - 1) >> 4;      // this wave's row tile (clamped like the rows)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float* gi = d.gi + (long)row * d.ldgi + jj[u];
        const float* hq = d.hprev + (long)row * d.ldh + jj[u];
        if (vec && d.gi_blk) {
          // blocked gate pre-activations (common.h gi_blk_offset): 1 KB per wave instruction, this lane's 16 bytes at lane * 16
          const float* gq = d.gi + (long)rtl * d.gi_blk + gi_blk_block(0, jb + u * 16) + lane * 4;
          gr[u] = *(const f32x4q*)gq; gz[u] = *(const f32x4q*)(gq + 256); gn[u] = *(const f32x4q*)(gq + 512);
          hp[u] = d.hp_blk ? *(const f32x4q*)(d.hprev_b + (long)rtl * d.hp_blk + ((jb + u * 16) >> 4) * 256 + lane * 4) : *(const f32x4q*)hq;
        } else if (vec) {
          gr[u] = *(const f32x4q*)gi; gz[u] = *(const f32x4q*)(gi + Hp); gn[u] = *(const f32x4q*)(gi + 2 * Hp);
          hp[u] = d.hp_blk ? *(const f32x4q*)(d.hprev_b + (long)rtl * d.hp_blk + ((jb + u * 16) >> 4) * 256 + lane * 4) : *(const f32x4q*)hq;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (d.gi_blk) {
              gr[u][c] = d.gi[gi_blk_offset(row, 0, jj[u] + c, d.gi_blk)]; gz[u][c] = d.gi[gi_blk_offset(row, 1, jj[u] + c, d.gi_blk)];
              gn[u][c] = d.gi[gi_blk_offset(row, 2, jj[u] + c, d.gi_blk)];
            } else { gr[u][c] = gi[c]; gz[u][c] = gi[Hp + c]; gn[u][c] = gi[2 * Hp + c]; }
            hp[u][c] = d.hp_blk ? d.hprev_b[st_blk_offset(row, jj[u] + c, d.hp_blk)] : hq[c];
          }
        }
      }
    }
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
        const float rg = s16_sigmoid(gr[u][c] + (hr + br[u][c]));
        const float zg = s16_sigmoid(gz[u][c] + (hz + bz[u][c]));
        const float ng = s16_tanh(gn[u][c] + rg * (hn + bn[u][c]));
        v[c] = (1.f - zg) * ng + zg * hp[u][c];
        const float sv = v[c] * batch.state_scale;
        hh[c] = (_Float16)sv;
        ll[c] = (_Float16)(sv - (float)hh[c]);
      }
      float* ho = d.hout + (long)row * d.ldo + j;
      const long o = (long)(j >> 4) * d.okst + plane16_index(row, j & 15, 0);
      if (vec) {
        // blocked fp32 state: one contiguous KB per instruction; with PLANES nobody reads it (the next step takes the planes): not written
        if (d.ho_blk) { if constexpr (!PLANES) *(f32x4q*)(d.hout_b + (long)(row >> 4) * d.ho_blk + ((jb + u * 16) >> 4) * 256 + lane * 4) = v; }
        else *(f32x4q*)ho = v;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (d.ho_blk) d.hout_b[st_blk_offset(row, j + c, d.ho_blk)] = v[c];
          else ho[c] = v[c];
        }
      }
      *(h16x4q*)((_Float16*)d.hout_hi + o) = h16x4q{hh[0], hh[1], hh[2], hh[3]};
      *(h16x4q*)((_Float16*)d.hout_lo + o) = h16x4q{ll[0], ll[1], ll[2], ll[3]};
    }
  }
}

bool gru_step16_ok(const H3SBatch& b) {
  if (b.n < 1 || b.n > 3 || b.Hp % 64 != 0) return false;
  for (int d = 0; d < b.n; ++d)
    if (b.p[d].Kp % 32 != 0 || b.p[d].Kp < 64 || b.p[d].M != b.p[0].M || b.p[d].M <= 0) return false;
  return true;
}

// the PLANES form needs every tile full and every operand in the blocked layouts (its vmcnt waits count the stores of every row tile); a forward's
// launches agree on this: it depends on the batch size, the hidden size and the layouts only (api.hip select_kernels decides, this is the launcher's check)
bool gru_step16_planes_ok(const H3SBatch& b) {
  if (!gru_step16_ok(b) || b.p[0].M % 128 != 0) return false;
  for (int d = 0; d < b.n; ++d) {
    const GateDir& g = b.gate[d];
    const long dist = (const char*)b.p[d].Al - (const char*)b.p[d].Ah;
    if (g.gi_blk == 0 || g.hp_blk == 0 || dist <= 0 || dist >= (1l << 31)) return false;
    if ((((size_t)g.hout | (size_t)g.gi | (size_t)g.hprev | (size_t)g.bhh) & 15) != 0 || (g.ldo & 3) != 0 || (g.ldgi & 3) != 0 || (g.ldh & 3) != 0) return false;
  }
  return true;
}

hipError_t launch_gru_step16(const H3SBatch& b, hipStream_t s, bool planes, int gm_opt) {
  if (b.n <= 0 || b.p[0].M <= 0) return hipSuccess;
  if (!gru_step16_ok(b) || (planes && !gru_step16_planes_ok(b))) return hipErrorInvalidValue;
  const int tm = (b.p[0].M + 127) / 128, tj = b.Hp / 64;
  // tile rows per XCD group (Options::gru_gm): the 64 workgroups resident on an XCD cover GM row tiles x 64 / GM unit tiles of one direction
  const int gm = gm_opt >= 1 && gm_opt <= 64 ? gm_opt : 4;
  if (planes) hipLaunchKernelGGL((gru_step16_kernel<true>), dim3(tm * tj, b.n), dim3(256), 0, s, b, tm, tj, gm);
  else hipLaunchKernelGGL((gru_step16_kernel<false>), dim3(tm * tj, b.n), dim3(256), 0, s, b, tm, tj, gm);
  return hipGetLastError();
}

}  // namespace tepose
