// EXPERIMENT, measured and dropped (profiles/r06_gru_bigtile.txt): not part of the library build.  To reproduce: copy to tepose_amd/csrc/gru_step16w.hip, add it to
// __graft_entry__.SOURCES, declare gru_step16w_ok / launch_gru_step16w in common.h and call launch_gru_step16w(b, s, gm) from api.hip's `step` lambda where
// plan.planes_state && B % 256 == 0 (commit "gru_step16w: fused GRU step on the barrier-free pipeline" has that wiring; tools/gru_wide_ab.py, tools/gru_wide_pmc.sh drive it).
//
// The fused GRU cell step of large batches on the BARRIER-FREE pipeline with a 256-row tile (round 6).
//
// Reference math (lib/models/tepose.py:53-64,73-76 = torch.nn.GRU, gates r, z, n), as gru_step16.hip:
//   gh = h_{t-1} W_hh^T + b_hh,  r = sig(gi_r + gh_r),  z = sig(gi_z + gh_z),  n = tanh(gi_n + r * gh_n),  h_t = (1 - z) n + z h_{t-1}
//
// What this form changes against gru_step16_kernel<true> (128 rows x 192 columns, four waves, two workgroups per CU, two workgroup barriers per pair):
//   * tile 256 rows x 192 columns (64 hidden units x 3 gates): EIGHT waves of 64 x 96 -- the same wave tile, the same fragments, the same K order and the
//     same accumulators, so the results are bit-identical -- ONE workgroup per CU.  Global->LDS traffic at full MFMA rate: 56 KB per 2304 clocks =
//     24.9 B/clk/CU instead of 35.5 (the projection kernel, 256 x 256: 21.3);
//   * the K loop of gemm_h3s16c.hip: no workgroup barrier, one LDS arrival counter per ring-slot pair, every wave confirms its own share of the next
//     pair's LDS-DMA requests; persistent workgroups walk the (direction, row tile, unit tile) list, pairs run on across tiles;
//   * the cell update is a per-wave epilogue with plain loads (gate pre-activations from the blocked layout: 1 KB per wave instruction; h_{t-1} rebuilt
//     from the hi / lo state planes) -- nothing of it goes through the ring, which the next tile's pairs are already using.
// Why not 256 x 384 (128 units x 3 gates, arithmetic intensity 153.6): eight waves of 64 x 192 hold 192 accumulator registers per lane; at two waves
// per SIMD a wave has 256 registers, and the A fragments (32) + one double-buffered W chunk (32) + addresses do not fit beside them (268+), and four
// waves of 128 x 192 need 384 + 64 + 32 of a lone wave's 512.  Streaming the W side twice instead costs 97 B/clk/CU of LDS reads of 128.
// Needs what the plane-fed form needs (full tiles, blocked layouts) with M % 256 == 0: gru_step16w_ok.
#include "common.h"

#ifndef TEPOSE_W_VAR
#define TEPOSE_W_VAR 1     // where the landing of the next pair is confirmed: 0 after chunk 0, 1 after chunk 1, 2 after chunk 2
#endif

namespace tepose {

typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8w __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float s16w_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float s16w_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

__device__ __forceinline__ void step16w_tile_of_block(int bid, int nwg, int tilesM, int tilesN, int& tm, int& tn, int GM) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int group = lin / (GM * tilesN), rem = lin - group * GM * tilesN;
  const int gm = min(GM, tilesM - group * GM);
  tm = group * GM + rem % gm;
  tn = rem / gm;
}

__global__ void __launch_bounds__(512) gru_step16w_kernel(H3SBatch batch, int tilesM, int tilesN, int GM, unsigned* err) {
  constexpr int NWN = 2, NST = 4, MT = 4, NT = 6;
  constexpr int HM = 256, HN = 192, HK = 16, RB = HK * 2, RPI = 1024 / RB;
  constexpr int STAGE = (2 * HM + 2 * HN) * RB;             // 28 KB: [A_hi | A_lo | W_hi | W_lo] rows of 32 bytes
  constexpr int TOT = STAGE / 1024, PQ = 2 * TOT / 8;       // 28 KB-instructions per stage; 7 requests per wave per PAIR of stages
  constexpr int SPIN = 1 << 18;
  static_assert(2 * TOT % 8 == 0 && NST * STAGE + 64 <= 160 * 1024, "request split / LDS budget");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE + 64];
  const int tpd = tilesM * tilesN, ntiles = tpd * batch.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int t = lane & 15, g = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)lds;
  const unsigned cnt0 = lds0 + NST * STAGE;                 // landed[0] at cnt0, landed[1] at cnt0 + 4
  if (tid < 16) ((unsigned*)(lds + NST * STAGE))[tid] = 0u;
  __syncthreads();
  const int Hp = batch.Hp;
  auto fresh_lane = [&]() __attribute__((always_inline)) {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  // this wave's 7 requests of a pair: request k moves KB-part (wave * 7 + k) % 28 of stage (wave * 7 + k) / 28 of the pair
  bool isA[PQ], isLo[PQ];
  int lrow0[PQ], stg[PQ];
  unsigned dsto[PQ];
#pragma unroll
  for (int k = 0; k < PQ; ++k) {
    const int r = wave * PQ + k;
    stg[k] = r / TOT;
    const int part = r - stg[k] * TOT;
    int ri = part * RPI;
    isA[k] = ri < 2 * HM;
    if (!isA[k]) ri -= 2 * HM;
    isLo[k] = ri >= (isA[k] ? HM : HN);
    lrow0[k] = isLo[k] ? ri - (isA[k] ? HM : HN) : ri;
    dsto[k] = (unsigned)(stg[k] * STAGE + part * 1024);
  }
  const char* sbase[PQ];
  long kst2[PQ];
  unsigned voff[PQ];
  int m0 = 0, n0 = 0, dir = 0;
  auto setup = [&](int tile) {
    dir = tile / tpd;
    int tm, tn;
    step16w_tile_of_block(tile - dir * tpd, tpd, tilesM, tilesN, tm, tn, GM);
    m0 = tm * HM; n0 = tn * HN;
    const H3SArgs& a = batch.p[dir];
    const int l = fresh_lane();
#pragma unroll
    for (int k = 0; k < PQ; ++k) {
      const int grow = isA[k] ? min(m0 + lrow0[k] + l / 2, a.M - 1) : n0 + lrow0[k] + l / 2;
      voff[k] = (unsigned)grow * RB + 16u * (l & 1);
      const long ks = (isA[k] ? a.a_kst : a.w_kst) * 2;     // bytes between K-tiles
      sbase[k] = (const char*)(isA[k] ? (isLo[k] ? a.Al : a.Ah) : (isLo[k] ? a.Wl : a.Wh)) + (long)stg[k] * ks;
      kst2[k] = 2 * ks;
    }
  };
  auto request_one = [&](int pair, int k) __attribute__((always_inline)) {
    const unsigned dst = lds0 + (unsigned)(pair & 1) * 2u * STAGE + dsto[k];
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[k]), "s"(sbase[k]), "s"(dst) : "m0", "memory");
    sbase[k] += kst2[k];
  };
  auto request_pair = [&](int pair) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PQ; ++k) request_one(pair, k);
  };
  auto bump = [&](int slot) __attribute__((always_inline)) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt0 + 4u * (unsigned)slot), "v"(1u) : "memory");
  };
  bool dead = false;
  const unsigned INJ = batch.p[0].inject;
  auto poll = [&](int slot, int target) __attribute__((always_inline)) {
    const unsigned addr = cnt0 + 4u * (unsigned)slot;
    for (int spins = 0;; ++spins) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - (unsigned)target - INJ) >= 0) break;
      if (spins > SPIN) { dead = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  };
  const unsigned sx = 16u * ((g & 1) ^ ((t >> 3) & 1)) + (unsigned)(g >> 1) * STAGE;
  const unsigned abase = lds0 + (unsigned)(wm * 16 * MT + t) * RB + sx;
  const unsigned bbase = lds0 + 2 * HM * RB + (unsigned)(wn * 16 * NT + t) * RB + sx;
  constexpr int A_LO = HM * RB, W_LO = HN * RB;
  const int NP = batch.p[0].Kp / (2 * HK);                  // the launcher guarantees equal Kp, Kp % 32 == 0, NP >= 2

  f32x4w acc[MT][NT];
  h16x8w ah[MT], al[MT], bh[2][2], bl[2][2];
#define TEPOSE_W_READ_A(AB)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                            \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[i]) : "v"(AB), "n"(i * 16 * RB));                                  \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[i]) : "v"(AB), "n"(i * 16 * RB + A_LO));                           \
  }
#define TEPOSE_W_READ_B(BB, C)                                                                                                \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                             \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[(C) & 1][u]) : "v"(BB), "n"((2 * (C) + u) * 16 * RB));            \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[(C) & 1][u]) : "v"(BB), "n"((2 * (C) + u) * 16 * RB + W_LO));     \
  }
#define TEPOSE_W_WAIT_B(N, X)                                                                                                 \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(bh[X][0]), "+v"(bh[X][1]), "+v"(bl[X][0]), "+v"(bl[X][1]) : : "memory")
  auto chunk = [&](int C, int X, bool dma = false, int pair = 0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        acc[i][2 * C + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[X][u], ah[i], acc[i][2 * C + u], 0, 0, 0);
        if (dma && i * 2 + u < PQ) request_one(pair, i * 2 + u);      // (compile-time subscripts)
      }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[i][2 * C + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[X][u], ah[i], acc[i][2 * C + u], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[i][2 * C + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[X][u], al[i], acc[i][2 * C + u], 0, 0, 0);
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  setup(tile);
  int dtile = tile, dpt = 0;                                // the tile / pair the NEXT request belongs to
  request_pair(0);
  ++dpt;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bump(0);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};
  int pt = 0, gp = 0;
  int tm0 = m0, tn0 = n0, dir0 = dir;
  for (;;) {
    poll(gp & 1, 8 * (gp / 2 + 1));
    if (dead) break;
    bool requested = false;
    if (dpt >= NP) {
      const int nt = dtile + (int)gridDim.x;
      if (nt < ntiles) { dtile = nt; dpt = 0; setup(nt); }
    }
    if (dpt < NP) { ++dpt; requested = true; }
    const unsigned par = (unsigned)(gp & 1) * 2u * STAGE;
    const unsigned ab = abase + par, bb = bbase + par;
    TEPOSE_W_READ_A(ab)
    TEPOSE_W_READ_B(bb, 0)
    TEPOSE_W_READ_B(bb, 1)
    asm volatile("s_waitcnt lgkmcnt(4)"
                 : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]),
                   "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bl[0][0]), "+v"(bl[0][1])
                 :
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (requested) chunk(0, 0, true, gp + 1); else chunk(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    TEPOSE_W_READ_B(bb, 2)
    auto confirm = [&]() __attribute__((always_inline)) {
      if (requested) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // own share of pair gp + 1 (and this wave's older stores) landed
        bump((gp + 1) & 1);
      }
    };
    // (the bump must follow this wave's last fragment READ of pair gp -- issued above: the LDS executes a wave's instructions in order)
    if (TEPOSE_W_VAR == 0) confirm();
    TEPOSE_W_WAIT_B(4, 1);
    __builtin_amdgcn_sched_barrier(0);
    chunk(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    TEPOSE_W_WAIT_B(0, 0);                                  // every fragment of pair gp is in registers
    if (TEPOSE_W_VAR == 1) confirm();
    __builtin_amdgcn_sched_barrier(0);
    chunk(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (TEPOSE_W_VAR == 2) confirm();
    ++gp;
    if (pt + 1 < NP) { ++pt; continue; }

    // ---- the finished tile (dir0, tm0, tn0): the cell update, per wave.  Lane (t, g) holds, for row tile i and unit tile u, row tm0 + wm * 64 + i * 16 + t
    // and the hidden units jb + u * 16 + 4 g .. + 3 of the three gates (W_hh tile j = gate * 2 + u of this wave's 96 rows).
    {
      const H3SArgs& a = batch.p[dir0];
      const GateDir& d = batch.gate[dir0];
      const int le = fresh_lane(), te = le & 15, ge = le >> 4;
      const int jb = (tn0 / HN) * (32 * NWN) + wn * 32;
      f32x4w br[2], bz[2], bn[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float* bp = d.bhh + jb + u * 16 + 4 * ge;
        br[u] = *(const f32x4w*)bp; bz[u] = *(const f32x4w*)(bp + Hp); bn[u] = *(const f32x4w*)(bp + 2 * Hp);
      }
      const float inv_ss = 1.f / batch.state_scale;
      const unsigned sp = (unsigned)te * 32u + (unsigned)((((ge >> 1) ^ (te >> 3)) & 1) * 16 + (ge & 1) * 8);
      // operands of one row tile: 6 KB of gate pre-activations (blocked: this lane's 16 bytes at lane * 16 of every KB block) + this lane's 4 units of the
      // hi and lo state planes.  Software-pipelined by hand: row tile i + 1 is requested BEFORE row tile i is computed and stored (left to itself hipcc
      // keeps every row tile's loads behind the previous one's stores -- they may alias -- and exposes the load latency four times per tile)
      f32x4w G[2][6];
      h16x4w QH[2][2], QL[2][2];
      auto load_rt = [&](int i, int bf) __attribute__((always_inline)) {
        const int rt = (tm0 + wm * 16 * MT + i * 16) >> 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float* gq = d.gi + (long)rt * d.gi_blk + gi_blk_block(0, jb + u * 16) + le * 4;
          G[bf][u * 3 + 0] = *(const f32x4w*)gq; G[bf][u * 3 + 1] = *(const f32x4w*)(gq + 256); G[bf][u * 3 + 2] = *(const f32x4w*)(gq + 512);
          const long po = (long)((jb >> 4) + u) * a.a_kst + (long)rt * 256;
          QH[bf][u] = *(const h16x4w*)((const char*)(a.Ah + po) + sp);
          QL[bf][u] = *(const h16x4w*)((const char*)(a.Al + po) + sp);
        }
      };
      load_rt(0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int bf = i & 1;
        if (i + 1 < MT) load_rt(i + 1, bf ^ 1);
        const int row = tm0 + wm * 16 * MT + i * 16 + te;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = jb + u * 16 + 4 * ge;
          f32x4w v;
          _Float16 hh[4], ll[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float hp = ((float)QH[bf][u][c] + (float)QL[bf][u][c]) * inv_ss;       // exact: 11 + 11 bits, power-of-two scale
            const float hr = acc[i][0 + u][c] * a.inv_scale, hz = acc[i][2 + u][c] * a.inv_scale, hn = acc[i][4 + u][c] * a.inv_scale;
            const float rg = s16w_sigmoid(G[bf][u * 3 + 0][c] + (hr + br[u][c]));
            const float zg = s16w_sigmoid(G[bf][u * 3 + 1][c] + (hz + bz[u][c]));
            const float ng = s16w_tanh(G[bf][u * 3 + 2][c] + rg * (hn + bn[u][c]));
            v[c] = (1.f - zg) * ng + zg * hp;
            const float sv = v[c] * batch.state_scale;
            hh[c] = (_Float16)sv;
            ll[c] = (_Float16)(sv - (float)hh[c]);
          }
          if (!d.ho_blk) *(f32x4w*)(d.hout + (long)row * d.ldo + j) = v;         // row-major fp32 state for a reader outside the step kernels
          const long o = (long)(j >> 4) * d.okst + plane16_index(row, j & 15, 0);
          *(h16x4w*)((_Float16*)d.hout_hi + o) = h16x4w{hh[0], hh[1], hh[2], hh[3]};
          *(h16x4w*)((_Float16*)d.hout_lo + o) = h16x4w{ll[0], ll[1], ll[2], ll[3]};
        }
      }
    }
    const int next = tile + (int)gridDim.x;
    if (next >= ntiles) break;
    tile = next;
    // (the request stream is one pair ahead: it switched to `next` in this tile's last interval, so m0 / n0 / dir describe it)
    tm0 = m0; tn0 = n0; dir0 = dir;
    pt = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};
  }
#undef TEPOSE_W_READ_A
#undef TEPOSE_W_READ_B
#undef TEPOSE_W_WAIT_B
  if (dead && lane == 0) {                                  // never a plausible-looking wrong result with rc 0: NaN + the failure channel
    if (err) atomicAdd(err, 1u);
    *(_Float16*)batch.gate[0].hout_hi = (_Float16)__builtin_nanf("");
    if (batch.p[0].status) __hip_atomic_store(batch.p[0].status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (batch.p[0].fault) __hip_atomic_store(batch.p[0].fault, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the plane-fed form's conditions (full tiles, blocked layouts, aligned operands) with 256-row tiles
bool gru_step16w_ok(const H3SBatch& b) {
  if (!gru_step16_planes_ok(b) || b.p[0].M % 256 != 0) return false;
  for (int d = 1; d < b.n; ++d)
    if (b.p[d].Kp != b.p[0].Kp) return false;
  return true;
}

hipError_t launch_gru_step16w(const H3SBatch& b, hipStream_t s, int gm) {
  if (b.n <= 0 || b.p[0].M <= 0) return hipSuccess;
  if (!gru_step16w_ok(b)) return hipErrorInvalidValue;
  const int tm = b.p[0].M / 256, tj = b.Hp / 64;
  const int nt = tm * tj * b.n;
  const int grid = nt < 256 ? nt : 256;                    // one persistent workgroup per CU (as gemm_h3s16c.hip)
  hipLaunchKernelGGL(gru_step16w_kernel, dim3(grid), dim3(512), 0, s, b, tm, tj, gm, h3s16c_err_of_device());
  return hipGetLastError();
}

}  // namespace tepose
