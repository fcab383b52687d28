// One GRU cell step of a layer >= 1 for large batches with the layer's INPUT PROJECTION FUSED IN (round 5).
//
// Reference math (lib/models/tepose.py:53-64,73-76 = torch.nn.GRU layer l >= 1, gates r, z, n):
//   gi = x_t W_ih^T + b_ih,  gh = h_{t-1} W_hh^T + b_hh,  r = sig(gi_r + gh_r),  z = sig(gi_z + gh_z),  n = tanh(gi_n + r * gh_n),
//   h_t = (1 - z) n + z h_{t-1}
// Until round 4 the two products were two kernels: a projection GEMM over all T x B rows wrote gi (fp32, 12 bytes per element) to HBM and every step
// kernel's cell update read it back with HBM-latency loads next to the other workgroup's LDS-DMA panel stream -- the half of the step that kept the
// matrix pipes 43 % idle (profiles/r04_shape_ab.txt: the K loop alone runs at 0.93 of the split roofline).  x_t of a layer >= 1 IS
// the previous layer's state of that step, which already exists as scaled hi / lo planes.  So here the step's K loop simply runs over [x_t | h_{t-1}]
// against the planes of [W_ih | W_hh] (one scale for the concatenated matrix, rows in the gate-interleaved tile order):
//   * r and z accumulate over both ranges in one accumulator; n keeps two (its x part is set aside when the K loop crosses from x to h);
//   * gi is never materialised: the layer's projection launches (5.5 ms of a 28.3 ms cfg-C step), their 3.3 GB of writes and the steps' 3.3 GB of
//     HBM-latency reads are gone; the projection's FLOPs run in this K loop instead;
//   * the cell update needs only h_{t-1} and four bias rows.  Both come through the SAME in-order LDS-DMA stream as the K panels (requested behind the
//     second-to-last pair step into the ring slots that pair has freed): h_{t-1} as the 16 x 16 blocks of the state PLANES (22 significant bits -- the
//     value the product consumed), the biases as 512 bytes per wave.  The update issues NO vector load.
// Same tile as gru_h3s16_kernel<0, 2> (gemm_h3s16.hip): 128 rows x 64 hidden units x 3 gates per workgroup, four waves of 64 x 96, transposed product on
// v_mfma_f32_16x16x32_f16 (lane (t, g) owns row t, units 4 g .. 4 g + 3 of every gate), 20 KB stages, 4-slot ring, two workgroups per CU.
// First steps (h = 0) are not this kernel's: their gi comes from one small projection of the step's slab and gru_first16_kernel (api.hip).
#include <type_traits>

#include "common.h"

namespace tepose {

typedef float f32x4f __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8f __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4f __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmf() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ float fz_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fz_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}
// a wave-uniform pointer as an SGPR pair (hipcc keeps values that went through an integer division in VGPRs; an "s" asm operand then gets a VGPR pair)
__device__ __forceinline__ const char* fz_uniform(const char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void fz_tile_of_block(int bid, int nwg, int tilesM, int tilesN, int& tm, int& tn, int GM) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int group = lin / (GM * tilesN), rem = lin - group * GM * tilesN;
  const int gm = min(GM, tilesM - group * GM);
  tm = group * GM + rem % gm;
  tn = rem / gm;
}

__global__ void __launch_bounds__(256, 2) gru_fuse16_kernel(FuseBatch batch, int tilesM, int tilesN, int GM) {
  constexpr int NWN = 2, NW = 4, NST = 4, MT = 4, NT = 6;
  constexpr int HM = 128, HN = 192, HK = 16, RB = HK * 2, RPI = 1024 / RB;
  constexpr int STAGE = (2 * HM + 2 * HN) * RB;            // 20 KB: [A_hi | A_lo | W_hi | W_lo] rows of 32 bytes
  constexpr int TOT = STAGE / 1024, Q = TOT / NW;          // 20 KB-instructions per stage, 5 per wave
  static_assert(TOT % NW == 0 && NST * STAGE <= 80 * 1024, "ring");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const FuseDir& d = batch.d[blockIdx.y];
  int tm, tn;
  fz_tile_of_block(blockIdx.x, gridDim.x, tilesM, tilesN, tm, tn, GM);
  const int m0 = tm * HM, n0 = tn * HN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int t = lane & 15, g = lane >> 4;
  const int i0 = wave * Q;
  const int Hp = batch.Hp;
  const int SX = d.Kx / HK;                                // stages of the x range
  const int NP = (d.Kx + Hp) / (2 * HK), PX = d.Kx / (2 * HK);

  // LDS-DMA requests: plane base + K position in SGPRs, one 32-bit lane offset per instruction.  A requests switch from the x planes to the
  // h planes when the stream crosses SX (hb / hk: where and with which stride they continue); W requests run on through the concatenated planes.
  const char* sbase[Q];
  const char* hb[Q];
  long kst[Q], hk[Q];
  unsigned voff[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    int ri = (i0 + q) * RPI;                               // first row of the stage image this instruction fills
    const bool isA = ri < 2 * HM;
    if (!isA) ri -= 2 * HM;
    const bool lo = ri >= (isA ? HM : HN);
    const int lrow0 = lo ? ri - (isA ? HM : HN) : ri;
    if (isA) {
      sbase[q] = (const char*)(lo ? d.Xl : d.Xh);
      kst[q] = d.x_kst * 2;
      hb[q] = (const char*)(lo ? d.Hl : d.Hh);
      hk[q] = d.h_kst * 2;
    } else {
      sbase[q] = (const char*)(lo ? d.Wl : d.Wh);
      kst[q] = d.w_kst * 2;
      hb[q] = sbase[q] + (long)SX * kst[q];
      hk[q] = kst[q];
    }
#ifndef TEPOSE_FZ_ABL
#define TEPOSE_FZ_ABL 0    // timing-only (WRONG results): 1 every workgroup streams the A rows of row tile 0 (L2-hot), 2 the W rows of unit tile 0, 4 no cell update
#endif
    const int grow = isA ? ((TEPOSE_FZ_ABL & 1) ? 0 : m0) + lrow0 + lane / 2 : ((TEPOSE_FZ_ABL & 2) ? 0 : n0) + lrow0 + lane / 2;
    voff[q] = (unsigned)grow * RB + 16u * (lane & 1);
  }
  auto dma_part = [&](int stage, int q) __attribute__((always_inline)) {
    const unsigned dst = (unsigned)(size_t)lds + (unsigned)((stage % NST) * STAGE + (i0 + q) * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[q]), "s"(sbase[q]), "s"(dst) : "m0", "memory");
    if (stage + 1 == SX) { sbase[q] = hb[q]; kst[q] = hk[q]; }
    else sbase[q] += kst[q];
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

  f32x4f acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4f{0.f, 0.f, 0.f, 0.f};
  request_pair(0);
  request_pair(1);
  // NEWER: pairs younger than pair p whose requests may stay in flight; EXTRA: other younger vector-memory requests that may stay in flight
  auto pairstep = [&](int p, auto dma, auto newer, auto extra) __attribute__((always_inline)) {
    constexpr bool DMA = decltype(dma)::value;
    constexpr int NEWER = decltype(newer)::value, EXTRA = decltype(extra)::value;
    wait_vmf<NEWER * 2 * Q + EXTRA>();
    __builtin_amdgcn_s_barrier();
    const unsigned par = (unsigned)(p & 1) * 2u * STAGE;
    const unsigned ab = abase + par, bb = bbase + par;
    // the W-side fragments stream through two 2-tile buffers (chunk c = the unit tiles of gate c), the next chunk requested before this chunk's
    // 24 MFMAs; only the last chunk runs behind B' (with the LDS-DMA requests of pair p + 2)
    h16x8f ah[MT], al[MT], bh[2][2], bl[2][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[i]) : "v"(ab), "n"(i * 16 * RB));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[i]) : "v"(ab), "n"(i * 16 * RB + A_LO));
    }
#define TEPOSE_FZ_READ_B(C)                                                                                                    \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[(C) & 1][u]) : "v"(bb), "n"((2 * (C) + u) * 16 * RB));             \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[(C) & 1][u]) : "v"(bb), "n"((2 * (C) + u) * 16 * RB + W_LO));      \
  }
    TEPOSE_FZ_READ_B(0)
    TEPOSE_FZ_READ_B(1)
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
      if (c == 0) { TEPOSE_FZ_READ_B(2) }
    }
#undef TEPOSE_FZ_READ_B
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I9 = std::integral_constant<int, 9>;
  const int jb = tn * (32 * NWN) + wn * 32;
  int p = 0;
  for (; p < PX; ++p) pairstep(p, T_{}, I1{}, I0{});       // x range (every pair of it is followed by >= 2 pairs of the h range: Hp >= 64)
  // the n gate's x part (gi_n) is complete: set it aside, its accumulators go on with gh_n
  f32x4f ni[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int u = 0; u < 2; ++u) { ni[i][u] = acc[i][4 + u]; acc[i][4 + u] = f32x4f{0.f, 0.f, 0.f, 0.f}; }
  for (; p + 2 < NP; ++p) pairstep(p, T_{}, I1{}, I0{});
  pairstep(NP - 2, F_{}, I1{}, I0{});
  // Cell operands behind pair NP - 2, into the two ring slots that pair has freed (40 KB; a wave owns 9 KB of them):
  //   8 requests: the previous state of this wave's 4 row tiles x 2 unit tiles as blocks of the state planes -- rows rt * 16 .. + 15 of K-tile
  //     (jb + u * 16) / 16: 512 contiguous bytes of the hi plane (lanes 0 .. 31) and of the lo plane (lanes 32 .. 63);
  //   1 request: the four bias rows of this wave's 32 units (lane L < 32: row L / 8, units jb + (L % 8) * 4 .. + 3).
  const unsigned cslot = (unsigned)(size_t)lds + (unsigned)(((NP - 2) & 1) * 2 * STAGE) + (unsigned)wave * 9216u;
  {
    const unsigned hoff = (unsigned)(lane & 31) * 16u + (lane >= 32 ? (unsigned)((const char*)d.Hl - (const char*)d.Hh) : 0u);
#pragma unroll
    for (int k = 0; k < 8; ++k) {                          // k = i * 2 + u
      const int rt = (m0 + wm * 16 * MT + (k >> 1) * 16) >> 4;
      const char* src = fz_uniform((const char*)(d.Hh + (long)((jb >> 4) + (k & 1)) * d.h_kst + (long)rt * 256));
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(cslot + (unsigned)k * 1024u));
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(hoff), "s"(src), "s"(dst) : "m0", "memory");
    }
    const unsigned boff = (unsigned)((((lane & 31) >> 3) * Hp + (lane & 7) * 4) * 4);
    const char* bsrc = fz_uniform((const char*)(d.bias4 + jb));
    const unsigned bdst = (unsigned)__builtin_amdgcn_readfirstlane((int)(cslot + 8192u));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(boff), "s"(bsrc), "s"(bdst) : "m0", "memory");
  }
  pairstep(NP - 1, F_{}, I0{}, I9{});                      // pair NP - 1 has landed; the 9 younger requests may still be in flight
  wait_vmf<0>();                                           // (they were issued one and a half pair steps ago and are L2 hits)

#if TEPOSE_FZ_ABL & 4
  if (acc[0][0][0] == 12345.678f) d.hout_hi[0] = (_Float16)(acc[1][1][1] + acc[2][2][2] + acc[3][3][3] + acc[0][5][0] + acc[3][4][1] + ni[0][0][0] + ni[3][1][2]);
  if (true) return;
#endif
  // ---- cell update: lane (t, g) holds, for row tile i and unit tile u, row m0 + wm * 64 + i * 16 + t and the hidden units jb + u * 16 + 4 g .. + 3
  const float inv = d.inv_scale, inv_ss = 1.f / batch.state_scale;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = m0 + wm * 16 * MT + i * 16 + t;
    // this lane's 4 units of row t: 8 bytes of the hi block and 8 of the lo block (plane16_index: 32 bytes per row, the two 16-byte slots swizzled by row bit 3)
    const unsigned sp = cslot + (unsigned)(i * 2) * 1024u + (unsigned)t * 32u + (unsigned)((((g >> 1) ^ (t >> 3)) & 1) * 16 + (g & 1) * 8);
    const unsigned sb = cslot + 8192u + (unsigned)g * 16u;
    h16x4f qh[2], ql[2];
    f32x4f brz[2][2], bni[2], bnh[2];
    asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(qh[0]) : "v"(sp));
    asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(ql[0]) : "v"(sp));
    asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(qh[1]) : "v"(sp));
    asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(ql[1]) : "v"(sp));
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(brz[0][0]) : "v"(sb));      // [kind][32 units]: kind * 128 + (u * 16 + 4 g) * 4 bytes
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(brz[0][1]) : "v"(sb));
    asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(brz[1][0]) : "v"(sb));
    asm volatile("ds_read_b128 %0, %1 offset:192" : "=v"(brz[1][1]) : "v"(sb));
    asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(bni[0]) : "v"(sb));
    asm volatile("ds_read_b128 %0, %1 offset:320" : "=v"(bni[1]) : "v"(sb));
    asm volatile("ds_read_b128 %0, %1 offset:384" : "=v"(bnh[0]) : "v"(sb));
    asm volatile("ds_read_b128 %0, %1 offset:448" : "=v"(bnh[1]) : "v"(sb));
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(qh[0]), "+v"(ql[0]), "+v"(qh[1]), "+v"(ql[1]), "+v"(brz[0][0]), "+v"(brz[0][1]), "+v"(brz[1][0]), "+v"(brz[1][1]),
                   "+v"(bni[0]), "+v"(bni[1]), "+v"(bnh[0]), "+v"(bnh[1])
                 :
                 : "memory");
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = jb + u * 16 + 4 * g;
      f32x4f v;
      _Float16 hh[4], ll[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float hprev = ((float)qh[u][c] + (float)ql[u][c]) * inv_ss;       // exact: 11 + 11 bits, power-of-two scale
        const float rg = fz_sigmoid(acc[i][0 + u][c] * inv + brz[0][u][c]);
        const float zg = fz_sigmoid(acc[i][2 + u][c] * inv + brz[1][u][c]);
        const float ng = fz_tanh((ni[i][u][c] * inv + bni[u][c]) + rg * (acc[i][4 + u][c] * inv + bnh[u][c]));
        v[c] = (1.f - zg) * ng + zg * hprev;
        const float sv = v[c] * batch.state_scale;
        hh[c] = (_Float16)sv;
        ll[c] = (_Float16)(sv - (float)hh[c]);
      }
      if (d.hout) *(f32x4f*)(d.hout + (long)row * d.ldo + j) = v;              // only where somebody reads the fp32 state (the layer's last step: the tail)
      const long o = (long)(j >> 4) * d.okst + plane16_index(row, j & 15, 0);
      *(h16x4f*)((_Float16*)d.hout_hi + o) = h16x4f{hh[0], hh[1], hh[2], hh[3]};
      *(h16x4f*)((_Float16*)d.hout_lo + o) = h16x4f{ll[0], ll[1], ll[2], ll[3]};
    }
  }
}

bool gru_fuse16_ok(const FuseBatch& b) {
  if (b.n < 1 || b.n > 3 || b.M <= 0 || b.M % 128 != 0 || b.Hp % 64 != 0) return false;
  for (int i = 0; i < b.n; ++i) {
    const FuseDir& d = b.d[i];
    const long dist = (const char*)d.Hl - (const char*)d.Hh;
    if (d.Kx < 32 || d.Kx % 32 != 0 || dist <= 0 || dist >= (1l << 31) || !d.bias4 || ((size_t)d.bias4 & 15) != 0) return false;
    if (d.hout && ((((size_t)d.hout) & 15) != 0 || (d.ldo & 3) != 0)) return false;
    if ((((size_t)d.hout_hi | (size_t)d.hout_lo) & 7) != 0 || (d.okst & 3) != 0) return false;
  }
  return true;
}

hipError_t launch_gru_fuse16(const FuseBatch& b, hipStream_t s) {
  if (!gru_fuse16_ok(b)) return hipErrorInvalidValue;
  const int tm = b.M / 128, tj = b.Hp / 64;
  static const int gm = [] { const char* e = getenv("TEPOSE_GRU_GM"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 64 ? v : 4; }();
  hipLaunchKernelGGL(gru_fuse16_kernel, dim3(tm * tj, b.n), dim3(256), 0, s, b, tm, tj, gm);
  return hipGetLastError();
}

// bias4[4][Hp] = [b_ir + b_hr | b_iz + b_hz | b_in | b_hn] from the packed [3][Hp] rows of b_ih and b_hh
__global__ void __launch_bounds__(256) bias_cat_kernel(const float* __restrict__ bih, const float* __restrict__ bhh, float* __restrict__ out, int Hp) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Hp) return;
  out[j] = bih[j] + bhh[j];
  out[Hp + j] = bih[Hp + j] + bhh[Hp + j];
  out[2 * Hp + j] = bih[2 * Hp + j];
  out[3 * Hp + j] = bhh[2 * Hp + j];
}
hipError_t launch_bias_cat(const float* bih, const float* bhh, float* out, int Hp, hipStream_t s) {
  hipLaunchKernelGGL(bias_cat_kernel, dim3((Hp + 255) / 256), dim3(256), 0, s, bih, bhh, out, Hp);
  return hipGetLastError();
}

}  // namespace tepose
