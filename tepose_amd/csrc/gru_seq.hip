// Persistent recurrent kernel for small batches: ALL T cell steps of one GRU layer (up to 3 independent directions)
// in ONE launch, W_hh stationary in registers.
//
// Why: at B <= 64 a cell step is a weight-streaming problem -- 37.7 MB of W_hh planes per 3-direction step at
// H = 1024 -- and the step-per-launch kernels (skinny_h3.hip) re-stream them T times from the Infinity Cache
// (10-12 us per step, 2T+1 dependent launches per forward: 57 % of the B = 1 forward, 38 % at B = 64).  The whole
// chip's register file holds one layer's W_hh: a workgroup owns 16 hidden units of one direction = the 48 gate rows
// r, z, n of those units x K = Hp, as fp16 hi / lo planes 196 KB at Hp = 1024 = 96 VGPRs per lane of its 8 waves
// (each wave one eighth of K, in the lane layout of the B operand of v_mfma_f32_16x16x32_f16).  Hp / 16 workgroups
// per direction (64 at H = 1024; 192 for a 3-direction layer, one per CU, all resident).
//
// Per step a workgroup: waits until every workgroup of ITS direction has published the previous state (one counter
// per direction -- directions are independent chains, so there is no chip-wide barrier), reads that state's hi / lo
// planes from L2 straight into MFMA A operands, runs its 3 x MT x KS x 3 MFMAs, adds the 8 waves' partial sums
// through LDS, applies the cell update for its 16 units and publishes them (fp32 state + planes).  The same
// arithmetic as skinny_gru_h3_kernel (three fp16 MFMAs per product, fp32 accumulate, gemm_h3.hip); only the grouping
// of the K partial sums differs (8 waves instead of 4).
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Valid forms", first row):
// every handed-off byte is stored write-through (sc1: relaxed agent-scope atomic stores of 4 / 8 bytes), every
// storing wave drains (s_waitcnt vmcnt(0)), the workgroup barriers, ONE lane adds to the direction's counter
// (agent-scope atomic); the consumer's lane 0 polls that counter with sc1 loads, the workgroup barriers, and every
// load of handed-off bytes is a buffer_load ... sc1 (bypasses this CU's L1, which other CUs' stores never refresh).
// Counters are zeroed by a memset node before every launch; spins are bounded: a launch that gives up sets the forward's
// status word (the other workgroups' waits test it and end at once), publishes NaN from then on, and writes the HANDLE's
// host-visible fault word (pinned host memory, system-scope store) -- tepose_status() / the next call on the handle
// return TEPOSE_E_TIMEOUT (api.hip), so a give-up is an error at the boundary, never silent NaN.
//
// 33..64 rows of a 2-direction layer: two workgroups share a unit slice and take 32 rows each (row slices are independent
// chains with their own counters).  <= 4 rows (template flag GR): no counters at all -- the state travels as 8-byte
// {tag, hi | lo << 16} granules, the data is the flag: a consumer wave sweeps its K-slice with sc1 loads until every tag
// equals the step's, re-packs the halves into A fragments through wave-private LDS and goes (one round trip per step
// instead of drain + atomic + poll + load).  Two granule buffers alternate; the forward's memset node zeroes them, and
// tags are layer * 64 + step + 1, so no granule of another step, layer or forward can match.
#include <type_traits>

#include "common.h"

#ifndef TEPOSE_SEQ_ABL
#define TEPOSE_SEQ_ABL 0   // timing-only ablations (wrong results): 1 no state loads, 2 no MFMA, 4 no wait, 8 no drain/arrive, 16 plain loads
#endif

#ifdef TEPOSE_SEQ_STAMPS
#define SEQ_STAMP(k) do { if (a.stamps && blockIdx.x == 5 && tid == 0) a.stamps[((long)dir * kSeqMaxT + st) * 8 + (k)] = wall_clock64(); } while (0)
#define SEQ_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define SEQ_STAMP(k) do { } while (0)
#define SEQ_DRAIN() do { } while (0)
#endif

namespace tepose {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sq_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float sq_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}
// relu that keeps NaN (fmaxf(NaN, 0) is 0: a poisoned state would reach the tail product as a plausible zero)
__device__ __forceinline__ float sq_relu(float x) { return x < 0.f ? 0.f : x; }
__device__ __forceinline__ int sq_slot(long row, int q) { return ((q ^ (int)((row >> 2) & 3)) << 3); }

// two consecutive values of one row as hi / lo plane pairs (4 bytes each)
__device__ __forceinline__ void store_planes2(half_t* hi, half_t* lo, float v0, float v1) {
  half_t h0, l0, h1, l1;
  split_hi_lo(v0, h0, l0);
  split_hi_lo(v1, h1, l1);
  *(h16x2*)hi = h16x2{h0, h1};
  *(h16x2*)lo = h16x2{l0, l1};
}

__device__ __forceinline__ h16x8 as_h8(u32x4 v) {
  union { u32x4 u; h16x8 h; } c;
  c.u = v;
  return c.h;
}

}  // namespace

// MT: 16-row tiles of the batch (M <= 16 MT); KS: K-tiles (of 32) per wave = Hp / 256.
template <int MT, int KS, bool GR = false>
__global__ void __launch_bounds__(512) gru_seq_kernel(GruSeqArgs a) {
  constexpr int NW = 8;
  __shared__ __attribute__((aligned(16))) float red[NW * MT * 3 * 256];
  // granule mode: per wave, its K-slice of the previous state as hi / lo rows [16][32 KS (+8 pad)] for the A fragments
  constexpr int GLD = 32 * KS + 8;
  __shared__ __attribute__((aligned(16))) half_t gst[GR ? NW * 2 * 16 * GLD : 8];
  // this direction's step table: a kernel-argument read per step is a scalar-cache miss (~0.7 us) on the critical path
  __shared__ __attribute__((aligned(16))) GruSeqStep tab[kSeqMaxT];
  __shared__ unsigned gave_up;       // a bounded wait expired: everything this workgroup publishes from now on is NaN
  if (threadIdx.x == 0) gave_up = 0;
  const int dir = blockIdx.z;
  const int Hp = a.Hp, M = a.M, T = a.T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int j0 = blockIdx.x * 16;
  const unsigned cpd = gridDim.x;                 // workgroups of this direction

  {
    const unsigned* src = (const unsigned*)&a.st[dir][0];
    unsigned* dst = (unsigned*)tab;
    for (int i = threadIdx.x; i < (int)(a.T * sizeof(GruSeqStep) / 4); i += 512) dst[i] = src[i];
  }
  // ---- W_hh slice -> registers (once): rows of the gate-interleaved tile order (ROW_GATES_TILED), this wave's K-tiles
  h16x8 wh[KS][3], wl[KS][3];
  {
    const int rbase = (j0 >> 6) * 192 + ((j0 & 63) >> 5) * 96 + (j0 & 31);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const long row = rbase + g * 32 + r16;
      const long o = row * 32 + sq_slot(row, q);
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const long ko = (long)(wave * KS + c) * a.w_kst;
        wh[c][g] = *(const h16x8*)(a.whi[dir] + o + ko);
        wl[c][g] = *(const h16x8*)(a.wlo[dir] + o + ko);
      }
    }
  }

  // ---- epilogue item of this thread: tile i, row rr of the tile, unit pair p (2 consecutive hidden units)
  const bool item = tid < MT * 128;
  const int ei = tid >> 7, err = (tid >> 3) & 15, ep = tid & 7;
  const int m0 = blockIdx.y * (MT * 16);          // first batch row of this workgroup (row slices are independent chains)
  const int erow = m0 + ei * 16 + err;            // batch row
  const int ej = j0 + 2 * ep;                     // first of the two hidden units
  const bool live = item && erow < M;
  float2 br = {0.f, 0.f}, bz = br, bn = br, hp = br;
  if (live) {
    br = *(const float2*)(a.bhh[dir] + ej);
    bz = *(const float2*)(a.bhh[dir] + Hp + ej);
    bn = *(const float2*)(a.bhh[dir] + 2 * Hp + ej);
  }
  // partial sums of (erow, ej..ej+1): MFMA D layout = lane (q' * 16 + col), register e = row q' * 4 + e
  const int rq = err >> 2, re = err & 3;
  const float* rbase_e = red + ((ei * 3) * 4 + re) * 64 + rq * 16 + 2 * ep;

  // ---- one row (the live stream, B = 1; granule mode): 8 threads used to finish the workgroup's 16 units -- 24 LDS reads
  // of partial sums and two units' gate math each, 0.6 us of a 2.2 us step (tools/seq_stamps.py).  Here every thread
  // takes ONE partial sum: thread = (unit u1 = tid / 32, gate g1 = (tid / 8) % 4 [3: idle], K-partial pw1 = tid % 8);
  // three DPP steps add the 8 partials, two more bring z and n to the r lane, and the 16 lanes tid % 32 == 0 finish one
  // unit each.  (The partials are then added as a tree, not in wave order: last-bit differences against B > 1.)
  const bool one = GR && M == 1;
  const int u1 = tid >> 5, g1 = (tid >> 3) & 3, pw1 = tid & 7;
  const bool ul = one && (tid & 31) == 0;
  const int j1 = j0 + u1;
  float b1r = 0.f, b1z = 0.f, b1n = 0.f, hp1 = 0.f;
  if (ul) { b1r = a.bhh[dir][j1]; b1z = a.bhh[dir][Hp + j1]; b1n = a.bhh[dir][2 * Hp + j1]; }

  // ---- the extra direction's single step from h = 0 (element-wise; consumed by later kernels only: plain stores)
  if (dir == 0 && a.x_gi && live) {
    const float* gi = a.x_gi + (long)erow * a.x_ldgi + ej;
    const float2 gr = *(const float2*)gi, gz = *(const float2*)(gi + Hp), gn = *(const float2*)(gi + 2 * Hp);
    const float2 xr = *(const float2*)(a.x_bhh + ej), xz = *(const float2*)(a.x_bhh + Hp + ej),
                 xn = *(const float2*)(a.x_bhh + 2 * Hp + ej);
    float hv[2];
    hv[0] = (1.f - sq_sigmoid(gz.x + xz.x)) * sq_tanh(gn.x + sq_sigmoid(gr.x + xr.x) * xn.x);
    hv[1] = (1.f - sq_sigmoid(gz.y + xz.y)) * sq_tanh(gn.y + sq_sigmoid(gr.y + xr.y) * xn.y);
    *(float2*)(a.x_hout + (long)erow * a.x_ldo + ej) = float2{hv[0], hv[1]};
    const long po = (long)(ej >> 5) * a.x_pkst + plane_index(erow, ej & 31, 0);
    store_planes2(a.phi + a.x_poff + po, a.plo + a.x_poff + po, hv[0], hv[1]);
    if (a.x_roff != kNoPlane) {
      const long ro = (long)a.x_roff + (long)(ej >> 5) * a.r_kst + plane_index(erow, ej & 31, 0);
      store_planes2(a.rhi + ro, a.rlo + ro, sq_relu(hv[0]), sq_relu(hv[1]));
    }
  }

  __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc((void*)a.phi, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc((void*)a.plo, 0, 0x7fffffff, 0x00020000);
  unsigned* counter = a.counters + dir * 32 + blockIdx.y * 16;   // one per (direction, row slice)
  if (GR) for (int i = tid; i < NW * 2 * 16 * GLD; i += 512) gst[i] = (half_t)0.f;
  __syncthreads();                                // `tab` is filled

  for (int st = 0; st < T; ++st) {
    const GruSeqStep s = tab[st];
    // gate pre-activations of this step (written by an earlier kernel: plain loads), issued before the wait
    float2 gr = {0.f, 0.f}, gz = gr, gn = gr;
    if (live && !one) {
      const float* gi = s.gi + (long)erow * s.ldgi + ej;
      gr = *(const float2*)gi; gz = *(const float2*)(gi + Hp); gn = *(const float2*)(gi + 2 * Hp);
    }
    float g1r = 0.f, g1z = 0.f, g1n = 0.f;
    if (ul) { const float* gi = s.gi + j1; g1r = gi[0]; g1z = gi[Hp]; g1n = gi[2 * Hp]; }     // row 0
    float2 hr = {0.f, 0.f}, hz = hr, hn = hr;
    float h1r = 0.f, h1z = 0.f, h1n = 0.f;
    SEQ_STAMP(0);
    if (st > 0) {
      if constexpr (GR) {
        // sweep this wave's K-slice of the previous state until every granule carries this step's tag
        const unsigned want = a.tag_base + (unsigned)st + a.inject;
        const unsigned long long* gsrc = a.gran + ((size_t)(dir * 2 + ((st - 1) & 1)) * kSeqGranRows) * Hp + wave * 32 * KS + 2 * lane;
        const bool mine = 2 * lane < 32 * KS;
        half_t* gh = gst + (wave * 2) * 16 * GLD;
        half_t* gl = gh + 16 * GLD;
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
          unsigned long long g0[kSeqGranRows], g1[kSeqGranRows];
#pragma unroll
          for (int r = 0; r < (int)kSeqGranRows; ++r)        // every row's loads in flight together (M is uniform)
            if (r < M && mine) {
              g0[r] = __hip_atomic_load(gsrc + (size_t)r * Hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              g1[r] = __hip_atomic_load(gsrc + (size_t)r * Hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
          for (int r = 0; r < (int)kSeqGranRows; ++r)
            if (r < M && mine) {
              ok &= (unsigned)(g0[r] >> 32) == want && (unsigned)(g1[r] >> 32) == want;
              // low word of a granule = hi half | lo half << 16
              *(unsigned*)(gh + r * GLD + 2 * lane) = ((unsigned)g0[r] & 0xffffu) | (((unsigned)g1[r] & 0xffffu) << 16);
              *(unsigned*)(gl + r * GLD + 2 * lane) = (((unsigned)g0[r] >> 16) & 0xffffu) | ((unsigned)g1[r] & 0xffff0000u);
            }
          if (__all(ok)) break;
          ++spins;
          // somebody else gave up already (status word, or a wave of this workgroup): do not sit out the rest of this wait,
          // nor any later one (tested every 1024 sweeps only: the sweep is the B <= 4 step's critical path)
          if ((spins & 1023u) == 0u && (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                                        *(volatile unsigned*)&gave_up != 0u)) {
            if (lane == 0) gave_up = 1;
            break;
          }
          if (spins > (a.spin_limit >> 1)) {                // ~1 s by default: a workgroup of the direction is not resident
            if (lane == 0) {
              __hip_atomic_store(a.status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (a.fault) __hip_atomic_store(a.fault, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              gave_up = 1;
            }
            break;
          }
        }
        SEQ_STAMP(1);
        SEQ_STAMP(2);
        h16x8 xh[KS], xl[KS];
#pragma unroll
        for (int c = 0; c < KS; ++c) {
          xh[c] = *(const h16x8*)(gh + r16 * GLD + 32 * c + 8 * q);
          xl[c] = *(const h16x8*)(gl + r16 * GLD + 32 * c + 8 * q);
        }
        f32x4 acc[3], accx[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < KS; ++c) {
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[c], wh[c][g], acc[g], 0, 0, 0);
            accx[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[c], wl[c][g], accx[g], 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 3; ++g)
            accx[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[c], wh[c][g], accx[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int ee = 0; ee < 4; ++ee)
            red[((wave * MT * 3 + g) * 4 + ee) * 64 + lane] = acc[g][ee] + accx[g][ee] * (1.f / kLoScale);
      } else {
      // every workgroup of this direction has published step st - 1
      if (tid == 0 && !(TEPOSE_SEQ_ABL & 4) && !gave_up) {
        const unsigned want = cpd * (unsigned)st + a.inject;
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          __builtin_amdgcn_s_sleep(1);
          ++spins;
          // another workgroup gave up already: this wait (and every later one, `gave_up`) ends at once
          if ((spins & 1023u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            gave_up = 1;
            break;
          }
          if (spins > a.spin_limit) {             // ~2 s by default: give up (a workgroup of the direction is not resident / died)
            __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.fault) __hip_atomic_store(a.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            gave_up = 1;
            break;
          }
        }
      }
      SEQ_STAMP(1);
      __syncthreads();
      SEQ_STAMP(2);
      const unsigned pbase = tab[st - 1].poff * 2u;   // byte offset of the previous state's view inside the planes
      const unsigned pkst2 = tab[st - 1].pkst * 2u;
      // A operands of tile i + 1 are in flight while tile i's MFMAs run (a tile's loads take ~1 us through L2)
      h16x8 ah[2][KS], al[2][KS];
      auto load_tile = [&](int i, h16x8 (&h)[KS], h16x8 (&l)[KS]) {
        const long row = m0 + i * 16 + r16;
        const unsigned ro = pbase + (unsigned)(row * 32 + sq_slot(row, q)) * 2u;
#pragma unroll
        for (int c = 0; c < KS; ++c) {
          const unsigned o = ro + (unsigned)(wave * KS + c) * pkst2;
#if TEPOSE_SEQ_ABL & 1
          h[c] = as_h8(u32x4{o, o, o, o}); l[c] = h[c];
#elif TEPOSE_SEQ_ABL & 16
          h[c] = as_h8(__builtin_amdgcn_raw_buffer_load_b128(rs_hi, o, 0, 0));
          l[c] = as_h8(__builtin_amdgcn_raw_buffer_load_b128(rs_lo, o, 0, 0));
#else
          h[c] = as_h8(__builtin_amdgcn_raw_buffer_load_b128(rs_hi, o, 0, 16));
          l[c] = as_h8(__builtin_amdgcn_raw_buffer_load_b128(rs_lo, o, 0, 16));
#endif
        }
      };
      load_tile(0, ah[0], al[0]);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i + 1 < MT) load_tile(i + 1, ah[(i + 1) & 1], al[(i + 1) & 1]);
        const h16x8 (&xh)[KS] = ah[i & 1];
        const h16x8 (&xl)[KS] = al[i & 1];
        f32x4 acc[3], accx[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#if TEPOSE_SEQ_ABL & 2
#pragma unroll
        for (int c = 0; c < KS; ++c)
#pragma unroll
          for (int g = 0; g < 3; ++g) { acc[g][0] += (float)xh[c][0] * (float)wh[c][g][0]; accx[g][0] += (float)xl[c][0] * (float)wl[c][g][0]; }
#else
#pragma unroll
        for (int c = 0; c < KS; ++c) {
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[c], wh[c][g], acc[g], 0, 0, 0);
            accx[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[c], wl[c][g], accx[g], 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 3; ++g)
            accx[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[c], wh[c][g], accx[g], 0, 0, 0);
        }
#endif
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int ee = 0; ee < 4; ++ee)
            red[((wave * MT * 3 + i * 3 + g) * 4 + ee) * 64 + lane] = acc[g][ee] + accx[g][ee] * (1.f / kLoScale);
      }
      }
      SEQ_DRAIN();
      SEQ_STAMP(3);
      __syncthreads();
      SEQ_STAMP(4);
      if (one) {
        static_assert(NW == 8, "8 K-partials per sum");
        // row 0 of the MFMA D layout: lanes 0..15 = units, register 0
        float v = g1 < 3 ? red[((pw1 * MT * 3 + g1) * 4) * 64 + u1] : 0.f;
        auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
          return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
        };
        v += dpp(v, std::integral_constant<int, 0xB1>{});        // quad_perm [1,0,3,2]
        v += dpp(v, std::integral_constant<int, 0x4E>{});        // quad_perm [2,3,0,1]
        v += dpp(v, std::integral_constant<int, 0x141>{});       // row_half_mirror: the other quad of the 8 lanes
        h1r = v;
        h1z = dpp(v, std::integral_constant<int, 0x108>{});      // row_shl:8  -> the z sum of this unit (lane + 8)
        h1n = __shfl(v, lane + 16);                              // the n sum (lane + 16: next DPP row)
      } else if (live) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const float* rp = rbase_e + (long)w * MT * 3 * 256;
          const float2 vr = *(const float2*)rp, vz = *(const float2*)(rp + 256), vn = *(const float2*)(rp + 512);
          hr.x += vr.x; hr.y += vr.y; hz.x += vz.x; hz.y += vz.y; hn.x += vn.x; hn.y += vn.y;
        }
      }
    }
    if (one) {
      if (ul) {
        const float rg = sq_sigmoid(g1r + (h1r + b1r));
        const float zg = sq_sigmoid(g1z + (h1z + b1z));
        const float ng = sq_tanh(g1n + rg * (h1n + b1n));
        float hv = (1.f - zg) * ng + zg * hp1;
        if (gave_up) hv = __builtin_nanf("");
        hp1 = hv;
        half_t h0, l0;
        split_hi_lo(hv, h0, l0);
        union { half_t h; unsigned short u; } c0, c1;
        c0.h = h0; c1.h = l0;
        if (st + 1 < T) {
          unsigned long long* gd = a.gran + ((size_t)(dir * 2 + (st & 1)) * kSeqGranRows) * Hp + j1;
          const unsigned long long tg = (unsigned long long)(a.tag_base + (unsigned)st + 1u) << 32;
          __hip_atomic_store(gd, tg | c0.u | ((unsigned)c1.u << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s.hout[j1] = hv;                                                       // row 0; later kernels: plain stores
        const long o = (long)s.poff + (long)(j1 >> 5) * s.pkst + plane_index(0, j1 & 31, 0);
        a.phi[o] = h0;
        a.plo[o] = l0;
        if (st == T - 1 && a.r_off[dir] != kNoPlane) {
          const long ro = (long)a.r_off[dir] + (long)(j1 >> 5) * a.r_kst + plane_index(0, j1 & 31, 0);
          half_t rh, rl;
          split_hi_lo(sq_relu(hv), rh, rl);
          a.rhi[ro] = rh;
          a.rlo[ro] = rl;
        }
      }
    } else if (live) {
      float hv[2];
      {
        const float rg = sq_sigmoid(gr.x + (hr.x + br.x));
        const float zg = sq_sigmoid(gz.x + (hz.x + bz.x));
        const float ng = sq_tanh(gn.x + rg * (hn.x + bn.x));
        hv[0] = (1.f - zg) * ng + zg * hp.x;
      }
      {
        const float rg = sq_sigmoid(gr.y + (hr.y + br.y));
        const float zg = sq_sigmoid(gz.y + (hz.y + bz.y));
        const float ng = sq_tanh(gn.y + rg * (hn.y + bn.y));
        hv[1] = (1.f - zg) * ng + zg * hp.y;
      }
      if (gave_up) { hv[0] = hv[1] = __builtin_nanf(""); }   // never a plausible-looking wrong state: NaN to the outputs
      hp.x = hv[0]; hp.y = hv[1];
      half_t h0, l0, h1, l1;
      split_hi_lo(hv[0], h0, l0);
      split_hi_lo(hv[1], h1, l1);
      union { h16x2 h; unsigned u; } ph, pl;
      ph.h = h16x2{h0, h1}; pl.h = h16x2{l0, l1};
      const long o = (long)s.poff + (long)(ej >> 5) * s.pkst + plane_index(erow, ej & 31, 0);
      if constexpr (GR) {
        // the other workgroups read granules {tag, hi | lo << 16}: ONE 8-byte write-through store each, nothing to drain;
        // the fp32 state and the planes are for later kernels (plain stores)
        if (st + 1 < T) {
          unsigned long long* gd = a.gran + ((size_t)(dir * 2 + (st & 1)) * kSeqGranRows + erow) * Hp + ej;
          const unsigned long long tg = (unsigned long long)(a.tag_base + (unsigned)st + 1u) << 32;
          union { half_t h; unsigned short u; } c0, c1, c2, c3;
          c0.h = h0; c1.h = l0; c2.h = h1; c3.h = l1;
          __hip_atomic_store(gd, tg | c0.u | ((unsigned)c1.u << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(gd + 1, tg | c2.u | ((unsigned)c3.u << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *(float2*)(s.hout + (long)erow * s.ldo + ej) = float2{hv[0], hv[1]};
        *(unsigned*)(a.phi + o) = ph.u;
        *(unsigned*)(a.plo + o) = pl.u;
      } else {
      // publish: fp32 state (8 bytes) and its planes (4 + 4 bytes), all write-through
      union { float f[2]; unsigned long long u; } pk;
      pk.f[0] = hv[0]; pk.f[1] = hv[1];
      __hip_atomic_store((unsigned long long*)(s.hout + (long)erow * s.ldo + ej), pk.u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((unsigned*)(a.phi + o), ph.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((unsigned*)(a.plo + o), pl.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (st == T - 1 && a.r_off[dir] != kNoPlane) {        // relu(final state): the tail linear's A operand
        const long ro = (long)a.r_off[dir] + (long)(ej >> 5) * a.r_kst + plane_index(erow, ej & 31, 0);
        store_planes2(a.rhi + ro, a.rlo + ro, sq_relu(hv[0]), sq_relu(hv[1]));
      }
    }
    SEQ_STAMP(5);
    if (st + 1 < T) {
      if constexpr (GR) {
        __syncthreads();                                    // `red` is free for the next step
      } else {
      if (!(TEPOSE_SEQ_ABL & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
      SEQ_STAMP(6);
      __syncthreads();                                      // (also: `red` is free for the next step)
      if (tid == 0 && !(TEPOSE_SEQ_ABL & 8)) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      SEQ_STAMP(7);
      }
    }
  }
}

static hipError_t launch_gran(const GruSeqArgs& a, dim3 grid, hipStream_t s) {
  switch (a.Hp / 256) {
    case 1: hipLaunchKernelGGL((gru_seq_kernel<1, 1, true>), grid, dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL((gru_seq_kernel<1, 2, true>), grid, dim3(512), 0, s, a); break;
    case 3: hipLaunchKernelGGL((gru_seq_kernel<1, 3, true>), grid, dim3(512), 0, s, a); break;
    case 4: hipLaunchKernelGGL((gru_seq_kernel<1, 4, true>), grid, dim3(512), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int MT>
static hipError_t launch_mt(const GruSeqArgs& a, dim3 grid, hipStream_t s) {
  switch (a.Hp / 256) {
    case 1: hipLaunchKernelGGL((gru_seq_kernel<MT, 1>), grid, dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL((gru_seq_kernel<MT, 2>), grid, dim3(512), 0, s, a); break;
    case 3: hipLaunchKernelGGL((gru_seq_kernel<MT, 3>), grid, dim3(512), 0, s, a); break;
    case 4: hipLaunchKernelGGL((gru_seq_kernel<MT, 4>), grid, dim3(512), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// rows up to which the state travels as tagged granules (Options::seq_gran_max_m, at most the kernel's granule capacity; 0: never)
int gru_seq_gran_rows(const Options& o) { return o.seq_gran_max_m > (int)kSeqGranRows ? (int)kSeqGranRows : o.seq_gran_max_m; }

// CUs of the current device (Options::assume_cus > 0: planning on a machine without one -- tests/test_dispatch.py, tepose_select_kernels).  A device
// property, not configuration: asked once per device.
static int device_cus(const Options& o) {
  if (o.assume_cus > 0) return o.assume_cus;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  static int cached[64] = {};
  if (dev >= 0 && dev < 64 && __atomic_load_n(&cached[dev], __ATOMIC_RELAXED) > 0) return __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (dev >= 0 && dev < 64) __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
  return n;
}

// usable for this layer shape on this device?  Every workgroup must be resident at once (one per CU).
bool gru_seq_ok(int ndir, int M, int Hp, int T, const Options& o) {
  const int cus = device_cus(o);
  const int max_m = o.seq_max_m > 64 ? 64 : o.seq_max_m;
  return M >= 1 && M <= max_m && T >= 2 && T <= kSeqMaxT && gru_seq_shape_ok(Hp) && ndir >= 1 &&
         ndir <= 3 && ndir * (Hp / 16) <= cus;
}

hipError_t launch_gru_seq(const GruSeqArgs& a0, hipStream_t s, const Options& o) {
  GruSeqArgs a = a0;
#ifdef TEPOSE_SEQ_STAMPS
  a.stamps = (unsigned long long*)o.seq_stamp_ptr;        // device buffer of 3 * kSeqMaxT * 8 uint64 (tools/seq_stamps.py)
#else
  a.stamps = nullptr;
#endif
  if (!gru_seq_ok(a.ndir, a.M, a.Hp, a.T, o)) return hipErrorInvalidValue;
  dim3 grid(a.Hp / 16, 1, a.ndir);
  if (a.M <= gru_seq_gran_rows(o) && a.gran) return launch_gran(a, grid, s);
  if (a.M <= 16) return launch_mt<1>(a, grid, s);
  if (a.M <= 32) return launch_mt<2>(a, grid, s);
  if (a.M <= 48) return launch_mt<3>(a, grid, s);      // 33..48 rows (37 clips in lock-step): three tiles -- a quarter less state through the L2 port per step
  // 49..64 rows: four 16-row tiles per workgroup.  (Round 2 had an opt-in row split for 2-direction layers -- two workgroups per unit slice, 256
  // workgroups, -15 us per forward at B = 64 -- removed in round 5: it needs EVERY CU of the chip resident at once, which a second process's launch on
  // the same GPU can deny until the bounded wait expires.)
  return launch_mt<4>(a, grid, s);
}

}  // namespace tepose
