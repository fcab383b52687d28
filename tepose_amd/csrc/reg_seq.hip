// Persistent kernel for the iterative regressor of small batches (N <= 64 rows): Regressor.forward's FC loop
// (lib/models/spin.py:250-261) -- fc1, fc2 and the three decoders, n_iter times -- in ONE launch.
//
// Step-per-launch this is 1 + 3 n_iter dependent products of at most 64 rows (plus state initialisation and plane
// conversion): ten launches of 7-15 us that keep 4-22 CUs busy each (74 us of the B = 1 forward, 190 us at B = 64).
// Here 64 workgroups each own 16 of the 1024 hidden columns:
//   base  = feat W1a^T + b1                 (own 16 columns, K = 2048: W1a streamed once, result stays in registers)
//   loop: h1 = base + xs W1b^T              (K = 160;  needs every column of xs)
//         h2 = h1 W2^T + b2                 (K = 1024; needs every column of h1)
//         xs = xs + h2 Wdec^T + bdec        (workgroups 0..9 own 16 of the 160 state columns; needs every column of h2)
// W2 / Wdec / W1b slices stay in registers for all iterations; h1, h2 and xs travel between workgroups as hi / lo planes
// through L2 with the hand-off protocol of gru_seq.hip (write-through stores, drained, one arrival counter per edge,
// sc1 loads).  Same arithmetic as the step-per-launch path: three fp16 MFMAs per product on the hi / lo halves, fp32
// accumulation, operands split exactly once where they are produced.  33..64 rows: two 32-row slices (grid.y = 2, 128
// workgroups), each an independent chain with its own counters.
#include "common.h"

namespace tepose {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ int rs_slot(long row, int q) { return ((q ^ (int)((row >> 2) & 3)) << 3); }
__device__ __forceinline__ h16x8 rs_h8(u32x4 v) {
  union { u32x4 u; h16x8 h; } c;
  c.u = v;
  return c.h;
}

// wait until *counter >= want (lane 0 of the workgroup polls, everybody leaves through the barrier).  A wait that gives up
// sets the forward's status word (the final store turns the state NaN; other waits see it and end at once) and the
// handle's host-visible fault word (tepose_status / the next call return TEPOSE_E_TIMEOUT).
__device__ __forceinline__ void rs_wait(unsigned* counter, unsigned want, const RegSeqArgs& a, int tid) {
  if (tid == 0) {
    unsigned spins = 0;
    want += a.inject;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      if ((spins & 1023u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
      if (spins > a.spin_limit) {                         // ~2 s by default: give up
        __hip_atomic_store(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.fault) __hip_atomic_store(a.fault, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void rs_arrive(unsigned* counter, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// publish two consecutive columns of one row as planes, write-through
__device__ __forceinline__ void rs_publish2(half_t* hi, half_t* lo, float v0, float v1) {
  half_t h0, l0, h1, l1;
  split_hi_lo(v0, h0, l0);
  split_hi_lo(v1, h1, l1);
  union { h16x2 h; unsigned u; } ph, pl;
  ph.h = h16x2{h0, h1}; pl.h = h16x2{l0, l1};
  __hip_atomic_store((unsigned*)hi, ph.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((unsigned*)lo, pl.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

template <int MT>
__global__ void __launch_bounds__(512) reg_seq_kernel(RegSeqArgs a) {
  constexpr int NW = 8;
  __shared__ __attribute__((aligned(16))) float red[NW * MT * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int blk = blockIdx.x, n0 = blk * 16;            // own hidden columns [n0, n0 + 16)
  const int N = a.N;
  const bool dec = blk < 10;                            // owns state columns [n0, n0 + 16) of the 160 as well

  // epilogue item: (row, column pair)
  const bool item = tid < MT * 128;
  const int ei = tid >> 7, err = (tid >> 3) & 15, ep = tid & 7;
  const int m0 = blockIdx.y * (MT * 16);               // row slices are independent chains with their own counters
  const int erow = m0 + ei * 16 + err, ec = n0 + 2 * ep;
  const bool live = item && erow < N;
  const int rq = err >> 2, re = err & 3;
  const float* rbase = red + (ei * 4 + re) * 64 + rq * 16 + 2 * ep;

  // K-partial sums of the 8 waves -> this thread's (row, 2 columns)
  auto reduce = [&](f32x4 (&acc)[MT], f32x4 (&accx)[MT]) -> float2 {
    __syncthreads();                                     // previous phase's readers are done with `red`
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) red[((wave * MT + i) * 4 + ee) * 64 + lane] = acc[i][ee] + accx[i][ee] * (1.f / kLoScale);
    __syncthreads();
    float2 v = {0.f, 0.f};
    if (live) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float2 t = *(const float2*)(rbase + w * MT * 256);
        v.x += t.x; v.y += t.y;
      }
    }
    return v;
  };
  auto zero = [&](f32x4 (&acc)[MT], f32x4 (&accx)[MT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  };
  auto mma = [&](const h16x8& ah, const h16x8& al, const h16x8& wh, const h16x8& wl, f32x4& acc, f32x4& accx) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, acc, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, accx, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, accx, 0, 0, 0);
  };

  // ---- stationary weight slices: rows n0 + r16 of W2 (4 K-tiles per wave), of Wdec (workgroups 0..9) and one K-tile
  // of W1b (waves 0..4)
  const long wrow = n0 + r16;
  const long wo = wrow * 32 + rs_slot(wrow, q);
  h16x8 w2h[4], w2l[4], wdh[4], wdl[4], w1bh, w1bl;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const long ko = (long)(wave * 4 + c) * (1024 * 32);
    w2h[c] = *(const h16x8*)(a.w2_h + wo + ko);
    w2l[c] = *(const h16x8*)(a.w2_l + wo + ko);
  }
  if (dec) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long ko = (long)(wave * 4 + c) * (256 * 32);
      wdh[c] = *(const h16x8*)(a.wd_h + wo + ko);
      wdl[c] = *(const h16x8*)(a.wd_l + wo + ko);
    }
  }
  if (wave < 5) {
    const long ko = (long)wave * (1024 * 32);
    w1bh = *(const h16x8*)(a.w1b_h + wo + ko);
    w1bl = *(const h16x8*)(a.w1b_l + wo + ko);
  }

  // ---- base = feat W1a^T + b1 for the own columns: K = 2048 = 8 K-tiles per wave, W1a streamed
  float2 base = {0.f, 0.f};
  {
    f32x4 acc[MT], accx[MT];
    zero(acc, accx);
#pragma unroll 2
    for (int c = 0; c < 8; ++c) {
      const int kt = wave * 8 + c;
      const h16x8 wh = *(const h16x8*)(a.w1a_h + wo + (long)kt * (1024 * 32));
      const h16x8 wl = *(const h16x8*)(a.w1a_l + wo + (long)kt * (1024 * 32));
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const long row = min(m0 + i * 16 + r16, N - 1);
        const long o = row * 32 + rs_slot(row, q) + (long)kt * a.f_kst;
        mma(*(const h16x8*)(a.fh + o), *(const h16x8*)(a.fl + o), wh, wl, acc[i], accx[i]);
      }
    }
    base = reduce(acc, accx);
    if (live) { base.x += a.b1[ec]; base.y += a.b1[ec + 1]; }
  }
  // initial state of a row: the caller's init_pose / init_shape / init_cam rows where given, else the model's means
  auto init_val = [&](int row, int col) -> float {
    if (col < kNPose) return a.ipose ? a.ipose[(long)row * kNPose + col] : a.init160[col];
    if (col < kNPose + 10) return a.ishape ? a.ishape[(long)row * 10 + (col - kNPose)] : a.init160[col];
    if (col < kNPose + 13) return a.icam ? a.icam[(long)row * 3 + (col - kNPose - 10)] : a.init160[col];
    return 0.f;
  };
  float2 b2v = {0.f, 0.f}, bdv = {0.f, 0.f}, xs = {0.f, 0.f};
  if (live) {
    b2v = float2{a.b2[ec], a.b2[ec + 1]};
    if (dec) {
      bdv = float2{a.bdec[ec], a.bdec[ec + 1]};
      xs = float2{init_val(erow, ec), init_val(erow, ec + 1)};
    }
  }
  // initial state as planes (every workgroup needs all 160 columns for its first h1): published like every later state
  if (dec && live) {
    const long o = (long)(ec >> 5) * a.x_kst + plane_index(erow, ec & 31, 0);
    rs_publish2(a.xh + o, a.xl + o, xs.x, xs.y);
  }
  unsigned* c_xs = a.counters + blockIdx.y * 8;          // arrivals: 10 per state version (per row slice)
  unsigned* c_h1 = a.counters + 32 + blockIdx.y * 8;     // 64 per iteration
  unsigned* c_h2 = a.counters + 64 + blockIdx.y * 8;
  if (dec) rs_arrive(c_xs, tid);

  __amdgpu_buffer_rsrc_t r_xh = __builtin_amdgcn_make_buffer_rsrc((void*)a.xh, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t r_xl = __builtin_amdgcn_make_buffer_rsrc((void*)a.xl, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t r_1h = __builtin_amdgcn_make_buffer_rsrc((void*)a.h1h, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t r_1l = __builtin_amdgcn_make_buffer_rsrc((void*)a.h1l, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t r_2h = __builtin_amdgcn_make_buffer_rsrc((void*)a.h2h, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t r_2l = __builtin_amdgcn_make_buffer_rsrc((void*)a.h2l, 0, 0x7fffffff, 0x00020000);

  for (int it = 0; it < a.n_iter; ++it) {
    // ---- h1 = base + xs W1b^T (K = 160: waves 0..4 one K-tile each)
    rs_wait(c_xs, 10u * (unsigned)(it + 1), a, tid);
    {
      f32x4 acc[MT], accx[MT];
      zero(acc, accx);
      if (wave < 5) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const long row = min(m0 + i * 16 + r16, N - 1);
          const unsigned o = (unsigned)(row * 32 + rs_slot(row, q) + (long)wave * a.x_kst) * 2u;
          mma(rs_h8(__builtin_amdgcn_raw_buffer_load_b128(r_xh, o, 0, 16)),
              rs_h8(__builtin_amdgcn_raw_buffer_load_b128(r_xl, o, 0, 16)), w1bh, w1bl, acc[i], accx[i]);
        }
      }
      const float2 v = reduce(acc, accx);
      if (live) {
        const long o = (long)(ec >> 5) * a.h_kst + plane_index(erow, ec & 31, 0);
        rs_publish2(a.h1h + o, a.h1l + o, v.x + base.x, v.y + base.y);
      }
    }
    rs_arrive(c_h1, tid);
    // ---- h2 = h1 W2^T + b2 (K = 1024: 4 K-tiles per wave)
    rs_wait(c_h1, 64u * (unsigned)(it + 1), a, tid);
    {
      f32x4 acc[MT], accx[MT];
      zero(acc, accx);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const long row = min(m0 + i * 16 + r16, N - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned o = (unsigned)(row * 32 + rs_slot(row, q) + (long)(wave * 4 + c) * a.h_kst) * 2u;
          mma(rs_h8(__builtin_amdgcn_raw_buffer_load_b128(r_1h, o, 0, 16)),
              rs_h8(__builtin_amdgcn_raw_buffer_load_b128(r_1l, o, 0, 16)), w2h[c], w2l[c], acc[i], accx[i]);
        }
      }
      const float2 v = reduce(acc, accx);
      if (live) {
        const long o = (long)(ec >> 5) * a.h_kst + plane_index(erow, ec & 31, 0);
        rs_publish2(a.h2h + o, a.h2l + o, v.x + b2v.x, v.y + b2v.y);
      }
    }
    rs_arrive(c_h2, tid);
    // ---- xs += h2 Wdec^T + bdec (workgroups 0..9)
    if (dec) {
      rs_wait(c_h2, 64u * (unsigned)(it + 1), a, tid);
      f32x4 acc[MT], accx[MT];
      zero(acc, accx);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const long row = min(m0 + i * 16 + r16, N - 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned o = (unsigned)(row * 32 + rs_slot(row, q) + (long)(wave * 4 + c) * a.h_kst) * 2u;
          mma(rs_h8(__builtin_amdgcn_raw_buffer_load_b128(r_2h, o, 0, 16)),
              rs_h8(__builtin_amdgcn_raw_buffer_load_b128(r_2l, o, 0, 16)), wdh[c], wdl[c], acc[i], accx[i]);
        }
      }
      const float2 v = reduce(acc, accx);
      if (live) {
        xs.x += v.x + bdv.x; xs.y += v.y + bdv.y;
        if (it + 1 < a.n_iter) {
          const long o = (long)(ec >> 5) * a.x_kst + plane_index(erow, ec & 31, 0);
          rs_publish2(a.xh + o, a.xl + o, xs.x, xs.y);
        }
      }
      if (it + 1 < a.n_iter) rs_arrive(c_xs, tid);
    }
  }
  // final state for the SMPL kernels (next launch: plain store); the 3 pad columns stay 0.  If any wait of this launch
  // gave up (status word set by whichever workgroup timed out), the state is NaN rather than plausible and wrong.
  if (dec && live) {
    if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) xs.x = xs.y = __builtin_nanf("");
    float* o = a.xs + (long)erow * kState + ec;
    o[0] = ec < kNPose + 13 ? xs.x : 0.f;
    o[1] = ec + 1 < kNPose + 13 ? xs.y : 0.f;
  }
}

hipError_t launch_reg_seq(const RegSeqArgs& a, hipStream_t s) {
  if (a.N < 1 || a.N > 64) return hipErrorInvalidValue;
  if (a.N <= 16) hipLaunchKernelGGL((reg_seq_kernel<1>), dim3(64), dim3(512), 0, s, a);
  else if (a.N <= 32) hipLaunchKernelGGL((reg_seq_kernel<2>), dim3(64), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((reg_seq_kernel<2>), dim3(64, 2), dim3(512), 0, s, a);   // 33..64 rows: two 32-row slices, 128 workgroups
  return hipGetLastError();
}

}  // namespace tepose
