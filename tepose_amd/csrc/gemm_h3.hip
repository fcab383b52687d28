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

namespace tepose {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int HM = 256, HN = 128, HK = 32;
constexpr int H_ROWB = HK * 2;                       // bytes per tile row of one plane
constexpr int H_STAGE = (2 * HM + 2 * HN) * H_ROWB;  // 49152 B
constexpr int H_NSTAGE = 3;                          // ring of 3 stages = 144 KB: two K-tiles of DMA in flight

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
    const _Float16 h = (_Float16)a;
    hi[idx] = h;
    lo[idx] = (_Float16)(a - (float)h);
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

__global__ void __launch_bounds__(512, 2) gemm_h3_kernel(H3Batch batch, int tilesM, int tilesN) {
  const H3Args& a = batch.p[blockIdx.y];
  __shared__ __attribute__((aligned(16))) char lds[H_NSTAGE * H_STAGE];
  int tm, tn;
  h3_tile_of_block(blockIdx.x, gridDim.x, tilesM, tilesN, tm, tn);
  const int m0 = tm * HM, n0 = tn * HN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // ---- DMA: 48 wave-instructions per stage (16 A_hi, 16 A_lo, 8 W_hi, 8 W_lo), 6 per wave -----------
  const int lrow = lane >> 2, lslot = lane & 3;
  const char* gsrc[6];
  int ldst[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int i = wave * 6 + q;
    const char* base;
    long ldb;
    int row, grow;
    if (i < 32) {                       // A planes
      row = (i & 15) * 16 + lrow;
      grow = min(m0 + row, a.M - 1);
      base = (const char*)(i < 16 ? a.Ah : a.Al);
      ldb = a.lda * 2;
    } else {                            // W planes
      row = (i & 7) * 16 + lrow;
      grow = n0 + row;
      base = (const char*)(i < 40 ? a.Wh : a.Wl);
      ldb = (long)a.Kp * 2;
    }
    gsrc[q] = base + (long)grow * ldb + 16 * (lslot ^ ((row >> 2) & 3));
    ldst[q] = i * 1024;                 // stage layout is exactly the instruction order
  }
  auto issue = [&](int kt, int buf) {
    char* st = lds + buf * H_STAGE;
#pragma unroll
    for (int q = 0; q < 6; ++q) glds16b(gsrc[q] + (long)kt * H_ROWB, st + ldst[q]);
  };

  // ---- fragment offsets (bytes inside a stage) ---------------------------------------------------------
  const int sw = (r >> 2) & 3;
  int aoff[2], boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aoff[i] = (wm * 64 + i * 32 + r) * H_ROWB;
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = 2 * HM * H_ROWB + (wn * 64 + j * 32 + r) * H_ROWB;
  constexpr int A_LO = HM * H_ROWB, W_LO = HN * H_ROWB;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  struct Frags { h16x8 ah[2], al[2], bh[2], bl[2]; };
  auto load_frags = [&](const char* st, int s, Frags& f) {
    const int sx = 16 * ((2 * s + h) ^ sw);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f.ah[i] = *(const h16x8*)(st + aoff[i] + sx);
      f.al[i] = *(const h16x8*)(st + A_LO + aoff[i] + sx);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f.bh[j] = *(const h16x8*)(st + boff[j] + sx);
      f.bl[j] = *(const h16x8*)(st + W_LO + boff[j] + sx);
    }
  };
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
      }
  };

  // 3-stage ring, two K-tiles of LDS-DMA in flight.  Per K-tile: every wave waits until its own DMA
  // instructions of stage kt have landed (counted vmcnt: the 6 of stage kt+1 may stay outstanding), the
  // raw barrier then makes the whole stage visible and also proves that every wave is done reading stage
  // kt-1, whose slot the next DMA (kt+2) overwrites.  __syncthreads() would drain vmcnt(0) instead.
  const int KT = a.Kp / HK;
  Frags f0, f1;
  issue(0, 0);
  if (KT > 1) issue(1, 1);
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < KT) issue(kt + 2, (kt + 2) % H_NSTAGE);
    const char* st = lds + (kt % H_NSTAGE) * H_STAGE;
    load_frags(st, 0, f0);
    load_frags(st, 1, f1);
    mma(f0);
    mma(f1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + r;
    if (col >= a.N) continue;
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < a.M) a.C[(long)row * a.ldc + col] = acc[i][j][e] + bv;
      }
    }
  }
}

// test / bench entry: fp32 A[M,K], W[N,K] -> planes in `ws` -> C (K multiple of 32)
hipError_t launch_gemm_h3_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C,
                              long ldc, int M, int N, int K, void* ws, hipStream_t s) {
  const int Np = round_up(N, 128);
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
  const int tilesM = (b.p[0].M + HM - 1) / HM, tilesN = (b.p[0].N + HN - 1) / HN;
  hipLaunchKernelGGL(gemm_h3_kernel, dim3(tilesM * tilesN, b.n), dim3(512), 0, s, b, tilesM, tilesN);
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
    const _Float16 h = (_Float16)a;
    hi[idx] = h;
    lo[idx] = (_Float16)(a - (float)h);
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
__device__ __forceinline__ float g_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float g_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}
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
    const _Float16 hh = (_Float16)hv;
    d.hout_hi[row * d.ldo + j] = hh;
    d.hout_lo[row * d.ldo + j] = (_Float16)(hv - (float)hh);
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
  return 2 * align_up((size_t)M * K * 2, 256) + 2 * align_up((size_t)round_up(N, 128) * K * 2, 256) + 256;
}

}  // namespace tepose
