// Small-M ("skinny") GEMM and GRU step: the latency / weight-streaming regime
// (live-stream B = 1, real-data evaluation B = #concurrent clips; BASELINE.json configs 1, 2, 5).
//
// With M <= a few hundred rows the 128-row tiles of gemm.hip leave most CUs idle and walk K
// serially (a B=64 GRU step ran 16 blocks per direction for 93 us).  Here the work is cut
// for *width*: a block owns 16*MT rows x 48 columns (GRU: 16 hidden units x 3 gates), so a
// 3-direction step is 192 blocks; the 4 waves of a block split K four ways, stream their
// slice of A and W straight from global memory into VGPRs (no LDS round trip: each weight
// byte is used by exactly one wave), keep two super-chunks in flight, and meet once in LDS
// to add their partial sums before the epilogue.  MFMA shape 16x16x4 (f32 in, exact):
// A lane l holds A[l&15][k = 4*(l>>4)+m] for its 16-byte load, register m feeds MFMA #m.
#include "common.h"

namespace tepose {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sk_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float sk_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

// MT: 16-row tiles per block (1, 2, 4).  U: 16-k chunks per super-chunk (loads issued together).
template <int MT, int U>
struct SkinnyCore {
  f32x4 acc[MT][3];

  __device__ __forceinline__ void run(const float* const (&ap)[MT], const float* const (&wp)[3], int c0,
                                      int c1, bool relu) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 pa[U][MT], pw[U][3], qa[U][MT], qw[U][3];
    auto load = [&](int c, f32x4 (&A)[U][MT], f32x4 (&W)[U][3]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = c + u < c1;
        const int k = (c + u) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) A[u][i] = ok ? *(const f32x4*)(ap[i] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) W[u][t] = ok ? *(const f32x4*)(wp[t] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto mma = [&](f32x4 (&A)[U][MT], f32x4 (&W)[U][3]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (relu) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) A[u][i][m] = fmaxf(A[u][i][m], 0.f);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t)
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u][i][m], W[u][t][m], acc[i][t], 0, 0, 0);
      }
    };
    if (c0 >= c1) return;
    load(c0, pa, pw);
    for (int c = c0; c < c1; c += 2 * U) {
      const bool more = c + U < c1;
      if (more) load(c + U, qa, qw);
      mma(pa, pw);
      if (more) {
        if (c + 2 * U < c1) load(c + 2 * U, pa, pw);
        mma(qa, qw);
      }
    }
  }
};

#ifndef TEPOSE_SK_NW1
#define TEPOSE_SK_NW1 4
#endif
#ifndef TEPOSE_SK_U4
#define TEPOSE_SK_U4 1
#endif
#ifndef TEPOSE_SK_U2
#define TEPOSE_SK_U2 2
#endif
constexpr int SK_U4 = TEPOSE_SK_U4;     // 16-k chunks per super-chunk of the MT=4 kernels (two super-chunks in flight)
constexpr int SK_U2 = TEPOSE_SK_U2;     // same for MT=2
constexpr int SK_NW1 = TEPOSE_SK_NW1;   // waves per block of the M <= 16 kernels (K split that many ways)

struct SkinnyGemmArgs {
  GemmArgs g;
  int n_alloc;   // packed W rows that may be read (round_up(N,128))
};

template <int MT, int U, int NW>
__global__ void __launch_bounds__(64 * NW) skinny_gemm_kernel(SkinnyGemmArgs sa) {
  __shared__ __attribute__((aligned(16))) float red[NW * MT * 3 * 256];
  const GemmArgs& a = sa.g;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 48, m0 = blockIdx.y * 16 * MT;
  const int r16 = lane & 15, q = lane >> 4;
  const float* ap[MT];
  const float* wp[3];
#pragma unroll
  for (int i = 0; i < MT; ++i) ap[i] = a.A + (long)min(m0 + i * 16 + r16, a.M - 1) * a.lda + 4 * q;
#pragma unroll
  for (int t = 0; t < 3; ++t) wp[t] = a.W + (long)min(n0 + t * 16 + r16, sa.n_alloc - 1) * a.Kp + 4 * q;
  const int NC = a.Kp / 16;
  SkinnyCore<MT, U> core;
  core.run(ap, wp, (wave * NC) / NW, ((wave + 1) * NC) / NW, a.relu_a != 0);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[((wave * MT * 3 + i * 3 + t) * 4 + e) * 64 + lane] = core.acc[i][t][e];
  __syncthreads();
  if (NW > 4 && threadIdx.x >= 256) return;   // 4 x 64 threads finish the 16x16 tiles
  const int e = threadIdx.x >> 6;          // accumulator register of the element this thread finishes
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[((w * MT * 3 + i * 3 + t) * 4 + e) * 64 + lane];
      const int row = m0 + i * 16 + q * 4 + e, col = n0 + t * 16 + r16;
      if (row < a.M && col < a.N) {
        if (a.bias) v += a.bias[col];
        if (a.addend) v += a.addend[(long)row * a.ldadd + col];
        a.C[(long)row * a.ldc + col] = v * a.scale;
      }
    }
  }
}

template <int MT, int U, int NW>
__global__ void __launch_bounds__(64 * NW) skinny_gru_kernel(GruArgs a) {
  __shared__ __attribute__((aligned(16))) float red[NW * MT * 3 * 256];
  const GruDir& d = a.d[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * MT;
  const int r16 = lane & 15, q = lane >> 4;
  const int Hp = a.Hp;
  // epilogue operands first: their latency hides under the weight stream below
  const int e = (threadIdx.x >> 6) & 3;
  const int j = j0 + r16;
  const float br = d.bhh[j], bz = d.bhh[Hp + j], bn = d.bhh[2 * Hp + j];
  float gr[MT], gz[MT], gn[MT], hp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = min(m0 + i * 16 + q * 4 + e, a.M - 1);
    const float* gi = d.gi + (long)row * d.ldgi + j;
    gr[i] = gi[0]; gz[i] = gi[Hp]; gn[i] = gi[2 * Hp];
    hp[i] = a.first ? 0.f : d.hprev[(long)row * d.ldh + j];
  }
  if (!a.first) {
    const float* ap[MT];
    const float* wp[3];
#pragma unroll
    for (int i = 0; i < MT; ++i) ap[i] = d.hprev + (long)min(m0 + i * 16 + r16, a.M - 1) * d.ldh + 4 * q;
    // W_hh rows live in the gate-interleaved tile order of pack_kernel(ROW_GATES_TILED)
    const int rbase = (j0 >> 6) * 192 + ((j0 & 63) >> 5) * 96 + (j0 & 31);
#pragma unroll
    for (int g = 0; g < 3; ++g) wp[g] = d.Whh + (long)(rbase + g * 32 + r16) * Hp + 4 * q;
    const int NC = Hp / 16;
    SkinnyCore<MT, U> core;
    core.run(ap, wp, (wave * NC) / NW, ((wave + 1) * NC) / NW, false);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) red[((wave * MT * 3 + i * 3 + g) * 4 + ee) * 64 + lane] = core.acc[i][g][ee];
  }
  __syncthreads();
  if (NW > 4 && threadIdx.x >= 256) return;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float hr = 0.f, hz = 0.f, hn = 0.f;
    if (!a.first) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        hr += red[((w * MT * 3 + i * 3 + 0) * 4 + e) * 64 + lane];
        hz += red[((w * MT * 3 + i * 3 + 1) * 4 + e) * 64 + lane];
        hn += red[((w * MT * 3 + i * 3 + 2) * 4 + e) * 64 + lane];
      }
    }
    const int row = m0 + i * 16 + q * 4 + e;
    if (row < a.M) {
      const float rg = sk_sigmoid(gr[i] + (hr + br));
      const float zg = sk_sigmoid(gz[i] + (hz + bz));
      const float ng = sk_tanh(gn[i] + rg * (hn + bn));
      d.hout[(long)row * d.ldo + j] = (1.f - zg) * ng + zg * hp[i];
    }
  }
}

hipError_t launch_skinny_gemm(const GemmArgs& g, hipStream_t s) {
  SkinnyGemmArgs sa{g, round_up(g.N, 128)};
  const int nt = (g.N + 47) / 48;
  if (g.M <= 16) {
    hipLaunchKernelGGL((skinny_gemm_kernel<1, 4, SK_NW1>), dim3(nt, 1), dim3(64 * SK_NW1), 0, s, sa);
  } else if (g.M <= 32) {
    hipLaunchKernelGGL((skinny_gemm_kernel<2, SK_U2, 4>), dim3(nt, 1), dim3(256), 0, s, sa);
  } else {
    hipLaunchKernelGGL((skinny_gemm_kernel<4, SK_U4, 4>), dim3(nt, (g.M + 63) / 64), dim3(256), 0, s, sa);
  }
  return hipGetLastError();
}

hipError_t launch_skinny_gru(const GruArgs& a, hipStream_t s) {
  const int jt = a.Hp / 16;
  if (a.M <= 16) {
    hipLaunchKernelGGL((skinny_gru_kernel<1, 4, SK_NW1>), dim3(jt, 1, a.ndir), dim3(64 * SK_NW1), 0, s, a);
  } else if (a.M <= 32) {
    hipLaunchKernelGGL((skinny_gru_kernel<2, SK_U2, 4>), dim3(jt, 1, a.ndir), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((skinny_gru_kernel<4, SK_U4, 4>), dim3(jt, (a.M + 63) / 64, a.ndir), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace tepose
