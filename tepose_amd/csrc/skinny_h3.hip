// Small-M GRU step and GEMM on the split-precision operands (fp16 hi / lo planes, three fp16 MFMAs per product, fp32
// accumulation; gemm_h3.hip explains the arithmetic): the 4 < B <= a few hundred regime of a split-mode handle
// (real-data evaluation with tens to hundreds of concurrent clips; the GEMM variant serves every product of <= 768 rows).
//
// At these batch sizes a GRU step is a weight-streaming problem (37.7 MB of W_hh planes per 3-direction step at
// H = 1024), the fp32 skinny kernel of skinny.hip is bound by the slow fp32 MFMA once M > 16, and the 64/128-row
// tiles of gemm_h3.hip leave most CUs idle (B = 64: 48 blocks).  Same cut as skinny.hip, for width: a block owns
// 16*MT rows x 16 hidden units x 3 gates, its 4 waves split K four ways and stream their slice of the A and W
// planes straight from global memory into VGPRs (one 16-byte load per lane = the lane's 8 k-values of a
// 16x16x32 MFMA operand: the blocked plane layout of common.h keeps a row's 32-wide K-tile in one 64-byte run),
// two K-tiles in flight, one LDS pass to add the 4 partial sums, then the cell update.
#include "common.h"

namespace tepose {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float sh_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float sh_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

// halfs from the start of a row's K-tile to the 8 k-values lane-quarter q reads (slot swizzle of plane_index)
__device__ __forceinline__ int slot_off(long row, int q) { return ((q ^ (int)((row >> 2) & 3)) << 3); }

template <int MT>
__global__ void __launch_bounds__(256) skinny_gru_h3_kernel(H3Batch batch, int M) {
  constexpr int NW = 4;
  __shared__ __attribute__((aligned(16))) float red[NW * MT * 3 * 256];
  const H3Args& a = batch.p[blockIdx.z];
  const GateDir& d = batch.gate[blockIdx.z];
  const int Hp = batch.Hp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = blockIdx.x * 16, m0 = blockIdx.y * 16 * MT;
  const int r16 = lane & 15, q = lane >> 4;

  // epilogue operands first: their latency hides under the weight stream below
  const int e = (threadIdx.x >> 6) & 3;          // accumulator register this thread finishes after the reduction
  const int j = j0 + r16;
  const float br = d.bhh[j], bz = d.bhh[Hp + j], bn = d.bhh[2 * Hp + j];
  float gr[MT], gz[MT], gn[MT], hp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = min(m0 + i * 16 + q * 4 + e, M - 1);
    const float* gi = d.gi + (long)row * d.ldgi + j;
    gr[i] = gi[0]; gz[i] = gi[Hp]; gn[i] = gi[2 * Hp];
    hp[i] = d.hprev[(long)row * d.ldh + j];
  }

  // operand pointers (halfs), K-tile 0: A rows of the hprev view, W rows in the gate-interleaved tile order
  const half_t *ah[MT], *al[MT], *wh[3], *wl[3];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const long row = min(m0 + i * 16 + r16, M - 1);
    const long o = row * 32 + slot_off(row, q);
    ah[i] = a.Ah + o; al[i] = a.Al + o;
  }
  const int rbase = (j0 >> 6) * 192 + ((j0 & 63) >> 5) * 96 + (j0 & 31);
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const long row = rbase + g * 32 + r16;
    const long o = row * 32 + slot_off(row, q);
    wh[g] = a.Wh + o; wl[g] = a.Wl + o;
  }
  const int KT = a.Kp / kPlaneK;
  const int c0 = (wave * KT) / NW, c1 = ((wave + 1) * KT) / NW;

  f32x4 acc[MT][3], accx[MT][3];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int g = 0; g < 3; ++g) { acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  struct Chunk { h16x8 ah[MT], al[MT], wh[3], wl[3]; };
  auto load = [&](int c, Chunk& k) {
    const long ao = (long)c * a.a_kst, wo = (long)c * a.w_kst;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      k.ah[i] = *(const h16x8*)(ah[i] + ao);
      k.al[i] = *(const h16x8*)(al[i] + ao);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      k.wh[g] = *(const h16x8*)(wh[g] + wo);
      k.wl[g] = *(const h16x8*)(wl[g] + wo);
    }
  };
  auto mma = [&](const Chunk& k) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k.ah[i], k.wh[g], acc[i][g], 0, 0, 0);
        accx[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k.ah[i], k.wl[g], accx[i][g], 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int g = 0; g < 3; ++g)
        accx[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k.al[i], k.wh[g], accx[i][g], 0, 0, 0);
  };
  if (c0 < c1) {
    Chunk p, n;
    load(c0, p);
    for (int c = c0; c < c1; c += 2) {
      const bool more = c + 1 < c1;
      if (more) load(c + 1, n);
      mma(p);
      if (more) {
        if (c + 2 < c1) load(c + 2, p);
        mma(n);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int ee = 0; ee < 4; ++ee)
        red[((wave * MT * 3 + i * 3 + g) * 4 + ee) * 64 + lane] = acc[i][g][ee] + accx[i][g][ee] * (1.f / kLoScale);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float hr = 0.f, hz = 0.f, hn = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      hr += red[((w * MT * 3 + i * 3 + 0) * 4 + e) * 64 + lane];
      hz += red[((w * MT * 3 + i * 3 + 1) * 4 + e) * 64 + lane];
      hn += red[((w * MT * 3 + i * 3 + 2) * 4 + e) * 64 + lane];
    }
    const int row = m0 + i * 16 + q * 4 + e;
    if (row < M) {
      const float rg = sh_sigmoid(gr[i] + (hr + br));
      const float zg = sh_sigmoid(gz[i] + (hz + bz));
      const float ng = sh_tanh(gn[i] + rg * (hn + bn));
      const float hv = (1.f - zg) * ng + zg * hp[i];
      d.hout[(long)row * d.ldo + j] = hv;
      const long o = (long)(j >> 5) * d.okst + plane_index(row, j & 31, 0);
      split_hi_lo(hv, d.hout_hi[o], d.hout_lo[o]);
    }
  }
}

// Small-M product C = (A W^T + bias + addend) * scale on the same operands (optionally also written as planes):
// a block owns 16*MT rows x 48 columns, K split over the 4 waves exactly as above.
// NT: 16-column tiles per block (3 = 48 columns; 1 = 16 columns for narrow products of few rows, where 48-column blocks
// would leave most CUs without a weight stream: N = 2048 -> 43 blocks vs 128)
// DEPTH: K-tiles a wave keeps in flight (chunk buffers in registers).  The narrow forms (NT = 1: a handful of blocks streaming a small weight matrix, each
// wave a chain of dependent round trips to the Infinity Cache) take 4: same arithmetic in the same order, half the round trips.
template <int MT, int NT = 3, int NW = 4, int DEPTH = 2>
__global__ void __launch_bounds__(64 * NW) skinny_gemm_h3_kernel(H3ArgsBatch batch) {
  const H3Args& a = batch.p[blockIdx.z];          // up to 3 independent products per launch (their own M, N, K)
  if ((int)blockIdx.x * (16 * NT) >= a.N || (int)blockIdx.y * 16 * MT >= a.M) return;   // the grid covers the largest one
  __shared__ __attribute__((aligned(16))) float red[NW * MT * NT * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * 16 * MT;
  const int r16 = lane & 15, q = lane >> 4;
  const long wrows = a.w_kst / 32;               // rows the W planes hold (padded to the 128-row tile, zero past N)

  const int gr = a.grp_rows, gs = a.grp_stride;
  auto phys = [&](int m) -> long { return gr ? (long)(m / gr) * gs + m % gr : (long)m; };
  const half_t *ah[MT], *al[MT], *wh[NT], *wl[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const long row = phys(min(m0 + i * 16 + r16, a.M - 1));
    const long o = row * 32 + slot_off(row, q);
    ah[i] = a.Ah + o; al[i] = a.Al + o;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const long row = min((long)(n0 + t * 16 + r16), wrows - 1);
    const long o = row * 32 + slot_off(row, q);
    wh[t] = a.Wh + o; wl[t] = a.Wl + o;
  }
  const int KT = a.Kp / kPlaneK;
  const int c0 = (wave * KT) / NW, c1 = ((wave + 1) * KT) / NW;

  f32x4 acc[MT][NT], accx[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  struct Chunk { h16x8 ah[MT], al[MT], wh[NT], wl[NT]; };
  auto load = [&](int c, Chunk& k) {
    const long ao = (long)c * a.a_kst, wo = (long)c * a.w_kst;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      k.ah[i] = *(const h16x8*)(ah[i] + ao);
      k.al[i] = *(const h16x8*)(al[i] + ao);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      k.wh[t] = *(const h16x8*)(wh[t] + wo);
      k.wl[t] = *(const h16x8*)(wl[t] + wo);
    }
  };
  auto mma = [&](const Chunk& k) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k.ah[i], k.wh[t], acc[i][t], 0, 0, 0);
        accx[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k.ah[i], k.wl[t], accx[i][t], 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        accx[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k.al[i], k.wh[t], accx[i][t], 0, 0, 0);
  };
  if constexpr (DEPTH == 2) {
  if (c0 < c1) {
    Chunk p, n;
    load(c0, p);
    for (int c = c0; c < c1; c += 2) {
      const bool more = c + 1 < c1;
      if (more) load(c + 1, n);
      mma(p);
      if (more) {
        if (c + 2 < c1) load(c + 2, p);
        mma(n);
      }
    }
  }
  } else {
    Chunk buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (c0 + d < c1) load(c0 + d, buf[d]);
    for (int c = c0; c < c1; c += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        if (c + d < c1) {
          mma(buf[d]);
          if (c + d + DEPTH < c1) load(c + d + DEPTH, buf[d]);
        }
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ee = 0; ee < 4; ++ee)
        red[((wave * MT * NT + i * NT + t) * 4 + ee) * 64 + lane] = acc[i][t][ee] + accx[i][t][ee] * (1.f / kLoScale);
  __syncthreads();
  if (NW > 4 && threadIdx.x >= 256) return;      // 4 accumulator registers: the first 256 threads finish one element each
  const int e = threadIdx.x >> 6;                // accumulator register of the element this thread finishes
  const float sc = a.scale != 0.f ? a.scale : 1.f;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[((w * MT * NT + i * NT + t) * 4 + e) * 64 + lane];
      const int lrow = m0 + i * 16 + q * 4 + e, col = n0 + t * 16 + r16;
      if (lrow < a.M && col < a.N) {
        const long row = phys(lrow);
        if (a.row_scale) v *= a.row_scale[row];
        if (a.bias) v += a.bias[col];
        if (a.addend) v += a.addend[row * a.ldadd + col];
        v *= sc;
        if (a.C2 && row >= a.c_split) a.C2[(row - a.c_split) * a.ldc2 + col] = v;
        else a.C[row * a.ldc + col] = v;
        if (a.Chi) {
          const long o = (long)(col >> 5) * a.c_kst + plane_index(row, col & 31, 0);
          split_hi_lo(v, a.Chi[o], a.Clo[o]);
        }
      }
    }
  }
}

hipError_t launch_skinny_gemm_h3_batch(const H3ArgsBatch& b, hipStream_t s, const Options& o) {
  int maxM = 0, maxN = 0;
  for (int i = 0; i < b.n; ++i) { maxM = b.p[i].M > maxM ? b.p[i].M : maxM; maxN = b.p[i].N > maxN ? b.p[i].N : maxN; }
  if (b.n <= 0 || maxM <= 0 || maxN <= 0) return hipSuccess;
  const int nt = (maxN + 47) / 48;
  const int nt1_max = o.skinny_nt1_below;
  {
    // narrow products (the collapsed regressor product: 160 columns = 10 blocks of 16; the 2048-column tail linears at <= 16 rows): a CU takes in ~50 GB/s,
    // so the launch lasts as long as its busiest block's bytes -- one 16-row tile per block (W re-read per row tile from L2, A never clamped-and-repeated)
    // puts them on rows x more CUs: 64 rows x 160 x 3072 on 40 blocks of 392 KB instead of 10 of 960 KB
    const bool mt1 = o.skinny_mt1 != 0;
    const int blocks16 = (maxN + 15) / 16 * b.n, rowtiles = (maxM + 15) / 16;
    if (mt1 && nt * b.n < nt1_max && blocks16 * rowtiles <= 256) {
      hipLaunchKernelGGL((skinny_gemm_h3_kernel<1, 1, 8, 4>), dim3((maxN + 15) / 16, rowtiles, b.n), dim3(512), 0, s, b);
      return hipGetLastError();
    }
  }
  if (maxM <= 32 && nt * b.n < nt1_max) {          // few rows, narrow product: 16-column blocks put a weight stream on 3x the CUs
    // (<= 128 such blocks -- the 2048-column tail linears: 8 waves split K, twice the weight bytes in flight per CU)
    const bool w8 = o.skinny_w8 != 0;
    if (w8 && (maxN + 15) / 16 * b.n <= 128)
      hipLaunchKernelGGL((skinny_gemm_h3_kernel<2, 1, 8, 4>), dim3((maxN + 15) / 16, 1, b.n), dim3(512), 0, s, b);
    else
      hipLaunchKernelGGL((skinny_gemm_h3_kernel<2, 1, 4, 4>), dim3((maxN + 15) / 16, 1, b.n), dim3(256), 0, s, b);
  } else if (maxM <= 64 && maxM > 32 && nt * b.n < 96 && o.skinny_narrow64 != 0) {
    // (as above: <= 128 blocks -- the collapsed regressor product, N = 160: 10 blocks -- split K over 8 waves: the block's chain of dependent weight
    // chunks halves)
    const bool w8 = o.skinny_w8 != 0;
    if (w8 && (maxN + 15) / 16 * b.n <= 128)
      hipLaunchKernelGGL((skinny_gemm_h3_kernel<4, 1, 8, 4>), dim3((maxN + 15) / 16, 1, b.n), dim3(512), 0, s, b);
    else
      hipLaunchKernelGGL((skinny_gemm_h3_kernel<4, 1, 4, 4>), dim3((maxN + 15) / 16, 1, b.n), dim3(256), 0, s, b);
  } else {
    // row tiles dealt evenly over the fewest passes of <= 4 tiles: a block never loads a clamped-and-repeated tile beyond the last one of the batch
    // (74 rows = 5 tiles: 3 + 2 instead of 4 + 1 and three repeats; 16 rows: 1 tile instead of 2)
    const int tiles = (maxM + 15) / 16, passes = (tiles + 3) / 4;
    const int mt = o.skinny_mt1 != 0 ? (tiles + passes - 1) / passes : (maxM <= 32 ? 2 : 4);
    const dim3 grid(nt, (tiles + mt - 1) / mt, b.n);
    if (mt <= 1) hipLaunchKernelGGL((skinny_gemm_h3_kernel<1>), grid, dim3(256), 0, s, b);
    else if (mt == 2) hipLaunchKernelGGL((skinny_gemm_h3_kernel<2>), grid, dim3(256), 0, s, b);
    else if (mt == 3) hipLaunchKernelGGL((skinny_gemm_h3_kernel<3>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((skinny_gemm_h3_kernel<4>), grid, dim3(256), 0, s, b);
  }
  return hipGetLastError();
}

hipError_t launch_skinny_gemm_h3(const H3Args& a, hipStream_t s, const Options& o) {
  H3ArgsBatch b{};
  b.p[0] = a; b.n = 1;
  return launch_skinny_gemm_h3_batch(b, s, o);
}

hipError_t launch_skinny_gru_h3(const H3Batch& b, hipStream_t s) {
  const int M = b.p[0].M;
  if (b.n <= 0 || M <= 0) return hipSuccess;
  const int jt = b.Hp / 16;
  if (M <= 32) {
    hipLaunchKernelGGL((skinny_gru_h3_kernel<2>), dim3(jt, 1, b.n), dim3(256), 0, s, b, M);
  } else {
    hipLaunchKernelGGL((skinny_gru_h3_kernel<4>), dim3(jt, (M + 63) / 64, b.n), dim3(256), 0, s, b, M);
  }
  return hipGetLastError();
}

}  // namespace tepose
