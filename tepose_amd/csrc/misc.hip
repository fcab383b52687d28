// Weight packing, input padding and small fills.  All HBM-bound element-wise kernels:
// one dword per lane, consecutive lanes on consecutive addresses, grid-stride loops.
#include "common.h"

namespace tepose {

// dst[np][kp] = src[rowmap(np)][colmap(kp)] or 0 (see RowMap / ColMap in common.h).
//   ROW_GATES        np = g*Hp + j                      -> source row g*H + j
//   ROW_GATES_TILED  np = jt*192 + wn*96 + g*32 + jj    -> j = jt*64 + wn*32 + jj, row g*H + j
//                    (the order gru_step_kernel's waves consume W_hh in)
//   COL_SPLIT2       kp < Hp -> k = kp ; else k = H + (kp - Hp)   (bi-GRU concat input)
__global__ void __launch_bounds__(256) pack_kernel(PackArgs a) {
  const long total = (long)a.Np * a.Kp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int np = (int)(idx / a.Kp), kp = (int)(idx - (long)np * a.Kp);
    int n = -1, k = -1;
    if (a.rowmap == ROW_PLAIN) {
      if (np < a.N) n = np;
    } else if (a.rowmap == ROW_GATES) {
      const int g = np / a.Hp, j = np - g * a.Hp;
      if (g < 3 && j < a.H) n = g * a.H + j;
    } else {
      const int jt = np / 192, rem = np - jt * 192;
      const int wn = rem / 96, g = (rem % 96) / 32, jj = rem & 31;
      const int j = jt * 64 + wn * 32 + jj;
      if (j < a.H) n = g * a.H + j;
    }
    if (a.colmap == COL_PLAIN) {
      if (kp < a.K) k = kp;
    } else {
      if (kp < a.Hp) {
        if (kp < a.H) k = kp;
      } else if (kp - a.Hp < a.H) {
        k = a.H + (kp - a.Hp);
      }
    }
    const float v = (n >= 0 && k >= 0) ? a.src[(long)n * a.ld_src + k] : 0.f;
    if (a.dst) {
      a.dst[idx] = v;
    } else {                              // blocked hi / lo fp16 planes for the split-precision GEMM
      const long o = (long)(kp >> 5) * a.dst_kst + plane_index(np, kp & 31, 0);   // dst row offset % 16 == 0
      split_hi_lo(v, a.dst_hi[o], a.dst_lo[o]);
    }
  }
}

hipError_t launch_pack(const PackArgs& a, hipStream_t s) {
  const long total = (long)a.Np * a.Kp;
  if (total <= 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

// x[rows][2133] (rows only 4-byte aligned) -> xp[rows][2144], pad columns zero, so the
// GEMM's 16-byte LDS-DMA can read it.
__global__ void __launch_bounds__(256) pad_input_kernel(const float* __restrict__ x,
                                                        float* __restrict__ xp, long rows) {
  const long total = rows * kInputP;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / kInputP;
    const int k = (int)(idx - row * kInputP);
    xp[idx] = k < kInput ? x[row * kInput + k] : 0.f;
  }
}

hipError_t launch_pad_input(const float* x, float* xp, long rows, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const long total = rows * kInputP;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(pad_input_kernel, dim3(blocks), dim3(256), 0, s, x, xp, rows);
  return hipGetLastError();
}

// One padded GEMM row per clip from separately strided feature / theta rows (theta == nullptr ->
// zeros: the newest frame of a window, evaluate.py:248-252).
__global__ void __launch_bounds__(256) pad_rows_kernel(const float* __restrict__ feat, long feat_ld,
                                                       const float* __restrict__ theta, long theta_ld,
                                                       float* __restrict__ xp, long rows) {
  const long total = rows * kInputP;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / kInputP;
    const int k = (int)(idx - row * kInputP);
    float v = 0.f;
    if (k < kFeat) v = feat[row * feat_ld + k];
    else if (k < kInput && theta) v = theta[row * theta_ld + (k - kFeat)];
    xp[idx] = v;
  }
}

hipError_t launch_pad_rows(const float* feat, long feat_ld, const float* theta, long theta_ld, float* xp,
                           long rows, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const long total = rows * kInputP;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks), dim3(256), 0, s, feat, feat_ld, theta, theta_ld, xp, rows);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) fill_kernel(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}

hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
  return hipGetLastError();
}

// xs[n][0..159] = init160 (init_pose | init_shape | init_cam | 0 0 0), spin.py:243-248
__global__ void __launch_bounds__(256) init_state_kernel(const float* __restrict__ init160,
                                                         float* __restrict__ xs, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x)
    xs[idx] = init160[idx % kState];
}

// per-call initial state (Regressor.forward(init_pose=, init_shape=, init_cam=), spin.py:240-251): each part comes
// from the caller's [N, 144 | 10 | 3] rows when given, else from the model's mean-parameter buffers
__global__ void __launch_bounds__(256) init_state_rows_kernel(const float* __restrict__ init160,
                                                              const float* __restrict__ pose,
                                                              const float* __restrict__ shape,
                                                              const float* __restrict__ cam, float* __restrict__ xs,
                                                              long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long n = idx / kState;
    const int c = (int)(idx % kState);
    float v = init160[c];
    if (c < kNPose) { if (pose) v = pose[n * kNPose + c]; }
    else if (c < kNPose + 10) { if (shape) v = shape[n * 10 + (c - kNPose)]; }
    else if (c < kNPose + 13) { if (cam) v = cam[n * 3 + (c - kNPose - 10)]; }
    xs[idx] = v;
  }
}

hipError_t launch_init_state_rows(const float* init160, const float* pose, const float* shape, const float* cam,
                                  float* xs, int N, hipStream_t s) {
  const long total = (long)N * kState;
  if (total <= 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(init_state_rows_kernel, dim3(blocks), dim3(256), 0, s, init160, pose, shape, cam, xs, total);
  return hipGetLastError();
}

// dst[r][c] = src[r][c] (+ add[r][c]) for c < cols: a padded state buffer out as the caller's unpadded rows
__global__ void __launch_bounds__(256) copy_cols_kernel(const float* __restrict__ src, long lds, const float* __restrict__ add,
                                                        long lda, float* __restrict__ dst, long ldd, long rows, int cols) {
  const long total = rows * cols;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / cols;
    const int c = (int)(idx - r * cols);
    float v = src[r * lds + c];
    if (add) v += add[r * lda + c];
    dst[r * ldd + c] = v;
  }
}

hipError_t launch_copy_cols(const float* src, long lds, const float* add, long lda, float* dst, long ldd, long rows,
                            int cols, hipStream_t s) {
  const long total = rows * cols;
  if (total <= 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks), dim3(256), 0, s, src, lds, add, lda, dst, ldd, rows, cols);
  return hipGetLastError();
}

// ---- pack-time fp64 algebra (the collapsed regressor, api.hip): small dense products, one thread per output element.
// C[i][j] = alpha * sum_k A[i][k] * B[k][j] (+ 1 on the diagonal) (+ Cadd[i][j]) (+ Cadd32[i][j]); A / B are fp32 or fp64, row-major
// with their own leading dimensions.  A few hundred MFLOP per pack: speed is irrelevant, fp64 accumulation is the point.
__global__ void __launch_bounds__(256) dmm_kernel(const void* __restrict__ A, int a64, long lda, const void* __restrict__ B,
                                                  int b64, long ldb, const double* __restrict__ Cadd, long ldadd,
                                                  const float* __restrict__ Cadd32, long ldadd32,
                                                  double* __restrict__ C, long ldc, int M, int N, int K, double alpha,
                                                  int add_identity) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)M * N) return;
  const int i = (int)(idx / N), j = (int)(idx - (long)i * N);
  double acc = 0.0;
  for (int k = 0; k < K; ++k) {
    const double a = a64 ? ((const double*)A)[i * lda + k] : (double)((const float*)A)[i * lda + k];
    const double b = b64 ? ((const double*)B)[k * ldb + j] : (double)((const float*)B)[k * ldb + j];
    acc += a * b;
  }
  acc *= alpha;
  if (add_identity && i == j) acc += 1.0;
  if (Cadd) acc += Cadd[i * ldadd + j];
  if (Cadd32) acc += (double)Cadd32[i * ldadd32 + j];
  C[i * ldc + j] = acc;
}

hipError_t launch_dmm(const void* A, int a64, long lda, const void* B, int b64, long ldb, const double* Cadd, long ldadd,
                      const float* Cadd32, long ldadd32, double* C, long ldc, int M, int N, int K, double alpha,
                      int add_identity, hipStream_t s) {
  const long total = (long)M * N;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(dmm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A, a64, lda, B, b64, ldb, Cadd, ldadd,
                     Cadd32, ldadd32, C, ldc, M, N, K, alpha, add_identity);
  return hipGetLastError();
}

// fp64 [rows][cols] (leading dimension ld) -> zero-padded fp32 [Rp][Cp]
__global__ void __launch_bounds__(256) d2f_pad_kernel(const double* __restrict__ src, long ld, int rows, int cols,
                                                      float* __restrict__ dst, int Rp, int Cp) {
  const long total = (long)Rp * Cp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int r = (int)(idx / Cp), c = (int)(idx - (long)r * Cp);
    dst[idx] = (r < rows && c < cols) ? (float)src[r * ld + c] : 0.f;
  }
}

hipError_t launch_d2f_pad(const double* src, long ld, int rows, int cols, float* dst, int Rp, int Cp, hipStream_t s) {
  const long total = (long)Rp * Cp;
  if (total <= 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(d2f_pad_kernel, dim3(blocks), dim3(256), 0, s, src, ld, rows, cols, dst, Rp, Cp);
  return hipGetLastError();
}

hipError_t launch_init_state(const float* init160, float* xs, int N, hipStream_t s) {
  const long total = (long)N * kState;
  if (total <= 0) return hipSuccess;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(init_state_kernel, dim3(blocks), dim3(256), 0, s, init160, xs, total);
  return hipGetLastError();
}

}  // namespace tepose
