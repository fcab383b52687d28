// Evaluation metrics on the device: MPJPE, PA-MPJPE (similarity Procrustes),
// acceleration error, MPVPE.  Reference: evaluate.py:413-457, lib/utils/eval_utils.py:110-138
// (accel), :141-175 (verts), :287-337 (batch_compute_similarity_transform_torch).
// One thread per frame for the joint metrics (14-17 joints: everything lives in registers,
// the 3x3 SVD is a cyclic Jacobi in fp64), one block per frame for the 6890-vertex mean.
#include "common.h"

namespace tepose {

constexpr int kMaxJ = 17;

// pelvis_mode 0: mean of joints 2 and 3 (LSP order, evaluate.py:424-425); 1: joint J-3 (evaluate.py:420-422)
__device__ __forceinline__ void load_aligned(const float* __restrict__ src, int J, int pelvis_mode,
                                             float (&P)[kMaxJ][3]) {
  for (int j = 0; j < J; ++j) { P[j][0] = src[3 * j]; P[j][1] = src[3 * j + 1]; P[j][2] = src[3 * j + 2]; }
  float c[3];
  for (int a = 0; a < 3; ++a) c[a] = pelvis_mode == 0 ? (P[2][a] + P[3][a]) / 2.0f : P[J - 3][a];
  for (int j = 0; j < J; ++j)
    for (int a = 0; a < 3; ++a) P[j][a] -= c[a];
}

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi): A = V diag(w) V^T
__device__ __forceinline__ void jacobi3(double A[3][3], double V[3][3], double w[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}

__global__ void __launch_bounds__(64) metrics_joints_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ target, int N, int J,
                                                            int pelvis_mode, float* __restrict__ mpjpe,
                                                            float* __restrict__ pa_mpjpe,
                                                            float* __restrict__ accel) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= N) return;
  float P[kMaxJ][3], T[kMaxJ][3];
  load_aligned(pred + (long)i * J * 3, J, pelvis_mode, P);
  load_aligned(target + (long)i * J * 3, J, pelvis_mode, T);
  // ---- MPJPE (mm)
  float e = 0.f;
  for (int j = 0; j < J; ++j) {
    const float dx = P[j][0] - T[j][0], dy = P[j][1] - T[j][1], dz = P[j][2] - T[j][2];
    e += sqrtf(dx * dx + dy * dy + dz * dz);
  }
  mpjpe[i] = e / J * 1000.f;
  // ---- PA-MPJPE: S1 = pred, S2 = target
  double mu1[3] = {0, 0, 0}, mu2[3] = {0, 0, 0};
  for (int j = 0; j < J; ++j)
    for (int a = 0; a < 3; ++a) { mu1[a] += P[j][a]; mu2[a] += T[j][a]; }
  for (int a = 0; a < 3; ++a) { mu1[a] /= J; mu2[a] /= J; }
  double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, var1 = 0.0;
  for (int j = 0; j < J; ++j) {
    double x1[3], x2[3];
    for (int a = 0; a < 3; ++a) { x1[a] = P[j][a] - mu1[a]; x2[a] = T[j][a] - mu2[a]; var1 += x1[a] * x1[a]; }
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) K[a][b] += x1[a] * x2[b];
  }
  // K = U S V^T  ->  K^T K = V S^2 V^T ; U = K V S^-1 ; R = V diag(1,1,det(U V^T)) U^T
  double A[3][3], V[3][3], w[3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) A[a][b] = K[0][a] * K[0][b] + K[1][a] * K[1][b] + K[2][a] * K[2][b];
  jacobi3(A, V, w);
  int o[3] = {0, 1, 2};                                   // sort descending
  if (w[o[0]] < w[o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  if (w[o[1]] < w[o[2]]) { int t = o[1]; o[1] = o[2]; o[2] = t; }
  if (w[o[0]] < w[o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
  double Vs[3][3], U[3][3];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][o[c]];
  for (int c = 0; c < 2; ++c) {
    double u[3], n = 0.0;
    for (int r = 0; r < 3; ++r) { u[r] = K[r][0] * Vs[0][c] + K[r][1] * Vs[1][c] + K[r][2] * Vs[2][c]; n += u[r] * u[r]; }
    n = sqrt(n);
    for (int r = 0; r < 3; ++r) U[r][c] = n > 0 ? u[r] / n : (r == c ? 1.0 : 0.0);
  }
  // third left singular vector: any unit vector orthogonal to the first two (its sign cancels in R)
  U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
  U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
  U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  auto det3 = [](const double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
  };
  const double d = det3(U) * det3(Vs) >= 0 ? 1.0 : -1.0;  // sign(det(U V^T))
  double R[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) R[a][b] = Vs[a][0] * U[b][0] + Vs[a][1] * U[b][1] + d * Vs[a][2] * U[b][2];
  double tr = 0.0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) tr += R[a][b] * K[b][a];
  const double scale = tr / var1;
  double t[3];
  for (int a = 0; a < 3; ++a) t[a] = mu2[a] - scale * (R[a][0] * mu1[0] + R[a][1] * mu1[1] + R[a][2] * mu1[2]);
  double epa = 0.0;
  for (int j = 0; j < J; ++j) {
    double dd = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double h = scale * (R[a][0] * P[j][0] + R[a][1] * P[j][1] + R[a][2] * P[j][2]) + t[a] - T[j][a];
      dd += h * h;
    }
    epa += sqrt(dd);
  }
  pa_mpjpe[i] = (float)(epa / J * 1000.0);
  // ---- acceleration error of frame i (second difference; 0 at the two ends, evaluate.py:439-440)
  float acc = 0.f;
  if (i > 0 && i < N - 1) {
    float Pm[kMaxJ][3], Tm[kMaxJ][3], Pp[kMaxJ][3], Tp[kMaxJ][3];
    load_aligned(pred + (long)(i - 1) * J * 3, J, pelvis_mode, Pm);
    load_aligned(target + (long)(i - 1) * J * 3, J, pelvis_mode, Tm);
    load_aligned(pred + (long)(i + 1) * J * 3, J, pelvis_mode, Pp);
    load_aligned(target + (long)(i + 1) * J * 3, J, pelvis_mode, Tp);
    for (int j = 0; j < J; ++j) {
      float s = 0.f;
      for (int a = 0; a < 3; ++a) {
        const float ap = Pm[j][a] - 2.f * P[j][a] + Pp[j][a];
        const float at = Tm[j][a] - 2.f * T[j][a] + Tp[j][a];
        s += (ap - at) * (ap - at);
      }
      acc += sqrtf(s);
    }
    acc = acc / J * 1000.f;
  }
  accel[i] = acc;
}

hipError_t launch_metrics_joints(const float* pred, const float* target, int N, int J, int pelvis_mode,
                                 float* mpjpe, float* pa, float* accel, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(metrics_joints_kernel, dim3((N + 63) / 64), dim3(64), 0, s, pred, target, N, J, pelvis_mode,
                     mpjpe, pa, accel);
  return hipGetLastError();
}

// MPVPE: mean over 6890 vertices of |pred - target| (mm), one block per frame
__global__ void __launch_bounds__(256) metrics_verts_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ target, int N,
                                                            float* __restrict__ mpvpe) {
  __shared__ float red[4];
  const long base = (long)blockIdx.x * kNV * 3;
  float e = 0.f;
  for (int v = threadIdx.x; v < kNV; v += 256) {
    const float dx = pred[base + 3 * v] - target[base + 3 * v];
    const float dy = pred[base + 3 * v + 1] - target[base + 3 * v + 1];
    const float dz = pred[base + 3 * v + 2] - target[base + 3 * v + 2];
    e += sqrtf(dx * dx + dy * dy + dz * dz);
  }
  for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) mpvpe[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) / kNV * 1000.f;
}

hipError_t launch_metrics_verts(const float* pred, const float* target, int N, float* mpvpe, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(metrics_verts_kernel, dim3(N), dim3(256), 0, s, pred, target, N, mpvpe);
  return hipGetLastError();
}

}  // namespace tepose
