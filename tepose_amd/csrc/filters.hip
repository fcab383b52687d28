// Temporal post-filters, both first-order recursions over the frames of one
// sequence: the OneEuro filter on the 72 axis-angle components (lib/utils/one_euro_filter.py,
// lib/utils/smooth_pose.py:24-67) and the MEVA-style quaternion slerp smoothing of the 24 joint
// rotations (evaluate.py:32-59 over lib/utils/slerp_filter_utils.py).  One thread per component /
// per joint walks the frames in order; neighbouring threads touch neighbouring addresses, so
// every step is one coalesced row read and one row write.
#include "common.h"

namespace tepose {

// x[N][D] in place; frame 0 passes through; timestamps are the frame indices (t_e = 1).
__global__ void __launch_bounds__(128) one_euro_kernel(float* __restrict__ x, int N, int D, float min_cutoff,
                                                       float beta, float d_cutoff) {
  const int d = blockIdx.x * 128 + threadIdx.x;
  if (d >= D) return;
  const float two_pi = 6.283185307179586f;
  float x_prev = x[d], dx_prev = 0.f;
  const float t_e = 1.f;
  for (int i = 1; i < N; ++i) {
    const float xi = x[(long)i * D + d];
    const float rd = two_pi * d_cutoff * t_e;
    const float a_d = rd / (rd + 1.f);
    const float dx = (xi - x_prev) / t_e;
    const float dx_hat = a_d * dx + (1.f - a_d) * dx_prev;
    const float cutoff = min_cutoff + beta * fabsf(dx_hat);
    const float r = two_pi * cutoff * t_e;
    const float a = r / (r + 1.f);
    const float x_hat = a * xi + (1.f - a) * x_prev;
    x[(long)i * D + d] = x_hat;
    x_prev = x_hat; dx_prev = dx_hat;
  }
}

hipError_t launch_one_euro(float* x, int N, int D, float min_cutoff, float beta, float d_cutoff, hipStream_t s) {
  if (N <= 1 || D <= 0) return hipSuccess;
  hipLaunchKernelGGL(one_euro_kernel, dim3((D + 127) / 128), dim3(128), 0, s, x, N, D, min_cutoff, beta, d_cutoff);
  return hipGetLastError();
}

__device__ __forceinline__ void mat_to_quat(const float* R, double q[4]) {   // wxyz, Shepperd's branches
  const double m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7],
               m22 = R[8];
  const double tr = m00 + m11 + m22;
  if (tr > 0.0) {
    const double s = sqrt(tr + 1.0) * 2.0;
    q[0] = 0.25 * s; q[1] = (m21 - m12) / s; q[2] = (m02 - m20) / s; q[3] = (m10 - m01) / s;
  } else if (m00 > m11 && m00 > m22) {
    const double s = sqrt(1.0 + m00 - m11 - m22) * 2.0;
    q[0] = (m21 - m12) / s; q[1] = 0.25 * s; q[2] = (m01 + m10) / s; q[3] = (m02 + m20) / s;
  } else if (m11 > m22) {
    const double s = sqrt(1.0 + m11 - m00 - m22) * 2.0;
    q[0] = (m02 - m20) / s; q[1] = (m01 + m10) / s; q[2] = 0.25 * s; q[3] = (m12 + m21) / s;
  } else {
    const double s = sqrt(1.0 + m22 - m00 - m11) * 2.0;
    q[0] = (m10 - m01) / s; q[1] = (m02 + m20) / s; q[2] = (m12 + m21) / s; q[3] = 0.25 * s;
  }
  // The reference (transformations.quaternion_from_matrix, isprecise=False) returns the dominant
  // eigenvector of the symmetric matrix K(R) -- the best-fit quaternion when R is not exactly
  // orthonormal.  K/3 has eigenvalues {1, -1/3, -1/3, -1/3} for a rotation, so K/3 + I/3 is rank one up
  // to the input's non-orthonormality and two power steps from the Shepperd start converge to it.
  double v[4] = {q[1], q[2], q[3], q[0]};                     // (x, y, z, w)
  const double K[4][4] = {
      {(m00 - m11 - m22) / 3 + 1.0 / 3, (m01 + m10) / 3, (m02 + m20) / 3, (m21 - m12) / 3},
      {(m01 + m10) / 3, (m11 - m00 - m22) / 3 + 1.0 / 3, (m12 + m21) / 3, (m02 - m20) / 3},
      {(m02 + m20) / 3, (m12 + m21) / 3, (m22 - m00 - m11) / 3 + 1.0 / 3, (m10 - m01) / 3},
      {(m21 - m12) / 3, (m02 - m20) / 3, (m10 - m01) / 3, (m00 + m11 + m22) / 3 + 1.0 / 3}};
  for (int it = 0; it < 2; ++it) {
    double u[4], n = 0.0;
    for (int a = 0; a < 4; ++a) { u[a] = K[a][0] * v[0] + K[a][1] * v[1] + K[a][2] * v[2] + K[a][3] * v[3]; n += u[a] * u[a]; }
    n = sqrt(n);
    for (int a = 0; a < 4; ++a) v[a] = u[a] / n;
  }
  q[0] = v[3]; q[1] = v[0]; q[2] = v[1]; q[3] = v[2];
}

__device__ __forceinline__ void quat_to_mat(const double qin[4], float* R) {  // quaternion_matrix()
  const double n = qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3];
  if (n < 8.881784197001252e-16) {
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.f : 0.f;
    return;
  }
  const double sc = sqrt(2.0 / n);
  const double w = qin[0] * sc, x = qin[1] * sc, y = qin[2] * sc, z = qin[3] * sc;
  R[0] = (float)(1.0 - y * y - z * z); R[1] = (float)(x * y - z * w); R[2] = (float)(x * z + y * w);
  R[3] = (float)(x * y + z * w); R[4] = (float)(1.0 - x * x - z * z); R[5] = (float)(y * z - x * w);
  R[6] = (float)(x * z - y * w); R[7] = (float)(y * z + x * w); R[8] = (float)(1.0 - x * x - y * y);
}

// in/out [N][J][3][3]; thread = joint.  c_t: sign-corrected quaternion of frame t (quat_correct),
// s_t = slerp(s_{t-1}, c_t, ratio) (quat_smooth), out_t = matrix(s_t).
__global__ void __launch_bounds__(64) slerp_smooth_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int N, int J, double ratio) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= J) return;
  const double eps = 8.881784197001252e-16;               // numpy.finfo(float).eps * 4
  double c_prev[4], s_prev[4];
  for (int t = 0; t < N; ++t) {
    double q[4];
    mat_to_quat(in + ((long)t * J + j) * 9, q);
    if (t == 0) {
      for (int k = 0; k < 4; ++k) { c_prev[k] = q[k]; s_prev[k] = q[k]; }
    } else {
      double dm = 0.0, dp = 0.0;
      for (int k = 0; k < 4; ++k) { dm += (c_prev[k] - q[k]) * (c_prev[k] - q[k]); dp += (c_prev[k] + q[k]) * (c_prev[k] + q[k]); }
      if (dm > dp)
        for (int k = 0; k < 4; ++k) q[k] = -q[k];
      for (int k = 0; k < 4; ++k) c_prev[k] = q[k];
      // quaternion_slerp(s_prev, q, ratio), spin = 0, shortest path
      double q0[4], q1[4], n0 = 0.0, n1 = 0.0;
      for (int k = 0; k < 4; ++k) { n0 += s_prev[k] * s_prev[k]; n1 += q[k] * q[k]; }
      n0 = sqrt(n0); n1 = sqrt(n1);
      for (int k = 0; k < 4; ++k) { q0[k] = s_prev[k] / n0; q1[k] = q[k] / n1; }
      double d = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
      if (!(fabs(fabs(d) - 1.0) < eps) && ratio != 0.0) {
        if (ratio == 1.0) {
          for (int k = 0; k < 4; ++k) q0[k] = q1[k];
        } else {
          if (d < 0.0) { d = -d; for (int k = 0; k < 4; ++k) q1[k] = -q1[k]; }
          const double ang = acos(d);
          if (!(fabs(ang) < eps)) {
            const double isin = 1.0 / sin(ang);
            const double a = sin((1.0 - ratio) * ang) * isin, b = sin(ratio * ang) * isin;
            for (int k = 0; k < 4; ++k) q0[k] = q0[k] * a + q1[k] * b;
          }
        }
      }
      for (int k = 0; k < 4; ++k) s_prev[k] = q0[k];
    }
    quat_to_mat(s_prev, out + ((long)t * J + j) * 9);
  }
}

hipError_t launch_slerp_smooth(const float* in, float* out, int N, int J, double ratio, hipStream_t s) {
  if (N <= 0 || J <= 0) return hipSuccess;
  hipLaunchKernelGGL(slerp_smooth_kernel, dim3((J + 63) / 64), dim3(64), 0, s, in, out, N, J, ratio);
  return hipGetLastError();
}

}  // namespace tepose
