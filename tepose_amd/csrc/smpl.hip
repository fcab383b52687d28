// SMPL forward for N persons: rot6d -> R, R -> axis-angle, 24-joint kinematic chain, pose
// feature, linear blend skinning, joint regression, projection.
//
// Reference: lib/utils/geometry.py:330-344 (rot6d), :68-233 (R -> aa);
// lib/models/smpl.py:72-84 + smplx.lbs (LBS); lib/models/spin.py:275-280,
// :307-351 (eval joint regressor, projection).
//
// Split (round 1): prep (one wave per person) -> blend-shape GEMM on the MFMA kernel
// ([N,224] x [224,20670], the only FLOP-heavy part) -> skin (thread per vertex, W row in
// registers, persons looped, A matrices read through the scalar cache) -> joints (block per
// person, sparse regressors in CSR).
#include "common.h"

namespace tepose {

__device__ const int kJointMap49[49] = {24, 12, 17, 19, 21, 16, 18, 20, 0,  2,  5,  8,  1,  4,  7,  25, 26,
                                        27, 28, 29, 30, 31, 32, 33, 34, 8,  5,  45, 46, 4,  7,  21, 19, 17,
                                        16, 18, 20, 47, 48, 49, 50, 51, 52, 53, 24, 26, 25, 28, 27};
__device__ const int kH36mToJ14[14] = {6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10};
__device__ const int kExtraVerts[21] = {332,  6260, 2800, 4071, 583,  3216, 3226, 3387, 6617, 6624, 6787,
                                        2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133};

// ------------------------------------------------------------------ pack-time constants
// J0 = J_regressor * v_template, JS = J_regressor * shapedirs (fp64 accumulate), so the rest
// joints are J0 + JS * beta instead of a 6890-long reduction per person.
__global__ void __launch_bounds__(256) smpl_joint_consts_kernel(const float* __restrict__ vt,
                                                                const float* __restrict__ sd,
                                                                const float* __restrict__ jr,
                                                                float* J0, float* JS) {
  // blockIdx.x = (j*3 + c)*11 + l, l = 0 -> template, l = 1..10 -> shapedir l-1
  const int l = blockIdx.x % 11, jc = blockIdx.x / 11, c = jc % 3, j = jc / 3;
  double acc = 0.0;
  for (int v = threadIdx.x; v < kNV; v += 256) {
    const double w = jr[(long)j * kNV + v];
    const double x = l == 0 ? vt[v * 3 + c] : sd[((long)v * 3 + c) * 10 + (l - 1)];
    acc += w * x;
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (l == 0) J0[j * 3 + c] = (float)red[0];
    else JS[(j * 3 + c) * 10 + (l - 1)] = (float)red[0];
  }
}

// blendW[n = 3v+c][k]: k=0 v_template, k=1..10 shapedirs, k=11..217 posedirs[k-11][n]
__global__ void __launch_bounds__(256) smpl_blendw_kernel(const float* __restrict__ vt,
                                                          const float* __restrict__ sd,
                                                          const float* __restrict__ pd, float* bw) {
  const long total = (long)kBlendN * kBlendK;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int n = (int)(idx / kBlendK), k = (int)(idx % kBlendK);
    float v = 0.f;
    if (n < 3 * kNV) {
      if (k == 0) v = vt[n];
      else if (k <= 10) v = sd[(long)n * 10 + (k - 1)];
      else if (k < 218) v = pd[(long)(k - 11) * (3 * kNV) + n];
    }
    bw[idx] = v;
  }
}

hipError_t launch_smpl_consts(const float* vt, const float* sd, const float* pd, const float* jr,
                              float* J0, float* JS, float* blendW, hipStream_t s) {
  hipLaunchKernelGGL(smpl_joint_consts_kernel, dim3(24 * 3 * 11), dim3(256), 0, s, vt, sd, jr, J0, JS);
  hipLaunchKernelGGL(smpl_blendw_kernel, dim3(4096), dim3(256), 0, s, vt, sd, pd, blendW);
  return hipGetLastError();
}

// Dense [rows][cols] -> CSR (ptr[rows+1], idx, val), column order kept.  One block per row.
__global__ void __launch_bounds__(256) csr_build_kernel(const float* __restrict__ dense, int rows,
                                                        int cols, int* ptr, int* idx, float* val,
                                                        int cap) {
  __shared__ int wsum[4];
  __shared__ int base_sh;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // offset of this row = non-zeros of all previous rows
  int cnt = 0;
  for (long i = tid; i < (long)row * cols; i += 256) cnt += dense[i] != 0.f;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  if (lane == 0) wsum[wave] = cnt;
  __syncthreads();
  if (tid == 0) base_sh = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  int base = base_sh;
  if (tid == 0) ptr[row] = base;
  for (int c0 = 0; c0 < cols; c0 += 256) {
    const int c = c0 + tid;
    const float v = c < cols ? dense[(long)row * cols + c] : 0.f;
    const bool nz = v != 0.f;
    const unsigned long long m = __ballot(nz);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base + before;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (nz && off < cap) { idx[off] = c; val[off] = v; }
    base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  if (row == rows - 1 && tid == 0) ptr[rows] = base;
}

hipError_t launch_csr_build(const float* dense, int rows, int cols, int* ptr, int* idx, float* val,
                            int cap, hipStream_t s) {
  hipLaunchKernelGGL(csr_build_kernel, dim3(rows), dim3(256), 0, s, dense, rows, cols, ptr, idx, val, cap);
  return hipGetLastError();
}

// ------------------------------------------------------------------ per-person prep
// geometry.py:191-233 on M = R^T, then :118-157.
__device__ __forceinline__ void rotmat_to_aa(const float R[3][3], float aa[3]) {
#pragma clang fp contract(off)   // same op sequence and roundings as the reference's elementwise torch ops
  const float m00 = R[0][0], m01 = R[1][0], m02 = R[2][0];
  const float m10 = R[0][1], m11 = R[1][1], m12 = R[2][1];
  const float m20 = R[0][2], m21 = R[1][2], m22 = R[2][2];
  const bool d2 = m22 < 1e-6f, d01 = m00 > m11, d0n1 = m00 < -m11;
  float w, x, y, z, t;
  if (d2 && d01) {
    t = 1.f + m00 - m11 - m22; w = m12 - m21; x = t; y = m01 + m10; z = m20 + m02;
  } else if (d2) {
    t = 1.f - m00 + m11 - m22; w = m20 - m02; x = m01 + m10; y = t; z = m12 + m21;
  } else if (d0n1) {
    t = 1.f - m00 - m11 + m22; w = m01 - m10; x = m20 + m02; y = m12 + m21; z = t;
  } else {
    t = 1.f + m00 + m11 + m22; w = t; x = m12 - m21; y = m20 - m02; z = m01 - m10;
  }
  const float st = sqrtf(t);
  w = (w / st) * 0.5f; x = (x / st) * 0.5f; y = (y / st) * 0.5f; z = (z / st) * 0.5f;
  const float s2 = x * x + y * y + z * z;
  const float sn = sqrtf(s2);
  const float two_theta = 2.f * (w < 0.f ? atan2f(-sn, -w) : atan2f(sn, w));
  const float k = s2 > 0.f ? two_theta / sn : 2.f;
  aa[0] = x * k; aa[1] = y * k; aa[2] = z * k;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (aa[i] != aa[i]) aa[i] = 0.f;
}

// geometry.py:330-344: x.view(-1,3,2): a1 = x[0::2], a2 = x[1::2]; F.normalize(v, eps=1e-6) = v / max(|v|, 1e-6)
__device__ __forceinline__ void rot6d_to_rotmat(const float* __restrict__ x, float R[3][3]) {
#pragma clang fp contract(off)   // the reference rounds every product (torch elementwise ops): no fused a2 - dp * b1
  const float a1[3] = {x[0], x[2], x[4]};
  const float a2[3] = {x[1], x[3], x[5]};
  const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-6f);
  const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
  const float dp = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  const float u[3] = {a2[0] - dp * b1[0], a2[1] - dp * b1[1], a2[2] - dp * b1[2]};
  const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-6f);
  const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2],
                       b1[0] * b2[1] - b1[1] * b2[0]};
#pragma unroll
  for (int r = 0; r < 3; ++r) { R[r][0] = b1[r]; R[r][1] = b2[r]; R[r][2] = b3[r]; }
}

// One wave per person, lane = joint (24 active).  xs row = [pose6d 144 | betas 10 | cam 3 | 0 0 0].
// mode 0: xs row = regressor state (6D pose, rot6d_to_rotmat).  mode 1: xs row = theta[85]
// (cam3 | axis-angle 72 | betas 10) and R = Rodrigues(aa) as smplx does for pose2rot=True
// (GT meshes of MPVPE, lib/utils/eval_utils.py:155-169).
// mode 2: pose row = 24 rotation matrices (pose2rot=False callers: evaluate.py:279-286).
struct PrepIn {
  const float* pose; int pose_ld;     // mode 0: 144 (6D) | mode 1: 72 (axis-angle) | mode 2: 216 (rotmats)
  const float* betas; int betas_ld;   // 10 per row
  const float* cam; int cam_ld;       // 3 per row or nullptr (only copied into theta)
  int mode;
  half_t *pf_hi, *pf_lo; long pf_kst; // optional: the pose-feature row also as hi / lo planes (the blend-shape product's A operand)
};
__device__ __forceinline__ void pf_put(const PrepIn& in, float* f, long p, int col, float v) {
  f[col] = v;
  if (in.pf_hi) {
    const long o = (long)(col >> 5) * in.pf_kst + plane_index(p, col & 31, 0);
    split_hi_lo(v, in.pf_hi[o], in.pf_lo[o]);
  }
}
// one person's preparation by one wave (lane = joint): A_p[24][12] skinning transforms, f[224] pose-feature row (also as
// planes when in.pf_hi is set), and -- where asked -- posed joints [24][3], rotation matrices [24][9], theta[85]
__device__ __forceinline__ void smpl_prep_person(const SmplConsts& c, int maxdepth, const PrepIn& in, int p, int lane,
                                                 float* A_p, float* f, float* posed_p, float* rotmat_p, float* th) {
  const float* x = in.pose + (long)p * in.pose_ld;
  const float* bx = in.betas + (long)p * in.betas_ld;
  const int mode = in.mode;
  const int j = lane < kNJ ? lane : 0;
  const bool act = lane < kNJ;
  float R[3][3];
  float beta[10];
#pragma unroll
  for (int l = 0; l < 10; ++l) beta[l] = bx[l];
  if (mode == 2) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) R[r][cc] = x[9 * j + 3 * r + cc];
  } else if (mode == 1) {
    // smplx.lbs.batch_rodrigues: angle = |aa + 1e-8|, R = I + sin K + (1 - cos) K^2
    const float ax = x[3 * j], ay = x[3 * j + 1], az = x[3 * j + 2];
    const float ex = ax + 1e-8f, ey = ay + 1e-8f, ez = az + 1e-8f;
    const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
    const float rx = ax / ang, ry = ay / ang, rz = az / ang;
    const float sn = sinf(ang), cs = 1.f - cosf(ang);
    const float K[3][3] = {{0.f, -rz, ry}, {rz, 0.f, -rx}, {-ry, rx, 0.f}};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const float kk = K[r][0] * K[0][cc] + K[r][1] * K[1][cc] + K[r][2] * K[2][cc];
        R[r][cc] = (r == cc ? 1.f : 0.f) + sn * K[r][cc] + cs * kk;
      }
  } else {
    rot6d_to_rotmat(x + 6 * j, R);
  }
  float aa[3] = {0.f, 0.f, 0.f};
  if (th) rotmat_to_aa(R, aa);

  // rest joint of this lane and of its parent
  float Jr[3];
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) {
    float v = c.J0[j * 3 + cc];
#pragma unroll
    for (int l = 0; l < 10; ++l) v += c.JS[(j * 3 + cc) * 10 + l] * beta[l];
    Jr[cc] = v;
  }
  const int par = act ? c.parents[j] : -1;
  const int dep = act ? c.depth[j] : 0;
  const int src = par < 0 ? 0 : par;
  float rel[3];
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) {
    const float pj = __shfl(Jr[cc], src);
    rel[cc] = par < 0 ? Jr[cc] : Jr[cc] - pj;
  }
  // chain: G = [R | rel] for the root, G_parent * [R | rel] below it, level by level
  float G[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r) { G[r][0] = R[r][0]; G[r][1] = R[r][1]; G[r][2] = R[r][2]; G[r][3] = rel[r]; }
  for (int lvl = 1; lvl <= maxdepth; ++lvl) {
    float P[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) P[r][cc] = __shfl(G[r][cc], src);
    if (dep == lvl) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          G[r][cc] = P[r][0] * R[0][cc] + P[r][1] * R[1][cc] + P[r][2] * R[2][cc];
        G[r][3] = P[r][0] * rel[0] + P[r][1] * rel[1] + P[r][2] * rel[2] + P[r][3];
      }
    }
  }
  if (act) {
    float* Ao = A_p + j * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      Ao[r * 4 + 0] = G[r][0]; Ao[r * 4 + 1] = G[r][1]; Ao[r * 4 + 2] = G[r][2];
      Ao[r * 4 + 3] = G[r][3] - (G[r][0] * Jr[0] + G[r][1] * Jr[1] + G[r][2] * Jr[2]);
      if (posed_p) posed_p[j * 3 + r] = G[r][3];
      if (rotmat_p) {
        float* Ro = rotmat_p + j * 9;
        Ro[r * 3 + 0] = R[r][0]; Ro[r * 3 + 1] = R[r][1]; Ro[r * 3 + 2] = R[r][2];
      }
    }
    if (th) { th[3 + 3 * j + 0] = aa[0]; th[3 + 3 * j + 1] = aa[1]; th[3 + 3 * j + 2] = aa[2]; }
    if (j >= 1) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) pf_put(in, f, p, 11 + 9 * (j - 1) + 3 * r + cc, R[r][cc] - (r == cc ? 1.f : 0.f));
    } else {
      pf_put(in, f, p, 0, 1.f);
#pragma unroll
      for (int l = 0; l < 10; ++l) { pf_put(in, f, p, 1 + l, beta[l]); if (th) th[75 + l] = beta[l]; }
#pragma unroll
      for (int l = 218; l < kBlendK; ++l) pf_put(in, f, p, l, 0.f);
      if (th && in.cam) { const float* cm = in.cam + (long)p * in.cam_ld; th[0] = cm[0]; th[1] = cm[1]; th[2] = cm[2]; }
    }
  }
}

__global__ void __launch_bounds__(256) smpl_prep_kernel(SmplConsts c, int maxdepth, PrepIn in, int N,
                                                        float* __restrict__ pf, float* __restrict__ Amat,
                                                        float* __restrict__ posed,
                                                        float* __restrict__ rotmat,
                                                        float* __restrict__ theta) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= N) return;                       // wave-uniform
  smpl_prep_person(c, maxdepth, in, p, lane, Amat + (long)p * kNJ * 12, pf + (long)p * kBlendK,
                   posed ? posed + (long)p * kNJ * 3 : nullptr, rotmat ? rotmat + (long)p * kNJ * 9 : nullptr,
                   theta ? theta + (long)p * kTheta : nullptr);
}

// ------------------------------------------------------------------ skinning
// thread = vertex, 24 skin weights in registers; loops over a group of persons whose A
// matrices are wave-uniform reads.  T = sum_j W[v,j] A_j ; vert = T[:3,:3] v_posed + T[:3,3].
constexpr int kSkinPG = 32;
__global__ void __launch_bounds__(256) smpl_skin_kernel(const float* __restrict__ lbsW,
                                                        const float* __restrict__ vposed,
                                                        const float* __restrict__ Amat, int N,
                                                        float* __restrict__ verts, int pg) {
  __shared__ float As[kNJ * 12];
  const int v = blockIdx.x * 256 + threadIdx.x;
  const bool ok = v < kNV;
  float w[kNJ];
#pragma unroll
  for (int j = 0; j < kNJ; ++j) w[j] = ok ? lbsW[(long)v * kNJ + j] : 0.f;
  const int p0 = blockIdx.y * pg;
  const int p1 = min(p0 + pg, N);
  for (int p = p0; p < p1; ++p) {
    __syncthreads();
    for (int i = threadIdx.x; i < kNJ * 12; i += 256) As[i] = Amat[(long)p * kNJ * 12 + i];
    __syncthreads();
    float t[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) t[e] = 0.f;
#pragma unroll
    for (int j = 0; j < kNJ; ++j)
#pragma unroll
      for (int e = 0; e < 12; ++e) t[e] += w[j] * As[j * 12 + e];
    if (ok) {
      const float* vp = vposed + (long)p * kVertLd + 3 * v;
      const float x = vp[0], y = vp[1], z = vp[2];
      float* o = verts + ((long)p * kNV + v) * 3;
      o[0] = t[0] * x + t[1] * y + t[2] * z + t[3];
      o[1] = t[4] * x + t[5] * y + t[6] * z + t[7];
      o[2] = t[8] * x + t[9] * y + t[10] * z + t[11];
    }
  }
}

// Compacted skin weights: the official SMPL model has at most 4 non-zero weights per vertex.
// compact[v] = 4 x (joint, weight) in ascending joint order (the dense sum's order with its
// exact zeros removed, so the result is bit-identical); *max_nnz tells the host whether the
// table qualifies.
__global__ void __launch_bounds__(256) lbs_compact_kernel(const float* __restrict__ lbsW, int* __restrict__ cidx,
                                                          float* __restrict__ cval, int* __restrict__ max_nnz) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= kNV) return;
  int n = 0;
  for (int j = 0; j < kNJ; ++j) {
    const float w = lbsW[(long)v * kNJ + j];
    if (w != 0.f) {
      if (n < 4) { cidx[v * 4 + n] = j; cval[v * 4 + n] = w; }
      ++n;
    }
  }
  for (int k = n; k < 4; ++k) { cidx[v * 4 + k] = 0; cval[v * 4 + k] = 0.f; }
  atomicMax(max_nnz, n);
}

hipError_t launch_lbs_compact(const float* lbsW, int* cidx, float* cval, int* max_nnz, hipStream_t s) {
  hipError_t e = hipMemsetAsync(max_nnz, 0, sizeof(int), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lbs_compact_kernel, dim3((kNV + 255) / 256), dim3(256), 0, s, lbsW, cidx, cval, max_nnz);
  return hipGetLastError();
}

// thread = vertex with its <= 4 (joint, weight) pairs in registers; persons looped; the 4 joint
// transforms are gathered from the LDS copy of A (3 x 16-byte reads per joint).  HBM-bound:
// 12 B in + 12 B out per (person, vertex).
#ifndef TEPOSE_SKIN_DIAG
#define TEPOSE_SKIN_DIAG 0   // diagnostic builds only (the race probes of round 4, profiles/r04_packed_fp32_erratum.txt): every wave of the skinning
#endif                       // kernel records when it ran and where (s_memrealtime at entry / exit, HW_ID at entry / exit)
#if TEPOSE_SKIN_DIAG
__device__ unsigned tepose_skin_check_buf[16 + 64 * 32];
extern "C" int tepose_debug_skin_check(unsigned* host, int n, int reset) {
  int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tepose_skin_check_buf), (size_t)n * 4);
  if (reset) { unsigned z = 0; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(tepose_skin_check_buf), &z, 4); }
  return rc;
}
__device__ unsigned long long tepose_skin_diag_buf[64 * 27 * 4 * 4];
extern "C" int tepose_debug_skin_diag(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tepose_skin_diag_buf), (size_t)n * 8);
}
#endif
__global__ void __launch_bounds__(256) smpl_skin4_kernel(const int* __restrict__ cidx,
                                                         const float* __restrict__ cval,
                                                         const float* __restrict__ vposed,
                                                         const float* __restrict__ Amat, int N,
                                                         float* __restrict__ verts, int pg) {
  __shared__ __attribute__((aligned(16))) float As[kNJ * 12];
  typedef float f4 __attribute__((ext_vector_type(4)));
#if TEPOSE_SKIN_DIAG
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned diag_hw0 = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, all 32 bits
#endif
  const int v = blockIdx.x * 256 + threadIdx.x;
  const bool ok = v < kNV;
  int jx[4];
  float wv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { jx[k] = ok ? cidx[v * 4 + k] * 12 : 0; wv[k] = ok ? cval[v * 4 + k] : 0.f; }
  const int p0 = blockIdx.y * pg;
  const int p1 = min(p0 + pg, N);
  // one person ahead: the next person's transforms and posed-vertex coordinates are requested before this person's blend (round 4: a person
  // used to cost two dependent global round trips -- transforms -> LDS, then the vertex -- with 32 persons per workgroup in sequence)
  static_assert(kNJ * 12 <= 512, "two loads per thread cover the transforms");
  float a0 = 0.f, a1 = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
  if (p0 < p1) {
    a0 = Amat[(long)p0 * kNJ * 12 + threadIdx.x];
    if (threadIdx.x + 256 < kNJ * 12) a1 = Amat[(long)p0 * kNJ * 12 + 256 + threadIdx.x];
    if (ok) { const float* vp = vposed + (long)p0 * kVertLd + 3 * v; nx = vp[0]; ny = vp[1]; nz = vp[2]; }
  }
  for (int p = p0; p < p1; ++p) {
    __syncthreads();
    As[threadIdx.x] = a0;
    if (threadIdx.x + 256 < kNJ * 12) As[256 + threadIdx.x] = a1;
    __syncthreads();
    const float x = nx, y = ny, z = nz;
    if (p + 1 < p1) {
      a0 = Amat[(long)(p + 1) * kNJ * 12 + threadIdx.x];
      if (threadIdx.x + 256 < kNJ * 12) a1 = Amat[(long)(p + 1) * kNJ * 12 + 256 + threadIdx.x];
      if (ok) { const float* vp = vposed + (long)(p + 1) * kVertLd + 3 * v; nx = vp[0]; ny = vp[1]; nz = vp[2]; }
    }
    float t[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) t[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f4 r0 = *(const f4*)(As + jx[k]), r1 = *(const f4*)(As + jx[k] + 4), r2 = *(const f4*)(As + jx[k] + 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { t[e] += wv[k] * r0[e]; t[4 + e] += wv[k] * r1[e]; t[8 + e] += wv[k] * r2[e]; }
    }
    if (ok) {
      float* o = verts + ((long)p * kNV + v) * 3;
      o[0] = t[0] * x + t[1] * y + t[2] * z + t[3];
      o[1] = t[4] * x + t[5] * y + t[6] * z + t[7];
      o[2] = t[8] * x + t[9] * y + t[10] * z + t[11];
#if TEPOSE_SKIN_DIAG >= 2
      // self-check: the first row of the blended transform again, from a SECOND read of the LDS copy and with scalar FMAs the
      // compiler cannot pack (asm); a lane whose packed accumulators differ records what it saw
      {
        float u[4] = {0.f, 0.f, 0.f, 0.f};
        f4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          q[k] = *(volatile const f4*)(As + jx[k]);
#pragma unroll
          for (int e = 0; e < 4; ++e) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(u[e]) : "v"(wv[k]), "v"(q[k][e]));
        }
        const bool diff = __float_as_uint(u[0]) != __float_as_uint(t[0]) || __float_as_uint(u[1]) != __float_as_uint(t[1]) ||
                          __float_as_uint(u[2]) != __float_as_uint(t[2]) || __float_as_uint(u[3]) != __float_as_uint(t[3]);
        if (diff) {
          const unsigned slot = atomicAdd(&tepose_skin_check_buf[0], 1u);
          if (slot < 64) {
            unsigned* d = tepose_skin_check_buf + 16 + slot * 32;
            d[0] = (unsigned)p; d[1] = (unsigned)v;
            for (int e = 0; e < 4; ++e) { d[2 + e] = __float_as_uint(t[e]); d[6 + e] = __float_as_uint(u[e]); }
            for (int k = 0; k < 4; ++k) { d[10 + k] = __float_as_uint(wv[k]); d[14 + k] = (unsigned)jx[k]; d[18 + k] = __float_as_uint(q[k][0]); }
          }
        }
      }
#endif
    }
  }
#if TEPOSE_SKIN_DIAG
  if ((threadIdx.x & 63) == 0 && blockIdx.y < 64) {
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const unsigned hw1 = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    unsigned long long* d = tepose_skin_diag_buf + (((size_t)blockIdx.y * 27 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 4;
    d[0] = diag_t0; d[1] = t1; d[2] = diag_hw0; d[3] = hw1;
  }
#endif
}

hipError_t launch_smpl_skin(const SmplConsts& c, const float* vposed, const float* Amat, int N,
                            float* verts, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  // persons per block: up to kSkinPG (the per-vertex weights are loaded once per block), fewer for small batches so
  // that the grid still covers the chip (N = 64: 27 x 22 blocks instead of 27 x 2: 36 -> 8 us)
  const int vb = (kNV + 255) / 256;
  int pg = (int)((long)N * vb / 512);
  pg = pg < 1 ? 1 : (pg > kSkinPG ? kSkinPG : pg);
  dim3 grid(vb, (N + pg - 1) / pg);
  if (c.lbs_sparse)
    hipLaunchKernelGGL(smpl_skin4_kernel, grid, dim3(256), 0, s, c.lbs_cidx, c.lbs_cval, vposed, Amat, N, verts, pg);
  else
    hipLaunchKernelGGL(smpl_skin_kernel, grid, dim3(256), 0, s, c.lbsW, vposed, Amat, N, verts, pg);
  return hipGetLastError();
}

// ------------------------------------------------------------------ few persons (N <= kSmallN): prep + blend shapes + skinning in one launch
// One window is one person: the four-launch chain above then costs four launch latencies for 19 MB of table reads.
// Here a workgroup owns 32 vertices: it first issues the loads of its 96 blend-shape rows (224 floats each; a wave owns
// 24 rows, 8 lanes share a row -- these do not depend on the pose), its waves < N then prepare one person
// each into LDS (every workgroup repeats the 24-joint chain: ~3 us of latency against a grid-wide hand-off), the rows
// are reduced against the pose features in LDS (fp32 FMA, exact-fp32 tables; 8 lanes per row, so 3 shuffle steps) and the 32 x N vertices skinned from LDS.
// Workgroup 0 also writes the per-person outputs (theta, rotation matrices, posed joints, transforms).
constexpr int kSmallN = 4;
constexpr int kSmallVB = 32;                       // vertices per workgroup
__global__ void __launch_bounds__(256) smpl_small_kernel(SmplConsts c, int maxdepth, PrepIn in, int N,
                                                         float* __restrict__ Amat, float* __restrict__ posed,
                                                         float* __restrict__ rotmat, float* __restrict__ theta,
                                                         float* __restrict__ verts) {
  __shared__ __attribute__((aligned(16))) float sA[kSmallN][kNJ * 12];
  __shared__ __attribute__((aligned(16))) float spf[kSmallN][kBlendK];
  __shared__ float svp[kSmallN][3 * kSmallVB];
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // blend rows: 8 lanes per row (28 consecutive k each = 7 x 16 bytes), 8 rows per wave and pass, 3 passes per wave
  constexpr int NPASS = 3 * kSmallVB / 32, KQ = kBlendK / 8 / 4;
  static_assert(kBlendK == 8 * 4 * KQ && NPASS * 32 == 3 * kSmallVB, "row cut");
  const int row0 = blockIdx.x * 3 * kSmallVB;
  const int sub = lane & 7;
  f4 wv[NPASS][KQ];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int gr = row0 + wave * (8 * NPASS) + i * 8 + (lane >> 3);
#pragma unroll
    for (int q = 0; q < KQ; ++q)
      wv[i][q] = gr < 3 * kNV ? *(const f4*)(c.blendW + (long)gr * kBlendK + sub * (4 * KQ) + 4 * q) : f4{0.f, 0.f, 0.f, 0.f};
  }
  // skinning operands of this thread's (person, vertex)
  const int sp = threadIdx.x / kSmallVB, sv = blockIdx.x * kSmallVB + (threadIdx.x % kSmallVB);
  const bool sok = sp < N && sv < kNV;
  int jx[4];
  float wj[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { jx[k] = sok ? c.lbs_cidx[sv * 4 + k] * 12 : 0; wj[k] = sok ? c.lbs_cval[sv * 4 + k] : 0.f; }
  if (wave < N) {
    const bool out = blockIdx.x == 0;
    PrepIn q = in;
    q.pf_hi = nullptr; q.pf_lo = nullptr;
    smpl_prep_person(c, maxdepth, q, wave, lane, sA[wave], spf[wave], out && posed ? posed + (long)wave * kNJ * 3 : nullptr,
                     out && rotmat ? rotmat + (long)wave * kNJ * 9 : nullptr, out && theta ? theta + (long)wave * kTheta : nullptr);
  }
  __syncthreads();
  if (blockIdx.x == 0 && Amat)
    for (int i = threadIdx.x; i < N * kNJ * 12; i += 256) Amat[i] = sA[i / (kNJ * 12)][i % (kNJ * 12)];
  for (int p = 0; p < N; ++p) {
    f4 f[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) f[q] = *(const f4*)(&spf[p][sub * (4 * KQ) + 4 * q]);
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < KQ; ++q) a += wv[i][q][0] * f[q][0] + wv[i][q][1] * f[q][1] + wv[i][q][2] * f[q][2] + wv[i][q][3] * f[q][3];
      a += __shfl_xor(a, 1);
      a += __shfl_xor(a, 2);
      a += __shfl_xor(a, 4);
      if (sub == 0) svp[p][wave * (8 * NPASS) + i * 8 + (lane >> 3)] = a;
    }
  }
  __syncthreads();
  if (sok) {
    float t[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) t[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f4 r0 = *(const f4*)(&sA[sp][jx[k]]), r1 = *(const f4*)(&sA[sp][jx[k] + 4]), r2 = *(const f4*)(&sA[sp][jx[k] + 8]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { t[e] += wj[k] * r0[e]; t[4 + e] += wj[k] * r1[e]; t[8 + e] += wj[k] * r2[e]; }
    }
    const int lv = threadIdx.x % kSmallVB;
    const float x = svp[sp][3 * lv], y = svp[sp][3 * lv + 1], z = svp[sp][3 * lv + 2];
    float* o = verts + ((long)sp * kNV + sv) * 3;
    o[0] = t[0] * x + t[1] * y + t[2] * z + t[3];
    o[1] = t[4] * x + t[5] * y + t[6] * z + t[7];
    o[2] = t[8] * x + t[9] * y + t[10] * z + t[11];
  }
}

bool smpl_small_rows_ok(int N, const Options& o) {
  const int max_n = o.smpl_small_max_n < 0 ? 0 : (o.smpl_small_max_n > kSmallN ? kSmallN : o.smpl_small_max_n);
  return N >= 1 && N <= max_n && 256 / kSmallVB >= N;
}
bool smpl_small_ok(const SmplConsts& c, int N, const Options& o) { return c.lbs_sparse && smpl_small_rows_ok(N, o); }

// in: PrepIn-style source (mode 0: regressor state rows xs; 1: axis-angle; 2: rotation matrices)
hipError_t launch_smpl_small(const SmplConsts& c, int mode, const float* pose, int pose_ld, const float* betas, int betas_ld,
                             const float* cam, int cam_ld, int N, float* Amat, float* posed, float* rotmat, float* theta,
                             float* verts, hipStream_t s) {
  PrepIn in{pose, pose_ld, betas, betas_ld, cam, cam_ld, mode, nullptr, nullptr, 0};
  hipLaunchKernelGGL(smpl_small_kernel, dim3((kNV + kSmallVB - 1) / kSmallVB), dim3(256), 0, s, c, c.maxdepth, in, N, Amat,
                     posed, rotmat, theta, verts);
  return hipGetLastError();
}

// ------------------------------------------------------------------ joints + projection
// block per person.  Regressed joints = CSR rows (9 extra rows always; 17 h36m rows when the
// evaluation regressor is given).  kp_3d = 14 (h36m path) or 49 joints; kp_2d = projection.
template <int NT>
__global__ void __launch_bounds__(NT) smpl_joints_kernel(SmplConsts c, JregPacked jr, int use_jr,
                                                          const float* __restrict__ verts,
                                                          const float* __restrict__ posed,
                                                          const float* __restrict__ xs, int N,
                                                          float* __restrict__ kp3d,
                                                          float* __restrict__ kp2d) {
  __shared__ float reg[17][3];
  const int p = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* V = verts + (long)p * kNV * 3;
  const int nrows = use_jr ? 17 : 9;
  const int* ptr = use_jr ? jr.ptr : c.xr_ptr;
  const int* idx = use_jr ? jr.idx : c.xr_idx;
  const float* val = use_jr ? jr.val : c.xr_val;
  for (int row = wave; row < nrows; row += NT / 64) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int e1 = ptr[row + 1];
    for (int e = ptr[row] + lane; e < e1; e += 64) {
      const float wv = val[e];
      const float* q = V + (long)idx[e] * 3;
      a0 += wv * q[0]; a1 += wv * q[1]; a2 += wv * q[2];
    }
    for (int o = 32; o > 0; o >>= 1) {
      a0 += __shfl_down(a0, o); a1 += __shfl_down(a1, o); a2 += __shfl_down(a2, o);
    }
    if (lane == 0) { reg[row][0] = a0; reg[row][1] = a1; reg[row][2] = a2; }
  }
  __syncthreads();
  const int nj = use_jr ? 14 : 49;
  const int k = threadIdx.x;
  if (k < nj) {
    float q[3];
    if (use_jr) {
      const int s = kH36mToJ14[k];
      q[0] = reg[s][0]; q[1] = reg[s][1]; q[2] = reg[s][2];
    } else {
      const int s = kJointMap49[k];
      if (s < 24) {
        const float* ps = posed + ((long)p * kNJ + s) * 3;
        q[0] = ps[0]; q[1] = ps[1]; q[2] = ps[2];
      } else if (s < 45) {
        const float* ps = V + (long)kExtraVerts[s - 24] * 3;
        q[0] = ps[0]; q[1] = ps[1]; q[2] = ps[2];
      } else {
        q[0] = reg[s - 45][0]; q[1] = reg[s - 45][1]; q[2] = reg[s - 45][2];
      }
    }
    float* o3 = kp3d + ((long)p * nj + k) * 3;
    o3[0] = q[0]; o3[1] = q[1]; o3[2] = q[2];
    // spin.py:307-351: t = [cam1, cam2, 2*5000/(224*cam0 + 1e-9)], R = I, centre 0
    if (kp2d) {
      const float* cam = xs + (long)p * kState + 154;
      const float tz = 10000.f / (224.f * cam[0] + 1e-9f);
      const float px = q[0] + cam[1], py = q[1] + cam[2], pz = q[2] + tz;
      float* o2 = kp2d + ((long)p * nj + k) * 2;
      o2[0] = (5000.f * (px / pz)) / 112.f;
      o2[1] = (5000.f * (py / pz)) / 112.f;
    }
  }
}

hipError_t launch_smpl_joints(const SmplConsts& c, const JregPacked* jr, const float* verts,
                              const float* posed, const float* xs, int N, float* kp3d, float* kp2d,
                              hipStream_t s) {
  if (N <= 0) return hipSuccess;
  JregPacked j = jr ? *jr : JregPacked{nullptr, nullptr, nullptr};
  if (N <= 16)          // a handful of persons: 16 waves per person share the 9 / 17 regressor rows (8.0 -> us at N = 1)
    hipLaunchKernelGGL(smpl_joints_kernel<1024>, dim3(N), dim3(1024), 0, s, c, j, jr ? 1 : 0, verts, posed, xs, N, kp3d, kp2d);
  else
    hipLaunchKernelGGL(smpl_joints_kernel<256>, dim3(N), dim3(256), 0, s, c, j, jr ? 1 : 0, verts, posed, xs, N, kp3d, kp2d);
  return hipGetLastError();
}

hipError_t launch_smpl_prep(const SmplConsts& c, const float* xs, int N, float* pf, float* Amat,
                            float* posed, float* rotmat, float* theta, hipStream_t s, void* pf_hi, void* pf_lo,
                            long pf_kst) {
  if (N <= 0) return hipSuccess;
  PrepIn in{xs, kState, xs + kNPose, kState, xs + 154, kState, 0, (half_t*)pf_hi, (half_t*)pf_lo, pf_kst};
  hipLaunchKernelGGL(smpl_prep_kernel, dim3((N + 3) / 4), dim3(256), 0, s, c, c.maxdepth, in, N, pf, Amat,
                     posed, rotmat, theta);
  return hipGetLastError();
}

// mode 1: pose = axis-angle [N,72]; mode 2: pose = rotation matrices [N,24,3,3]
hipError_t launch_smpl_prep_pose(const SmplConsts& c, int mode, const float* pose, int pose_ld,
                                 const float* betas, int betas_ld, int N, float* pf, float* Amat, float* posed,
                                 hipStream_t s, void* pf_hi, void* pf_lo, long pf_kst) {
  if (N <= 0) return hipSuccess;
  PrepIn in{pose, pose_ld, betas, betas_ld, nullptr, 0, mode, (half_t*)pf_hi, (half_t*)pf_lo, pf_kst};
  hipLaunchKernelGGL(smpl_prep_kernel, dim3((N + 3) / 4), dim3(256), 0, s, c, c.maxdepth, in, N, pf, Amat, posed,
                     (float*)nullptr, (float*)nullptr);
  return hipGetLastError();
}

// ------------------------------------------------------------------ measurement only: blend shapes + skinning per person
// BASELINE.json's north star sketches SMPL as "a single wavefront-per-person kernel".  This is that kernel for the two
// heavy stages (the prep kernel already is one wave per person): a wave owns a person, walks the 6890 vertices 64 at a
// time, forms v_posed = blendW[3v+c][:] . pf[p][:] (224 terms per coordinate, the person's pose feature in LDS) and
// skins it with the person's 24 transforms (LDS).  Every person streams the whole 18.5 MB blend table through L2; the
// product path instead reads it once per 128-person tile as a GEMM on the matrix cores.  Exported only through
// tepose_smpl_fwd_per_person for tools/smpl_per_person_bench.py: never on the product path.
__global__ void __launch_bounds__(256) smpl_person_kernel(const float* __restrict__ blendW, const int* __restrict__ cidx,
                                                          const float* __restrict__ cval, const float* __restrict__ pf,
                                                          const float* __restrict__ Amat, int N, float* __restrict__ verts) {
  __shared__ __attribute__((aligned(16))) float spf[4][kBlendK];
  __shared__ __attribute__((aligned(16))) float sA[4][kNJ * 12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + wave;
  if (p >= N) return;                                       // wave-uniform; no block-wide barrier below
  for (int i = lane; i < kBlendK; i += 64) spf[wave][i] = pf[(long)p * kBlendK + i];
  for (int i = lane; i < kNJ * 12; i += 64) sA[wave][i] = Amat[(long)p * kNJ * 12 + i];
  __builtin_amdgcn_wave_barrier();
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (int v = lane; v < kNV; v += 64) {
    float x[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f4* w = (const f4*)(blendW + (long)(3 * v + c) * kBlendK);
      float acc = 0.f;
      for (int k = 0; k < kBlendK / 4; ++k) {
        const f4 wv = w[k], fv = *(const f4*)(spf[wave] + 4 * k);
        acc += wv[0] * fv[0] + wv[1] * fv[1] + wv[2] * fv[2] + wv[3] * fv[3];
      }
      x[c] = acc;
    }
    float t[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) t[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = cidx[v * 4 + k] * 12;
      const float wgt = cval[v * 4 + k];
#pragma unroll
      for (int e = 0; e < 12; ++e) t[e] += wgt * sA[wave][j + e];
    }
    float* o = verts + ((long)p * kNV + v) * 3;
    o[0] = t[0] * x[0] + t[1] * x[1] + t[2] * x[2] + t[3];
    o[1] = t[4] * x[0] + t[5] * x[1] + t[6] * x[2] + t[7];
    o[2] = t[8] * x[0] + t[9] * x[1] + t[10] * x[2] + t[11];
  }
}

hipError_t launch_smpl_person(const SmplConsts& c, const float* pf, const float* Amat, int N, float* verts, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  if (!c.lbs_sparse) return hipErrorInvalidValue;
  hipLaunchKernelGGL(smpl_person_kernel, dim3((N + 3) / 4), dim3(256), 0, s, c.blendW, c.lbs_cidx, c.lbs_cval, pf, Amat, N,
                     verts);
  return hipGetLastError();
}

// ------------------------------------------------------------------ geometry building blocks
// lib/utils/geometry.py:68-233 and :330-344 as callers use them on their own (lib/utils/demo_utils.py:112,
// lib/data_utils/threedpw_utils.py:98); the same device functions the prep kernel inlines.
__global__ void __launch_bounds__(256) rotmat_to_aa_kernel(const float* __restrict__ R, int N, float* __restrict__ aa) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float M[3][3], a[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) M[r][c] = R[(long)i * 9 + 3 * r + c];
  rotmat_to_aa(M, a);
  aa[(long)i * 3 + 0] = a[0]; aa[(long)i * 3 + 1] = a[1]; aa[(long)i * 3 + 2] = a[2];
}

__global__ void __launch_bounds__(256) rot6d_to_rotmat_kernel(const float* __restrict__ x6, int N, float* __restrict__ R) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float M[3][3];
  rot6d_to_rotmat(x6 + (long)i * 6, M);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[(long)i * 9 + 3 * r + c] = M[r][c];
}

hipError_t launch_rotmat_to_aa(const float* R, int N, float* aa, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(rotmat_to_aa_kernel, dim3((N + 255) / 256), dim3(256), 0, s, R, N, aa);
  return hipGetLastError();
}

hipError_t launch_rot6d_to_rotmat(const float* x6, int N, float* R, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  hipLaunchKernelGGL(rot6d_to_rotmat_kernel, dim3((N + 255) / 256), dim3(256), 0, s, x6, N, R);
  return hipGetLastError();
}

}  // namespace tepose
