/* A caller of the C ABI with no Python, no torch and no C++ in it: plain C + the HIP runtime for device memory.
 * It is what a maintainer's FFI (cgo / JNI / ctypes ...) does, in order: create the handle, give it a device blob, pack the
 * three weight groups from device tensors in the reference's state-dict order, ask for the workspace, run TePose.forward.
 * tests/test_gpu_c_client.py writes the tensors of a tepose_amd Python model to a file, runs this program on it and
 * requires the outputs to be bit-identical to the Python drop-in's (same library, same kernels, same bits).
 *
 *   input file:  int32 L, H, B, T, use_jreg;  then float32 tensors back to back:
 *     12*L+4 encoder tensors | 13 regressor tensors | v_template, shapedirs, posedirs, J_regressor, lbs_weights,
 *     J_regressor_extra | int32 parents[24] | J_regressor_h36m[17*6890] (when use_jreg) | x[B*T*2133]
 *   output file: float32 theta[B*85] verts[B*6890*3] kp_3d[B*J*3] kp_2d[B*J*2] rotmat[B*216]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/tepose_amd.h"

#define CHECK_HIP(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define CHECK_TE(e) do { int e_ = (e); if (e_ != 0) { fprintf(stderr, "tepose error %d (%s) at line %d\n", e_, tepose_error_string(e_), __LINE__); return 3; } } while (0)

static float* upload(FILE* f, size_t n) {
  float* h = (float*)malloc(n * sizeof(float));
  float* d = NULL;
  if (!h || fread(h, sizeof(float), n, f) != n) { fprintf(stderr, "short input file\n"); exit(4); }
  if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess ||
      hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); exit(5); }
  free(h);
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s <input> <output>\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[5];
  if (fread(hdr, sizeof(int32_t), 5, f) != 5) return 1;
  const int L = hdr[0], H = hdr[1], B = hdr[2], T = hdr[3], use_jreg = hdr[4];
  if (tepose_version() != TEPOSE_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

  tepose_model* m = NULL;
  CHECK_TE(tepose_create(L, H, &m));
  const size_t blob_bytes = tepose_packed_bytes(m);
  void* blob = NULL;
  CHECK_HIP(hipMalloc(&blob, blob_bytes));
  CHECK_HIP(hipMemset(blob, 0, blob_bytes));
  CHECK_TE(tepose_set_blob(m, blob, blob_bytes));
  hipStream_t s;
  CHECK_HIP(hipStreamCreate(&s));

  /* encoder: nn.GRU / nn.Linear shapes (lib/models/tepose.py:53-69) */
  const int n_enc = 12 * L + 4;
  const float** enc = (const float**)malloc(n_enc * sizeof(float*));
  int k = 0;
  for (int l = 0; l < L; ++l) {                                       /* gru_fwd */
    const size_t in = l == 0 ? TEPOSE_INPUT : (size_t)H;
    enc[k++] = upload(f, 3 * (size_t)H * in); enc[k++] = upload(f, 3 * (size_t)H * H);
    enc[k++] = upload(f, 3 * (size_t)H);      enc[k++] = upload(f, 3 * (size_t)H);
  }
  for (int l = 0; l < L; ++l)                                         /* gru_rec: forward, then reverse */
    for (int d = 0; d < 2; ++d) {
      const size_t in = l == 0 ? TEPOSE_INPUT : 2 * (size_t)H;
      enc[k++] = upload(f, 3 * (size_t)H * in); enc[k++] = upload(f, 3 * (size_t)H * H);
      enc[k++] = upload(f, 3 * (size_t)H);      enc[k++] = upload(f, 3 * (size_t)H);
    }
  enc[k++] = upload(f, (size_t)TEPOSE_FEAT * H);     enc[k++] = upload(f, TEPOSE_FEAT);
  enc[k++] = upload(f, (size_t)TEPOSE_FEAT * 2 * H); enc[k++] = upload(f, TEPOSE_FEAT);
  CHECK_TE(tepose_pack_encoder(m, enc, n_enc, s));

  const size_t reg_n[13] = {1024 * 2205, 1024, 1024 * 1024, 1024, 144 * 1024, 144, 10 * 1024, 10, 3 * 1024, 3, 144, 10, 3};
  const float* reg[13];
  for (int i = 0; i < 13; ++i) reg[i] = upload(f, reg_n[i]);
  CHECK_TE(tepose_pack_regressor(m, reg, 13, s));

  const size_t NV = TEPOSE_NVERT;
  const size_t smpl_n[6] = {NV * 3, NV * 3 * 10, 207 * NV * 3, 24 * NV, NV * 24, 9 * NV};
  const float* sm[6];
  for (int i = 0; i < 6; ++i) sm[i] = upload(f, smpl_n[i]);
  int32_t parents[24];
  if (fread(parents, sizeof(int32_t), 24, f) != 24) return 1;
  CHECK_TE(tepose_pack_smpl(m, sm[0], sm[1], sm[2], sm[3], sm[4], sm[5], parents, s));

  void* jreg = NULL;
  if (use_jreg) {
    const float* J = upload(f, 17 * NV);
    CHECK_HIP(hipMalloc(&jreg, tepose_jreg_packed_bytes()));
    CHECK_HIP(hipMemset(jreg, 0, tepose_jreg_packed_bytes()));
    CHECK_TE(tepose_pack_jreg(J, jreg, s));
  }
  const float* x = upload(f, (size_t)B * T * TEPOSE_INPUT);
  fclose(f);

  const int J3 = use_jreg ? 14 : 49;
  const size_t out_n[5] = {(size_t)B * TEPOSE_THETA, (size_t)B * NV * 3, (size_t)B * J3 * 3, (size_t)B * J3 * 2, (size_t)B * 216};
  float* out[5];
  for (int i = 0; i < 5; ++i) CHECK_HIP(hipMalloc((void**)&out[i], out_n[i] * sizeof(float)));
  const size_t ws_bytes = tepose_workspace_bytes(m, B, T);
  void* ws = NULL;
  CHECK_HIP(hipMalloc(&ws, ws_bytes));
  for (int rep = 0; rep < 2; ++rep)                                    /* the second call reuses handle and workspace */
    CHECK_TE(tepose_forward(m, x, B, T, jreg, out[0], out[1], out[2], out[3], out[4], ws, ws_bytes, s));
  /* a workspace one byte short is refused, not overrun */
  if (tepose_forward(m, x, B, T, jreg, out[0], out[1], out[2], out[3], out[4], ws, ws_bytes - 1, s) != TEPOSE_E_WORKSPACE) {
    fprintf(stderr, "short workspace was accepted\n");
    return 6;
  }
  /* before the outputs are trusted: synchronise and ask whether a persistent small-batch kernel gave up (TEPOSE_E_TIMEOUT);
     the reference would have raised an exception (evaluate.py:255) -- a C caller gets the code here and may retry after
     tepose_set_persistent(m, 0) */
  CHECK_TE(tepose_status(m, s));

  FILE* g = fopen(argv[2], "wb");
  if (!g) return 1;
  for (int i = 0; i < 5; ++i) {
    float* h = (float*)malloc(out_n[i] * sizeof(float));
    CHECK_HIP(hipMemcpy(h, out[i], out_n[i] * sizeof(float), hipMemcpyDeviceToHost));
    fwrite(h, sizeof(float), out_n[i], g);
    free(h);
  }
  fclose(g);
  tepose_destroy(m);
  printf("tepose_client ok: L=%d H=%d B=%d T=%d J=%d blob=%zu ws=%zu\n", L, H, B, T, J3, blob_bytes, ws_bytes);
  return 0;
}
