/* Host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer, on a machine WITHOUT a GPU (tools/sanitize_host.sh builds csrc/ with
 * -fsanitize=address,undefined on the host side -- device code as shipped -- and links this program against it; tests/test_sanitize_host.py runs both).  csrc/api.hip is ~2 400 lines of pointer arithmetic
 * over a 770 MB blob layout, paddings, CSR packing and kernel selection, all of it reachable without a device: handle creation and layout for many (layers,
 * hidden) pairs, packed / workspace sizes over the batch-class grid, the kernel selection at every class boundary, the option table, the fp32 section list,
 * and every argument-error path of every entry point (NULL handles, NULL buffers, nothing packed yet) -- which must answer with an error code, never with a
 * crash or a sanitizer report.  No compute call can succeed here and none is expected to; a call that gets as far as the HIP runtime returns its error.
 * Prints "host check ok: N calls" and exits 0; any unexpected return value exits 1. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tepose_amd.h"

static long calls = 0;
#define EXPECT(cond) do { ++calls; if (!(cond)) { fprintf(stderr, "host check FAILED at line %d: %s\n", __LINE__, #cond); return 1; } } while (0)
#define NEG(expr) do { ++calls; const long r_ = (long)(expr); if (r_ >= 0) { fprintf(stderr, "host check: expected an error, got %ld at line %d: %s\n", r_, __LINE__, #expr); return 1; } } while (0)
#define ANY(expr) do { ++calls; (void)(expr); } while (0)

int main(void) {
  EXPECT(tepose_version() == TEPOSE_ABI_VERSION);
  EXPECT(tepose_build_info() != NULL && strstr(tepose_build_info(), "packed_fp32") != NULL);
  for (int c = -12; c <= 4; ++c) EXPECT(tepose_error_string(c) != NULL);

  tepose_model* m = NULL;
  NEG(tepose_create(0, 1024, &m));
  NEG(tepose_create(2, 0, &m));
  NEG(tepose_create(2, 1024, NULL));
  NEG(tepose_create(9, 1024, &m));
  NEG(tepose_create(2, 8193, &m));
  tepose_destroy(NULL);

  /* layouts: odd hidden sizes (padding to 64), one to eight layers, the largest handle the ABI admits */
  static const int shapes[][2] = {{1, 1}, {1, 64}, {2, 1024}, {3, 100}, {2, 1000}, {2, 192}, {2, 2048}, {4, 320}, {8, 513}, {2, 8192}};
  static const int Bs[] = {1, 2, 4, 5, 16, 17, 32, 33, 37, 48, 49, 64, 65, 128, 129, 192, 193, 511, 512, 639, 640, 648, 656, 768, 1024, 2048, 8192, 20000};
  static const int Ts[] = {1, 2, 3, 6, 16, 32, 36, 37, 64};
  for (size_t si = 0; si < sizeof(shapes) / sizeof(shapes[0]); ++si) {
    m = NULL;
    EXPECT(tepose_create(shapes[si][0], shapes[si][1], &m) == 0 && m != NULL);
    EXPECT(tepose_set_option(m, "ASSUME_CUS", 256) == 0);
    const size_t bytes = tepose_packed_bytes(m);
    EXPECT(bytes > 0 && bytes % 4 == 0);
    EXPECT(tepose_kernel_info(m) != NULL);
    EXPECT(tepose_workspace_bytes(m, 0, 16) == 0 && tepose_workspace_bytes(m, 8, 0) == 0 && tepose_workspace_bytes(NULL, 8, 8) == 0);
    size_t prev = 0;
    for (size_t bi = 0; bi < sizeof(Bs) / sizeof(Bs[0]); ++bi)
      for (size_t ti = 0; ti < sizeof(Ts) / sizeof(Ts[0]); ++ti) {
        const size_t w = tepose_workspace_bytes(m, Bs[bi], Ts[ti]);
        EXPECT(w > 0);
        prev ^= w;
        const char* sel = tepose_select_kernels(m, Bs[bi], Ts[ti]);
        EXPECT(sel != NULL && strstr(sel, "projection=") != NULL && strstr(sel, "gru_step=") != NULL && strstr(sel, "smpl=") != NULL);
        ANY(tepose_uses_persistent(m, Bs[bi], Ts[ti]));
      }
    (void)prev;
    for (size_t bi = 0; bi < sizeof(Bs) / sizeof(Bs[0]); ++bi) EXPECT(tepose_project_frames_workspace_bytes(m, Bs[bi]) > 0);
    /* the fp32 section list: ascending, inside the blob, disjoint */
    size_t off[8], len[8];
    NEG(tepose_fp32_ranges(m, NULL, NULL, 0));
    NEG(tepose_fp32_ranges(m, off, len, 1));                           /* a buffer that is too small is refused, not overrun */
    const int n = tepose_fp32_ranges(m, off, len, 8);
    EXPECT(n > 1 && n <= 8);
    size_t end = 0, total = 0;
    for (int i = 0; i < n; ++i) {
      EXPECT(off[i] >= end && len[i] > 0 && off[i] + len[i] <= bytes);
      end = off[i] + len[i];
      total += len[i];
    }
    EXPECT(total < bytes);
    /* nothing packed, no blob: every compute entry point answers with an error code before it touches a device */
    float* fake = (float*)(uintptr_t)4096;     /* never dereferenced on the host */
    NEG(tepose_encoder_fwd(m, fake, 1, 6, 0, fake, fake, 1 << 20, NULL));
    NEG(tepose_forward(m, fake, 1, 6, NULL, fake, fake, fake, fake, fake, fake, 1 << 20, NULL));
    NEG(tepose_regressor_fwd(m, fake, 1, 3, NULL, fake, fake, fake, fake, fake, fake, 1 << 20, NULL));
    NEG(tepose_regressor_fwd_init(m, fake, 1, 3, NULL, NULL, NULL, NULL, fake, fake, fake, fake, fake, fake, 1 << 20, NULL));
    NEG(tepose_project_frames(m, fake, 2048, fake, 85, 1, fake, 9216, fake, 1 << 20, NULL));
    NEG(tepose_smpl_fwd(m, 1, fake, fake, 1, fake, fake, fake, 1 << 20, NULL));
    NEG(tepose_smpl_verts_from_theta(m, fake, 1, fake, fake, 1 << 20, NULL));
    NEG(tepose_adopt_blob(m));
    NEG(tepose_derive_planes(m, NULL));
    NEG(tepose_pack_encoder(m, NULL, 12 * shapes[si][0] + 4, NULL));
    NEG(tepose_pack_regressor(m, NULL, 13, NULL));
    NEG(tepose_set_blob(m, NULL, bytes));
    EXPECT(tepose_status_peek(m) == 0 && tepose_fault_code(m) == 0);
    EXPECT(tepose_set_persistent(m, 0) == 0 && tepose_uses_persistent(m, 1, 16) == 0 && tepose_set_persistent(m, 1) == 0);
    EXPECT(tepose_debug_set_test_fault(m, 3u) == 0 && tepose_debug_set_test_fault(m, 0u) == 0);
    EXPECT(tepose_profile_enable(m, 1) == 0 && tepose_profile_enable(m, 0) == 0);
    double ms = 0, fl = 0;
    int cnt = 0;
    EXPECT(tepose_profile_read(m, &ms, &cnt, &fl) == 0 && cnt == 0);
    EXPECT(tepose_profile_read_l1proj(m, &ms, &cnt, &fl) == 0 && cnt == 0);
    EXPECT(tepose_profile_read_gru(m, &ms, &cnt, &fl) == 0 && cnt == 0);
    NEG(tepose_profile_read(m, NULL, &cnt, &fl));
    tepose_destroy(m);
  }

  /* the option table: every name, with and without the prefix; per handle */
  static const char* names[] = {"SKINNY_MAX_M", "SKINNY_MAX_M_GEMM", "SPLIT_MIN_M", "SKINNY_H3_MAX_M", "GEMM_HALF_MAX_BLOCKS", "SPLIT_FEW_MAX_ROWS", "H3_TILE",
                                "H3_TILE64", "H3_TILE192", "S16_GM", "GRU_GM", "SEQ_GRAN_MAX_M", "SEQ_MAX_M", "REG_SEQ_MAX_N", "ASSUME_CUS", "SKINNY_NARROW64", "SKINNY_MT1",
                                "SKINNY_NT1_BELOW", "SKINNY_W8", "SMPL_SMALL_MAX_N", "L1_SKINNY_MAX_ROWS", "G0_MID_MIN_ROWS", "G0_SKINNY_MAX_M", "S_MIN_B",
                                "BLEND16_MIN_N", "GI_BLK", "SEQ_SPIN_LIMIT"};
  tepose_model *a = NULL, *b = NULL;
  EXPECT(tepose_create(2, 1024, &a) == 0 && tepose_create(2, 1024, &b) == 0);
  for (size_t i = 0; i < sizeof(names) / sizeof(names[0]); ++i) {
    char pref[64];
    snprintf(pref, sizeof(pref), "TEPOSE_%s", names[i]);
    const long before = tepose_get_option(a, names[i]);
    EXPECT(before == tepose_get_option(a, pref));
    EXPECT(tepose_set_option(b, pref, 7) == 0 && tepose_get_option(b, names[i]) == 7);
    EXPECT(tepose_get_option(a, names[i]) == before);                 /* the other handle is untouched */
    for (int Bq = 1; Bq <= 8192; Bq *= 2) EXPECT(tepose_select_kernels(b, Bq, 16) != NULL && tepose_workspace_bytes(b, Bq, 16) > 0);   /* any value plans without UB */
  }
  NEG(tepose_set_option(a, "NO_SUCH_OPTION", 1));
  NEG(tepose_set_option(a, NULL, 1));
  NEG(tepose_set_option(NULL, "S_MIN_B", 1));
  NEG(tepose_set_option(a, "SEQ_SPIN_LIMIT", 0));
  EXPECT(tepose_get_option(a, "NO_SUCH_OPTION") == -1 && tepose_get_option(NULL, "S_MIN_B") == -1 && tepose_get_option(a, NULL) == -1);
  /* extreme option values must not break the planners */
  static const long extremes[] = {-2147483647L, -1, 0, 1, 2147483647L};
  for (size_t i = 0; i < sizeof(names) / sizeof(names[0]); ++i)
    for (size_t e = 0; e < sizeof(extremes) / sizeof(extremes[0]); ++e) {
      ANY(tepose_set_option(b, names[i], extremes[e]));
      for (size_t bi = 0; bi < sizeof(Bs) / sizeof(Bs[0]); ++bi) {
        EXPECT(tepose_select_kernels(b, Bs[bi], 16) != NULL);
        ANY(tepose_workspace_bytes(b, Bs[bi], 16));
        ANY(tepose_uses_persistent(b, Bs[bi], 16));
      }
      ANY(tepose_set_option(b, names[i], tepose_get_option(a, names[i])));
    }
  tepose_destroy(a);
  tepose_destroy(b);

  /* VIBE bootstrap handles: every constructor flag */
  for (int bidir = 0; bidir <= 1; ++bidir)
    for (int lin = 0; lin <= 1; ++lin) {
      tepose_model* v = NULL;
      EXPECT(tepose_create_vibe_ex(2, 100, bidir, lin, &v) == 0 && v != NULL);
      EXPECT(tepose_packed_bytes(v) > 0 && tepose_vibe_feature_dim(v) > 0);
      for (int Bq = 1; Bq <= 64; Bq *= 4) EXPECT(tepose_vibe_workspace_bytes(v, Bq, 16) > 0);
      float* fake = (float*)(uintptr_t)4096;
      NEG(tepose_vibe_encoder_fwd(v, fake, 1, 16, 1, fake, fake, 1 << 20, NULL));
      NEG(tepose_pack_vibe_encoder(v, NULL, 4, NULL));
      NEG(tepose_encoder_fwd(v, fake, 1, 6, 0, fake, fake, 1 << 20, NULL));      /* the wrong kind of handle */
      tepose_destroy(v);
    }
  tepose_model* v = NULL;
  EXPECT(tepose_create_vibe(2, 1024, &v) == 0);
  tepose_destroy(v);
  NEG(tepose_create_vibe(0, 1024, &v));
  NEG(tepose_create_vibe_ex(2, 1024, 0, 1, NULL));

  /* NULL handles and buffers on every entry point */
  float* fake = (float*)(uintptr_t)4096;
  NEG(tepose_encoder_fwd(NULL, fake, 1, 6, 0, fake, fake, 1 << 20, NULL));
  NEG(tepose_forward(NULL, fake, 1, 6, NULL, fake, fake, fake, fake, fake, fake, 1 << 20, NULL));
  NEG(tepose_regressor_fwd(NULL, fake, 1, 3, NULL, fake, fake, fake, fake, fake, fake, 1 << 20, NULL));
  NEG(tepose_forward_status(NULL, fake, NULL));
  NEG(tepose_status(NULL, NULL));
  NEG(tepose_status_peek(NULL));
  NEG(tepose_fault_code(NULL));
  NEG(tepose_set_persistent(NULL, 1));
  EXPECT(tepose_uses_persistent(NULL, 1, 16) == 0);
  NEG(tepose_debug_set_test_fault(NULL, 1u));
  EXPECT(tepose_kernel_info(NULL) == NULL || tepose_kernel_info(NULL)[0] == 0);
  EXPECT(tepose_select_kernels(NULL, 8, 16) == NULL || tepose_select_kernels(NULL, 8, 16)[0] == 0);
  EXPECT(tepose_packed_bytes(NULL) == 0);
  NEG(tepose_fp32_ranges(NULL, NULL, NULL, 0));
  NEG(tepose_pack_jreg(NULL, fake, NULL));
  NEG(tepose_metrics_joints(NULL, fake, 4, 14, 0, fake, fake, fake, NULL));
  NEG(tepose_metrics_verts(NULL, fake, 4, fake, NULL));
  NEG(tepose_filter_one_euro(NULL, 4, 72, 0.004f, 0.7f, 1.0f, NULL));
  NEG(tepose_filter_slerp(NULL, fake, 4, 24, 0.3, NULL));
  NEG(tepose_rotmat_to_angle_axis(NULL, 4, fake, NULL));
  NEG(tepose_rot6d_to_rotmat(NULL, 4, fake, NULL));
  NEG(tepose_gemm_f32(NULL, 32, fake, 32, NULL, fake, 32, 4, 4, 32, 0, fake, 1 << 20, NULL));
  NEG(tepose_gemm_h3_f32(NULL, 32, fake, 32, NULL, fake, 32, 4, 4, 32, fake, 1 << 20, NULL));
  NEG(tepose_profile_enable(NULL, 1));
  EXPECT(tepose_gemm_workspace_bytes(0, 32) == 0 && tepose_gemm_workspace_bytes(100, 2144) > 0);
  EXPECT(tepose_gemm_h3_workspace_bytes(0, 4, 32) == 0 && tepose_gemm_h3_workspace_bytes(100, 300, 2144) > 0);
  EXPECT(tepose_jreg_packed_bytes() > 0);
  ANY(tepose_debug_kernel_errors());
  printf("host check ok: %ld calls\n", calls);
  return 0;
}
