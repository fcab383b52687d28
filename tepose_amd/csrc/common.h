// Internal declarations shared by the HIP translation units of libtepose_hip.so.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts, fp32-input MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tepose {

typedef _Float16 half_t;
// split-precision operands: a = hi + lo / kLoScale with hi = fp16(a), lo = fp16((a - hi) * kLoScale); the power-of-two
// scale keeps the low half out of the fp16 subnormal range for every |a| >= ~1e-7 (22 significant bits)
constexpr float kLoScale = 2048.f;
__host__ __device__ inline void split_hi_lo(float a, half_t& hi, half_t& lo) {
  hi = (half_t)a;
  lo = (half_t)((a - (float)hi) * kLoScale);
}

constexpr int kFeat = 2048;
constexpr int kTheta = 85;
constexpr int kInput = 2133;
constexpr int kInputP = 2144;   // 2133 padded to a multiple of the GEMM K-tile (32)
constexpr int kNV = 6890;
constexpr int kNJ = 24;
constexpr int kNPose = 144;
constexpr int kState = 160;     // regressor state row: pose6d(144) | shape(10) | cam(3) | pad(3)
constexpr int kBlendK = 224;    // [1 | betas(10) | pose_feature(207) | pad(6)]
constexpr int kBlendN = 20736;  // 3*6890 = 20670 padded to a multiple of 128
constexpr int kVertLd = 20672;  // row stride of the v_posed scratch (>= 20670, multiple of 4)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- per-handle options
// Every launch threshold of the library.  A handle owns ONE Options object: tepose_create fills it from the environment (options_from_env, the only
// place the library reads TEPOSE_* threshold variables), tepose_set_option(m, "NAME", value) changes a field before anything is packed, and every
// launcher that consults a threshold takes the handle's object as an argument -- no function-local static, no process-wide state: two handles of one
// process can differ (SURVEY.md 8b "no global mutable state").  Defaults = the measured best.
struct Options {
  int skinny_max_m = 768;            // TEPOSE_SKINNY_MAX_M: rows at or below which the width-first fp32 kernels (skinny.hip) win over 128-row tiles (measured crossover)
  int skinny_max_m_gemm = -1;        // TEPOSE_SKINNY_MAX_M_GEMM: the same for plain products only (-1: skinny_max_m)
  int split_min_m = 0;               // TEPOSE_SPLIT_MIN_M: rows ABOVE which a split-mode handle runs its matmuls on the fp16x3 kernels (round 2: they win at every batch size)
  int skinny_h3_max_m = 128;         // TEPOSE_SKINNY_H3_MAX_M: rows up to which a split-mode GRU step uses the width-first kernel of skinny_h3.hip
  int gemm_half_max_blocks = 1024;   // TEPOSE_GEMM_HALF_MAX_BLOCKS: 64-row tiles of gemm_f32_kernel while 128-row tiles would make at most this many blocks
  int split_few_max_rows = 1024;     // TEPOSE_SPLIT_FEW_MAX_ROWS: rows up to which launch_split_rows runs one workgroup per row (222 rows 12.3 -> 5 us, 1024 rows 13 -> 8)
  int h3_tile = 0;                   // TEPOSE_H3_TILE: A/B, force a tile shape of gemm_h3_kernel (64 / 192 / 256)
  int h3_tile64 = 1;                 // TEPOSE_H3_TILE64: 0 = never pick 64-row tiles (A/B)
  int h3_tile192 = 1;                // TEPOSE_H3_TILE192: 0 = never pick 128 x 192 tiles (A/B)
  int s16_gm = 8;                    // TEPOSE_S16_GM: row tiles per XCD group of the barrier-free projection's walk (layer-0 projection ms at 2 / 4 / 8 / 16 / 32: 10.92 / 10.82 / 10.74 / 11.37 / 12.37)
  int gru_gm = 4;                    // TEPOSE_GRU_GM: the same for the fused step (recurrent ms per forward at 1 / 2 / 4 / 8 / 16 / 32: 11.31 / 11.27 / 11.22 / 11.41 / 11.41 / 11.77)
  int seq_gran_max_m = 4;            // TEPOSE_SEQ_GRAN_MAX_M: rows up to which the persistent recurrent kernel hands its state over as tagged granules (B = 1 -14 %, 4 -5 %, 8 +20 %)
  int seq_max_m = 64;                // TEPOSE_SEQ_MAX_M: rows up to which a layer's T steps run as ONE persistent launch (0: never)
  int reg_seq_max_n = 64;            // TEPOSE_REG_SEQ_MAX_N: the same for the persistent FC-loop kernel
  int assume_cus = 0;                // TEPOSE_ASSUME_CUS: plan as if the device had this many CUs (planning without a device: tests/test_dispatch.py)
  int skinny_narrow64 = 1;           // TEPOSE_SKINNY_NARROW64, TEPOSE_SKINNY_MT1, TEPOSE_SKINNY_NT1_BELOW, TEPOSE_SKINNY_W8: block shapes of the width-first
  int skinny_mt1 = 1;                //   split kernel for narrow products (skinny_h3.hip launch_skinny_gemm_h3_batch)
  int skinny_nt1_below = 96;
  int skinny_w8 = 1;
  int smpl_small_max_n = 4;          // TEPOSE_SMPL_SMALL_MAX_N: persons up to which prep + blend shapes + skinning run as one launch
  int l1_skinny_max_rows = 192;      // TEPOSE_L1_SKINNY_MAX_ROWS: real rows up to which a layer >= 1 projection runs on the width-first kernel
  int g0_mid_min_rows = 512;         // TEPOSE_G0_MID_MIN_ROWS: rows from which the layer-0 projection may take the 128 x 288 tiles
  int g0_skinny_max_m = 128;         // TEPOSE_G0_SKINNY_MAX_M: rows up to which the layer-0 projection runs on the width-first kernel
  unsigned long long seq_stamp_ptr = 0;   // TEPOSE_SEQ_STAMP_PTR: device buffer for the per-step time stamps of a -DTEPOSE_SEQ_STAMPS build (tools/seq_stamps.py)
};
// Shape predicates shared by the launchers and by api.hip's describe_plan (tepose_select_kernels), so that the description cannot drift from what
// launches: which input-split kernel, which first-step kernel, granule hand-off, the one-launch SMPL form (the last two additionally need
// what only a packed handle / a launch knows: operand alignment, <= 4 skin weights per vertex).
inline bool split_rows_few_ok(long rows, int Kp, int permT, const Options& o) { return rows <= o.split_few_max_rows && Kp <= 4096 && !permT; }
inline bool gru_first16_shape_ok(int Hp) { return Hp % 128 == 0; }
inline bool gru_seq_shape_ok(int Hp) { return Hp % 256 == 0 && Hp <= 1024; }
Options options_from_env();          // gemm.hip
int* option_field(Options& o, const char* name);   // "SKINNY_MAX_M" (or "TEPOSE_SKINNY_MAX_M") -> &o.skinny_max_m; nullptr: no such option

// ---------------------------------------------------------------- GEMM (gemm.hip)
// C[M,N] = epi( A[M,Kp] * W[Np,Kp]^T ), fp32 operands, fp32 MFMA accumulate.
struct GemmArgs {
  const float* A; long lda;      // rows 16-byte aligned, K zero-padded to Kp by the producer
  const float* W; int Kp;        // packed [Np][Kp], Np multiple of 128, Kp multiple of 32
  float* C; long ldc;
  const float* bias;             // [N] or nullptr
  const float* addend; long ldadd;  // [M,N] or nullptr
  float scale;                   // applied last
  int M, N;
  int relu_a;                    // apply max(0,.) to A on the fly
};
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s, const Options& o);

// One GRU cell step for up to 3 independent directions in one launch:
//   gh = hprev * Whh^T ; r,z,n gate math ; hout = (1-z)*n + z*hprev      (torch.nn.GRU)
struct GruDir {
  const float* hprev; long ldh;  // [M][Hp]; ignored when first
  const float* Whh;              // packed, gate-interleaved tiles (see pack_whh_kernel)
  const float* bhh;              // [3*Hp], natural order g*Hp + j
  const float* gi; long ldgi;    // gi[row*ldgi + g*Hp + j] = x W_ih^T + b_ih
  float* hout; long ldo;
  long gi_blk = 0;               // != 0: gi is in the 16 x 16-blocked layout (gi_blk_offset below), floats between 16-row tiles
  // != nullptr: the fp32 previous / new state in the same 16 x 16 blocks (st_blk_offset below; base of the view's first row tile, floats between row tiles);
  // hprev / hout stay the row-major addresses (they also name the state's planes: EncWs::view16)
  const float* hprev_b = nullptr; long hp_blk = 0;
  float* hout_b = nullptr; long ho_blk = 0;
};
struct GruArgs {
  GruDir d[3];
  int ndir;
  int M, Hp;
  int first;                     // h_{-1} = 0: skip the matmul
};
hipError_t launch_gru_step(const GruArgs& a, hipStream_t s, const Options& o);
// small-M variants (skinny.hip); the launchers above dispatch to them when M <= Options::skinny_max_m
hipError_t launch_skinny_gemm(const GemmArgs& a, hipStream_t s);
hipError_t launch_skinny_gru(const GruArgs& a, hipStream_t s);

// ---------------------------------------------------------------- misc.hip
// dst[np][kp] (row-major [Np][Kp]) = src[rowmap(np)][colmap(kp)] or 0.
enum RowMap { ROW_PLAIN = 0, ROW_GATES = 1, ROW_GATES_TILED = 2 };
enum ColMap { COL_PLAIN = 0, COL_SPLIT2 = 1 };
struct PackArgs {
  const float* src; long ld_src; int N, K;   // logical source [N][K] (col offset folded into src)
  float* dst; int Np, Kp;                     // fp32 destination, or nullptr with dst_hi / dst_lo set
  half_t* dst_hi; half_t* dst_lo;             // blocked split-precision planes (gemm_h3.hip): pointer to this
  long dst_kst;                               // matrix's first row inside the plane, halfs between K-tiles
  int rowmap, colmap;
  int H, Hp;                                  // for the gate / split maps
};
hipError_t launch_pack(const PackArgs& a, hipStream_t s);
hipError_t launch_pad_input(const float* x, float* xp, long rows, hipStream_t s);
hipError_t launch_pad_rows(const float* feat, long feat_ld, const float* theta, long theta_ld, float* xp,
                           long rows, hipStream_t s);
hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s);
hipError_t launch_dmm(const void* A, int a64, long lda, const void* B, int b64, long ldb, const double* Cadd, long ldadd,
                      const float* Cadd32, long ldadd32, double* C, long ldc, int M, int N, int K, double alpha,
                      int add_identity, hipStream_t s);
hipError_t launch_d2f_pad(const double* src, long ld, int rows, int cols, float* dst, int Rp, int Cp, hipStream_t s);
hipError_t launch_copy_cols(const float* src, long lds, const float* add, long lda, float* dst, long ldd, long rows,
                            int cols, hipStream_t s);
hipError_t launch_init_state(const float* init160, float* xs, int N, hipStream_t s);
hipError_t launch_init_state_rows(const float* init160, const float* pose, const float* shape, const float* cam,
                                  float* xs, int N, hipStream_t s);
hipError_t launch_rotmat_to_aa(const float* R, int N, float* aa, hipStream_t s);
hipError_t launch_rot6d_to_rotmat(const float* x6, int N, float* R, hipStream_t s);

// ---------------------------------------------------------------- smpl.hip
struct SmplConsts {              // device pointers into the packed blob
  const float* J0;               // [24][3]      J_regressor * v_template
  const float* JS;               // [24][3][10]  J_regressor * shapedirs
  const float* blendW;           // [kBlendN][kBlendK]  rows (v,c): [v_template | shapedirs | posedirs^T]
  const float* lbsW;             // [6890][24]
  const int* lbs_cidx;           // [6890][4] compacted skin weights (joint index)
  const float* lbs_cval;         // [6890][4]
  int lbs_sparse;                // 1: every vertex has <= 4 non-zero weights -> compact kernel
  const int* parents;            // [24]
  const int* depth;              // [24]
  int maxdepth;                  // deepest level of the tree (host-computed)
  const int* xr_ptr;             // CSR of J_regressor_extra: [10]
  const int* xr_idx;             // [nnz]
  const float* xr_val;
};
struct JregPacked {              // CSR of the optional 17-row evaluation regressor
  const int* ptr; const int* idx; const float* val;
};
// pf_hi / pf_lo: optionally also write the pose-feature rows as blocked planes (view base, pf_kst halfs between 32-column groups)
hipError_t launch_smpl_prep(const SmplConsts& c, const float* xs, int N, float* pf, float* Amat,
                            float* posed, float* rotmat, float* theta, hipStream_t s, void* pf_hi = nullptr,
                            void* pf_lo = nullptr, long pf_kst = 0);
hipError_t launch_smpl_prep_pose(const SmplConsts& c, int mode, const float* pose, int pose_ld,
                                 const float* betas, int betas_ld, int N, float* pf, float* Amat, float* posed,
                                 hipStream_t s, void* pf_hi = nullptr, void* pf_lo = nullptr, long pf_kst = 0);
// measurement only: blend shapes + skinning as one wave per person
bool smpl_small_rows_ok(int N, const Options& o);        // the person-count half of smpl_small_ok (describe_plan: a handle whose tables are not packed yet)
bool smpl_small_ok(const SmplConsts& c, int N, const Options& o);
hipError_t launch_smpl_small(const SmplConsts& c, int mode, const float* pose, int pose_ld, const float* betas, int betas_ld,
                             const float* cam, int cam_ld, int N, float* Amat, float* posed, float* rotmat, float* theta,
                             float* verts, hipStream_t s);
hipError_t launch_smpl_person(const SmplConsts& c, const float* pf, const float* Amat, int N, float* verts, hipStream_t s);
hipError_t launch_lbs_compact(const float* lbsW, int* cidx, float* cval, int* max_nnz, hipStream_t s);
hipError_t launch_smpl_skin(const SmplConsts& c, const float* vposed, const float* Amat, int N,
                            float* verts, hipStream_t s);
hipError_t launch_smpl_joints(const SmplConsts& c, const JregPacked* jr, const float* verts,
                              const float* posed, const float* xs, int N, float* kp3d,
                              float* kp2d, hipStream_t s);
// CSR build (device): counts then fill, rows x cols dense -> ptr/idx/val
hipError_t launch_csr_build(const float* dense, int rows, int cols, int* ptr, int* idx, float* val,
                            int cap, hipStream_t s);
hipError_t launch_smpl_consts(const float* v_template, const float* shapedirs, const float* posedirs,
                              const float* J_regressor, float* J0, float* JS, float* blendW,
                              hipStream_t s);

// ---------------------------------------------------------------- metrics.hip
hipError_t launch_metrics_joints(const float* pred, const float* target, int N, int J, int pelvis_mode,
                                 float* mpjpe, float* pa, float* accel, hipStream_t s);
hipError_t launch_metrics_verts(const float* pred, const float* target, int N, float* mpvpe, hipStream_t s);

// ---------------------------------------------------------------- filters.hip
hipError_t launch_one_euro(float* x, int N, int D, float min_cutoff, float beta, float d_cutoff, hipStream_t s);
hipError_t launch_slerp_smooth(const float* in, float* out, int N, int J, double ratio, hipStream_t s);

// ---------------------------------------------------------------- gemm_h3.hip (split-precision fp16x3 GEMM)
// Operand planes are stored K-tile-blocked and slot-swizzled: element (row, c) of an [R x C] matrix (C multiple
// of 32) lives at
//   ((c / 32) * R + row) * 32 + (((c / 8) % 4) ^ ((row / 4) % 4)) * 8 + c % 8        (halfs)
// * blocked: the 64 bytes a K-tile takes from each of a block's rows form one contiguous run, so an LDS-DMA
//   instruction (16 rows) moves one contiguous KiB = 8 whole 128-byte lines;
// * swizzled in memory: the XOR that makes the LDS image bank-conflict-free for ds_read_b128 is already in
//   the data, so the DMA's global addresses simply ascend with the lane number.
// Measured against row-major planes with the XOR applied in the lane addresses, this layout is worth 0-3 % on
// the layer-0 projection (within box-to-box noise; profiles/r01_README.md) -- it is kept because it makes the
// DMA addressing trivial, not because it is faster.
// A view of rows [r0, r0+M) x columns [c0, c0+K), r0 % 16 == 0 and c0 % 32 == 0, is the pointer to block
// (r0, c0) plus the K-tile stride R * 32.
constexpr int kPlaneK = 32;
__host__ __device__ inline long plane_index(long row, long c, long R) {
  return ((c >> 5) * R + row) * 32 + ((((c >> 3) & 3) ^ ((row >> 2) & 3)) << 3) + (c & 7);
}
struct H3Args {
  const half_t *Ah, *Al; long a_kst;  // A hi / lo planes: pointer to the view's first block, halfs between K-tiles
  const half_t *Wh, *Wl; long w_kst;  // W planes: rows padded to the N-tile, zero beyond N
  int Kp;                             // multiple of 32
  float* C; long ldc;
  const float* bias;                  // [N] or nullptr
  int M, N;
  // optional epilogue extras (zero-initialised when omitted): C = (A W^T + bias + addend) * scale, and the same
  // values written as blocked hi / lo planes (the next product's A operand)
  const float* addend; long ldadd;
  float scale;                        // 0 = no scaling
  half_t *Chi, *Clo; long c_kst;      // view base (row 0, column 0), halfs between 32-column groups
  const float* row_scale;             // optional [M]: the A planes hold row m divided by row_scale[m] (launch_split_rows);
                                      // the product of row m is multiplied back before bias / addend
  // optional row grouping (width-first kernel only): logical row m lives at physical row (m / grp_rows) * grp_stride +
  // m % grp_rows of A and C -- the time slabs of the state buffers are padded to 16 rows, and at B = 1 fifteen of
  // every sixteen rows are padding that a product over the physical rows would stream W against
  int grp_rows, grp_stride;           // 0: identity
  // optional second destination (width-first kernel only; no planes out, no addend): rows >= c_split are written to C2 + (row - c_split) * ldc2 instead --
  // two row groups of ONE product land in two differently strided buffers (the clip driver's cached projections: ring slot + newest-frame row)
  float* C2; long ldc2; int c_split;  // C2 == nullptr: none
};
struct GateDir {
  const float* gi; long ldgi;         // x W_ih^T + b_ih, [row*ldgi + g*Hp + j]
  const float* bhh;                   // [3Hp]
  const float* hprev; long ldh;       // fp32 previous state (unused by the first step)
  float* hout; long ldo;              // fp32 new state
  half_t *hout_hi, *hout_lo; long okst;   // its planes: view base (row % 16 == 0), halfs between 32-column groups
  long gi_blk = 0;                    // as GruDir::gi_blk
  const float* hprev_b = nullptr; long hp_blk = 0;    // as GruDir::hprev_b / hout_b
  float* hout_b = nullptr; long ho_blk = 0;
};
// fp32 recurrent state of large batches between two cell steps (its only reader is the next step's cell update): element (row, unit) of a view
__host__ __device__ inline long st_blk_offset(long row, int unit, long rt_stride) {
  return (row >> 4) * rt_stride + (long)(unit >> 4) * 256 + ((((unit & 15) >> 2) * 16 + (row & 15)) << 2) + (unit & 3);
}
// Blocked layout of the gate pre-activations of large batches (round 4; internal scratch between the barrier-free projection kernel
// and the fused GRU step / first-step kernels): 16 rows x 16 hidden units of one gate = one 1 KB block, stored in the order of the
// 16x16x32 MFMA's transposed C fragment (lane = (unit % 16 / 4) * 16 + row % 16 holds 4 consecutive units) -- a wave instruction of the
// producer's tile store and of the consumer's cell-update load moves ONE contiguous KB instead of 16 rows x 64 bytes.  Inside a
// 16-row tile the blocks are ordered [unit / 32][unit / 16 & 1][gate]: the six blocks a wave reads per row tile are 6 KB in a row.
// Offset (floats) of element (row, gate, unit) relative to the view's first row tile; rt_stride = 3 Hp * 16 for a [rows][3 Hp] matrix.
__host__ __device__ inline long gi_blk_block(int gate, int unit) { return ((long)(unit >> 5) * 6 + ((unit >> 4) & 1) * 3 + gate) * 256; }
// column `col` of a [rows][ND * 3 Hp] matrix ([dir][gate][unit]): the block's offset inside its row tile (directions are 3 Hp * 16 floats apart)
__host__ __device__ inline long gi_blk_col_block(int col, int hp) {
  const int dgi = col / hp, unit = col - dgi * hp, dir = dgi / 3;
  return (long)dir * 3 * hp * 16 + gi_blk_block(dgi - 3 * dir, unit);
}
__host__ __device__ inline long gi_blk_offset(long row, int gate, int unit, long rt_stride) {
  return (row >> 4) * rt_stride + gi_blk_block(gate, unit) + ((((unit & 15) >> 2) * 16 + (row & 15)) << 2) + (unit & 3);
}
struct H3Batch { H3Args p[3]; GateDir gate[3]; int n; int Hp; };
// up to 3 independent products of the same M, N, Kp in one launch
hipError_t launch_gemm_h3(const H3Batch& b, hipStream_t s, const Options& o);
// recurrent product with the GRU cell update fused into the epilogue: W planes in the gate-interleaved tile
// order (ROW_GATES_TILED), p[d].C unused, gate[d] describes the cell operands / outputs
hipError_t launch_gru_h3(const H3Batch& b, hipStream_t s);
// the same step for small M (skinny_h3.hip): width-first blocks, K split over the waves, operands streamed to VGPRs
hipError_t launch_skinny_gru_h3(const H3Batch& b, hipStream_t s);
hipError_t launch_skinny_gemm_h3(const H3Args& a, hipStream_t s, const Options& o);
struct H3ArgsBatch { H3Args p[3]; int n; };                       // independent products (own M, N, K) in one launch
hipError_t launch_skinny_gemm_h3_batch(const H3ArgsBatch& b, hipStream_t s, const Options& o);
// gru_seq.hip: all T cell steps of one layer (up to 3 directions) in one persistent launch for M <= 64 rows, W_hh
// planes stationary in registers.  Per (direction, step): where the step's gate pre-activations come from and where
// its new state goes (fp32 + the planes view, as GateDir); step t reads the planes step t-1 wrote.
constexpr int kSeqMaxT = 36;             // the table travels in the kernel arguments (4 KB)
struct GruSeqStep {
  const float* gi; float* hout;          // gi[row*ldgi + g*Hp + j], hout[row*ldo + j]
  int ldgi, ldo;
  unsigned poff, pkst;                   // hout's planes: view base offset inside phi / plo (halfs), halfs between 32-column groups
};
struct GruSeqArgs {
  GruSeqStep st[3][kSeqMaxT];
  const half_t* whi[3]; const half_t* wlo[3]; long w_kst;   // W_hh planes (gate-interleaved tile order), as H3Args
  const float* bhh[3];
  half_t *phi, *plo;                     // state planes (hi / lo) that every poff refers to
  unsigned* counters;                    // 3 x 32 uints, zeroed before the launch: arrivals per direction
  unsigned* status;                      // set to non-zero when a bounded wait gave up (device word, cleared per forward)
  unsigned* fault;                       // the handle's host-visible fault word (pinned host memory, sticky until
                                         // tepose_status reads it) or nullptr: written with system scope on give-up
  unsigned spin_limit;                   // polls before a wait gives up (~1 us each)
  unsigned inject;                       // test knob (TEPOSE_TEST_FAULT): added to what a wait expects, so that it cannot be met
  int ndir, T, M, Hp;
  // optional single cell step from h = 0 of one more direction (the top bi-GRU layer's forward direction consumes only
  // its first step, tepose.py:80), done by direction 0's workgroups before their loop; x_gi == nullptr: none
  const float* x_gi; const float* x_bhh; float* x_hout;
  int x_ldgi, x_ldo;
  unsigned x_poff, x_pkst;
  // optional relu(final state) planes = the A operand of the tail linears: view offsets inside rhi / rlo per
  // direction (and for the extra step), kNoPlane = none; r_kst halfs between 32-column groups
  half_t *rhi, *rlo;
  unsigned r_off[3], x_roff, r_kst;
  unsigned long long* stamps;            // diagnostic builds (-DTEPOSE_SEQ_STAMPS): per-step wall-clock stamps of one workgroup
  // M <= 16: the state travels between workgroups as 8-byte {tag, hi|lo} granules -- the data is the flag (one round trip
  // per step instead of store drain + counter + poll + load).  gran: [3 directions][2 buffers][4 rows][Hp] uint64, zeroed
  // before the forward; a step's tag = tag_base + step + 1 (tag_base separates the layers of one forward).  nullptr: counters.
  unsigned long long* gran;
  unsigned tag_base;
};
constexpr size_t kSeqGranRows = 4;       // granule mode serves <= 4 rows (gru_seq_gran_max_m)
constexpr unsigned kNoPlane = 0xffffffffu;
// reg_seq.hip: the regressor's FC loop (fc1 / fc2 / decoders x n_iter) for N <= 64 rows in one persistent launch
struct RegSeqArgs {
  const half_t *fh, *fl; long f_kst;     // feature planes [N x 2048]
  const half_t *w1a_h, *w1a_l, *w1b_h, *w1b_l, *w2_h, *w2_l, *wd_h, *wd_l;   // weight planes ([1024] rows; decoders [256])
  const float *b1, *b2, *bdec;
  const float *init160, *ipose, *ishape, *icam;   // model means [160]; optional per-call rows [N,144] / [N,10] / [N,3]
  half_t *h1h, *h1l, *h2h, *h2l; long h_kst;      // exchange planes [N x 1024]
  half_t *xh, *xl; long x_kst;                    // state planes [N x 160]
  float* xs;                                      // final state [N][160]
  unsigned* counters;                             // 3 x 32 uints, zeroed before the launch
  unsigned* status;
  unsigned* fault; unsigned spin_limit; unsigned inject;      // as GruSeqArgs
  int N, n_iter;
};
hipError_t launch_reg_seq(const RegSeqArgs& a, hipStream_t s);
bool gru_seq_ok(int ndir, int M, int Hp, int T, const Options& o);
int gru_seq_gran_rows(const Options& o);        // min(Options::seq_gran_max_m, the kernel's granule capacity)
hipError_t launch_gru_seq(const GruSeqArgs& a, hipStream_t s, const Options& o);
// first cell step of a direction (h = 0: no product), writing the same outputs
struct GateBatch { GateDir d[3]; };
hipError_t launch_gru_first(const GateBatch& gb, int ndir, int M, int Hp, hipStream_t s, int scaled16 = 0);
// x[rows][2133] fp32 -> blocked hi / lo planes of [rows x 2144]
hipError_t launch_pad_input_planes(const float* x, void* hi, void* lo, long rows, hipStream_t s);
// (relu?)src[rows][ld] fp32 (K valid columns) -> blocked planes of [R x Kp], rows < R
hipError_t launch_split_planes(const float* src, long ld, long rows, int K, int Kp, long R, void* hi, void* lo,
                               hipStream_t s, int relu = 0);
// Caller-supplied rows (the [.,2133] windows) -> planes with one power-of-two scale per row: row m is stored as
// x[m] * 2^e_m with max|x[m]| * 2^e_m in [2^13, 2^14) (zero / non-finite rows: e = 0), row_scale[m] = 2^-e_m.  Any
// finite fp32 input is representable (no fp16 overflow at |x| >= 65504, no precision loss for tiny features); the
// consuming product multiplies row m's result by row_scale[m].  fmt16: the scaled [K/16][R][16] format of gemm_h3s.hip
// (hi = fp16(v), lo = fp16(v - hi)), else the blocked [K/32][R][32] format (lo = fp16((v - hi) * 2^11)).
// zero / zero_bytes (multiple of 16): the kernel also clears that block -- the forward's arrival counters and granules,
// when this is the forward's first kernel (saves the memset node).
// optional gathered source of launch_split_rows (the clip driver's two per-step projections as one product): row r < B = features f0[r] | theta th0[r],
// row r >= B = features f1[r - B] | zeros -- what two launch_pad_rows calls would have written to a padded buffer first (evaluate.py:248-252)
struct RowPairSrc { const float* f0; const float* th0; const float* f1; long fld, thld; long B; };
hipError_t launch_split_rows(const float* src, long ld, long rows, int K, int Kp, long R, int fmt16, void* hi, void* lo,
                             float* row_scale, hipStream_t s, const Options& o, void* zero = nullptr, size_t zero_bytes = 0,
                             int permT = 0, const RowPairSrc* pair = nullptr);   // permT = T: source rows [B][T] -> plane rows / row_scale frame-major (t * B + b)
hipError_t launch_gemm_h3_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C,
                              long ldc, int M, int N, int K, void* ws, hipStream_t s, const Options& o, int kind = 0);
size_t gemm_h3_ws_bytes(int M, int N, int K);
// gemm_h3s.hip: single accumulator, 256 x 256 tiles, scaled planes in the [K/16][R][16] layout:
// element (row, k) of an [R x Kp] matrix (Kp multiple of 16), value v stored as hi = fp16(v p), lo = fp16(v p - hi)
__host__ __device__ inline long plane16_index(long row, long k, long R) {
  return ((k >> 4) * R + row) * 16 + ((((k >> 3) & 1) ^ ((row >> 3) & 1)) << 3) + (k & 7);
}
constexpr float kStateScale = 16384.f;   // scale of recurrent-state planes in that format (|h| < 1)
struct H3SArgs {
  const half_t *Ah, *Al; long a_kst;     // halfs between K-tiles (R * 16)
  const half_t *Wh, *Wl; long w_kst;     // W rows padded to 256
  int Kp;                                // multiple of 16
  float* C; long ldc;
  const float* bias;                     // [N] or nullptr
  float inv_scale;                       // 1 / (pA * pW)
  int M, N;
  const float* row_scale;                // optional [M], as H3Args::row_scale (then pA = 1)
  // barrier-free kernels (gemm_h3s16c.hip): where a wave whose bounded LDS poll expired reports it -- the forward's status word
  // (workspace sync region, agent scope) and the handle's host-visible fault word (system scope); nullptr: debug counter + NaN only
  unsigned* status = nullptr;
  unsigned* fault = nullptr;
  unsigned inject = 0;                   // tests: added to every poll target (1 = no poll can ever be met)
  int c_blk_hp = 0;                      // != 0 (= Hp): C is a [rows][3 Hp] gate pre-activation matrix, written in the blocked layout (gi_blk_offset)
};
struct H3SBatch { H3SArgs p[3]; GateDir gate[3]; int n; int Hp; float state_scale; };   // state_scale: scale of the
hipError_t launch_gemm_h3s(const H3SArgs& a, hipStream_t s, const Options& o, int tag = 1);   // every plain scaled-plane product of large batches (tag 0: the layer-0 projection: own kernel symbol for profiles)
bool gemm_h3s16_ok(const H3SArgs& a);
hipError_t launch_gemm_h3s16c(const H3SArgs& a, hipStream_t s, int tag, int gm);  // gemm_h3s16c.hip: persistent, no workgroup barriers in the K loop
unsigned h3s16c_read_err();
void h3s16c_warm();                    // allocates the current device's debug error counter (tepose_set_blob: never inside a stream capture)
// gru_step16.hip: the fused GRU cell step of large batches; planes = the instantiation that takes its cell operands through the LDS-DMA stream and
// rebuilds h_{t-1} from the state planes (full tiles + blocked layouts only: gru_step16_planes_ok)
bool gru_step16_ok(const H3SBatch& b);
bool gru_step16_planes_ok(const H3SBatch& b);
hipError_t launch_gru_step16(const H3SBatch& b, hipStream_t s, bool planes, int gm);
unsigned* h3s16c_err_of_device();      // the current device's debug counter of give-ups of the barrier-free kernels (nullptr: none)
bool gemm_h3s_mid_ok(const H3SArgs& a);
hipError_t launch_gemm_h3s_mid(const H3SArgs& a, hipStream_t s);   // 128 x 288 tiles, N % 288 == 0
hipError_t launch_split_planes16(const float* src, long ld, long rows, int K, int Kp, long R, float p, void* hi,
                                 void* lo, hipStream_t s);
hipError_t launch_absmax(const float* src, size_t n, float* out, hipStream_t s);
size_t gemm_h3s_ws_bytes(int M, int N, int K);
hipError_t launch_gemm_h3s_f32(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N,
                               int K, float pA, float pW, void* ws, hipStream_t s, const Options& o, const float* bias = nullptr,
                               int mid = 0);   // mid: the 128 x 288-tile kernel (N % 288 == 0) instead of the barrier-free persistent one

}  // namespace tepose
