// C ABI of libtepose_hip.so (include/tepose_amd.h): model handle, weight packing, and the
// launch sequence of the TePose forward.  Host code only; every kernel lives in gemm.hip,
// misc.hip, smpl.hip.  Nothing here allocates device memory or synchronises the device.
#include "../../include/tepose_amd.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

using namespace tepose;

namespace {

struct DirW {                     // one GRU layer/direction inside the blob (float offsets)
  size_t wih = 0, bih = 0;        // input projection (layer-0 ones live in the stacked block)
  size_t whh = 0, bhh = 0;
  size_t wih_p = 0, whh_p = 0;    // blocked hi|lo fp16 planes of the same matrices (whh: gate-tiled rows), float offsets
  size_t wih_s = 0, whh_s = 0;    // the same as scaled [K/16][R][16] planes (gemm_h3s.hip; rows padded to 256 / 384)
  size_t scales = 0;              // blob slot: [0] = W_ih scale, [1] = W_hh scale
  float wih_scale = 1.f, whh_scale = 1.f;   // host copies
};

struct SmplOff {
  size_t J0, JS, blendW, lbsW, lbs_cidx, lbs_cval, lbs_nnz, parents, depth, xr_ptr, xr_idx, xr_val;
};

}  // namespace

struct tepose_model {
  int kind = 0;                                 // 0 = TePose, 1 = VIBE bootstrap encoder
  std::vector<DirW> vibe;                       // VIBE: per-layer GRU weights; wih / bih hold the stacked rows of both
                                                // directions ([dir][3Hp]), whh / bhh of direction d sit at + d * their size
  bool vibe_bidir = false, vibe_linear = true;  // vibe.py:27-47: bidirectional GRU; Linear(D*hidden -> 2048) on relu(y)
  size_t vlin_w = 0, vlin_b = 0;
  bool vibe_packed = false;
  int L = 0, H = 0, Hp = 0;
  size_t hdr = 0;                               // blob header (BlobHeader): what the blob holds, checked by tepose_adopt_blob
  float* blob = nullptr;
  size_t blob_floats = 0;
  bool enc_packed = false, reg_packed = false, smpl_packed = false;
  // encoder offsets
  size_t wih0 = 0, bih0 = 0;                    // stacked [9Hp][2144]: fwd | rec_reverse | rec
  size_t wih0_p = 0;                            // its hi|lo planes
  size_t wih0_s = 0, wih0_scale = 0;            // the same block as scaled [K/16][R][16] planes (gemm_h3s.hip) + its scale
  float w0_scale = 1.f;                         // host copy of blob[wih0_scale]
  // kernel-family knobs (named options, read once per handle; defaults = the measured best):
  bool large_scaled = true;                     // TEPOSE_LARGE_BATCH_KERNELS=scaled|twoacc: large batches (layer-0 projection from B * T >= 8192 / mid tiles from 512 rows,
                                                // recurrent path from s_min_b windows) on the scaled-plane single-accumulator kernels (gemm_h3s16c.hip, gru_step16.hip,
                                                // gemm_h3s.hip) -- or, `twoacc`, on the two-accumulator family of gemm_h3.hip at every batch size
  bool state_planes = true;                     // TEPOSE_GRU_STATE=planes|fp32: the large-batch step kernel rebuilds h_{t-1} from the state planes and takes its cell
                                                // operands through the LDS-DMA stream (gru_step16_kernel<true>) -- or, `fp32`, keeps a separate fp32 state copy (<false>)
  std::string kinfo;                            // tepose_kernel_info(): the kernel symbols the knobs select for the dominant launches of cfg-C
  std::vector<DirW> fwd, rec_f, rec_r;          // per layer
  size_t wlf = 0, blf = 0, wlr = 0, blr = 0;
  size_t wlf_p = 0, wlr_p = 0;                  // blocked hi|lo planes of the tail linears
  size_t wlfr_p = 0;                            // planes of [W_lf | W_lr] ([2048][3Hp]): eval mode's (y_fwd + y_rec)/2 as ONE product
  // regressor offsets
  size_t w1a = 0, b1 = 0, w1b = 0, w2 = 0, b2 = 0, wdec = 0, bdec = 0, init = 0;
  size_t w1a_p = 0, w1b_p = 0, w2_p = 0, wdec_p = 0, blendW_p = 0;   // blocked hi|lo planes (split path)
  size_t blendW_s = 0, blend_scale = 0;         // the blend-shape matrix as scaled [K/16][R][16] planes (large batches: barrier-free persistent kernel) + its scale
  float blend_sc = 1.f;                         // host copy of blob[blend_scale]
  // collapsed regressor: the eval-mode FC loop is affine in (feature, initial state), so with the model's own
  // initial state and n_iter = 3 the final state is  xs = feat Mf^T + k0  and, through the (affine) tail linears,
  // xs = [relu(h_fwd) | relu(y_rec0)] Mt^T + kt.  fp64 algebra at pack time; [256][K] fp32 + planes, bias rows of 160.
  size_t mf = 0, mf_p = 0, k0 = 0, mt = 0, mt_p = 0, kt = 0;
  bool reg_collapsed = false, tail_collapsed = false;
  bool collapse_env = true;                     // TEPOSE_COLLAPSE_REGRESSOR=0: always run the FC loop
  SmplOff smpl{};
  int maxdepth = 0;
  int lbs_sparse = 0;                           // skin-weight table has <= 4 non-zeros per vertex
  bool split = true;                            // batches of more than m->opt.split_min_m rows run their matmuls on the fp16x3 split kernels
  bool split_env = true;                        // what the environment asked for; `split` also needs every packed weight inside
  bool enc_range_ok = true, reg_range_ok = true, smpl_range_ok = true;   // the fp16 range (|w| < 2^15), checked at pack time
  int s_min_b = 640;                            // scaled-format recurrent path from this batch size
  Options opt;                                  // every launch threshold (common.h Options): from the environment at tepose_create, tepose_set_option before packing
  // fault channel of the persistent kernels (gru_seq.hip, reg_seq.hip): one word of pinned host memory that a kernel
  // whose bounded wait expired writes with system scope; sticky until tepose_status() reads it
  unsigned* fault = nullptr;
  bool persist = true;                          // false: step-per-launch kernels at every batch size (tepose_set_persistent)
  unsigned spin_limit = 1u << 21;               // polls (~1 us each) before a wait gives up
  int blend16_min_n = 512;                      // TEPOSE_BLEND16_MIN_N: rows from which the blend-shape product runs on gemm_h3s_persist16c_kernel (0x7fffffff = never)
  int gi_blk = 1;                               // TEPOSE_GI_BLK: large batches keep the layer >= 1 gate pre-activations in the 16 x 16-blocked layout (common.h gi_blk_offset)
  int last_fault_code = 0;                      // the kernel code of the last fault a status call collected (tepose_fault_code)
  // The fault word is shared by every stream and thread of the handle and ANY status call clears it, so "the word is clear" says nothing about one
  // particular forward once somebody else has collected: `collected` counts the clears that found the word raised, and every forward notes the count
  // it was queued under, per workspace (its status words live there).  tepose_forward_status trusts the clear word only while the count stands still.
  mutable std::mutex q_mu;
  mutable std::unordered_map<const void*, unsigned> q_gen;
  mutable unsigned collected = 0;               // guarded by q_mu
  unsigned test_fault = 0;                      // TEPOSE_TEST_FAULT: bit 0 recurrent kernel, bit 1 regressor kernel wait for arrivals that never come
  // profiling of the dominant kernel (layer-0 input-projection GEMM)
  bool prof = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  double prof_flops = 0.0;
  std::vector<hipEvent_t> ev_gru;               // pairs around each layer's sequence of GRU-step launches
  size_t ev_gru_used = 0;
  double prof_gru_flops = 0.0;                  // algorithmic FLOPs of all GRU steps of one forward
  double prof_l1_flops = 0.0;                   // algorithmic FLOPs of the layer >= 1 input projections (launched between two GRU intervals)
};

namespace {

constexpr size_t kAlignF = 64;   // 256-byte sections

size_t take(size_t& cur, size_t n) {
  const size_t o = cur;
  cur = align_up(cur + n, kAlignF);
  return o;
}

void layout_tail(tepose_model* m, size_t cur);

void layout_vibe(tepose_model* m) {
  const size_t Hp = m->Hp, L = m->L, D = m->vibe_bidir ? 2 : 1;
  size_t cur = 0;
  m->hdr = take(cur, 64);
  const size_t n128 = round_up(3 * (int)(D * Hp), 128);
  m->vibe.assign(L, DirW());
  for (size_t l = 0; l < L; ++l) {
    m->vibe[l].wih = take(cur, n128 * (l == 0 ? (size_t)kFeat : D * Hp));
    m->vibe[l].bih = take(cur, D * 3 * Hp);
    m->vibe[l].whh = take(cur, D * 3 * Hp * Hp);
    m->vibe[l].bhh = take(cur, D * 3 * Hp);
  }
  if (m->vibe_linear) {
    m->vlin_w = take(cur, (size_t)kFeat * D * Hp);
    m->vlin_b = take(cur, kFeat);
  }
  layout_tail(m, cur);
}

void layout(tepose_model* m) {
  const size_t Hp = m->Hp, L = m->L;
  size_t cur = 0;
  m->hdr = take(cur, 64);
  m->wih0 = take(cur, (size_t)round_up(9 * (int)Hp, 128) * kInputP);
  m->bih0 = take(cur, 9 * Hp);
  m->fwd.assign(L, DirW());
  m->rec_f.assign(L, DirW());
  m->rec_r.assign(L, DirW());
  for (size_t l = 0; l < L; ++l) {
    const size_t n128 = round_up(3 * (int)Hp, 128);
    if (l > 0) {
      m->fwd[l].wih = take(cur, n128 * Hp);
      m->fwd[l].bih = take(cur, 3 * Hp);
      m->rec_f[l].wih = take(cur, n128 * 2 * Hp);
      m->rec_f[l].bih = take(cur, 3 * Hp);
      m->rec_r[l].wih = take(cur, n128 * 2 * Hp);
      m->rec_r[l].bih = take(cur, 3 * Hp);
    }
    for (DirW* d : {&m->fwd[l], &m->rec_f[l], &m->rec_r[l]}) {
      d->whh = take(cur, 3 * Hp * Hp);
      d->bhh = take(cur, 3 * Hp);
    }
  }
  m->wlf = take(cur, (size_t)kFeat * Hp);
  m->blf = take(cur, kFeat);
  m->wlr = take(cur, (size_t)kFeat * 2 * Hp);
  m->blr = take(cur, kFeat);
  // split-precision copies (hi plane then lo plane, fp16): same float count as the fp32 matrix
  m->wih0_p = take(cur, (size_t)round_up(9 * (int)Hp, 128) * kInputP);
  m->wih0_s = take(cur, (size_t)round_up(9 * (int)Hp, 256) * kInputP);
  m->wih0_scale = take(cur, 16);
  for (size_t l = 0; l < L; ++l) {
    const size_t n128 = round_up(3 * (int)Hp, 128);
    if (l > 0) {
      m->fwd[l].wih_p = take(cur, n128 * Hp);
      m->rec_f[l].wih_p = take(cur, n128 * 2 * Hp);
      m->rec_r[l].wih_p = take(cur, n128 * 2 * Hp);
    }
    for (DirW* d : {&m->fwd[l], &m->rec_f[l], &m->rec_r[l]}) d->whh_p = take(cur, n128 * Hp);
    const size_t r256 = round_up(3 * (int)Hp, 256), r384 = round_up(3 * (int)Hp, 384);
    if (l > 0) {
      m->fwd[l].wih_s = take(cur, r256 * Hp);
      m->rec_f[l].wih_s = take(cur, r256 * 2 * Hp);
      m->rec_r[l].wih_s = take(cur, r256 * 2 * Hp);
    }
    for (DirW* d : {&m->fwd[l], &m->rec_f[l], &m->rec_r[l]}) {
      d->whh_s = take(cur, r384 * Hp);
      d->scales = take(cur, 16);
    }
  }
  m->wlf_p = take(cur, (size_t)kFeat * Hp);
  m->wlr_p = take(cur, (size_t)kFeat * 2 * Hp);
  m->wlfr_p = take(cur, (size_t)kFeat * 3 * Hp);
  layout_tail(m, cur);
}

void layout_tail(tepose_model* m, size_t cur) {   // regressor + SMPL sections, shared by both kinds
  m->w1a = take(cur, 1024 * (size_t)kFeat);
  m->b1 = take(cur, 1024);
  m->w1b = take(cur, 1024 * (size_t)kState);
  m->w2 = take(cur, 1024 * 1024);
  m->b2 = take(cur, 1024);
  m->wdec = take(cur, 256 * 1024);
  m->bdec = take(cur, kState);
  m->init = take(cur, kState);
  m->smpl.J0 = take(cur, 72);
  m->smpl.JS = take(cur, 720);
  m->smpl.blendW = take(cur, (size_t)kBlendN * kBlendK);
  m->smpl.lbsW = take(cur, (size_t)kNV * kNJ);
  m->smpl.lbs_cidx = take(cur, (size_t)kNV * 4);
  m->smpl.lbs_cval = take(cur, (size_t)kNV * 4);
  m->smpl.lbs_nnz = take(cur, 16);
  m->smpl.parents = take(cur, 32);
  m->smpl.depth = take(cur, 32);
  m->smpl.xr_ptr = take(cur, 16);
  m->smpl.xr_idx = take(cur, (size_t)9 * kNV);
  m->smpl.xr_val = take(cur, (size_t)9 * kNV);
  m->w1a_p = take(cur, 1024 * (size_t)kFeat);
  m->w1b_p = take(cur, 1024 * (size_t)kState);
  m->w2_p = take(cur, 1024 * 1024);
  m->wdec_p = take(cur, 256 * 1024);
  m->blendW_p = take(cur, (size_t)kBlendN * kBlendK);
  m->blendW_s = take(cur, (size_t)kBlendN * kBlendK);
  m->blend_scale = take(cur, 16);
  m->mf = take(cur, 256 * (size_t)kFeat);
  m->mf_p = take(cur, 256 * (size_t)kFeat);
  m->k0 = take(cur, kState);
  if (m->kind == 0) {
    m->mt = take(cur, 256 * (size_t)3 * m->Hp);
    m->mt_p = take(cur, 256 * (size_t)3 * m->Hp);
    m->kt = take(cur, kState);
  }
  m->blob_floats = cur;
}

#define CK(expr)                      \
  do {                                \
    hipError_t e__ = (expr);          \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)

// First 256 bytes of the blob: identifies the model the packed sections belong to, so that a blob that travelled
// (RCCL broadcast, copy) is only adopted by a handle of the same kind / size / library layout.
struct BlobHeader {
  uint32_t magic, abi, kind, L, H, Hp, sections;   // sections: bit 0 encoder, 1 regressor, 2 SMPL tables, 3 range flag, 4 / 5 collapsed regressor / tail
  uint32_t layout_floats_lo, layout_floats_hi;      // blob_floats of the layout that wrote it
};
constexpr uint32_t kBlobMagic = 0x54455031u;        // "TEP1"

// model kind as the header records it: a VIBE handle also carries its constructor flags
uint32_t header_kind(const tepose_model* m) {
  return (uint32_t)m->kind | (m->kind == 1 ? (m->vibe_bidir ? 0x100u : 0u) | (m->vibe_linear ? 0x200u : 0u) : 0u);
}

int write_header(tepose_model* m, hipStream_t s) {
  BlobHeader h{};
  h.magic = kBlobMagic; h.abi = TEPOSE_ABI_VERSION; h.kind = header_kind(m); h.L = (uint32_t)m->L; h.H = (uint32_t)m->H;
  h.Hp = (uint32_t)m->Hp;
  h.sections = ((m->kind == 0 ? m->enc_packed : m->vibe_packed) ? 1u : 0u) | (m->reg_packed ? 2u : 0u) |
               (m->smpl_packed ? 4u : 0u) | ((m->enc_range_ok && m->reg_range_ok && m->smpl_range_ok) ? 0u : 8u) |
               (m->reg_collapsed ? 16u : 0u) | (m->tail_collapsed ? 32u : 0u);
  h.layout_floats_lo = (uint32_t)(m->blob_floats & 0xffffffffu); h.layout_floats_hi = (uint32_t)((uint64_t)m->blob_floats >> 32);
  CK(hipMemcpyAsync(m->blob + m->hdr, &h, sizeof(h), hipMemcpyHostToDevice, s));
  CK(hipStreamSynchronize(s));                      // h is a stack object (pack time only)
  return 0;
}

// Pack-time range guard of the split-precision path: a weight of magnitude >= 2^15 (or inf) has no fp16 hi half, so a
// handle holding one runs every product on the exact-fp32 kernels instead (as TEPOSE_EXACT_FP32=1) -- never a silent inf.
// One device reduction + read-back over the section's fp32 copy; pack time only.
int range_check(tepose_model* m, size_t first, size_t end, bool* ok, hipStream_t s) {
  float* scratch = m->blob + m->hdr + 32;
  CK(launch_absmax(m->blob + first, end - first, scratch, s));
  float wmax = 0.f;
  CK(hipMemcpyAsync(&wmax, scratch, sizeof(float), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  *ok = wmax < 32768.f;
  m->split = m->split_env && m->enc_range_ok && m->reg_range_ok && m->smpl_range_ok;
  return 0;
}

int pack(const float* src, long ld, int N, int K, float* dst, int Np, int Kp, int rowmap, int colmap,
         int H, int Hp, hipStream_t s) {
  PackArgs a{src, ld, N, K, dst, Np, Kp, nullptr, nullptr, 0, rowmap, colmap, H, Hp};
  return (int)launch_pack(a, s);
}

// same mapping, written as K-tile-blocked hi / lo fp16 planes of [Np x Kp] (lo plane follows the hi plane)
int pack_planes(const float* src, long ld, int N, int K, float* dst_planes, int Np, int Kp, int rowmap, int colmap,
                int H, int Hp, hipStream_t s) {
  half_t* hi = (half_t*)dst_planes;
  PackArgs a{src, ld, N, K, nullptr, Np, Kp, hi, hi + (size_t)Np * Kp, (long)Np * 32, rowmap, colmap, H, Hp};
  return (int)launch_pack(a, s);
}

// blocked hi / lo planes of an already packed fp32 blob matrix [Np][Kp] (lo plane follows the hi plane)
int planes_of(const float* packed, int Np, int Kp, float* dst_planes, hipStream_t s) {
  half_t* hi = (half_t*)dst_planes;
  return (int)launch_split_planes(packed, Kp, Np, Kp, Kp, Np, hi, hi + (size_t)Np * Kp, s);
}

// scaled [K/16][R][16] planes (gemm_h3s.hip) of an already packed fp32 blob matrix [rows][Kp], R >= rows rows
// allocated (zero beyond); one power-of-two scale for the matrix (largest |w| * p in [2^13, 2^14)), written to
// *scale_dev and *scale_host.  Pack time only: one device reduction + read-back.
int scaled_planes_of(const float* packed, int rows, int Kp, float* dst_planes, int R, float* scale_dev, float* scale_host,
                     hipStream_t s) {
  hipError_t e = launch_absmax(packed, (size_t)rows * Kp, scale_dev, s);
  if (e != hipSuccess) return (int)e;
  float wmax = 0.f;
  if ((e = hipMemcpyAsync(&wmax, scale_dev, sizeof(float), hipMemcpyDeviceToHost, s)) != hipSuccess) return (int)e;
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return (int)e;
  float p = 1.f;
  if (wmax > 0.f && wmax < 3e38f) {
    int ex = 0;
    (void)frexpf(wmax, &ex);              // wmax = f * 2^ex, f in [0.5, 1)
    p = ldexpf(1.f, 14 - ex);             // wmax * p in [2^13, 2^14)
  }
  *scale_host = p;
  if ((e = hipMemcpyAsync(scale_dev, scale_host, sizeof(float), hipMemcpyHostToDevice, s)) != hipSuccess) return (int)e;
  if ((e = launch_fill(dst_planes, (size_t)R * Kp, 0.f, s)) != hipSuccess) return (int)e;
  half_t* hi = (half_t*)dst_planes;
  if ((e = launch_split_planes16(packed, Kp, rows, Kp, Kp, (long)R, p, hi, hi + (size_t)R * Kp, s)) != hipSuccess)
    return (int)e;
  return (int)hipStreamSynchronize(s);    // *scale_host is read by the async copy above
}

struct Carver {
  char* base; size_t cur = 0, cap;
  Carver(void* p, size_t c) : base((char*)p), cap(c) {}
  float* f(size_t n) {
    const size_t o = cur;
    cur = align_up(cur + n * sizeof(float), 256);
    return base ? (float*)(base + o) : nullptr;
  }
};

struct Planes { half_t *hi = nullptr, *lo = nullptr; long kst = 0; };   // blocked planes of an [R x C] matrix
Planes carve_planes(Carver& c, size_t R, size_t C, bool on) {
  Planes p;
  p.hi = (half_t*)c.f(on ? R * C / 2 + 64 : 0);
  p.lo = (half_t*)c.f(on ? R * C / 2 + 64 : 0);
  p.kst = (long)R * 32;
  return p;
}

// Buffers of one encoder forward (shared between sizing and execution).
struct EncWs {
  float *xp, *g0, *g0c, *gf, *grr, *grf, *sf[2], *sr[2], *pf[2], *pr[2], *ytop, *y1;
  // split-precision path: every state buffer [T][B][C] has fp16 hi / lo mirror planes holding the
  // [T * Bs x C] matrix in the K-tile-blocked layout of common.h (slabs of Bs = B rounded up to 16 rows, in the
  // fp32 buffers too, so that every time slab starts on a swizzle period; the pad rows are never consumed); x0h / x0l: compact planes of the frames a 1-layer model's rec.l0
  // forward direction consumes
  half_t *state_hi, *state_lo, *x0h, *x0l;
  float *rs = nullptr, *rs0 = nullptr;   // per-row scales of the input planes (launch_split_rows): [B*T] and, 1-layer models, [B]
  unsigned long long* gran = nullptr;    // {tag, hi|lo} granule buffers of the persistent kernel's B <= 16 mode
  unsigned* sync = nullptr;   // persistent recurrent kernel (gru_seq.hip): per layer 3 x 32 arrival counters, then a status word
  Planes tailA, tailF, tailR;   // [relu(last forward state) | relu(ytop)] = [B x 3Hp], A operand of the tail linears; tailF /
                                // tailR: its K-tile ranges [0, Hp/32) and [Hp/32, 3Hp/32)
  size_t Bs = 0;       // rows per time slab of gf/grr/grf/sf/sr: B, or B rounded up to 16 on the split path
  struct Buf { const float* base; size_t T, B, C, poff; };   // poff: first half of its mirror inside state_hi/lo
  Buf bufs[9]; int nbufs = 0;
  size_t plane_halfs = 0;
  void add(const float* base, size_t T, size_t B, size_t C) {     // B = slab rows (multiple of 16 when split)
    bufs[nbufs++] = Buf{base, T, B, C, plane_halfs};
    plane_halfs += T * B * C;
  }
  struct View { half_t *hi, *lo; long kst; };
  // planes of the sub-matrix that starts at fp32 element p = (slab t, row 0, column c0)
  View view(const float* p) const {
    for (int i = 0; i < nbufs; ++i) {
      const Buf& b = bufs[i];
      if (p >= b.base && p < b.base + b.T * b.B * b.C) {
        const size_t off = (size_t)(p - b.base), t = off / (b.B * b.C), rem = off % (b.B * b.C);
        if (rem / b.C != 0 || (rem % b.C) % 32 != 0) break;
        const size_t R = b.T * b.B;
        const size_t e = b.poff + (size_t)plane_index((long)(t * b.B), (long)(rem % b.C), (long)R);
        return View{state_hi + e, state_lo + e, (long)R * 32};
      }
    }
    return View{nullptr, nullptr, 0};
  }
  // the same sub-matrix in the scaled [K/16][R][16] format of gemm_h3s.hip (the mirrors hold ONE of the two formats
  // per forward: every kernel of a forward agrees on it)
  View view16(const float* p) const {
    for (int i = 0; i < nbufs; ++i) {
      const Buf& b = bufs[i];
      if (p >= b.base && p < b.base + b.T * b.B * b.C) {
        const size_t off = (size_t)(p - b.base), t = off / (b.B * b.C), rem = off % (b.B * b.C);
        if (rem / b.C != 0 || (rem % b.C) % 16 != 0) break;
        const size_t R = b.T * b.B;
        const size_t e = b.poff + (size_t)plane16_index((long)(t * b.B), (long)(rem % b.C), (long)R);
        return View{state_hi + e, state_lo + e, (long)R * 16};
      }
    }
    return View{nullptr, nullptr, 0};
  }
};

// Arrival counters of the persistent kernels (gru_seq.hip, reg_seq.hip), ONE block zeroed by one memset node per
// forward.  It is the first carve of the encoder's and of the regressor's workspace, so that inside tepose_forward
// (both share one region) it is the same memory: [L x 3 x 32 recurrent arrivals | 32 status | 3 x 32 regressor | 32 status].
inline size_t sync_words(const tepose_model* m) { return (size_t)m->L * 96 + 32 + 96 + 32; }
// floats of the granule buffers: [3 directions][2 buffers][16 rows][Hp] uint64, only where the persistent kernel can run
inline size_t seq_gran_words(const tepose_model* m, int B) {
  return (m->split && B <= gru_seq_gran_rows(m->opt) && gru_seq_shape_ok(m->Hp)) ? (size_t)3 * 2 * kSeqGranRows * m->Hp * 2 : 0;
}
inline size_t sync_zero_bytes(const tepose_model* m, int B) {      // counters + granules: the block a forward clears
  return align_up(sync_words(m) * sizeof(unsigned), 256) + seq_gran_words(m, B) * sizeof(float);
}
inline unsigned* sync_gru(unsigned* sy, int l) { return sy + (size_t)l * 96; }
inline unsigned* sync_gru_status(const tepose_model* m, unsigned* sy) { return sy + (size_t)m->L * 96; }
inline unsigned* sync_reg(const tepose_model* m, unsigned* sy) { return sy + (size_t)m->L * 96 + 32; }
inline unsigned* sync_reg_status(const tepose_model* m, unsigned* sy) { return sy + (size_t)m->L * 96 + 32 + 96; }

void carve_encoder(const tepose_model* m, int B, int T, Carver& c, EncWs& w) {
  const size_t Hp = m->Hp, BT = (size_t)B * T;
  const int L = m->L;
  const bool h3 = m->split && B > m->opt.split_min_m;
  const size_t Bs = h3 ? (size_t)round_up(B, 16) : (size_t)B, BTs = Bs * T;
  w.Bs = Bs;
  w.sync = (unsigned*)c.f(sync_words(m));
  // granule buffers of the persistent recurrent kernel (B <= 16), right behind the counters: one memset zeroes both
  w.gran = (unsigned long long*)c.f(seq_gran_words(m, B));
  w.xp = c.f(BT * kInputP);
  w.g0 = c.f(BT * (L >= 2 ? 9 : 6) * Hp);
  w.g0c = c.f(L >= 2 ? 0 : (size_t)B * 3 * Hp);
  w.gf = c.f(L >= 2 ? BTs * 3 * Hp : 0);
  w.grr = c.f(L >= 2 ? BTs * 3 * Hp : 0);
  w.grf = c.f(L >= 3 ? BTs * 3 * Hp : (L == 2 ? (size_t)B * 3 * Hp : 0));
  for (int i = 0; i < 2; ++i) {
    const bool need = (i == 0 && L >= 2) || (i == 1 && L >= 3);
    w.sf[i] = c.f(need ? BTs * Hp : 0);
    w.sr[i] = c.f(need ? BTs * 2 * Hp : 0);
    w.pf[i] = c.f(Bs * Hp);
    w.pr[i] = c.f(Bs * Hp);
    if (need) {
      w.add(w.sf[i], T, Bs, Hp);
      w.add(w.sr[i], T, Bs, 2 * Hp);
    }
    w.add(w.pf[i], 1, Bs, Hp);
    w.add(w.pr[i], 1, Bs, Hp);
  }
  w.ytop = c.f(Bs * 2 * Hp);
  w.add(w.ytop, 1, Bs, 2 * Hp);
  w.y1 = c.f((size_t)B * kFeat);
  w.state_hi = (half_t*)c.f(h3 ? w.plane_halfs / 2 + 64 : 0);
  w.state_lo = (half_t*)c.f(h3 ? w.plane_halfs / 2 + 64 : 0);
  w.x0h = (half_t*)c.f(h3 && L == 1 ? (size_t)B * kInputP / 2 + 64 : 0);
  w.x0l = (half_t*)c.f(h3 && L == 1 ? (size_t)B * kInputP / 2 + 64 : 0);
  w.rs = c.f(h3 ? BT : 0);
  w.rs0 = c.f(h3 && L == 1 ? (size_t)B : 0);
  w.tailA = carve_planes(c, B, 3 * Hp, h3);
  w.tailF = w.tailA;
  w.tailR = w.tailA;
  if (h3 && w.tailA.hi) {
    w.tailR.hi = w.tailA.hi + (size_t)(Hp / 32) * w.tailA.kst;
    w.tailR.lo = w.tailA.lo + (size_t)(Hp / 32) * w.tailA.kst;
  }
}

struct RegWs {
  unsigned* sync;                  // see sync_words()
  float *base, *h1, *h2, *xs, *pf, *amat, *posed, *vposed;
  bool split, split_fc;            // split-mode handle with N > 4 rows: blend-shape GEMM / FC stack on the fp16x3 kernels
  Planes featP, xsP, h1P, h2P, pfP;
  // large batches: the pose features again as scaled [K/16][N][16] planes + per-row scales, for the blend-shape product on the barrier-free kernel
  bool blend16 = false; half_t *pf16h = nullptr, *pf16l = nullptr; float* pfrs = nullptr;
};

// regressor / SMPL side of the kernel selection (select_kernels below computes the same two predicates for its description)
inline bool reg_split_for(const tepose_model* m, int N) { return m->split && N > m->opt.split_min_m; }
inline bool blend16_for(const tepose_model* m, int N) { return reg_split_for(m, N) && m->large_scaled && N >= m->blend16_min_n; }

void carve_regressor(const tepose_model* m, int N, Carver& c, RegWs& w) {
  w.split = reg_split_for(m, N);
  w.split_fc = w.split;            // h3_mm picks the width-first kernel for <= 768 rows, 256-row tiles above
  w.sync = (unsigned*)c.f(sync_words(m));
  w.featP = carve_planes(c, N, kFeat, w.split_fc);
  w.xsP = carve_planes(c, N, kState, w.split_fc);
  w.h1P = carve_planes(c, N, 1024, w.split_fc);
  w.h2P = carve_planes(c, N, 1024, w.split_fc);
  w.pfP = carve_planes(c, N, kBlendK, w.split);
  w.base = c.f((size_t)N * 1024);
  w.h1 = c.f((size_t)N * 1024);
  w.h2 = c.f((size_t)N * 1024);
  w.xs = c.f((size_t)N * kState);
  w.pf = c.f((size_t)N * kBlendK);
  w.amat = c.f((size_t)N * kNJ * 12);
  w.posed = c.f((size_t)N * kNJ * 3);
  w.vposed = c.f((size_t)N * kVertLd);
  w.blend16 = blend16_for(m, N);
  if (w.blend16) {
    w.pf16h = (half_t*)c.f((size_t)N * kBlendK / 2);
    w.pf16l = (half_t*)c.f((size_t)N * kBlendK / 2);
    w.pfrs = c.f((size_t)N);
  }
}

hipError_t init_state(const float* init160, const float* pose, const float* shape, const float* cam, float* xs, int N,
                      hipStream_t s) {
  if (pose || shape || cam) return launch_init_state_rows(init160, pose, shape, cam, xs, N, s);
  return launch_init_state(init160, xs, N, s);
}

GemmArgs gemm(const float* A, long lda, const float* W, int Kp, float* C, long ldc, const float* bias,
              int M, int N) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.Kp = Kp; g.C = C; g.ldc = ldc; g.bias = bias;
  g.addend = nullptr; g.ldadd = 0; g.scale = 1.f; g.M = M; g.N = N; g.relu_a = 0;
  return g;
}

// C = (A W^T + bias + addend) * scale on the split-precision kernel: A as blocked planes, W = blocked planes of a
// packed [Np][Kp] blob matrix (hi plane, then lo plane); `out`: also write C as planes (the next product's A)
int h3_mm(const tepose_model* m, const Planes& A, const float* w_planes, int Np, int Kp, float* C, long ldc, const float* bias, int M, int N,
          const float* addend, long ldadd, float scale, const Planes* out, hipStream_t s, const float* row_scale = nullptr) {
  H3Batch b{};
  const half_t* wh = (const half_t*)w_planes;
  H3Args& p = b.p[0];
  p.Ah = A.hi; p.Al = A.lo; p.a_kst = A.kst;
  p.Wh = wh; p.Wl = wh + (size_t)Np * Kp; p.w_kst = (long)Np * 32; p.Kp = Kp;
  p.C = C; p.ldc = ldc; p.bias = bias; p.M = M; p.N = N;
  p.addend = addend; p.ldadd = ldadd; p.scale = scale; p.row_scale = row_scale;
  if (out) { p.Chi = out->hi; p.Clo = out->lo; p.c_kst = out->kst; }
  // few rows: width-first kernel (skinny_h3.hip), except the short-K / very wide blend-shape product, which already
  // makes 162 tiles of the big kernel
  // (and any product with <= 256 columns -- the stacked decoders -- at every M: 2 column tiles of the big kernel
  // would use 64 CUs)
  if ((M <= m->opt.skinny_max_m && !(N > 4096 && Kp < 512)) || N <= 256) return (int)launch_skinny_gemm_h3(p, s, m->opt);
  b.n = 1;
  return (int)launch_gemm_h3(b, s, m->opt);
}

// v_posed = v_template + shapedirs beta + posedirs^T pose_feature as one GEMM, K = 224
int blend_shapes(const tepose_model* m, const RegWs& w, int N, hipStream_t s) {
  const float* Bl = m->blob;
  if (w.blend16) {
    // large batches: K = 224 is 7 pairs of K-tiles -- on the one-workgroup-per-tile kernel every tile pays pipeline fill, drain and a 128 KB store burst
    // (0.40 ms for 677 MB of output); the persistent barrier-free kernel streams the next tile's stages under the finished tile's stores
    CK(launch_split_rows(w.pf, kBlendK, N, kBlendK, kBlendK, N, 1, w.pf16h, w.pf16l, w.pfrs, s, m->opt));
    const half_t* sh = (const half_t*)(Bl + m->blendW_s);
    H3SArgs a{w.pf16h, w.pf16l, (long)N * 16, sh, sh + (size_t)kBlendN * kBlendK, (long)kBlendN * 16, kBlendK, w.vposed, (long)kVertLd,
              nullptr, 1.f / m->blend_sc, N, 3 * kNV, w.pfrs};
    if (w.sync) a.status = sync_reg_status(m, w.sync);
    a.fault = m->fault;
    return (int)launch_gemm_h3s(a, s, m->opt, 1);
  }
  if (w.split) {      // the prep kernel wrote the pose-feature planes next to the fp32 rows
    return h3_mm(m, w.pfP, Bl + m->blendW_p, kBlendN, kBlendK, w.vposed, kVertLd, nullptr, N, 3 * kNV, nullptr, 0, 0.f,
                 nullptr, s);
  }
  GemmArgs gv = gemm(w.pf, kBlendK, Bl + m->smpl.blendW, kBlendK, w.vposed, kVertLd, nullptr, N, 3 * kNV);
  return (int)launch_gemm(gv, s, m->opt);
}

}  // namespace

// ---- fault channel of the persistent kernels: the two tests every entry point and the kernel selection use
static inline bool fault_pending(const tepose_model* m) {
  return m->fault && __atomic_load_n(m->fault, __ATOMIC_RELAXED) != 0u;
}
// entry of a forward that owns status words in `workspace`: refused while the word is raised, else noted with the collection count it starts under
static inline int forward_begin(const tepose_model* m, const void* workspace) {
  if (fault_pending(m)) return TEPOSE_E_TIMEOUT;
  if (workspace) {
    std::lock_guard<std::mutex> g(m->q_mu);
    if (m->q_gen.size() > 4096) m->q_gen.clear();        // callers that never reuse a workspace: an unknown workspace takes the slow path, which is always right
    m->q_gen[workspace] = m->collected;
  }
  return 0;
}
// rows up to which the persistent kernels may run (the option, capped by what the kernels hold: 64 rows)
static inline int seq_rows_cap(const tepose_model* m) { return m->opt.seq_max_m > 64 ? 64 : m->opt.seq_max_m; }
static inline int reg_seq_rows_cap(const tepose_model* m) { return m->opt.reg_seq_max_n > 64 ? 64 : m->opt.reg_seq_max_n; }
static inline void fault_collected(const tepose_model* m) {
  std::lock_guard<std::mutex> g(m->q_mu);
  ++m->collected;
}
static inline bool persist_on(const tepose_model* m) {
  return __atomic_load_n(&m->persist, __ATOMIC_RELAXED) && m->fault != nullptr;   // (tepose_set_persistent may run on another thread)
}

namespace {

// ---- kernel selection: EVERY batch-class decision of a forward in one place ------------------------------------------------------------
// (the predicates used to be spread over encoder_fwd_impl / encoder_core / carve_regressor as conjunctions of knob bits.)
// Pure host function of (handle knobs, L, Hp, B, T): no device call, so tests/test_dispatch.py pins every class boundary on a machine without a GPU
// (tepose_select_kernels).  The launch code below consumes these fields; nothing else decides a kernel family.
struct KernelPlan {
  bool h3 = false;            // split-precision kernels (split-mode handle, B > TEPOSE_SPLIT_MIN_M); else the exact-fp32 kernels of gemm.hip / skinny.hip
  bool scaled = false;        // large batch: recurrent-state planes in the scaled format, layer >= 1 projections and cell steps on the scaled-plane kernels
  bool gblk = false;          // ... with the layer >= 1 gate pre-activations and the fp32 states between steps in the 16 x 16-blocked layout
  bool planes_state = false;  // ... and the step kernel's PLANES instantiation (every tile full: B % 128 == 0)
  bool g0big = false, g0mid = false, g0blk = false, g0skinny = false;     // layer-0 projection class
  bool seq2 = false, seq3 = false;   // the persistent recurrent kernel serves 2- / 3-direction layers of this (B, T)
  bool step_skinny = false;   // (not scaled, not seq) width-first step kernel
  bool reg_split = false, reg_seq = false, blend16 = false;   // regressor / SMPL side at N = B persons (1 - 4 persons: smpl_small_kernel, decided by smpl_small_ok at launch)
};

KernelPlan select_kernels(const tepose_model* m, int B, int T, bool assume_ready = false) {
  KernelPlan k;
  const int L = m->L, Hp = m->Hp;
  const long BT = (long)B * T;
  k.h3 = m->split && B > m->opt.split_min_m;
  k.scaled = k.h3 && m->large_scaled && B >= m->s_min_b;
  k.gblk = k.scaled && m->gi_blk && Hp % 32 == 0;
  k.planes_state = k.gblk && m->state_planes && B % 128 == 0;
  // layer-0 projection
  k.g0big = k.h3 && m->large_scaled && L >= 2 && BT >= 8192;
  const int g0mid_min = m->opt.g0_mid_min_rows;
  k.g0mid = k.h3 && m->large_scaled && L >= 2 && !k.g0big && BT >= g0mid_min && BT > 128 && (9 * Hp) % 288 == 0;
  if (k.g0mid) {   // whichever tile shape needs less time in whole rounds of the 256 CUs (a 128 x 288 tile takes ~2.1x a 128 x 128 one)
    const long rt = (BT + 127) / 128;
    const long r_mid = (rt * (9 * Hp / 288) + 255) / 256, r_old = (rt * ((9 * Hp + 127) / 128) + 255) / 256;
    k.g0mid = 2.1 * (double)r_mid <= (double)r_old + 0.15;
  }
  // frame-major + blocked layer-0 gate pre-activations: the same condition as gblk, plus whole row tiles per frame
  k.g0blk = k.g0big && k.gblk && B % 16 == 0;
  const int g0_skinny_max = m->opt.g0_skinny_max_m;
  k.g0skinny = k.h3 && !k.g0big && !k.g0mid && BT <= g0_skinny_max;
  // recurrent part of small batches
  const bool persist = assume_ready ? m->persist : persist_on(m);
  const bool seq_ok = k.h3 && !k.scaled && persist;
  k.seq3 = seq_ok && gru_seq_ok(3, B, Hp, T, m->opt);
  k.seq2 = seq_ok && gru_seq_ok(2, B, Hp, T, m->opt);
  k.step_skinny = k.h3 && !k.scaled && B <= m->opt.skinny_h3_max_m;
  // regressor / SMPL
  k.reg_split = reg_split_for(m, B);
  k.reg_seq = k.reg_split && B <= reg_seq_rows_cap(m) && persist;
  k.blend16 = blend16_for(m, B);
  return k;
}

// the kernel symbols (as a rocprofv3 trace names them) a default eval forward of B windows x T frames launches, family by family
std::string describe_plan(const tepose_model* m, int B, int T) {
  const KernelPlan k = select_kernels(m, B, T, true);
  const int L = m->L;
  const long BT = (long)B * T;
  std::string s = "input=";
  s += !k.h3 ? "pad_input_kernel" : split_rows_few_ok(BT, kInputP, k.g0blk ? T : 0, m->opt) ? "split_rows_few_kernel" : "split_rows_kernel";
  s += ";projection=";
  s += !k.h3 ? (BT <= m->opt.skinny_max_m ? "skinny_gemm_kernel" : "gemm_f32_kernel")
       : k.g0big ? "gemm_h3s_persist16c_kernel<0>" : k.g0mid ? "gemm_h3s_kernel<1, 3, 4, 3, 4>" : k.g0skinny ? "skinny_gemm_h3_kernel" : "gemm_h3_kernel";
  s += std::string(";gi0_layout=") + (k.g0blk ? "frame_major_blocked" : "row_major");
  const bool seq_l0 = L == 1 ? k.seq2 : k.seq3;
  s += ";gru_step=";
  s += !k.h3 ? (B <= m->opt.skinny_max_m ? "skinny_gru_kernel" : "gru_step_kernel")
       : k.scaled ? (k.planes_state && k.g0blk ? "gru_step16_kernel<true>" : "gru_step16_kernel<false>")
       : seq_l0 ? (B <= gru_seq_gran_rows(m->opt) && gru_seq_shape_ok(m->Hp) ? "gru_seq_kernel(granules)" : "gru_seq_kernel")
       : k.step_skinny ? "skinny_gru_h3_kernel" : "gemm_h3_kernel<GRU>";
  s += ";gru_first=";
  s += !k.h3 ? "gru_step_kernel" : (seq_l0 ? "(in gru_seq_kernel)" : (k.scaled && gru_first16_shape_ok(m->Hp) ? "gru_first16_kernel" : "gru_first_kernel"));
  if (L >= 2) {
    s += ";projection_l1=";
    s += !k.h3 ? "gemm_f32_kernel" : k.scaled ? "gemm_h3s_persist16c_kernel<1>" : (BT <= m->opt.l1_skinny_max_rows ? "skinny_gemm_h3_kernel" : "gemm_h3_kernel");
    s += std::string(";gi1_layout=") + (k.gblk ? "blocked" : "row_major");
    if (k.scaled) s += std::string(";gru_step_l1=") + (k.planes_state ? "gru_step16_kernel<true>" : "gru_step16_kernel<false>");
  }
  s += ";tail_regressor=";
  s += !k.reg_split ? "gemm_f32_kernel x (2 + 1 + 9)" : (m->tail_collapsed || !m->enc_packed) && m->collapse_env ? "collapsed: one product (skinny_gemm_h3_kernel / gemm_h3_kernel)"
       : k.reg_seq ? "reg_seq_kernel" : "gemm_h3_kernel loop";
  s += ";smpl=";
  s += (smpl_small_rows_ok(B, m->opt) && (m->lbs_sparse || !m->smpl_packed)) ? "smpl_small_kernel" : k.blend16 ? "smpl_prep_kernel+gemm_h3s_persist16c_kernel<1>+smpl_skin4_kernel" : k.reg_split ? "smpl_prep_kernel+gemm_h3_kernel+smpl_skin4_kernel"
       : "smpl_prep_kernel+gemm_f32_kernel+smpl_skin4_kernel";
  return s;
}

void refresh_kernel_info(tepose_model* m) {
  // the symbols a rocprofv3 kernel trace of cfg-C (B = 8192, T = 16) lists for the two dominant launch families -- what a committed profile must
  // name to describe THIS binary with THESE knobs (bench.py checks)
  const std::string d = describe_plan(m, 8192, 16);
  auto field = [&](const char* key) {
    const std::string kk = std::string(key) + "=";
    const size_t i = d.find(kk);
    if (i == std::string::npos) return std::string("?");
    const size_t j = d.find(';', i);
    return d.substr(i + kk.size(), j == std::string::npos ? std::string::npos : j - i - kk.size());
  };
  m->kinfo = "projection=" + field("projection") + ";gru_step=" + field("gru_step");
}
}  // namespace

// numerics / dispatch knobs, read once per handle at creation (both model kinds)
static void read_env_knobs(tepose_model* m) {
  m->opt = options_from_env();                      // the launch thresholds (common.h Options)
  const char* e = getenv("TEPOSE_EXACT_FP32");      // 1: keep every product on the exact-fp32 MFMA
  m->split = m->split_env = !(e && atoi(e) != 0);
  e = getenv("TEPOSE_LARGE_BATCH_KERNELS");         // scaled (default) | twoacc
  m->large_scaled = !(e && std::string(e) == "twoacc");
  e = getenv("TEPOSE_GRU_STATE");                   // planes (default) | fp32
  m->state_planes = !(e && std::string(e) == "fp32");
  e = getenv("TEPOSE_COLLAPSE_REGRESSOR");          // 0: the regressor's FC loop is always run as a loop
  m->collapse_env = !(e && atoi(e) == 0);
  e = getenv("TEPOSE_S_MIN_B");                     // batch threshold of the scaled-format recurrent path
  m->s_min_b = e ? atoi(e) : 640;                   // measured crossover against the two-accumulator recurrent path (round 4, 16x16x32 step + blocked
                                                    // operands: B = 512 2.38 vs 2.63 ms, 640 3.17 vs 2.94, 1024 4.29 vs 3.90, 1536 6.70 vs 5.91; it was 2048)
  e = getenv("TEPOSE_PERSISTENT");                  // 0: never launch the persistent small-batch kernels (the remedy for GPUs
  m->persist = !(e && atoi(e) == 0);                // that are shared or CU-masked: they need all their workgroups resident)
  e = getenv("TEPOSE_SEQ_SPIN_LIMIT");              // polls before a persistent kernel's wait gives up
  if (e && atol(e) > 0) m->spin_limit = (unsigned)atol(e);
  e = getenv("TEPOSE_TEST_FAULT");                  // tests only: make the persistent kernels' waits unmeetable
  m->test_fault = e ? (unsigned)atoi(e) : 0u;
  e = getenv("TEPOSE_BLEND16_MIN_N");
  m->blend16_min_n = e ? atoi(e) : 512;
  e = getenv("TEPOSE_GI_BLK");
  m->gi_blk = e ? atoi(e) : 1;
  refresh_kernel_info(m);
}

namespace {
// The regressor's loop (spin.py:252-261) in eval mode, with s = [pose6d | shape | cam] (157 values):
//   h1 = W1a f + W1b s + b1,  h2 = W2 h1 + b2,  s' = s + Wd h2 + bd        (no activation; Dropout is the identity)
// is affine:  s' = G s + F f + c  with  P = Wd W2,  G = I + P W1b,  F = P W1a,  c = P b1 + Wd b2 + bd,  so after three
// iterations from the model's own initial state s0:  s3 = (I + G + G^2)(F f + c) + G^3 s0 = Mf f + k0.
// All products in fp64 on the device (a few hundred MFLOP, pack time only), rounded to fp32 once.
int collapse_regressor(tepose_model* m, hipStream_t s) {
  m->reg_collapsed = false;
  m->tail_collapsed = false;
  if (!m->collapse_env || !m->reg_packed) return 0;
  constexpr int S = 157;
  float* B = m->blob;
  const size_t nP = (size_t)S * 1024, nF = (size_t)S * kFeat, nG = (size_t)S * S;
  double* d = nullptr;
  CK(hipMalloc((void**)&d, (nP + 2 * nF + 4 * nG + 5 * S) * sizeof(double)));
  double *P = d, *F = P + nP, *Mf = F + nF, *G = Mf + nF, *G2 = G + nG, *G3 = G2 + nG, *Ss = G3 + nG;
  double *t1 = Ss + nG, *c = t1 + S, *t2 = c + S, *k0 = t2 + S;
  auto run = [&]() -> int {
    CK(launch_dmm(B + m->wdec, 0, 1024, B + m->w2, 0, 1024, nullptr, 0, nullptr, 0, P, 1024, S, 1024, 1024, 1.0, 0, s));
    CK(launch_dmm(P, 1, 1024, B + m->w1a, 0, kFeat, nullptr, 0, nullptr, 0, F, kFeat, S, kFeat, 1024, 1.0, 0, s));
    CK(launch_dmm(P, 1, 1024, B + m->w1b, 0, kState, nullptr, 0, nullptr, 0, G, S, S, S, 1024, 1.0, 1, s));
    CK(launch_dmm(B + m->wdec, 0, 1024, B + m->b2, 0, 1, nullptr, 0, nullptr, 0, t1, 1, S, 1, 1024, 1.0, 0, s));
    CK(launch_dmm(P, 1, 1024, B + m->b1, 0, 1, t1, 1, B + m->bdec, 1, c, 1, S, 1, 1024, 1.0, 0, s));
    CK(launch_dmm(G, 1, S, G, 1, S, nullptr, 0, nullptr, 0, G2, S, S, S, S, 1.0, 0, s));
    CK(launch_dmm(G2, 1, S, G, 1, S, nullptr, 0, nullptr, 0, G3, S, S, S, S, 1.0, 0, s));
    CK(launch_dmm(G, 1, S, G, 1, S, G, S, nullptr, 0, Ss, S, S, S, S, 1.0, 1, s));                    // I + G + G^2
    CK(launch_dmm(Ss, 1, S, F, 1, kFeat, nullptr, 0, nullptr, 0, Mf, kFeat, S, kFeat, S, 1.0, 0, s));
    CK(launch_dmm(G3, 1, S, B + m->init, 0, 1, nullptr, 0, nullptr, 0, t2, 1, S, 1, S, 1.0, 0, s));
    CK(launch_dmm(Ss, 1, S, c, 1, 1, t2, 1, nullptr, 0, k0, 1, S, 1, S, 1.0, 0, s));
    CK(launch_d2f_pad(Mf, kFeat, S, kFeat, B + m->mf, 256, kFeat, s));
    CK(launch_d2f_pad(k0, S, 1, S, B + m->k0, 1, kState, s));
    CK((hipError_t)planes_of(B + m->mf, 256, kFeat, B + m->mf_p, s));
    CK(hipStreamSynchronize(s));
    return 0;
  };
  const int rc = run();
  (void)hipFree(d);
  if (rc) return rc;
  bool ok = false;                                        // the collapsed matrix must fit the fp16 planes like any weight
  CK((hipError_t)range_check(m, m->mf, m->mf_p, &ok, s));
  m->reg_collapsed = ok;
  return 0;
}

// ... and through the tail linears (tepose.py:81-86, eval mode: feat = (relu(h_fwd) W_lf^T + b_lf + relu(y_rec0) W_lr^T + b_lr) / 2):
//   xs = [relu(h_fwd) | relu(y_rec0)] Mt^T + kt,   Mt = Mf [W_lf | W_lr] / 2,   kt = Mf (b_lf + b_lr) / 2 + k0
int collapse_tail(tepose_model* m, hipStream_t s) {
  m->tail_collapsed = false;
  if (m->kind != 0 || !m->collapse_env || !m->reg_collapsed || !m->enc_packed) return 0;
  constexpr int S = 157;
  const int Hp = m->Hp, K3 = 3 * Hp;
  float* B = m->blob;
  double* d = nullptr;
  CK(hipMalloc((void**)&d, ((size_t)S * K3 + 2 * S) * sizeof(double)));
  double *Mt = d, *t = Mt + (size_t)S * K3, *kt = t + S;
  auto run = [&]() -> int {
    CK(launch_dmm(B + m->mf, 0, kFeat, B + m->wlf, 0, Hp, nullptr, 0, nullptr, 0, Mt, K3, S, Hp, kFeat, 0.5, 0, s));
    CK(launch_dmm(B + m->mf, 0, kFeat, B + m->wlr, 0, 2 * Hp, nullptr, 0, nullptr, 0, Mt + Hp, K3, S, 2 * Hp, kFeat, 0.5, 0, s));
    CK(launch_dmm(B + m->mf, 0, kFeat, B + m->blf, 0, 1, nullptr, 0, nullptr, 0, t, 1, S, 1, kFeat, 0.5, 0, s));
    CK(launch_dmm(B + m->mf, 0, kFeat, B + m->blr, 0, 1, t, 1, B + m->k0, 1, kt, 1, S, 1, kFeat, 0.5, 0, s));
    CK(launch_d2f_pad(Mt, K3, S, K3, B + m->mt, 256, K3, s));
    CK(launch_d2f_pad(kt, S, 1, S, B + m->kt, 1, kState, s));
    CK((hipError_t)planes_of(B + m->mt, 256, K3, B + m->mt_p, s));
    CK(hipStreamSynchronize(s));
    return 0;
  };
  const int rc = run();
  (void)hipFree(d);
  if (rc) return rc;
  bool ok = false;
  CK((hipError_t)range_check(m, m->mt, m->mt_p, &ok, s));
  m->tail_collapsed = ok;
  return 0;
}
}  // namespace

extern "C" {

int tepose_version(void) { return TEPOSE_ABI_VERSION; }

const char* tepose_build_info(void) {
#ifdef TEPOSE_NO_PACKED_FP32
  return "gfx950 packed_fp32=off";
#else
  return "gfx950 packed_fp32=on";
#endif
}

const char* tepose_kernel_info(const tepose_model* m) { return m ? m->kinfo.c_str() : ""; }

const char* tepose_select_kernels(const tepose_model* m, int B, int T) {
  static thread_local std::string buf;
  if (!m || B < 1 || T < 1) return "";
  buf = describe_plan(m, B, T);
  return buf.c_str();
}

const char* tepose_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case TEPOSE_E_ARG: return "tepose: bad argument";
    case TEPOSE_E_SHAPE: return "tepose: unsupported shape";
    case TEPOSE_E_WORKSPACE: return "tepose: workspace too small";
    case TEPOSE_E_STATE: return "tepose: model not packed";
    case TEPOSE_E_TIMEOUT: return "tepose: a persistent kernel's bounded wait expired (GPU shared or CU-masked?); "
                                  "outputs of that forward are NaN -- tepose_status() clears, TEPOSE_PERSISTENT=0 / "
                                  "tepose_set_persistent(m, 0) selects the step-per-launch kernels";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "tepose: unknown error";
  }
}

int tepose_create(int n_layers, int hidden, tepose_model** out) {
  if (!out || n_layers < 1 || hidden < 1) return TEPOSE_E_ARG;
  if (n_layers > 8 || hidden > 8192) return TEPOSE_E_SHAPE;
  tepose_model* m = new (std::nothrow) tepose_model();
  if (!m) return TEPOSE_E_ARG;
  m->L = n_layers; m->H = hidden; m->Hp = round_up(hidden, 64);
  read_env_knobs(m);
  layout(m);
  *out = m;
  return 0;
}

void tepose_destroy(tepose_model* m) {
  if (!m) return;
  for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : m->ev_gru) (void)hipEventDestroy(e);
  if (m->fault) (void)hipHostFree(m->fault);
  delete m;
}

size_t tepose_packed_bytes(const tepose_model* m) { return m ? m->blob_floats * sizeof(float) : 0; }

int tepose_set_blob(tepose_model* m, void* blob, size_t bytes) {
  if (!m || !blob) return TEPOSE_E_ARG;
  if (bytes < m->blob_floats * sizeof(float)) return TEPOSE_E_WORKSPACE;
  m->blob = (float*)blob;
  h3s16c_warm();                                 // the blob's device is current (callers run under it): its debug counter of the barrier-free kernels
  if (!m->fault) {                               // first blob = first moment a device is certain to exist
    unsigned* f = nullptr;
    if (hipHostMalloc((void**)&f, 64, hipHostMallocDefault) == hipSuccess && f) { *f = 0u; m->fault = f; }
    else (void)hipGetLastError();                // no fault word: the persistent kernels are not used (uses_persistent)
  }
  return 0;
}

// ---- fault channel of the persistent kernels ------------------------------------------------------------------

int tepose_status_peek(const tepose_model* m) {
  if (!m) return TEPOSE_E_ARG;
  return fault_pending(m) ? TEPOSE_E_TIMEOUT : 0;
}

int tepose_status(tepose_model* m, void* stream) {
  if (!m) return TEPOSE_E_ARG;
  CK(hipStreamSynchronize((hipStream_t)stream));
  if (!m->fault) return 0;
  const unsigned code = __atomic_exchange_n(m->fault, 0u, __ATOMIC_RELAXED);
  if (code) { m->last_fault_code = (int)code; fault_collected(m); }
  return code != 0u ? TEPOSE_E_TIMEOUT : 0;
}

int tepose_forward_status(tepose_model* m, void* workspace, void* stream) {
  if (!m || !workspace) return TEPOSE_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  CK(hipStreamSynchronize(s));
  // every give-up raises BOTH the forward's status word (workspace) and the handle's fault word (pinned host memory, system scope): with the stream
  // drained, a clear fault word means no forward of this handle gave up -- one host-memory read, no copy (the D2H copy + two null-stream
  // memsets per small-batch forward of sync mode) -- PROVIDED nobody has collected a fault since this forward was queued (another thread's
  // tepose_status / tepose_forward_status clears the shared word; this forward's own status word is then the only trace of its give-up)
  if (m->fault && __atomic_load_n(m->fault, __ATOMIC_RELAXED) == 0u) {
    std::lock_guard<std::mutex> g(m->q_mu);
    auto it = m->q_gen.find(workspace);
    if (it != m->q_gen.end() && it->second == m->collected) return 0;
  }
  // the sync region is the first carve of every workspace (carve_encoder / carve_regressor): [.. | gru status | .. | reg status]
  unsigned* sy = (unsigned*)workspace;
  // one copy of the span [recurrent status .. regressor status] (129 words: the regressor's arrival counters lie between them), on the caller's stream
  unsigned span[32 + 96 + 1];
  static_assert(sizeof(span) == (32 + 96 + 1) * sizeof(unsigned), "span");
  CK(hipMemcpyAsync(span, sync_gru_status(m, sy), sizeof(span), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  const unsigned st[2] = {span[0], span[32 + 96]};
  if ((st[0] | st[1]) == 0u) return 0;          // somebody else's forward raised the handle's word: theirs to collect
  // once per faulted forward: a later forward that launches no persistent kernel does not clear the words itself
  CK(hipMemsetAsync(sync_gru_status(m, sy), 0, sizeof(unsigned), s));
  CK(hipMemsetAsync(sync_reg_status(m, sy), 0, sizeof(unsigned), s));
  CK(hipStreamSynchronize(s));
  m->last_fault_code = (int)(st[0] ? st[0] : st[1]);
  if (m->fault && __atomic_exchange_n(m->fault, 0u, __ATOMIC_RELAXED) != 0u) fault_collected(m);   // (cleared, or every entry point would go on refusing)
  return TEPOSE_E_TIMEOUT;
}

// One named option of this handle (common.h Options + the handle's own batch thresholds), before anything is packed: workspace sizes and the packed
// planes depend on them.  name = the environment variable's, with or without the TEPOSE_ prefix.
int tepose_set_option(tepose_model* m, const char* name, long value) {
  if (!m || !name) return TEPOSE_E_ARG;
  if (m->enc_packed || m->reg_packed || m->smpl_packed || m->vibe_packed) return TEPOSE_E_STATE;
  const char* n = strncmp(name, "TEPOSE_", 7) == 0 ? name + 7 : name;
  if (int* f = option_field(m->opt, n)) *f = (int)value;
  else if (strcmp(n, "S_MIN_B") == 0) m->s_min_b = (int)value;
  else if (strcmp(n, "BLEND16_MIN_N") == 0) m->blend16_min_n = (int)value;
  else if (strcmp(n, "GI_BLK") == 0) m->gi_blk = (int)value;
  else if (strcmp(n, "SEQ_SPIN_LIMIT") == 0) { if (value <= 0) return TEPOSE_E_ARG; m->spin_limit = (unsigned)value; }
  else return TEPOSE_E_ARG;
  refresh_kernel_info(m);
  return 0;
}

long tepose_get_option(const tepose_model* m, const char* name) {
  if (!m || !name) return -1;
  const char* n = strncmp(name, "TEPOSE_", 7) == 0 ? name + 7 : name;
  Options o = m->opt;
  if (const int* f = option_field(o, n)) return *f;
  if (strcmp(n, "S_MIN_B") == 0) return m->s_min_b;
  if (strcmp(n, "BLEND16_MIN_N") == 0) return m->blend16_min_n;
  if (strcmp(n, "GI_BLK") == 0) return m->gi_blk;
  if (strcmp(n, "SEQ_SPIN_LIMIT") == 0) return (long)m->spin_limit;
  return -1;
}

int tepose_fault_code(const tepose_model* m) {
  if (!m) return TEPOSE_E_ARG;
  const unsigned live = m->fault ? __atomic_load_n(m->fault, __ATOMIC_RELAXED) : 0u;
  return live ? (int)live : m->last_fault_code;
}

int tepose_set_persistent(tepose_model* m, int on) {
  if (!m) return TEPOSE_E_ARG;
  __atomic_store_n(&m->persist, on != 0, __ATOMIC_RELAXED);
  return 0;
}

unsigned tepose_debug_kernel_errors(void) { return h3s16c_read_err(); }

int tepose_debug_set_test_fault(tepose_model* m, unsigned bits) {
  if (!m) return TEPOSE_E_ARG;
  m->test_fault = bits;
  return 0;
}

int tepose_uses_persistent(const tepose_model* m, int B, int T) {
  if (!m || B < 1) return 0;
  if (!persist_on(m) || !m->split || B <= m->opt.split_min_m) return 0;
  const int cap = seq_rows_cap(m) > reg_seq_rows_cap(m) ? seq_rows_cap(m) : reg_seq_rows_cap(m);
  (void)T;
  return B <= cap ? 1 : 0;
}

int tepose_create_vibe_ex(int n_layers, int hidden, int bidirectional, int add_linear, tepose_model** out) {
  if (!out || n_layers < 1 || hidden < 1) return TEPOSE_E_ARG;
  if (n_layers > 8 || hidden > 8192) return TEPOSE_E_SHAPE;
  tepose_model* m = new (std::nothrow) tepose_model();
  if (!m) return TEPOSE_E_ARG;
  m->kind = 1; m->L = n_layers; m->H = hidden; m->Hp = round_up(hidden, 64);
  m->vibe_bidir = bidirectional != 0;
  m->vibe_linear = bidirectional != 0 || add_linear != 0;      // vibe.py:43-47: a bidirectional encoder always has the linear
  read_env_knobs(m);        // TEPOSE_EXACT_FP32 covers the bootstrap model's regressor / blend-shape products too
  layout_vibe(m);
  *out = m;
  return 0;
}

int tepose_create_vibe(int n_layers, int hidden, tepose_model** out) {
  return tepose_create_vibe_ex(n_layers, hidden, 0, 1, out);
}

int tepose_vibe_feature_dim(const tepose_model* m) {
  if (!m || m->kind != 1) return 0;
  return m->vibe_linear ? kFeat : m->H;
}

int tepose_adopt_blob(tepose_model* m) {
  if (!m) return TEPOSE_E_ARG;
  if (!m->blob) return TEPOSE_E_STATE;
  BlobHeader h{};
  CK(hipMemcpy(&h, m->blob + m->hdr, sizeof(h), hipMemcpyDeviceToHost));                      // set-up time only
  const uint64_t lf = ((uint64_t)h.layout_floats_hi << 32) | h.layout_floats_lo;
  if (h.magic != kBlobMagic || h.abi != TEPOSE_ABI_VERSION || h.kind != header_kind(m) || (int)h.L != m->L ||
      (int)h.H != m->H || (int)h.Hp != m->Hp || lf != (uint64_t)m->blob_floats)
    return TEPOSE_E_STATE;                        // not a blob of this model kind / size / library layout
  m->enc_packed = m->kind == 0 && (h.sections & 1u);
  m->vibe_packed = m->kind == 1 && (h.sections & 1u);
  m->reg_packed = (h.sections & 2u) != 0;
  m->smpl_packed = (h.sections & 4u) != 0;
  m->enc_range_ok = m->reg_range_ok = m->smpl_range_ok = !(h.sections & 8u);   // bit 3: a weight outside the fp16 range
  m->reg_collapsed = m->collapse_env && (h.sections & 16u) != 0;      // (a handle created with TEPOSE_COLLAPSE_REGRESSOR=0 keeps the loop)
  m->tail_collapsed = m->collapse_env && m->kind == 0 && (h.sections & 32u) != 0;
  m->split = m->split_env && m->enc_range_ok;
  m->maxdepth = kNJ - 1;   // upper bound; chain levels past the real depth are no-ops
  int max_nnz = kNJ;
  CK(hipMemcpy(&max_nnz, m->blob + m->smpl.lbs_nnz, sizeof(int), hipMemcpyDeviceToHost));   // set-up time only
  m->lbs_sparse = max_nnz <= 4 ? 1 : 0;
  CK(hipMemcpy(&m->blend_sc, m->blob + m->blend_scale, sizeof(float), hipMemcpyDeviceToHost));
  if (!(m->blend_sc > 0.f)) m->blend_sc = 1.f;
  if (m->kind == 0 && m->enc_packed) {
    CK(hipMemcpy(&m->w0_scale, m->blob + m->wih0_scale, sizeof(float), hipMemcpyDeviceToHost));
    if (!(m->w0_scale > 0.f)) m->w0_scale = 1.f;
    for (int l = 0; l < m->L; ++l)
      for (DirW* d : {&m->fwd[l], &m->rec_f[l], &m->rec_r[l]}) {
        float sc[2] = {1.f, 1.f};
        CK(hipMemcpy(sc, m->blob + d->scales, sizeof(sc), hipMemcpyDeviceToHost));
        d->wih_scale = sc[0] > 0.f ? sc[0] : 1.f;
        d->whh_scale = sc[1] > 0.f ? sc[1] : 1.f;
      }
  }
  return 0;
}

// ---- broadcast less (round 3): every hi / lo plane in the blob is a function of the fp32 sections next to it --------------
// The blob interleaves source-of-truth fp32 sections (packed matrices, biases, SMPL tables, the fp64-derived collapsed maps
// rounded to fp32, the header) with the plane copies the split-precision kernels read.  tepose_fp32_ranges lists the former as
// byte ranges; a rank that received only those (276 of 770 MB at L = 2 / H = 1024) rebuilds the rest with tepose_derive_planes,
// which also does what tepose_adopt_blob does.  The planes come out bit-identical to the packing rank's
// (tests/test_gpu_multirank.py::test_planes_derived_from_the_fp32_sections_are_bit_identical).
int tepose_fp32_ranges(const tepose_model* m, size_t* offsets, size_t* sizes, int cap) {
  if (!m || !offsets || !sizes) return TEPOSE_E_ARG;
  size_t r[6][2];
  int n = 0;
  const size_t enc_planes = m->kind == 0 ? m->wih0_p : m->w1a;   // VIBE handles keep no encoder planes
  r[n][0] = 0; r[n][1] = enc_planes; ++n;                          // header + encoder fp32
  if (m->kind == 0) { r[n][0] = m->w1a; r[n][1] = m->w1a_p; ++n; } else r[0][1] = m->w1a_p;   // regressor + SMPL fp32
  r[n][0] = m->mf; r[n][1] = m->mf_p; ++n;
  if (m->kind == 0) {
    r[n][0] = m->k0; r[n][1] = m->mt_p; ++n;                       // k0 | mt
    r[n][0] = m->kt; r[n][1] = m->blob_floats; ++n;
  } else {
    r[n][0] = m->k0; r[n][1] = m->blob_floats; ++n;
  }
  if (n > cap) return TEPOSE_E_ARG;
  for (int i = 0; i < n; ++i) { offsets[i] = r[i][0] * sizeof(float); sizes[i] = (r[i][1] - r[i][0]) * sizeof(float); }
  return n;
}

int tepose_derive_planes(tepose_model* m, void* stream) {
  if (!m) return TEPOSE_E_ARG;
  if (!m->blob) return TEPOSE_E_STATE;
  int rc = tepose_adopt_blob(m);                 // header check, packed / range / collapse flags
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* B = m->blob;
  const int Hp = m->Hp, L = m->L;
  const int n128 = round_up(3 * Hp, 128);
  // blocked planes of a packed [rows][Kp] matrix whose plane copy has R >= rows rows (zero beyond)
  auto planes_rows = [&](size_t src, int rows, int Kp, size_t dst, int R) -> int {
    CK(launch_fill(B + dst, (size_t)R * Kp, 0.f, s));
    half_t* hi = (half_t*)(B + dst);
    return (int)launch_split_planes(B + src, Kp, rows, Kp, Kp, R, hi, hi + (size_t)R * Kp, s);
  };
  if (m->kind == 0 && m->enc_packed) {
    CK((hipError_t)planes_of(B + m->wih0, round_up(9 * Hp, 128), kInputP, B + m->wih0_p, s));
    CK((hipError_t)scaled_planes_of(B + m->wih0, 9 * Hp, kInputP, B + m->wih0_s, round_up(9 * Hp, 256), B + m->wih0_scale,
                                    &m->w0_scale, s));
    for (int l = 0; l < L; ++l) {
      DirW* dirs[3] = {&m->fwd[l], &m->rec_f[l], &m->rec_r[l]};
      for (int k = 0; k < 3; ++k) {
        DirW* d = dirs[k];
        const int Kp = k == 0 ? Hp : 2 * Hp;
        if (l > 0) {
          CK((hipError_t)planes_of(B + d->wih, n128, Kp, B + d->wih_p, s));
          CK((hipError_t)scaled_planes_of(B + d->wih, 3 * Hp, Kp, B + d->wih_s, round_up(3 * Hp, 256), B + d->scales,
                                          &d->wih_scale, s));
        }
        CK((hipError_t)planes_rows(d->whh, 3 * Hp, Hp, d->whh_p, n128));
        CK((hipError_t)scaled_planes_of(B + d->whh, 3 * Hp, Hp, B + d->whh_s, round_up(3 * Hp, 384), B + d->scales + 1,
                                        &d->whh_scale, s));
      }
    }
    CK((hipError_t)planes_of(B + m->wlf, kFeat, Hp, B + m->wlf_p, s));
    CK((hipError_t)planes_of(B + m->wlr, kFeat, 2 * Hp, B + m->wlr_p, s));
    {  // [W_lf | W_lr] side by side along K
      half_t* hi = (half_t*)(B + m->wlfr_p);
      half_t* lo = hi + (size_t)kFeat * 3 * Hp;
      const long kst = (long)kFeat * 32;
      CK(launch_split_planes(B + m->wlf, Hp, kFeat, Hp, Hp, kFeat, hi, lo, s));
      CK(launch_split_planes(B + m->wlr, 2 * Hp, kFeat, 2 * Hp, 2 * Hp, kFeat, hi + (size_t)(Hp / 32) * kst,
                             lo + (size_t)(Hp / 32) * kst, s));
    }
    if (m->tail_collapsed) CK((hipError_t)planes_of(B + m->mt, 256, 3 * Hp, B + m->mt_p, s));
  }
  if (m->reg_packed) {
    CK((hipError_t)planes_of(B + m->w1a, 1024, kFeat, B + m->w1a_p, s));
    CK((hipError_t)planes_of(B + m->w1b, 1024, kState, B + m->w1b_p, s));
    CK((hipError_t)planes_of(B + m->w2, 1024, 1024, B + m->w2_p, s));
    CK((hipError_t)planes_of(B + m->wdec, 256, 1024, B + m->wdec_p, s));
    if (m->reg_collapsed) CK((hipError_t)planes_of(B + m->mf, 256, kFeat, B + m->mf_p, s));
  }
  if (m->smpl_packed) CK((hipError_t)planes_of(B + m->smpl.blendW, kBlendN, kBlendK, B + m->blendW_p, s));
  if (m->smpl_packed) CK((hipError_t)scaled_planes_of(B + m->smpl.blendW, kBlendN, kBlendK, B + m->blendW_s, kBlendN, B + m->blend_scale, &m->blend_sc, s));
  CK(hipStreamSynchronize(s));
  return 0;
}

int tepose_pack_vibe_encoder(tepose_model* m, const float* const* w, int n_w, void* stream) {
  if (!m || !w || m->kind != 1) return TEPOSE_E_ARG;
  if (!m->blob) return TEPOSE_E_STATE;
  const int L = m->L, H = m->H, Hp = m->Hp, D = m->vibe_bidir ? 2 : 1;
  if (n_w != 4 * L * D + (m->vibe_linear ? 2 : 0)) return TEPOSE_E_ARG;
  for (int i = 0; i < n_w; ++i)
    if (!w[i]) return TEPOSE_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* B = m->blob;
  const int n128 = round_up(3 * D * Hp, 128);
  const int cmap = D == 2 ? COL_SPLIT2 : COL_PLAIN;            // layer >= 1 inputs and the linear read [fwd Hp | bwd Hp]
  for (int l = 0; l < L; ++l) {
    const int K = l == 0 ? kFeat : D * H, Kp = l == 0 ? kFeat : D * Hp;
    for (int d = 0; d < D; ++d) {
      const float* const* q = w + 4 * (l * D + d);              // weight_ih, weight_hh, bias_ih, bias_hh (nn.GRU's order)
      const int rows = d == D - 1 ? n128 - d * 3 * Hp : 3 * Hp; // the last direction also zeroes the padding rows
      CK((hipError_t)pack(q[0], K, 3 * H, K, B + m->vibe[l].wih + (size_t)d * 3 * Hp * Kp, rows, Kp, ROW_GATES,
                          l == 0 ? COL_PLAIN : cmap, H, Hp, s));
      CK((hipError_t)pack(q[2], 1, 3 * H, 1, B + m->vibe[l].bih + (size_t)d * 3 * Hp, 3 * Hp, 1, ROW_GATES, COL_PLAIN, H, Hp, s));
      CK((hipError_t)pack(q[1], H, 3 * H, H, B + m->vibe[l].whh + (size_t)d * 3 * Hp * Hp, 3 * Hp, Hp, ROW_GATES_TILED,
                          COL_PLAIN, H, Hp, s));
      CK((hipError_t)pack(q[3], 1, 3 * H, 1, B + m->vibe[l].bhh + (size_t)d * 3 * Hp, 3 * Hp, 1, ROW_GATES, COL_PLAIN, H, Hp, s));
    }
  }
  if (m->vibe_linear) {
    CK((hipError_t)pack(w[4 * L * D], D * H, kFeat, D * H, B + m->vlin_w, kFeat, D * Hp, ROW_PLAIN, cmap, H, Hp, s));
    CK((hipError_t)pack(w[4 * L * D + 1], 1, kFeat, 1, B + m->vlin_b, kFeat, 1, ROW_PLAIN, COL_PLAIN, H, Hp, s));
  }
  m->vibe_packed = true;
  CK((hipError_t)range_check(m, m->vibe[0].wih, m->w1a, &m->enc_range_ok, s));
  return write_header(m, s);
}

size_t tepose_vibe_workspace_bytes(const tepose_model* m, int B, int N) {
  if (!m || m->kind != 1 || B < 1 || N < 1) return 0;
  const size_t BN = (size_t)B * N, Hp = m->Hp, D = m->vibe_bidir ? 2 : 1;
  return align_up(BN * D * 3 * Hp * 4, 256) + 2 * align_up(BN * D * Hp * 4, 256) + 256;
}

int tepose_vibe_encoder_fwd(const tepose_model* m, const float* x, int B, int N, int use_residual, float* feat,
                            void* workspace, size_t ws_bytes, void* stream) {
  if (!m || m->kind != 1 || !x || !feat || !workspace || B < 1 || N < 1) return TEPOSE_E_ARG;
  if (!m->vibe_packed) return TEPOSE_E_STATE;
  if (ws_bytes < tepose_vibe_workspace_bytes(m, B, N)) return TEPOSE_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int L = m->L, Hp = m->Hp, H3 = 3 * Hp, D = m->vibe_bidir ? 2 : 1;
  const long BN = (long)B * N;
  Carver c(workspace, ws_bytes);
  float* G = c.f((size_t)BN * D * H3);
  float* S[2] = {c.f((size_t)BN * D * Hp), c.f((size_t)BN * D * Hp)};
  const float* Bl = m->blob;
  // everything is batch-major (row = b*N + t), like the caller's [B,N,2048]: time steps are a
  // column offset t*ld with row stride N*ld, so no permute (vibe.py:53,62) is ever materialised; a layer's
  // output row is [forward Hp | backward Hp], the backward direction walking t = N-1 .. 0
  const float* in = x;
  int ldin = kFeat;
  for (int l = 0; l < L; ++l) {
    GemmArgs g = gemm(in, ldin, Bl + m->vibe[l].wih, ldin, G, (long)D * H3, Bl + m->vibe[l].bih, (int)BN, D * H3);
    CK(launch_gemm(g, s, m->opt));
    float* So = S[l & 1];
    for (int t = 0; t < N; ++t) {
      GruArgs a{};
      a.M = B; a.Hp = Hp; a.first = t == 0; a.ndir = D;
      for (int d = 0; d < D; ++d) {
        const int td = d ? N - 1 - t : t, tp = d ? td + 1 : td - 1;
        GruDir& q = a.d[d];
        q.Whh = Bl + m->vibe[l].whh + (size_t)d * H3 * Hp; q.bhh = Bl + m->vibe[l].bhh + (size_t)d * H3;
        q.gi = G + (long)td * D * H3 + (long)d * H3; q.ldgi = (long)N * D * H3;
        q.hprev = So + (long)tp * D * Hp + (long)d * Hp; q.ldh = (long)N * D * Hp;
        q.hout = So + (long)td * D * Hp + (long)d * Hp; q.ldo = (long)N * D * Hp;
      }
      CK(launch_gru_step(a, s, m->opt));
    }
    in = So; ldin = D * Hp;
  }
  if (!m->vibe_linear)                                        // y = gru(x) (+ x when it is 2048 wide, vibe.py:55-61)
    return (int)launch_copy_cols(in, ldin, (use_residual && m->H == kFeat) ? x : nullptr, kFeat, feat, m->H, BN, m->H, s);
  GemmArgs g = gemm(in, ldin, Bl + m->vlin_w, ldin, feat, kFeat, Bl + m->vlin_b, (int)BN, kFeat);
  g.relu_a = 1;
  if (use_residual) { g.addend = x; g.ldadd = kFeat; }
  CK(launch_gemm(g, s, m->opt));
  return 0;
}

int tepose_pack_encoder(tepose_model* m, const float* const* w, int n_w, void* stream) {
  if (!m || !w || m->kind != 0) return TEPOSE_E_ARG;
  if (!m->blob) return TEPOSE_E_STATE;
  const int L = m->L, H = m->H, Hp = m->Hp;
  if (n_w != 12 * L + 4) return TEPOSE_E_ARG;
  for (int i = 0; i < n_w; ++i)
    if (!w[i]) return TEPOSE_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* B = m->blob;
  // zero the stacked layer-0 block first (rows beyond 9Hp up to the 128 multiple)
  CK(launch_fill(B + m->wih0, (size_t)round_up(9 * Hp, 128) * kInputP, 0.f, s));
  auto fwd_w = [&](int l, int k) { return w[4 * l + k]; };                       // ih, hh, bih, bhh
  auto rec_w = [&](int l, int rev, int k) { return w[4 * L + 8 * l + 4 * rev + k]; };
  const int n128 = round_up(3 * Hp, 128);
  // layer 0 input projections, stacked [fwd | rec_reverse | rec]
  const float* l0[3] = {fwd_w(0, 0), rec_w(0, 1, 0), rec_w(0, 0, 0)};
  const float* l0b[3] = {fwd_w(0, 2), rec_w(0, 1, 2), rec_w(0, 0, 2)};
  for (int d = 0; d < 3; ++d) {
    CK((hipError_t)pack(l0[d], kInput, 3 * H, kInput, B + m->wih0 + (size_t)d * 3 * Hp * kInputP, 3 * Hp,
                        kInputP, ROW_GATES, COL_PLAIN, H, Hp, s));
    CK((hipError_t)pack(l0b[d], 1, 3 * H, 1, B + m->bih0 + (size_t)d * 3 * Hp, 3 * Hp, 1, ROW_GATES,
                        COL_PLAIN, H, Hp, s));
  }
  {  // planes of the stacked layer-0 block: pack each direction's rows into the hi and the lo plane
    const size_t rows0 = (size_t)round_up(9 * Hp, 128);
    CK(launch_fill(B + m->wih0_p, rows0 * kInputP, 0.f, s));
    half_t* hi = (half_t*)(B + m->wih0_p);
    half_t* lo = hi + rows0 * kInputP;
    for (int d = 0; d < 3; ++d) {
      PackArgs a{l0[d], kInput, 3 * H, kInput, nullptr, 3 * Hp, kInputP, hi + (size_t)d * 3 * Hp * 32,
                 lo + (size_t)d * 3 * Hp * 32, (long)rows0 * 32, ROW_GATES, COL_PLAIN, H, Hp};
      CK(launch_pack(a, s));
    }
  }
  // the same block as scaled planes for the single-accumulator kernel (gemm_h3s.hip)
  CK((hipError_t)scaled_planes_of(B + m->wih0, 9 * Hp, kInputP, B + m->wih0_s, round_up(9 * Hp, 256),
                                  B + m->wih0_scale, &m->w0_scale, s));
  for (int l = 0; l < L; ++l) {
    struct { DirW* d; const float *ih, *hh, *bih, *bhh; bool split; } dirs[3] = {
        {&m->fwd[l], fwd_w(l, 0), fwd_w(l, 1), fwd_w(l, 2), fwd_w(l, 3), false},
        {&m->rec_f[l], rec_w(l, 0, 0), rec_w(l, 0, 1), rec_w(l, 0, 2), rec_w(l, 0, 3), true},
        {&m->rec_r[l], rec_w(l, 1, 0), rec_w(l, 1, 1), rec_w(l, 1, 2), rec_w(l, 1, 3), true}};
    for (auto& d : dirs) {
      if (l > 0) {
        const int K = d.split ? 2 * H : H, Kp = d.split ? 2 * Hp : Hp;
        CK((hipError_t)pack(d.ih, K, 3 * H, K, B + d.d->wih, n128, Kp, ROW_GATES,
                            d.split ? COL_SPLIT2 : COL_PLAIN, H, Hp, s));
        CK((hipError_t)pack(d.bih, 1, 3 * H, 1, B + d.d->bih, 3 * Hp, 1, ROW_GATES, COL_PLAIN, H, Hp, s));
        CK((hipError_t)pack_planes(d.ih, K, 3 * H, K, B + d.d->wih_p, n128, Kp, ROW_GATES,
                                   d.split ? COL_SPLIT2 : COL_PLAIN, H, Hp, s));
      }
      CK((hipError_t)pack_planes(d.hh, H, 3 * H, H, B + d.d->whh_p, n128, Hp, ROW_GATES_TILED, COL_PLAIN, H, Hp, s));
      CK((hipError_t)pack(d.hh, H, 3 * H, H, B + d.d->whh, 3 * Hp, Hp, ROW_GATES_TILED, COL_PLAIN, H, Hp, s));
      CK((hipError_t)pack(d.bhh, 1, 3 * H, 1, B + d.d->bhh, 3 * Hp, 1, ROW_GATES, COL_PLAIN, H, Hp, s));
      // scaled planes of the packed fp32 matrices (W_ih: natural gate order; W_hh: gate-tiled rows)
      if (l > 0) {
        const int Kp = d.split ? 2 * Hp : Hp;
        CK((hipError_t)scaled_planes_of(B + d.d->wih, 3 * Hp, Kp, B + d.d->wih_s, round_up(3 * Hp, 256),
                                        B + d.d->scales, &d.d->wih_scale, s));
      }
      CK((hipError_t)scaled_planes_of(B + d.d->whh, 3 * Hp, Hp, B + d.d->whh_s, round_up(3 * Hp, 384),
                                      B + d.d->scales + 1, &d.d->whh_scale, s));
    }
  }
  const float* const* t = w + 12 * L;
  CK((hipError_t)pack(t[0], H, kFeat, H, B + m->wlf, kFeat, Hp, ROW_PLAIN, COL_PLAIN, H, Hp, s));
  CK((hipError_t)pack(t[1], 1, kFeat, 1, B + m->blf, kFeat, 1, ROW_PLAIN, COL_PLAIN, H, Hp, s));
  CK((hipError_t)pack(t[2], 2 * H, kFeat, 2 * H, B + m->wlr, kFeat, 2 * Hp, ROW_PLAIN, COL_SPLIT2, H, Hp, s));
  CK((hipError_t)pack(t[3], 1, kFeat, 1, B + m->blr, kFeat, 1, ROW_PLAIN, COL_PLAIN, H, Hp, s));
  CK((hipError_t)planes_of(B + m->wlf, kFeat, Hp, B + m->wlf_p, s));
  CK((hipError_t)planes_of(B + m->wlr, kFeat, 2 * Hp, B + m->wlr_p, s));
  {  // [W_lf | W_lr] side by side along K: K-tiles [0, Hp/32) from linear_fwd, the rest from linear_rec
    half_t* hi = (half_t*)(B + m->wlfr_p);
    half_t* lo = hi + (size_t)kFeat * 3 * Hp;
    const long kst = (long)kFeat * 32;
    PackArgs a1{t[0], H, kFeat, H, nullptr, kFeat, Hp, hi, lo, kst, ROW_PLAIN, COL_PLAIN, H, Hp};
    CK(launch_pack(a1, s));
    PackArgs a2{t[2], 2 * H, kFeat, 2 * H, nullptr, kFeat, 2 * Hp, hi + (size_t)(Hp / 32) * kst, lo + (size_t)(Hp / 32) * kst,
                kst, ROW_PLAIN, COL_SPLIT2, H, Hp};
    CK(launch_pack(a2, s));
  }
  m->enc_packed = true;
  CK((hipError_t)range_check(m, m->wih0, m->wih0_p, &m->enc_range_ok, s));
  CK((hipError_t)collapse_tail(m, s));
  return write_header(m, s);
}

int tepose_pack_regressor(tepose_model* m, const float* const* w, int n_w, void* stream) {
  if (!m || !w || n_w != 13) return TEPOSE_E_ARG;
  if (!m->blob) return TEPOSE_E_STATE;
  for (int i = 0; i < n_w; ++i)
    if (!w[i]) return TEPOSE_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* B = m->blob;
  const int ld1 = kFeat + kNPose + 13;   // 2205
  CK((hipError_t)pack(w[0], ld1, 1024, kFeat, B + m->w1a, 1024, kFeat, 0, 0, 0, 1, s));
  CK((hipError_t)pack(w[1], 1, 1024, 1, B + m->b1, 1024, 1, 0, 0, 0, 1, s));
  CK((hipError_t)pack(w[0] + kFeat, ld1, 1024, 157, B + m->w1b, 1024, kState, 0, 0, 0, 1, s));
  CK((hipError_t)pack(w[2], 1024, 1024, 1024, B + m->w2, 1024, 1024, 0, 0, 0, 1, s));
  CK((hipError_t)pack(w[3], 1, 1024, 1, B + m->b2, 1024, 1, 0, 0, 0, 1, s));
  // decoders stacked: rows 0..143 decpose, 144..153 decshape, 154..156 deccam, rest zero
  CK(launch_fill(B + m->wdec, 256 * 1024, 0.f, s));
  CK(launch_fill(B + m->bdec, kState, 0.f, s));
  CK(launch_fill(B + m->init, kState, 0.f, s));
  const int rows[3] = {kNPose, 10, 3}, off[3] = {0, kNPose, kNPose + 10};
  for (int i = 0; i < 3; ++i) {
    CK((hipError_t)pack(w[4 + 2 * i], 1024, rows[i], 1024, B + m->wdec + (size_t)off[i] * 1024, rows[i], 1024,
                        0, 0, 0, 1, s));
    CK((hipError_t)pack(w[5 + 2 * i], 1, rows[i], 1, B + m->bdec + off[i], rows[i], 1, 0, 0, 0, 1, s));
    CK((hipError_t)pack(w[10 + i], 1, rows[i], 1, B + m->init + off[i], rows[i], 1, 0, 0, 0, 1, s));
  }
  CK((hipError_t)planes_of(B + m->w1a, 1024, kFeat, B + m->w1a_p, s));
  CK((hipError_t)planes_of(B + m->w1b, 1024, kState, B + m->w1b_p, s));
  CK((hipError_t)planes_of(B + m->w2, 1024, 1024, B + m->w2_p, s));
  CK((hipError_t)planes_of(B + m->wdec, 256, 1024, B + m->wdec_p, s));
  m->reg_packed = true;
  CK((hipError_t)range_check(m, m->w1a, m->smpl.J0, &m->reg_range_ok, s));
  CK((hipError_t)collapse_regressor(m, s));
  CK((hipError_t)collapse_tail(m, s));
  return write_header(m, s);
}

int tepose_pack_smpl(tepose_model* m, const float* v_template, const float* shapedirs,
                     const float* posedirs, const float* J_regressor, const float* lbs_weights,
                     const float* J_regressor_extra, const int32_t* parents_host, void* stream) {
  if (!m || !v_template || !shapedirs || !posedirs || !J_regressor || !lbs_weights ||
      !J_regressor_extra || !parents_host)
    return TEPOSE_E_ARG;
  if (!m->blob) return TEPOSE_E_STATE;
  hipStream_t s = (hipStream_t)stream;
  float* B = m->blob;
  int par[kNJ], dep[kNJ], maxd = 0;
  for (int j = 0; j < kNJ; ++j) {
    par[j] = parents_host[j];
    if (j == 0) { dep[j] = 0; par[j] = -1; continue; }
    if (par[j] < 0 || par[j] >= j) return TEPOSE_E_ARG;   // parents must precede children
    dep[j] = dep[par[j]] + 1;
    if (dep[j] > maxd) maxd = dep[j];
  }
  m->maxdepth = maxd;
  CK(hipMemcpyAsync(B + m->smpl.parents, par, sizeof(par), hipMemcpyHostToDevice, s));
  CK(hipMemcpyAsync(B + m->smpl.depth, dep, sizeof(dep), hipMemcpyHostToDevice, s));
  CK(hipStreamSynchronize(s));   // par/dep are stack arrays (pack time only, never on the forward path)
  CK(launch_smpl_consts(v_template, shapedirs, posedirs, J_regressor, B + m->smpl.J0, B + m->smpl.JS,
                        B + m->smpl.blendW, s));
  CK((hipError_t)planes_of(B + m->smpl.blendW, kBlendN, kBlendK, B + m->blendW_p, s));
  CK((hipError_t)scaled_planes_of(B + m->smpl.blendW, kBlendN, kBlendK, B + m->blendW_s, kBlendN, B + m->blend_scale, &m->blend_sc, s));
  CK((hipError_t)pack(lbs_weights, kNJ, kNV, kNJ, B + m->smpl.lbsW, kNV, kNJ, 0, 0, 0, 1, s));
  CK(launch_lbs_compact(lbs_weights, (int*)(B + m->smpl.lbs_cidx), B + m->smpl.lbs_cval, (int*)(B + m->smpl.lbs_nnz), s));
  int max_nnz = 0;
  CK(hipMemcpyAsync(&max_nnz, B + m->smpl.lbs_nnz, sizeof(int), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));                 // pack time only
  m->lbs_sparse = max_nnz <= 4 ? 1 : 0;
  CK(launch_csr_build(J_regressor_extra, 9, kNV, (int*)(B + m->smpl.xr_ptr), (int*)(B + m->smpl.xr_idx),
                      B + m->smpl.xr_val, 9 * kNV, s));
  m->smpl_packed = true;
  CK((hipError_t)range_check(m, m->smpl.J0, m->smpl.lbs_cidx, &m->smpl_range_ok, s));
  return write_header(m, s);
}

size_t tepose_jreg_packed_bytes(void) { return (32 + (size_t)17 * kNV * 2) * 4; }

int tepose_pack_jreg(const float* J, void* packed, void* stream) {
  if (!J || !packed) return TEPOSE_E_ARG;
  int* p = (int*)packed;
  CK(launch_csr_build(J, 17, kNV, p, p + 32, (float*)(p + 32 + 17 * kNV), 17 * kNV, (hipStream_t)stream));
  return 0;
}

size_t tepose_workspace_bytes(const tepose_model* m, int B, int T) {
  if (!m || B < 1 || T < 1) return 0;
  Carver c(nullptr, 0);
  EncWs e;
  carve_encoder(m, B, T, c, e);
  RegWs r;
  c.f((size_t)B * 2 * kFeat);          // feature buffer of tepose_forward
  carve_regressor(m, 2 * B, c, r);        // is_train regresses 2 rows per window
  return c.cur + 256;
}

int tepose_profile_enable(tepose_model* m, int on) {
  if (!m) return TEPOSE_E_ARG;
  m->prof = on != 0;
  m->ev_used = 0;
  m->prof_flops = 0.0;
  m->ev_gru_used = 0;
  m->prof_gru_flops = 0.0;
  m->prof_l1_flops = 0.0;
  return 0;
}

int tepose_profile_read(tepose_model* m, double* total_ms, int* n_launches, double* flops_per_launch) {
  if (!m || !total_ms || !n_launches || !flops_per_launch) return TEPOSE_E_ARG;
  double tot = 0.0;
  for (size_t i = 0; i + 1 < m->ev_used; i += 2) {
    CK(hipEventSynchronize(m->ev[i + 1]));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, m->ev[i], m->ev[i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *n_launches = (int)(m->ev_used / 2);
  *flops_per_launch = m->prof_flops;
  m->ev_used = 0;
  return 0;
}

namespace {

// Where the layer-0 gate pre-activations (x W_ih^T + b_ih, 9Hp columns: fwd | rec_reverse | rec) of a
// window's frames live.  Regular forward: one buffer, frame t at base + t*frame_stride.  Cached driver:
// frame t of the window sits in slot (first + t) % ring of a per-clip ring, except the newest frame
// (theta slots still zero), which has its own buffer.
struct G0Src {
  const float* base; long frame_stride, row_stride;
  int first, ring;                 // ring == 0: no wrap
  const float* last; long last_ld; // newest frame's projections or nullptr
  const float* single; long single_ld;   // L == 1: source of the one consumed rec.l0 forward step
  long blk = 0;                    // != 0: base is in the blocked layout (common.h gi_blk_offset), floats between 16-row tiles; frames are row_stride * B apart
};

int prof_mark(tepose_model* mm, hipStream_t s) {     // next event of the GRU-interval list
  if (mm->ev_gru.size() < mm->ev_gru_used + 1) {
    hipEvent_t e;
    CK(hipEventCreate(&e));
    mm->ev_gru.push_back(e);
  }
  CK(hipEventRecord(mm->ev_gru[mm->ev_gru_used++], s));
  return 0;
}

// xs_out (eval mode, tail_collapsed handles only): instead of the feature, write the regressor's final state rows
// [B][160] = [relu(h_fwd) | relu(y_rec0)] Mt^T + kt -- the tail linears and the three FC iterations as one product
int encoder_core(const tepose_model* m, const G0Src& src, int B, int T, int is_train, float* feat, EncWs& w,
                 hipStream_t s, const Planes* feat_planes = nullptr, bool sync_zeroed = false, float* xs_out = nullptr) {
  tepose_model* mm = const_cast<tepose_model*>(m);
  const int L = m->L, Hp = m->Hp;
  const float* Bl = m->blob;
  const long BT = (long)B * T;
  const int H3 = 3 * Hp;
  const KernelPlan plan = select_kernels(m, B, T);     // every batch-class decision (see select_kernels)
  const bool h3 = plan.h3;
  const long Bs = (long)w.Bs;          // rows per time slab of the layer >= 1 buffers
  // large batches: recurrent-state planes in the scaled format, layer >= 1 projections and GRU steps on the scaled-plane kernels
  const bool sf = plan.scaled;
  const size_t n128 = (size_t)round_up(H3, 128);
  // layer >= 1 gate pre-activations in the blocked layout (common.h gi_blk_offset): producer = the barrier-free projection kernel, consumers =
  // gru_step16_kernel / gru_first16_kernel / gru_first_kernel
  const bool gblk = plan.gblk;
  if (src.blk && !gblk) return (int)hipErrorInvalidValue;   // the caller projected layer 0 into the blocked layout: every consumer here must read it
  // input projection of a layer >= 1: fp32 kernel, or split kernel on the hi/lo mirrors of the input states
  auto proj = [&](const float* in, int K, size_t w_f32, size_t w_planes, size_t w_s, float w_scale, size_t bias,
                  float* out, int M) -> int {
    if (!h3) {
      GemmArgs g = gemm(in, K, Bl + w_f32, K, out, H3, Bl + bias, M, H3);
      return (int)launch_gemm(g, s, m->opt);
    }
    if (sf) {
      const EncWs::View v = w.view16(in);
      if (!v.hi) return (int)hipErrorInvalidValue;
      const size_t r256 = (size_t)round_up(H3, 256);
      const half_t* sh = (const half_t*)(Bl + w_s);
      H3SArgs a{v.hi, v.lo, v.kst, sh, sh + r256 * K, (long)r256 * 16, K, out, (long)H3, Bl + bias,
                1.f / (kStateScale * w_scale), M, H3};
      if (w.sync) a.status = sync_gru_status(m, w.sync);
      a.fault = m->fault;
      a.inject = (m->test_fault >> 2) & 1u;
      a.c_blk_hp = gblk ? Hp : 0;
      return (int)launch_gemm_h3s(a, s, m->opt);
    }
    const EncWs::View v = w.view(in);
    if (!v.hi) return (int)hipErrorInvalidValue;
    Planes A; A.hi = v.hi; A.lo = v.lo; A.kst = v.kst;
    return h3_mm(m, A, Bl + w_planes, (int)n128, K, out, (long)H3, Bl + bias, M, H3, nullptr, 0, 0.f, nullptr, s);
  };
  // one GRU step of up to 3 directions: fused fp32 kernel; or the split product with the cell update in its
  // epilogue (first step: h = 0, element-wise kernel)
  auto step = [&](const GruArgs& a, const size_t (&whh_planes)[3], const DirW* const (&dw)[3]) -> int {
    if (!h3) return (int)launch_gru_step(a, s, m->opt);
    if (sf) {
      H3SBatch b{};
      GateBatch gb{};
      const size_t r384 = (size_t)round_up(H3, 384);
      for (int d = 0; d < a.ndir; ++d) {
        const EncWs::View vo = w.view16(a.d[d].hout);
        if (!vo.hi) return (int)hipErrorInvalidValue;
        GateDir g{a.d[d].gi, a.d[d].ldgi, a.d[d].bhh, a.d[d].hprev, a.d[d].ldh, a.d[d].hout, a.d[d].ldo,
                  vo.hi, vo.lo, vo.kst};
        g.gi_blk = a.d[d].gi_blk;
        g.hprev_b = a.first ? nullptr : a.d[d].hprev_b; g.hp_blk = a.first ? 0 : a.d[d].hp_blk;
        g.hout_b = a.d[d].hout_b; g.ho_blk = a.d[d].ho_blk;
        b.gate[d] = g;
        gb.d[d] = g;
        if (!a.first) {
          const EncWs::View vi = w.view16(a.d[d].hprev);
          if (!vi.hi) return (int)hipErrorInvalidValue;
          const half_t* sh = (const half_t*)(Bl + dw[d]->whh_s);
          b.p[d] = H3SArgs{vi.hi, vi.lo, vi.kst, sh, sh + r384 * Hp, (long)r384 * 16, Hp, nullptr, 0, nullptr,
                           1.f / (kStateScale * dw[d]->whh_scale), B, H3};
          if (w.sync) b.p[d].status = sync_gru_status(m, w.sync);
          b.p[d].fault = m->fault;
          b.p[d].inject = (m->test_fault >> 2) & 1u;
        }
      }
      if (a.first) return (int)launch_gru_first(gb, a.ndir, B, Hp, s, 1);
      b.n = a.ndir; b.Hp = Hp; b.state_scale = kStateScale;
      // the plane-fed instantiation wants this layer's gate pre-activations blocked: layers >= 1 always are (gblk), layer 0 only where the projection
      // wrote them frame-major + blocked (g0blk; not from the driver's cache ring).  One decision per layer: every step of a layer runs the same kernel.
      bool planes = plan.planes_state;
      for (int d = 0; d < a.ndir; ++d) planes = planes && a.d[d].gi_blk != 0;
      planes = planes && gru_step16_planes_ok(b);          // (a misaligned view or a ragged tile the plan did not foresee: the general instantiation, not an error)
      return (int)launch_gru_step16(b, s, planes, m->opt.gru_gm);
    }
    H3Batch b{};
    GateBatch gb{};
    for (int d = 0; d < a.ndir; ++d) {
      const EncWs::View vo = w.view(a.d[d].hout);
      if (!vo.hi) return (int)hipErrorInvalidValue;
      const GateDir g{a.d[d].gi, a.d[d].ldgi, a.d[d].bhh, a.d[d].hprev, a.d[d].ldh, a.d[d].hout, a.d[d].ldo,
                      vo.hi, vo.lo, vo.kst};
      b.gate[d] = g;
      gb.d[d] = g;
      if (!a.first) {
        const half_t* wh = (const half_t*)(Bl + whh_planes[d]);
        const EncWs::View vi = w.view(a.d[d].hprev);
        if (!vi.hi) return (int)hipErrorInvalidValue;
        b.p[d] = H3Args{vi.hi, vi.lo, vi.kst, wh, wh + n128 * Hp, (long)n128 * 32, Hp, nullptr, 0, nullptr, B, H3};
      }
    }
    if (a.first) return (int)launch_gru_first(gb, a.ndir, B, Hp, s);
    b.n = a.ndir; b.Hp = Hp;
    if (plan.step_skinny) return (int)launch_skinny_gru_h3(b, s);
    return (int)launch_gru_h3(b, s);
  };
  auto gi0 = [&](int t, int dir, const float*& p, long& ld) {
    if (src.last && t == T - 1) { p = src.last + (long)dir * H3; ld = src.last_ld; return; }
    const int slot = src.ring ? (src.first + t) % src.ring : t;
    p = src.base + (long)slot * src.frame_stride + (long)dir * H3 * (src.blk ? 16 : 1);
    ld = src.row_stride;
  };
  // small batches: all T steps of a layer in one persistent launch (gru_seq.hip); its arrival counters are zeroed
  // by a memset node in front of the first launch of every forward
  const bool scaled_fmt = sf;         // (the layer loop below reuses the name `sf` for a state buffer)
  bool tail_planes_done = false;      // the persistent kernel of the top layer wrote relu(final states) as planes
  const bool seq = w.sync && (L == 1 ? plan.seq2 : plan.seq3);
  const size_t gran_bytes = seq_gran_words(m, B) * sizeof(float);
  // every forward clears its sync region -- arrival counters, granules, and the two STATUS words that tepose_forward_status
  // reads -- whether or not a persistent kernel will run (a stale or uninitialised status word would read as a give-up)
  if (!sync_zeroed && w.sync) CK(hipMemsetAsync(w.sync, 0, seq ? sync_zero_bytes(m, B) : sync_words(m) * sizeof(unsigned), s));
  for (int l = 0; l < L; ++l) {
    const bool top = l == L - 1;
    float* sf = w.sf[l & 1];
    float* sr = w.sr[l & 1];
    const float *gf, *grr, *grf;     // gate pre-activation sources of this layer
    long ldg;                        // row stride of the gate pre-activations
    if (l == 0) {
      gf = grr = grf = nullptr; ldg = 0;      // layer 0 reads through gi0()
    } else {
      const float* inf = w.sf[(l - 1) & 1];
      const float* inr = w.sr[(l - 1) & 1];
      const int MT = (int)(Bs * T);       // every slab row, pad rows included (their results are never read)
      const int Mf = top ? B : MT;         // the top layer's forward direction of gru_rec consumes one step only
      if (h3 && !scaled_fmt) {
        // the three products of a layer in as few launches as their shapes allow (each alone under-fills the chip:
        // 64-192 workgroups): width-first kernel for <= m->opt.skinny_max_m rows, 128/256-row tiles above
        const EncWs::View vf = w.view(inf), vr = w.view(inr);
        if (!vf.hi || !vr.hi) return (int)hipErrorInvalidValue;
        auto mk = [&](const EncWs::View& v, int K, size_t w_planes, size_t bias, float* out, int M) {
          H3Args a{};
          const half_t* wh = (const half_t*)(Bl + w_planes);
          a.Ah = v.hi; a.Al = v.lo; a.a_kst = v.kst; a.Wh = wh; a.Wl = wh + n128 * K; a.w_kst = (long)n128 * 32; a.Kp = K;
          a.C = out; a.ldc = H3; a.bias = Bl + bias; a.M = M; a.N = H3;
          return a;
        };
        // longest K first: the blocks of a batched launch are dealt product by product, and the chip finishes a mix of
        // K = 2Hp and K = Hp tiles sooner when the long ones start first (1.5 -> 1.0 long-tile times at 1024 rows)
        H3Args pa[3] = {mk(vr, 2 * Hp, m->rec_r[l].wih_p, m->rec_r[l].bih, w.grr, MT),
                        mk(vr, 2 * Hp, m->rec_f[l].wih_p, m->rec_f[l].bih, w.grf, Mf),
                        mk(vf, Hp, m->fwd[l].wih_p, m->fwd[l].bih, w.gf, MT)};
        H3ArgsBatch sk{};
        H3Batch big{};
        // width-first kernel up to 192 real rows (three 64-row passes over the weights), tiles above: with 64-row tiles (launch_gemm_h3, round 5) the tile
        // kernel is flat at ~35 us up to a round of the chip, the width-first one costs ~12-16 us per pass (222 rows: 48.7 -> 37 us; 150 rows: stays)
        for (H3Args& a : pa) {
          const long real_rows = a.M == MT ? (long)B * T : (long)a.M;
          if (a.M <= m->opt.skinny_max_m && real_rows <= m->opt.l1_skinny_max_rows) {
            // width-first kernel: only the B real rows of every 16-row-padded time slab (B = 1: 16 rows instead of 256)
            if (a.M == MT && Bs != B) { a.M = B * T; a.grp_rows = B; a.grp_stride = (int)Bs; }
            sk.p[sk.n++] = a;
          }
          else if (big.n == 0 || (big.p[0].M == a.M && big.p[0].N == a.N)) big.p[big.n++] = a;
          else {                           // a big product of another shape: its own launch
            H3Batch one{};
            one.p[0] = a; one.n = 1;
            CK(launch_gemm_h3(one, s, m->opt));
          }
        }
        if (big.n) CK(launch_gemm_h3(big, s, m->opt));
        if (sk.n) CK(launch_skinny_gemm_h3_batch(sk, s, m->opt));
      } else {
      CK((hipError_t)proj(inf, Hp, m->fwd[l].wih, m->fwd[l].wih_p, m->fwd[l].wih_s, m->fwd[l].wih_scale, m->fwd[l].bih,
                          w.gf, MT));
      CK((hipError_t)proj(inr, 2 * Hp, m->rec_r[l].wih, m->rec_r[l].wih_p, m->rec_r[l].wih_s, m->rec_r[l].wih_scale,
                          m->rec_r[l].bih, w.grr, MT));
      CK((hipError_t)proj(inr, 2 * Hp, m->rec_f[l].wih, m->rec_f[l].wih_p, m->rec_f[l].wih_s, m->rec_f[l].wih_scale,
                          m->rec_f[l].bih, w.grf, Mf));
      }
      gf = w.gf; grr = w.grr; grf = w.grf;
      ldg = H3;
      if (m->prof) mm->prof_l1_flops += 2.0 * 3.0 * m->H * ((double)B * T * m->H + (double)B * T * 2.0 * m->H + (double)(top ? B : B * T) * 2.0 * m->H);
    }
    // offset of sequence position q (a frame t for layer 0, a time-major slab otherwise)
    auto goff = [&](int q) -> long { return (long)q * Bs * H3; };

    if (m->prof) { int rc = prof_mark(mm, s); if (rc) return rc; }
    const bool use_seq = seq && (top ? plan.seq2 : plan.seq3);
    GruSeqArgs sq{};
    for (int st = 0; st < T; ++st) {
      GruArgs a{};
      a.M = B; a.Hp = Hp; a.first = st == 0;
      int nd = 0;
      {  // gru_fwd layer l, frame t = st
        GruDir& d = a.d[nd++];
        d.Whh = Bl + m->fwd[l].whh; d.bhh = Bl + m->fwd[l].bhh;
        if (l == 0) { gi0(st, 0, d.gi, d.ldgi); d.gi_blk = src.blk; }
        else { d.gi = gf + goff(st); d.ldgi = ldg; d.gi_blk = gblk ? (long)H3 * 16 : 0; }
        if (!top) {
          d.hprev = sf + (long)(st - 1) * Bs * Hp; d.ldh = Hp;
          d.hout = sf + (long)st * Bs * Hp; d.ldo = Hp;
          if (gblk) { d.hprev_b = d.hprev; d.hp_blk = (long)Hp * 16; d.hout_b = d.hout; d.ho_blk = (long)Hp * 16; }
        } else {
          d.hprev = w.pf[(st + 1) & 1]; d.ldh = Hp;
          d.hout = w.pf[st & 1]; d.ldo = Hp;
          // (the last state is read row-major by the tail; every earlier one only by the next step)
          if (gblk) { d.hprev_b = d.hprev; d.hp_blk = (long)Hp * 16; if (st < T - 1) { d.hout_b = d.hout; d.ho_blk = (long)Hp * 16; } }
        }
      }
      {  // gru_rec layer l, reverse direction: flipped index i = T-1-st (frame st for layer 0)
        GruDir& d = a.d[nd++];
        const int i = T - 1 - st;
        d.Whh = Bl + m->rec_r[l].whh; d.bhh = Bl + m->rec_r[l].bhh;
        if (l == 0) { gi0(st, 1, d.gi, d.ldgi); d.gi_blk = src.blk; }
        else { d.gi = grr + goff(i); d.ldgi = ldg; d.gi_blk = gblk ? (long)H3 * 16 : 0; }
        if (!top) {
          d.hprev = sr + (long)(i + 1) * Bs * 2 * Hp + Hp; d.ldh = 2 * Hp;
          d.hout = sr + (long)i * Bs * 2 * Hp + Hp; d.ldo = 2 * Hp;
          if (gblk) {      // second half of the [., 2 Hp] slab: its blocks start Hp * 16 floats into every row tile
            d.hprev_b = sr + (long)(i + 1) * Bs * 2 * Hp + (long)Hp * 16; d.hp_blk = (long)2 * Hp * 16;
            d.hout_b = sr + (long)i * Bs * 2 * Hp + (long)Hp * 16; d.ho_blk = (long)2 * Hp * 16;
          }
        } else {
          d.hprev = w.pr[(st + 1) & 1]; d.ldh = Hp;
          if (st == T - 1) { d.hout = w.ytop + Hp; d.ldo = 2 * Hp; }
          else { d.hout = w.pr[st & 1]; d.ldo = Hp; }
          if (gblk) { d.hprev_b = d.hprev; d.hp_blk = (long)Hp * 16; if (st < T - 1) { d.hout_b = d.hout; d.ho_blk = (long)Hp * 16; } }
        }
      }
      if (!top) {  // gru_rec layer l, forward direction: flipped index i = st (frame T-1-st)
        GruDir& d = a.d[nd++];
        d.Whh = Bl + m->rec_f[l].whh; d.bhh = Bl + m->rec_f[l].bhh;
        if (l == 0) { gi0(T - 1 - st, 2, d.gi, d.ldgi); d.gi_blk = src.blk; }
        else { d.gi = grf + goff(st); d.ldgi = ldg; d.gi_blk = gblk ? (long)H3 * 16 : 0; }
        d.hprev = sr + (long)(st - 1) * Bs * 2 * Hp; d.ldh = 2 * Hp;
        d.hout = sr + (long)st * Bs * 2 * Hp; d.ldo = 2 * Hp;
        if (gblk) { d.hprev_b = d.hprev; d.hp_blk = (long)2 * Hp * 16; d.hout_b = d.hout; d.ho_blk = (long)2 * Hp * 16; }
      }
      a.ndir = nd;
      const size_t wp[3] = {m->fwd[l].whh_p, m->rec_r[l].whh_p, m->rec_f[l].whh_p};
      const DirW* const dw[3] = {&m->fwd[l], &m->rec_r[l], &m->rec_f[l]};
      if (use_seq) {                       // record the step; one launch after the loop
        for (int d = 0; d < nd; ++d) {
          const EncWs::View vo = w.view(a.d[d].hout);
          if (!vo.hi) return (int)hipErrorInvalidValue;
          GruSeqStep& e = sq.st[d][st];
          e.gi = a.d[d].gi; e.ldgi = (int)a.d[d].ldgi; e.hout = a.d[d].hout; e.ldo = (int)a.d[d].ldo;
          e.poff = (unsigned)(vo.hi - w.state_hi); e.pkst = (unsigned)vo.kst;
          if (st == 0) {
            const half_t* wh = (const half_t*)(Bl + wp[d]);
            sq.whi[d] = wh; sq.wlo[d] = wh + n128 * Hp; sq.bhh[d] = a.d[d].bhh;
          }
        }
        if (st == T - 1) {
          sq.w_kst = (long)n128 * 32; sq.phi = w.state_hi; sq.plo = w.state_lo;
          sq.counters = sync_gru(w.sync, l); sq.status = sync_gru_status(m, w.sync);
          sq.fault = m->fault; sq.spin_limit = m->spin_limit; sq.inject = (m->test_fault & 1u) ? 1u : 0u;
          sq.ndir = nd; sq.T = T; sq.M = B; sq.Hp = Hp;
          sq.gran = gran_bytes ? w.gran : nullptr; sq.tag_base = (unsigned)l * 64u;
          sq.rhi = w.tailA.hi; sq.rlo = w.tailA.lo; sq.r_kst = (unsigned)w.tailA.kst;
          sq.r_off[0] = sq.r_off[1] = sq.r_off[2] = sq.x_roff = kNoPlane;
          if (top) {
            // relu(final states) straight into the tail product's A planes: [fwd | rec forward | rec reverse]
            sq.r_off[0] = 0;
            sq.r_off[1] = (unsigned)((size_t)(2 * Hp / 32) * w.tailA.kst);
            // the top layer's forward direction of gru_rec: one cell step from h = 0 (element-wise), same launch
            const EncWs::View vy = w.view(w.ytop);
            if (!vy.hi) return (int)hipErrorInvalidValue;
            sq.x_gi = l == 0 ? src.single : grf; sq.x_ldgi = l == 0 ? (int)src.single_ld : H3;
            sq.x_bhh = Bl + m->rec_f[l].bhh; sq.x_hout = w.ytop; sq.x_ldo = 2 * Hp;
            sq.x_poff = (unsigned)(vy.hi - w.state_hi); sq.x_pkst = (unsigned)vy.kst;
            sq.x_roff = (unsigned)((size_t)(Hp / 32) * w.tailA.kst);
            tail_planes_done = true;
          }
          CK(launch_gru_seq(sq, s, m->opt));
        }
        continue;
      }
      CK((hipError_t)step(a, wp, dw));
    }
    if (top && !use_seq) {  // forward direction of the top bi-GRU layer: one cell step from h = 0
      GruArgs a{};
      a.M = B; a.Hp = Hp; a.first = 1; a.ndir = 1;
      GruDir& d = a.d[0];
      d.Whh = Bl + m->rec_f[l].whh; d.bhh = Bl + m->rec_f[l].bhh;
      if (l == 0) { d.gi = src.single; d.ldgi = src.single_ld; }
      else { d.gi = grf; d.ldgi = H3; d.gi_blk = gblk ? (long)H3 * 16 : 0; }
      d.hprev = w.ytop; d.ldh = 2 * Hp;
      d.hout = w.ytop; d.ldo = 2 * Hp;
      const size_t wp[3] = {m->rec_f[l].whh_p, 0, 0};
      const DirW* const dw[3] = {&m->rec_f[l], nullptr, nullptr};
      CK((hipError_t)step(a, wp, dw));
    }
    if (m->prof) {
      int rc = prof_mark(mm, s);
      if (rc) return rc;
      // consumed cell steps of this layer: fwd T + rec_reverse T + rec forward (T, or 1 on the top layer)
      mm->prof_gru_flops += 2.0 * B * 3.0 * m->H * m->H * (2.0 * T + (top ? 1 : T));
    }
  }
  // ---- y_fwd = linear_fwd(relu(y[-1])), y_rec = linear_rec(relu(y_rec[0])) -------------------
  const float* hlast = w.pf[(T - 1) & 1];
  if (h3) {
    if (!tail_planes_done) {
      CK(launch_split_planes(hlast, Hp, B, Hp, Hp, B, w.tailF.hi, w.tailF.lo, s, 1));
      CK(launch_split_planes(w.ytop, 2 * Hp, B, 2 * Hp, 2 * Hp, B, w.tailR.hi, w.tailR.lo, s, 1));
    }
    if (!is_train && xs_out && m->tail_collapsed) {
      CK((hipError_t)h3_mm(m, w.tailA, Bl + m->mt_p, 256, 3 * Hp, xs_out, kState, Bl + m->kt, B, kState, nullptr, 0, 0.f,
                           nullptr, s));
    } else if (!is_train) {
      // (y_fwd + y_rec) / 2 = ([relu(h_fwd) | relu(y_rec0)] [W_lf | W_lr]^T + b_lf + b_lr) / 2: one product, K = 3Hp
      // (b_lr rides in as an addend row with stride 0)
      CK((hipError_t)h3_mm(m, w.tailA, Bl + m->wlfr_p, kFeat, 3 * Hp, feat, kFeat, Bl + m->blf, B, kFeat, Bl + m->blr, 0,
                           0.5f, feat_planes, s));
    } else {
      CK((hipError_t)h3_mm(m, w.tailF, Bl + m->wlf_p, kFeat, Hp, feat, 2 * kFeat, Bl + m->blf, B, kFeat, nullptr, 0,
                           0.f, nullptr, s));
      CK((hipError_t)h3_mm(m, w.tailR, Bl + m->wlr_p, kFeat, 2 * Hp, feat + kFeat, 2 * kFeat, Bl + m->blr, B, kFeat,
                           nullptr, 0, 0.f, nullptr, s));
    }
  } else if (!is_train) {
    GemmArgs g1 = gemm(hlast, Hp, Bl + m->wlf, Hp, w.y1, kFeat, Bl + m->blf, B, kFeat);
    g1.relu_a = 1;
    CK(launch_gemm(g1, s, m->opt));
    GemmArgs g2 = gemm(w.ytop, 2 * Hp, Bl + m->wlr, 2 * Hp, feat, kFeat, Bl + m->blr, B, kFeat);
    g2.relu_a = 1; g2.addend = w.y1; g2.ldadd = kFeat; g2.scale = 0.5f;
    CK(launch_gemm(g2, s, m->opt));
  } else {
    GemmArgs g1 = gemm(hlast, Hp, Bl + m->wlf, Hp, feat, 2 * kFeat, Bl + m->blf, B, kFeat);
    g1.relu_a = 1;
    CK(launch_gemm(g1, s, m->opt));
    GemmArgs g2 = gemm(w.ytop, 2 * Hp, Bl + m->wlr, 2 * Hp, feat + kFeat, 2 * kFeat, Bl + m->blr, B, kFeat);
    g2.relu_a = 1;
    CK(launch_gemm(g2, s, m->opt));
  }
  return 0;
}

}  // namespace


namespace {
int encoder_fwd_impl(const tepose_model* m, const float* x, int B, int T, int is_train, float* feat,
                     void* workspace, size_t ws_bytes, void* stream, const Planes* feat_planes, bool* wrote_planes,
                     bool zero_sync, float* xs_out);
int regressor_impl(const tepose_model* m, const float* feat, int N, int n_iter, const float* init_pose,
                   const float* init_shape, const float* init_cam, const void* jreg_packed, float* theta, float* verts,
                   float* kp_3d, float* kp_2d, float* rotmat, void* workspace, size_t ws_bytes, void* stream,
                   bool feat_planes_ready, bool sync_zeroed, const float* xs_ready = nullptr);
}  // namespace

int tepose_profile_read_gru(tepose_model* m, double* total_ms, int* n_forwards, double* flops_per_forward) {
  if (!m || !total_ms || !n_forwards || !flops_per_forward) return TEPOSE_E_ARG;
  double tot = 0.0;
  for (size_t i = 0; i + 1 < m->ev_gru_used; i += 2) {
    CK(hipEventSynchronize(m->ev_gru[i + 1]));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, m->ev_gru[i], m->ev_gru[i + 1]));
    tot += ms;
  }
  const int nf = (int)(m->ev_gru_used / (2 * (size_t)m->L));
  *total_ms = tot;
  *n_forwards = nf;
  *flops_per_forward = nf > 0 ? m->prof_gru_flops / nf : 0.0;
  m->ev_gru_used = 0;
  m->prof_gru_flops = 0.0;
  m->prof_l1_flops = 0.0;
  return 0;
}

// the layer >= 1 input projections are what runs between the end of layer l-1's step sequence and the start of layer l's:
// the gaps of the GRU interval list.  Does not reset (tepose_profile_read_gru does): call it first.
int tepose_profile_read_l1proj(tepose_model* m, double* total_ms, int* n_forwards, double* flops_per_forward) {
  if (!m || !total_ms || !n_forwards || !flops_per_forward) return TEPOSE_E_ARG;
  const size_t per = 2 * (size_t)m->L;
  double tot = 0.0;
  for (size_t f = 0; (f + 1) * per <= m->ev_gru_used; ++f)
    for (int l = 1; l < m->L; ++l) {
      CK(hipEventSynchronize(m->ev_gru[f * per + 2 * l]));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, m->ev_gru[f * per + 2 * l - 1], m->ev_gru[f * per + 2 * l]));
      tot += ms;
    }
  const int nf = (int)(m->ev_gru_used / per);
  *total_ms = tot;
  *n_forwards = nf;
  *flops_per_forward = nf > 0 ? m->prof_l1_flops / nf : 0.0;
  return 0;
}

int tepose_encoder_fwd(const tepose_model* m, const float* x, int B, int T, int is_train, float* feat,
                       void* workspace, size_t ws_bytes, void* stream) {
  if (m) { const int rc = forward_begin(m, workspace); if (rc) return rc; }   // an earlier forward on this handle gave up: say so before more work is queued
  return encoder_fwd_impl(m, x, B, T, is_train, feat, workspace, ws_bytes, stream, nullptr, nullptr, false, nullptr);
}

namespace {
// feat_planes: also leave the feature as hi / lo planes there (the regressor's first A operand), when that region does
// not overlap a buffer the tail product still reads
int encoder_fwd_impl(const tepose_model* m, const float* x, int B, int T, int is_train, float* feat,
                     void* workspace, size_t ws_bytes, void* stream, const Planes* feat_planes, bool* wrote_planes,
                     bool zero_sync, float* xs_out) {
  if (wrote_planes) *wrote_planes = false;
  if (!m || m->kind != 0 || !x || !feat || !workspace || B < 1 || T < 1) return TEPOSE_E_ARG;
  if (!m->enc_packed) return TEPOSE_E_STATE;
  if ((size_t)B * T > (1u << 30) / 4) return TEPOSE_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  Carver c(workspace, ws_bytes);
  EncWs w;
  carve_encoder(m, B, T, c, w);
  if (c.cur > ws_bytes) return TEPOSE_E_WORKSPACE;
  const int L = m->L, Hp = m->Hp;
  const float* Bl = m->blob;
  const long BT = (long)B * T;
  const int H3 = 3 * Hp;

  // ---- layer-0 input projections: one GEMM for every direction that runs all T steps --------
  const int ld0 = (L >= 2 ? 9 : 6) * Hp;
  const KernelPlan plan = select_kernels(m, B, T);     // every batch-class decision (see select_kernels)
  const bool h3 = plan.h3;
  half_t* xh = (half_t*)w.xp;                       // hi / lo planes share the padded-input buffer
  half_t* xl = xh + (size_t)BT * kInputP;
  const size_t rows0 = (size_t)round_up(9 * Hp, 128);
  const half_t* w0h = (const half_t*)(Bl + m->wih0_p);
  const half_t* w0l = w0h + rows0 * kInputP;
  // g0big: large batches of an L >= 2 model on the barrier-free scaled-plane kernel (its input planes carry scale 1: same fp16 range as the other
  // layout; elements below 2^-3 keep an absolute error <= 2^-25 instead of a relative one).  g0mid: mid-size batches (cfg-B: 64 windows x 16 frames =
  // 1024 rows) on 128 x 288 tiles, which cut the 9 Hp columns into whole rounds of the chip.  g0blk: gate pre-activations FRAME-major
  // (plane row t * B + b: a GRU step then reads B consecutive rows) and 16 x 16-blocked (common.h gi_blk_offset).
  const bool g0big = plan.g0big, g0mid = plan.g0mid, g0blk = plan.g0blk;
  const bool g0s = g0big || g0mid;
  // the caller's windows -> planes with one power-of-two scale per row (any finite fp32 magnitude; DESIGN 4b "range")
  // (with zero_sync the kernel also clears the forward's arrival counters / granules: it is the forward's first kernel)
  // (the forward's first kernel also clears its sync region -- arrival counters, granules, STATUS words -- so that a give-up of the
  // layer-0 projection (barrier-free kernel, gemm_h3s16c.hip) is not wiped by a clearing that comes after it)
  (void)zero_sync;
  if (h3) CK(launch_split_rows(x, kInput, BT, kInput, kInputP, BT, g0s ? 1 : 0, xh, xl, w.rs, s, m->opt, (void*)w.sync, w.sync ? sync_zero_bytes(m, B) : 0,
                               g0blk ? T : 0));
  else CK(launch_pad_input(x, w.xp, BT, s));
  if (!h3 && w.sync) CK(hipMemsetAsync(w.sync, 0, sync_zero_bytes(m, B), s));
  {
    tepose_model* mm = const_cast<tepose_model*>(m);
    if (m->prof) {
      if (mm->ev.size() < mm->ev_used + 2) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        mm->ev.push_back(a);
        mm->ev.push_back(b);
      }
      CK(hipEventRecord(mm->ev[mm->ev_used], s));
    }
    if (h3 && g0s) {          // 256 x 256 tiles, one accumulator per tile, scaled planes (gemm_h3s.hip)
      const size_t rows256 = (size_t)round_up(9 * Hp, 256);
      const half_t* sh = (const half_t*)(Bl + m->wih0_s);
      H3SArgs a{xh, xl, BT * 16, sh, sh + rows256 * kInputP, (long)rows256 * 16, kInputP, w.g0, (long)ld0,
                Bl + m->bih0, 1.f / m->w0_scale, (int)BT, ld0, w.rs};
      if (w.sync) a.status = sync_gru_status(m, w.sync);
      a.fault = m->fault;
      a.inject = (m->test_fault >> 2) & 1u;
      a.c_blk_hp = g0blk ? Hp : 0;
      // (the barrier-free 256 x 256 kernel loses on mid-size batches: 1024 rows are 144 of its tiles -- 0.138 against 0.119 ms, profiles/r05_mid_rows_gemm.txt)
      if (g0mid) CK(launch_gemm_h3s_mid(a, s));
      else CK(launch_gemm_h3s(a, s, m->opt, 0));
    } else if (h3) {
      H3Batch b{};
      b.p[0] = H3Args{xh, xl, BT * 32, w0h, w0l, (long)rows0 * 32, kInputP, w.g0, (long)ld0, Bl + m->bih0, (int)BT,
                      ld0};
      b.p[0].row_scale = w.rs;
      b.n = 1;
      // few rows (live stream, a handful of clips): the width-first kernel streams the 79 MB of W_ih planes with
      // N / 48 = 192 workgroups instead of 72 tiles of 128 rows
      if (plan.g0skinny) CK(launch_skinny_gemm_h3(b.p[0], s, m->opt));
      else CK(launch_gemm_h3(b, s, m->opt));
    } else {
      GemmArgs g = gemm(w.xp, kInputP, Bl + m->wih0, kInputP, w.g0, ld0, Bl + m->bih0, (int)BT, ld0);
      CK(launch_gemm(g, s, m->opt));
    }
    if (m->prof) {
      CK(hipEventRecord(mm->ev[mm->ev_used + 1], s));
      mm->ev_used += 2;
      mm->prof_flops = 2.0 * (double)BT * (double)(L >= 2 ? 9 : 6) * m->H * kInput;
    }
  }
  if (L == 1) {  // rec.l0 forward direction: only flipped index 0 (= frame T-1) is consumed
    if (h3) {
      H3Batch b{};
      // frames T-1 of every window as compact planes; W rows 6Hp.. of the stacked layer-0 block
      CK(launch_split_rows(x + (long)(T - 1) * kInput, (long)T * kInput, B, kInput, kInputP, B, 0, w.x0h, w.x0l, w.rs0, s, m->opt));
      b.p[0] = H3Args{w.x0h, w.x0l, (long)B * 32, w0h + (size_t)6 * Hp * 32, w0l + (size_t)6 * Hp * 32,
                      (long)rows0 * 32, kInputP, w.g0c, (long)H3, Bl + m->bih0 + 6 * Hp, B, H3};
      b.p[0].row_scale = w.rs0;
      b.n = 1;
      CK(launch_gemm_h3(b, s, m->opt));
    } else {
      GemmArgs g = gemm(w.xp + (long)(T - 1) * kInputP, (long)T * kInputP, Bl + m->wih0 + (size_t)6 * Hp * kInputP,
                        kInputP, w.g0c, H3, Bl + m->bih0 + 6 * Hp, B, H3);
      CK(launch_gemm(g, s, m->opt));
    }
  }

  G0Src src{w.g0, ld0, (long)T * ld0, 0, 0, nullptr, 0, w.g0c, H3};
  if (g0blk) { src.frame_stride = (long)B * ld0; src.row_stride = ld0; src.blk = (long)ld0 * 16; }
  if (feat_planes) {
    // live at tail time: the tail product's A planes and the fp32 final states; everything carved before them is dead
    const char* end = (const char*)(feat_planes->lo + (size_t)B * kFeat + 128);
    const char* first_live = (const char*)(L >= 2 ? w.gf : w.pf[0]);
    if (!h3 || is_train || end > first_live) feat_planes = nullptr;
  }
  if (wrote_planes) *wrote_planes = feat_planes != nullptr;
  return encoder_core(m, src, B, T, is_train, feat, w, s, feat_planes, true, xs_out);     // cleared above
}
}  // namespace

size_t tepose_project_frames_workspace_bytes(const tepose_model* m, int B) {
  if (!m || B < 1) return 0;
  const size_t xbytes = align_up((size_t)B * kInputP * sizeof(float), 256);
  // padded fp32 rows, plus their hi / lo planes and per-row scales when the product runs on the split-precision kernel
  return (m->split && B > m->opt.split_min_m) ? 2 * xbytes + 512 + align_up((size_t)B * sizeof(float), 256) : xbytes;
}

int tepose_project_frames(const tepose_model* m, const float* feat, long feat_ld, const float* theta, long theta_ld,
                          int B, float* out, long out_ld, void* workspace, size_t ws_bytes, void* stream) {
  if (!m || m->kind != 0 || !feat || !out || !workspace || B < 1) return TEPOSE_E_ARG;
  if (!m->enc_packed) return TEPOSE_E_STATE;
  if (ws_bytes < tepose_project_frames_workspace_bytes(m, B)) return TEPOSE_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* xp = (float*)workspace;
  CK(launch_pad_rows(feat, feat_ld, theta, theta_ld, xp, B, s));
  const size_t xbytes = align_up((size_t)B * kInputP * sizeof(float), 256);
  if (m->split && B > m->opt.split_min_m) {   // split-precision product, same numerics as tepose_forward's
    Planes P;
    P.hi = (half_t*)((char*)workspace + xbytes);
    P.lo = (half_t*)((char*)workspace + xbytes + xbytes / 2);
    P.kst = (long)B * 32;
    float* rs = (float*)((char*)workspace + 2 * xbytes + 512);
    CK(launch_split_rows(xp, kInputP, B, kInputP, kInputP, B, 0, P.hi, P.lo, rs, s, m->opt));
    CK((hipError_t)h3_mm(m, P, m->blob + m->wih0_p, round_up(9 * m->Hp, 128), kInputP, out, out_ld, m->blob + m->bih0, B,
                         9 * m->Hp, nullptr, 0, 0.f, nullptr, s, rs));
    return 0;
  }
  GemmArgs g = gemm(xp, kInputP, m->blob + m->wih0, kInputP, out, out_ld, m->blob + m->bih0, B, 9 * m->Hp);
  CK(launch_gemm(g, s, m->opt));
  return 0;
}

// Both projections of a window step of the clip driver as ONE product of 2 B rows (rows [0, B): the previous newest frame with its now-known theta ->
// its ring slot; rows [B, 2 B): the newest frame with zero theta -> the `newest` rows): the 79 MB of layer-0 W_ih planes are streamed once per
// step instead of twice, one input split (which gathers the rows itself) instead of two pads and two splits.  Same GEMM rows on the same operands as
// two tepose_project_frames calls; the width-first kernel may split K over 4 or 8 waves depending on the row count, so results agree to rounding
// (bit for bit at the published width).
namespace {
int project_frame_pair_impl(const tepose_model* m, const float* feat_prev, const float* feat_new, long feat_ld, const float* theta_prev,
                            long theta_ld, int B, float* out_prev, long out_prev_ld, float* out_new, long out_new_ld, void* workspace,
                            size_t ws_bytes, void* stream, void* zero, size_t zero_bytes, bool* zeroed);
}

int tepose_project_frame_pair(const tepose_model* m, const float* feat_prev, const float* feat_new, long feat_ld, const float* theta_prev,
                              long theta_ld, int B, float* out_prev, long out_prev_ld, float* out_new, long out_new_ld, void* workspace,
                              size_t ws_bytes, void* stream) {
  return project_frame_pair_impl(m, feat_prev, feat_new, feat_ld, theta_prev, theta_ld, B, out_prev, out_prev_ld, out_new, out_new_ld, workspace, ws_bytes,
                                 stream, nullptr, 0, nullptr);
}

namespace {
// `zero` / `zero_bytes`: a region the input-split kernel clears on its way (the following forward's sync region: tepose_window_step); *zeroed says
// whether it did (the two-call fallbacks do not)
int project_frame_pair_impl(const tepose_model* m, const float* feat_prev, const float* feat_new, long feat_ld, const float* theta_prev,
                            long theta_ld, int B, float* out_prev, long out_prev_ld, float* out_new, long out_new_ld, void* workspace,
                            size_t ws_bytes, void* stream, void* zero, size_t zero_bytes, bool* zeroed) {
  if (zeroed) *zeroed = false;
  if (!m || m->kind != 0 || !feat_prev || !feat_new || !theta_prev || !out_prev || !out_new || !workspace || B < 1) return TEPOSE_E_ARG;
  if (!m->enc_packed) return TEPOSE_E_STATE;
  if (ws_bytes < tepose_project_frames_workspace_bytes(m, 2 * B)) return TEPOSE_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int M = 2 * B;
  const bool h3 = m->split && B > m->opt.split_min_m;       // (B, not 2 B: the same arithmetic class as tepose_project_frames at this B)
  if (!h3 || M > m->opt.skinny_max_m) {            // exact-fp32 products / more rows than the width-first kernel takes: the two products one after the other
    int rc = tepose_project_frames(m, feat_prev, feat_ld, theta_prev, theta_ld, B, out_prev, out_prev_ld, workspace, ws_bytes, stream);
    if (rc) return rc;
    return tepose_project_frames(m, feat_new, feat_ld, nullptr, 0, B, out_new, out_new_ld, workspace, ws_bytes, stream);
  }
  const size_t xbytes = align_up((size_t)M * kInputP * sizeof(float), 256);
  half_t* hi = (half_t*)((char*)workspace + xbytes);
  half_t* lo = (half_t*)((char*)workspace + xbytes + xbytes / 2);
  float* rs = (float*)((char*)workspace + 2 * xbytes + 512);
  // the split kernel gathers the 2 B rows itself (features | theta, features | zeros): no padded fp32 copy, one launch instead of three
  const RowPairSrc pr{feat_prev, theta_prev, feat_new, feat_ld, theta_ld, B};
  const bool z = zero && zero_bytes && zero_bytes % 16 == 0;
  CK(launch_split_rows(nullptr, 0, M, kInput, kInputP, M, 0, hi, lo, rs, s, m->opt, z ? zero : nullptr, z ? zero_bytes : 0, 0, &pr));
  if (zeroed) *zeroed = z;
  const int Np = round_up(9 * m->Hp, 128);
  const half_t* wh = (const half_t*)(m->blob + m->wih0_p);
  H3Args p{};
  p.Ah = hi; p.Al = lo; p.a_kst = (long)M * 32;
  p.Wh = wh; p.Wl = wh + (size_t)Np * kInputP; p.w_kst = (long)Np * 32; p.Kp = kInputP;
  p.C = out_prev; p.ldc = out_prev_ld; p.bias = m->blob + m->bih0; p.M = M; p.N = 9 * m->Hp;
  p.row_scale = rs;
  p.C2 = out_new; p.ldc2 = out_new_ld; p.c_split = B;
  CK(launch_skinny_gemm_h3(p, s, m->opt));
  return 0;
}

int forward_cached_impl(const tepose_model* m, const float* ring_base, int ring, int first_slot, long clip_stride, const float* newest, long newest_ld, int B,
                        int T, const void* jreg_packed, float* theta, float* verts, float* kp_3d, float* kp_2d, float* rotmat, void* workspace,
                        size_t ws_bytes, void* stream, bool sync_zeroed);
}  // namespace

// One iteration of the reference's window loop (evaluate.py:247-269, demo.py:238-252) for B clips in lock-step, as ONE call: both layer-0 projections of
// the step (tepose_project_frame_pair: the previous newest frame with its now-known theta -> its ring slot `out_prev`, the newest frame with zero theta ->
// `newest`) and then TePose.forward of the window from the cached projections (tepose_forward_cached).  Same results as the two calls; the forward's
// sync region is cleared by the projection's input-split kernel instead of a memset node of its own, and a host loop makes one call per step.
int tepose_window_step(const tepose_model* m, const float* feat_prev, const float* feat_new, long feat_ld, const float* theta_prev, long theta_ld,
                       float* out_prev, long out_prev_ld, float* newest, long newest_ld, const float* ring_base, int ring, int first_slot,
                       long clip_stride, int B, int T, const void* jreg_packed, float* theta, float* verts, float* kp_3d, float* kp_2d, float* rotmat,
                       void* workspace, size_t ws_bytes, void* pair_workspace, size_t pair_ws_bytes, void* stream) {
  if (!m || m->kind != 0 || !workspace || B < 1 || T < 1) return TEPOSE_E_ARG;
  if (!m->enc_packed) return TEPOSE_E_STATE;
  { const int rc = forward_begin(m, workspace); if (rc) return rc; }
  if (ws_bytes < tepose_workspace_bytes(m, B, T)) return TEPOSE_E_WORKSPACE;
  // the forward's sync region: the first carve of its workspace (as tepose_forward_cached lays it out)
  void* zero = nullptr;
  size_t zero_bytes = 0;
  {
    const size_t feat_bytes = align_up((size_t)B * 2 * kFeat * sizeof(float), 256);
    const size_t rest_bytes = (ws_bytes & ~(size_t)255) - feat_bytes;
    Carver c(workspace, rest_bytes);
    EncWs w;
    carve_encoder(m, B, T, c, w);
    if (c.cur > rest_bytes) return TEPOSE_E_WORKSPACE;
    if (w.sync) { zero = (void*)w.sync; zero_bytes = sync_zero_bytes(m, B); }
  }
  bool zeroed = false;
  int rc = project_frame_pair_impl(m, feat_prev, feat_new, feat_ld, theta_prev, theta_ld, B, out_prev, out_prev_ld, newest, newest_ld, pair_workspace,
                                   pair_ws_bytes, stream, zero, zero_bytes, &zeroed);
  if (rc) return rc;
  return forward_cached_impl(m, ring_base, ring, first_slot, clip_stride, newest, newest_ld, B, T, jreg_packed, theta, verts, kp_3d, kp_2d, rotmat,
                             workspace, ws_bytes, stream, zeroed);
}

int tepose_forward_cached(const tepose_model* m, const float* ring_base, int ring, int first_slot, long clip_stride,
                          const float* newest, long newest_ld, int B, int T, const void* jreg_packed, float* theta,
                          float* verts, float* kp_3d, float* kp_2d, float* rotmat, void* workspace, size_t ws_bytes,
                          void* stream) {
  return forward_cached_impl(m, ring_base, ring, first_slot, clip_stride, newest, newest_ld, B, T, jreg_packed, theta, verts, kp_3d, kp_2d, rotmat,
                             workspace, ws_bytes, stream, false);
}

namespace {
int forward_cached_impl(const tepose_model* m, const float* ring_base, int ring, int first_slot, long clip_stride, const float* newest, long newest_ld, int B,
                        int T, const void* jreg_packed, float* theta, float* verts, float* kp_3d, float* kp_2d, float* rotmat, void* workspace,
                        size_t ws_bytes, void* stream, bool sync_zeroed) {
  if (!m || m->kind != 0 || !ring_base || !newest || !workspace || B < 1 || T < 1 || ring < T - 1 || ring < 1 ||
      first_slot < 0 || first_slot >= ring)
    return TEPOSE_E_ARG;
  if (!m->enc_packed) return TEPOSE_E_STATE;
  { const int rc = forward_begin(m, workspace); if (rc) return rc; }
  if (ws_bytes < tepose_workspace_bytes(m, B, T)) return TEPOSE_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // [shared scratch | feature]: the scratch comes FIRST, so that its first carve -- the sync region with the forward's status
  // words -- sits at the workspace base for every entry point (tepose_forward_status reads it there)
  const size_t feat_bytes = align_up((size_t)B * 2 * kFeat * sizeof(float), 256);
  char* rest = (char*)workspace;
  const size_t rest_bytes = (ws_bytes & ~(size_t)255) - feat_bytes;
  float* feat = (float*)(rest + rest_bytes);
  Carver c(rest, rest_bytes);
  EncWs w;
  carve_encoder(m, B, T, c, w);
  if (c.cur > rest_bytes) return TEPOSE_E_WORKSPACE;
  const int ld0 = 9 * m->Hp;
  G0Src src{ring_base, ld0, clip_stride, first_slot, ring, newest, newest_ld, newest + 6 * m->Hp, newest_ld};
  const bool col = m->tail_collapsed && m->split && B > m->opt.split_min_m;
  int rc = encoder_core(m, src, B, T, 0, feat, w, s, nullptr, sync_zeroed, col ? feat : nullptr);
  if (rc) return rc;
  return regressor_impl(m, feat, B, 3, nullptr, nullptr, nullptr, jreg_packed, theta, verts, kp_3d, kp_2d, rotmat, rest,
                        rest_bytes, stream, false, true, col ? feat : nullptr);    // (encoder_core cleared the shared sync region)
}
}  // namespace

int tepose_regressor_fwd(const tepose_model* m, const float* feat, int N, int n_iter, const void* jreg_packed,
                         float* theta, float* verts, float* kp_3d, float* kp_2d, float* rotmat,
                         void* workspace, size_t ws_bytes, void* stream) {
  return tepose_regressor_fwd_init(m, feat, N, n_iter, nullptr, nullptr, nullptr, jreg_packed, theta, verts, kp_3d, kp_2d,
                                   rotmat, workspace, ws_bytes, stream);
}

int tepose_regressor_fwd_init(const tepose_model* m, const float* feat, int N, int n_iter, const float* init_pose,
                              const float* init_shape, const float* init_cam, const void* jreg_packed, float* theta,
                              float* verts, float* kp_3d, float* kp_2d, float* rotmat, void* workspace, size_t ws_bytes,
                              void* stream) {
  if (m) { const int rc = forward_begin(m, workspace); if (rc) return rc; }
  return regressor_impl(m, feat, N, n_iter, init_pose, init_shape, init_cam, jreg_packed, theta, verts, kp_3d, kp_2d,
                        rotmat, workspace, ws_bytes, stream, false, false);
}

namespace {
int regressor_impl(const tepose_model* m, const float* feat, int N, int n_iter, const float* init_pose,
                   const float* init_shape, const float* init_cam, const void* jreg_packed, float* theta, float* verts,
                   float* kp_3d, float* kp_2d, float* rotmat, void* workspace, size_t ws_bytes, void* stream,
                   bool feat_planes_ready, bool sync_zeroed, const float* xs_ready) {
  if (!m || !feat || !theta || !verts || !kp_3d || !kp_2d || !rotmat || !workspace || N < 1 || n_iter < 0)
    return TEPOSE_E_ARG;
  if (!m->reg_packed || !m->smpl_packed) return TEPOSE_E_STATE;
  hipStream_t s = (hipStream_t)stream;
  Carver c(workspace, ws_bytes);
  RegWs w;
  carve_regressor(m, N, c, w);
  if (c.cur > ws_bytes) return TEPOSE_E_WORKSPACE;
  const float* Bl = m->blob;
  // a stand-alone regressor call clears its sync region (counters + the status words tepose_forward_status reads); inside
  // tepose_forward / tepose_forward_cached the encoder part has done it (sync_zeroed) and may have left a give-up there
  if (!sync_zeroed && w.sync) CK(hipMemsetAsync(w.sync, 0, sync_words(m) * sizeof(unsigned), s));
  // xc = cat[x, pose, shape, cam]; fc1(xc) = x W1a^T + b1 (iteration-invariant) + state W1b^T
  if (xs_ready) {
    // the encoder's last product already produced the final state rows (collapsed regressor + tail)
    w.xs = const_cast<float*>(xs_ready);
  } else if (m->reg_collapsed && w.split_fc && n_iter == 3 && !init_pose && !init_shape && !init_cam) {
    // the three iterations from the model's own initial state as ONE product: xs = feat Mf^T + k0
    if (!feat_planes_ready) CK(launch_split_planes(feat, kFeat, N, kFeat, kFeat, N, w.featP.hi, w.featP.lo, s));
    CK((hipError_t)h3_mm(m, w.featP, Bl + m->mf_p, 256, kFeat, w.xs, kState, Bl + m->k0, N, kState, nullptr, 0, 0.f, nullptr, s));
  } else if (select_kernels(m, N, 1).reg_seq) {
    // small batches: the whole FC loop in one persistent launch (reg_seq.hip)
    if (!feat_planes_ready) CK(launch_split_planes(feat, kFeat, N, kFeat, kFeat, N, w.featP.hi, w.featP.lo, s));
    RegSeqArgs ra{};
    ra.fh = w.featP.hi; ra.fl = w.featP.lo; ra.f_kst = w.featP.kst;
    const half_t* p;
    p = (const half_t*)(Bl + m->w1a_p); ra.w1a_h = p; ra.w1a_l = p + (size_t)1024 * kFeat;
    p = (const half_t*)(Bl + m->w1b_p); ra.w1b_h = p; ra.w1b_l = p + (size_t)1024 * kState;
    p = (const half_t*)(Bl + m->w2_p); ra.w2_h = p; ra.w2_l = p + (size_t)1024 * 1024;
    p = (const half_t*)(Bl + m->wdec_p); ra.wd_h = p; ra.wd_l = p + (size_t)256 * 1024;
    ra.b1 = Bl + m->b1; ra.b2 = Bl + m->b2; ra.bdec = Bl + m->bdec;
    ra.init160 = Bl + m->init; ra.ipose = init_pose; ra.ishape = init_shape; ra.icam = init_cam;
    ra.h1h = w.h1P.hi; ra.h1l = w.h1P.lo; ra.h2h = w.h2P.hi; ra.h2l = w.h2P.lo; ra.h_kst = w.h1P.kst;
    ra.xh = w.xsP.hi; ra.xl = w.xsP.lo; ra.x_kst = w.xsP.kst;
    ra.xs = w.xs; ra.counters = sync_reg(m, w.sync); ra.status = sync_reg_status(m, w.sync);
    ra.fault = m->fault; ra.spin_limit = m->spin_limit; ra.inject = (m->test_fault & 2u) ? 1u : 0u;
    ra.N = N; ra.n_iter = n_iter;
    CK(launch_reg_seq(ra, s));
  } else if (w.split_fc) {
    if (!feat_planes_ready) CK(launch_split_planes(feat, kFeat, N, kFeat, kFeat, N, w.featP.hi, w.featP.lo, s));
    CK((hipError_t)h3_mm(m, w.featP, Bl + m->w1a_p, 1024, kFeat, w.base, 1024, Bl + m->b1, N, 1024, nullptr, 0, 0.f,
                         nullptr, s));
    CK(init_state(Bl + m->init, init_pose, init_shape, init_cam, w.xs, N, s));
    CK(launch_split_planes(w.xs, kState, N, kState, kState, N, w.xsP.hi, w.xsP.lo, s));
    for (int it = 0; it < n_iter; ++it) {
      CK((hipError_t)h3_mm(m, w.xsP, Bl + m->w1b_p, 1024, kState, w.h1, 1024, nullptr, N, 1024, w.base, 1024, 0.f,
                           &w.h1P, s));
      CK((hipError_t)h3_mm(m, w.h1P, Bl + m->w2_p, 1024, 1024, w.h2, 1024, Bl + m->b2, N, 1024, nullptr, 0, 0.f,
                           &w.h2P, s));
      CK((hipError_t)h3_mm(m, w.h2P, Bl + m->wdec_p, 256, 1024, w.xs, kState, Bl + m->bdec, N, kState, w.xs, kState,
                           0.f, &w.xsP, s));
    }
  } else {
    GemmArgs gb = gemm(feat, kFeat, Bl + m->w1a, kFeat, w.base, 1024, Bl + m->b1, N, 1024);
    CK(launch_gemm(gb, s, m->opt));
    CK(init_state(Bl + m->init, init_pose, init_shape, init_cam, w.xs, N, s));
    for (int it = 0; it < n_iter; ++it) {
      GemmArgs g1 = gemm(w.xs, kState, Bl + m->w1b, kState, w.h1, 1024, nullptr, N, 1024);
      g1.addend = w.base; g1.ldadd = 1024;
      CK(launch_gemm(g1, s, m->opt));
      GemmArgs g2 = gemm(w.h1, 1024, Bl + m->w2, 1024, w.h2, 1024, Bl + m->b2, N, 1024);
      CK(launch_gemm(g2, s, m->opt));
      GemmArgs g3 = gemm(w.h2, 1024, Bl + m->wdec, 1024, w.xs, kState, Bl + m->bdec, N, kState);
      g3.addend = w.xs; g3.ldadd = kState;
      CK(launch_gemm(g3, s, m->opt));
    }
  }
  SmplConsts sc{};
  sc.J0 = Bl + m->smpl.J0; sc.JS = Bl + m->smpl.JS; sc.blendW = Bl + m->smpl.blendW;
  sc.lbsW = Bl + m->smpl.lbsW; sc.lbs_cidx = (const int*)(Bl + m->smpl.lbs_cidx); sc.lbs_cval = Bl + m->smpl.lbs_cval;
  sc.lbs_sparse = m->lbs_sparse; sc.parents = (const int*)(Bl + m->smpl.parents);
  sc.depth = (const int*)(Bl + m->smpl.depth); sc.maxdepth = m->maxdepth;
  sc.xr_ptr = (const int*)(Bl + m->smpl.xr_ptr); sc.xr_idx = (const int*)(Bl + m->smpl.xr_idx);
  sc.xr_val = Bl + m->smpl.xr_val;
  if (smpl_small_ok(sc, N, m->opt)) {     // a window or a few: prep + blend shapes + skinning as one launch (smpl.hip)
    CK(launch_smpl_small(sc, 0, w.xs, kState, w.xs + kNPose, kState, w.xs + 154, kState, N, w.amat, w.posed, rotmat, theta,
                         verts, s));
  } else {
    CK(launch_smpl_prep(sc, w.xs, N, w.pf, w.amat, w.posed, rotmat, theta, s, w.split ? w.pfP.hi : nullptr,
                        w.split ? w.pfP.lo : nullptr, w.pfP.kst));
    CK((hipError_t)blend_shapes(m, w, N, s));
    CK(launch_smpl_skin(sc, w.vposed, w.amat, N, verts, s));
  }
  JregPacked jr{};
  if (jreg_packed) {
    const int* p = (const int*)jreg_packed;
    jr.ptr = p; jr.idx = p + 32; jr.val = (const float*)(p + 32 + 17 * kNV);
  }
  CK(launch_smpl_joints(sc, jreg_packed ? &jr : nullptr, verts, w.posed, w.xs, N, kp_3d, kp_2d, s));
  return 0;
}
}  // namespace

int tepose_forward(const tepose_model* m, const float* x, int B, int T, const void* jreg_packed, float* theta,
                   float* verts, float* kp_3d, float* kp_2d, float* rotmat, void* workspace, size_t ws_bytes,
                   void* stream) {
  if (!m || !workspace || B < 1 || T < 1) return TEPOSE_E_ARG;
  { const int rc = forward_begin(m, workspace); if (rc) return rc; }
  if (ws_bytes < tepose_workspace_bytes(m, B, T)) return TEPOSE_E_WORKSPACE;
  // [shared scratch | feature]: the encoder's scratch is dead once `feat` exists; the scratch comes FIRST, so that its first
  // carve -- the sync region with the forward's status words -- sits at the workspace base for every entry point
  // (tepose_forward_status reads it there)
  const size_t feat_bytes = align_up((size_t)B * 2 * kFeat * sizeof(float), 256);
  char* rest = (char*)workspace;
  const size_t rest_bytes = (ws_bytes & ~(size_t)255) - feat_bytes;
  float* feat = (float*)(rest + rest_bytes);
  // the regressor's first A operand (planes of the feature) is written by the encoder's tail product
  RegWs rw;
  {
    Carver c(rest, rest_bytes);
    carve_regressor(m, B, c, rw);
  }
  if (!rw.sync) return TEPOSE_E_WORKSPACE;
  // every arrival counter (and, for B <= 4, every granule) of this forward is cleared by its first kernel (the input
  // split), or by one memset node where that kernel does not run
  bool wrote = false;
  if (m->tail_collapsed && rw.split_fc) {
    // the tail linears and the regressor's three iterations are one product on the relu(final states): the
    // state rows land in the (otherwise unused) feature buffer
    int rc = encoder_fwd_impl(m, x, B, T, 0, feat, rest, rest_bytes, stream, nullptr, &wrote, true, feat);
    if (rc) return rc;
    return regressor_impl(m, feat, B, 3, nullptr, nullptr, nullptr, jreg_packed, theta, verts, kp_3d, kp_2d, rotmat, rest,
                          rest_bytes, stream, false, true, feat);
  }
  int rc = encoder_fwd_impl(m, x, B, T, 0, feat, rest, rest_bytes, stream, rw.split_fc ? &rw.featP : nullptr, &wrote, true, nullptr);
  if (rc) return rc;
  return regressor_impl(m, feat, B, 3, nullptr, nullptr, nullptr, jreg_packed, theta, verts, kp_3d, kp_2d, rotmat, rest,
                        rest_bytes, stream, wrote, true);
}

int tepose_metrics_joints(const float* pred, const float* target, int N, int J, int pelvis_mode, float* mpjpe,
                          float* pa_mpjpe, float* accel, void* stream) {
  if (!pred || !target || !mpjpe || !pa_mpjpe || !accel || N < 1) return TEPOSE_E_ARG;
  if (J < 4 || J > 17 || (pelvis_mode != 0 && pelvis_mode != 1)) return TEPOSE_E_SHAPE;
  CK(launch_metrics_joints(pred, target, N, J, pelvis_mode, mpjpe, pa_mpjpe, accel, (hipStream_t)stream));
  return 0;
}

int tepose_metrics_verts(const float* pred_verts, const float* target_verts, int N, float* mpvpe, void* stream) {
  if (!pred_verts || !target_verts || !mpvpe || N < 1) return TEPOSE_E_ARG;
  CK(launch_metrics_verts(pred_verts, target_verts, N, mpvpe, (hipStream_t)stream));
  return 0;
}

namespace {
SmplConsts smpl_consts(const tepose_model* m) {
  const float* Bl = m->blob;
  SmplConsts sc{};
  sc.J0 = Bl + m->smpl.J0; sc.JS = Bl + m->smpl.JS; sc.blendW = Bl + m->smpl.blendW;
  sc.lbsW = Bl + m->smpl.lbsW; sc.lbs_cidx = (const int*)(Bl + m->smpl.lbs_cidx); sc.lbs_cval = Bl + m->smpl.lbs_cval;
  sc.lbs_sparse = m->lbs_sparse; sc.parents = (const int*)(Bl + m->smpl.parents);
  sc.depth = (const int*)(Bl + m->smpl.depth); sc.maxdepth = m->maxdepth;
  sc.xr_ptr = (const int*)(Bl + m->smpl.xr_ptr); sc.xr_idx = (const int*)(Bl + m->smpl.xr_idx);
  sc.xr_val = Bl + m->smpl.xr_val;
  return sc;
}
}  // namespace

int tepose_smpl_fwd(const tepose_model* m, int pose2rot, const float* pose, const float* betas, int N, float* verts,
                    float* joints49, void* workspace, size_t ws_bytes, void* stream) {
  if (!m || !pose || !betas || !verts || !workspace || N < 1) return TEPOSE_E_ARG;
  if (!m->smpl_packed) return TEPOSE_E_STATE;
  hipStream_t s = (hipStream_t)stream;
  Carver c(workspace, ws_bytes);
  RegWs w;
  carve_regressor(m, N, c, w);
  if (c.cur > ws_bytes) return TEPOSE_E_WORKSPACE;
  SmplConsts sc = smpl_consts(m);
  if (smpl_small_ok(sc, N, m->opt)) {
    CK(launch_smpl_small(sc, pose2rot ? 1 : 2, pose, pose2rot ? 72 : 216, betas, 10, nullptr, 0, N, w.amat, w.posed, nullptr,
                         nullptr, verts, s));
  } else {
    CK(launch_smpl_prep_pose(sc, pose2rot ? 1 : 2, pose, pose2rot ? 72 : 216, betas, 10, N, w.pf, w.amat, w.posed, s,
                             w.split ? w.pfP.hi : nullptr, w.split ? w.pfP.lo : nullptr, w.pfP.kst));
    CK((hipError_t)blend_shapes(m, w, N, s));
    CK(launch_smpl_skin(sc, w.vposed, w.amat, N, verts, s));
  }
  if (joints49) CK(launch_smpl_joints(sc, nullptr, verts, w.posed, nullptr, N, joints49, nullptr, s));
  return 0;
}

// evaluate.py:289-291 (the --filter branch): the H36M regressor applied to given vertices, 14 LSP joints per person
int tepose_joints_from_verts(const tepose_model* m, const void* jreg_packed, const float* verts, int N, float* kp_3d, void* stream) {
  if (!m || !jreg_packed || !verts || !kp_3d || N < 1) return TEPOSE_E_ARG;
  if (!m->smpl_packed) return TEPOSE_E_STATE;
  SmplConsts sc = smpl_consts(m);
  const int* p = (const int*)jreg_packed;
  JregPacked jr{p, p + 32, (const float*)(p + 32 + 17 * kNV)};
  CK(launch_smpl_joints(sc, &jr, verts, nullptr, nullptr, N, kp_3d, nullptr, (hipStream_t)stream));
  return 0;
}

int tepose_smpl_fwd_per_person(const tepose_model* m, const float* pose, const float* betas, int N, float* verts,
                               void* workspace, size_t ws_bytes, void* stream) {
  if (!m || !pose || !betas || !verts || !workspace || N < 1) return TEPOSE_E_ARG;
  if (!m->smpl_packed) return TEPOSE_E_STATE;
  hipStream_t s = (hipStream_t)stream;
  Carver c(workspace, ws_bytes);
  RegWs w;
  carve_regressor(m, N, c, w);
  if (c.cur > ws_bytes) return TEPOSE_E_WORKSPACE;
  SmplConsts sc = smpl_consts(m);
  CK(launch_smpl_prep_pose(sc, 1, pose, 72, betas, 10, N, w.pf, w.amat, w.posed, s));
  CK(launch_smpl_person(sc, w.pf, w.amat, N, verts, s));
  return 0;
}

int tepose_smpl_verts_from_theta(const tepose_model* m, const float* theta, int N, float* verts, void* workspace,
                                 size_t ws_bytes, void* stream) {
  if (!m || !theta || !verts || !workspace || N < 1) return TEPOSE_E_ARG;
  if (!m->smpl_packed) return TEPOSE_E_STATE;
  hipStream_t s = (hipStream_t)stream;
  Carver c(workspace, ws_bytes);
  RegWs w;
  carve_regressor(m, N, c, w);
  if (c.cur > ws_bytes) return TEPOSE_E_WORKSPACE;
  SmplConsts sc = smpl_consts(m);
  if (smpl_small_ok(sc, N, m->opt)) {
    CK(launch_smpl_small(sc, 1, theta + 3, kTheta, theta + 75, kTheta, nullptr, 0, N, w.amat, nullptr, nullptr, nullptr, verts, s));
    return 0;
  }
  CK(launch_smpl_prep_pose(sc, 1, theta + 3, kTheta, theta + 75, kTheta, N, w.pf, w.amat, nullptr, s,
                           w.split ? w.pfP.hi : nullptr, w.split ? w.pfP.lo : nullptr, w.pfP.kst));
  CK((hipError_t)blend_shapes(m, w, N, s));
  CK(launch_smpl_skin(sc, w.vposed, w.amat, N, verts, s));
  return 0;
}

int tepose_filter_one_euro(float* x, int N, int D, float min_cutoff, float beta, float d_cutoff, void* stream) {
  if (!x || N < 1 || D < 1) return TEPOSE_E_ARG;
  CK(launch_one_euro(x, N, D, min_cutoff, beta, d_cutoff, (hipStream_t)stream));
  return 0;
}

int tepose_filter_slerp(const float* rotmat_in, float* rotmat_out, int N, int J, double ratio, void* stream) {
  if (!rotmat_in || !rotmat_out || N < 1 || J < 1) return TEPOSE_E_ARG;
  CK(launch_slerp_smooth(rotmat_in, rotmat_out, N, J, ratio, (hipStream_t)stream));
  return 0;
}

int tepose_rotmat_to_angle_axis(const float* R, int N, float* aa, void* stream) {
  if (!R || !aa || N < 1) return TEPOSE_E_ARG;
  CK(launch_rotmat_to_aa(R, N, aa, (hipStream_t)stream));
  return 0;
}

int tepose_rot6d_to_rotmat(const float* x6, int N, float* R, void* stream) {
  if (!x6 || !R || N < 1) return TEPOSE_E_ARG;
  CK(launch_rot6d_to_rotmat(x6, N, R, (hipStream_t)stream));
  return 0;
}

size_t tepose_gemm_h3_workspace_bytes(int M, int N, int K) {
  if (M < 1 || N < 1 || K < 1) return 0;
  const size_t a = gemm_h3_ws_bytes(M, N, K), b = gemm_h3s_ws_bytes(M, N, K);
  return a > b ? a : b;
}

int tepose_gemm_h3_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc,
                       int M, int N, int K, void* workspace, size_t ws_bytes, void* stream) {
  if (!A || !W || !C || !workspace || M < 1 || N < 1 || K < 1) return TEPOSE_E_ARG;
  if (K % 32 != 0) return TEPOSE_E_SHAPE;
  if (ws_bytes < gemm_h3_ws_bytes(M, N, K)) return TEPOSE_E_WORKSPACE;
  // TEPOSE_H3S=1 (read per call: this is the test / bench entry): the scaled-plane barrier-free kernel of gemm_h3s16c.hip, operand scales for
  // the ranges of tools/h3_loop.py / h3_bench.py and tests/test_gpu_stress.py; TEPOSE_H3S=mid: the 128 x 288-tile kernel of gemm_h3s.hip (N % 288 == 0)
  const char* pe = getenv("TEPOSE_H3S");
  const Options opt = options_from_env();          // no handle here: the environment's thresholds, read per call (test / bench entry)
  if (pe && pe[0] && std::string(pe) != "0" && std::string(pe).rfind("skinny", 0) != 0 && ws_bytes >= gemm_h3s_ws_bytes(M, N, K)) {
    CK(launch_gemm_h3s_f32(A, lda, W, ldw, C, ldc, M, N, K, 256.f, 16384.f, workspace, (hipStream_t)stream, opt, bias, std::string(pe) == "mid" ? 1 : 0));
    return 0;
  }
  // TEPOSE_H3S=skinny: the width-first kernel of skinny_h3.hip on the same planes, at any M (tools/mid_rows_gemm_bench.py)
  CK(launch_gemm_h3_f32(A, lda, W, ldw, bias, C, ldc, M, N, K, workspace, (hipStream_t)stream, opt, pe && std::string(pe) == "skinny" ? 1 : 0));
  return 0;
}

size_t tepose_gemm_workspace_bytes(int N, int K) {
  if (N < 1 || K < 1) return 0;
  return (size_t)round_up(N, 128) * round_up(K, 32) * sizeof(float) + 256;
}

int tepose_gemm_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc,
                    int M, int N, int K, int relu_a, void* workspace, size_t ws_bytes, void* stream) {
  if (!A || !W || !C || !workspace || M < 1 || N < 1 || K < 1) return TEPOSE_E_ARG;
  if (lda % 4 != 0 || ((uintptr_t)A & 15) != 0 || K % 32 != 0) return TEPOSE_E_SHAPE;
  if (ws_bytes < tepose_gemm_workspace_bytes(N, K)) return TEPOSE_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* wp = (float*)workspace;
  CK((hipError_t)pack(W, ldw, N, K, wp, round_up(N, 128), K, 0, 0, 0, 1, s));
  GemmArgs g = gemm(A, lda, wp, K, C, ldc, bias, M, N);
  g.relu_a = relu_a;
  CK(launch_gemm(g, s, options_from_env()));       // no handle here: the environment's thresholds, read per call (test / bench entry)
  return 0;
}

}  // extern "C"
