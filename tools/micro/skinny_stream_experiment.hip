// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 5; results: profiles/r05_mid_rows_gemm.txt).  It compiled as tepose_amd/csrc/skinny_stream.hip behind
// launch_skinny_gemm_h3_batch and was bit-identical to skinny_gemm_h3_kernel on every shape tried; it lost on time: an LDS-DMA instruction costs the issuing
// wave 60-185 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), so a wave that requests every KB it consumes is issue-bound at ~10 B/clk, and a CU takes
// in 45-65 GB/s whatever the tile shape.  222 x 9216 x 2144: 68-137 us against 50 us (128-row tiles) / 67 us (register-streamed, 4 passes).
// skinny_stream_h3_kernel: products of 65 .. ~800 rows on the split-precision planes of gemm_h3.hip (three fp16 MFMAs per product, fp32
// accumulate, hi / lo planes [K/32][R][32]) -- the regime between the width-first kernel of skinny_h3.hip (<= 64 rows per pass: at 222 rows
// = 37 clips x 6 frames it streams the weights four times) and the 128-row tiles of gemm_h3.hip (2 x 72 workgroups of 67 barriered K-tiles).
//
// At these sizes a product is bound by what a CU can pull through its 64 B/clk port to the L2, and by how many bytes it keeps in flight:
//   * a workgroup owns 16 MT rows x 16 NT columns and all of K; its 4 waves split K four ways (no A or W byte is fetched twice inside a
//     workgroup, no barrier in the K loop);
//   * every wave STREAMS its K-slice through a private LDS ring with LDS-DMA (global_load_lds_dwordx4: one instruction = the 1 KB block
//     of a 16-row fragment of one K-tile, which the blocked plane layout of common.h keeps contiguous): slot s of K-tile k + 1 is requested
//     into the ring position of slot s of K-tile k the moment that fragment is in registers, so 2 (MT + NT) - 4 KB per wave are in flight
//     at any time without a single VGPR holding them (the register-streamed kernel of skinny_h3.hip has at most two chunks in flight
//     and needs their registers);
//   * loads return in order, so "slot s has landed" is a counted s_waitcnt vmcnt with a literal that the fixed request order makes the
//     same for every slot (2 (MT + NT) - 4; the last K-tile counts down);
//   * the W fragments of a K-tile (2 NT) stay in registers while the A fragments (2 per row tile) pass through a double buffer;
//   * after the loop the four partial sums meet in LDS (the rings are dead by then) and each thread finishes MT NT elements with the
//     epilogue of skinny_gemm_h3_kernel (row scale, bias, addend, scale, optional planes out, optional second destination).
// Tile shapes are template parameters (MT NT <= 36: the accumulators take 8 MT NT registers of the 512 a wave has at one wave per SIMD);
// the launcher picks the shape and the number of row passes with a port-traffic model (skinny_stream_plan) -- squarer tiles move fewer
// bytes per CU, but the workgroup count has to fit the chip in whole rounds.  Same K order per wave and the same 4-way partial-sum
// grouping as skinny_gemm_h3_kernel: results are bit-identical to it (tests/test_gpu_skinny_stream.py).
#include <type_traits>

#include "common.h"

namespace tepose {

namespace {

typedef float f32x4q __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8q __attribute__((ext_vector_type(8)));

__device__ __forceinline__ const char* ss_uniform(const char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

// one LDS-DMA instruction: lane l moves 16 bytes from sbase + voff to LDS dst + 16 l
__device__ __forceinline__ void ss_dma(unsigned voff, const char* sbase, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(dst) : "m0", "memory");
}

template <int N>
__device__ __forceinline__ void ss_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// counted wait with a count that is a constant only after unrolling (the switch folds away)
__device__ __forceinline__ void ss_wait_vm_n(int n) {
#define TEPOSE_SS_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    TEPOSE_SS_CASE(0) TEPOSE_SS_CASE(1) TEPOSE_SS_CASE(2) TEPOSE_SS_CASE(3) TEPOSE_SS_CASE(4) TEPOSE_SS_CASE(5) TEPOSE_SS_CASE(6) TEPOSE_SS_CASE(7)
    TEPOSE_SS_CASE(8) TEPOSE_SS_CASE(9) TEPOSE_SS_CASE(10) TEPOSE_SS_CASE(11) TEPOSE_SS_CASE(12) TEPOSE_SS_CASE(13) TEPOSE_SS_CASE(14) TEPOSE_SS_CASE(15)
    TEPOSE_SS_CASE(16) TEPOSE_SS_CASE(17) TEPOSE_SS_CASE(18) TEPOSE_SS_CASE(19) TEPOSE_SS_CASE(20) TEPOSE_SS_CASE(21) TEPOSE_SS_CASE(22) TEPOSE_SS_CASE(23)
    TEPOSE_SS_CASE(24) TEPOSE_SS_CASE(25) TEPOSE_SS_CASE(26) TEPOSE_SS_CASE(27) TEPOSE_SS_CASE(28) TEPOSE_SS_CASE(29) TEPOSE_SS_CASE(30) TEPOSE_SS_CASE(31)
    TEPOSE_SS_CASE(32) TEPOSE_SS_CASE(33) TEPOSE_SS_CASE(34) TEPOSE_SS_CASE(35) TEPOSE_SS_CASE(36)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef TEPOSE_SS_CASE
}

}  // namespace

template <int MT, int NT>
__global__ void __launch_bounds__(256) skinny_stream_h3_kernel(H3ArgsBatch batch, int passes) {
  constexpr int NW = 4, D = 32;                 // D: ring slots (KB) per wave = requests in flight per wave
  constexpr int RING = D * 1024, RED = MT * NT * 1024;
  constexpr int LDSB = NW * (RING > RED ? RING : RED);
  static_assert(LDSB <= 152 * 1024 && MT * NT <= 36 && MT >= 2 && D % 2 == 0 && D - 2 - 2 * NT >= 0, "ring / accumulators fit a CU at one workgroup");
  __shared__ __attribute__((aligned(1024))) char lds[LDSB];
  const H3Args& a = batch.p[blockIdx.z];
  const int n0 = blockIdx.x * (16 * NT);
  const int tiles = (a.M + 15) >> 4;
  const int t0 = (int)((long)blockIdx.y * tiles / passes), t1 = (int)((long)(blockIdx.y + 1) * tiles / passes);   // balanced row passes
  if (n0 >= a.N || t0 >= t1) return;          // (the grid covers the largest product of the batch)
  const int m0 = t0 * 16;
  const int mt = __builtin_amdgcn_readfirstlane(min(t1 - t0, MT));   // row tiles of this pass (the launcher guarantees <= MT)
  const int row_end = min(t1 * 16, a.M);      // rows [m0, row_end) are this workgroup's; a ragged last fragment re-reads row row_end - 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)lds;
  const unsigned ring = lds0 + (unsigned)wave * RING;
  const unsigned frag = ring + (unsigned)(r16 * 64 + 16 * (q ^ ((r16 >> 2) & 3)));      // this lane's 16 bytes of a fragment block
  const long wrows = a.w_kst / 32;            // rows the W planes hold (padded, zero past N)

  const int KT = a.Kp / kPlaneK;
  const int c0 = (wave * KT) / NW, c1 = ((wave + 1) * KT) / NW;

  f32x4q acc[MT][NT], accx[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[i][t] = f32x4q{0.f, 0.f, 0.f, 0.f}; accx[i][t] = f32x4q{0.f, 0.f, 0.f, 0.f}; }

  if (c0 < c1) {
    // The wave's K-slice as ONE request stream of 1 KB slots, K-tile after K-tile: [W hi / lo of column tile 0 .. NT - 1, A hi / lo of row tile 0 .. mt - 1].
    // Flat slot f lives at ring position f % D; it is consumed in order, and the moment a pair of slots is in registers the pair D slots further down the
    // stream is requested into its place: D slots are in flight whatever mt is, and every "has landed" is the same counted wait.
    const int S = 2 * NT + 2 * mt;                               // slots per K-tile
    const int total = (c1 - c0) * S;
    const long a_step = a.a_kst * 2, w_step = a.w_kst * 2;       // bytes between K-tiles
    const char* ahp = ss_uniform((const char*)a.Ah + ((long)c0 * a.a_kst + (long)m0 * 32) * 2);     // the PRODUCER's K-tile
    const char* alp = ss_uniform((const char*)a.Al + ((long)c0 * a.a_kst + (long)m0 * 32) * 2);
    const char* whp = ss_uniform((const char*)a.Wh + ((long)c0 * a.w_kst + (long)n0 * 32) * 2);
    const char* wlp = ss_uniform((const char*)a.Wl + ((long)c0 * a.w_kst + (long)n0 * 32) * 2);
    const int l4 = lane >> 2;
    const unsigned lp = (unsigned)(lane & 3) * 16u;
    int pf = 0, ps = 0;                                          // producer: slots requested so far, slot inside its K-tile
    unsigned pring = 0;                                          // ... and ring position (bytes)
    bool tail = false;                                           // the producer has run out: from now on waits are vmcnt(0)
    auto issue_pair = [&]() __attribute__((always_inline)) {
      if (pf >= total) { tail = true; return; }
      unsigned off;
      const char *bh, *bl;
      if (ps < 2 * NT) {
        const int t = ps >> 1;
        const long row = min((long)(n0 + 16 * t + l4), wrows - 1);
        off = (unsigned)((row - n0) * 64) + lp;
        bh = whp; bl = wlp;
      } else {
        const int i = (ps - 2 * NT) >> 1;
        const int row = min(m0 + 16 * i + l4, row_end - 1);
        off = (unsigned)((row - m0) * 64) + lp;
        bh = ahp; bl = alp;
      }
      ss_dma(off, ss_uniform(bh), ring + pring);
      ss_dma(off, ss_uniform(bl), ring + pring + 1024u);
      pf += 2;
      pring = pring + 2048u == (unsigned)RING ? 0u : pring + 2048u;
      ps += 2;
      if (ps == S) {
        ps = 0;
        ahp = ss_uniform(ahp + a_step); alp = ss_uniform(alp + a_step);
        whp = ss_uniform(whp + w_step); wlp = ss_uniform(wlp + w_step);
      }
    };
#pragma unroll 1
    for (int k = 0; k < D / 2; ++k) issue_pair();
    tail = pf >= total;                                          // (a stream shorter than the ring)

    unsigned cring = 0;                                          // consumer ring position (bytes)
    h16x8q wh[NT], wl[NT], ah[2], al[2];
    // next pair of the stream -> registers; `behind`: slots requested behind this pair when the stream is in steady state
    auto read_pair = [&](h16x8q& hi, h16x8q& lo, int behind) __attribute__((always_inline)) {
      if (tail) ss_wait_vm<0>(); else ss_wait_vm_n(behind);
      const unsigned ad = frag + cring;
      asm volatile("ds_read_b128 %0, %1" : "=v"(hi) : "v"(ad) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(lo) : "v"(ad) : "memory");
      cring = cring + 2048u == (unsigned)RING ? 0u : cring + 2048u;
    };
#pragma unroll 1
    for (int kc = c0; kc < c1; ++kc) {
#pragma unroll
      for (int t = 0; t < NT; ++t) read_pair(wh[t], wl[t], D - 2 - 2 * t);
      read_pair(ah[0], al[0], D - 2 - 2 * NT);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0]), "+v"(al[0]) : : "memory");
#pragma unroll
      for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(wh[t]), "+v"(wl[t]));
#pragma unroll
      for (int t = 0; t < NT; ++t) issue_pair();                 // the W slots are free: their fragments are in registers
      __builtin_amdgcn_sched_barrier(0);
      // (the MFMAs of all MT row tiles are issued whatever mt is -- tiles >= mt multiply stale fragments into accumulators nobody reads: a branch around
      // an accumulator update makes the compiler shuttle the whole accumulator file between AGPRs, VGPRs and scratch every K-tile)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int b = i & 1;
        if (i + 1 < mt) read_pair(ah[b ^ 1], al[b ^ 1], D - 4);
        if (i < mt) issue_pair();                                // row tile i is in registers: its slots take the stream's next pair
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], wh[t], acc[i][t], 0, 0, 0);
          accx[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], wl[t], accx[i][t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) accx[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b], wh[t], accx[i][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[b ^ 1]), "+v"(al[b ^ 1]) : : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                                               // every wave's ring is dead: the partial sums take the space
  float* red = (float*)lds;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ee = 0; ee < 4; ++ee)
        red[((wave * MT * NT + i * NT + t) * 4 + ee) * 64 + lane] = acc[i][t][ee] + accx[i][t][ee] * (1.f / kLoScale);
    __builtin_amdgcn_sched_barrier(0);        // (one row tile at a time: the scheduler would otherwise hold all 8 MT NT sums in registers)
  }
  __syncthreads();
  const int e = tid >> 6;                                        // accumulator register of the elements this thread finishes
  const float sc = a.scale != 0.f ? a.scale : 1.f;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    const int row = m0 + i * 16 + q * 4 + e;
    if (row >= row_end) continue;
    const float rsc = a.row_scale ? a.row_scale[row] : 1.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n0 + t * 16 + r16;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[((w * MT * NT + i * NT + t) * 4 + e) * 64 + lane];
      if (col < a.N) {
        if (a.row_scale) v *= rsc;
        if (a.bias) v += a.bias[col];
        if (a.addend) v += a.addend[(long)row * a.ldadd + col];
        v *= sc;
        if (a.C2 && row >= a.c_split) a.C2[(long)(row - a.c_split) * a.ldc2 + col] = v;
        else a.C[(long)row * a.ldc + col] = v;
        if (a.Chi) {
          const long o = (long)(col >> 5) * a.c_kst + plane_index(row, col & 31, 0);
          split_hi_lo(v, a.Chi[o], a.Clo[o]);
        }
      }
    }
  }
}

// ---- shape choice.  Cost of a candidate (MT, NT) = rounds of the chip x bytes one workgroup pulls through its CU's L2 port
// (16 (rows + columns) x Kp x 4 bytes of planes), rounds = ceil(workgroups / CUs) with the workgroups of shorter products counted by their K.
namespace {
struct SsShape { int mt, nt; };
constexpr SsShape kSsShapes[] = {{16, 2}, {12, 3}, {9, 4}, {6, 6}, {4, 9}};

double ss_cost(const H3ArgsBatch& b, SsShape sh, int* passes_out) {
  int maxTiles = 0;
  for (int i = 0; i < b.n; ++i) { const int t = (b.p[i].M + 15) / 16; maxTiles = t > maxTiles ? t : maxTiles; }
  const int passes = (maxTiles + sh.mt - 1) / sh.mt;
  *passes_out = passes;
  // one workgroup per CU at a time (the ring takes most of the LDS): makespan ~ longest workgroup x whole rounds of the chip, the workgroups of
  // shorter products counted by their share of the longest one (the launch deals the longest K first)
  double work = 0., longest = 0.;          // in port bytes: sum over workgroups, and the longest one
  for (int i = 0; i < b.n; ++i) {
    const H3Args& a = b.p[i];
    if (a.M <= 0 || a.N <= 0) continue;
    const int tiles = (a.M + 15) / 16;
    const int p = passes < tiles ? passes : tiles;         // (passes beyond a product's tiles return at once)
    const int nb = (a.N + 16 * sh.nt - 1) / (16 * sh.nt);
    const double rows = 16. * ((tiles + p - 1) / p), cols = 16. * sh.nt;
    // bytes through the port + the MFMA issue time of a K-tile expressed in bytes (MT NT 3 MFMAs of 16 cycles against 64 B/clk; all MT tiles are issued)
    const double wg = ((rows + cols) * 4. + 0.25 * (sh.mt * sh.nt * 3 * 16 * 64 / 32.)) * a.Kp;
    work += wg * p * nb;
    longest = wg > longest ? wg : longest;
  }
  if (longest <= 0.) return 0.;
  const double rounds = work / longest / 256.;
  const double whole = (double)(long)(rounds + 0.999);
  return longest * (whole < 1. ? 1. : whole);
}
}  // namespace

int skinny_stream_min_m() {
  static const int v = [] { const char* e = getenv("TEPOSE_SKINNY_STREAM_MIN_M"); return e ? atoi(e) : 65; }();
  return v;
}

bool skinny_stream_ok(const H3Args& a) {
  return a.Kp % 32 == 0 && a.Kp >= 128 && a.grp_rows == 0 && a.M > 0 && a.N > 0;
}

// rows of the largest product -> (shape, passes); forced by TEPOSE_SKINNY_STREAM_SHAPE=<mt>x<nt> (A/B runs)
hipError_t launch_skinny_stream_h3_batch(const H3ArgsBatch& b, hipStream_t s) {
  if (b.n <= 0) return hipSuccess;
  for (int i = 0; i < b.n; ++i)
    if (!skinny_stream_ok(b.p[i])) return hipErrorInvalidValue;
  static const SsShape forced = [] {
    SsShape f{0, 0};
    const char* e = getenv("TEPOSE_SKINNY_STREAM_SHAPE");
    if (e) sscanf(e, "%dx%d", &f.mt, &f.nt);
    return f;
  }();
  SsShape best = kSsShapes[0];
  int passes = 1;
  double bc = -1.;
  for (const SsShape& sh : kSsShapes) {
    if (forced.mt && (sh.mt != forced.mt || sh.nt != forced.nt)) continue;
    int p;
    const double c = ss_cost(b, sh, &p);
    if (bc < 0. || c < bc) { bc = c; best = sh; passes = p; }
  }
  int maxN = 0;
  for (int i = 0; i < b.n; ++i) maxN = b.p[i].N > maxN ? b.p[i].N : maxN;
  const dim3 grid((maxN + 16 * best.nt - 1) / (16 * best.nt), passes, b.n);
#define TEPOSE_SS_LAUNCH(MTV, NTV) \
  if (best.mt == MTV && best.nt == NTV) hipLaunchKernelGGL((skinny_stream_h3_kernel<MTV, NTV>), grid, dim3(256), 0, s, b, passes)
  TEPOSE_SS_LAUNCH(16, 2);
  TEPOSE_SS_LAUNCH(12, 3);
  TEPOSE_SS_LAUNCH(9, 4);
  TEPOSE_SS_LAUNCH(6, 6);
  TEPOSE_SS_LAUNCH(4, 9);
#undef TEPOSE_SS_LAUNCH
  return hipGetLastError();
}

}  // namespace tepose
