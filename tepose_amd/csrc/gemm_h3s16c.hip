// gemm_h3s_persist16c_kernel: the persistent scaled-plane product on v_mfma_f32_16x16x32_f16 WITHOUT workgroup barriers in its K loop
// (round 4; every plain product of large batches since; the barrier form it replaced was removed in round 5, DESIGN_history.md).
//
// What the stamps of the barrier form said (round 4, profiles/r04_shape_ab.txt): per pair of stages a wave spends 4776 cycles for 3072 of MFMA; the
// two waves of a SIMD leave every barrier together, read their fragments together (matrix pipe idle) and multiply together.  Here the
// eight waves of a workgroup are coupled only through DATA: per ring slot one LDS counter `landed` that every wave bumps when its own
// share of a pair's LDS-DMA requests has landed.  A wave's interval for pair p:
//     poll landed[p & 1] >= 8 x (uses of that slot so far)     -- all eight shares of pair p are in LDS.  Every wave bumped it in ITS
//                                                                 interval p - 1 BEHIND its last fragment read of pair p - 1 (the LDS
//                                                                 executes a wave's instructions in order), so the slot of pair p - 1
//                                                                 is free as well
//     Q0, with this wave's 8 requests of pair p + 1 (into that slot) between its MFMAs
//     Q1 | s_waitcnt vmcnt(0); landed[(p + 1) & 1] += 1 | Q2 | Q3
// No s_barrier: waves drift (half an interval before anybody waits), the partners on a SIMD fall into opposite phases -- one reads
// fragments while the other multiplies.  Pairs run on across tiles (a tile is set up where its first pair is requested); the epilogue
// is per wave (bias and row scales straight from global memory), so nothing synchronises at a tile boundary either.  Same fragments,
// same K order, same accumulators as the barrier form had: its results were bit-identical.
// Measured (profiles/r04_shape_ab.txt): MFMA pipes 71 % busy instead of 64 %, 1.76-1.79 GHz instead of 1.86-1.92 (the chip gives part
// of it back), layer-0 projection 10.91 ms against 11.65 (same box, two rounds); where the landing is confirmed matters: after Q1
// 10.91, after Q2 11.18, after Q3 11.56 (drift tolerance beats request lead time); requests in one burst at the top: 11.08.
// No wave waits for anything but its own workgroup's waves, which are resident by construction, so the polls cannot hang; they are
// bounded all the same (a wave that gives up writes NaN to C[0], raises a counter that tepose_debug_kernel_errors() reads, and sets the
// forward's status word / the handle's fault word: H3SArgs::status, ::fault -- the library's failure channel, include/tepose_amd.h).
// Round 5, tried and removed (profiles/r05_throttle_ab.txt): a bounded-drift throttle between the 32 workgroups of an XCD -- every 4 pairs
// wave 0 published the workgroup's pair count in its XCD's line of a progress array and looked, asynchronously, at the line fetched 4 pairs earlier; a workgroup
// more than D pairs ahead of a peer slept.  Layer-0 projection, same box: no throttle 10.64 ms, D = 6: 11.15, D = 12: 10.76, D = 24: 10.77 -- the sleeping
// leaders cost more than the shared panels they keep in L2 save (as the per-tile rendezvous of round 4 did).
// Gate pre-activation outputs (H3SArgs::c_blk_hp) are written in the 16 x 16-blocked layout of common.h gi_blk_offset: the lane order
// of this kernel's C fragment IS that layout's block order, so every tile store is one contiguous KB.
#include "common.h"

#ifndef TEPOSE_C_VAR
#define TEPOSE_C_VAR 5     // A/B builds (bits 0-1: where the landing is confirmed: 0 after Q2, 1 after Q1, 2 after Q3; bit 2: the requests
#endif                     // between the MFMAs of Q0 instead of one burst at the top).  5 = the measured best

namespace tepose {

typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8c __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void h3s16c_tile_of_block(int bid, int nwg, int tilesM, int tilesN, int& tm, int& tn, int GM) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int group = lin / (GM * tilesN), rem = lin - group * GM * tilesN;
  const int gm = min(GM, tilesM - group * GM);
  tm = group * GM + rem % gm;
  tn = rem / gm;
}

template <int TAG>
__global__ void __launch_bounds__(512) gemm_h3s_persist16c_kernel(H3SArgs a, int tilesM, int tilesN, int GM, unsigned* err) {
  constexpr int NWN = 2, NST = 4, MT = 4, NT = 8;
  constexpr int HM = 256, HN = 256, HK = 16, RB = HK * 2, RPI = 1024 / RB;
  constexpr int STAGE = (2 * HM + 2 * HN) * RB, TOT = STAGE / 1024, Q = TOT / 8;
  constexpr int SPIN = 1 << 18;
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE + 64];
  const int ntiles = tilesM * tilesN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int t = lane & 15, g = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)lds;
  const unsigned cnt0 = lds0 + NST * STAGE;               // landed[0] at cnt0, landed[1] at cnt0 + 4
  if (tid < 16) ((unsigned*)(lds + NST * STAGE))[tid] = 0u;
  __syncthreads();
  auto fresh_lane = [&]() __attribute__((always_inline)) {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  const int i0 = wave * Q;
  bool isA[Q];
  int lrow0[Q];
  long kst[Q];
  const char* pbase[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    int ri = (i0 + q) * RPI;
    isA[q] = ri < 2 * HM;
    if (!isA[q]) ri -= 2 * HM;
    const bool lo = ri >= (isA[q] ? HM : HN);
    lrow0[q] = lo ? ri - (isA[q] ? HM : HN) : ri;
    pbase[q] = (const char*)(isA[q] ? (lo ? a.Al : a.Ah) : (lo ? a.Wl : a.Wh));
    kst[q] = (isA[q] ? a.a_kst : a.w_kst) * 2;
  }
  const char* sbase[Q];
  unsigned voff[Q];
  int m0 = 0, n0 = 0;
  auto setup = [&](int tile) {
    int tm, tn;
    h3s16c_tile_of_block(tile, ntiles, tilesM, tilesN, tm, tn, GM);
    m0 = tm * HM; n0 = tn * HN;
    const int l = fresh_lane();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int grow = isA[q] ? min(m0 + lrow0[q] + l / 2, a.M - 1) : n0 + lrow0[q] + l / 2;
      voff[q] = (unsigned)grow * RB + 16u * (l & 1);
      sbase[q] = pbase[q];
    }
  };
  // this wave's share of one pair of stages -> ring slot (pair & 1)
  auto request_pair = [&](int pair) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const unsigned dst = lds0 + (unsigned)((((2 * pair + s) % NST) * STAGE) + (i0 + q) * 1024);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[q]), "s"(sbase[q]), "s"(dst) : "m0", "memory");
        sbase[q] += kst[q];
      }
  };
  auto bump = [&](int slot) __attribute__((always_inline)) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(cnt0 + 4u * (unsigned)slot), "v"(1u) : "memory");
  };
  bool dead = false;
  const unsigned INJ = a.inject;
  auto poll = [&](int slot, int target) __attribute__((always_inline)) {
    const unsigned addr = cnt0 + 4u * (unsigned)slot;
    for (int spins = 0;; ++spins) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      if ((int)((unsigned)__builtin_amdgcn_readfirstlane((int)v) - (unsigned)target - INJ) >= 0) break;
      if (spins > SPIN) { dead = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  };
  const unsigned sx = 16u * ((g & 1) ^ ((t >> 3) & 1)) + (unsigned)(g >> 1) * STAGE;
  const unsigned abase = lds0 + (unsigned)(wm * 16 * MT + t) * RB + sx;
  const unsigned bbase = lds0 + 2 * HM * RB + (unsigned)(wn * 16 * NT + t) * RB + sx;
  constexpr int A_LO = HM * RB, W_LO = HN * RB;
  const int NP = a.Kp / (2 * HK);
  const bool vec = (((size_t)a.C | (size_t)a.bias) & 15) == 0 && (a.ldc & 3) == 0;

  f32x4c acc[MT][NT];
  h16x8c ah[MT], al[MT], bh[2][2], bl[2][2];
#define TEPOSE_C_READ_A(AB)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                                            \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ah[i]) : "v"(AB), "n"(i * 16 * RB));                                  \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(al[i]) : "v"(AB), "n"(i * 16 * RB + A_LO));                           \
  }
#define TEPOSE_C_READ_B(BB, QD)                                                                                               \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                             \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[(QD) & 1][u]) : "v"(BB), "n"((2 * (QD) + u) * 16 * RB));          \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bl[(QD) & 1][u]) : "v"(BB), "n"((2 * (QD) + u) * 16 * RB + W_LO));   \
  }
#define TEPOSE_C_WAIT_B(N, X)                                                                                                 \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(bh[X][0]), "+v"(bh[X][1]), "+v"(bl[X][0]), "+v"(bl[X][1]) : : "memory")
  auto request_one = [&](int pair, int k) __attribute__((always_inline)) {       // k-th of the 2 Q requests of a pair
    const int sgi = k / Q, q = k % Q;
    const unsigned dst = lds0 + (unsigned)((((2 * pair + sgi) % NST) * STAGE) + (i0 + q) * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[q]), "s"(sbase[q]), "s"(dst) : "m0", "memory");
    sbase[q] += kst[q];
  };
  auto quarter = [&](int QD, int X, bool dma = false, int pair = 0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        acc[i][2 * QD + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[X][u], ah[i], acc[i][2 * QD + u], 0, 0, 0);
        if (dma) request_one(pair, i * 2 + u);               // (compile-time subscripts into sbase / voff / kst)
      }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[i][2 * QD + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[X][u], ah[i], acc[i][2 * QD + u], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[i][2 * QD + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[X][u], al[i], acc[i][2 * QD + u], 0, 0, 0);
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  setup(tile);
  int dtile = tile, dpt = 0;                              // the tile / pair the NEXT request belongs to
  request_pair(0);
  ++dpt;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bump(0);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4c{0.f, 0.f, 0.f, 0.f};
  int pt = 0, gp = 0;
  int tm0 = m0, tn0 = n0;
  for (;;) {
    poll(gp & 1, 8 * (gp / 2 + 1));
    if (dead) break;
    // the next pair in request order: this tile's, or the first one of this workgroup's next tile
    bool requested = false;
    if (dpt >= NP) {
      const int nt = dtile + (int)gridDim.x;
      if (nt < ntiles) { dtile = nt; dpt = 0; setup(nt); }
    }
    if (dpt < NP) {
      if (!(TEPOSE_C_VAR & 4)) request_pair(gp + 1);
      ++dpt; requested = true;
    }
    const unsigned par = (unsigned)(gp & 1) * 2u * STAGE;
    const unsigned ab = abase + par, bb = bbase + par;
    TEPOSE_C_READ_A(ab)
    TEPOSE_C_READ_B(bb, 0)
    TEPOSE_C_READ_B(bb, 1)
    asm volatile("s_waitcnt lgkmcnt(4)"
                 : "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]),
                   "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bl[0][0]), "+v"(bl[0][1])
                 :
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if ((TEPOSE_C_VAR & 4) && requested) quarter(0, 0, true, gp + 1); else quarter(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    TEPOSE_C_READ_B(bb, 2)
    TEPOSE_C_WAIT_B(4, 1);
    __builtin_amdgcn_sched_barrier(0);
    quarter(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    TEPOSE_C_READ_B(bb, 3)
    TEPOSE_C_WAIT_B(4, 0);
    auto confirm = [&]() __attribute__((always_inline)) {
      if (requested) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own share of pair gp + 1 (and older stores) landed
        bump((gp + 1) & 1);
      }
    };
    // (the bump must follow this wave's last fragment READ of pair gp -- issued above -- not its arrival: the LDS executes a wave's
    // instructions in order, so the ds_add is behind the reads)
    if ((TEPOSE_C_VAR & 3) == 1) confirm();
    __builtin_amdgcn_sched_barrier(0);
    quarter(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    TEPOSE_C_WAIT_B(0, 1);                                  // every fragment of pair gp is in registers
    if ((TEPOSE_C_VAR & 3) == 0) confirm();
    __builtin_amdgcn_sched_barrier(0);
    quarter(3, 1);
    __builtin_amdgcn_sched_barrier(0);
    if ((TEPOSE_C_VAR & 3) == 2) confirm();
    ++gp;
    if (pt + 1 < NP) { ++pt; continue; }

    // ---- the finished tile (tm0, tn0): C = acc * row scale + bias, per wave, operands straight from global memory
    const int le = fresh_lane(), te = le & 15, ge = le >> 4;
    const bool full = tm0 + HM <= a.M && tn0 + HN <= a.N && vec;
    if (full) {
      float rs[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = tm0 + wm * 16 * MT + i * 16 + te;
        rs[i] = a.row_scale ? a.row_scale[row] * a.inv_scale : a.inv_scale;
      }
      float* c0 = a.C + (long)(tm0 + wm * 16 * MT + te) * a.ldc + tn0 + wn * 16 * NT + 4 * ge;
      const float* b0 = a.bias ? a.bias + tn0 + wn * 16 * NT + 4 * ge : nullptr;
      if (a.c_blk_hp) {
        // blocked gate pre-activations (common.h gi_blk_offset): every tile store is ONE contiguous KB (lane order = this fragment's)
        const long rt_stride = (long)a.N * 16;
        float* cb = a.C + (long)((tm0 + wm * 16 * MT) >> 4) * rt_stride + le * 4;
        const int col0 = tn0 + wn * 16 * NT;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4c bq = b0 ? *(const f32x4c*)(b0 + j * 16) : f32x4c{0.f, 0.f, 0.f, 0.f};
          float* cj = cb + gi_blk_col_block(col0 + j * 16, a.c_blk_hp);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            f32x4c v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = acc[i][j][c] * rs[i] + bq[c];
            *(f32x4c*)(cj + (long)i * rt_stride) = v;
          }
        }
      } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x4c bq = b0 ? *(const f32x4c*)(b0 + j * 16) : f32x4c{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          f32x4c v;
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = acc[i][j][c] * rs[i] + bq[c];
          *(f32x4c*)(c0 + (long)i * 16 * a.ldc + j * 16) = v;
        }
      }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = tm0 + wm * 16 * MT + i * 16 + te;
        if (row >= a.M) continue;
        const float rsv = a.row_scale ? a.row_scale[row] * a.inv_scale : a.inv_scale;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int col = tn0 + wn * 16 * NT + j * 16 + 4 * ge + c;
            if (col < a.N) {
              const float v = acc[i][j][c] * rsv + (a.bias ? a.bias[col] : 0.f);
              if (a.c_blk_hp) a.C[(long)(row >> 4) * a.N * 16 + gi_blk_col_block(col & ~15, a.c_blk_hp) + ((((col & 15) >> 2) * 16 + (row & 15)) << 2) + (col & 3)] = v;
              else a.C[(long)row * a.ldc + col] = v;
            }
          }
      }
    }
    const int next = tile + (int)gridDim.x;
    if (next >= ntiles) break;
    tile = next;
    // (the request stream is one pair ahead: it switched to `next` in this tile's last interval, so m0 / n0 describe it)
    tm0 = m0; tn0 = n0;
    pt = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4c{0.f, 0.f, 0.f, 0.f};
  }
#undef TEPOSE_C_READ_A
#undef TEPOSE_C_READ_B
#undef TEPOSE_C_WAIT_B
  if (dead && lane == 0) {                                   // never a plausible-looking wrong result with rc 0: NaN + the failure channel
    if (err) atomicAdd(err, 1u);
    a.C[0] = __builtin_nanf("");
    if (a.status) __hip_atomic_store(a.status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.fault) __hip_atomic_store(a.fault, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Debug counter of give-ups, ONE PER DEVICE (a process-wide counter allocated on whichever device was current at the first tepose_create was
// handed to launches on every other GPU of the process, whose atomicAdd would then fault instead of reporting).  Allocated by h3s16c_warm(), which
// tepose_set_blob calls with the blob's device current -- never inside a stream capture; a device without a counter launches with nullptr (the kernel
// then reports through the status / fault words only).
static constexpr int kMaxDev = 64;
static unsigned* g_h3s16c_err[kMaxDev] = {};

unsigned* h3s16c_err_of_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { (void)hipGetLastError(); return nullptr; }
  return __atomic_load_n(&g_h3s16c_err[dev], __ATOMIC_ACQUIRE);
}

void h3s16c_warm() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) { (void)hipGetLastError(); return; }
  if (__atomic_load_n(&g_h3s16c_err[dev], __ATOMIC_ACQUIRE)) return;
  unsigned* q = nullptr;
  if (hipMalloc((void**)&q, 256) != hipSuccess || !q) { (void)hipGetLastError(); return; }
  (void)hipMemset(q, 0, 256);
  unsigned* expected = nullptr;
  if (!__atomic_compare_exchange_n(&g_h3s16c_err[dev], &expected, q, false, __ATOMIC_RELEASE, __ATOMIC_ACQUIRE)) (void)hipFree(q);   // another thread was first
}

bool gemm_h3s16_ok(const H3SArgs& a) { return a.Kp % 32 == 0 && a.Kp >= 64; }

hipError_t launch_gemm_h3s16c(const H3SArgs& a, hipStream_t s, int tag, int gm_opt) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  if (!gemm_h3s16_ok(a)) return hipErrorInvalidValue;
  const int tilesM = (a.M + 255) / 256, tilesN = (a.N + 255) / 256;
  const int nt = tilesM * tilesN;
  // row tiles per XCD group of the walk (the 32 workgroups of an XCD take GM x 32 / GM tiles at a time; Options::s16_gm)
  const int gm = gm_opt >= 1 && gm_opt <= 32 ? gm_opt : 8;
  unsigned* err = h3s16c_err_of_device();
  if (tag == 0) hipLaunchKernelGGL(gemm_h3s_persist16c_kernel<0>, dim3(nt < 256 ? nt : 256), dim3(512), 0, s, a, tilesM, tilesN, gm, err);
  else hipLaunchKernelGGL(gemm_h3s_persist16c_kernel<1>, dim3(nt < 256 ? nt : 256), dim3(512), 0, s, a, tilesM, tilesN, gm, err);
  return hipGetLastError();
}

unsigned h3s16c_read_err() {            // sum over the devices that have a counter (unified addressing: readable from any current device)
  unsigned total = 0;
  for (int d = 0; d < kMaxDev; ++d) {
    unsigned* p = __atomic_load_n(&g_h3s16c_err[d], __ATOMIC_ACQUIRE);
    unsigned v = 0;
    if (p && hipMemcpy(&v, p, 4, hipMemcpyDeviceToHost) == hipSuccess) total += v;
  }
  return total;
}

}  // namespace tepose
