// Micro-benchmark (measurement only, not part of the library): the per-step cost of an all-to-all granule hand-off among P
// workgroups -- every step each participant publishes one 8-byte {tag, payload} granule and then sweeps all P granules
// until every tag is the step's -- with the participants (a) spread over the 8 XCDs and (b) all on ONE XCD, and with
// write-through (sc1) or plain stores.  Decides whether a persistent recurrent kernel should keep a direction inside an XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_handoff.hip -o gpurun_out/xcd_handoff && gpurun_out/xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// mode bit 0: participants = the workgroups of ONE XCD (the XCD of block 0); else blocks 0..P-1 (round-robin over XCDs)
// mode bit 1: plain stores instead of sc1
__global__ void __launch_bounds__(64) handoff(unsigned long long* gran, unsigned* slots, unsigned* xcd0, int P, int steps, int mode,
                                              long long* cycles, unsigned* where) {
  __shared__ int slot_s;
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) {
    int slot = -1;
    if (mode & 1) {
      if (blockIdx.x == 0) __hip_atomic_store(xcd0, x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned want;
      while ((want = __hip_atomic_load(xcd0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(2);
      if (x + 1 == want) slot = (int)atomicAdd(slots, 1u);
    } else {
      slot = blockIdx.x;
    }
    if (slot >= P) slot = -1;
    slot_s = slot;
    if (slot >= 0) where[slot] = x;
  }
  __syncthreads();
  const int slot = slot_s;
  if (slot < 0) return;
  const int lane = threadIdx.x;
  long long t0 = 0;
  for (int st = 1; st <= steps; ++st) {
    if (st == 9) t0 = wall_clock64();
    unsigned long long* buf = gran + (size_t)(st & 1) * 256;
    if (lane == 0) {
      const unsigned long long g = ((unsigned long long)st << 32) | (unsigned)slot;
      if (mode & 2) *(volatile unsigned long long*)(buf + slot) = g;
      else __hip_atomic_store(buf + slot, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (unsigned spins = 0;; ++spins) {
      if (spins > (1u << 22)) { if (lane == 0) *cycles = -1; return; }      // a participant is missing: give up
      bool ok = true;
      for (int i = lane; i < P; i += 64) {
        const unsigned long long g = __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok &= (unsigned)(g >> 32) == (unsigned)st;
      }
      if (__all(ok)) break;
    }
  }
  if (lane == 0 && slot == 0) *cycles = wall_clock64() - t0;
}

int main() {
  unsigned long long* gran; unsigned *slots, *xcd0, *where; long long* cyc;
  hipMalloc(&gran, 512 * 8); hipMalloc(&slots, 4); hipMalloc(&xcd0, 4); hipMalloc(&cyc, 8); hipMalloc(&where, 256 * 4);
  const int steps = 2008;
  for (int P : {8, 16, 32}) {
    for (int mode = 0; mode < 4; ++mode) {
      double best = 1e30;
      std::vector<unsigned> w(256);
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(gran, 0, 512 * 8); hipMemset(slots, 0, 4); hipMemset(xcd0, 0, 4); hipMemset(where, 0xff, 256 * 4);
        hipLaunchKernelGGL(handoff, dim3(256), dim3(64), 0, 0, gran, slots, xcd0, P, steps, mode, cyc, where);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(w.data(), where, 256 * 4, hipMemcpyDeviceToHost);
        const double us = c / 100.0 / (steps - 8);            // wall_clock64: 100 MHz
        if (us < best) best = us;
      }
      int nx[16] = {0};
      for (int i = 0; i < P; ++i) if (w[i] < 16) nx[w[i]]++;
      printf("P=%2d  %-10s %-6s stores: %.3f us per step   participants per XCD:", P, (mode & 1) ? "one XCD" : "all XCDs",
             (mode & 2) ? "plain" : "sc1", best);
      for (int i = 0; i < 8; ++i) printf(" %d", nx[i]);
      printf("\n");
    }
  }
  return 0;
}
