// Diagnostic (not shipped): what does a grid barrier that carries one fp64 value per workgroup cost among 32 workgroups
//   (A) spread over all eight XCDs, agent-scope (sc1) 16-byte store + polling loads — the barrier of the cooperative SDE solve
//       (k_sde_eh_fast<DT, HT, true>, csrc/lrnde_sde_fast.hpp), and
//   (B) placed on ONE XCD (workgroup k of a 256-workgroup launch goes to XCD k mod 8: the workers are the workgroups with
//       k mod 8 == 0, the rest exit), exchanging through that XCD's L2 with atomics that carry no scope bit: they execute in
//       the L2, which every CU of the XCD shares — publish = two 64-bit swaps {sum bits; tag | checksum}, poll = 64-bit
//       atomic OR of 0 with return (a load would be served by the CU's own L1).
// Every worker checks every value it receives; the host prints microseconds per barrier and the XCC_ID of the workers.
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_barrier_probe.hip -o tools/tmp/xcd_barrier_probe && tools/tmp/xcd_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NWORK = 32, STRIDE = 8;   // 64-bit words per slot (one 64-byte line each)

struct Args {
  unsigned long long* slots;   // [2 parities][NWORK][STRIDE]
  int* xcc; int* err; unsigned long long* ticks;
  int niter, mode, spread;     // mode 0: agent scope, 1: L2 atomics; spread 1: workers = workgroups 0..31, 0: workgroups 0, 8, 16, ...
};

__device__ __forceinline__ double expected(int w, int it) { return (double)(w + 1) * 0.5 + (double)it; }

__global__ __launch_bounds__(64) void k_probe(Args a) {
  const int lane = threadIdx.x;
  int id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (lane == 0) a.xcc[blockIdx.x] = id & 0xf;
  int w;
  if (a.spread) { if ((int)blockIdx.x >= NWORK) return; w = blockIdx.x; }
  else { if (blockIdx.x % 8 != 0 || (int)(blockIdx.x / 8) >= NWORK) return; w = blockIdx.x / 8; }
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  int bad = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool dead = false;   // a barrier timed out: stop (the others time out once, too)
  for (int it = 0; it < a.niter && !dead; ++it) {
    unsigned long long* blk = a.slots + (size_t)(it & 1) * NWORK * STRIDE;
    const double mine = expected(w, it);
    const unsigned long long vb = __builtin_bit_cast(unsigned long long, mine);
    const unsigned tag = (unsigned)(it + 1);
    const unsigned long long w1 = (unsigned long long)tag | ((unsigned long long)((unsigned)(vb >> 32) ^ (unsigned)vb) << 32);
    if (a.mode == 0) {
      if (lane == 0) {
        const u32x4_ v = {(unsigned)vb, (unsigned)(vb >> 32), (unsigned)(vb >> 32) ^ (unsigned)vb, tag};
        unsigned long long* p = blk + (size_t)w * STRIDE;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
      }
      const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        u32x4_ v;
        const unsigned long long* p = blk + (size_t)(lane < NWORK ? lane : 0) * STRIDE;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        const bool good = lane >= NWORK || (v.w == tag && v.z == (v.x ^ v.y));
        if (__ballot(good) == ~0ull) {
          const double got = __builtin_bit_cast(double, (unsigned long long)v.x | ((unsigned long long)v.y << 32));
          if (lane < NWORK && got != expected(lane, it)) ++bad;
          break;
        }
        if (__builtin_amdgcn_s_memrealtime() - ts > 5000000ull) { bad += 1000; dead = true; break; }
      }
    } else {
      if (lane == 0) {
        unsigned long long* p = blk + (size_t)w * STRIDE;
        unsigned long long r0, r1;
        asm volatile("global_atomic_swap_x2 %0, %1, %2, off sc0" : "=v"(r0) : "v"(p), "v"(vb) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("global_atomic_swap_x2 %0, %1, %2, off sc0" : "=v"(r1) : "v"(p + 1), "v"(w1) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(r0), "v"(r1));
      }
      const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
      const unsigned long long zero = 0ull;
      for (;;) {
        const unsigned long long* p = blk + (size_t)(lane < NWORK ? lane : 0) * STRIDE;
        unsigned long long q1, q0;
        asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0" : "=v"(q1) : "v"(p + 1), "v"(zero) : "memory");
        asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0" : "=v"(q0) : "v"(p), "v"(zero) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(q0), "+v"(q1));
        const unsigned chk = (unsigned)(q0 >> 32) ^ (unsigned)q0;
        const bool good = lane >= NWORK || ((unsigned)q1 == tag && (unsigned)(q1 >> 32) == chk);
        if (__ballot(good) == ~0ull) {
          const double got = __builtin_bit_cast(double, q0);
          if (lane < NWORK && got != expected(lane, it)) ++bad;
          break;
        }
        if (__builtin_amdgcn_s_memrealtime() - ts > 5000000ull) { bad += 1000; dead = true; break; }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (bad) atomicAdd(a.err, bad);
  if (lane == 0 && w == 0) a.ticks[0] = t1 - t0;
}

int main() {
  Args a{};
  const int G = 256, niter = 2000;
  CHECK(hipMalloc(&a.slots, sizeof(unsigned long long) * 2 * NWORK * STRIDE));
  CHECK(hipMalloc(&a.xcc, sizeof(int) * G));
  CHECK(hipMalloc(&a.err, sizeof(int)));
  CHECK(hipMalloc(&a.ticks, sizeof(unsigned long long)));
  a.niter = niter;
  std::vector<int> hx(G);
  for (int spread = 1; spread >= 0; --spread)
    for (int mode = 0; mode < 2; ++mode) {
      if (spread == 1 && mode == 1) continue;   // L2 atomics across XCDs cannot work: not tried
      a.mode = mode; a.spread = spread;
      CHECK(hipMemset(a.slots, 0, sizeof(unsigned long long) * 2 * NWORK * STRIDE));
      CHECK(hipMemset(a.err, 0, sizeof(int)));
      void* args[] = {&a};
      CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_probe), dim3(G), dim3(64), args, 0, nullptr));
      CHECK(hipDeviceSynchronize());
      int err; unsigned long long tk;
      CHECK(hipMemcpy(&err, a.err, sizeof(int), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&tk, a.ticks, sizeof(tk), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hx.data(), a.xcc, sizeof(int) * G, hipMemcpyDeviceToHost));
      int off = 0;
      for (int i = 0; i < NWORK; ++i) { const int wg = spread ? i : 8 * i; if (hx[wg] != hx[0]) ++off; }
      printf("%s, %s: %.3f us per barrier (%d rounds), errors %d, workers off worker 0's XCD: %d\n",
             spread ? "workers on workgroups 0..31 (all XCDs)" : "workers on workgroups 0, 8, 16, ... (one XCD)",
             mode ? "L2 atomics (no scope bit)" : "agent scope (sc1 store + loads)", (double)tk / 100.0 / niter, niter, err, off);
    }
  printf("XCC_ID of workgroups 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
  printf("\n");
  return 0;
}
