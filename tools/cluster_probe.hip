// Diagnostic (not shipped): prices the ROW-SPLIT CLUSTER decomposition of the MLP vector field that DESIGN.md 4.1 proposes as
// the step kernel's successor.  Eight workgroups (one per CU) share 16 batch columns: member c owns rows [13c, 13c+13) of
// Dense-1 and [98c, 98c+98) of Dense-2; its weights (95 KB) stay in LDS for the whole launch; an evaluation costs two all-gathers inside the cluster (x: 784 x 16, h: 100 x 16) through
// global memory with agent-scope accesses and a bounded spin barrier of eight arrivals.  The launch runs NEV chained
// evaluations x <- f(x, t) — the data flow of a Runge-Kutta step's stages without the stage algebra — and is compared BIT FOR
// BIT with NEV chained lrnde_rhs calls (same canonical sums: every dot product is one workgroup's k-ordered MFMA chain over
// segments of 112 rows, partials added left to right, then the time column by fma, then the bias).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I include tools/cluster_probe.hip -o tools/tmp/cluster_probe -lrccl
#include "../localregneuralde.jl_amd/csrc/lrnde_kernels.hip"
#include <vector>
#include <cstdio>

namespace {
constexpr int CD = 784, CH = 100, CB = 16, CM = 8;     // state rows, hidden rows, columns per cluster, members
constexpr int R1 = 13, R2 = 98;                         // rows of Dense-1 / Dense-2 per member
constexpr int KS1 = CD / 4, KS2 = CH / 4;               // k-steps of 4: 196, 25
constexpr int T2 = 7;                                   // 16-row tiles of a member's Dense-2 rows (6 full + 2 rows)
constexpr int SEGS = 7, SEGK4 = 28;                     // canonical segments of Dense-1: 7 x 112 rows = 7 x 28 k-steps
constexpr int CNT = 256;

struct ClArgs {
  const float* W1f;   // [CM][KS1][64]  A fragments of the member's Dense-1 tile (rows beyond 13 / beyond H are zero)
  const float* W2f;   // [CM][T2][KS2][64]
  const float *w1t, *b1, *w2t, *b2;   // time columns and biases in natural order
  const float* u; float* out;         // (B, D)
  float* xg; float* hg;               // exchange: [cluster][CD][CB], [cluster][CH][CB]
  int* bar;                           // [cluster] arrival counters (zero at launch)
  int* err; int* xcc;                 // [gridDim.x]: every workgroup's XCC_ID
  float t; int nev, B;
};

// CL_L2 (compile-time): the exchange stays inside the XCD's L2 — plain stores (the CU's L1 writes through), sc0 loads (past the L1,
// served by the L2), atomics without a scope bit (executed by the L2).  Valid only if the eight members of a cluster sit on
// ONE XCD (the probe records every workgroup's XCC_ID and the host checks).  Default: agent-scope (sc1) accesses, valid anywhere.
#ifdef CL_L2
__device__ __forceinline__ void stcc_(float* p, float v) { *p = v; }
#define CL_LOAD4 "global_load_dwordx4 %0, %1, off sc0"
#define CL_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
__device__ __forceinline__ int bar_load(const int* p) { int v; asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
#else
__device__ __forceinline__ void stcc_(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define CL_LOAD4 "global_load_dwordx4 %0, %1, off sc1"
#define CL_SCOPE __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ int bar_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// a block of n4 float4 from global (written by other workgroups of the running launch: past the L2, one 16-byte sc1 load per
// lane per trip, all trips in flight before the one wait) into LDS
template <int TRIPS>
__device__ __forceinline__ void cl_gather(const float* src, float* dst, int n4) {
  f32x4 v[TRIPS];
#pragma unroll
  for (int i = 0; i < TRIPS; ++i) {
    const int q = (int)threadIdx.x + i * CNT;
    const float* p = src + (size_t)(q < n4 ? q : 0) * 4;
    asm volatile(CL_LOAD4 : "=v"(v[i]) : "v"(p) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < TRIPS; ++i) {
    asm volatile("" : "+v"(v[i]));   // (the value is defined by the wait above, not by the load's issue)
    const int q = (int)threadIdx.x + i * CNT;
    if (q < n4) *reinterpret_cast<f32x4*>(dst + (size_t)q * 4) = v[i];
  }
}

__device__ __forceinline__ bool cl_barrier(int* bar, int want, int* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, CL_SCOPE);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int good = 1;
    while (bar_load(bar) - want < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { good = 0; *err = 1; break; }   // 20 ms
    }
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

__global__ __launch_bounds__(CNT) void k_rhs_cluster(ClArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* w1s = sm;                         // [KS1][64]
  float* xs = w1s + KS1 * 64;              // [CD][CB]
  float* hs = xs + CD * CB;                // [CH][CB]
  float* ps = hs + CH * CB;                // [SEGS][256] segment partials of the Dense-1 tile
  float* w2s = ps + SEGS * 256;            // [T2][KS2][64]
  const int wg = blockIdx.x, xcd = wg & 7, j = wg >> 3;
  const int cluster = xcd * 4 + (j >> 3), c = j & 7;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col0 = cluster * CB;
  float* xg = a.xg + (size_t)cluster * CD * CB;
  float* hg = a.hg + (size_t)cluster * CH * CB;
  int* bar = a.bar + cluster;
  if (threadIdx.x == 0) { int id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); a.xcc[blockIdx.x] = id & 0xf; }
  for (int i = threadIdx.x; i < KS1 * 64; i += CNT) w1s[i] = a.W1f[(size_t)c * KS1 * 64 + i];
  for (int i = threadIdx.x; i < T2 * KS2 * 64; i += CNT) w2s[i] = a.W2f[(size_t)c * T2 * KS2 * 64 + i];
  // my Dense-2 rows of the state: tile tt, C fragment: row = 98 c + 16 tt + 4 (lane / 16) + r, column = lane % 16
  f32x4 xr[2];   // this wave's two tiles (tt = wave, wave + 4)
  const int sc = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tt = wave + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = 16 * tt + 4 * rq + r;   // row within the member's 98
      xr[q][r] = (tt < T2 && lr < R2 && col0 + sc < a.B) ? a.u[(size_t)(col0 + sc) * CD + R2 * c + lr] : 0.f;
    }
  }
  int phase = 0;
  for (int ev = 0; ev < a.nev; ++ev) {
    // ---- all-gather x: my 98 rows out, everybody's in ----
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int tt = wave + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int lr = 16 * tt + 4 * rq + r; if (tt < T2 && lr < R2) stcc_(xg + (size_t)(R2 * c + lr) * CB + sc, xr[q][r]); }
    }
    if (!cl_barrier(bar, CM * (++phase), a.err)) return;
    cl_gather<(CD * CB / 4 + CNT - 1) / CNT>(xg, xs, CD * CB / 4);
    __syncthreads();
    // ---- Dense-1: my 16-row tile, wave w takes segments w and w + 4 (each a chain from zero) ----
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int seg = wave + 4 * q;
      if (seg < SEGS) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k0 = 0; k0 < SEGK4; k0 += 7) {   // seven k-steps' operands first, then their MFMAs
          float av[7], bv[7];
#pragma unroll
          for (int u = 0; u < 7; ++u) { const int ks = seg * SEGK4 + k0 + u; av[u] = w1s[ks * 64 + lane]; bv[u] = xs[(4 * ks + rq) * CB + sc]; }
#pragma unroll
          for (int u = 0; u < 7; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ps[seg * 256 + (4 * rq + r) * 16 + sc] = acc[r];
      }
    }
    __syncthreads();
    {  // 16 x 16 outputs, one per thread: partials left to right, time column, bias, activation
      const int row = threadIdx.x >> 4, s2 = threadIdx.x & 15;
      float v = ps[row * 16 + s2];
#pragma unroll
      for (int seg = 1; seg < SEGS; ++seg) v = v + ps[seg * 256 + row * 16 + s2];
      const int o = R1 * c + row;
      if (row < R1 && o < CH) {
        float pre = fma_(a.w1t[o], a.t, v);
        pre = pre + a.b1[o];
        stcc_(hg + (size_t)o * CB + s2, act_apply(LRNDE_ACT_TANH, pre));
      }
    }
    if (!cl_barrier(bar, CM * (++phase), a.err)) return;
    cl_gather<(CH * CB / 4 + CNT - 1) / CNT>(hg, hs, CH * CB / 4);
    __syncthreads();
    // ---- Dense-2: my 7 tiles, one chain over K = H each (K <= 112: a single segment) ----
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int tt = wave + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (tt < T2) {
        const float* wf = w2s + (size_t)tt * KS2 * 64 + lane;
        float av[KS2], bv[KS2];
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) { av[ks] = wf[ks * 64]; bv[ks] = hs[(4 * ks + rq) * CB + sc]; }
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = 16 * tt + 4 * rq + r, o = R2 * c + lr;
          if (lr < R2) { const float pre = fma_(a.w2t[o], a.t, acc[r]); acc[r] = pre + a.b2[o]; }
        }
      }
      xr[q] = acc;
    }
    __syncthreads();   // xs / hs / ps are rewritten by the next evaluation
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tt = wave + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = 16 * tt + 4 * rq + r;
      if (tt < T2 && lr < R2 && col0 + sc < a.B) a.out[(size_t)(col0 + sc) * CD + R2 * c + lr] = xr[q][r];
    }
  }
}
}  // namespace

int main(int argc, char** argv) {
  const int B = 512, nev = argc > 1 ? atoi(argv[1]) : 6;
  lrnde_model_desc d{CD, CH, 1, 1};
  lrnde_ctx* c = nullptr;
  if (lrnde_create(&c, &d, 0, nullptr)) return 1;
  const size_t np = lrnde_param_count(&d);
  std::vector<float> hp(np);
  for (size_t i = 0; i < np; ++i) hp[i] = 0.08f * (float)((i * 2654435761u) % 1000) / 1000.f - 0.04f;
  const size_t n = (size_t)B * CD;
  std::vector<float> hu(n);
  for (size_t i = 0; i < n; ++i) hu[i] = (float)((i * 40503u) % 997) / 997.f;
  float *p, *u, *ra, *rb, *out;
  hipMalloc(&p, np * 4); hipMalloc(&u, n * 4); hipMalloc(&ra, n * 4); hipMalloc(&rb, n * 4); hipMalloc(&out, n * 4);
  hipMemcpy(p, hp.data(), np * 4, hipMemcpyHostToDevice); hipMemcpy(u, hu.data(), n * 4, hipMemcpyHostToDevice);
  lrnde_set_params(c, p, np);
  const float t = 0.37f;
  // reference: nev chained evaluations through the library
  const float* src = u; float* dst = ra;
  for (int e = 0; e < nev; ++e) { if (lrnde_rhs(c, src, t, B, dst)) return 2; src = dst; dst = (dst == ra) ? rb : ra; }
  hipDeviceSynchronize();
  std::vector<float> href(n), hout(n);
  hipMemcpy(href.data(), src, n * 4, hipMemcpyDeviceToHost);
  // pack: flat Lux layout [W1 (H x (D+1), column-major); b1; W2 (D x (H+1)); b2], the time column is the last column
  const float* W1 = hp.data(); const float* b1 = W1 + (size_t)CH * (CD + 1); const float* W2 = b1 + CH; const float* b2 = W2 + (size_t)CD * (CH + 1);
  std::vector<float> W1f((size_t)CM * KS1 * 64, 0.f), W2f((size_t)CM * T2 * KS2 * 64, 0.f), w1t(CH), w2t(CD);
  for (int m = 0; m < CM; ++m)
    for (int ks = 0; ks < KS1; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int row = l % 16, k = 4 * ks + l / 16, o = R1 * m + row;
        if (row < R1 && o < CH) W1f[((size_t)m * KS1 + ks) * 64 + l] = W1[o + (size_t)CH * k];
      }
  for (int m = 0; m < CM; ++m)
    for (int tt = 0; tt < T2; ++tt)
      for (int ks = 0; ks < KS2; ++ks)
        for (int l = 0; l < 64; ++l) {
          const int lr = 16 * tt + l % 16, k = 4 * ks + l / 16, o = R2 * m + lr;
          if (lr < R2 && k < CH) W2f[(((size_t)m * T2 + tt) * KS2 + ks) * 64 + l] = W2[o + (size_t)CD * k];
        }
  for (int o = 0; o < CH; ++o) w1t[o] = W1[o + (size_t)CH * CD];
  for (int o = 0; o < CD; ++o) w2t[o] = W2[o + (size_t)CD * CH];
  ClArgs a{};
  float *dW1f, *dW2f, *dw1t, *db1, *dw2t, *db2, *xg, *hg; int *bar, *err;
  hipMalloc(&dW1f, W1f.size() * 4); hipMalloc(&dW2f, W2f.size() * 4); hipMalloc(&dw1t, CH * 4); hipMalloc(&db1, CH * 4);
  hipMalloc(&dw2t, CD * 4); hipMalloc(&db2, CD * 4); hipMalloc(&xg, (size_t)32 * CD * CB * 4); hipMalloc(&hg, (size_t)32 * CH * CB * 4);
  hipMalloc(&bar, 32 * 4); hipMalloc(&err, 4); int* xcc; hipMalloc(&xcc, 256 * 4); a.xcc = xcc;
  hipMemcpy(dW1f, W1f.data(), W1f.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW2f, W2f.data(), W2f.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw1t, w1t.data(), CH * 4, hipMemcpyHostToDevice); hipMemcpy(db1, b1, CH * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw2t, w2t.data(), CD * 4, hipMemcpyHostToDevice); hipMemcpy(db2, b2, CD * 4, hipMemcpyHostToDevice);
  a.W1f = dW1f; a.W2f = dW2f; a.w1t = dw1t; a.b1 = db1; a.w2t = dw2t; a.b2 = db2; a.u = u; a.out = out; a.xg = xg; a.hg = hg;
  a.bar = bar; a.err = err; a.t = t; a.nev = nev; a.B = B;
  const size_t smem = sizeof(float) * ((size_t)KS1 * 64 + CD * CB + CH * CB + SEGS * 256 + (size_t)T2 * KS2 * 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_rhs_cluster), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  void* args[] = {&a};
  auto launch = [&]() {
    hipMemsetAsync(bar, 0, 32 * 4, nullptr); hipMemsetAsync(err, 0, 4, nullptr);
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_rhs_cluster), dim3(256), dim3(CNT), args, smem, nullptr);
  };
  hipError_t le = launch();
  if (le != hipSuccess) { printf("cooperative launch failed: %s\n", hipGetErrorString(le)); return 3; }
  hipDeviceSynchronize();
  int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
  hipMemcpy(hout.data(), out, n * 4, hipMemcpyDeviceToHost);
  size_t bad = 0; double md = 0;
  for (size_t i = 0; i < n; ++i) { if (memcmp(&hout[i], &href[i], 4)) ++bad; const double dd = fabs((double)hout[i] - href[i]); if (dd > md) md = dd; }
  {
    int hx[256]; hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
    int mixed = 0;
    for (int cl = 0; cl < 32; ++cl) { const int x0 = cl / 4, j0 = (cl % 4) * 8; for (int m = 1; m < 8; ++m) if (hx[x0 + 8 * (j0 + m)] != hx[x0 + 8 * j0]) ++mixed; }
    printf("XCC_ID of workgroups 0..15: "); for (int i = 0; i < 16; ++i) printf("%d ", hx[i]); printf("| members off their cluster's XCD: %d\n", mixed);
  }
  printf("%d chained evaluations, B=%d: %zu of %zu values differ from the library's (max |diff| %.3g), barrier timeouts %d\n", nev, B, bad, n, md, herr);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 50;
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("cluster form: %.2f us per launch of %d evaluations = %.2f us per evaluation (two memsets per launch included)\n", ms / reps * 1e3, nev, ms / reps * 1e3 / nev);
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < reps; ++i) { const float* s2 = u; float* d2 = ra; for (int e = 0; e < nev; ++e) { lrnde_rhs(c, s2, t, B, d2); s2 = d2; d2 = (d2 == ra) ? rb : ra; } }
  hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("library k_rhs_q: %.2f us per %d chained launches = %.2f us per evaluation (a launch each; inside k_step_q an evaluation is 7.7 us)\n", ms / reps * 1e3, nev, ms / reps * 1e3 / nev);
  return 0;
}
