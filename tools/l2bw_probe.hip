// Probe (diagnostic): per-CU bandwidth of streaming a small L2-resident table (the packed weights)
// into registers with global_load_dwordx4, as the f-eval does.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U> __global__ __launch_bounds__(512) void k_stream(const f32x4* w, size_t n4, int reps, float* sink, unsigned long long* cyc) {
  const int tid = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    for (size_t i = tid; i + (U - 1) * 512 < n4; i += U * 512) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = w[i + u * 512];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  const size_t bytes = 694 * 1024, n4 = bytes / 16;
  f32x4* w; hipMalloc(&w, bytes); hipMemset(w, 0, bytes);
  float* sink; hipMalloc(&sink, 512 * 4 * 1024); unsigned long long* cyc; hipMalloc(&cyc, 8); unsigned long long h;
  for (int nwg : {1, 32, 128, 256, 512}) {
    for (int U : {4, 8, 16}) {
      const int reps = 20;
      if (U == 4) k_stream<4><<<nwg, 512>>>(w, n4, reps, sink, cyc);
      else if (U == 8) k_stream<8><<<nwg, 512>>>(w, n4, reps, sink, cyc);
      else k_stream<16><<<nwg, 512>>>(w, n4, reps, sink, cyc);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("nwg=%3d U=%2d: %.1f B/clk per CU (%.1f us per 694 KB pass at 2.4 GHz)\n", nwg, U, (double)bytes * reps / h, h / (double)reps / 2400.0);
    }
  }
  return 0;
}
