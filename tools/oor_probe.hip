// Probe (diagnostic): what does a buffer_load_dwordx4 whose lanes are all outside the descriptor's range cost?
// The step kernel keeps its weight stream a fixed instruction sequence (counted vmcnt waits) by issuing the loads of
// absent row groups / padded k-quads with an out-of-range per-lane offset: they return 0 without touching memory.
// This measures whether they still take the CU's address-issue slot that a real 1-KiB wave-load takes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/oor_probe tools/oor_probe.hip && tools/oor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 wload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

// MODE 0: U real loads per trip; 1: U real + U out-of-range; 2: U out-of-range only; 3: U real + U with one active lane
template <int MODE> __global__ __launch_bounds__(512) void k(const float* w, int nbytes, int reps, float* sink, unsigned long long* cyc) {
  constexpr int U = 8;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int vreal = lane * 16, voor = 0x7ffffff0;
  const int vone = lane == 0 ? 0 : 0x7ffffff0;
  const int ntrip = nbytes / (8 * U * 1024);
  f32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    for (int i = 0; i < ntrip; ++i) {
      const int so = (i * 8 + wave) * U * 1024;
      f32x4 a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (MODE != 2) a[u] = wload(rs, vreal, so + u * 1024);
        if (MODE == 1 || MODE == 2) b[u] = wload(rs, voor, so + u * 1024);
        if (MODE == 3) b[u] = wload(rs, vone, so + u * 1024);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (MODE != 2) acc += a[u];
        if (MODE != 0) acc += b[u];
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  const int nbytes = 704 * 1024;  // the packed MNIST weights' size: L2 resident
  float* w; (void)hipMalloc(&w, nbytes); (void)hipMemset(w, 0, nbytes);
  float* sink; (void)hipMalloc(&sink, 512 * 4 * 256);
  unsigned long long* cyc; (void)hipMalloc(&cyc, 8);
  const int reps = 50, nwg = 128;
  const char* names[4] = {"real only            ", "real + out-of-range  ", "out-of-range only    ", "real + one-lane loads"};
  for (int mode = 0; mode < 4; ++mode) {
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) k<0><<<nwg, 512>>>(w, nbytes, reps, sink, cyc);
      if (mode == 1) k<1><<<nwg, 512>>>(w, nbytes, reps, sink, cyc);
      if (mode == 2) k<2><<<nwg, 512>>>(w, nbytes, reps, sink, cyc);
      if (mode == 3) k<3><<<nwg, 512>>>(w, nbytes, reps, sink, cyc);
      unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      if (h < best) best = h;
    }
    const double per = (double)best / reps;
    printf("%s: %9.0f ticks per pass over %d KiB (%d wave-loads per CU) = %.1f real B/tick\n", names[mode], per, nbytes / 1024,
           nbytes / 1024, mode == 2 ? 0.0 : nbytes / per);
  }
  return 0;
}
