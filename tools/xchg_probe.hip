// Probe (diagnostic): latency of an activation exchange between two resident workgroups of one launch, as a
// row-split of the MLP step over a pair of CUs would need it ("what would move it").
// Each word on the wire is 8 bytes = (fp32 value, epoch): the flag travels with the data, so there is no fence, no
// separate flag store and no ordering requirement between stores.  Both directions use relaxed agent-scope atomics
// (global_store/global_load ... sc1), which are coherent whichever XCD the partner landed on.
// Every poll loop is bounded: a partner that never shows up sets the error word and the kernel still drains.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/xchg_probe tools/xchg_probe.hip && tools/xchg_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int NT = 512;
constexpr int RSLOTS = 4;
constexpr int MAXPOLL = 1 << 20;

__device__ __forceinline__ void st_word(uint64_t* p, float v, unsigned epoch) {
  const uint64_t w = ((uint64_t)epoch << 32) | (uint64_t)__float_as_uint(v);
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_word(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// L2-scope variant (only meaningful when both workgroups sit on the same XCD, i.e. share an L2): plain write-through
// store, load that bypasses the CU's L1 (sc0) but may hit in the XCD's L2
__device__ __forceinline__ void st_word_l2(uint64_t* p, float v, unsigned epoch) {
  const uint64_t w = ((uint64_t)epoch << 32) | (uint64_t)__float_as_uint(v);
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ uint64_t ld_word_l2(const uint64_t* p) {
  uint64_t w;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  return w;
}

// buf: [nblocks][RSLOTS][nvals] words.  Round r: write my nvals words (epoch r+1) into my slot r % RSLOTS, then read
// the partner's, then a workgroup barrier (the consumer of the real kernel is an LDS tile).  The partner can be at most
// one round away, so RSLOTS >= 2 slots never see a write-after-read hazard.
template <int L2ONLY> __global__ __launch_bounds__(NT) void k_xchg(uint64_t* buf, int rounds, int nvals, int pair_xor, unsigned epoch0,
                                             unsigned* err, unsigned long long* cyc, float* sink) {
  __shared__ float tile[4096];
  const int me = blockIdx.x, other = blockIdx.x ^ pair_xor;
  uint64_t* mine = buf + (size_t)me * RSLOTS * nvals;
  const uint64_t* theirs = buf + (size_t)other * RSLOTS * nvals;
  float acc = 0.f;
  bool dead = false;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    const unsigned ep = epoch0 + (unsigned)r;
    const int slot = r % RSLOTS;
    for (int i = threadIdx.x; i < nvals; i += NT) { if (L2ONLY) st_word_l2(mine + (size_t)slot * nvals + i, (float)(me + i) + acc * 0.f, ep); else st_word(mine + (size_t)slot * nvals + i, (float)(me + i) + acc * 0.f, ep); }
    for (int i = threadIdx.x; i < nvals; i += NT) {
      uint64_t w = 0;
      int n = 0;
      if (!dead) {
        do { w = L2ONLY ? ld_word_l2(theirs + (size_t)slot * nvals + i) : ld_word(theirs + (size_t)slot * nvals + i); } while ((unsigned)(w >> 32) != ep && ++n < MAXPOLL);
        if ((unsigned)(w >> 32) != ep) { dead = true; atomicAdd(err, 1u); }
      }
      tile[i & 4095] = __uint_as_float((unsigned)w);
    }
    __syncthreads();
    acc += tile[(threadIdx.x * 7 + r) & 4095 & (nvals - 1)];
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  sink[(size_t)blockIdx.x * NT + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same loop without the partner: the cost of the stores, one read of one's own words and the barriers
__global__ __launch_bounds__(NT) void k_self(uint64_t* buf, int rounds, int nvals, unsigned epoch0, float* sink) {
  __shared__ float tile[4096];
  uint64_t* mine = buf + (size_t)blockIdx.x * RSLOTS * nvals;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    const unsigned ep = epoch0 + (unsigned)r;
    const int slot = r % RSLOTS;
    for (int i = threadIdx.x; i < nvals; i += NT) st_word(mine + (size_t)slot * nvals + i, (float)i + acc * 0.f, ep);
    for (int i = threadIdx.x; i < nvals; i += NT) {
      uint64_t w; int n = 0;
      do { w = ld_word(mine + (size_t)slot * nvals + i); } while ((unsigned)(w >> 32) != ep && ++n < MAXPOLL);
      tile[i & 4095] = __uint_as_float((unsigned)w);
    }
    __syncthreads();
    acc += tile[(threadIdx.x * 7 + r) & 4095 & (nvals - 1)];
    __syncthreads();
  }
  sink[(size_t)blockIdx.x * NT + threadIdx.x] = acc;
}

int main() {
  const int nblk = 256, rounds = 200, maxvals = 2048;
  uint64_t* buf; hipMalloc(&buf, sizeof(uint64_t) * nblk * RSLOTS * maxvals);
  hipMemset(buf, 0, sizeof(uint64_t) * nblk * RSLOTS * maxvals);
  unsigned* err; hipMalloc(&err, 4); hipMemset(err, 0, 4);
  unsigned long long* cyc; hipMalloc(&cyc, 8 * nblk);
  float* sink; hipMalloc(&sink, sizeof(float) * nblk * NT);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned epoch = 1;
  std::vector<unsigned long long> hc(nblk);
  printf("pairs of workgroups, %d rounds per launch, 256 workgroups of 512 threads (one per CU)\n", rounds);
  for (int mode : {0, 1})
  for (int pair_xor : {8, 1, 128}) {
    for (int nvals : {256, 1024, 2048}) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode) k_xchg<1><<<nblk, NT>>>(buf, rounds, nvals, pair_xor, epoch, err, cyc, sink);
        else k_xchg<0><<<nblk, NT>>>(buf, rounds, nvals, pair_xor, epoch, err, cyc, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        epoch += rounds;
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      unsigned herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      hipMemcpy(hc.data(), cyc, 8 * nblk, hipMemcpyDeviceToHost);
      unsigned long long mx = 0; for (auto c : hc) mx = c > mx ? c : mx;
      printf("%s partner = block ^ %3d  %5d words (%5d B payload): %7.3f us per exchange round (launch %8.1f us, "
             "%llu memtime ticks per round, timeouts %u)\n",
             mode ? "L2-scope (sc0 load)  " : "agent-scope (sc1)    ", pair_xor, nvals, nvals * 4, best * 1e3f / rounds, best * 1e3f, mx / rounds, herr);
    }
  }
  for (int nvals : {256, 1024, 2048}) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      k_self<<<nblk, NT>>>(buf, rounds, nvals, epoch, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      epoch += rounds;
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("no partner (own words)  %5d words: %7.3f us per round\n", nvals, best * 1e3f / rounds);
  }
  return 0;
}
