// lrnde_cls_fused.hpp — the MNIST experiment's head on the MLP handle in TWO launches (it was three kernels, two
// hipMallocs and a host-side sum per training step): Dense(D => K) + logitcrossentropy forward together with the state
// cotangent (k_cls_fwd_bwdx), then the classifier cotangent as one batch-reduction GEMM plus the loss sum
// (k_cls_bwdw_loss).  experiments/src/construct.jl:199, experiments/src/utils.jl:88.  Included by lrnde_kernels.hip.
struct ClsOut { double loss_sum; int bad_label; int pad; };

// one wave per sample: logits[c] = sum_k W[c][k] u[k] + b[c] (the arithmetic and order of k_cls_fwd), softmax, the
// logit cotangent dl = (softmax - onehot) / Bnorm, and du[k] = sum_c dl[c] W[c][k] (the arithmetic of k_cls_bwd_x)
// WLDS: the K x (D+1) parameter block is staged in LDS once per workgroup (coalesced) and both passes read it from
// there — read straight from memory a lane's ten weights per k are ten 4-byte loads 40 bytes apart from its neighbour's
// (36 us for 16 MFLOP).  Same values, same order of operations.
// KT: the number of classes as a compile-time constant (10: the experiments' heads) or 0 = the runtime K.  With the runtime
// bound every `if (c < K)` is a branch around one LDS read and one fma — 260 of them in a row per lane, each waited for.
template <bool WLDS, int KT>
__global__ __launch_bounds__(256) void k_cls_fwd_bwdx(const float* u, const float* pcg, const int32_t* labels, int B, int D, int K_,
                                                      float Bnorm, float* logits, float* dl, float* loss_b, float* du, ClsOut* out) {
  const int K = KT ? KT : K_;
  extern __shared__ __attribute__((aligned(16))) float cls_w[];
  if (WLDS) {
    // (all of a thread's loads — 31 for the MNIST head — in flight before the first LDS store: one after the other — a load, its wait, a store, 31
    //  times for the MNIST head — this staging was most of the launch's 27 us)
    const int nw = K * (D + 1);
    for (int i0 = threadIdx.x; i0 < nw; i0 += 256 * 32) {
      float v[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) { const int i = i0 + r * 256; v[r] = i < nw ? pcg[i] : 0.f; }
#pragma unroll
      for (int r = 0; r < 32; ++r) { const int i = i0 + r * 256; if (i < nw) cls_w[i] = v[r]; }
    }
    __syncthreads();
  }
  const float* pc = WLDS ? cls_w : pcg;  // (a compile-time choice: the LDS pointer keeps its address space)
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* ub = u + (size_t)b * D;
  float lg[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) lg[c] = 0.f;
  // (the lane's u values eight at a time, all requested before the first is used: the rolled loop waited for every one of
  //  its 13 loads in turn — u comes from memory, the solve's last launch wrote it.  Same k order, same fma chain per lane.)
  for (int k0 = lane; k0 < D; k0 += 64 * 8) {
    float uu[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { const int k = k0 + 64 * r; uu[r] = k < D ? ub[k] : 0.f; }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = k0 + 64 * r;
      if (k < D) {
        const float* wk = pc + (size_t)K * k;
#pragma unroll
        for (int c = 0; c < 16; ++c) if (c < K) lg[c] = fma_(wk[c], uu[r], lg[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < K) {
      float s = lg[c];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
      lg[c] = s + pc[(size_t)K * D + c];
    }
  }
  // (every loop over the classes is unrolled over 16 with a guard: a loop with the runtime bound K indexes lg[] / dlv[]
  //  dynamically, which put both arrays into scratch memory — a round trip to memory per element, most of the launch's 28 us)
  float mx = lg[0];
#pragma unroll
  for (int c = 1; c < 16; ++c) if (c < K) mx = fmaxf_(mx, lg[c]);
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) if (c < K) se += expf_c(lg[c] - mx);
  const float lse = mx + logf(se);
  int y = labels[b];
  const bool bad = y < 0 || y >= K;
  if (bad) y = 0;
  float dlv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    dlv[c] = 0.f;
    if (c < K) dlv[c] = (expf_c(lg[c] - lse) - (c == y ? 1.f : 0.f)) / Bnorm;
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < K) {
        if (logits) logits[(size_t)b * K + c] = lg[c];
        dl[(size_t)b * K + c] = dlv[c];
      }
    }
    float ly = lg[0];
#pragma unroll
    for (int c = 1; c < 16; ++c) if (c < K && c == y) ly = lg[c];
    loss_b[b] = lse - ly;
    if (bad) atomicExch(&out->bad_label, 1);
  }
  if (du) {
    float* db = du + (size_t)b * D;
    for (int k = lane; k < D; k += 64) {
      const float* wk = pc + (size_t)K * k;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) if (c < K) s = fma_(dlv[c], wk[c], s);
      db[k] = s;
    }
  }
}

// workgroups 0..nt-1: the tiles of dpc = [dl^T u, dl^T 1] (pgrad_tile, first form); workgroup nt: the loss sum
__global__ __launch_bounds__(256) void k_cls_bwdw_loss(PgradArgs g, int nt, const float* loss_b, int B, ClsOut* out) {
  if ((int)blockIdx.x < nt) { pgrad_tile(g, (int)blockIdx.x); return; }
  __shared__ double red[4];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) acc += (double)loss_b[b];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out->loss_sum = ((red[0] + red[1]) + red[2]) + red[3];
}
