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
template <bool WLDS>
__global__ __launch_bounds__(256) void k_cls_fwd_bwdx(const float* u, const float* pcg, const int32_t* labels, int B, int D, int K,
                                                      float Bnorm, float* logits, float* dl, float* loss_b, float* du, ClsOut* out) {
  extern __shared__ __attribute__((aligned(16))) float cls_w[];
  if (WLDS) {
    const int nw = K * (D + 1);
    for (int i = threadIdx.x; i < nw; i += 256) cls_w[i] = pcg[i];
    __syncthreads();
  }
  const float* pc = WLDS ? cls_w : pcg;  // (a compile-time choice: the LDS pointer keeps its address space)
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* ub = u + (size_t)b * D;
  float lg[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) lg[c] = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float uk = ub[k];
    const float* wk = pc + (size_t)K * k;
#pragma unroll
    for (int c = 0; c < 16; ++c) if (c < K) lg[c] = fma_(wk[c], uk, lg[c]);
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
  float mx = lg[0];
  for (int c = 1; c < K; ++c) mx = fmaxf_(mx, lg[c]);
  float se = 0.f;
  for (int c = 0; c < K; ++c) se += expf_c(lg[c] - mx);
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
    for (int c = 0; c < K; ++c) {
      if (logits) logits[(size_t)b * K + c] = lg[c];
      dl[(size_t)b * K + c] = dlv[c];
    }
    float ly = lg[0];
    for (int c = 1; c < K; ++c) if (c == y) ly = lg[c];
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
