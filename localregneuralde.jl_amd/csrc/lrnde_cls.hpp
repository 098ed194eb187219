// lrnde_cls.hpp — Dense(D => K) + logitcrossentropy kernels (experiments/src/construct.jl:199, utils.jl:88), included
// inside the anonymous namespace of both translation units (MNIST head on the MLP handle, CIFAR head on the conv one).
// one wave per sample: logits[c] = sum_k W[c][k] u[k] + b[c] (fixed lane-strided order, butterfly reduce)
__global__ __launch_bounds__(256) void k_cls_fwd(const float* u, const float* pc, const int32_t* labels, int B, int D, int K,
                                                 float* logits, float* dl, float* loss_b) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* ub = u + (size_t)b * D;
  float lg[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) lg[c] = 0.f;
  // k outermost: one trip reads u[k] and the K weights of row k (contiguous) — 13 trips of independent loads instead of
  // K x 13 dependent ones; each class still accumulates over k = lane, lane + 64, ... in that order
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
  const int y = labels[b];
  if (lane == 0) {
    for (int c = 0; c < K; ++c) {
      if (logits) logits[(size_t)b * K + c] = lg[c];
      const float sm = expf_c(lg[c] - lse);
      dl[(size_t)b * K + c] = (sm - (c == y ? 1.f : 0.f)) / (float)B;
    }
    loss_b[b] = lse - lg[y];
  }
}
// du[b][k] = sum_c dl[b][c] W[c][k]
__global__ void k_cls_bwd_x(const float* dl, const float* pc, int B, int D, int K, float* du) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)B * D) return;
  const int b = i / D, k = i % D;
  float s = 0.f;
  for (int c = 0; c < K; ++c) s = fma_(dl[(size_t)b * K + c], pc[(size_t)c + (size_t)K * k], s);
  du[i] = s;
}
// dW[c][k] = sum_b dl[b][c] u[b][k] ; db[c] = sum_b dl[b][c]   (thread per (c,k); k == D is the bias)
__global__ void k_cls_bwd_w(const float* dl, const float* u, int B, int D, int K, float* dpc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * (D + 1)) return;
  const int c = i % K, k = i / K;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 16 <= B; b += 16) {  // sixteen samples' loads in flight; the four accumulators take them in the same order
    float dv[16], uv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { dv[j] = dl[(size_t)(b + j) * K + c]; uv[j] = k < D ? u[(size_t)(b + j) * D + k] : 1.f; }
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
      s0 = fma_(dv[j + 0], uv[j + 0], s0);
      s1 = fma_(dv[j + 1], uv[j + 1], s1);
      s2 = fma_(dv[j + 2], uv[j + 2], s2);
      s3 = fma_(dv[j + 3], uv[j + 3], s3);
    }
  }
  for (; b + 4 <= B; b += 4) {
    s0 = fma_(dl[(size_t)(b + 0) * K + c], k < D ? u[(size_t)(b + 0) * D + k] : 1.f, s0);
    s1 = fma_(dl[(size_t)(b + 1) * K + c], k < D ? u[(size_t)(b + 1) * D + k] : 1.f, s1);
    s2 = fma_(dl[(size_t)(b + 2) * K + c], k < D ? u[(size_t)(b + 2) * D + k] : 1.f, s2);
    s3 = fma_(dl[(size_t)(b + 3) * K + c], k < D ? u[(size_t)(b + 3) * D + k] : 1.f, s3);
  }
  for (; b < B; ++b) s0 = fma_(dl[(size_t)b * K + c], k < D ? u[(size_t)b * D + k] : 1.f, s0);
  dpc[(size_t)c + (size_t)K * k] = (s0 + s1) + (s2 + s3);
}
