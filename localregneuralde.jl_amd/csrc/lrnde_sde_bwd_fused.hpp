// lrnde_sde_bwd_fused.hpp — the reverse sweep of the NeuralDSDE layer's recorded Euler-Heun solve as ONE launch
// (lrnde_sde_node_backward_recorded, src/layers/neural_sde.jl:12: the reference differentiates the solve by taping the solver's
// own arithmetic; the tape of src/perform_step.jl:172-191 per recorded step is what runs here).  Included by lrnde_kernels.hip
// after lrnde_sde_node.hpp.
//
// The forward solve needs one launch per attempted step: the controller's error norm couples the batch.  The reverse sweep does
// not — the steps (start index and length on the caller's Brownian grid, the state at each step's start) are recorded, the
// weights are constants of the sweep, and a sample's cotangent runs backwards through ITS OWN states only.  What couples the
// batch is the parameter cotangent, a sum over samples.  So: a workgroup takes four samples (one wave each) through ALL K
// steps, newest first, with the three weight matrices in LDS (one copy each, leading dimension + 1: both W x and W^T x read it
// without bank conflicts), the cotangent and the step's elementwise quantities in the lanes' registers (lane d = row d), and
// the parameter cotangent of its four samples in LDS (each entry owned by one thread), summed over all steps; at the end it
// leaves one partial vector, and k_sde_bwd_reduce adds the workgroups' partials in workgroup order (fixed: the result does
// not depend on scheduling).  The generic path it replaces enqueued 15 launches per recorded step (two f-evals, four VJP +
// GEMM pairs, elementwise joins): 112 us per step at the MNIST-SDE shape against ~1 us here.
//
// Shapes: drift Chain(Dense(D => H, act), Dense(H => D)) without a time input, diffusion Dense(D => D), D <= 64, H <= 128 —
// those of the one-launch forward kernel (lrnde_sde_fast.hpp).  Arithmetic: plain fp32 fma chains over the input index in
// increasing order; the forward values are RECOMPUTED here in that order (the forward kernel's canonical MFMA chains differ
// in the last bits), which is inside the gradient's bar — tests/test_gpu_sde_layer.py holds the result to 5e-6 of float64
// autograd over the recorded grid, and to the generic path (LRNDE_NO_SDE_BWD_FUSED=1).

namespace {

constexpr int SBF_NT = 256, SBF_NS = 4, SBF_MAXSER = 512;   // (a longer series takes the generic path)

struct SdeBwdFusedArgs {
  const float* pdr;                 // flat drift parameters [vec(W1) (H x D, column-major); b1; vec(W2) (D x H); b2]
  const float* Wg; const float* bg; // diffusion: D x D column-major, D (zeros without a bias)
  int D, H, act, B, K;
  const float* x;                   // (B, D) the layer's input = start state of step 0
  const float* rec_u;               // (K, B, D) end state of accepted step k
  const int2* im;                   // (start index, length) of step k on the path's grid
  const float* W;                   // the caller's Brownian path ((nfine + 1), B, D)
  float h;                          // grid interval
  const float* du_series; int nseries; const int* ser_k; const float* ser_theta;   // cotangents of the caller's series
  float* dx;                        // (B, D): cotangent of the input
  float* part;                      // [gridDim.x][Ptot] parameter-cotangent partials of the workgroups
  int Pf, Ptot;                     // drift parameters; drift + diffusion (D*D + D)
};

// offsets inside a sample's vector block of one evaluation point: x (D), the constant 1, dpre (H), h (H), lam (D), lam_g (D);
// every vector padded to whole quads (the pads stay zero) so that a product's input is read four elements at a time
struct SbfOff { int Dq, Hq, X, ONE, DPRE, HV, LAM, LAMG, VS; };
__host__ __device__ inline SbfOff sbf_off(int D, int H) {
  SbfOff o; o.Dq = (D + 3) & ~3; o.Hq = (H + 3) & ~3;
  o.X = 0; o.ONE = o.Dq; o.DPRE = o.Dq + 4; o.HV = o.DPRE + o.Hq; o.LAM = o.HV + o.Hq; o.LAMG = o.LAM + o.Dq; o.VS = o.LAMG + o.Dq;
  return o;
}
__host__ __device__ inline int sbf_up4(int n) { return (n + 3) & ~3; }
inline size_t sbf_smem_bytes(int D, int H) {
  const SbfOff o = sbf_off(D, H);
  return sizeof(float) * ((size_t)sbf_up4((H + 1) * o.Dq + 4) + sbf_up4((D + 1) * o.Hq + 4) + sbf_up4((D + 1) * o.Dq + 4) + o.Hq + 2 * o.Dq   // weights, biases
                          + (size_t)SBF_NS * 2 * o.VS                                                  // two evaluation points per sample
                          + (size_t)((size_t)D * H * 2 + H + D + (size_t)D * D + D));                  // the workgroup's parameter cotangent
}

__global__ __launch_bounds__(SBF_NT) void k_sde_eh_bwd_fused(SdeBwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = a.D, H = a.H;
  const SbfOff o = sbf_off(D, H);
  const int Dq = o.Dq, Hq = o.Hq;
  const int ld1 = H + 1, ld2 = D + 1;          // leading dimensions of W1 (H x D) and of W2 (D x H) / Wg (D x D)
  float* W1s = sm;                             // W1[h + ld1 * d], d < Dq (zero columns beyond D)
  float* W2s = W1s + sbf_up4(ld1 * Dq + 4);    // W2[d + ld2 * h], h < Hq
  float* Wgs = W2s + sbf_up4(ld2 * Hq + 4);    // Wg[i + ld2 * j], j < Dq
  float* b1s = Wgs + sbf_up4(ld2 * Dq + 4);
  float* b2s = b1s + Hq;
  float* bgs = b2s + Dq;
  float* V = bgs + Dq;                         // [NS][2][VS]: evaluation point 0 = tmp, 1 = u
  float* accL = V + (size_t)SBF_NS * 2 * o.VS; // [Ptot] parameter cotangent of this workgroup's samples, summed over the steps
  __shared__ int sk[SBF_MAXSER];
  __shared__ float sth[SBF_MAXSER];
  const int tid = threadIdx.x, lane = tid & 63;
  const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's sample of the workgroup
  const int b = blockIdx.x * SBF_NS + sw;
  const bool valid = b < a.B;
  // ---- weights into LDS (pads zero), vector blocks zeroed (a sample beyond the batch stays all zero: it adds nothing) ----
  {
    const float* W1 = a.pdr; const float* b1 = W1 + (size_t)H * D; const float* W2 = b1 + H; const float* b2 = W2 + (size_t)D * H;
    for (int e = tid; e < (int)(V - sm); e += SBF_NT) sm[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < H * D; e += SBF_NT) { const int hh = e % H, d = e / H; W1s[hh + ld1 * d] = W1[e]; }
    for (int e = tid; e < D * H; e += SBF_NT) { const int d = e % D, hh = e / D; W2s[d + ld2 * hh] = W2[e]; }
    for (int e = tid; e < D * D; e += SBF_NT) { const int i = e % D, j = e / D; Wgs[i + ld2 * j] = a.Wg[e]; }
    for (int e = tid; e < H; e += SBF_NT) b1s[e] = b1[e];
    for (int e = tid; e < D; e += SBF_NT) { b2s[e] = b2[e]; bgs[e] = a.bg[e]; }
    for (int e = tid; e < SBF_NS * 2 * o.VS; e += SBF_NT) V[e] = 0.f;
    for (int e = tid; e < a.Ptot; e += SBF_NT) accL[e] = 0.f;
    for (int e = tid; e < a.nseries; e += SBF_NT) { sk[e] = a.ser_k[e]; sth[e] = a.ser_theta[e]; }
  }
  // which two vectors of a sample's block the parameter entry e multiplies (flat layout: W1, b1, W2, b2, Wg, bg);
  // q / n by a float reciprocal (exact for the q < 2^15 that occur here)
  const float rH = 1.0f / (float)H, rD = 1.0f / (float)D;
  auto decode = [&](int e, int& ao, int& bo) {
    const int n1 = H * D, n2 = n1 + H, n3 = n2 + D * H, n4 = a.Pf, n5 = n4 + D * D;
    if (e < n1) { const int d = (int)(((float)e + 0.5f) * rH); ao = o.DPRE + (e - d * H); bo = o.X + d; }
    else if (e < n2) { ao = o.DPRE + (e - n1); bo = o.ONE; }
    else if (e < n3) { const int q = e - n2; const int hh = (int)(((float)q + 0.5f) * rD); ao = o.LAM + (q - hh * D); bo = o.HV + hh; }
    else if (e < n4) { ao = o.LAM + (e - n3); bo = o.ONE; }
    else if (e < n5) { const int q = e - n4; const int jj = (int)(((float)q + 0.5f) * rD); ao = o.LAMG + (q - jj * D); bo = o.X + jj; }
    else { ao = o.LAMG + (e - n5); bo = o.ONE; }
  };
  __syncthreads();
  float* V0 = V + (size_t)(sw * 2 + 0) * o.VS;   // evaluation point tmp
  float* V1 = V + (size_t)(sw * 2 + 1) * o.VS;   // evaluation point u
  if (valid && lane == 0) { V0[o.ONE] = 1.0f; V1[o.ONE] = 1.0f; }
  const bool row = lane < D;                      // this lane owns row `lane` of every D-vector
  const bool k0 = lane < H, k1 = lane + 64 < H;   // ... and rows lane, lane + 64 of every H-vector
  const size_t nst = (size_t)a.B * D;
  const size_t g = valid ? (size_t)b * D + (row ? lane : 0) : 0;
  // Matrix-vector products of this wave's sample: the input is a vector of the sample's block in LDS (written by this wave
  // just before: a wave's LDS operations execute in order), read four elements at a time; the weight element of lane l and
  // input index k is W[l + ld k] (W x) or W[k + ld l] (W^T x) — consecutive lanes hit consecutive banks either way, the
  // leading dimensions being odd.  One fma chain per output over the input index in increasing order (the pads add 0 * w).
  // S = 1: rows `lane`; S = 2: rows `lane` and `lane + 64`.
  auto mv = [&](const float* Wm, int ld, const float* in, int nq, int base0, int base1, int kstride, bool two, float& r0, float& r1) {
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < nq; k += 4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(in + k);
      const float* w0 = Wm + base0 + kstride * k;
      const float a0 = w0[0], a1 = w0[kstride], a2 = w0[2 * kstride], a3 = w0[3 * kstride];
      s0 = fma_(a0, xv.x, s0); s0 = fma_(a1, xv.y, s0); s0 = fma_(a2, xv.z, s0); s0 = fma_(a3, xv.w, s0);
      if (two) {
        const float* w1 = Wm + base1 + kstride * k;
        const float c0 = w1[0], c1 = w1[kstride], c2 = w1[2 * kstride], c3 = w1[3 * kstride];
        s1 = fma_(c0, xv.x, s1); s1 = fma_(c1, xv.y, s1); s1 = fma_(c2, xv.z, s1); s1 = fma_(c3, xv.w, s1);
      }
    }
    r0 = s0; r1 = s1;
    (void)ld;
  };
  const bool twoH = H > 64;
  const int lh0 = k0 ? lane : 0, lh1 = k1 ? lane + 64 : 0, ld_ = row ? lane : 0;   // (lanes without a row compute row 0's value and drop it)
  // hidden layer at the point whose x is in Vp[X]: h to Vp[HV], act' in registers
  auto hidden = [&](float* Vp, float& a0, float& a1) {
    float p0, p1;
    mv(W1s, ld1, Vp + o.X, Dq, lh0, lh1, ld1, twoH, p0, p1);                    // W1 x: element (h, d) at h + ld1 d
    a0 = a1 = 0.f;
    if (k0) { const float pre = p0 + b1s[lane]; const float hv = act_apply(a.act, pre); a0 = act_deriv_c(a.act, pre, hv); if (valid) Vp[o.HV + lane] = hv; }
    if (k1) { const float pre = p1 + b1s[lane + 64]; const float hv = act_apply(a.act, pre); a1 = act_deriv_c(a.act, pre, hv); if (valid) Vp[o.HV + lane + 64] = hv; }
  };
  auto w2_h = [&](const float* Vp) { float r0, r1; mv(W2s, ld2, Vp + o.HV, Hq, ld_, 0, ld2, false, r0, r1); return row ? r0 : 0.f; };      // (W2 h)[d]
  auto wg_x = [&](const float* in) { float r0, r1; mv(Wgs, ld2, in, Dq, ld_, 0, ld2, false, r0, r1); return row ? r0 : 0.f; };             // (Wg x)[i]
  auto wgt_x = [&](const float* in) { float r0, r1; mv(Wgs, ld2, in, Dq, ld2 * ld_, 0, 1, false, r0, r1); return row ? r0 : 0.f; };        // (Wg^T x)[j]
  // J_f^T lam at the point of block Vp (lam in Vp[LAM], act' in registers): dpre to Vp[DPRE], returns row `lane`
  auto drift_vjp = [&](float* Vp, float a0, float a1) {
    float d0, d1;
    mv(W2s, ld2, Vp + o.LAM, Dq, ld2 * lh0, ld2 * lh1, 1, twoH, d0, d1);         // (W2^T lam)[h]: element (d, h) at d + ld2 h
    if (valid && k0) Vp[o.DPRE + lane] = d0 * a0;
    if (valid && k1) Vp[o.DPRE + lane + 64] = d1 * a1;
    float r0, r1;
    mv(W1s, ld1, Vp + o.DPRE, Hq, ld1 * ld_, 0, 1, false, r0, r1);               // (W1^T dpre)[d]: element (h, d) at h + ld1 d
    return row ? r0 : 0.f;
  };

  float ub = 0.f;   // cotangent of the state at the end of the step being undone (row `lane` of this wave's sample)
  // the step's data: start state, the two path values, its length; the next (older) step's are requested while this one is worked on
  auto step_src = [&](int k, float& u, float& wlo, float& whi, int& m) {
    const int2 im = a.im[k];
    const float* up = (k == 0) ? a.x : a.rec_u + (size_t)(k - 1) * nst;
    u = up[g]; wlo = a.W[(size_t)im.x * nst + g]; whi = a.W[(size_t)(im.x + im.y) * nst + g]; m = im.y;
  };
  float u_n = 0.f, wlo_n = 0.f, whi_n = 0.f;
  int m_n = 0;
  if (a.K > 0) step_src(a.K - 1, u_n, wlo_n, whi_n, m_n);
  for (int k = a.K - 1; k >= 0; --k) {
    const float u = (valid && row) ? u_n : 0.f;
    const float dW = (valid && row) ? whi_n - wlo_n : 0.f;
    const float dt = (float)m_n * a.h;
    if (k > 0) step_src(k - 1, u_n, wlo_n, whi_n, m_n);
    // cotangents of the series values taken inside step k: theta of each onto the step's end state ...
    for (int j = 0; j < a.nseries; ++j)
      if (sk[j] == k) { const float th = sth[j]; if (th != 0.f && valid && row) ub = ub + th * a.du_series[(size_t)j * nst + g]; }
    const float hdt = dt / 2.0f;
    // ---- forward pieces (src/perform_step.jl:175,179,183): du1 = f(u), L = g(u), tmp = (u + dt du1) + L dW ----
    if (valid && row) V1[o.X + lane] = u;
    float a1a, a1b, a2a, a2b;
    hidden(V1, a1a, a1b);
    const float du1 = w2_h(V1) + (row ? b2s[lane] : 0.f);
    const float L = wg_x(V1 + o.X) + (row ? bgs[lane] : 0.f);
    const float tmp = (valid && row) ? (u + dt * du1) + L * dW : 0.f;
    const float fb2 = hdt * ub, gb2 = (0.5f * dW) * ub;
    if (valid && row) { V0[o.X + lane] = tmp; V0[o.LAM + lane] = fb2; V0[o.LAMG + lane] = gb2; }
    hidden(V0, a2a, a2b);   // h(tmp), act'(tmp); f(tmp) itself is not needed
    // ---- second half backwards: cotangent of tmp ----
    const float dtf = drift_vjp(V0, a2a, a2b);
    const float dtg = wgt_x(V0 + o.LAMG);
    const float tb = dtf + dtg;
    const float du1b = hdt * ub + dt * tb;
    const float Lb = (0.5f * dW) * ub + dW * tb;
    const float up_ = ub + tb;
    if (valid && row) { V1[o.LAM + lane] = du1b; V1[o.LAMG + lane] = Lb; }
    const float duf = drift_vjp(V1, a1a, a1b);
    const float dug = wgt_x(V1 + o.LAMG);
    ub = (valid && row) ? (up_ + duf) + dug : 0.f;
    __syncthreads();
    // ---- the step's parameter cotangent: every entry is a product of two of the vectors now in LDS, summed over the
    //      workgroup's samples and the step's two evaluation points ----
#pragma unroll 1
    for (int e = tid; e < a.Ptot; e += SBF_NT) {
      int ao, bo;
      decode(e, ao, bo);
      float s = accL[e];
#pragma unroll
      for (int q = 0; q < SBF_NS * 2; ++q) s = fma_(V[q * o.VS + ao], V[q * o.VS + bo], s);
      accL[e] = s;
    }
    // ... and 1 - theta of the series values onto its start state
    for (int j = 0; j < a.nseries; ++j)
      if (sk[j] == k) { const float th = sth[j]; if (th != 1.0f && valid && row) ub = ub + (1.0f - th) * a.du_series[(size_t)j * nst + g]; }
    __syncthreads();
  }
  for (int j = 0; j < a.nseries; ++j)   // a saved start value is the input itself
    if (sk[j] < 0 && valid && row) ub = ub + a.du_series[(size_t)j * nst + g];
  if (valid && row) a.dx[g] = ub;
  float* pp = a.part + (size_t)blockIdx.x * a.Ptot;
  for (int e = tid; e < a.Ptot; e += SBF_NT) pp[e] = accL[e];   // (each entry is its owner thread's: no barrier needed)
}

// dp = sum over the workgroups' partials, in workgroup order; the diffusion part is [vec(Wg); bg] with bg present or not
__global__ void k_sde_bwd_reduce(const float* part, int nwg, int Ptot, int Pf, int Pg, float* dp_drift, float* dp_diff) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Ptot) return;
  float s = 0.f;
  for (int w = 0; w < nwg; ++w) s = s + part[(size_t)w * Ptot + e];
  if (e < Pf) dp_drift[e] = s;
  else if (e - Pf < Pg) dp_diff[e - Pf] = s;
}

}  // namespace
