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
//
// At the MNIST-SDE shape class (D <= 32, H <= 64) both kernels run in the compile-time form further down (SbfR: per-product
// weight images read 16 bytes at a time, the parameter cotangent in registers, no barrier inside the sweep); the kernels
// right below are the general form (LRNDE_SDE_BWD_LDSACC=1 forces it).
//
// The regulariser's part — d(EEst*dt)/dp of the ONE local Euler-Heun step at (sol(t1), t1), src/perform_step.jl:193-205 with
// uprev, dW, dt constant (neural_sde.jl:42) — is a second kernel of the same construction (k_sde_eh_reg_fused): three
// evaluation points per sample (K and utilde, tmp, uprev), EEst taken from the forward's record (the norm is the one thing
// that couples the batch, and the forward has it).

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
  int dw_direct;                    // 1: W is the array of INCREMENTS, one (B, D) block per step (fixed-grid solve): dW = W[im.x]
  const float* du_series; int nseries; const int* ser_k; const float* ser_theta;   // cotangents of the caller's series
  float* dx;                        // (B, D): cotangent of the input
  float* part;                      // [gridDim.x][Ptot] parameter-cotangent partials of the workgroups
  int Pf, Ptot;                     // drift parameters; drift + diffusion (D*D + D)
  // the regulariser's kernel: the local step's start state, increment and end state, its dt, EEst and tolerances
  const float *u1, *dW1, *un1; float dt1, eest, abstol, reltol, delta;
  // the deferred form of the sweep (k_sde_eh_bwd_fused_r<.., true> + k_sde_bwd_hist_gemm): one record of SbfR::HREC floats per
  // (step, sample, evaluation point) — x, dpre, h, lam, lam_g — from which the parameter cotangent is formed after the sweep
  float* hist; int nrec;
  int rec0; float w_reg;   // the regulariser's kernel in the deferred form: its first record, the weight its seeds carry
};

// offsets inside a sample's vector block of one evaluation point: x (D), the constant 1, dpre (H), h (H), lam (D), lam_g (D)
// [, x_g (D): the diffusion's input where it is not the drift's]; every vector padded to whole quads (the pads stay zero) so
// that a product's input is read four elements at a time
struct SbfOff { int Dq, Hq, X, ONE, DPRE, HV, LAM, LAMG, XG, VS; };
__host__ __device__ inline SbfOff sbf_off(int D, int H, bool own_xg) {
  SbfOff o; o.Dq = (D + 3) & ~3; o.Hq = (H + 3) & ~3;
  o.X = 0; o.ONE = o.Dq; o.DPRE = o.Dq + 4; o.HV = o.DPRE + o.Hq; o.LAM = o.HV + o.Hq; o.LAMG = o.LAM + o.Dq; o.VS = o.LAMG + o.Dq;
  o.XG = o.X;
  if (own_xg) { o.XG = o.VS; o.VS += o.Dq; }
  return o;
}
__host__ __device__ inline int sbf_up4(int n) { return (n + 3) & ~3; }
// nev evaluation points per sample: 2 in the sweep, 3 (with their own x_g) in the regulariser's kernel
inline size_t sbf_smem_bytes(int D, int H, int nev) {
  const SbfOff o = sbf_off(D, H, nev == 3);
  return sizeof(float) * ((size_t)sbf_up4((H + 1) * o.Dq + 4) + sbf_up4((D + 1) * o.Hq + 4) + sbf_up4((D + 1) * o.Dq + 4) + o.Hq + 2 * o.Dq   // weights, biases
                          + (size_t)SBF_NS * nev * o.VS                                                // the evaluation points' vectors
                          + (size_t)((size_t)D * H * 2 + H + D + (size_t)D * D + D));                  // the workgroup's parameter cotangent
}

// what both kernels share: the LDS image of the weights, this wave's sample, the products
struct SbfDev {
  int D, H, Dq, Hq, ld1, ld2, act, Pf, Ptot, nev;
  SbfOff o;
  float *W1s, *W2s, *Wgs, *b1s, *b2s, *bgs, *V, *accL;
  int tid, lane, sw, b;
  bool valid, row, k0, k1, twoH;
  int lh0, lh1, ld_;
  size_t nst, g;

  __device__ __forceinline__ void setup(const SdeBwdFusedArgs& a, float* sm, int nev_) {
    D = a.D; H = a.H; act = a.act; Pf = a.Pf; Ptot = a.Ptot; nev = nev_;
    o = sbf_off(D, H, nev == 3);
    Dq = o.Dq; Hq = o.Hq; ld1 = H + 1; ld2 = D + 1;   // leading dimensions of W1 (H x D) and of W2 (D x H) / Wg (D x D)
    W1s = sm;                                  // W1[h + ld1 * d], d < Dq (zero columns beyond D)
    W2s = W1s + sbf_up4(ld1 * Dq + 4);         // W2[d + ld2 * h], h < Hq
    Wgs = W2s + sbf_up4(ld2 * Hq + 4);         // Wg[i + ld2 * j], j < Dq
    b1s = Wgs + sbf_up4(ld2 * Dq + 4);
    b2s = b1s + Hq;
    bgs = b2s + Dq;
    V = bgs + Dq;                              // [NS][nev][VS]
    accL = V + (size_t)SBF_NS * nev * o.VS;    // [Ptot] parameter cotangent of this workgroup's samples
    tid = threadIdx.x; lane = tid & 63;
    sw = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's sample of the workgroup
    b = blockIdx.x * SBF_NS + sw;
    valid = b < a.B;
    // weights into LDS (pads zero), vector blocks zeroed (a sample beyond the batch stays all zero: it adds nothing)
    const float* W1 = a.pdr; const float* b1 = W1 + (size_t)H * D; const float* W2 = b1 + H; const float* b2 = W2 + (size_t)D * H;
    for (int e = tid; e < (int)(V - sm); e += SBF_NT) sm[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < H * D; e += SBF_NT) { const int hh = e % H, d = e / H; W1s[hh + ld1 * d] = W1[e]; }
    for (int e = tid; e < D * H; e += SBF_NT) { const int d = e % D, hh = e / D; W2s[d + ld2 * hh] = W2[e]; }
    for (int e = tid; e < D * D; e += SBF_NT) { const int i = e % D, j = e / D; Wgs[i + ld2 * j] = a.Wg[e]; }
    for (int e = tid; e < H; e += SBF_NT) b1s[e] = b1[e];
    for (int e = tid; e < D; e += SBF_NT) { b2s[e] = b2[e]; bgs[e] = a.bg[e]; }
    for (int e = tid; e < SBF_NS * nev * o.VS; e += SBF_NT) V[e] = 0.f;
    for (int e = tid; e < Ptot; e += SBF_NT) accL[e] = 0.f;
    row = lane < D;                       // this lane owns row `lane` of every D-vector
    k0 = lane < H; k1 = lane + 64 < H;    // ... and rows lane, lane + 64 of every H-vector
    twoH = H > 64;
    lh0 = k0 ? lane : 0; lh1 = k1 ? lane + 64 : 0; ld_ = row ? lane : 0;   // (lanes without a row compute row 0's value and drop it)
    nst = (size_t)a.B * D;
    g = valid ? (size_t)b * D + (row ? lane : 0) : 0;
  }
  __device__ __forceinline__ float* block(int ev) const { return V + (size_t)(sw * nev + ev) * o.VS; }
  __device__ __forceinline__ void ones() const { if (valid && lane == 0) for (int ev = 0; ev < nev; ++ev) block(ev)[o.ONE] = 1.0f; }

  // Matrix-vector product of this wave's sample: the input is a vector of the sample's block in LDS (written by this wave
  // just before: a wave's LDS operations execute in order), read four elements at a time; the weight element of lane l and
  // input index k is W[l + ld k] (W x: kstride = ld) or W[k + ld l] (W^T x: kstride = 1) — consecutive lanes hit consecutive
  // banks either way, the leading dimensions being odd.  One fma chain per output over the input index in increasing order
  // (the pads add 0 * w).  two: rows `lane` and `lane + 64`.
  __device__ __forceinline__ void mv(const float* Wm, const float* in, int nq, int base0, int base1, int kstride, bool two, float& r0, float& r1) const {
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
  }
  // hidden layer at the point whose x is in Vp[X]: h to Vp[HV], act' in registers
  __device__ __forceinline__ void hidden(float* Vp, float& a0, float& a1) const {
    float p0, p1;
    mv(W1s, Vp + o.X, Dq, lh0, lh1, ld1, twoH, p0, p1);                        // W1 x: element (h, d) at h + ld1 d
    a0 = a1 = 0.f;
    if (k0) { const float pre = p0 + b1s[lane]; const float hv = act_apply(act, pre); a0 = act_deriv_c(act, pre, hv); if (valid) Vp[o.HV + lane] = hv; }
    if (k1) { const float pre = p1 + b1s[lane + 64]; const float hv = act_apply(act, pre); a1 = act_deriv_c(act, pre, hv); if (valid) Vp[o.HV + lane + 64] = hv; }
  }
  __device__ __forceinline__ float f_out(const float* Vp) const {   // (W2 h + b2)[d] from Vp[HV]
    float r0, r1; mv(W2s, Vp + o.HV, Hq, ld_, 0, ld2, false, r0, r1); return row ? r0 + b2s[lane] : 0.f;
  }
  __device__ __forceinline__ float g_out(const float* in) const {   // (Wg x + bg)[i]
    float r0, r1; mv(Wgs, in, Dq, ld_, 0, ld2, false, r0, r1); return row ? r0 + bgs[lane] : 0.f;
  }
  __device__ __forceinline__ float wgt_x(const float* in) const {   // (Wg^T x)[j]
    float r0, r1; mv(Wgs, in, Dq, ld2 * ld_, 0, 1, false, r0, r1); return row ? r0 : 0.f;
  }
  // J_f^T lam at the point of block Vp (lam in Vp[LAM], act' in registers): dpre to Vp[DPRE], returns row `lane`
  __device__ __forceinline__ float drift_vjp(float* Vp, float a0, float a1) const {
    float d0, d1;
    mv(W2s, Vp + o.LAM, Dq, ld2 * lh0, ld2 * lh1, 1, twoH, d0, d1);            // (W2^T lam)[h]: element (d, h) at d + ld2 h
    if (valid && k0) Vp[o.DPRE + lane] = d0 * a0;
    if (valid && k1) Vp[o.DPRE + lane + 64] = d1 * a1;
    float r0, r1;
    mv(W1s, Vp + o.DPRE, Hq, ld1 * ld_, 0, 1, false, r0, r1);                  // (W1^T dpre)[d]: element (h, d) at h + ld1 d
    return row ? r0 : 0.f;
  }
  // The parameter cotangent of the vectors now in LDS: entry e of the flat layout (W1, b1, W2, b2, Wg, bg) is a product of two
  // vectors of a block, summed over the workgroup's samples and evaluation points.  q / n by a float reciprocal (exact for
  // the q < 2^15 that occur here).  Call between two barriers.
  __device__ __forceinline__ void accumulate() const {
    const float rH = 1.0f / (float)H, rD = 1.0f / (float)D;
    const int n1 = H * D, n2 = n1 + H, n3 = n2 + D * H, n4 = Pf, n5 = n4 + D * D;
    const int nb = SBF_NS * nev;
#pragma unroll 1
    for (int e = tid; e < Ptot; e += SBF_NT) {
      int ao, bo;
      if (e < n1) { const int d = (int)(((float)e + 0.5f) * rH); ao = o.DPRE + (e - d * H); bo = o.X + d; }
      else if (e < n2) { ao = o.DPRE + (e - n1); bo = o.ONE; }
      else if (e < n3) { const int q = e - n2; const int hh = (int)(((float)q + 0.5f) * rD); ao = o.LAM + (q - hh * D); bo = o.HV + hh; }
      else if (e < n4) { ao = o.LAM + (e - n3); bo = o.ONE; }
      else if (e < n5) { const int q = e - n4; const int jj = (int)(((float)q + 0.5f) * rD); ao = o.LAMG + (q - jj * D); bo = o.XG + jj; }
      else { ao = o.LAMG + (e - n5); bo = o.ONE; }
      float s = accL[e];
      for (int q = 0; q < nb; ++q) s = fma_(V[q * o.VS + ao], V[q * o.VS + bo], s);
      accL[e] = s;
    }
  }
  __device__ __forceinline__ void store_partial(float* part) const {
    float* pp = part + (size_t)blockIdx.x * Ptot;
    for (int e = tid; e < Ptot; e += SBF_NT) pp[e] = accL[e];   // (each entry is its owner thread's: no barrier needed)
  }
};

__global__ __launch_bounds__(SBF_NT) void k_sde_eh_bwd_fused(SdeBwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int sk[SBF_MAXSER];
  __shared__ float sth[SBF_MAXSER];
  SbfDev c;
  c.setup(a, sm, 2);
  for (int e = c.tid; e < a.nseries; e += SBF_NT) { sk[e] = a.ser_k[e]; sth[e] = a.ser_theta[e]; }
  __syncthreads();
  c.ones();
  const SbfOff o = c.o;
  const int lane = c.lane;
  const bool live = c.valid && c.row;
  const size_t nst = c.nst, g = c.g;
  float* V0 = c.block(0);   // evaluation point tmp
  float* V1 = c.block(1);   // evaluation point u
  float ub = 0.f;   // cotangent of the state at the end of the step being undone (row `lane` of this wave's sample)
  // the step's data: start state, the two path values, its length; the next (older) step's are requested while this one is worked on
  auto step_src = [&](int k, float& u, float& wlo, float& whi, int& m) {
    const int2 im = a.im[k];
    const float* up = (k == 0) ? a.x : a.rec_u + (size_t)(k - 1) * nst;
    u = up[g]; m = im.y;
    if (a.dw_direct) { wlo = 0.f; whi = a.W[(size_t)im.x * nst + g]; }
    else { wlo = a.W[(size_t)im.x * nst + g]; whi = a.W[(size_t)(im.x + im.y) * nst + g]; }
  };
  float u_n = 0.f, wlo_n = 0.f, whi_n = 0.f;
  int m_n = 0;
  if (a.K > 0) step_src(a.K - 1, u_n, wlo_n, whi_n, m_n);
  for (int k = a.K - 1; k >= 0; --k) {
    const float u = live ? u_n : 0.f;
    const float dW = live ? whi_n - wlo_n : 0.f;
    const float dt = (float)m_n * a.h;
    if (k > 0) step_src(k - 1, u_n, wlo_n, whi_n, m_n);
    // cotangents of the series values taken inside step k: theta of each onto the step's end state ...
    for (int j = 0; j < a.nseries; ++j)
      if (sk[j] == k) { const float th = sth[j]; if (th != 0.f && live) ub = ub + th * a.du_series[(size_t)j * nst + g]; }
    const float hdt = dt / 2.0f;
    // ---- forward pieces (src/perform_step.jl:175,179,183): du1 = f(u), L = g(u), tmp = (u + dt du1) + L dW ----
    if (live) V1[o.X + lane] = u;
    float a1a, a1b, a2a, a2b;
    c.hidden(V1, a1a, a1b);
    const float du1 = c.f_out(V1);
    const float L = c.g_out(V1 + o.X);
    const float tmp = live ? (u + dt * du1) + L * dW : 0.f;
    const float fb2 = hdt * ub, gb2 = (0.5f * dW) * ub;
    if (live) { V0[o.X + lane] = tmp; V0[o.LAM + lane] = fb2; V0[o.LAMG + lane] = gb2; }
    c.hidden(V0, a2a, a2b);   // h(tmp), act'(tmp); f(tmp) itself is not needed
    // ---- second half backwards: cotangent of tmp ----
    const float dtf = c.drift_vjp(V0, a2a, a2b);
    const float dtg = c.wgt_x(V0 + o.LAMG);
    const float tb = dtf + dtg;
    const float du1b = hdt * ub + dt * tb;
    const float Lb = (0.5f * dW) * ub + dW * tb;
    const float up_ = ub + tb;
    if (live) { V1[o.LAM + lane] = du1b; V1[o.LAMG + lane] = Lb; }
    const float duf = c.drift_vjp(V1, a1a, a1b);
    const float dug = c.wgt_x(V1 + o.LAMG);
    ub = live ? (up_ + duf) + dug : 0.f;
    __syncthreads();
    c.accumulate();
    // ... and 1 - theta of the series values onto its start state
    for (int j = 0; j < a.nseries; ++j)
      if (sk[j] == k) { const float th = sth[j]; if (th != 1.0f && live) ub = ub + (1.0f - th) * a.du_series[(size_t)j * nst + g]; }
    __syncthreads();
  }
  for (int j = 0; j < a.nseries; ++j)   // a saved start value is the input itself
    if (sk[j] < 0 && live) ub = ub + a.du_series[(size_t)j * nst + g];
  if (live) a.dx[g] = ub;
  c.store_partial(a.part);
}

// The two kernels for D <= DM, H <= HM = 64 with every size a compile-time constant (vectors and weight images padded to
// DM / HM, pads zero): the products unroll completely — a product's LDS reads are all in flight before its fma chain starts —
// and the parameter cotangent of a wave's sample lives in the wave's REGISTERS for the whole kernel (lane h: row h of dW1 and
// column h of dW2, DM values each; lane j: column j of dWg; one bias entry per lane), so a step of the sweep has no workgroup
// barrier at all: the four waves run their samples independently.  At the end the waves add their registers into the
// workgroup's LDS vector one after the other (wave order: fixed) and the partial goes out as in the kernels above.  Same
// expressions as k_sde_eh_bwd_fused / k_sde_eh_reg_fused; the sums over samples and steps associate differently (per sample
// over its steps first).
#ifdef LRNDE_SBF_STAMPS
__device__ unsigned long long g_sbf_stamps[4][8];
#define SBF_STAMP(k, i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sbf_stamps[(k)][(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SBF_STAMP(k, i) do { } while (0)
#endif
// The waves' register accumulators (lane h: g1[d] = dW1[h][d], g2[d] = dW2[d][h]; lane j: gg[i] = dWg[i][j]; one bias entry per
// lane) to part[workgroup][flat layout (W1, b1, W2, b2, Wg, bg), column-major], the four waves added in wave order.  Every wave
// writes its values into its OWN copy of a padded LDS vector (no read-modify-write, no wave waiting for another), one barrier,
// then every thread adds the four copies of its output entries.  Inside LDS the W2 and Wg parts have leading dimension D + 1:
// lane l writes at d + (D + 1) l — with the flat leading dimension D = 32 all 64 lanes of a store hit ONE bank.  (The first form
// — one LDS vector, `acc[i] += g` wave after wave — took 23 us of a 36-us launch.)
template <int DM>
__device__ __forceinline__ void sbf_sum_out(float* acc, const float (&g1)[DM], const float (&g2)[DM], const float (&gg)[DM], float gb1, float gb2,
                                            float gbg, int D, int H, int Pf, int Ptot, float* part) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool row = lane < D, k0 = lane < H;
  const int LP = D + 1;
  const int m1 = H * D, m2 = m1 + H, m3 = m2 + LP * H, m4 = m3 + D, m5 = m4 + LP * D, mtot = (m5 + D + 3) & ~3;   // padded layout
  __syncthreads();   // (the caller's LDS use is over: acc may alias it)
  float* mine = acc + (size_t)sw * mtot;
#pragma unroll
  for (int d = 0; d < DM; ++d) {
    if (d < D) {
      if (k0) { mine[lane + H * d] = g1[d]; mine[m2 + d + LP * lane] = g2[d]; }
      if (row) mine[m4 + d + LP * lane] = gg[d];
    }
  }
  if (k0) mine[m1 + lane] = gb1;
  if (row) { mine[m3 + lane] = gb2; mine[m5 + lane] = gbg; }
  __syncthreads();
  const int n1 = H * D, n2 = n1 + H, n3 = n2 + D * H, n4 = Pf, n5 = n4 + D * D;
  const float rD = 1.0f / (float)D;
  float* pp = part + (size_t)blockIdx.x * Ptot;
  auto src_of = [&](int e) {
    if (e < n2) return e;                                                                                    // W1, b1: as they are
    if (e < n3) { const int q = e - n2; const int hh = (int)(((float)q + 0.5f) * rD); return m2 + (q - hh * D) + LP * hh; }
    if (e < n4) return m3 + (e - n3);
    if (e < n5) { const int q = e - n4; const int jj = (int)(((float)q + 0.5f) * rD); return m4 + (q - jj * D) + LP * jj; }
    return m5 + (e - n5);
  };
  for (int e0 = tid; e0 < Ptot; e0 += SBF_NT * 4) {   // four entries per trip: sixteen LDS reads in flight
    float v[4][SBF_NS];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * SBF_NT;
      const int sidx = e < Ptot ? src_of(e) : 0;
#pragma unroll
      for (int w = 0; w < SBF_NS; ++w) v[q][w] = acc[(size_t)w * mtot + sidx];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * SBF_NT;
      if (e < Ptot) pp[e] = ((v[q][0] + v[q][1]) + v[q][2]) + v[q][3];
    }
  }
}
static_assert(SBF_NS == 4, "sbf_sum_out adds four waves");
inline size_t sbf_acc_floats(int D, int H) { return (size_t)SBF_NS * ((((size_t)H * D + H + (size_t)(D + 1) * H + D + (size_t)(D + 1) * D + D) + 3) & ~(size_t)3); }

template <int DM, int HM> struct SbfR {
  static_assert(DM % 4 == 0 && DM <= 64 && HM == 64, "one row per lane; the h vector is written by all 64 lanes");
  // Every product reads ITS OWN image of the matrix, one row of the product per lane, contiguous: lane l's row at l * LD with
  // LD = K + 4 — 16-byte reads, and LD mod 32 = 4 spreads eight lanes' quads over all 32 banks.  Six images (W1, W1^T, W2,
  // W2^T, Wg, Wg^T): 45 KB at 32 / 64; the workgroup's parameter cotangent reuses that space after the last step.
  static constexpr int LDD = DM + 4, LDH = HM + 4;                 // leading dimensions of images whose rows run over D / over H
  static_assert(LDD % 32 == 4 && LDH % 32 == 4, "conflict-free 16-byte reads");
  static constexpr int X = 0, DPRE = DM, HV = DM + HM, LAM = DM + 2 * HM, LAMG = LAM + DM, XG = LAMG + DM, VS = XG + DM;
  static constexpr int NIMG = 2 * HM * LDD + 2 * DM * LDH + 2 * DM * LDD;
  static constexpr int HREC = XG;   // a history record = the block's first five vectors (x, dpre, h, lam, lam_g), in that order
  static size_t smem_bytes(int nev, int D, int H) {   // (the four waves' padded cotangent vectors reuse the space at the end)
    const size_t work = (size_t)NIMG + HM + 2 * DM + (size_t)SBF_NS * nev * VS, fin = sbf_acc_floats(D, H);
    return sizeof(float) * (work > fin ? work : fin);
  }
  float *A1, *A1T, *A2, *A2T, *AG, *AGT, *V, *accL;
  int D, H, Ptot, Pf, act, nev, tid, lane, sw, b, lh, ld;
  bool valid, row, k0;
  float b1l, b2l, bgl;
  float g1[DM], g2[DM], gg[DM], gb1, gb2, gbg;
  size_t nst, g;

  __device__ __forceinline__ void setup(const SdeBwdFusedArgs& a, float* sm, int nev_) {
    D = a.D; H = a.H; Ptot = a.Ptot; Pf = a.Pf; act = a.act; nev = nev_;
    const int nimg = NIMG + HM + 2 * DM;
    A1 = sm;                     // W1[h][d] at h LDD + d       (W1 x)
    A1T = A1 + HM * LDD;         // W1[h][d] at d LDH + h       (W1^T dpre)
    A2 = A1T + DM * LDH;         // W2[d][h] at d LDH + h       (W2 h)
    A2T = A2 + DM * LDH;         // W2[d][h] at h LDD + d       (W2^T lam)
    AG = A2T + HM * LDD;         // Wg[i][j] at i LDD + j       (Wg x)
    AGT = AG + DM * LDD;         // Wg[i][j] at j LDD + i       (Wg^T lam_g)
    float* b1s = AGT + DM * LDD;
    float* b2s = b1s + HM;
    float* bgs = b2s + DM;
    accL = sm;                   // the padded cotangent vector, after the last step (finish)
    V = sm + nimg;               // [NS][nev][VS]
    tid = threadIdx.x; lane = tid & 63;
    sw = __builtin_amdgcn_readfirstlane(tid >> 6);
    b = blockIdx.x * SBF_NS + sw;
    valid = b < a.B;
    const float* W1 = a.pdr; const float* b1 = W1 + (size_t)H * D; const float* W2 = b1 + H; const float* b2 = W2 + (size_t)D * H;
    const int nz = (int)(V - sm) + SBF_NS * nev * VS;
    for (int e = tid; e < nz; e += SBF_NT) sm[e] = 0.f;
    __syncthreads();
    // (eight loads in flight per thread before the first store: a load per loop trip made the launch's start 20 global round trips long)
    auto fill = [&](const float* src, int n, auto&& put) {
      for (int e0 = tid; e0 < n; e0 += SBF_NT * 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = e0 + q * SBF_NT; v[q] = e < n ? src[e] : 0.f; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int e = e0 + q * SBF_NT; if (e < n) put(e, v[q]); }
      }
    };
    fill(W1, H * D, [&](int e, float w) { const int hh = e % H, d = e / H; A1[hh * LDD + d] = w; A1T[d * LDH + hh] = w; });
    fill(W2, D * H, [&](int e, float w) { const int d = e % D, hh = e / D; A2[d * LDH + hh] = w; A2T[hh * LDD + d] = w; });
    fill(a.Wg, D * D, [&](int e, float w) { const int i = e % D, jj = e / D; AG[i * LDD + jj] = w; AGT[jj * LDD + i] = w; });
    fill(b1, H, [&](int e, float w) { b1s[e] = w; });
    fill(b2, D, [&](int e, float w) { b2s[e] = w; });
    fill(a.bg, D, [&](int e, float w) { bgs[e] = w; });
    __syncthreads();
    row = lane < D; k0 = lane < H;
    lh = k0 ? lane : 0; ld = row ? lane : 0;   // (lanes without a row compute row 0's value and drop it)
    b1l = b1s[lh]; b2l = b2s[ld]; bgl = bgs[ld];
    nst = (size_t)a.B * D;
    g = valid ? (size_t)b * D + ld : 0;
#pragma unroll
    for (int d = 0; d < DM; ++d) { g1[d] = 0.f; g2[d] = 0.f; gg[d] = 0.f; }
    gb1 = gb2 = gbg = 0.f;
  }
  __device__ __forceinline__ float* block(int ev) const { return V + (size_t)(sw * nev + ev) * VS; }
  // sum over k < N of rowp[k] * in[k]: one fma chain in increasing k, row and input read four elements at a time (four
  // interleaved chains measured the same: a step is bound by the lone wave's LDS round trips, not by the chain's depth)
  template <int N> __device__ __forceinline__ float mv(const float* rowp, const float* in) const {
    f32x4 wv[N / 4];
#pragma unroll
    for (int q = 0; q < N / 4; ++q) wv[q] = *reinterpret_cast<const f32x4*>(rowp + 4 * q);
    float sacc = 0.f;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(in + 4 * q);
      sacc = fma_(wv[q].x, xv.x, sacc); sacc = fma_(wv[q].y, xv.y, sacc); sacc = fma_(wv[q].z, xv.z, sacc); sacc = fma_(wv[q].w, xv.w, sacc);
    }
    return sacc;
  }
  __device__ __forceinline__ void put(float* Vp, int off, float v) const { if (lane < DM) Vp[off + lane] = v; }   // a D-vector's row `lane` (0 beyond D)
  // hidden layer at the point whose x is in Vp[X]: h to Vp[HV] and returned, act' returned
  __device__ __forceinline__ void hidden(float* Vp, float& hv, float& da) const {
    const float pre = mv<DM>(A1 + lh * LDD, Vp + X) + b1l;
    hv = act_apply(act, pre); da = act_deriv_c(act, pre, hv);
    if (!k0) { hv = 0.f; da = 0.f; }
    Vp[HV + lane] = hv;
  }
  __device__ __forceinline__ float f_out(const float* Vp) const { return row ? mv<HM>(A2 + ld * LDH, Vp + HV) + b2l : 0.f; }   // (W2 h + b2)[d]
  __device__ __forceinline__ float g_out(const float* in) const { return row ? mv<DM>(AG + ld * LDD, in) + bgl : 0.f; }       // (Wg x + bg)[i]
  __device__ __forceinline__ float wgt_x(const float* in) const { return row ? mv<DM>(AGT + ld * LDD, in) : 0.f; }            // (Wg^T x)[j]
  // J_f^T lam at the point of block Vp (lam in Vp[LAM]): dpre to Vp[DPRE] and returned, row `lane` of W1^T dpre returned
  __device__ __forceinline__ float drift_vjp(float* Vp, float da, float& dpre) const {
    dpre = mv<DM>(A2T + lh * LDD, Vp + LAM) * da;          // (W2^T lam)[h] act'
    if (!k0) dpre = 0.f;
    Vp[DPRE + lane] = dpre;
    const float r = mv<HM>(A1T + ld * LDH, Vp + DPRE);     // (W1^T dpre)[d]
    return row ? r : 0.f;
  }
  // the parameter cotangent of one evaluation point: outer products of the vectors in LDS (x, lam, lam_g) with this lane's
  // entries (dpre[h], h[h]; the diffusion's input x_g[j]; lam[d], lam_g[d] for the biases)
  __device__ __forceinline__ void accumulate(const float* Vp, float dpre, float hv, float xgl, float laml, float lamgl) {
#pragma unroll
    for (int d = 0; d < DM; d += 4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(Vp + X + d);
      const f32x4 lv = *reinterpret_cast<const f32x4*>(Vp + LAM + d);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(Vp + LAMG + d);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        g1[d + r] = fma_(dpre, xv[r], g1[d + r]);      // dW1[h][d] += dpre[h] x[d]
        g2[d + r] = fma_(hv, lv[r], g2[d + r]);        // dW2[d][h] += lam[d] h[h]
        gg[d + r] = fma_(xgl, gv[r], gg[d + r]);       // dWg[i][j] += lam_g[i] x_g[j]     (lane j, register i)
      }
    }
    gb1 = gb1 + dpre; gb2 = gb2 + laml; gbg = gbg + lamgl;
  }
  // the block's record to the history buffer: 16 bytes per lane, one contiguous HREC-float row
  __device__ __forceinline__ void store_hist(float* dst, const float* Vp) const {
    if (lane < HREC / 4) *reinterpret_cast<f32x4*>(dst + 4 * lane) = *reinterpret_cast<const f32x4*>(Vp + 4 * lane);
  }
  __device__ __forceinline__ void finish(float* part) { sbf_sum_out<DM>(accL, g1, g2, gg, gb1, gb2, gbg, D, H, Pf, Ptot, part); }
};

template <int DM, int HM, bool DEFER = false>
__global__ __launch_bounds__(SBF_NT) void k_sde_eh_bwd_fused_r(SdeBwdFusedArgs a) {
  using R = SbfR<DM, HM>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int sk[SBF_MAXSER];
  __shared__ float sth[SBF_MAXSER];
  for (int e = threadIdx.x; e < a.nseries; e += SBF_NT) { sk[e] = a.ser_k[e]; sth[e] = a.ser_theta[e]; }
  SBF_STAMP(2, 0);
  R c;
  c.setup(a, sm, 2);   // (its barriers cover sk / sth)
  SBF_STAMP(2, 1);
  const bool row = c.row;
  const size_t nst = c.nst, g = c.g;
  float* V0 = c.block(0);   // evaluation point tmp
  float* V1 = c.block(1);   // evaluation point u
  float ub = 0.f;
  if (c.valid) {
    auto step_src = [&](int k, float& u, float& wlo, float& whi, int& m) {
      const int2 im = a.im[k];
      const float* up = (k == 0) ? a.x : a.rec_u + (size_t)(k - 1) * nst;
      u = up[g]; m = im.y;
      if (a.dw_direct) { wlo = 0.f; whi = a.W[(size_t)im.x * nst + g]; }
      else { wlo = a.W[(size_t)im.x * nst + g]; whi = a.W[(size_t)(im.x + im.y) * nst + g]; }
    };
    float u_n = 0.f, wlo_n = 0.f, whi_n = 0.f;
    int m_n = 0;
    if (a.K > 0) step_src(a.K - 1, u_n, wlo_n, whi_n, m_n);
    for (int k = a.K - 1; k >= 0; --k) {
      const float u = row ? u_n : 0.f;
      const float dW = row ? whi_n - wlo_n : 0.f;
      const float dt = (float)m_n * a.h;
      if (k > 0) step_src(k - 1, u_n, wlo_n, whi_n, m_n);
      // cotangents of the series values taken inside step k: theta of each onto the step's end state ...
      for (int j = 0; j < a.nseries; ++j)
        if (sk[j] == k) { const float th = sth[j]; if (th != 0.f && row) ub = ub + th * a.du_series[(size_t)j * nst + g]; }
      const float hdt = dt / 2.0f;
      // ---- forward pieces (src/perform_step.jl:175,179,183): du1 = f(u), L = g(u), tmp = (u + dt du1) + L dW ----
      c.put(V1, R::X, u);
      float hv1, da1, hv0, da0;
      c.hidden(V1, hv1, da1);
      const float du1 = c.f_out(V1);
      const float L = c.g_out(V1 + R::X);
      const float tmp = row ? (u + dt * du1) + L * dW : 0.f;
      const float fb2 = hdt * ub, gb2v = (0.5f * dW) * ub;
      c.put(V0, R::X, tmp); c.put(V0, R::LAM, fb2); c.put(V0, R::LAMG, gb2v);
      c.hidden(V0, hv0, da0);   // h(tmp), act'(tmp); f(tmp) itself is not needed
      // ---- second half backwards: cotangent of tmp ----
      float dpre0, dpre1;
      const float dtf = c.drift_vjp(V0, da0, dpre0);
      const float dtg = c.wgt_x(V0 + R::LAMG);
      const float tb = dtf + dtg;
      const float du1b = hdt * ub + dt * tb;
      const float Lb = (0.5f * dW) * ub + dW * tb;
      const float up_ = ub + tb;
      c.put(V1, R::LAM, du1b); c.put(V1, R::LAMG, Lb);
      const float duf = c.drift_vjp(V1, da1, dpre1);
      const float dug = c.wgt_x(V1 + R::LAMG);
      ub = row ? (up_ + duf) + dug : 0.f;
      if constexpr (DEFER) {   // the step's two records out; k_sde_bwd_hist_gemm forms the parameter cotangent from them
        float* hr = a.hist + ((size_t)k * a.B + c.b) * 2 * R::HREC;
        c.store_hist(hr, V0);
        c.store_hist(hr + R::HREC, V1);
      } else {
        c.accumulate(V0, dpre0, hv0, tmp, fb2, gb2v);
        c.accumulate(V1, dpre1, hv1, u, du1b, Lb);
      }
      // ... and 1 - theta of the series values onto its start state
      for (int j = 0; j < a.nseries; ++j)
        if (sk[j] == k) { const float th = sth[j]; if (th != 1.0f && row) ub = ub + (1.0f - th) * a.du_series[(size_t)j * nst + g]; }
    }
    for (int j = 0; j < a.nseries; ++j)   // a saved start value is the input itself
      if (sk[j] < 0 && row) ub = ub + a.du_series[(size_t)j * nst + g];
    if (row) a.dx[g] = ub;
  }
  SBF_STAMP(2, 2);
  if constexpr (!DEFER) c.finish(a.part);
  SBF_STAMP(2, 3);
}

// The deferred sweep with the weights in REGISTERS (DM = 32, HM = 64).  Without the cotangent's 96 accumulators a lane has room
// for its rows of all six products for the whole sweep: lane h keeps row h of W1 and column h of W2 (the products with H
// outputs); the products with D <= 32 outputs run on BOTH half-waves — lane d and lane d + 32 each keep one half of row d's
// input range (32 of W2's and W1^T's 64, 16 of Wg's and Wg^T's 32) and the halves meet in one cross-half add — 160 registers.
// A step then reads only its input vectors from LDS (68 16-byte reads instead of 208) and issues 272 fma instead of 416.  The
// D-output sums associate as (first half) + (second half): dx differs from the kernels above in the last bits.
template <int DM, int HM>
__global__ __launch_bounds__(SBF_NT) void k_sde_eh_bwd_sweep_res(SdeBwdFusedArgs a) {
  static_assert(DM == 32 && HM == 64, "half-wave split of the D-output products");
  using R = SbfR<DM, HM>;
  constexpr int HH = HM / 2, DH = DM / 2;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int sk[SBF_MAXSER];
  __shared__ float sth[SBF_MAXSER];
  for (int e = threadIdx.x; e < a.nseries; e += SBF_NT) { sk[e] = a.ser_k[e]; sth[e] = a.ser_theta[e]; }
  SBF_STAMP(2, 0);
  R c;
  c.setup(a, sm, 2);   // (the LDS images are the source of the register copies; its barriers cover sk / sth)
  const int lane = c.lane, half = lane >> 5, dl = lane & 31;
  const bool rowd = dl < c.D;               // this lane's D-row (both half-waves hold every D-vector's row dl)
  const int ldl = rowd ? dl : 0;
  float rw1[DM], rw2t[DM], rw2[HH], rw1t[HH], rwg[DH], rwgt[DH];
#pragma unroll
  for (int k = 0; k < DM; k += 4) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(c.A1 + c.lh * R::LDD + k), q = *reinterpret_cast<const f32x4*>(c.A2T + c.lh * R::LDD + k);
#pragma unroll
    for (int r = 0; r < 4; ++r) { rw1[k + r] = p[r]; rw2t[k + r] = q[r]; }
  }
#pragma unroll
  for (int k = 0; k < HH; k += 4) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(c.A2 + ldl * R::LDH + HH * half + k), q = *reinterpret_cast<const f32x4*>(c.A1T + ldl * R::LDH + HH * half + k);
#pragma unroll
    for (int r = 0; r < 4; ++r) { rw2[k + r] = p[r]; rw1t[k + r] = q[r]; }
  }
#pragma unroll
  for (int k = 0; k < DH; k += 4) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(c.AG + ldl * R::LDD + DH * half + k), q = *reinterpret_cast<const f32x4*>(c.AGT + ldl * R::LDD + DH * half + k);
#pragma unroll
    for (int r = 0; r < 4; ++r) { rwg[k + r] = p[r]; rwgt[k + r] = q[r]; }
  }
  const float b1l = c.b1l;
  const float b2l = __shfl(c.b2l, ldl, 64), bgl = __shfl(c.bgl, ldl, 64);   // (setup left them in lane d; lane d + 32 needs them too)
  SBF_STAMP(2, 1);
  // sum over k < N of w[k] * in[k], the input read four elements at a time
  auto dotr = [&](auto nk, const float* w, const float* in) {
    constexpr int N = decltype(nk)::value;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four chains (k mod 4), added pairwise: the step is a chain of seven such sums
#pragma unroll
    for (int k = 0; k < N; k += 4) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(in + k);
      s0 = fma_(w[k], xv.x, s0); s1 = fma_(w[k + 1], xv.y, s1); s2 = fma_(w[k + 2], xv.z, s2); s3 = fma_(w[k + 3], xv.w, s3);
    }
    return (s0 + s1) + (s2 + s3);
  };
  using IC_D = std::integral_constant<int, DM>; using IC_HH = std::integral_constant<int, HH>; using IC_DH = std::integral_constant<int, DH>;
  // the two half-waves' sums: the same value on both (v_permlane32_swap instead of the cross-lane read gave wrong sums here — not pursued)
  auto both = [&](float part) { return part + __shfl_xor(part, 32, 64); };
  auto hidden = [&](float* Vp, float& hv, float& da) {
    const float pre = dotr(IC_D{}, rw1, Vp + R::X) + b1l;
    hv = act_apply(c.act, pre); da = act_deriv_c(c.act, pre, hv);
    if (!c.k0) { hv = 0.f; da = 0.f; }
    Vp[R::HV + lane] = hv;
  };
  auto f_out = [&](const float* Vp) { const float t = both(dotr(IC_HH{}, rw2, Vp + R::HV + HH * half)); return rowd ? t + b2l : 0.f; };
  auto g_out = [&](const float* in) { const float t = both(dotr(IC_DH{}, rwg, in + DH * half)); return rowd ? t + bgl : 0.f; };
  auto wgt_x = [&](const float* in) { const float t = both(dotr(IC_DH{}, rwgt, in + DH * half)); return rowd ? t : 0.f; };
  auto drift_vjp = [&](float* Vp, float da) {
    float dpre = dotr(IC_D{}, rw2t, Vp + R::LAM) * da;
    if (!c.k0) dpre = 0.f;
    Vp[R::DPRE + lane] = dpre;
    const float t = both(dotr(IC_HH{}, rw1t, Vp + R::DPRE + HH * half));
    return rowd ? t : 0.f;
  };
  auto put = [&](float* Vp, int off, float v) { if (lane < DM) Vp[off + lane] = v; };
  const size_t nst = c.nst;
  const size_t g = c.valid ? (size_t)c.b * c.D + ldl : 0;
  float* V0 = c.block(0);   // evaluation point tmp
  float* V1 = c.block(1);   // evaluation point u
  float ub = 0.f;
  if (c.valid) {
    auto step_src = [&](int k, float& u, float& wlo, float& whi, int& m) {
      const int2 im = a.im[k];
      const float* up = (k == 0) ? a.x : a.rec_u + (size_t)(k - 1) * nst;
      u = up[g]; m = im.y;
      if (a.dw_direct) { wlo = 0.f; whi = a.W[(size_t)im.x * nst + g]; }
      else { wlo = a.W[(size_t)im.x * nst + g]; whi = a.W[(size_t)(im.x + im.y) * nst + g]; }
    };
    float u_n = 0.f, wlo_n = 0.f, whi_n = 0.f;
    int m_n = 0;
    if (a.K > 0) step_src(a.K - 1, u_n, wlo_n, whi_n, m_n);
    for (int k = a.K - 1; k >= 0; --k) {
      const float u = rowd ? u_n : 0.f;
      const float dW = rowd ? whi_n - wlo_n : 0.f;
      const float dt = (float)m_n * a.h;
      if (k > 0) step_src(k - 1, u_n, wlo_n, whi_n, m_n);
      for (int j = 0; j < a.nseries; ++j)
        if (sk[j] == k) { const float th = sth[j]; if (th != 0.f && rowd) ub = ub + th * a.du_series[(size_t)j * nst + g]; }
      const float hdt = dt / 2.0f;
      // ---- forward pieces (src/perform_step.jl:175,179,183) ----
      put(V1, R::X, u);
      float hv1, da1, hv0, da0;
      hidden(V1, hv1, da1);
      const float du1 = f_out(V1);
      const float L = g_out(V1 + R::X);
      const float tmp = rowd ? (u + dt * du1) + L * dW : 0.f;
      const float fb2 = hdt * ub, gb2v = (0.5f * dW) * ub;
      put(V0, R::X, tmp); put(V0, R::LAM, fb2); put(V0, R::LAMG, gb2v);
      hidden(V0, hv0, da0);
      // ---- second half backwards: cotangent of tmp ----
      const float dtf = drift_vjp(V0, da0);
      const float dtg = wgt_x(V0 + R::LAMG);
      const float tb = dtf + dtg;
      const float du1b = hdt * ub + dt * tb;
      const float Lb = (0.5f * dW) * ub + dW * tb;
      const float up_ = ub + tb;
      put(V1, R::LAM, du1b); put(V1, R::LAMG, Lb);
      const float duf = drift_vjp(V1, da1);
      const float dug = wgt_x(V1 + R::LAMG);
      ub = rowd ? (up_ + duf) + dug : 0.f;
      float* hr = a.hist + ((size_t)k * a.B + c.b) * 2 * R::HREC;
      c.store_hist(hr, V0);
      c.store_hist(hr + R::HREC, V1);
      for (int j = 0; j < a.nseries; ++j)
        if (sk[j] == k) { const float th = sth[j]; if (th != 1.0f && rowd) ub = ub + (1.0f - th) * a.du_series[(size_t)j * nst + g]; }
    }
    for (int j = 0; j < a.nseries; ++j)   // a saved start value is the input itself
      if (sk[j] < 0 && rowd) ub = ub + a.du_series[(size_t)j * nst + g];
    if (lane < c.D) a.dx[g] = ub;
  }
  SBF_STAMP(2, 2);
  SBF_STAMP(2, 3);
}

// The parameter cotangent from the sweep's history: record r = {x (DM), dpre (HM), h (HM), lam (DM), lam_g (DM)} of one
// (step, sample, evaluation point); dW1 = sum dpre x^T (H x D), dW2 = sum lam h^T (D x H), dWg = sum lam_g x^T (D x D), the
// biases = sum dpre, lam, lam_g — three GEMMs whose inner dimension is the record index: v_mfma_f32_16x16x4_f32, four records
// per instruction (A[i][k] = left vector of record k, B[k][j] = right vector).  A workgroup takes a contiguous chunk of the
// records in batches of 32 through LDS (record stride 232 floats: the four k-groups of a fragment read land on different bank
// octets), the next batch's quads in registers while this one is worked on; the 20 output tiles (8 + 8 + 4) go five to a wave,
// every tile owned by ONE wave — no cross-wave sum — and leave as the workgroup's partial straight from the C fragments
// (k_sde_bwd_reduce adds the partials).  The biases: waves 0..2 add one vector each over the batch.  Sums: the MFMA's k-ordered
// chain over the records of the chunk, in order.  (The VALU form this replaces: 21 us for 31 744 records; the stamps showed it
// issue-bound with one wave per SIMD.)
constexpr int SBF_GEMM_BATCH = 32, SBF_GEMM_RS = 232;
template <int DM, int HM>
__global__ __launch_bounds__(SBF_NT) void k_sde_bwd_hist_gemm(SdeBwdFusedArgs a) {
  static_assert(DM == 32 && HM == 64, "tile assignment");
  using R = SbfR<DM, HM>;
  constexpr int NBATCH = SBF_GEMM_BATCH, Q = R::HREC / 4, RS = SBF_GEMM_RS;   // records per batch; 16-byte quads per record; LDS stride of a record
  extern __shared__ __attribute__((aligned(16))) float bat[];   // [NBATCH][RS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = a.D, H = a.H, Ptot = a.Ptot;
  SBF_STAMP(1, 0);
  // tiles: job t < 8: dW1 tile (mt = t >> 1 of H, nt = t & 1 of D); 8 <= t < 16: dW2 tile (mt = (t - 8) >> 2 of D, nt = (t - 8) & 3 of H);
  // 16 <= t < 20: dWg tile (mt, nt of D).  Wave w owns jobs w, w + 4, ..., w + 16: two of dW1, two of dW2, one of dWg.
  int offA[5], offB[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int t = sw + 4 * q;
    if (t < 8) { offA[q] = R::DPRE + 16 * (t >> 1); offB[q] = R::X + 16 * (t & 1); }
    else if (t < 16) { offA[q] = R::LAM + 16 * ((t - 8) >> 2); offB[q] = R::HV + 16 * ((t - 8) & 3); }
    else { offA[q] = R::LAMG + 16 * ((t - 16) >> 1); offB[q] = R::X + 16 * ((t - 16) & 1); }
  }
  f32x4 acc[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs0 = 0.f;   // waves 0..2: this lane's entry of b1 (lane h) / b2 / bg (lane d < 32)
  const int per = (a.nrec + (int)gridDim.x - 1) / (int)gridDim.x;
  const int r0 = (int)blockIdx.x * per, r1 = min(a.nrec, r0 + per);
  constexpr int NQ = (NBATCH * Q + SBF_NT - 1) / SBF_NT;
  f32x4 pre[NQ];
  auto fetch = [&](int b0) {
    const int nb = min(NBATCH, r1 - b0);
    const f32x4* src = reinterpret_cast<const f32x4*>(a.hist + (size_t)b0 * R::HREC);
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int i = tid + q * SBF_NT; pre[q] = i < nb * Q ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
  };
  if (r0 < r1) fetch(r0);
  const int kl = lane >> 4, il = lane & 15;   // this lane's record of a k-step and its row / column of a fragment
  const int ld = lane < DM ? lane : 0;
  for (int b0 = r0; b0 < r1; b0 += NBATCH) {
    __syncthreads();
    // quad i of the batch = record i / Q, quad i % Q; records beyond nb are written as zeros (they add nothing)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int i = tid + q * SBF_NT;
      if (i < NBATCH * Q) { const int rr = i / Q, qq = i - rr * Q; *reinterpret_cast<f32x4*>(bat + rr * RS + 4 * qq) = pre[q]; }
    }
    __syncthreads();
    if (b0 + NBATCH < r1) fetch(b0 + NBATCH);
#pragma unroll 2
    for (int k0 = 0; k0 < NBATCH; k0 += 4) {
      const float* rec = bat + (k0 + kl) * RS + il;
#pragma unroll
      for (int q = 0; q < 5; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(rec[offA[q]], rec[offB[q]], acc[q], 0, 0, 0);
    }
    if (sw < 3) {   // the biases: wave 0 adds dpre (b1), wave 1 lam (b2), wave 2 lam_g (bg) over the batch, eight reads in flight
      const float* col = bat + (sw == 0 ? R::DPRE + lane : (sw == 1 ? R::LAM + ld : R::LAMG + ld));
      for (int j = 0; j < NBATCH; j += 8) {   // (records beyond nb are zeros)
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = col[(j + q) * RS];
#pragma unroll
        for (int q = 0; q < 8; ++q) bs0 = bs0 + v[q];
      }
    }
  }
  SBF_STAMP(1, 1);
  // C fragment of a tile: rows 16 mt + 4 (lane >> 4) + r, column 16 nt + (lane & 15); flat layouts are column-major
  float* pp = a.part + (size_t)blockIdx.x * Ptot;
  const int n1 = H * D, n2 = n1 + H, n3 = n2 + D * H, n4 = a.Pf, n5 = n4 + D * D;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int t = sw + 4 * q;
    int base, ldm, nrow, ncol, mt, nt;
    if (t < 8) { base = 0; ldm = H; nrow = H; ncol = D; mt = t >> 1; nt = t & 1; }
    else if (t < 16) { base = n2; ldm = D; nrow = D; ncol = H; mt = (t - 8) >> 2; nt = (t - 8) & 3; }
    else { base = n4; ldm = D; nrow = D; ncol = D; mt = (t - 16) >> 1; nt = (t - 16) & 1; }
    const int col = 16 * nt + il, row0 = 16 * mt + 4 * kl;
    if (col < ncol) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (row0 + r < nrow) pp[base + (row0 + r) + ldm * col] = acc[q][r];
    }
  }
  if (sw == 0 && lane < H) pp[n1 + lane] = bs0;
  if (sw == 1 && lane < D) pp[n3 + lane] = bs0;
  if (sw == 2 && lane < D) pp[n5 + lane] = bs0;
  SBF_STAMP(1, 2);
  SBF_STAMP(1, 3);
}

// k_sde_eh_reg_fused in that form.  Evaluation points: 0 = (K for the drift, utilde for the diffusion), 1 = tmp, 2 = uprev.
// DEFER: the step's cotangents are seeded with w_reg and its records join the sweep's history (four per sample: evaluation point 0
// leaves two, one for the drift at K and one for the diffusion at utilde — a record has ONE x), so that the sweep's GEMM and
// reduction cover the regulariser too: no closing sum here, no second reduction.
template <int DM, int HM, bool DEFER = false>
__global__ __launch_bounds__(SBF_NT) void k_sde_eh_reg_fused_r(SdeBwdFusedArgs a) {
  using R = SbfR<DM, HM>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  SBF_STAMP(0, 0);
  R c;
  c.setup(a, sm, DEFER ? 4 : 3);
  SBF_STAMP(0, 1);
  const bool row = c.row;
  const size_t g = c.g;
  float* V0 = c.block(0); float* V1 = c.block(1); float* V2 = c.block(2);
  if (c.valid) {
    const float u = row ? a.u1[g] : 0.f, dW = row ? a.dW1[g] : 0.f, un = row ? a.un1[g] : 0.f;
    const float dt = a.dt1, sqdt = sqrtf(dt), hdt = dt / 2.0f, nf = (float)((size_t)a.B * a.D);
    c.put(V2, R::X, u);
    float hvu, dau, hvk, dak, hvt, dat;
    c.hidden(V2, hvu, dau);
    const float du1 = c.f_out(V2);
    const float L = c.g_out(V2 + R::X);
    const float ut = row ? u + L * sqdt : 0.f;            // :196
    const float K = row ? u + dt * du1 : 0.f;             // :175
    const float tmp = row ? K + L * dW : 0.f;
    c.put(V0, R::X, K); c.put(V0, R::XG, ut);
    c.hidden(V0, hvk, dak);
    const float du2 = c.f_out(V0);
    const float g3 = c.g_out(V0 + R::XG);
    // seeds (k_sder_seed)
    float du2b = 0.f, du1b0 = 0.f, g3b = 0.f, Lb0 = 0.f, unb = 0.f;
    if (row) {
      const float Ed = (dt * (du2 - du1)) / 2.0f;
      const float ggp = (g3 - L) / sqdt;
      const float w2 = dW * dW;
      const float En = (ggp * w2) / 2.0f;
      const float sc = a.abstol + fmaxf_(__builtin_fabsf(u), __builtin_fabsf(un)) * a.reltol;
      const float num = a.delta * Ed + En;
      const float r = num / sc;
      float rb = (a.eest > 0.f) ? dt * r / (nf * a.eest) : 0.f;   // reg = dt * sqrt(mean r^2)
      if constexpr (DEFER) rb = a.w_reg * rb;
      const float numb = rb / sc;
      const float scb = -rb * num / (sc * sc);
      unb = (__builtin_fabsf(un) > __builtin_fabsf(u)) ? scb * a.reltol * (un >= 0.f ? 1.f : -1.f) : 0.f;
      const float Edb = a.delta * numb, ggpb = numb * w2 * 0.5f;
      du2b = hdt * Edb; du1b0 = -hdt * Edb;
      g3b = ggpb / sqdt; Lb0 = -ggpb / sqdt;
    }
    // f at K with du2b, g at utilde with g3b
    c.put(V0, R::LAM, du2b); c.put(V0, R::LAMG, g3b);
    float dprek, dpret, dpreu;
    const float Kb = c.drift_vjp(V0, dak, dprek);
    const float utb = c.wgt_x(V0 + R::LAMG);
    // u_new's cotangent through f, g at tmp
    const float fb2 = hdt * unb, gb2v = (0.5f * dW) * unb;
    c.put(V1, R::X, tmp); c.put(V1, R::LAM, fb2); c.put(V1, R::LAMG, gb2v);
    c.hidden(V1, hvt, dat);
    const float dtf = c.drift_vjp(V1, dat, dpret);
    const float dtg = c.wgt_x(V1 + R::LAMG);
    const float tb = dtf + dtg;
    const float du1b = ((du1b0 + dt * Kb) + hdt * unb) + dt * tb;
    const float Lb = ((Lb0 + sqdt * utb) + (0.5f * dW) * unb) + dW * tb;
    // f, g at uprev: only their parameter cotangents count
    c.put(V2, R::LAM, du1b); c.put(V2, R::LAMG, Lb);
    (void)c.drift_vjp(V2, dau, dpreu);
    if constexpr (DEFER) {
      float* V3 = c.block(3);   // (zeroed by setup) the diffusion's record of evaluation point 0: x = utilde, lam_g = g3b
      c.put(V3, R::X, ut); c.put(V3, R::LAMG, g3b);
      c.put(V0, R::LAMG, 0.f);  // ... and the drift's: x = K, no lam_g (every product that reads it is done)
      float* hr = a.hist + ((size_t)a.rec0 + (size_t)c.b * 4) * R::HREC;
      c.store_hist(hr, V0); c.store_hist(hr + R::HREC, V3); c.store_hist(hr + 2 * R::HREC, V1); c.store_hist(hr + 3 * R::HREC, V2);
    } else {
      c.accumulate(V0, dprek, hvk, ut, du2b, g3b);
      c.accumulate(V1, dpret, hvt, tmp, fb2, gb2v);
      c.accumulate(V2, dpreu, hvu, u, du1b, Lb);
    }
  }
  SBF_STAMP(0, 2);
  if constexpr (!DEFER) c.finish(a.part);
  SBF_STAMP(0, 3);
}

// d(EEst*dt)/dp of the local step (the arithmetic of k_sder_seed / k_sdeb_seed / k_sder_join and of the six products of
// lrnde_sde_euler_heun_reg_grad), one launch.  Evaluation points: 0 = (K for the drift, utilde for the diffusion), 1 = tmp,
// 2 = uprev.  uprev, dW, dt and the step's end state are constants of the tape.
__global__ __launch_bounds__(SBF_NT) void k_sde_eh_reg_fused(SdeBwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  SbfDev c;
  c.setup(a, sm, 3);
  __syncthreads();
  c.ones();
  const SbfOff o = c.o;
  const int lane = c.lane;
  const bool live = c.valid && c.row;
  const size_t g = c.g;
  float* V0 = c.block(0); float* V1 = c.block(1); float* V2 = c.block(2);
  const float u = live ? a.u1[g] : 0.f, dW = live ? a.dW1[g] : 0.f, un = live ? a.un1[g] : 0.f;
  const float dt = a.dt1, sqdt = sqrtf(dt), hdt = dt / 2.0f, nf = (float)((size_t)a.B * a.D);
  if (live) { V2[o.X + lane] = u; V2[o.XG + lane] = u; }
  float aua, aub, aka, akb, ata, atb;
  c.hidden(V2, aua, aub);
  const float du1 = c.f_out(V2);
  const float L = c.g_out(V2 + o.XG);
  const float ut = live ? u + L * sqdt : 0.f;            // :196
  const float K = live ? u + dt * du1 : 0.f;             // :175
  const float tmp = live ? K + L * dW : 0.f;
  if (live) { V0[o.X + lane] = K; V0[o.XG + lane] = ut; }
  c.hidden(V0, aka, akb);
  const float du2 = c.f_out(V0);
  const float g3 = c.g_out(V0 + o.XG);
  // seeds (k_sder_seed)
  float du2b = 0.f, du1b0 = 0.f, g3b = 0.f, Lb0 = 0.f, unb = 0.f;
  if (live) {
    const float Ed = (dt * (du2 - du1)) / 2.0f;
    const float ggp = (g3 - L) / sqdt;
    const float w2 = dW * dW;
    const float En = (ggp * w2) / 2.0f;
    const float sc = a.abstol + fmaxf_(__builtin_fabsf(u), __builtin_fabsf(un)) * a.reltol;
    const float num = a.delta * Ed + En;
    const float r = num / sc;
    const float rb = (a.eest > 0.f) ? dt * r / (nf * a.eest) : 0.f;   // reg = dt * sqrt(mean r^2)
    const float numb = rb / sc;
    const float scb = -rb * num / (sc * sc);
    unb = (__builtin_fabsf(un) > __builtin_fabsf(u)) ? scb * a.reltol * (un >= 0.f ? 1.f : -1.f) : 0.f;
    const float Edb = a.delta * numb, ggpb = numb * w2 * 0.5f;
    du2b = hdt * Edb; du1b0 = -hdt * Edb;
    g3b = ggpb / sqdt; Lb0 = -ggpb / sqdt;
  }
  // f at K with du2b, g at utilde with g3b
  if (live) { V0[o.LAM + lane] = du2b; V0[o.LAMG + lane] = g3b; }
  const float Kb = c.drift_vjp(V0, aka, akb);
  const float utb = c.wgt_x(V0 + o.LAMG);
  // u_new's cotangent through f, g at tmp
  const float fb2 = hdt * unb, gb2 = (0.5f * dW) * unb;
  if (live) { V1[o.X + lane] = tmp; V1[o.XG + lane] = tmp; V1[o.LAM + lane] = fb2; V1[o.LAMG + lane] = gb2; }
  c.hidden(V1, ata, atb);
  const float dtf = c.drift_vjp(V1, ata, atb);
  const float dtg = c.wgt_x(V1 + o.LAMG);
  const float tb = dtf + dtg;
  const float du1b = ((du1b0 + dt * Kb) + hdt * unb) + dt * tb;
  const float Lb = ((Lb0 + sqdt * utb) + (0.5f * dW) * unb) + dW * tb;
  // f, g at uprev: only their parameter cotangents count
  if (live) { V2[o.LAM + lane] = du1b; V2[o.LAMG + lane] = Lb; }
  (void)c.drift_vjp(V2, aua, aub);
  __syncthreads();
  c.accumulate();
  c.store_partial(a.part);
}

// dp (+)= scale * sum over the workgroups' partials; the diffusion part is [vec(Wg); bg] with bg present or not.  A block takes
// 32 entries, eight threads per entry add an eighth of the workgroups each (in workgroup order, loads unrolled), and the
// eight sums are added in slice order: the association is fixed by nwg alone, never by scheduling.
__global__ __launch_bounds__(256) void k_sde_bwd_reduce(const float* part, int nwg, int Ptot, int Pf, int Pg, float* dp_drift, float* dp_diff, float scale, int add) {
  __shared__ float red[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  const int cnt = (nwg + 7) >> 3;
  const int w0 = sl * cnt, w1 = min(nwg, w0 + cnt);
  float s = 0.f;
  if (e < Ptot) {
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = part[(size_t)(w + q) * Ptot + e];
#pragma unroll
      for (int q = 0; q < 8; ++q) s = s + v[q];
    }
    for (; w < w1; ++w) s = s + part[(size_t)w * Ptot + e];
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl != 0 || e >= Ptot) return;
  float t = red[0][el];
#pragma unroll
  for (int q = 1; q < 8; ++q) t = t + red[q][el];
  float* dst = e < Pf ? dp_drift + e : (e - Pf < Pg ? dp_diff + (e - Pf) : nullptr);
  if (dst) *dst = add ? *dst + scale * t : scale * t;
}

}  // namespace
