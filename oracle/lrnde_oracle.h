/*
 * lrnde_oracle.h — CPU restatement ("oracle") of the LocalRegNeuralDE.jl hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product path (localregneuralde.jl_amd/): only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference is Julia and cannot run in
 * this image (no julia binary), its tests hold no numeric golden vectors
 * (test/runtests.jl asserts only type / finite / non-zero), and most of the
 * arithmetic on the path lives in un-vendored, un-pinned dependencies
 * (OrdinaryDiffEq 6, DiffEqBase 6, Lux 0.4/0.5, NNlib 0.8/0.9).  The oracle is
 * therefore pinned by (a) the algebraic identities of the Tsit5 tableau and its
 * dense interpolant, (b) order-of-convergence known-answer tests, (c) an
 * independent numpy restatement and scipy cross-checks, and (d) the behavioural
 * assertions of test/runtests.jl — see tests/ and DESIGN.md.
 *
 * What each function follows (all paths relative to /root/reference):
 *   lro_tsit5_step        src/perform_step.jl:3-32   (stage order, operation order)
 *   lro_reg_error         src/perform_step.jl:34-38, 210-212
 *   lro_reg_stiffness     src/perform_step.jl:40-47
 *   lro_mlp_rhs           src/layers/common.jl:10-40 (TDChain: t appended as the
 *                         last input row of every Dense), src/utils.jl:12-23,
 *                         experiments/src/construct.jl:180-189 (shapes)
 *   lro_init_dt           OrdinaryDiffEq ode_determine_initdt (un-vendored)
 *   lro_solve             OrdinaryDiffEq solve!/loopheader!/loopfooter!/PIController/
 *                         savevalues!/Tsit5 interpolant (un-vendored),
 *                         as called from src/layers/neural_ode.jl:42-54
 *   lro_node_forward      src/layers/neural_ode.jl:56-116 (none / unbiased / biased)
 *   lro_euler_heun_step   src/perform_step.jl:172-206, 214-216
 *   lro_rkmil_step        src/perform_step.jl:108-170 (diagonal noise, Ito), 218-220
 *   lro_conv_rhs          experiments/src/construct.jl:213-218 (the CIFAR10 node_core:
 *                         TDChain(Chain(Conv3x3 C+1=>Hc no-bias, BatchNorm(Hc, act)),
 *                         Chain(Conv Hc+1=>Hc, BatchNorm(Hc, act)), Conv Hc+1=>C)),
 *                         src/layers/common.jl:10-45 (t plane concatenated as the LAST
 *                         channel before every sub-layer, and therefore zero padded at the
 *                         image border like any other channel); Lux Conv = NNlib.conv (true
 *                         convolution: flipped kernel) and Lux BatchNorm (batch statistics
 *                         with the biased variance in training, running statistics in test
 *                         mode, epsilon 1f-5) are un-vendored: UPSTREAM-RECALL.
 *
 * Canonical arithmetic (shared definition with the HIP kernels, so that both
 * produce the same bits): every Dense dot product is a sum of fp32 fma chains
 * over consecutive segments of 112 rows (each chain starts from 0 and runs in
 * increasing k; the segment partials are added left to right — for K <= 112
 * this is one plain chain), then the time column by fma, then "+ bias"; tanh/gelu
 * are the fixed fp32 polynomial/exp forms below; every norm accumulates the
 * fp32 squares in fp64 and rounds the final sqrt to fp32.  The reference does
 * these with OpenBLAS sgemm / Julia Base tanh / Float32 pairwise sums, whose
 * summation orders are unspecified; the differences are O(1e-7) relative, well
 * inside the rtol=1e-5 the north star asks for.
 */
#ifndef LRNDE_ORACLE_H
#define LRNDE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { LRO_ACT_IDENTITY = 0, LRO_ACT_TANH = 1, LRO_ACT_GELU = 2 };
enum { LRO_REG_ERROR_ESTIMATE = 0, LRO_REG_STIFFNESS_ESTIMATE = 1 };
enum { LRO_MODE_NONE = 0, LRO_MODE_UNBIASED = 1, LRO_MODE_BIASED = 2 };
enum {
  LRO_OK = 0,
  LRO_MAXITERS = 1,
  LRO_DT_LESS_THAN_MIN = 2,
  LRO_DT_NAN = 3,
  LRO_BADARG = 4,
  LRO_CAPACITY = 5
};

/* out-of-place vector field du = f(u, t) on a column-major (D x B) state */
typedef void (*lro_field_fn)(void* ctx, const float* u, float t, int B, float* du);
typedef struct {
  lro_field_fn fn;
  void* ctx;
  int D; /* state rows per sample */
} lro_field;

/* 2-layer MLP field: Dense(D+td -> H, act) -> Dense(H+td -> D); flat params in
 * Lux/ComponentArray order [vec(W1) (H x (D+td), column-major); b1; vec(W2) (D x (H+td)); b2] */
typedef struct {
  int D, H, time_dep, act;
  const float* p;
  int nthreads;
} lro_mlp;

/* conv field on a (W x H x C) image state per sample, Julia WHCN order (w fastest), i.e. the
 * (B, W*H*C) sample-major state of everything else here.  Flat params in Lux/ComponentArray
 * order: conv1.weight (3x3x(C+1)xHc, column-major kx,ky,ci,co), bn1.scale (Hc), bn1.bias (Hc),
 * conv2.weight (3x3x(Hc+1)xHc), bn2.scale, bn2.bias, conv3.weight (3x3x(Hc+1)xC).
 * Arithmetic (fp32): each output = one fma chain over (ky, kx, ci) in that nesting order,
 * out-of-image taps skipped; BN statistics in fp64; y = ((x-mean)*inv)*scale + bias; act.
 * Parity of the HIP conv path with this is BY TOLERANCE (the batch statistics couple all
 * samples through sums whose order the GPU does not reproduce), see tests/test_gpu_conv.py. */
typedef struct {
  int W, H, C, Hc, act;
  int bn_train;          /* 1: batch statistics; 0: bn_state = [mean1 var1 mean2 var2] (4*Hc) */
  float eps;
  const float* p;
  const float* bn_state; /* may be NULL in test mode: mean 0, var 1 */
  int nthreads;
  float* bn_run;         /* optional (4*Hc): running [mean1 var1 mean2 var2], advanced by every training-mode
                            lro_conv_rhs call as Lux's BatchNorm does (momentum 0.1, n/(n-1) variance correction) —
                            what the reference's dudt closure does to its captured st_ (src/layers/neural_ode.jl:44-48);
                            lro_conv_vjp's recomputation leaves it alone */
  int bf16;              /* 1: emulate the bf16 compute mode of the HIP path: y1, y2 and the activated
                            conv inputs (the state too) rounded to bf16 (RNE), conv weights (not the t
                            plane's) rounded to bf16, fp32 accumulation, statistics and t-plane term */
} lro_conv;

typedef struct {
  float abstol, reltol;
  int maxiters;
  int save_start;     /* push (t0,u0) first */
  int save_everystep; /* saveat empty: push every accepted step */
  int exact_pow;      /* 0: DiffEqBase fastpow (default); 1: pow() in double */
  int alg;            /* n.solver of the global solve (experiments/src/construct.jl:154-164): 0 Tsit5, 1 VCAB3, 2 VCABM3 */
} lro_opts;
enum { LRO_ALG_TSIT5 = 0, LRO_ALG_VCAB3 = 1, LRO_ALG_VCABM3 = 2 };

typedef struct {
  int retcode;
  int nf, naccept, nreject, iters;
  int nsaved;
  float t_final, dt_final, eest_last, dt_init;
} lro_stats;

typedef struct { /* one row per attempted step */
  float t, dt, eest;
  int accepted;
} lro_trace_row;

/* ---- canonical fp32 math ---- */
float lro_expf(float x);
float lro_tanhf(float x);
float lro_geluf(float x);
float lro_fastlog2(float x);
float lro_fastpow2(float x);
float lro_fastpow(float x, float y);
int lro_tsit5_tableau(double* a, double* c, double* btilde, double* r); /* a[21] c[6] btilde[7] r[28] */

/* ---- vector field ---- */
int lro_mlp_param_count(int D, int H, int time_dep);
void lro_mlp_rhs(const lro_mlp* m, const float* u, float t, int B, float* du);
void lro_mlp_as_field(const lro_mlp* m, lro_field* out);
int lro_conv_param_count(int C, int Hc);
void lro_conv_rhs(const lro_conv* m, const float* u, float t, int B, float* du);
void lro_conv_as_field(const lro_conv* m, lro_field* out);
/* vector-Jacobian product of the conv field at (y, t): dy = (df/dy)^T lam, gp (may be NULL) = (df/dp)^T lam in
 * the flat parameter layout (train-mode BatchNorm differentiated through its batch statistics, as Zygote does
 * through Lux's batchnorm).  fp32 mode only. */
void lro_conv_vjp(const lro_conv* m, const float* y, float t, const float* lam, int B, float* dy, float* gp);
/* the conv-field instances of lro_tsit5_step_reg_grad / lro_node_backward (same drivers, field-agnostic) */
int lro_conv_step_reg_grad(const lro_conv* m, const float* uprev, const float* k1, float t, float dt,
                           float abstol, float reltol, int B, int reg_type, float* gp, float* reg_val);
int lro_conv_node_backward(const lro_conv* m, const float* x, int B, float t0, float t2, const lro_opts* o, int mode,
                           int reg_type, float t1_or_rand, const float* du_end, float w_reg, float* dx, float* dp,
                           lro_stats* st_fwd, lro_stats* st_bwd);

/* ---- step kernels ---- */
/* ks: optional (5*D*B) k2..k6; g6: optional (D*B) */
int lro_tsit5_step(const lro_field* f, const float* uprev, const float* k1, float t, float dt,
                   float abstol, float reltol, int B, float* u, float* k7, float* ks, float* g6,
                   float* eest, float* reg_error, float* reg_stiff);
int lro_tsit5_step_sums(const lro_field* f, const float* uprev, const float* k1, float t, float dt,
                        float abstol, float reltol, int B, float* u, float* k7, float* ks,
                        float* g6, double* sums3);
int lro_init_dt(const lro_field* f, const float* u0, float t0, float tend, float abstol,
                float reltol, int B, float* f0_out, float* dt_out);
void lro_tsit5_interp(float theta, float dt, const float* y0, const float* const k[7], long n,
                      float* out);

/* ---- adaptive solve ---- */
int lro_solve(const lro_field* f, const float* u0, int B, float t0, float t1, const lro_opts* o,
              const float* saveat, int nsave, float* u_saved, float* t_saved, int cap_saved,
              lro_stats* st, lro_trace_row* trace, int cap_trace);

typedef struct { /* dense forward storage: per accepted step t, dt, then [uprev, k1..k7] (8*n floats) */
  int nsteps, cap;
  long n;
  float *t, *dt, *data;
} lro_dense;
void lro_dense_free(lro_dense* d);
void lro_dense_eval(const lro_dense* d, float t, float* out);
int lro_solve_ex(const lro_field* f, const float* u0, int B, float t0, float t1, const lro_opts* o,
                 const float* saveat, int nsave, float* u_saved, float* t_saved, int cap_saved,
                 lro_stats* st, lro_trace_row* trace, int cap_trace, const float* tstops, int ntstops,
                 lro_dense* dense);

/* ---- NeuralODE layer forward (src/layers/neural_ode.jl:56-100) ----
 * mode none/test: saveat=[t1_end]; unbiased: caller passes t1 (host RNG draw);
 * biased: caller passes rand_index in [0,1) used as floor(r*(n-1)) over sol.t[1:end-1].
 * Outputs: u_end (D*B), reg_val, nfe. */
int lro_node_forward(const lro_field* f, const float* x, int B, float t0, float t2,
                     const lro_opts* o, int mode, int reg_type, float t1_or_rand, float* u_end,
                     float* reg_val, int* nfe, lro_stats* st, float* t1_used);

/* ---- backward pass ---- */
void lro_mlp_vjp(const lro_mlp* m, const float* y, float t, const float* lam, int B, float* dy, float* gp);
int lro_tsit5_step_reg_grad(const lro_mlp* m, const float* uprev, const float* k1, float t, float dt,
                            float abstol, float reltol, int B, int reg_type, float* gp, float* reg_val);
int lro_node_backward(const lro_mlp* m, const float* x, int B, float t0, float t2, const lro_opts* o, int mode,
                      int reg_type, float t1_or_rand, const float* du_end, float w_reg, float* dx, float* dp,
                      lro_stats* st_fwd, lro_stats* st_bwd);
int lro_node_backward_traced(const lro_mlp* m, const float* x, int B, float t0, float t2, const lro_opts* o, int mode,
                             int reg_type, float t1_or_rand, const float* du_end, float w_reg, float* dx, float* dp,
                             lro_stats* st_fwd, lro_stats* st_bwd, lro_trace_row* btrace, int cap_btrace);

/* ---- SDE: adaptive Euler-Heun local step with supplied dW (src/perform_step.jl:172-206) ---- */
/* ---- layers around the CIFAR10 NeuralODE (experiments/src/construct.jl:224-227) ----
 * stem: AugmenterLayer(Conv((3,3), 3=>5; pad=1), 3) (src/layers/common.jl:80-92: cat(x, conv(x); dims=3)) then
 * BatchNorm(8) (no activation).  Flat parameters ps = [conv.weight (3x3x3x5 column-major kx,ky,ci,co); conv.bias (5);
 * bn.scale (8); bn.bias (8)].  x: (B,3,H,W).  Forward writes u0 (B,8,H,W); backward takes du0 and returns dps (156).
 * head: Chain(Conv((3,3), 8=>1, gelu; pad=1), FlattenLayer(), Dense(H*W => K)) + logitcrossentropy; ph = [conv.weight
 * (3x3x8x1); conv.bias (1); dense.weight (K x H*W column-major); dense.bias (K)].  Returns the mean loss and, optionally,
 * logits (B,K), du (B,8,H,W) and dph. */
int lro_cifar_stem_param_count(void);
int lro_cifar_head_param_count(int H, int W, int K);
void lro_cifar_stem_forward(const float* x, int B, int H, int W, const float* ps, int bn_train, const float* bn_state, float eps,
                            float* u0, float* bn_state_out /* running statistics after the call, or NULL */);
void lro_cifar_stem_backward(const float* x, int B, int H, int W, const float* ps, int bn_train, const float* bn_state, float eps,
                             const float* du0, float* dps);
float lro_cifar_head_ce(const float* u, int B, int H, int W, const float* ph, int K, const int* labels, float* logits, float* du,
                        float* dph);
int lro_rkmil_step(const lro_field* drift, const lro_field* diffusion, const float* uprev, const float* dW, float t,
                   float dt, float abstol, float reltol, int B, float* u, float* eest, float* reg_val);

/* four-stage SRI step (src/perform_step.jl:49-106, `FourStageSRIConstantCache`: the step SOSRI runs), diagonal noise.
 * The tableau lives in un-vendored StochasticDiffEq (its SOSRI constructor) and is NOT restated here: the caller
 * supplies it, in the order the reference unpacks the cache (:51-55). */
typedef struct lro_sri_tableau {
  float a021, a031, a032, a041, a042, a043, a121, a131, a132, a141, a142, a143;
  float b021, b031, b032, b041, b042, b043, b121, b131, b132, b141, b142, b143;
  float c02, c03, c04, c11, c12, c13, c14, alpha1, alpha2, alpha3, alpha4;
  float beta11, beta12, beta13, beta14, beta21, beta22, beta23, beta24, beta31, beta32, beta33, beta34, beta41, beta42, beta43, beta44;
} lro_sri_tableau;
int lro_sri_step(const lro_field* drift, const lro_field* diffusion, const lro_sri_tableau* tab, const float* uprev,
                 const float* dW, const float* dZ, float t, float dt, float abstol, float reltol, float delta, int B,
                 float* u, float* eest, float* reg_val);
/* classifier head + logitcrossentropy (experiments/src/construct.jl:199, experiments/src/utils.jl:88):
 * pc = [vec(W) (K x D column-major); b]; returns mean CE; optional logits (B,K), du (B,D), dpc (K*(D+1)) */
float lro_classifier_ce(const float* u, int B, int D, const float* pc, int K, const int* labels, float* logits,
                        float* du, float* dpc);
int lro_euler_heun_step(const lro_field* drift, const lro_field* diffusion, const float* uprev,
                        const float* dW, float t, float dt, float abstol, float reltol, float delta,
                        int B, float* u, float* eest, float* reg_val);

#ifdef __cplusplus
}
#endif
#endif
