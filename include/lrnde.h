/*
 * lrnde.h — C ABI of liblrnde: the MI355X (gfx950) implementation of the
 * LocalRegNeuralDE.jl adaptive Tsit5 neural-ODE path.
 *
 * The reference has no FFI seam of its own (it is pure Julia); the seam this
 * library replaces is the Julia-level one listed in SURVEY.md §8(b).  Each
 * entry point names the reference interface it stands in for (paths relative
 * to the reference repository).  A Julia maintainer binds these with `ccall`
 * (see INTEGRATION.md); this repo's host mirror binds them with ctypes.
 *
 * Conventions
 *   - opaque handle, one per (device, stream); not thread-safe; distinct
 *     handles are independent;
 *   - every array pointer is a DEVICE pointer (hipMalloc'd / torch.cuda) unless
 *     the name ends in _host; the caller owns all buffers;
 *   - states are fp32, column-major (D x B) "batch last" exactly as the Julia
 *     arrays are laid out, i.e. sample-major contiguous rows of D floats;
 *   - parameters are the flat Lux/ComponentArray vector
 *     [vec(W1) (H x (D+td)); b1 (H); vec(W2) (D x (H+td)); b2 (D)];
 *   - every call returns an lrnde_status (0 = ok) and never throws; the text of
 *     the last error is available from lrnde_last_error();
 *   - calls are stream-ordered on the handle's HIP stream; calls that return
 *     host-side results (stats, scalars) synchronise that stream.
 */
#ifndef LRNDE_H
#define LRNDE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrnde_ctx lrnde_ctx;

typedef enum {
  LRNDE_OK = 0,
  LRNDE_MAXITERS = 1,          /* ReturnCode.MaxIters  (iter > maxiters)            */
  LRNDE_DT_LESS_THAN_MIN = 2,  /* ReturnCode.DtLessThanMin                          */
  LRNDE_DT_NAN = 3,            /* ReturnCode.DtNaN / Unstable (NaN reached the norm) */
  LRNDE_BADARG = 4,            /* ArgumentError (src/utils.jl:53-58 and shape checks) */
  LRNDE_CAPACITY = 5,          /* caller's save buffer too small                     */
  LRNDE_HIP_ERROR = 6,
  LRNDE_NCCL_ERROR = 7,
  LRNDE_UNSUPPORTED = 8
} lrnde_status;

enum { LRNDE_ACT_IDENTITY = 0, LRNDE_ACT_TANH = 1, LRNDE_ACT_GELU = 2 };
enum { LRNDE_REG_ERROR_ESTIMATE = 0, LRNDE_REG_STIFFNESS_ESTIMATE = 1 };
enum { LRNDE_MODE_NONE = 0, LRNDE_MODE_UNBIASED = 1, LRNDE_MODE_BIASED = 2 };

/* Vector field description.  Replaces the Lux model captured by the `dudt`
 * closure, src/layers/neural_ode.jl:45-48, for the field the MNIST experiment
 * builds: TDChain(Dense(D+td => H, act), Dense(H+td => D)),
 * experiments/src/construct.jl:180-189 and src/layers/common.jl:10-40. */
typedef struct {
  int32_t state_dim;   /* D: rows of the state per sample (784 for MNIST)        */
  int32_t hidden_dim;  /* H (100)                                               */
  int32_t time_dep;    /* 1: TDChain appends t as the last input row of each Dense */
  int32_t act;         /* LRNDE_ACT_* applied after the first Dense              */
} lrnde_model_desc;

/* Solver options.  Replaces the kwargs NeuralODE forwards to `solve`,
 * src/layers/neural_ode.jl:10-13,51 and experiments/src/construct.jl:192-197. */
typedef struct {
  float abstol, reltol;
  int32_t maxiters;
  int32_t save_start;      /* Julia kwarg save_start (experiments pass false)     */
  int32_t save_everystep;  /* saveat == [] (the :biased mode)                     */
  int32_t exact_pow;       /* 0: DiffEqBase fastpow in the PI controller; 1: pow  */
} lrnde_solve_opts;

/* sol.destats / retcode / the last attempted step (src/utils.jl:7-9). */
typedef struct {
  int32_t retcode;  /* lrnde_status of the solve itself */
  int32_t nf, naccept, nreject, iters;
  int32_t nsaved;
  float t_final, dt_final, eest_last, dt_init;
} lrnde_stats;

typedef struct { /* one row per attempted step (diagnostics / parity tests) */
  float t, dt, eest;
  int32_t accepted;
} lrnde_trace_row;

/* lifecycle.  `stream` is a hipStream_t (NULL = the null stream). */
int lrnde_create(lrnde_ctx** out, const lrnde_model_desc* desc, int device, void* stream);
int lrnde_destroy(lrnde_ctx* ctx);
const char* lrnde_last_error(const lrnde_ctx* ctx);
size_t lrnde_param_count(const lrnde_model_desc* desc);
const char* lrnde_version(void);

/* Hands the flat parameter vector `ps` (what ODEProblem(dudt, x, tspan, ps)
 * carries, src/layers/neural_ode.jl:50) to the library; repacked on device
 * into the MFMA operand layout.  Call again whenever ps changes. */
int lrnde_set_params(lrnde_ctx* ctx, const float* p, size_t n);

/* The `solver` field of NeuralODE (src/layers/neural_ode.jl:4,51: `solve(prob, n.solver; ...)`) for the choices the
 * experiments offer (experiments/src/construct.jl:154-164 `_ode_solver`): 0 = Tsit5 (default), 1 = VCAB3, 2 = VCABM3.
 * It selects the method of the layer's GLOBAL solve in lrnde_solve / lrnde_node_forward* / lrnde_node_backward*; the
 * local regularisation step stays Tsit5, as in the reference (neural_ode.jl:75,93 build that integrator with Tsit5()).
 * VCAB3 / VCABM3: restated from the published algorithm, UPSTREAM-RECALL (csrc/lrnde_adams.hpp says what is recalled);
 * the backward pass of a recorded Adams forward interpolates its Hermite record and integrates the adjoint with Tsit5. */
enum { LRNDE_ALG_TSIT5 = 0, LRNDE_ALG_VCAB3 = 1, LRNDE_ALG_VCABM3 = 2 };
int lrnde_set_solver(lrnde_ctx* ctx, int32_t alg);

/* du = dudt(u, p, t)  — src/layers/neural_ode.jl:45-48, src/layers/common.jl:10-40. */
int lrnde_rhs(lrnde_ctx* ctx, const float* u, float t, int32_t B, float* du);

/* `init(prob, Tsit5(); ...)` as used by _get_ode_integrator,
 * src/layers/neural_ode.jl:33-38: the automatic initial dt (2 f-evals) and
 * fsalfirst = f(u0, t0).  dt_host receives integrator.dt. */
int lrnde_init_dt(lrnde_ctx* ctx, const float* u0, int32_t B, float t0, float tend, float abstol,
                  float reltol, float* k1, float* dt_host);

/* `_perform_step(integrator, cache::Tsit5ConstantCache, p, Val(reg_type))`,
 * src/perform_step.jl:3-47.  Reads uprev, k1 (= integrator.fsalfirst), t, dt;
 * writes u and k7 (= integrator.fsallast); returns EEst and both regularisation
 * values on the host (reg_error = EEst*dt, :34-38; reg_stiff, :40-47).  The
 * reference's tuple (u, reg_val, 6 + sol.destats.nf, dt), :31, is these plus two
 * constants of the call: the step costs 6 f-evals and does not change dt
 * (julia/LRNDEBackend.jl `_perform_step`; lrnde_node_forward adds the 6 + 3 itself). */
int lrnde_perform_step(lrnde_ctx* ctx, const float* uprev, const float* k1, int32_t B, float t,
                       float dt, float abstol, float reltol, float* u, float* k7,
                       float* eest_host, float* reg_error_host, float* reg_stiff_host);

/* `solve(ODEProblem(dudt, x, tspan, ps), Tsit5(); maxiters, saveat, reltol,
 * abstol, save_start)`, src/layers/neural_ode.jl:50-51.  saveat_host: nsave
 * ascending times (may be NULL/0 with save_everystep).  u_saved: device buffer
 * of cap_saved states; t_saved_host: cap_saved floats (sol.t).  trace_host may be
 * NULL. */
int lrnde_solve(lrnde_ctx* ctx, const float* u0, int32_t B, float t0, float t1,
                const lrnde_solve_opts* opts, const float* saveat_host, int32_t nsave,
                float* u_saved, float* t_saved_host, int32_t cap_saved, lrnde_stats* stats_host,
                lrnde_trace_row* trace_host, int32_t cap_trace);

/* `(n::NeuralODE{regularize, regularize_type})(x, ps, st)`,
 * src/layers/neural_ode.jl:56-100: solve on (t0,t2), pick t1 (mode unbiased:
 * t1_or_rand is t1 itself, drawn by the host RNG as :71; mode biased: a uniform
 * [0,1) draw used as an index into sol.t[1:end-1], :92), fresh init at
 * (sol(t1), t1), one local _perform_step, nfe = sol.destats.nf + 6 + 3.
 * u_end receives sol.u[end] (diffeqsol_to_array, src/utils.jl:37). */
int lrnde_node_forward(lrnde_ctx* ctx, const float* x, int32_t B, float t0, float t2,
                       const lrnde_solve_opts* opts, int32_t mode, int32_t reg_type,
                       float t1_or_rand, float* u_end, float* reg_val_host, int32_t* nfe_host,
                       lrnde_stats* stats_host, float* t1_used_host);

/* Batch sharding across GPUs (one process per GPU).  Not in the reference
 * (SURVEY.md §2 rows 16-17): each rank owns B columns of a global batch of
 * nranks*B; the only exchange is one RCCL all-reduce of the per-tile fp64
 * partial sums of the error norm per attempted step (plus two at init).
 * Backward pass on a sharded handle: the parameter cotangent (gp of lrnde_vjp,
 * dp of lrnde_node_backward*) is all-reduced over the ranks — it is a sum over
 * all samples — and comes out replicated; dx stays sharded; the adjoint's error
 * norm runs over [lambda of all ranks; mu once].
 * unique_id: the 128 bytes of an ncclUniqueId made by lrnde_comm_unique_id on
 * rank 0 and broadcast by the host (torch.distributed). */
int lrnde_comm_unique_id(void* unique_id_128_host);
int lrnde_comm_init(lrnde_ctx* ctx, const void* unique_id_128_host, int32_t rank, int32_t nranks);
int lrnde_comm_destroy(lrnde_ctx* ctx);
/* The size of the communicator the handle actually holds (ncclCommCount; 1 for an unsharded handle) and, optionally, its
 * kind (0 none, 1 RCCL, 2 the in-process local communicator of lrnde_hooks.h): what a launcher checks against the number
 * of ranks it asked for. */
int lrnde_comm_count(lrnde_ctx* ctx, int32_t* nranks_host, int32_t* kind_host);

/* ---- SDE: `_perform_step(integrator, cache::LambaEulerHeunConstantCache, p)`,
 * src/perform_step.jl:172-206 (residual :214-216), as called by NeuralDSDE,
 * src/layers/neural_sde.jl:98,118.  drift: Chain(Dense(D=>H,act), Dense(H=>D)); diffusion:
 * Dense(D=>D) (experiments/src/construct.jl:204-205), diagonal noise; parameters are the two
 * halves of the ComponentArray (p.drift, p.diffusion = [vec(Wg); bg]).  dW (= W.dW) is supplied by
 * the caller, delta is integrator.opts.delta.  Returns u, EEst and the regularisation value
 * EEst*dt on the host.  3 drift + 3 diffusion evaluations per call. */
typedef struct lrnde_sde lrnde_sde;
int lrnde_sde_create(lrnde_sde** out, const lrnde_model_desc* drift, int32_t diffusion_bias, int device,
                     void* stream);
int lrnde_sde_destroy(lrnde_sde* sde);
const char* lrnde_sde_last_error(const lrnde_sde* sde);
int lrnde_sde_set_params(lrnde_sde* sde, const float* p_drift, size_t n_drift, const float* p_diffusion,
                         size_t n_diffusion);
int lrnde_sde_euler_heun_step(lrnde_sde* sde, const float* uprev, const float* dW, int32_t B, float t,
                              float dt, float abstol, float reltol, float delta, float* u,
                              float* eest_host, float* reg_val_host);

/* The same in two halves, for a training step that needs sol.u[end] before it can form du_end
 * (experiments/src/utils.jl:104-115: Zygote.pullback forward, then back(...)): the forward keeps the dense
 * record, the backward consumes it (one backward per record). */
int lrnde_node_forward_record(lrnde_ctx* ctx, const float* x, int32_t B, float t0, float t2,
                              const lrnde_solve_opts* opts, int32_t mode, int32_t reg_type, float t1_or_rand,
                              float* u_end, float* reg_val_host, int32_t* nfe_host, lrnde_stats* stats_host,
                              float* t1_used_host);
int lrnde_node_backward_recorded(lrnde_ctx* ctx, int32_t B, const float* du_end, float w_reg, float* dx, float* dp,
                                 lrnde_stats* stats_bwd_host);

/* The recorded forward with the layer's `saveat` kwarg, and its pullback for cotangents on EVERY state of the returned
 * solution — what the reference's time-series consumers differentiate (diffeqsol_to_timeseries, src/utils.jl:42-46;
 * experiments/src/construct.jl:244-249).  saveat_host: nsave ascending times (nsave = 0: the default of
 * src/layers/neural_ode.jl:102-116).  Mode unbiased appends t1 for the solve and drops every saved time equal to t1 from
 * the result again (_CorrectedDESolution, src/utils.jl:25-33); mode biased draws t1 among the saved times but the last.
 * u_series (device, cap_series x B x D) / t_series_host receive sol.u / sol.t as the caller of the layer sees them,
 * *nseries_host their count.  lrnde_node_backward_recorded_ts: du_series (device, nseries x B x D), one cotangent per
 * state of that series; each enters the reversed-time adjoint solve as an impulse on lambda at its time (SciMLSensitivity's
 * callbacks at the saved times, with the first-same-as-last derivative re-evaluated after each), the one at the end time is
 * lambda's start value. */
int lrnde_node_forward_record_ts(lrnde_ctx* ctx, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* opts,
                                 int32_t mode, int32_t reg_type, float t1_or_rand, const float* saveat_host, int32_t nsave,
                                 float* u_series, float* t_series_host, int32_t cap_series, int32_t* nseries_host,
                                 float* reg_val_host, int32_t* nfe_host, lrnde_stats* stats_host, float* t1_used_host);
int lrnde_node_backward_recorded_ts(lrnde_ctx* ctx, int32_t B, const float* du_series, int32_t nseries, float w_reg,
                                    float* dx, float* dp, lrnde_stats* stats_bwd_host);

/* Classifier head + loss of the MNIST experiments: logits = Dense(D => K)(u) with flat Lux
 * parameters pc = [vec(W) (K x D, column-major); b] (experiments/src/construct.jl:199), loss =
 * logitcrossentropy(logits, onehot(labels)) = mean over the batch (experiments/src/utils.jl:88).  Returns the
 * loss on the host and, where the pointers are non-NULL, logits (B,K), du = d loss / d u (B,D) and
 * dpc = d loss / d pc (device).  labels: device int32 (B), 0-based.  D is the handle's state_dim. */
int lrnde_classifier_ce(lrnde_ctx* ctx, const float* u, int32_t B, const float* pc, int32_t K, const int32_t* labels,
                        float* loss_host, float* logits, float* du, float* dpc);

/* The forward half of the reference's training step in one call (experiments/src/utils.jl:104-115: model forward + loss inside
 * one timed pullback): lrnde_node_forward_record followed by lrnde_classifier_ce on its sol.u[end], same arguments and
 * results as the two calls, same bits.  The head's launches are enqueued when the solve's last report is in, ahead of the
 * solve's final synchronisation, so the call ends with ONE synchronisation and no host round trip between the layer and
 * the head (the two calls: two synchronisations and ~60 us of idle GPU between them). */
int lrnde_node_forward_record_ce(lrnde_ctx* ctx, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* opts,
                                 int32_t mode, int32_t reg_type, float t1_or_rand, float* u_end, float* reg_val_host,
                                 int32_t* nfe_host, lrnde_stats* stats_host, float* t1_used_host, const float* pc, int32_t K,
                                 const int32_t* labels, float* loss_host, float* logits, float* du, float* dpc);

/* ---- conv vector field (experiments/src/construct.jl:213-218) ----
 * node_core = TDChain(Chain(Conv3x3(C+1=>Hc, no bias), BatchNorm(Hc, act)),
 *                     Chain(Conv3x3(Hc+1=>Hc), BatchNorm(Hc, act)), Conv3x3(Hc+1=>C))
 * on a (W x H x C x B) image state in the reference's own (Julia WHCN, w fastest) order — the same
 * (B, W*H*C) sample-major device array as everywhere else in this header.  Flat parameters in
 * Lux/ComponentArray order: conv1.weight (3x3x(C+1)xHc, column-major), bn1.scale, bn1.bias,
 * conv2.weight (3x3x(Hc+1)xHc), bn2.scale, bn2.bias, conv3.weight (3x3x(Hc+1)xC).
 * The t plane is concatenated as the last channel before every conv (src/layers/common.jl:10-45) and
 * is zero padded at the border like any other channel.  bn_train = 1: batch statistics (Lux training
 * mode); 0: the running statistics given to lrnde_conv_set_bn_state (default mean 0 / var 1).
 * compute_dtype LRNDE_F32: fp32 MFMA; LRNDE_BF16: bf16 MFMA operands, fp32 accumulation, fp32 state,
 * stage combination and error norm (BASELINE.json config 4); derivatives (lrnde_conv_vjp, _step_reg_grad, _node_backward) of a
 * bf16 handle are taken in fp32 at the states of its bf16 forward solve ("bf16 forward, fp32 adjoint").  LRNDE_F32_SPLIT: fp32 results for conv2 / conv3 on the fp16
 * MFMA pipe — operands written as hi + lo fp16 pairs (22 significant bits), products hi*hi + hi*lo + lo*hi accumulated in
 * fp32; the f-eval agrees with LRNDE_F32 to ~1e-6 of its scale and runs 1.6x faster, but its rounding errors are less
 * correlated between RK stages, so the embedded error estimate carries more noise when it is far below 1.
 * The solver entry points have the meaning of their lrnde_* namesakes above (same reference lines);
 * the adaptive loop's controller runs on the host for this field (an f-eval is 10^2..10^3 us). */
enum { LRNDE_F32 = 0, LRNDE_BF16 = 1, LRNDE_F32_SPLIT = 2 };
typedef struct {
  int32_t width, height, channels; /* state image W, H, C (CIFAR block: 32, 32, 8).  Supported: C = 8, Hc = 64, W % 4 == 0,
                                    * 4 <= W <= 128 (forward; the backward pass up to W = 124), H >= 2; otherwise
                                    * lrnde_conv_create / the call returns LRNDE_UNSUPPORTED */
  int32_t hidden;                  /* Hc (64) */
  int32_t act;                     /* LRNDE_ACT_*: activation inside the BatchNorm layers (gelu) */
  int32_t bn_train;
  int32_t compute_dtype;
  float bn_eps;                    /* Lux default 1e-5 */
} lrnde_conv_desc;
typedef struct lrnde_conv lrnde_conv;
size_t lrnde_conv_param_count(const lrnde_conv_desc* d);
int lrnde_conv_create(lrnde_conv** out, const lrnde_conv_desc* d, int device, void* stream);
int lrnde_conv_destroy(lrnde_conv* c);
const char* lrnde_conv_last_error(const lrnde_conv* c);
int lrnde_conv_set_params(lrnde_conv* c, const float* p, size_t n);               /* device pointer */
int lrnde_conv_set_bn_state(lrnde_conv* c, const float* mean_var, size_t n);      /* device, [mean1 var1 mean2 var2] */
/* The running statistics (st.model of the layer).  With bn_train = 1 every f-eval of rhs / solve / node_forward
 * advances them as Lux's training-mode BatchNorm does on each call (momentum 0.1, unbiased variance), which is
 * what the reference's dudt closure does to its captured st_ (src/layers/neural_ode.jl:44-48); the backward
 * pass's recomputations do not.  Initial value: mean 0, var 1 (Lux initialstates). */
int lrnde_conv_get_bn_state(lrnde_conv* c, float* mean_var, size_t n);            /* device */
/* Lux.trainmode / Lux.testmode for the field's BatchNorm layers after creation (overrides desc.bn_train) */
int lrnde_conv_set_bn_mode(lrnde_conv* c, int32_t bn_train);
int lrnde_conv_rhs(lrnde_conv* c, const float* u, float t, int32_t B, float* du);
int lrnde_conv_init_dt(lrnde_conv* c, const float* u0, int32_t B, float t0, float t1, float abstol,
                       float reltol, float* k1, float* dt_host);
int lrnde_conv_perform_step(lrnde_conv* c, const float* uprev, const float* k1, int32_t B, float t, float dt,
                            float abstol, float reltol, float* u, float* k7, float* eest_host,
                            float* reg_error_host, float* reg_stiff_host);
int lrnde_conv_solve(lrnde_conv* c, const float* u0, int32_t B, float t0, float t1, const lrnde_solve_opts* o,
                     const float* saveat_host, int32_t nsave, float* u_saved, float* t_saved_host,
                     int32_t cap_saved, lrnde_stats* st, lrnde_trace_row* trace_host, int32_t cap_trace);
int lrnde_conv_node_forward(lrnde_conv* c, const float* x, int32_t B, float t0, float t2,
                            const lrnde_solve_opts* o, int32_t mode, int32_t reg_type, float t1_or_rand,
                            float* u_end, float* reg_val_host, int32_t* nfe_host, lrnde_stats* st,
                            float* t1_used_host);
/* Zygote.pullback(dudt, y, p, t) of the conv field (the adjoint RHS building block, as lrnde_vjp): dy = (df/dy)^T lam,
 * gp (device, flat parameter layout, may be NULL) = (df/dp)^T lam; train-mode BatchNorm is differentiated through its
 * batch statistics.  fp32 compute only. */
int lrnde_conv_vjp(lrnde_conv* c, const float* y, float t, const float* lam, int32_t B, float* dy, float* gp);
/* the conv-field instances of lrnde_step_reg_grad / lrnde_node_backward (same meaning, same reference lines) */
int lrnde_conv_step_reg_grad(lrnde_conv* c, const float* uprev, const float* k1, int32_t B, float t, float dt, float abstol,
                             float reltol, int32_t reg_type, float* gp, float* reg_val_host);
int lrnde_conv_node_backward(lrnde_conv* c, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* o,
                             int32_t mode, int32_t reg_type, float t1_or_rand, const float* du_end, float w_reg,
                             float* dx, float* dp, lrnde_stats* st_fwd, lrnde_stats* st_bwd);
/* The same in two calls, for a training step that runs the forward once (experiments/src/utils.jl:104-123):
 * lrnde_conv_node_forward_record = lrnde_conv_node_forward that keeps the dense record of the main solve and u(t1);
 * lrnde_conv_node_backward_recorded = the backward pass from that record (invalidated by any later solve on the handle). */
int lrnde_conv_node_forward_record(lrnde_conv* c, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* o,
                                   int32_t mode, int32_t reg_type, float t1_or_rand, float* u_end, float* reg_val_host,
                                   int32_t* nfe_host, lrnde_stats* st, float* t1_used_host);
int lrnde_conv_node_backward_recorded(lrnde_conv* c, int32_t B, const float* du_end, float w_reg, float* dx, float* dp,
                                      lrnde_stats* st_bwd);
/* ---- layers around the CIFAR10 NeuralODE (experiments/src/construct.jl:224-227) ----
 * stem: AugmenterLayer(Conv((3,3), 3=>5; pad=1), 3) (src/layers/common.jl:80-92: cat(x, conv(x); dims=3)) + BatchNorm(8);
 * ps (device, 156) = [conv.weight 3x3x3x5 column-major; conv.bias 5; bn.scale 8; bn.bias 8]; x (B,3,H,W) -> u0 (B,8,H,W).
 * BatchNorm(8) follows the handle's bn_train; in test mode bn_state (device, [mean 8; var 8], may be NULL = 0/1) is used.
 * In training mode bn_state_out (device, 16 floats, may be NULL) receives the running statistics Lux's BatchNorm
 * returns in its state: bn_state advanced by this batch (momentum 0.1, n/(n-1) variance correction, as the field's
 * layers, lrnde_conv_get_bn_state); in test mode it receives a copy of bn_state.
 * head: Chain(Conv((3,3), 8=>1, gelu; pad=1), FlattenLayer(), Dense(H*W=>K)) + logitcrossentropy
 * (experiments/src/utils.jl:88); ph (device) = [conv.weight 3x3x8x1; conv.bias 1; dense.weight K x H*W column-major;
 * dense.bias K]; returns the mean loss on the host and, where non-NULL, logits (B,K), du (B,8,H,W), dph.
 * H, W are the handle's image size.  Run once per batch: simple direct kernels. */
size_t lrnde_cifar_stem_param_count(void);
size_t lrnde_cifar_head_param_count(int32_t H, int32_t W, int32_t K);
int lrnde_cifar_stem_forward(lrnde_conv* c, const float* x, int32_t B, const float* ps, const float* bn_state, float* u0,
                             float* bn_state_out);
int lrnde_cifar_stem_backward(lrnde_conv* c, const float* x, int32_t B, const float* ps, const float* bn_state,
                              const float* du0, float* dps);
int lrnde_cifar_head_ce(lrnde_conv* c, const float* u, int32_t B, const float* ph, int32_t K, const int32_t* labels,
                        float* loss_host, float* logits, float* du, float* dph);

/* `_perform_step(integrator, cache::RKMilCommuteConstantCache, p)`, src/perform_step.jl:108-170, diagonal noise,
 * Ito interpretation: u = K + L*dW + Dgj*J with J = dW^2/2 - |dt|/2, EEst from the 4-argument
 * _calculate_residuals (:218-220); returns u, EEst and EEst*dt.  The reference's du2/En are dead code there
 * (their `tmp` is overwritten at :166) and are not evaluated: 1 drift + 2 diffusion evaluations. */
int lrnde_sde_rkmil_step(lrnde_sde* sde, const float* uprev, const float* dW, int32_t B, float t, float dt,
                         float abstol, float reltol, float* u, float* eest_host, float* reg_val_host);

/* nsteps steps of size dt on a fixed grid, step i from t0 + i*dt with the increments dW[i] (device, nsteps x B x D):
 * the loop a NeuralDSDE forward (src/layers/neural_sde.jl:56-86) runs with a fixed-step solver, without a host round
 * trip per step (Euler-Heun at the one-launch step's shape: ONE launch marches the whole grid).  which: 0 Euler-Heun (src/perform_step.jl:172-206, delta used), 1 Milstein (:108-170).
 * u_traj (device, nsteps x B x D): every step's u; eest_host / reg_val_host (host, nsteps, may be NULL): every step's
 * EEst and EEst*dt.  Step i is bit-identical to the corresponding single-step call. */
int lrnde_sde_solve_fixed(lrnde_sde* sde, int32_t which, const float* u0, const float* dW, int32_t B, float t0, float dt,
                          int32_t nsteps, float abstol, float reltol, float delta, float* u_traj, float* eest_host,
                          float* reg_val_host);

/* Adaptive Euler-Heun solve: the loop that consumes the step's error estimate (src/perform_step.jl:200-205), which the
 * reference reaches through `solve(prob, solver; ...)` in src/layers/neural_sde.jl:60-72 (StochasticDiffEq, un-vendored).
 * The Brownian path is the CALLER's: W (device, (nfine+1) x B x D) holds it on a uniform grid of nfine intervals over
 * (t0, t1), W[0] = 0.  Steps are whole numbers of grid intervals, so every increment dW = W[j] - W[i] is the path's own
 * and a rejected step is retried over a shorter piece of the SAME path — what StochasticDiffEq's rejection sampling with
 * memory guarantees in distribution, made exact by fixing the path up front.  Step-size control: the PI form of the ODE
 * controller on EEst; StochasticDiffEq's own constants could not be read in this image, so they are
 * options (suggested: gamma 0.9, qmin 0.2, qmax 1.125, beta1 0.14, beta2 0.08).  u_end: the state at t1; the trace
 * (may be NULL) gets one row per attempted step.  On the MNIST-SDE shape (state 32, hidden 64, no time input, unsharded)
 * the controller runs on the device — in the footer of the one-launch step kernel; the host keeps launches enqueued and
 * watches a pinned progress word (12 us per attempted step) — elsewhere the loop is host-controlled, one stream sync per
 * attempted step. */
typedef struct {
  float abstol, reltol, delta;   /* integrator.opts.abstol / reltol / delta */
  float dt0;                     /* first step (rounded down to whole grid intervals, at least one) */
  float gamma, qmin, qmax, beta1, beta2;
  int32_t maxiters;
} lrnde_sde_adapt_opts;
int lrnde_sde_solve_adaptive(lrnde_sde* sde, const float* u0, const float* W, int32_t nfine, int32_t B, float t0, float t1,
                             const lrnde_sde_adapt_opts* opts, float* u_end, lrnde_stats* stats_host,
                             lrnde_trace_row* trace_host, int32_t cap_trace);

/* Gradient path of the NeuralDSDE layer.  The reference differentiates the SDE solve with TrackerAdjoint — a tape of the
 * solver's own arithmetic (src/layers/neural_sde.jl:12; test/runtests.jl:361-365,386-397) — and reg_val w.r.t. the
 * parameters only (the local step's integrator is built under CRC.@non_differentiable, neural_sde.jl:42).
 * lrnde_sde_solve_fixed_backward: pullback of lrnde_sde_solve_fixed (which = 0, Euler-Heun) for
 * loss = <du_end, u_traj[nsteps-1]>: the reverse sweep through the nsteps steps with the same increments dW
 * (discretise-then-differentiate); dx (B x D), dp_drift (flat drift parameters), dp_diff ([vec(Wg); bg]): device.
 * lrnde_sde_euler_heun_reg_grad: d (EEst*dt) / d (p_drift, p_diffusion) of one local Euler-Heun step
 * (src/perform_step.jl:172-206) with uprev, dW, dt constant; reg_val_host receives EEst*dt. */
int lrnde_sde_solve_fixed_backward(lrnde_sde* sde, const float* u0, const float* u_traj, const float* dW, int32_t B, float t0,
                                   float dt, int32_t nsteps, const float* du_end, float* dx, float* dp_drift, float* dp_diff);
int lrnde_sde_euler_heun_reg_grad(lrnde_sde* sde, const float* uprev, const float* dW, int32_t B, float t, float dt, float abstol,
                                  float reltol, float delta, float* dp_drift, float* dp_diff, float* reg_val_host);
/* The same two for the Milstein step (src/perform_step.jl:108-170; `solver = RKMilCommute()`): pullback of
 * lrnde_sde_solve_fixed(which = 1) and d (EEst*dt) / d parameters of one local step (EEst from the 4-argument residual). */
int lrnde_sde_solve_fixed_backward_rkmil(lrnde_sde* sde, const float* u0, const float* u_traj, const float* dW, int32_t B, float t0,
                                         float dt, int32_t nsteps, const float* du_end, float* dx, float* dp_drift, float* dp_diff);
int lrnde_sde_rkmil_reg_grad(lrnde_sde* sde, const float* uprev, const float* dW, int32_t B, float t, float dt, float abstol,
                             float reltol, float* dp_drift, float* dp_diff, float* reg_val_host);

/* The NeuralDSDE layer itself, forward and pullback: `(n::NeuralDSDE{R})(x, ps, st)` (src/layers/neural_sde.jl:74-123) =
 * `_solve_neuraldsde_generic` (:50-72: ADAPTIVE solve of drift / diffusion from x over tspan with the layer's saveat rules,
 * src/layers/neural_ode.jl:102-116) + for R = unbiased / biased in training mode the local step: a fresh integrator at
 * (sol(t1), t1) over (t1, t2) and one `_perform_step` (:94-98, :116-118; src/perform_step.jl:172-206) whose EEst*dt is
 * reg_val — and what `Zygote.gradient` of the reference's tests takes through it (test/runtests.jl:340-433: TrackerAdjoint
 * tapes the solve's own steps; reg_val w.r.t. the parameters only, neural_sde.jl:42).
 *   forward : x (B x D, device); W: the caller's Brownian path on a uniform grid of nfine intervals over (t0, t2), as for
 *     lrnde_sde_solve_adaptive (device, must stay alive until the backward call); opts: tolerances, delta and the
 *     controller's constants; opts->dt0 <= 0 selects the automatic initial dt (StochasticDiffEq's sde_determine_initdt
 *     restated, UPSTREAM-RECALL), for the main solve (rounded down to whole grid intervals) and for the local step.
 *     mode LRNDE_MODE_* (pass NONE for test mode); t1_or_rand: t1 itself (unbiased; the caller's host RNG draw
 *     rand*(t2-t0)+t0, :92) or the uniform draw r in [0,1) that picks sol.t[floor(r*(len-1))] (biased, :114).
 *     z_local (B x D, device, standard normal): the local step's increment is sqrt(dt_local) * z_local.
 *     save_start: 1 / 0, or -1 for DiffEq's default rule; saveat_host / nsave: the layer's `saveat` kwarg (ascending, may be
 *     empty).  Saved values between step ends come from the SDE solvers' linear interpolant.
 *     u_series (device, cap_series x B x D), t_series_host, nseries_host: sol.u / sol.t as the layer's caller sees them —
 *     after `_CorrectedDESolution` when the layer added t1 to a user saveat (the reference's own method is typed for
 *     ODESolution only, src/utils.jl:31; the behaviour here is the ODE layer's).  sol.u[end] is the last entry.
 *     nfe_drift_host / nfe_diffusion_host: the closures' call counts (:55-65).  stats: the main solve's.
 *   backward: du_series (device, nseries x B x D) = the cotangent of every state of that series; loss = sum_j <du_j, u_j> +
 *     w_reg * reg_val.  dx (B x D), dp_drift (flat drift parameters), dp_diff ([vec(Wg); bg]): device.  The reverse sweep walks
 *     the recorded accepted steps with their own dt and dW. */
int lrnde_sde_node_forward_record(lrnde_sde* sde, const float* x, const float* W, int32_t nfine, int32_t B, float t0, float t2,
                                  const lrnde_sde_adapt_opts* opts, int32_t mode, float t1_or_rand, const float* z_local,
                                  int32_t save_start, const float* saveat_host, int32_t nsave, float* u_series,
                                  float* t_series_host, int32_t cap_series, int32_t* nseries_host, float* reg_val_host,
                                  int32_t* nfe_drift_host, int32_t* nfe_diffusion_host, lrnde_stats* stats_host, float* t1_used_host);
int lrnde_sde_node_backward_recorded(lrnde_sde* sde, int32_t B, const float* du_series, int32_t nseries, float w_reg, float* dx,
                                     float* dp_drift, float* dp_diff);

/* Which recorded forward a handle's record belongs to: the count of successful lrnde_*_forward_record* calls on it (0 = no
 * usable record).  A binding's pullback closure keeps the value it saw after ITS forward and compares before calling
 * lrnde_*_backward_recorded*: a later forward on the same handle (an evaluation pass between forward and pullback, nested
 * AD) has replaced the record, and the backward would silently differentiate the other input.  julia/LRNDELayer.jl re-runs
 * its forward in that case. */
int lrnde_record_generation(lrnde_ctx* ctx, uint64_t* gen_host);
int lrnde_conv_record_generation(lrnde_conv* c, uint64_t* gen_host);
int lrnde_sde_record_generation(lrnde_sde* sde, uint64_t* gen_host);

/* `_perform_step(integrator, cache::FourStageSRIConstantCache, p)`, src/perform_step.jl:49-106 — the step the
 * reference's default SDE solver SOSRI runs — diagonal noise: four drift and four diffusion evaluations, the
 * increments dW and dZ of the caller's noise process (device, B x D each), u, EEst from the 7-argument
 * _calculate_residuals (:214-216) and EEst*dt.  The tableau is NOT part of this library: SOSRI's coefficients live in
 * un-vendored StochasticDiffEq; the caller passes the cache's fields in the order the reference unpacks them (:51-55). */
typedef struct lrnde_sri_tableau {
  float a021, a031, a032, a041, a042, a043, a121, a131, a132, a141, a142, a143;
  float b021, b031, b032, b041, b042, b043, b121, b131, b132, b141, b142, b143;
  float c02, c03, c04, c11, c12, c13, c14, alpha1, alpha2, alpha3, alpha4;
  float beta11, beta12, beta13, beta14, beta21, beta22, beta23, beta24, beta31, beta32, beta33, beta34, beta41, beta42, beta43, beta44;
} lrnde_sri_tableau;
int lrnde_sde_sri_step(lrnde_sde* sde, const lrnde_sri_tableau* tab, const float* uprev, const float* dW, const float* dZ,
                       int32_t B, float t, float dt, float abstol, float reltol, float delta, float* u, float* eest_host,
                       float* reg_val_host);
/* Reverse sweep of ONE four-stage SRI step (src/perform_step.jl:49-106; lrnde_sde_sri_step with the same arguments):
 * loss = <du_new, u'> + w_reg * EEst*dt.  du_new (device, may be NULL = 0): cotangent of the step's result; dx (device, may be
 * NULL): cotangent of uprev — NULL when uprev is a constant of the tape, as for the local step's regulariser
 * (src/layers/neural_sde.jl:42); dp_drift / dp_diff (device): the parameter cotangents are ADDED (zero them first; a solve's
 * pullback calls this once per step, newest first).  dW, dZ, dt are constants.  reg_val_host (may be NULL): EEst*dt. */
int lrnde_sde_sri_step_backward(lrnde_sde* sde, const lrnde_sri_tableau* tab, const float* uprev, const float* dW, const float* dZ,
                                int32_t B, float t, float dt, float abstol, float reltol, float delta, const float* du_new,
                                float w_reg, float* dx, float* dp_drift, float* dp_diff, float* reg_val_host);


/* ---- backward pass ----
 * lrnde_vjp: the vector-Jacobian product Zygote.pullback(dudt, y, p, t) computes inside the adjoint
 * RHS (SciMLSensitivity ZygoteVJP): dy = (df/dy)^T lam, gp = (df/dp)^T lam (flat Lux layout, may be
 * NULL).  All device pointers. */
int lrnde_vjp(lrnde_ctx* ctx, const float* y, float t, const float* lam, int32_t B, float* dy, float* gp);

/* d reg_val / d ps for one local step: Zygote through `_perform_step` with integrator, k1, dt, uprev
 * constant (src/layers/neural_ode.jl:40, src/utils.jl:60; asserted by test/runtests.jl:127-131:
 * no gradient w.r.t. x).  gp: device vector of lrnde_param_count floats. */
int lrnde_step_reg_grad(lrnde_ctx* ctx, const float* uprev, const float* k1, int32_t B, float t, float dt,
                        float abstol, float reltol, int32_t reg_type, float* gp, float* reg_val_host);

/* Pullback of the NeuralODE layer for  loss = <du_end, sol.u[end]> + w_reg * reg_val
 * (what Zygote.pullback computes in experiments/src/utils.jl:104-115): forward re-solve with a dense
 * record, continuous adjoint of `solve` (SciMLSensitivity InterpolatingAdjoint(autojacvec=ZygoteVJP()):
 * reversed-time adaptive Tsit5 on [lambda; mu], same tolerances, cotangent times as tstops), plus the
 * regulariser's reverse sweep.  dx (B,D) and dp (P) are device outputs. */
int lrnde_node_backward(lrnde_ctx* ctx, const float* x, int32_t B, float t0, float t2,
                        const lrnde_solve_opts* opts, int32_t mode, int32_t reg_type, float t1_or_rand,
                        const float* du_end, float w_reg, float* dx, float* dp, lrnde_stats* stats_fwd_host,
                        lrnde_stats* stats_bwd_host);

/* ---- optimiser update rules of the experiments: experiments/src/construct.jl:104-126 builds
 * Adam / AdamW / AdaMax / Descent / Momentum / Nesterov (Optimisers.jl, un-vendored: rules restated from its documented
 * update formulas) and chains WeightDecay when cfg.weight_decay != 0.  One fused pass over a flat parameter vector:
 * x -= rule(grad) + weight_decay * x.  state1 / state2: the rule's moment vectors (device, n floats, zero-initialised by
 * the caller; unused ones may be NULL); step: 1-based update count (bias correction beta^step); eta: the learning rate
 * the scheduler returned for this step (experiments/src/utils.jl:1-68; host mirror: localregneuralde.jl_amd/optim.py).
 * AdamW(eta) = LRNDE_OPT_ADAM with weight_decay = its decay. */
enum { LRNDE_OPT_DESCENT = 0, LRNDE_OPT_MOMENTUM = 1, LRNDE_OPT_NESTEROV = 2, LRNDE_OPT_ADAM = 3, LRNDE_OPT_ADAMAX = 4 };
int lrnde_opt_update(int32_t kind, float* x, const float* grad, float* state1, float* state2, size_t n, float eta,
                     float rho_or_beta1, float beta2, float eps, int32_t step, float weight_decay, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif
