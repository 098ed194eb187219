/*
 * lrnde_hooks.h — entry points of liblrnde that are NOT part of the drop-in boundary (include/lrnde.h): the timing
 * hooks bench.py uses for its roofline leg, and the in-process local communicator that lets the batch-sharded
 * (nranks > 1) code of the library run inside one process — several handles, one host thread each, on ONE GPU —
 * where RCCL cannot (it refuses two ranks on one device).  Nothing in the reference corresponds to
 * these (it has neither a benchmark harness nor a collective: SURVEY.md §2 rows 16-18).
 */
#ifndef LRNDE_HOOKS_H
#define LRNDE_HOOKS_H

#include "lrnde.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- in-process local communicator ----
 * lrnde_local_comm_create(nranks) makes the rendezvous object; every participating handle joins it with
 * lrnde_comm_init_local(ctx, lc, rank) (instead of lrnde_comm_init) and is then driven by ITS OWN host thread: the
 * sharded entry points (lrnde_solve, lrnde_node_forward*, lrnde_node_backward*, lrnde_vjp, lrnde_classifier_ce ...)
 * must be called by all ranks concurrently, exactly as one process per GPU would call them.  Each collective is a
 * SUM all-reduce done by stream-ordered kernels and events (rank-order sum, identical on every rank); a rank that
 * does not arrive within 60 s breaks the communicator and every pending call returns LRNDE_NCCL_ERROR.  Handles
 * must leave the communicator (lrnde_comm_destroy or lrnde_destroy) before lrnde_local_comm_destroy. */
typedef struct lrnde_local_comm lrnde_local_comm;
int lrnde_local_comm_create(lrnde_local_comm** out, int32_t nranks);
int lrnde_local_comm_destroy(lrnde_local_comm* lc);
int lrnde_comm_init_local(lrnde_ctx* ctx, lrnde_local_comm* lc, int32_t rank);

/* ---- timing hooks (bench.py) ---- */
/* `reps` back-to-back launches of the full Tsit5 step kernel on fixed (uprev, k1, t, dt), timed
 * with HIP events on the handle's stream; avg_us_host = microseconds per launch (roofline leg). */
int lrnde_bench_step(lrnde_ctx* ctx, const float* uprev, const float* k1, int32_t B, float t, float dt,
                     float abstol, float reltol, int32_t reps, float* avg_us_host);
/* `reps` back-to-back exchanges of the per-step kind (the SUM all-reduce of the error norm's partial sums, the buffers
 * and count of a solve at batch B) between two HIP events on the handle's stream: microseconds per exchange.  Collective
 * (every rank calls it); 0 for an unsharded handle.  bench.py --gpus N reports it as allreduce_us_per_step. */
int lrnde_bench_exchange(lrnde_ctx* ctx, int32_t B, int32_t reps, float* avg_us_host);
/* HIP events on the handle's stream around the kernels of the last solve (ms), and its step-kernel launches.  The first call
 * ARMS the clock (solves record their two events from then on — each a marker packet in the queue — and this call returns
 * 0 ms): call it once, run a solve, call it again. */
int lrnde_last_solve_kernel_ms(lrnde_ctx* ctx, float* total_ms_host, int32_t* step_launches_host);
/* Diagnostic: 0 = the solve loop does not use the per-launch reports in pinned host memory and steers by polled copies of the
 * control block instead — its fall-back for a platform where no report arrives.  Same results
 * (tests/test_gpu_overlap.py::test_solve_loop_without_reports_gives_the_same_bits). */
int lrnde_set_reports(lrnde_ctx* ctx, int32_t on);
/* Diagnostic switches, process-wide: `name` is the switch's environment-variable name (LRNDE_NO_QTILE,
 * LRNDE_QTILE_MAX_B, LRNDE_NO_FUSE, LRNDE_DENSE_COPY, LRNDE_NO_OVERLAP, LRNDE_NO_SDE_FAST, LRNDE_SDE_HOST_LOOP, LRNDE_NO_QVJP,
 * LRNDE_ADJ_ERR_ONE_LAUNCH, LRNDE_ADJ_MU_FOLD, LRNDE_ADJ_OVERLAP, LRNDE_ADJ_HOST, LRNDE_VJP_QCOLS, LRNDE_PGRAD_TS, LRNDE_ADJ_NO_REUSE, LRNDE_NO_SDE_BWD_FUSED, LRNDE_SDE_NO_PERSIST, LRNDE_SDE_COOP_LAUNCH, LRNDE_SDE_PERSIST_STALL, LRNDE_SDE_HOST_INITDT, LRNDE_SDE_BWD_LDSACC, LRNDE_SDE_BWD_NO_DEFER, LRNDE_SDE_BWD_NO_RESIDENT, LRNDE_SDE_NO_MARCH, LRNDE_FEED_T / _E / _M, LRNDE_GATHER_TILES, LRNDE_FORCE_COMM); the environment gives
 * the initial value, this call overrides it from then on (handles created earlier pick it up at their next call, except
 * the two communicator switches, which are read when a communicator is made).  Each selects an alternative path that
 * must give the default path's bits: tests/test_gpu_switches.py runs every one of them in the GPU suite.
 * LRNDE_BADARG for an unknown name. */
int lrnde_set_option(const char* name, int32_t value);
/* Diagnostic: the adjoint solves of the following backward calls write one row per ATTEMPTED step — (s, dt) in reversed time
 * s = -t, the attempt's error estimate, accepted 1/0 — into rows_host (host memory of the caller, cap rows; cap = 0 switches
 * the trace off); lrnde_adjoint_trace_rows returns how many were written since the last lrnde_set_adjoint_trace.  This is how
 * tests/test_gpu_backward.py holds the adjoint's controller to the oracle's step sequence attempt by attempt. */
int lrnde_set_adjoint_trace(lrnde_ctx* ctx, lrnde_trace_row* rows_host, int32_t cap);
int lrnde_adjoint_trace_rows(lrnde_ctx* ctx, int32_t* n_host);
/* Diagnostic: mean host-side microseconds per lrnde_node_forward call since the last reset, by phase: [0] entry -> the main
 * solve's init launches enqueued, [1] -> its last report read (the feed loop: the GPU is busy throughout), [2] -> the final
 * synchronisation returned, [3] -> the call returned (local-step results, bookkeeping).  tools/bench/host_phases.py */
int lrnde_host_phases(lrnde_ctx* ctx, double* us4_host, int32_t reset);
/* Diagnostic: 0 = the layer forward keeps its local step (and the recorded forward's regulariser sweep) in order on the
 * handle's stream instead of its companion stream; 1 = overlap (the default for unsharded handles).
 * Results are the same bits either way — tests/test_gpu_overlap.py holds the library to that. */
int lrnde_set_overlap(lrnde_ctx* ctx, int32_t on);
/* average microseconds of one f-eval (3 conv + 2 batch-norm statistics launches), HIP events */
int lrnde_conv_bench_rhs(lrnde_conv* c, const float* u, float t, int32_t B, int32_t reps, float* us_host);

#ifdef __cplusplus
}
#endif
#endif
