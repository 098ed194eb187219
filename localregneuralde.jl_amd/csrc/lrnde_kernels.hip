// lrnde_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the adaptive Tsit5
// neural-ODE path of LocalRegNeuralDE.jl, and the C ABI around them.
//
// Reference behaviour (paths relative to the reference repository):
//   src/perform_step.jl:3-47      one Tsit5 step + regularisation values
//   src/layers/common.jl:10-40    TDChain: t appended to the input of every Dense
//   src/layers/neural_ode.jl:33-100  init / solve / local step orchestration
//   OrdinaryDiffEq (un-vendored)  initdt, PI controller, loop header/footer,
//                                 saveat interpolation — SURVEY.md §3.5
//
// Design (DESIGN.md has the long form):
//   * The vector field is per-sample independent, so a workgroup owns a tile of
//     NB = 16 batch columns and runs a WHOLE attempted Tsit5 step for them in one
//     launch: six vector-field evaluations (two fp32-MFMA GEMMs each, weights
//     streamed from L2 in a pre-packed fragment layout, activations in LDS),
//     the stage combinations, the embedded error residual and the
//     regularisation residuals.  No inter-workgroup traffic inside a step.
//   * The only cross-sample coupling is the RMS error norm.  Each workgroup
//     writes one fp64 partial sum; the NEXT launch's prologue (every workgroup
//     redundantly, in the same fixed order) reduces them and runs the PI
//     controller / accept-reject / saveat logic on the device.  The host just
//     enqueues step launches and polls a status word once per chunk.
//   * Sharded batches exchange only those partial sums (one RCCL all-reduce of
//     a zero-padded fp64 vector per attempted step), so every rank takes the
//     same decisions bit for bit.
//   * Numerics are canonical (lrnde_math.hpp): each dot product is one fp32 fma
//     chain in increasing k — exactly what v_mfma_f32_16x16x4_f32 computes —
//     so results are reproducible and equal to the CPU oracle's.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <chrono>
#include <mutex>
#include <string>
#include <functional>
#include <vector>

#include "lrnde.h"
#include "lrnde_hooks.h"
#include "lrnde_math.hpp"
#include "lrnde_comm.hpp"

namespace {

using namespace lrnde;

constexpr int NB = 16;       // batch columns per workgroup tile = MFMA N
constexpr int NW = 8;        // waves per workgroup
constexpr int NT = NW * 64;  // threads per workgroup
constexpr int PSTRIDE = 4;   // doubles per workgroup in a partial-sum vector

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ST_RUNNING = 0, ST_DONE = 100 };                // Ctrl.status: else an lrnde_status error
enum { MODE_SOLVE = 0, MODE_SINGLE_GIVEN_DT = 1, MODE_SINGLE_INIT_DT = 2, MODE_BENCH = 3 };

struct ModelDev {
  int D, H, Dp, Hp, MT1, KG1, MT2, KG2, act, td;
  const f32x4* W1p;  // [MT1][KG1][64] x 4: A fragments of Dense-1, 4 k-steps per load
  const f32x4* W2p;  // [MT2][KG2][64] x 4
  const float* w1t;  // [Hp] time column of W1
  const float* b1;   // [Hp]
  const float* w2t;  // [Dp]
  const float* b2;   // [Dp]
  // 4-column tile family (lrnde_qtile.hpp)
  const float* W1q;  // [RG1][KQ1p][64][4]
  const float* W2q;  // [RG2][KQ2p][64][4]
  int KQ1p, KQ2p, RG1, RG2;
};

#ifdef LRNDE_STAMPS
// Diagnostic build only (tools/probe.hip): per-phase s_memtime/s_memrealtime stamps of
// workgroup 0 into a buffer of their own.  The shipped library is built without this macro.
__device__ unsigned long long g_stamps[64];
__device__ unsigned long long g_pre[8];  // stamps 9..11 (launch entry .. prologue) are kept only by launches that go on to a
                                         // full step (stamp 12): the trailing launches of a solve do not overwrite them
#define STAMP(i)                                                              \
  do {                                                                        \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                \
      const unsigned long long m_ = __builtin_amdgcn_s_memtime(), r_ = __builtin_amdgcn_s_memrealtime(); \
      if ((i) >= 9 && (i) <= 11) { g_pre[2 * ((i) - 9)] = m_; g_pre[2 * ((i) - 9) + 1] = r_; }           \
      else { g_stamps[2 * (i)] = m_; g_stamps[2 * (i) + 1] = r_; }            \
      if ((i) == 12) for (int k_ = 0; k_ < 6; ++k_) g_stamps[18 + k_] = g_pre[k_];                      \
    }                                                                         \
  } while (0)
__device__ unsigned long long g_pstamps[8];  // inside step_prologue (workgroup 0, wave 0), kept by full steps only
#define PSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_pstamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_wstamps[8 * 8];
#define STAMPW(i)                                                             \
  do {                                                                        \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                           \
      g_wstamps[(threadIdx.x >> 6) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
// Phase boundaries are also scheduling fences: without them the machine scheduler moves loads and
// LDS traffic across the phases in ways that cost ~20 % (measured: the stamped diagnostic build,
// whose stamps split basic blocks at exactly these points, was that much faster).
#define STAMP(i) __builtin_amdgcn_sched_barrier(0)
#define STAMPW(i) __builtin_amdgcn_sched_barrier(0)
#define PSTAMP(i) do { } while (0)
#endif

struct Ctrl {  // device-resident integrator state, double-buffered by attempt parity
  int status, first;
  int iter, naccept, nreject, nf;
  int cur;  // uprev = ubuf[cur], k1 = kfsal[cur]; the step writes u -> ubuf[cur^1], k7 -> kfsal[cur^1]
  int isave, nsaved;
  float t, dt;  // start time and dt of the last attempted step
  float qold, q11, dtpropose;
  float eest_last, dt_init;
  float reg_error, reg_stiff;  // single-step modes: filled by k_finalize
  float stiff_num, stiff_den;  // ||k7-k6||_rms, ||u-g6||_rms (for the regulariser's reverse sweep)
};

// The controller's part of Ctrl: exactly 16 dwords, one s_load_dwordx16.  The step prologue reads and writes only this
// (as a whole Ctrl the last four floats came through a vector load that was waited for — and spilled — before the
// partial-sum loads were even issued: one more memory round trip at the head of every step).
struct CtrlHead {
  int status, first;
  int iter, naccept, nreject, nf;
  int cur;
  int isave, nsaved;
  float t, dt;
  float qold, q11, dtpropose;
  float eest_last, dt_init;
};
static_assert(sizeof(CtrlHead) == 64 && offsetof(Ctrl, reg_error) == 64 && offsetof(Ctrl, dt_init) == offsetof(CtrlHead, dt_init),
              "CtrlHead must be the first 64 bytes of Ctrl");

struct SaveInit { float* saveat; int n; float v[8]; float* tsaved; int save_start; };

struct StepArgs {
  ModelDev m;
  float* state;     // one allocation: ubuf[2] kfsal[2] ks[5] g6, each n_local floats
  long n_local;     // B * D
  int fused;        // 1: fused Dense-2 epilogue path (D % 16 == 0 and 32-bit offsets suffice)
  float* ubuf[2];
  float* kfsal[2];
  float* ks[5];  // k2..k6
  float* g6;
  int B;          // local batch columns
  int wg_offset;  // index of this rank's first tile in the global partial vector (prered: this rank's slot)
  int nwg_global; // entries of the exchanged partial vector: all ranks' tiles, or (prered) one slot per rank
  // sharded handles, per-rank pre-reduction: every tile writes its partial to tile_part / tile_pinit (rank local), the
  // LAST workgroup of the launch to arrive (arrive[] counters) reduces them in the fixed order of reduce_partials and
  // publishes ONE triple in this rank's slot of part_send / pinit_send: the all-reduce carries nranks triples
  int force_store_k;   // 1: keep k2..k6 of a single step in global memory (the recorded forward's local step: its reverse sweep reads them)
  int prered;
  int* arrive;         // [4]: step parity 0/1, init phase 1/2
  double* tile_part;   // [2][nwg_local*PSTRIDE]
  double* tile_pinit;  // [2][nwg_local*PSTRIDE]
  double n_global;  // D * B_global, the norm's element count
  float t0, t1, abstol, reltol, bench_dt;
  int maxiters, save_everystep, exact_pow, want_stiff, mode;
  int nsave, cap_saved, cap_trace;
  const float* saveat;  // device copy
  float* u_saved;
  float* also_dst;  // save slot `also_slot` is written here as well (the caller's u_end: no copy packet after the solve); or NULL
  int also_slot;
  float* t_saved;  // device view of a pinned host array: the few saved times are written straight to the host
  // unsharded solves: progress word in pinned host memory (one 64-bit posted store per launch, see solve_progress) and
  // the final control block, both written by workgroup 0's prologue; NULL elsewhere
  unsigned long long* prog;
  Ctrl* fin_host;
  // k_init1_q as the first launch of a solve / local step (4-column family): the start state is read from init_u0 (the
  // caller's array; also copied to ubuf[0]) and workgroup 0 initialises the control blocks, saveat times and the
  // save_start time itself — no copy packet and no one-thread kernel ahead of it
  const float* init_u0; int init_fresh; int init_nsaved; SaveInit init_si;
  Ctrl* ctrl;      // [2]
  double* part_send;        // [2][nwg_global*PSTRIDE] (this rank writes its own segment)
  const double* part_recv;  // [2][...]  == part_send when nranks == 1
  double* pinit_send;       // [2][...]  init phases 1 and 2
  const double* pinit_recv;
  lrnde_trace_row* trace;
  // SDE Euler-Heun step (k_sde_step): second model = diffusion, Brownian increments, scratch
  ModelDev m2;
  const float* dW;
  float delta;
  float* sde_scratch;  // 9 arrays of n_local floats
  const float* sde_uprev;
  float* sde_u;
  // dense forward record for the adjoint: per accepted step [uprev, k1, P2, P3, P4] (REC_ARRAYS * n_local floats; lrnde_math.hpp)
  float* dense;
  float* dense_t;   // device [dense_cap]
  float* dense_dt;  // device [dense_cap]
  int dense_cap;
  int dense_direct;  // 1 (4-column kernel): every attempted step writes its own record slot [uprev,k1,P2,P3,P4] straight from LDS /
                     // registers at its end (slot = accepted steps so far; a rejected attempt's slot is rewritten by the
                     // retry) instead of the next launch's prologue copying eight state arrays through global memory
};

// control blocks, the (few) saveat times and the save_start time of a fresh solve (k_solve_init, k_init1_q)
__device__ __forceinline__ void solve_init_body(Ctrl* ctrl, float t0, int nsaved, const SaveInit& si) {
  Ctrl c;
  memset(&c, 0, sizeof(c));
  c.status = ST_RUNNING; c.first = 1; c.cur = 0; c.nsaved = nsaved;
  c.t = t0; c.dt = 0.f; c.qold = 1e-4f; c.q11 = 1.0f; c.dtpropose = 0.f;
  ctrl[0] = c;
  ctrl[1] = c;
  for (int i = 0; i < si.n; ++i) si.saveat[i] = si.v[i];
  if (si.save_start) si.tsaved[0] = t0;
}

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

template <int W> struct Vec { float v[W]; };

template <int W> __device__ __forceinline__ Vec<W> vload(const float* p) {
  Vec<W> r;
  if constexpr (W == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int W> __device__ __forceinline__ Vec<W> vzero() {
  Vec<W> r;
#pragma unroll
  for (int h = 0; h < W; ++h) r.v[h] = 0.f;
  return r;
}
template <int W> __device__ __forceinline__ void vstore(float* p, const Vec<W>& r) {
  if constexpr (W == 4) {
    f32x4 t; t.x = r.v[0]; t.y = r.v[1]; t.z = r.v[2]; t.w = r.v[3];
    *reinterpret_cast<f32x4*>(p) = t;
  } else {
    *p = r.v[0];
  }
}

// x-tile / h-tile LDS image: [kgroup][lane = (k&3)*16 + n][q = (k>>2)&3], i.e. exactly the
// B operand of four consecutive 16x16x4 MFMA k-steps per ds_read_b128.
__device__ __forceinline__ int lds_index(int row, int n) {
  return (((row >> 4) * 64 + (row & 3) * 16 + n) << 2) + ((row >> 2) & 3);
}

// Visit every (row-chunk, sample) of this workgroup's tile.  W=4: lanes read 16 B of 4
// consecutive rows; 16 lanes cover the 16 samples, 4 lane groups cover 64 B of each sample.
template <int W, class F>
__device__ __forceinline__ void tile_foreach(const ModelDev& m, int b0, int nvalid, F&& fn) {
  const int n = threadIdx.x & 15;
  const int cc = threadIdx.x >> 4;  // 0..31
  const int nch = m.Dp / W;
  for (int c = cc; c < nch; c += NT / 16) {
    const int row = c * W;
    const bool valid = (n < nvalid) && (row < m.D);
    const size_t g = (size_t)(b0 + n) * m.D + row;
    fn(row, n, valid, g);
  }
}

template <int W>
__device__ __forceinline__ void lds_put(float* xl, int row, int n, const Vec<W>& x) {
#pragma unroll
  for (int h = 0; h < W; ++h) xl[lds_index(row + h, n)] = x.v[h];
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// The same sum with DPP row operations instead of ds_bpermute shuffles (a double is two of them per step, six dependent
// steps): four v_add_f64 on DPP-moved halves give every lane its 16-lane row total, the four row totals are read with
// v_readlane and added in row order.  The total is wave-uniform.  (A different — still fixed — order of fp64 adds.)
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return ((readlane_f64(v, 0) + readlane_f64(v, 16)) + readlane_f64(v, 32)) + readlane_f64(v, 48);
}

// fixed-order block reduction of up to 3 doubles; result valid on thread 0
__device__ __forceinline__ void block_sum3(double* red, double& a, double& b, double& c) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  if (lane == 0) { red[wave * 3 + 0] = a; red[wave * 3 + 1] = b; red[wave * 3 + 2] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0, sc = 0.0;
    for (int w = 0; w < NW; ++w) { sa += red[w * 3]; sb += red[w * 3 + 1]; sc += red[w * 3 + 2]; }
    a = sa; b = sb; c = sc;
  }
  __syncthreads();
}

// every workgroup reduces the global partial vector in the same fixed order (wave 0)
struct Sum3 { double a, b, c; };
__device__ __forceinline__ Sum3 reduce_partials3(const double* p, int nwg);
__device__ __forceinline__ void reduce_partials(const double* p, int nwg, double out[3]) {
  const Sum3 r = reduce_partials3(p, nwg);
  out[0] = r.a; out[1] = r.b; out[2] = r.c;
}
// (the loads and the sum are separate calls so that a caller can put other loads in flight between them; ALL = false
// reads the first sum of each triple only — the step prologue's error norm.  A loaded-but-unused value is not free here:
// its destination register is reused at once and the reuse waits for the load.)
struct PartLoads { double va[4], vb[4], vc[4]; };
template <bool ALL> __device__ __forceinline__ void part_issue(PartLoads& L, const double* p, int nwg) {
  const int lane = threadIdx.x & 63;
  // agent-scope loads (the vector was written by the previous launch / by RCCL), four workgroups' triples per lane in
  // flight at once: this reduction opens every step launch, one round trip per loop trip was on its critical path.
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = lane + 64 * u;
    const size_t o = (size_t)(i < nwg ? i : 0) * PSTRIDE;
    L.va[u] = __hip_atomic_load(p + o + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ALL) {
      L.vb[u] = __hip_atomic_load(p + o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      L.vc[u] = __hip_atomic_load(p + o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
template <bool ALL> __device__ __forceinline__ Sum3 part_finish(const PartLoads& L, const double* p, int nwg) {
  const int lane = threadIdx.x & 63;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  // The adds run in the same order as a plain loop over i = lane, lane + 64, ...
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (lane + 64 * u < nwg) { s0 += L.va[u]; if (ALL) { s1 += L.vb[u]; s2 += L.vc[u]; } }
  for (int i0 = lane + 256; i0 < nwg; i0 += 256) {
    double va[4], vb[4] = {0.0, 0.0, 0.0, 0.0}, vc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 64 * u;
      const size_t o = (size_t)(i < nwg ? i : i0) * PSTRIDE;
      va[u] = __hip_atomic_load(p + o + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ALL) {
        vb[u] = __hip_atomic_load(p + o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vc[u] = __hip_atomic_load(p + o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + 64 * u < nwg) { s0 += va[u]; if (ALL) { s1 += vb[u]; s2 += vc[u]; } }
  }
  // lane totals -> wave total with DPP row operations (wave-uniform result; fp64 sums of fp32 squares: the rounded fp32
  // norm does not depend on the order of these adds)
  Sum3 r;
  r.a = wave_sum_dpp(s0);
  r.b = ALL ? wave_sum_dpp(s1) : 0.0;
  r.c = ALL ? wave_sum_dpp(s2) : 0.0;
  return r;
}
__device__ __forceinline__ Sum3 reduce_partials3(const double* p, int nwg) {
  PartLoads L;
  part_issue<true>(L, p, nwg);
  return part_finish<true>(L, p, nwg);
}

// A workgroup hands in its three fp64 partial sums (valid on thread 0).  which = 0/1: the step's parity block of
// part_send; 2/3: init phase 1/2 block of pinit_send.  Gather mode: thread 0 writes the tile's entry of the exchanged
// vector.  prered mode: see StepArgs — the last workgroup to arrive reduces the rank's tiles and writes the rank's slot.
__device__ __forceinline__ float rms_from(double sumsq, double n) { return (float)sqrt(sumsq / n); }

__device__ __forceinline__ void publish_partial(const StepArgs& a, int which, double s0, double s1, double s2) {
  const size_t blk = (size_t)(which & 1);
  double* send = (which < 2 ? a.part_send : a.pinit_send) + blk * (size_t)a.nwg_global * PSTRIDE;
  if (!a.prered) {
    if (threadIdx.x == 0) {
      double* p = send + (size_t)(a.wg_offset + blockIdx.x) * PSTRIDE;
      p[0] = s0; p[1] = s1; p[2] = s2;
    }
    return;
  }
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  double* tiles = (which < 2 ? a.tile_part : a.tile_pinit) + blk * (size_t)gridDim.x * PSTRIDE;
  int last = 0;
  if (lane == 0) {
    double* p = tiles + (size_t)blockIdx.x * PSTRIDE;
    __hip_atomic_store(p + 0, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int old = __hip_atomic_fetch_add(a.arrive + which, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (old == (int)gridDim.x - 1);
  }
  last = __shfl(last, 0, 64);
  if (!last) return;
  const Sum3 sr = reduce_partials3(tiles, (int)gridDim.x);  // agent-scope loads, the fixed order every reduction here uses
  if (lane == 0) {
    double* q = send + (size_t)a.wg_offset * PSTRIDE;
    q[0] = sr.a; q[1] = sr.b; q[2] = sr.c;
    __hip_atomic_store(a.arrive + which, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch of this kind
  }
}

// ---------------------------------------------------------------------------
// vector field on one tile:  k = W2 * act(W1 * [x; t] + b1) (+ t column) + b2
// xl holds the 16-sample x tile; results go to kout (global, sample-major).
//
// Canonical dot product (same definition as the oracle): fp32 fma chains over consecutive
// segments of SEGK*16 = 112 rows, each from 0 in increasing k (exactly what a run of
// v_mfma_f32_16x16x4_f32 computes), segment partials added left to right.
//   Dense 1 (K = D, M = H): wave w owns segment w of K and runs all M tiles of it as
//     independent accumulator chains (B fragments read once per k-group, weights streamed
//     straight from L2 into registers, double-buffered one k-group ahead); the segment
//     partials meet in LDS and are summed in order by the epilogue.
//   Dense 2 (K = H <= 112 in the reference models => one segment, M = D): waves split the M
//     tiles; weights for the next tile are prefetched while the current one runs.
// ---------------------------------------------------------------------------
#ifdef LRNDE_ABL_NOLOAD  // diagnostic: every weight load hits the same (L1-resident) line
#define LRNDE_ABL_KG(kg) 0
#define LRNDE_ABL_I(i) 0
#else
#define LRNDE_ABL_KG(kg) (kg)
#define LRNDE_ABL_I(i) (i)
#endif
#ifdef LRNDE_STAMPS
#define LRNDE_D2ACC() do { d2c += u1 - u0; d2p += u2 - u1; } while (0)
#else
#define LRNDE_D2ACC() do {} while (0)
#endif
constexpr int SEGK = 7;  // k-groups (of 16 rows) per canonical segment
constexpr int TG = 7;    // M tiles run concurrently by one wave in Dense 1

__device__ __forceinline__ f32x4 mfma4(const f32x4& a, const f32x4& b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}

struct Smem {
  float* xl;   // x tile, B-operand image of Dense 1            [KG1][64][4]
  float* hl;   // h tile, B-operand image of Dense 2            [KG2p][64][4] (zero beyond KG2)
  float* pl;   // Dense-1 segment partials                      [nseg1][MT1][64][4]
  float* bias; // w1t[Hp] b1[Hp] w2t[Dp] b2[Dp]
  double* red;
  struct Bcast* bc;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Weights are streamed with buffer loads: address = SGPR descriptor + per-lane VGPR offset
// (lane*16, constant) + SGPR offset (tile / k-group, computed on the scalar unit), so a load
// costs ONE vector-issue slot and no VALU address arithmetic.  That matters because the two
// waves of a SIMD share its vector issue: every non-MFMA vector instruction of one wave queues
// behind the partner's MFMA stream (measured: 30-instruction load blocks took 600-1700 cycles).
__device__ __forceinline__ f32x4 wload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

struct FevalCtx {
  __amdgpu_buffer_rsrc_t rs1, rs2;
  // kept in registers for the whole launch, so that neither GEMM phase starts on an exposed
  // L2 round trip: Dense-1 fragments of (segment = wave, first k-group, tiles 0..TG-1) and the
  // Dense-2 fragments of this wave's first tile
  f32x4 r1[TG];
  f32x4 r2[SEGK];
};

__device__ __forceinline__ void feval_ctx_init(const ModelDev& m, FevalCtx& fc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int MT1p = ((m.MT1 + TG - 1) / TG) * TG, KG2p = ((m.KG2 + SEGK - 1) / SEGK) * SEGK;
  fc.rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)m.W1p, 0, MT1p * m.KG1 * 1024, 0x00020000);
  fc.rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)m.W2p, 0, m.MT2 * KG2p * 1024, 0x00020000);
  const int voff = lane * 16;
  const int nseg1 = (m.KG1 + SEGK - 1) / SEGK;
  const int kg0 = (wave < nseg1) ? wave * SEGK : 0;
#pragma unroll
  for (int i = 0; i < TG; ++i) fc.r1[i] = wload(fc.rs1, voff, (i * m.KG1 + kg0) * 1024);
  const int mt0 = (wave < m.MT2) ? wave : 0;
#pragma unroll
  for (int j = 0; j < SEGK; ++j) fc.r2[j] = wload(fc.rs2, voff, (mt0 * KG2p + j) * 1024);
}

// per-lane access to the state workspace (ubuf / kfsal / ks / g6 live in ONE allocation) through
// a buffer descriptor: byte address = base + SGPR offset (array, tile) + per-lane VGPR offset.
// Lanes whose sample column is beyond the batch get an out-of-range offset: their loads return 0
// and their stores are dropped by the hardware range check.
#ifdef LRNDE_DBG_PLAIN_IO
__device__ const char* g_dbg_base;
#endif
struct TileIO {
  __amdgpu_buffer_rsrc_t rs;
  int voff;
};
__device__ __forceinline__ TileIO make_tile_io(const StepArgs& a, int b0, int nvalid) {
  const int lane = threadIdx.x & 63, n = lane & 15, rq = lane >> 4;
  TileIO io;
  io.rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.state, 0, (int)(a.n_local * 40), 0x00020000);
  io.voff = (n < nvalid) ? ((b0 + n) * a.m.D + rq * 4) * 4 : 0x7ffffff0;
#ifdef LRNDE_DBG_PLAIN_IO
  if (threadIdx.x == 0) g_dbg_base = reinterpret_cast<const char*>(a.state);
  __syncthreads();
#endif
  return io;
}
#ifdef LRNDE_DBG_PLAIN_IO
__device__ __forceinline__ f32x4 sload(const TileIO& io, int soff) {
  if (io.voff == 0x7ffffff0) return f32x4{0, 0, 0, 0};
  return *reinterpret_cast<const f32x4*>(g_dbg_base + soff + io.voff);
}
__device__ __forceinline__ void sstore(const TileIO& io, int soff, const f32x4& v) {
  if (io.voff == 0x7ffffff0) return;
  *reinterpret_cast<f32x4*>(const_cast<char*>(g_dbg_base) + soff + io.voff) = v;
}
#else
__device__ __forceinline__ f32x4 sload(const TileIO& io, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(io.rs, io.voff, soff, 0));
}
// NOTE (ROCm 7.2 / gfx950): with an SGPR soffset hipcc places VALU writes of the store-data
// registers directly behind buffer_store_dwordx4 (its hazard recognizer assumes that form is
// safe); on this chip it is not — lanes 12-15 of dword 1 were stored with the overwritten value.
// With a literal soffset the compiler inserts the required wait state, so stores fold the uniform
// offset into the per-lane offset (one v_add) instead.
__device__ __forceinline__ void sstore(const TileIO& io, int soff, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), io.rs, io.voff + soff, 0, 0);
}
#endif
// byte offset of state array `idx` (0,1: ubuf; 2,3: kfsal; 4..8: ks; 9: g6)
__device__ __forceinline__ int arr_off(const StepArgs& a, int idx) { return (int)(a.n_local * 4) * idx; }
// ubuf[i] / kfsal[i] for a runtime i (the ping-pong parity): the two buffers of a pair are n_local floats apart
// (fill_args), so this is arithmetic — indexing the kernel argument's pointer arrays with a runtime value made hipcc copy
// them to scratch (40 bytes per lane in k_step_q) and fetch the pointer from there at the head of every launch
__device__ __forceinline__ float* ubuf_at(const StepArgs& a, int i) { return a.ubuf[0] + (size_t)i * a.n_local; }
__device__ __forceinline__ float* kfsal_at(const StepArgs& a, int i) { return a.kfsal[0] + (size_t)i * a.n_local; }

// ---- Dense-2 epilogue policies -------------------------------------------------------------
// pre(mt, pb): issue the loads this tile's epilogue needs (one tile ahead of use);
// post(mt, kv, pb): kv = the four k values of this lane (rows mt*16 + rq*4 .. +3, column n).
struct EpiStoreK {  // k only (lrnde_rhs, init phases, generic path)
  static constexpr int NPRE = 1;
  static constexpr bool DBUF = false;
  const ModelDev* m; float* kout; int b0, nvalid, w;
  __device__ __forceinline__ void pre(int, f32x4 (&)[NPRE]) const {}
  __device__ __forceinline__ void post(int mt, const f32x4& kv, f32x4 (&)[NPRE]) const {
    const int lane = threadIdx.x & 63, n = lane & 15, rq = lane >> 4;
    const int row0 = mt * 16 + rq * 4;
    if (n < nvalid) {
      float* dst = kout + (size_t)(b0 + n) * m->D + row0;
      if (w == 4) {
        if (row0 < m->D) *reinterpret_cast<f32x4*>(dst) = kv;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (row0 + r < m->D) dst[r] = kv[r];
      }
    }
  }
};

// after k_S (S = 2..6): store k_S and build the NEXT stage input
//   x_{S+1} = uprev + dt*(a_{S+1,1} k1 + ... + a_{S+1,S} k_S)        (src/perform_step.jl:13-18)
// left to right, k_S (still in registers) last; x goes to the LDS x tile (and to u / g6).
template <int S> struct EpiStage {
  static constexpr int NPRE = S;  // uprev + k1..k_{S-1}
  static constexpr bool DBUF = true;  // operands prefetched one tile ahead
  TileIO io;
  int off_up, off_k[6], off_out, off_x;  // off_x: u (S==6) or g6 (S==5, stiffness) or -1
  float dt;
  float* xl;
  __device__ __forceinline__ void pre(int mt, f32x4 (&pb)[NPRE]) const {
    pb[0] = sload(io, off_up + mt * 64);
#pragma unroll
    for (int j = 0; j < S - 1; ++j) pb[1 + j] = sload(io, off_k[j] + mt * 64);
  }
  __device__ __forceinline__ void post(int mt, const f32x4& kv, f32x4 (&pb)[NPRE]) const {
    const int lane = threadIdx.x & 63, n = lane & 15, rq = lane >> 4;
    constexpr int off = (S - 1) * S / 2;  // row S+1 of the tableau
    sstore(io, off_out + mt * 64, kv);  // always: this tile shape's later stages re-read k_S from global memory (Bcast::store_k is the 4-column kernel's)
    f32x4 x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float o[S];
#pragma unroll
      for (int j = 0; j < S - 1; ++j) o[j] = pb[1 + j][r];
      o[S - 1] = kv[r];
      float sum = (float)Tsit5::A[off] * o[0] + (float)Tsit5::A[off + 1] * o[1];
#pragma unroll
      for (int j = 2; j < S; ++j) sum = sum + (float)Tsit5::A[off + j] * o[j];
      x[r] = pb[0][r] + dt * sum;
    }
    if (off_x >= 0) sstore(io, off_x + mt * 64, x);
    float* dst = xl + ((mt * 64 + n) << 2) + rq;   // lds_index(mt*16 + rq*4 + r, n) = dst + r*64
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r * 64] = x[r];
  }
};

// after k7: store k7; utilde, scaled residual and the stiffness differences, accumulated in
// fp64 per lane (src/perform_step.jl:21-47, 210-212)
struct EpiFinal {
  static constexpr int NPRE = 9;  // uprev, u, k1..k6, g6
  static constexpr bool DBUF = false;  // 9 quads: issued right before the tile's MFMA chain
  TileIO io;
  int off_up, off_u, off_k[6], off_g6, off_out;
  float dt, abstol, reltol;
  int want_stiff, nvalid;
  double *aerr, *anum, *aden;
  __device__ __forceinline__ void pre(int mt, f32x4 (&pb)[NPRE]) const {
    pb[0] = sload(io, off_up + mt * 64);
    pb[1] = sload(io, off_u + mt * 64);
#pragma unroll
    for (int j = 0; j < 6; ++j) pb[2 + j] = sload(io, off_k[j] + mt * 64);
    if (want_stiff) pb[8] = sload(io, off_g6 + mt * 64);
  }
  __device__ __forceinline__ void post(int mt, const f32x4& kv, f32x4 (&pb)[NPRE]) const {
    const int lane = threadIdx.x & 63, n = lane & 15;
    sstore(io, off_out + mt * 64, kv);
    if (n >= nvalid) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sum = (float)Tsit5::BT[0] * pb[2][r] + (float)Tsit5::BT[1] * pb[3][r];
      sum = sum + (float)Tsit5::BT[2] * pb[4][r];
      sum = sum + (float)Tsit5::BT[3] * pb[5][r];
      sum = sum + (float)Tsit5::BT[4] * pb[6][r];
      sum = sum + (float)Tsit5::BT[5] * pb[7][r];
      sum = sum + (float)Tsit5::BT[6] * kv[r];
      const float utilde = dt * sum;
      const float sc = abstol + fmaxf_(__builtin_fabsf(pb[0][r]), __builtin_fabsf(pb[1][r])) * reltol;
      const float rr = utilde / sc;
      const float sq = rr * rr;
      *aerr += (double)sq;
      if (want_stiff) {
        const float d1 = pb[1][r] - pb[8][r];
        const float d2 = kv[r] - pb[7][r];
        const float q1 = d1 * d1, q2 = d2 * d2;
        *aden += (double)q1; *anum += (double)q2;
      }
    }
  }
};

template <int W, class Epi>
__device__ __forceinline__ void feval_tile(const ModelDev& m, const Smem& sm, const FevalCtx& fc,
                                           float ts, const Epi& epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform
  const int rq = lane >> 4;
  const int voff = lane * 16;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float* w1t = sm.bias; const float* b1 = w1t + m.Hp;
  const float* w2t = b1 + m.Hp; const float* b2 = w2t + m.Dp;
  STAMP(1);
  STAMPW(0);
  // ---- Dense 1: [Hp x Dp] * [Dp x 16], K-split by segment; W1p is padded to MT1p tiles ----
  const int nseg1 = (m.KG1 + SEGK - 1) / SEGK;
  {
    const f32x4* xp = reinterpret_cast<const f32x4*>(sm.xl) + lane;
    for (int seg = wave; seg < nseg1; seg += NW) {
      const int kg_lo = seg * SEGK, kg_hi = min(m.KG1, kg_lo + SEGK);
      for (int mt0 = 0; mt0 < m.MT1; mt0 += TG) {
        const int tbase = mt0 * m.KG1;
        f32x4 acc[TG], aX[TG], aY[TG], bX, bY;
#pragma unroll
        for (int i = 0; i < TG; ++i) acc[i] = zero4;
#define LRNDE_LOAD1(a, b, kg)                                                     \
  do {                                                                            \
    b = xp[(kg) * 64];                                                            \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) a[i] = wload(fc.rs1, voff, (tbase + i * m.KG1 + LRNDE_ABL_KG(kg)) * 1024); \
    __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ABOVE the MFMA block it overlaps */ \
  } while (0)
#define LRNDE_MMA1(a, b)                                                          \
  do {                                                                            \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b.x, acc[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b.y, acc[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b.z, acc[i], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b.w, acc[i], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                            \
  } while (0)
        // The steady state prefetches unconditionally (a conditional prefetch forces vmcnt(0) at
        // the join); the first k-group of the wave's own segment comes from the resident
        // registers, the last one or two k-groups are peeled.
        int kg = kg_lo;
        if (seg == wave && mt0 == 0) {
          bX = xp[kg_lo * 64];
          if (kg_lo + 1 < kg_hi) {
            LRNDE_LOAD1(aY, bY, kg_lo + 1);
            LRNDE_MMA1(fc.r1, bX);
#pragma unroll
            for (int i = 0; i < TG; ++i) aX[i] = aY[i];
            bX = bY;
            kg = kg_lo + 1;
          } else {
            LRNDE_MMA1(fc.r1, bX);
            kg = kg_hi;
          }
        } else {
          LRNDE_LOAD1(aX, bX, kg_lo);
        }
#pragma unroll 1
        for (; kg + 2 < kg_hi; kg += 2) {
          LRNDE_LOAD1(aY, bY, kg + 1);
          LRNDE_MMA1(aX, bX);
          LRNDE_LOAD1(aX, bX, kg + 2);
          LRNDE_MMA1(aY, bY);
        }
        if (kg + 1 < kg_hi) {
          LRNDE_LOAD1(aY, bY, kg + 1);
          LRNDE_MMA1(aX, bX);
          LRNDE_MMA1(aY, bY);
        } else if (kg < kg_hi) {
          LRNDE_MMA1(aX, bX);
        }
#undef LRNDE_LOAD1
#undef LRNDE_MMA1
        f32x4* pp = reinterpret_cast<f32x4*>(sm.pl) + ((size_t)seg * m.MT1 + mt0) * 64 + lane;
#pragma unroll
        for (int i = 0; i < TG; ++i) if (mt0 + i < m.MT1) pp[i * 64] = acc[i];
      }
    }
  }
  STAMP(2);
  STAMPW(1);
  __syncthreads();
  STAMP(3);
  // epilogue 1: sum the segment partials in order, time column, bias, activation -> h tile
  // (B operand layout of Dense 2).  element e = (mt*64 + l)*4 + r of the C fragment:
  // row o = mt*16 + (l>>4)*4 + r, column n = l&15.
  {
    const int nquad = m.MT1 * 64;  // one C-fragment quad (4 rows x 1 column) per thread
    const f32x4* pl4 = reinterpret_cast<const f32x4*>(sm.pl);
    for (int q = threadIdx.x; q < nquad; q += NT) {
      const int l = q & 63, mt = q >> 6;
      f32x4 v = pl4[q];
      for (int sgi = 1; sgi < nseg1; ++sgi) {
        const f32x4 pv = pl4[(size_t)sgi * nquad + q];
        v.x = v.x + pv.x; v.y = v.y + pv.y; v.z = v.z + pv.z; v.w = v.w + pv.w;
      }
      const int o0 = mt * 16 + (l >> 4) * 4;
      const f32x4 wt = *reinterpret_cast<const f32x4*>(w1t + o0);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + o0);
      float* dst = sm.hl + ((mt * 64 + (l & 15)) << 2) + (l >> 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pre = m.td ? fma_(wt[r], ts, v[r]) : v[r];
        pre = pre + bb[r];
        dst[r * 64] = act_apply(m.act, pre);
      }
    }
  }
  STAMPW(2);
  __syncthreads();
  STAMP(4);
  STAMPW(3);
  // ---- Dense 2: [Dp x Hp] * [Hp x 16]; W2p is padded to KG2p = nseg2*SEGK k-groups ----
  {
    const f32x4* hp = reinterpret_cast<const f32x4*>(sm.hl) + lane;
    const int nseg2 = (m.KG2 + SEGK - 1) / SEGK;
    const int KG2p = nseg2 * SEGK;
    auto finish = [&](int mt, const f32x4& tot) {  // time column + bias -> k values of this lane
      const int row0 = mt * 16 + rq * 4;
      const f32x4 wt = *reinterpret_cast<const f32x4*>(w2t + row0);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + row0);
      f32x4 kv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pre = m.td ? fma_(wt[r], ts, tot[r]) : tot[r];
        kv[r] = pre + bb[r];
      }
      return kv;
    };
    if (nseg2 == 1) {
      // one canonical segment (H <= 112): B fragments live in registers for all tiles of this
      // wave; the first tile's A fragments are resident, the next tile's are in flight while the
      // current tile's chain runs.
      f32x4 b[SEGK], aX[SEGK], aY[SEGK], pX[Epi::NPRE], pY[Epi::NPRE];
#pragma unroll
      for (int j = 0; j < SEGK; ++j) b[j] = hp[j * 64];
      const int ntile = (m.MT2 > wave) ? (m.MT2 - wave + NW - 1) / NW : 0;
#define LRNDE_LOAD2(a, pb, i)                                                                      \
  do {                                                                                             \
    _Pragma("unroll") for (int j = 0; j < SEGK; ++j) a[j] = wload(fc.rs2, voff, ((wave + LRNDE_ABL_I(i) * NW) * KG2p + j) * 1024); \
    if constexpr (Epi::DBUF) epi.pre(wave + (i) * NW, pb);                                         \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#ifdef LRNDE_STAMPS
#define TQ2(x) do { __builtin_amdgcn_sched_barrier(0); x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
      unsigned long long d2c = 0, d2p = 0, u0, u1, u2;
#else
#define TQ2(x) do {} while (0)
#endif
#define LRNDE_MMA2(a, pb, i)                                      \
  do {                                                            \
    f32x4 acc = zero4;                                            \
    if constexpr (!Epi::DBUF) { epi.pre(wave + (i) * NW, pb); __builtin_amdgcn_sched_barrier(0); } \
    TQ2(u0);                                                      \
    _Pragma("unroll") for (int j = 0; j < SEGK; ++j) acc = mfma4(a[j], b[j], acc); \
    { const f32x4 kv_ = finish(wave + (i) * NW, acc);             \
    TQ2(u1);                                                      \
    epi.post(wave + (i) * NW, kv_, pb); }                         \
    TQ2(u2);                                                      \
    LRNDE_D2ACC();                                                \
    __builtin_amdgcn_sched_barrier(0);                            \
  } while (0)
      if (ntile > 0) {
        int i = 1;
        if constexpr (Epi::DBUF) epi.pre(wave, pX);
        if (ntile > 1) {
          LRNDE_LOAD2(aY, pY, 1);
          LRNDE_MMA2(fc.r2, pX, 0);
#pragma unroll 1
          for (; i + 2 < ntile; i += 2) {
            LRNDE_LOAD2(aX, pX, i + 1);
            LRNDE_MMA2(aY, pY, i);
            LRNDE_LOAD2(aY, pY, i + 2);
            LRNDE_MMA2(aX, pX, i + 1);
          }
          if (i + 1 < ntile) {
            LRNDE_LOAD2(aX, pX, i + 1);
            LRNDE_MMA2(aY, pY, i);
            LRNDE_MMA2(aX, pX, i + 1);
          } else {
            LRNDE_MMA2(aY, pY, i);
          }
        } else {
          LRNDE_MMA2(fc.r2, pX, 0);
        }
      }
#undef LRNDE_LOAD2
#undef LRNDE_MMA2
#ifdef LRNDE_STAMPS
      if (blockIdx.x == 0 && lane == 0) { g_wstamps[wave * 8 + 5] = d2c; g_wstamps[wave * 8 + 6] = d2p; g_wstamps[wave * 8 + 7] = ntile; }
#endif
    } else {
      // general hidden width: segment chains summed left to right in registers
      for (int mt = wave; mt < m.MT2; mt += NW) {
        f32x4 tot = zero4;
        for (int sg = 0; sg < nseg2; ++sg) {
          f32x4 acc = zero4;
#pragma unroll
          for (int j = 0; j < SEGK; ++j)
            acc = mfma4(wload(fc.rs2, voff, (mt * KG2p + sg * SEGK + j) * 1024), hp[(sg * SEGK + j) * 64], acc);
          if (sg == 0) tot = acc;
          else { tot.x = tot.x + acc.x; tot.y = tot.y + acc.y; tot.z = tot.z + acc.z; tot.w = tot.w + acc.w; }
        }
        f32x4 pb[Epi::NPRE];
        epi.pre(mt, pb);
        epi.post(mt, finish(mt, tot), pb);
      }
    }
  }
  STAMP(5);
  STAMPW(4);
  __syncthreads();  // k stores are visible to the whole workgroup; xl/hl/pl may be overwritten
  STAMP(6);
}

// vector field with the plain "store k" epilogue
template <int W>
__device__ __forceinline__ void feval_store(const ModelDev& m, const Smem& sm, const FevalCtx& fc, float ts,
                                            float* kout, int b0, int nvalid) {
  EpiStoreK e;
  e.m = &m; e.kout = kout; e.b0 = b0; e.nvalid = nvalid; e.w = W;
  feval_tile<W, EpiStoreK>(m, sm, fc, ts, e);
}

// ---------------------------------------------------------------------------
// stage inputs (src/perform_step.jl:11-18), left-to-right, no contraction
// ---------------------------------------------------------------------------
template <int S> __device__ __forceinline__ float stage_value(float up, const float* kv, float dt) {
  constexpr int off = (S - 2) * (S - 1) / 2;
  if constexpr (S == 2) {
    const float a = dt * (float)Tsit5::A[0];
    return up + a * kv[0];
  } else {
    float s = (float)Tsit5::A[off] * kv[0] + (float)Tsit5::A[off + 1] * kv[1];
#pragma unroll
    for (int j = 2; j < S - 1; ++j) s = s + (float)Tsit5::A[off + j] * kv[j];
    return up + dt * s;
  }
}

template <int S, int W>
__device__ __forceinline__ void stage_combine(const StepArgs& a, float* xl, const float* uprev,
                                              const float* k1, float* uout, float dt, int b0,
                                              int nvalid) {
  tile_foreach<W>(a.m, b0, nvalid, [&](int row, int n, bool valid, size_t g) {
    Vec<W> x;
    if (valid) {
      const Vec<W> up = vload<W>(uprev + g);
      Vec<W> kk[S - 1];
      kk[0] = vload<W>(k1 + g);
#pragma unroll
      for (int j = 1; j < S - 1; ++j) kk[j] = vload<W>(a.ks[j - 1] + g);
#pragma unroll
      for (int h = 0; h < W; ++h) {
        float kv[S - 1];
#pragma unroll
        for (int j = 0; j < S - 1; ++j) kv[j] = kk[j].v[h];
        x.v[h] = stage_value<S>(up.v[h], kv, dt);
      }
      if constexpr (S == 6) { if (a.want_stiff) vstore<W>(a.g6 + g, x); }
      if constexpr (S == 7) vstore<W>(uout + g, x);
    } else {
      x = vzero<W>();
    }
    lds_put<W>(xl, row, n, x);
  });
  __syncthreads();
}

// ---------------------------------------------------------------------------
// device-side integrator logic (OrdinaryDiffEq loopfooter!/loopheader!, PI
// controller, ode_determine_initdt).  Run by wave 0 of every
// workgroup on identical inputs; block 0 publishes the next control block.
// ---------------------------------------------------------------------------
struct Bcast {
  int do_step, cur;
  float t, dt;
  // footer of the previous attempt (save actions)
  int accepted_prev, cur_prev, isave0, nsaved0;
  int isave1;  // the accepted step saves the saveat points [isave0, isave1) (counted by the prologue: no thread of the
               // launch has to read the saveat list to find out that, as for most steps, there is nothing to save)
  float t_new, tprev, dt_prev;
  float dt0;  // init phase 2
  int dense_idx;  // index of the accepted step in the dense record (-1: none)
  // 1: this step's k2..k6 must reach global memory — the next launch's prologue reads them (dense record of the step, or
  // a saveat point inside the step to interpolate) or the caller does (single-step entry points).  Otherwise they live in
  // LDS for the launch only: five state-sized stores per attempted step (8 MB of 21 at B=512) that nothing would read
  int store_k;
  int dense_slot;  // dense_direct: the record slot this attempt writes (-1: none)
};

__device__ __forceinline__ float init_dt0(const double s[3], double n, float dtmax) {
  const float d0 = rms_from(s[0], n), d1 = rms_from(s[1], n);
  float dt0;
  if ((double)d0 < 1e-5 || (double)d1 < 1e-5) dt0 = 1e-6f;
  else dt0 = (d0 / d1) / 100.0f;
  return fminf_(dt0, dtmax);
}

__device__ __forceinline__ float init_dt_final(const double s1[3], const double s2[3], double n,
                                               float dtmax) {
  const float d1 = rms_from(s1[1], n);
  const float dt0 = init_dt0(s1, n, dtmax);
  const float d2 = rms_from(s2[0], n) / dt0;
  const float maxd = fmaxf_(d1, d2);
  float dt1;
  if ((double)maxd <= 1e-15) {
    dt1 = fmaxf_(1e-6f, dt0 * 1e-3f);
  } else {
    const float l10 = (float)log10((double)maxd);
    const float e = (-(2.0f + l10)) / 5.0f;
    dt1 = (float)pow(10.0, (double)e);
  }
  return fminf_(fminf_(100.0f * dt0, dt1), dtmax);
}

__device__ __forceinline__ float init_dt_final3(const Sum3& r1, const Sum3& r2, double n, float dtmax) {
  const float d0 = rms_from(r1.a, n), d1 = rms_from(r1.b, n);
  float dt0;
  if ((double)d0 < 1e-5 || (double)d1 < 1e-5) dt0 = 1e-6f;
  else dt0 = (d0 / d1) / 100.0f;
  dt0 = fminf_(dt0, dtmax);
  const float d2 = rms_from(r2.a, n) / dt0;
  const float maxd = fmaxf_(d1, d2);
  float dt1;
  if ((double)maxd <= 1e-15) {
    dt1 = fmaxf_(1e-6f, dt0 * 1e-3f);
  } else {
    const float l10 = (float)log10((double)maxd);
    const float e = (-(2.0f + l10)) / 5.0f;
    dt1 = (float)pow(10.0, (double)e);
  }
  return fminf_(fminf_(100.0f * dt0, dt1), dtmax);
}

// Progress of a solve, for the host loop that keeps the stream fed (lrnde_solve): ONE 64-bit store per launch into
// pinned host memory — [launches run : 24][status : 8][saves completed : 16][steps still to go at this dt : 16] — into
// slot (launch index mod PROG_RING) of a small ring, so the host can read the report of EVERY launch, in order: a
// batch-sharded run needs that (all ranks must steer by the same launch's report or their collective counts drift
// apart), and each entry validates itself by the launch count it carries.  A posted write: the wave does not wait for
// it, and the host only steers by it (how many launches to enqueue next, when to start the companion's local step);
// everything it reports is read after the stream has been synchronised.  A finished solve also leaves its control block
// in host memory, so the host needs no copy packet on the stream.
constexpr int PROG_RING = 32;  // > the deepest queue the host keeps (16 launches ahead of the last report it has read)
template <class A> __device__ __forceinline__ void solve_progress(const A& a, int j, const CtrlHead& c, int nsaved_done, float steps_left) {
  if (!a.prog) return;
  if (c.status != ST_RUNNING) *reinterpret_cast<CtrlHead*>(a.fin_host) = c;
  const unsigned long long cnt = (unsigned long long)(j + 1) & 0xffffffull;
  const unsigned long long stt = (unsigned long long)(c.status & 0xff);
  const unsigned long long nsv = (unsigned long long)(nsaved_done > 65535 ? 65535 : nsaved_done);
  const float sl = __builtin_ceilf(steps_left);
  const unsigned long long rem = (unsigned long long)(sl > 65535.f ? 65535 : (sl > 0.f ? (int)sl : 0));
  __hip_atomic_store(a.prog + (j & (PROG_RING - 1)), cnt | (stt << 24) | (nsv << 32) | (rem << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the launch arguments step_prologue works with (see there), each forced into a vector register
template <class T> __device__ __forceinline__ T pin_v(T x) { asm volatile("" : "+v"(x)); return x; }
struct PrologueArgs {
  int mode, maxiters, nsave, cap_saved, save_everystep, exact_pow, nwg_global, dense_cap, dense_direct, force_store_k, cap_trace;
  float t0, t1;
  double n_global;
  Ctrl* ctrl; const double* part_recv; const double* pinit_recv; const float* saveat; float* dense; lrnde_trace_row* trace;
  unsigned long long* prog; Ctrl* fin_host;
};
__device__ __forceinline__ PrologueArgs pin_args(const StepArgs& a) {
  PrologueArgs p{};
  p.mode = pin_v(a.mode); p.maxiters = pin_v(a.maxiters); p.nsave = pin_v(a.nsave); p.cap_saved = pin_v(a.cap_saved);
  p.save_everystep = pin_v(a.save_everystep); p.exact_pow = pin_v(a.exact_pow); p.nwg_global = pin_v(a.nwg_global);
  p.dense_cap = pin_v(a.dense_cap); p.dense_direct = pin_v(a.dense_direct); p.force_store_k = pin_v(a.force_store_k);
  p.cap_trace = pin_v(a.cap_trace); p.t0 = pin_v(a.t0); p.t1 = pin_v(a.t1); p.n_global = pin_v(a.n_global);
  p.ctrl = pin_v(a.ctrl); p.part_recv = pin_v(a.part_recv); p.pinit_recv = pin_v(a.pinit_recv); p.saveat = pin_v(a.saveat);
  p.dense = pin_v(a.dense); p.trace = pin_v(a.trace); p.prog = pin_v(a.prog); p.fin_host = pin_v(a.fin_host);
  return p;
}

__device__ __forceinline__ void step_prologue(const StepArgs& a_, int j, Bcast* bc) {
  // wave 0 only
  const int lane = threadIdx.x & 63;
  if (a_.mode == MODE_BENCH) {  // timing hook: every launch is a full step on fixed inputs
    if (lane == 0) { bc->do_step = 1; bc->cur = 0; bc->t = a_.t0; bc->dt = a_.bench_dt; bc->accepted_prev = 0; bc->store_k = 0; bc->dense_idx = -1; bc->dense_slot = -1; bc->isave0 = 0; bc->isave1 = 0; }
    return;
  }
  // The launch arguments the decision uses, pinned in vector registers once.  Left to itself the compiler re-reads each of
  // them from the kernel-argument segment at every use (the step kernel has no scalar registers to spare): some two dozen
  // scalar loads, each waited for on the spot, in the one stretch of the launch that every wave is waiting on.
  const PrologueArgs a = pin_args(a_);
  PSTAMP(0);
  const CtrlHead* cin = reinterpret_cast<const CtrlHead*>(a.ctrl + (j & 1));
  CtrlHead c = *cin;
  CtrlHead* cout = reinterpret_cast<CtrlHead*>(a.ctrl + ((j + 1) & 1));
  // Everything the decision needs from memory is put in flight at once: the control block, the previous attempt's
  // partial sums (unused on the first launch) and the saveat times.  One after the other these round trips (each to
  // memory: the writers were other launches) were the start of every step.
  const double* ppart = a.part_recv + (size_t)(j & 1) * a.nwg_global * PSTRIDE;
  PartLoads pl;
  part_issue<false>(pl, ppart, a.nwg_global);
  // (the first 64 saveat times ride in the lanes: no second, dependent round trip once the save index is known)
  float svl = 0.f;
  if (lane < a.nsave) svl = a.saveat[lane];
  auto saveat_at = [&](int is) -> float {  // a.saveat[is] for is < a.nsave (is: wave-uniform)
    return is < 64 ? __shfl(svl, is, 64) : a.saveat[is];
  };
  if (c.status != ST_RUNNING) {
    if (blockIdx.x == 0 && lane == 0) { *cout = c; solve_progress(a, j, c, c.nsaved, 0.f); }
    if (lane == 0) bc->do_step = 0, bc->accepted_prev = 0;
    return;
  }
  const float gamma = 0.9f, qmin = 0.2f, qmax = 10.0f, qoldinit = 1e-4f;
  const float beta1 = (float)(7.0 / 50.0), beta2 = (float)(2.0 / 25.0);
  const float dtmax = a.t1 - a.t0;
  const float dtmin = fmaxf_(eps_f(a.t1), eps_f(a.t0));
  int accepted = 0, do_step = 0;
  float t = c.t, dt = c.dt;
  Bcast b;
  b.accepted_prev = 0; b.cur_prev = c.cur; b.isave0 = c.isave; b.isave1 = c.isave; b.nsaved0 = c.nsaved;
  b.t_new = c.t; b.tprev = c.t; b.dt_prev = c.dt; b.dt0 = 0.f; b.dense_idx = -1;

  if (c.first) {
    if (a.mode != MODE_SINGLE_GIVEN_DT) {
      const Sum3 r1 = reduce_partials3(a.pinit_recv, a.nwg_global);
      const Sum3 r2 = reduce_partials3(a.pinit_recv + (size_t)a.nwg_global * PSTRIDE, a.nwg_global);
      dt = init_dt_final3(r1, r2, a.n_global, dtmax);
      c.nf = 3;  // initdt: 2 f-evals, initialize!: fsalfirst
    }
    c.dt_init = dt;
    c.dtpropose = dt;
  } else {
    PSTAMP(1);
    // What does not depend on the error norm is computed while the partial sums are still on their way: the controller's
    // qold^beta2, and everything the ACCEPTED branch needs about the new time (the snap onto t1, its eps floor, the saveat
    // points it passes).  Same expressions as before, earlier.
    const float pq = a.exact_pow ? (float)pow((double)c.qold, (double)beta2) : fastpow(c.qold, beta2);
    const float ttmp = c.t + c.dt;
    const float t_acc = (__builtin_fabsf(ttmp - a.t1) < 100.0f * eps_f(fmaxf_(c.t, a.t1))) ? a.t1 : ttmp;
    const float floor_acc = fmaxf_(eps_f(t_acc), dtmin);
    int is_acc = c.isave, ns_acc = c.nsaved;
    while (is_acc < a.nsave && saveat_at(is_acc) <= t_acc) { ++is_acc; ++ns_acc; }
    if (a.save_everystep) ++ns_acc;
    __builtin_amdgcn_sched_barrier(0);
    const Sum3 sr = part_finish<false>(pl, ppart, a.nwg_global);
    PSTAMP(2);
    const float eest = rms_from(sr.a, a.n_global);
    PSTAMP(3);
    c.eest_last = eest;
    float q;
    if (eest == 0.0f) {
      q = 1.0f / qmax;
    } else {
      c.q11 = a.exact_pow ? (float)pow((double)eest, (double)beta1) : fastpow(eest, beta1);
      q = c.q11 / pq;
      q = fmaxf_(1.0f / qmax, fminf_(1.0f / qmin, q / gamma));
    }
    accepted = (eest <= 1.0f);
    const int ntr = c.naccept + c.nreject;
    if (blockIdx.x == 0 && lane == 0 && a.trace && ntr < a.cap_trace) {
      lrnde_trace_row r; r.t = c.t; r.dt = c.dt; r.eest = eest; r.accepted = accepted;
      a.trace[ntr] = r;
    }
    if (eest != eest) {
      c.status = LRNDE_DT_NAN;
    } else if (accepted) {
      c.naccept++;
      const float dtnew = c.dt / q;
      c.qold = fmaxf_(eest, qoldinit);
      b.tprev = c.t; b.dt_prev = c.dt;
      t = t_acc;
      c.dtpropose = fmaxf_(fminf_(dtmax, dtnew), floor_acc);
      b.accepted_prev = 1; b.t_new = t;
      if (a.dense) {
        b.dense_idx = c.naccept - 1;
        if (b.dense_idx >= a.dense_cap) { c.status = LRNDE_CAPACITY; b.accepted_prev = 0; b.dense_idx = -1; }
      }
      // savevalues!: count what this step saves (performed by all threads afterwards)
      const int is = is_acc, ns = ns_acc;
      if (ns > a.cap_saved) c.status = LRNDE_CAPACITY, b.accepted_prev = 0;
      else { c.isave = is; c.nsaved = ns; b.isave1 = is; }
    } else {
      c.nreject++;
    }
  }

  if (c.status == ST_RUNNING && a.mode == MODE_SOLVE) {
    if (!(t < a.t1)) {
      c.status = ST_DONE;
    } else {
      // loopheader!
      if (!c.first) {
        if (accepted) { c.cur ^= 1; dt = c.dtpropose; }
        else dt = c.dt / fminf_(1.0f / qmin, c.q11 / gamma);
      }
      c.iter++;
      dt = fminf_(dtmax, dt);
      dt = fmaxf_(dt, dtmin);
      dt = fminf_(__builtin_fabsf(dt), __builtin_fabsf(a.t1 - t));
      if (c.iter > a.maxiters) c.status = LRNDE_MAXITERS;
      else if (dt != dt) c.status = LRNDE_DT_NAN;
      else if (__builtin_fabsf(dt) <= __builtin_fabsf(dtmin)) c.status = LRNDE_DT_LESS_THAN_MIN;
      else { do_step = 1; c.nf += 6; }
    }
  } else if (c.status == ST_RUNNING) {  // single-step modes: integrator.dt as is
    if (c.first) { do_step = 1; c.iter = 1; c.nf += 6; }
    else c.status = ST_DONE;
  }
  PSTAMP(4);
  c.t = t; c.dt = dt; c.first = 0;
  b.do_step = do_step; b.cur = c.cur; b.t = t; b.dt = dt;
  // (the margin covers the snap of t + dt onto t1 within 100 eps; a pending saveat equal to the new time is a copy of u)
  b.dense_slot = -1;
  if (a.dense && a.dense_direct && do_step) {
    if (c.naccept >= a.dense_cap) { c.status = LRNDE_CAPACITY; b.do_step = 0; }  // the host retries with a larger record
    else b.dense_slot = c.naccept;
  }
  b.store_k = (a.mode == MODE_SINGLE_GIVEN_DT) || (a.dense != nullptr && !a.dense_direct) || a.force_store_k ||
              (a.mode == MODE_SOLVE && c.isave < a.nsave && saveat_at(c.isave) < t + dt * 1.001f);
  if (lane == 0) {
    *bc = b;
    if (blockIdx.x == 0) {
      *cout = c;
      // (nsaved0: the saves of COMPLETED launches — what this launch saves is written by its body, after this word)
      solve_progress(a, j, c, b.nsaved0, (do_step && dt > 0.f) ? (a.t1 - t) / dt : 0.f);
    }
  }
  PSTAMP(5);
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
__device__ __forceinline__ Smem carve(const ModelDev& m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nseg1 = (m.KG1 + SEGK - 1) / SEGK, KG2p = ((m.KG2 + SEGK - 1) / SEGK) * SEGK;
  Smem s;
  s.xl = reinterpret_cast<float*>(smem);
  s.hl = s.xl + (size_t)m.Dp * NB;
  s.pl = s.hl + (size_t)KG2p * 256;
  s.bias = s.pl + (size_t)nseg1 * m.Hp * NB;
  s.red = reinterpret_cast<double*>(s.bias + 2 * (size_t)(m.Hp + m.Dp));
  s.bc = reinterpret_cast<Bcast*>(s.red + NW * 3);
  return s;
}
static size_t smem_bytes(int Dp, int Hp) {
  const size_t nseg1 = (size_t)((Dp / 16 + SEGK - 1) / SEGK);
  const size_t KG2p = (size_t)((Hp / 16 + SEGK - 1) / SEGK) * SEGK;
  return ((size_t)Dp * NB + KG2p * 256 + nseg1 * Hp * NB + 2 * (size_t)(Hp + Dp)) * sizeof(float) +
         NW * 3 * sizeof(double) + sizeof(Bcast) + 16;
}
// once per launch: zero the h tile (its padded k-groups must stay zero) and stage the bias /
// time-column vectors in LDS
__device__ __forceinline__ void smem_init(const ModelDev& m, const Smem& s) {
  const int KG2p = ((m.KG2 + SEGK - 1) / SEGK) * SEGK;
  for (int i = threadIdx.x; i < KG2p * 256; i += NT) s.hl[i] = 0.f;
  for (int i = threadIdx.x; i < m.Hp; i += NT) { s.bias[i] = m.w1t[i]; s.bias[m.Hp + i] = m.b1[i]; }
  for (int i = threadIdx.x; i < m.Dp; i += NT) { s.bias[2 * m.Hp + i] = m.w2t[i]; s.bias[2 * m.Hp + m.Dp + i] = m.b2[i]; }
}

__device__ __forceinline__ void smem_init_bias_only(const ModelDev& m, const Smem& s) {
  const int KG2p = ((m.KG2 + SEGK - 1) / SEGK) * SEGK;
  for (int i = m.KG2 * 256 + threadIdx.x; i < KG2p * 256; i += NT) s.hl[i] = 0.f;  // padded k-groups
  for (int i = threadIdx.x; i < m.Hp; i += NT) { s.bias[i] = m.w1t[i]; s.bias[m.Hp + i] = m.b1[i]; }
  for (int i = threadIdx.x; i < m.Dp; i += NT) { s.bias[2 * m.Hp + i] = m.w2t[i]; s.bias[2 * m.Hp + m.Dp + i] = m.b2[i]; }
}

// du = f(u, t) for the whole batch (lrnde_rhs)
template <int W> __global__ __launch_bounds__(NT) void k_rhs(StepArgs a, const float* u, float t, float* du) {
  STAMP(0);
  const Smem s = carve(a.m);
  smem_init(a.m, s);
  FevalCtx fc;
  feval_ctx_init(a.m, fc);
  const int b0 = blockIdx.x * NB, nvalid = min(NB, a.B - b0);
  tile_foreach<W>(a.m, b0, nvalid, [&](int row, int n, bool valid, size_t g) {
    const Vec<W> x = valid ? vload<W>(u + g) : vzero<W>();
    lds_put<W>(s.xl, row, n, x);
  });
  __syncthreads();
  feval_store<W>(a.m, s, fc, t, du, b0, nvalid);
}

// init phase 1: f0 = f(u0, t0) -> k1; partial sums of (u0/sk)^2 and (f0/sk)^2
template <int W> __global__ __launch_bounds__(NT) void k_init1(StepArgs a) {
  const Smem s = carve(a.m);
  smem_init(a.m, s);
  FevalCtx fc;
  feval_ctx_init(a.m, fc);
  const int b0 = blockIdx.x * NB, nvalid = min(NB, a.B - b0);
  const Ctrl c = a.ctrl[0];
  const float* u0 = ubuf_at(a, c.cur);
  float* f0 = kfsal_at(a, c.cur);
  tile_foreach<W>(a.m, b0, nvalid, [&](int row, int n, bool valid, size_t g) {
    const Vec<W> x = valid ? vload<W>(u0 + g) : vzero<W>();
    lds_put<W>(s.xl, row, n, x);
  });
  __syncthreads();
  feval_store<W>(a.m, s, fc, c.t, f0, b0, nvalid);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
    if (!valid) return;
    const Vec<W> u = vload<W>(u0 + g), f = vload<W>(f0 + g);
#pragma unroll
    for (int h = 0; h < W; ++h) {
      const float sk = a.abstol + __builtin_fabsf(u.v[h]) * a.reltol;
      const float r0 = u.v[h] / sk, r1 = f.v[h] / sk;
      const float q0 = r0 * r0, q1 = r1 * r1;
      a0 += (double)q0; a1 += (double)q1;
    }
  });
  block_sum3(s.red, a0, a1, a2);
  publish_partial(a, 2, a0, a1, 0.0);
}

// init phase 2: u1 = u0 + dt0*f0, f1 = f(u1, t0+dt0); partial sum of ((f1-f0)/sk)^2
template <int W> __global__ __launch_bounds__(NT) void k_init2(StepArgs a) {
  const Smem s = carve(a.m);
  smem_init(a.m, s);
  FevalCtx fc;
  feval_ctx_init(a.m, fc);
  const int b0 = blockIdx.x * NB, nvalid = min(NB, a.B - b0);
  const Ctrl c = a.ctrl[0];
  if (threadIdx.x < 64) {
    double s1[3];
    reduce_partials(a.pinit_recv, a.nwg_global, s1);
    if (threadIdx.x == 0) s.bc->dt0 = init_dt0(s1, a.n_global, a.t1 - a.t0);
  }
  __syncthreads();
  const float dt0 = s.bc->dt0;
  const float* u0 = ubuf_at(a, c.cur);
  const float* f0 = kfsal_at(a, c.cur);
  float* f1 = a.ks[0];
  tile_foreach<W>(a.m, b0, nvalid, [&](int row, int n, bool valid, size_t g) {
    Vec<W> x;
    if (valid) {
      const Vec<W> u = vload<W>(u0 + g), f = vload<W>(f0 + g);
#pragma unroll
      for (int h = 0; h < W; ++h) x.v[h] = u.v[h] + dt0 * f.v[h];
    } else {
      x = vzero<W>();
    }
    lds_put<W>(s.xl, row, n, x);
  });
  __syncthreads();
  feval_store<W>(a.m, s, fc, c.t + dt0, f1, b0, nvalid);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
    if (!valid) return;
    const Vec<W> u = vload<W>(u0 + g), f = vload<W>(f0 + g), ff = vload<W>(f1 + g);
#pragma unroll
    for (int h = 0; h < W; ++h) {
      const float sk = a.abstol + __builtin_fabsf(u.v[h]) * a.reltol;
      const float r2 = (ff.v[h] - f.v[h]) / sk;
      const float q2 = r2 * r2;
      a0 += (double)q2;
    }
  });
  block_sum3(s.red, a0, a1, a2);
  publish_partial(a, 3, a0, 0.0, 0.0);
}

// one attempted Tsit5 step for the whole batch (src/perform_step.jl:3-47), preceded by the
// device-side footer of the previous attempt and header of this one.
// SPEC only changes the kernel's NAME: launches the host is not yet sure are needed (they may
// find the solve finished and exit in the prologue) are issued as k_step<W,true>, so that the
// kernel-trace statistics of k_step<W,false> describe full steps.
template <int W, bool SPEC> __global__ __launch_bounds__(NT) void k_step(StepArgs a, int j) {
  const Smem s = carve(a.m);
  smem_init(a.m, s);
  FevalCtx fc;
  feval_ctx_init(a.m, fc);
  const int b0 = blockIdx.x * NB, nvalid = min(NB, a.B - b0);
  STAMP(10);
  if (threadIdx.x < 64) step_prologue(a, j, s.bc);
  __syncthreads();
  STAMP(11);
  const Bcast bc = *s.bc;

  // savevalues! of the step accepted by the prologue (Tsit5 dense output / copy)
  if (bc.accepted_prev) {
    const float* up = ubuf_at(a, bc.cur_prev);
    const float* un = ubuf_at(a, bc.cur_prev ^ 1);
    const float* k1 = kfsal_at(a, bc.cur_prev);
    const float* k7 = kfsal_at(a, bc.cur_prev ^ 1);
    int slot = bc.nsaved0;
    for (int is = bc.isave0; is < bc.isave1; ++is, ++slot) {
      const float ts = a.saveat[is];
      float* dst = a.u_saved + (size_t)slot * a.B * a.m.D;
      float* dst2 = slot == a.also_slot ? a.also_dst : nullptr;
      if (ts != bc.t_new) {
        const float theta = (ts - bc.tprev) / bc.dt_prev;
        float bw[7];
        tsit5_bweights(theta, bw);
        tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
          if (!valid) return;
          const Vec<W> y0 = vload<W>(up + g), v1 = vload<W>(k1 + g), v2 = vload<W>(a.ks[0] + g),
                       v3 = vload<W>(a.ks[1] + g), v4 = vload<W>(a.ks[2] + g),
                       v5 = vload<W>(a.ks[3] + g), v6 = vload<W>(a.ks[4] + g), v7 = vload<W>(k7 + g);
          Vec<W> o;
#pragma unroll
          for (int h = 0; h < W; ++h) {
            float sum = v1.v[h] * bw[0] + v2.v[h] * bw[1];
            sum = sum + v3.v[h] * bw[2];
            sum = sum + v4.v[h] * bw[3];
            sum = sum + v5.v[h] * bw[4];
            sum = sum + v6.v[h] * bw[5];
            sum = sum + v7.v[h] * bw[6];
            o.v[h] = y0.v[h] + bc.dt_prev * sum;
          }
          vstore<W>(dst + g, o);
          if (dst2) vstore<W>(dst2 + g, o);
        });
      } else {
        tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
          if (!valid) return;
          const Vec<W> o = vload<W>(un + g);
          vstore<W>(dst + g, o);
          if (dst2) vstore<W>(dst2 + g, o);
        });
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) a.t_saved[slot] = ts;
    }
    if (a.save_everystep) {
      float* dst = a.u_saved + (size_t)slot * a.B * a.m.D;
      tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
        if (valid) vstore<W>(dst + g, vload<W>(un + g));
      });
      if (blockIdx.x == 0 && threadIdx.x == 0) a.t_saved[slot] = bc.t_new;
    }
    if (bc.dense_idx >= 0) {  // dense record of the accepted step (InterpolatingAdjoint keeps u and k1..k7)
      const size_t nst = (size_t)a.n_local;
      float* dd = a.dense + (size_t)bc.dense_idx * REC_ARRAYS * nst;
      const float* src[8] = {up, k1, a.ks[0], a.ks[1], a.ks[2], a.ks[3], a.ks[4], k7};
      tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
        if (!valid) return;
        Vec<W> v[8], P2, P3, P4;   // polynomial form [uprev, k1, P2, P3, P4] (lrnde_math.hpp tsit5_rec_poly)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = vload<W>(src[q] + g);
#pragma unroll
        for (int h = 0; h < W; ++h) {
          const float kk[6] = {v[2].v[h], v[3].v[h], v[4].v[h], v[5].v[h], v[6].v[h], v[7].v[h]};
          float P[3];
          tsit5_rec_poly(v[1].v[h], kk, P);
          P2.v[h] = P[0]; P3.v[h] = P[1]; P4.v[h] = P[2];
        }
        vstore<W>(dd + g, v[0]); vstore<W>(dd + nst + g, v[1]);
        vstore<W>(dd + 2 * nst + g, P2); vstore<W>(dd + 3 * nst + g, P3); vstore<W>(dd + 4 * nst + g, P4);
      });
      if (blockIdx.x == 0 && threadIdx.x == 0) { a.dense_t[bc.dense_idx] = bc.tprev; a.dense_dt[bc.dense_idx] = bc.dt_prev; }
    }
  }
  if (!bc.do_step) return;

  const float t = bc.t, dt = bc.dt;
  const float* uprev = ubuf_at(a, bc.cur);
  float* unew = ubuf_at(a, bc.cur ^ 1);
  const float* k1 = kfsal_at(a, bc.cur);
  float* k7 = kfsal_at(a, bc.cur ^ 1);
  const float c1 = (float)Tsit5::C[0], c2 = (float)Tsit5::C[1], c3 = (float)Tsit5::C[2],
              c4 = (float)Tsit5::C[3];

  double aerr = 0.0, anum = 0.0, aden = 0.0;
  if (a.fused) {
    // Fused path: each Dense-2 epilogue stores k_S and builds the next stage input (or, after k7,
    // the error residuals) from operands prefetched through the state buffer descriptor; only
    // the first stage input needs a stand-alone pass.
    const TileIO io = make_tile_io(a, b0, nvalid);
    const int tb = 0;  // tile offsets are added by the policies (mt * 64 bytes)
    const int o_up = arr_off(a, bc.cur) + tb, o_un = arr_off(a, bc.cur ^ 1) + tb;
    const int o_k1 = arr_off(a, 2 + bc.cur) + tb, o_k7 = arr_off(a, 2 + (bc.cur ^ 1)) + tb;
    const int o_g6 = arr_off(a, 9) + tb;
    stage_combine<2, W>(a, s.xl, uprev, k1, unew, dt, b0, nvalid);
    STAMP(12);
#define LRNDE_STAGE(S, TS)                                                              \
  do {                                                                                  \
    EpiStage<S> e;                                                                      \
    e.io = io; e.off_up = o_up; e.off_k[0] = o_k1;                                                     \
    _Pragma("unroll") for (int q = 0; q < 5; ++q) e.off_k[1 + q] = arr_off(a, 4 + q) + tb; \
    e.off_out = arr_off(a, 4 + (S - 2)) + tb;                                           \
    e.off_x = (S == 6) ? o_un : ((S == 5 && a.want_stiff) ? o_g6 : -1);                 \
    e.dt = dt; e.xl = s.xl;                                                             \
    feval_tile<W, EpiStage<S>>(a.m, s, fc, (TS), e);                                    \
    STAMP(11 + S);                                                                      \
  } while (0)
    LRNDE_STAGE(2, t + c1 * dt);
    LRNDE_STAGE(3, t + c2 * dt);
    LRNDE_STAGE(4, t + c3 * dt);
    LRNDE_STAGE(5, t + c4 * dt);
    LRNDE_STAGE(6, t + dt);
#undef LRNDE_STAGE
    EpiFinal ef;
    ef.io = io; ef.off_up = o_up; ef.off_u = o_un; ef.off_k[0] = o_k1;
#pragma unroll
    for (int q = 0; q < 5; ++q) ef.off_k[1 + q] = arr_off(a, 4 + q) + tb;
    ef.off_g6 = o_g6; ef.off_out = o_k7;
    ef.dt = dt; ef.abstol = a.abstol; ef.reltol = a.reltol; ef.want_stiff = a.want_stiff; ef.nvalid = nvalid;
    ef.aerr = &aerr; ef.anum = &anum; ef.aden = &aden;
    feval_tile<W, EpiFinal>(a.m, s, fc, t + dt, ef);
    STAMP(18);
  } else {
  stage_combine<2, W>(a, s.xl, uprev, k1, unew, dt, b0, nvalid);
  feval_store<W>(a.m, s, fc, t + c1 * dt, a.ks[0], b0, nvalid);
  stage_combine<3, W>(a, s.xl, uprev, k1, unew, dt, b0, nvalid);
  feval_store<W>(a.m, s, fc, t + c2 * dt, a.ks[1], b0, nvalid);
  stage_combine<4, W>(a, s.xl, uprev, k1, unew, dt, b0, nvalid);
  feval_store<W>(a.m, s, fc, t + c3 * dt, a.ks[2], b0, nvalid);
  stage_combine<5, W>(a, s.xl, uprev, k1, unew, dt, b0, nvalid);
  feval_store<W>(a.m, s, fc, t + c4 * dt, a.ks[3], b0, nvalid);
  stage_combine<6, W>(a, s.xl, uprev, k1, unew, dt, b0, nvalid);
  feval_store<W>(a.m, s, fc, t + dt, a.ks[4], b0, nvalid);
  stage_combine<7, W>(a, s.xl, uprev, k1, unew, dt, b0, nvalid);
  feval_store<W>(a.m, s, fc, t + dt, k7, b0, nvalid);

  // utilde, scaled residual, regularisation residuals (src/perform_step.jl:21-47, 210-212)
  tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
    if (!valid) return;
    const Vec<W> up = vload<W>(uprev + g), un = vload<W>(unew + g), v1 = vload<W>(k1 + g),
                 v2 = vload<W>(a.ks[0] + g), v3 = vload<W>(a.ks[1] + g), v4 = vload<W>(a.ks[2] + g),
                 v5 = vload<W>(a.ks[3] + g), v6 = vload<W>(a.ks[4] + g), v7 = vload<W>(k7 + g);
    Vec<W> gg;
    if (a.want_stiff) gg = vload<W>(a.g6 + g);
#pragma unroll
    for (int h = 0; h < W; ++h) {
      float sum = (float)Tsit5::BT[0] * v1.v[h] + (float)Tsit5::BT[1] * v2.v[h];
      sum = sum + (float)Tsit5::BT[2] * v3.v[h];
      sum = sum + (float)Tsit5::BT[3] * v4.v[h];
      sum = sum + (float)Tsit5::BT[4] * v5.v[h];
      sum = sum + (float)Tsit5::BT[5] * v6.v[h];
      sum = sum + (float)Tsit5::BT[6] * v7.v[h];
      const float utilde = dt * sum;
      const float sc = a.abstol + fmaxf_(__builtin_fabsf(up.v[h]), __builtin_fabsf(un.v[h])) * a.reltol;
      const float r = utilde / sc;
      const float sq = r * r;
      aerr += (double)sq;
      if (a.want_stiff) {
        const float d1 = un.v[h] - gg.v[h];
        const float d2 = v7.v[h] - v6.v[h];
        const float q1 = d1 * d1, q2 = d2 * d2;
        aden += (double)q1; anum += (double)q2;
      }
    }
  });
  }
  block_sum3(s.red, aerr, anum, aden);
  STAMP(19);
  publish_partial(a, (j + 1) & 1, aerr, anum, aden);
}


// ---------------------------------------------------------------------------------------------
// Adaptive Euler-Heun SDE step with its error estimate, diagonal noise, supplied dW
// (src/perform_step.jl:172-206, residual :214-216): 3 drift + 3 diffusion evaluations.
// Drift = a.m (Chain(Dense, Dense)); diffusion = a.m2 (Dense(D=>D) held as identity-Dense + Dense,
// which is the same canonical arithmetic).  Sizes are tiny (MNIST-SDE: 32 x B state), so this is a
// latency kernel: one launch per step, 16 columns per workgroup, plain epilogues.
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void sde_feval(const ModelDev& m, float ts, float* out, int b0, int nvalid) {
  const Smem s = carve(m);
  __syncthreads();
  smem_init_bias_only(m, s);
  __syncthreads();
  FevalCtx fc;
  feval_ctx_init(m, fc);
  feval_store<W>(m, s, fc, ts, out, b0, nvalid);
}

template <int W> __global__ __launch_bounds__(NT) void k_sde_step(StepArgs a) {
  const Smem s = carve(a.m);  // the x tile image depends only on D: shared by both models
  const int b0 = blockIdx.x * NB, nvalid = min(NB, a.B - b0);
  const size_t n = (size_t)a.n_local;
  float *du1 = a.sde_scratch, *Kb = du1 + n, *L = du1 + 2 * n, *g2 = du1 + 3 * n, *f2 = du1 + 4 * n,
        *du2 = du1 + 5 * n, *g3 = du1 + 6 * n;
  const float* up = a.sde_uprev;
  const float t = a.t0, dt = a.bench_dt;
  const float sqdt = __builtin_sqrtf(dt);
  // zero both h tiles' padded k-groups once (the larger model's extent covers both)
  {
    const int KG2a = ((a.m.KG2 + SEGK - 1) / SEGK) * SEGK, KG2b = ((a.m2.KG2 + SEGK - 1) / SEGK) * SEGK;
    const int kmax = KG2a > KG2b ? KG2a : KG2b;
    for (int i = threadIdx.x; i < kmax * 256; i += NT) s.hl[i] = 0.f;
  }
  auto fill = [&](auto&& fn) {
    __syncthreads();
    tile_foreach<W>(a.m, b0, nvalid, [&](int row, int nn, bool valid, size_t g) {
      Vec<W> x = vzero<W>();
      if (valid) x = fn(g);
      lds_put<W>(s.xl, row, nn, x);
    });
    __syncthreads();
  };
  fill([&](size_t g) { return vload<W>(up + g); });
  sde_feval<W>(a.m, t, du1, b0, nvalid);                      // :174 du1 = f(uprev, t)
  sde_feval<W>(a.m2, t, L, b0, nvalid);                       // :176 L = g(uprev, t)   (x tile unchanged)
  fill([&](size_t g) {                                        // :175,179,183 tmp = (uprev + dt*du1) + L*dW
    const Vec<W> u = vload<W>(up + g), d = vload<W>(du1 + g), l = vload<W>(L + g), w = vload<W>(a.dW + g);
    Vec<W> k, x;
#pragma unroll
    for (int h = 0; h < W; ++h) { k.v[h] = u.v[h] + dt * d.v[h]; x.v[h] = k.v[h] + l.v[h] * w.v[h]; }
    vstore<W>(Kb + g, k);
    return x;
  });
  sde_feval<W>(a.m2, t + dt, g2, b0, nvalid);                 // :184
  sde_feval<W>(a.m, t + dt, f2, b0, nvalid);                  // :191
  const float hdt = dt / 2.0f;
  fill([&](size_t g) {                                        // :191 u ; x <- K
    const Vec<W> u = vload<W>(up + g), d = vload<W>(du1 + g), l = vload<W>(L + g), w = vload<W>(a.dW + g),
                 gg = vload<W>(g2 + g), ff = vload<W>(f2 + g);
    Vec<W> un;
#pragma unroll
    for (int h = 0; h < W; ++h) {
      const float gtmp2 = 0.5f * (l.v[h] + gg.v[h]);
      const float noise2 = gtmp2 * w.v[h];
      un.v[h] = (u.v[h] + hdt * (d.v[h] + ff.v[h])) + noise2;
    }
    vstore<W>(a.sde_u + g, un);
    return vload<W>(Kb + g);
  });
  sde_feval<W>(a.m, t + dt, du2, b0, nvalid);                 // :193 du2 = f(K, t+dt)
  fill([&](size_t g) {                                        // :196 utilde = uprev + L*sqdt
    const Vec<W> u = vload<W>(up + g), l = vload<W>(L + g);
    Vec<W> x;
#pragma unroll
    for (int h = 0; h < W; ++h) x.v[h] = u.v[h] + l.v[h] * sqdt;
    return x;
  });
  sde_feval<W>(a.m2, t, g3, b0, nvalid);                      // :197
  double acc = 0.0, z1 = 0.0, z2 = 0.0;
  tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
    if (!valid) return;
    const Vec<W> u = vload<W>(up + g), un = vload<W>(a.sde_u + g), d1 = vload<W>(du1 + g), d2 = vload<W>(du2 + g),
                 l = vload<W>(L + g), gg = vload<W>(g3 + g), w = vload<W>(a.dW + g);
#pragma unroll
    for (int h = 0; h < W; ++h) {
      const float Ed = (dt * (d2.v[h] - d1.v[h])) / 2.0f;                       // :194
      const float ggp = (gg.v[h] - l.v[h]) / sqdt;                              // :197
      const float En = (ggp * (w.v[h] * w.v[h])) / 2.0f;                        // :198
      const float sc = a.abstol + fmaxf_(__builtin_fabsf(u.v[h]), __builtin_fabsf(un.v[h])) * a.reltol;
      const float r = (a.delta * Ed + En) / sc;                                 // :214-216
      const float sq = r * r;
      acc += (double)sq;
    }
  });
  block_sum3(s.red, acc, z1, z2);
  if (threadIdx.x == 0) {
    double* p = a.part_send + ((size_t)a.nwg_global + a.wg_offset + blockIdx.x) * PSTRIDE;  // parity-1 slot
    p[0] = acc; p[1] = 0.0; p[2] = 0.0;
  }
}

// Milstein step, diagonal noise, Ito (src/perform_step.jl:108-170): 1 drift + 2 diffusion evaluations.  The
// reference's du2 = f(K, t+dt) and En only feed a `tmp` that is overwritten before EEst (:163-166), so they
// are not evaluated.  EEst = rms((u - uprev) / (abstol + max(|uprev|,|u|) reltol)) (:166, :218-220).
template <int W> __global__ __launch_bounds__(NT) void k_sde_rkmil(StepArgs a) {
  const Smem s = carve(a.m);
  const int b0 = blockIdx.x * NB, nvalid = min(NB, a.B - b0);
  const size_t n = (size_t)a.n_local;
  float *du1 = a.sde_scratch, *Kb = du1 + n, *L = du1 + 2 * n, *gt = du1 + 3 * n;
  const float* up = a.sde_uprev;
  const float t = a.t0, dt = a.bench_dt;
  const float sqdt = __builtin_sqrtf(dt);
  {
    const int KG2a = ((a.m.KG2 + SEGK - 1) / SEGK) * SEGK, KG2b = ((a.m2.KG2 + SEGK - 1) / SEGK) * SEGK;
    const int kmax = KG2a > KG2b ? KG2a : KG2b;
    for (int i = threadIdx.x; i < kmax * 256; i += NT) s.hl[i] = 0.f;
  }
  auto fill = [&](auto&& fn) {
    __syncthreads();
    tile_foreach<W>(a.m, b0, nvalid, [&](int row, int nn, bool valid, size_t g) {
      Vec<W> x = vzero<W>();
      if (valid) x = fn(g);
      lds_put<W>(s.xl, row, nn, x);
    });
    __syncthreads();
  };
  fill([&](size_t g) { return vload<W>(up + g); });
  sde_feval<W>(a.m, t, du1, b0, nvalid);                      // :130 du1 = f(uprev, t)
  sde_feval<W>(a.m2, t, L, b0, nvalid);                       // :131 L = g(uprev, t)
  fill([&](size_t g) {                                        // :133, :136-137 tmp = K + sqdt*L (Ito)
    const Vec<W> u = vload<W>(up + g), d = vload<W>(du1 + g), l = vload<W>(L + g);
    Vec<W> k, x;
#pragma unroll
    for (int h = 0; h < W; ++h) { k.v[h] = u.v[h] + dt * d.v[h]; x.v[h] = k.v[h] + sqdt * l.v[h]; }
    vstore<W>(Kb + g, k);
    return x;
  });
  sde_feval<W>(a.m2, t, gt, b0, nvalid);                      // :138 gtmp = g(tmp, t)
  __syncthreads();
  const float hdt = 0.5f * __builtin_fabsf(dt);
  double acc = 0.0, z1 = 0.0, z2 = 0.0;
  tile_foreach<W>(a.m, b0, nvalid, [&](int, int, bool valid, size_t g) {
    if (!valid) return;
    const Vec<W> u = vload<W>(up + g), k = vload<W>(Kb + g), l = vload<W>(L + g), gg = vload<W>(gt + g), w = vload<W>(a.dW + g);
    Vec<W> un;
#pragma unroll
    for (int h = 0; h < W; ++h) {
      const float J = (0.5f * w.v[h]) * w.v[h] - hdt;                         // :117, :122
      const float Dgj = (gg.v[h] - l.v[h]) / sqdt;                            // :139
      un.v[h] = (k.v[h] + l.v[h] * w.v[h]) + Dgj * J;                         // :141
      const float sc = a.abstol + fmaxf_(__builtin_fabsf(u.v[h]), __builtin_fabsf(un.v[h])) * a.reltol;
      const float r = (un.v[h] - u.v[h]) / sc;                                // :166, :218-220
      const float sq = r * r;
      acc += (double)sq;
    }
    vstore<W>(a.sde_u + g, un);
  });
  block_sum3(s.red, acc, z1, z2);
  if (threadIdx.x == 0) {
    double* p = a.part_send + ((size_t)a.nwg_global + a.wg_offset + blockIdx.x) * PSTRIDE;
    p[0] = acc; p[1] = 0.0; p[2] = 0.0;
  }
}

#include "lrnde_sde_fast.hpp"
#include "lrnde_qtile.hpp"
#include "lrnde_adjoint.hpp"
#include "lrnde_backward.hpp"

// single-step modes: EEst and the two regularisation values from the partial sums
__global__ void k_finalize(StepArgs a, int j) {
  if (threadIdx.x >= 64) return;
  double s[3];
  reduce_partials(a.part_recv + (size_t)(j & 1) * a.nwg_global * PSTRIDE, a.nwg_global, s);
  if (threadIdx.x == 0) {
    Ctrl* c = a.ctrl + (j & 1);
    const float eest = rms_from(s[0], a.n_global);
    c->eest_last = eest;
    c->reg_error = eest * c->dt;
    const float den = rms_from(s[2], a.n_global);
    float rs = 0.0f;
    if (den != 0.0f) {
      const float num = rms_from(s[1], a.n_global);
      rs = __builtin_fabsf(num / (den + 1.1920929e-7f)) / 3.5068f;
    }
    c->reg_stiff = rs;
    c->stiff_den = den;
    c->stiff_num = rms_from(s[1], a.n_global);
    c->status = ST_DONE;
  }
}

// the SDE steps' own footer for lrnde_sde_solve_fixed: EEst and EEst*dt of the step just run into a record slot (no
// integrator state is involved: the step kernels take t and dt as arguments)
__global__ void k_sde_record(StepArgs a, float dt, Ctrl* rec) {
  if (threadIdx.x >= 64) return;
  double s[3];
  reduce_partials(a.part_recv + (size_t)a.nwg_global * PSTRIDE, a.nwg_global, s);
  if (threadIdx.x == 0) {
    const float eest = rms_from(s[0], a.n_global);
    rec->eest_last = eest;
    rec->reg_error = eest * dt;
    rec->status = ST_DONE;
  }
}

__global__ void k_ctrl_init(Ctrl* ctrl, float t0, float dt, int cur, int nsaved) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Ctrl c;
  memset(&c, 0, sizeof(c));
  c.status = ST_RUNNING; c.first = 1; c.cur = cur; c.nsaved = nsaved;
  c.t = t0; c.dt = dt; c.qold = 1e-4f; c.q11 = 1.0f; c.dtpropose = dt;
  ctrl[0] = c;
  ctrl[1] = c;
}

// start of a solve in one launch: control blocks, the (few) saveat times and the save_start time — no copy packets
__global__ void k_solve_init(Ctrl* ctrl, float t0, int nsaved, SaveInit si) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  solve_init_body(ctrl, t0, nsaved, si);
}

// Dense(D=>D) parameters [vec(Wg); bg] -> the 2-layer form [vec(I); 0; vec(Wg); bg] (identity first layer)
__global__ void k_diff_expand(const float* pd, int D, int has_bias, float* p2) {
  const size_t nI = (size_t)D * D;
  const size_t total = 2 * nI + 2 * (size_t)D;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float v;
    if (i < nI) v = ((i % D) == (i / D)) ? 1.0f : 0.0f;       // W1 = I (column-major)
    else if (i < nI + D) v = 0.0f;                             // b1 = 0
    else if (i < 2 * nI + D) v = pd[i - nI - D];               // W2 = Wg
    else v = has_bias ? pd[nI + (i - 2 * nI - D)] : 0.0f;      // b2 = bg
    p2[i] = v;
  }
}

// flat Lux parameter vector -> MFMA A-fragment layout (zero padded)
__global__ void k_pack(const float* p, int D, int H, int td, int Dp, int Hp, float* W1p, float* w1t,
                       float* b1, float* W2p, float* w2t, float* b2) {
  // W1p: [MT1p][KG1][64][4] with MT1p a multiple of TG; W2p: [MT2][KG2p][64][4] with KG2p a
  // multiple of SEGK; everything outside the real (H x D) / (D x H) blocks is zero.
  const int KG1 = Dp / 16, KG2 = ((Hp / 16 + SEGK - 1) / SEGK) * SEGK;
  const int MT1p = ((Hp / 16 + TG - 1) / TG) * TG;
  const size_t n1 = (size_t)MT1p * KG1 * 256, n2 = (size_t)(Dp / 16) * KG2 * 256;
  const size_t base2 = (size_t)H * (D + td) + H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n1 + n2 + Hp + Dp;
       i += (size_t)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int q = i & 3, lane = (i >> 2) & 63;
      const size_t blk = i >> 8;
      const int kg = blk % KG1, mt = blk / KG1;
      const int o = mt * 16 + (lane & 15), k = kg * 16 + q * 4 + (lane >> 4);
      W1p[i] = (o < H && k < D) ? p[(size_t)o + (size_t)H * k] : 0.f;
    } else if (i < n1 + n2) {
      const size_t e = i - n1;
      const int q = e & 3, lane = (e >> 2) & 63;
      const size_t blk = e >> 8;
      const int kg = blk % KG2, mt = blk / KG2;
      const int o = mt * 16 + (lane & 15), k = kg * 16 + q * 4 + (lane >> 4);
      W2p[e] = (o < D && k < H) ? p[base2 + (size_t)o + (size_t)D * k] : 0.f;
    } else if (i < n1 + n2 + Hp) {
      const int o = i - n1 - n2;
      w1t[o] = (td && o < H) ? p[(size_t)o + (size_t)H * D] : 0.f;
      b1[o] = (o < H) ? p[(size_t)H * (D + td) + o] : 0.f;
    } else {
      const int o = i - n1 - n2 - Hp;
      w2t[o] = (td && o < D) ? p[base2 + (size_t)o + (size_t)D * H] : 0.f;
      b2[o] = (o < D) ? p[base2 + (size_t)D * (H + td) + o] : 0.f;
    }
  }
}

}  // namespace

// ===========================================================================
// host side
// ===========================================================================

struct lrnde_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  lrnde_model_desc desc{};
  ModelDev m{};
  bool have_params = false;
  // packed weights
  float *W1p = nullptr, *W2p = nullptr, *w1t = nullptr, *b1 = nullptr, *w2t = nullptr, *b2 = nullptr;
  float *W1q = nullptr, *W2q = nullptr;
  float *V1p = nullptr, *U2p = nullptr;          // transposed weights for the backward pass
  float *V1q = nullptr, *U2q = nullptr;          // the same in the 4-column layouts
  float *bw_y = nullptr, *bw_h = nullptr, *bw_dp = nullptr, *bw_da = nullptr;  // VJP scratch (B*D, B*Hp, B*Hp, B*Hp), three sets each
  // deferred parameter-gradient GEMM (adjoint Tsit5 loop): the GEMM of RHS evaluation e rides in the launch of the VJP
  // of evaluation e+1 (k_vjp_q_pg); scratch set bw_cur is the one the next VJP writes
  bool pg_defer = false, pg_pending = false; int bw_cur = 0; PgradArgs pg_args;
  bool pg_accumulate = false;  // the next parameter-gradient GEMM adds to gp instead of overwriting it (regulariser sweep)
  int bwB = 0;
  // dense forward record + adjoint work vectors
  float *dense = nullptr, *dense_t = nullptr, *dense_dt = nullptr;
  int dense_cap = 0; size_t dense_n = 0; bool dense_on = false;
  float* adj = nullptr; size_t adj_elems = 0;   // 11 vectors of N = B*D + P floats
  double* adj_part = nullptr; double* adj_part_host = nullptr;
  // device-side adjoint controller (lrnde_adjoint.hpp): control blocks, initdt partial sums, tstops, pinned read-back slots
  AdjCtrl* adj_ctl = nullptr; AdjCtrl* adj_ctl_host = nullptr; double* adj_ipart = nullptr; float* adj_stops = nullptr; int adj_stops_cap = 0;
  hipEvent_t adj_ev[2] = {nullptr, nullptr};
  // overlapped stage launches of the adjoint loop (LRNDE_ADJ_OVERLAP): second stream, cross-stream events, device sync words
  hipStream_t adj_stream2 = nullptr; hipEvent_t adj_evA[2] = {nullptr, nullptr}, adj_evB[2] = {nullptr, nullptr};
  int* adj_sync = nullptr; int adj_launch_id = 0;
  int* adj_hstat = nullptr; int* adj_hstat_dev = nullptr; int adj_seq = 0;  // pinned progress word of the adjoint loop (host / device view)
  std::vector<float> last_ts;  // sol.t of the last node_forward (cotangent times of the adjoint)
  std::vector<int> series_idx; std::vector<float> series_t;  // the caller's view of that solution: save slots and times
  float last_t1 = 0.f; int last_i1 = 0;
  // backward workspace kept across calls: u(t1) of the recorded forward, k1 and the regulariser's gradient
  float* rec_gr = nullptr; size_t rec_n = 0;
  float rec_dt1 = 0.f, rec_eest = 0.f, rec_snum = 0.f, rec_sden = 0.f;  // the local step's dt and scalars (forward's)
  float loc_dt = 0.f, loc_eest = 0.f, loc_snum = 0.f, loc_sden = 0.f;    // the same of the LAST layer forward
  // arguments of the last lrnde_node_forward_record (what lrnde_node_backward_recorded differentiates)
  bool rec_valid = false; int rec_B = 0, rec_mode = 0, rec_reg_type = 0, rec_naccept = 0;
  unsigned long long rec_gen = 0;  // counts the recorded forwards of this handle (lrnde_record_generation)
  float rec_t0 = 0.f, rec_t2 = 0.f, rec_t1 = 0.f; lrnde_solve_opts rec_opts{};
  int wsNB = 0;  // tile width the workspace (partial-sum vectors) was sized for
  // workspace
  int wsB = 0;
  float* state = nullptr;  // 10 * B * D floats: ubuf[2], kfsal[2], ks[5], g6
  Ctrl* ctrl = nullptr;
  double* part = nullptr;      // send [2][nwg_global*PSTRIDE]
  double* part_rx = nullptr;   // recv (nranks > 1)
  double* pinit = nullptr;
  double* pinit_rx = nullptr;
  float* saveat_dev = nullptr;
  int saveat_cap = 0;
  float* tsaved_dev = nullptr;   // saved times: pinned host array (tsaved_host) as the device sees it
  float* tsaved_host = nullptr;
  int tsaved_cap = 0;
  // pinned host block the step prologue reports to (solve_progress): [0] the progress word, +64 B the final control block
  unsigned long long* prog_host = nullptr; unsigned long long* prog_dev = nullptr;
  lrnde_trace_row* trace_dev = nullptr;
  int trace_cap = 0;
  float* usave = nullptr;  // internal save slots for node_forward
  size_t usave_slots = 0, usave_slot_elems = 0;
  Ctrl* ctrl_host = nullptr;  // pinned [2]
  hipEvent_t ev_norm = nullptr;  // vec_norm's read-back
  void* cls_ws = nullptr; size_t cls_ws_bytes = 0; void* cls_host = nullptr;  // lrnde_classifier_ce workspace (device / pinned)
  // comm
  ncclComm_t comm = nullptr;           // RCCL communicator (one process per GPU)
  lrnde_local_comm* lcomm = nullptr;   // or: in-process local communicator (lrnde_hooks.h), never both
  int rank = 0, nranks = 1;
  // per-rank pre-reduction of the error-norm partial sums (StepArgs::prered): default for sharded handles; with
  // LRNDE_GATHER_TILES=1 in the environment every tile's partial is exchanged instead (the exact gather)
  bool prered = false;
  int* arrive = nullptr; double* tile_part = nullptr; double* tile_pinit = nullptr;
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evp[2] = {nullptr, nullptr};
  float last_ms = 0.f;
  bool time_solves = false;  // lrnde_last_solve_kernel_ms has been called once: solves bracket their kernels with events
  int last_launches = 0;
  // companion context on its own (non-blocking) stream: shares the packed weights, owns a second state workspace.  The
  // layer forward runs its local step there (and, recording, the regulariser's reverse sweep) WHILE the main solve
  // finishes [t1, t2] on the handle's stream: at B <= 512 a step kernel is 128 workgroups, half of the chip.
  lrnde_ctx* side = nullptr;
  bool is_side = false, side_busy = false, rec_gr_ready = false, overlap_off = false;
  // a recorded forward whose t1 came too late to enqueue the sweep beside the solve leaves it to the backward pass, which
  // enqueues it on the companion's stream once its first adjoint attempt is on the handle's (the host would otherwise
  // spend the sweep's ~30 launch calls inside the forward, with the device idle)
  bool sweep_pending = false; int sw_B = 0, sw_reg_type = 0; float sw_t1 = 0.f, sw_abstol = 0.f, sw_reltol = 0.f;
  std::function<int()> after_first_attempt;  // adj_solve_device calls it once, after enqueuing its first attempt
  hipEvent_t ev_side_local = nullptr, ev_side_sweep = nullptr;
  // host-side phase clock of the layer forward (lrnde_host_phases, diagnostics): time points of the call in flight, sums over calls
  std::chrono::steady_clock::time_point hp_t[8];
  double hp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long hp_n = 0;
  float* tail_copy_dst = nullptr; int tail_copy_slot = -1;  // lrnde_solve: the step that fills this save slot writes it to tail_copy_dst too
  std::function<int(int, hipEvent_t)> poll_hook;  // lrnde_solve calls it after every status poll (event: that poll's)
  // lrnde_solve calls it once its last report is in and sol.u[end] is being written to the caller's array by the queued
  // launches (StepArgs::also_dst), BEFORE its final synchronisation: work enqueued here follows the solve without a host
  // round trip and is covered by that synchronisation (lrnde_node_forward_record_ce: the classifier head)
  const float* adj_init_src = nullptr;  // adj_solve_device: its first launch also sets z = [this; 0] (k_adj_begin)
  std::function<int()> final_hook;
  bool final_hook_fired = false, last_u_end_done = false;
  bool adj_stage7_reused = false;   // the last stage-7 launch of the adjoint loop took y / h from stage 6's scratch set (its GEMM must too)
  int solver_alg = 0;        // lrnde_set_solver: 0 Tsit5 (k_step_q / k_step), 1 VCAB3, 2 VCABM3 (lrnde_adams.hpp)
  bool hung = false;         // a host loop waited LRNDE_SPIN_DEADLINE_S for a report while the queue stayed busy: only lrnde_destroy is safe
  bool reports_off = false;  // lrnde_set_reports(ctx, 0): the solve loop polls by copies (its fall-back when no report arrives)
  // lrnde_set_adjoint_trace: per-attempt (s, dt, EEst, accepted) rows of the next adjoint solves, host memory of the caller
  lrnde_trace_row* adj_trace = nullptr; int adj_trace_cap = 0; int adj_trace_n = 0;
  std::string err;
};

namespace {

// ---- diagnostic switches -------------------------------------------------------------------------
// One process-wide table: each entry starts from its environment variable (read once) and can be set from the host
// language with the hook lrnde_set_option(name, value) — which is how tests/test_gpu_switches.py runs every alternative
// path in the same process as the default one and holds the two to the same bits.
enum {
  OPT_NO_QTILE, OPT_QTILE_MAX_B, OPT_NO_FUSE, OPT_DENSE_COPY, OPT_NO_OVERLAP, OPT_NO_SDE_FAST, OPT_SDE_HOST_LOOP, OPT_NO_QVJP,
  OPT_ADJ_ERR_ONE_LAUNCH, OPT_ADJ_MU_FOLD, OPT_ADJ_OVERLAP, OPT_ADJ_HOST, OPT_VJP_QCOLS, OPT_PGRAD_TS, OPT_ADJ_NO_REUSE, OPT_NO_SDE_BWD_FUSED, OPT_SDE_NO_PERSIST, OPT_SDE_COOP_LAUNCH, OPT_SDE_PERSIST_STALL, OPT_SDE_HOST_INITDT, OPT_SDE_BWD_LDSACC, OPT_SDE_BWD_NO_DEFER, OPT_SDE_BWD_NO_RESIDENT, OPT_SDE_NO_MARCH, OPT_FEED_T, OPT_FEED_E, OPT_FEED_M, OPT_GATHER_TILES, OPT_FORCE_COMM, N_OPT
};
struct OptDef { const char* name; int dflt; bool flag; };   // flag: present in the environment = 1
const OptDef g_optdef[N_OPT] = {
    {"LRNDE_NO_QTILE", 0, true}, {"LRNDE_QTILE_MAX_B", 2048, false}, {"LRNDE_NO_FUSE", 0, true}, {"LRNDE_DENSE_COPY", 0, true},
    {"LRNDE_NO_OVERLAP", 0, true}, {"LRNDE_NO_SDE_FAST", 0, true}, {"LRNDE_SDE_HOST_LOOP", 0, true}, {"LRNDE_NO_QVJP", 0, true},
    {"LRNDE_ADJ_ERR_ONE_LAUNCH", 0, true}, {"LRNDE_ADJ_MU_FOLD", 0, true}, {"LRNDE_ADJ_OVERLAP", 0, false}, {"LRNDE_ADJ_HOST", 0, true}, {"LRNDE_VJP_QCOLS", 4, false}, {"LRNDE_PGRAD_TS", 0, false}, {"LRNDE_ADJ_NO_REUSE", 0, true}, {"LRNDE_NO_SDE_BWD_FUSED", 0, true}, {"LRNDE_SDE_NO_PERSIST", 0, true}, {"LRNDE_SDE_COOP_LAUNCH", 0, true}, {"LRNDE_SDE_PERSIST_STALL", 0, true}, {"LRNDE_SDE_HOST_INITDT", 0, true}, {"LRNDE_SDE_BWD_LDSACC", 0, true}, {"LRNDE_SDE_BWD_NO_DEFER", 0, true}, {"LRNDE_SDE_BWD_NO_RESIDENT", 0, true}, {"LRNDE_SDE_NO_MARCH", 0, true}, {"LRNDE_FEED_T", 3, false}, {"LRNDE_FEED_E", 1, false},
    {"LRNDE_FEED_M", 2, false}, {"LRNDE_GATHER_TILES", 0, true}, {"LRNDE_FORCE_COMM", 0, true}};
int g_opt[N_OPT];
bool g_opt_set[N_OPT];     // set by the hook: the environment no longer counts
std::once_flag g_opt_once;
void opt_init() {
  for (int i = 0; i < N_OPT; ++i) {
    const char* e = getenv(g_optdef[i].name);
    g_opt[i] = e ? (g_optdef[i].flag ? 1 : atoi(e)) : g_optdef[i].dflt;
  }
  if (const char* e = getenv("LRNDE_FEED")) sscanf(e, "%d,%d,%d", &g_opt[OPT_FEED_T], &g_opt[OPT_FEED_E], &g_opt[OPT_FEED_M]);
}
int opt(int i) {
  std::call_once(g_opt_once, opt_init);
  // (the two communicator switches are read when a communicator is made and tests flip them through the environment)
  if ((i == OPT_GATHER_TILES || i == OPT_FORCE_COMM) && !g_opt_set[i]) return getenv(g_optdef[i].name) != nullptr;
  return g_opt[i];
}

// batch columns per workgroup of the 4-column VJP kernel: LRNDE_VJP_QCOLS = 2 spreads a B <= 512 launch over all 256 CUs
inline int vjp_qcols(int B) {
  const int q = opt(OPT_VJP_QCOLS);
  return (q == 1 || q == 2) && B <= 256 * q ? q : QNB;
}

int fail(lrnde_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define HIPCHK(c, x)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess)                                                                     \
      return fail(c, LRNDE_HIP_ERROR, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_),     \
                  __FILE__, __LINE__);                                                        \
  } while (0)
#define NCCLCHK(c, x)                                                                         \
  do {                                                                                        \
    ncclResult_t r_ = (x);                                                                    \
    if (r_ != ncclSuccess)                                                                    \
      return fail(c, LRNDE_NCCL_ERROR, "%s failed: %s (%s:%d)", #x, ncclGetErrorString(r_),   \
                  __FILE__, __LINE__);                                                        \
  } while (0)

#define HPT(c, i) ((c)->hp_t[i] = std::chrono::steady_clock::now())
// for the host loops that spin on a report word in pinned memory: true once a wait has lasted 20 ms (then every 20 ms) — the
// caller then asks the runtime whether the queue is still alive.  `spin` is the caller's spin count of THIS wait.
inline bool spin_stalled(long spin) {
  thread_local std::chrono::steady_clock::time_point t_last;
  const auto now = std::chrono::steady_clock::now();
  if (spin <= 0x4000) { t_last = now; return false; }   // first check of this wait (the callers ask every 0x4000 spins): start the clock
  if (now - t_last < std::chrono::milliseconds(20)) return false;
  t_last = now;
  return true;
}

// ... and the bound on such a wait: the loops spin while hipStreamQuery says "not ready", which a hung queue says for ever.
// After LRNDE_SPIN_DEADLINE_S of one wait the call fails, the handle is marked (check_ready refuses it) and nothing is
// restarted in place.
constexpr int LRNDE_SPIN_DEADLINE_S = 90;  // (above the local communicator's 60-s rendezvous timeout)
struct SpinDeadline {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void restart() { t0 = std::chrono::steady_clock::now(); }
  bool expired() const { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(LRNDE_SPIN_DEADLINE_S); }
};
#define LRNDE_HUNG(c, what) ((c)->hung = true, fail((c), LRNDE_HIP_ERROR, "%s: no report for %d s while the queue stayed busy (hung queue); destroy the handle", (what), LRNDE_SPIN_DEADLINE_S))

// a batch-sharded handle: its collectives run (RCCL or the in-process local communicator, lrnde_comm.hpp)
inline bool sharded(const lrnde_ctx* c) { return c->comm != nullptr || c->lcomm != nullptr; }
// every collective of the library: SUM all-reduce of `count` doubles / floats on the handle's stream (send may equal recv)
int comm_allreduce(lrnde_ctx* c, const void* send, void* recv, size_t count, bool is_double) {
  if (c->comm) {
    NCCLCHK(c, ncclAllReduce(send, recv, count, is_double ? ncclDouble : ncclFloat, ncclSum, c->comm, c->stream));
    return LRNDE_OK;
  }
  if (c->lcomm) {
    const int rc = lc_allreduce(c->lcomm, c->rank, c->stream, send, recv, count, is_double);
    if (rc) return fail(c, rc, "local communicator all-reduce failed (rank %d of %d): %s", c->rank, c->nranks,
                        rc == LRNDE_NCCL_ERROR ? "a peer did not arrive (timeout) or the communicator is broken" : "HIP error");
  }
  return LRNDE_OK;
}

inline int ceil16(int x) { return (x + 15) & ~15; }
inline int vecw(const lrnde_ctx* c) { return (c->desc.state_dim % 4 == 0) ? 4 : 1; }

// Which tile shape runs this batch: the 4-column family (lrnde_qtile.hpp) when there would be too
// few 16-column workgroups to fill the chip, the 16-column family otherwise.
bool use_qtile(const lrnde_ctx* c, int B) {
  const int qmax = opt(OPT_QTILE_MAX_B);  // two rounds of 4-column workgroups (97 us at B=2048) beat one of 16-column ones (114 us)
  const int D = c->desc.state_dim, H = c->desc.hidden_dim;
  // streaming path shape limits: one Dense-1 segment and one Dense-2 pass per wave
  const bool shape_ok = (D % 4 == 0) && (H <= 112) && (c->m.KQ1p / QSEG <= QNW) && (c->m.RG1 <= 2) &&
                        (c->m.RG2 <= 2 * QNW);
  const bool no_qtile = opt(OPT_NO_QTILE) != 0;
  return shape_ok && ((double)B * D * 40.0 < 2147483000.0) && B <= qmax && !no_qtile;
}
inline int tile_nb(const lrnde_ctx* c, int B) { return use_qtile(c, B) ? QNB : NB; }

int ensure_workspace(lrnde_ctx* c, int B) {
  const size_t n = (size_t)B * c->desc.state_dim;
  const int nwg = (B + QNB - 1) / QNB;  // sized for the finer tile: covers both shapes
  const int nwg_global = nwg * c->nranks;
  if (B != c->wsB) {
    if (c->state) HIPCHK(c, hipFree(c->state));
    if (c->part) HIPCHK(c, hipFree(c->part));
    if (c->part_rx) HIPCHK(c, hipFree(c->part_rx));
    if (c->pinit) HIPCHK(c, hipFree(c->pinit));
    if (c->pinit_rx) HIPCHK(c, hipFree(c->pinit_rx));
    if (c->tile_part) HIPCHK(c, hipFree(c->tile_part));
    if (c->tile_pinit) HIPCHK(c, hipFree(c->tile_pinit));
    c->state = nullptr; c->part = c->part_rx = c->pinit = c->pinit_rx = c->tile_part = c->tile_pinit = nullptr;
    HIPCHK(c, hipMalloc(&c->state, sizeof(float) * n * 10));
    const size_t pb = sizeof(double) * 2 * (size_t)nwg_global * PSTRIDE;
    HIPCHK(c, hipMalloc(&c->part, pb));
    HIPCHK(c, hipMalloc(&c->pinit, pb));
    HIPCHK(c, hipMemsetAsync(c->part, 0, pb, c->stream));
    HIPCHK(c, hipMemsetAsync(c->pinit, 0, pb, c->stream));
    if (sharded(c)) {  // nranks > 1 (or LRNDE_FORCE_COMM: the same path on one rank)
      HIPCHK(c, hipMalloc(&c->part_rx, pb));
      HIPCHK(c, hipMalloc(&c->pinit_rx, pb));
      HIPCHK(c, hipMemsetAsync(c->part_rx, 0, pb, c->stream));
      HIPCHK(c, hipMemsetAsync(c->pinit_rx, 0, pb, c->stream));
      if (c->prered) {
        const size_t tb = sizeof(double) * 2 * (size_t)nwg * PSTRIDE;
        HIPCHK(c, hipMalloc(&c->tile_part, tb));
        HIPCHK(c, hipMalloc(&c->tile_pinit, tb));
        if (!c->arrive) HIPCHK(c, hipMalloc(&c->arrive, sizeof(int) * 4));
        HIPCHK(c, hipMemsetAsync(c->arrive, 0, sizeof(int) * 4, c->stream));
      }
    }
    c->wsB = B;
  }
  if (!c->ctrl) HIPCHK(c, hipMalloc(&c->ctrl, sizeof(Ctrl) * 2));
  if (!c->ctrl_host) HIPCHK(c, hipHostMalloc(&c->ctrl_host, sizeof(Ctrl) * 2));
  return LRNDE_OK;
}

// force_nb: the SDE kernels' fixed tile shape; they exchange per-tile partials (no pre-reduction)
void fill_args(lrnde_ctx* c, StepArgs& a, int B, int force_nb = 0) {
  memset(&a, 0, sizeof(a));
  const size_t n = (size_t)B * c->desc.state_dim;
  const int nb = force_nb ? force_nb : tile_nb(c, B);
  const int nwg = (B + nb - 1) / nb;
  a.m = c->m;
  a.state = c->state; a.n_local = (long)n;
  const bool no_fuse = opt(OPT_NO_FUSE) != 0;
  a.fused = (c->desc.state_dim % 16 == 0) && ((double)n * 40.0 < 2147483000.0) && !no_fuse;
  a.ubuf[0] = c->state; a.ubuf[1] = c->state + n;
  a.kfsal[0] = c->state + 2 * n; a.kfsal[1] = c->state + 3 * n;
  for (int i = 0; i < 5; ++i) a.ks[i] = c->state + (4 + i) * n;
  a.g6 = c->state + 9 * n;
  a.B = B;
  a.nwg_global = nwg * c->nranks;
  a.wg_offset = nwg * c->rank;
  if (sharded(c) && c->prered && !force_nb) {
    a.prered = 1; a.arrive = c->arrive; a.tile_part = c->tile_part; a.tile_pinit = c->tile_pinit;
    a.nwg_global = c->nranks; a.wg_offset = c->rank;
  }
  a.n_global = (double)c->desc.state_dim * (double)B * (double)c->nranks;
  a.ctrl = c->ctrl;
  a.part_send = c->part;
  a.part_recv = sharded(c) ? c->part_rx : c->part;
  a.pinit_send = c->pinit;
  a.pinit_recv = sharded(c) ? c->pinit_rx : c->pinit;
}

size_t smem_q(const lrnde_ctx* c) { return smem_bytes_q(c->m.KQ1p, c->m.KQ2p, c->m.RG1, c->m.RG2); }

template <class K> int launch_tile_kernel(lrnde_ctx* c, K kern, int B, const StepArgs& a) {
  const int nwg = (B + NB - 1) / NB;
  const size_t sm = smem_bytes(c->m.Dp, c->m.Hp);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), sm, c->stream, a);
  HIPCHK(c, hipGetLastError());
  return LRNDE_OK;
}

int launch_step(lrnde_ctx* c, int B, const StepArgs& a, int j, bool spec = false) {
  if (use_qtile(c, B)) {
    const int nq = (B + QNB - 1) / QNB;
    const size_t smq = smem_q(c) + (size_t)9 * c->m.KQ1p * 4 * 16 + 64 * 16 + 16;  // + the LDS-resident stage operands (both candidate (uprev, k1) pairs, k2..k6; one padded row group of slack)
    // KT: real k-quads in the last Dense-2 stream block (lrnde_qtile.hpp); 1 for H = 97..100, else the generic form
    const bool kt1 = (c->desc.hidden_dim + 3) / 4 == (QSB2 - 1) * QSQ + 1;
    if (kt1) {
      if (spec) hipLaunchKernelGGL((k_step_q<true, 1>), dim3(nq), dim3(QNT), smq, c->stream, a, j);
      else hipLaunchKernelGGL((k_step_q<false, 1>), dim3(nq), dim3(QNT), smq, c->stream, a, j);
    } else {
      if (spec) hipLaunchKernelGGL((k_step_q<true, 4>), dim3(nq), dim3(QNT), smq, c->stream, a, j);
      else hipLaunchKernelGGL((k_step_q<false, 4>), dim3(nq), dim3(QNT), smq, c->stream, a, j);
    }
    HIPCHK(c, hipGetLastError());
    return LRNDE_OK;
  }
  const int nwg = (B + NB - 1) / NB;
  const size_t sm = smem_bytes(c->m.Dp, c->m.Hp);
  if (vecw(c) == 4) {
    if (spec) hipLaunchKernelGGL((k_step<4, true>), dim3(nwg), dim3(NT), sm, c->stream, a, j);
    else hipLaunchKernelGGL((k_step<4, false>), dim3(nwg), dim3(NT), sm, c->stream, a, j);
  } else {
    if (spec) hipLaunchKernelGGL((k_step<1, true>), dim3(nwg), dim3(NT), sm, c->stream, a, j);
    else hipLaunchKernelGGL((k_step<1, false>), dim3(nwg), dim3(NT), sm, c->stream, a, j);
  }
  HIPCHK(c, hipGetLastError());
  return LRNDE_OK;
}

// exchange of the per-tile fp64 partial sums between ranks: ONE all-reduce (sum) of a vector
// in which every rank has zeros outside its own segment, so the result is the exact gather.
int exchange(lrnde_ctx* c, double* send, double* recv, size_t count) {
  if (!sharded(c)) return LRNDE_OK;
  return comm_allreduce(c, send, recv, count, true);
}

int run_init(lrnde_ctx* c, int B, const StepArgs& a) {
  int rc;
  const size_t cnt = (size_t)a.nwg_global * PSTRIDE;
  const bool qt = use_qtile(c, B);
  const int nq = (B + QNB - 1) / QNB;
  if (qt) { hipLaunchKernelGGL(k_init1_q, dim3(nq), dim3(QNT), smem_q(c), c->stream, a); rc = LRNDE_OK; HIPCHK(c, hipGetLastError()); }
  else if (vecw(c) == 4) rc = launch_tile_kernel(c, k_init1<4>, B, a);
  else rc = launch_tile_kernel(c, k_init1<1>, B, a);
  if (rc) return rc;
  if ((rc = exchange(c, c->pinit, c->pinit_rx, cnt))) return rc;
  if (qt) { hipLaunchKernelGGL(k_init2_q, dim3(nq), dim3(QNT), smem_q(c), c->stream, a); rc = LRNDE_OK; HIPCHK(c, hipGetLastError()); }
  else if (vecw(c) == 4) rc = launch_tile_kernel(c, k_init2<4>, B, a);
  else rc = launch_tile_kernel(c, k_init2<1>, B, a);
  if (rc) return rc;
  return exchange(c, c->pinit + cnt, c->pinit_rx + cnt, cnt);
}

int set_smem_attr() {
  static bool done = false;
  if (done) return 0;
  const int maxb = 160 * 1024;
  hipFuncSetAttribute((const void*)k_step<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_step<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_step<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_step<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_init1<4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_init1<1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_init2<4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_init2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_rhs<4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_rhs<1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_step_q<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_step_q<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_step_q<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_step_q<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_init1_q, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_init2_q, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_rhs_q, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_vjp<4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_vjp<1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_sde_step<4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_sde_step<1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_sde_rkmil<4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  hipFuncSetAttribute((const void*)k_sde_rkmil<1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
  done = true;
  return 0;
}

int check_ready(lrnde_ctx* c, int B) {
  if (!c) return LRNDE_BADARG;
  if (c->hung) return fail(c, LRNDE_HIP_ERROR, "the handle's queue stopped making progress in an earlier call: destroy the handle");
  if (!c->have_params) return fail(c, LRNDE_BADARG, "lrnde_set_params has not been called");
  if (B <= 0) return fail(c, LRNDE_BADARG, "batch must be positive (got %d)", B);
  HIPCHK(c, hipSetDevice(c->device));
  return LRNDE_OK;
}

void stats_from_ctrl(const Ctrl& k, lrnde_stats* st) {
  st->retcode = (k.status == ST_DONE || k.status == ST_RUNNING) ? LRNDE_OK : k.status;
  st->nf = k.nf; st->naccept = k.naccept; st->nreject = k.nreject; st->iters = k.iter;
  st->nsaved = k.nsaved; st->t_final = k.t; st->dt_final = k.dt; st->eest_last = k.eest_last;
  st->dt_init = k.dt_init;
}

// ---- companion context (lrnde_ctx::side) ----
// A shallow clone: same descriptor and packed-weight pointers, own stream, state workspace, partial sums, control blocks
// and backward scratch.  Unsharded handles only (a communicator's collectives must stay in one stream order).
int side_get(lrnde_ctx* c, int B, lrnde_ctx** out) {
  if (!c->side) {
    lrnde_ctx* s = new lrnde_ctx();
    s->is_side = true; s->device = c->device;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_side_local, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_side_sweep, hipEventDisableTiming) != hipSuccess) {
      delete s;
      return fail(c, LRNDE_HIP_ERROR, "companion stream / events could not be created");
    }
    c->side = s;
  }
  lrnde_ctx* s = c->side;
  s->desc = c->desc; s->m = c->m; s->have_params = c->have_params;
  s->W1p = c->W1p; s->W2p = c->W2p; s->w1t = c->w1t; s->b1 = c->b1; s->w2t = c->w2t; s->b2 = c->b2;
  s->W1q = c->W1q; s->W2q = c->W2q; s->V1p = c->V1p; s->U2p = c->U2p; s->V1q = c->V1q; s->U2q = c->U2q;
  const int rc = ensure_workspace(s, B);
  if (rc) { c->err = s->err; return rc; }
  *out = s;
  return LRNDE_OK;
}
// nothing of the companion's is in flight after this (before the weights are repacked, a new forward, destroy)
int side_quiesce(lrnde_ctx* c) {
  if (c->side && c->side_busy) {
    HIPCHK(c, hipStreamSynchronize(c->side->stream));
    c->side_busy = false;
  }
  return LRNDE_OK;
}

}  // namespace
namespace {  // (defined in lrnde_adams.hpp, behind the vector helpers)
int adams_solve(lrnde_ctx* c, const float* u0, int32_t B, float t0, float t1, const lrnde_solve_opts* o,
                const float* saveat_host, int32_t nsave, float* u_saved, float* t_saved_host, int32_t cap_saved,
                lrnde_stats* st, lrnde_trace_row* trace_host, int32_t cap_trace);
}

extern "C" {

const char* lrnde_version(void) { return "lrnde-mi355x 0.1 (gfx950)"; }

size_t lrnde_param_count(const lrnde_model_desc* d) {
  if (!d) return 0;
  const size_t D = d->state_dim, H = d->hidden_dim, td = d->time_dep ? 1 : 0;
  return H * (D + td) + H + D * (H + td) + D;
}

const char* lrnde_last_error(const lrnde_ctx* c) { return c ? c->err.c_str() : "null context"; }

int lrnde_create(lrnde_ctx** out, const lrnde_model_desc* d, int device, void* stream) {
  if (!out || !d) return LRNDE_BADARG;
  *out = nullptr;
  if (d->state_dim <= 0 || d->hidden_dim <= 0 || d->act < 0 || d->act > 2) return LRNDE_BADARG;
  lrnde_ctx* c = new lrnde_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  c->desc = *d;
  const int Dp = ceil16(d->state_dim), Hp = ceil16(d->hidden_dim);
  if (smem_bytes(Dp, Hp) > 160 * 1024) {
    delete c;
    return LRNDE_UNSUPPORTED;  // state tile does not fit LDS
  }
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { delete c; return LRNDE_HIP_ERROR; }
  set_smem_attr();
  ModelDev& m = c->m;
  m.D = d->state_dim; m.H = d->hidden_dim; m.Dp = Dp; m.Hp = Hp;
  m.MT1 = Hp / 16; m.KG1 = Dp / 16; m.MT2 = Dp / 16; m.KG2 = Hp / 16;
  m.act = d->act; m.td = d->time_dep ? 1 : 0;
  const size_t nW1 = (size_t)(((Hp / 16 + TG - 1) / TG) * TG) * (Dp / 16) * 256;
  const size_t nW2 = (size_t)(Dp / 16) * (((Hp / 16 + SEGK - 1) / SEGK) * SEGK) * 256;
  bool ok = hipMalloc(&c->W1p, sizeof(float) * nW1) == hipSuccess &&
            hipMalloc(&c->W2p, sizeof(float) * nW2) == hipSuccess &&
            hipMalloc(&c->V1p, sizeof(float) * nW1) == hipSuccess &&
            hipMalloc(&c->U2p, sizeof(float) * nW2) == hipSuccess &&
            hipMalloc(&c->w1t, sizeof(float) * Hp) == hipSuccess &&
            hipMalloc(&c->b1, sizeof(float) * Hp) == hipSuccess &&
            hipMalloc(&c->w2t, sizeof(float) * Dp) == hipSuccess &&
            hipMalloc(&c->b2, sizeof(float) * Dp) == hipSuccess &&
            hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
            hipEventCreate(&c->evp[0]) == hipSuccess && hipEventCreate(&c->evp[1]) == hipSuccess;
  m.KQ1p = (((d->state_dim + 3) / 4 + QSEG - 1) / QSEG) * QSEG;  // whole canonical segments
  m.KQ2p = QSEG;                                                  // H <= 112 (q-tile eligibility)
  m.RG1 = (d->hidden_dim + 63) / 64;
  m.RG2 = (d->state_dim + 63) / 64;
  ok = ok && hipMalloc(&c->W1q, sizeof(float) * (size_t)m.RG1 * m.KQ1p * 256) == hipSuccess &&
       hipMalloc(&c->W2q, sizeof(float) * (size_t)m.RG2 * m.KQ2p * 256) == hipSuccess &&
       hipMalloc(&c->V1q, sizeof(float) * (size_t)m.RG1 * m.KQ1p * 256) == hipSuccess &&
       hipMalloc(&c->U2q, sizeof(float) * (size_t)m.RG2 * m.KQ2p * 256) == hipSuccess;
  if (!ok) { lrnde_destroy(c); return LRNDE_HIP_ERROR; }
  m.W1q = c->W1q; m.W2q = c->W2q;
  m.W1p = reinterpret_cast<const f32x4*>(c->W1p);
  m.W2p = reinterpret_cast<const f32x4*>(c->W2p);
  m.w1t = c->w1t; m.b1 = c->b1; m.w2t = c->w2t; m.b2 = c->b2;
  *out = c;
  return LRNDE_OK;
}

int lrnde_destroy(lrnde_ctx* c) {
  if (!c) return LRNDE_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream); else hipDeviceSynchronize();
  if (c->side) {  // the companion borrows the packed weights: drop the borrowed pointers, free what it owns
    lrnde_ctx* s = c->side;
    hipStream_t ss = s->stream;
    hipStreamSynchronize(ss);
    s->W1p = s->W2p = s->w1t = s->b1 = s->w2t = s->b2 = s->W1q = s->W2q = s->V1p = s->U2p = s->V1q = s->U2q = nullptr;
    lrnde_destroy(s);
    hipStreamDestroy(ss);
    if (c->ev_side_local) hipEventDestroy(c->ev_side_local);
    if (c->ev_side_sweep) hipEventDestroy(c->ev_side_sweep);
    c->side = nullptr;
  }
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->adj_part_host) hipHostFree(c->adj_part_host);
  void* ptrs[] = {c->dense, c->dense_t, c->dense_dt, c->adj, c->adj_part, c->V1p, c->U2p, c->V1q, c->U2q, c->bw_y, c->bw_h, c->bw_dp, c->bw_da, c->W1q, c->W2q, c->W1p, c->W2p, c->w1t, c->b1, c->w2t, c->b2, c->state, c->ctrl, c->part,
                  c->part_rx, c->pinit, c->pinit_rx, c->arrive, c->tile_part, c->tile_pinit, c->rec_gr, c->saveat_dev, c->trace_dev,
                  c->usave};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->ctrl_host) hipHostFree(c->ctrl_host);
  if (c->tsaved_host) hipHostFree(c->tsaved_host);
  if (c->prog_host) hipHostFree(c->prog_host);
  if (c->adj_ctl) hipFree(c->adj_ctl);
  if (c->adj_ctl_host) hipHostFree(c->adj_ctl_host);
  if (c->adj_hstat) hipHostFree(c->adj_hstat);
  if (c->adj_stream2) hipStreamDestroy(c->adj_stream2);
  for (int i = 0; i < 2; ++i) { if (c->adj_evA[i]) hipEventDestroy(c->adj_evA[i]); if (c->adj_evB[i]) hipEventDestroy(c->adj_evB[i]); }
  if (c->adj_sync) hipFree(c->adj_sync);
  if (c->adj_ipart) hipFree(c->adj_ipart);
  if (c->adj_stops) hipFree(c->adj_stops);
  if (c->adj_ev[0]) hipEventDestroy(c->adj_ev[0]);
  if (c->adj_ev[1]) hipEventDestroy(c->adj_ev[1]);
  if (c->cls_ws) hipFree(c->cls_ws);
  if (c->cls_host) hipHostFree(c->cls_host);
  if (c->ev_norm) hipEventDestroy(c->ev_norm);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->evp[0]) hipEventDestroy(c->evp[0]);
  if (c->evp[1]) hipEventDestroy(c->evp[1]);
  delete c;
  return LRNDE_OK;
}

int lrnde_set_params(lrnde_ctx* c, const float* p, size_t n) {
  if (!c || !p) return LRNDE_BADARG;
  if (n != lrnde_param_count(&c->desc))
    return fail(c, LRNDE_BADARG, "parameter count %zu != expected %zu", n, lrnde_param_count(&c->desc));
  HIPCHK(c, hipSetDevice(c->device));
  { const int rq = side_quiesce(c); if (rq) return rq; }
  c->rec_gr_ready = false; c->sweep_pending = false;
  const ModelDev& m = c->m;
  hipLaunchKernelGGL(k_pack, dim3(256), dim3(256), 0, c->stream, p, m.D, m.H, m.td, m.Dp, m.Hp, c->W1p,
                     c->w1t, c->b1, c->W2p, c->w2t, c->b2);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_pack_q, dim3(256), dim3(256), 0, c->stream, p, m.D, m.H, m.td, m.KQ1p, m.KQ2p, m.RG1, m.RG2,
                     c->W1q, c->W2q);
  hipLaunchKernelGGL(k_pack_t, dim3(256), dim3(256), 0, c->stream, p, m.D, m.H, m.td, m.Dp, m.Hp, c->V1p, c->U2p);
  hipLaunchKernelGGL(k_pack_tq, dim3(256), dim3(256), 0, c->stream, p, m.D, m.H, m.td, m.KQ1p, m.KQ2p, m.RG1, m.RG2,
                     c->V1q, c->U2q);
  HIPCHK(c, hipGetLastError());
  c->have_params = true;
  return LRNDE_OK;
}

int lrnde_set_solver(lrnde_ctx* c, int32_t alg) {
  if (!c) return LRNDE_BADARG;
  if (alg < 0 || alg > 2) return fail(c, LRNDE_BADARG, "solver must be 0 (Tsit5), 1 (VCAB3) or 2 (VCABM3)");
  c->solver_alg = alg;
  c->rec_valid = false;
  return LRNDE_OK;
}

int lrnde_rhs(lrnde_ctx* c, const float* u, float t, int32_t B, float* du) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!u || !du) return fail(c, LRNDE_BADARG, "null state pointer");
  StepArgs a{};
  memset(&a, 0, sizeof(a));
  a.m = c->m; a.B = B;
  if (use_qtile(c, B)) {
    hipLaunchKernelGGL(k_rhs_q, dim3((B + QNB - 1) / QNB), dim3(QNT), smem_q(c), c->stream, a, u, t, du);
    HIPCHK(c, hipGetLastError());
    return LRNDE_OK;
  }
  const int nwg = (B + NB - 1) / NB;
  const size_t sm = smem_bytes(c->m.Dp, c->m.Hp);
  if (vecw(c) == 4) hipLaunchKernelGGL(k_rhs<4>, dim3(nwg), dim3(NT), sm, c->stream, a, u, t, du);
  else hipLaunchKernelGGL(k_rhs<1>, dim3(nwg), dim3(NT), sm, c->stream, a, u, t, du);
  HIPCHK(c, hipGetLastError());
  return LRNDE_OK;
}

int lrnde_init_dt(lrnde_ctx* c, const float* u0, int32_t B, float t0, float tend, float abstol,
                  float reltol, float* k1, float* dt_host) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!u0 || !dt_host) return fail(c, LRNDE_BADARG, "null pointer");
  if (!(tend > t0)) return fail(c, LRNDE_BADARG, "tspan must be increasing");
  if ((rc = ensure_workspace(c, B))) return rc;
  c->rec_valid = false;
  StepArgs a{};
  fill_args(c, a, B);
  a.t0 = t0; a.t1 = tend; a.abstol = abstol; a.reltol = reltol; a.mode = MODE_SINGLE_INIT_DT;
  const size_t n = (size_t)B * c->desc.state_dim;
  HIPCHK(c, hipMemcpyAsync(a.ubuf[0], u0, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, c->stream, c->ctrl, t0, 0.f, 0, 0);
  if ((rc = run_init(c, B, a))) return rc;
  // the first-step prologue computes dt; run it through the single-step kernel path would cost a
  // step, so evaluate the (deterministic) formula on the host from the reduced partial sums.
  const size_t cnt = (size_t)a.nwg_global * PSTRIDE;
  std::vector<double> h(2 * cnt);
  HIPCHK(c, hipMemcpyAsync(h.data(), a.pinit_recv, sizeof(double) * 2 * cnt, hipMemcpyDeviceToHost, c->stream));
  if (k1) HIPCHK(c, hipMemcpyAsync(k1, a.kfsal[0], sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // same fixed reduction order as reduce_partials (lane-strided, then shuffle tree)
  auto reduce = [&](const double* p, double out[3]) {
    double lanes[3][64];
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 3; ++q) {
      double s = 0.0;
      for (int i = l; i < a.nwg_global; i += 64) s += p[(size_t)i * PSTRIDE + q];
      lanes[q][l] = s;
    }
    for (int q = 0; q < 3; ++q) {
      for (int off = 32; off > 0; off >>= 1)
        for (int l = 0; l < off; ++l) lanes[q][l] += lanes[q][l + off];
      out[q] = lanes[q][0];
    }
  };
  double s1[3], s2[3];
  reduce(h.data(), s1);
  reduce(h.data() + cnt, s2);
  {
    const float dtmax = tend - t0;
    const float d0 = (float)sqrt(s1[0] / a.n_global), d1 = (float)sqrt(s1[1] / a.n_global);
    float dt0 = ((double)d0 < 1e-5 || (double)d1 < 1e-5) ? 1e-6f : (d0 / d1) / 100.0f;
    dt0 = fminf(dt0, dtmax);
    const float d2 = (float)sqrt(s2[0] / a.n_global) / dt0;
    const float maxd = fmaxf(d1, d2);
    float dt1;
    if ((double)maxd <= 1e-15) dt1 = fmaxf(1e-6f, dt0 * 1e-3f);
    else {
      const float l10 = (float)log10((double)maxd);
      const float e = (-(2.0f + l10)) / 5.0f;
      dt1 = (float)pow(10.0, (double)e);
    }
    *dt_host = fminf(fminf(100.0f * dt0, dt1), dtmax);
  }
  return LRNDE_OK;
}

int lrnde_perform_step(lrnde_ctx* c, const float* uprev, const float* k1, int32_t B, float t,
                       float dt, float abstol, float reltol, float* u, float* k7, float* eest_host,
                       float* reg_error_host, float* reg_stiff_host) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!uprev || !k1) return fail(c, LRNDE_BADARG, "null state pointer");
  if ((rc = ensure_workspace(c, B))) return rc;
  c->rec_valid = false;  // the state workspace a recorded forward left for its backward is overwritten
  StepArgs a{};
  fill_args(c, a, B);
  a.t0 = t; a.t1 = t + 1.0f; a.abstol = abstol; a.reltol = reltol; a.mode = MODE_SINGLE_GIVEN_DT;
  a.want_stiff = 1; a.maxiters = 1;
  const size_t n = (size_t)B * c->desc.state_dim;
  HIPCHK(c, hipMemcpyAsync(a.ubuf[0], uprev, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(a.kfsal[0], k1, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, c->stream, c->ctrl, t, dt, 0, 0);
  if ((rc = launch_step(c, B, a, 0))) return rc;
  const size_t cnt = (size_t)a.nwg_global * PSTRIDE;
  if ((rc = exchange(c, c->part + cnt, c->part_rx + cnt, cnt))) return rc;
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, c->stream, a, 1);
  HIPCHK(c, hipGetLastError());
  if (u) HIPCHK(c, hipMemcpyAsync(u, a.ubuf[1], sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  if (k7) HIPCHK(c, hipMemcpyAsync(k7, a.kfsal[1], sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->ctrl_host, c->ctrl + 1, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (eest_host) *eest_host = c->ctrl_host[0].eest_last;
  if (reg_error_host) *reg_error_host = c->ctrl_host[0].reg_error;
  if (reg_stiff_host) *reg_stiff_host = c->ctrl_host[0].reg_stiff;
  return LRNDE_OK;
}

int lrnde_solve(lrnde_ctx* c, const float* u0, int32_t B, float t0, float t1,
                const lrnde_solve_opts* o, const float* saveat_host, int32_t nsave, float* u_saved,
                float* t_saved_host, int32_t cap_saved, lrnde_stats* st, lrnde_trace_row* trace_host,
                int32_t cap_trace) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!u0 || !o || !st) return fail(c, LRNDE_BADARG, "null pointer");
  memset(st, 0, sizeof(*st));
  if (!(t1 > t0)) return fail(c, LRNDE_BADARG, "tspan must be increasing");
  if (nsave < 0 || (nsave > 0 && !saveat_host)) return fail(c, LRNDE_BADARG, "bad saveat");
  for (int i = 1; i < nsave; ++i)
    if (!(saveat_host[i] >= saveat_host[i - 1])) return fail(c, LRNDE_BADARG, "saveat must be ascending");
  if (cap_saved > 0 && !u_saved) return fail(c, LRNDE_BADARG, "null save buffer");
  if ((rc = ensure_workspace(c, B))) return rc;
  if (!c->dense_on) c->rec_valid = false;
  if (c->solver_alg != 0)   // n.solver = VCAB3() / VCABM3() (experiments/src/construct.jl:154-164)
    return adams_solve(c, u0, B, t0, t1, o, saveat_host, nsave, u_saved, t_saved_host, cap_saved, st, trace_host, cap_trace);
  const size_t n = (size_t)B * c->desc.state_dim;
  if (nsave > c->saveat_cap) {
    if (c->saveat_dev) HIPCHK(c, hipFree(c->saveat_dev));
    HIPCHK(c, hipMalloc(&c->saveat_dev, sizeof(float) * nsave));
    c->saveat_cap = nsave;
  }
  if (cap_saved > c->tsaved_cap || !c->tsaved_host) {
    if (c->tsaved_host) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipHostFree(c->tsaved_host)); }
    c->tsaved_host = nullptr; c->tsaved_cap = 0;
    const int cap = cap_saved > 16 ? cap_saved : 16;
    HIPCHK(c, hipHostMalloc(&c->tsaved_host, sizeof(float) * cap, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->tsaved_dev, c->tsaved_host, 0));
    c->tsaved_cap = cap;
  }
  if (!c->prog_host) {
    HIPCHK(c, hipHostMalloc(&c->prog_host, PROG_RING * 8 + sizeof(Ctrl), hipHostMallocMapped));
    memset(c->prog_host, 0, PROG_RING * 8 + sizeof(Ctrl));
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->prog_dev, c->prog_host, 0));
  }
  if (trace_host && cap_trace > c->trace_cap) {
    if (c->trace_dev) HIPCHK(c, hipFree(c->trace_dev));
    HIPCHK(c, hipMalloc(&c->trace_dev, sizeof(lrnde_trace_row) * cap_trace));
    c->trace_cap = cap_trace;
  }
  StepArgs a{};
  fill_args(c, a, B);
  a.t0 = t0; a.t1 = t1; a.abstol = o->abstol; a.reltol = o->reltol;
  a.maxiters = o->maxiters; a.save_everystep = o->save_everystep; a.exact_pow = o->exact_pow;
  a.want_stiff = 0; a.mode = MODE_SOLVE;
  if (c->dense_on) {
    a.dense = c->dense; a.dense_t = c->dense_t; a.dense_dt = c->dense_dt; a.dense_cap = c->dense_cap;
    const bool no_direct = opt(OPT_DENSE_COPY) != 0;  // diagnostic: round 1's prologue copy
    a.dense_direct = (use_qtile(c, B) && !no_direct) ? 1 : 0;
  }
  a.cap_saved = cap_saved; a.u_saved = u_saved; a.t_saved = c->tsaved_dev;
  // (round 2 enqueued a D2D copy of that slot once the last report was in: 30 us of host latency plus a blit kernel
  //  between the last launch and the caller's synchronisation)
  a.also_dst = c->tail_copy_dst; a.also_slot = c->tail_copy_dst ? c->tail_copy_slot : -1;
  a.trace = trace_host ? c->trace_dev : nullptr; a.cap_trace = trace_host ? cap_trace : 0;
  // saveat points at/before t0 are the start value (save_start), as in the oracle
  int skip = 0;
  while (skip < nsave && saveat_host[skip] <= t0) ++skip;
  a.nsave = nsave - skip;
  a.saveat = c->saveat_dev;
  SaveInit si;
  memset(&si, 0, sizeof(si));
  si.saveat = c->saveat_dev; si.tsaved = c->tsaved_dev;
  if (a.nsave > 8)
    HIPCHK(c, hipMemcpyAsync(c->saveat_dev, saveat_host + skip, sizeof(float) * a.nsave,
                             hipMemcpyHostToDevice, c->stream));
  else
    for (si.n = 0; si.n < a.nsave; ++si.n) si.v[si.n] = saveat_host[skip + si.n];   // written by k_solve_init
  int nsaved0 = 0;
  if (o->save_start) {
    if (cap_saved < 1) return fail(c, LRNDE_CAPACITY, "save buffer too small for save_start");
    HIPCHK(c, hipMemcpyAsync(u_saved, u0, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
    si.save_start = 1;
    nsaved0 = 1;
  }
  // the prologue reports to pinned host memory (solve_progress) and the loop below steers by those reports, one per
  // launch, read in launch order — the same sequence of decisions on every rank of a sharded run
  volatile unsigned long long* pw = c->prog_host;
  Ctrl* fin_host = reinterpret_cast<Ctrl*>(reinterpret_cast<char*>(c->prog_host) + PROG_RING * 8);
  for (int i = 0; i < PROG_RING; ++i) pw[i] = 0ull;
  a.prog = c->reports_off ? nullptr : c->prog_dev;
  a.fin_host = reinterpret_cast<Ctrl*>(reinterpret_cast<char*>(c->prog_dev) + PROG_RING * 8);
  if (c->time_solves) HIPCHK(c, hipEventRecord(c->ev0, c->stream));  // (an event is a marker packet in the queue: only on request)
  if (use_qtile(c, B)) {
    // (4-column family: the first init launch reads the caller's array, copies it to ubuf[0] and writes the control blocks)
    a.init_u0 = u0; a.init_fresh = 1; a.init_nsaved = nsaved0; a.init_si = si;
  } else {
    HIPCHK(c, hipMemcpyAsync(a.ubuf[0], u0, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_solve_init, dim3(1), dim3(1), 0, c->stream, c->ctrl, t0, nsaved0, si);
  }
  if ((rc = run_init(c, B, a))) return rc;
  a.init_u0 = nullptr; a.init_fresh = 0;
  HPT(c, 1);

  // Attempted steps are enqueued ahead of the device's decisions, steered by the per-launch reports (below); the copy-polled
  // chunk loop further down is the fall-back if no report ever arrives.  Launches beyond what a report makes certain are
  // speculative (k_step<.,true>): they find the solve finished and do nothing.
  const size_t cnt = (size_t)a.nwg_global * PSTRIDE;
  int j = 0, pending = -1, pending_j = 0, launches = 0, target = 4, nchunk = 0;
  bool done = false, word_ok = !c->reports_off;
  const long hard_cap = (long)o->maxiters + 8;
  if (word_ok) {
    // Report-driven feed: after the report of launch `seen - 1` keep enqueued what it says is still to come at the current
    // dt (an over-estimate while dt grows, exact for the last step, whose dt is clipped to t1 - t), the launch that will find
    // the solve finished, and never fewer than two launches beyond the reporting one — so the stream neither runs dry nor
    // ends with a tail of launches that have nothing to do.
    int seen = 0, rem = 3;
    while (!done) {
      // (rem + 1 is exact when dt stays put; while the controller still grows dt the estimate is high and every launch
      //  enqueued on its strength beyond the real end is a 6-us launch with nothing to do — seven of them per pass on the
      //  MNIST field.  Far from the end half the estimate plus two keeps the queue two launches deep at the least.)
      const int fT = opt(OPT_FEED_T), fE = opt(OPT_FEED_E), fM = opt(OPT_FEED_M);
      int ahead = rem <= fT ? rem + fE : rem / 2 + fE + 1;
      if (ahead < fM) ahead = fM;
      if (ahead > 16) ahead = 16;
      const int certain = seen + (rem > 1 ? rem / 2 : 1);  // launches beyond it carry the speculative kernel name
      for (const int want = seen + ahead; j < want; ++j) {
        if ((rc = launch_step(c, B, a, j, j >= certain))) return rc;
        ++launches;
        const size_t par = (size_t)((j + 1) & 1);
        if ((rc = exchange(c, c->part + par * cnt, c->part_rx + par * cnt, cnt))) return rc;
      }
      if (j > hard_cap + 64) break;
      // the report of launch `seen` (it carries seen + 1 as its launch count)
      volatile unsigned long long* slot = pw + (seen & (PROG_RING - 1));
      unsigned long long w = *slot;
      const SpinDeadline deadline;
      for (long spin = 1; (int)(w & 0xffffffull) != seen + 1; ++spin) {
        // bounded: a faulted queue must not hang the caller.  The query is kept for a report that is LATE (the runtime
        // answers it by putting a marker packet into the queue: asked every few thousand spins, one landed between two
        // steps whenever the queue was only a launch or two deep — a 6-us bubble each, five per pass at the end of a solve)
        if ((spin & 0x3fff) == 0 && spin_stalled(spin)) {
          const hipError_t qe = hipStreamQuery(c->stream);
          if (qe != hipSuccess && qe != hipErrorNotReady) return fail(c, LRNDE_HIP_ERROR, "solve loop: %s", hipGetErrorString(qe));
          if (qe == hipErrorNotReady && deadline.expired()) return LRNDE_HUNG(c, "solve loop");
          if (qe == hipSuccess) {
            w = *slot;
            if ((int)(w & 0xffffffull) != seen + 1) word_ok = false;  // the stream drained and no report came: poll by copies
            break;
          }
        }
        w = *slot;
      }
      if (!word_ok) break;
      ++seen;
      rem = (int)((w >> 48) & 0xffff);
      if (c->poll_hook && (rc = c->poll_hook((int)((w >> 32) & 0xffff), nullptr))) return rc;
      if ((int)((w >> 24) & 0xff) != (ST_RUNNING & 0xff)) done = true;
    }
    target = j;
  }
  HPT(c, 2);
  while (!done && !word_ok) {
    int ch = target - j;
    if (ch < 2) ch = 2;
    if (ch > 16) ch = 16;
    for (int i = 0; i < ch; ++i, ++j) {
      if ((rc = launch_step(c, B, a, j, j >= target))) return rc;
      ++launches;
      const size_t par = (size_t)((j + 1) & 1);
      if ((rc = exchange(c, c->part + par * cnt, c->part_rx + par * cnt, cnt))) return rc;
    }
    const int slot = (nchunk++) & 1;
    HIPCHK(c, hipMemcpyAsync(c->ctrl_host + slot, c->ctrl + (j & 1), sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->evp[slot], c->stream));
    if (pending >= 0) {
      HIPCHK(c, hipEventSynchronize(c->evp[pending]));
      const Ctrl& k = c->ctrl_host[pending];
      if (c->poll_hook && (rc = c->poll_hook(k.nsaved, c->evp[pending]))) return rc;
      if (k.status != ST_RUNNING) done = true;
      else if (k.dt > 0.f) {
        double est = ceil((double)(t1 - k.t) / (double)k.dt);
        if (est > 1e6) est = 1e6;
        // dt usually grows along the solve, so (t1-t)/dt over-estimates: count half of it as
        // "certainly needed"; launches beyond that carry the speculative kernel name
        const int tg = pending_j - 1 + (int)(est / 2);
        if (tg > target) target = tg;
      }
    }
    pending = slot; pending_j = j;
    if (target < j) target = j;  // never re-label launches already issued
    if (j > hard_cap + 64) break;
  }
  c->tail_copy_dst = nullptr;
  if (c->time_solves) HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  Ctrl fin;
  if (word_ok && done && c->final_hook && a.also_dst) {
    c->final_hook_fired = true;
    if ((rc = c->final_hook())) return rc;
  }
  if (word_ok && done) {  // the finished solve left its control block in host memory (solve_progress)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HPT(c, 3);
    fin = *fin_host;
  } else {
    HIPCHK(c, hipMemcpyAsync(c->ctrl_host, c->ctrl + (j & 1), sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fin = c->ctrl_host[0];
  }
  stats_from_ctrl(fin, st);
  if (fin.status == ST_RUNNING) st->retcode = LRNDE_MAXITERS;
  if (t_saved_host && fin.nsaved > 0) memcpy(t_saved_host, c->tsaved_host, sizeof(float) * fin.nsaved);
  if (trace_host) {
    int nt = fin.naccept + fin.nreject;
    if (nt > cap_trace) nt = cap_trace;
    if (nt > 0) HIPCHK(c, hipMemcpy(trace_host, c->trace_dev, sizeof(lrnde_trace_row) * nt, hipMemcpyDeviceToHost));
  }
  if (c->time_solves) hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1);
  c->last_launches = launches;
  if (st->retcode != LRNDE_OK)
    return fail(c, st->retcode, "solve stopped with retcode %d at t=%g (iter %d)", st->retcode, (double)fin.t, fin.iter);
  return LRNDE_OK;
}

}  // extern "C"
// `(n::NeuralODE)(x, ps, st)` for every mode, with or without a user `saveat` (src/layers/neural_ode.jl:56-116).
// user_sv / nuser: the layer's `saveat` kwarg (ascending); nuser == 0 is the default of :102-116 ([t2] / [t1, t2] / every
// step).  With a user saveat the :unbiased mode appends t1 for the solve and the returned series leaves every saved
// time equal to t1 out again (_CorrectedDESolution, src/utils.jl:31-33: `sol.u[t1 .!= sol.t]`); :biased draws t1 from
// the saved times but the last.  The series (the times the caller sees as sol.t) is kept in c->series_* for the caller
// and for lrnde_node_backward_recorded_ts.
static int step_reg_sweep(lrnde_ctx* c, const float* uprev, int32_t B, float t, float dt, float abstol, float reltol,
                          int32_t reg_type, float eest, float stiff_num, float stiff_den, float* gp);
static int node_forward_impl(lrnde_ctx* c, const float* x, int32_t B, float t0, float t2,
                             const lrnde_solve_opts* o, int32_t mode, int32_t reg_type, float t1_or_rand,
                             const float* user_sv, int nuser, float* u_end, float* reg_val_host, int32_t* nfe_host,
                             lrnde_stats* st, float* t1_used_host) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!x || !o || !u_end || !st) return fail(c, LRNDE_BADARG, "null pointer");
  // src/utils.jl:53-58 _check_valid_regularize
  if (mode < LRNDE_MODE_NONE || mode > LRNDE_MODE_BIASED)
    return fail(c, LRNDE_BADARG, "regularize must be one of (:none, :unbiased, :biased)");
  if (reg_type != LRNDE_REG_ERROR_ESTIMATE && reg_type != LRNDE_REG_STIFFNESS_ESTIMATE)
    return fail(c, LRNDE_BADARG, "regularize must be one of (:error_estimate, :stiffness_estimate)");
  const size_t n = (size_t)B * c->desc.state_dim;
  // a plain forward overwrites what a recorded one left for lrnde_node_backward_recorded (usave, last_ts, the state
  // workspace): the record is gone
  if (!c->dense_on) c->rec_valid = false;
  if ((rc = side_quiesce(c))) return rc;
  c->rec_gr_ready = false; c->sweep_pending = false;
  lrnde_solve_opts oo = *o;
  size_t need = 3;
  if (mode == LRNDE_MODE_BIASED) need = (size_t)(oo.maxiters < 510 ? oo.maxiters + 2 : 512);
  if (nuser > 0) {
    if (!user_sv) return fail(c, LRNDE_BADARG, "null saveat");
    for (int i = 1; i < nuser; ++i) if (!(user_sv[i] >= user_sv[i - 1])) return fail(c, LRNDE_BADARG, "saveat must be ascending");
    need = (size_t)nuser + 3;
  }
  c->series_idx.clear(); c->series_t.clear();
  if (c->usave_slots < need || c->usave_slot_elems != n) {
    if (c->usave) HIPCHK(c, hipFree(c->usave));
    c->usave = nullptr;
    HIPCHK(c, hipMalloc(&c->usave, sizeof(float) * n * need));
    c->usave_slots = need; c->usave_slot_elems = n;
  }
  if (reg_val_host) *reg_val_host = 0.0f;
  if (t1_used_host) *t1_used_host = t2;
  std::vector<float> ts(c->usave_slots);
  float t1 = t2;
  const float* u1 = nullptr;
  bool u_end_done = false;  // u_end already copied by the solve (early_slot)
  c->final_hook_fired = false; c->last_u_end_done = false;
  auto series_all = [&](int nsaved, float drop) {  // the caller's sol: every saved entry (drop: t1 of the corrected solution)
    for (int i = 0; i < nsaved; ++i)
      if (!(ts[i] == drop)) { c->series_idx.push_back(i); c->series_t.push_back(ts[i]); }
  };
  const float no_drop = nanf("");
  // sol.u[end] is the last save slot; when every saveat time lies in (t0, t2] that slot is known before the solve and
  // the solve itself copies it to u_end ahead of its final synchronisation (-1: not known, copy afterwards)
  auto early_slot = [&](const float* sv_, int nsv_) -> int {
    if (oo.save_start || nsv_ < 1) return -1;
    for (int i = 0; i < nsv_; ++i) if (!(sv_[i] > t0 && sv_[i] <= t2)) return -1;
    return nsv_ - 1;
  };
  // _get_ode_integrator (neural_ode.jl:33-38): fresh init on (t1, t2); then _perform_step (:77).  `L` is the context the
  // step runs on: the handle itself, or its companion (own stream) while the main solve is still going.
  StepArgs a{};
  auto enqueue_local = [&](lrnde_ctx* L, const float* u_at_t1, float t1v) -> int {
    int r;
    fill_args(L, a, B);
    a.t0 = t1v; a.t1 = t2; a.abstol = oo.abstol; a.reltol = oo.reltol; a.mode = MODE_SINGLE_INIT_DT;
    a.want_stiff = (reg_type == LRNDE_REG_STIFFNESS_ESTIMATE); a.maxiters = 1;
    a.force_store_k = c->dense_on ? 1 : 0;  // recorded forward: the regulariser's reverse sweep starts from this step's k2..k6
    if (use_qtile(L, B)) {
      a.init_u0 = u_at_t1; a.init_fresh = 1; a.init_nsaved = 0; memset(&a.init_si, 0, sizeof(a.init_si));
    } else {
      HIPCHK(c, hipMemcpyAsync(a.ubuf[0], u_at_t1, sizeof(float) * n, hipMemcpyDeviceToDevice, L->stream));
      hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, L->stream, L->ctrl, t1v, 0.f, 0, 0);
    }
    if ((r = run_init(L, B, a))) return r;
    a.init_u0 = nullptr; a.init_fresh = 0;
    if ((r = launch_step(L, B, a, 0))) return r;
    const size_t cnt = (size_t)a.nwg_global * PSTRIDE;
    if ((r = exchange(L, L->part + cnt, L->part_rx + cnt, cnt))) return r;
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, L->stream, a, 1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(L->ctrl_host, L->ctrl + 1, sizeof(Ctrl), hipMemcpyDeviceToHost, L->stream));
    return LRNDE_OK;
  };
  auto local_results = [&](const Ctrl& k) {
    c->loc_dt = k.dt; c->loc_eest = k.eest_last; c->loc_snum = k.stiff_num; c->loc_sden = k.stiff_den;
    if (reg_val_host) *reg_val_host = a.want_stiff ? k.reg_stiff : k.reg_error;
    if (nfe_host) *nfe_host = st->nf + k.nf;  // sol.destats.nf + (6 + 3), perform_step.jl:31
  };
  if (mode == LRNDE_MODE_NONE) {  // _vanilla_node_fallback, neural_ode.jl:56-60
    const float sv1[1] = {t2};
    oo.save_everystep = 0;
    const int early = early_slot(nuser ? user_sv : sv1, nuser ? nuser : 1);
    if (early >= 0) { c->tail_copy_dst = u_end; c->tail_copy_slot = early; }
    rc = lrnde_solve(c, x, B, t0, t2, &oo, nuser ? user_sv : sv1, nuser ? nuser : 1, c->usave, ts.data(), (int)c->usave_slots, st, nullptr, 0);
    c->tail_copy_dst = nullptr;
    if (rc) return rc;
    if (st->nsaved < 1) return fail(c, LRNDE_BADARG, "the solve saved nothing (saveat outside the time span)");
    series_all(st->nsaved, no_drop);
    c->last_ts.assign(ts.begin(), ts.begin() + st->nsaved);
    if (nfe_host) *nfe_host = st->nf;
    c->last_u_end_done = (st->nsaved - 1 == early);
    if (st->nsaved - 1 != early) {
      HIPCHK(c, hipMemcpyAsync(u_end, c->usave + (size_t)(st->nsaved - 1) * n, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return LRNDE_OK;
  } else if (mode == LRNDE_MODE_UNBIASED) {  // neural_ode.jl:68-84, saveat = [t1, t2] or vcat(user saveat, t1)
    t1 = t1_or_rand;
    std::vector<float> sv;
    if (nuser) { sv.assign(user_sv, user_sv + nuser); sv.insert(std::upper_bound(sv.begin(), sv.end(), t1), t1); }
    else { sv.push_back(t1); sv.push_back(t2); }
    oo.save_everystep = 0;
    // Overlap (unsharded handles): t1 is known before the solve, so as soon as a status poll shows sol(t1) in its save slot
    // the local step is enqueued on the companion's stream — and, recording, the regulariser's reverse sweep once the
    // step's scalars have come back — while this stream goes on with [t1, t2].  LRNDE_NO_OVERLAP=1: everything in order
    // on the handle's stream (the results are the same bits either way: same kernels, same inputs).
    const bool no_overlap = opt(OPT_NO_OVERLAP) != 0;
    lrnde_ctx* sd = nullptr;
    int side_state = 0;  // 0: nothing enqueued, 1: local step enqueued, 2: + sweep
    // save slot of the LAST saveat entry equal to t1: its index among the entries inside the span (those at or before t0 are
    // the start value and take no slot), behind the save_start slot if there is one
    const int kpos = (int)(std::upper_bound(sv.begin(), sv.end(), t1) - sv.begin()) - 1;
    int nskip = 0;
    while (nskip < (int)sv.size() && sv[nskip] <= t0) ++nskip;
    const int pos = (kpos >= nskip) ? kpos - nskip + (oo.save_start ? 1 : 0) : -1;
    auto enqueue_sweep = [&]() -> int {
      const Ctrl k = sd->ctrl_host[0];
      const int r = step_reg_sweep(sd, sd->state, B, t1, k.dt, oo.abstol, oo.reltol, reg_type, k.eest_last, k.stiff_num, k.stiff_den,
                                   c->rec_gr);
      if (r) { c->err = sd->err; return r; }
      HIPCHK(c, hipEventRecord(c->ev_side_sweep, sd->stream));
      c->rec_gr_ready = true;
      side_state = 2;
      return LRNDE_OK;
    };
    auto side_advance = [&](int nsaved_now, hipEvent_t ev) -> int {
      if (side_state == 0 && nsaved_now > pos) {
        if (ev) HIPCHK(c, hipStreamWaitEvent(sd->stream, ev, 0));
        c->side_busy = true;
        const int r = enqueue_local(sd, c->usave + (size_t)pos * n, t1);
        if (r) { if (!sd->err.empty()) c->err = sd->err; return r; }
        HIPCHK(c, hipEventRecord(c->ev_side_local, sd->stream));
        side_state = 1;
      } else if (side_state == 1 && c->dense_on && hipEventQuery(c->ev_side_local) == hipSuccess) {
        return enqueue_sweep();
      }
      return LRNDE_OK;
    };
    if (!sharded(c) && !no_overlap && !c->overlap_off && pos >= 0 && sv[kpos] == t1) {
      if ((rc = side_get(c, B, &sd))) return rc;
      c->poll_hook = [&](int nsaved_done, hipEvent_t ev) { return side_advance(nsaved_done, ev); };
    }
    const int early = early_slot(sv.data(), (int)sv.size());
    if (early >= 0) { c->tail_copy_dst = u_end; c->tail_copy_slot = early; }
    rc = lrnde_solve(c, x, B, t0, t2, &oo, sv.data(), (int)sv.size(), c->usave, ts.data(), (int)c->usave_slots, st, nullptr, 0);
    c->poll_hook = nullptr; c->tail_copy_dst = nullptr;
    if (rc) return rc;
    u_end_done = (st->nsaved - 1 == early);
    c->last_u_end_done = u_end_done;
    int i1 = -1;
    for (int i = 0; i < st->nsaved; ++i) if (ts[i] == t1) i1 = i;   // the last entry saved at t1 is sol(t1)
    if (i1 < 0) return fail(c, LRNDE_BADARG, "t1 = %g is not inside the time span", (double)t1);
    u1 = c->usave + (size_t)i1 * n;
    c->last_i1 = i1;
    series_all(st->nsaved, nuser ? t1 : no_drop);
    if (sd && i1 == pos) {
      c->last_ts.assign(ts.begin(), ts.begin() + st->nsaved);
      c->last_t1 = t1;
      if (!u_end_done)
        HIPCHK(c, hipMemcpyAsync(u_end, c->usave + (size_t)(st->nsaved - 1) * n, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
      if (t1_used_host) *t1_used_host = t1;
      if ((rc = side_advance(st->nsaved, nullptr))) return rc;   // (a t1 reported last: the handle's stream is idle by now)
      HIPCHK(c, hipEventSynchronize(c->ev_side_local));
      local_results(sd->ctrl_host[0]);
      if (c->dense_on && side_state == 1) {  // too late to run beside the solve: the backward pass enqueues it (sweep_pending)
        c->sweep_pending = true; c->sw_B = B; c->sw_reg_type = reg_type; c->sw_t1 = t1; c->sw_abstol = oo.abstol; c->sw_reltol = oo.reltol;
      }
      if (!c->dense_on || side_state == 1) c->side_busy = false;  // nothing of the companion's is left in flight
      if (!u_end_done) HIPCHK(c, hipStreamSynchronize(c->stream));
      return LRNDE_OK;
    }
    // (not reached with a companion step in flight unless the slot prediction was wrong: then nothing of it may be used)
    if ((rc = side_quiesce(c))) return rc;
    c->rec_gr_ready = false; c->sweep_pending = false;
  } else {  // neural_ode.jl:88-100, saveat = [] => every accepted step (or the user's saveat)
    oo.save_everystep = nuser ? 0 : 1;
    rc = lrnde_solve(c, x, B, t0, t2, &oo, nuser ? user_sv : nullptr, nuser, c->usave, ts.data(), (int)c->usave_slots, st, nullptr, 0);
    if (rc) return rc;
    if (st->nsaved < 2) return fail(c, LRNDE_BADARG, "biased mode needs at least two saved times");
    const int mm = st->nsaved - 1;
    int idx = (int)(t1_or_rand * (float)mm);
    if (idx >= mm) idx = mm - 1;
    if (idx < 0) idx = 0;
    t1 = ts[idx];
    u1 = c->usave + (size_t)idx * n;
    c->last_i1 = idx;
    series_all(st->nsaved, no_drop);
  }
  c->last_ts.assign(ts.begin(), ts.begin() + st->nsaved);
  c->last_t1 = t1;
  if (!u_end_done)
    HIPCHK(c, hipMemcpyAsync(u_end, c->usave + (size_t)(st->nsaved - 1) * n, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  if (t1_used_host) *t1_used_host = t1;
  // the local step in order on the handle's stream (biased t1, sharded handles, LRNDE_NO_OVERLAP)
  if ((rc = enqueue_local(c, u1, t1))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  local_results(c->ctrl_host[0]);
  return LRNDE_OK;
}
extern "C" {

int lrnde_node_forward(lrnde_ctx* c, const float* x, int32_t B, float t0, float t2,
                       const lrnde_solve_opts* o, int32_t mode, int32_t reg_type, float t1_or_rand,
                       float* u_end, float* reg_val_host, int32_t* nfe_host, lrnde_stats* st,
                       float* t1_used_host) {
  if (!c) return LRNDE_BADARG;
  HPT(c, 0);
  const int rc = node_forward_impl(c, x, B, t0, t2, o, mode, reg_type, t1_or_rand, nullptr, 0, u_end, reg_val_host, nfe_host, st, t1_used_host);
  HPT(c, 4);
  if (rc == LRNDE_OK) {
    for (int i = 1; i <= 4; ++i) c->hp_sum[i] += std::chrono::duration<double, std::micro>(c->hp_t[i] - c->hp_t[i - 1]).count();
    ++c->hp_n;
  }
  return rc;
}

int lrnde_comm_unique_id(void* out) {
  if (!out) return LRNDE_BADARG;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return LRNDE_NCCL_ERROR;
  memcpy(out, &id, sizeof(id));
  return LRNDE_OK;
}

int lrnde_comm_init(lrnde_ctx* c, const void* uid, int32_t rank, int32_t nranks) {
  if (!c || !uid || nranks < 1 || rank < 0 || rank >= nranks) return LRNDE_BADARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
  c->lcomm = nullptr;
  c->rank = rank; c->nranks = nranks;
  c->wsB = 0;  // partial vectors are sized by nranks
  if (nranks > 1 || opt(OPT_FORCE_COMM)) {
    ncclUniqueId id;
    memcpy(&id, uid, sizeof(id));
    NCCLCHK(c, ncclCommInitRank(&c->comm, nranks, id, rank));
    c->prered = !opt(OPT_GATHER_TILES);
  }
  return LRNDE_OK;
}

int lrnde_comm_destroy(lrnde_ctx* c) {
  if (!c) return LRNDE_BADARG;
  if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
  c->lcomm = nullptr;
  c->rank = 0; c->nranks = 1; c->wsB = 0;
  return LRNDE_OK;
}

// how many ranks the handle's communicator really has: ncclCommCount for RCCL, the rendezvous size for the local
// communicator, 1 for an unsharded handle — what bench.py prints as `rccl_nranks` (WORLD_SIZE says what was asked for,
// this says what was built)
int lrnde_comm_count(lrnde_ctx* c, int32_t* nranks_host, int32_t* kind_host) {
  if (!c || !nranks_host) return LRNDE_BADARG;
  int n = 1, kind = 0;
  if (c->comm) { NCCLCHK(c, ncclCommCount(c->comm, &n)); kind = 1; }
  else if (c->lcomm) { n = c->lcomm->n; kind = 2; }
  *nranks_host = n;
  if (kind_host) *kind_host = kind;  // 0 none, 1 RCCL, 2 in-process local communicator
  return LRNDE_OK;
}

// bench.py --gpus N: `reps` back-to-back exchanges of the per-step kind (the all-reduce of the error norm's partial sums,
// same buffers and count as inside a solve of batch B) between two HIP events on the handle's stream -> microseconds per
// exchange.  Collective: every rank must call it.  An unsharded handle reports 0.
int lrnde_bench_exchange(lrnde_ctx* c, int32_t B, int32_t reps, float* avg_us_host) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (reps < 1 || !avg_us_host) return fail(c, LRNDE_BADARG, "bad bench arguments");
  *avg_us_host = 0.f;
  if (!sharded(c)) return LRNDE_OK;
  if ((rc = ensure_workspace(c, B))) return rc;
  StepArgs a{};
  fill_args(c, a, B);
  const size_t cnt = (size_t)a.nwg_global * PSTRIDE;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  for (int i = 0; i < 3 && !rc; ++i) rc = exchange(c, c->part, c->part_rx, cnt);  // warm-up (first collective builds its channels)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipEventRecord(e0, c->stream));
  for (int i = 0; i < reps && !rc; ++i) rc = exchange(c, c->part + (size_t)(i & 1) * cnt, c->part_rx + (size_t)(i & 1) * cnt, cnt);
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (rc) return rc;
  *avg_us_host = ms * 1e3f / (float)reps;
  return LRNDE_OK;
}

// ---- in-process local communicator (include/lrnde_hooks.h; lrnde_comm.hpp) ----
int lrnde_local_comm_create(lrnde_local_comm** out, int32_t nranks) {
  if (!out || nranks < 1 || nranks > LRNDE_LC_MAXR) return LRNDE_BADARG;
  lrnde_local_comm* lc = new lrnde_local_comm();
  lc->n = nranks;
  *out = lc;
  return LRNDE_OK;
}

int lrnde_local_comm_destroy(lrnde_local_comm* lc) {
  if (!lc) return LRNDE_OK;
  for (int r = 0; r < lc->n; ++r) {
    if (!lc->joined[r]) continue;
    hipSetDevice(lc->device[r]);
    if (lc->tmp[r]) hipFree(lc->tmp[r]);
    if (lc->ready[r]) hipEventDestroy(lc->ready[r]);
    if (lc->done[r]) hipEventDestroy(lc->done[r]);
  }
  delete lc;
  return LRNDE_OK;
}

int lrnde_comm_init_local(lrnde_ctx* c, lrnde_local_comm* lc, int32_t rank) {
  if (!c || !lc || rank < 0 || rank >= lc->n) return LRNDE_BADARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->comm) { ncclCommDestroy(c->comm); c->comm = nullptr; }
  {
    std::lock_guard<std::mutex> lk(lc->mu);
    if (lc->joined[rank]) return fail(c, LRNDE_BADARG, "rank %d has already joined this local communicator", rank);
    // one device only: the sum kernel reads the peers' send buffers directly, and no peer access is set up between devices
    for (int r = 0; r < lc->n; ++r)
      if (lc->joined[r] && lc->device[r] != c->device)
        return fail(c, LRNDE_UNSUPPORTED, "the in-process local communicator is single-device (rank %d is on device %d, rank %d on %d); "
                    "use lrnde_comm_init (RCCL) across devices", r, lc->device[r], rank, c->device);
    HIPCHK(c, hipEventCreateWithFlags(&lc->ready[rank], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&lc->done[rank], hipEventDisableTiming));
    lc->device[rank] = c->device;
    lc->joined[rank] = true;
  }
  c->lcomm = lc; c->rank = rank; c->nranks = lc->n;
  c->prered = !opt(OPT_GATHER_TILES);
  c->wsB = 0;  // partial vectors are sized by nranks
  return LRNDE_OK;
}

int lrnde_set_overlap(lrnde_ctx* c, int32_t on) {
  if (!c) return LRNDE_BADARG;
  const int rc = side_quiesce(c);
  if (rc) return rc;
  c->overlap_off = !on;
  return LRNDE_OK;
}

int lrnde_bench_step(lrnde_ctx* c, const float* uprev, const float* k1, int32_t B, float t, float dt,
                     float abstol, float reltol, int32_t reps, float* avg_us_host) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!uprev || !k1 || reps < 1 || !avg_us_host) return fail(c, LRNDE_BADARG, "bad bench arguments");
  if ((rc = ensure_workspace(c, B))) return rc;
  StepArgs a{};
  fill_args(c, a, B);
  a.t0 = t; a.t1 = t + 1.0f; a.bench_dt = dt; a.abstol = abstol; a.reltol = reltol; a.mode = MODE_BENCH;
  const size_t n = (size_t)B * c->desc.state_dim;
  HIPCHK(c, hipMemcpyAsync(a.ubuf[0], uprev, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(a.kfsal[0], k1, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  for (int i = 0; i < 3; ++i) if ((rc = launch_step(c, B, a, i))) return rc;  // warm
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < reps; ++i) if ((rc = launch_step(c, B, a, i))) return rc;
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *avg_us_host = ms * 1000.0f / (float)reps;
  return LRNDE_OK;
}

// ---- SDE (src/layers/neural_sde.jl, src/perform_step.jl:172-206) ----
namespace { struct SdeNodeRecord; }   // the layer's forward record (lrnde_sde_node.hpp)
struct lrnde_sde {
  SdeNodeRecord* node = nullptr;
  double *idt_part = nullptr, *idt_part_host = nullptr;   // sde_init_dt's norm partials (device / pinned)
  double* march_part = nullptr; size_t march_part_n = 0;   // marched fixed-grid solve: [step][workgroup] partial sums
  double* idt_pp = nullptr; int idt_pp_nwg = 0;           // sde_init_dt_dev's per-workgroup partial sums (two phases)
  float *idt_scal = nullptr, *idt_scal_host = nullptr;    // sde_init_dt_dev's results: {dt0, d1, dt} of the solve, then of the local step
  lrnde_ctx* drift = nullptr;
  lrnde_ctx* diff = nullptr;
  float* p2 = nullptr;  // expanded diffusion parameters
  int diff_bias = 1;
  Ctrl *traj_host = nullptr, *traj_dev = nullptr; int traj_cap = 0;  // per-step records of lrnde_sde_solve_fixed (pinned / device)
  float* sri_ws = nullptr; size_t sri_n = 0; double *sri_part = nullptr, *sri_part_host = nullptr;  // lrnde_sde_sri_step scratch
  float* bwd_ws = nullptr; size_t bwd_n = 0;  // lrnde_sde_*_backward / _reg_grad scratch
  // the one-launch reverse sweep (lrnde_sde_bwd_fused.hpp): raw drift parameters as the caller gave them, the workgroups'
  // parameter-cotangent partials, the recorded steps and the series table on the device
  float* pdr = nullptr; float* bwf_part = nullptr; size_t bwf_part_n = 0; int* bwf_meta = nullptr; int* bwf_meta_pin = nullptr; size_t bwf_meta_n = 0;
  float* bwf_hist = nullptr; size_t bwf_hist_n = 0;   // the deferred sweep's history records
  int* arrive = nullptr;                      // arrival counter of the one-launch step's footer (lrnde_sde_fast.hpp)
  float* ad_ws = nullptr; size_t ad_n = 0;    // lrnde_sde_solve_adaptive: two states + the current increment
  SdeCtl* ad_ctl = nullptr; SdeCtl* ad_ctl_host = nullptr;             // device-controlled adaptive loop: control block (device / pinned), heading ...
  int ad_blob_cap = 0;                                                 // ... room for this many (start, length) pairs 64 bytes in
  unsigned long long* ad_prog = nullptr; unsigned long long* ad_prog_dev = nullptr;  // its pinned progress word
  lrnde_trace_row* ad_trace = nullptr; int ad_trace_cap = 0;
};

int lrnde_sde_create(lrnde_sde** out, const lrnde_model_desc* drift, int32_t diffusion_bias, int device, void* stream) {
  if (!out || !drift) return LRNDE_BADARG;
  *out = nullptr;
  lrnde_sde* s = new lrnde_sde();
  s->diff_bias = diffusion_bias ? 1 : 0;
  int rc = lrnde_create(&s->drift, drift, device, stream);
  if (rc) { delete s; return rc; }
  lrnde_model_desc dd{drift->state_dim, drift->state_dim, 0, LRNDE_ACT_IDENTITY};
  rc = lrnde_create(&s->diff, &dd, device, stream);
  if (rc) { lrnde_destroy(s->drift); delete s; return rc; }
  const size_t n2 = lrnde_param_count(&dd);
  if (hipMalloc(&s->p2, sizeof(float) * n2) != hipSuccess) { lrnde_destroy(s->drift); lrnde_destroy(s->diff); delete s; return LRNDE_HIP_ERROR; }
  *out = s;
  return LRNDE_OK;
}

namespace { void sde_node_release(lrnde_sde* s); unsigned long long sde_node_generation(const lrnde_sde* s); }
int lrnde_sde_destroy(lrnde_sde* s) {
  if (!s) return LRNDE_OK;
  sde_node_release(s);
  if (s->idt_part) hipFree(s->idt_part);
  if (s->idt_part_host) hipHostFree(s->idt_part_host);
  if (s->idt_pp) hipFree(s->idt_pp);
  if (s->march_part) hipFree(s->march_part);
  if (s->idt_scal) hipFree(s->idt_scal);
  if (s->idt_scal_host) hipHostFree(s->idt_scal_host);
  lrnde_destroy(s->drift);
  lrnde_destroy(s->diff);
  if (s->p2) hipFree(s->p2);
  if (s->traj_host) hipHostFree(s->traj_host);
  if (s->traj_dev) hipFree(s->traj_dev);
  if (s->sri_ws) hipFree(s->sri_ws);
  if (s->bwd_ws) hipFree(s->bwd_ws);
  if (s->pdr) hipFree(s->pdr);
  if (s->bwf_part) hipFree(s->bwf_part);
  if (s->bwf_meta) hipFree(s->bwf_meta);
  if (s->bwf_meta_pin) hipHostFree(s->bwf_meta_pin);
  if (s->bwf_hist) hipFree(s->bwf_hist);
  if (s->ad_ctl) hipFree(s->ad_ctl);
  if (s->ad_ctl_host) hipHostFree(s->ad_ctl_host);
  if (s->ad_prog) hipHostFree(s->ad_prog);
  if (s->ad_trace) hipFree(s->ad_trace);
  if (s->arrive) hipFree(s->arrive);
  if (s->ad_ws) hipFree(s->ad_ws);
  if (s->sri_part) hipFree(s->sri_part);
  if (s->sri_part_host) hipHostFree(s->sri_part_host);
  delete s;
  return LRNDE_OK;
}

const char* lrnde_sde_last_error(const lrnde_sde* s) { return s ? lrnde_last_error(s->drift) : "null sde handle"; }

int lrnde_sde_set_params(lrnde_sde* s, const float* p_drift, size_t n_drift, const float* p_diff, size_t n_diff) {
  if (!s || !p_drift || !p_diff) return LRNDE_BADARG;
  const int D = s->drift->desc.state_dim;
  if (n_diff != (size_t)D * D + (s->diff_bias ? D : 0))
    return fail(s->drift, LRNDE_BADARG, "diffusion parameter count %zu != %zu", n_diff, (size_t)D * D + (s->diff_bias ? D : 0));
  int rc = lrnde_set_params(s->drift, p_drift, n_drift);
  if (rc) return rc;
  if (!s->pdr) HIPCHK(s->drift, hipMalloc(&s->pdr, sizeof(float) * n_drift));
  HIPCHK(s->drift, hipMemcpyAsync(s->pdr, p_drift, sizeof(float) * n_drift, hipMemcpyDeviceToDevice, s->drift->stream));
  hipLaunchKernelGGL(k_diff_expand, dim3(64), dim3(256), 0, s->diff->stream, p_diff, D, s->diff_bias, s->p2);
  return lrnde_set_params(s->diff, s->p2, lrnde_param_count(&s->diff->desc));
}

static int sde_step_impl(lrnde_sde* s, int which, const float* uprev, const float* dW, int32_t B, float t, float dt,
                         float abstol, float reltol, float delta, float* u, float* eest_host, float* reg_val_host);
// the one-launch Euler-Heun kernel (lrnde_sde_fast.hpp) serves this handle: unsharded, no time input, D <= 64, H <= 128
static bool sde_uses_fast(const lrnde_sde* s) {
  const lrnde_ctx* c = s->drift;
  return !opt(OPT_NO_SDE_FAST) && !sharded(c) && !c->desc.time_dep && sde_fast_shape(c->desc.state_dim, c->desc.hidden_dim);
}
static void sde_fast_args(lrnde_sde* s, SdeFastArgs& f) {
  lrnde_ctx* c = s->drift;
  memset(&f, 0, sizeof(f));
  f.W1p = c->m.W1p; f.KG1 = c->m.KG1;
  f.W2p = c->m.W2p; f.KG2p = ((c->m.KG2 + SEGK - 1) / SEGK) * SEGK;
  f.Wgp = s->diff->m.W2p; f.KGgp = ((s->diff->m.KG2 + SEGK - 1) / SEGK) * SEGK;
  f.b1 = c->m.b1; f.b2 = c->m.b2; f.bg = s->diff->m.b2; f.act = c->m.act; f.D = c->desc.state_dim;
}
int lrnde_sde_euler_heun_step(lrnde_sde* s, const float* uprev, const float* dW, int32_t B, float t, float dt,
                              float abstol, float reltol, float delta, float* u, float* eest_host,
                              float* reg_val_host) {
  return sde_step_impl(s, 0, uprev, dW, B, t, dt, abstol, reltol, delta, u, eest_host, reg_val_host);
}
int lrnde_sde_rkmil_step(lrnde_sde* s, const float* uprev, const float* dW, int32_t B, float t, float dt,
                         float abstol, float reltol, float* u, float* eest_host, float* reg_val_host) {
  return sde_step_impl(s, 1, uprev, dW, B, t, dt, abstol, reltol, 0.f, u, eest_host, reg_val_host);
}
// one step on the stream, its Ctrl record copied to `rec` (pinned host) without waiting
// one step on the stream; its record goes to `rec` (pinned host: integrator-state footer k_finalize + async copy) or to
// the device slot `rec_dev` (k_sde_record: two launches per step, no copy)
static int sde_step_enqueue(lrnde_sde* s, int which, const float* uprev, const float* dW, int32_t B, float t, float dt,
                            float abstol, float reltol, float delta, float* u, Ctrl* rec, Ctrl* rec_dev = nullptr,
                            const float* dt_dev = nullptr, float* dW_scaled = nullptr) {
  lrnde_ctx* c = s->drift;
  int rc;
  StepArgs a{};
  fill_args(c, a, B, NB);
  a.m2 = s->diff->m;
  a.t0 = t; a.bench_dt = dt; a.abstol = abstol; a.reltol = reltol; a.delta = delta;
  a.dW = dW; a.sde_scratch = c->state; a.sde_uprev = uprev; a.sde_u = u;
  const int nwg = (B + NB - 1) / NB;
  // the MNIST-SDE shape (state 32, hidden 64, no time input) has a one-launch small-latency kernel (lrnde_sde_fast.hpp)
  if (which == 0 && sde_uses_fast(s)) {
    SdeFastArgs f{};
    sde_fast_args(s, f);
    f.u = uprev; f.dW = dW; f.un = u; f.B = B; f.dt = dt; f.abstol = abstol; f.reltol = reltol; f.delta = delta;
    f.part = c->part + (size_t)a.nwg_global * PSTRIDE;  // the parity-1 block k_finalize reads
    f.n_norm = a.n_global;
    f.dt_dev = dt_dev; f.dW_scaled = dW_scaled;   // (the layer's local step: dt from the device, dW = sqrt(dt) z formed in the launch)
    if (rec_dev) {  // fixed-grid solve / the layer's local step: the step writes its own record (no footer launch)
      if (!s->arrive) { HIPCHK(c, hipMalloc(&s->arrive, sizeof(int))); HIPCHK(c, hipMemsetAsync(s->arrive, 0, sizeof(int), c->stream)); }
      f.arrive = s->arrive; f.rec = rec_dev;
      sde_fast_launch(f.D, c->desc.hidden_dim, nwg, c->stream, f);
      HIPCHK(c, hipGetLastError());
      return LRNDE_OK;
    }
    if (!dt_dev)
      hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, c->stream, c->ctrl, t, dt, 0, 0);
    sde_fast_launch(f.D, c->desc.hidden_dim, nwg, c->stream, f);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, c->stream, a, 1);
    HIPCHK(c, hipGetLastError());
    if (rec) HIPCHK(c, hipMemcpyAsync(rec, c->ctrl + 1, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
    return LRNDE_OK;
  }
  size_t sm = smem_bytes(c->m.Dp, c->m.Hp);
  const size_t sm2 = smem_bytes(s->diff->m.Dp, s->diff->m.Hp);
  if (sm2 > sm) sm = sm2;
  if (!rec_dev) hipLaunchKernelGGL(k_ctrl_init, dim3(1), dim3(1), 0, c->stream, c->ctrl, t, dt, 0, 0);
  if (which == 1) {
    if (vecw(c) == 4) hipLaunchKernelGGL(k_sde_rkmil<4>, dim3(nwg), dim3(NT), sm, c->stream, a);
    else hipLaunchKernelGGL(k_sde_rkmil<1>, dim3(nwg), dim3(NT), sm, c->stream, a);
  } else if (vecw(c) == 4) hipLaunchKernelGGL(k_sde_step<4>, dim3(nwg), dim3(NT), sm, c->stream, a);
  else hipLaunchKernelGGL(k_sde_step<1>, dim3(nwg), dim3(NT), sm, c->stream, a);
  HIPCHK(c, hipGetLastError());
  const size_t cnt = (size_t)a.nwg_global * PSTRIDE;
  if ((rc = exchange(c, c->part + cnt, c->part_rx + cnt, cnt))) return rc;
  if (rec_dev) hipLaunchKernelGGL(k_sde_record, dim3(1), dim3(64), 0, c->stream, a, dt, rec_dev);  // two launches per step
  else hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, c->stream, a, 1);
  HIPCHK(c, hipGetLastError());
  if (rec) HIPCHK(c, hipMemcpyAsync(rec, c->ctrl + 1, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
  return LRNDE_OK;
}
static int sde_check(lrnde_sde* s, const float* uprev, const float* dW, const float* u, int32_t B, float dt) {
  if (!s) return LRNDE_BADARG;
  lrnde_ctx* c = s->drift;
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!s->diff->have_params) return fail(c, LRNDE_BADARG, "lrnde_sde_set_params has not been called");
  if (!uprev || !dW || !u) return fail(c, LRNDE_BADARG, "null pointer");
  if (!(dt > 0.f)) return fail(c, LRNDE_BADARG, "dt must be positive");
  return ensure_workspace(c, B);
}
static int sde_step_impl(lrnde_sde* s, int which, const float* uprev, const float* dW, int32_t B, float t, float dt,
                         float abstol, float reltol, float delta, float* u, float* eest_host, float* reg_val_host) {
  int rc = sde_check(s, uprev, dW, u, B, dt);
  if (rc) return rc;
  lrnde_ctx* c = s->drift;
  if ((rc = sde_step_enqueue(s, which, uprev, dW, B, t, dt, abstol, reltol, delta, u, c->ctrl_host))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (eest_host) *eest_host = c->ctrl_host[0].eest_last;
  if (reg_val_host) *reg_val_host = c->ctrl_host[0].reg_error;  // EEst * dt (src/perform_step.jl:205)
  return LRNDE_OK;
}
// nsteps steps of size dt from t0 on a fixed grid, step i from t0 + i*dt with the increments dW[i]: the loop a
// NeuralDSDE forward runs (src/layers/neural_sde.jl with a fixed-step solver), enqueued without a host round trip
// per step.  u_traj (device, nsteps x B x D) receives every step's u; eest_host / reg_val_host (host, nsteps, may
// be NULL) every step's EEst and EEst*dt.  which: 0 Euler-Heun (delta used), 1 Milstein.
// every step's record of a marched fixed-grid solve: EEst and EEst*dt from the workgroups' partial sums (one wave per step)
__global__ void k_sde_march_records(const double* part, int nwg, double n_norm, float dt, Ctrl* rec) {
  if (threadIdx.x >= 64) return;
  const Sum3 s = reduce_partials3(part + (size_t)blockIdx.x * nwg * PSTRIDE, nwg);
  if (threadIdx.x == 0) {
    const float eest = rms_from(s.a, n_norm);
    Ctrl* r = rec + blockIdx.x;
    r->eest_last = eest;
    r->reg_error = eest * dt;
    r->status = ST_DONE;
  }
}

int lrnde_sde_solve_fixed(lrnde_sde* s, int32_t which, const float* u0, const float* dW, int32_t B, float t0, float dt,
                          int32_t nsteps, float abstol, float reltol, float delta, float* u_traj, float* eest_host,
                          float* reg_val_host) {
  int rc = sde_check(s, u0, dW, u_traj, B, dt);
  if (rc) return rc;
  lrnde_ctx* c = s->drift;
  if (nsteps <= 0) return fail(c, LRNDE_BADARG, "nsteps must be positive");
  if (which != 0 && which != 1) return fail(c, LRNDE_BADARG, "which: 0 Euler-Heun, 1 Milstein");
  if (s->traj_cap < nsteps) {
    if (s->traj_host) hipHostFree(s->traj_host);
    if (s->traj_dev) hipFree(s->traj_dev);
    s->traj_host = nullptr; s->traj_dev = nullptr; s->traj_cap = 0;
    HIPCHK(c, hipHostMalloc(&s->traj_host, sizeof(Ctrl) * (size_t)nsteps));
    HIPCHK(c, hipMalloc(&s->traj_dev, sizeof(Ctrl) * (size_t)nsteps));
    s->traj_cap = nsteps;
  }
  const size_t n = (size_t)B * c->desc.state_dim;
  if (which == 0 && sde_uses_fast(s) && !opt(OPT_SDE_NO_MARCH)) {
    // the one-launch step's shape: the whole grid in ONE launch, no step waiting for another workgroup (lrnde_sde_fast.hpp,
    // march_n), then one launch for the nsteps records.  LRNDE_SDE_NO_MARCH=1: a launch per step (same bits)
    const int nwg = (B + NB - 1) / NB;
    const size_t need = (size_t)nsteps * nwg * PSTRIDE;
    if (s->march_part_n < need) {
      if (s->march_part) HIPCHK(c, hipFree(s->march_part));
      s->march_part = nullptr; s->march_part_n = 0;
      HIPCHK(c, hipMalloc(&s->march_part, sizeof(double) * need));
      s->march_part_n = need;
    }
    SdeFastArgs f{};
    sde_fast_args(s, f);
    f.u = u0; f.dW = dW; f.un = u_traj; f.B = B; f.dt = dt; f.abstol = abstol; f.reltol = reltol; f.delta = delta;
    f.n_norm = (double)n;
    f.march_n = nsteps; f.march_part = s->march_part;
    sde_fast_launch(f.D, c->desc.hidden_dim, nwg, c->stream, f);
    hipLaunchKernelGGL(k_sde_march_records, dim3(nsteps), dim3(64), 0, c->stream, (const double*)s->march_part, nwg, (double)n, dt, s->traj_dev);
    HIPCHK(c, hipGetLastError());
  } else
  for (int i = 0; i < nsteps; ++i) {
    const float t = t0 + (float)i * dt;
    if ((rc = sde_step_enqueue(s, which, i == 0 ? u0 : u_traj + (size_t)(i - 1) * n, dW + (size_t)i * n, B, t, dt, abstol, reltol,
                               delta, u_traj + (size_t)i * n, nullptr, s->traj_dev + i))) return rc;
  }
  HIPCHK(c, hipMemcpyAsync(s->traj_host, s->traj_dev, sizeof(Ctrl) * (size_t)nsteps, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < nsteps; ++i) {
    if (eest_host) eest_host[i] = s->traj_host[i].eest_last;
    if (reg_val_host) reg_val_host[i] = s->traj_host[i].reg_error;
  }
  return LRNDE_OK;
}

// ---- adaptive Euler-Heun solve on a caller-supplied Brownian path ----
__global__ void k_sde_dw(size_t n, const float* Wlo, const float* Whi, float* dW) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dW[i] = Whi[i] - Wlo[i];
}

__global__ void k_sde_ctl_init(SdeCtl* ctl, int m0, float dtc0) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  SdeCtl c;
  c.status = ST_RUNNING; c.i = 0; c.m = m0; c.cur = 0; c.naccept = 0; c.nreject = 0; c.iters = 1; c.nf = 0;
  c.qold = 1e-4f; c.eest_last = 0.f; c.dtc = dtc0;
  *ctl = c;
}

// ... the initial dt read from the device (sde_init_dt_dev's result), quantised to the path's grid as the host does: the rerun of
// a solve whose control block k_sde_initdt_fin had initialised
__global__ void k_sde_ctl_init_dtdev(SdeCtl* ctl, const float* dt_dev, float h, int nfine) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float dt0 = *dt_dev;
  int m0 = (int)(dt0 / h); if (m0 < 1) m0 = 1;
  if (m0 > nfine) m0 = nfine;
  SdeCtl c;
  c.status = ST_RUNNING; c.i = 0; c.m = m0; c.cur = 0; c.naccept = 0; c.nreject = 0; c.iters = 1; c.nf = 0;
  c.qold = 1e-4f; c.eest_last = 0.f; c.dtc = dt0;
  *ctl = c;
}
// the end state of a device-controlled solve (the control block says which of the two buffers holds it) -> out
__global__ void k_sde_pick_end(size_t n, const SdeCtl* ctl, const float* ua, const float* ub, float* out) {
  const float* src = ctl->cur ? ub : ua;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = src[i];
}

// lrnde_sde_solve_adaptive on the one-launch kernel's shape: the controller runs in the step kernel's footer (SdeCtl,
// lrnde_sde_fast.hpp), the host only keeps launches enqueued and watches a pinned progress word — one launch per attempted
// step, no synchronisation inside the solve (the host-controlled loop below paid one per step: ~45 us for a 10-us step).
// rec_u / rec_im / rec_cap: the layer's dense record of the accepted steps (device; NULL / 0 for a plain solve)
// the device-controlled loop's control block and progress word.  The block heads one allocation with room for the record's
// (start, length) pairs behind it (64 bytes in): control block and pairs come home in ONE copy.  rec_cap: pairs to hold
// (a call that grows the allocation must come before anything initialises the block: the layer calls it ahead of its initial dt).
static int sde_adaptive_prepare(lrnde_sde* s, int rec_cap = 0) {
  static_assert(sizeof(SdeCtl) <= 64, "the pairs start 64 bytes in");
  lrnde_ctx* c = s->drift;
  if (!s->ad_prog) {
    HIPCHK(c, hipHostMalloc(&s->ad_prog, 64, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer((void**)&s->ad_prog_dev, s->ad_prog, 0));
  }
  if (!s->ad_ctl || s->ad_blob_cap < rec_cap) {
    if (s->ad_ctl) HIPCHK(c, hipFree(s->ad_ctl));
    if (s->ad_ctl_host) HIPCHK(c, hipHostFree(s->ad_ctl_host));
    s->ad_ctl = nullptr; s->ad_ctl_host = nullptr; s->ad_blob_cap = 0;
    const size_t bytes = 64 + sizeof(int2) * (size_t)rec_cap;
    HIPCHK(c, hipMalloc((void**)&s->ad_ctl, bytes));
    HIPCHK(c, hipHostMalloc((void**)&s->ad_ctl_host, bytes));
    s->ad_blob_cap = rec_cap;
  }
  return LRNDE_OK;
}
static int sde_adaptive_device(lrnde_sde* s, const float* u0, const float* W, int32_t nfine, int32_t B, float t0, float t1,
                               const lrnde_sde_adapt_opts* o, float* u_end, lrnde_stats* st, lrnde_trace_row* trace_host,
                               int32_t cap_trace, float* ua, float* ub, float* rec_u = nullptr, int2* rec_im = nullptr,
                               int rec_cap = 0, int2* rec_im_host = nullptr, const float* dt0_dev = nullptr, bool no_persist = false) {
  lrnde_ctx* c = s->drift;
  const size_t n = (size_t)B * c->desc.state_dim;
  const bool pairs_home = rec_im_host && rec_im && rec_cap > 0;   // the layer's record: its pairs live behind the control block
  int rc0 = sde_adaptive_prepare(s, pairs_home ? rec_cap : 0);
  if (rc0) return rc0;
  if (pairs_home) rec_im = reinterpret_cast<int2*>(reinterpret_cast<char*>(s->ad_ctl) + 64);
  if (trace_host && cap_trace > s->ad_trace_cap) {
    if (s->ad_trace) HIPCHK(c, hipFree(s->ad_trace));
    s->ad_trace = nullptr; s->ad_trace_cap = 0;
    HIPCHK(c, hipMalloc(&s->ad_trace, sizeof(lrnde_trace_row) * cap_trace));
    s->ad_trace_cap = cap_trace;
  }
  if (!s->arrive) { HIPCHK(c, hipMalloc(&s->arrive, sizeof(int))); HIPCHK(c, hipMemsetAsync(s->arrive, 0, sizeof(int), c->stream)); }
  const float h = (t1 - t0) / (float)nfine;
  int m0 = (int)(o->dt0 / h); if (m0 < 1) m0 = 1;
  if (m0 > nfine) m0 = nfine;
  if (1 > o->maxiters) { st->iters = 1; st->retcode = LRNDE_MAXITERS; return fail(c, LRNDE_MAXITERS, "adaptive SDE solve stopped with retcode %d at t=%g", LRNDE_MAXITERS, (double)t0); }
  HIPCHK(c, hipMemcpyAsync(ua, u0, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  if (dt0_dev && no_persist) hipLaunchKernelGGL(k_sde_ctl_init_dtdev, dim3(1), dim3(1), 0, c->stream, s->ad_ctl, dt0_dev, h, nfine);  // (the rerun)
  else if (!dt0_dev)   // (dt0_dev: sde_init_dt_dev's closing launch has initialised the control block from its dt)
    hipLaunchKernelGGL(k_sde_ctl_init, dim3(1), dim3(1), 0, c->stream, s->ad_ctl, m0, o->dt0);
  volatile unsigned long long* pw = s->ad_prog;
  *pw = 0ull;
  StepArgs a{};
  fill_args(c, a, B, NB);
  SdeFastArgs f{};
  sde_fast_args(s, f);
  f.rec_u = rec_u; f.rec_im = rec_im; f.rec_cap = rec_cap;
  f.B = B; f.abstol = o->abstol; f.reltol = o->reltol; f.delta = o->delta;
  f.part = c->part + (size_t)a.nwg_global * PSTRIDE;
  f.n_norm = a.n_global;
  f.arrive = s->arrive;
  f.ctl = s->ad_ctl; f.Wpath = W; f.ua = ua; f.ub = ub; f.nfine = nfine; f.t0 = t0; f.h = h;
  f.gamma = o->gamma; f.qmin = o->qmin; f.qmax = o->qmax; f.beta1 = o->beta1; f.beta2 = o->beta2; f.maxiters = o->maxiters;
  f.trace = trace_host ? s->ad_trace : nullptr; f.cap_trace = trace_host ? cap_trace : 0;
  f.prog = s->ad_prog_dev;
  const int nwg = (B + NB - 1) / NB;
  // The whole solve as ONE cooperative launch (k_sde_eh_fast<DT, HT, true>: state and weights stay in registers, a grid barrier
  // per step) when every workgroup fits on the chip at once; LRNDE_SDE_NO_PERSIST=1, a launch the runtime refuses or more than
  // 256 workgroups: the launch-per-step loop below.  Same arithmetic, same controller: same bits.
  bool persisted = false, plain = false;
  // (under rocprofv3 a process that made a cooperative launch dies in the tool's exit handler after the trace is written —
  //  observed with ROCm 7.2; the profiler preloads its tool library, and profiled runs take the launch-per-step loop)
  static const bool profiled = [] { const char* p = getenv("LD_PRELOAD"); return p && strstr(p, "rocprofiler") != nullptr; }();
  if (!opt(OPT_SDE_NO_PERSIST) && !profiled && !no_persist && nwg <= 256) {
    // up to half the chip's CUs: a plain launch (every workgroup finds a CU on an idle device; should the device be busy for longer
    // than the barrier's 50 ms bound, the solve comes back with an error status and is rerun as the loop below);
    // LRNDE_SDE_COOP_LAUNCH=1 or a larger grid: the cooperative API (25-30 us more per solve)
    plain = nwg <= 128 && !opt(OPT_SDE_COOP_LAUNCH);
    f.dbg_stall = (plain && opt(OPT_SDE_PERSIST_STALL)) ? 1 : 0;
    f.part2 = c->part;
    HIPCHK(c, hipMemsetAsync(c->part, 0, sizeof(double) * 2 * (size_t)nwg * PSTRIDE, c->stream));   // (tags of an earlier solve)
    f.jlaunch = 0;
    const hipError_t le = sde_persist_launch(f.D, c->desc.hidden_dim, nwg, c->stream, f, !plain);
    if (le == hipSuccess) persisted = true;
    else (void)hipGetLastError();   // (not resident / not supported: fall through to the loop)
  }
  // launches are enqueued eight at a time, the next eight when four of them have reported; launches that find the solve
  // finished return at once (at most eight of them)
  int j = 0;
  bool done = persisted;
  const long cap = (long)o->maxiters + 16;
  while (!done && j <= cap) {
    for (int k = 0; k < 8; ++k, ++j) {
      f.jlaunch = j;
      sde_fast_launch(f.D, c->desc.hidden_dim, nwg, c->stream, f);
    }
    HIPCHK(c, hipGetLastError());
    const unsigned want = (unsigned)(j - 4);
    const SpinDeadline deadline;
    for (long spin = 1;; ++spin) {
      const unsigned long long w = *pw;
      if ((unsigned)(w >> 32) != (unsigned)ST_RUNNING) { done = true; break; }
      if ((unsigned)(w & 0xffffffffull) >= want) break;
      if ((spin & 0x3fff) == 0 && spin_stalled(spin)) {  // (a stream query is a marker packet in the queue: only when the report is late)
        const hipError_t qe = hipStreamQuery(c->stream);
        if (qe != hipSuccess && qe != hipErrorNotReady) return fail(c, LRNDE_HIP_ERROR, "adaptive SDE loop: %s", hipGetErrorString(qe));
        if (qe == hipErrorNotReady && deadline.expired()) return LRNDE_HUNG(c, "adaptive SDE loop");
        if (qe == hipSuccess) break;  // everything enqueued has run: look at the word again, enqueue more
      }
    }
  }
  // one synchronisation ends the solve: the control block, the end state (picked on the device) and the record's
  // (start, length) pairs are all enqueued before it
  HIPCHK(c, hipMemcpyAsync(s->ad_ctl_host, s->ad_ctl, 64 + (pairs_home ? sizeof(int2) * (size_t)rec_cap : 0), hipMemcpyDeviceToHost, c->stream));
  if (u_end) {   // (the layer does not ask for it: the end state is its record's last slot)
    int nb = (int)((n + 255) / 256); if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_sde_pick_end, dim3(nb), dim3(256), 0, c->stream, n, (const SdeCtl*)s->ad_ctl, (const float*)ua, (const float*)ub, u_end);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const SdeCtl fin = *s->ad_ctl_host;
  if (persisted && plain && (fin.status == LRNDE_HIP_ERROR || fin.status == ST_RUNNING))   // the barrier gave up waiting (busy device): the loop
    return sde_adaptive_device(s, u0, W, nfine, B, t0, t1, o, u_end, st, trace_host, cap_trace, ua, ub, rec_u, rec_im, rec_cap, rec_im_host, dt0_dev, true);
  if (pairs_home && fin.naccept > 0)
    memcpy(rec_im_host, reinterpret_cast<const char*>(s->ad_ctl_host) + 64, sizeof(int2) * (size_t)(fin.naccept < rec_cap ? fin.naccept : rec_cap));
  st->naccept = fin.naccept; st->nreject = fin.nreject; st->iters = fin.iters; st->nf = fin.nf; st->eest_last = fin.eest_last;
  st->t_final = t0 + (float)fin.i * h; st->dt_final = (float)fin.m * h;
  st->retcode = (fin.status == ST_DONE) ? LRNDE_OK : (fin.status == ST_RUNNING ? LRNDE_MAXITERS : fin.status);
  if (trace_host) {
    int nt = fin.naccept + fin.nreject;
    if (nt > cap_trace) nt = cap_trace;
    if (nt > 0) HIPCHK(c, hipMemcpy(trace_host, s->ad_trace, sizeof(lrnde_trace_row) * nt, hipMemcpyDeviceToHost));
  }
  if (st->retcode != LRNDE_OK) return fail(c, st->retcode, "adaptive SDE solve stopped with retcode %d at t=%g", st->retcode, (double)st->t_final);
  return LRNDE_OK;
}

// rec_u (device, rec_cap x B x D) / rec_im_dev (device) / rec_im_host (host): the dense record of the accepted steps —
// end state, (start index, length) on the path's grid — for the layer's recorded forward; all NULL for a plain solve
static int sde_solve_adaptive_impl(lrnde_sde* s, const float* u0, const float* W, int32_t nfine, int32_t B, float t0, float t1,
                                   const lrnde_sde_adapt_opts* o, float* u_end, lrnde_stats* st, lrnde_trace_row* trace_host,
                                   int32_t cap_trace, float* rec_u, int2* rec_im_dev, int2* rec_im_host, int rec_cap,
                                   const float* dt0_dev = nullptr);
int lrnde_sde_solve_adaptive(lrnde_sde* s, const float* u0, const float* W, int32_t nfine, int32_t B, float t0, float t1,
                             const lrnde_sde_adapt_opts* o, float* u_end, lrnde_stats* st, lrnde_trace_row* trace_host,
                             int32_t cap_trace) {
  if (s && !u_end) return fail(s->drift, LRNDE_BADARG, "null pointer");
  return sde_solve_adaptive_impl(s, u0, W, nfine, B, t0, t1, o, u_end, st, trace_host, cap_trace, nullptr, nullptr, nullptr, 0);
}
static int sde_solve_adaptive_impl(lrnde_sde* s, const float* u0, const float* W, int32_t nfine, int32_t B, float t0, float t1,
                                   const lrnde_sde_adapt_opts* o, float* u_end, lrnde_stats* st, lrnde_trace_row* trace_host,
                                   int32_t cap_trace, float* rec_u, int2* rec_im_dev, int2* rec_im_host, int rec_cap,
                                   const float* dt0_dev) {
  int rc = sde_check(s, u0, W, u_end ? u_end : u0, B, 1.0f);   // (u_end may be NULL from the layer: no end state asked for)
  if (rc) return rc;
  lrnde_ctx* c = s->drift;
  if (!o || !st || nfine < 1 || !(t1 > t0)) return fail(c, LRNDE_BADARG, "bad arguments (nfine >= 1, t1 > t0)");
  memset(st, 0, sizeof(*st));
  const size_t n = (size_t)B * c->desc.state_dim;
  if (s->ad_n != n) {
    if (s->ad_ws) HIPCHK(c, hipFree(s->ad_ws));
    s->ad_ws = nullptr; s->ad_n = 0;
    HIPCHK(c, hipMalloc(&s->ad_ws, sizeof(float) * 3 * n));
    s->ad_n = n;
  }
  float *ua = s->ad_ws, *ub = s->ad_ws + n, *dW = s->ad_ws + 2 * n;
  {
    const bool host_loop = opt(OPT_SDE_HOST_LOOP) != 0;  // diagnostic: the host-controlled loop below
    if (sde_uses_fast(s) && !host_loop) {
      return sde_adaptive_device(s, u0, W, nfine, B, t0, t1, o, u_end, st, trace_host, cap_trace, ua, ub, rec_u, rec_im_dev, rec_cap,
                                 rec_im_host, dt0_dev);
    }
  }
  HIPCHK(c, hipMemcpyAsync(ua, u0, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  const float h = (t1 - t0) / (float)nfine;
  const float gamma = o->gamma, qmin = o->qmin, qmax = o->qmax, qoldinit = 1e-4f;
  int i = 0;                                                  // position on the path's grid
  int m = (int)(o->dt0 / h); if (m < 1) m = 1;                // step length in grid intervals
  float qold = qoldinit;
  float dtc = o->dt0;   // the controller's proposal as a real number; the step taken is its floor on the grid (SdeCtl::dtc)
  int nb = (int)((n + 255) / 256); if (nb > 1024) nb = 1024;
  while (i < nfine) {
    if (m > nfine - i) m = nfine - i;
    if (++st->iters > o->maxiters) { st->retcode = LRNDE_MAXITERS; break; }
    const float t = t0 + (float)i * h, dt = (float)m * h;
    hipLaunchKernelGGL(k_sde_dw, dim3(nb), dim3(256), 0, c->stream, n, W + (size_t)i * n, W + (size_t)(i + m) * n, dW);
    if ((rc = sde_step_enqueue(s, 0, ua, dW, B, t, dt, o->abstol, o->reltol, o->delta, ub, c->ctrl_host))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const float eest = c->ctrl_host[0].eest_last;
    st->nf += 3; st->eest_last = eest;
    if (eest != eest) { st->retcode = LRNDE_DT_NAN; break; }
    // PI controller on EEst (the form of SURVEY.md 3.5; StochasticDiffEq's constants are the caller's options)
    float q;
    if (eest == 0.0f) q = 1.0f / qmax;
    else {
      const float q11 = fastpow(eest, o->beta1);
      q = q11 / fastpow(qold, o->beta2);
      q = fmaxf(1.0f / qmax, fminf(1.0f / qmin, q / gamma));
    }
    const int accepted = eest <= 1.0f;
    const int ntr = st->naccept + st->nreject;
    if (trace_host && ntr < cap_trace) { trace_host[ntr].t = t; trace_host[ntr].dt = dt; trace_host[ntr].eest = eest; trace_host[ntr].accepted = accepted; }
    dtc = (accepted ? fmaxf(dtc, dt) : dt) / q;
    int mnew = (int)(dtc / h);
    if (mnew < 1) mnew = 1;
    if (accepted) {
      if (rec_u) {
        if (st->naccept >= rec_cap) { st->retcode = LRNDE_CAPACITY; break; }
        HIPCHK(c, hipMemcpyAsync(rec_u + (size_t)st->naccept * n, ub, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
        rec_im_host[st->naccept] = make_int2(i, m);
      }
      st->naccept++;
      qold = fmaxf(eest, qoldinit);
      i += m;
      std::swap(ua, ub);
      m = mnew;
    } else {
      st->nreject++;
      if (m == 1) { st->retcode = LRNDE_DT_LESS_THAN_MIN; break; }  // the path's grid cannot be refined further
      m = mnew < m ? mnew : m - 1;
    }
  }
  st->t_final = t0 + (float)i * h; st->dt_final = (float)m * h;
  if (u_end) HIPCHK(c, hipMemcpyAsync(u_end, ua, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (st->retcode != LRNDE_OK) return fail(c, st->retcode, "adaptive SDE solve stopped with retcode %d at t=%g", st->retcode, (double)st->t_final);
  return LRNDE_OK;
}

// ---- four-stage SRI step (src/perform_step.jl:49-106), diagonal noise, caller-supplied tableau ----
// Composed of the two contexts' f-eval launches and three elementwise kernels that evaluate the reference's expressions
// in their own association order (the arithmetic of lro_sri_step): 15.7 MFLOP per step, latency only.
struct SriPtrs { const float *uprev, *dW, *dZ; float *k[4], *g[4], *H0, *H1, *chi1, *chi2, *chi3; };
__global__ void k_sri_chi(size_t n, SriPtrs p, float dt, float sqdt) {
  const float sqrt3 = sqrtf(3.0f), two_sqdt = 2.0f * sqdt, six_dt = 6.0f * dt, adt = fabsf(dt);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float w = p.dW[i];
    p.chi1[i] = (w * w - adt) / two_sqdt;
    p.chi2[i] = (w + p.dZ[i] / sqrt3) / 2.0f;
    p.chi3[i] = ((w * w) * w - (3.0f * w) * dt) / six_dt;
  }
}
__global__ void k_sri_stage(size_t n, SriPtrs p, lrnde_sri_tableau T, int stage, float dt, float sqdt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float up = p.uprev[i], c2 = p.chi2[i];
    float h0, h1;
    if (stage == 1) {
      const float da = dt * T.a021, db = dt * T.a121, sb = sqdt * T.b121;
      h0 = (up + da * p.k[0][i]) + (T.b021 * c2) * p.g[0][i];
      h1 = (up + db * p.k[0][i]) + sb * p.g[0][i];
    } else if (stage == 2) {
      h0 = (up + dt * (T.a031 * p.k[0][i] + T.a032 * p.k[1][i])) + c2 * (T.b031 * p.g[0][i] + T.b032 * p.g[1][i]);
      h1 = (up + dt * (T.a131 * p.k[0][i] + T.a132 * p.k[1][i])) + sqdt * (T.b131 * p.g[0][i] + T.b132 * p.g[1][i]);
    } else {
      h0 = (up + dt * ((T.a041 * p.k[0][i] + T.a042 * p.k[1][i]) + T.a043 * p.k[2][i])) +
           c2 * ((T.b041 * p.g[0][i] + T.b042 * p.g[1][i]) + T.b043 * p.g[2][i]);
      h1 = (up + dt * ((T.a141 * p.k[0][i] + T.a142 * p.k[1][i]) + T.a143 * p.k[2][i])) +
           sqdt * ((T.b141 * p.g[0][i] + T.b142 * p.g[1][i]) + T.b143 * p.g[2][i]);
    }
    p.H0[i] = h0; p.H1[i] = h1;
  }
}
constexpr int SRI_NB = 64;
__global__ __launch_bounds__(256) void k_sri_final(size_t n, SriPtrs p, lrnde_sri_tableau T, float dt, float abstol, float reltol,
                                                   float delta, float* u, double* part) {
  __shared__ double red[256];
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float k1 = p.k[0][i], k2 = p.k[1][i], k3 = p.k[2][i], k4 = p.k[3][i];
    const float g1 = p.g[0][i], g2 = p.g[1][i], g3 = p.g[2][i], g4 = p.g[3][i];
    const float s3 = ((T.beta31 * g1 + T.beta32 * g2) + T.beta33 * g3) + T.beta34 * g4;
    const float s4 = ((T.beta41 * g1 + T.beta42 * g2) + T.beta43 * g3) + T.beta44 * g4;
    const float E2 = p.chi2[i] * s3 + p.chi3[i] * s4;
    const float sa = ((T.alpha1 * k1 + T.alpha2 * k2) + T.alpha3 * k3) + T.alpha4 * k4;
    const float s1 = ((T.beta11 * g1 + T.beta12 * g2) + T.beta13 * g3) + T.beta14 * g4;
    const float s2 = ((T.beta21 * g1 + T.beta22 * g2) + T.beta23 * g3) + T.beta24 * g4;
    const float up = p.uprev[i];
    const float un = (((up + dt * sa) + E2) + p.dW[i] * s1) + p.chi1[i] * s2;
    u[i] = un;
    const float E1 = dt * (((k1 + k2) + k3) + k4);
    const float sc = abstol + fmaxf_(__builtin_fabsf(up), __builtin_fabsf(un)) * reltol;
    const float r = (delta * E1 + E2) / sc;
    const float sq = r * r;
    acc += (double)sq;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

int lrnde_sde_sri_step(lrnde_sde* s, const lrnde_sri_tableau* tab, const float* uprev, const float* dW, const float* dZ,
                       int32_t B, float t, float dt, float abstol, float reltol, float delta, float* u, float* eest_host,
                       float* reg_val_host) {
  int rc = sde_check(s, uprev, dW, u, B, dt);
  if (rc) return rc;
  lrnde_ctx* c = s->drift;
  if (!tab || !dZ) return fail(c, LRNDE_BADARG, "null pointer");
  if (s->diff->stream != c->stream) return fail(c, LRNDE_BADARG, "drift and diffusion contexts must share a stream");
  const size_t n = (size_t)B * c->desc.state_dim;
  if (s->sri_n != n) {
    if (s->sri_ws) HIPCHK(c, hipFree(s->sri_ws));
    s->sri_ws = nullptr; s->sri_n = 0;
    HIPCHK(c, hipMalloc(&s->sri_ws, sizeof(float) * 13 * n));
    s->sri_n = n;
  }
  if (!s->sri_part) {
    HIPCHK(c, hipMalloc(&s->sri_part, sizeof(double) * SRI_NB));
    HIPCHK(c, hipHostMalloc(&s->sri_part_host, sizeof(double) * SRI_NB));
  }
  SriPtrs p;
  p.uprev = uprev; p.dW = dW; p.dZ = dZ;
  float* w = s->sri_ws;
  for (int j = 0; j < 4; ++j) { p.k[j] = w + (size_t)j * n; p.g[j] = w + (size_t)(4 + j) * n; }
  p.H0 = w + 8 * n; p.H1 = w + 9 * n; p.chi1 = w + 10 * n; p.chi2 = w + 11 * n; p.chi3 = w + 12 * n;
  const float sqdt = sqrtf(fabsf(dt));
  const lrnde_sri_tableau& T = *tab;
  int nb = (int)((n + 255) / 256); if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_sri_chi, dim3(nb), dim3(256), 0, c->stream, n, p, dt, sqdt);
  if ((rc = lrnde_rhs(c, uprev, t, B, p.k[0]))) return rc;                                   // :62
  if ((rc = lrnde_rhs(s->diff, uprev, t + T.c11 * dt, B, p.g[0]))) return rc;                // :63
  const float cf[3] = {T.c02, T.c03, T.c04}, cg[3] = {T.c12, T.c13, T.c14};
  for (int st = 1; st <= 3; ++st) {
    hipLaunchKernelGGL(k_sri_stage, dim3(nb), dim3(256), 0, c->stream, n, p, T, st, dt, sqdt);  // :65-66, :71-72, :77-82
    if ((rc = lrnde_rhs(c, p.H0, t + cf[st - 1] * dt, B, p.k[st]))) return rc;
    if ((rc = lrnde_rhs(s->diff, p.H1, t + cg[st - 1] * dt, B, p.g[st]))) return rc;
  }
  hipLaunchKernelGGL(k_sri_final, dim3(SRI_NB), dim3(256), 0, c->stream, n, p, T, dt, abstol, reltol, delta, u, s->sri_part);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(s->sri_part_host, s->sri_part, sizeof(double) * SRI_NB, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double acc = 0.0;
  for (int i = 0; i < SRI_NB; ++i) acc += s->sri_part_host[i];
  const float ee = (float)sqrt(acc / (double)n);
  if (eest_host) *eest_host = ee;
  if (reg_val_host) *reg_val_host = ee * dt;  // :105
  return LRNDE_OK;
}

// ---- backward building blocks ----
static int ensure_bw(lrnde_ctx* c, int B) {
  if (B == c->bwB) return LRNDE_OK;
  if (c->bw_y) HIPCHK(c, hipFree(c->bw_y));
  if (c->bw_h) HIPCHK(c, hipFree(c->bw_h));
  if (c->bw_dp) HIPCHK(c, hipFree(c->bw_dp));
  if (c->bw_da) HIPCHK(c, hipFree(c->bw_da));
  c->bw_y = c->bw_h = c->bw_dp = c->bw_da = nullptr;
  // (three sets: two alternate between consecutive evaluations; overlapped stage launches rotate through all three)
  HIPCHK(c, hipMalloc(&c->bw_y, sizeof(float) * 3 * (size_t)B * c->desc.state_dim));
  HIPCHK(c, hipMalloc(&c->bw_h, sizeof(float) * 3 * (size_t)B * c->m.Hp));
  HIPCHK(c, hipMalloc(&c->bw_dp, sizeof(float) * 3 * (size_t)B * c->m.Hp));
  HIPCHK(c, hipMalloc(&c->bw_da, sizeof(float) * 3 * (size_t)B * c->m.Hp));   // act'(pre) of a stage-6 evaluation, for stage 7
  c->bwB = B;
  return LRNDE_OK;
}

// (df/dp)^T lam from the scratch left by the last VJP launch (y, h, dpre); gp may be NULL
// riding: the tiles share a VJP launch's grid (k_vjp_q_pg).  Every workgroup of that launch reserves the VJP's ~127 KB of
// dynamic LDS, so a CU holds ONE of them: at B = 512 the 128 VJP workgroups leave 128 CUs, and tiles beyond 128 run in a
// second round after the first (workgroup stamps of tools/vjp_probe: 200 tiles of 32 x 32 started at 0 and at ~9.5 us and the
// launch ended with them, not with the VJP).  32 x 64 tiles are 102 workgroups at the MNIST shape: one round.  In a launch of
// their own 16 x 16 tiles (700 workgroups at 6 KB of LDS, several per CU) hide the loads' latency best (9.5 us vs 11.2 for
// 32 x 32).  LRNDE_PGRAD_TS = 1 | 2 | 3 forces one shape everywhere.
static void pgrad_shape(PgradArgs& g, bool riding) {
  const int ots = opt(OPT_PGRAD_TS);
  g.ts = (ots >= 1 && ots <= 3) ? ots : (riding ? 3 : 1);
  if (!riding && g.ts == 3) g.ts = 2;   // the 32 x 64 shape exists in the VJP launch only (pgrad_tile_any<CC, WIDE>)
  const int em = g.ts == 1 ? 16 : 32, en = g.ts == 1 ? 16 : (g.ts == 2 ? 32 : 64);
  const int th = (g.H + em - 1) / em, td16 = (g.D + em - 1) / em;
  g.nt1c = (g.D + 2 + en - 1) / en; g.nt2c = (g.H + 2 + en - 1) / en;
  g.ntile1 = th * g.nt1c; g.ntile2 = td16 * g.nt2c;
}
static PgradArgs pgrad_args(const lrnde_ctx* c, int B, float t, const float* lam, float* gp, int set, bool riding = false) {
  PgradArgs g{};
  memset(&g, 0, sizeof(g));  // adj_mode = ADJ_HOST: t / lam / gp as given here
  g.accumulate = c->pg_accumulate ? 1 : 0;
  g.D = c->m.D; g.H = c->m.H; g.Hp = c->m.Hp; g.td = c->m.td; g.B = B; g.t = t;
  g.lam = lam; g.gp = gp;
  g.y = c->bw_y + (size_t)set * B * c->desc.state_dim; g.h = c->bw_h + (size_t)set * B * c->m.Hp; g.dpre = c->bw_dp + (size_t)set * B * c->m.Hp;
  // output tiles incl. the two virtual columns (time column, bias): gW1 is H x (D+2), gW2 is D x (H+2)
  pgrad_shape(g, riding);
  return g;
}
static int launch_pgrad_args(lrnde_ctx* c, const PgradArgs& g0) {
  PgradArgs g = g0;
  pgrad_shape(g, false);   // (a deferred GEMM that no VJP launch came to carry: the shape of a launch of its own)
  hipLaunchKernelGGL(k_pgrad, dim3(g.ntile1 + g.ntile2), dim3(256), 0, c->stream, g);
  HIPCHK(c, hipGetLastError());
  // batch-sharded run: the parameter cotangent is a sum over all samples (SURVEY.md §8e caveat 1)
  if (sharded(c)) return comm_allreduce(c, g.gp, g.gp, lrnde_param_count(&c->desc), false);
  return LRNDE_OK;
}
// the deferred GEMM, if one is waiting (before anything reads the mu part of its K vector)
static int flush_pgrad(lrnde_ctx* c) {
  if (!c->pg_pending) return LRNDE_OK;
  c->pg_pending = false;
  return launch_pgrad_args(c, c->pg_args);
}
static int launch_pgrad(lrnde_ctx* c, int B, float t, const float* lam, float* gp) {
  if (!gp) return LRNDE_OK;
  const PgradArgs g = pgrad_args(c, B, t, lam, gp, 0);
  hipLaunchKernelGGL(k_pgrad, dim3(g.ntile1 + g.ntile2), dim3(256), 0, c->stream, g);
  HIPCHK(c, hipGetLastError());
  // batch-sharded run: the parameter cotangent is a sum over all samples (SURVEY.md §8e caveat 1)
  if (sharded(c)) return comm_allreduce(c, gp, gp, lrnde_param_count(&c->desc), false);
  return LRNDE_OK;
}

// fused stage combination handed to the 4-column VJP kernel (lambda part of z_stage = z + dt * sum c_j K_j)
struct StageIn { const float* base; float dt; int nk; const float* k[6]; float c[6]; float* lam_out; };

// dy = J^T lam at (y or the interpolated dense step, t);  gp (optional) = (df/dp)^T lam
static bool vjp_uses_qtile(const lrnde_ctx* c, int B) {
  const bool no_qvjp = opt(OPT_NO_QVJP) != 0;
  return use_qtile(c, B) && !no_qvjp;
}
static int launch_vjp(lrnde_ctx* c, const float* y, const float* dense, float theta, float dense_dt, float t,
                      const float* lam, int B, float* dy, float* gp, const StageIn* sin = nullptr) {
  int rc = ensure_bw(c, B);
  if (rc) return rc;
  if (vjp_uses_qtile(c, B)) {
    VjpQArgs a{};
    memset(&a, 0, sizeof(a));
    a.m = c->m; a.V1q = c->V1q; a.U2q = c->U2q;
    a.B = B; a.t = t; a.y = y; a.dense = dense; a.theta = theta; a.dense_dt = dense_dt; a.lam = lam; a.dy = dy;
    const int set = c->pg_defer ? c->bw_cur : 0;
    a.ysc = c->bw_y + (size_t)set * B * c->desc.state_dim; a.hsc = c->bw_h + (size_t)set * B * c->m.Hp; a.dpsc = c->bw_dp + (size_t)set * B * c->m.Hp;
    if (sin) {
      a.lbase = sin->base; a.ldt = sin->dt; a.lnk = sin->nk; a.lam_out = sin->lam_out;
      for (int j = 0; j < sin->nk; ++j) { a.lk[j] = sin->k[j]; a.lc[j] = sin->c[j]; }
      for (int j = sin->nk; j < 6; ++j) { a.lk[j] = sin->base; a.lc[j] = 0.0f; }  // the kernel always loads six terms (adds +-0)
      lam = sin->lam_out;  // what the parameter-gradient GEMM reads
    }
    const size_t smq = smem_bytes_vq(c->m.KQ1p, c->m.KQ2p, c->m.RG1, c->m.RG2);
    const int qcv = vjp_qcols(B); a.qcols = qcv;
    const int nvjp = (B + qcv - 1) / qcv;
    const bool kt1 = (c->desc.hidden_dim + 3) / 4 == (QSB2 - 1) * QSQ + 1;  // real k-quads in the last phase-3 block (see launch_step)
    if (c->pg_defer) {
      // this VJP's launch carries the GEMM of the previous evaluation; its own GEMM waits for the next launch (or flush_pgrad)
      const bool had = c->pg_pending;
      const PgradArgs prev = c->pg_args;
      if (kt1) {
        if (had) hipLaunchKernelGGL(k_vjp_q_pg<1>, dim3(nvjp + prev.ntile1 + prev.ntile2), dim3(QNT), smq, c->stream, a, prev, nvjp);
        else hipLaunchKernelGGL(k_vjp_q<1>, dim3(nvjp), dim3(QNT), smq, c->stream, a);
      } else {
        if (had) hipLaunchKernelGGL(k_vjp_q_pg<4>, dim3(nvjp + prev.ntile1 + prev.ntile2), dim3(QNT), smq, c->stream, a, prev, nvjp);
        else hipLaunchKernelGGL(k_vjp_q<4>, dim3(nvjp), dim3(QNT), smq, c->stream, a);
      }
      HIPCHK(c, hipGetLastError());
      if (had && sharded(c)) { const int rcc = comm_allreduce(c, prev.gp, prev.gp, lrnde_param_count(&c->desc), false); if (rcc) return rcc; }
      c->pg_pending = gp != nullptr;
      if (gp) c->pg_args = pgrad_args(c, B, t, lam, gp, set, true);
      c->bw_cur ^= 1;
      return LRNDE_OK;
    }
    if (kt1) hipLaunchKernelGGL(k_vjp_q<1>, dim3(nvjp), dim3(QNT), smq, c->stream, a);
    else hipLaunchKernelGGL(k_vjp_q<4>, dim3(nvjp), dim3(QNT), smq, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return launch_pgrad(c, B, t, lam, gp);
  }
  if (sin) return fail(c, LRNDE_BADARG, "fused stage input needs the 4-column VJP kernel");
  VjpArgs a{};
  memset(&a, 0, sizeof(a));
  a.m = c->m; a.V1p = reinterpret_cast<const f32x4*>(c->V1p); a.U2p = reinterpret_cast<const f32x4*>(c->U2p);
  a.B = B; a.t = t; a.y = y; a.dense = dense; a.theta = theta; a.dense_dt = dense_dt; a.lam = lam; a.dy = dy;
  a.ysc = c->bw_y; a.hsc = c->bw_h; a.dpsc = c->bw_dp;
  const int nwg = (B + NB - 1) / NB;
  const size_t sm = smem_bytes(c->m.Dp, c->m.Hp) + (size_t)c->m.Hp * NB * sizeof(float) + 64;
  if (vecw(c) == 4) hipLaunchKernelGGL(k_vjp<4>, dim3(nwg), dim3(NT), sm, c->stream, a);
  else hipLaunchKernelGGL(k_vjp<1>, dim3(nwg), dim3(NT), sm, c->stream, a);
  HIPCHK(c, hipGetLastError());
  return launch_pgrad(c, B, t, lam, gp);
}

int lrnde_vjp(lrnde_ctx* c, const float* y, float t, const float* lam, int32_t B, float* dy, float* gp) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!y || !lam || !dy) return fail(c, LRNDE_BADARG, "null pointer");
  return launch_vjp(c, y, nullptr, 0.f, 0.f, t, lam, B, dy, gp);
}

}  // extern "C"
#include "lrnde_sde_bwd.hpp"
#include "lrnde_sde_bwd_fused.hpp"
#include "lrnde_sde_node.hpp"
extern "C" {
// ---- backward drivers -----------------------------------------------------------------------
}  // extern "C"
namespace {

struct AdjVec {  // device vectors of the augmented adjoint state [lambda (local columns); mu (replicated)]
  float *z, *zn, *zs, *ut, *K[7];
  size_t N;         // local length n_lam + P
  size_t n_lam, P;  // split of the vector; the norm runs over n_lam * nranks + P elements
};

int adj_alloc(lrnde_ctx* c, size_t N, AdjVec& v) {
  if (c->adj_elems < 12 * N) {   // (the 12th vector: third stage-lambda buffer of the overlapped launches)
    if (c->adj) HIPCHK(c, hipFree(c->adj));
    c->adj = nullptr;
    HIPCHK(c, hipMalloc(&c->adj, sizeof(float) * 12 * N));
    c->adj_elems = 12 * N;
  }
  if (!c->adj_part) {  // [256 lambda partials][256 mu partials][64 per-rank lambda sums]
    HIPCHK(c, hipMalloc(&c->adj_part, sizeof(double) * (512 + 64 + ADJ_MU_TILE_MAX)));  // + the per-tile mu partials (ADJ_MU_TILE_OFF)
    HIPCHK(c, hipHostMalloc(&c->adj_part_host, sizeof(double) * (512 + 64)));
  }
  v.N = N; v.n_lam = N; v.P = 0; v.z = c->adj; v.zn = c->adj + N; v.zs = c->adj + 2 * N; v.ut = c->adj + 3 * N;
  for (int j = 0; j < 7; ++j) v.K[j] = c->adj + (4 + j) * N;
  return LRNDE_OK;
}

int vec_axpy(lrnde_ctx* c, float* out, const float* base, float dt, int nk, const float* const* k, const float* coef, size_t n) {
  AxArgs a{};
  a.out = out; a.base = base; a.dt = dt; a.nk = nk; a.n = n;
  for (int j = 0; j < 7; ++j) { a.k[j] = j < nk ? k[j] : nullptr; a.c[j] = j < nk ? coef[j] : 0.f; }
  int nb = (int)((n + 255) / 256); if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_axpy, dim3(nb), dim3(256), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  return LRNDE_OK;
}

// sqrt(sum(((num[-num2]) / (abstol + max(|sa|,|sb|)*reltol))^2) / n_total), fp64 accumulation, over the augmented
// vector [lambda; mu]: the lambda part is summed over all ranks (exact gather of one fp64 sum per rank, added in
// rank order), the replicated mu part is counted once; n_total = n_lam * nranks + P.
int norm_readback(lrnde_ctx* c, size_t n_lam, size_t P, float* out);
int vec_norm(lrnde_ctx* c, const float* num, const float* num2, const float* sa, const float* sb, float abstol,
             float reltol, size_t n_lam, size_t P, float* out) {
  NormArgs a{};
  a.num = num; a.num2 = num2; a.sa = sa; a.sb = sb; a.abstol = abstol; a.reltol = reltol; a.n = n_lam; a.part = c->adj_part;
  if (P) {  // one launch for both parts (same per-block sums as two k_norm launches)
    NormArgs b = a;
    b.num = num + n_lam; b.num2 = num2 ? num2 + n_lam : nullptr; b.sa = sa + n_lam; b.sb = sb ? sb + n_lam : nullptr;
    b.n = P; b.part = c->adj_part + 256;
    hipLaunchKernelGGL(k_norm2, dim3(512), dim3(256), 0, c->stream, a, b);
  } else {
    hipLaunchKernelGGL(k_norm, dim3(256), dim3(256), 0, c->stream, a);
  }
  HIPCHK(c, hipGetLastError());
  return norm_readback(c, n_lam, P, out);
}
// the per-block sums in c->adj_part (lambda part [0,256), mu part [256,512)) -> the rms over [lambda of all ranks; mu]
int norm_readback(lrnde_ctx* c, size_t n_lam, size_t P, float* out) {
  const int nr = sharded(c) ? c->nranks : 1;
  if (sharded(c)) {
    if (nr > 64) return fail(c, LRNDE_UNSUPPORTED, "more than 64 ranks");
    hipLaunchKernelGGL(k_rank_slot, dim3(1), dim3(64), 0, c->stream, c->adj_part, c->adj_part + 512, c->rank, nr);
    { const int rcc = comm_allreduce(c, c->adj_part + 512, c->adj_part + 512, nr, true); if (rcc) return rcc; }
  }
  HIPCHK(c, hipMemcpyAsync(c->adj_part_host, c->adj_part, sizeof(double) * (512 + 64), hipMemcpyDeviceToHost, c->stream));
  // the controller waits for this read-back once per adjoint step: poll an event instead of a blocking stream wait
  if (!c->ev_norm) HIPCHK(c, hipEventCreateWithFlags(&c->ev_norm, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_norm, c->stream));
  for (;;) {
    const hipError_t q = hipEventQuery(c->ev_norm);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return fail(c, LRNDE_HIP_ERROR, "waiting for the norm read-back: %s", hipGetErrorString(q));
  }
  double s = 0.0;
  if (sharded(c)) { for (int r = 0; r < nr; ++r) s += c->adj_part_host[512 + r]; }
  else { for (int i = 0; i < 256; ++i) s += c->adj_part_host[i]; }
  if (P) for (int i = 0; i < 256; ++i) s += c->adj_part_host[256 + i];
  *out = (float)sqrt(s / ((double)n_lam * (double)nr + (double)P));
  return LRNDE_OK;
}

#include "lrnde_adams.hpp"

// adjoint RHS in reversed time s = -t: K = [J^T lambda; (df/dp)^T lambda] at y(t) from the dense record
int adj_rhs(lrnde_ctx* c, const std::vector<float>& dt_, const std::vector<float>& dd_, int B, size_t n,
            const float* zs, float sgt, float* K, const StageIn* sin = nullptr) {
  const float t = -sgt;
  int lo = 0, hi = (int)dt_.size() - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (dt_[mid] <= t) lo = mid; else hi = mid - 1; }
  const float theta = (t - dt_[lo]) / dd_[lo];
  return launch_vjp(c, nullptr, c->dense + (size_t)lo * REC_ARRAYS * n, theta, dd_[lo], t, zs, B, K, K + n, sin);
}

struct AdjImpulse { float s; const float* du; };  // a cotangent added to lambda when the reversed solve reaches s = -t_saved

// adaptive Tsit5 on device vectors, host-side controller (mirror of the forward loop / the oracle's
// lro_solve_ex), integrating s from s0 to s1 with tstops; only the end state is kept
template <class RHS, class RHSF>
int vec_tsit5_solve(lrnde_ctx* c, AdjVec& v, RHS rhs, RHSF rhs_fused, bool fuse_stage, float s0, float s1, float abstol,
                    float reltol, int maxiters, int exact_pow, const std::vector<float>& tstops,
                    const std::vector<AdjImpulse>& impulses, lrnde_stats* st) {
  const size_t N = v.N;
  const float gamma = 0.9f, qmin = 0.2f, qmax = 10.0f, qoldinit = 1e-4f;
  const float beta1 = (float)(7.0 / 50.0), beta2 = (float)(2.0 / 25.0);
  float A[21], BT[7];
  for (int i = 0; i < 21; ++i) A[i] = (float)Tsit5::A[i];
  for (int i = 0; i < 7; ++i) BT[i] = (float)Tsit5::BT[i];
  const float cs[6] = {(float)Tsit5::C[0], (float)Tsit5::C[1], (float)Tsit5::C[2], (float)Tsit5::C[3], 1.0f, 1.0f};
  memset(st, 0, sizeof(*st));
  struct DeferGuard { lrnde_ctx* c; ~DeferGuard() { c->pg_defer = false; c->pg_pending = false; } } defer_guard{c};
  int rc;
  float t = s0;
  const float dtmax = s1 - s0;
  const float dtmin = fmaxf(eps_f(s1), eps_f(s0));
  float *z = v.z, *zn = v.zn;
  float* K[7]; for (int j = 0; j < 7; ++j) K[j] = v.K[j];
  // ode_determine_initdt
  float dt;
  {
    if ((rc = rhs(z, t, K[0]))) return rc;
    float d0, d1, d2;
    if ((rc = vec_norm(c, z, nullptr, z, nullptr, abstol, reltol, v.n_lam, v.P, &d0))) return rc;
    if ((rc = vec_norm(c, K[0], nullptr, z, nullptr, abstol, reltol, v.n_lam, v.P, &d1))) return rc;
    float dt0 = ((double)d0 < 1e-5 || (double)d1 < 1e-5) ? 1e-6f : (d0 / d1) / 100.0f;
    dt0 = fminf(dt0, dtmax);
    const float one = 1.0f; const float* kk[1] = {K[0]};
    if ((rc = vec_axpy(c, v.zs, z, dt0, 1, kk, &one, N))) return rc;
    if ((rc = rhs(v.zs, t + dt0, K[1]))) return rc;
    if ((rc = vec_norm(c, K[1], K[0], z, nullptr, abstol, reltol, v.n_lam, v.P, &d2))) return rc;
    d2 = d2 / dt0;
    const float maxd = fmaxf(d1, d2);
    float dt1;
    if ((double)maxd <= 1e-15) dt1 = fmaxf(1e-6f, dt0 * 1e-3f);
    else { const float l10 = (float)log10((double)maxd); const float e = (-(2.0f + l10)) / 5.0f; dt1 = (float)pow(10.0, (double)e); }
    dt = fminf(fminf(100.0f * dt0, dt1), dtmax);
    st->nf = 3; st->dt_init = dt;
  }
  float qold = qoldinit, q11 = 1.0f, dtpropose = dt;
  int accept = 0, iter = 0;
  size_t istop = 0, iimp = 0;
  while (iimp < impulses.size() && impulses[iimp].s <= s0) ++iimp;
  while (istop < tstops.size() && tstops[istop] <= s0) ++istop;
  rc = LRNDE_OK;
  while (t < s1) {
    while (istop < tstops.size() && tstops[istop] <= t) ++istop;
    const float tstop = (istop < tstops.size() && tstops[istop] < s1) ? tstops[istop] : s1;
    if (iter > 0) {
      if (accept) {
        std::swap(z, zn); std::swap(K[0], K[6]); dt = dtpropose;
        // a cotangent impulse at the saved time just reached: lambda += du, K1 re-evaluated at the modified state
        while (iimp < impulses.size() && impulses[iimp].s < t) ++iimp;
        bool hit = false;
        for (; iimp < impulses.size() && impulses[iimp].s == t && t < s1; ++iimp) {
          const float* gi[1] = {impulses[iimp].du}; const float one = 1.0f;
          if ((rc = vec_axpy(c, z, z, 1.0f, 1, gi, &one, v.n_lam))) return rc;
          hit = true;
        }
        if (hit) { if ((rc = rhs(z, t, K[0]))) return rc; st->nf += 1; }
      }
      else dt = dt / fminf(1.0f / qmin, q11 / gamma);
    }
    ++iter;
    dt = fminf(dtmax, dt); dt = fmaxf(dt, dtmin); dt = fminf(fabsf(dt), fabsf(tstop - t));
    if (iter > maxiters) { rc = LRNDE_MAXITERS; break; }
    if (dt != dt) { rc = LRNDE_DT_NAN; break; }
    if (fabsf(dt) <= fabsf(dtmin)) { rc = LRNDE_DT_LESS_THAN_MIN; break; }
    // stages 2..7 (src/perform_step.jl:11-20 on the augmented state)
    c->pg_defer = fuse_stage;
    for (int sidx = 2; sidx <= 7; ++sidx) {
      const int off = (sidx - 2) * (sidx - 1) / 2;
      float* out = (sidx == 7) ? zn : v.zs;
      if (fuse_stage) {
        // the RHS only reads the lambda part of the stage state: it is formed inside the VJP kernel (same arithmetic as
        // k_axpy) and left for the parameter-gradient GEMM, which is deferred into the next stage's launch; the stage
        // lambdas therefore alternate between v.zs and v.ut (free until the error estimate), stage 7's is zn itself.
        // The mu part of a stage state is never needed; zn's is formed after the last GEMM.
        StageIn sin{};
        sin.base = z; sin.dt = dt; sin.nk = sidx - 1; sin.lam_out = (sidx == 7) ? zn : ((sidx & 1) ? v.ut : v.zs);
        for (int j = 0; j < sidx - 1; ++j) { sin.k[j] = K[j]; sin.c[j] = A[off + j]; }
        if ((rc = rhs_fused(sin, t + cs[sidx - 2] * dt, K[sidx - 1]))) return rc;
        continue;
      }
      if ((rc = vec_axpy(c, out, z, dt, sidx - 1, K, A + off, N))) return rc;
      if ((rc = rhs(out, t + cs[sidx - 2] * dt, K[sidx - 1]))) return rc;
    }
    st->nf += 6;
    float eest;
    if (fuse_stage && v.P) {
      if ((rc = flush_pgrad(c))) return rc;
      c->pg_defer = false;
      // mu part of u_{n+1}, utilde and the error norm's sums in one launch (k_adj_err: the values of k_axpy + k_norm2)
      AdjErrArgs e{};
      for (int j = 0; j < 7; ++j) { e.K[j] = K[j]; e.BT[j] = BT[j]; }
      for (int j = 0; j < 6; ++j) e.A7[j] = A[15 + j];
      e.dt = dt; e.z = z; e.zn = zn; e.n_lam = v.n_lam; e.P = v.P; e.abstol = abstol; e.reltol = reltol; e.part = c->adj_part;
      hipLaunchKernelGGL(k_adj_err, dim3(512), dim3(256), 0, c->stream, e);
      HIPCHK(c, hipGetLastError());
      if ((rc = norm_readback(c, v.n_lam, v.P, &eest))) return rc;
    } else {
      if (fuse_stage) { if ((rc = flush_pgrad(c))) return rc; c->pg_defer = false; }
      if ((rc = vec_axpy(c, v.ut, nullptr, dt, 7, K, BT, N))) return rc;
      if ((rc = vec_norm(c, v.ut, nullptr, z, zn, abstol, reltol, v.n_lam, v.P, &eest))) return rc;
    }
    st->eest_last = eest;
    if (eest != eest) { rc = LRNDE_DT_NAN; break; }
    const float ttmp = t + dt;
    float q;
    if (eest == 0.0f) q = 1.0f / qmax;
    else {
      if (exact_pow) { q11 = (float)pow((double)eest, (double)beta1); q = q11 / (float)pow((double)qold, (double)beta2); }
      else { q11 = fastpow(eest, beta1); q = q11 / fastpow(qold, beta2); }
      q = fmaxf(1.0f / qmax, fminf(1.0f / qmin, q / gamma));
    }
    accept = (eest <= 1.0f);
    if (c->adj_trace && c->adj_trace_n < c->adj_trace_cap) {
      lrnde_trace_row& r = c->adj_trace[c->adj_trace_n++];
      r.t = t; r.dt = dt; r.eest = eest; r.accepted = accept;
    }
    if (accept) {
      st->naccept++;
      const float dtnew = dt / q;
      qold = fmaxf(eest, qoldinit);
      // (magnitudes: reversed time s = -t <= 0 — a signed max would take the time nearer zero, and with it an eps far below
      //  the rounding of t + dt; the reference's adjoint runs t from t2 down to t0 with positive times)
      t = (fabsf(ttmp - tstop) < 100.0f * eps_f(fmaxf(fabsf(t), fabsf(tstop)))) ? tstop : ttmp;
      dtpropose = fmaxf(fminf(dtmax, dtnew), fmaxf(eps_f(t), dtmin));
    } else {
      st->nreject++;
    }
  }
  if (accept && rc == LRNDE_OK) std::swap(z, zn);  // z now holds the end state
  if (rc == LRNDE_OK)  // cotangents at the end time itself (a saved start value)
    for (; iimp < impulses.size(); ++iimp) {
      if (impulses[iimp].s < s1) continue;
      const float* gi[1] = {impulses[iimp].du}; const float one = 1.0f;
      int r2 = vec_axpy(c, z, z, 1.0f, 1, gi, &one, v.n_lam);
      if (r2) return r2;
    }
  if (z != v.z) HIPCHK(c, hipMemcpyAsync(v.z, z, sizeof(float) * N, hipMemcpyDeviceToDevice, c->stream));
  st->retcode = rc; st->iters = iter; st->t_final = t; st->dt_final = dt;
  return rc;
}


// ---- the adjoint solve with the controller on the device (lrnde_adjoint.hpp) --------------------------------------
// One segment = s from the current time of the control block to s_end (tstops inside it are handled on the device).
// `impulses`: cotangents added to lambda when the reversed solve reaches a saved time — (reversed time, device pointer
// to a (B, D) cotangent) pairs in ascending s; each ends a segment: the host waits for it, adds the impulse, has K1
// re-evaluated at the modified state (what a callback's u_modified! does upstream) and lets the integrator go on.
int adj_enqueue_eval(lrnde_ctx* c, int B, const AdjArgs& g, int mode, int stage, int j, bool with_prev_pgrad,
                     int prev_mode, int prev_stage, hipStream_t st = nullptr, int ovl = 0) {
  // one VJP launch in device-resolved form; with_prev_pgrad: the launch also carries the parameter-gradient GEMM of the
  // previous evaluation (prev_mode / prev_stage of the same attempt), whose scratch set is the one written last
  VjpQArgs a{};
  memset(&a, 0, sizeof(a));
  a.m = c->m; a.V1q = c->V1q; a.U2q = c->U2q; a.B = B;
  a.adj_mode = mode; a.adj_stage = stage; a.adj_j = j; a.adj = g;
  if (!st) st = c->stream;
  // scratch sets: two alternate; with overlapped stage launches (g.sync) three rotate, because launch id + 1 writes its set while
  // the tiles of launch id still read the set of launch id - 1
  const bool rot3 = g.sync != nullptr && mode == ADJ_STAGE;
  const int set = c->bw_cur, prev_set = rot3 ? (set + 2) % 3 : set ^ 1;
  a.ysc = c->bw_y + (size_t)set * B * c->desc.state_dim; a.hsc = c->bw_h + (size_t)set * B * c->m.Hp; a.dpsc = c->bw_dp + (size_t)set * B * c->m.Hp;
  if (rot3) { a.sync_id = c->adj_launch_id++; a.ovl = ovl; }
  // stages 6 and 7 of an attempt are evaluations at the same point (c6 = c7 = 1): stage 7 takes y, h and act' from stage 6
  const bool reuse_on = !rot3 && opt(OPT_ADJ_NO_REUSE) == 0 && mode == ADJ_STAGE && with_prev_pgrad && prev_mode == ADJ_STAGE;
  const bool reuse7 = reuse_on && stage == 7 && prev_stage == 6;
  if (reuse_on && stage == 6) a.dact_out = c->bw_da + (size_t)set * B * c->m.Hp;
  if (reuse7) a.dact_in = c->bw_da + (size_t)prev_set * B * c->m.Hp;
  c->adj_stage7_reused = reuse7;
  const size_t smq = smem_bytes_vq(c->m.KQ1p, c->m.KQ2p, c->m.RG1, c->m.RG2);
  const int qcv = vjp_qcols(B); a.qcols = qcv;
  const int nvjp = (B + qcv - 1) / qcv;
  const bool kt1 = (c->desc.hidden_dim + 3) / 4 == (QSB2 - 1) * QSQ + 1;
  if (with_prev_pgrad) {
    PgradArgs pg = pgrad_args(c, B, 0.f, nullptr, nullptr, prev_set, true);
    pg.adj_mode = prev_mode; pg.adj_stage = prev_stage; pg.adj_j = j;
    const dim3 grid(nvjp + pg.ntile1 + pg.ntile2);
    if (rot3) {
      if (kt1) hipLaunchKernelGGL((k_vjp_q_pg<1, true>), grid, dim3(QNT), smq, st, a, pg, nvjp);
      else hipLaunchKernelGGL((k_vjp_q_pg<4, true>), grid, dim3(QNT), smq, st, a, pg, nvjp);
    } else if (reuse7) {
      if (kt1) hipLaunchKernelGGL((k_vjp_q_pg<1, false, true>), grid, dim3(QNT), smq, st, a, pg, nvjp);
      else hipLaunchKernelGGL((k_vjp_q_pg<4, false, true>), grid, dim3(QNT), smq, st, a, pg, nvjp);
    } else {
      if (kt1) hipLaunchKernelGGL(k_vjp_q_pg<1>, grid, dim3(QNT), smq, st, a, pg, nvjp);
      else hipLaunchKernelGGL(k_vjp_q_pg<4>, grid, dim3(QNT), smq, st, a, pg, nvjp);
    }
  } else if (rot3) {
    if (kt1) hipLaunchKernelGGL((k_vjp_q<1, true>), dim3(nvjp), dim3(QNT), smq, st, a);
    else hipLaunchKernelGGL((k_vjp_q<4, true>), dim3(nvjp), dim3(QNT), smq, st, a);
  } else {
    if (kt1) hipLaunchKernelGGL(k_vjp_q<1>, dim3(nvjp), dim3(QNT), smq, st, a);
    else hipLaunchKernelGGL(k_vjp_q<4>, dim3(nvjp), dim3(QNT), smq, st, a);
  }
  HIPCHK(c, hipGetLastError());
  c->bw_cur = rot3 ? (set + 1) % 3 : set ^ 1;
  return LRNDE_OK;
}

// stage 7's GEMM after a REUSE launch: y and h are stage 6's (the set written before the last one), dpre and lambda its own
void pgrad_stage7_operands(const lrnde_ctx* c, int B, PgradArgs& pg) {
  if (!c->adj_stage7_reused) return;
  const int s6 = c->bw_cur;   // (bw_cur has moved on: bw_cur ^ 1 is stage 7's set, bw_cur stage 6's)
  pg.y = c->bw_y + (size_t)s6 * B * c->desc.state_dim; pg.h = c->bw_h + (size_t)s6 * B * c->m.Hp;
}

// the parameter-gradient GEMM of the evaluation whose scratch was written by the LAST VJP launch, by itself; on a
// sharded handle followed by the all-reduce of mu's slot (every rank's sum over its own columns -> the batch sum)
int adj_enqueue_pgrad(lrnde_ctx* c, int B, const AdjArgs& g, int mode, int stage, int j) {
  PgradArgs pg = pgrad_args(c, B, 0.f, nullptr, nullptr, c->bw_cur ^ 1);
  pg.adj_mode = mode; pg.adj_stage = stage; pg.adj_j = j;
  if (mode == ADJ_STAGE && stage == 7) pgrad_stage7_operands(c, B, pg);
  hipLaunchKernelGGL(k_pgrad_adj, dim3(pg.ntile1 + pg.ntile2), dim3(256), 0, c->stream, pg, g);
  HIPCHK(c, hipGetLastError());
  return LRNDE_OK;
}

// per-rank lambda sum -> rank slots, exact gather (sharded handles); the partial block `part` is [512 + 64] doubles
int adj_enqueue_slots(lrnde_ctx* c, double* part) {
  if (!sharded(c)) return LRNDE_OK;
  if (c->nranks > 64) return fail(c, LRNDE_UNSUPPORTED, "more than 64 ranks");
  hipLaunchKernelGGL(k_rank_slot, dim3(1), dim3(64), 0, c->stream, part, part + 512, c->rank, c->nranks);
  return comm_allreduce(c, part + 512, part + 512, c->nranks, true);
}

int adj_norm_into(lrnde_ctx* c, const float* num, const float* num2, const float* sa, float abstol, float reltol, size_t n_lam,
                  size_t P, double* part) {
  NormArgs a{};
  a.num = num; a.num2 = num2; a.sa = sa; a.sb = nullptr; a.abstol = abstol; a.reltol = reltol; a.n = n_lam; a.part = part;
  NormArgs b = a;
  b.num = num + n_lam; b.num2 = num2 ? num2 + n_lam : nullptr; b.sa = sa + n_lam; b.n = P; b.part = part + 256;
  hipLaunchKernelGGL(k_norm2, dim3(512), dim3(256), 0, c->stream, a, b);
  HIPCHK(c, hipGetLastError());
  return adj_enqueue_slots(c, part);
}

int adj_solve_device(lrnde_ctx* c, AdjVec& v, int B, float s0, float s1, float abstol, float reltol, int maxiters, int exact_pow,
                     const std::vector<float>& tstops, const std::vector<AdjImpulse>& impulses, int nrec, lrnde_stats* st) {
  int rc;
  memset(st, 0, sizeof(*st));
  if (sharded(c)) return fail(c, LRNDE_UNSUPPORTED, "sharded handles use the host-controlled adjoint loop");
  if ((rc = ensure_bw(c, B))) return rc;
  if (!c->adj_ctl) {
    HIPCHK(c, hipMalloc(&c->adj_ctl, sizeof(AdjCtrl) * 2));
    HIPCHK(c, hipHostMalloc(&c->adj_ctl_host, sizeof(AdjCtrl) * 2));
    HIPCHK(c, hipMalloc(&c->adj_ipart, sizeof(double) * 3 * 576));
    HIPCHK(c, hipEventCreateWithFlags(&c->adj_ev[0], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->adj_ev[1], hipEventDisableTiming));
    HIPCHK(c, hipHostMalloc(&c->adj_hstat, sizeof(int) * 16, hipHostMallocMapped));
    memset(c->adj_hstat, 0, sizeof(int) * 16);
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->adj_hstat_dev, c->adj_hstat, 0));
  }
  if ((int)tstops.size() > c->adj_stops_cap) {
    if (c->adj_stops) HIPCHK(c, hipFree(c->adj_stops));
    c->adj_stops = nullptr; c->adj_stops_cap = 0;
    HIPCHK(c, hipMalloc(&c->adj_stops, sizeof(float) * (tstops.size() + 8)));
    c->adj_stops_cap = (int)tstops.size() + 8;
  }
  const bool one_begin = c->adj_init_src != nullptr && tstops.size() <= 8;
  if (!tstops.empty() && !one_begin)
    HIPCHK(c, hipMemcpyAsync(c->adj_stops, tstops.data(), sizeof(float) * tstops.size(), hipMemcpyHostToDevice, c->stream));
  AdjArgs g{};
  memset(&g, 0, sizeof(g));
  g.ctl = c->adj_ctl; g.base = c->adj; g.N = v.N; g.n_lam = v.n_lam; g.P = v.P;
  g.dense = c->dense; g.dense_t = c->dense_t; g.dense_dt = c->dense_dt; g.nrec = nrec;
  g.stops = c->adj_stops; g.nstops = (int)tstops.size();
  g.s0 = s0; g.dtmax = s1 - s0; g.dtmin = fmaxf(eps_f(s1), eps_f(s0));
  g.abstol = abstol; g.reltol = reltol; g.maxiters = maxiters; g.exact_pow = exact_pow;
  g.part = c->adj_part; g.ipart = c->adj_ipart; g.nranks = 1; g.use_slots = 0;
  // LRNDE_ADJ_MU_FOLD=1: the mu part of an attempt's error norm and of z_new rides in the last GEMM's tiles (no launch of its
  // own).  Built as DESIGN 4.4's "next lever" and measured (round 3): the tiles' tail — a dependent round trip for K1..K6 and
  // z after the GEMM — makes that launch longer than k_adj_err_dev was (+6.5 us per attempt net), so it stays opt-in.
  bool fold_mu = false;
  {
    const PgradArgs pg0 = pgrad_args(c, B, 0.f, nullptr, nullptr, 0);
    const int nt0 = pg0.ntile1 + pg0.ntile2;
    fold_mu = v.P != 0 && !opt(OPT_ADJ_ERR_ONE_LAUNCH) && opt(OPT_ADJ_MU_FOLD) && nt0 <= ADJ_MU_TILE_MAX;
    g.mu_tiles = fold_mu ? nt0 : 0;
  }
  // LRNDE_ADJ_OVERLAP=1: the stage launches of an attempt alternate between the handle's stream and a second one, so that
  // launch s + 1 starts WHILE launch s runs — on the half of the chip a stage launch leaves idle — and does everything that
  // needs the record only (entry, y, the first GEMM phase) before it waits, on the device, for launch s's arrivals
  // (lrnde_adjoint.hpp).  Not with cotangent impulses (segments), which keep the one-stream order.
  const bool ovl_on = opt(OPT_ADJ_OVERLAP) != 0 && impulses.empty() && !opt(OPT_ADJ_ERR_ONE_LAUNCH);
  if (ovl_on) {
    if (!c->adj_stream2) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->adj_stream2, hipStreamNonBlocking));
      for (int i = 0; i < 2; ++i) {
        HIPCHK(c, hipEventCreateWithFlags(&c->adj_evA[i], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->adj_evB[i], hipEventDisableTiming));
      }
      HIPCHK(c, hipMalloc(&c->adj_sync, sizeof(int) * 32));
    }
    HIPCHK(c, hipMemsetAsync(c->adj_sync, 0, sizeof(int) * 32, c->stream));
    g.sync = c->adj_sync;
    c->adj_launch_id = 0;
    c->adj_hstat[11] = 0;   // (the prologue copies sync[1], the timeout word, here)
  }
  hipStream_t const sA = c->stream, sB = ovl_on ? c->adj_stream2 : c->stream;
  const size_t N = v.N, n = v.n_lam;
  float* const zb0 = c->adj; float* const K0 = c->adj + 4 * N; float* const K1 = c->adj + 5 * N;
  AdjErrArgs e{};
  memset(&e, 0, sizeof(e));
  for (int q = 0; q < 7; ++q) e.BT[q] = (float)Tsit5::BT[q];
  for (int q = 0; q < 6; ++q) e.A7[q] = (float)Tsit5::A[15 + q];
  e.n_lam = v.n_lam; e.P = v.P; e.abstol = abstol; e.reltol = reltol; e.part = c->adj_part;

  if (one_begin) {
    AdjBegin b{};
    memset(&b, 0, sizeof(b));
    b.z = v.z; b.src = c->adj_init_src; b.n = v.n_lam; b.N = v.N; b.ctl = c->adj_ctl; b.s0 = s0;
    b.stops = c->adj_stops; b.nstops = (int)tstops.size();
    for (size_t k = 0; k < tstops.size(); ++k) b.sv[k] = tstops[k];
    int nb = (int)((v.N + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_adj_begin, dim3(nb), dim3(256), 0, c->stream, b);
  } else {
    if (c->adj_init_src) {
      HIPCHK(c, hipMemsetAsync(v.z, 0, sizeof(float) * v.N, c->stream));
      HIPCHK(c, hipMemcpyAsync(v.z, c->adj_init_src, sizeof(float) * v.n_lam, hipMemcpyDeviceToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_adj_ctrl_init, dim3(1), dim3(1), 0, c->stream, c->adj_ctl, s0);
  }
  c->adj_init_src = nullptr;
  size_t iseg = 0;
  while (iseg < impulses.size() && impulses[iseg].s <= s0) ++iseg;  // a cotangent at the start time is the caller's lambda(s0)
  bool first_seg = true;
  int extra_nf = 0;
  AdjCtrl fin;
  memset(&fin, 0, sizeof(fin));
  for (;;) {
    const bool last_seg = iseg >= impulses.size() || !(impulses[iseg].s < s1);
    g.s1 = last_seg ? s1 : impulses[iseg].s;
    // (re-)evaluate K1 = rhs(z, t) at the state of ctl[0]; first segment: the rest of ode_determine_initdt
    c->bw_cur = 0;
    if ((rc = adj_enqueue_eval(c, B, g, ADJ_FSAL, 0, 0, false, 0, 0))) return rc;
    if ((rc = adj_enqueue_pgrad(c, B, g, ADJ_FSAL, 0, 0))) return rc;
    if (first_seg) {
      if ((rc = adj_norm_into(c, zb0, nullptr, zb0, abstol, reltol, v.n_lam, v.P, c->adj_ipart))) return rc;          // d0
      if ((rc = adj_norm_into(c, K0, nullptr, zb0, abstol, reltol, v.n_lam, v.P, c->adj_ipart + 576))) return rc;     // d1
      if ((rc = adj_enqueue_eval(c, B, g, ADJ_INIT_B, 0, 0, false, 0, 0))) return rc;
      if ((rc = adj_enqueue_pgrad(c, B, g, ADJ_INIT_B, 0, 0))) return rc;
      if ((rc = adj_norm_into(c, K1, K0, zb0, abstol, reltol, v.n_lam, v.P, c->adj_ipart + 2 * 576))) return rc;      // d2 * dt0
    } else {
      ++extra_nf;
    }
    first_seg = false;
    // Attempts are enqueued ONE ahead of what the device has decided: the first launch of attempt j publishes the
    // integrator's status in pinned host memory (AdjArgs::hstat), and attempt j+1 is enqueued when attempt j is known
    // to be running (its remaining seven launches, ~130 us, cover the host's enqueue); no copy packet sits between the
    // kernels, and an attempt that would end the segment is followed by the next one's first launch only (maybe_last).
    g.hstat = c->adj_hstat_dev; g.seq0 = c->adj_seq;
    volatile int* hs = c->adj_hstat;
    int j = 0;
    bool done = false;
    // every early exit of this segment: nothing of it stays in flight, and the sequence numbers its launches may still have
    // written (up to seq0 + j + 1) are retired, so the next solve's first wait cannot be satisfied by a stale report
    auto bail = [&](int code) -> int {
      hipStreamSynchronize(c->stream);
      if (c->adj_stream2) hipStreamSynchronize(c->adj_stream2);
      c->adj_seq += j + 1;
      c->after_first_attempt = nullptr;
      return code;
    };
    // maybe_last: the attempt whose prologue reported last reaches the end of the segment if it is accepted.  The next
    // attempt is then enqueued as its FIRST launch only (whose prologue takes that decision); its other seven follow
    // once the report says the solve goes on (a rejection: the stream idles for one host round trip) — otherwise the
    // solve would always end with eight launches that find nothing to do.
    bool maybe_last = false;
    int trace_prev = -1, trace_nacc = 0;
    auto enqueue_rest = [&](int jj) -> int {
      if (ovl_on) HIPCHK(c, hipStreamWaitEvent(sB, c->adj_evA[jj & 1], 0));   // (recorded on the handle's stream ahead of this attempt's stage 2)
      for (int sidx = 3; sidx <= 7; ++sidx) {
        const int r = adj_enqueue_eval(c, B, g, ADJ_STAGE, sidx, jj, true, ADJ_STAGE, sidx - 1, (sidx & 1) ? sB : sA, ovl_on ? 1 : 0);
        if (r) return r;
      }
      if (ovl_on) {   // the end of the attempt needs stage 7 (second stream) and its tiles
        HIPCHK(c, hipEventRecord(c->adj_evB[jj & 1], sB));
        HIPCHK(c, hipStreamWaitEvent(sA, c->adj_evB[jj & 1], 0));
      }
      const bool split_off = opt(OPT_ADJ_ERR_ONE_LAUNCH) != 0;  // diagnostic: the error norm in one launch of its own
      if (split_off) {
        const int r = adj_enqueue_pgrad(c, B, g, ADJ_STAGE, 7, jj);
        if (r) return r;
        hipLaunchKernelGGL(k_adj_err_dev, dim3(512), dim3(256), 0, c->stream, e, g, jj, 0);
      } else {
        PgradArgs pg = pgrad_args(c, B, 0.f, nullptr, nullptr, ovl_on ? (c->bw_cur + 2) % 3 : c->bw_cur ^ 1);
        pg.adj_mode = ADJ_STAGE; pg.adj_stage = 7; pg.adj_j = jj;
        if (!ovl_on) pgrad_stage7_operands(c, B, pg);
        const int nt = pg.ntile1 + pg.ntile2;
        hipLaunchKernelGGL(k_pgrad_adj_err, dim3(nt + 256), dim3(256), 0, c->stream, pg, g, e, nt, jj, fold_mu ? 1 : 0);
        if (!fold_mu) hipLaunchKernelGGL(k_adj_err_dev, dim3(256), dim3(256), 0, c->stream, e, g, jj, 1);
      }
      HIPCHK(c, hipGetLastError());
      return LRNDE_OK;
    };
    while (!done) {
      if (ovl_on) { c->bw_cur = 0; HIPCHK(c, hipEventRecord(c->adj_evA[j & 1], sA)); }   // (the second stream starts this attempt's stage 3 from here)
      if ((rc = adj_enqueue_eval(c, B, g, ADJ_STAGE, 2, j, false, ADJ_STAGE, 1, sA, 0))) return bail(rc);
      const bool rest_ahead = !maybe_last;
      if (rest_ahead && (rc = enqueue_rest(j))) return bail(rc);
      if (c->after_first_attempt) {  // (the handle's stream now holds ~150 us of work: time for the caller's side enqueues)
        auto fn = std::move(c->after_first_attempt);
        c->after_first_attempt = nullptr;
        if ((rc = fn())) return bail(rc);
      }
      ++j;
      // wait for the prologue of the attempt just enqueued (bounded: a faulted or hung queue must not hang the caller)
      const int want = g.seq0 + j;
      long spins = 0;
      const SpinDeadline deadline;
      while ((int)(__atomic_load_n(hs, __ATOMIC_ACQUIRE) - want) < 0) {
        if (((++spins) & 0xFFFFF) == 0) {
          const hipError_t q = hipStreamQuery(c->stream);
          if (q != hipSuccess && q != hipErrorNotReady) return bail(fail(c, LRNDE_HIP_ERROR, "adjoint loop: %s", hipGetErrorString(q)));
          if (q == hipSuccess && (int)(__atomic_load_n(hs, __ATOMIC_ACQUIRE) - want) < 0)
            return bail(fail(c, LRNDE_HIP_ERROR, "adjoint loop: the stream drained without the status of attempt %d", j - 1));
          if (q == hipErrorNotReady && deadline.expired()) { c->adj_seq += j + 1; return LRNDE_HUNG(c, "adjoint loop"); }
        }
      }
      if (c->adj_trace) {
        // the report of attempt j-1's prologue: its (s, dt), and the error estimate / decision of the attempt before it
        if (j > 1 && trace_prev >= 0) {
          lrnde_trace_row& r = c->adj_trace[trace_prev];
          r.eest = __builtin_bit_cast(float, (int)hs[9]); r.accepted = (hs[6] > trace_nacc);
        }
        trace_prev = -1; trace_nacc = hs[6];
        if (hs[1] == ST_RUNNING && c->adj_trace_n < c->adj_trace_cap) {
          trace_prev = c->adj_trace_n++;
          lrnde_trace_row& r = c->adj_trace[trace_prev];
          r.t = __builtin_bit_cast(float, (int)hs[2]); r.dt = __builtin_bit_cast(float, (int)hs[3]); r.eest = 0.f; r.accepted = -1;
        }
      }
      if (hs[1] != ST_RUNNING) done = true;
      else {
        if (!rest_ahead && (rc = enqueue_rest(j - 1))) return bail(rc);
        const float te = __builtin_bit_cast(float, (int)hs[2]) + __builtin_bit_cast(float, (int)hs[3]);
        maybe_last = fabsf(te - g.s1) <= 100.0f * eps_f(fmaxf(fabsf(te), fabsf(g.s1)));
      }
      if (j > maxiters + 16) break;
    }
    c->adj_seq += j;
    if (ovl_on && hs[11]) return bail(fail(c, LRNDE_HIP_ERROR, "adjoint loop: a wait between overlapped stage launches timed out"));
    if (done) {
      // the report of the last attempt's prologue carries the integrator's final state (adj_hstat_fill): no read-back
      // copy on the stream, no synchronisation here — the caller's own, after its output copies, is the only one
      fin.status = hs[1]; fin.t = __builtin_bit_cast(float, (int)hs[2]); fin.dt = __builtin_bit_cast(float, (int)hs[3]);
      fin.cur = hs[4]; fin.nf = hs[5]; fin.naccept = hs[6]; fin.nreject = hs[7]; fin.iter = hs[8];
      fin.eest_last = __builtin_bit_cast(float, (int)hs[9]); fin.dt_init = __builtin_bit_cast(float, (int)hs[10]);
    } else {  // launch cap reached with the solve still running
      HIPCHK(c, hipMemcpyAsync(c->adj_ctl_host, c->adj_ctl + (j & 1), sizeof(AdjCtrl), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      fin = c->adj_ctl_host[0];
    }
    if (fin.status != ST_DONE) break;  // error status (or still running after the launch cap: MaxIters)
    if (last_seg) break;
    // cotangent impulse at the saved time just reached, then go on
    const float* gi[1] = {impulses[iseg].du}; const float one = 1.0f;
    float* zc = c->adj + (size_t)fin.cur * N;
    if ((rc = vec_axpy(c, zc, zc, 1.0f, 1, gi, &one, n))) return rc;
    ++iseg;
    hipLaunchKernelGGL(k_adj_ctrl_continue, dim3(1), dim3(1), 0, c->stream, c->adj_ctl, j & 1);
  }
  st->retcode = (fin.status == ST_DONE) ? LRNDE_OK : (fin.status == ST_RUNNING ? LRNDE_MAXITERS : fin.status);
  st->nf = fin.nf + extra_nf; st->naccept = fin.naccept; st->nreject = fin.nreject; st->iters = fin.iter;
  st->t_final = fin.t; st->dt_final = fin.dt; st->eest_last = fin.eest_last; st->dt_init = fin.dt_init;
  // the end state is zb[cur]; cotangents at the end time itself (a saved start value) are added to it
  float* zend = c->adj + (size_t)fin.cur * N;
  if (st->retcode == LRNDE_OK)
    for (; iseg < impulses.size(); ++iseg) {
      const float* gi[1] = {impulses[iseg].du}; const float one = 1.0f;
      if ((rc = vec_axpy(c, zend, zend, 1.0f, 1, gi, &one, n))) return rc;
    }
  v.z = zend;  // (the caller copies dx / dp out of it; no move to the first buffer)
  if (st->retcode != LRNDE_OK) hipStreamSynchronize(c->stream);  // nothing of a failed solve is left in flight
  return st->retcode;
}

}  // namespace
extern "C" {

// gradient of the local regularisation value w.r.t. p (reverse sweep through one Tsit5 step with
// k1, dt, uprev constant: src/layers/neural_ode.jl:40, src/perform_step.jl:3-47).  gp: device (P).
}  // extern "C"
// the reverse sweep proper, enqueued without a host synchronisation: the forward step's state (uprev, u, k1..k7, g6) is
// in the state workspace (by stream order), its scalars (EEst, the two stiffness rms values) are given
static int step_reg_sweep(lrnde_ctx* c, const float* uprev, int32_t B, float t, float dt, float abstol, float reltol,
                          int32_t reg_type, float eest, float stiff_num, float stiff_den, float* gp);
extern "C" {
int lrnde_step_reg_grad(lrnde_ctx* c, const float* uprev, const float* k1, int32_t B, float t, float dt,
                        float abstol, float reltol, int32_t reg_type, float* gp, float* reg_val_host) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!uprev || !k1 || !gp) return fail(c, LRNDE_BADARG, "null pointer");
  // forward step (keeps k2..k6, g6, u, k7 in the state workspace) and its scalars
  float ee, re, rs;
  if ((rc = lrnde_perform_step(c, uprev, k1, B, t, dt, abstol, reltol, nullptr, nullptr, &ee, &re, &rs))) return rc;
  const Ctrl fin = c->ctrl_host[0];
  if (reg_val_host) *reg_val_host = (reg_type == LRNDE_REG_STIFFNESS_ESTIMATE) ? rs : re;
  if ((rc = step_reg_sweep(c, uprev, B, t, dt, abstol, reltol, reg_type, fin.eest_last, fin.stiff_num, fin.stiff_den, gp))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}
}  // extern "C"
static int step_reg_sweep(lrnde_ctx* c, const float* uprev, int32_t B, float t, float dt, float abstol, float reltol,
                          int32_t reg_type, float eest, float stiff_num, float stiff_den, float* gp) {
  int rc;
  const size_t n = (size_t)B * c->desc.state_dim;
  const size_t P = lrnde_param_count(&c->desc);
  float* S = c->state;  // ubuf0 (uprev) ubuf1 (u) kfsal0 (k1) kfsal1 (k7) ks0..4 (k2..k6) g6
  const float* kk[7] = {S + 2 * n, S + 4 * n, S + 5 * n, S + 6 * n, S + 7 * n, S + 8 * n, S + 3 * n};
  const float* u = S + n; const float* g6 = S + 9 * n;
  // work vectors: kbar[1..6] (k2..k7), ub, g6b, xs, xb  -> reuse the adjoint buffer
  AdjVec v;
  if ((rc = adj_alloc(c, n > P ? n : P, v))) return rc;
  const size_t N = v.N;
  float* kb[7] = {nullptr, c->adj + 0 * N, c->adj + 1 * N, c->adj + 2 * N, c->adj + 3 * N, c->adj + 4 * N, c->adj + 5 * N};
  float* ub = c->adj + 6 * N; float* g6b = c->adj + 7 * N; float* xs = c->adj + 8 * N; float* xb = c->adj + 9 * N;
  float* gtmp = c->adj + 10 * N;
  HIPCHK(c, hipMemsetAsync(c->adj, 0, sizeof(float) * 8 * N, c->stream));
  HIPCHK(c, hipMemsetAsync(gp, 0, sizeof(float) * P, c->stream));
  RegSeedArgs sa{};
  sa.n = n; sa.n_norm = n * (size_t)(sharded(c) ? c->nranks : 1); sa.uprev = uprev; sa.u = u; sa.g6 = g6;
  for (int j = 0; j < 7; ++j) sa.k[j] = kk[j];
  for (int j = 1; j < 7; ++j) sa.kb[j] = kb[j];
  sa.ub = ub; sa.g6b = g6b; sa.dt = dt; sa.abstol = abstol; sa.reltol = reltol; sa.reg_type = reg_type;
  sa.eest = eest; sa.num = stiff_num; sa.den = stiff_den;
  { int nb = (int)((n + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_reg_seed, dim3(nb), dim3(256), 0, c->stream, sa); HIPCHK(c, hipGetLastError()); }
  float A[21];
  for (int i = 0; i < 21; ++i) A[i] = (float)Tsit5::A[i];
  const float cs[6] = {(float)Tsit5::C[0], (float)Tsit5::C[1], (float)Tsit5::C[2], (float)Tsit5::C[3], 1.0f, 1.0f};
  // per stage, newest first: stage input (one launch), vector-Jacobian product, parameter-gradient GEMM ACCUMULATING
  // into gp, and one fused join (xbar += ubar / g6bar; kbar_j += dt a_sj xbar for every earlier stage): 4 launches
  // (the sweep is launch-bound: it was 7..9 small launches per stage)
  for (int sidx = 7; sidx >= 2; --sidx) {
    const int off = (sidx - 2) * (sidx - 1) / 2;
    const float* x;
    if (sidx == 7) x = u;
    else { if ((rc = vec_axpy(c, xs, uprev, dt, sidx - 1, kk, A + off, n))) return rc; x = xs; }
    if (sharded(c)) {  // every evaluation's cotangent is all-reduced over the ranks by itself, then added
      if ((rc = launch_vjp(c, x, nullptr, 0.f, 0.f, t + cs[sidx - 2] * dt, kb[sidx - 1], B, xb, gtmp))) return rc;
      const float* g1[2] = {gp, gtmp}; const float cc[2] = {1.0f, 1.0f};
      if ((rc = vec_axpy(c, gp, nullptr, 1.0f, 2, g1, cc, P))) return rc;
    } else {
      c->pg_accumulate = true;
      rc = launch_vjp(c, x, nullptr, 0.f, 0.f, t + cs[sidx - 2] * dt, kb[sidx - 1], B, xb, gp);
      c->pg_accumulate = false;
      if (rc) return rc;
    }
    SweepJoinArgs ja{};
    ja.n = n; ja.xb = xb; ja.extra = (sidx == 7) ? ub : ((sidx == 6) ? g6b : nullptr); ja.dt = dt; ja.nk = sidx - 2;
    for (int j = 1; j < 6; ++j) { ja.kb[j - 1] = (j < sidx - 1) ? kb[j] : nullptr; ja.c[j - 1] = (j < sidx - 1) ? A[off + j] : 0.f; }
    if (ja.nk > 0 || ja.extra) {
      int nb = (int)((n + 255) / 256); if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(k_sweep_join, dim3(nb), dim3(256), 0, c->stream, ja);
      HIPCHK(c, hipGetLastError());
    }
  }
  return LRNDE_OK;
}
extern "C" {

// node_forward that also keeps what the backward pass needs: the dense record of every accepted
// step (retry with a larger record if it overflows) and the solve's arguments.
}  // extern "C"
static int node_forward_record_impl(lrnde_ctx* c, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* o,
                                    int32_t mode, int32_t reg_type, float t1_or_rand, const float* user_sv, int nuser,
                                    float* u_end, float* reg_val_host, int32_t* nfe_host, lrnde_stats* st, float* t1_used_host) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!x || !o || !u_end || !reg_val_host || !nfe_host || !st) return fail(c, LRNDE_BADARG, "null pointer");
  const size_t n = (size_t)B * c->desc.state_dim;
  c->rec_valid = false;
  float t1 = t2;
  if (c->rec_n != n) {  // the regulariser's parameter gradient (written by the forward's overlapped sweep, or by the backward)
    if ((rc = side_quiesce(c))) return rc;
    if (c->rec_gr) HIPCHK(c, hipFree(c->rec_gr));
    c->rec_gr = nullptr; c->rec_n = 0;
    HIPCHK(c, hipMalloc(&c->rec_gr, sizeof(float) * lrnde_param_count(&c->desc)));
    c->rec_n = n;
  }
  for (int attempt = 0;; ++attempt) {
    if (c->dense_cap == 0 || c->dense_n != n) {
      if (c->dense) { hipFree(c->dense); hipFree(c->dense_t); hipFree(c->dense_dt); c->dense = nullptr; }
      if (c->dense_cap == 0) c->dense_cap = 64;
      if (hipMalloc(&c->dense, sizeof(float) * (size_t)c->dense_cap * REC_ARRAYS * n) != hipSuccess ||
          hipMalloc(&c->dense_t, sizeof(float) * c->dense_cap) != hipSuccess ||
          hipMalloc(&c->dense_dt, sizeof(float) * c->dense_cap) != hipSuccess)
        return fail(c, LRNDE_HIP_ERROR, "dense record allocation failed");
      c->dense_n = n;
    }
    c->dense_on = true;
    rc = node_forward_impl(c, x, B, t0, t2, o, mode, reg_type, t1_or_rand, user_sv, nuser, u_end, reg_val_host, nfe_host, st, &t1);
    c->dense_on = false;
    if (rc == LRNDE_CAPACITY && attempt < 8) { c->dense_cap *= 2; c->dense_n = 0; continue; }
    break;
  }
  if (rc) return rc;
  if (t1_used_host) *t1_used_host = t1;
  if (mode != LRNDE_MODE_NONE) {
    // the local step's operands (uprev = u(t1), u, k1..k7, g6) stay in the state workspace it ran in and its scalars are
    // kept here: the regulariser's reverse sweep starts from them without re-running the step or asking the device again
    c->rec_dt1 = c->loc_dt; c->rec_eest = c->loc_eest; c->rec_snum = c->loc_snum; c->rec_sden = c->loc_sden;
  }
  c->rec_valid = true; ++c->rec_gen; c->rec_B = B; c->rec_t0 = t0; c->rec_t2 = t2; c->rec_opts = *o; c->rec_mode = mode;
  c->rec_reg_type = reg_type; c->rec_t1 = t1; c->rec_naccept = st->naccept;
  return LRNDE_OK;
}
extern "C" {

int lrnde_node_forward_record(lrnde_ctx* c, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* o,
                              int32_t mode, int32_t reg_type, float t1_or_rand, float* u_end, float* reg_val_host,
                              int32_t* nfe_host, lrnde_stats* st, float* t1_used_host) {
  return node_forward_record_impl(c, x, B, t0, t2, o, mode, reg_type, t1_or_rand, nullptr, 0, u_end, reg_val_host, nfe_host, st,
                                  t1_used_host);
}

// the recorded forward with the layer's `saveat` kwarg: the solution the caller sees (sol.u / sol.t after
// _CorrectedDESolution) goes to u_series (device, cap_series x B x D) / t_series_host
int lrnde_node_forward_record_ts(lrnde_ctx* c, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* o,
                                 int32_t mode, int32_t reg_type, float t1_or_rand, const float* saveat_host, int32_t nsave,
                                 float* u_series, float* t_series_host, int32_t cap_series, int32_t* nseries_host,
                                 float* reg_val_host, int32_t* nfe_host, lrnde_stats* st, float* t1_used_host) {
  if (!c) return LRNDE_BADARG;
  if (nsave < 0 || !u_series || !t_series_host || !nseries_host) return fail(c, LRNDE_BADARG, "null pointer / negative count");
  const size_t n = (size_t)B * c->desc.state_dim;
  float* u_end = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMalloc(&u_end, sizeof(float) * n));
  float regv = 0.f; int nfe = 0;
  int rc = node_forward_record_impl(c, x, B, t0, t2, o, mode, reg_type, t1_or_rand, saveat_host, nsave, u_end, &regv, &nfe, st,
                                    t1_used_host);
  hipFree(u_end);
  if (rc) return rc;
  if (reg_val_host) *reg_val_host = regv;
  if (nfe_host) *nfe_host = nfe;
  const int ns = (int)c->series_idx.size();
  *nseries_host = ns;
  if (ns > cap_series) return fail(c, LRNDE_CAPACITY, "series buffer too small (%d > %d)", ns, cap_series);
  for (int i = 0; i < ns; ++i) {
    HIPCHK(c, hipMemcpyAsync(u_series + (size_t)i * n, c->usave + (size_t)c->series_idx[i] * n, sizeof(float) * n,
                             hipMemcpyDeviceToDevice, c->stream));
    t_series_host[i] = c->series_t[i];
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LRNDE_OK;
}

// backward of  loss = <du_end, sol.u[end]> + w_reg * reg_val  from the record of the last
// lrnde_node_forward_record: continuous adjoint (InterpolatingAdjoint restatement) + the regulariser's
// reverse sweep.  dx (B,D), dp (P): device.
}  // extern "C"
// du_end: the cotangent of sol.u[end] (nser == 0), or du_series: one cotangent per state of the caller's series
// (c->series_t, ascending): each enters the reversed solve as an impulse on lambda at its time
static int node_backward_recorded_impl(lrnde_ctx* c, int32_t B, const float* du_end, const float* du_series, int nser,
                                       float w_reg, float* dx, float* dp, lrnde_stats* st_bwd) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if ((!du_end && !du_series) || !dx || !dp || !st_bwd) return fail(c, LRNDE_BADARG, "null pointer");
  if (!c->rec_valid || c->rec_B != B) return fail(c, LRNDE_BADARG, "no forward record for this batch (call lrnde_node_forward_record first)");
  const lrnde_solve_opts* o = &c->rec_opts;
  const float t0 = c->rec_t0, t2 = c->rec_t2, t1 = c->rec_t1;
  const int mode = c->rec_mode, reg_type = c->rec_reg_type;
  const size_t n = (size_t)B * c->desc.state_dim;
  const size_t P = lrnde_param_count(&c->desc);
  const int nsteps = c->rec_naccept;
  // adjoint solve on z = [lambda; mu] in s = -t from -t2 to -t0, tstops at the saved times
  const size_t N = n + P;
  AdjVec v;
  if ((rc = adj_alloc(c, N, v))) return rc;
  v.n_lam = n; v.P = P;
  const bool adj_host = opt(OPT_ADJ_HOST) != 0;  // diagnostic: the round-1 host-controlled loop
  const bool dev_loop = vjp_uses_qtile(c, B) && !sharded(c) && !adj_host;
  const bool begin_in_solve = dev_loop && !du_series;  // the device loop's first launch sets z = [du_end; 0] itself
  if (!begin_in_solve) HIPCHK(c, hipMemsetAsync(v.z, 0, sizeof(float) * N, c->stream));
  std::vector<AdjImpulse> impulses;  // ascending in s = -t
  if (du_series) {
    if (nser != (int)c->series_t.size()) return fail(c, LRNDE_BADARG, "%d cotangents for a series of %zu states", nser, c->series_t.size());
    for (int i = nser - 1; i >= 0; --i) {
      const float tv = c->series_t[i];
      const float* du = du_series + (size_t)i * n;
      if (tv >= t2) {  // cotangents at the end time: lambda(s0)
        const float* gi[1] = {du}; const float one = 1.0f;
        if ((rc = vec_axpy(c, v.z, v.z, 1.0f, 1, gi, &one, n))) return rc;
      } else {
        impulses.push_back(AdjImpulse{-tv, du});
      }
    }
  } else if (!begin_in_solve) {
    HIPCHK(c, hipMemcpyAsync(v.z, du_end, sizeof(float) * n, hipMemcpyDeviceToDevice, c->stream));
  }
  std::vector<float> stops;
  if (mode != LRNDE_MODE_NONE || du_series)
    for (int i = (int)c->last_ts.size() - 1; i >= 0; --i) {
      const float tv = c->last_ts[i];
      if (tv > t0 && tv < t2) stops.push_back(-tv);
    }
  auto pending_sweep = [c, t1]() -> int {  // the forward left the regulariser's sweep to us (lrnde_ctx::sweep_pending)
    lrnde_ctx* sd = c->side;
    c->sweep_pending = false;
    c->side_busy = true;
    const int r = step_reg_sweep(sd, sd->state, c->sw_B, t1, c->rec_dt1, c->sw_abstol, c->sw_reltol, c->sw_reg_type, c->rec_eest,
                                 c->rec_snum, c->rec_sden, c->rec_gr);
    if (r) { c->err = sd->err; return r; }
    HIPCHK(c, hipEventRecord(c->ev_side_sweep, sd->stream));
    c->rec_gr_ready = true;
    return LRNDE_OK;
  };
  const bool want_sweep = mode != LRNDE_MODE_NONE && w_reg != 0.0f && c->sweep_pending && c->side != nullptr;
  if (want_sweep) {
    if (vjp_uses_qtile(c, B) && !sharded(c) && !adj_host) c->after_first_attempt = pending_sweep;
    else if ((rc = pending_sweep())) return rc;
  }
  if (dev_loop) {
    c->adj_init_src = begin_in_solve ? du_end : nullptr;
    rc = adj_solve_device(c, v, B, -t2, -t0, o->abstol, o->reltol, o->maxiters, o->exact_pow, stops, impulses, nsteps, st_bwd);
    c->adj_init_src = nullptr;
  } else {
    std::vector<float> dts(nsteps), dds(nsteps);
    HIPCHK(c, hipMemcpy(dts.data(), c->dense_t, sizeof(float) * nsteps, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(dds.data(), c->dense_dt, sizeof(float) * nsteps, hipMemcpyDeviceToHost));
    auto rhs = [&](const float* zs, float sg, float* K) { return adj_rhs(c, dts, dds, B, n, zs, sg, K); };
    auto rhs_fused = [&](const StageIn& sin, float sg, float* K) { return adj_rhs(c, dts, dds, B, n, nullptr, sg, K, &sin); };
    rc = vec_tsit5_solve(c, v, rhs, rhs_fused, vjp_uses_qtile(c, B), -t2, -t0, o->abstol, o->reltol, o->maxiters, o->exact_pow,
                         stops, impulses, st_bwd);
  }
  c->after_first_attempt = nullptr;
  if (rc) return fail(c, rc, "adjoint solve stopped with retcode %d", rc);
  auto adj_out = [&](const float* grad_reg) -> int {   // dx = lambda, dp = mu [+ w_reg * grad_reg]: one launch
    int nb = (int)((N + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_adj_out, dim3(nb), dim3(256), 0, c->stream, (const float*)v.z, n, P, dx, dp, grad_reg, w_reg);
    HIPCHK(c, hipGetLastError());
    return LRNDE_OK;
  };
  if (!(mode != LRNDE_MODE_NONE && w_reg != 0.0f)) {
    if ((rc = adj_out(nullptr))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  // regulariser: dp += w_reg * d reg_val / d p   (no gradient w.r.t. x: test/runtests.jl:129)
  if (mode != LRNDE_MODE_NONE && w_reg != 0.0f) {
    float* gr = c->rec_gr;
    if (c->rec_gr_ready) {
      // the forward already ran the sweep on the companion's stream (it depends on the forward alone): wait for it there
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_side_sweep, 0));
      rc = adj_out(gr);
    } else {
      // The forward's own local step is still in the state workspace (uprev = u(t1), u, k1..k7, g6: the recorded forward
      // keeps k2..k6 in memory, StepArgs::force_store_k), and its scalars are in the record: the reverse sweep starts from
      // them, nothing is re-run and nothing is read back.
      // (the sweep works in the adjoint vector's buffers: lambda and mu leave them first)
      if ((rc = adj_out(nullptr))) return rc;
      rc = step_reg_sweep(c, c->state, B, t1, c->rec_dt1, o->abstol, o->reltol, reg_type, c->rec_eest, c->rec_snum, c->rec_sden, gr);
      if (!rc) { const float* g1[2] = {dp, gr}; const float cc[2] = {1.0f, w_reg}; rc = vec_axpy(c, dp, nullptr, 1.0f, 2, g1, cc, P); }
    }
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (c->rec_gr_ready && mode != LRNDE_MODE_NONE && w_reg != 0.0f) c->side_busy = false;  // waited for above, then synchronised
  c->rec_gr_ready = false;
  c->rec_valid = false;  // the regulariser sweep reused the state workspace
  return LRNDE_OK;
}
extern "C" {

int lrnde_node_backward_recorded(lrnde_ctx* c, int32_t B, const float* du_end, float w_reg, float* dx, float* dp,
                                 lrnde_stats* st_bwd) {
  return node_backward_recorded_impl(c, B, du_end, nullptr, 0, w_reg, dx, dp, st_bwd);
}

int lrnde_node_backward_recorded_ts(lrnde_ctx* c, int32_t B, const float* du_series, int32_t nseries, float w_reg, float* dx,
                                    float* dp, lrnde_stats* st_bwd) {
  if (!c) return LRNDE_BADARG;
  if (!du_series || nseries <= 0) return fail(c, LRNDE_BADARG, "null pointer / empty series");
  return node_backward_recorded_impl(c, B, nullptr, du_series, nseries, w_reg, dx, dp, st_bwd);
}

// forward (with record) + backward in one call
int lrnde_node_backward(lrnde_ctx* c, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* o,
                        int32_t mode, int32_t reg_type, float t1_or_rand, const float* du_end, float w_reg,
                        float* dx, float* dp, lrnde_stats* st_fwd, lrnde_stats* st_bwd) {
  int rc = check_ready(c, B);
  if (rc) return rc;
  if (!x || !o || !du_end || !dx || !dp || !st_fwd || !st_bwd) return fail(c, LRNDE_BADARG, "null pointer");
  const size_t n = (size_t)B * c->desc.state_dim;
  float* u_end = nullptr;
  HIPCHK(c, hipMalloc(&u_end, sizeof(float) * n));
  float regv = 0.f; int nfe = 0;
  rc = lrnde_node_forward_record(c, x, B, t0, t2, o, mode, reg_type, t1_or_rand, u_end, &regv, &nfe, st_fwd, nullptr);
  hipFree(u_end);
  if (rc) return rc;
  return lrnde_node_backward_recorded(c, B, du_end, w_reg, dx, dp, st_bwd);
}

// ---- classifier head + loss of the MNIST experiment (experiments/src/construct.jl:199 Dense(D => K),
// experiments/src/utils.jl:88 logitcrossentropy = mean(-sum(y .* logsoftmax(logits)))): forward value
// and the cotangents the pullback of `ce` sends to sol.u[end] and to the classifier parameters.
}  // extern "C"
namespace {
#include "lrnde_cls.hpp"
#include "lrnde_cls_fused.hpp"
}  // namespace
extern "C" {

}  // extern "C"
namespace {
// the classifier head's launches and the read-back of its loss sum, enqueued on the handle's stream (no synchronisation)
int cls_enqueue(lrnde_ctx* c, const float* u, int32_t B, const float* pc, int32_t K, const int32_t* labels, float* logits, float* du,
                float* dpc) {
  if (!u || !pc || !labels || B <= 0 || K <= 0 || K > 16) return fail(c, LRNDE_BADARG, "bad argument (1 <= K <= 16)");
  HIPCHK(c, hipSetDevice(c->device));
  const int D = c->desc.state_dim;
  // cached workspace: dl (B x K), per-sample losses (B), {double loss sum, int bad-label flag}
  const size_t need = sizeof(float) * ((size_t)B * K + (size_t)B) + 64;
  if (c->cls_ws_bytes < need) {
    if (c->cls_ws) HIPCHK(c, hipFree(c->cls_ws));
    c->cls_ws = nullptr; c->cls_ws_bytes = 0;
    HIPCHK(c, hipMalloc(&c->cls_ws, need));
    c->cls_ws_bytes = need;
  }
  if (!c->cls_host) HIPCHK(c, hipHostMalloc(&c->cls_host, sizeof(ClsOut)));
  ClsOut* out = reinterpret_cast<ClsOut*>(c->cls_ws);
  float* dl = reinterpret_cast<float*>(reinterpret_cast<char*>(c->cls_ws) + 64);
  float* lb = dl + (size_t)B * K;
  // the loss is the mean over the GLOBAL batch: on a sharded handle every rank normalises by B * nranks, and the loss
  // and the classifier cotangent (sums over all samples) are all-reduced; du stays sharded like dx
  const int nr = sharded(c) ? c->nranks : 1;
  const float Bnorm = (float)B * (float)nr;
  HIPCHK(c, hipMemsetAsync(out, 0, sizeof(ClsOut), c->stream));
  const size_t wbytes = sizeof(float) * (size_t)K * (D + 1);
  const int wlds = wbytes <= 60 * 1024 ? 1 : 0;  // (the parameter block in LDS when it fits the default dynamic limit)
  if (wlds && K == 10) hipLaunchKernelGGL((k_cls_fwd_bwdx<true, 10>), dim3((B + 3) / 4), dim3(256), wbytes, c->stream, u, pc, labels, B, D, K, Bnorm, logits, dl, lb, du, out);
  else if (wlds) hipLaunchKernelGGL((k_cls_fwd_bwdx<true, 0>), dim3((B + 3) / 4), dim3(256), wbytes, c->stream, u, pc, labels, B, D, K, Bnorm, logits, dl, lb, du, out);
  else hipLaunchKernelGGL((k_cls_fwd_bwdx<false, 0>), dim3((B + 3) / 4), dim3(256), 0, c->stream, u, pc, labels, B, D, K, Bnorm, logits, dl, lb, du, out);
  {  // dW = dl^T u and db = dl^T 1 as ONE batch-reduction GEMM (the parameter-gradient tiles of the adjoint, first form,
     // H := K, no time column) + one extra workgroup that adds the per-sample losses in a fixed order
    PgradArgs g{};
    memset(&g, 0, sizeof(g));
    g.D = D; g.H = K; g.Hp = K; g.td = 0; g.B = B; g.t = 0.f; g.dpre = dl; g.y = u; g.gp = dpc;
    g.nt1c = (D + 2 + 15) / 16; g.ntile1 = ((K + 15) / 16) * g.nt1c; g.ntile2 = 0; g.nt2c = 1;
    const int nt = dpc ? g.ntile1 : 0;
    hipLaunchKernelGGL(k_cls_bwdw_loss, dim3(nt + 1), dim3(256), 0, c->stream, g, nt, (const float*)lb, B, out);
  }
  HIPCHK(c, hipGetLastError());
  if (sharded(c)) {
    int rc;
    if (dpc && (rc = comm_allreduce(c, dpc, dpc, (size_t)K * (D + 1), false))) return rc;
    if ((rc = comm_allreduce(c, &out->loss_sum, &out->loss_sum, 1, true))) return rc;
  }
  HIPCHK(c, hipMemcpyAsync(c->cls_host, out, sizeof(ClsOut), hipMemcpyDeviceToHost, c->stream));
  return LRNDE_OK;
}
// after the stream has been synchronised
int cls_finish(lrnde_ctx* c, int32_t B, int32_t K, float* loss_host) {
  const int nr = sharded(c) ? c->nranks : 1;
  const ClsOut* ho = reinterpret_cast<const ClsOut*>(c->cls_host);
  if (ho->bad_label) return fail(c, LRNDE_BADARG, "a label is outside [0, %d)", K);
  *loss_host = (float)(ho->loss_sum / ((double)B * (double)nr));
  return LRNDE_OK;
}
}  // namespace
extern "C" {

int lrnde_classifier_ce(lrnde_ctx* c, const float* u, int32_t B, const float* pc, int32_t K, const int32_t* labels,
                        float* loss_host, float* logits, float* du, float* dpc) {
  if (!c) return LRNDE_BADARG;
  if (!loss_host) return fail(c, LRNDE_BADARG, "null pointer");
  int rc = cls_enqueue(c, u, B, pc, K, labels, logits, du, dpc);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return cls_finish(c, B, K, loss_host);
}

int lrnde_node_forward_record_ce(lrnde_ctx* c, const float* x, int32_t B, float t0, float t2, const lrnde_solve_opts* o,
                                 int32_t mode, int32_t reg_type, float t1_or_rand, float* u_end, float* reg_val_host,
                                 int32_t* nfe_host, lrnde_stats* st, float* t1_used_host, const float* pc, int32_t K,
                                 const int32_t* labels, float* loss_host, float* logits, float* du, float* dpc) {
  if (!c) return LRNDE_BADARG;
  if (!loss_host || !u_end) return fail(c, LRNDE_BADARG, "null pointer");
  // the head's launches go into the queue as soon as the solve's last report is in, ahead of its final synchronisation
  c->final_hook = [&]() { return cls_enqueue(c, u_end, B, pc, K, labels, logits, du, dpc); };
  int rc = node_forward_record_impl(c, x, B, t0, t2, o, mode, reg_type, t1_or_rand, nullptr, 0, u_end, reg_val_host, nfe_host, st,
                                    t1_used_host);
  c->final_hook = nullptr;
  if (rc) return rc;
  if (!(c->final_hook_fired && c->last_u_end_done)) {  // (sol.u[end] reached the caller's array by a copy after the solve)
    if ((rc = cls_enqueue(c, u_end, B, pc, K, labels, logits, du, dpc))) return rc;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));   // (returns at once when the solve's own synchronisation covered the head)
  return cls_finish(c, B, K, loss_host);
}

// ---- optimiser update rules of the experiments (experiments/src/construct.jl:104-126) ----
}  // extern "C"
namespace {
// one fused elementwise pass: gradient -> update direction (Optimisers.jl rules, UPSTREAM-RECALL) -> optional
// WeightDecay (OptimiserChain(rule, WeightDecay(gamma)): gamma * x added to the direction) -> x -= direction
__global__ void k_opt_update(size_t n, int kind, float* x, const float* g, float* s1, float* s2, float eta, float rho_b1, float b2,
                             float eps, float b1t, float b2t, float wd) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float dx = g[i];
    float d;
    if (kind == LRNDE_OPT_DESCENT) {
      d = eta * dx;
    } else if (kind == LRNDE_OPT_MOMENTUM) {          // v = rho v - eta dx ; dx' = -v
      const float v = rho_b1 * s1[i] - eta * dx;
      s1[i] = v; d = -v;
    } else if (kind == LRNDE_OPT_NESTEROV) {          // dx' = -rho^2 v + (1 + rho) eta dx ; v = rho v - eta dx
      const float v0 = s1[i];
      d = -(rho_b1 * rho_b1) * v0 + (1.0f + rho_b1) * eta * dx;
      s1[i] = rho_b1 * v0 - eta * dx;
    } else if (kind == LRNDE_OPT_ADAM) {              // mt, vt moments; dx' = mt/(1-b1^t) / (sqrt(vt/(1-b2^t)) + eps) * eta
      const float mt = rho_b1 * s1[i] + (1.0f - rho_b1) * dx;
      const float vt = b2 * s2[i] + (1.0f - b2) * (dx * dx);
      s1[i] = mt; s2[i] = vt;
      d = mt / (1.0f - b1t) / (sqrtf(vt / (1.0f - b2t)) + eps) * eta;
    } else {                                           // AdaMax: ut = max(b2 ut, |dx|); dx' = eta/(1-b1^t) * mt / (ut + eps)
      const float mt = rho_b1 * s1[i] + (1.0f - rho_b1) * dx;
      const float ut = fmaxf(b2 * s2[i], fabsf(dx));
      s1[i] = mt; s2[i] = ut;
      d = (eta / (1.0f - b1t)) * mt / (ut + eps);
    }
    if (wd != 0.0f) d = d + wd * x[i];
    x[i] = x[i] - d;
  }
}
}  // namespace
extern "C" {

int lrnde_opt_update(int32_t kind, float* x, const float* grad, float* state1, float* state2, size_t n, float eta, float rho_or_beta1,
                     float beta2, float eps, int32_t step, float weight_decay, int device, void* stream) {
  if (!x || !grad || kind < LRNDE_OPT_DESCENT || kind > LRNDE_OPT_ADAMAX || step < 1) return LRNDE_BADARG;
  if (kind != LRNDE_OPT_DESCENT && !state1) return LRNDE_BADARG;
  if (kind >= LRNDE_OPT_ADAM && !state2) return LRNDE_BADARG;
  if (hipSetDevice(device) != hipSuccess) return LRNDE_HIP_ERROR;
  if (n == 0) return LRNDE_OK;
  const float b1t = powf(rho_or_beta1, (float)step), b2t = powf(beta2, (float)step);
  int nb = (int)((n + 255) / 256); if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_opt_update, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, (int)kind, x, grad, state1, state2, eta,
                     rho_or_beta1, beta2, eps, b1t, b2t, weight_decay);
  return hipGetLastError() == hipSuccess ? LRNDE_OK : LRNDE_HIP_ERROR;
}

int lrnde_set_adjoint_trace(lrnde_ctx* c, lrnde_trace_row* rows_host, int32_t cap) {
  if (!c || cap < 0) return LRNDE_BADARG;
  c->adj_trace = cap > 0 ? rows_host : nullptr; c->adj_trace_cap = cap; c->adj_trace_n = 0;
  return LRNDE_OK;
}

int lrnde_adjoint_trace_rows(lrnde_ctx* c, int32_t* n_host) {
  if (!c || !n_host) return LRNDE_BADARG;
  *n_host = c->adj_trace_n;
  return LRNDE_OK;
}

// which recorded forward the handle's record belongs to: it counts the successful lrnde_node_forward_record* calls.  A
// binding whose pullback closure captured generation g checks it before lrnde_node_backward_recorded*: a later forward of
// the same layer (an evaluation pass, a second pullback in flight) has replaced the record, and the backward would
// return the gradients of the OTHER input without an error (the C side can only see that some record is valid).
int lrnde_record_generation(lrnde_ctx* c, uint64_t* gen_host) {
  if (!c || !gen_host) return LRNDE_BADARG;
  *gen_host = c->rec_valid ? c->rec_gen : 0;   // 0: no usable record
  return LRNDE_OK;
}
int lrnde_sde_record_generation(lrnde_sde* s, uint64_t* gen_host) {
  if (!s || !gen_host) return LRNDE_BADARG;
  *gen_host = sde_node_generation(s);
  return LRNDE_OK;
}

int lrnde_set_option(const char* name, int32_t value) {
  if (!name) return LRNDE_BADARG;
  opt(0);
  for (int i = 0; i < N_OPT; ++i)
    if (strcmp(name, g_optdef[i].name) == 0) { g_opt[i] = value; g_opt_set[i] = true; return LRNDE_OK; }
  return LRNDE_BADARG;
}

int lrnde_set_reports(lrnde_ctx* c, int32_t on) {
  if (!c) return LRNDE_BADARG;
  c->reports_off = (on == 0);
  return LRNDE_OK;
}

int lrnde_host_phases(lrnde_ctx* c, double* us, int32_t reset) {
  if (!c || !us) return LRNDE_BADARG;
  for (int i = 0; i < 4; ++i) us[i] = c->hp_n ? c->hp_sum[i + 1] / (double)c->hp_n : 0.0;
  if (reset) { for (double& v : c->hp_sum) v = 0; c->hp_n = 0; }
  return LRNDE_OK;
}

int lrnde_last_solve_kernel_ms(lrnde_ctx* c, float* ms, int32_t* launches) {
  if (!c) return LRNDE_BADARG;
  c->time_solves = true;
  if (ms) *ms = c->last_ms;
  if (launches) *launches = c->last_launches;
  return LRNDE_OK;
}

}  // extern "C"
