// lrnde_adjoint.hpp — device-side controller of the continuous adjoint's reversed-time Tsit5 solve (SURVEY.md §3.3:
// SciMLSensitivity InterpolatingAdjoint, un-vendored; what the reference reaches through Zygote.pullback in
// experiments/src/utils.jl:104-115).  Included by lrnde_kernels.hip inside its anonymous namespace, before
// lrnde_backward.hpp (whose kernels resolve their per-attempt arguments through it).
//
// Round 1 kept this loop's controller on the host: one norm read-back per adjoint step (a stream sync and ~35 us of
// idle GPU per 145-us step).  Here it is the forward solve's scheme (step_prologue): the integrator state AdjCtrl lives
// on the device, double-buffered by attempt parity; the FIRST launch of attempt j (the stage-2 VJP) runs, in wave 0 of
// every workgroup and on identical inputs, the footer of attempt j-1 (error norm from the per-block fp64 partial sums,
// PI controller, accept/reject, FSAL swap, tstop handling) and the header of attempt j (dt, the six stage times and the
// dense-record step / interpolation parameter each of them falls into); block 0 publishes the result, the other
// launches of the attempt (stages 3..7, the parameter-gradient GEMMs, the end-of-step norm kernel) just read it.  The
// host enqueues one attempt ahead and reads the integrator's progress (and, at the end, its final state) from a block in
// pinned host memory that the prologue writes; nothing waits for the host.
//
// z = [lambda (n_lam local columns); mu (P, replicated)], N = n_lam + P, reversed time s = -t.
// Buffers (one allocation, c->adj): zb[2] (z / z_new ping-pong), zs, ut (stage lambdas), K[0..6] with K[0], K[6] the
// FSAL pair that swaps with z on an accepted step.

struct AdjStage { int lo; float theta, ddt, t; };  // dense-record step, interpolation parameter, that step's dt, forward time

struct AdjCtrl {
  int status, first, do_step;
  int resume;  // 1: first attempt of a later segment (after a cotangent impulse): no footer, dt = dtpropose
  int iter, naccept, nreject, nf;
  int cur;     // z = zb[cur], z_new = zb[cur^1]; K1 = (cur ? K[6] : K[0]), K7 = (cur ? K[0] : K[6])
  int istop;
  float t, dt;  // reversed time and dt of the attempt this block describes
  float tstop;  // end of that attempt's tstop interval (the accept snap needs it)
  float qold, q11, dtpropose, eest_last, dt_init;
  float dt0;    // initdt's first guess (init phase B)
  AdjStage st[6];  // stages 2..7 (init phases use st[0])
};

struct AdjArgs {
  AdjCtrl* ctl;  // [2]
  float* base;   // the 11-vector allocation: zb0 zb1 zs ut K0..K6, each N floats
  size_t N, n_lam, P;
  const float* dense; const float* dense_t; const float* dense_dt; int nrec;  // forward record: [uprev,k1,P2,P3,P4] per accepted step (lrnde_math.hpp)
  const float* stops; int nstops;  // tstops in reversed time (device, ascending)
  float s0, s1;        // start of the whole solve; end of the CURRENT segment (the next cotangent impulse, or the end)
  float dtmax, dtmin;  // of the whole solve (s_end - s0; eps)
  float abstol, reltol;
  int maxiters, exact_pow;
  const double* part;   // [512 + 64]: per-block sums of the error norm (lambda 256, mu 256) and the per-rank lambda sums
  const double* ipart;  // [3][512 + 64]: the same for initdt's d0, d1, d2
  int nranks;           // > 1 (or a forced communicator): the lambda part's sum is the rank slots', added in rank order
  int use_slots;
  // pinned host memory (device-mapped), written by block 0 of every attempt's first launch: {seq, status, t, dt} — the
  // host driver reads the integrator's progress from it without putting copy packets between the kernels
  int* hstat;
  int seq0;             // seq written by attempt j is seq0 + j + 1
  // mu part of the ATTEMPT's error norm: 0 = the 256 per-block sums at part[256..512) (k_adj_err_dev / k_adj_err);
  // > 0 = one partial per tile of the last parameter-gradient GEMM at part[ADJ_MU_TILE_OFF ..) (k_pgrad_adj_err, which
  // then also forms the mu part of z_new: no launch of its own for the end of an adjoint step)
  int mu_tiles;
  // OVERLAPPED stage launches (LRNDE_ADJ_OVERLAP, adj_solve_device): device words shared by the stage launches of one solve —
  // sync[0] = attempt whose control block is published (seq numbering of hstat), sync[1] = a wait timed out,
  // sync[8 + (id & 7)] = VJP workgroups of stage launch `id` that have stored their outputs.  NULL: launches run one after
  // the other and none of this is touched.
  int* sync;
};
constexpr int ADJ_MU_TILE_OFF = 576;   // behind [256 lambda][256 mu blocks][64 rank slots]
constexpr int ADJ_MU_TILE_MAX = 2048;

// how a backward kernel gets its per-launch arguments: from the host (round-1 path, single calls), or from AdjCtrl:
// ADJ_FSAL = K1 := rhs(z, t) at the state of ctl[0] (init phase A; re-evaluation after an impulse), ADJ_INIT_B = initdt's
// second evaluation at z + dt0*K1, ADJ_STAGE = stage sidx of attempt j
enum { ADJ_HOST = 0, ADJ_FSAL = 1, ADJ_INIT_B = 2, ADJ_STAGE = 3 };

__device__ __forceinline__ float* adj_zb(const AdjArgs& g, int i) { return g.base + (size_t)i * g.N; }
__device__ __forceinline__ float* adj_zs(const AdjArgs& g) { return g.base + 2 * g.N; }
__device__ __forceinline__ float* adj_ut(const AdjArgs& g) { return g.base + 3 * g.N; }
// K_j of the attempt (j = 0..6) under FSAL parity cur
__device__ __forceinline__ float* adj_K(const AdjArgs& g, int j, int cur) {
  const int slot = (j == 0) ? (cur ? 6 : 0) : ((j == 6) ? (cur ? 0 : 6) : j);
  return g.base + (size_t)(4 + slot) * g.N;
}
// lambda part of the stage state of stage sidx (2..7): the buffers the host loop of round 1 alternated between
// (overlapped launches: THREE stage buffers in rotation — the tiles of launch s still read lambda_{s-1} while launch s+1
//  forms lambda_{s+1}; the third is the 12th vector of the allocation)
__device__ __forceinline__ float* adj_stage_lam(const AdjArgs& g, int sidx, int cur) {
  if (sidx == 7) return adj_zb(g, cur ^ 1);
  if (g.sync) { const int r = sidx % 3; return r == 0 ? g.base + 11 * g.N : (r == 1 ? adj_ut(g) : adj_zs(g)); }
  return (sidx & 1) ? adj_ut(g) : adj_zs(g);
}

// ---- overlapped launches: L2-bypassing accesses (agent scope: the per-XCD L2s are not coherent with each other inside a
// kernel) and the bounded wait.  Pattern measured in tools/xchg_probe.hip: relaxed agent-scope stores / loads, no fences.
__device__ __forceinline__ float ldcc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ldcc(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte forms: one sc1 instruction per lane (four scalar agent-scope accesses cost four address-path slots)
__device__ __forceinline__ f32x4 ld4cc(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void stwt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stwt(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4wt(float* p, const f32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
// wait until *w - want >= 0 (30 ms at most: a wait that long is a lost launch, not a slow one; sync[1] tells the host)
__device__ __forceinline__ bool adj_spin(const int* w, int want, int* err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  for (;;) {
    if (ldcc(w) - want >= 0) return true;
    __builtin_amdgcn_s_sleep(4);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 3000000ull) { stwt(err, 1); return false; }
  }
}
// all of this workgroup's stores have been acknowledged -> one arrival on the launch's counter (every wave, then thread 0)
__device__ __forceinline__ void adj_arrive(int* sync, int id) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(sync + 8 + (id & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum of the norm's partial sums (wave 0, all lanes return the total): lambda part = the 256 block sums, or, on a sharded
// handle, the per-rank sums in rank order; then the 256 mu block sums.  Fixed order: lane-strided, DPP/readlane tree.
__device__ __forceinline__ double adj_norm_sum(const double* p, int use_slots, int nranks, bool has_mu, int mu_tiles = 0) {
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  if (use_slots) {
    if (lane < nranks) s = __hip_atomic_load(p + 512 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __hip_atomic_load(p + lane + 64 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = ((v[0] + v[1]) + v[2]) + v[3];
  }
  if (has_mu && mu_tiles > 0) {   // per-tile partials of the last GEMM launch: lane-strided, in tile order per lane
    double m = 0.0;
    for (int i = lane; i < mu_tiles; i += 64) m += __hip_atomic_load(p + ADJ_MU_TILE_OFF + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s += m;
  } else if (has_mu) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __hip_atomic_load(p + 256 + lane + 64 * u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s += ((v[0] + v[1]) + v[2]) + v[3];
  }
  return wave_sum_dpp(s);
}

// the forward step the time tt falls into: the largest lo with dense_t[lo] <= tt (0 if none), by ballots over the
// (ascending) start times; wave 0, uniform result
__device__ __forceinline__ AdjStage adj_lookup(const AdjArgs& g, float tt) {
  const int lane = threadIdx.x & 63;
  int cnt = 0;
  for (int base = 0; base < g.nrec; base += 64) {
    const int i = base + lane;
    const float v = (i < g.nrec) ? g.dense_t[i] : 3.0e38f;
    cnt += __popcll(__ballot(v <= tt));
  }
  AdjStage s;
  s.lo = cnt > 0 ? cnt - 1 : 0;
  s.ddt = g.dense_dt[s.lo];
  s.theta = (tt - g.dense_t[s.lo]) / s.ddt;
  s.t = tt;
  return s;
}

// The same for several times at once without a memory round trip per time: lane i holds (dense_t[i], dense_dt[i]) of
// the record (nrec <= 64: one load each, issued with the prologue's other loads), a lookup is a ballot and two readlanes.
struct AdjRecLanes { float t, dt; };
__device__ __forceinline__ AdjRecLanes adj_rec_load(const AdjArgs& g) {
  const int lane = threadIdx.x & 63;
  AdjRecLanes r;
  r.t = (lane < g.nrec) ? g.dense_t[lane] : 3.0e38f;
  r.dt = (lane < g.nrec) ? g.dense_dt[lane] : 1.0f;
  return r;
}
__device__ __forceinline__ AdjStage adj_lookup_lanes(const AdjArgs& g, const AdjRecLanes& r, float tt) {
  if (g.nrec > 64) return adj_lookup(g, tt);
  const int cnt = __popcll(__ballot(r.t <= tt));
  AdjStage s;
  s.lo = cnt > 0 ? cnt - 1 : 0;
  const float t_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.t), s.lo));
  s.ddt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.dt), s.lo));
  s.theta = (tt - t_lo) / s.ddt;
  s.t = tt;
  return s;
}

__device__ __forceinline__ float adj_dt0(float d0, float d1, float dtmax) {
  float dt0 = ((double)d0 < 1e-5 || (double)d1 < 1e-5) ? 1e-6f : (d0 / d1) / 100.0f;
  return fminf_(dt0, dtmax);
}

// the pinned progress block (host view: volatile int hs[16]): [0] seq (written last, release), [1] status, [2] t, [3] dt,
// [4] cur, [5] nf, [6] naccept, [7] nreject, [8] iter, [9] eest_last, [10] dt_init — everything the host driver needs of
// the integrator's state, so that a finished solve costs no read-back copy and no synchronisation of its own
__device__ __forceinline__ void adj_hstat_fill(int* hs, const AdjCtrl& c) {
  hs[1] = c.status; hs[2] = __builtin_bit_cast(int, c.t); hs[3] = __builtin_bit_cast(int, c.dt);
  hs[4] = c.cur; hs[5] = c.nf; hs[6] = c.naccept; hs[7] = c.nreject; hs[8] = c.iter;
  hs[9] = __builtin_bit_cast(int, c.eest_last); hs[10] = __builtin_bit_cast(int, c.dt_init);
}

// overlapped launches: the control block of attempt j is read by the stage-3 launch WHILE this (stage-2) launch runs, from
// other XCDs — field by field through the L2 (agent-scope stores), then the attempt's number in sync[0]
__device__ __forceinline__ void adj_publish(const AdjArgs& g, AdjCtrl* cout, const AdjCtrl& c, int j) {
  int* d = reinterpret_cast<int*>(cout);
#define LRNDE_PUB(f) stwt(d + offsetof(AdjCtrl, f) / 4, __builtin_bit_cast(int, c.f))
  LRNDE_PUB(status); LRNDE_PUB(first); LRNDE_PUB(do_step); LRNDE_PUB(resume); LRNDE_PUB(iter); LRNDE_PUB(naccept); LRNDE_PUB(nreject);
  LRNDE_PUB(nf); LRNDE_PUB(cur); LRNDE_PUB(istop); LRNDE_PUB(t); LRNDE_PUB(dt); LRNDE_PUB(tstop); LRNDE_PUB(qold); LRNDE_PUB(q11);
  LRNDE_PUB(dtpropose); LRNDE_PUB(eest_last); LRNDE_PUB(dt_init); LRNDE_PUB(dt0);
#undef LRNDE_PUB
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    int* sp = d + (offsetof(AdjCtrl, st) + q * sizeof(AdjStage)) / 4;
    stwt(sp + 0, c.st[q].lo); stwt(sp + 1, __builtin_bit_cast(int, c.st[q].theta)); stwt(sp + 2, __builtin_bit_cast(int, c.st[q].ddt));
    stwt(sp + 3, __builtin_bit_cast(int, c.st[q].t));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stwt(g.sync, g.seq0 + j + 1);
}

// footer of attempt j-1 + header of attempt j (wave 0 of every workgroup; identical inputs => identical results).
// Returns the control block of attempt j; block 0 also publishes it to ctl[(j+1)&1].
__device__ __forceinline__ AdjCtrl adj_prologue(const AdjArgs& g, int j) {
  const int lane = threadIdx.x & 63;
  AdjCtrl c = g.ctl[j & 1];
  const AdjRecLanes rec = adj_rec_load(g);  // (independent of the control block: in flight with it)
  // so are the error norm's partial sums and the tstops: loaded (and summed) before the control block says whether they
  // are needed — behind its branches each was one more memory round trip of the launch every other launch waits for
  const float stop_l = (lane < g.nstops) ? g.stops[lane] : 3.0e38f;
  const double esum = adj_norm_sum(g.part, g.use_slots, g.nranks, g.P != 0, g.mu_tiles);
  AdjCtrl* cout = g.ctl + ((j + 1) & 1);
  c.do_step = 0;
  if (c.status != ST_RUNNING) {
    if (blockIdx.x == 0 && lane == 0) {
      if (g.sync) adj_publish(g, cout, c, j);
      else *cout = c;
      if (g.hstat) {
        adj_hstat_fill(g.hstat, c);
        if (g.sync) g.hstat[11] = ldcc(g.sync + 1);   // a wait of an overlapped launch timed out: the results are not to be used
        __hip_atomic_store(g.hstat, g.seq0 + j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return c;
  }
  const float gamma = 0.9f, qmin = 0.2f, qmax = 10.0f, qoldinit = 1e-4f;
  const float beta1 = (float)(7.0 / 50.0), beta2 = (float)(2.0 / 25.0);
  const float dtmax = g.dtmax, dtmin = g.dtmin;
  const double ntot = (double)g.n_lam * (double)(g.use_slots ? g.nranks : 1) + (double)g.P;
  float t = c.t, dt = c.dt;
  int accepted = 0;
  if (c.first) {
    // ode_determine_initdt from the partial sums of d1 and d2; dt0 was fixed by init phase B
    const float d1 = (float)sqrt(adj_norm_sum(g.ipart + 576, g.use_slots, g.nranks, g.P != 0) / ntot);
    const float d2 = (float)sqrt(adj_norm_sum(g.ipart + 2 * 576, g.use_slots, g.nranks, g.P != 0) / ntot) / c.dt0;
    const float maxd = fmaxf_(d1, d2);
    float dt1;
    if ((double)maxd <= 1e-15) dt1 = fmaxf_(1e-6f, c.dt0 * 1e-3f);
    else {
      const float l10 = (float)log10((double)maxd);
      const float e = (-(2.0f + l10)) / 5.0f;
      dt1 = (float)pow(10.0, (double)e);
    }
    dt = fminf_(fminf_(100.0f * c.dt0, dt1), dtmax);
    c.nf = 3; c.dt_init = dt; c.dtpropose = dt;
    c.qold = qoldinit; c.q11 = 1.0f;
  } else if (c.resume) {
    dt = c.dtpropose;
    c.resume = 0;
  } else {
    const float eest = (float)sqrt(esum / ntot);
    c.eest_last = eest;
    if (eest != eest) {
      c.status = LRNDE_DT_NAN;
    } else {
      float q;
      if (eest == 0.0f) q = 1.0f / qmax;
      else {
        if (g.exact_pow) { c.q11 = (float)pow((double)eest, (double)beta1); q = c.q11 / (float)pow((double)c.qold, (double)beta2); }
        else { c.q11 = fastpow(eest, beta1); q = c.q11 / fastpow(c.qold, beta2); }
        q = fmaxf_(1.0f / qmax, fminf_(1.0f / qmin, q / gamma));
      }
      accepted = (eest <= 1.0f);
      if (accepted) {
        c.naccept++;
        const float dtnew = c.dt / q;
        c.qold = fmaxf_(eest, qoldinit);
        const float ttmp = c.t + c.dt;
        // (magnitudes: see vec_tsit5_solve)
        t = (__builtin_fabsf(ttmp - c.tstop) < 100.0f * eps_f(fmaxf_(__builtin_fabsf(c.t), __builtin_fabsf(c.tstop)))) ? c.tstop : ttmp;
        c.dtpropose = fmaxf_(fminf_(dtmax, dtnew), fmaxf_(eps_f(t), dtmin));
        c.cur ^= 1;  // z <- z_new, K1 <- K7 (FSAL)
        dt = c.dtpropose;
      } else {
        c.nreject++;
        dt = c.dt / fminf_(1.0f / qmin, c.q11 / gamma);
      }
    }
  }
  if (c.status == ST_RUNNING) {
    if (!(t < g.s1)) {
      c.status = ST_DONE;
    } else {
      float tstop = g.s1;
      if (g.nstops <= 64) {   // (ascending: the entries from istop on that t has reached are a prefix)
        c.istop += __popcll(__ballot(lane >= c.istop && lane < g.nstops && stop_l <= t));
        if (c.istop < g.nstops) {
          const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, stop_l), c.istop));
          if (sv < g.s1) tstop = sv;
        }
      } else {
        while (c.istop < g.nstops && g.stops[c.istop] <= t) ++c.istop;
        if (c.istop < g.nstops && g.stops[c.istop] < g.s1) tstop = g.stops[c.istop];
      }
      c.iter++;
      dt = fminf_(dtmax, dt);
      dt = fmaxf_(dt, dtmin);
      dt = fminf_(__builtin_fabsf(dt), __builtin_fabsf(tstop - t));
      if (c.iter > g.maxiters) c.status = LRNDE_MAXITERS;
      else if (dt != dt) c.status = LRNDE_DT_NAN;
      else if (__builtin_fabsf(dt) <= __builtin_fabsf(dtmin)) c.status = LRNDE_DT_LESS_THAN_MIN;
      else {
        c.do_step = 1; c.nf += 6; c.tstop = tstop;
        const float cs[6] = {(float)Tsit5::C[0], (float)Tsit5::C[1], (float)Tsit5::C[2], (float)Tsit5::C[3], 1.0f, 1.0f};
#pragma unroll
        for (int q = 0; q < 6; ++q) c.st[q] = adj_lookup_lanes(g, rec, -(t + cs[q] * dt));
      }
    }
  }
  c.t = t; c.dt = dt; c.first = 0;
  if (blockIdx.x == 0 && lane == 0) {
    if (g.sync) adj_publish(g, cout, c, j);
    else *cout = c;
    if (g.hstat) {
      adj_hstat_fill(g.hstat, c);
      if (g.sync) g.hstat[11] = ldcc(g.sync + 1);
      __hip_atomic_store(g.hstat, g.seq0 + j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  return c;
}

__global__ void k_adj_ctrl_init(AdjCtrl* ctl, float s0) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  AdjCtrl c;
  memset(&c, 0, sizeof(c));
  c.status = ST_RUNNING; c.first = 1; c.t = s0; c.qold = 1e-4f; c.q11 = 1.0f;
  ctl[0] = c; ctl[1] = c;
}

// The start of a backward pass in ONE launch: z = [du_end; 0], the control blocks, and the (few) tstops — round 2 enqueued a
// memset, a D2D copy, a H2D copy and k_adj_ctrl_init for these, four packets with the queue's idle gaps between them.
struct AdjBegin { float* z; const float* src; size_t n, N; AdjCtrl* ctl; float s0; float* stops; int nstops; float sv[8]; };
__global__ __launch_bounds__(256) void k_adj_begin(AdjBegin a) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.N; i += (size_t)gridDim.x * blockDim.x)
    a.z[i] = i < a.n ? a.src[i] : 0.f;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  AdjCtrl c;
  memset(&c, 0, sizeof(c));
  c.status = ST_RUNNING; c.first = 1; c.t = a.s0; c.qold = 1e-4f; c.q11 = 1.0f;
  a.ctl[0] = c; a.ctl[1] = c;
  for (int k = 0; k < a.nstops; ++k) a.stops[k] = a.sv[k];
}
// ... and its end: dx = lambda, dp = mu (+ w_reg * the regulariser's gradient, in k_axpy's operation order) in one launch
// instead of two copy packets and an axpy
__global__ __launch_bounds__(256) void k_adj_out(const float* z, size_t n, size_t P, float* dx, float* dp, const float* gr, float w_reg) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n + P; i += (size_t)gridDim.x * blockDim.x) {
    if (i < n) { dx[i] = z[i]; continue; }
    const size_t j = i - n;
    if (gr) {
      float s = 1.0f * z[i];
      s = s + w_reg * gr[j];
      dp[j] = 0.f + 1.0f * s;
    } else {
      dp[j] = z[i];
    }
  }
}

// a later segment of the same solve (after a cotangent impulse at a saved time): the integrator goes on with its
// proposed dt and controller memory, K1 has been re-evaluated by the host driver at the modified state
__global__ void k_adj_ctrl_continue(AdjCtrl* ctl, int from) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  AdjCtrl c = ctl[from];
  c.status = ST_RUNNING; c.first = 0; c.do_step = 0; c.resume = 1;
  ctl[0] = c; ctl[1] = c;
}
