// lrnde_backward.hpp — device kernels of the backward pass, included by
// lrnde_kernels.hip inside its anonymous namespace.
//
//   k_vjp      per 16-column batch tile: y (given, or the Tsit5 interpolant of the stored forward
//              step — SciMLSensitivity InterpolatingAdjoint), h = act(W1[y;t]+b1), then the
//              vector-Jacobian product  dy = W1^T ((W2^T lam) .* act'(pre))  with the two transposed
//              weight matrices pre-packed in the same MFMA fragment layouts as the forward ones.
//              Also leaves y, h, dpre in scratch for the parameter gradient.
//   k_pgrad    (df/dp)^T lam: the batch-reduction GEMMs  gW1 = dpre^T [y;t], gW2 = lam^T [h;t]
//              (K = batch, output-tiled: no cross-workgroup reduction) and the bias column sums.
//   elementwise helpers for the augmented state z = [lambda (B*D); mu (P)].
// Numerics: fp32 with fixed (deterministic) summation orders; parity with the oracle is by
// tolerance (the oracle sums in a different order), see tests/test_gpu_backward.py.

__device__ __forceinline__ float act_deriv_c(int act, float pre, float h) {
  if (act == 1) return 1.0f - h * h;
  if (act == 2) {
    const float two_lambda = 1.5957691216057308f;
    const float x2 = pre * pre;
    const float a = (two_lambda * pre) * fma_(x2, 0.044715f, 1.0f);
    const float sg = 1.0f / (1.0f + expf_c(-a));
    const float da = two_lambda * fma_(x2, 3.0f * 0.044715f, 1.0f);
    return sg + pre * sg * (1.0f - sg) * da;
  }
  return 1.0f;
}

struct VjpArgs {
  ModelDev m;
  const f32x4* V1p;  // W2^T in the Dense-1 layout  [MT1p][KG1][64][4]
  const f32x4* U2p;  // W1^T in the Dense-2 layout  [MT2][KG2p][64][4]
  int B;
  float t;
  const float* y;      // (B,D) or NULL -> interpolate from the dense record
  const float* dense;  // [uprev, k1, P2, P3, P4] of one forward step (polynomial form, lrnde_math.hpp), REC_ARRAYS arrays of B*D
  float theta, dense_dt;
  const float* lam;    // (B,D)
  float* dy;           // (B,D)
  float* ysc;          // scratch (B,D): y
  float* hsc;          // scratch (B,Hp): h
  float* dpsc;         // scratch (B,Hp): dpre
};

// Dense-1-shaped GEMM phase: K split by 112-row segments over waves, partial sums into pl
__device__ __forceinline__ void gemm_ksplit(const ModelDev& m, __amdgpu_buffer_rsrc_t rs, const float* xl, float* pl) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int voff = lane * 16;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int nseg1 = (m.KG1 + SEGK - 1) / SEGK;
  const f32x4* xp = reinterpret_cast<const f32x4*>(xl) + lane;
  for (int seg = wave; seg < nseg1; seg += NW) {
    const int kg_lo = seg * SEGK, kg_hi = min(m.KG1, kg_lo + SEGK);
    for (int mt0 = 0; mt0 < m.MT1; mt0 += TG) {
      const int tbase = mt0 * m.KG1;
      f32x4 acc[TG], aX[TG], aY[TG], bX, bY;
#pragma unroll
      for (int i = 0; i < TG; ++i) acc[i] = zero4;
#define LRNDE_BL(a, b, kg)                                                                 \
  do {                                                                                     \
    b = xp[(kg) * 64];                                                                     \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) a[i] = wload(rs, voff, (tbase + i * m.KG1 + (kg)) * 1024); \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
#define LRNDE_BM(a, b)                                                                     \
  do {                                                                                     \
    _Pragma("unroll") for (int i = 0; i < TG; ++i) acc[i] = mfma4(a[i], b, acc[i]);        \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
      LRNDE_BL(aX, bX, kg_lo);
      int kg = kg_lo;
#pragma unroll 1
      for (; kg + 2 < kg_hi; kg += 2) {
        LRNDE_BL(aY, bY, kg + 1);
        LRNDE_BM(aX, bX);
        LRNDE_BL(aX, bX, kg + 2);
        LRNDE_BM(aY, bY);
      }
      if (kg + 1 < kg_hi) { LRNDE_BL(aY, bY, kg + 1); LRNDE_BM(aX, bX); LRNDE_BM(aY, bY); }
      else { LRNDE_BM(aX, bX); }
#undef LRNDE_BL
#undef LRNDE_BM
      f32x4* pp = reinterpret_cast<f32x4*>(pl) + ((size_t)seg * m.MT1 + mt0) * 64 + lane;
#pragma unroll
      for (int i = 0; i < TG; ++i) if (mt0 + i < m.MT1) pp[i * 64] = acc[i];
    }
  }
}

// sum of the segment partials of C-fragment element e
__device__ __forceinline__ float seg_sum(const ModelDev& m, const float* pl, int e) {
  const int nseg1 = (m.KG1 + SEGK - 1) / SEGK, pstride = m.MT1 * 256;
  float v = pl[e];
  for (int s = 1; s < nseg1; ++s) v = v + pl[(size_t)s * pstride + e];
  return v;
}

// vector-Jacobian product on one 16-column tile
template <int W> __global__ __launch_bounds__(NT) void k_vjp(VjpArgs a) {
  const ModelDev& m = a.m;
  const Smem s = carve(m);
  float* dact = reinterpret_cast<float*>(reinterpret_cast<char*>(s.bc + 1) + 16);  // [Hp*16] act'(pre), C-fragment order (extra LDS)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, rq = lane >> 4;
  const int b0 = blockIdx.x * NB, nvalid = min(NB, a.B - b0);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  smem_init(m, s);
  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)m.W1p, 0, (((m.MT1 + TG - 1) / TG) * TG) * m.KG1 * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.V1p, 0, (((m.MT1 + TG - 1) / TG) * TG) * m.KG1 * 1024, 0x00020000);
  const int KG2p = ((m.KG2 + SEGK - 1) / SEGK) * SEGK;
  const __amdgpu_buffer_rsrc_t rsU2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.U2p, 0, m.MT2 * KG2p * 1024, 0x00020000);
  // ---- phase 0: y tile (given or interpolated) -> LDS x tile and scratch ----
  __syncthreads();
  tile_foreach<W>(m, b0, nvalid, [&](int row, int nn, bool valid, size_t g) {
    Vec<W> x = vzero<W>();
    if (valid) {
      if (a.y) {
        x = vload<W>(a.y + g);
      } else {
        const size_t nst = (size_t)a.B * m.D;
        // the record is in polynomial form: [uprev, k1, P2, P3, P4] (lrnde_math.hpp tsit5_rec_eval)
        const Vec<W> y0 = vload<W>(a.dense + g), v1 = vload<W>(a.dense + nst + g), v2 = vload<W>(a.dense + 2 * nst + g),
                     v3 = vload<W>(a.dense + 3 * nst + g), v4 = vload<W>(a.dense + 4 * nst + g);
#pragma unroll
        for (int h = 0; h < W; ++h) x.v[h] = tsit5_rec_eval(y0.v[h], v1.v[h], v2.v[h], v3.v[h], v4.v[h], a.theta, a.dense_dt);
      }
      vstore<W>(a.ysc + g, x);
    }
    lds_put<W>(s.xl, row, nn, x);
  });
  __syncthreads();
  // ---- phase 1: pre = W1 [y;t] + b1 ; h, act' ----
  gemm_ksplit(m, rsW1, s.xl, s.pl);
  __syncthreads();
  {
    const float* w1t = s.bias; const float* b1 = w1t + m.Hp;
    for (int e = threadIdx.x; e < m.MT1 * 256; e += NT) {
      const int r = e & 3, l = (e >> 2) & 63, mt = e >> 8;
      const int o = mt * 16 + (l >> 4) * 4 + r, nn = l & 15;
      float pre = seg_sum(m, s.pl, e);
      pre = m.td ? fma_(w1t[o], a.t, pre) : pre;
      pre = pre + b1[o];
      const float h = act_apply(m.act, pre);
      dact[e] = act_deriv_c(m.act, pre, h);
      if (nn < nvalid) a.hsc[(size_t)(b0 + nn) * m.Hp + o] = h;
    }
  }
  __syncthreads();
  // ---- phase 2: dh = W2^T lam ; dpre = dh .* act' -> h tile image + scratch ----
  tile_foreach<W>(m, b0, nvalid, [&](int row, int nn, bool valid, size_t g) {
    const Vec<W> x = valid ? vload<W>(a.lam + g) : vzero<W>();
    lds_put<W>(s.xl, row, nn, x);
  });
  __syncthreads();
  gemm_ksplit(m, rsV1, s.xl, s.pl);
  __syncthreads();
  for (int e = threadIdx.x; e < m.MT1 * 256; e += NT) {
    const int r = e & 3, l = (e >> 2) & 63, mt = e >> 8;
    const int o = mt * 16 + (l >> 4) * 4 + r, nn = l & 15;
    const float dpre = seg_sum(m, s.pl, e) * dact[e];
    s.hl[((mt * 64 + r * 16 + nn) << 2) + (l >> 4)] = dpre;
    if (nn < nvalid) a.dpsc[(size_t)(b0 + nn) * m.Hp + o] = dpre;
  }
  __syncthreads();
  // ---- phase 3: dy = W1^T dpre ----
  {
    const f32x4* hp = reinterpret_cast<const f32x4*>(s.hl) + lane;
    const int voff = lane * 16;
    const int nseg2 = (m.KG2 + SEGK - 1) / SEGK;
    for (int mt = wave; mt < m.MT2; mt += NW) {
      f32x4 tot = zero4;
      for (int sg = 0; sg < nseg2; ++sg) {
        f32x4 acc = zero4;
        f32x4 av[SEGK], bv[SEGK];
#pragma unroll
        for (int j = 0; j < SEGK; ++j) { av[j] = wload(rsU2, voff, (mt * KG2p + sg * SEGK + j) * 1024); bv[j] = hp[(sg * SEGK + j) * 64]; }
#pragma unroll
        for (int j = 0; j < SEGK; ++j) acc = mfma4(av[j], bv[j], acc);
        if (sg == 0) tot = acc;
        else { tot.x = tot.x + acc.x; tot.y = tot.y + acc.y; tot.z = tot.z + acc.z; tot.w = tot.w + acc.w; }
      }
      const int row0 = mt * 16 + rq * 4;
      if (n < nvalid) {
        float* dst = a.dy + (size_t)(b0 + n) * m.D + row0;
        if constexpr (W == 4) { if (row0 < m.D) *reinterpret_cast<f32x4*>(dst) = tot; }
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (row0 + r < m.D) dst[r] = tot[r];
        }
      }
    }
  }
}

// (df/dp)^T lam in the flat Lux layout: gW1 (H x (D+td)), gb1, gW2 (D x (H+td)), gb2.
// One wave per 16x16 output tile, fp32 MFMA chain over the batch (K = B); bias / time columns by
// column sums.  out = scale_old*out + result is NOT done here: the kernel overwrites gp.
struct PgradArgs {
  int D, H, Hp, td, B;
  float t;
  const float* lam;   // (B,D)
  const float* y;     // (B,D)
  const float* h;     // (B,Hp)
  const float* dpre;  // (B,Hp)
  float* gp;          // flat (P)
  int ntile1, ntile2, nt1c, nt2c;  // tiles of gW1: ceil(H/em) x ceil((D+2)/en); gW2: ceil(D/em) x ceil((H+2)/en) for an em x en tile
  int ts;          // shape code of a workgroup's output tile: 1 = 16 x 16, 2 = 32 x 32, 3 = 32 x 64 (pgrad_tile_any)
  // device-resolved form (lrnde_adjoint.hpp): t, lam and gp of the evaluation come from the control block
  int adj_mode, adj_stage, adj_j;
  int accumulate;  // 1: gp += result (the regulariser's reverse sweep sums six evaluations' cotangents), 0: overwrite
  int cc;          // 1: operands (scratch y / h / dpre, lam) were written by a launch that is still running elsewhere on the
                   // chip (overlapped stage launches): read them past the L2 (agent scope)
};

// t / lam / gp of a parameter-gradient GEMM that belongs to the adjoint loop's evaluation (mode, stage) of attempt j
__device__ __forceinline__ bool pgrad_resolve(PgradArgs& a, const AdjArgs& g) {
  if (a.adj_mode == ADJ_HOST) return true;
  if (a.adj_mode == ADJ_FSAL || a.adj_mode == ADJ_INIT_B) {
    const AdjCtrl* cp = g.ctl;
    const int cur = cp->cur;
    if (a.adj_mode == ADJ_FSAL) { a.t = -cp->t; a.lam = adj_zb(g, cur); a.gp = adj_K(g, 0, cur) + g.n_lam; }
    else { a.t = cp->st[0].t; a.lam = adj_zs(g); a.gp = adj_K(g, 1, cur) + g.n_lam; }
    return true;
  }
  // (fields read through the pointer: a private copy of the block indexed by the runtime stage would live in scratch)
  const AdjCtrl* cp = g.ctl + ((a.adj_j + 1) & 1);
  int do_step, cur; float tt;
  if (g.sync) {   // published by a launch that may still be running: past the L2
    const int* d = reinterpret_cast<const int*>(cp);
    do_step = ldcc(d + offsetof(AdjCtrl, do_step) / 4); cur = ldcc(d + offsetof(AdjCtrl, cur) / 4);
    tt = __builtin_bit_cast(float, ldcc(d + (offsetof(AdjCtrl, st) + (a.adj_stage - 2) * sizeof(AdjStage) + offsetof(AdjStage, t)) / 4));
  } else { do_step = cp->do_step; cur = cp->cur; tt = cp->st[a.adj_stage - 2].t; }
  if (!do_step) return false;
  a.t = tt;
  a.lam = adj_stage_lam(g, a.adj_stage, cur);
  a.gp = adj_K(g, a.adj_stage - 1, cur) + g.n_lam;
  return true;
}

// the end of an adjoint step folded into the last GEMM's tiles (fold != nullptr): the lane that holds an entry of K7's mu part
// forms z_new's and the residual's entry on the spot (K1..K6 and z at the same index; the arithmetic of k_adj_err) and the
// tile leaves ONE fp64 partial — no launch of its own for the mu part of the error norm
struct AdjMuFold { const float* K[6]; const float* z; float* zn; float A7[6], BT[7]; float dt, abstol, reltol; double* part; };
// TSM x TSN MFMA tiles of 16 x 16 per workgroup (the launch's shape code a.ts: 1 = 1 x 1, 2 = 2 x 2, 3 = 2 x 4).  A wave's loads
// per k-step are TSM A values + TSN B values for TSM*TSN MFMAs: at 1 x 1 the GEMM re-read its operands 45 MB per launch at
// B = 512 through the L2 (700 workgroups, two loads per MFMA), at 2 x 2 half of that with 200 workgroups, at 2 x 4 (32 x 64
// outputs) 102 workgroups.  The sums are the same chains in the same order in every shape.
template <bool CC = false, int TSM = 1, int TSN = 1>
__device__ __forceinline__ void pgrad_tile(const PgradArgs& a, const int tile, const AdjMuFold* fold = nullptr) {  // one output tile per workgroup, the batch (K) split over its 4 waves
  __shared__ f32x4 red[3][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const size_t ob1 = (size_t)a.H * (a.D + a.td), oW2 = ob1 + a.H, ob2 = oW2 + (size_t)a.D * (a.H + a.td);
  // C[i][j] = sum_b A[b][i] * Bm[b][j]:  gW1: A = dpre (rows o), Bm = [y, t, 1];  gW2: A = lam (rows i), Bm = [h, t, 1].
  // The two virtual columns (value t and value 1 for every sample) give the time column of the
  // weight gradient and the bias gradient from the same MFMA chain.
  const bool first = tile < a.ntile1;
  const int tt = first ? tile : tile - a.ntile1;
  const int ncol = first ? a.nt1c : a.nt2c;
  const int ti = tt / ncol, tj = tt % ncol;
  const float* A = first ? a.dpre : a.lam;
  const float* Bm = first ? a.y : a.h;
  const int lda = first ? a.Hp : a.D, ldb = first ? a.D : a.Hp;
  const int M = first ? a.H : a.D, N = first ? a.D : a.H;
  bool rok[TSM], cok[TSN];
  float cconst[TSN];
  const float* Ap[TSM];
  const float* Bp[TSN];
#pragma unroll
  for (int s = 0; s < TSM; ++s) {
    const int row = (ti * TSM + s) * 16 + li;
    rok[s] = row < M;
    Ap[s] = A + (rok[s] ? row : 0);
  }
#pragma unroll
  for (int s = 0; s < TSN; ++s) {
    const int col = (tj * TSN + s) * 16 + li;
    cok[s] = col < N;
    cconst[s] = (col == N) ? a.t : ((col == N + 1) ? 1.0f : 0.0f);
    Bp[s] = Bm + (cok[s] ? col : 0);
  }
  f32x4 acc[TSM][TSN];
#pragma unroll
  for (int si = 0; si < TSM; ++si)
#pragma unroll
    for (int sj = 0; sj < TSN; ++sj) acc[si][sj] = f32x4{0.f, 0.f, 0.f, 0.f};
  // D fragment: row = ti*16 + lk*4 + r, col = tj*16 + li ; flat weight index = row + M*col
  float* const gW = a.gp + (first ? (size_t)0 : oW2);
  float* const gb = a.gp + (first ? ob1 : ob2);
  auto dst_of = [&](int si, int sj, int r) -> float* {
    const int c = (tj * TSN + sj) * 16 + li, rr = (ti * TSM + si) * 16 + lk * 4 + r;
    if (rr >= M) return nullptr;
    if (c < N) return gW + (size_t)rr + (size_t)M * c;
    if (c == N) return a.td ? gW + (size_t)rr + (size_t)M * N : nullptr;
    if (c == N + 1) return gb + rr;
    return nullptr;
  };
  // fold: K1..K6 and z at the lane's (up to) four entries of a sub-tile.  The four entries are consecutive rows of one column:
  // one 16-byte load per array when the layout allows it (M % 4 == 0, as for the MNIST field) — as scalar loads a tile touched
  // 2048 cache lines for 1024 entries and the launch got 14 us longer
  struct FoldOps { float kv[4][6], zv[4]; };
  auto fold_load = [&](int si, int sj, FoldOps& o) {
    float* d0 = dst_of(si, sj, 0);
    float* d3 = dst_of(si, sj, 3);
    const bool vec4 = (M & 3) == 0 && d0 && d3 == d0 + 3 && ((d0 - a.gp) & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(fold->z) | reinterpret_cast<uintptr_t>(fold->K[0])) & 15) == 0;
    if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(vec4 || !d0)) == 64)) {
      const size_t i = d0 ? (size_t)(d0 - a.gp) : 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(fold->K[q] + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) o.kv[r][q] = v[r];
      }
      const f32x4 zz = *reinterpret_cast<const f32x4*>(fold->z + i);
#pragma unroll
      for (int r = 0; r < 4; ++r) o.zv[r] = zz[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* d = dst_of(si, sj, r);
        const size_t i = d ? (size_t)(d - a.gp) : 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) o.kv[r][q] = fold->K[q][i];
        o.zv[r] = fold->z[i];
      }
    }
  };
  // one tile per workgroup: the fold's operands are requested BEFORE the GEMM's (they were written by other launches — nothing
  // of them is in the L2; behind the GEMM they were a round trip of their own at the end of every tile)
  FoldOps pre;
  constexpr bool hoist = TSM * TSN == 1;
  if (hoist && fold && wave == 0) fold_load(0, 0, pre);
  constexpr int UN = 8;   // 8 MFMA k-steps (32 samples) per block
  constexpr int GB = TSM + TSN == 2 ? 4 : 2;   // (2 x 4 with all four blocks in flight: 256 registers and spills, a tile 14.6 -> 16.4 us)   // blocks whose loads are all in flight before the first MFMA (64 independent loads per lane):
                          // at B = 512 and 1 x 1 a wave's whole share; block after block the kernel paid one L2 round trip per block
  // blocks of 32 samples go round-robin to the 4 waves (fixed, so the summation order is fixed)
  const int nblk = (a.B + 4 * UN - 1) / (4 * UN);
  for (int blk0 = wave < 4 ? wave : nblk; blk0 < nblk; blk0 += 4 * GB) {  // waves beyond the fourth (512-thread launch) only join the barrier
    float av[GB][UN][TSM], bv[GB][UN][TSN];
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const int b0 = (blk0 + 4 * g) * 4 * UN;
      if (b0 + 4 * UN <= a.B) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const size_t b = (size_t)(b0 + 4 * u + lk);
#pragma unroll
          for (int s = 0; s < TSM; ++s) {
            if constexpr (CC) av[g][u][s] = rok[s] ? ldcc(Ap[s] + b * lda) : 0.f;
            else av[g][u][s] = rok[s] ? Ap[s][b * lda] : 0.f;
          }
#pragma unroll
          for (int s = 0; s < TSN; ++s) {
            if constexpr (CC) bv[g][u][s] = cok[s] ? ldcc(Bp[s] + b * ldb) : cconst[s];
            else bv[g][u][s] = cok[s] ? Bp[s][b * ldb] : cconst[s];
          }
        }
      } else {  // the ragged last block, or no block at all (zeros add nothing)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int b = b0 + 4 * u + lk;
          const bool bok = b < a.B;
#pragma unroll
          for (int s = 0; s < TSM; ++s) {
            if constexpr (CC) av[g][u][s] = (rok[s] && bok) ? ldcc(Ap[s] + (size_t)b * lda) : 0.f;
            else av[g][u][s] = (rok[s] && bok) ? Ap[s][(size_t)b * lda] : 0.f;
          }
#pragma unroll
          for (int s = 0; s < TSN; ++s) {
            if constexpr (CC) bv[g][u][s] = bok ? (cok[s] ? ldcc(Bp[s] + (size_t)b * ldb) : cconst[s]) : 0.f;
            else bv[g][u][s] = bok ? (cok[s] ? Bp[s][(size_t)b * ldb] : cconst[s]) : 0.f;
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < GB; ++g)
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int si = 0; si < TSM; ++si)
#pragma unroll
          for (int sj = 0; sj < TSN; ++sj) acc[si][sj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][u][si], bv[g][u][sj], acc[si][sj], 0, 0, 0);
  }
  // chains 1..3 join chain 0 in wave 0, sub-tile by sub-tile through the one staging array
#pragma unroll
  for (int si = 0; si < TSM; ++si)
#pragma unroll
    for (int sj = 0; sj < TSN; ++sj) {
      if (TSM * TSN > 1 && (si | sj)) __syncthreads();
      if (wave > 0 && wave < 4) red[wave - 1][lane] = acc[si][sj];
      __syncthreads();
      if (wave == 0) {
        acc[si][sj] = acc[si][sj] + red[0][lane];
        acc[si][sj] = acc[si][sj] + red[1][lane];
        acc[si][sj] = acc[si][sj] + red[2][lane];
      }
    }
  if (wave > 0) return;
  double esum = 0.0;
#pragma unroll
  for (int si = 0; si < TSM; ++si)
#pragma unroll
    for (int sj = 0; sj < TSN; ++sj) {
      const f32x4 accv = acc[si][sj];
      float* dsts[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dst = dst_of(si, sj, r);
        dsts[r] = dst;
        if (dst) *dst = a.accumulate ? *dst + accv[r] : accv[r];
      }
      if (fold) {
        FoldOps late;
        if (!hoist) fold_load(si, sj, late);
        const FoldOps& o = hoist ? pre : late;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (!dsts[r]) continue;
          const size_t i = (size_t)(dsts[r] - a.gp);
          float sacc = fold->A7[0] * o.kv[r][0];
#pragma unroll
          for (int q = 1; q < 6; ++q) sacc = sacc + fold->A7[q] * o.kv[r][q];
          const float znv = o.zv[r] + fold->dt * sacc;
          fold->zn[i] = znv;
          float sb = fold->BT[0] * o.kv[r][0];
#pragma unroll
          for (int q = 1; q < 6; ++q) sb = sb + fold->BT[q] * o.kv[r][q];
          sb = sb + fold->BT[6] * accv[r];
          const float ut = 0.f + fold->dt * sb;
          const float sc = fold->abstol + fmaxf_(__builtin_fabsf(o.zv[r]), __builtin_fabsf(znv)) * fold->reltol;
          const float rres = ut / sc;
          esum += (double)(rres * rres);
        }
      }
    }
  if (fold) {
    esum = wave_sum_dpp(esum);
    if (lane == 0) fold->part[tile] = esum;
  }
}
// the tile shape is the launch's (PgradArgs::ts, set with the tile counts by pgrad_args)
// (WIDE: the 32 x 64 shape exists in this kernel — the VJP launch that carries the tiles; it holds 192 operand registers per
//  lane in flight, which a kernel of tiles alone would pay for in occupancy)
template <bool CC = false, bool WIDE = false>
__device__ __forceinline__ void pgrad_tile_any(const PgradArgs& a, const int tile, const AdjMuFold* fold = nullptr) {
  if constexpr (WIDE) { if (a.ts == 3) { pgrad_tile<CC, 2, 4>(a, tile, fold); return; } }
  if (a.ts == 2) pgrad_tile<CC, 2, 2>(a, tile, fold);
  else pgrad_tile<CC, 1, 1>(a, tile, fold);
}

__global__ __launch_bounds__(256) void k_pgrad(PgradArgs a) { pgrad_tile_any(a, blockIdx.x); }
__global__ __launch_bounds__(256) void k_pgrad_adj(PgradArgs a, AdjArgs g) {
  if (!pgrad_resolve(a, g)) return;
  pgrad_tile_any(a, blockIdx.x);
}

// W2^T / W1^T in the forward fragment layouts (see k_pack)
__global__ void k_pack_t(const float* p, int D, int H, int td, int Dp, int Hp, float* V1p, float* U2p) {
  const int KG1 = Dp / 16, KG2 = ((Hp / 16 + SEGK - 1) / SEGK) * SEGK;
  const int MT1p = ((Hp / 16 + TG - 1) / TG) * TG;
  const size_t n1 = (size_t)MT1p * KG1 * 256, n2 = (size_t)(Dp / 16) * KG2 * 256;
  const size_t base2 = (size_t)H * (D + td) + H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.x * blockDim.x) {
    if (i < n1) {  // V1[o][k] = W2[k][o]
      const int q = i & 3, lane = (i >> 2) & 63;
      const size_t blk = i >> 8;
      const int kg = blk % KG1, mt = blk / KG1;
      const int o = mt * 16 + (lane & 15), k = kg * 16 + q * 4 + (lane >> 4);
      V1p[i] = (o < H && k < D) ? p[base2 + (size_t)k + (size_t)D * o] : 0.f;
    } else {  // U2[i][k] = W1[k][i]
      const size_t e = i - n1;
      const int q = e & 3, lane = (e >> 2) & 63;
      const size_t blk = e >> 8;
      const int kg = blk % KG2, mt = blk / KG2;
      const int o = mt * 16 + (lane & 15), k = kg * 16 + q * 4 + (lane >> 4);
      U2p[e] = (o < D && k < H) ? p[(size_t)k + (size_t)H * o] : 0.f;
    }
  }
}

// ---- elementwise kernels on flat vectors (augmented adjoint state, cotangents) ----
struct AxArgs { float* out; const float* base; const float* k[7]; float c[7]; int nk; float dt; size_t n; };
// out = base + dt * (c0*k0 + c1*k1 + ...)   (left to right); ONE term: base + (dt*c0)*k0 — the operation order of the
// Tsit5 step's second stage (src/perform_step.jl:11-12: `a = dt * a21; uprev + a * k1`), which is also what upstream's own
// perform_step does in the adjoint's reversed solve
__global__ void k_axpy(AxArgs a) {
  if (a.nk == 1) {
    const float c = a.dt * a.c[0];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x)
      a.out[i] = (a.base ? a.base[i] : 0.f) + c * a.k[0][i];
    return;
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    float s = a.c[0] * a.k[0][i];
    for (int j = 1; j < a.nk; ++j) s = s + a.c[j] * a.k[j][i];
    a.out[i] = (a.base ? a.base[i] : 0.f) + a.dt * s;
  }
}
// sum over i of ((num_i [- num2_i]) / (abstol + max(|a_i|,|b_i|)*reltol))^2 -> per-block doubles
struct NormArgs { const float* num; const float* num2; const float* sa; const float* sb; float abstol, reltol; size_t n; double* part; };
__global__ __launch_bounds__(256) void k_norm(NormArgs a) {
  __shared__ double red[4];
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = a.num2 ? a.num[i] - a.num2[i] : a.num[i];
    const float sa = __builtin_fabsf(a.sa[i]), sb = a.sb ? __builtin_fabsf(a.sb[i]) : sa;
    const float sc = a.abstol + fmaxf_(sa, sb) * a.reltol;
    const float r = v / sc;
    acc += (double)(r * r);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// both parts of the augmented vector in one launch: blocks 0..255 the lambda part (a), blocks 256..511 the mu part (b)
__global__ __launch_bounds__(256) void k_norm2(NormArgs a, NormArgs b) {
  __shared__ double red[4];
  const bool second = blockIdx.x >= 256;
  const NormArgs& g = second ? b : a;
  const unsigned bx = second ? blockIdx.x - 256 : blockIdx.x;
  double acc = 0.0;
  for (size_t i = bx * (size_t)blockDim.x + threadIdx.x; i < g.n; i += (size_t)256 * blockDim.x) {
    const float v = g.num2 ? g.num[i] - g.num2[i] : g.num[i];
    const float sa = __builtin_fabsf(g.sa[i]), sb = g.sb ? __builtin_fabsf(g.sb[i]) : sa;
    const float sc = g.abstol + fmaxf_(sa, sb) * g.reltol;
    const float r = v / sc;
    acc += (double)(r * r);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) g.part[bx] = red[0] + red[1] + red[2] + red[3];
}

// regulariser sweep, end of a stage: xbar (+= ubar / g6bar), then kbar_j += dt * a_sj * xbar for the earlier stages j
struct SweepJoinArgs { size_t n; float* xb; const float* extra; float dt; int nk; float* kb[5]; float c[5]; };
__global__ void k_sweep_join(SweepJoinArgs a) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    float x = a.xb[i];
    if (a.extra) { x = x + a.extra[i]; a.xb[i] = x; }
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (j < a.nk) a.kb[j][i] = a.kb[j][i] + a.c[j] * (a.dt * x);  // pullback of uprev + dt*(sum_j a_sj k_j): through `dt *` first, then a_sj
  }
}

// End of an adjoint Tsit5 step in one launch: utilde = dt * sum btilde_j K_j is formed on the fly (the arithmetic of
// k_axpy without a base) and goes straight into the error norm's per-block sums (those of k_norm2) without being stored;
// the mu part of u_{n+1} (its lambda part is stage 7's input, already written by the VJP) is formed here too.
struct AdjErrArgs {
  const float* K[7]; float BT[7], A7[6]; float dt;
  const float* z; float* zn; size_t n_lam, P; float abstol, reltol; double* part;
};
__global__ __launch_bounds__(256) void k_adj_err(AdjErrArgs a) {
  __shared__ double red[4];
  const bool second = blockIdx.x >= 256;
  const unsigned bx = second ? blockIdx.x - 256 : blockIdx.x;
  const size_t off = second ? a.n_lam : 0, cnt = second ? a.P : a.n_lam;
  double acc = 0.0;
  for (size_t j = bx * (size_t)blockDim.x + threadIdx.x; j < cnt; j += (size_t)256 * blockDim.x) {
    const size_t i = off + j;
    float kv[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) kv[q] = a.K[q][i];
    const float zv = a.z[i];
    float znv;
    if (second) {
      float s = a.A7[0] * kv[0];
#pragma unroll
      for (int q = 1; q < 6; ++q) s = s + a.A7[q] * kv[q];
      znv = zv + a.dt * s;
      a.zn[i] = znv;
    } else {
      znv = a.zn[i];
    }
    float s = a.BT[0] * kv[0];
#pragma unroll
    for (int q = 1; q < 7; ++q) s = s + a.BT[q] * kv[q];
    const float ut = 0.f + a.dt * s;
    const float sc = a.abstol + fmaxf_(__builtin_fabsf(zv), __builtin_fabsf(znv)) * a.reltol;
    const float r = ut / sc;
    acc += (double)(r * r);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.part[(second ? 256 : 0) + bx] = red[0] + red[1] + red[2] + red[3];
}

// the same with z / z_new / K_j / dt taken from the control block of attempt j (device-side controller)
// (block bx of 256 of the lambda part (second = false) or of the mu part: the two halves may run in different launches)
__device__ __forceinline__ void adj_err_blocks(const AdjErrArgs& a, const AdjArgs& g, int j, const unsigned bx, const bool second) {
  const AdjCtrl c = g.ctl[(j + 1) & 1];
  if (!c.do_step) return;
  __shared__ double red[4];
  const size_t off = second ? a.n_lam : 0, cnt = second ? a.P : a.n_lam;
  const float* K[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) K[q] = adj_K(g, q, c.cur);
  const float* z = adj_zb(g, c.cur);
  float* zn = adj_zb(g, c.cur ^ 1);
  const float dt = c.dt;
  double acc = 0.0;
  if (((a.n_lam | a.P) & 3) == 0) {
    // Four elements per lane and two such quads requested before the first is used: the scalar loop below is a chain of
    // memory round trips (nine 4-byte loads, then the next iteration's; the K's were written by other launches), six of
    // them per thread for the MNIST field.  Per-thread order of the fp64 adds: quad by quad, x y z w.
    const size_t nq = cnt >> 2, stride = (size_t)256 * blockDim.x;
    auto ld = [&](size_t qi, f32x4* kv, f32x4& zv, f32x4& znv) {
      const size_t i = off + (qi << 2);
#pragma unroll
      for (int q = 0; q < 7; ++q) kv[q] = *reinterpret_cast<const f32x4*>(K[q] + i);
      zv = *reinterpret_cast<const f32x4*>(z + i);
      if (!second) znv = *reinterpret_cast<const f32x4*>(zn + i);
    };
    auto use = [&](size_t qi, const f32x4* kv, const f32x4& zv, f32x4 znv) {
      const size_t i = off + (qi << 2);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (second) {
          float s = a.A7[0] * kv[0][h];
#pragma unroll
          for (int q = 1; q < 6; ++q) s = s + a.A7[q] * kv[q][h];
          znv[h] = zv[h] + dt * s;
        }
        float s = a.BT[0] * kv[0][h];
#pragma unroll
        for (int q = 1; q < 7; ++q) s = s + a.BT[q] * kv[q][h];
        const float ut = 0.f + dt * s;
        const float sc = a.abstol + fmaxf_(__builtin_fabsf(zv[h]), __builtin_fabsf(znv[h])) * a.reltol;
        const float r = ut / sc;
        acc += (double)(r * r);
      }
      if (second) *reinterpret_cast<f32x4*>(zn + i) = znv;
    };
    for (size_t q0 = bx * (size_t)blockDim.x + threadIdx.x; q0 < nq; q0 += 2 * stride) {
      const size_t q1 = q0 + stride;
      f32x4 ka[7], kb[7], za, zb, zna = {0.f, 0.f, 0.f, 0.f}, znb = {0.f, 0.f, 0.f, 0.f};
      ld(q0, ka, za, zna);
      if (q1 < nq) ld(q1, kb, zb, znb);
      use(q0, ka, za, zna);
      if (q1 < nq) use(q1, kb, zb, znb);
    }
  } else
  for (size_t jj = bx * (size_t)blockDim.x + threadIdx.x; jj < cnt; jj += (size_t)256 * blockDim.x) {
    const size_t i = off + jj;
    float kv[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) kv[q] = K[q][i];
    const float zv = z[i];
    float znv;
    if (second) {
      float s = a.A7[0] * kv[0];
#pragma unroll
      for (int q = 1; q < 6; ++q) s = s + a.A7[q] * kv[q];
      znv = zv + dt * s;
      zn[i] = znv;
    } else {
      znv = zn[i];
    }
    float s = a.BT[0] * kv[0];
#pragma unroll
    for (int q = 1; q < 7; ++q) s = s + a.BT[q] * kv[q];
    const float ut = 0.f + dt * s;
    const float sc = a.abstol + fmaxf_(__builtin_fabsf(zv), __builtin_fabsf(znv)) * a.reltol;
    const float r = ut / sc;
    acc += (double)(r * r);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.part[(second ? 256 : 0) + bx] = red[0] + red[1] + red[2] + red[3];
}
// 512 blocks: both parts; only_mu: 256 blocks, the mu part alone (the lambda part rode in k_pgrad_adj_err's launch)
__global__ __launch_bounds__(256) void k_adj_err_dev(AdjErrArgs a, AdjArgs g, int j, int only_mu) {
  const bool second = only_mu || blockIdx.x >= 256;
  adj_err_blocks(a, g, j, (!only_mu && second) ? blockIdx.x - 256 : blockIdx.x, second);
}
// The last evaluation's parameter-gradient GEMM of an attempt (nt tiles) and, on 256 more workgroups of the same launch,
// the LAMBDA part of the attempt's error norm: it needs nothing of this GEMM (K7's lambda part and z_new's were written by
// the VJP launches before it), so it no longer waits for it in a launch of its own.  Same blocks, same sums, same bits.
// fold_mu: the GEMM's tiles also form the mu part of z_new and of the error norm (AdjMuFold; g.mu_tiles = nt tells the next
// prologue where the partials are) — then k_adj_err_dev is not launched at all
__global__ __launch_bounds__(256) void k_pgrad_adj_err(PgradArgs a, AdjArgs g, AdjErrArgs e, int nt, int j, int fold_mu) {
  if ((int)blockIdx.x >= nt) { adj_err_blocks(e, g, j, blockIdx.x - nt, false); return; }
  if (!pgrad_resolve(a, g)) return;
  if (!fold_mu) { pgrad_tile_any(a, blockIdx.x); return; }
  const AdjCtrl* cp = g.ctl + ((j + 1) & 1);
  const int cur = cp->cur;
  AdjMuFold f;
#pragma unroll
  for (int q = 0; q < 6; ++q) { f.K[q] = adj_K(g, q, cur) + g.n_lam; f.A7[q] = e.A7[q]; }
#pragma unroll
  for (int q = 0; q < 7; ++q) f.BT[q] = e.BT[q];
  f.z = adj_zb(g, cur) + g.n_lam; f.zn = adj_zb(g, cur ^ 1) + g.n_lam;
  f.dt = cp->dt; f.abstol = e.abstol; f.reltol = e.reltol; f.part = const_cast<double*>(g.part) + ADJ_MU_TILE_OFF;
  pgrad_tile_any(a, blockIdx.x, &f);
}

// the rank's own fp64 sum (256 block partials, fixed order) into slot[rank] of a zeroed per-rank vector: the
// all-reduce (sum) of that vector is the exact gather
__global__ void k_rank_slot(const double* part, double* slots, int rank, int nranks) {
  if (threadIdx.x < nranks) slots[threadIdx.x] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += part[i];
    slots[rank] = s;
  }
}

#include "lrnde_regseed.hpp"

// ===========================================================================================
// 4-column (q-tile) vector-Jacobian product: the same three GEMM phases as k_vjp on the tile shape
// and weight stream of lrnde_qtile.hpp (128 workgroups at B=512 instead of 32).  One stream of
// 21 blocks per wave: 7 of W1q (pre-activation), 7 of V1q = W2^T (dh), 7 of U2q = W1^T (dy).
// ===========================================================================================
__global__ void k_pack_tq(const float* p, int D, int H, int td, int KQ1p, int KQ2p, int RG1, int RG2,
                          float* V1q, float* U2q) {
  const size_t n1 = (size_t)RG1 * KQ1p * 256, n2 = (size_t)RG2 * KQ2p * 256;
  const size_t base2 = (size_t)H * (D + td) + H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = i < n1 ? i : i - n1;
    const int jj = e & 3, l = (e >> 2) & 63;
    const size_t blk = e >> 8;
    if (i < n1) {  // V1[o][k] = W2[k][o]
      const int kq = blk % KQ1p, rg = blk / KQ1p;
      const int o = rg * 64 + l, k = kq * 4 + jj;
      V1q[e] = (o < H && k < D) ? p[base2 + (size_t)k + (size_t)D * o] : 0.f;
    } else {       // U2[o][k] = W1[k][o]
      const int kq = blk % KQ2p, rg = blk / KQ2p;
      const int o = rg * 64 + l, k = kq * 4 + jj;
      U2q[e] = (o < D && k < H) ? p[(size_t)k + (size_t)H * o] : 0.f;
    }
  }
}

struct VjpQArgs {
  ModelDev m;
  const float* V1q;  // [RG1][KQ1p][64][4]
  const float* U2q;  // [RG2][KQ2p][64][4]
  int B;
  float t;
  const float* y; const float* dense; float theta, dense_dt;
  const float* lam;
  float* dy; float* ysc; float* hsc; float* dpsc;
  // optional fused stage combination of the adjoint's Tsit5 loop: lambda = base + dt*(c0 k0 + c1 k1 + ...) (the
  // arithmetic of k_axpy) formed while the tile is staged and written to lam_out for the parameter-gradient GEMM
  const float* lbase; const float* lk[6]; float lc[6]; int lnk; float ldt; float* lam_out;
  // device-resolved form: the fields above that depend on the integrator's decisions (which record step and theta, t,
  // dt, the z / K buffers) are filled in from the control block (lrnde_adjoint.hpp) at the head of the kernel
  int adj_mode, adj_stage, adj_j;
  AdjArgs adj;
  int qcols;   // batch columns per workgroup: QNB (4), or 2 to put a B <= 512 launch on all 256 CUs (0 = QNB)
  // overlapped stage launches (adj.sync != NULL): this launch's id (its arrival counter is sync[8 + (id & 7)]); ovl = 1: the
  // launch started while its producer (id - 1) was still running — it forms y, h and act' first, then waits for the
  // producer's arrivals before it touches lambda and the newest K
  int sync_id, ovl;
  // act'(pre) of this evaluation, (B, Hp) like hsc: written when dact_out != NULL (stage 6 of an adjoint attempt), read by the
  // REUSE variant (stage 7: same time t + dt, same record step, same theta -> the same y, h and act' bit for bit)
  float* dact_out; const float* dact_in;
};

constexpr int VQB = 3 * QSB1;  // stream blocks of one VJP (QSB2 == QSB1)

struct StreamV {
  __amdgpu_buffer_rsrc_t rs1, rsv, rsu;
  int v1[2], v2[2], s1[2], s2[2];
  int kq2_real;
  f32x4 ring[VRING][QSQ][2];
};

template <int B, int SLOT>
__device__ __forceinline__ void vq_stream_load(StreamV& st) {
  if constexpr (B < 2 * QSB1) {
    constexpr int BB = B % QSB1;
#pragma unroll
    for (int j = 0; j < QSQ; ++j) {
      st.ring[SLOT][j][0] = wload(B < QSB1 ? st.rs1 : st.rsv, st.v1[0], st.s1[0] + (BB * QSQ + j) * 1024);
      st.ring[SLOT][j][1] = wload(B < QSB1 ? st.rs1 : st.rsv, st.v1[1], st.s1[1] + (BB * QSQ + j) * 1024);
    }
  } else if constexpr (B < VQB) {
#pragma unroll
    for (int j = 0; j < QSQ; ++j) {
      constexpr int kq = (B - 2 * QSB1) * QSQ;
      const bool real = (kq + j) < st.kq2_real;  // wave-uniform
      st.ring[SLOT][j][0] = wload(st.rsu, real ? st.v2[0] : 0x7ffffff0, st.s2[0] + (kq + j) * 1024);
      st.ring[SLOT][j][1] = wload(st.rsu, real ? st.v2[1] : 0x7ffffff0, st.s2[1] + (kq + j) * 1024);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// the two loads (row groups 0 and 1) of quad J of stream block B
template <int B, int SLOT, int J, int KT>
__device__ __forceinline__ void vq_stream_load_quad(StreamV& st) {
  if constexpr (B < 2 * QSB1) {
    constexpr int BB = B % QSB1;
    st.ring[SLOT][J][0] = wload(B < QSB1 ? st.rs1 : st.rsv, st.v1[0], st.s1[0] + (BB * QSQ + J) * 1024);
    st.ring[SLOT][J][1] = wload(B < QSB1 ? st.rs1 : st.rsv, st.v1[1], st.s1[1] + (BB * QSQ + J) * 1024);
  } else if constexpr (B < VQB - 1 || (B == VQB - 1 && J < KT)) {  // KT: see q_stream_load_quad
    constexpr int kq = (B - 2 * QSB1) * QSQ;
    const bool real = (kq + J) < st.kq2_real;  // wave-uniform
    st.ring[SLOT][J][0] = wload(st.rsu, real ? st.v2[0] : 0x7ffffff0, st.s2[0] + (kq + J) * 1024);
    st.ring[SLOT][J][1] = wload(st.rsu, real ? st.v2[1] : 0x7ffffff0, st.s2[1] + (kq + J) * 1024);
  }
}

// 4 k-quads of the stream block in ring slot SL against the B operands b_[0..3], with the loads of block LB (two blocks
// ahead, into slot NSL) issued two at a time between the quads (see feval_qs)
template <int LB, int SL, int NSL, int KT = 4, int NQ = QSQ>
__device__ __forceinline__ void vq_block_mfma(StreamV& st, const f32x4 (&b_)[QSQ], f32x4& acc0, f32x4& acc1) {
#ifdef LRNDE_QBURST
  vq_stream_load<LB, NSL>(st);
#endif
  static_for<0, QSQ>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value;
#ifndef LRNDE_QBURST
    vq_stream_load_quad<LB, NSL, j, KT>(st);
#endif
    if constexpr (j < NQ) {  // NQ < 4: the last phase-3 block, whose quads beyond the real matrix are not loaded
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(st.ring[SL][j][0].x, b_[j].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(st.ring[SL][j][1].x, b_[j].x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(st.ring[SL][j][0].y, b_[j].y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(st.ring[SL][j][1].y, b_[j].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(st.ring[SL][j][0].z, b_[j].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(st.ring[SL][j][1].z, b_[j].z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(st.ring[SL][j][0].w, b_[j].w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(st.ring[SL][j][1].w, b_[j].w, acc1, 0, 0, 0);
    }
#ifndef LRNDE_QBURST
    __builtin_amdgcn_sched_barrier(0);
#endif
  });
  __builtin_amdgcn_sched_barrier(0);
}

// Dense-1-shaped phase: stream blocks BASE..BASE+6, segment = wave, both row groups; partials -> pl
template <int BASE, int KT>
__device__ __forceinline__ void vq_phase_ksplit(const ModelDev& m, const SmemQ& sm, StreamV& st, const f32x4* tile) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sidx = lane & 3;
  const int nseg1 = q_nseg1(m);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const f32x4* xp = tile + (size_t)((wave < nseg1 ? wave : 0) * QSEG) * 4 + sidx;
  static_for<0, QSB1>([&](auto Bc) {
    constexpr int B = decltype(Bc)::value;
    constexpr int SL = (BASE + B) % VRING, NSL = (BASE + B + 2) % VRING;
    f32x4 b_[QSQ];
#pragma unroll
    for (int j = 0; j < QSQ; ++j) b_[j] = xp[(B * QSQ + j) * 4];
    vq_block_mfma<BASE + B + 2, SL, NSL, KT>(st, b_, acc0, acc1);
  });
  if (wave < nseg1) {
    f32x4* pp = sm.pl + ((size_t)wave * m.RG1) * 64 + lane;
    pp[0] = acc0;
    if (m.RG1 > 1) pp[64] = acc1;
  }
}

static size_t smem_bytes_vq(int KQ1p, int KQ2p, int RG1, int RG2) {
  return smem_bytes_q(KQ1p, KQ2p, RG1, RG2) + (size_t)KQ1p * 4 * 16 + (size_t)RG1 * 256 * 4 + 32;
}

// fills the decision-dependent fields of `a` for a launch of the adjoint loop; false: nothing to do (solve finished)
// For stages 3..7 the control block is requested at the head of the kernel (`early`, thread 0) and consumed here, after
// the LDS initialisation and the first weight-stream requests: its round trip hides behind them.
// what a stage-3..7 launch needs of the attempt's control block (published by the stage-2 launch): read by every thread
// through the uniform pointer at the head of the kernel — scalar loads, no private copy of the 172-byte block (it lived
// in scratch), no hop through LDS, no barrier
struct AdjEarly { int do_step, cur; float dt; int lo; float theta, ddt, t; };
__device__ __forceinline__ AdjEarly adj_early_load(const VjpQArgs& a) {
  AdjEarly e;
  e.do_step = 1; e.cur = 0; e.dt = 0.f; e.lo = 0; e.theta = 0.f; e.ddt = 0.f; e.t = 0.f;
  if (a.adj_mode == ADJ_STAGE && a.adj_stage > 2) {
    const AdjCtrl* cp = a.adj.ctl + ((a.adj_j + 1) & 1);
    if (a.adj.sync) {
      // the stage-2 launch of this attempt may not have published yet (it runs beside us), and it sits on other XCDs
      if (!adj_spin(a.adj.sync, a.adj.seq0 + a.adj_j + 1, a.adj.sync + 1)) { e.do_step = 0; return e; }
      const int* d = reinterpret_cast<const int*>(cp);
      const int so = (offsetof(AdjCtrl, st) + (a.adj_stage - 2) * sizeof(AdjStage)) / 4;
      e.do_step = ldcc(d + offsetof(AdjCtrl, do_step) / 4); e.cur = ldcc(d + offsetof(AdjCtrl, cur) / 4);
      e.dt = __builtin_bit_cast(float, ldcc(d + offsetof(AdjCtrl, dt) / 4));
      e.lo = ldcc(d + so); e.theta = __builtin_bit_cast(float, ldcc(d + so + 1)); e.ddt = __builtin_bit_cast(float, ldcc(d + so + 2));
      e.t = __builtin_bit_cast(float, ldcc(d + so + 3));
      return e;
    }
    const AdjStage* sp = &cp->st[a.adj_stage - 2];
    e.do_step = cp->do_step; e.cur = cp->cur; e.dt = cp->dt;
    e.lo = sp->lo; e.theta = sp->theta; e.ddt = sp->ddt; e.t = sp->t;
  }
  return e;
}
__device__ __forceinline__ bool vjp_q_resolve(VjpQArgs& a, const AdjEarly& early) {
  if (a.adj_mode == ADJ_HOST) return true;
  __shared__ AdjCtrl sh_c;
  const AdjArgs& g = a.adj;
  if (a.adj_mode == ADJ_STAGE && a.adj_stage > 2) {
    if (!early.do_step) return false;
    const int cur = early.cur;
    const size_t nst = g.n_lam;
    const int sidx = a.adj_stage;
    a.y = nullptr;
    a.dense = g.dense + (size_t)early.lo * REC_ARRAYS * nst; a.theta = early.theta; a.dense_dt = early.ddt; a.t = early.t;
    a.lbase = adj_zb(g, cur); a.ldt = early.dt; a.lnk = sidx - 1;
    // row sidx of the tableau; terms beyond the row: the base vector with coefficient 0 (adds +-0, as the host path does)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      float v = 0.0f;
#pragma unroll
      for (int ss = 3; ss <= 7; ++ss) if (ss == sidx && q < ss - 1) v = (float)Tsit5::A[(ss - 2) * (ss - 1) / 2 + q];
      const bool on = q < sidx - 1;
      a.lk[q] = on ? adj_K(g, q, cur) : a.lbase;
      a.lc[q] = v;
    }
    a.lam_out = adj_stage_lam(g, sidx, cur);
    a.lam = a.lam_out;
    a.dy = adj_K(g, sidx - 1, cur);
    return true;
  }
  if (threadIdx.x < 64) {
    AdjCtrl c;
    if (a.adj_mode == ADJ_STAGE) {
      c = adj_prologue(g, a.adj_j);
    } else {
      c = g.ctl[0];
      if (a.adj_mode == ADJ_FSAL) {
        c.st[0] = adj_lookup(g, -c.t);
      } else {  // ADJ_INIT_B: dt0 from the partial sums of d0 and d1, evaluation at z + dt0*K1, time s0 + dt0
        const double ntot = (double)g.n_lam * (double)(g.use_slots ? g.nranks : 1) + (double)g.P;
        const float d0 = (float)sqrt(adj_norm_sum(g.ipart, g.use_slots, g.nranks, g.P != 0) / ntot);
        const float d1 = (float)sqrt(adj_norm_sum(g.ipart + 576, g.use_slots, g.nranks, g.P != 0) / ntot);
        c.dt0 = adj_dt0(d0, d1, g.dtmax);
        c.st[0] = adj_lookup(g, -(c.t + c.dt0));
      }
      c.do_step = 1;
      if (blockIdx.x == 0 && threadIdx.x == 0) { g.ctl[0] = c; g.ctl[1] = c; }
    }
    if (threadIdx.x == 0) sh_c = c;
  }
  __syncthreads();
  // (read field by field from LDS: a private copy of the block indexed by the runtime stage would live in scratch)
  if (!sh_c.do_step) return false;
  const int cur = sh_c.cur;
  const size_t nst = g.n_lam;
  a.y = nullptr;
  const int sidx = a.adj_mode == ADJ_STAGE ? a.adj_stage : 2;
  const AdjStage* sp = &sh_c.st[sidx - 2];
  a.dense = g.dense + (size_t)sp->lo * REC_ARRAYS * nst; a.theta = sp->theta; a.dense_dt = sp->ddt; a.t = sp->t;
  if (a.adj_mode == ADJ_STAGE) {
    a.lbase = adj_zb(g, cur); a.ldt = sh_c.dt; a.lnk = sidx - 1;
    // row sidx of the tableau; terms beyond the row: the base vector with coefficient 0 (adds +-0, as the host path does)
    float cf[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      float v = 0.0f;
#pragma unroll
      for (int ss = 2; ss <= 7; ++ss) if (ss == sidx && q < ss - 1) v = (float)Tsit5::A[(ss - 2) * (ss - 1) / 2 + q];
      cf[q] = v;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const bool on = q < sidx - 1;
      a.lk[q] = on ? adj_K(g, q, cur) : a.lbase;
      a.lc[q] = cf[q];
    }
    a.lam_out = adj_stage_lam(g, sidx, cur);
    a.lam = a.lam_out;
    a.dy = adj_K(g, sidx - 1, cur);
  } else if (a.adj_mode == ADJ_FSAL) {
    a.lnk = 0; a.lam = adj_zb(g, cur); a.dy = adj_K(g, 0, cur);
  } else {
    a.lbase = adj_zb(g, cur); a.ldt = sh_c.dt0; a.lnk = 1;
#pragma unroll
    for (int q = 0; q < 6; ++q) { a.lk[q] = q == 0 ? adj_K(g, 0, cur) : a.lbase; a.lc[q] = q == 0 ? 1.0f : 0.0f; }
    a.lam_out = adj_zs(g); a.lam = a.lam_out; a.dy = adj_K(g, 1, cur);
  }
  return true;
}

#ifndef LRNDE_STAMP_STAGE
#define LRNDE_STAMP_STAGE 5
#endif
#ifdef LRNDE_STAMPS
__device__ unsigned long long g_vstamps[16];  // tools/vjp_probe: phases of a stage-5 launch of the adjoint loop (workgroup 0)
#define VSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && vst_on) g_vstamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VSTAMP(i) do { } while (0)
#endif
// REUSE (stage 7 of an adjoint attempt, one-stream loop): the evaluation point is stage 6's (c6 = c7 = 1), so y, h and act' are
// stage 6's — its scratch set still holds y and h for the parameter-gradient GEMM and its act' comes from dact_in.  The launch
// skips the record's five arrays, the y tile and all of phase 1 (a third of the weight stream); same results, bit for bit.
template <int KT, bool SYNC = false, bool REUSE = false> __device__ __forceinline__ void vjp_q_body(VjpQArgs a) {
#ifdef LRNDE_STAMPS
  const bool vst_on = a.adj_mode == ADJ_STAGE && a.adj_stage == LRNDE_STAMP_STAGE;
#endif
  VSTAMP(0);
  // SYNC: a stage launch of the overlapped adjoint loop (compile-time: a run-time flag around every store and in the operand
  // loads cost the one-stream path 40 % — the loads of a phase no longer went out together)
  int* const sync = SYNC ? a.adj.sync : nullptr;
  constexpr bool wt = SYNC;              // outputs past the L2: a consumer on another XCD reads them while this launch runs
  const bool ovl = SYNC && a.ovl != 0;
  if (SYNC && blockIdx.x == 0 && threadIdx.x == 0) stwt(sync + 8 + ((a.sync_id + 2) & 7), 0);   // the launch after next's counter
  const AdjEarly early = adj_early_load(a);
  const ModelDev& m = a.m;
  const BiasPreQ bpre = bias_issue_q(m);  // in flight while the weight stream is set up; written to LDS below
  const SmemQ s = carve_q(m);
  // extra LDS behind the forward layout: the lambda tile and act'(pre)
  f32x4* ll = q_extra_smem(s);
  float* dact = reinterpret_cast<float*>(ll + (size_t)m.KQ1p * 4);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sidx = lane & 3, q = lane >> 2;
  const int qc = a.qcols > 0 ? a.qcols : QNB;
  const int b0 = blockIdx.x * qc, nvalid = min(qc, a.B - b0);
  const int nprod = (a.B + qc - 1) / qc;   // VJP workgroups of a stage launch (the producer's arrival count)
  const int KQ1 = m.D / 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  StreamV st;
  {
    const int voff = lane * 16;
    st.rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)m.W1q, 0, m.RG1 * m.KQ1p * 1024, 0x00020000);
    st.rsv = __builtin_amdgcn_make_buffer_rsrc((void*)a.V1q, 0, m.RG1 * m.KQ1p * 1024, 0x00020000);
    st.rsu = __builtin_amdgcn_make_buffer_rsrc((void*)a.U2q, 0, m.RG2 * m.KQ2p * 1024, 0x00020000);
    const bool has1 = wave < q_nseg1(m);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      st.v1[c] = (has1 && c < m.RG1 && c * 64 + lane < m.H) ? voff : 0x7ffffff0;
      st.s1[c] = (c * m.KQ1p + (has1 ? wave * QSEG : 0)) * 1024;
      const int g = wave + c * QNW;
      st.v2[c] = (g < m.RG2 && g * 64 + lane < m.D) ? voff : 0x7ffffff0;
      st.s2[c] = (g < m.RG2 ? g : 0) * m.KQ2p * 1024;
    }
    st.kq2_real = (m.H + 3) / 4;
    if constexpr (REUSE) {   // the stream starts at phase 2's first block
      vq_stream_load<QSB1, QSB1 % VRING>(st);
      vq_stream_load<QSB1 + 1, (QSB1 + 1) % VRING>(st);
    } else {
      vq_stream_load<0, 0>(st);
      vq_stream_load<1, 1>(st);
    }
  }
  smem_zero_q(m, s);
  for (int i = threadIdx.x; i < (m.KQ1p - KQ1) * 4; i += QNT) ll[KQ1 * 4 + i] = zero4;
  // adjoint loop: decision-dependent arguments from the control block (a barrier inside; nothing to do -> leave, the
  // outstanding weight requests are simply dropped with the wave)
  VSTAMP(1);
  if (!vjp_q_resolve(a, early)) return;
  VSTAMP(2);
  // ---- phase 0: y tile (given or interpolated) and lambda tile -> LDS; y -> scratch ----
  // Both passes of the tile walk (KQ1 * 4 <= 784 quads over 448 threads) put ALL their loads in flight — up to 15 arrays
  // each: the record's (y0, k1..k7), the stage base and six K vectors — before the first store.  As a loop with the
  // stores in program order it was four memory round trips in series (per pass: record, store y, then the lambda terms,
  // whose loads could not move above a store that might alias them): 13.5 k of the launch's 44 k cycles.
  __syncthreads();
  // the lambda half of phase 0, by itself (overlapped launches run it AFTER phase 1, once the producer has arrived):
  // lambda_s = base + dt * sum a_sj K_j, the newest K past the L2
  auto lambda_tile = [&]() {
    f32x4 bs[2], kv[2][6]; bool in[2], ok[2]; size_t gg[2]; int li[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int i = (int)threadIdx.x + r * QNT;
      const int sx = i & 3, kq = i >> 2;
      in[r] = i < KQ1 * 4; ok[r] = in[r] && sx < nvalid; li[r] = i;
      gg[r] = ok[r] ? (size_t)(b0 + sx) * m.D + kq * 4 : 0;
      const size_t g = gg[r];
      bs[r] = ld4(a.lbase + g);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if constexpr (wt) kv[r][j] = (j < a.lnk) ? (j == a.lnk - 1 ? ld4cc(a.lk[j] + g) : ld4(a.lk[j] + g)) : zero4;
        else kv[r][j] = (j < a.lnk) ? ld4(a.lk[j] + g) : zero4;
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (!in[r]) continue;
      f32x4 lv = zero4;
      if (ok[r]) {
        if (a.lnk == 1) {
          const float c0 = a.ldt * a.lc[0];
#pragma unroll
          for (int h = 0; h < 4; ++h) lv[h] = bs[r][h] + c0 * kv[r][0][h];
        } else {
          f32x4 sacc;
#pragma unroll
          for (int h = 0; h < 4; ++h) sacc[h] = a.lc[0] * kv[r][0][h];
#pragma unroll
          for (int j = 1; j < 6; ++j)
#pragma unroll
            for (int h = 0; h < 4; ++h) sacc[h] = sacc[h] + a.lc[j] * kv[r][j][h];
#pragma unroll
          for (int h = 0; h < 4; ++h) lv[h] = bs[r][h] + a.ldt * sacc[h];
        }
        if constexpr (wt) st4wt(a.lam_out + gg[r], lv); else st4(a.lam_out + gg[r], lv);
      }
      ll[li[r]] = lv;
    }
  };
  if constexpr (REUSE) {
    lambda_tile();
  } else
  if (ovl) {   // y tile alone: record -> y -> LDS x tile and scratch
    const size_t nst = (size_t)a.B * m.D;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int i = (int)threadIdx.x + r * QNT;
      const int sx = i & 3, kq = i >> 2;
      if (i >= KQ1 * 4) continue;
      f32x4 x = zero4;
      if (sx < nvalid) {
        const size_t g = (size_t)(b0 + sx) * m.D + kq * 4;
        f32x4 dv[5];
#pragma unroll
        for (int qq = 0; qq < 5; ++qq) dv[qq] = ld4(a.dense + (size_t)qq * nst + g);
#pragma unroll
        for (int h = 0; h < 4; ++h) x[h] = tsit5_rec_eval(dv[0][h], dv[1][h], dv[2][h], dv[3][h], dv[4][h], a.theta, a.dense_dt);
        st4wt(a.ysc + g, x);
      }
      s.xl[i] = x;
    }
  } else
  {
    const size_t nst = (size_t)a.B * m.D;
    bool in[2], ok[2]; size_t gg[2]; int li[2];
    f32x4 dv[2][5], bs[2], kv[2][6];   // the record is in polynomial form: [uprev, k1, P2, P3, P4] (lrnde_math.hpp)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int i = (int)threadIdx.x + r * QNT;
      const int sx = i & 3, kq = i >> 2;
      in[r] = i < KQ1 * 4; ok[r] = in[r] && sx < nvalid; li[r] = i;
      gg[r] = ok[r] ? (size_t)(b0 + sx) * m.D + kq * 4 : 0;   // (masked lanes read element 0 and are zeroed below)
      const size_t g = gg[r];
      if (a.y) {
        dv[r][0] = ld4(a.y + g);
      } else {
#pragma unroll
        for (int qq = 0; qq < 5; ++qq) dv[r][qq] = ld4(a.dense + (size_t)qq * nst + g);
      }
      if (a.lnk > 0) {
        // the lnk real terms only (lnk is launch-uniform: plain branches, no waits between the loads); the others enter
        // the sum as coefficient 0 times +0
        bs[r] = ld4(a.lbase + g);
#pragma unroll
        for (int j = 0; j < 6; ++j) kv[r][j] = (j < a.lnk) ? ld4(a.lk[j] + g) : zero4;
      } else {
        bs[r] = ld4(a.lam + g);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (!in[r]) continue;
      f32x4 x = zero4, lv = zero4;
      if (ok[r]) {
        const size_t g = gg[r];
        if (a.y) {
          x = dv[r][0];
        } else {
#pragma unroll
          for (int h = 0; h < 4; ++h)
            x[h] = tsit5_rec_eval(dv[r][0][h], dv[r][1][h], dv[r][2][h], dv[r][3][h], dv[r][4][h], a.theta, a.dense_dt);
        }
        if constexpr (wt) st4wt(a.ysc + g, x); else st4(a.ysc + g, x);
        if (a.lnk == 1) {  // one term (stage 2, initdt's Euler step): base + (dt*c0)*k0, the order of k_axpy / perform_step.jl:11-12
          const float c0 = a.ldt * a.lc[0];
#pragma unroll
          for (int h = 0; h < 4; ++h) lv[h] = bs[r][h] + c0 * kv[r][0][h];
          if constexpr (wt) st4wt(a.lam_out + g, lv); else st4(a.lam_out + g, lv);
        } else if (a.lnk > 0) {
          f32x4 sacc;
#pragma unroll
          for (int h = 0; h < 4; ++h) sacc[h] = a.lc[0] * kv[r][0][h];
#pragma unroll
          for (int j = 1; j < 6; ++j)
#pragma unroll
            for (int h = 0; h < 4; ++h) sacc[h] = sacc[h] + a.lc[j] * kv[r][j][h];
#pragma unroll
          for (int h = 0; h < 4; ++h) lv[h] = bs[r][h] + a.ldt * sacc[h];
          if constexpr (wt) st4wt(a.lam_out + g, lv); else st4(a.lam_out + g, lv);
        } else {
          lv = bs[r];
        }
      }
      s.xl[li[r]] = x;
      ll[li[r]] = lv;
    }
  }
  bias_write_q(m, s, bpre);  // (first read in epilogue 1)
  __syncthreads();
  VSTAMP(3);
  const int h64 = m.RG1 * 64;
  const float* w1t = s.bias; const float* b1 = w1t + h64;
  const float* plf = reinterpret_cast<const float*>(s.pl);
  float* hlf = reinterpret_cast<float*>(s.hl);
  const int ne = m.RG1 * 256;
  const int nseg1 = q_nseg1(m);
  if constexpr (REUSE) {
    // act' of the evaluation at the same point (stage 6), in the epilogue's element order
    for (int e = threadIdx.x; e < ne; e += QNT) {
      const int r = e & 3, l = (e >> 2) & 63, rg = e >> 8;
      const int o = rg * 64 + (l >> 2) * 4 + r, sx = l & 3;
      if ((o >> 2) >= m.KQ2p) continue;
      dact[e] = (sx < nvalid && o < m.Hp) ? a.dact_in[(size_t)(b0 + sx) * m.Hp + o] : 0.f;
    }
  } else {
  // ---- phase 1: pre = W1 [y;t] + b1 ; h, act' ----
  vq_phase_ksplit<0, KT>(m, s, st, s.xl);
  q_barrier();
  VSTAMP(4);
  for (int e = threadIdx.x; e < ne; e += QNT) {
    const int r = e & 3, l = (e >> 2) & 63, rg = e >> 8;
    const int o = rg * 64 + (l >> 2) * 4 + r, sx = l & 3;
    if ((o >> 2) >= m.KQ2p) continue;
    const float v = q_segment_sum(plf, ne, e, nseg1);
    float pre = m.td ? fma_(w1t[o], a.t, v) : v;
    pre = pre + b1[o];
    const float h = act_apply(m.act, pre);
    const float da = act_deriv_c(m.act, pre, h);
    dact[e] = da;
    if (sx < nvalid && o < m.Hp) {
      float* hp_ = a.hsc + (size_t)(b0 + sx) * m.Hp + o; if constexpr (wt) stwt(hp_, h); else *hp_ = h;
      if (a.dact_out) a.dact_out[(size_t)(b0 + sx) * m.Hp + o] = da;
    }
  }
  }
  if (ovl) {
    // everything above needed the record only.  Now the producer (launch id - 1: the previous stage) must have stored its
    // K and lambda: one thread waits for its arrivals, then the lambda tile is formed (phase 0's other half)
    if (threadIdx.x == 0) adj_spin(sync + 8 + ((a.sync_id - 1) & 7), nprod, sync + 1);
    __syncthreads();
    lambda_tile();
  }
  q_barrier();
  VSTAMP(5);
  // ---- phase 2: dh = W2^T lam ; dpre = dh .* act' -> h tile image + scratch ----
  vq_phase_ksplit<QSB1, KT>(m, s, st, ll);
  q_barrier();
  VSTAMP(6);
  for (int e = threadIdx.x; e < ne; e += QNT) {
    const int r = e & 3, l = (e >> 2) & 63, rg = e >> 8;
    const int o = rg * 64 + (l >> 2) * 4 + r, sx = l & 3;
    if ((o >> 2) >= m.KQ2p) continue;
    const float v = q_segment_sum(plf, ne, e, nseg1);
    const float dpre = (o < m.H) ? v * dact[e] : 0.f;
    hlf[((o >> 2) * 4 + sx) * 4 + r] = dpre;
    if (sx < nvalid && o < m.Hp) { float* dp_ = a.dpsc + (size_t)(b0 + sx) * m.Hp + o; if constexpr (wt) stwt(dp_, dpre); else *dp_ = dpre; }
  }
  q_barrier();
  VSTAMP(7);
  // ---- phase 3: dy = W1^T dpre (row groups wave and wave + QNW, one chain over K = H) ----
  {
    const f32x4* hp = s.hl + sidx;
    f32x4 acc0 = zero4, acc1 = zero4;
    static_for<0, QSB2>([&](auto Bc) {
      constexpr int B = decltype(Bc)::value;
      constexpr int SL = (2 * QSB1 + B) % VRING, NSL = (2 * QSB1 + B + 2) % VRING;
      f32x4 b_[QSQ];
#pragma unroll
      for (int j = 0; j < QSQ; ++j) b_[j] = hp[(B * QSQ + j) * 4];
      vq_block_mfma<2 * QSB1 + B + 2, SL, NSL, KT, (B == QSB2 - 1 ? KT : QSQ)>(st, b_, acc0, acc1);
    });
    const int g0 = wave, g1 = wave + QNW;
    if (sidx < nvalid) {
      float* dst = a.dy + (size_t)(b0 + sidx) * m.D + q * 4;
      if (g0 < m.RG2 && g0 * 64 + q * 4 < m.D) { if constexpr (wt) st4wt(dst + g0 * 64, acc0); else st4(dst + g0 * 64, acc0); }
      if (g1 < m.RG2 && g1 * 64 + q * 4 < m.D) { if constexpr (wt) st4wt(dst + g1 * 64, acc1); else st4(dst + g1 * 64, acc1); }
    }
  }
  VSTAMP(8);
  if constexpr (wt) adj_arrive(sync, a.sync_id);   // this workgroup's K, lambda and scratch are out: the next stage may read them
}

#ifndef LRNDE_OVL_TILE_CC
#define LRNDE_OVL_TILE_CC false
#endif
template <int KT, bool SYNC = false> __global__ __launch_bounds__(QNT) void k_vjp_q(VjpQArgs a) { vjp_q_body<KT, SYNC>(a); }

// The VJP of one adjoint RHS evaluation and, on the CUs it leaves idle (it has B/4 workgroups: 128 at B = 512), the
// parameter-gradient GEMM of the PREVIOUS evaluation (its tiles are workgroups nvjp, nvjp+1, ...).  The two touch
// disjoint buffers: the scratch (y, h, dpre) and the stage lambda are double buffered by the host (launch_vjp).
#ifdef LRNDE_STAMPS
__device__ unsigned long long g_wgstamps[1024][2];   // tools/vjp_probe: start / end (100-MHz clock) of every workgroup of the stamped stage launch
struct WgStamp {
  bool on; unsigned long long t0;
  __device__ WgStamp(bool o) : on(o && threadIdx.x == 0 && blockIdx.x < 1024), t0(__builtin_amdgcn_s_memrealtime()) {}
  __device__ ~WgStamp() { if (on) { g_wgstamps[blockIdx.x][0] = t0; g_wgstamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime(); } }
};
#endif
template <int KT, bool SYNC = false, bool REUSE = false> __global__ __launch_bounds__(QNT) void k_vjp_q_pg(VjpQArgs a, PgradArgs pg, int nvjp) {
#ifdef LRNDE_STAMPS
  WgStamp wgs(a.adj_mode == ADJ_STAGE && a.adj_stage == LRNDE_STAMP_STAGE);
#endif
  if ((int)blockIdx.x >= nvjp) {
    constexpr bool synced = SYNC;
    if (synced) {   // the attempt's control block may not be published yet (its stage-2 launch runs beside this one)
      if (threadIdx.x == 0) adj_spin(a.adj.sync, a.adj.seq0 + a.adj_j + 1, a.adj.sync + 1);
      __syncthreads();
    }
    if (!pgrad_resolve(pg, a.adj)) return;
    if (synced) {
      // the scratch set and lambda these tiles read belong to launch id - 1, which may still be running elsewhere on the chip
      if (threadIdx.x == 0) adj_spin(a.adj.sync + 8 + ((a.sync_id - 1) & 7), nvjp, a.adj.sync + 1);
      __syncthreads();
      pgrad_tile_any<LRNDE_OVL_TILE_CC, true>(pg, (int)blockIdx.x - nvjp);
      return;
    }
    pgrad_tile_any<false, true>(pg, (int)blockIdx.x - nvjp);
    return;
  }
  vjp_q_body<KT, SYNC, REUSE>(a);
}
