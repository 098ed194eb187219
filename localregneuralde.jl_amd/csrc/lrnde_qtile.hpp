// lrnde_qtile.hpp — the small-batch kernel family: 4 batch columns per workgroup.
// Included by lrnde_kernels.hip inside its anonymous namespace (shares StepArgs, Ctrl, the device
// prologue, the partial-sum protocol and the canonical arithmetic).
//
// Why a second tile shape: the vector field is per-sample independent, so the batch is the only
// inter-workgroup parallelism.  With 16 columns per workgroup MNIST-ODE B=512 gives 32 workgroups
// on a 256-CU chip.  v_mfma_f32_4x4x1_16B_f32 multiplies 16 independent 4x4 blocks per
// instruction; mapping block b to rows 4b..4b+3 of a 64-row group and the 4 columns to 4 samples
// gives a 64x4 output tile per instruction with K = 1 — still one fp32 fma per product in
// increasing k (measured: tools/mfma_probe.hip), i.e. the same canonical dot product, on a
// 4-column tile: 128 workgroups at B=512.  A workgroup is 4 waves, ONE per SIMD (512 VGPRs
// each): weights stream from L2 with buffer loads one block ahead, the first block of each
// GEMM phase stays resident in registers for the whole launch.
//
// Lane roles (lane l): sample s = l & 3, row quad q = l >> 2.  A operand: W[row 64*rg + l][k];
// B operand: x[k][s]; D register r: out[row 64*rg + 4q + r][s].

constexpr int QNB = 4;    // batch columns per workgroup
constexpr int QNW = 8;    // waves per workgroup
constexpr int QNT = QNW * 64;
constexpr int QB1 = 7;    // k-quads per Dense-1 block (a canonical segment = 28 quads = 4 blocks)
constexpr int QSEG = 28;  // k-quads per canonical segment (112 rows)
constexpr int QB2 = 5;    // k-quads per Dense-2 block
constexpr int QRGC = 4;   // Dense-2 row groups run concurrently by one wave

struct SmemQ {
  f32x4* xl;   // x tile: [KQ1p][4 samples] quads (4 consecutive rows of one sample)
  f32x4* hl;   // h tile: [KQ2p][4]
  f32x4* pl;   // Dense-1 segment partials [nseg1][RG1][64 lanes]
  float* bias; // w1t[64*RG1] b1[64*RG1] w2t[64*RG2] b2[64*RG2]
  double* red;
  Bcast* bc;
};

__device__ __forceinline__ int q_nseg1(const ModelDev& m) { return (m.KQ1p + QSEG - 1) / QSEG; }

__device__ __forceinline__ SmemQ carve_q(const ModelDev& m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SmemQ s;
  s.xl = reinterpret_cast<f32x4*>(smem);
  s.hl = s.xl + (size_t)m.KQ1p * 4;
  s.pl = s.hl + (size_t)m.KQ2p * 4;
  s.bias = reinterpret_cast<float*>(s.pl + (size_t)q_nseg1(m) * m.RG1 * 64);
  s.red = reinterpret_cast<double*>(s.bias + 128 * (size_t)(m.RG1 + m.RG2));
  s.bc = reinterpret_cast<Bcast*>(s.red + QNW * 3);
  return s;
}
static size_t smem_bytes_q(int KQ1p, int KQ2p, int RG1, int RG2) {
  const size_t nseg1 = (size_t)((KQ1p + QSEG - 1) / QSEG);
  return ((size_t)KQ1p * 4 + (size_t)KQ2p * 4 + nseg1 * RG1 * 64) * 16 + 128 * (size_t)(RG1 + RG2) * 4 +
         QNW * 3 * sizeof(double) + sizeof(Bcast) + 16;
}
__device__ __forceinline__ void smem_init_q(const ModelDev& m, const SmemQ& s) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < (m.KQ1p + m.KQ2p) * 4; i += QNT) s.xl[i] = z;  // xl and hl are adjacent
  const int h64 = m.RG1 * 64, d64 = m.RG2 * 64;
  for (int i = threadIdx.x; i < h64; i += QNT) {
    s.bias[i] = (i < m.Hp) ? m.w1t[i] : 0.f;
    s.bias[h64 + i] = (i < m.Hp) ? m.b1[i] : 0.f;
  }
  for (int i = threadIdx.x; i < d64; i += QNT) {
    s.bias[2 * h64 + i] = (i < m.Dp) ? m.w2t[i] : 0.f;
    s.bias[2 * h64 + d64 + i] = (i < m.Dp) ? m.b2[i] : 0.f;
  }
}

// ---- per-lane access to the state workspace for this tile shape ----------------------------
struct TileIOQ {
  __amdgpu_buffer_rsrc_t rs;
  int voff;  // ((b0 + s) * D + 4q) * 4, or out of range for columns beyond the batch
  int row_limit_bytes;  // D*4: rows at/after it (last row group) are masked
  int q4;    // 16 * q: byte offset of this lane's row quad inside a row group
};
__device__ __forceinline__ TileIOQ make_tile_io_q(const StepArgs& a, int b0, int nvalid) {
  const int lane = threadIdx.x & 63, sidx = lane & 3, q = lane >> 2;
  TileIOQ io;
  io.rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.state, 0, (int)(a.n_local * 40), 0x00020000);
  io.voff = (sidx < nvalid) ? ((b0 + sidx) * a.m.D + q * 4) * 4 : 0x7ffffff0;
  io.row_limit_bytes = a.m.D * 4;
  io.q4 = q * 16;
  return io;
}
// offset of row group rg for this lane (out of range when the lane's rows are beyond D)
__device__ __forceinline__ int q_voff(const TileIOQ& io, int rg) {
  return (rg * 256 + io.q4 < io.row_limit_bytes) ? io.voff + rg * 256 : 0x7ffffff0;
}
__device__ __forceinline__ f32x4 qload(const TileIOQ& io, int voff_rg, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(io.rs, voff_rg, soff, 0));
}
__device__ __forceinline__ void qstore(const TileIOQ& io, int voff_rg, int soff, const f32x4& v) {
  // literal soffset: see the store-data hazard note at sstore()
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), io.rs, voff_rg + soff, 0, 0);
}

// ---- Dense-2 epilogue policies (same contract as the 16-column ones; tile index = row group) ----
struct EpiStoreKQ {
  static constexpr int NPRE = 1;
  const ModelDev* m; float* kout; int b0, nvalid;
  __device__ __forceinline__ void pre(int, f32x4 (&)[NPRE]) const {}
  __device__ __forceinline__ void post(int rg, const f32x4& kv, f32x4 (&)[NPRE]) const {
    const int lane = threadIdx.x & 63, sidx = lane & 3, q = lane >> 2;
    const int row0 = rg * 64 + q * 4;
    if (sidx < nvalid && row0 < m->D) *reinterpret_cast<f32x4*>(kout + (size_t)(b0 + sidx) * m->D + row0) = kv;
  }
};

template <int S> struct EpiStageQ {
  static constexpr int NPRE = S;
  TileIOQ io;
  int off_up, off_k[6], off_out, off_x;
  float dt;
  f32x4* xl; int KQ1;
  __device__ __forceinline__ void pre(int rg, f32x4 (&pb)[NPRE]) const {
    const int vo = q_voff(io, rg);
    pb[0] = qload(io, vo, off_up);
#pragma unroll
    for (int j = 0; j < S - 1; ++j) pb[1 + j] = qload(io, vo, off_k[j]);
  }
  __device__ __forceinline__ void post(int rg, const f32x4& kv, f32x4 (&pb)[NPRE]) const {
    const int lane = threadIdx.x & 63, sidx = lane & 3, q = lane >> 2;
    constexpr int off = (S - 1) * S / 2;
    const int vo = q_voff(io, rg);
    qstore(io, vo, off_out, kv);
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
    if (off_x >= 0) qstore(io, vo, off_x, x);
    const int kq = rg * 16 + q;
    if (kq < KQ1) xl[kq * 4 + sidx] = x;  // one quad = the B operand of 4 Dense-1 k-steps
  }
};

struct EpiFinalQ {
  static constexpr int NPRE = 9;
  TileIOQ io;
  int off_up, off_u, off_k[6], off_g6, off_out;
  float dt, abstol, reltol;
  int want_stiff, nvalid, D;
  double *aerr, *anum, *aden;
  __device__ __forceinline__ void pre(int rg, f32x4 (&pb)[NPRE]) const {
    const int vo = q_voff(io, rg);
    pb[0] = qload(io, vo, off_up);
    pb[1] = qload(io, vo, off_u);
#pragma unroll
    for (int j = 0; j < 6; ++j) pb[2 + j] = qload(io, vo, off_k[j]);
    if (want_stiff) pb[8] = qload(io, vo, off_g6);
  }
  __device__ __forceinline__ void post(int rg, const f32x4& kv, f32x4 (&pb)[NPRE]) const {
    const int lane = threadIdx.x & 63, sidx = lane & 3, q = lane >> 2;
    const int vo = q_voff(io, rg);
    qstore(io, vo, off_out, kv);
    if (sidx >= nvalid || rg * 64 + q * 4 >= D) return;
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

// ---- weights resident in registers for the whole launch ------------------------------------
constexpr int QRG2 = 2;  // Dense-2 row groups per pass (chains sharing the B operand)

struct FevalCtxQ {
  __amdgpu_buffer_rsrc_t rs1, rs2;
  f32x4 r1[QB1][2];  // Dense 1: first block of (segment = wave, row groups 0,1), kept all launch
};

// W1q: [RG1][KQ1p][64][4]   element (rg,kq,l,j) = W1[row 64rg+l][k 4kq+j]
// W2q: [RG2][KQ2p][64][4]
// Loads for row groups beyond RG1 / RG2 are out of the descriptor's range: they return 0 without
// touching memory, so the unrolled slots need no guards.
__device__ __forceinline__ int q_w1_off(const ModelDev& m, int rg, int kq) { return (rg * m.KQ1p + kq) * 1024; }
__device__ __forceinline__ int q_w2_off(const ModelDev& m, int rg, int kq) { return (rg * m.KQ2p + kq) * 1024; }

__device__ __forceinline__ void feval_ctx_init_q(const ModelDev& m, FevalCtxQ& fc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  fc.rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)m.W1q, 0, m.RG1 * m.KQ1p * 1024, 0x00020000);
  fc.rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)m.W2q, 0, m.RG2 * m.KQ2p * 1024, 0x00020000);
  const int voff = lane * 16;
  const int kq0 = (wave < q_nseg1(m)) ? wave * QSEG : 0;
#pragma unroll
  for (int j = 0; j < QB1; ++j)
#pragma unroll
    for (int c = 0; c < 2; ++c) fc.r1[j][c] = wload(fc.rs1, voff, q_w1_off(m, c, kq0 + j));
}

// one vector-field evaluation on the 4-column tile
template <class Epi>
__device__ __forceinline__ void feval_q(const ModelDev& m, const SmemQ& sm, const FevalCtxQ& fc, float ts,
                                        const Epi& epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sidx = lane & 3, q = lane >> 2;
  const int voff = lane * 16;
  // weight rows beyond the real matrix are all-zero padding: give those lanes an out-of-range
  // offset so the row group's tail is never fetched (H = 100: 28 of 128 Dense-1 rows)
  auto voff1 = [&](int rg) { return (rg * 64 + lane < m.H) ? voff : 0x7ffffff0; };
  auto voff2 = [&](int rg) { return (rg * 64 + lane < m.D) ? voff : 0x7ffffff0; };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int h64 = m.RG1 * 64, d64 = m.RG2 * 64;
  const float* w1t = sm.bias; const float* b1 = w1t + h64;
  const float* w2t = b1 + h64; const float* b2 = w2t + d64;
  const int nseg1 = q_nseg1(m);
  // Dense-2 work of this wave: passes of QRG2 row groups {g, g+QNW}: g = wave + 2*QNW*pass
  const int nblk2 = m.KQ2p / QB2;
  const int npass2 = (m.RG2 > wave) ? (m.RG2 - wave + 2 * QNW - 1) / (2 * QNW) : 0;
  const int nitem2 = npass2 * nblk2;
  f32x4 a2X[QB2][QRG2], a2Y[QB2][QRG2];
#define LRNDE_QLOADA2(a, it)                                                                    \
  do {                                                                                          \
    const int g_ = wave + 2 * QNW * ((it) / nblk2), k0_ = ((it) % nblk2) * QB2;                 \
    const int v0_ = voff2(g_), v1_ = voff2(g_ + QNW);                                           \
    _Pragma("unroll") for (int j = 0; j < QB2; ++j) {                                           \
      a[j][0] = wload(fc.rs2, v0_, q_w2_off(m, g_, k0_ + j));                                   \
      a[j][1] = wload(fc.rs2, v1_, q_w2_off(m, g_ + QNW, k0_ + j));                             \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
  STAMP(1); STAMPW(0);
  // ---- Dense 1: waves own canonical segments w, w+QNW, ...; two row groups run as independent
  // chains sharing the B operand; weights one block (7 k-quads) ahead ----
  {
    const f32x4* xp = sm.xl + sidx;  // x[k-quad][sample]
    for (int seg = wave; seg < nseg1; seg += QNW) {
      const int kq_lo = seg * QSEG, kq_hi = min(m.KQ1p, kq_lo + QSEG);
      const int nblk = (kq_hi - kq_lo) / QB1;
      for (int rg0 = 0; rg0 < m.RG1; rg0 += 2) {
        f32x4 acc0 = zero4, acc1 = zero4;
        f32x4 aX[QB1][2], aY[QB1][2];
#define LRNDE_QLOAD1(a, blk)                                                                    \
  do {                                                                                          \
    const int v0_ = voff1(rg0), v1_ = voff1(rg0 + 1);                                           \
    _Pragma("unroll") for (int j = 0; j < QB1; ++j) {                                           \
      a[j][0] = wload(fc.rs1, v0_, q_w1_off(m, rg0, kq_lo + (blk) * QB1 + j));                  \
      a[j][1] = wload(fc.rs1, v1_, q_w1_off(m, rg0 + 1, kq_lo + (blk) * QB1 + j));              \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
#define LRNDE_QMMA1(a, blk)                                                                     \
  do {                                                                                          \
    f32x4 b_[QB1];                                                                              \
    _Pragma("unroll") for (int j = 0; j < QB1; ++j) b_[j] = xp[(kq_lo + (blk) * QB1 + j) * 4];  \
    _Pragma("unroll") for (int j = 0; j < QB1; ++j) {                                           \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][0].x, b_[j].x, acc0, 0, 0, 0);             \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][1].x, b_[j].x, acc1, 0, 0, 0);             \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][0].y, b_[j].y, acc0, 0, 0, 0);             \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][1].y, b_[j].y, acc1, 0, 0, 0);             \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][0].z, b_[j].z, acc0, 0, 0, 0);             \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][1].z, b_[j].z, acc1, 0, 0, 0);             \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][0].w, b_[j].w, acc0, 0, 0, 0);             \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][1].w, b_[j].w, acc1, 0, 0, 0);             \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
        int blk = 0;
        if (seg == wave && rg0 == 0) {  // first block from the resident registers
          if (nblk > 1) {
            LRNDE_QLOAD1(aX, 1);
            LRNDE_QMMA1(fc.r1, 0);
            blk = 1;
          } else {
            LRNDE_QMMA1(fc.r1, 0);
            blk = nblk;
          }
        } else {
          LRNDE_QLOAD1(aX, 0);
        }
#pragma unroll 1
        for (; blk + 2 < nblk; blk += 2) {
          LRNDE_QLOAD1(aY, blk + 1);
          LRNDE_QMMA1(aX, blk);
          LRNDE_QLOAD1(aX, blk + 2);
          LRNDE_QMMA1(aY, blk + 1);
        }
        if (blk + 1 < nblk) {
          LRNDE_QLOAD1(aY, blk + 1);
          LRNDE_QMMA1(aX, blk);
          LRNDE_QMMA1(aY, blk + 1);
        } else if (blk < nblk) {
          LRNDE_QMMA1(aX, blk);
        }
#undef LRNDE_QLOAD1
#undef LRNDE_QMMA1
        f32x4* pp = sm.pl + ((size_t)seg * m.RG1 + rg0) * 64 + lane;
        pp[0] = acc0;
        if (rg0 + 1 < m.RG1) pp[64] = acc1;
      }
    }
  }
  // the first Dense-2 weight block does not depend on h: in flight across epilogue 1
  if (nitem2 > 0) LRNDE_QLOADA2(a2X, 0);
  STAMP(2); STAMPW(1);
  __syncthreads();
  STAMP(3);
  // epilogue 1: segment partials in order, time column, bias, activation -> h tile
  // (one C-fragment element per thread: element e = (rg*64 + l)*4 + r -> row 64rg + 4(l>>2) + r)
  {
    const float* plf = reinterpret_cast<const float*>(sm.pl);
    float* hlf = reinterpret_cast<float*>(sm.hl);
    const int ne = m.RG1 * 256;
    for (int e = threadIdx.x; e < ne; e += QNT) {
      const int r = e & 3, l = (e >> 2) & 63, rg = e >> 8;
      const int o = rg * 64 + (l >> 2) * 4 + r;
      if ((o >> 2) >= m.KQ2p) continue;
      float v = plf[e];
      for (int sgi = 1; sgi < nseg1; ++sgi) v = v + plf[(size_t)sgi * ne + e];
      float pre = m.td ? fma_(w1t[o], ts, v) : v;
      pre = pre + b1[o];
      hlf[((o >> 2) * 4 + (l & 3)) * 4 + r] = act_apply(m.act, pre);
    }
  }
  STAMPW(2);
  __syncthreads();
  STAMP(4); STAMPW(3);
  // ---- Dense 2 (H <= 112, host-checked: one canonical segment = one chain over all KQ2p
  // k-quads).  Wave w runs passes of two row groups {g, g+4}, g = w + 8*pass; the block loop is
  // flattened over passes so that the next block (also the next pass's first) is always in flight.
  {
    const f32x4* hp = sm.hl + sidx;
    auto finish = [&](int rg, const f32x4& tot) {
      const int row0 = rg * 64 + q * 4;
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
    f32x4 acc0 = zero4, acc1 = zero4;
#define LRNDE_QMMA2(a, it)                                                                      \
  do {                                                                                          \
    const int blk_ = (it) % nblk2, g_ = wave + 2 * QNW * ((it) / nblk2);                        \
    f32x4 b_[QB2];                                                                              \
    _Pragma("unroll") for (int j = 0; j < QB2; ++j) b_[j] = hp[(blk_ * QB2 + j) * 4];           \
    _Pragma("unroll") for (int j = 0; j < QB2; ++j) {                                           \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][0].x, b_[j].x, acc0, 0, 0, 0);             \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][1].x, b_[j].x, acc1, 0, 0, 0);             \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][0].y, b_[j].y, acc0, 0, 0, 0);             \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][1].y, b_[j].y, acc1, 0, 0, 0);             \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][0].z, b_[j].z, acc0, 0, 0, 0);             \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][1].z, b_[j].z, acc1, 0, 0, 0);             \
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][0].w, b_[j].w, acc0, 0, 0, 0);             \
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][1].w, b_[j].w, acc1, 0, 0, 0);             \
    }                                                                                           \
    if (blk_ == nblk2 - 1) { /* pass complete: epilogue of its two row groups */                \
      f32x4 pb0[Epi::NPRE], pb1[Epi::NPRE];                                                     \
      epi.pre(g_, pb0);                                                                         \
      if (g_ + QNW < m.RG2) epi.pre(g_ + QNW, pb1);                                             \
      epi.post(g_, finish(g_, acc0), pb0);                                                      \
      if (g_ + QNW < m.RG2) epi.post(g_ + QNW, finish(g_ + QNW, acc1), pb1);                    \
      acc0 = zero4; acc1 = zero4;                                                               \
    }                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
    if (nitem2 > 0) {
      int it = 0;
#pragma unroll 1
      for (; it + 2 < nitem2; it += 2) {
        LRNDE_QLOADA2(a2Y, it + 1);
        LRNDE_QMMA2(a2X, it);
        LRNDE_QLOADA2(a2X, it + 2);
        LRNDE_QMMA2(a2Y, it + 1);
      }
      if (it + 1 < nitem2) {
        LRNDE_QLOADA2(a2Y, it + 1);
        LRNDE_QMMA2(a2X, it);
        LRNDE_QMMA2(a2Y, it + 1);
      } else {
        LRNDE_QMMA2(a2X, it);
      }
    }
#undef LRNDE_QMMA2
  }
#undef LRNDE_QLOADA2
  STAMP(5); STAMPW(4);
  __syncthreads();
  STAMP(6);
}

template <class F>
__device__ __forceinline__ void q_tile_foreach(const ModelDev& m, int b0, int nvalid, int KQ1, F&& fn) {
  // quads of the tile: sample fastest (4), then k-quad: 256 B contiguous per sample per wave
  for (int i = threadIdx.x; i < KQ1 * 4; i += QNT) {
    const int sidx = i & 3, kq = i >> 2;
    fn(kq, sidx, sidx < nvalid, (size_t)(b0 + sidx) * m.D + kq * 4);
  }
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ void block_sum3_q(double* red, double& a, double& b, double& c) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  if (lane == 0) { red[wave * 3 + 0] = a; red[wave * 3 + 1] = b; red[wave * 3 + 2] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0, sc = 0.0;
    for (int w = 0; w < QNW; ++w) { sa += red[w * 3]; sb += red[w * 3 + 1]; sc += red[w * 3 + 2]; }
    a = sa; b = sb; c = sc;
  }
  __syncthreads();
}

__device__ __forceinline__ void q_feval_store(const ModelDev& m, const SmemQ& sm, const FevalCtxQ& fc, float ts,
                                              float* kout, int b0, int nvalid) {
  EpiStoreKQ e;
  e.m = &m; e.kout = kout; e.b0 = b0; e.nvalid = nvalid;
  feval_q<EpiStoreKQ>(m, sm, fc, ts, e);
}

__global__ __launch_bounds__(QNT) void k_rhs_q(StepArgs a, const float* u, float t, float* du) {
  STAMP(0);
  const SmemQ s = carve_q(a.m);
  smem_init_q(a.m, s);
  FevalCtxQ fc;
  feval_ctx_init_q(a.m, fc);
  const int b0 = blockIdx.x * QNB, nvalid = min(QNB, a.B - b0);
  const int KQ1 = a.m.D / 4;
  __syncthreads();
  q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int kq, int sidx, bool valid, size_t g) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    s.xl[kq * 4 + sidx] = valid ? ld4(u + g) : z;
  });
  __syncthreads();
  q_feval_store(a.m, s, fc, t, du, b0, nvalid);
}

__global__ __launch_bounds__(QNT) void k_init1_q(StepArgs a) {
  const SmemQ s = carve_q(a.m);
  smem_init_q(a.m, s);
  FevalCtxQ fc;
  feval_ctx_init_q(a.m, fc);
  const int b0 = blockIdx.x * QNB, nvalid = min(QNB, a.B - b0);
  const int KQ1 = a.m.D / 4;
  const Ctrl c = a.ctrl[0];
  const float* u0 = a.ubuf[c.cur];
  float* f0 = a.kfsal[c.cur];
  __syncthreads();
  q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int kq, int sidx, bool valid, size_t g) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    s.xl[kq * 4 + sidx] = valid ? ld4(u0 + g) : z;
  });
  __syncthreads();
  q_feval_store(a.m, s, fc, c.t, f0, b0, nvalid);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int, int, bool valid, size_t g) {
    if (!valid) return;
    const f32x4 u = ld4(u0 + g), f = ld4(f0 + g);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float sk = a.abstol + __builtin_fabsf(u[h]) * a.reltol;
      const float r0 = u[h] / sk, r1 = f[h] / sk;
      const float q0 = r0 * r0, q1 = r1 * r1;
      a0 += (double)q0; a1 += (double)q1;
    }
  });
  block_sum3_q(s.red, a0, a1, a2);
  if (threadIdx.x == 0) {
    double* p = a.pinit_send + (size_t)(a.wg_offset + blockIdx.x) * PSTRIDE;
    p[0] = a0; p[1] = a1; p[2] = 0.0;
  }
}

__global__ __launch_bounds__(QNT) void k_init2_q(StepArgs a) {
  const SmemQ s = carve_q(a.m);
  smem_init_q(a.m, s);
  FevalCtxQ fc;
  feval_ctx_init_q(a.m, fc);
  const int b0 = blockIdx.x * QNB, nvalid = min(QNB, a.B - b0);
  const int KQ1 = a.m.D / 4;
  const Ctrl c = a.ctrl[0];
  if (threadIdx.x < 64) {
    double s1[3];
    reduce_partials(a.pinit_recv, a.nwg_global, s1);
    if (threadIdx.x == 0) s.bc->dt0 = init_dt0(s1, a.n_global, a.t1 - a.t0);
  }
  __syncthreads();
  const float dt0 = s.bc->dt0;
  const float* u0 = a.ubuf[c.cur];
  const float* f0 = a.kfsal[c.cur];
  float* f1 = a.ks[0];
  q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int kq, int sidx, bool valid, size_t g) {
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      const f32x4 u = ld4(u0 + g), f = ld4(f0 + g);
#pragma unroll
      for (int h = 0; h < 4; ++h) x[h] = u[h] + dt0 * f[h];
    }
    s.xl[kq * 4 + sidx] = x;
  });
  __syncthreads();
  q_feval_store(a.m, s, fc, c.t + dt0, f1, b0, nvalid);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int, int, bool valid, size_t g) {
    if (!valid) return;
    const f32x4 u = ld4(u0 + g), f = ld4(f0 + g), ff = ld4(f1 + g);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float sk = a.abstol + __builtin_fabsf(u[h]) * a.reltol;
      const float r2 = (ff[h] - f[h]) / sk;
      const float q2 = r2 * r2;
      a0 += (double)q2;
    }
  });
  block_sum3_q(s.red, a0, a1, a2);
  if (threadIdx.x == 0) {
    double* p = a.pinit_send + (size_t)(a.nwg_global + a.wg_offset + blockIdx.x) * PSTRIDE;
    p[0] = a0; p[1] = 0.0; p[2] = 0.0;
  }
}

// one attempted Tsit5 step, 4 columns per workgroup (same flow as k_step's fused path)
template <bool SPEC> __global__ __launch_bounds__(QNT) void k_step_q(StepArgs a, int j) {
  STAMP(9);
  const SmemQ s = carve_q(a.m);
  smem_init_q(a.m, s);
  FevalCtxQ fc;
  feval_ctx_init_q(a.m, fc);
  const int b0 = blockIdx.x * QNB, nvalid = min(QNB, a.B - b0);
  const int KQ1 = a.m.D / 4;
  STAMP(10);
  if (threadIdx.x < 64) step_prologue(a, j, s.bc);
  __syncthreads();
  STAMP(11);
  const Bcast bc = *s.bc;

  if (bc.accepted_prev) {  // savevalues! of the step accepted by the prologue
    const float* up = a.ubuf[bc.cur_prev];
    const float* un = a.ubuf[bc.cur_prev ^ 1];
    const float* k1 = a.kfsal[bc.cur_prev];
    const float* k7 = a.kfsal[bc.cur_prev ^ 1];
    int slot = bc.nsaved0;
    for (int is = bc.isave0; is < a.nsave && a.saveat[is] <= bc.t_new; ++is, ++slot) {
      const float ts = a.saveat[is];
      float* dst = a.u_saved + (size_t)slot * a.B * a.m.D;
      if (ts != bc.t_new) {
        const float theta = (ts - bc.tprev) / bc.dt_prev;
        float bw[7];
        tsit5_bweights(theta, bw);
        q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int, int, bool valid, size_t g) {
          if (!valid) return;
          const f32x4 y0 = ld4(up + g), v1 = ld4(k1 + g), v2 = ld4(a.ks[0] + g), v3 = ld4(a.ks[1] + g),
                      v4 = ld4(a.ks[2] + g), v5 = ld4(a.ks[3] + g), v6 = ld4(a.ks[4] + g), v7 = ld4(k7 + g);
          f32x4 o;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            float sum = v1[h] * bw[0] + v2[h] * bw[1];
            sum = sum + v3[h] * bw[2];
            sum = sum + v4[h] * bw[3];
            sum = sum + v5[h] * bw[4];
            sum = sum + v6[h] * bw[5];
            sum = sum + v7[h] * bw[6];
            o[h] = y0[h] + bc.dt_prev * sum;
          }
          st4(dst + g, o);
        });
      } else {
        q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int, int, bool valid, size_t g) {
          if (valid) st4(dst + g, ld4(un + g));
        });
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) a.t_saved[slot] = ts;
    }
    if (a.save_everystep) {
      float* dst = a.u_saved + (size_t)slot * a.B * a.m.D;
      q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int, int, bool valid, size_t g) {
        if (valid) st4(dst + g, ld4(un + g));
      });
      if (blockIdx.x == 0 && threadIdx.x == 0) a.t_saved[slot] = bc.t_new;
    }
  }
  if (!bc.do_step) return;

  const float t = bc.t, dt = bc.dt;
  const float* uprev = a.ubuf[bc.cur];
  const float* k1 = a.kfsal[bc.cur];
  const float c1 = (float)Tsit5::C[0], c2 = (float)Tsit5::C[1], c3 = (float)Tsit5::C[2],
              c4 = (float)Tsit5::C[3];
  double aerr = 0.0, anum = 0.0, aden = 0.0;
  const TileIOQ io = make_tile_io_q(a, b0, nvalid);
  const int o_up = arr_off(a, bc.cur), o_un = arr_off(a, bc.cur ^ 1);
  const int o_k1 = arr_off(a, 2 + bc.cur), o_k7 = arr_off(a, 2 + (bc.cur ^ 1));
  const int o_g6 = arr_off(a, 9);
  __syncthreads();  // smem_init_q is complete before the x tile is written
  {  // x2 = uprev + (dt*a21)*k1   (src/perform_step.jl:11-12)
    const float a21dt = dt * (float)Tsit5::A[0];
    q_tile_foreach(a.m, b0, nvalid, KQ1, [&](int kq, int sidx, bool valid, size_t g) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        const f32x4 u = ld4(uprev + g), f = ld4(k1 + g);
#pragma unroll
        for (int h = 0; h < 4; ++h) x[h] = u[h] + a21dt * f[h];
      }
      s.xl[kq * 4 + sidx] = x;
    });
  }
  __syncthreads();
  STAMP(12);
#define LRNDE_QSTAGE(S, TS)                                                             \
  do {                                                                                  \
    EpiStageQ<S> e;                                                                     \
    e.io = io; e.off_up = o_up; e.off_k[0] = o_k1;                                      \
    _Pragma("unroll") for (int qq = 0; qq < 5; ++qq) e.off_k[1 + qq] = arr_off(a, 4 + qq); \
    e.off_out = arr_off(a, 4 + (S - 2));                                                \
    e.off_x = (S == 6) ? o_un : ((S == 5 && a.want_stiff) ? o_g6 : -1);                 \
    e.dt = dt; e.xl = s.xl; e.KQ1 = KQ1;                                                \
    feval_q<EpiStageQ<S>>(a.m, s, fc, (TS), e);                                         \
    STAMP(11 + S);                                                                      \
  } while (0)
  LRNDE_QSTAGE(2, t + c1 * dt);
  LRNDE_QSTAGE(3, t + c2 * dt);
  LRNDE_QSTAGE(4, t + c3 * dt);
  LRNDE_QSTAGE(5, t + c4 * dt);
  LRNDE_QSTAGE(6, t + dt);
#undef LRNDE_QSTAGE
  EpiFinalQ ef;
  ef.io = io; ef.off_up = o_up; ef.off_u = o_un; ef.off_k[0] = o_k1;
#pragma unroll
  for (int qq = 0; qq < 5; ++qq) ef.off_k[1 + qq] = arr_off(a, 4 + qq);
  ef.off_g6 = o_g6; ef.off_out = o_k7;
  ef.dt = dt; ef.abstol = a.abstol; ef.reltol = a.reltol; ef.want_stiff = a.want_stiff; ef.nvalid = nvalid;
  ef.D = a.m.D;
  ef.aerr = &aerr; ef.anum = &anum; ef.aden = &aden;
  feval_q<EpiFinalQ>(a.m, s, fc, t + dt, ef);
  STAMP(18);
  block_sum3_q(s.red, aerr, anum, aden);
  STAMP(19);
  if (threadIdx.x == 0) {
    double* p = a.part_send + ((size_t)((j + 1) & 1) * a.nwg_global + a.wg_offset + blockIdx.x) * PSTRIDE;
    p[0] = aerr; p[1] = anum; p[2] = aden;
  }
}

// flat Lux parameter vector -> quad-tile A layouts (zero padded in k; row groups NOT padded)
__global__ void k_pack_q(const float* p, int D, int H, int td, int KQ1p, int KQ2p, int RG1, int RG2,
                         float* W1q, float* W2q) {
  const size_t n1 = (size_t)RG1 * KQ1p * 256, n2 = (size_t)RG2 * KQ2p * 256;
  const size_t base2 = (size_t)H * (D + td) + H;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int jj = i & 3, l = (i >> 2) & 63;
      const size_t blk = i >> 8;
      const int kq = blk % KQ1p, rg = blk / KQ1p;
      const int o = rg * 64 + l, k = kq * 4 + jj;
      W1q[i] = (o < H && k < D) ? p[(size_t)o + (size_t)H * k] : 0.f;
    } else {
      const size_t e = i - n1;
      const int jj = e & 3, l = (e >> 2) & 63;
      const size_t blk = e >> 8;
      const int kq = blk % KQ2p, rg = blk / KQ2p;
      const int o = rg * 64 + l, k = kq * 4 + jj;
      W2q[e] = (o < D && k < H) ? p[base2 + (size_t)o + (size_t)D * k] : 0.f;
    }
  }
}
