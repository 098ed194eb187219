// lrnde_regseed.hpp — included inside the anonymous namespace of both translation units.
// cotangent seeds of the regulariser's reverse sweep (src/perform_step.jl:34-47): kbar_2..7, ubar, g6bar
struct RegSeedArgs {
  size_t n;       // local elements
  size_t n_norm;  // elements of the (global) norm the regularisation value was taken over
  const float *uprev, *u, *g6;
  const float* k[7];
  float* kb[7];  // kb[1..6] <-> k2..k7
  float *ub, *g6b;
  float dt, abstol, reltol, eest, num, den;
  int reg_type;
};
__global__ void k_reg_seed(RegSeedArgs a) {
  const float nf = (float)a.n_norm;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    if (a.reg_type == 0) {  // reg = EEst*dt, EEst = sqrt(mean(r^2)), r = utilde / sc
      float sum = (float)Tsit5::BT[0] * a.k[0][i] + (float)Tsit5::BT[1] * a.k[1][i];
#pragma unroll
      for (int j = 2; j < 7; ++j) sum = sum + (float)Tsit5::BT[j] * a.k[j][i];
      const float ut = a.dt * sum;
      const float up = a.uprev[i], un = a.u[i];
      const float sc = a.abstol + fmaxf_(__builtin_fabsf(up), __builtin_fabsf(un)) * a.reltol;
      const float r = ut / sc;
      const float rb = (a.eest > 0.f) ? a.dt * r / (nf * a.eest) : 0.f;
      const float utb = rb / sc;
      const float scb = -rb * ut / (sc * sc);
      if (__builtin_fabsf(un) > __builtin_fabsf(up)) a.ub[i] += scb * a.reltol * (un >= 0.f ? 1.f : -1.f);
#pragma unroll
      for (int j = 1; j < 7; ++j) a.kb[j][i] += (float)Tsit5::BT[j] * (a.dt * utb);  // pullback of dt*(sum_j btilde_j k_j): through `dt *` first
    } else if (a.den != 0.f) {  // reg = |num/(den+eps)| / 3.5068
      const float eps = 1.1920929e-7f;
      const float qv = a.num / (a.den + eps);
      const float sgn = (qv >= 0.f ? 1.f : -1.f) / 3.5068f;
      const float numb = sgn / (a.den + eps), denb = -sgn * a.num / ((a.den + eps) * (a.den + eps));
      const float dk = a.k[6][i] - a.k[5][i], du = a.u[i] - a.g6[i];
      const float ca = (a.num > 0.f) ? numb * dk / (nf * a.num) : 0.f;
      const float cb = denb * du / (nf * a.den);
      a.kb[6][i] += ca; a.kb[5][i] -= ca;
      a.ub[i] += cb; a.g6b[i] -= cb;
    }
  }
}

