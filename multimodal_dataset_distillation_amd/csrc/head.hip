// Text ProjectionHead pointwise/normalisation pieces (reference networks.py:639-646: GELU,
// dropout-mask, residual, LayerNorm) and the contrastive head (reference distill.py:533,546-551:
// row L2-normalise, logits = scale * X Y^T, symmetric softmax cross-entropy), each with backward
// and -- via S = Dual -- the tangent of forward and backward.  fp32 throughout (the reference
// calls .float() here); wavefront/LDS reductions for LayerNorm and softmax.
#include "kernels.h"

namespace {

template <class S> DEVI S ldz(const float* pv, const float* pt, size_t i) {
  if constexpr (IsDual<S>::v) return Dual(pv[i], pt ? pt[i] : 0.f);
  else return pv[i];
}
// store both parts (scratch that the tangent pass re-derives)
template <class S> DEVI void st2(float* pv, float* pt, size_t i, S x) {
  pv[i] = val(x);
  if constexpr (IsDual<S>::v) pt[i] = x.t;
}

template <class S>
__global__ void k_gelu(float* __restrict__ g, float* __restrict__ g_t, const float* __restrict__ p,
                       const float* __restrict__ p_t, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    stS<S>(g, g_t, i, gelu_(ldS<S>(p, p_t, i)));
}
template <class S>
__global__ void k_gelu_bwd(float* __restrict__ pbar, float* __restrict__ pbar_t,
                           const float* __restrict__ rbar, const float* __restrict__ rbar_t,
                           const float* __restrict__ gbar, const float* __restrict__ gbar_t,
                           const float* __restrict__ p, const float* __restrict__ p_t, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    S r = ldS<S>(rbar, rbar_t, i) + ldS<S>(gbar, gbar_t, i) * dgelu_(ldS<S>(p, p_t, i));
    stS<S>(pbar, pbar_t, i, r);
  }
}

template <class S> struct LnStats { S mu, rstd; };
template <class S>
DEVI LnStats<S> ln_stats(const float* r, const float* r_t, int d, float eps, float* scratch) {
  S s = mk<S>(0.f, 0.f);
  for (int i = threadIdx.x; i < d; i += blockDim.x) s = s + ldS<S>(r, r_t, i);
  s = block_sum(s, scratch);
  S mu = s * (1.f / d);
  S q = mk<S>(0.f, 0.f);
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    S c = ldS<S>(r, r_t, i) - mu;
    q = q + c * c;
  }
  q = block_sum(q, scratch);
  LnStats<S> o;
  o.mu = mu;
  o.rstd = rsqrt_(q * (1.f / d) + eps);
  return o;
}

// one block per row: r = f*mask + p ; y = (r-mu)*rstd*gamma + beta
template <class S>
__global__ void k_ln_fwd(float* __restrict__ y, float* __restrict__ y_t, float* __restrict__ r,
                         float* __restrict__ r_t, const float* __restrict__ f,
                         const float* __restrict__ f_t, const float* __restrict__ mask,
                         const float* __restrict__ p, const float* __restrict__ p_t,
                         const float* __restrict__ gamma, const float* __restrict__ gamma_t,
                         const float* __restrict__ beta, const float* __restrict__ beta_t, int d,
                         float eps) {
  __shared__ float scratch[16];
  size_t base = (size_t)blockIdx.x * d;
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    S v = ldS<S>(f, f_t, base + i) * (mask ? mask[base + i] : 1.f) + ldS<S>(p, p_t, base + i);
    stS<S>(r, r_t, base + i, v);
  }
  __syncthreads();
  LnStats<S> st = ln_stats<S>(r + base, IsDual<S>::v ? r_t + base : nullptr, d, eps, scratch);
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    S v = (ldS<S>(r, r_t, base + i) - st.mu) * st.rstd * ldz<S>(gamma, gamma_t, i) +
          ldz<S>(beta, beta_t, i);
    stS<S>(y, y_t, base + i, v);
  }
}

// one block per row: gy = ybar*gamma ; rbar = rstd*(gy - mean(gy) - rhat*mean(gy*rhat)) ;
// fbar = rbar*mask ; stats[row] = {mu.v, rstd.v, mu.t, rstd.t} for the column kernel
template <class S>
__global__ void k_ln_bwd_rows(float* __restrict__ rbar, float* __restrict__ rbar_t,
                              float* __restrict__ fbar, float* __restrict__ fbar_t,
                              float* __restrict__ stats, const float* __restrict__ ybar,
                              const float* __restrict__ ybar_t, const float* __restrict__ r,
                              const float* __restrict__ r_t, const float* __restrict__ mask,
                              const float* __restrict__ gamma, const float* __restrict__ gamma_t,
                              int d, float eps) {
  __shared__ float scratch[16];
  size_t base = (size_t)blockIdx.x * d;
  LnStats<S> st = ln_stats<S>(r + base, IsDual<S>::v ? r_t + base : nullptr, d, eps, scratch);
  S s1 = mk<S>(0.f, 0.f), s2 = mk<S>(0.f, 0.f);
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    S gy = ldS<S>(ybar, ybar_t, base + i) * ldz<S>(gamma, gamma_t, i);
    S rh = (ldS<S>(r, r_t, base + i) - st.mu) * st.rstd;
    s1 = s1 + gy;
    s2 = s2 + gy * rh;
  }
  s1 = block_sum(s1, scratch);
  s2 = block_sum(s2, scratch);
  S m1 = s1 * (1.f / d), m2 = s2 * (1.f / d);
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    S gy = ldS<S>(ybar, ybar_t, base + i) * ldz<S>(gamma, gamma_t, i);
    S rh = (ldS<S>(r, r_t, base + i) - st.mu) * st.rstd;
    S rb = st.rstd * (gy - m1 - rh * m2);
    stS<S>(rbar, rbar_t, base + i, rb);
    stS<S>(fbar, fbar_t, base + i, rb * (mask ? mask[base + i] : 1.f));
  }
  if (threadIdx.x == 0) {
    stats[blockIdx.x * 4 + 0] = val(st.mu);
    stats[blockIdx.x * 4 + 1] = val(st.rstd);
    stats[blockIdx.x * 4 + 2] = tan_(st.mu);
    stats[blockIdx.x * 4 + 3] = tan_(st.rstd);
  }
}
// thread per column: dgamma[i] = sum_n ybar*rhat ; dbeta[i] = sum_n ybar
template <class S>
__global__ void k_ln_bwd_cols(float* __restrict__ dgamma, float* __restrict__ dbeta,
                              const float* __restrict__ stats, const float* __restrict__ ybar,
                              const float* __restrict__ ybar_t, const float* __restrict__ r,
                              const float* __restrict__ r_t, int n, int d) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  S a = mk<S>(0.f, 0.f), b = mk<S>(0.f, 0.f);
  for (int nn = 0; nn < n; ++nn) {
    S mu = mk<S>(stats[nn * 4 + 0], stats[nn * 4 + 2]);
    S rs = mk<S>(stats[nn * 4 + 1], stats[nn * 4 + 3]);
    S yb = ldS<S>(ybar, ybar_t, (size_t)nn * d + i);
    a = a + yb * (ldS<S>(r, r_t, (size_t)nn * d + i) - mu) * rs;
    b = b + yb;
  }
  dgamma[i] = IsDual<S>::v ? tan_(a) : val(a);
  dbeta[i] = IsDual<S>::v ? tan_(b) : val(b);
}

// ------------------------------------------------------------------ contrastive head
// rows 0..n-1: x ; rows n..2n-1: y.   xh = v / ||v||, rn = 1/||v||   (no eps, as the reference)
template <class S>
__global__ void k_rownorm(LossWork w, const float* __restrict__ x, const float* __restrict__ y,
                          const float* __restrict__ x_t, const float* __restrict__ y_t, int n,
                          int d) {
  __shared__ float scratch[16];
  int row = blockIdx.x;
  bool isy = row >= n;
  int rr = isy ? row - n : row;
  const float* v = (isy ? y : x) + (size_t)rr * d;
  const float* v_t = IsDual<S>::v ? (isy ? y_t : x_t) + (size_t)rr * d : nullptr;
  float* oh = (isy ? w.yh : w.xh) + (size_t)rr * d;
  float* oh_t = (isy ? w.yh_t : w.xh_t) + (size_t)rr * d;
  S s = mk<S>(0.f, 0.f);
  for (int i = threadIdx.x; i < d; i += blockDim.x) {
    S a = ldS<S>(v, v_t, i);
    s = s + a * a;
  }
  s = block_sum(s, scratch);
  S rn = rsqrt_(s);
  for (int i = threadIdx.x; i < d; i += blockDim.x) st2<S>(oh, oh_t, i, ldS<S>(v, v_t, i) * rn);
  if (threadIdx.x == 0) st2<S>(isy ? w.rny : w.rnx, isy ? w.rny_t : w.rnx_t, rr, rn);
}
// G[i,j] = xh_i . yh_j ; one wave per (i,j)
template <class S>
__global__ void k_logits(LossWork w, int n, int d) {
  int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wid >= (int64_t)n * n) return;
  int lane = threadIdx.x & 63;
  int i = (int)(wid / n), j = (int)(wid - (int64_t)i * n);
  S acc = mk<S>(0.f, 0.f);
  for (int f = lane; f < d; f += WAVE)
    acc = acc + ldS<S>(w.xh, w.xh_t, (size_t)i * d + f) * ldS<S>(w.yh, w.yh_t, (size_t)j * d + f);
  acc = wave_sum(acc);
  if (lane == 0) st2<S>(w.G, w.G_t, (size_t)i * n + j, acc);
}
// single block.  S_ij = s*G_ij ; L = (1/2n) sum_i (lse_row_i - S_ii) + (1/2n) sum_j (lse_col_j - S_jj)
// Sbar = (P + Q - 2I)/(2n) ; sbar = sum Sbar*G ; Gbar = s*Sbar
template <class S>
__global__ void k_ce(LossWork w, float* __restrict__ loss, float* __restrict__ sbar,
                     float* __restrict__ sbar_t, const float* __restrict__ scale_ptr,
                     float scale_const, int n) {
  extern __shared__ float lds[];  // lse_r[n] v,t ; lse_c[n] v,t ; scratch[16]
  float* lr_v = lds;
  float* lr_t = lds + n;
  float* lc_v = lds + 2 * n;
  float* lc_t = lds + 3 * n;
  float* scratch = lds + 4 * n;
  S s = mk<S>(scale_ptr ? scale_ptr[0] : scale_const, 0.f);
  for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) {
    bool col = i >= n;
    int r = col ? i - n : i;
    float m = -3.0e38f;
    for (int j = 0; j < n; ++j) {
      size_t gi = col ? (size_t)j * n + r : (size_t)r * n + j;
      m = fmaxf(m, val(s) * w.G[gi]);
    }
    S acc = mk<S>(0.f, 0.f);
    for (int j = 0; j < n; ++j) {
      size_t gi = col ? (size_t)j * n + r : (size_t)r * n + j;
      acc = acc + exp_(s * ldS<S>(w.G, w.G_t, gi) - m);
    }
    S lse = log_(acc) + m;
    (col ? lc_v : lr_v)[r] = val(lse);
    (col ? lc_t : lr_t)[r] = tan_(lse);
  }
  __syncthreads();
  S lacc = mk<S>(0.f, 0.f), sacc = mk<S>(0.f, 0.f);
  float inv2n = 0.5f / n;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    int i = e / n, j = e - i * n;
    S g = ldS<S>(w.G, w.G_t, e);
    S sv = s * g;
    S P = exp_(sv - mk<S>(lr_v[i], lr_t[i]));
    S Q = exp_(sv - mk<S>(lc_v[j], lc_t[j]));
    S sb = (P + Q - (i == j ? 2.f : 0.f)) * inv2n;
    sacc = sacc + sb * g;
    st2<S>(w.Gb, w.Gb_t, e, s * sb);
    if (i == j) lacc = lacc + (mk<S>(lr_v[i], lr_t[i]) + mk<S>(lc_v[i], lc_t[i]) - 2.f * sv) * inv2n;
  }
  lacc = block_sum(lacc, scratch);
  sacc = block_sum(sacc, scratch);
  if (threadIdx.x == 0) {
    if constexpr (IsDual<S>::v) {
      if (sbar_t) sbar_t[0] = sacc.t;
    } else {
      loss[0] = lacc;
      sbar[0] = sacc;
    }
  }
}
// block per row (x rows then y rows): hb = Gbar(row,:) . other_hat ; out = rn*(hb - hat*(hat.hb))
constexpr int FG_MAX = 16;
template <class S>
__global__ void k_featgrad(LossWork w, float* __restrict__ xbar, float* __restrict__ ybar,
                           float* __restrict__ xbar_t, float* __restrict__ ybar_t, int n, int d) {
  __shared__ float scratch[16];
  int row = blockIdx.x;
  bool isy = row >= n;
  int rr = isy ? row - n : row;
  const float* oth = isy ? w.xh : w.yh;
  const float* oth_t = isy ? w.xh_t : w.yh_t;
  const float* own = (isy ? w.yh : w.xh) + (size_t)rr * d;
  const float* own_t = (isy ? w.yh_t : w.xh_t) + (size_t)rr * d;
  S hb[FG_MAX];
#pragma unroll
  for (int q = 0; q < FG_MAX; ++q) hb[q] = mk<S>(0.f, 0.f);
  // 4 rows of the other side per trip, all their loads issued before the FMAs (the accumulation is a
  // loop-carried chain; one row per trip left every load waiting for the previous trip)
  constexpr int JB = 4;
  for (int j0 = 0; j0 < n; j0 += JB) {
    S gb[JB];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      int j = j0 + jj;
      size_t gi = isy ? (size_t)j * n + rr : (size_t)rr * n + j;
      gb[jj] = j < n ? ldS<S>(w.Gb, w.Gb_t, gi) : mk<S>(0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < FG_MAX; ++q) {
      int f = threadIdx.x + q * blockDim.x;
      if (f < d) {
        S o[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          int j = min(j0 + jj, n - 1);
          o[jj] = ldS<S>(oth, oth_t, (size_t)j * d + f);
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) hb[q] = hb[q] + gb[jj] * o[jj];
      }
    }
  }
  S dotp = mk<S>(0.f, 0.f);
#pragma unroll
  for (int q = 0; q < FG_MAX; ++q) {
    int f = threadIdx.x + q * blockDim.x;
    if (f < d) dotp = dotp + hb[q] * ldS<S>(own, own_t, f);
  }
  dotp = block_sum(dotp, scratch);
  S rn = ldS<S>(isy ? w.rny : w.rnx, isy ? w.rny_t : w.rnx_t, rr);
  float* o = (isy ? ybar : xbar);
  float* o_t = (isy ? ybar_t : xbar_t);
#pragma unroll
  for (int q = 0; q < FG_MAX; ++q) {
    int f = threadIdx.x + q * blockDim.x;
    if (f < d) stS<S>(o, o_t, (size_t)rr * d + f, rn * (hb[q] - ldS<S>(own, own_t, f) * dotp));
  }
}

// out[r, :] = in[idx[r], :]   (reference distill.py:510-513: this_y = syn_texts[these_indices])
__global__ void k_gather_rows(float* __restrict__ out, const float* __restrict__ in,
                              const int64_t* __restrict__ idx, int n, int d) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)n * d;
       i += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(i / d), c = (int)(i - (int64_t)r * d);
    out[i] = in[(idx ? idx[r] : r) * (int64_t)d + c];
  }
}
// out[idx[r], :] += coef*mul * in[r, :]   (idx rows distinct within a step)
__global__ void k_scatter_rows_axpy(float* __restrict__ out, const float* __restrict__ in,
                                    const int64_t* __restrict__ idx, const float* __restrict__ coef,
                                    float mul, int n, int d) {
  const float a = mul * (coef ? coef[0] : 1.f);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)n * d;
       i += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(i / d), c = (int)(i - (int64_t)r * d);
    out[(idx ? idx[r] : r) * (int64_t)d + c] += a * in[i];
  }
}

inline int pgrid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

void launch_gather_rows(float* out, const float* in, const int64_t* idx, int n, int d,
                        hipStream_t st) {
  k_gather_rows<<<pgrid((int64_t)n * d), 256, 0, st>>>(out, in, idx, n, d);
}
void launch_scatter_rows_axpy(float* out, const float* in, const int64_t* idx, const float* coef,
                              float mul, int n, int d, hipStream_t st) {
  k_scatter_rows_axpy<<<pgrid((int64_t)n * d), 256, 0, st>>>(out, in, idx, coef, mul, n, d);
}
void launch_gelu(float* g, float* g_t, const float* p, const float* p_t, int64_t n, hipStream_t st) {
  if (g_t) k_gelu<Dual><<<pgrid(n), 256, 0, st>>>(g, g_t, p, p_t, n);
  else k_gelu<float><<<pgrid(n), 256, 0, st>>>(g, nullptr, p, nullptr, n);
}
void launch_gelu_bwd(float* pbar, float* pbar_t, const float* rbar, const float* rbar_t,
                     const float* gbar, const float* gbar_t, const float* p, const float* p_t,
                     int64_t n, hipStream_t st) {
  if (pbar_t) k_gelu_bwd<Dual><<<pgrid(n), 256, 0, st>>>(pbar, pbar_t, rbar, rbar_t, gbar, gbar_t, p, p_t, n);
  else k_gelu_bwd<float><<<pgrid(n), 256, 0, st>>>(pbar, nullptr, rbar, nullptr, gbar, nullptr, p, nullptr, n);
}
void launch_ln_fwd(float* y, float* y_t, float* r, float* r_t, const float* f, const float* f_t,
                   const float* mask, const float* p, const float* p_t, const float* gamma,
                   const float* gamma_t, const float* beta, const float* beta_t, int n, int d,
                   float eps, hipStream_t st) {
  if (y_t) k_ln_fwd<Dual><<<n, 256, 0, st>>>(y, y_t, r, r_t, f, f_t, mask, p, p_t, gamma, gamma_t, beta, beta_t, d, eps);
  else k_ln_fwd<float><<<n, 256, 0, st>>>(y, nullptr, r, nullptr, f, nullptr, mask, p, nullptr, gamma, nullptr, beta, nullptr, d, eps);
}
void launch_ln_bwd(float* rbar, float* rbar_t, float* fbar, float* fbar_t, float* dgamma,
                   float* dbeta, float* stats, const float* ybar, const float* ybar_t,
                   const float* r, const float* r_t, const float* mask, const float* gamma,
                   const float* gamma_t, int n, int d, float eps, hipStream_t st) {
  if (rbar_t) {
    k_ln_bwd_rows<Dual><<<n, 256, 0, st>>>(rbar, rbar_t, fbar, fbar_t, stats, ybar, ybar_t, r, r_t, mask, gamma, gamma_t, d, eps);
    k_ln_bwd_cols<Dual><<<(d + 255) / 256, 256, 0, st>>>(dgamma, dbeta, stats, ybar, ybar_t, r, r_t, n, d);
  } else {
    k_ln_bwd_rows<float><<<n, 256, 0, st>>>(rbar, nullptr, fbar, nullptr, stats, ybar, nullptr, r, nullptr, mask, gamma, nullptr, d, eps);
    k_ln_bwd_cols<float><<<(d + 255) / 256, 256, 0, st>>>(dgamma, dbeta, stats, ybar, nullptr, r, nullptr, n, d);
  }
}

int64_t loss_work_floats(int n, int d) { return 4ll * n * d + 4ll * n + 4ll * n * n + 64; }
LossWork loss_work_carve(float* b, int n, int d) {
  LossWork w;
  int64_t nd = (int64_t)n * d, nn = (int64_t)n * n;
  w.xh = b; b += nd; w.yh = b; b += nd; w.xh_t = b; b += nd; w.yh_t = b; b += nd;
  w.rnx = b; b += n; w.rny = b; b += n; w.rnx_t = b; b += n; w.rny_t = b; b += n;
  w.G = b; b += nn; w.G_t = b; b += nn; w.Gb = b; b += nn; w.Gb_t = b; b += nn;
  return w;
}
void launch_contrastive(const LossWork& w, float* loss, float* xbar, float* ybar, float* sbar,
                        float* xbar_t, float* ybar_t, float* sbar_t, const float* x,
                        const float* y, const float* x_t, const float* y_t, const float* scale_ptr,
                        float scale_const, int n, int d, hipStream_t st) {
  bool dual = x_t != nullptr;
  int64_t waves = (int64_t)n * n;
  int lg = (int)((waves + 3) / 4);
  size_t shm = (4 * n + 16) * sizeof(float);
  if (dual) {
    k_rownorm<Dual><<<2 * n, 256, 0, st>>>(w, x, y, x_t, y_t, n, d);
    k_logits<Dual><<<lg, 256, 0, st>>>(w, n, d);
    k_ce<Dual><<<1, 1024, shm, st>>>(w, loss, sbar, sbar_t, scale_ptr, scale_const, n);
    k_featgrad<Dual><<<2 * n, 256, 0, st>>>(w, xbar, ybar, xbar_t, ybar_t, n, d);
  } else {
    k_rownorm<float><<<2 * n, 256, 0, st>>>(w, x, y, nullptr, nullptr, n, d);
    k_logits<float><<<lg, 256, 0, st>>>(w, n, d);
    k_ce<float><<<1, 1024, shm, st>>>(w, loss, sbar, nullptr, scale_ptr, scale_const, n);
    k_featgrad<float><<<2 * n, 256, 0, st>>>(w, xbar, ybar, nullptr, nullptr, n, d);
  }
}
