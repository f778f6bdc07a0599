// Weight standardisation of timm's ScaledStdConv2d, done on EVERY forward (it is inside the
// theta-graph and is double-differentiated by reference distill.py:562-567/:606):
//   w_hat = gain*scale * (w - mean_row(w)) * rsqrt(var_row(w) + eps),  scale = gamma*fan_in^-1/2
// One launch covers all convs of the net: one wave per output channel (row of fan_in weights).
// Generic over S in {float, Dual}: the Dual instantiation is the tangent pass.
#include "kernels.h"

namespace {

DEVI int find_conv(const WsDesc* d, int nconv, int row) {
  int lo = 0, hi = nconv - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (d[mid].row_start <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <class S> struct RowStats { S mu, rstd; };

template <class S>
DEVI RowStats<S> row_stats(const float* w, const float* w_t, int fan, float eps, int lane) {
  S s = mk<S>(0.f, 0.f);
  for (int e = lane; e < fan; e += WAVE) s = s + ldS<S>(w, w_t, e);
  s = wave_sum(s);
  S mu = s * (1.f / fan);
  S q = mk<S>(0.f, 0.f);
  for (int e = lane; e < fan; e += WAVE) {
    S d = ldS<S>(w, w_t, e) - mu;
    q = q + d * d;
  }
  q = wave_sum(q);
  RowStats<S> r;
  r.mu = mu;
  r.rstd = rsqrt_(q * (1.f / fan) + eps);
  return r;
}

template <class S, class AT>
__global__ void k_ws_forward(const WsDesc* __restrict__ descs, int nconv, int total_rows,
                             const float* __restrict__ theta, const float* __restrict__ theta_t,
                             AT* __restrict__ wf, AT* __restrict__ wt, AT* __restrict__ wf_t,
                             AT* __restrict__ wt_t) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  int lane = threadIdx.x & 63;
  const WsDesc d = descs[find_conv(descs, nconv, row)];
  int co = row - d.row_start;
  int fan = d.cin_g * d.ksq;
  const float* w = theta + d.off_w + (int64_t)co * fan;
  const float* w_t = IsDual<S>::v ? theta_t + d.off_w + (int64_t)co * fan : nullptr;
  RowStats<S> rs = row_stats<S>(w, w_t, fan, d.eps, lane);
  S gain = ldS<S>(theta + d.off_g, IsDual<S>::v ? theta_t + d.off_g : nullptr, co);
  S gm = gain * d.scale * rs.rstd;
  int g = co / d.cout_g, cog = co - g * d.cout_g;
  for (int e = lane; e < fan; e += WAVE) {
    int ci = e / d.ksq, tap = e - ci * d.ksq;
    S x = (ldS<S>(w, w_t, e) - rs.mu) * gm;
    int64_t of = d.off_wf + ((int64_t)co * d.ksq + tap) * d.cin_pad_g + ci;
    int64_t ot = d.off_wt + (((int64_t)g * d.cin_pad_g + ci) * d.ksq + tap) * d.cout_g + cog;
    if constexpr (IsDual<S>::v) {
      wf_t[of] = from_f<AT>(x.t);
      wt_t[ot] = from_f<AT>(x.t);
      if (wf) wf[of] = from_f<AT>(x.v);
      if (wt) wt[ot] = from_f<AT>(x.v);
    } else {
      wf[of] = from_f<AT>(x);
      wt[ot] = from_f<AT>(x);
    }
  }
}

// u = (w-mu)*rstd, gm = gain*scale:
//   dgain = scale * sum(dwh*u) ;  dw = gm*rstd*(dwh - mean(dwh) - u*mean(dwh*u))
template <class S>
__global__ void k_ws_backward(const WsDesc* __restrict__ descs, int nconv, int total_rows,
                              const float* __restrict__ theta, const float* __restrict__ theta_t,
                              const float* __restrict__ dwf, const float* __restrict__ dwf_t,
                              float* __restrict__ gtheta) {
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  int lane = threadIdx.x & 63;
  const WsDesc d = descs[find_conv(descs, nconv, row)];
  int co = row - d.row_start;
  int fan = d.cin_g * d.ksq;
  const float* w = theta + d.off_w + (int64_t)co * fan;
  const float* w_t = IsDual<S>::v ? theta_t + d.off_w + (int64_t)co * fan : nullptr;
  RowStats<S> rs = row_stats<S>(w, w_t, fan, d.eps, lane);
  S gain = ldS<S>(theta + d.off_g, IsDual<S>::v ? theta_t + d.off_g : nullptr, co);
  const float* dw = dwf + d.off_wf + (int64_t)co * d.ksq * d.cin_pad_g;
  const float* dw_t = IsDual<S>::v ? dwf_t + d.off_wf + (int64_t)co * d.ksq * d.cin_pad_g : nullptr;
  S s1 = mk<S>(0.f, 0.f), s2 = mk<S>(0.f, 0.f);
  for (int e = lane; e < fan; e += WAVE) {
    int ci = e / d.ksq, tap = e - ci * d.ksq;
    S g = ldS<S>(dw, dw_t, (size_t)tap * d.cin_pad_g + ci);
    S u = (ldS<S>(w, w_t, e) - rs.mu) * rs.rstd;
    s1 = s1 + g;
    s2 = s2 + g * u;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  S m1 = s1 * (1.f / fan), m2 = s2 * (1.f / fan);
  S coef = gain * d.scale * rs.rstd;
  float* out_w = gtheta + d.off_w + (int64_t)co * fan;
  for (int e = lane; e < fan; e += WAVE) {
    int ci = e / d.ksq, tap = e - ci * d.ksq;
    S g = ldS<S>(dw, dw_t, (size_t)tap * d.cin_pad_g + ci);
    S u = (ldS<S>(w, w_t, e) - rs.mu) * rs.rstd;
    S r = coef * (g - m1 - u * m2);
    out_w[e] = IsDual<S>::v ? tan_(r) : val(r);
  }
  if (lane == 0) {
    S dg = s2 * d.scale;
    gtheta[d.off_g + co] = IsDual<S>::v ? tan_(dg) : val(dg);
  }
}

}  // namespace

template <class AT>
void launch_ws_forward(const WsDesc* descs, int nconv, int total_rows, const float* theta,
                       const float* theta_t, AT* wf, AT* wt, AT* wf_t, AT* wt_t, hipStream_t st) {
  int grid = (total_rows + 3) / 4;
  if (theta_t)
    k_ws_forward<Dual, AT><<<grid, 256, 0, st>>>(descs, nconv, total_rows, theta, theta_t, wf, wt,
                                                  wf_t, wt_t);
  else
    k_ws_forward<float, AT><<<grid, 256, 0, st>>>(descs, nconv, total_rows, theta, nullptr, wf, wt,
                                                   nullptr, nullptr);
}
template void launch_ws_forward<float>(const WsDesc*, int, int, const float*, const float*, float*,
                                       float*, float*, float*, hipStream_t);
template void launch_ws_forward<bf16>(const WsDesc*, int, int, const float*, const float*, bf16*,
                                      bf16*, bf16*, bf16*, hipStream_t);

void launch_ws_backward(const WsDesc* descs, int nconv, int total_rows, const float* theta,
                        const float* theta_t, const float* dwf, const float* dwf_t, float* gtheta,
                        hipStream_t st) {
  int grid = (total_rows + 3) / 4;
  if (theta_t)
    k_ws_backward<Dual><<<grid, 256, 0, st>>>(descs, nconv, total_rows, theta, theta_t, dwf, dwf_t,
                                              gtheta);
  else
    k_ws_backward<float><<<grid, 256, 0, st>>>(descs, nconv, total_rows, theta, nullptr, dwf,
                                               nullptr, gtheta);
}
