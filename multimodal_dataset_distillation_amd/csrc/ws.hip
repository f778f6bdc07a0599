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

// Loads are issued in batches of LB per lane before they are consumed: a running sum is a loop-carried
// dependency, and without the batching every 4-byte load of a row waited for the previous one.
constexpr int LB = 8;
template <class S>
DEVI RowStats<S> row_stats(const float* w, const float* w_t, int fan, float eps, int lane) {
  S s = mk<S>(0.f, 0.f);
  for (int e0 = lane; e0 < fan; e0 += WAVE * LB) {
    S v[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      int e = e0 + j * WAVE;
      v[j] = e < fan ? ldS<S>(w, w_t, e) : mk<S>(0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) s = s + v[j];
  }
  s = wave_sum(s);
  S mu = s * (1.f / fan);
  S q = mk<S>(0.f, 0.f);
  for (int e0 = lane; e0 < fan; e0 += WAVE * LB) {
    S v[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      int e = e0 + j * WAVE;
      v[j] = e < fan ? ldS<S>(w, w_t, e) - mu : mk<S>(0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) q = q + v[j] * v[j];
  }
  q = wave_sum(q);
  RowStats<S> r;
  r.mu = mu;
  r.rstd = rsqrt_(q * (1.f / fan) + eps);
  return r;
}

// Forward packing.  A block owns TR consecutive output channels of one group: its 4 waves first take
// the row statistics (one row per wave at a time, two passes, fp32 -- the same arithmetic as the
// reference's var_mean), then the block streams the rows 64 weights at a time: wf ([co][tap][ci]) is
// written straight from the row-major read, and the transposed wt ([g][ci][tap][co]) goes through an
// LDS tile so that TR consecutive output channels are stored together instead of one 2-byte element
// per cache line.
constexpr int TR = 16;

DEVI int find_tile_conv(const WsDesc* d, int nconv, int tile) {
  int lo = 0, hi = nconv - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (d[mid].tile_start <= tile) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <class S, class AT>
__global__ __launch_bounds__(256) void k_ws_forward(const WsDesc* __restrict__ descs, int nconv,
                                                    const float* __restrict__ theta,
                                                    const float* __restrict__ theta_t,
                                                    AT* __restrict__ wf, AT* __restrict__ wt,
                                                    AT* __restrict__ wf_t, AT* __restrict__ wt_t) {
  constexpr bool D = IsDual<S>::v;
  constexpr int RPW = TR / 4;                       // rows per wave
  __shared__ float tv[64][TR + 1];
  __shared__ float tt[D ? 64 : 1][TR + 1];
  __shared__ float s_mu[TR], s_gm[TR], s_mu_t[TR], s_gm_t[TR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const WsDesc d = descs[find_tile_conv(descs, nconv, blockIdx.x)];
  const int tpg = (d.cout_g + TR - 1) / TR;         // tiles per group
  const int tl = blockIdx.x - d.tile_start;
  const int g = tl / tpg, cog0 = (tl - g * tpg) * TR;
  const int nrows = min(TR, d.cout_g - cog0);
  const int co0 = g * d.cout_g + cog0;
  const int fan = d.cin_g * d.ksq;
  const float* wbase = theta + d.off_w + (int64_t)co0 * fan;
  const float* wbase_t = D ? theta_t + d.off_w + (int64_t)co0 * fan : nullptr;

  // ---- row statistics
  for (int i = 0; i < RPW; ++i) {
    const int r = wave + 4 * i;
    if (r >= nrows) break;
    RowStats<S> rs = row_stats<S>(wbase + (int64_t)r * fan, D ? wbase_t + (int64_t)r * fan : nullptr,
                                  fan, d.eps, lane);
    S gain = ldS<S>(theta + d.off_g, D ? theta_t + d.off_g : nullptr, co0 + r);
    S gm = gain * d.scale * rs.rstd;
    if (lane == 0) {
      s_mu[r] = val(rs.mu); s_gm[r] = val(gm);
      s_mu_t[r] = tan_(rs.mu); s_gm_t[r] = tan_(gm);
    }
  }
  __syncthreads();

  // ---- normalise + pack (the next 64-weight slab is loaded while this one is being stored)
  const bool pw = d.ksq == 1;
  S v[RPW], vn[RPW];
  auto load_slab = [&](int e0, S* dst) {
    const int e = e0 + lane;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + 4 * i;
      dst[i] = (r < nrows && e < fan) ? ldS<S>(wbase, wbase_t, (int64_t)r * fan + e) : mk<S>(0.f, 0.f);
    }
  };
  load_slab(0, v);
  for (int e0 = 0; e0 < fan; e0 += 64) {
    const int e = e0 + lane;
    const bool eok = e < fan;
    const int ci = pw ? e : e / d.ksq, tap = pw ? 0 : e - ci * d.ksq;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + 4 * i;
      if (r < nrows && eok) {
        S x = (v[i] - mk<S>(s_mu[r], s_mu_t[r])) * mk<S>(s_gm[r], s_gm_t[r]);
        const int64_t of = d.off_wf + ((int64_t)(co0 + r) * d.ksq + tap) * d.cin_pad_g + ci;
        if constexpr (D) {
          wf_t[of] = from_f<AT>(x.t);
          if (wf) wf[of] = from_f<AT>(x.v);
          tt[lane][r] = x.t;
        } else {
          wf[of] = from_f<AT>(x);
        }
        tv[lane][r] = val(x);
      }
    }
    if (e0 + 64 < fan) load_slab(e0 + 64, vn);
    __syncthreads();
    // transposed store: thread -> (weight e0 + el, channel r): TR consecutive channels per weight
    for (int el = tid / TR; el < 64; el += 256 / TR) {
      const int r = tid % TR, ee = e0 + el;
      if (r < nrows && ee < fan) {
        const int ci2 = pw ? ee : ee / d.ksq, tap2 = pw ? 0 : ee - ci2 * d.ksq;
        const int64_t ot = d.off_wt + (((int64_t)g * d.cin_pad_g + ci2) * d.ksq + tap2) * d.cout_g + cog0 + r;
        if constexpr (D) {
          wt_t[ot] = from_f<AT>(tt[el][r]);
          if (wt) wt[ot] = from_f<AT>(tv[el][r]);
        } else {
          wt[ot] = from_f<AT>(tv[el][r]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPW; ++i) v[i] = vn[i];
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
  for (int e0 = lane; e0 < fan; e0 += WAVE * LB) {
    S gv[LB], uv[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      int e = e0 + j * WAVE;
      bool ok = e < fan;
      int ee = ok ? e : 0;
      int ci = ee / d.ksq, tap = ee - ci * d.ksq;
      S g = ldS<S>(dw, dw_t, (size_t)tap * d.cin_pad_g + ci);
      S u = (ldS<S>(w, w_t, ee) - rs.mu) * rs.rstd;
      gv[j] = ok ? g : mk<S>(0.f, 0.f);
      uv[j] = u;
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      s1 = s1 + gv[j];
      s2 = s2 + gv[j] * uv[j];
    }
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

int ws_tile_rows() { return TR; }

template <class AT>
void launch_ws_forward(const WsDesc* descs, int nconv, int total_rows, int total_tiles, const float* theta,
                       const float* theta_t, AT* wf, AT* wt, AT* wf_t, AT* wt_t, hipStream_t st) {
  (void)total_rows;
  if (theta_t)
    k_ws_forward<Dual, AT><<<total_tiles, 256, 0, st>>>(descs, nconv, theta, theta_t, wf, wt, wf_t, wt_t);
  else
    k_ws_forward<float, AT><<<total_tiles, 256, 0, st>>>(descs, nconv, theta, nullptr, wf, wt, nullptr,
                                                         nullptr);
}
template void launch_ws_forward<float>(const WsDesc*, int, int, int, const float*, const float*, float*,
                                       float*, float*, float*, hipStream_t);
template void launch_ws_forward<bf16>(const WsDesc*, int, int, int, const float*, const float*, bf16*,
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
