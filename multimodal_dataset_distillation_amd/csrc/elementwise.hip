// HBM-bound NHWC elementwise / reduction kernels of the NFNet image path (everything that is not a
// conv contraction): input gather, shortcut average pool, squeeze-excite pooling / gating /
// residual, final global pool -- each with its backward, and (S = Dual) its tangent.
// All tensors are [n, h*w, c] with c contiguous; every access is a 16-byte chunk per lane.
#include "kernels.h"

#ifndef MDD_EW_TILED
#define MDD_EW_TILED 1   // per-image tiled SE apply / gradient kernels (0: the flat grid-stride versions)
#endif

namespace {

template <class AT> DEVI void ld_chunk(const AT* p, int64_t chunk_idx, float* f) {
  uint4 v = ((const uint4*)p)[chunk_idx];
  Chunk<AT>::unpack(v, f);
}
template <class AT> DEVI void st_chunk(AT* p, int64_t chunk_idx, const float* f) {
  ((uint4*)p)[chunk_idx] = Chunk<AT>::pack(f);
}
template <class S, class AT>
DEVI void ld_chunkS(const AT* pv, const AT* pt, int64_t ci, S* out) {
  constexpr int CE = Chunk<AT>::N;
  float a[CE], b[CE];
  ld_chunk<AT>(pv, ci, a);
  if constexpr (IsDual<S>::v) {
    ld_chunk<AT>(pt, ci, b);
#pragma unroll
    for (int i = 0; i < CE; ++i) out[i] = Dual(a[i], b[i]);
  } else {
#pragma unroll
    for (int i = 0; i < CE; ++i) out[i] = a[i];
  }
}
// Dual: store tangent to pt ; float: store to pv
template <class S, class AT> DEVI void st_chunkS(AT* pv, AT* pt, int64_t ci, const S* x) {
  constexpr int CE = Chunk<AT>::N;
  float a[CE];
  if constexpr (IsDual<S>::v) {
#pragma unroll
    for (int i = 0; i < CE; ++i) a[i] = x[i].t;
    st_chunk<AT>(pt, ci, a);
  } else {
#pragma unroll
    for (int i = 0; i < CE; ++i) a[i] = x[i];
    st_chunk<AT>(pv, ci, a);
  }
}

inline int egrid(int64_t items, int block = 256) {
  int64_t g = (items + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------ image gather / scatter
// reference distill.py:510-513 (x = syn_images[these_indices]) fused with NCHW fp32 -> NHWC AT.
template <class AT>
__global__ void k_img_gather(AT* __restrict__ x0, const float* __restrict__ image,
                             const int64_t* __restrict__ idx, int n, int c, int h, int w, int cpad) {
  int64_t total = (int64_t)n * h * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t hw = i % ((int64_t)h * w);
    int ni = (int)(i / ((int64_t)h * w));
    int64_t src_n = idx ? idx[ni] : ni;
    const float* src = image + (src_n * c) * (int64_t)h * w + hw;
    for (int cc = 0; cc < cpad; ++cc)
      x0[i * cpad + cc] = from_f<AT>(cc < c ? src[(int64_t)cc * h * w] : 0.f);
  }
}
// dimage[idx[n], c, h, w] += coef*mul * x0bar[n, h, w, c]   (idx rows are distinct within a step)
template <class AT>
__global__ void k_img_scatter(float* __restrict__ dimage, const AT* __restrict__ x0bar,
                              const int64_t* __restrict__ idx, const float* __restrict__ coef,
                              float mul, int n, int c, int h, int w, int cpad) {
  const float a = mul * (coef ? coef[0] : 1.f);
  int64_t total = (int64_t)n * h * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t hw = i % ((int64_t)h * w);
    int ni = (int)(i / ((int64_t)h * w));
    int64_t dst_n = idx ? idx[ni] : ni;
    float* dst = dimage + (dst_n * c) * (int64_t)h * w + hw;
    for (int cc = 0; cc < c; ++cc) dst[(int64_t)cc * h * w] += a * to_f(x0bar[i * cpad + cc]);
  }
}

// ------------------------------------------------------------------ first stem conv: data gradient straight into d image
// dimage[idx[n], c, y, x] += a * sum_{src} sum_{ty,tx,co} dy_src[n, (y+1-ty)/2, (x+1-tx)/2, co] * wt_src[c][ty*3+tx][co]
// (3x3, stride 2, pad 1, 3 real input channels; a tap contributes only where its source coordinate is even and
// inside: even y -> ty = 1, odd y -> ty in {0, 2}, the same in x).
// The implicit-GEMM path spent 370 us on this layer per tangent pass (a 128x32 MFMA tile for 3 useful
// columns, 16-byte stores on every other pixel of a parity class) plus 120 us to scatter its padded NHWC
// result into the NCHW fp32 image gradient.  Here a 128-thread block owns 8 rows x 128 pixels of one image:
// the 5 x 65 dy pixels those touch are brought into LDS in one batch of loads (both sources), each of the
// two waves takes ONE x parity (row parity is a compile-time property of the unrolled row index), so only the
// taps that hit are visited, with their 3 x COUT weights fetched once per tap (broadcast LDS reads) and reused
// for the 4 rows they apply to.
template <class AT, int COUT>
__global__ __launch_bounds__(128, 4) void k_stem_dgrad_image(float* __restrict__ dimage, const AT* __restrict__ dy1,
                                                          const AT* __restrict__ w1, const AT* __restrict__ dy2,
                                                          const AT* __restrict__ w2, const int64_t* __restrict__ idx,
                                                          const float* __restrict__ coef, float mul, int n, int s) {
  constexpr int CE = Chunk<AT>::N, NCH = COUT / CE;
  constexpr int PR = 5, PC = 65, NPIX = PR * PC;            // dy patch: rows y0/2 .. y0/2+4, cols x0/2 .. x0/2+64
  __shared__ __attribute__((aligned(16))) float wsh[2][9][3][COUT];
  __shared__ uint4 patch[2][NPIX * NCH];
  const int tid = threadIdx.x, lane = tid & 63, xa = tid >> 6;
  for (int i = tid; i < 2 * 27 * COUT; i += 128) {
    const int src = i / (27 * COUT), r = i - src * 27 * COUT;
    const int tap = r / (3 * COUT), c = (r / COUT) % 3, co = r % COUT;
    const AT* w = src ? w2 : w1;
    wsh[src][tap][c][co] = w ? to_f(w[((size_t)c * 9 + tap) * COUT + co]) : 0.f;   // wt layout [cin_pad][9][cout]
  }
  const int ho = s >> 1, spans = (s + 127) >> 7, yblocks = s >> 3;
  int t = blockIdx.x;
  const int xs = t % spans; t /= spans;
  const int yb = t % yblocks;
  const int ni = t / yblocks;
  const int x0 = xs * 128, y0 = yb * 8, oy0 = y0 >> 1, ox0 = x0 >> 1;
  // ---- patch -> LDS (zeros outside the dy tensor), every thread's loads issued before its LDS stores
  constexpr int PER = (NPIX * NCH + 127) / 128;
#pragma unroll
  for (int src = 0; src < 2; ++src) {
    const AT* dy = src ? dy2 : dy1;
    if (dy == nullptr) continue;
    uint4 r[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int ci = u * 128 + tid, pl = ci / NCH, k = ci - pl * NCH;
      const int pr = pl / PC, pc = pl - pr * PC;
      const bool ok = ci < NPIX * NCH && oy0 + pr < ho && ox0 + pc < ho;
      r[u] = ok ? ((const uint4*)dy)[(((int64_t)ni * ho + oy0 + pr) * ho + ox0 + pc) * NCH + k] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int ci = u * 128 + tid;
      if (ci < NPIX * NCH) patch[src][ci] = r[u];
    }
  }
  __syncthreads();
  const int x = x0 + 2 * lane + xa;
  float acc[8][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
  const int ntx = xa ? 2 : 1;
#pragma unroll
  for (int src = 0; src < 2; ++src) {
    if ((src ? dy2 : dy1) == nullptr) continue;                   // uniform
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      for (int txi = 0; txi < ntx; ++txi) {                       // wave-uniform trip count
        const int tx = xa ? 2 * txi : 1;
        const int pc = ((x + 1 - tx) >> 1) - ox0;                 // 0..64; x + 1 - tx is even and >= 0 here
        float w[3][COUT];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int co = 0; co < COUT; co += 4) {
            const float4 v = *(const float4*)&wsh[src][ty * 3 + tx][c][co];
            w[c][co] = v.x; w[c][co + 1] = v.y; w[c][co + 2] = v.z; w[c][co + 3] = v.w;
          }
        // rows y0 + j with (j + 1 - ty) even: even rows take ty == 1, odd rows ty in {0, 2}: four rows per tap;
        // a source row above the tensor only occurs for y = 0, ty = 2 -> j + 1 - ty < 0
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = 2 * jj + ((ty + 1) & 1);
          if (j + 1 - ty < 0) continue;                           // compile-time
          const int pr = (j + 1 - ty) >> 1;                       // 0..4 (rows past the tensor were zero-filled)
#pragma unroll
          for (int k = 0; k < NCH; ++k) {
            float v[CE];
            Chunk<AT>::unpack(patch[src][(pr * PC + pc) * NCH + k], v);
#pragma unroll
            for (int e = 0; e < CE; ++e) {
              acc[j][0] = fmaf(v[e], w[0][k * CE + e], acc[j][0]);
              acc[j][1] = fmaf(v[e], w[1][k * CE + e], acc[j][1]);
              acc[j][2] = fmaf(v[e], w[2][k * CE + e], acc[j][2]);
            }
          }
        }
      }
    }
  }
  if (x >= s) return;
  const float a = mul * (coef ? coef[0] : 1.f);
  const int64_t dst_n = idx ? idx[ni] : ni;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float* dst = dimage + (dst_n * 3) * (int64_t)s * s + (int64_t)(y0 + j) * s + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(int64_t)c * s * s] += a * acc[j][c];
  }
}

// ------------------------------------------------------------------ AvgPool2d(2, stride, ceil_mode, count_include_pad=False)
// One block per (image, output row); threads walk the row's (ox, chunk column) pairs -- 32-bit index
// arithmetic only (a flat 64-bit index costs three 64-bit divisions per 16-byte chunk).
template <class AT>
__global__ void k_avgpool2(AT* __restrict__ out, const AT* __restrict__ in, int n, int h, int w,
                           int c, int stride, int ho, int wo) {
  constexpr int CE = Chunk<AT>::N;
  const int cch = c / CE;
  const int ni = blockIdx.x / ho, oy = blockIdx.x - ni * ho;
  const int iy = oy * stride;
  const int ny = (iy + 1 < h) ? 2 : 1;
  const int64_t in_row = ((int64_t)ni * h + iy) * w * cch;
  const int64_t out_row = ((int64_t)ni * ho + oy) * wo * cch;
  const int items = wo * cch;
  for (int j = threadIdx.x; j < items; j += blockDim.x) {
    const int ox = j / cch, cc = j - ox * cch;
    const int ix = ox * stride;
    const int nx = (ix + 1 < w) ? 2 : 1;
    // the four window taps are loaded unconditionally (clamped to the last valid row / column, weight 0 when
    // outside) so that all four loads are in flight together
    const int64_t b00 = in_row + (int64_t)ix * cch + cc;
    const int64_t dxo = (nx == 2) ? cch : 0, dyo = (ny == 2) ? (int64_t)w * cch : 0;
    float f00[CE], f01[CE], f10[CE], f11[CE];
    ld_chunk<AT>(in, b00, f00);
    ld_chunk<AT>(in, b00 + dxo, f01);
    ld_chunk<AT>(in, b00 + dyo, f10);
    ld_chunk<AT>(in, b00 + dyo + dxo, f11);
    const float wx = nx == 2 ? 1.f : 0.f, wy = ny == 2 ? 1.f : 0.f;
    const float r = 1.f / (ny * nx);
    float acc[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) acc[e] = (f00[e] + wx * f01[e] + wy * f10[e] + wx * wy * f11[e]) * r;
    st_chunk<AT>(out, out_row + j, acc);
  }
}
// stride == 2 only (non-overlapping windows): din[iy,ix] = dout[iy/2, ix/2] / count.  One block per
// (image, input row).
template <class AT>
__global__ void k_avgpool2_bwd(AT* __restrict__ din, const AT* __restrict__ dout, int n, int h,
                               int w, int c, int ho, int wo) {
  constexpr int CE = Chunk<AT>::N;
  const int cch = c / CE;
  const int ni = blockIdx.x / h, iy = blockIdx.x - ni * h;
  const int oy = iy >> 1;
  const int cy = (oy * 2 + 1 < h) ? 2 : 1;
  const int64_t in_row = ((int64_t)ni * h + iy) * w * cch;
  const int64_t out_row = ((int64_t)ni * ho + oy) * wo * cch;
  const int items = w * cch;
  for (int j = threadIdx.x; j < items; j += blockDim.x) {
    const int ix = j / cch, cc = j - ix * cch;
    const int ox = ix >> 1;
    const int cx = (ox * 2 + 1 < w) ? 2 : 1;
    float f[CE];
    ld_chunk<AT>(dout, out_row + (int64_t)ox * cch + cc, f);
    const float r = 1.f / (cy * cx);
#pragma unroll
    for (int e = 0; e < CE; ++e) f[e] *= r;
    st_chunk<AT>(din, in_row + j, f);
  }
}

// ------------------------------------------------------------------ per-(n,c) reductions over h*w
// block = 256 threads = 8 chunk columns x 32 hw lanes.  MODE 0: mean x ; 1: mean silu(x) ;
// 2: d = ga * sum a*b, then the sigmoid backward of the SE gate: out = d*g*(1-g)
// 3: MODE 2, and the gradient w.r.t. conv3's output is written in the same pass: cb = a * g * ga
//    (a = grad of the block output, b = conv3 output; the pooled path's share of that gradient reaches the
//    residual branch through the mid-channel pooled vector instead -- see Eng::img_backward)
template <class S, class AT, int MODE>
__global__ void k_hw_reduce(float* __restrict__ out, float* __restrict__ out_t,
                            const AT* __restrict__ a, const AT* __restrict__ a_t,
                            const AT* __restrict__ b, const AT* __restrict__ b_t,
                            const float* __restrict__ gate, const float* __restrict__ gate_t,
                            float mul, int hw, int c, AT* __restrict__ cb = nullptr,
                            AT* __restrict__ cb_t = nullptr) {
  constexpr int CE = Chunk<AT>::N;
  int cch = c / CE;
  int colgroups = (cch + 7) / 8;
  int ni = blockIdx.x / colgroups;
  int cg = blockIdx.x - ni * colgroups;
  int col = cg * 8 + (threadIdx.x & 7);
  int lane = threadIdx.x >> 3;  // 0..31
  S acc[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) acc[e] = mk<S>(0.f, 0.f);
  if (col < cch) {
    S gs[MODE == 3 ? CE : 1];
    if constexpr (MODE == 3) {
#pragma unroll
      for (int e = 0; e < CE; ++e) gs[e] = ldS<S>(gate, gate_t, (size_t)ni * c + (size_t)col * CE + e) * mul;
    }
    // two rows per trip: both rows' chunks (2 or 4 sixteen-byte loads per operand pair) are in flight together
    auto body = [&](const S* x, const S* y, int64_t ci) __attribute__((always_inline)) {
      if constexpr (MODE == 3) {
        S o[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) o[e] = x[e] * gs[e];
        st_chunkS<S, AT>(cb, cb_t, ci, o);
      }
      if constexpr (MODE == 0) {
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] = acc[e] + x[e];
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] = acc[e] + silu_(x[e]);
      } else {
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] = acc[e] + x[e] * y[e];
      }
    };
    int p = lane;
    for (; p + 32 < hw; p += 64) {
      const int64_t c0 = ((int64_t)ni * hw + p) * cch + col, c1 = c0 + (int64_t)32 * cch;
      S x0[CE], x1[CE], y0[MODE >= 2 ? CE : 1], y1[MODE >= 2 ? CE : 1];
      ld_chunkS<S, AT>(a, a_t, c0, x0);
      ld_chunkS<S, AT>(a, a_t, c1, x1);
      if constexpr (MODE >= 2) { ld_chunkS<S, AT>(b, b_t, c0, y0); ld_chunkS<S, AT>(b, b_t, c1, y1); }
      body(x0, y0, c0);
      body(x1, y1, c1);
    }
    if (p < hw) {
      const int64_t c0 = ((int64_t)ni * hw + p) * cch + col;
      S x0[CE], y0[MODE >= 2 ? CE : 1];
      ld_chunkS<S, AT>(a, a_t, c0, x0);
      if constexpr (MODE >= 2) ld_chunkS<S, AT>(b, b_t, c0, y0);
      body(x0, y0, c0);
    }
  }
  // reduce over the 32 hw lanes through LDS: sh[lane][colslot][e]
  __shared__ float sh[32][8][CE * 2 + 1];
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    sh[lane][threadIdx.x & 7][e] = val(acc[e]);
    sh[lane][threadIdx.x & 7][CE + e] = tan_(acc[e]);
  }
  __syncthreads();
  // 8 cols * CE elems * (1 or 2) outputs
  int nout = 8 * CE;
  for (int o = threadIdx.x; o < nout; o += blockDim.x) {
    int cs = o / CE, e = o - cs * CE;
    int ccol = cg * 8 + cs;
    if (ccol >= cch) continue;
    float sv = 0.f, stn = 0.f;
    for (int l = 0; l < 32; ++l) {
      sv += sh[l][cs][e];
      stn += sh[l][cs][CE + e];
    }
    int64_t oi = (int64_t)ni * c + (int64_t)ccol * CE + e;
    S r = mk<S>(sv * mul, stn * mul);
    if constexpr (MODE >= 2) {
      S g = ldS<S>(gate, gate_t, oi);
      r = r * g * (1.f - g);
    }
    stS<S>(out, out_t, oi, r);
  }
}

// ------------------------------------------------------------------ SE apply + residual + next pre-activation
// X' = C3*gate*ga + SC ; A' = beta*silu(X')        (timm NormFreeBlock: out*alpha + shortcut with
// out = attn_gain * SE(conv3(.)); ga = attn_gain*alpha; beta = next block's 1/sqrt(expected var))
template <class S, class AT>
__global__ void k_se_apply(const AT* __restrict__ c3, const AT* __restrict__ c3_t,
                           const float* __restrict__ gate, const float* __restrict__ gate_t,
                           const AT* __restrict__ sc, const AT* __restrict__ sc_t,
                           AT* __restrict__ xo, AT* __restrict__ xo_t, AT* __restrict__ ao,
                           AT* __restrict__ ao_t, float ga, float beta, int64_t total_chunks,
                           int hw, int c) {
  constexpr int CE = Chunk<AT>::N;
  int cch = c / CE;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks;
       i += (int64_t)gridDim.x * blockDim.x) {
    int cc = (int)(i % cch);
    int ni = (int)(i / ((int64_t)cch * hw));
    S x[CE], s[CE], xo_[CE], ao_[CE];
    ld_chunkS<S, AT>(c3, c3_t, i, x);
    ld_chunkS<S, AT>(sc, sc_t, i, s);
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      size_t gi = (size_t)ni * c + cc * CE + e;
      S g = ldS<S>(gate, gate_t, gi);
      xo_[e] = x[e] * g * ga + s[e];
      ao_[e] = silu_(xo_[e]) * beta;
    }
    st_chunkS<S, AT>(xo, xo_t, i, xo_);
    if (ao || ao_t) st_chunkS<S, AT>(ao, ao_t, i, ao_);
  }
}
// c3bar = xbar*gate*ga + pbar/hw
template <class S, class AT>
__global__ void k_se_apply_bwd(AT* __restrict__ c3bar, AT* __restrict__ c3bar_t,
                               const AT* __restrict__ xbar, const AT* __restrict__ xbar_t,
                               const float* __restrict__ gate, const float* __restrict__ gate_t,
                               const float* __restrict__ pbar, const float* __restrict__ pbar_t,
                               float ga, int64_t total_chunks, int hw, int c) {
  constexpr int CE = Chunk<AT>::N;
  int cch = c / CE;
  float rhw = 1.f / hw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks;
       i += (int64_t)gridDim.x * blockDim.x) {
    int cc = (int)(i % cch);
    int ni = (int)(i / ((int64_t)cch * hw));
    S x[CE], o[CE];
    ld_chunkS<S, AT>(xbar, xbar_t, i, x);
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      size_t gi = (size_t)ni * c + cc * CE + e;
      S g = ldS<S>(gate, gate_t, gi);
      S pb = ldS<S>(pbar, pbar_t, gi);
      o[e] = x[e] * g * ga + pb * rhw;
    }
    st_chunkS<S, AT>(c3bar, c3bar_t, i, o);
  }
}
// cfbar = ybar/hw * dsilu(cf)
template <class S, class AT>
__global__ void k_final_pool_bwd(AT* __restrict__ cfbar, AT* __restrict__ cfbar_t,
                                 const float* __restrict__ ybar, const float* __restrict__ ybar_t,
                                 const AT* __restrict__ cf, const AT* __restrict__ cf_t,
                                 int64_t total_chunks, int hw, int c) {
  constexpr int CE = Chunk<AT>::N;
  int cch = c / CE;
  float rhw = 1.f / hw;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks;
       i += (int64_t)gridDim.x * blockDim.x) {
    int cc = (int)(i % cch);
    int ni = (int)(i / ((int64_t)cch * hw));
    S x[CE], o[CE];
    ld_chunkS<S, AT>(cf, cf_t, i, x);
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      size_t gi = (size_t)ni * c + cc * CE + e;
      S yb = ldS<S>(ybar, ybar_t, gi);
      o[e] = yb * rhw * dsilu_(x[e]);
    }
    st_chunkS<S, AT>(cfbar, cfbar_t, i, o);
  }
}


// ------------------------------------------------------------------ tiled variants of the per-(image, channel) broadcasts
// block = 32 chunk columns x 8 row lanes of ONE image (grid: image x column tile x row split).  The flat
// grid-stride versions above pay two 64-bit integer divisions per 16-byte chunk (image and column of a flat
// index) and re-load the [n, c] operands (gate, pooled gradient) for every chunk; here they are loaded once per
// thread, and the row loop is 32-bit address arithmetic only.  Used whenever c is a multiple of 32 chunks.
struct RowTile {
  int ni, cc, p0, p1;      // image, chunk column, first row of this thread, end row; rows advance by 8
  size_t base;             // chunk index of (ni, row 0, cc)
};
template <int CE> DEVI RowTile row_tile(int hw, int c, int rows_per_block) {
  RowTile t;
  const int cch = c / CE;
  t.ni = blockIdx.x;
  t.cc = blockIdx.y * 32 + (threadIdx.x & 31);
  const int r0 = blockIdx.z * rows_per_block;
  t.p0 = r0 + (threadIdx.x >> 5);
  t.p1 = min(hw, r0 + rows_per_block);
  t.base = (size_t)t.ni * hw * cch + t.cc;
  return t;
}
template <class S, class AT>
__global__ __launch_bounds__(256) void k_se_apply_tiled(
    const AT* __restrict__ c3, const AT* __restrict__ c3_t, const float* __restrict__ gate,
    const float* __restrict__ gate_t, const AT* __restrict__ sc, const AT* __restrict__ sc_t,
    AT* __restrict__ xo, AT* __restrict__ xo_t, AT* __restrict__ ao, AT* __restrict__ ao_t, float ga,
    float beta, int hw, int c, int rows_per_block) {
  constexpr int CE = Chunk<AT>::N;
  const int cch = c / CE;
  const RowTile t = row_tile<CE>(hw, c, rows_per_block);
  S g[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) g[e] = ldS<S>(gate, gate_t, (size_t)t.ni * c + t.cc * CE + e) * ga;
  const bool act = ao || ao_t;
  for (int p = t.p0; p < t.p1; p += 8) {
    const size_t i = t.base + (size_t)p * cch;
    S x[CE], s_[CE], xo_[CE], ao_[CE];
    ld_chunkS<S, AT>(c3, c3_t, i, x);
    ld_chunkS<S, AT>(sc, sc_t, i, s_);
#pragma unroll
    for (int e = 0; e < CE; ++e) {
      xo_[e] = x[e] * g[e] + s_[e];
      ao_[e] = silu_(xo_[e]) * beta;
    }
    st_chunkS<S, AT>(xo, xo_t, i, xo_);
    if (act) st_chunkS<S, AT>(ao, ao_t, i, ao_);
  }
}
template <class S, class AT>
__global__ __launch_bounds__(256) void k_se_apply_bwd_tiled(
    AT* __restrict__ c3bar, AT* __restrict__ c3bar_t, const AT* __restrict__ xbar,
    const AT* __restrict__ xbar_t, const float* __restrict__ gate, const float* __restrict__ gate_t,
    const float* __restrict__ pbar, const float* __restrict__ pbar_t, float ga, int hw, int c,
    int rows_per_block) {
  constexpr int CE = Chunk<AT>::N;
  const int cch = c / CE;
  const RowTile t = row_tile<CE>(hw, c, rows_per_block);
  const float rhw = 1.f / hw;
  S g[CE], pb[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) {
    const size_t gi = (size_t)t.ni * c + t.cc * CE + e;
    g[e] = ldS<S>(gate, gate_t, gi) * ga;
    pb[e] = ldS<S>(pbar, pbar_t, gi) * rhw;
  }
  for (int p = t.p0; p < t.p1; p += 8) {
    const size_t i = t.base + (size_t)p * cch;
    S x[CE], o[CE];
    ld_chunkS<S, AT>(xbar, xbar_t, i, x);
#pragma unroll
    for (int e = 0; e < CE; ++e) o[e] = x[e] * g[e] + pb[e];
    st_chunkS<S, AT>(c3bar, c3bar_t, i, o);
  }
}
template <class S, class AT>
__global__ __launch_bounds__(256) void k_final_pool_bwd_tiled(
    AT* __restrict__ cfbar, AT* __restrict__ cfbar_t, const float* __restrict__ ybar,
    const float* __restrict__ ybar_t, const AT* __restrict__ cf, const AT* __restrict__ cf_t, int hw,
    int c, int rows_per_block) {
  constexpr int CE = Chunk<AT>::N;
  const int cch = c / CE;
  const RowTile t = row_tile<CE>(hw, c, rows_per_block);
  const float rhw = 1.f / hw;
  S yb[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) yb[e] = ldS<S>(ybar, ybar_t, (size_t)t.ni * c + t.cc * CE + e) * rhw;
  for (int p = t.p0; p < t.p1; p += 8) {
    const size_t i = t.base + (size_t)p * cch;
    S x[CE], o[CE];
    ld_chunkS<S, AT>(cf, cf_t, i, x);
#pragma unroll
    for (int e = 0; e < CE; ++e) o[e] = yb[e] * dsilu_(x[e]);
    st_chunkS<S, AT>(cfbar, cfbar_t, i, o);
  }
}
// rows per block: about 64 (eight per thread), the row range split evenly and rounded up to the 8 row lanes
inline bool tiled_ok(int c, int ce) { return (c % ce) == 0 && ((c / ce) % 32) == 0; }
inline dim3 tiled_grid(int n, int hw, int c, int ce, int* rows_per_block) {
  const int splits = (hw + 63) / 64;
  const int rpb = (((hw + splits - 1) / splits) + 7) & ~7;
  *rows_per_block = rpb;
  return dim3((unsigned)n, (unsigned)(c / ce / 32), (unsigned)((hw + rpb - 1) / rpb));
}

}  // namespace

// ====================================================================== launchers
template <class AT>
void launch_img_gather_nhwc(AT* x0, const float* image, const int64_t* idx, int n, int c, int h,
                            int w, int cpad, hipStream_t st) {
  k_img_gather<AT><<<egrid((int64_t)n * h * w), 256, 0, st>>>(x0, image, idx, n, c, h, w, cpad);
}
template <class AT>
void launch_img_scatter_grad(float* dimage, const AT* x0bar, const int64_t* idx, const float* coef,
                             float mul, int n, int c, int h, int w, int cpad, hipStream_t st) {
  k_img_scatter<AT><<<egrid((int64_t)n * h * w), 256, 0, st>>>(dimage, x0bar, idx, coef, mul, n, c,
                                                               h, w, cpad);
}
template <class AT>
bool launch_stem_dgrad_image(float* dimage, const AT* dy1, const AT* wt1, const AT* dy2, const AT* wt2,
                             const int64_t* idx, const float* coef, float mul, int n, int s, int cout,
                             int cpad, hipStream_t st) {
  (void)cpad;
  if ((s & 7) != 0) return false;
  const int blocks = n * (s >> 3) * ((s + 127) >> 7);            // (image, 8-row block, 128-pixel span)
  if (cout == 16) k_stem_dgrad_image<AT, 16><<<blocks, 128, 0, st>>>(dimage, dy1, wt1, dy2, wt2, idx, coef, mul, n, s);
  else if (cout == 8) k_stem_dgrad_image<AT, 8><<<blocks, 128, 0, st>>>(dimage, dy1, wt1, dy2, wt2, idx, coef, mul, n, s);
  else return false;     // other stem widths keep the implicit-GEMM path
  return true;
}
template <class AT>
void launch_avgpool2(AT* out, const AT* in, int n, int h, int w, int c, int stride,
                     hipStream_t st) {
  int ho = (h + stride - 1) / stride, wo = (w + stride - 1) / stride;
  k_avgpool2<AT><<<n * ho, 256, 0, st>>>(out, in, n, h, w, c, stride, ho, wo);
}
template <class AT>
void launch_avgpool2_bwd(AT* din, const AT* dout, int n, int h, int w, int c, int stride,
                         hipStream_t st) {
  int ho = (h + stride - 1) / stride, wo = (w + stride - 1) / stride;
  k_avgpool2_bwd<AT><<<n * h, 256, 0, st>>>(din, dout, n, h, w, c, ho, wo);
}
template <class AT>
void launch_pool_mean(float* p, const AT* x, int n, int hw, int c, hipStream_t st) {
  int colgroups = (c / Chunk<AT>::N + 7) / 8;
  k_hw_reduce<float, AT, 0><<<n * colgroups, 256, 0, st>>>(p, nullptr, x, nullptr, nullptr,
                                                           nullptr, nullptr, nullptr, 1.f / hw, hw, c);
}
template <class AT>
void launch_se_apply(const AT* c3, const AT* c3_t, const float* gate, const float* gate_t,
                     const AT* sc, const AT* sc_t, AT* xo, AT* xo_t, AT* ao, AT* ao_t, float ga,
                     float beta, int n, int hw, int c, hipStream_t st) {
  int64_t total = (int64_t)n * hw * (c / Chunk<AT>::N);
  if (MDD_EW_TILED && tiled_ok(c, Chunk<AT>::N)) {
    int rpb;
    const dim3 grid = tiled_grid(n, hw, c, Chunk<AT>::N, &rpb);
    if (c3_t)
      k_se_apply_tiled<Dual, AT><<<grid, 256, 0, st>>>(c3, c3_t, gate, gate_t, sc, sc_t, xo, xo_t, ao, ao_t, ga,
                                                       beta, hw, c, rpb);
    else
      k_se_apply_tiled<float, AT><<<grid, 256, 0, st>>>(c3, nullptr, gate, nullptr, sc, nullptr, xo, nullptr, ao,
                                                        nullptr, ga, beta, hw, c, rpb);
    return;
  }
  if (c3_t)
    k_se_apply<Dual, AT><<<egrid(total), 256, 0, st>>>(c3, c3_t, gate, gate_t, sc, sc_t, xo, xo_t,
                                                       ao, ao_t, ga, beta, total, hw, c);
  else
    k_se_apply<float, AT><<<egrid(total), 256, 0, st>>>(c3, nullptr, gate, nullptr, sc, nullptr,
                                                        xo, nullptr, ao, nullptr, ga, beta, total,
                                                        hw, c);
}
template <class AT>
void launch_se_gate_grad(float* zbar, float* zbar_t, const AT* xbar, const AT* xbar_t,
                         const AT* c3, const AT* c3_t, const float* gate, const float* gate_t,
                         float ga, int n, int hw, int c, hipStream_t st) {
  int colgroups = (c / Chunk<AT>::N + 7) / 8;
  if (xbar_t)
    k_hw_reduce<Dual, AT, 2><<<n * colgroups, 256, 0, st>>>(zbar, zbar_t, xbar, xbar_t, c3, c3_t,
                                                            gate, gate_t, ga, hw, c);
  else
    k_hw_reduce<float, AT, 2><<<n * colgroups, 256, 0, st>>>(zbar, nullptr, xbar, nullptr, c3,
                                                             nullptr, gate, nullptr, ga, hw, c);
}
template <class AT>
void launch_se_gate_grad_c3b(float* zbar, float* zbar_t, AT* c3bar, AT* c3bar_t, const AT* xbar, const AT* xbar_t,
                             const AT* c3, const AT* c3_t, const float* gate, const float* gate_t, float ga, int n,
                             int hw, int c, hipStream_t st) {
  int colgroups = (c / Chunk<AT>::N + 7) / 8;
  if (xbar_t)
    k_hw_reduce<Dual, AT, 3><<<n * colgroups, 256, 0, st>>>(zbar, zbar_t, xbar, xbar_t, c3, c3_t, gate, gate_t, ga,
                                                            hw, c, c3bar, c3bar_t);
  else
    k_hw_reduce<float, AT, 3><<<n * colgroups, 256, 0, st>>>(zbar, nullptr, xbar, nullptr, c3, nullptr, gate,
                                                             nullptr, ga, hw, c, c3bar, nullptr);
}
template <class AT>
void launch_se_apply_bwd(AT* c3bar, AT* c3bar_t, const AT* xbar, const AT* xbar_t,
                         const float* gate, const float* gate_t, const float* pbar,
                         const float* pbar_t, float ga, int n, int hw, int c, hipStream_t st) {
  int64_t total = (int64_t)n * hw * (c / Chunk<AT>::N);
  if (MDD_EW_TILED && tiled_ok(c, Chunk<AT>::N)) {
    int rpb;
    const dim3 grid = tiled_grid(n, hw, c, Chunk<AT>::N, &rpb);
    if (xbar_t)
      k_se_apply_bwd_tiled<Dual, AT><<<grid, 256, 0, st>>>(c3bar, c3bar_t, xbar, xbar_t, gate, gate_t, pbar, pbar_t,
                                                           ga, hw, c, rpb);
    else
      k_se_apply_bwd_tiled<float, AT><<<grid, 256, 0, st>>>(c3bar, nullptr, xbar, nullptr, gate, nullptr, pbar,
                                                            nullptr, ga, hw, c, rpb);
    return;
  }
  if (xbar_t)
    k_se_apply_bwd<Dual, AT><<<egrid(total), 256, 0, st>>>(c3bar, c3bar_t, xbar, xbar_t, gate,
                                                           gate_t, pbar, pbar_t, ga, total, hw, c);
  else
    k_se_apply_bwd<float, AT><<<egrid(total), 256, 0, st>>>(c3bar, nullptr, xbar, nullptr, gate,
                                                            nullptr, pbar, nullptr, ga, total, hw,
                                                            c);
}
template <class AT>
void launch_final_pool(float* y, float* y_t, const AT* cf, const AT* cf_t, int n, int hw, int c,
                       hipStream_t st) {
  int colgroups = (c / Chunk<AT>::N + 7) / 8;
  if (cf_t)
    k_hw_reduce<Dual, AT, 1><<<n * colgroups, 256, 0, st>>>(y, y_t, cf, cf_t, nullptr, nullptr,
                                                            nullptr, nullptr, 1.f / hw, hw, c);
  else
    k_hw_reduce<float, AT, 1><<<n * colgroups, 256, 0, st>>>(y, nullptr, cf, nullptr, nullptr,
                                                             nullptr, nullptr, nullptr, 1.f / hw, hw, c);
}
template <class AT>
void launch_final_pool_bwd(AT* cfbar, AT* cfbar_t, const float* ybar, const float* ybar_t,
                           const AT* cf, const AT* cf_t, int n, int hw, int c, hipStream_t st) {
  int64_t total = (int64_t)n * hw * (c / Chunk<AT>::N);
  if (MDD_EW_TILED && tiled_ok(c, Chunk<AT>::N)) {
    int rpb;
    const dim3 grid = tiled_grid(n, hw, c, Chunk<AT>::N, &rpb);
    if (cf_t)
      k_final_pool_bwd_tiled<Dual, AT><<<grid, 256, 0, st>>>(cfbar, cfbar_t, ybar, ybar_t, cf, cf_t, hw, c, rpb);
    else
      k_final_pool_bwd_tiled<float, AT><<<grid, 256, 0, st>>>(cfbar, nullptr, ybar, nullptr, cf, nullptr, hw, c, rpb);
    return;
  }
  if (cf_t)
    k_final_pool_bwd<Dual, AT><<<egrid(total), 256, 0, st>>>(cfbar, cfbar_t, ybar, ybar_t, cf,
                                                             cf_t, total, hw, c);
  else
    k_final_pool_bwd<float, AT><<<egrid(total), 256, 0, st>>>(cfbar, nullptr, ybar, nullptr, cf,
                                                              nullptr, total, hw, c);
}
#define INST(AT)                                                                                   \
  template void launch_img_gather_nhwc<AT>(AT*, const float*, const int64_t*, int, int, int, int,  \
                                           int, hipStream_t);                                      \
  template void launch_img_scatter_grad<AT>(float*, const AT*, const int64_t*, const float*,       \
                                            float, int, int, int, int, int, hipStream_t);          \
  template bool launch_stem_dgrad_image<AT>(float*, const AT*, const AT*, const AT*, const AT*,    \
                                            const int64_t*, const float*, float, int, int, int,    \
                                            int, hipStream_t);                                     \
  template void launch_avgpool2<AT>(AT*, const AT*, int, int, int, int, int, hipStream_t);         \
  template void launch_avgpool2_bwd<AT>(AT*, const AT*, int, int, int, int, int, hipStream_t);     \
  template void launch_pool_mean<AT>(float*, const AT*, int, int, int, hipStream_t);               \
  template void launch_se_apply<AT>(const AT*, const AT*, const float*, const float*, const AT*,   \
                                    const AT*, AT*, AT*, AT*, AT*, float, float, int, int, int,    \
                                    hipStream_t);                                                  \
  template void launch_se_gate_grad<AT>(float*, float*, const AT*, const AT*, const AT*,           \
                                        const AT*, const float*, const float*, float, int, int,    \
                                        int, hipStream_t);                                         \
  template void launch_se_gate_grad_c3b<AT>(float*, float*, AT*, AT*, const AT*, const AT*, const AT*, const AT*, \
                                            const float*, const float*, float, int, int, int, hipStream_t);      \
  template void launch_se_apply_bwd<AT>(AT*, AT*, const AT*, const AT*, const float*,              \
                                        const float*, const float*, const float*, float, int, int, \
                                        int, hipStream_t);                                         \
  template void launch_final_pool<AT>(float*, float*, const AT*, const AT*, int, int, int,         \
                                      hipStream_t);                                                \
  template void launch_final_pool_bwd<AT>(AT*, AT*, const float*, const float*, const AT*,         \
                                          const AT*, int, int, int, hipStream_t);
INST(float)
INST(bf16)
