// Building blocks of the ViT image encoder (BASELINE configs[4]): the operators a transformer
// block has and an NFNet block does not -- LayerNorm over token rows, exact GELU, softmax rows and the batched
// attention contractions -- each in the four forms the unrolled matching loop needs: forward, backward, and
// (S = Dual) the tangents of both, obtained by running the SAME kernel body on dual numbers (common.h), exactly as
// the NFNet elementwise kernels do.  Tangent calls read the primal operands and write ONLY the tangent outputs.
// Reference: the reference reaches its transformers through timm / clip (networks.py:661,668); what is implemented
// is timm 0.6.7's VisionTransformer block as restated in oracle/vit_ref.py.
// Status: op-level entry points (include/mdd_hip.h `mdd_op_*`), parity-tested against torch; the engine topology
// that walks them is the next step.
#include <algorithm>
#include <type_traits>

#include "kernels.h"
#include "mdd_hip.h"

#define CHECK_ARG(cond, msg) \
  do { if (!(cond)) return mdd_set_error_msg(2, "mdd: invalid argument: " msg); } while (0)
#ifndef MDD_VIT_MFMA_ATTENTION
#define MDD_VIT_MFMA_ATTENTION 1   // bf16 storage: attention contractions on v_mfma_f32_32x32x16_bf16 (0: the S-generic FMA kernel)
#endif

#ifndef MDD_LN_BWD_ROWS
#define MDD_LN_BWD_ROWS 16   // rows per block of k_ln_bwd (four waves, one row per wave at a time): the parameter-gradient sums cost
                             // one fp32 atomic per column and block
#endif

namespace {

template <class AT> DEVI void ldc(const AT* p, int64_t ci, float* f) {
  uint4 v = ((const uint4*)p)[ci];
  Chunk<AT>::unpack(v, f);
}
template <class AT> DEVI void stc(AT* p, int64_t ci, const float* f) { ((uint4*)p)[ci] = Chunk<AT>::pack(f); }
template <class S, class AT> DEVI void ldcS(const AT* pv, const AT* pt, int64_t ci, S* out) {
  constexpr int CE = Chunk<AT>::N;
  float a[CE], b[CE];
  ldc<AT>(pv, ci, a);
  if constexpr (IsDual<S>::v) {
    ldc<AT>(pt, ci, b);
#pragma unroll
    for (int i = 0; i < CE; ++i) out[i] = Dual(a[i], b[i]);
  } else {
#pragma unroll
    for (int i = 0; i < CE; ++i) out[i] = a[i];
  }
}
template <class S, class AT> DEVI void stcS(AT* pv, AT* pt, int64_t ci, const S* x) {   // Dual: the tangent only
  constexpr int CE = Chunk<AT>::N;
  float a[CE];
#pragma unroll
  for (int i = 0; i < CE; ++i) a[i] = IsDual<S>::v ? tan_(x[i]) : val(x[i]);
  stc<AT>(IsDual<S>::v ? pt : pv, ci, a);
}
// fp32 parameter vectors (gamma, beta): four at a time
template <class S> DEVI void ldp4(const float* pv, const float* pt, int64_t i, S* out) {
  const float4 a = *(const float4*)(pv + i);
  if constexpr (IsDual<S>::v) {
    const float4 b = *(const float4*)(pt + i);
    out[0] = Dual(a.x, b.x); out[1] = Dual(a.y, b.y); out[2] = Dual(a.z, b.z); out[3] = Dual(a.w, b.w);
  } else {
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
  }
}

// ------------------------------------------------------------------ LayerNorm over rows of `dim` (a wave per row)
// lane l owns chunks l, l+64, ... (MAXC of them at most): the whole row sits in registers between the two reductions
template <class S, class AT, int MAXC>
__global__ __launch_bounds__(256) void k_ln_fwd(const AT* __restrict__ x, const AT* __restrict__ x_t,
                                                const float* __restrict__ g, const float* __restrict__ g_t,
                                                const float* __restrict__ b, const float* __restrict__ b_t,
                                                AT* __restrict__ y, AT* __restrict__ y_t, int rows, int dim, float eps,
                                                const AT* __restrict__ ra, const AT* __restrict__ rb, AT* __restrict__ rs) {
  // ra / rb / rs (mdd_op_add_layernorm): the normalised tensor is the residual sum ra + rb, which is formed here, stored to rs
  // in the storage type and normalised AS STORED (what a separate add kernel followed by this one would see).  Primal call:
  // x is unused, rs = ra + rb.  Tangent call: x = the primal sum (stashed), ra / rb / rs = the tangents.
  constexpr int CE = Chunk<AT>::N;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;                       // wave-uniform
  const int cch = dim / CE;
  const float rd = 1.f / dim;
  S v[MAXC][CE];
  S sum = mk<S>(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < cch) {
      const int64_t ci = (int64_t)row * cch + c;
      if (ra) {
        float p0[CE], p1[CE], q[CE];
        ldc<AT>(ra, ci, p0); ldc<AT>(rb, ci, p1);
#pragma unroll
        for (int e = 0; e < CE; ++e) q[e] = p0[e] + p1[e];
        stc<AT>(rs, ci, q);
        if constexpr (sizeof(AT) == 2) Chunk<AT>::unpack(Chunk<AT>::pack(q), q);      // as stored
        if constexpr (IsDual<S>::v) {
          float pv[CE];
          ldc<AT>(x, ci, pv);
#pragma unroll
          for (int e = 0; e < CE; ++e) v[i][e] = Dual(pv[e], q[e]);
        } else {
#pragma unroll
          for (int e = 0; e < CE; ++e) v[i][e] = q[e];
        }
      } else {
        ldcS<S, AT>(x, x_t, ci, v[i]);
      }
#pragma unroll
      for (int e = 0; e < CE; ++e) sum = sum + v[i][e];
    }
  }
  const S mean = wave_sum(sum) * rd;
  S sq = mk<S>(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < cch) {
#pragma unroll
      for (int e = 0; e < CE; ++e) { const S d = v[i][e] - mean; sq = sq + d * d; }
    }
  const S rstd = rsqrt_(wave_sum(sq) * rd + eps);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < cch) {
      S o[CE];
#pragma unroll
      for (int e = 0; e < CE; e += 4) {
        S gg[4], bb[4];
        ldp4<S>(g, g_t, (int64_t)c * CE + e, gg);
        ldp4<S>(b, b_t, (int64_t)c * CE + e, bb);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[e + k] = (v[i][e + k] - mean) * rstd * gg[k] + bb[k];
      }
      stcS<S, AT>(y, y_t, (int64_t)row * cch + c, o);
    }
  }
}

// dx = rstd * (gy - mean(gy) - xh * mean(gy * xh)), gy = dy * gamma, xh = (x - mean) * rstd;
// dgamma += sum_rows dy * xh, dbeta += sum_rows dy (per-block partial sums over its rows, then fp32 atomics).
// The statistics are recomputed from x (x is read anyway).
template <class S, class AT, int MAXC>
__global__ __launch_bounds__(256) void k_ln_bwd(const AT* __restrict__ x, const AT* __restrict__ x_t,
                                                const AT* __restrict__ dy, const AT* __restrict__ dy_t,
                                                const float* __restrict__ g, const float* __restrict__ g_t,
                                                const AT* __restrict__ res, const AT* __restrict__ res_t,
                                                AT* __restrict__ dx, AT* __restrict__ dx_t, float* __restrict__ dg,
                                                float* __restrict__ dg_t, float* __restrict__ db,
                                                float* __restrict__ db_t, int rows, int dim, float eps,
                                                int rows_per_block) {
  constexpr int CE = Chunk<AT>::N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cch = dim / CE;
  const float rd = 1.f / dim;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  S gam[MAXC][CE], ag[MAXC][CE], ab[MAXC][CE];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < CE; ++e) { ag[i][e] = mk<S>(0.f, 0.f); ab[i][e] = mk<S>(0.f, 0.f); gam[i][e] = mk<S>(0.f, 0.f); }
    if (c < cch) {
#pragma unroll
      for (int e = 0; e < CE; e += 4) ldp4<S>(g, g_t, (int64_t)c * CE + e, &gam[i][e]);
    }
  }
  for (int row = r0 + wave; row < r1; row += 4) {
    S v[MAXC][CE], d[MAXC][CE];
    S sum = mk<S>(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < cch) {
        ldcS<S, AT>(x, x_t, (int64_t)row * cch + c, v[i]);
        ldcS<S, AT>(dy, dy_t, (int64_t)row * cch + c, d[i]);
#pragma unroll
        for (int e = 0; e < CE; ++e) sum = sum + v[i][e];
      }
    }
    const S mean = wave_sum(sum) * rd;
    S sq = mk<S>(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < cch) {
#pragma unroll
        for (int e = 0; e < CE; ++e) { const S t = v[i][e] - mean; sq = sq + t * t; }
      }
    const S rstd = rsqrt_(wave_sum(sq) * rd + eps);
    S s1 = mk<S>(0.f, 0.f), s2 = mk<S>(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
      if (lane + 64 * i < cch) {
#pragma unroll
        for (int e = 0; e < CE; ++e) {
          v[i][e] = (v[i][e] - mean) * rstd;                 // xh
          ag[i][e] = ag[i][e] + d[i][e] * v[i][e];
          ab[i][e] = ab[i][e] + d[i][e];
          d[i][e] = d[i][e] * gam[i][e];                     // gy
          s1 = s1 + d[i][e];
          s2 = s2 + d[i][e] * v[i][e];
        }
      }
    const S m1 = wave_sum(s1) * rd, m2 = wave_sum(s2) * rd;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < cch) {
        S o[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) o[e] = rstd * (d[i][e] - m1 - v[i][e] * m2);
        if (res) {             // the gradient arriving over the residual connection around this LayerNorm's branch
          S r[CE];
          ldcS<S, AT>(res, res_t, (int64_t)row * cch + c, r);
#pragma unroll
          for (int e = 0; e < CE; ++e) o[e] = o[e] + r[e];
        }
        stcS<S, AT>(dx, dx_t, (int64_t)row * cch + c, o);
      }
    }
  }
  // the four waves' column sums -> one atomic per column and block.  Dual: the tangent sums only (the primal sums
  // were produced by the primal call)
  __shared__ float red[4][64 * MAXC * CE + 1];
  auto reduce_to = [&](S (&acc)[MAXC][CE], float* outp) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int e = 0; e < CE; ++e) red[wave][(i * 64 + lane) * CE + e] = IsDual<S>::v ? tan_(acc[i][e]) : val(acc[i][e]);
    __syncthreads();
    for (int o = threadIdx.x; o < 64 * MAXC * CE; o += 256) {
      const int e = o % CE, slot = o / CE, i = slot / 64, l = slot - i * 64;
      const int c = l + 64 * i;
      if (c < cch) atomicAdd(outp + (int64_t)c * CE + e, red[0][o] + red[1][o] + red[2][o] + red[3][o]);
    }
  };
  reduce_to(ag, IsDual<S>::v ? dg_t : dg);
  reduce_to(ab, IsDual<S>::v ? db_t : db);
}

// ------------------------------------------------------------------ exact GELU on chunks
template <class S, class AT>
__global__ void k_gelu_fwd(const AT* __restrict__ c, const AT* __restrict__ c_t, AT* __restrict__ a,
                           AT* __restrict__ a_t, int64_t chunks) {
  constexpr int CE = Chunk<AT>::N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * blockDim.x) {
    S x[CE], o[CE];
    ldcS<S, AT>(c, c_t, i, x);
#pragma unroll
    for (int e = 0; e < CE; ++e) o[e] = gelu_(x[e]);
    stcS<S, AT>(a, a_t, i, o);
  }
}
template <class S, class AT>
__global__ void k_gelu_bwd(const AT* __restrict__ c, const AT* __restrict__ c_t, const AT* __restrict__ ab,
                           const AT* __restrict__ ab_t, AT* __restrict__ cb, AT* __restrict__ cb_t, int64_t chunks) {
  constexpr int CE = Chunk<AT>::N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * blockDim.x) {
    S x[CE], g[CE], o[CE];
    ldcS<S, AT>(c, c_t, i, x);
    ldcS<S, AT>(ab, ab_t, i, g);
#pragma unroll
    for (int e = 0; e < CE; ++e) o[e] = g[e] * dgelu_(x[e]);
    stcS<S, AT>(cb, cb_t, i, o);
  }
}

// ------------------------------------------------------------------ softmax over rows of fp32 scores (a wave per row)
constexpr int SM_MAXE = 8;     // columns per lane: cols <= 512
template <class S, class TS>
// (no __restrict__: the engine calls these in place, p == s / ds == dp; every element of a row is loaded before any
// is stored -- the wave reduction sits in between)
__global__ __launch_bounds__(256) void k_softmax_fwd(const TS* s, const TS* s_t, TS* p, TS* p_t, int64_t rows,
                                                     int cols, int ld, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  S v[SM_MAXE];
  float m = -3.0e38f;
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    const int c = lane + 64 * i;
    if (c < cols) { v[i] = ldS<S>(s, s_t, (size_t)(row * ld + c)); m = fmaxf(m, val(v[i])); }
  }
  m = wave_max(m);                                  // the shift is a constant of the row: it carries no tangent
  S sum = mk<S>(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i)
    if (lane + 64 * i < cols) { v[i] = exp_((v[i] - m) * scale); sum = sum + v[i]; }
  sum = wave_sum(sum);
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    const int c = lane + 64 * i;
    if (c < cols) stS<S>(p, p_t, (size_t)(row * ld + c), v[i] / sum);
  }
}
// ds = scale * p * (dp - sum_j p_j dp_j)
template <class S, class TS>
__global__ __launch_bounds__(256) void k_softmax_bwd(const TS* p, const TS* p_t, const TS* dp, const TS* dp_t,
                                                     TS* ds, TS* ds_t, int64_t rows, int cols, int ld, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  S pv[SM_MAXE], dv[SM_MAXE];
  S dot = mk<S>(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    const int c = lane + 64 * i;
    if (c < cols) {
      pv[i] = ldS<S>(p, p_t, (size_t)(row * ld + c));
      dv[i] = ldS<S>(dp, dp_t, (size_t)(row * ld + c));
      dot = dot + pv[i] * dv[i];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < SM_MAXE; ++i) {
    const int c = lane + 64 * i;
    if (c < cols) stS<S>(ds, ds_t, (size_t)(row * ld + c), pv[i] * (dv[i] - dot) * scale);
  }
}

// ------------------------------------------------------------------ strided batched contraction, S-generic
// C[b](i, j) = alpha * sum_k A[b](i, k) * B[b](k, j), batch b = (o, q) with o < outer, q < inner; every operand is
// addressed by element strides, so transposes and the head slices of a fused qkv tensor need no copies.
// Dual: C_t = alpha * (A_t B + A B_t); a null tangent pointer is a zero tangent.
struct BG {
  const void *A, *A_t, *B, *B_t;
  void *C, *C_t;
  int M, N, K, inner;
  int64_t sam, sak, sbk, sbn, scm, scn;     // element strides
  int64_t ao, aq, bo, bq, co, cq;           // batch strides (outer, inner)
  float alpha;
};
template <class S, class T> DEVI S ldS0(const T* pv, const T* pt, int64_t i) {
  if constexpr (IsDual<S>::v) return Dual(to_f(pv[i]), pt ? to_f(pt[i]) : 0.f);
  else return to_f(pv[i]);
}
template <class S, class TA, class TB, class TC>
__global__ __launch_bounds__(256) void k_bgemm(const BG p) {
  constexpr int BM = 64, BN = 64, BK = 16;
  __shared__ float As[IsDual<S>::v ? 2 : 1][BK][BM + 1];
  __shared__ float Bs[IsDual<S>::v ? 2 : 1][BK][BN + 1];
  const int b = blockIdx.z, o = b / p.inner, q = b - o * p.inner;
  const TA* A = (const TA*)p.A + o * p.ao + q * p.aq;
  const TA* At = p.A_t ? (const TA*)p.A_t + o * p.ao + q * p.aq : nullptr;
  const TB* B = (const TB*)p.B + o * p.bo + q * p.bq;
  const TB* Bt = p.B_t ? (const TB*)p.B_t + o * p.bo + q * p.bq : nullptr;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  S acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = mk<S>(0.f, 0.f);
  for (int k0 = 0; k0 < p.K; k0 += BK) {
    // stage 16 x 64 of each operand: thread -> (k = tid / 16, four consecutive m / n)
    for (int t = tid; t < BK * BM; t += 256) {
      const int kk = t / BM, mm = t - kk * BM;
      const int gm = m0 + mm, gk = k0 + kk;
      S a = mk<S>(0.f, 0.f);
      if (gm < p.M && gk < p.K) a = ldS0<S>(A, At, gm * p.sam + gk * p.sak);
      As[0][kk][mm] = val(a);
      if constexpr (IsDual<S>::v) As[1][kk][mm] = tan_(a);
    }
    for (int t = tid; t < BK * BN; t += 256) {
      const int kk = t / BN, nn = t - kk * BN;
      const int gn = n0 + nn, gk = k0 + kk;
      S bb = mk<S>(0.f, 0.f);
      if (gn < p.N && gk < p.K) bb = ldS0<S>(B, Bt, gk * p.sbk + gn * p.sbn);
      Bs[0][kk][nn] = val(bb);
      if constexpr (IsDual<S>::v) Bs[1][kk][nn] = tan_(bb);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      S a[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = mk<S>(As[0][kk][ty * 4 + i], IsDual<S>::v ? As[IsDual<S>::v ? 1 : 0][kk][ty * 4 + i] : 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = mk<S>(Bs[0][kk][tx * 4 + j], IsDual<S>::v ? Bs[IsDual<S>::v ? 1 : 0][kk][tx * 4 + j] : 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = acc[i][j] + a[i] * bv[j];
    }
    __syncthreads();
  }
  TC* C = (TC*)(IsDual<S>::v ? p.C_t : p.C) + o * p.co + q * p.cq;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gm = m0 + ty * 4 + i, gn = n0 + tx * 4 + j;
      if (gm < p.M && gn < p.N) {
        const float r = (IsDual<S>::v ? tan_(acc[i][j]) : val(acc[i][j])) * p.alpha;
        C[gm * p.scm + gn * p.scn] = from_f<TC>(r);
      }
    }
}

// ------------------------------------------------------------------ patches <-> image
// cols[(n*np + py*g + px), c*P*P + dy*P + dx] = image[idx[n], c, py*P + dy, px*P + dx]  (the K order of a
// Conv2d(3, D, P, stride P) weight flattened [D][3*P*P]): the patch embedding becomes a pointwise contraction
template <class AT>
__global__ void k_patchify(AT* __restrict__ cols, const float* __restrict__ image, const int64_t* __restrict__ idx,
                           int n, int s, int patch) {
  const int g = s / patch, kk = 3 * patch * patch;
  const int64_t total = (int64_t)n * g * g * kk;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % kk);
    const int64_t r = i / kk;
    const int px = (int)(r % g), py = (int)((r / g) % g), ni = (int)(r / ((int64_t)g * g));
    const int dx = k % patch, dy = (k / patch) % patch, c = k / (patch * patch);
    const int64_t src_n = idx ? idx[ni] : ni;
    cols[i] = from_f<AT>(image[((src_n * 3 + c) * s + py * patch + dy) * (int64_t)s + px * patch + dx]);
  }
}
// dimage[idx[n], c, y, x] += coef*mul * colsbar[...]   (patches do not overlap: every pixel is written by one thread)
template <class AT>
__global__ void k_unpatchify_accum(float* __restrict__ dimage, const AT* __restrict__ colsbar,
                                   const int64_t* __restrict__ idx, const float* __restrict__ coef, float mul, int n,
                                   int s, int patch) {
  const int g = s / patch, kk = 3 * patch * patch;
  const float a = mul * (coef ? coef[0] : 1.f);
  const int64_t total = (int64_t)n * g * g * kk;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % kk);
    const int64_t r = i / kk;
    const int px = (int)(r % g), py = (int)((r / g) % g), ni = (int)(r / ((int64_t)g * g));
    const int dx = k % patch, dy = (k / patch) % patch, c = k / (patch * patch);
    const int64_t dst_n = idx ? idx[ni] : ni;
    dimage[((dst_n * 3 + c) * s + py * patch + dy) * (int64_t)s + px * patch + dx] += a * to_f(colsbar[i]);
  }
}

// ------------------------------------------------------------------ token assembly: x0[n, 0] = cls + pos[0], x0[n, 1+p] = pe[n, p] + pos[1+p]
template <class S, class AT>
__global__ void k_vit_embed(AT* __restrict__ x0, AT* __restrict__ x0_t, const AT* __restrict__ pe,
                            const AT* __restrict__ pe_t, const float* __restrict__ cls, const float* __restrict__ cls_t,
                            const float* __restrict__ pos, const float* __restrict__ pos_t, int n, int tok, int dim) {
  const int64_t total = (int64_t)n * tok * dim;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % dim);
    const int t = (int)((i / dim) % tok);
    const int ni = (int)(i / ((int64_t)dim * tok));
    S v = ldS<S>(pos, pos_t, (size_t)t * dim + d);
    if (t == 0) v = v + ldS<S>(cls, cls_t, (size_t)d);
    else v = v + ldS<S>(pe, pe_t, (size_t)(((int64_t)ni * (tok - 1) + t - 1) * dim + d));
    stS<S>(x0, x0_t, (size_t)i, v);
  }
}
// dpos[t] = sum_n xb[n, t] ; dcls = sum_n xb[n, 0] ; peb[n, p] = xb[n, 1+p].   One block per token position;
// OVERWRITES its slices of the flat gradient (Dual: the tangent sums into the tangent gradient).
template <class S, class AT>
__global__ void k_vit_embed_bwd(const AT* __restrict__ xb, const AT* __restrict__ xb_t, AT* __restrict__ peb,
                                AT* __restrict__ peb_t, float* __restrict__ dcls, float* __restrict__ dpos, int n,
                                int tok, int dim) {
  const int t = blockIdx.x;
  for (int d = threadIdx.x; d < dim; d += blockDim.x) {
    S acc = mk<S>(0.f, 0.f);
    for (int ni = 0; ni < n; ++ni) {
      const size_t src = (size_t)(((int64_t)ni * tok + t) * dim + d);
      const S v = ldS<S>(xb, xb_t, src);
      acc = acc + v;
      if (t > 0) stS<S>(peb, peb_t, (size_t)(((int64_t)ni * (tok - 1) + t - 1) * dim + d), v);
    }
    const float r = IsDual<S>::v ? tan_(acc) : val(acc);
    dpos[(size_t)t * dim + d] = r;
    if (t == 0) dcls[d] = r;
  }
}

// ------------------------------------------------------------------ small elementwise helpers
template <class S, class AT>
__global__ void k_add2(AT* __restrict__ o, AT* __restrict__ o_t, const AT* __restrict__ a, const AT* __restrict__ a_t,
                       const AT* __restrict__ b, const AT* __restrict__ b_t, int64_t chunks) {
  constexpr int CE = Chunk<AT>::N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * blockDim.x) {
    S x[CE], y[CE];
    ldcS<S, AT>(a, a_t, i, x);
    ldcS<S, AT>(b, b_t, i, y);
#pragma unroll
    for (int e = 0; e < CE; ++e) x[e] = x[e] + y[e];
    stcS<S, AT>(o, o_t, i, x);
  }
}
// rows (n, 0) of a [n, tok, dim] activation tensor <-> a dense [n, dim] one; fp32 <-> activation type
template <class AT>
__global__ void k_cls_gather(AT* __restrict__ out, const AT* __restrict__ x, int n, int tok, int dim) {
  const int64_t total = (int64_t)n * dim;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = x[(i / dim) * (int64_t)tok * dim + (i % dim)];
}
template <class AT>
__global__ void k_cls_scatter(AT* __restrict__ xb, const AT* __restrict__ in, int n, int tok, int dim) {
  const int64_t total = (int64_t)n * tok * dim;      // every row of xb is written: zero except the class rows
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)((i / dim) % tok);
    xb[i] = t == 0 ? in[(i / ((int64_t)tok * dim)) * dim + (i % dim)] : from_f<AT>(0.f);
  }
}
template <class AT> __global__ void k_act_to_f32(float* __restrict__ o, const AT* __restrict__ a, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = to_f(a[i]);
}
template <class AT> __global__ void k_f32_to_act(AT* __restrict__ o, const float* __restrict__ a, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = from_f<AT>(a[i]);
}
// packed operands of a linear W [out][in] (fp32 in theta): wf = W as stored (B operand of the forward contraction),
// wt = W^T [in][out] (B operand of the data gradient); 32x32 tiles through LDS
template <class AT>
__global__ void k_lin_pack(AT* __restrict__ wf, AT* __restrict__ wt, const float* __restrict__ w, int out, int in) {
  __shared__ float tile[32][33];
  const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    const int o = o0 + r, i = i0 + tx;
    float v = 0.f;
    if (o < out && i < in) { v = w[(int64_t)o * in + i]; wf[(int64_t)o * in + i] = from_f<AT>(v); }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, o = o0 + tx;
    if (o < out && i < in) wt[(int64_t)i * out + o] = from_f<AT>(tile[tx][r]);
  }
}

// ------------------------------------------------------------------ the same contraction on the matrix cores (bf16 storage)
// 64 x 64 output tile, four waves of 32 x 32 (v_mfma_f32_32x32x16_bf16), K-step 32.  Operands are staged through LDS
// as bf16 rows of K (fp32 operands -- probabilities, score gradients -- are rounded on the way in: this is the bf16
// mode, whose every other contraction rounds its operands the same way).  A tangent call accumulates
// A_t B + A B_t only (the primal product was written by the primal call).  The element strides are arbitrary, so the
// global loads are scalar; the thread -> element map follows whichever index is contiguous in memory.
template <bool DUAL, class TA, class TC>
__global__ __launch_bounds__(256) void k_bgemm_mfma(const BG p) {
  constexpr int BM = 64, BN = 64, BK = 32, PITCH = BK + 8;     // bf16 elements; 80-byte rows: conflict-free 16-byte reads
  __shared__ __attribute__((aligned(16))) bf16 As[DUAL ? 2 : 1][BM][PITCH];
  __shared__ __attribute__((aligned(16))) bf16 Bs[DUAL ? 2 : 1][BN][PITCH];
  const int b = blockIdx.z, o = b / p.inner, q = b - o * p.inner;
  const TA* A = (const TA*)p.A + o * p.ao + q * p.aq;
  const TA* At = (DUAL && p.A_t) ? (const TA*)p.A_t + o * p.ao + q * p.aq : nullptr;
  const bf16* B = (const bf16*)p.B + o * p.bo + q * p.bq;
  const bf16* Bt = (DUAL && p.B_t) ? (const bf16*)p.B_t + o * p.bo + q * p.bq : nullptr;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  // Staging: a thread fetches EIGHT consecutive elements along the index that is contiguous in memory with one or two
  // 16-byte loads when that run is aligned and inside the matrix (callers' strides: heads are 64 wide, score rows are
  // padded to a multiple of four floats), element by element otherwise.  vec_* are block-uniform.
  constexpr int AE = 16 / (int)sizeof(TA);                        // elements per 16-byte load of A
  const bool a_kfast = p.sak == 1, b_kfast = p.sbk == 1;
  const bool a_vec = (a_kfast ? (p.sam % AE == 0) : (p.sak % AE == 0 && p.sam == 1)) &&
                     (((uintptr_t)A | (uintptr_t)(At ? At : A)) & 15) == 0;
  const bool b_vec = (b_kfast ? (p.sbn % 8 == 0) : (p.sbk % 8 == 0 && p.sbn == 1)) &&
                     (((uintptr_t)B | (uintptr_t)(Bt ? Bt : B)) & 15) == 0;
  // run r of thread tid: (fixed index f, first running index g0) ; 64 x 32 tile = 256 runs of 8
  const int a_f = a_kfast ? (tid >> 2) : (tid >> 3), a_g = a_kfast ? (tid & 3) * 8 : (tid & 7) * 8;   // kfast: f = m, g = k
  const int b_f = b_kfast ? (tid >> 2) : (tid >> 3), b_g = b_kfast ? (tid & 3) * 8 : (tid & 7) * 8;   // else  f = k, g = m / n
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto fetch8 = [&](const auto* src, bool vec, int64_t off, int64_t step, int valid, float* out) {
    using T = std::remove_cv_t<std::remove_pointer_t<decltype(src)>>;
    constexpr int E = 16 / (int)sizeof(T);
    if (vec && valid == 8) {
      if constexpr (E == 8) {
        float f[8];
        Chunk<bf16>::unpack(*(const uint4*)(src + off), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = f[e];
      } else {
        const float4 lo = *(const float4*)(src + off), hi = *(const float4*)(src + off + 4);
        out[0] = lo.x; out[1] = lo.y; out[2] = lo.z; out[3] = lo.w; out[4] = hi.x; out[5] = hi.y; out[6] = hi.z; out[7] = hi.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = e < valid ? to_f(src[off + e * step]) : 0.f;
    }
  };
  for (int k0 = 0; k0 < p.K; k0 += BK) {
    {   // A tile -> As[m][k]
      const int gm = a_kfast ? m0 + a_f : m0 + a_g, gk = a_kfast ? k0 + a_g : k0 + a_f;
      const int lim = a_kfast ? p.K - gk : p.M - gm;                       // elements of the run that exist
      const bool row_ok = a_kfast ? gm < p.M : gk < p.K;
      const int valid = row_ok ? max(0, min(8, lim)) : 0;
      const int64_t off = (int64_t)gm * p.sam + (int64_t)gk * p.sak, step = a_kfast ? 1 : p.sam;
      float v[8], vt[8];
      fetch8(A, a_vec, off, step, valid, v);
      if constexpr (DUAL) { if (At) fetch8(At, a_vec, off, step, valid, vt); else { for (int e = 0; e < 8; ++e) vt[e] = 0.f; } }
      if (a_kfast) {
        float pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = v[e];
        *(uint4*)&As[0][a_f][a_g] = Chunk<bf16>::pack(pk);
        if constexpr (DUAL) *(uint4*)&As[1][a_f][a_g] = Chunk<bf16>::pack(vt);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          As[0][a_g + e][a_f] = (bf16)v[e];
          if constexpr (DUAL) As[1][a_g + e][a_f] = (bf16)vt[e];
        }
      }
    }
    {   // B tile -> Bs[n][k]
      const int gn = b_kfast ? n0 + b_f : n0 + b_g, gk = b_kfast ? k0 + b_g : k0 + b_f;
      const int lim = b_kfast ? p.K - gk : p.N - gn;
      const bool row_ok = b_kfast ? gn < p.N : gk < p.K;
      const int valid = row_ok ? max(0, min(8, lim)) : 0;
      const int64_t off = (int64_t)gk * p.sbk + (int64_t)gn * p.sbn, step = b_kfast ? 1 : p.sbn;
      float v[8], vt[8];
      fetch8(B, b_vec, off, step, valid, v);
      if constexpr (DUAL) { if (Bt) fetch8(Bt, b_vec, off, step, valid, vt); else { for (int e = 0; e < 8; ++e) vt[e] = 0.f; } }
      if (b_kfast) {
        *(uint4*)&Bs[0][b_f][b_g] = Chunk<bf16>::pack(v);
        if constexpr (DUAL) *(uint4*)&Bs[1][b_f][b_g] = Chunk<bf16>::pack(vt);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          Bs[0][b_g + e][b_f] = (bf16)v[e];
          if constexpr (DUAL) Bs[1][b_g + e][b_f] = (bf16)vt[e];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int kc = ks * 16 + (lane >> 5) * 8;
      const bf16x8 a = *(const bf16x8*)&As[0][wm * 32 + (lane & 31)][kc];
      const bf16x8 bb = *(const bf16x8*)&Bs[0][wn * 32 + (lane & 31)][kc];
      if constexpr (DUAL) {
        const bf16x8 at = *(const bf16x8*)&As[1][wm * 32 + (lane & 31)][kc];
        const bf16x8 bt = *(const bf16x8*)&Bs[1][wn * 32 + (lane & 31)][kc];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, bb, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bt, acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  TC* C = (TC*)(DUAL ? p.C_t : p.C) + o * p.co + q * p.cq;
  const int gn = n0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (gm < p.M && gn < p.N) C[gm * p.scm + gn * p.scn] = from_f<TC>(acc[r] * p.alpha);
  }
}

// ------------------------------------------------------------------ split-bf16 ("bf16x2") form of the same kernel, fp32 storage
// every operand element x is staged as hi = bf16(x), lo = bf16(x - hi) in two LDS planes and each product is
// hh + hl + lh on the matrix cores with fp32 accumulation (the ll term is below 2^-16 relative), as k_conv_gemm /
// k_conv_wgrad do in this mode: ~1e-5 of the exact result at a third of the FMA kernel's time.
template <bool DUAL>
__global__ __launch_bounds__(256) void k_bgemm_mfma_split(const BG p) {
  constexpr int BM = 64, BN = 64, BK = 32, PITCH = BK + 8;
  constexpr int NP = DUAL ? 4 : 2;                                  // planes: hi, lo (, tangent hi, tangent lo)
  __shared__ __attribute__((aligned(16))) bf16 As[NP][BM][PITCH];
  __shared__ __attribute__((aligned(16))) bf16 Bs[NP][BN][PITCH];
  const int b = blockIdx.z, o = b / p.inner, q = b - o * p.inner;
  const float* A = (const float*)p.A + o * p.ao + q * p.aq;
  const float* At = (DUAL && p.A_t) ? (const float*)p.A_t + o * p.ao + q * p.aq : nullptr;
  const float* B = (const float*)p.B + o * p.bo + q * p.bq;
  const float* Bt = (DUAL && p.B_t) ? (const float*)p.B_t + o * p.bo + q * p.bq : nullptr;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const bool a_kfast = p.sak == 1, b_kfast = p.sbk == 1;
  const bool a_vec = (a_kfast ? (p.sam % 4 == 0) : (p.sak % 4 == 0 && p.sam == 1)) &&
                     (((uintptr_t)A | (uintptr_t)(At ? At : A)) & 15) == 0;
  const bool b_vec = (b_kfast ? (p.sbn % 4 == 0) : (p.sbk % 4 == 0 && p.sbn == 1)) &&
                     (((uintptr_t)B | (uintptr_t)(Bt ? Bt : B)) & 15) == 0;
  const int a_f = a_kfast ? (tid >> 2) : (tid >> 3), a_g = a_kfast ? (tid & 3) * 8 : (tid & 7) * 8;
  const int b_f = b_kfast ? (tid >> 2) : (tid >> 3), b_g = b_kfast ? (tid & 3) * 8 : (tid & 7) * 8;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto fetch8 = [&](const float* src, bool vec, int64_t off, int64_t step, int valid, float* out) {
    if (vec && valid == 8) {
      const float4 lo = *(const float4*)(src + off), hi = *(const float4*)(src + off + 4);
      out[0] = lo.x; out[1] = lo.y; out[2] = lo.z; out[3] = lo.w; out[4] = hi.x; out[5] = hi.y; out[6] = hi.z; out[7] = hi.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = e < valid ? src[off + e * step] : 0.f;
    }
  };
  // eight fp32 values -> their hi / lo bf16 planes, along k (one 16-byte store each) or along the other index
  auto put8 = [&](bf16 (*plane)[BM][PITCH], int ph, bool kfast, int f, int g, const float* v) {
    float h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bf16 hb = (bf16)v[e];
      h[e] = (float)hb;
      l[e] = v[e] - h[e];
    }
    if (kfast) {
      *(uint4*)&plane[ph][f][g] = Chunk<bf16>::pack(h);
      *(uint4*)&plane[ph + 1][f][g] = Chunk<bf16>::pack(l);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { plane[ph][g + e][f] = (bf16)h[e]; plane[ph + 1][g + e][f] = (bf16)l[e]; }
    }
  };
  for (int k0 = 0; k0 < p.K; k0 += BK) {
    {
      const int gm = a_kfast ? m0 + a_f : m0 + a_g, gk = a_kfast ? k0 + a_g : k0 + a_f;
      const int lim = a_kfast ? p.K - gk : p.M - gm;
      const bool row_ok = a_kfast ? gm < p.M : gk < p.K;
      const int valid = row_ok ? max(0, min(8, lim)) : 0;
      const int64_t off = (int64_t)gm * p.sam + (int64_t)gk * p.sak, step = a_kfast ? 1 : p.sam;
      float v[8];
      fetch8(A, a_vec, off, step, valid, v);
      put8(As, 0, a_kfast, a_f, a_g, v);
      if constexpr (DUAL) {
        if (At) fetch8(At, a_vec, off, step, valid, v); else { for (int e = 0; e < 8; ++e) v[e] = 0.f; }
        put8(As, 2, a_kfast, a_f, a_g, v);
      }
    }
    {
      const int gn = b_kfast ? n0 + b_f : n0 + b_g, gk = b_kfast ? k0 + b_g : k0 + b_f;
      const int lim = b_kfast ? p.K - gk : p.N - gn;
      const bool row_ok = b_kfast ? gn < p.N : gk < p.K;
      const int valid = row_ok ? max(0, min(8, lim)) : 0;
      const int64_t off = (int64_t)gk * p.sbk + (int64_t)gn * p.sbn, step = b_kfast ? 1 : p.sbn;
      float v[8];
      fetch8(B, b_vec, off, step, valid, v);
      put8(Bs, 0, b_kfast, b_f, b_g, v);
      if constexpr (DUAL) {
        if (Bt) fetch8(Bt, b_vec, off, step, valid, v); else { for (int e = 0; e < 8; ++e) v[e] = 0.f; }
        put8(Bs, 2, b_kfast, b_f, b_g, v);
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int kc = ks * 16 + (lane >> 5) * 8, ar = wm * 32 + (lane & 31), br = wn * 32 + (lane & 31);
      const bf16x8 ah = *(const bf16x8*)&As[0][ar][kc], al = *(const bf16x8*)&As[1][ar][kc];
      const bf16x8 bh = *(const bf16x8*)&Bs[0][br][kc], bl = *(const bf16x8*)&Bs[1][br][kc];
      if constexpr (DUAL) {
        const bf16x8 ath = *(const bf16x8*)&As[2][ar][kc], atl = *(const bf16x8*)&As[3][ar][kc];
        const bf16x8 bth = *(const bf16x8*)&Bs[2][br][kc], btl = *(const bf16x8*)&Bs[3][br][kc];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ath, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ath, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(atl, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bth, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, btl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bth, acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* C = (float*)(DUAL ? p.C_t : p.C) + o * p.co + q * p.cq;
  const int gn = n0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (gm < p.M && gn < p.N) C[gm * p.scm + gn * p.scn] = acc[r] * p.alpha;
  }
}

inline int vgrid(int64_t items, int block = 256) {
  int64_t g = (items + block - 1) / block;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, 8192));
}

template <class AT>
int ln_fwd(int rows, int dim, float eps, const void* x, const void* x_t, const float* g, const float* g_t,
           const float* b, const float* b_t, void* y, void* y_t, hipStream_t st, const void* ra = nullptr,
           const void* rb = nullptr, void* rs = nullptr) {
  const int cch = dim / Chunk<AT>::N, maxc = (cch + 63) / 64;
  const int grid = (rows + 3) / 4;
#define LN_F(MC)                                                                                                \
  do {                                                                                                          \
    if (g_t) k_ln_fwd<Dual, AT, MC><<<grid, 256, 0, st>>>((const AT*)x, (const AT*)x_t, g, g_t, b, b_t, (AT*)y,  \
                                                          (AT*)y_t, rows, dim, eps, (const AT*)ra, (const AT*)rb, \
                                                          (AT*)rs);                                             \
    else k_ln_fwd<float, AT, MC><<<grid, 256, 0, st>>>((const AT*)x, nullptr, g, nullptr, b, nullptr, (AT*)y,    \
                                                       nullptr, rows, dim, eps, (const AT*)ra, (const AT*)rb,   \
                                                       (AT*)rs);                                                \
  } while (0)
  if (maxc == 1) LN_F(1); else if (maxc == 2) LN_F(2); else if (maxc == 3) LN_F(3); else LN_F(4);
#undef LN_F
  return 0;
}
template <class AT>
int ln_bwd(int rows, int dim, float eps, const void* x, const void* x_t, const void* dy, const void* dy_t,
           const float* g, const float* g_t, const void* res, const void* res_t, void* dx, void* dx_t, float* dg,
           float* dg_t, float* db, float* db_t, hipStream_t st) {
  const int cch = dim / Chunk<AT>::N, maxc = (cch + 63) / 64;
  const int rpb = MDD_LN_BWD_ROWS, grid = (rows + rpb - 1) / rpb;
#define LN_B(MC)                                                                                                 \
  do {                                                                                                           \
    if (x_t) k_ln_bwd<Dual, AT, MC><<<grid, 256, 0, st>>>((const AT*)x, (const AT*)x_t, (const AT*)dy,            \
                                                          (const AT*)dy_t, g, g_t, (const AT*)res,               \
                                                          (const AT*)res_t, (AT*)dx, (AT*)dx_t, dg, dg_t,        \
                                                          db, db_t, rows, dim, eps, rpb);                        \
    else k_ln_bwd<float, AT, MC><<<grid, 256, 0, st>>>((const AT*)x, nullptr, (const AT*)dy, nullptr, g, nullptr, \
                                                       (const AT*)res, nullptr, (AT*)dx, nullptr, dg, nullptr,   \
                                                       db, nullptr, rows, dim, eps, rpb);                        \
  } while (0)
  if (maxc == 1) LN_B(1); else if (maxc == 2) LN_B(2); else LN_B(3);
#undef LN_B
  return 0;
}
// bf16 storage: the matrix-core kernel for the two operand-type patterns attention uses
template <bool DUAL>
bool bgemm_mfma(const BG& p, int a_is_f32, int c_is_f32, dim3 grid, hipStream_t st) {
  if (!a_is_f32 && c_is_f32) k_bgemm_mfma<DUAL, bf16, float><<<grid, 256, 0, st>>>(p);
  else if (a_is_f32 && !c_is_f32) k_bgemm_mfma<DUAL, float, bf16><<<grid, 256, 0, st>>>(p);
  else if (!a_is_f32 && !c_is_f32) k_bgemm_mfma<DUAL, bf16, bf16><<<grid, 256, 0, st>>>(p);   // scores stored as bf16 too
  else return false;
  return true;
}
template <class S, class AT>
void bgemm_types(const BG& p, int a_is_f32, int c_is_f32, dim3 grid, hipStream_t st) {
  // operand types in use: (A, B, C) = (AT, AT, fp32) scores / score gradients; (fp32, AT, AT) everything that
  // multiplies by the probabilities or their gradients
  if (!a_is_f32 && c_is_f32) k_bgemm<S, AT, AT, float><<<grid, 256, 0, st>>>(p);
  else if (a_is_f32 && !c_is_f32) k_bgemm<S, float, AT, AT><<<grid, 256, 0, st>>>(p);
  else k_bgemm<S, AT, AT, AT><<<grid, 256, 0, st>>>(p);
}

}  // namespace

template <class AT>
void launch_patchify(AT* cols, const float* image, const int64_t* idx, int n, int s, int patch, hipStream_t st) {
  k_patchify<AT><<<vgrid((int64_t)n * s * s * 3), 256, 0, st>>>(cols, image, idx, n, s, patch);
}
template <class AT>
void launch_unpatchify_accum(float* dimage, const AT* colsbar, const int64_t* idx, const float* coef, float mul,
                             int n, int s, int patch, hipStream_t st) {
  k_unpatchify_accum<AT><<<vgrid((int64_t)n * s * s * 3), 256, 0, st>>>(dimage, colsbar, idx, coef, mul, n, s, patch);
}
template <class AT>
void launch_vit_embed(AT* x0, AT* x0_t, const AT* pe, const AT* pe_t, const float* cls, const float* cls_t,
                      const float* pos, const float* pos_t, int n, int tok, int dim, hipStream_t st) {
  const int g = vgrid((int64_t)n * tok * dim);
  if (x0_t) k_vit_embed<Dual, AT><<<g, 256, 0, st>>>(x0, x0_t, pe, pe_t, cls, cls_t, pos, pos_t, n, tok, dim);
  else k_vit_embed<float, AT><<<g, 256, 0, st>>>(x0, nullptr, pe, nullptr, cls, nullptr, pos, nullptr, n, tok, dim);
}
template <class AT>
void launch_vit_embed_bwd(const AT* xb, const AT* xb_t, AT* peb, AT* peb_t, float* dcls, float* dpos, int n, int tok,
                          int dim, hipStream_t st) {
  if (xb_t) k_vit_embed_bwd<Dual, AT><<<tok, 256, 0, st>>>(xb, xb_t, peb, peb_t, dcls, dpos, n, tok, dim);
  else k_vit_embed_bwd<float, AT><<<tok, 256, 0, st>>>(xb, nullptr, peb, nullptr, dcls, dpos, n, tok, dim);
}
template <class AT>
void launch_add2(AT* o, AT* o_t, const AT* a, const AT* a_t, const AT* b, const AT* b_t, int64_t elems, hipStream_t st) {
  const int64_t ch = elems / Chunk<AT>::N;
  if (o_t) k_add2<Dual, AT><<<vgrid(ch), 256, 0, st>>>(o, o_t, a, a_t, b, b_t, ch);
  else k_add2<float, AT><<<vgrid(ch), 256, 0, st>>>(o, nullptr, a, nullptr, b, nullptr, ch);
}
template <class AT> void launch_cls_gather(AT* out, const AT* x, int n, int tok, int dim, hipStream_t st) {
  k_cls_gather<AT><<<vgrid((int64_t)n * dim), 256, 0, st>>>(out, x, n, tok, dim);
}
template <class AT> void launch_cls_scatter(AT* xb, const AT* in, int n, int tok, int dim, hipStream_t st) {
  k_cls_scatter<AT><<<vgrid((int64_t)n * tok * dim), 256, 0, st>>>(xb, in, n, tok, dim);
}
template <class AT> void launch_act_to_f32(float* o, const AT* a, int64_t n, hipStream_t st) {
  k_act_to_f32<AT><<<vgrid(n), 256, 0, st>>>(o, a, n);
}
template <class AT> void launch_f32_to_act(AT* o, const float* a, int64_t n, hipStream_t st) {
  k_f32_to_act<AT><<<vgrid(n), 256, 0, st>>>(o, a, n);
}
template <class AT> void launch_lin_pack(AT* wf, AT* wt, const float* w, int out, int in, hipStream_t st) {
  k_lin_pack<AT><<<dim3((in + 31) / 32, (out + 31) / 32), 256, 0, st>>>(wf, wt, w, out, in);
}
// every linear of the network (and, with theta_t, its tangent) in ONE launch: block -> (layer, 32x32 tile) by the
// layers' tile prefix sums
template <class AT>
__global__ __launch_bounds__(256) void k_lin_pack_all(const LinPackDesc* __restrict__ descs, int nd, const float* __restrict__ th,
                                                      const float* __restrict__ th_t, AT* __restrict__ wf, AT* __restrict__ wt,
                                                      AT* __restrict__ wf_t, AT* __restrict__ wt_t) {
  // 64 x 64 tiles; a thread moves PAIRS of consecutive elements (8-byte loads, 4- or 8-byte stores: full 128 / 256-byte
  // rows per wave); the layer of a block by bisection of the tile prefix sums (a linear scan was ~50 dependent loads)
  __shared__ float tile[2][64][65];
  int lo = 0, hi = nd - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= descs[mid].tile_start) lo = mid; else hi = mid - 1;
  }
  const LinPackDesc d = descs[lo];
  const int t = blockIdx.x - d.tile_start, tpr = (d.in + 63) / 64;
  const int i0 = (t % tpr) * 64, o0 = (t / tpr) * 64;
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const float* w = th + d.off_w; const float* w_t = th_t ? th_t + d.off_w : nullptr;
  const bool even_in = (d.in & 1) == 0, even_out = (d.out & 1) == 0;
  auto put2 = [](AT* p, float a, float b, bool two, bool aligned) __attribute__((always_inline)) {
    if constexpr (sizeof(AT) == 2) {
      if (two && aligned) { *(unsigned*)p = Chunk<bf16>::pk(a, b); return; }
    } else {
      if (two && aligned) { *(float2*)p = make_float2(a, b); return; }
    }
    p[0] = from_f<AT>(a);
    if (two) p[1] = from_f<AT>(b);
  };
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int r = ry + 8 * k, o = o0 + r, i = i0 + 2 * cx;
    float v0 = 0.f, v1 = 0.f, u0 = 0.f, u1 = 0.f;
    if (o < d.out && i < d.in) {
      const int64_t idx = (int64_t)o * d.in + i;
      const bool two = i + 1 < d.in, al = even_in && ((d.off_w & 1) == 0) && ((d.off_p & 1) == 0);
      if (two && al) { const float2 q = *(const float2*)(w + idx); v0 = q.x; v1 = q.y; }
      else { v0 = w[idx]; if (two) v1 = w[idx + 1]; }
      put2(wf + d.off_p + idx, v0, v1, two, al);
      if (w_t) {
        if (two && al) { const float2 q = *(const float2*)(w_t + idx); u0 = q.x; u1 = q.y; }
        else { u0 = w_t[idx]; if (two) u1 = w_t[idx + 1]; }
        put2(wf_t + d.off_p + idx, u0, u1, two, al);
      }
    }
    tile[0][r][2 * cx] = v0; tile[0][r][2 * cx + 1] = v1;
    tile[1][r][2 * cx] = u0; tile[1][r][2 * cx + 1] = u1;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int r = ry + 8 * k, i = i0 + r, o = o0 + 2 * cx;
    if (o < d.out && i < d.in) {
      const int64_t idx = (int64_t)i * d.out + o;
      const bool two = o + 1 < d.out, al = even_out && ((d.off_p & 1) == 0);
      put2(wt + d.off_p + idx, tile[0][2 * cx][r], tile[0][2 * cx + 1][r], two, al);
      if (w_t) put2(wt_t + d.off_p + idx, tile[1][2 * cx][r], tile[1][2 * cx + 1][r], two, al);
    }
  }
}
template <class AT>
void launch_lin_pack_all(const LinPackDesc* descs, int nd, int total_tiles, const float* th, const float* th_t,
                         AT* wf, AT* wt, AT* wf_t, AT* wt_t, hipStream_t st) {
  k_lin_pack_all<AT><<<total_tiles, 256, 0, st>>>(descs, nd, th, th_t, wf, wt, wf_t, wt_t);
}
#define VIT_INST(AT)                                                                                              \
  template void launch_patchify<AT>(AT*, const float*, const int64_t*, int, int, int, hipStream_t);               \
  template void launch_unpatchify_accum<AT>(float*, const AT*, const int64_t*, const float*, float, int, int, int, \
                                            hipStream_t);                                                         \
  template void launch_vit_embed<AT>(AT*, AT*, const AT*, const AT*, const float*, const float*, const float*,     \
                                     const float*, int, int, int, hipStream_t);                                   \
  template void launch_vit_embed_bwd<AT>(const AT*, const AT*, AT*, AT*, float*, float*, int, int, int, hipStream_t); \
  template void launch_add2<AT>(AT*, AT*, const AT*, const AT*, const AT*, const AT*, int64_t, hipStream_t);       \
  template void launch_cls_gather<AT>(AT*, const AT*, int, int, int, hipStream_t);                                \
  template void launch_cls_scatter<AT>(AT*, const AT*, int, int, int, hipStream_t);                               \
  template void launch_act_to_f32<AT>(float*, const AT*, int64_t, hipStream_t);                                   \
  template void launch_f32_to_act<AT>(AT*, const float*, int64_t, hipStream_t);                                   \
  template void launch_lin_pack<AT>(AT*, AT*, const float*, int, int, hipStream_t);                               \
  template void launch_lin_pack_all<AT>(const LinPackDesc*, int, int, const float*, const float*, AT*, AT*, AT*, AT*, \
                                        hipStream_t);
VIT_INST(float)
VIT_INST(bf16)
#undef VIT_INST

extern "C" {

int mdd_op_layernorm(int dtype, int rows, int dim, float eps, const void* x, const void* x_t, const float* gamma,
                     const float* gamma_t, const float* beta, const float* beta_t, void* y, void* y_t, void* stream) {
  CHECK_ARG(rows > 0 && dim > 0 && x && gamma && beta, "null pointer / empty problem");
  CHECK_ARG((x_t != nullptr) == (gamma_t != nullptr) && (x_t != nullptr) == (beta_t != nullptr) &&
                (x_t != nullptr) == (y_t != nullptr), "tangent operands come together");
  CHECK_ARG(x_t || y, "output is null");
  CHECK_ARG(dtype == MDD_DTYPE_F32 || dtype == MDD_DTYPE_BF16, "dtype");
  const int ce = dtype == MDD_DTYPE_F32 ? 4 : 8;
  CHECK_ARG(dim % ce == 0 && dim / ce <= 256, "dim must be a multiple of the 16-byte chunk and at most 256 chunks");
  if (dtype == MDD_DTYPE_F32) ln_fwd<float>(rows, dim, eps, x, x_t, gamma, gamma_t, beta, beta_t, y, y_t, (hipStream_t)stream);
  else ln_fwd<bf16>(rows, dim, eps, x, x_t, gamma, gamma_t, beta, beta_t, y, y_t, (hipStream_t)stream);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int mdd_op_add_layernorm(int dtype, int rows, int dim, float eps, const void* a, const void* a_t, const void* b,
                         const void* b_t, void* s, void* s_t, const float* gamma, const float* gamma_t, const float* beta,
                         const float* beta_t, void* y, void* y_t, void* stream) {
  CHECK_ARG(rows > 0 && dim > 0 && s && gamma && beta, "null pointer / empty problem");
  const bool T = gamma_t != nullptr;
  CHECK_ARG(T ? (a_t && b_t && s_t && beta_t && y_t) : (a && b && y), "primal call: a, b, s, y; tangent call: s (primal sum), a_t, b_t, s_t, y_t");
  CHECK_ARG(dtype == MDD_DTYPE_F32 || dtype == MDD_DTYPE_BF16, "dtype");
  const int ce = dtype == MDD_DTYPE_F32 ? 4 : 8;
  CHECK_ARG(dim % ce == 0 && dim / ce <= 256, "dim must be a multiple of the 16-byte chunk and at most 256 chunks");
  const void* ra = T ? a_t : a; const void* rb = T ? b_t : b; void* rs = T ? s_t : s;
  if (dtype == MDD_DTYPE_F32) ln_fwd<float>(rows, dim, eps, s, s_t, gamma, gamma_t, beta, beta_t, y, y_t, (hipStream_t)stream, ra, rb, rs);
  else ln_fwd<bf16>(rows, dim, eps, s, s_t, gamma, gamma_t, beta, beta_t, y, y_t, (hipStream_t)stream, ra, rb, rs);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int mdd_op_layernorm_bwd(int dtype, int rows, int dim, float eps, const void* x, const void* x_t, const void* dy,
                         const void* dy_t, const float* gamma, const float* gamma_t, const void* res,
                         const void* res_t, void* dx, void* dx_t, float* dgamma, float* dgamma_t, float* dbeta,
                         float* dbeta_t, void* stream) {
  CHECK_ARG(rows > 0 && dim > 0 && x && dy && gamma, "null pointer / empty problem");
  const bool T = x_t != nullptr;
  CHECK_ARG(T == (dy_t != nullptr) && T == (gamma_t != nullptr) && T == (dx_t != nullptr) &&
                T == (dgamma_t != nullptr) && T == (dbeta_t != nullptr), "tangent operands come together");
  CHECK_ARG(!res || !T || res_t, "the residual gradient needs its tangent in a tangent call");
  CHECK_ARG(T || (dx && dgamma && dbeta), "output is null");
  CHECK_ARG(dtype == MDD_DTYPE_F32 || dtype == MDD_DTYPE_BF16, "dtype");
  const int ce = dtype == MDD_DTYPE_F32 ? 4 : 8;
  CHECK_ARG(dim % ce == 0 && dim / ce <= 192, "dim must be a multiple of the 16-byte chunk and at most 192 chunks");
  if (dtype == MDD_DTYPE_F32) ln_bwd<float>(rows, dim, eps, x, x_t, dy, dy_t, gamma, gamma_t, res, res_t, dx, dx_t, dgamma, dgamma_t, dbeta, dbeta_t, (hipStream_t)stream);
  else ln_bwd<bf16>(rows, dim, eps, x, x_t, dy, dy_t, gamma, gamma_t, res, res_t, dx, dx_t, dgamma, dgamma_t, dbeta, dbeta_t, (hipStream_t)stream);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int mdd_op_gelu(int dtype, int64_t n, const void* c, const void* c_t, void* a, void* a_t, void* stream) {
  CHECK_ARG(n > 0 && c && (c_t ? a_t != nullptr : a != nullptr), "null pointer / empty problem");
  CHECK_ARG(dtype == MDD_DTYPE_F32 || dtype == MDD_DTYPE_BF16, "dtype");
  const int ce = dtype == MDD_DTYPE_F32 ? 4 : 8;
  CHECK_ARG(n % ce == 0, "element count must be a multiple of the 16-byte chunk");
  const int64_t ch = n / ce;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDD_DTYPE_F32) {
    if (c_t) k_gelu_fwd<Dual, float><<<vgrid(ch), 256, 0, st>>>((const float*)c, (const float*)c_t, (float*)a, (float*)a_t, ch);
    else k_gelu_fwd<float, float><<<vgrid(ch), 256, 0, st>>>((const float*)c, nullptr, (float*)a, nullptr, ch);
  } else {
    if (c_t) k_gelu_fwd<Dual, bf16><<<vgrid(ch), 256, 0, st>>>((const bf16*)c, (const bf16*)c_t, (bf16*)a, (bf16*)a_t, ch);
    else k_gelu_fwd<float, bf16><<<vgrid(ch), 256, 0, st>>>((const bf16*)c, nullptr, (bf16*)a, nullptr, ch);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int mdd_op_gelu_bwd(int dtype, int64_t n, const void* c, const void* c_t, const void* abar, const void* abar_t,
                    void* cbar, void* cbar_t, void* stream) {
  CHECK_ARG(n > 0 && c && abar, "null pointer / empty problem");
  const bool T = c_t != nullptr;
  CHECK_ARG(T == (abar_t != nullptr) && (T ? cbar_t != nullptr : cbar != nullptr), "tangent operands come together");
  CHECK_ARG(dtype == MDD_DTYPE_F32 || dtype == MDD_DTYPE_BF16, "dtype");
  const int ce = dtype == MDD_DTYPE_F32 ? 4 : 8;
  CHECK_ARG(n % ce == 0, "element count must be a multiple of the 16-byte chunk");
  const int64_t ch = n / ce;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDD_DTYPE_F32) {
    if (T) k_gelu_bwd<Dual, float><<<vgrid(ch), 256, 0, st>>>((const float*)c, (const float*)c_t, (const float*)abar, (const float*)abar_t, (float*)cbar, (float*)cbar_t, ch);
    else k_gelu_bwd<float, float><<<vgrid(ch), 256, 0, st>>>((const float*)c, nullptr, (const float*)abar, nullptr, (float*)cbar, nullptr, ch);
  } else {
    if (T) k_gelu_bwd<Dual, bf16><<<vgrid(ch), 256, 0, st>>>((const bf16*)c, (const bf16*)c_t, (const bf16*)abar, (const bf16*)abar_t, (bf16*)cbar, (bf16*)cbar_t, ch);
    else k_gelu_bwd<float, bf16><<<vgrid(ch), 256, 0, st>>>((const bf16*)c, nullptr, (const bf16*)abar, nullptr, (bf16*)cbar, nullptr, ch);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int mdd_op_softmax(int dtype, int64_t rows, int cols, int ld, float scale, const void* s, const void* s_t, void* p,
                   void* p_t, void* stream) {
  CHECK_ARG(rows > 0 && cols > 0 && cols <= 64 * SM_MAXE && ld >= cols && s, "rows / cols (<= 512) / ld");
  CHECK_ARG(s_t ? p_t != nullptr : p != nullptr, "output is null");
  CHECK_ARG(dtype == MDD_DTYPE_F32 || dtype == MDD_DTYPE_BF16, "dtype");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDD_DTYPE_F32) {
    if (s_t) k_softmax_fwd<Dual, float><<<grid, 256, 0, st>>>((const float*)s, (const float*)s_t, (float*)p, (float*)p_t, rows, cols, ld, scale);
    else k_softmax_fwd<float, float><<<grid, 256, 0, st>>>((const float*)s, nullptr, (float*)p, nullptr, rows, cols, ld, scale);
  } else {
    if (s_t) k_softmax_fwd<Dual, bf16><<<grid, 256, 0, st>>>((const bf16*)s, (const bf16*)s_t, (bf16*)p, (bf16*)p_t, rows, cols, ld, scale);
    else k_softmax_fwd<float, bf16><<<grid, 256, 0, st>>>((const bf16*)s, nullptr, (bf16*)p, nullptr, rows, cols, ld, scale);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int mdd_op_softmax_bwd(int dtype, int64_t rows, int cols, int ld, float scale, const void* p, const void* p_t,
                       const void* dp, const void* dp_t, void* ds, void* ds_t, void* stream) {
  CHECK_ARG(rows > 0 && cols > 0 && cols <= 64 * SM_MAXE && ld >= cols && p && dp, "rows / cols (<= 512) / ld");
  const bool T = p_t != nullptr;
  CHECK_ARG(T == (dp_t != nullptr) && (T ? ds_t != nullptr : ds != nullptr), "tangent operands come together");
  CHECK_ARG(dtype == MDD_DTYPE_F32 || dtype == MDD_DTYPE_BF16, "dtype");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDD_DTYPE_F32) {
    if (T) k_softmax_bwd<Dual, float><<<grid, 256, 0, st>>>((const float*)p, (const float*)p_t, (const float*)dp, (const float*)dp_t, (float*)ds, (float*)ds_t, rows, cols, ld, scale);
    else k_softmax_bwd<float, float><<<grid, 256, 0, st>>>((const float*)p, nullptr, (const float*)dp, nullptr, (float*)ds, nullptr, rows, cols, ld, scale);
  } else {
    if (T) k_softmax_bwd<Dual, bf16><<<grid, 256, 0, st>>>((const bf16*)p, (const bf16*)p_t, (const bf16*)dp, (const bf16*)dp_t, (bf16*)ds, (bf16*)ds_t, rows, cols, ld, scale);
    else k_softmax_bwd<float, bf16><<<grid, 256, 0, st>>>((const bf16*)p, nullptr, (const bf16*)dp, nullptr, (bf16*)ds, nullptr, rows, cols, ld, scale);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int mdd_op_bgemm(int dtype, int a_is_f32, int c_is_f32, const mdd_bgemm_desc* d, const void* A, const void* A_t,
                 const void* B, const void* B_t, void* C, void* C_t, void* stream) {
  CHECK_ARG(d && A && B, "null pointer");
  CHECK_ARG(d->m > 0 && d->n > 0 && d->k > 0 && d->outer > 0 && d->inner > 0, "empty problem");
  CHECK_ARG((int64_t)d->outer * d->inner <= 65535, "more than 65535 batches");
  CHECK_ARG(dtype == MDD_DTYPE_F32 || dtype == MDD_DTYPE_BF16 || dtype == MDD_DTYPE_BF16X2, "dtype");
  const bool T = C_t != nullptr;
  CHECK_ARG(T || C, "output is null");
  CHECK_ARG(!T || A_t || B_t, "a tangent call needs at least one tangent operand");
  BG p;
  p.A = A; p.A_t = A_t; p.B = B; p.B_t = B_t; p.C = C; p.C_t = C_t;
  p.M = d->m; p.N = d->n; p.K = d->k; p.inner = d->inner;
  p.sam = d->a_row; p.sak = d->a_col; p.sbk = d->b_row; p.sbn = d->b_col; p.scm = d->c_row; p.scn = d->c_col;
  p.ao = d->a_outer; p.aq = d->a_inner; p.bo = d->b_outer; p.bq = d->b_inner; p.co = d->c_outer; p.cq = d->c_inner;
  p.alpha = d->alpha;
  dim3 grid((unsigned)((d->n + 63) / 64), (unsigned)((d->m + 63) / 64), (unsigned)(d->outer * d->inner));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MDD_DTYPE_BF16X2) {          // fp32 storage, split-bf16 contraction on the matrix cores
    if (T) k_bgemm_mfma_split<true><<<grid, 256, 0, st>>>(p);
    else k_bgemm_mfma_split<false><<<grid, 256, 0, st>>>(p);
  } else if (dtype == MDD_DTYPE_F32) {
    if (T) k_bgemm<Dual, float, float, float><<<grid, 256, 0, st>>>(p);
    else k_bgemm<float, float, float, float><<<grid, 256, 0, st>>>(p);
  } else {
    const bool done = MDD_VIT_MFMA_ATTENTION && (T ? bgemm_mfma<true>(p, a_is_f32, c_is_f32, grid, st)
                                                     : bgemm_mfma<false>(p, a_is_f32, c_is_f32, grid, st));
    if (!done) {
      if (T) bgemm_types<Dual, bf16>(p, a_is_f32, c_is_f32, grid, st);
      else bgemm_types<float, bf16>(p, a_is_f32, c_is_f32, grid, st);
    }
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // extern "C"
