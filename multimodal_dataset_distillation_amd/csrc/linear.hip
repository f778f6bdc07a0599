// Small-M fp32 linear layers (M = number of synthetic pairs, 10..500): squeeze-excite fc1/fc2 of
// every NFNet block and the text ProjectionHead (reference networks.py:625-646).  They read the
// flat theta directly (nn.Linear / 1x1-conv weight is [out, in] row-major) -- no repacking.
//
// One LDS-tiled SGEMM kernel serves the three contractions through operand strides:
//   forward  y[n,j]  = sum_k x[n,k]  W[j,k]      A = x  (k contiguous), B = W (k contiguous)
//   dgrad    dx[n,k] = sum_j dy[n,j] W[j,k]      A = dy (j contiguous), B = W (n' contiguous)
//   wgrad    dW[j,k] = sum_n dy[n,j] x[n,k]      A = dy (m contiguous), B = x (n' contiguous)
// 64x64 output tile, 16-deep K step, 4x4 register micro-tile per thread, K split over blockIdx.z
// with fp32 atomics when the tile count alone cannot fill 256 CUs (M is tiny).
// Tangent pass (forward-over-reverse): C_t = A_t*B + A*B_t in the same loop; either tangent
// operand may be absent (= 0).
#include <algorithm>

#include "kernels.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDP = 68;  // LDS row pitch (floats): 16B-aligned, 2-way max on writes

struct GArgs {
  const float *A, *A_t, *B, *B_t;
  float* C;             // primal result (MODE 0) or tangent result (MODE 1)
  int M, N, K;
  int64_t sAm, sAk, sBk, sBn;   // element strides
  int ksplit_len;       // K range per blockIdx.z (multiple of BK)
  int atomic;           // 1: atomicAdd into zeroed C, 0: plain store
};

// MODE 0: C = A*B ; MODE 1: C = A_t*B + A*B_t (null tangent = 0)
template <int MODE>
__global__ __launch_bounds__(256) void k_sgemm_small(const GArgs p) {
  __shared__ __attribute__((aligned(16))) float As[BK][LDP], Bs[BK][LDP];
  __shared__ __attribute__((aligned(16))) float Ats[MODE ? BK : 1][LDP], Bts[MODE ? BK : 1][LDP];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.ksplit_len;
  const int kend = min(p.K, kbeg + p.ksplit_len);
  const int tm = (tid >> 4) * 4, tn = (tid & 15) * 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const bool a_kfast = p.sAk == 1;   // pick the loader mapping that coalesces along the unit stride
  const bool b_kfast = p.sBk == 1;
  const bool hasAt = MODE && p.A_t != nullptr, hasBt = MODE && p.B_t != nullptr;

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int e = tid + 256 * i;
      int kk = a_kfast ? (e & (BK - 1)) : (e >> 6);
      int mm = a_kfast ? (e >> 4) : (e & 63);
      bool ok = (m0 + mm) < p.M && (k0 + kk) < kend;
      int64_t off = (int64_t)(m0 + mm) * p.sAm + (int64_t)(k0 + kk) * p.sAk;
      As[kk][mm] = ok ? p.A[off] : 0.f;
      if (MODE) Ats[kk][mm] = (ok && hasAt) ? p.A_t[off] : 0.f;
      int kb = b_kfast ? (e & (BK - 1)) : (e >> 6);
      int nn = b_kfast ? (e >> 4) : (e & 63);
      bool okb = (n0 + nn) < p.N && (k0 + kb) < kend;
      int64_t offb = (int64_t)(k0 + kb) * p.sBk + (int64_t)(n0 + nn) * p.sBn;
      Bs[kb][nn] = okb ? p.B[offb] : 0.f;
      if (MODE) Bts[kb][nn] = (okb && hasBt) ? p.B_t[offb] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      float4 a = *(const float4*)&As[kk][tm];
      float4 b = *(const float4*)&Bs[kk][tn];
      float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
      } else {
        float4 at = *(const float4*)&Ats[kk][tm];
        float4 bt = *(const float4*)&Bts[kk][tn];
        float atv[4] = {at.x, at.y, at.z, at.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = fmaf(av[i], btv[j], fmaf(atv[i], bv[j], acc[i][j]));
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + tm + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = n0 + tn + j;
      if (n >= p.N) continue;
      float* dst = p.C + (int64_t)m * p.N + n;
      if (p.atomic) atomicAdd(dst, acc[i][j]);
      else *dst = acc[i][j];
    }
  }
}

template <class S> DEVI S ldz(const float* pv, const float* pt, size_t i) {
  if constexpr (IsDual<S>::v) return Dual(pv[i], pt ? pt[i] : 0.f);
  else return pv[i];
}
// y = act(acc + b)    acc: primal GEMM result in y (MODE 0) / tangent GEMM result in y_t (MODE 1)
template <class S>
__global__ void k_bias_act(float* __restrict__ y, float* __restrict__ y_t,
                           const float* __restrict__ b, const float* __restrict__ b_t, int n, int j,
                           int act) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)n * j;
       i += (int64_t)gridDim.x * blockDim.x) {
    int jj = (int)(i % j);
    S s;
    if constexpr (IsDual<S>::v) {
      // y holds the stashed POST-activation primal; recover what the tangent rule needs from it
      float yt = y_t[i] + (b_t ? b_t[jj] : 0.f);
      float yv = y[i];
      float d = act == 1 ? (yv > 0.f ? 1.f : 0.f) : (act == 2 ? yv * (1.f - yv) : 1.f);
      y_t[i] = d * yt;
      continue;
    } else {
      s = y[i] + (b ? b[jj] : 0.f);
      if (act == 1) s = relu_(s);
      else if (act == 2) s = sigmoid_(s);
      y[i] = s;
    }
  }
}
// db[j] = sum_n dy[n,j]   (tangent: of dy_t)
__global__ void k_colsum(float* __restrict__ db, const float* __restrict__ dy, int n, int j) {
  int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj >= j) return;
  float s = 0.f;
  for (int nn = 0; nn < n; ++nn) s += dy[(size_t)nn * j + jj];
  db[jj] = s;
}

void run_gemm(bool tangent, const float* A, const float* A_t, const float* B, const float* B_t,
              float* C, int M, int N, int K, int64_t sAm, int64_t sAk, int64_t sBk, int64_t sBn,
              hipStream_t st) {
  GArgs g;
  g.A = A; g.A_t = A_t; g.B = B; g.B_t = B_t; g.C = C; g.M = M; g.N = N; g.K = K;
  g.sAm = sAm; g.sAk = sAk; g.sBk = sBk; g.sBn = sBn;
  int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int splits = 1;
  if (tiles < 256) {
    splits = (512 + tiles - 1) / tiles;
    int maxs = (K + 4 * BK - 1) / (4 * BK);   // at least 4 K-steps per block
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  int len = (K + splits - 1) / splits;
  len = (len + BK - 1) / BK * BK;
  splits = (K + len - 1) / len;
  g.ksplit_len = len;
  g.atomic = splits > 1;
  if (g.atomic) (void)hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), st);
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, splits);
  if (tangent) k_sgemm_small<1><<<grid, 256, 0, st>>>(g);
  else k_sgemm_small<0><<<grid, 256, 0, st>>>(g);
}

}  // namespace

// y[n,j] = act(sum_k x[n,k] W[j,k] + b[j])
void launch_linear_fwd(float* y, float* y_t, const float* x, const float* x_t, const float* W,
                       const float* W_t, const float* b, const float* b_t, int n, int k, int j,
                       int act, hipStream_t st) {
  int grid = (int)std::min<int64_t>(((int64_t)n * j + 255) / 256, 2048);
  if (y_t) {
    run_gemm(true, x, x_t, W, W_t, y_t, n, j, k, k, 1, 1, k, st);
    k_bias_act<Dual><<<grid, 256, 0, st>>>(y, y_t, b, b_t, n, j, act);
  } else {
    run_gemm(false, x, nullptr, W, nullptr, y, n, j, k, k, 1, 1, k, st);
    k_bias_act<float><<<grid, 256, 0, st>>>(y, nullptr, b, nullptr, n, j, act);
  }
}
// dx[n,k] = sum_j dy[n,j] W[j,k]
void launch_linear_dgrad(float* dx, float* dx_t, const float* dy, const float* dy_t,
                         const float* W, const float* W_t, int n, int k, int j, hipStream_t st) {
  if (dx_t) run_gemm(true, dy, dy_t, W, W_t, dx_t, n, k, j, j, 1, k, 1, st);
  else run_gemm(false, dy, nullptr, W, nullptr, dx, n, k, j, j, 1, k, 1, st);
}
// dW[j,k] = sum_n dy[n,j] x[n,k] ; db[j] = sum_n dy[n,j]   (tangent pass writes tangents only)
void launch_linear_wgrad(float* dW, float* db, const float* dy, const float* dy_t, const float* x,
                         const float* x_t, int n, int k, int j, hipStream_t st) {
  bool tangent = dy_t || x_t;
  if (tangent) run_gemm(true, dy, dy_t, x, x_t, dW, j, k, n, 1, j, k, 1, st);
  else run_gemm(false, dy, nullptr, x, nullptr, dW, j, k, n, 1, j, k, 1, st);
  if (db) {
    if (tangent && !dy_t) (void)hipMemsetAsync(db, 0, (size_t)j * sizeof(float), st);
    else k_colsum<<<(j + 255) / 256, 256, 0, st>>>(db, tangent ? dy_t : dy, n, j);
  }
}
