// Small-M fp32 linear layers (M = number of synthetic pairs, 10..500): squeeze-excite fc1/fc2 of
// every NFNet block and the text ProjectionHead (reference networks.py:625-646).  These read the
// flat theta directly (nn.Linear / 1x1 conv weight is [out, in] row-major) -- no repacking.
// Weight-bandwidth bound (each weight is used M times), so: coalesced weight reads, fp32 FMA.
// Generic over S in {float, Dual}; in the Dual instantiation any tangent pointer may be null (=0).
#include "kernels.h"

namespace {

template <class S> DEVI S ldz(const float* pv, const float* pt, size_t i) {
  if constexpr (IsDual<S>::v) return Dual(pv[i], pt ? pt[i] : 0.f);
  else return pv[i];
}

constexpr int NB = 8;  // rows of x per wave

// y[n,j] = act(sum_k x[n,k] W[j,k] + b[j]) ; one wave per (j, block of NB rows)
template <class S>
__global__ void k_linear_fwd(float* __restrict__ y, float* __restrict__ y_t,
                             const float* __restrict__ x, const float* __restrict__ x_t,
                             const float* __restrict__ W, const float* __restrict__ W_t,
                             const float* __restrict__ b, const float* __restrict__ b_t, int n,
                             int k, int j, int act) {
  int nblocks = (n + NB - 1) / NB;
  int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wid >= (int64_t)j * nblocks) return;
  int lane = threadIdx.x & 63;
  int jj = (int)(wid / nblocks);
  int n0 = (int)(wid - (int64_t)jj * nblocks) * NB;
  S acc[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) acc[r] = mk<S>(0.f, 0.f);
  for (int kk = lane; kk < k; kk += WAVE) {
    S w = ldz<S>(W, W_t, (size_t)jj * k + kk);
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      int nn = n0 + r;
      if (nn < n) acc[r] = acc[r] + ldz<S>(x, x_t, (size_t)nn * k + kk) * w;
    }
  }
  S bias = b ? ldz<S>(b, b_t, jj) : mk<S>(0.f, 0.f);
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    S s = wave_sum(acc[r]) + bias;
    if (act == 1) s = relu_(s);
    else if (act == 2) s = sigmoid_(s);
    int nn = n0 + r;
    if (lane == 0 && nn < n) stS<S>(y, y_t, (size_t)nn * j + jj, s);
  }
}

constexpr int DB = 4;  // rows per thread in dgrad
// dx[n,k] = sum_j dy[n,j] W[j,k] ; thread per (k, block of DB rows)
template <class S>
__global__ void k_linear_dgrad(float* __restrict__ dx, float* __restrict__ dx_t,
                               const float* __restrict__ dy, const float* __restrict__ dy_t,
                               const float* __restrict__ W, const float* __restrict__ W_t, int n,
                               int k, int j) {
  int kk = blockIdx.x * blockDim.x + threadIdx.x;
  int n0 = blockIdx.y * DB;
  if (kk >= k) return;
  S acc[DB];
#pragma unroll
  for (int r = 0; r < DB; ++r) acc[r] = mk<S>(0.f, 0.f);
  for (int jj = 0; jj < j; ++jj) {
    S w = ldz<S>(W, W_t, (size_t)jj * k + kk);
#pragma unroll
    for (int r = 0; r < DB; ++r) {
      int nn = n0 + r;
      if (nn < n) acc[r] = acc[r] + ldz<S>(dy, dy_t, (size_t)nn * j + jj) * w;
    }
  }
#pragma unroll
  for (int r = 0; r < DB; ++r) {
    int nn = n0 + r;
    if (nn < n) stS<S>(dx, dx_t, (size_t)nn * k + kk, acc[r]);
  }
}

// dW[j,k] = sum_n dy[n,j] x[n,k] ; db[j] = sum_n dy[n,j] ; thread per (j,k)
template <class S>
__global__ void k_linear_wgrad(float* __restrict__ dW, float* __restrict__ db,
                               const float* __restrict__ dy, const float* __restrict__ dy_t,
                               const float* __restrict__ x, const float* __restrict__ x_t, int n,
                               int k, int j) {
  int kk = blockIdx.x * blockDim.x + threadIdx.x;
  int jj = blockIdx.y;
  if (kk >= k) return;
  S acc = mk<S>(0.f, 0.f), accb = mk<S>(0.f, 0.f);
  for (int nn = 0; nn < n; ++nn) {
    S d = ldz<S>(dy, dy_t, (size_t)nn * j + jj);
    acc = acc + d * ldz<S>(x, x_t, (size_t)nn * k + kk);
    accb = accb + d;
  }
  dW[(size_t)jj * k + kk] = IsDual<S>::v ? tan_(acc) : val(acc);
  if (kk == 0 && db) db[jj] = IsDual<S>::v ? tan_(accb) : val(accb);
}

}  // namespace

void launch_linear_fwd(float* y, float* y_t, const float* x, const float* x_t, const float* W,
                       const float* W_t, const float* b, const float* b_t, int n, int k, int j,
                       int act, hipStream_t st) {
  int64_t waves = (int64_t)j * ((n + NB - 1) / NB);
  int grid = (int)((waves + 3) / 4);
  if (y_t) k_linear_fwd<Dual><<<grid, 256, 0, st>>>(y, y_t, x, x_t, W, W_t, b, b_t, n, k, j, act);
  else k_linear_fwd<float><<<grid, 256, 0, st>>>(y, nullptr, x, nullptr, W, nullptr, b, nullptr, n, k, j, act);
}
void launch_linear_dgrad(float* dx, float* dx_t, const float* dy, const float* dy_t,
                         const float* W, const float* W_t, int n, int k, int j, hipStream_t st) {
  dim3 grid((k + 255) / 256, (n + DB - 1) / DB);
  if (dx_t) k_linear_dgrad<Dual><<<grid, 256, 0, st>>>(dx, dx_t, dy, dy_t, W, W_t, n, k, j);
  else k_linear_dgrad<float><<<grid, 256, 0, st>>>(dx, nullptr, dy, nullptr, W, nullptr, n, k, j);
}
void launch_linear_wgrad(float* dW, float* db, const float* dy, const float* dy_t, const float* x,
                         const float* x_t, int n, int k, int j, hipStream_t st) {
  int bs = k >= 256 ? 256 : 64;
  dim3 grid((k + bs - 1) / bs, j);
  if (dy_t || x_t) k_linear_wgrad<Dual><<<grid, bs, 0, st>>>(dW, db, dy, dy_t, x, x_t, n, k, j);
  else k_linear_wgrad<float><<<grid, bs, 0, st>>>(dW, db, dy, nullptr, x, nullptr, n, k, j);
}
