// HBM-bound streaming kernels over the flattened student parameters (fp32 master copies).
// reference distill.py:582-583 (theta' = theta - lr*g, lr a differentiable device scalar),
// :588-598 (sum-MSE ratios), :233-241/:611-613 (SGD momentum 0.5).
// All scalars (lr, distances) stay on the device: no host sync inside an iteration.
#include "kernels.h"

namespace {

constexpr int kBlock = 256;
inline int grid_for(int64_t n4) {  // n4 = number of float4 items
  int64_t g = (n4 + kBlock - 1) / kBlock;
  if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void k_axpy_out(float* __restrict__ y, const float* __restrict__ x,
                           const float* __restrict__ g, const float* __restrict__ lr, float sign,
                           int64_t n) {
  const float a = sign * lr[0];
  int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 xv = ((const float4*)x)[i], gv = ((const float4*)g)[i];
    ((float4*)y)[i] = make_float4(xv.x + a * gv.x, xv.y + a * gv.y, xv.z + a * gv.z, xv.w + a * gv.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    y[i] = x[i] + a * g[i];
  }
}

__global__ void k_scale_out(float* __restrict__ y, const float* __restrict__ x,
                            const float* __restrict__ coef, float mul, int64_t n) {
  const float a = mul * (coef ? coef[0] : 1.f);
  int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 xv = ((const float4*)x)[i];
    ((float4*)y)[i] = make_float4(a * xv.x, a * xv.y, a * xv.z, a * xv.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    y[i] = a * x[i];
  }
}

// out[0] += sum (a-b)^2  (double accumulator; out zeroed by the caller)
__global__ void k_sqdist(const float* __restrict__ a, const float* __restrict__ b,
                         double* __restrict__ out, int64_t n) {
  double acc = 0.0;
  int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 av = ((const float4*)a)[i], bv = ((const float4*)b)[i];
    float d0 = av.x - bv.x, d1 = av.y - bv.y, d2 = av.z - bv.z, d3 = av.w - bv.w;
    acc += (double)(d0 * d0 + d1 * d1) + (double)(d2 * d2 + d3 * d3);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    float d = a[i] - b[i];
    acc += (double)d * d;
  }
  acc = wave_sum_d(acc);
  __shared__ double sh[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int i = 0; i < kBlock / 64; ++i) s += sh[i];
    atomicAdd(out, s);
  }
}

// out[0] += sign * sum a*b
__global__ void k_dot(const float* __restrict__ a, const float* __restrict__ b,
                      double* __restrict__ out, double sign, int64_t n) {
  double acc = 0.0;
  int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 av = ((const float4*)a)[i], bv = ((const float4*)b)[i];
    acc += (double)(av.x * bv.x + av.y * bv.y) + (double)(av.z * bv.z + av.w * bv.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    acc += (double)a[i] * b[i];
  }
  acc = wave_sum_d(acc);
  __shared__ double sh[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int i = 0; i < kBlock / 64; ++i) s += sh[i];
    atomicAdd(out, sign * s);
  }
}

// lam = 2*(thetaK - target)/dist0     (d grand_loss / d theta_K, reference distill.py:588-597)
__global__ void k_lambda_init(float* __restrict__ lam, const float* __restrict__ thK,
                              const float* __restrict__ tgt, const double* __restrict__ dist0,
                              int64_t n) {
  const float c = (float)(2.0 / dist0[0]);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    lam[i] = c * (thK[i] - tgt[i]);
}

__global__ void k_sub_inplace(float* __restrict__ y, const float* __restrict__ x, int64_t n) {
  int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 yv = ((float4*)y)[i], xv = ((const float4*)x)[i];
    ((float4*)y)[i] = make_float4(yv.x - xv.x, yv.y - xv.y, yv.z - xv.z, yv.w - xv.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    y[i] -= x[i];
  }
}

// torch.optim.SGD(momentum=m, dampening=0): buf = g (first) | m*buf+g ; p -= lr*buf
__global__ void k_sgd_momentum(float* __restrict__ p, const float* __restrict__ g,
                               float* __restrict__ buf, float lr, float mom, int first, int64_t n,
                               const float* __restrict__ skip) {
  // skip (optional, device): non-zero => the step is a no-op.  Lets the host enqueue the next iteration
  // before it has read this one's NaN flag (reference distill.py:599 breaks BEFORE the optimiser steps).
  if (skip != nullptr && skip[0] != 0.f) return;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float b = first ? g[i] : mom * buf[i] + g[i];
    buf[i] = b;
    p[i] -= lr * b;
  }
}

// scalars[0..3] = {img_numer, img_denom, txt_numer, txt_denom} (double)
// out[0] = grand_loss, out[1] = img_loss, out[2] = txt_loss     (reference distill.py:596-598)
__global__ void k_match_finalize(const double* __restrict__ s, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    double il = s[0] / s[1], tl = s[2] / s[3];
    out[0] = (float)(il + tl);
    out[1] = (float)il;
    out[2] = (float)tl;
  }
}

// out_f32[i] (+)= (float) in_f64[i] * mul   -- scalar hand-over (lr grads)
__global__ void k_d2f(float* __restrict__ out, const double* __restrict__ in, float mul, int accum,
                      int n) {
  int i = threadIdx.x;
  if (i < n) out[i] = (accum ? out[i] : 0.f) + (float)(in[i] * mul);
}

__global__ void k_accum_f2d(double* __restrict__ out, const float* __restrict__ in, double sign) {
  if (threadIdx.x == 0) out[0] += sign * (double)in[0];
}

}  // namespace

void launch_accum_f2d(double* out, const float* in, double sign, hipStream_t st) {
  k_accum_f2d<<<1, 64, 0, st>>>(out, in, sign);
}
void launch_axpy_out(float* y, const float* x, const float* g, const float* lr, float sign,
                     int64_t n, hipStream_t st) {
  k_axpy_out<<<grid_for(n >> 2), kBlock, 0, st>>>(y, x, g, lr, sign, n);
}
void launch_scale_out(float* y, const float* x, const float* coef, float mul, int64_t n,
                      hipStream_t st) {
  k_scale_out<<<grid_for(n >> 2), kBlock, 0, st>>>(y, x, coef, mul, n);
}
void launch_sqdist(const float* a, const float* b, double* out, int64_t n, hipStream_t st) {
  k_sqdist<<<grid_for(n >> 2), kBlock, 0, st>>>(a, b, out, n);
}
void launch_dot(const float* a, const float* b, double* out, double sign, int64_t n,
                hipStream_t st) {
  k_dot<<<grid_for(n >> 2), kBlock, 0, st>>>(a, b, out, sign, n);
}
void launch_lambda_init(float* lam, const float* thK, const float* tgt, const double* dist0,
                        int64_t n, hipStream_t st) {
  k_lambda_init<<<grid_for(n), kBlock, 0, st>>>(lam, thK, tgt, dist0, n);
}
void launch_sub_inplace(float* y, const float* x, int64_t n, hipStream_t st) {
  k_sub_inplace<<<grid_for(n >> 2), kBlock, 0, st>>>(y, x, n);
}
void launch_sgd_momentum(float* p, const float* g, float* buf, float lr, float mom, int first,
                         int64_t n, const float* skip, hipStream_t st) {
  k_sgd_momentum<<<grid_for(n), kBlock, 0, st>>>(p, g, buf, lr, mom, first, n, skip);
}
void launch_match_finalize(const double* s, float* out, hipStream_t st) {
  k_match_finalize<<<1, 64, 0, st>>>(s, out);
}
void launch_d2f(float* out, const double* in, float mul, int accum, int n, hipStream_t st) {
  k_d2f<<<1, 64, 0, st>>>(out, in, mul, accum, n);
}
