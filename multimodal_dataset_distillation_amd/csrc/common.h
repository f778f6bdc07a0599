// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the bi-trajectory
// co-distillation hot path.  wave = 64 lanes everywhere.
//
// Second-order support: every non-linear op on the path is written ONCE, generic over a scalar
// type S that is either `float` (primal pass) or `Dual` (value + directional derivative).  The
// "tangent" pass R_v{.} of the forward and of the hand-written backward formulas -- i.e. the
// Hessian-vector / mixed second derivatives that reference distill.py:562-567 + :606 obtain by
// autograd-through-autograd -- falls out of Dual arithmetic (forward-over-reverse).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DEVI __device__ __forceinline__
#define WAVE 64

// Work-skipping timing switches (drop K-steps / epilogue / loads / math / stores) exist ONLY in
// experiment builds (build_ext.build_variant(..., ["MDD_DEBUG_SWITCHES"])): the product library
// compiles them out, so no environment variable can turn a kernel into a partial no-op.
#ifdef MDD_DEBUG_SWITCHES
#define MDD_DBG_BITS(p) ((p).dbg)
#else
#define MDD_DBG_BITS(p) 0
#endif

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ------------------------------------------------------------------ storage <-> float
template <class T> DEVI float to_f(T x) { return (float)x; }
template <class T> DEVI T from_f(float x) { return (T)x; }
template <class T> struct ElemsPer16B;
template <> struct ElemsPer16B<float> { static constexpr int v = 4; };
template <> struct ElemsPer16B<bf16> { static constexpr int v = 8; };

DEVI float bf16_bits_to_f(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// unpack a 16-byte chunk into floats (8 for bf16, 4 for f32)
template <class T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  static DEVI void unpack(const uint4& c, float* f) {
    f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y);
    f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
  }
  static DEVI uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                      __float_as_uint(f[3]));
  }
};
template <> struct Chunk<bf16> {
  static constexpr int N = 8;
  static DEVI void unpack(const uint4& c, float* f) {
    f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
    f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
    f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
    f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
  }
  static DEVI unsigned pk(float lo, float hi) {
    bf16 a = (bf16)lo, b = (bf16)hi;  // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
    unsigned short ua = *(unsigned short*)&a, ub = *(unsigned short*)&b;
    return (unsigned)ua | ((unsigned)ub << 16);
  }
  static DEVI uint4 pack(const float* f) {
    return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
  }
};

// ------------------------------------------------------------------ dual numbers
struct Dual {
  float v, t;
  DEVI Dual() : v(0.f), t(0.f) {}
  DEVI Dual(float v_) : v(v_), t(0.f) {}
  DEVI Dual(float v_, float t_) : v(v_), t(t_) {}
};
DEVI Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.t + b.t); }
DEVI Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.t - b.t); }
DEVI Dual operator-(Dual a) { return Dual(-a.v, -a.t); }
DEVI Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.t * b.v + a.v * b.t); }
DEVI Dual operator*(Dual a, float b) { return Dual(a.v * b, a.t * b); }
DEVI Dual operator*(float a, Dual b) { return Dual(a * b.v, a * b.t); }
DEVI Dual operator+(Dual a, float b) { return Dual(a.v + b, a.t); }
DEVI Dual operator+(float a, Dual b) { return Dual(a + b.v, b.t); }
DEVI Dual operator-(Dual a, float b) { return Dual(a.v - b, a.t); }
DEVI Dual operator-(float a, Dual b) { return Dual(a - b.v, -b.t); }
DEVI Dual operator/(Dual a, Dual b) {
  float r = 1.f / b.v, q = a.v * r;
  return Dual(q, (a.t - q * b.t) * r);
}
DEVI Dual operator/(Dual a, float b) { float r = 1.f / b; return Dual(a.v * r, a.t * r); }
DEVI Dual& operator+=(Dual& a, Dual b) { a.v += b.v; a.t += b.t; return a; }

template <class S> struct IsDual { static constexpr bool v = false; };
template <> struct IsDual<Dual> { static constexpr bool v = true; };

DEVI float val(float x) { return x; }
DEVI float val(Dual x) { return x.v; }
DEVI float tan_(float) { return 0.f; }
DEVI float tan_(Dual x) { return x.t; }
template <class S> DEVI S mk(float v, float t);
template <> DEVI float mk<float>(float v, float) { return v; }
template <> DEVI Dual mk<Dual>(float v, float t) { return Dual(v, t); }

// generic loads / stores: primal pointer pv, tangent pointer pt (pt ignored for S=float).
// Dual stores write ONLY the tangent (primal values already sit in the stash).
template <class S, class T> DEVI S ldS(const T* pv, const T* pt, size_t i) {
  if constexpr (IsDual<S>::v) return Dual(to_f(pv[i]), to_f(pt[i]));
  else return to_f(pv[i]);
}
template <class S, class T> DEVI void stS(T* pv, T* pt, size_t i, S x) {
  if constexpr (IsDual<S>::v) pt[i] = from_f<T>(x.t);
  else pv[i] = from_f<T>(x);
}

// ------------------------------------------------------------------ scalar math, float + Dual
// v_exp_f32 + v_rcp_f32 (1 ulp): the full-precision division sequence costs ~10 VALU ops per element
// and the conv epilogues (SiLU and its derivatives on every output element) are VALU-bound.
DEVI float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
DEVI float rsqrt_(float x) { return rsqrtf(x); }
DEVI Dual rsqrt_(Dual x) { float r = rsqrtf(x.v); return Dual(r, -0.5f * r * r * r * x.t); }
DEVI float sqrt_(float x) { return sqrtf(x); }
DEVI Dual sqrt_(Dual x) { float r = sqrtf(x.v); return Dual(r, 0.5f * x.t / r); }
DEVI float exp_(float x) { return __expf(x); }
DEVI Dual exp_(Dual x) { float e = __expf(x.v); return Dual(e, e * x.t); }
DEVI float log_(float x) { return __logf(x); }
DEVI Dual log_(Dual x) { return Dual(__logf(x.v), x.t / x.v); }
DEVI float sigmoid_(float x) { return sigmoidf_(x); }
DEVI Dual sigmoid_(Dual x) { float s = sigmoidf_(x.v); return Dual(s, s * (1.f - s) * x.t); }
DEVI float relu_(float x) { return x > 0.f ? x : 0.f; }
DEVI Dual relu_(Dual x) { return x.v > 0.f ? x : Dual(0.f, 0.f); }

// SiLU f(x)=x*sigmoid(x);  f' = s(1+x(1-s));  f'' = s(1-s)(2+x(1-2s))
DEVI float silu_(float x) { return x * sigmoidf_(x); }
DEVI Dual silu_(Dual x) {
  float s = sigmoidf_(x.v);
  return Dual(x.v * s, s * (1.f + x.v * (1.f - s)) * x.t);
}
DEVI float dsilu_(float x) { float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }
DEVI Dual dsilu_(Dual x) {
  float s = sigmoidf_(x.v);
  return Dual(s * (1.f + x.v * (1.f - s)), s * (1.f - s) * (2.f + x.v * (1.f - 2.f * s)) * x.t);
}
// exact (erf) GELU, as nn.GELU() default: g = x*Phi(x); g' = Phi + x*phi; g'' = phi*(2 - x^2)
DEVI float gelu_(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
DEVI float dgelu_(float x) {
  float Phi = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  float phi = 0.3989422804014327f * __expf(-0.5f * x * x);
  return Phi + x * phi;
}
DEVI Dual gelu_(Dual x) { return Dual(gelu_(x.v), dgelu_(x.v) * x.t); }
DEVI Dual dgelu_(Dual x) {
  float phi = 0.3989422804014327f * __expf(-0.5f * x.v * x.v);
  return Dual(dgelu_(x.v), phi * (2.f - x.v * x.v) * x.t);
}

// the same functions for bf16 storage (k_gemm_pipe's epilogues): Phi from the rational approximation of erf (Abramowitz &
// Stegun 7.1.26, |error| < 1.5e-7: three orders below bf16's 2^-9) with ONE exp(-x^2/2) shared by Phi and phi -- the
// library erff plus a second exponential were 19 ms of a 440 ms configs[4] iteration
DEVI void gelu_parts_fast(float x, float& Phi, float& phi) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float e = __expf(-0.5f * x * x);                     // = exp(-(x / sqrt 2)^2)
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  Phi = 0.5f * (1.f + copysignf(1.f - poly * e, x));
  phi = 0.3989422804014327f * e;
}
DEVI float gelu_fast_(float x) { float Phi, phi; gelu_parts_fast(x, Phi, phi); return x * Phi; }
DEVI float dgelu_fast_(float x) { float Phi, phi; gelu_parts_fast(x, Phi, phi); return Phi + x * phi; }
DEVI Dual dgelu_fast_(Dual x) {
  float Phi, phi;
  gelu_parts_fast(x.v, Phi, phi);
  return Dual(Phi + x.v * phi, phi * (2.f - x.v * x.v) * x.t);
}

// ------------------------------------------------------------------ wave / block reductions
DEVI float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
DEVI Dual wave_sum(Dual x) { return Dual(wave_sum(x.v), wave_sum(x.t)); }
DEVI float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
DEVI double wave_sum_d(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
// block-wide sum (blockDim.x multiple of 64, <= 1024); scratch >= 16 floats; result in all threads
DEVI float block_sum(float x, float* scratch) {
  x = wave_sum(x);
  int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = x;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}
DEVI Dual block_sum(Dual x, float* scratch) {
  return Dual(block_sum(x.v, scratch), block_sum(x.t, scratch));
}

#define HIP_CHECK_RET(expr)                                   \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return mdd_set_error(_e, #expr);    \
  } while (0)
