// Small-M fp32 linear layers (M = number of synthetic pairs, 10..500): squeeze-excite fc1/fc2 of
// every NFNet block and the text ProjectionHead (reference networks.py:625-646).  They read the
// flat theta directly (nn.Linear / 1x1-conv weight is [out, in] row-major) -- no repacking.
//
// One LDS-tiled SGEMM kernel serves the three contractions through operand strides:
//   forward  y[n,j]  = sum_k x[n,k]  W[j,k]      A = x  (k contiguous), B = W (k contiguous)
//   dgrad    dx[n,k] = sum_j dy[n,j] W[j,k]      A = dy (j contiguous), B = W (n' contiguous)
//   wgrad    dW[j,k] = sum_n dy[n,j] x[n,k]      A = dy (m contiguous), B = x (n' contiguous)
// 64x64 output tile, 16-deep K step, 4x4 register micro-tile per thread, 16-byte global loads along
// whichever operand dimension is contiguous, next K slab prefetched into registers while the current
// one is multiplied out of (double-buffered) LDS.
//
// M is tiny, so the tile count alone cannot fill 256 CUs: K is split over blockIdx.z.  Partial tiles
// are added (fp32 atomics) into a per-stream scratch accumulator; the LAST block to arrive at a tile
// (device-scope counter, self-resetting) swaps the sums out and applies the epilogue -- bias +
// activation, the tangent of that, the ReLU mask of the SE backward, the bias-gradient row sums --
// so a layer is ONE launch (no zero-fill + separate pointwise kernels).
//
// Tangent pass (forward-over-reverse): C_t = A_t*B + A*B_t in the same loop; either tangent
// operand may be absent (= 0).
#include <algorithm>

#include "kernels.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDP = 68;  // LDS row pitch (floats): 16B-aligned rows

enum { EP_PLAIN = 0, EP_BIAS_ACT = 1, EP_BIAS_ACT_T = 2, EP_RELU_MASK = 3 };

struct GArgs {
  const float *A, *A_t, *B, *B_t;
  float* C;             // primal result (MODE 0) or tangent result (MODE 1)
  int M, N, K;
  int64_t sAm, sAk, sBk, sBn;   // element strides
  int ksplit_len;       // K range per blockIdx.z (multiple of BK)
  int splits;
  int vecA, vecB, vecC; // 16-byte access legal for the operand (alignment + extents)
  int epi, act;
  const float* bias;    // EP_BIAS_ACT: b ; EP_BIAS_ACT_T: b_t (may be null)
  const float* aux;     // EP_BIAS_ACT_T: stashed post-activation y ; EP_RELU_MASK: h
  float* rowsum;        // optional: rowsum[m] = sum_k (MODE ? A_t : A)[m,k]   (bias gradient)
  float* part; float* part_rs; unsigned* ctr;
};

// MODE 0: C = A*B ; MODE 1: C = A_t*B + A*B_t (null tangent = 0)
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_sgemm_small(const GArgs p) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDP], Bs[2][BK][LDP];
  __shared__ __attribute__((aligned(16))) float Ats[MODE ? 2 : 1][MODE ? BK : 1][LDP],
      Bts[MODE ? 2 : 1][MODE ? BK : 1][LDP];
  __shared__ int s_last;
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
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool a_kfast = p.sAk == 1;   // which operand dimension is the unit stride
  const bool b_kfast = p.sBk == 1;
  const bool hasAt = MODE && p.A_t != nullptr, hasBt = MODE && p.B_t != nullptr;
  const bool do_rs = p.rowsum != nullptr && blockIdx.x == 0;

  float ra[4], rb[4], rat[4], rbt[4];
  auto load_op = [&](const float* P, const float* Pt, bool hasT, bool vec, bool kfast, int r0,
                     int rmax, int64_t sr, int64_t sk, int k0, float* r, float* rt) {
    if (vec) {
      // one float4 per thread along the contiguous dimension
      int row = kfast ? (tid >> 2) : ((tid & 15) * 4);
      int kk = kfast ? ((tid & 3) * 4) : (tid >> 4);
      bool ok = (r0 + row) < rmax && (k0 + kk) < kend;
      int64_t off = (int64_t)(r0 + row) * sr + (int64_t)(k0 + kk) * sk;
      float4 v = ok ? *(const float4*)(P + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
      if (MODE) {
        float4 t = (ok && hasT) ? *(const float4*)(Pt + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        rt[0] = t.x; rt[1] = t.y; rt[2] = t.z; rt[3] = t.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int e = tid + 256 * i;
        int kk = kfast ? (e & (BK - 1)) : (e >> 6);
        int row = kfast ? (e >> 4) : (e & 63);
        bool ok = (r0 + row) < rmax && (k0 + kk) < kend;
        int64_t off = (int64_t)(r0 + row) * sr + (int64_t)(k0 + kk) * sk;
        r[i] = ok ? P[off] : 0.f;
        if (MODE) rt[i] = (ok && hasT) ? Pt[off] : 0.f;
      }
    }
  };
  auto store_op = [&](float (*L)[LDP], float (*Lt)[LDP], bool vec, bool kfast, const float* r,
                      const float* rt) {
    if (vec) {
      if (kfast) {
        int row = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          L[kk + i][row] = r[i];
          if (MODE) Lt[kk + i][row] = rt[i];
        }
      } else {
        int row = (tid & 15) * 4, kk = tid >> 4;
        *(float4*)&L[kk][row] = make_float4(r[0], r[1], r[2], r[3]);
        if (MODE) *(float4*)&Lt[kk][row] = make_float4(rt[0], rt[1], rt[2], rt[3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int e = tid + 256 * i;
        int kk = kfast ? (e & (BK - 1)) : (e >> 6);
        int row = kfast ? (e >> 4) : (e & 63);
        L[kk][row] = r[i];
        if (MODE) Lt[kk][row] = rt[i];
      }
    }
  };
  auto load_tile = [&](int k0) {
    load_op(p.A, p.A_t, hasAt, p.vecA, a_kfast, m0, p.M, p.sAm, p.sAk, k0, ra, rat);
    load_op(p.B, p.B_t, hasBt, p.vecB, b_kfast, n0, p.N, p.sBn, p.sBk, k0, rb, rbt);
  };
  auto store_tile = [&](int buf) {
    store_op(As[buf], Ats[MODE ? buf : 0], p.vecA, a_kfast, ra, rat);
    store_op(Bs[buf], Bts[MODE ? buf : 0], p.vecB, b_kfast, rb, rbt);
  };

  int buf = 0;
  if (kbeg < kend) { load_tile(kbeg); store_tile(0); }
  __syncthreads();
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) load_tile(k0 + BK);          // in flight while this slab is multiplied
#pragma unroll 4
    for (int kk = 0; kk < BK; ++kk) {
      float4 a = *(const float4*)&As[buf][kk][tm];
      float4 b = *(const float4*)&Bs[buf][kk][tn];
      float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        if (do_rs) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rs[i] += av[i];
        }
      } else {
        float4 at = *(const float4*)&Ats[buf][kk][tm];
        float4 bt = *(const float4*)&Bts[buf][kk][tn];
        float atv[4] = {at.x, at.y, at.z, at.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = fmaf(av[i], btv[j], fmaf(atv[i], bv[j], acc[i][j]));
        if (do_rs) {
#pragma unroll
          for (int i = 0; i < 4; ++i) rs[i] += atv[i];
        }
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- split-K: fp32 atomics into this tile's (always-zero-between-launches) scratch accumulator;
  // the last block to arrive swaps the sums out (re-zeroing the scratch) and runs the epilogue.
  // No device-scope fence: on gfx950 that is a whole-L2 write-back + invalidate per block.  The
  // accumulators are only ever touched by device-scope atomics (executed at the memory side), and EVERY
  // wave drains its own atomics with an explicit s_waitcnt vmcnt(0) BEFORE the workgroup barrier that
  // precedes thread 0's arrival-counter bump: s_barrier itself does not wait for outstanding VMEM, and a
  // workgroup-scope fence lowers to nothing for global memory, so without the wait another block could
  // see splits-1 arrivals and swap the sums out while these adds are still in flight.
  if (p.splits > 1) {
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* mine = p.part + (size_t)tile * (BM * BN) + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(mine + (i * 4 + j) * 256, acc[i][j]);
    float* rsp = p.part_rs + (size_t)blockIdx.y * BM + tm;
    if (do_rs && (tid & 15) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) atomicAdd(rsp + i, rs[i]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's atomic adds have been performed
    __syncthreads();
    if (tid == 0) {
      unsigned prev = atomicAdd(&p.ctr[tile], 1u);
      s_last = prev == (unsigned)(p.splits - 1);
      if (s_last) atomicExch(&p.ctr[tile], 0u);   // ready for the next launch on this stream
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = atomicExch(mine + (i * 4 + j) * 256, 0.f);
    if (do_rs && (tid & 15) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[i] = atomicExch(rsp + i, 0.f);
    }
  }

  // ---- epilogue
  if (do_rs && (tid & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (m0 + tm + i < p.M) p.rowsum[m0 + tm + i] = rs[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + tm + i;
    if (m >= p.M) continue;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = n0 + tn + j;
      float v = acc[i][j];
      if (n < p.N) {
        if (p.epi == EP_BIAS_ACT) {
          v += p.bias ? p.bias[n] : 0.f;
          if (p.act == 1) v = relu_(v);
          else if (p.act == 2) v = sigmoid_(v);
        } else if (p.epi == EP_BIAS_ACT_T) {
          // aux holds the stashed POST-activation primal; recover what the tangent rule needs from it
          v += p.bias ? p.bias[n] : 0.f;
          float yv = p.aux ? p.aux[(int64_t)m * p.N + n] : 0.f;
          float d = p.act == 1 ? (yv > 0.f ? 1.f : 0.f) : (p.act == 2 ? yv * (1.f - yv) : 1.f);
          v *= d;
        } else if (p.epi == EP_RELU_MASK) {
          v = p.aux[(int64_t)m * p.N + n] > 0.f ? v : 0.f;
        }
      }
      o[j] = v;
    }
    float* dst = p.C + (int64_t)m * p.N + n0 + tn;
    if (p.vecC && n0 + tn + 3 < p.N) {
      *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + tn + j < p.N) dst[j] = o[j];
    }
  }
}


// ====================================================================== small-batch linears on the matrix cores
// One of the three GEMM dimensions is the number of synthetic pairs (10..500): the squeeze-excite MLPs, the
// pooled-feature -> conv3 shortcut of the SE path and the text head.  These layers are latency, not
// throughput: a [100 x 1536] x [1536 x 384] product is 0.12 GFLOP but sits in the dependent chain of every
// residual block of every pass.  k_sgemm_small above split K over blocks and combined the partial tiles with
// fp32 atomics + an arrival counter + an exchange: three dependent memory round trips per layer (20-30 us
// beside the weight-gradient stream).  Here ONE workgroup owns a 16 x 16 output tile and splits K over its
// NW waves (exact-fp32 v_mfma_f32_16x16x4_f32, operands straight from global memory into registers: every
// lane loads 4 consecutive k of its row / column, the same k-permutation on both operands); the partial
// accumulators meet in LDS, the epilogue (bias, activation or its tangent rule, ReLU mask, bias-gradient row
// sums, accumulate-into) runs on the reduced tile, one store.  One memory round trip, no atomics, no scratch,
// deterministic.  The B operand may be stored in bf16 (the packed standardised conv3 weights in bf16 mode).
struct LArgs {
  const float *A, *A_t;
  const void *B, *B_t;
  float* C;
  int M, N, K;
  int64_t sAm, sAk, sBn, sBk;   // element strides
  int kw_len;                   // K range per wave (multiple of 8)
  int vecA, vecB;               // 16-byte (8-byte for bf16) loads along k are legal
  int epi, act, accum;
  float alpha;
  const float* bias;
  const float* aux;
  float* rowsum;
};

template <class TB> DEVI void ld4(const TB* p, float* o);
template <> DEVI void ld4<float>(const float* p, float* o) {
  const float4 v = *(const float4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> DEVI void ld4<bf16>(const bf16* p, float* o) {
  const uint2 v = *(const uint2*)p;
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}

// Register diet on purpose: these layers run beside the weight-gradient contractions (2 blocks of 188 registers per
// CU leave 136 per SIMD lane) -- a wave that needs more waits for a whole contraction block to retire (50-90 us).
template <int NW, int MODE, class TB, bool AKC, bool BKC>
__global__ __launch_bounds__(NW * 64, NW >= 4 ? 4 : (NW == 2 ? 4 : 4)) void k_lin_mfma(const LArgs p) {
  // 16 x 16 output tile per workgroup, v_mfma_f32_16x16x4_f32: lane = (r = lane & 15, kq = lane >> 4) supplies
  // A[row r][k] / B[col r][k] for k = 16 s + 4 kq + j (one 16-byte load per operand and step s, four MFMAs j = 0..3
  // -- the same k-permutation on both operands); the four lanes of a row read 64 contiguous bytes per load.
  __shared__ float red[NW][4][64];
  __shared__ float rsum[NW][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  const int row = min(m0 + l15, p.M - 1), col = min(n0 + l15, p.N - 1);   // clamped: never stored
  const int k0 = wave * p.kw_len, k1 = min(p.K, k0 + p.kw_len);
  const bool hasAt = MODE && p.A_t != nullptr, hasBt = MODE && p.B_t != nullptr;
  const bool do_rs = p.rowsum != nullptr && blockIdx.x == 0;
  const float* Ar = p.A + (AKC ? (int64_t)row * p.sAm : (int64_t)row);
  const float* Atr = hasAt ? p.A_t + (AKC ? (int64_t)row * p.sAm : (int64_t)row) : nullptr;
  const TB* Br = (const TB*)p.B + (BKC ? (int64_t)col * p.sBn : (int64_t)col);
  const TB* Btr = hasBt ? (const TB*)p.B_t + (BKC ? (int64_t)col * p.sBn : (int64_t)col) : nullptr;

  constexpr int U = 2;                 // steps (of 16 k) per register chunk (two chunks in flight)
  struct Frag { float a[U][4], b[U][4], at[MODE ? U : 1][4], bt[MODE ? U : 1][4]; };
  auto ldA = [&](const float* P, int kk, float* o) __attribute__((always_inline)) {
    if constexpr (AKC) {
      if (kk < k1) {
        if (p.vecA) ld4<float>(P + kk, o);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (kk + i < k1) ? P[kk + i] : 0.f;
        }
      } else { o[0] = o[1] = o[2] = o[3] = 0.f; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (kk + i < k1) ? P[(int64_t)(kk + i) * p.sAk] : 0.f;
    }
  };
  auto ldB = [&](const TB* P, int kk, float* o) __attribute__((always_inline)) {
    if constexpr (BKC) {
      if (kk < k1) {
        if (p.vecB) ld4<TB>(P + kk, o);
        else {
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (kk + i < k1) ? to_f(P[kk + i]) : 0.f;
        }
      } else { o[0] = o[1] = o[2] = o[3] = 0.f; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (kk + i < k1) ? to_f(P[(int64_t)(kk + i) * p.sBk]) : 0.f;
    }
  };
  auto load = [&](Frag& f, int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = kc + 16 * u + 4 * kq;
      ldA(Ar, kk, f.a[u]);
      ldB(Br, kk, f.b[u]);
      if constexpr (MODE != 0) {
        if (hasAt) ldA(Atr, kk, f.at[u]); else { f.at[u][0] = f.at[u][1] = f.at[u][2] = f.at[u][3] = 0.f; }
        if (hasBt) ldB(Btr, kk, f.bt[u]); else { f.bt[u][0] = f.bt[u][1] = f.bt[u][2] = f.bt[u][3] = 0.f; }
      }
    }
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float rs = 0.f;
  Frag cur, nxt;
  if (k0 < k1) load(cur, k0);
  for (int kc = k0; kc < k1; kc += 16 * U) {
    const bool more = kc + 16 * U < k1;
    if (more) load(nxt, kc + 16 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (MODE == 0) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[u][i], cur.b[u][i], acc, 0, 0, 0);
          if (do_rs) rs += cur.a[u][i];
        } else {
          if (hasAt) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.at[u][i], cur.b[u][i], acc, 0, 0, 0);
          if (hasBt) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[u][i], cur.bt[u][i], acc, 0, 0, 0);
          if (do_rs) rs += cur.at[u][i];
        }
      }
    }
    if (more) cur = nxt;
  }
  // ---- the NW partial tiles meet in LDS (C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + r)
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
  if (do_rs) {
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    if (kq == 0) rsum[wave][l15] = rs;
  }
  __syncthreads();
  if (do_rs && tid < 16 && m0 + tid < p.M) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += rsum[w][tid];
    p.rowsum[m0 + tid] = p.accum ? p.rowsum[m0 + tid] + s : s;
  }
  const int n = n0 + l15;
  // 256 outputs: thread (wave w, lane) takes accumulator register r = w (+ NW, ...) of every wave's tile
  for (int r = wave; r < 4; r += NW) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][r][lane];
    const int m = m0 + 4 * kq + r;
    if (m >= p.M || n >= p.N) continue;
    v *= p.alpha;
    const int64_t ci = (int64_t)m * p.N + n;
    if (p.epi == EP_BIAS_ACT) {
      v += p.bias ? p.bias[n] : 0.f;
      if (p.act == 1) v = relu_(v);
      else if (p.act == 2) v = sigmoid_(v);
    } else if (p.epi == EP_BIAS_ACT_T) {
      v += p.bias ? p.bias[n] : 0.f;
      const float yv = p.aux ? p.aux[ci] : 0.f;
      v *= p.act == 1 ? (yv > 0.f ? 1.f : 0.f) : (p.act == 2 ? yv * (1.f - yv) : 1.f);
    } else if (p.epi == EP_RELU_MASK) {
      v = p.aux[ci] > 0.f ? v : 0.f;
    }
    if (p.accum) v += p.C[ci];
    p.C[ci] = v;
  }
}

template <int NW, int MODE, class TB>
void lin_mfma_launch(const LArgs& a, dim3 grid, hipStream_t st) {
  const bool akc = a.sAk == 1, bkc = a.sBk == 1;
  if (akc && bkc) k_lin_mfma<NW, MODE, TB, true, true><<<grid, NW * 64, 0, st>>>(a);
  else if (akc) k_lin_mfma<NW, MODE, TB, true, false><<<grid, NW * 64, 0, st>>>(a);
  else if (bkc) k_lin_mfma<NW, MODE, TB, false, true><<<grid, NW * 64, 0, st>>>(a);
  else k_lin_mfma<NW, MODE, TB, false, false><<<grid, NW * 64, 0, st>>>(a);
}

// C[M,N] (+)= alpha * sum_k A[m,k] B[n,k]  (tangent: A_t B + A B_t), epilogue as in run_gemm
template <class TB>
void run_lin_mfma(bool tangent, const float* A, const float* A_t, const TB* B, const TB* B_t, float* C, int M,
                  int N, int K, int64_t sAm, int64_t sAk, int64_t sBk, int64_t sBn, int epi, int act,
                  const float* bias, const float* aux, float* rowsum, int accum, float alpha, hipStream_t st) {
  LArgs a;
  a.A = A; a.A_t = A_t; a.B = B; a.B_t = B_t; a.C = C; a.M = M; a.N = N; a.K = K;
  a.sAm = sAm; a.sAk = sAk; a.sBn = sBn; a.sBk = sBk;
  a.epi = epi; a.act = act; a.accum = accum; a.alpha = alpha; a.bias = bias; a.aux = aux; a.rowsum = rowsum;
  auto al = [](const void* q, size_t b) { return q == nullptr || ((uintptr_t)q & (b - 1)) == 0; };
  a.vecA = sAk == 1 && al(A, 16) && al(A_t, 16) && sAm % 4 == 0 && K % 4 == 0;
  a.vecB = sBk == 1 && al(B, 4 * sizeof(TB)) && al(B_t, 4 * sizeof(TB)) && sBn % 4 == 0 && K % 4 == 0;
  const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
  // waves per tile: fill the 1024 SIMDs about twice, at least 64 k per wave, at most 8 waves
  int nw = 1;
  while (nw < 8 && tiles * nw < 2048 && K / (nw * 2) >= 64) nw *= 2;
  a.kw_len = ((K + nw - 1) / nw + 15) / 16 * 16;
  dim3 grid((N + 15) / 16, (M + 15) / 16, 1);
#define LIN_GO(NWV)                                                      \
  do {                                                                   \
    if (tangent) lin_mfma_launch<NWV, 1, TB>(a, grid, st);               \
    else lin_mfma_launch<NWV, 0, TB>(a, grid, st);                       \
  } while (0)
  switch (nw) {
    case 1: LIN_GO(1); break;
    case 2: LIN_GO(2); break;
    case 4: LIN_GO(4); break;
    default: LIN_GO(8); break;
  }
#undef LIN_GO
}

inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; }

void run_gemm(bool tangent, const float* A, const float* A_t, const float* B, const float* B_t,
              float* C, int M, int N, int K, int64_t sAm, int64_t sAk, int64_t sBk, int64_t sBn,
              int epi, int act, const float* bias, const float* aux, float* rowsum,
              const LinScratch& ws, hipStream_t st) {
  if (ws.mfma) {   // small-batch layer: one workgroup per 32x32 tile on the matrix cores, no split-K scratch
    run_lin_mfma<float>(tangent, A, A_t, B, B_t, C, M, N, K, sAm, sAk, sBk, sBn, epi, act, bias, aux, rowsum, 0,
                        1.f, st);
    return;
  }
  GArgs g;
  g.A = A; g.A_t = A_t; g.B = B; g.B_t = B_t; g.C = C; g.M = M; g.N = N; g.K = K;
  g.sAm = sAm; g.sAk = sAk; g.sBk = sBk; g.sBn = sBn;
  g.epi = epi; g.act = act; g.bias = bias; g.aux = aux; g.rowsum = rowsum;
  g.part = ws.part; g.part_rs = ws.part_rs; g.ctr = ws.ctr;
  // 16-byte loads run along the unit-stride dimension and must not straddle the operand's edge
  g.vecA = al16(A) && al16(A_t) && (sAk == 1 ? (sAm % 4 == 0 && K % 4 == 0) : (sAm == 1 && sAk % 4 == 0 && M % 4 == 0));
  g.vecB = al16(B) && al16(B_t) && (sBk == 1 ? (sBn % 4 == 0 && K % 4 == 0) : (sBn == 1 && sBk % 4 == 0 && N % 4 == 0));
  g.vecC = al16(C) && N % 4 == 0;
  int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
  int tiles = mt * nt;
  int splits = 1;
  if (tiles < 256 && ws.part) {
    splits = (512 + tiles - 1) / tiles;
    int maxs = (K + 4 * BK - 1) / (4 * BK);   // at least 4 K-steps per block
    if (splits > maxs) splits = maxs;
    if (tiles > ws.ctr_n || (int64_t)tiles * BM * BN > ws.part_floats ||
        (int64_t)mt * BM > ws.part_rs_floats) splits = 1;
    if (splits < 1) splits = 1;
  }
  int len = (K + splits - 1) / splits;
  len = (len + BK - 1) / BK * BK;
  if (len < BK) len = BK;
  splits = std::max(1, (K + len - 1) / len);
  g.ksplit_len = len;
  g.splits = splits;
  dim3 grid(nt, mt, splits);
  if (tangent) k_sgemm_small<1><<<grid, 256, 0, st>>>(g);
  else k_sgemm_small<0><<<grid, 256, 0, st>>>(g);
}

}  // namespace

int64_t lin_scratch_bytes() {
  return (LIN_PART_FLOATS + LIN_PART_RS_FLOATS) * 4 + LIN_CTR_N * 4;
}
LinScratch lin_scratch_carve(void* base) {
  LinScratch s;
  s.part = (float*)base; s.part_floats = LIN_PART_FLOATS;
  s.part_rs = s.part + LIN_PART_FLOATS; s.part_rs_floats = LIN_PART_RS_FLOATS;
  s.ctr = (unsigned*)(s.part_rs + LIN_PART_RS_FLOATS); s.ctr_n = LIN_CTR_N;
  s.mfma = true;
  return s;
}

// y[n,j] = act(sum_k x[n,k] W[j,k] + b[j])
void launch_linear_fwd(float* y, float* y_t, const float* x, const float* x_t, const float* W,
                       const float* W_t, const float* b, const float* b_t, int n, int k, int j,
                       int act, const LinScratch& ws, hipStream_t st) {
  if (y_t) run_gemm(true, x, x_t, W, W_t, y_t, n, j, k, k, 1, 1, k, EP_BIAS_ACT_T, act, b_t, y, nullptr, ws, st);
  else run_gemm(false, x, nullptr, W, nullptr, y, n, j, k, k, 1, 1, k, EP_BIAS_ACT, act, b, nullptr, nullptr, ws, st);
}
// dx[n,k] = sum_j dy[n,j] W[j,k]   (* [relu_of > 0] when given: the ReLU backward of the SE MLP)
void launch_linear_dgrad(float* dx, float* dx_t, const float* dy, const float* dy_t,
                         const float* W, const float* W_t, const float* relu_of, int n, int k,
                         int j, const LinScratch& ws, hipStream_t st) {
  int epi = relu_of ? EP_RELU_MASK : EP_PLAIN;
  if (dx_t) run_gemm(true, dy, dy_t, W, W_t, dx_t, n, k, j, j, 1, k, 1, epi, 0, nullptr, relu_of, nullptr, ws, st);
  else run_gemm(false, dy, nullptr, W, nullptr, dx, n, k, j, j, 1, k, 1, epi, 0, nullptr, relu_of, nullptr, ws, st);
}
// dW[j,k] = sum_n dy[n,j] x[n,k] ; db[j] = sum_n dy[n,j]   (tangent pass writes tangents only)
void launch_linear_wgrad(float* dW, float* db, const float* dy, const float* dy_t, const float* x,
                         const float* x_t, int n, int k, int j, const LinScratch& ws,
                         hipStream_t st) {
  bool tangent = dy_t || x_t;
  if (tangent) run_gemm(true, dy, dy_t, x, x_t, dW, j, k, n, 1, j, k, 1, EP_PLAIN, 0, nullptr, nullptr, db, ws, st);
  else run_gemm(false, dy, nullptr, x, nullptr, dW, j, k, n, 1, j, k, 1, EP_PLAIN, 0, nullptr, nullptr, db, ws, st);
}

// ---- typed-weight forms (W in the activation storage type: the packed standardised conv weights)
// y[n,j] = sum_k x[n,k] W[j,k] + b[j]   (tangent: x_t W + x W_t + b_t, written to y_t)
template <class TW>
void launch_linear_fwd_w(float* y, float* y_t, const float* x, const float* x_t, const TW* W, const TW* W_t,
                         const float* b, const float* b_t, int n, int k, int j, float alpha, hipStream_t st) {
  if (y_t) run_lin_mfma<TW>(true, x, x_t, W, W_t, y_t, n, j, k, k, 1, 1, k, EP_BIAS_ACT_T, 0, b_t, nullptr, nullptr,
                            0, alpha, st);
  else run_lin_mfma<TW>(false, x, nullptr, W, nullptr, y, n, j, k, k, 1, 1, k, EP_BIAS_ACT, 0, b, nullptr, nullptr, 0,
                        alpha, st);
}
template void launch_linear_fwd_w<float>(float*, float*, const float*, const float*, const float*, const float*,
                                         const float*, const float*, int, int, int, float, hipStream_t);
template void launch_linear_fwd_w<bf16>(float*, float*, const float*, const float*, const bf16*, const bf16*,
                                        const float*, const float*, int, int, int, float, hipStream_t);
// dW[j,k] += sum_n dy[n,j] x[n,k] ; db[j] += sum_n dy[n,j]   (tangent: dy_t x + dy x_t ; db += sum dy_t)
void launch_linear_wgrad_accum(float* dW, float* db, const float* dy, const float* dy_t, const float* x,
                               const float* x_t, int n, int k, int j, hipStream_t st) {
  const bool tangent = dy_t || x_t;
  run_lin_mfma<float>(tangent, dy, dy_t, x, x_t, dW, j, k, n, 1, j, k, 1, EP_PLAIN, 0, nullptr, nullptr, db, 1, 1.f,
                      st);
}

// out[j] = sum_c W[j,c] v[c] + b[j]   (tangent: out_t[j] = sum_c (W_t[j,c] v[c] + W[j,c] v_t[c]) + b_t[j]); a wave per row
__global__ __launch_bounds__(256) void k_matvec_bias(float* __restrict__ out, float* __restrict__ out_t,
                                                     const float* __restrict__ W, const float* __restrict__ W_t,
                                                     const float* __restrict__ v, const float* __restrict__ v_t,
                                                     const float* __restrict__ b, const float* __restrict__ b_t, int rows,
                                                     int cols) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float a = 0.f, at = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float w = W[(size_t)row * cols + c], x = v[c];
    a += w * x;
    if (out_t) at += (W_t ? W_t[(size_t)row * cols + c] : 0.f) * x + w * (v_t ? v_t[c] : 0.f);
  }
  a = wave_sum(a);
  if (out_t) at = wave_sum(at);
  if (lane == 0) {
    if (out_t) out_t[row] = at + (b_t ? b_t[row] : 0.f);
    else out[row] = a + (b ? b[row] : 0.f);
  }
}
void launch_matvec_bias(float* out, float* out_t, const float* W, const float* W_t, const float* v, const float* v_t,
                        const float* b, const float* b_t, int rows, int cols, hipStream_t st) {
  k_matvec_bias<<<(rows + 3) / 4, 256, 0, st>>>(out, out_t, W, W_t, v, v_t, b, b_t, rows, cols);
}
