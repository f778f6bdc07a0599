// Fused multi-head attention for the ViT image encoder (BASELINE configs[4]: "pure-MFMA attention path"), all four
// passes of the unrolled loop: forward, inner gradient and the tangents of both (reference networks.py:661,668 reach
// timm / clip attention; distill.py:562-567 and :606 differentiate it once and twice).
//
// The unfused path (vit.hip: k_bgemm_mfma + k_softmax_*) writes the [tokens x tokens] scores, probabilities and their
// gradients to HBM -- 95 MB per tensor and layer at configs[4], touched ~40 times per layer and inner step.  With 197
// tokens and head dimension 64 a head's K, V (and tangents) fit in LDS, so nothing of that size has to leave the CU:
// every pass RECOMPUTES the score tiles it needs on the matrix cores from the fused qkv tensor and keeps only per-row
// statistics (max m, sum l, and the row sums r, D, D_t below) in HBM.
//
//   S = Q K^T,  P = softmax(s S) = exp(s (S - m)) / l,  O = P V
//   tangent:  S_t = Q_t K^T + Q K_t^T,  r = rowsum(P S_t),  P_t = s P (S_t - r),  O_t = P_t V + P V_t
//   backward: dP = dO V^T,  D = rowsum(P dP),  dS = s P (dP - D),  dV = P^T dO,  dQ = dS K,  dK = dS^T Q
//   tangent of backward:  dP_t = dO_t V^T + dO V_t^T,  D_t = rowsum(P_t dP + P dP_t),
//             dS_t = s (P_t (dP - D) + P (dP_t - D_t)),  dV_t = P_t^T dO + P^T dO_t,
//             dQ_t = dS_t K + dS K_t,  dK_t = dS_t^T Q + dS^T Q_t
//
// ONE kernel template serves every product.  A workgroup (8 waves) handles one (image, head); wave w OWNS 32 rows of
// the "owner" index -- queries for O / O_t / dQ / dQ_t, keys for dV / dK and their tangents -- and walks the other
// index in tiles of 32.  Score-type tiles are v_mfma_f32_32x32x16_bf16(other-side rows from LDS, owner-side rows from
// registers): the accumulator holds tile[other (registers), owner (lane)], which IS the A-operand layout of the second
// product  out[owner, :] = sum_other W[owner, other] R[other, :]  (rows = owner = lane, eight `other` per lane), so the
// probabilities / score gradients go from accumulator registers to the matrix cores without touching LDS; R (V, K, dO
// or Q) is staged transposed in LDS so that a lane's eight `other` values of one column are two 8-byte reads.
#include <atomic>
#include <cstdint>

#include "kernels.h"
#include "mdd_hip.h"

#ifndef MDD_ATTN_DBG
#define MDD_ATTN_DBG 0     // experiment builds: bit 0 skip pass A, 1 skip pass B, 2 skip stores, 3 skip the LDS fill stores
#endif

namespace {

constexpr int HD = 64;                 // head dimension (vit_tiny16 / vit_b16)
constexpr int TP = 224;                // tokens padded to whole 32-row tiles (197 -> 7 tiles)
constexpr int NT = TP / 32;
constexpr int RTP = 228;               // pitch (elements) of the transposed tensors: conflict-free 8-byte reads
constexpr int ROWB = TP * 128;         // bytes of a row-major [TP x 64] bf16 tensor in LDS
constexpr int TRB = HD * RTP * 2;      // bytes of a transposed one

struct AttnArgs {
  const bf16 *qkv, *qkv_t;             // [n, tok, 3, heads, 64]
  const bf16 *dout, *dout_t;           // [n, tok, heads, 64]   gradient of the attention output
  bf16 *out;                           // O / O_t [n, tok, heads, 64]  or  d qkv / d qkv_t [n, tok, 3, heads, 64]
  const bf16 *o, *o_t;                 // attention output and its tangent (modes 2, 3: D = rowsum(O dO), D_t)
  float *m, *l, *r, *D, *D_t;          // row statistics [n, heads, tok]
  int n, tok, heads;
  int r_tan;                           // modes 2 / 4: the R operand is the TANGENT tensor (k_t / q_t) ...
  int accum;                           // ... and the result is added to what `out` holds
  int skip0;                           // mode 4: only out1 (dK-type product)
  float scale;
};

DEVI int lds_row_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
DEVI int rowmap(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

// MODE: 0 F (O), 1 TF (O_t), 2 BQ (dQ | part 2 of dQ_t), 3 TBQ (dS_t K), 4 BKV (dV, dK | part 2 of dK_t),
//       5 TBV (dV_t), 6 TBK (dS_t^T Q)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_attn(const AttnArgs p) {
  constexpr bool OQ = MODE <= 3;                       // the owner index is the query index
  constexpr bool TAN = MODE == 1 || MODE == 3 || MODE == 5 || MODE == 6;
  constexpr bool DP = MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6;
  constexpr bool PASS_A = MODE <= 3;                   // a row reduction over the other index comes first
  constexpr int NY = (TAN ? 2 : 1) * (DP ? 2 : 1);     // row-major other-side tensors
  constexpr int UNR = (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 5) ? 2 : 1;   // tiles in flight (register headroom)
  constexpr int NR = (MODE == 1 || MODE == 5) ? 2 : (MODE == 4 ? 2 : 1);   // transposed tensors
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Y1 = smem;
  char* const Y1t = TAN ? smem + ROWB : smem;
  char* const Y2 = smem + (TAN ? 2 : 1) * ROWB;
  char* const Y2t = smem + 3 * ROWB;
  char* const R0 = smem + NY * ROWB;
  char* const R1 = R0 + TRB;                           // second transposed tensor (R0_t of modes 1 / 5, q of mode 4)
  float4* const st4 = (float4*)(R0 + NR * TRB);        // [256] {-m c1, 1/l, r, D} of the QUERY rows
  float* const stDt = (float*)(st4 + 256);             // [256] D_t

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int head = blockIdx.x, img = blockIdx.y, T = p.tok, H = p.heads, Dm = H * HD;
  const size_t qrow = (size_t)3 * Dm;                  // row stride of the fused qkv tensor
  const bf16* qkv = p.qkv + (size_t)img * T * qrow + (size_t)head * HD;
  const bf16* qkv_t = (TAN || p.r_tan) ? p.qkv_t + (size_t)img * T * qrow + (size_t)head * HD : nullptr;
  const bf16* dob = DP || MODE == 5 ? p.dout + (size_t)img * T * Dm + (size_t)head * HD : nullptr;
  const bf16* dob_t = (MODE == 3 || MODE == 5 || MODE == 6) ? p.dout_t + (size_t)img * T * Dm + (size_t)head * HD : nullptr;
  const size_t sbase = ((size_t)img * H + head) * T;   // statistics of this (image, head)

  // ---- which global tensor plays which role (row stride, component offset inside the qkv row)
  // owner-side: U1 (scores), U2 (dP-type); other-side row-major: Y1, Y2; transposed: R0 (+ R0_t), R1
  const int cq = 0, ck = Dm, cv = 2 * Dm;
  const bf16 *gU1, *gU1t = nullptr, *gU2 = nullptr, *gU2t = nullptr, *gY1, *gY1t = nullptr, *gY2 = nullptr, *gY2t = nullptr;
  size_t sU1 = qrow, sU2 = qrow, sY1 = qrow, sY2 = qrow;
  if constexpr (OQ) {
    gU1 = qkv + cq; gY1 = qkv + ck;
    if constexpr (TAN) { gU1t = qkv_t + cq; gY1t = qkv_t + ck; }
    if constexpr (DP) { gU2 = dob; sU2 = Dm; gY2 = qkv + cv; if constexpr (TAN) { gU2t = dob_t; gY2t = qkv_t + cv; } }
  } else {
    gU1 = qkv + ck; gY1 = qkv + cq;
    if constexpr (TAN) { gU1t = qkv_t + ck; gY1t = qkv_t + cq; }
    if constexpr (DP) { gU2 = qkv + cv; gY2 = dob; sY2 = Dm; if constexpr (TAN) { gU2t = qkv_t + cv; gY2t = dob_t; } }
  }
  const bf16 *gR0, *gR1 = nullptr; size_t sR0 = qrow, sR1 = qrow;
  if constexpr (MODE == 0) gR0 = qkv + cv;
  else if constexpr (MODE == 1) { gR0 = qkv + cv; gR1 = qkv_t + cv; }
  else if constexpr (MODE == 2) gR0 = (p.r_tan ? qkv_t : qkv) + ck;
  else if constexpr (MODE == 3) gR0 = qkv + ck;
  else if constexpr (MODE == 4) { gR0 = dob; sR0 = Dm; gR1 = (p.r_tan ? qkv_t : qkv) + cq; }
  else if constexpr (MODE == 5) { gR0 = dob; sR0 = Dm; gR1 = dob_t; sR1 = Dm; }
  else gR0 = qkv + cq;

  // ---- LDS fill: row-major tensors (16-byte chunks, swizzled), transposed tensors (2-byte scatter), statistics.
  // ALL global loads of the block are issued before the first LDS store (a load -> store loop per tensor made every
  // block pay 4 dependent HBM round trips per tensor: 20-30 us of a 30-55 us block).
  constexpr int NCH = (TP * 8 + 511) / 512;            // 16-byte chunks per thread and tensor
  constexpr int NTEN = NY + NR;
  const bf16* gsrc[6] = {gY1, TAN ? gY1t : nullptr, DP ? gY2 : nullptr, (TAN && DP) ? gY2t : nullptr, gR0, NR == 2 ? gR1 : nullptr};
  const size_t gstr[6] = {sY1, sY1, sY2, sY2, sR0, sR1};
  char* const gdst[6] = {Y1, Y1t, Y2, Y2t, R0, R1};
  constexpr bool gtr[6] = {false, false, false, false, true, true};
  const bool guse[6] = {true, TAN, DP, TAN && DP, true, NR == 2};
  // row-major tensors: chunk c = tid + 512 j -> (row c >> 3, chunk c & 7); transposed tensors: item c -> (row PAIR
  // c >> 3, chunk c & 7), both rows' chunks loaded so that a 32-bit LDS store holds [row 2 j | row 2 j + 1] of a column
  constexpr int NCHT = (TP / 2 * 8 + 511) / 512;       // (row pair, chunk) items per thread and transposed tensor
  u32x4 stg[NY][NCH];
  u32x4 stt[NR][NCHT][2];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!guse[i]) continue;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = tid + 512 * j, row = c >> 3, ch = c & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (c < TP * 8 && row < T) v = *(const u32x4*)(gsrc[i] + (size_t)row * gstr[i] + ch * 8);
        stg[k][j] = v;
      }
      ++k;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
      for (int j = 0; j < NCHT; ++j) {
        const int c = tid + 512 * j, rp = c >> 3, ch = c & 7;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = 2 * rp + u;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (c < TP / 2 * 8 && row < T) v = *(const u32x4*)(gsrc[4 + i] + (size_t)row * gstr[4 + i] + ch * 8);
          stt[i][j][u] = v;
        }
      }
    }
  }
  // statistics of the QUERY rows in LDS: st4[q] = {-m c1, 1/l, r, D}, stDt[q]; rows beyond the tokens get 1/l = 0, so a
  // probability formed from them is 0 without a per-element mask
  const float c1 = p.scale * 1.4426950408889634f;      // exp(s x) = exp2(c1 x)
  if (tid < 256) {
    const bool ok = tid < T;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MODE != 0) { if (ok) { v.x = -p.m[sbase + tid] * c1; v.y = 1.f / p.l[sbase + tid]; } }
    if constexpr (MODE == 3 || MODE == 5 || MODE == 6) { if (ok) v.z = p.r[sbase + tid]; }
    if constexpr (MODE == 4 || MODE == 6) { if (ok) v.w = p.D[sbase + tid]; }
    st4[tid] = v;
    if constexpr (MODE == 6) stDt[tid] = ok ? p.D_t[sbase + tid] : 0.f;
  }
  // ---- owner-side fragments: lane (row 32 w + l31, half lh) holds elements 16 s + 8 lh .. + 8 of its row, s = 0..3
  const int own = wave * 32 + l31;
  const bool own_ok = own < T;
  auto load_frag = [&](const bf16* src, size_t stride, u32x4* f) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (own_ok) v = *(const u32x4*)(src + (size_t)own * stride + 16 * s + 8 * lh);
      f[s] = v;
    }
  };
  u32x4 fU1[4], fU1t[TAN ? 4 : 1], fU2[DP ? 4 : 1], fU2t[(TAN && DP) ? 4 : 1];
  load_frag(gU1, sU1, fU1);
  if constexpr (TAN) load_frag(gU1t, sU1, fU1t);
  if constexpr (DP) { load_frag(gU2, sU2, fU2); if constexpr (TAN) load_frag(gU2t, sU2, fU2t); }
  if (!(MDD_ATTN_DBG & 8)) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!guse[i]) continue;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = tid + 512 * j, row = c >> 3, ch = c & 7;
        if (c < TP * 8) *(u32x4*)(gdst[i] + lds_row_off(row, ch)) = stg[k][j];
      }
      ++k;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
      for (int j = 0; j < NCHT; ++j) {
        const int c = tid + 512 * j, rp = c >> 3, ch = c & 7;
        if (c < TP / 2 * 8) {
          unsigned* d = (unsigned*)gdst[4 + i];
          const u32x4 a = stt[i][j][0], b = stt[i][j][1];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const unsigned lo = (a[e >> 1] >> ((e & 1) * 16)) & 0xffffu, hi = (b[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
            d[((ch * 8 + e) * RTP + 2 * rp) >> 1] = lo | (hi << 16);
          }
        }
      }
    }
  }
  __syncthreads();

  const float sc = p.scale;
  const int nt = (T + 31) >> 5;                        // tiles that hold tokens (the others contribute nothing)
  auto mm = [&](const u32x4& a, const u32x4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  };
  // score-type tile t: acc[r] = sum_d Y[32 t + rowmap(r, lh)][d] * U[own][d]
  auto tile = [&](const char* Y, const u32x4* fU, int t, f32x16& acc, bool first) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32x4 a = *(const u32x4*)(Y + lds_row_off(32 * t + l31, 2 * s + lh));
      if (first && s == 0) {        // the product starts from the constant 0 (no register zero-fill per tile)
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, fU[s]), z, 0, 0, 0);
      } else {
        mm(a, fU[s], acc);
      }
    }
  };
  struct Tiles { f32x16 S, St, dP, dPt; };
  auto zero = [](f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
  };
  auto compute_tiles = [&](int t, Tiles& X) {
    tile(Y1, fU1, t, X.S, true);
    if constexpr (TAN) { tile(Y1t, fU1, t, X.St, true); tile(Y1, fU1t, t, X.St, false); }
    if constexpr (DP) {
      tile(Y2, fU2, t, X.dP, true);
      if constexpr (TAN) { tile(Y2t, fU2, t, X.dPt, true); tile(Y2, fU2t, t, X.dPt, false); }
    }
  };
  auto dot8 = [](const u32x4& a, const u32x4& b) {     // sum of the eight bf16 products of two 16-byte fragments
    float fa[8], fb[8], s_ = 0.f;
    Chunk<bf16>::unpack(__builtin_bit_cast(uint4, a), fa);
    Chunk<bf16>::unpack(__builtin_bit_cast(uint4, b), fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) s_ += fa[e] * fb[e];
    return s_;
  };

  // ================================================================== statistics of the owner's query row (q-owner modes)
  // The vector ALU, not the matrix cores, bounds this kernel (157 TFLOP/s against 2.5 PFLOP/s): every instruction per
  // score element counts.  D and D_t are row dots of [tokens x 64] tensors (D = rowsum(P dP) = rowsum(O dO)), so only
  // the softmax maximum (mode 0) and r = rowsum(P S_t) (mode 1) still walk the score tiles twice.
  float q_nm = 0.f, q_linv = 1.f, q_r = 0.f, q_D = 0.f, q_Dt = 0.f;
  if constexpr (OQ && MODE != 0) { const float4 v = st4[own & 255]; q_nm = v.x; q_linv = v.y; q_r = v.z; }
  if constexpr (MODE == 2 || MODE == 3) {
    const bf16* ob = p.o + (size_t)img * T * Dm + (size_t)head * HD;
    u32x4 fO[4];
    load_frag(ob, Dm, fO);
    float d_ = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) d_ += dot8(fO[s], fU2[s]);
    d_ += __shfl_xor(d_, 32, 64);
    q_D = d_;
    if constexpr (MODE == 2) { if (lh == 0 && own_ok) p.D[sbase + own] = d_; }
    if constexpr (MODE == 3) {
      const bf16* obt = p.o_t + (size_t)img * T * Dm + (size_t)head * HD;
      u32x4 fOt[4];
      load_frag(obt, Dm, fOt);
      float e_ = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) e_ += dot8(fOt[s], fU2[s]) + dot8(fO[s], fU2t[s]);
      e_ += __shfl_xor(e_, 32, 64);
      q_Dt = e_;
      if (lh == 0 && own_ok) p.D_t[sbase + own] = e_;
    }
  }
  if constexpr (MODE == 0) {
    if (!(MDD_ATTN_DBG & 1)) {
      float mx = -3.0e38f;
#pragma unroll UNR
      for (int t = 0; t < nt; ++t) {
        Tiles X; compute_tiles(t, X);
        if (32 * t + 32 > T) {         // the partial tile (block-uniform branch)
#pragma unroll
          for (int r = 0; r < 16; ++r) if (32 * t + rowmap(r, lh) < T) mx = fmaxf(mx, X.S[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, X.S[r]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      q_nm = -mx * c1;
      if (lh == 0 && own_ok) p.m[sbase + own] = mx;
    }
  }
  if constexpr (MODE == 1) {
    if (!(MDD_ATTN_DBG & 1)) {
      float red = 0.f;
#pragma unroll UNR
      for (int t = 0; t < nt; ++t) {
        Tiles X; compute_tiles(t, X);
        if (32 * t + 32 > T) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (32 * t + rowmap(r, lh) < T) red = fmaf(__builtin_amdgcn_exp2f(fmaf(X.S[r], c1, q_nm)), X.St[r], red);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) red = fmaf(__builtin_amdgcn_exp2f(fmaf(X.S[r], c1, q_nm)), X.St[r], red);
        }
      }
      red *= q_linv;
      red += __shfl_xor(red, 32, 64);
      q_r = red;
      if (lh == 0 && own_ok) p.r[sbase + own] = red;
    }
  }

  // ================================================================== W tiles -> second product
  constexpr int NOUT = MODE == 4 ? 2 : 1;
  f32x16 o[NOUT][2];
#pragma unroll
  for (int k = 0; k < NOUT; ++k) { zero(o[k][0]); zero(o[k][1]); }
  float l_run = 0.f;                                   // mode 0: the softmax denominator, accumulated on the way
  auto pk8 = [](const float* w) {
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = Chunk<bf16>::pk(w[2 * e], w[2 * e + 1]);
    return v;
  };
  auto rfrag = [&](const char* Rt, int dt, int base) {   // 8 `other` values of column d = 32 dt + l31
    const unsigned short* row = (const unsigned short*)Rt + (size_t)(32 * dt + l31) * RTP;
    const uint2 a = *(const uint2*)(row + base), b = *(const uint2*)(row + base + 8);
    u32x4 v; v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    return v;
  };
#pragma unroll UNR
  for (int t = 0; t < ((MDD_ATTN_DBG & 2) ? 0 : nt); ++t) {
    Tiles X; compute_tiles(t, X);
    const float4* const sp4 = st4 + 32 * t + 4 * lh;     // statistics of query row 32 t + rowmap(r, lh): sp4[rowmap(r, 0)]
    const float* const spD = stDt + 32 * t + 4 * lh;
    if constexpr (OQ) {
      if (32 * t + 32 > T) {           // partial tile: the padded keys must not count (block-uniform branch): S := -inf
#pragma unroll
        for (int r = 0; r < 16; ++r) if (32 * t + rowmap(r, lh) >= T) X.S[r] = -3.0e38f;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // the eight W values of this lane for the 16-wide half h of the tile (registers 8 h .. 8 h + 7)
      float wP[8], wPt[TAN ? 8 : 1], wdS[DP ? 8 : 1], wdSt[(TAN && DP) ? 8 : 1];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 8 * h + i;
        float nm_ = q_nm, li = q_linv, r_ = q_r, D_ = q_D, Dt_ = q_Dt;
        if constexpr (!OQ) {        // the query is the OTHER index: its statistics come from LDS (1/l = 0 beyond the tokens)
          const float4 v = sp4[rowmap(r, 0)];
          nm_ = v.x; li = v.y; r_ = v.z; D_ = v.w;
          if constexpr (MODE == 6) Dt_ = spD[rowmap(r, 0)];
        }
        float P = __builtin_amdgcn_exp2f(fmaf(X.S[r], c1, nm_));
        if constexpr (MODE == 0) l_run += P; else P *= li;
        wP[i] = P;
        if constexpr (TAN || DP) {
          const float sP = sc * P;
          float dd = 0.f;
          if constexpr (DP) { dd = X.dP[r] - D_; wdS[i] = sP * dd; }
          if constexpr (TAN) {
            const float Pt = sP * (X.St[r] - r_);
            wPt[i] = Pt;
            if constexpr (DP) wdSt[i] = fmaf(sc * Pt, dd, sP * (X.dPt[r] - Dt_));
          }
        }
      }
      const int base = 32 * t + 16 * h + 4 * lh;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        if constexpr (MODE == 0) mm(pk8(wP), rfrag(R0, dt, base), o[0][dt]);
        else if constexpr (MODE == 1) { mm(pk8(wPt), rfrag(R0, dt, base), o[0][dt]); mm(pk8(wP), rfrag(R1, dt, base), o[0][dt]); }
        else if constexpr (MODE == 2) mm(pk8(wdS), rfrag(R0, dt, base), o[0][dt]);
        else if constexpr (MODE == 3) mm(pk8(wdSt), rfrag(R0, dt, base), o[0][dt]);
        else if constexpr (MODE == 4) {
          if (!p.skip0) mm(pk8(wP), rfrag(R0, dt, base), o[0][dt]);
          mm(pk8(wdS), rfrag(R1, dt, base), o[1][dt]);
        }
        else if constexpr (MODE == 5) { mm(pk8(wPt), rfrag(R0, dt, base), o[0][dt]); mm(pk8(wP), rfrag(R1, dt, base), o[0][dt]); }
        else mm(pk8(wdSt), rfrag(R0, dt, base), o[0][dt]);
      }
    }
  }
  if constexpr (MODE == 0) {
    // normalise: row rho of the output belongs to the lane with l31 = rho
    l_run += __shfl_xor(l_run, 32, 64);
    if (lh == 0 && own_ok) p.l[sbase + own] = l_run;
    const float linv = 1.f / l_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float f = __shfl(linv, rowmap(r, lh), 64);
      o[0][0][r] *= f; o[0][1][r] *= f;
    }
  }

  // ---- store.  The accumulators hold out[owner row 32 w + rowmap(r, lh)][32 dt + l31]: one 2-byte element per lane and
  // register.  They go through this wave's 4 KB of LDS (the row-major regions are free once every wave has left the tile
  // loop) so that global memory sees 16-byte chunks: 4 store instructions per wave and output instead of 32.
  __syncthreads();
  auto store = [&](const f32x16 (&acc)[2], char* region, bf16* dst, size_t stride, bool accum) {
    if (wave >= NT) return;                               // rows >= TP: never tokens
    unsigned short* stg16 = (unsigned short*)(region + wave * 4096);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bf16 v = (bf16)acc[dt][r];
        stg16[rowmap(r, lh) * 64 + 32 * dt + l31] = __builtin_bit_cast(unsigned short, v);
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this wave's LDS stores have landed
    __builtin_amdgcn_wave_barrier();
    if (MDD_ATTN_DBG & 4) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j, row = c >> 3, ch = c & 7, grow = wave * 32 + row;
      if (grow < T) {
        uint4 v = *(const uint4*)((const char*)stg16 + row * 128 + ch * 16);
        bf16* q = dst + (size_t)grow * stride + ch * 8;
        if (accum) {
          float a[8], b[8];
          Chunk<bf16>::unpack(v, a);
          Chunk<bf16>::unpack(*(const uint4*)q, b);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += b[e];
          v = Chunk<bf16>::pack(a);
        }
        *(uint4*)q = v;
      }
    }
  };
  if constexpr (MODE <= 1) {
    store(o[0], Y1, p.out + (size_t)img * T * Dm + (size_t)head * HD, Dm, false);
  } else {
    bf16* dq = p.out + (size_t)img * T * qrow + (size_t)head * HD;
    if constexpr (MODE == 2 || MODE == 3) store(o[0], Y1, dq + cq, qrow, p.accum != 0);
    else if constexpr (MODE == 4) { if (!p.skip0) store(o[0], Y1, dq + cv, qrow, false); store(o[1], Y2, dq + ck, qrow, p.accum != 0); }
    else if constexpr (MODE == 5) store(o[0], Y1, dq + cv, qrow, false);
    else store(o[0], Y1, dq + ck, qrow, false);
  }
}

template <int MODE> size_t attn_lds() {
  constexpr bool TAN = MODE == 1 || MODE == 3 || MODE == 5 || MODE == 6;
  constexpr bool DP = MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6;
  constexpr int NY = (TAN ? 2 : 1) * (DP ? 2 : 1);
  constexpr int NR = (MODE == 1 || MODE == 5) ? 2 : (MODE == 4 ? 2 : 1);
  return (size_t)NY * ROWB + (size_t)NR * TRB + 5 * 256 * sizeof(float);   // + st4[256] + stDt[256]
}
template <int MODE> int attn_launch(const AttnArgs& a, hipStream_t st) {
  const size_t shm = attn_lds<MODE>();
  // the dynamic-LDS limit is a per-device function attribute: set it once per (instance, device)
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)k_attn<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  k_attn<MODE><<<dim3((unsigned)a.heads, (unsigned)a.n), 512, shm, st>>>(a);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

// host entry (engine + C ABI).  pass: MDD_ATTN_*; the statistics buffers are [n, heads, tokens] fp32.
int launch_attention(int mode, const void* qkv, const void* qkv_t, const void* dout, const void* dout_t, void* out,
                     const void* o, const void* o_t, float* m, float* l, float* r, float* D, float* D_t, int n, int tokens,
                     int heads, float scale, int r_tan, int accum, int skip0, hipStream_t st) {
  AttnArgs a;
  a.qkv = (const bf16*)qkv; a.qkv_t = (const bf16*)qkv_t; a.dout = (const bf16*)dout; a.dout_t = (const bf16*)dout_t;
  a.out = (bf16*)out; a.o = (const bf16*)o; a.o_t = (const bf16*)o_t; a.m = m; a.l = l; a.r = r; a.D = D; a.D_t = D_t;
  a.n = n; a.tok = tokens; a.heads = heads; a.scale = scale; a.r_tan = r_tan; a.accum = accum; a.skip0 = skip0;
  switch (mode) {
    case 0: return attn_launch<0>(a, st);
    case 1: return attn_launch<1>(a, st);
    case 2: return attn_launch<2>(a, st);
    case 3: return attn_launch<3>(a, st);
    case 4: return attn_launch<4>(a, st);
    case 5: return attn_launch<5>(a, st);
    case 6: return attn_launch<6>(a, st);
  }
  return mdd_set_error_msg(2, "mdd: invalid argument: attention mode");
}
bool attention_fused_supported(int tokens, int head_dim) { return head_dim == HD && tokens <= TP && tokens >= 1; }

extern "C" int mdd_op_attention(int mode, int n, int tokens, int heads, float scale, const void* qkv, const void* qkv_t,
                                const void* dout, const void* dout_t, const void* o, const void* o_t, void* out, float* m,
                                float* l, float* r, float* D, float* D_t, int r_tan, int accum, int skip0, void* stream) {
  if (mode < 0 || mode > 6) return mdd_set_error_msg(2, "mdd: invalid argument: attention mode");
  if (!qkv || !out || !m || !l || n < 1 || heads < 1 || heads > 65535 || n > 65535)
    return mdd_set_error_msg(2, "mdd: invalid argument: attention operands");
  if (!attention_fused_supported(tokens, HD)) return mdd_set_error_msg(2, "mdd: invalid argument: attention needs tokens <= 224");
  const bool tan = mode == 1 || mode == 3 || mode == 5 || mode == 6;
  const bool dp = mode == 2 || mode == 3 || mode == 4 || mode == 6;
  if ((tan || r_tan) && !qkv_t) return mdd_set_error_msg(2, "mdd: invalid argument: attention tangent operand is null");
  if ((dp || mode == 5) && !dout) return mdd_set_error_msg(2, "mdd: invalid argument: attention output gradient is null");
  if ((mode == 3 || mode == 5 || mode == 6) && !dout_t) return mdd_set_error_msg(2, "mdd: invalid argument: attention dout_t is null");
  if ((mode == 1 || mode == 3 || mode == 5 || mode == 6) && !r) return mdd_set_error_msg(2, "mdd: invalid argument: attention r is null");
  if ((mode == 2 || mode == 3 || mode == 4 || mode == 6) && !D) return mdd_set_error_msg(2, "mdd: invalid argument: attention D is null");
  if ((mode == 3 || mode == 6) && !D_t) return mdd_set_error_msg(2, "mdd: invalid argument: attention D_t is null");
  if ((mode == 2 || mode == 3) && !o) return mdd_set_error_msg(2, "mdd: invalid argument: attention output o is null");
  if (mode == 3 && !o_t) return mdd_set_error_msg(2, "mdd: invalid argument: attention o_t is null");
  return launch_attention(mode, qkv, qkv_t, dout, dout_t, out, o, o_t, m, l, r, D, D_t, n, tokens, heads, scale, r_tan,
                          accum, skip0, (hipStream_t)stream);
}
