// Weight-gradient contraction of a convolution on NHWC activations, gfx950 MFMA.
//
//   dW[g][co][tap][kc] += sum_m dy[m, g*nc+co] * x[src(m,tap), g*kc_+kc]      (m = output pixel)
//
// Both operands have the REDUCTION index m as their slow (strided) dimension, so MFMA fragments
// need "k along rows" data:
//   f32 : v_mfma_f32_32x32x2_f32 takes one f32 per lane -- lane (i, k) reads LDS[row 2t+k][col i],
//         consecutive lanes hit consecutive dwords (conflict-free, no transpose needed).
//   bf16: v_mfma_f32_16x16x32_bf16 fragments are fetched with ds_read_b64_tr_b16 (the CDNA4
//         LDS transpose read): a 16-lane group reads a 4-row x 16-col block and each lane gets one
//         column.  Which 4-row block of a 32-row K-step a group reads per instruction is a free
//         k-permutation (same for A and B); it is chosen so the two groups of a half-wave read
//         rows 8x..8x+3 and 8x+4..8x+7, which with the XOR swizzle below is bank-conflict-free.
// Output tile BCO x BKP (64x128 for group width 64, 128x128 otherwise), BKM pixels per barrier,
// one LDS stage, the next tile's global loads in flight (staging registers) during the MFMA phase.
// The M range is split across blockIdx.y.  With a slab workspace (the engine's path) every split
// writes its partial tile with plain 16-byte stores into slab[split][g][co][k'] and k_wgrad_reduce sums
// the slabs into dW (overwriting: no zero-fill, no atomics -- fp32 atomics execute at the memory side at
// ~1.3 TB/s chip-wide and the 16x16 accumulator layout gave them 64-byte segments); without a slab
// (single-op entry point) the partials are combined with fp32 atomics into a dW zeroed by the caller.
// The MFMA operand roles are (x, dy): D rows = k', columns = co, so a lane's four accumulator registers
// are four CONSECUTIVE k' of one output channel = one 16-byte store.  An optional second pair (dy2, x2) is accumulated too (tangent pass), and the bias
// gradient (column sums of dy1) is taken from the staging registers of the k'-tile-0 blocks.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <type_traits>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

#ifndef MDD_WG_SINGLE_BUF
#define MDD_WG_SINGLE_BUF 1
#endif
#ifndef MDD_WG_PF2
#define MDD_WG_PF2 0
#endif
#ifndef MDD_WG_MIN_WAVES
#define MDD_WG_MIN_WAVES 1
#endif
#ifndef MDD_WG_PIPE_MIN
#define MDD_WG_PIPE_MIN 768  // narrowest layer (channels in and out) taken by the 256 x 256 pipelined kernel
#endif
#ifndef MDD_WG_SLAB
#define MDD_WG_SLAB 1     // grouped 3x3 stride-1 convolutions with 64-channel groups on k_wgrad_slab (all nine taps per pass)
#endif
#ifndef MDD_WGP_ABL
#define MDD_WGP_ABL 0   // timing experiments on k_wgrad_pipe (wrong results): 1 no MFMAs, 2 no LDS-DMA, 4 no fragment reads, 8 no write-out
#endif
#ifndef MDD_WG_TSTEP_SCALE
#define MDD_WG_TSTEP_SCALE 1.0   // split policy of k_conv_wgrad: < 1 weighs a K-step less = fewer, longer blocks (less slab traffic, less CU time)
#endif
#ifndef MDD_WGP_GROUP_SPLITS
#define MDD_WGP_GROUP_SPLITS 1   // pixel chunks of a grouped launch (0 = the cost model's choice): with ONE chunk the 108 tiles of a ViT layer go
                                 // straight into the gradient -- no partial tiles, no combine launch, and 148 CUs stay with the other stream
                                 // (measured: 1 chunk 407.0, 2 chunks (the model) 409.8, 3 chunks 418.4 ms per configs[4] iteration)
#endif
#ifndef MDD_WGP_SLOTS
#define MDD_WGP_SLOTS 8    // half-tile slots of k_wgrad_pipe's LDS ring (16 KB each); half-tiles are issued SLOTS-2 phases ahead.  The K loop is
                           // unrolled over EIGHT slots (static_assert): the 10-slot experiment of profiles/r03_experiments.md predates that
#endif
#ifndef MDD_WG_BKM
#define MDD_WG_BKM 64      // pixels per K-step of the bf16 instances
#endif

namespace {

struct WArgs {
  const void* dy1; const void* x1; const void* dy2; const void* x2;
  float* dW; float* dbias;
  float* slab;              // [splits][groups*nc*ktot] partial sums, or null: atomics into dW
  int64_t slab_stride;      // floats per split
  ConvGeom g;
  int M, cotiles, kptiles, mchunk, dbg;
};

typedef unsigned __attribute__((ext_vector_type(4))) u32x4;   // first-class 16-byte register value

// LDS chunk swizzle (bf16 path only): rows of ROWB bytes
template <int ROWB> DEVI int wswz(int row) { return ROWB == 128 ? ((row >> 1) & 3) : (row & 7); }

// PW: pointwise (1x1, stride 1, no padding) instance -- no gather state, no per-step pixel decode
// PF2: two tiles in flight in staging registers (loads issued two K-steps before their LDS store)
// SPLIT (fp32 storage only): split-bf16 math -- each fp32 element is split into hi = bf16(x) and
// lo = bf16(x - hi) between the global load and the LDS store, into two bf16 planes with the bf16 path's
// layout; the contraction is hh + hl + lh on v_mfma_f32_16x16x32_bf16 (the ll term is below 2^-16 relative).
// F16OPS (with SPLIT; ConvGeom::prec == 3): the fp32 operands are rounded to ONE fp16 each (11 significand bits)
// and multiplied by v_mfma_f32_16x16x32_f16 -- the precision experiment of DESIGN.md section 5
template <class AT, int BCO, int BKP, int BKM, bool PW, bool PF2, bool SPLIT, bool F16OPS = false>
__global__ __launch_bounds__(256, MDD_WG_MIN_WAVES) void k_conv_wgrad(const WArgs p) {
  constexpr int CE = 16 / (int)sizeof(AT);
  constexpr bool BF = sizeof(AT) == 2;
  constexpr bool TR = BF || SPLIT;                       // fragments by ds_read_b64_tr_b16 from bf16 planes
  constexpr int TD = BCO * 2, TX = BKP * 2;              // bf16 plane row bytes
  static_assert(!SPLIT || sizeof(AT) == 4, "split-bf16 math is a mode of the fp32-storage engine");
  constexpr int DCH = BCO / CE, XCH = BKP / CE;          // chunks per row
  constexpr int DROWB = BCO * sizeof(AT), XROWB = BKP * sizeof(AT);
  constexpr int DSL = (BKM * DCH) / 256, XSL = (BKM * XCH) / 256;   // chunks per thread
  constexpr int DSTEP = 256 / DCH, XSTEP = 256 / XCH;
  constexpr int DTILE = BKM * DROWB, XTILE = BKM * XROWB;
  constexpr int WCO = BCO / 64, WKP = 4 / WCO;           // wave grid over (co, k')
  constexpr int WKW = BKP / WKP;                         // k' width per wave
  static_assert(DSL >= 1 && XSL >= 1, "tile too small for 256 threads");
  static_assert(TR || (BCO == 64 && BKP == 128), "f32 path: 64x128 tile only");
  constexpr int NBUF = MDD_WG_SINGLE_BUF ? 1 : 2;   // LDS stages (registers hold the tile in flight)
  __shared__ __attribute__((aligned(16))) char smem[NBUF * (DTILE + XTILE)];
  char* Ds = smem;
  char* Xs = smem + NBUF * DTILE;

  const ConvGeom& G = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / WKP, wk = wave - wc * WKP;
  // XCD-aware remap of the linear block id (8 XCDs, private L2s, round-robin dispatch): all output
  // tiles of ONE M-chunk get consecutive logical ids -> one XCD -> its dy / x rows are fetched
  // through the fabric once instead of once per XCD.  Bijective for any grid size.
  int bid, bsplit;
  {
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.x + blockIdx.y * gridDim.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    bsplit = lin / gridDim.x;
    bid = lin - bsplit * gridDim.x;
  }
  const int kpt = bid % p.kptiles; bid /= p.kptiles;
  const int cot = bid % p.cotiles;
  const int grp = bid / p.cotiles;
  const int co0 = cot * BCO, kp0 = kpt * BKP;
  const int ktot = G.k * G.k * G.kc;
  const int mbeg = bsplit * p.mchunk;
  const int mend = min(p.M, mbeg + p.mchunk);
  const int niter1 = mend > mbeg ? (mend - mbeg + BKM - 1) / BKM : 0;
  const int niter = (MDD_DBG_BITS(p) & 1) ? (niter1 > 0 ? 1 : 0) : (p.dy2 ? 2 * niter1 : niter1);   // dbg bit0: timing only

  // ---- staging geometry.  dy: chunk col dcol, rows drow + DSTEP*i ; x: xcol, rows xrow + XSTEP*i.
  // All global loads are BUFFER loads: a wave-uniform descriptor + a 32-bit per-lane byte offset that is
  // fixed for the whole launch + a scalar offset that advances BKM pixels per step.  Elements that must
  // read as zero (rows past the M chunk, taps outside the image, K / channel tails) get an
  // out-of-range offset -- the hardware returns 0 -- so a load costs one compare + one select instead
  // of 64-bit address arithmetic and a mask (this loop is instruction-issue bound).
  const int dcol = tid % DCH, drow = tid / DCH;
  const int xcol = tid % XCH, xrow = tid / XCH;
  // x chunk -> (tap, kc) is iteration-invariant
  const int kp = kp0 + xcol * CE;
  const bool kp_ok = kp < ktot;
  const int tap = kp / G.kc, kcq = kp - tap * G.kc;
  const int ty = tap / G.k, tx = tap - ty * G.k;
  const bool dco_ok = (co0 + dcol * CE) < G.nc;
  constexpr bool pointwise = PW;
  // source pixel index is (output pixel index + constant) for stride-1 "same" convolutions
  const bool lin = pointwise || (G.stride == 1 && G.ha == G.ho && G.wa == G.wo);
  constexpr int ESZ = (int)sizeof(AT);
  constexpr unsigned OOB = 0x80000000u;
  const int margin = lin ? G.pad * G.wa + G.pad : 0;          // keeps every per-lane offset >= 0
  auto rsrc = [&](const void* base, int64_t elem_off) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + elem_off * ESZ), (short)0, 0x7fffffff,
                                             0x00020000);
  };
  const int64_t dy_base = (int64_t)mbeg * G.co_tot + (int64_t)grp * G.nc + co0;
  const int64_t x_base = lin ? ((int64_t)mbeg - margin) * G.ca_tot + (int64_t)grp * G.kc : (int64_t)grp * G.kc;
  const __amdgpu_buffer_rsrc_t rs_dy1 = rsrc(p.dy1, dy_base), rs_x1 = rsrc(p.x1, x_base);
  const __amdgpu_buffer_rsrc_t rs_dy2 = rsrc(p.dy2 ? p.dy2 : p.dy1, dy_base), rs_x2 = rsrc(p.x2 ? p.x2 : p.x1, x_base);
  unsigned dvo[DSL], xvo[XSL];
#pragma unroll
  for (int i = 0; i < DSL; ++i)
    dvo[i] = dco_ok ? (unsigned)(((drow + DSTEP * i) * G.co_tot + dcol * CE) * ESZ) : OOB;
#pragma unroll
  for (int i = 0; i < XSL; ++i) {
    int shift = lin ? margin + (ty - G.pad) * G.wa + (tx - G.pad) : 0;
    xvo[i] = kp_ok ? (unsigned)(((xrow + XSTEP * i + shift) * G.ca_tot + kcq) * ESZ) : OOB;
  }
  const int dstep = BKM * G.co_tot * ESZ, xstep = BKM * G.ca_tot * ESZ;   // bytes per K-step

  // running (image, oy, ox) of each x row slot (gathered convs only); advanced by BKM pixels per iteration
  int sox[XSL], soy[XSL], sni[XSL];
  auto init_rows = [&]() {
#pragma unroll
    for (int i = 0; i < XSL; ++i) {
      int m = mbeg + xrow + XSTEP * i;
      sox[i] = m % G.wo;
      int t = m / G.wo;
      soy[i] = t % G.ho;
      sni[i] = t / G.ho;
    }
  };
  if constexpr (!pointwise) init_rows();

  constexpr int NST = PF2 ? 2 : 1;
  u32x4 rdS[NST][DSL], rxS[NST][XSL];
  bool biasS[NST] = {};
  float bsum[CE];
#pragma unroll
  for (int e = 0; e < CE; ++e) bsum[e] = 0.f;
  const bool do_bias = p.dbias != nullptr && kpt == 0;
  const bool hi_only = SPLIT && p.g.prec == 2;   // attribution experiment: bf16 operands from an fp32 stash

  auto load_tile = [&](auto SI, int it) {
    constexpr int si = decltype(SI)::value;
    u32x4* rd = rdS[si]; u32x4* rx = rxS[si];
    const bool second = it >= niter1;
    if constexpr (!pointwise) { if (it == niter1) init_rows(); }
    const int itl = second ? it - niter1 : it;
    const int rem = mend - mbeg - itl * BKM;          // rows of this chunk still ahead (uniform)
    const __amdgpu_buffer_rsrc_t rd_ = second ? rs_dy2 : rs_dy1;
    const __amdgpu_buffer_rsrc_t rx_ = second ? rs_x2 : rs_x1;
#pragma unroll
    for (int i = 0; i < DSL; ++i) {
      unsigned vo = (drow + DSTEP * i) < rem ? dvo[i] : OOB;
      rd[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd_, (int)vo, itl * dstep, 0));
    }
#pragma unroll
    for (int i = 0; i < XSL; ++i) {
      bool ok = (xrow + XSTEP * i) < rem;
      unsigned vo = xvo[i];
      int so = itl * xstep;
      if constexpr (!pointwise) {
        int iy = soy[i] * G.stride - G.pad + ty, ix = sox[i] * G.stride - G.pad + tx;
        ok = ok && (unsigned)iy < (unsigned)G.ha && (unsigned)ix < (unsigned)G.wa;
        if (!lin) {   // strided: the source pixel is not linear in m -- absolute offset, no scalar advance
          vo = kp_ok ? (unsigned)((((sni[i] * G.ha + iy) * G.wa + ix) * G.ca_tot + kcq) * ESZ) : OOB;
          so = 0;
        }
        // advance this slot to the next iteration's pixel
        sox[i] += BKM;
        while (sox[i] >= G.wo) { sox[i] -= G.wo; ++soy[i]; }
        while (soy[i] >= G.ho) { soy[i] -= G.ho; ++sni[i]; }
      }
      rx[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, (int)(ok ? vo : OOB), so, 0));
    }
    biasS[si] = do_bias && !second;
  };
  auto store_tile = [&](auto SI, int buf) {
    constexpr int si = decltype(SI)::value;
    const u32x4* rd = rdS[si]; const u32x4* rx = rxS[si];
    const bool bias_now = biasS[si];
    char* d = Ds + buf * DTILE;
    char* x = Xs + buf * XTILE;
    auto split_store = [&](char* plane_hi, int plane_bytes, int row_bytes, int r, int unit, const u32x4& raw) {
      unsigned h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float f = __uint_as_float(raw[e]);
        if constexpr (F16OPS) {
          _Float16 hh = (_Float16)f;
          h[e] = (unsigned)__builtin_bit_cast(unsigned short, hh);
          l[e] = 0u;
        } else {
          bf16 hb = (bf16)f;
          h[e] = (unsigned)__builtin_bit_cast(unsigned short, hb);
          bf16 lb = (bf16)(f - __uint_as_float(h[e] << 16));
          l[e] = hi_only ? 0u : (unsigned)__builtin_bit_cast(unsigned short, lb);
        }
      }
      char* a = plane_hi + r * row_bytes + unit * 8;
      *(uint2*)a = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
      *(uint2*)(a + plane_bytes) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    };
#pragma unroll
    for (int i = 0; i < DSL; ++i) {
      int r = drow + DSTEP * i;
      if constexpr (SPLIT) {
        split_store(d, BKM * TD, TD, r, dcol ^ (wswz<TD>(r) << 2), rd[i]);
      } else {
        int c = BF ? (dcol ^ (wswz<DROWB>(r) << 1)) : dcol;
        *(u32x4*)(d + r * DROWB + c * 16) = rd[i];
      }
      if (bias_now) {
        float f[CE];
        Chunk<AT>::unpack(__builtin_bit_cast(uint4, rd[i]), f);
#pragma unroll
        for (int e = 0; e < CE; ++e) bsum[e] += f[e];
      }
    }
#pragma unroll
    for (int i = 0; i < XSL; ++i) {
      int r = xrow + XSTEP * i;
      if constexpr (SPLIT) {
        split_store(x, BKM * TX, TX, r, xcol ^ (wswz<TX>(r) << 2), rx[i]);
      } else {
        int c = BF ? (xcol ^ (wswz<XROWB>(r) << 1)) : xcol;
        *(u32x4*)(x + r * XROWB + c * 16) = rx[i];
      }
    }
  };

  // accumulators: f32 -> 2 blocks of 32x32 (co 0..31, 32..63) x (32 k' of this wave)
  //               bf16 -> 4 x (WKW/16) blocks of 16x16 (64 co x WKW k' per wave)
  constexpr int NBK = TR ? WKW / 16 : 1;
  f32x16 accf[2];
  f32x4 accb[4][NBK];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accf[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) accb[i][j][r] = 0.f;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, PF2 ? 1 : 0>;
  if (niter > 0) {
    load_tile(S0{}, 0);
    if (PF2 && niter > 1) load_tile(S1{}, 1);
    store_tile(S0{}, 0);
  }
  __syncthreads();

  // one K-step: tile `it` is in LDS; the staging set SN holds tile it+1; (PF2) the free set SF takes it+2
  auto step = [&](auto SF, auto SN, int it) {
    const int buf = NBUF == 2 ? (it & 1) : 0;
    if constexpr (PF2) { if (it + 2 < niter) load_tile(SF, it + 2); }
    else { if (it + 1 < niter) load_tile(SN, it + 1); }
    const char* d = Ds + buf * DTILE;
    const char* x = Xs + buf * XTILE;
    if constexpr (SPLIT) {
      const int gq = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
#pragma unroll
      for (int ks = 0; ks < BKM / 32; ++ks) {
        s16x4 ah[4][2], al[4][2], bh[NBK][2], bl[NBK][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          int row = ks * 32 + 4 * (2 * (2 * e + (gq >> 1)) + (gq & 1)) + q;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int unit = (wc * 64 + 16 * i + 4 * pp) >> 2;
            const char* a = d + row * TD + (unit ^ (wswz<TD>(row) << 2)) * 8;
            ah[i][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
            al[i][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + BKM * TD));
          }
#pragma unroll
          for (int j = 0; j < NBK; ++j) {
            int unit = (wk * WKW + 16 * j + 4 * pp) >> 2;
            const char* a = x + row * TX + (unit ^ (wswz<TX>(row) << 2)) * 8;
            bh[j][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
            bl[j][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + BKM * TX));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NBK; ++j) {
            bf16x8 a8h = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(ah[i][0], ah[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
            bf16x8 a8l = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(al[i][0], al[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
            bf16x8 b8h = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(bh[j][0], bh[j][1], 0, 1, 2, 3, 4, 5, 6, 7));
            bf16x8 b8l = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(bl[j][0], bl[j][1], 0, 1, 2, 3, 4, 5, 6, 7));
            if constexpr (F16OPS) {
              typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
              accb[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b8h), __builtin_bit_cast(f16x8, a8h),
                                                                  accb[i][j], 0, 0, 0);
            } else {
              accb[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b8h, a8h, accb[i][j], 0, 0, 0);
              accb[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b8l, a8h, accb[i][j], 0, 0, 0);
              accb[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b8h, a8l, accb[i][j], 0, 0, 0);
            }
          }
      }
    } else if constexpr (!BF) {
      const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
      for (int t = 0; t < BKM / 2; ++t) {
        int r = 2 * t + lh;
        float a0 = *(const float*)(d + r * DROWB + l31 * 4);
        float a1 = *(const float*)(d + r * DROWB + (32 + l31) * 4);
        float b = *(const float*)(x + r * XROWB + (wk * 32 + l31) * 4);
        accf[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a0, accf[0], 0, 0, 0);
        accf[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a1, accf[1], 0, 0, 0);
      }
    } else {
      const int gq = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
      typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
      for (int ks = 0; ks < BKM / 32; ++ks) {
        // 4-row block read by this lane group in instruction e: rho(g,e) = 2*(2e + (g>>1)) + (g&1)
        s16x4 af[4][2], bfr[NBK][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          int row = ks * 32 + 4 * (2 * (2 * e + (gq >> 1)) + (gq & 1)) + q;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int unit = (wc * 64 + 16 * i + 4 * pp) >> 2;  // 8-byte unit index within the row
            int u = unit ^ (wswz<DROWB>(row) << 2);
            af[i][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(d + row * DROWB + u * 8));
          }
#pragma unroll
          for (int j = 0; j < NBK; ++j) {
            int unit = (wk * WKW + 16 * j + 4 * pp) >> 2;
            int u = unit ^ (wswz<XROWB>(row) << 2);
            bfr[j][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(x + row * XROWB + u * 8));
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NBK; ++j) {
            s16x8 a8 = __builtin_shufflevector(af[i][0], af[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
            s16x8 b8 = __builtin_shufflevector(bfr[j][0], bfr[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
            accb[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, b8), __builtin_bit_cast(bf16x8, a8), accb[i][j], 0, 0, 0);
          }
      }
    }
    if (NBUF == 1) __syncthreads();      // everyone is done reading the tile before it is replaced
    if (it + 1 < niter) store_tile(SN, NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  };
  if constexpr (PF2) {
    for (int it = 0; it < niter; it += 2) {
      step(S0{}, S1{}, it);
      if (it + 1 < niter) step(S1{}, S0{}, it + 1);
    }
  } else {
    for (int it = 0; it < niter; ++it) step(S0{}, S0{}, it);
  }

  if (MDD_DBG_BITS(p) & 2) return;   // dbg bit1: no write-out (timing only)
  // ---- write-out.  Accumulator maps (operands (x, dy): rows = k', columns = co):
  //   32x32 (f32)  : col = lane&31 -> co ; row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> k'
  //   16x16 (bf16) : col = lane&15 -> co ; row = 4*(lane>>4) + r -> k'
  // so registers r = 4t..4t+3 are four consecutive k' of one channel: one float4.
  const size_t gbase = (size_t)grp * G.nc * ktot;
  float* dst = p.slab ? p.slab + (size_t)bsplit * p.slab_stride + gbase : p.dW + gbase;
  const bool plain = p.slab != nullptr;
  auto put4 = [&](int co, int kcol, float v0, float v1, float v2, float v3) {
    if (co >= G.nc || kcol >= ktot) return;     // ktot and kcol are multiples of 4
    float* a = dst + (size_t)co * ktot + kcol;
    if (plain) {
      *(float4*)a = make_float4(v0, v1, v2, v3);
    } else {
      atomicAdd(a, v0); atomicAdd(a + 1, v1); atomicAdd(a + 2, v2); atomicAdd(a + 3, v3);
    }
  };
  if constexpr (!TR) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        put4(co0 + 32 * i + l31, kp0 + wk * 32 + 8 * t + 4 * lh, accf[i][4 * t], accf[i][4 * t + 1],
             accf[i][4 * t + 2], accf[i][4 * t + 3]);
  } else {
    const int gq = lane >> 4, li = lane & 15;
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        put4(co0 + wc * 64 + 16 * i + li, kp0 + wk * WKW + 16 * j + 4 * gq, accb[i][j][0], accb[i][j][1],
             accb[i][j][2], accb[i][j][3]);
  }

  // ---- bias gradient: reduce the per-thread column sums over the rows that share a column
  if (do_bias) {
    __syncthreads();
    float* sh = (float*)smem;  // [DSTEP rows][DCH cols][CE]
#pragma unroll
    for (int e = 0; e < CE; ++e) sh[(drow * DCH + dcol) * CE + e] = bsum[e];
    __syncthreads();
    if (tid < BCO) {
      int c = tid / CE, e = tid - c * CE;
      float s = 0.f;
      for (int r = 0; r < DSTEP; ++r) s += sh[(r * DCH + c) * CE + e];
      if (co0 + tid < G.nc) atomicAdd(p.dbias + grp * G.nc + co0 + tid, s);
    }
  }
}

// dW[i] = sum_s slab[s][i].  A block owns 16 consecutive float4 columns; its 256 threads are 16 split lanes
// x 16 columns: lane l sums splits l, l+16, ... (8 independent 16-byte loads in flight), the 16 partial sums
// of a column meet in LDS.  Layers with tiny outputs run thousands of splits (stem: M = 1.25 M pixels onto a
// 4.6 K-float dW): one thread per column walking all of them serially took >100 us on the pass's tail.
__global__ __launch_bounds__(256) void k_wgrad_reduce(float* __restrict__ dW, const float* __restrict__ slab,
                                                        int64_t n4, int64_t stride4, int splits) {
  __shared__ float4 part[16][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + col;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* src = (const float4*)slab + i;
    for (int k = sl; k < splits; k += 16 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = (k + 16 * u < splits) ? src[(int64_t)(k + 16 * u) * stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  part[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < n4) {
    float4 t = part[0][col];
#pragma unroll
    for (int l = 1; l < 16; ++l) { float4 v = part[l][col]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    ((float4*)dW)[i] = t;
  }
}

// ======================================================================================================
// Wide pointwise layers in bf16 (the ViT linears: 19,700 token rows onto 768..3072 x 768..3072 weights): a
// 256 (co) x 256 (k') output tile per 512-thread block, one block per CU.
//   * 8 waves as 2 (co) x 4 (k'), wave tile 128 x 64 = 8 x 4 accumulator blocks of v_mfma_f32_16x16x32_bf16
//   * a K-tile = 64 pixels = four HALF-TILES of 64 rows x 256 B (16 KB): D0, X0, X1, D1.  D_h holds channels
//     [64h, 64h+64) of BOTH wave rows, X_h columns [32h, 32h+32) of all four wave columns: every wave needs the
//     half-tiles in the same order.  They live in an 8-slot LDS ring (128 KB) filled by global_load_lds_dwordx4
//     -- no staging registers, no ds_write -- six phases ahead of their use; the chunk swizzle of the one-stage
//     kernel is applied on the SOURCE address (the hardware writes lane-linear).  Pixels past the block's chunk
//     and channels past the tensor read a block of zeros.
//   * a K-tile is four PHASES: {ds_read_b64_tr_b16 this phase's fragments, issue one half-tile, counted vmcnt}
//     barrier {16 MFMAs: one 64 x 32 quadrant x 64 pixels} barrier.  Wave row 1 runs ONE barrier behind wave
//     row 0, so of the two waves of a SIMD one issues MFMAs while the other reads LDS.
//   Hazards (g = phase index; both wave rows counted): a slot is read in ONE phase g_r and refilled by the
//   issue of phase g_r + 2 or later (the lagging row's reads are retired by its lgkmcnt(0) two barriers before);
//   a half-tile is read one phase after the phase whose vmcnt retired it (the lagging row's wait precedes the
//   barrier the leading row passes before reading).
// tools/micro/gemm_pipe.hip holds the same schedule on K-contiguous operands with its timings.
__device__ __attribute__((aligned(16))) const unsigned g_wzero[64] = {0};

template <int N_> DEVI void wait_vm() {
  __builtin_amdgcn_s_waitcnt((N_ & 0xF) | ((N_ >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}
DEVI void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

constexpr int WGP_NS = MDD_WGP_SLOTS, WGP_D = WGP_NS - 2;   // ring slots, issue distance in phases (see "Hazards")
// one to four layers with the same pixel count in one launch (a ViT layer's four linears: 108 tiles instead of 9-36,
// two or three pixel chunks instead of seven -- a third of the partial-tile traffic)
struct WProb {
  const void *dy1, *x1, *dy2, *x2;
  float* dW; float* dbias;
  int64_t slab_off;            // floats, inside one split's slab
  int nc, kc, co_tot, ca_tot, kptiles, tile_start;
};
struct WGroup {
  WProb pr[4];
  int n, M, mchunk, tiles;
  int direct;                  // one pixel chunk: the tiles are stored straight into dW (no slab, no combine)
  float* slab; int64_t slab_stride;
};
__global__ __launch_bounds__(512, 1) void k_wgrad_pipe(const WGroup pg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = WGP_NS, D = WGP_D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wc = wave >> 2, wk = wave & 3;
  int bid, bsplit;
  {
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.x + blockIdx.y * gridDim.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    bsplit = lin / gridDim.x;
    bid = lin - bsplit * gridDim.x;
  }
  int pi = 0;
  while (pi + 1 < pg.n && bid >= pg.pr[pi + 1].tile_start) ++pi;
  const WProb& q_ = pg.pr[pi];
  bid -= q_.tile_start;
  const int kpt = bid % q_.kptiles, cot = bid / q_.kptiles;
  const int co0 = cot * 256, kp0 = kpt * 256;
  const int ktot = q_.kc;
  const int mbeg = bsplit * pg.mchunk;
  const int mend = min(pg.M, mbeg + pg.mchunk);
  const int rows = mend - mbeg;
  const int nk1 = (rows + 63) >> 6;
  const int nk = q_.dy2 ? 2 * nk1 : nk1;
  const int H = 4 * nk;

  // ---- loader: two 1-KB wave instructions per half-tile; instruction j covers pixel rows j*32 + wave*4 + (lane>>4),
  // this lane's 16-byte slot lane&15 of the row receives the chunk (slot ^ swizzle(row)) of the half-tile's 128 columns
  // (both instructions and both halves share one offset: rows 32 apart have the same swizzle, halves are 64 / 32 columns apart)
  const int lrow0 = wave * 4 + (lane >> 4);
  const int lcidx = (lane & 15) ^ ((lrow0 & 7) << 1);
  const int dch0 = co0 + lcidx * 8;       // channel of half 0; half 1 is + 128: every half-tile row is ONE 256-byte segment
  const int xkk0 = kp0 + lcidx * 8;
  const unsigned dv0 = (unsigned)((lrow0 * q_.co_tot + dch0) * 2), xv0 = (unsigned)((lrow0 * q_.ca_tot + xkk0) * 2);
  const unsigned drow32 = (unsigned)(32 * q_.co_tot * 2), xrow32 = (unsigned)(32 * q_.ca_tot * 2);
  const char* const zsrc = (const char*)g_wzero + (lane & 15) * 16;
  const size_t dstep = (size_t)64 * q_.co_tot * 2, xstep = (size_t)64 * q_.ca_tot * 2;
  const char* const dy1b = (const char*)q_.dy1 + (size_t)mbeg * q_.co_tot * 2;
  const char* const x1b = (const char*)q_.x1 + (size_t)mbeg * q_.ca_tot * 2;
  const char* const dy2b = (const char*)(q_.dy2 ? q_.dy2 : q_.dy1) + (size_t)mbeg * q_.co_tot * 2;
  const char* const x2b = (const char*)(q_.x2 ? q_.x2 : q_.x1) + (size_t)mbeg * q_.ca_tot * 2;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  // half-tile hh = 4*kt + {0: D0, 1: X0, 2: X1, 3: D1} -> ring slot hh % NS (the caller passes it)
  auto issue = [&](int hh, int slot) __attribute__((always_inline)) {
    const int kt = hh >> 2, jj = hh & 3;
    const bool second = kt >= nk1;
    const int ktl = second ? kt - nk1 : kt;
    const bool isd = jj == 0 || jj == 3;
    const int h = jj >= 2 ? 1 : 0;
    const char* base = isd ? (second ? dy2b : dy1b) + ktl * dstep : (second ? x2b : x1b) + ktl * xstep;
    const int rem = rows - ktl * 64;
    if constexpr (MDD_WGP_ABL & 2) return;
    char* dst = smem + slot * 16384 + wave * 1024;
    const bool colok = isd ? (dch0 + 128 * h) < q_.nc : (xkk0 + 128 * h) < ktot;
    const unsigned vo = (isd ? dv0 : xv0) + 256u * h;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* src = (colok && lrow0 + 32 * j < rem) ? base + vo + (j ? (isd ? drow32 : xrow32) : 0u) : zsrc;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + j * 8192), 16, 0, 0);
    }
  };

  // ---- fragment geometry (as the one-stage kernel): 16-lane group gq, lane li = 4q + pp reads the 4-row block
  // rho(gq, e) of a 32-pixel K-step, rows 8x..8x+3 / 8x+4..8x+7 per half-wave: conflict-free with the swizzle
  const int gq = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  // byte offsets inside a half-tile slot of block 0 (e = 0, 1); block i is XOR 32*i (the swizzle and the block index
  // occupy the same bits of the 8-byte unit index), the second 32-pixel step is + 8192
  unsigned rdD0[2], rdX0[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 4 * (2 * (2 * e + (gq >> 1)) + (gq & 1)) + q;
    const int sw = (row & 7) << 2;
    rdD0[e] = (unsigned)(row * 256 + ((((wc * 64 + 4 * pp) >> 2) ^ sw) << 3));
    rdX0[e] = (unsigned)(row * 256 + ((((wk * 32 + 4 * pp) >> 2) ^ sw) << 3));
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // bias gradient (column sums of dy1) from the D half-tiles in LDS, by the blocks of k'-tile 0: thread t adds the
  // chunks (row t>>4 + 32u, slot t&15) -- one fixed 8-channel chunk of the half-tile per thread
  const bool do_bias = q_.dbias != nullptr && kpt == 0;
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
  const int brow = (tid & 255) >> 4, bslot = tid & 15;     // waves 0-3 sum D0, waves 4-7 D1: rows brow + 16u
  auto bias_add = [&](unsigned slot) __attribute__((always_inline)) {     // asm reads, as the fragments: no vmcnt(0)
    const unsigned a = slot + (unsigned)(brow * 256 + bslot * 16);
    u32x4 c[4];
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\t"
                 "ds_read_b128 %3, %4 offset:12288\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]) : "v"(a));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      Chunk<bf16>::unpack(__builtin_bit_cast(uint4, c[u]), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum[e] += f[e];
    }
  };

  // The transposing reads are inline asm: hipcc orders a ds_read_tr builtin behind EVERY outstanding LDS-DMA
  // (s_waitcnt vmcnt(0) in front of it), which would drain the ring each phase.  Their completion is counted by the
  // wait statements below, which name every destination register so that no consumer is scheduled above them.
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  s16x4 af[2][2][4], b0[2][2][2], b1[2][2][2];
  // fragment addresses inside slot 0, one register per (e, block): a read then costs no vector-ALU instruction -- the slot
  // (compile-time inside the K loop, unrolled by two K-tiles = the eight ring slots) and the second 32-pixel step go into
  // the instruction's 16-bit offset, slots 4..7 through + 65536 on the registers once per K-tile.  With the address
  // arithmetic per read, phase 0 of a wave row (24 reads) outlasted the other row's 16 MFMAs.
  unsigned aD[2][4], aX[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
#pragma unroll
    for (int i = 0; i < 4; ++i) aD[e][i] = (lds0 + rdD0[e]) ^ (unsigned)(32 * i);
#pragma unroll
    for (int j = 0; j < 2; ++j) aX[e][j] = (lds0 + rdX0[e]) ^ (unsigned)(32 * j);
  }
  auto trd = [](unsigned a, auto OFF) __attribute__((always_inline)) {
    s16x4 v;
    if constexpr (MDD_WGP_ABL & 4) asm volatile("" : "=v"(v) : "v"(a)); else
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(decltype(OFF)::value));
    return v;
  };
  // SL: slot inside the half of the ring (0..3); hi: 0 or 65536 (slots 4..7)
  auto read_d = [&](auto SL, unsigned hi) __attribute__((always_inline)) {
    constexpr int o = decltype(SL)::value * 16384;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[0][e][i] = trd(aD[e][i] + hi, std::integral_constant<int, o>{});
        af[1][e][i] = trd(aD[e][i] + hi, std::integral_constant<int, o + 8192>{});
      }
  };
  auto read_x = [&](auto SL, unsigned hi, s16x4 (&b)[2][2][2]) __attribute__((always_inline)) {
    constexpr int o = decltype(SL)::value * 16384;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b[0][e][j] = trd(aX[e][j] + hi, std::integral_constant<int, o>{});
        b[1][e][j] = trd(aX[e][j] + hi, std::integral_constant<int, o + 8192>{});
      }
  };
#define MDD_B8(b) "+v"(b[0][0][0]), "+v"(b[0][0][1]), "+v"(b[0][1][0]), "+v"(b[0][1][1]), "+v"(b[1][0][0]), "+v"(b[1][0][1]), \
                  "+v"(b[1][1][0]), "+v"(b[1][1][1])
#define MDD_A16 "+v"(af[0][0][0]), "+v"(af[0][0][1]), "+v"(af[0][0][2]), "+v"(af[0][0][3]), "+v"(af[0][1][0]), "+v"(af[0][1][1]), \
                "+v"(af[0][1][2]), "+v"(af[0][1][3]), "+v"(af[1][0][0]), "+v"(af[1][0][1]), "+v"(af[1][0][2]), "+v"(af[1][0][3]),  \
                "+v"(af[1][1][0]), "+v"(af[1][1][1]), "+v"(af[1][1][2]), "+v"(af[1][1][3])
  auto quad = [&](const s16x4 (&b)[2][2][2], int ia, int jb) __attribute__((always_inline)) {
    if constexpr (MDD_WGP_ABL & 1) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const s16x8 a8 = __builtin_shufflevector(af[ks][0][i], af[ks][1][i], 0, 1, 2, 3, 4, 5, 6, 7);
          const s16x8 b8 = __builtin_shufflevector(b[ks][0][j], b[ks][1][j], 0, 1, 2, 3, 4, 5, 6, 7);
          acc[ia + i][jb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b8), __builtin_bit_cast(bf16x8, a8),
                                                                        acc[ia + i][jb + j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };

#pragma unroll
  for (int hh = 0; hh < D; ++hh)
    if (hh < H) issue(hh, hh);
  if (H >= D + 2) wait_vm<2 * (D - 2)>(); else wait_vm<0>();    // D0, X0 of K-tile 0 have landed (this wave's share)
  raw_barrier();
  if (wc == 1) raw_barrier();
  static_assert(NS == 8 && D == 6, "the K loop below is unrolled over the eight ring slots");
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  // one K-tile whose half-tiles sit in slots 4 PAR .. 4 PAR + 3 (PAR = kt & 1)
  auto ktile = [&](const int kt, auto PARc) __attribute__((always_inline)) {
    constexpr int PAR = decltype(PARc)::value;
    const unsigned hi = PAR ? 65536u : 0u;
    const int g0 = 4 * kt;
    const bool bias_now = do_bias && kt < nk1;
    // ---- phase 0: D0, X0 ; quadrant (0, 0)
    if (bias_now && wc == 0) bias_add(lds0 + (4 * PAR + 0) * 16384);
    read_x(I1{}, hi, b0);
    read_d(I0{}, hi);
    if (g0 + 6 < H) { issue(g0 + 6, (4 * PAR + 6) & 7); wait_vm<8>(); } else wait_vm<0>();
    raw_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : MDD_B8(b0), MDD_A16);
    quad(b0, 0, 0);
    raw_barrier();
    // ---- phase 1: X1 ; quadrant (0, 1)
    read_x(I2{}, hi, b1);
    if (g0 + 7 < H) { issue(g0 + 7, (4 * PAR + 7) & 7); wait_vm<8>(); } else wait_vm<0>();
    raw_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : MDD_B8(b1));
    quad(b1, 0, 2);
    raw_barrier();
    // ---- phase 2: D1 ; quadrant (1, 1)
    if (bias_now && wc == 1) bias_add(lds0 + (4 * PAR + 3) * 16384);
    read_d(I3{}, hi);
    if (g0 + 8 < H) { issue(g0 + 8, (4 * PAR + 8) & 7); wait_vm<10>(); } else wait_vm<0>();
    raw_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" : MDD_A16);
    quad(b1, 4, 2);
    raw_barrier();
    // ---- phase 3: nothing new ; quadrant (1, 0)
    if (g0 + 9 < H) { issue(g0 + 9, (4 * PAR + 9) & 7); wait_vm<8>(); } else wait_vm<0>();
    raw_barrier();
    quad(b0, 4, 0);
    raw_barrier();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(kt, I0{});
    if (kt + 1 < nk) ktile(kt + 1, I1{});
  }
#undef MDD_B8
#undef MDD_A16
  if (wc == 0) raw_barrier();
  __syncthreads();

  // ---- write-out: rows = k' (4 consecutive per lane), columns = co, as the one-stage kernel
  float* dst = pg.slab ? pg.slab + (size_t)bsplit * pg.slab_stride + q_.slab_off : q_.dW;
  const bool plain = pg.slab != nullptr || pg.direct != 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (MDD_WGP_ABL & 8) { asm volatile("" :: "v"(acc[i][j])); continue; }
      const int co = co0 + (i >> 2) * 128 + wc * 64 + 16 * (i & 3) + li, kcol = kp0 + (j >> 1) * 128 + wk * 32 + 16 * (j & 1) + 4 * gq;
      if (co >= q_.nc || kcol >= ktot) continue;
      float* a = dst + (size_t)co * ktot + kcol;
      if (plain) {
        *(float4*)a = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      } else {
        atomicAdd(a, acc[i][j][0]); atomicAdd(a + 1, acc[i][j][1]); atomicAdd(a + 2, acc[i][j][2]); atomicAdd(a + 3, acc[i][j][3]);
      }
    }
  if (do_bias) {
    float* sh = (float*)smem;     // [2 halves][16 rows][128 columns]
    const int cidx = bslot ^ ((brow & 7) << 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[(wc * 16 + brow) * 128 + cidx * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 256) {
      const int h = tid >> 7, col = tid & 127;
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += sh[(h * 16 + r) * 128 + col];
      const int ch = co0 + h * 128 + col;
      if (ch < q_.nc) atomicAdd(q_.dbias + ch, s);
    }
  }
}

// ======================================================================================================
// Grouped 3x3 stride-1 "same" convolutions with 64-channel groups (conv2 / conv2b of every NFNet block), bf16:
// ALL nine taps from ONE pass over the pixels.  The general kernel tiles k' = (tap, channel) into 128-wide blocks, so a
// group's 576 columns are 4.5 blocks that each re-read dy, and every tap is its own gather of x: 46 bytes through L1 per
// kMAC against 31 for a pointwise layer.  Here a block owns (group, pixel chunk): the 64-channel dy rows go to LDS as in
// the general kernel, the x rows [m - W - 1, m + 64 + W] live in a 256-row LDS RING that advances 64 pixels per step, and
// tap (ty, tx) of pixel m is ring row m + (ty-1) W + (tx-1) -- or a row of zeros where the tap leaves the image (per-pixel
// 9-bit mask, tracked incrementally): 16 KB of loads per step for 64 x 64 x 576 MACs = 7 bytes per kMAC.
// Wave w owns input channels [16w, 16w+16) of every tap x all 64 output channels: 36 accumulator blocks of 16 x 16.
__global__ __launch_bounds__(256, 2) void k_wgrad_slab(const WArgs p) {
  constexpr int RING = 256;
  __shared__ __attribute__((aligned(16))) char smem[64 * 128 + (RING + 1) * 128];
  char* const Ds = smem;
  char* const Xs = smem + 64 * 128;
  const ConvGeom& G = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int grp, bsplit;
  {
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.x + blockIdx.y * gridDim.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    bsplit = lin / gridDim.x;
    grp = lin - bsplit * gridDim.x;
  }
  const int W = G.wo, H = G.ho;
  const int mbeg = bsplit * p.mchunk;
  const int mend = min(p.M, mbeg + p.mchunk);
  const int rows = mend - mbeg;
  const int niter = (rows + 63) >> 6;
  constexpr unsigned OOB = 0x80000000u;
  auto rsrc = [&](const void* base, int64_t elem_off) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + elem_off * 2), (short)0, 0x7fffffff, 0x00020000);
  };
  // staging: thread -> 16-byte chunk column scol of rows srow, srow + 32
  const int scol = tid & 7, srow = tid >> 3;
  const int wsw_s = ((srow >> 1) & 3) << 1;      // rows 32 apart share it
  if (tid < 8) *(uint4*)(Xs + RING * 128 + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // fragment geometry: lane (gq, q, pp); its four pixel positions of a 64-pixel step
  const int gq = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  int lp[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) lp[ks][e] = ks * 32 + 4 * (2 * (2 * e + (gq >> 1)) + (gq & 1)) + q;
  const int dq64 = 64 / W, dr64 = 64 - dq64 * W;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][i][r] = 0.f;
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  auto run = [&](const void* dyp, const void* xp, const bool bias) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs_dy = rsrc(dyp, (int64_t)mbeg * G.co_tot + (int64_t)grp * 64);
    const __amdgpu_buffer_rsrc_t rs_x = rsrc(xp, (int64_t)grp * 64);
    // ---- prologue: x pixels [mbeg - W - 1, mbeg + 64 + W] -> ring rows 0 .., dy rows of step 0
    const int org = mbeg - W - 1;                 // pixel of ring row 0 (mod RING)
    const int npro = 66 + 2 * W;
    for (int j0 = 0; j0 < npro; j0 += 32) {
      const int j = j0 + srow, px = org + j;
      const bool ok = j < npro && px >= 0 && px < p.M;
      const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rs_x, (int)(ok ? (unsigned)((px * G.ca_tot) * 2 + scol * 16) : OOB), 0, 0));
      if (j < npro) *(u32x4*)(Xs + (j & (RING - 1)) * 128 + ((scol ^ ((((j & (RING - 1)) >> 1) & 3) << 1)) << 4)) = v;
    }
    u32x4 rd[2], rx[2];
    auto load_step = [&](int it) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = it * 64 + srow + 32 * i;
        rd[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs_dy, (int)(r < rows ? (unsigned)((r * G.co_tot) * 2 + scol * 16) : OOB), 0, 0));
      }
    };
    auto load_x = [&](int it) __attribute__((always_inline)) {     // the 64 new pixels step it needs beyond step it-1's
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int px = mbeg + it * 64 + W + 1 + srow + 32 * i;
        rx[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs_x, (int)(px < p.M ? (unsigned)((px * G.ca_tot) * 2 + scol * 16) : OOB), 0, 0));
      }
    };
    auto store_dy = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *(u32x4*)(Ds + (srow + 32 * i) * 128 + ((scol ^ wsw_s) << 4)) = rd[i];
        if (bias) {
          float f[8];
          Chunk<bf16>::unpack(__builtin_bit_cast(uint4, rd[i]), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[e] += f[e];
        }
      }
    };
    auto store_x = [&](int it) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rr = (it * 64 + 2 * W + 2 + srow + 32 * i) & (RING - 1);       // = pixel - org
        *(u32x4*)(Xs + rr * 128 + ((scol ^ (((rr >> 1) & 3) << 1)) << 4)) = rx[i];
      }
    };
    // running (oy, ox) of this lane's four pixel positions
    int oy[2][2], ox[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int m = mbeg + lp[ks][e];
        ox[ks][e] = m % W;
        oy[ks][e] = (m / W) % H;
      }
    if (niter > 0) { load_step(0); store_dy(); }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
      if (it + 1 < niter) { load_step(it + 1); load_x(it + 1); }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        s16x8 a8[4];
        unsigned tm[2];
        int rb[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int row = lp[ks][e];
          const unsigned ym = (oy[ks][e] > 0 ? 1u : 0u) | 2u | (oy[ks][e] < H - 1 ? 4u : 0u);
          const unsigned xm = (ox[ks][e] > 0 ? 1u : 0u) | 2u | (ox[ks][e] < W - 1 ? 4u : 0u);
          tm[e] = ((ym & 1u) ? xm : 0u) | ((ym & 2u) ? xm << 3 : 0u) | ((ym & 4u) ? xm << 6 : 0u);
          rb[e] = it * 64 + W + 1 + row;          // ring row of the centre tap (before the mask)
        }
        s16x4 af[4][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int row = lp[ks][e];
          const int sw = ((row >> 1) & 3) << 2;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            af[i][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Ds + row * 128 + (((4 * i + pp) ^ sw) << 3)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a8[i] = __builtin_shufflevector(af[i][0], af[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          s16x4 bf[3][2];
#pragma unroll
          for (int tx = 0; tx < 3; ++tx)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int t = ty * 3 + tx;
              const int rr = (rb[e] + (ty - 1) * W + (tx - 1)) & (RING - 1);
              const int off = ((tm[e] >> t) & 1u) ? rr * 128 + (((4 * wave + pp) ^ (((rr >> 1) & 3) << 2)) << 3)
                                                  : RING * 128 + pp * 8;
              bf[tx][e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(Xs + off));
            }
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const s16x8 b8 = __builtin_shufflevector(bf[tx][0], bf[tx][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              acc[ty * 3 + tx][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b8), __builtin_bit_cast(bf16x8, a8[i]),
                                                                            acc[ty * 3 + tx][i], 0, 0, 0);
          }
        }
      }
      // advance the pixel positions by 64
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          ox[ks][e] += dr64; oy[ks][e] += dq64;
          if (ox[ks][e] >= W) { ox[ks][e] -= W; ++oy[ks][e]; }
          while (oy[ks][e] >= H) oy[ks][e] -= H;
        }
      __syncthreads();                   // every wave is done with the dy tile (and with ring rows older than this step's)
      if (it + 1 < niter) { store_dy(); store_x(it + 1); }
      __syncthreads();
    }
  };
  run(p.dy1, p.x1, p.dbias != nullptr);
  if (p.dy2) run(p.dy2, p.x2, false);

  // ---- write-out: rows = k' (4 consecutive per lane), columns = co
  const int ktot = 576;
  const size_t gbase = (size_t)grp * 64 * ktot;
  float* dst = p.slab ? p.slab + (size_t)bsplit * p.slab_stride + gbase : p.dW + gbase;
  const bool plain = p.slab != nullptr;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* a = dst + (size_t)(16 * i + li) * ktot + t * 64 + 16 * wave + 4 * gq;
      if (plain) {
        *(float4*)a = make_float4(acc[t][i][0], acc[t][i][1], acc[t][i][2], acc[t][i][3]);
      } else {
        atomicAdd(a, acc[t][i][0]); atomicAdd(a + 1, acc[t][i][1]); atomicAdd(a + 2, acc[t][i][2]); atomicAdd(a + 3, acc[t][i][3]);
      }
    }
  if (p.dbias) {
    __syncthreads();
    float* sh = (float*)smem;      // [32 rows][8 chunk columns][8]
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[(srow * 8 + scol) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 64) {
      float s_ = 0.f;
      for (int r = 0; r < 32; ++r) s_ += sh[(r * 8 + (tid >> 3)) * 8 + (tid & 7)];
      atomicAdd(p.dbias + grp * 64 + tid, s_);
    }
  }
}

bool launch_slab(WArgs a, float* slab, int64_t slab_floats, hipEvent_t ev_mid, hipStream_t st) {
  const ConvGeom& g = a.g;
  if (!MDD_WG_SLAB || g.k != 3 || g.stride != 1 || g.pad != 1 || g.kc != 64 || g.nc != 64 || g.ha != g.ho || g.wa != g.wo ||
      g.wo > 90 || g.wo < 3 || (g.ca_tot & 7) || (g.co_tot & 7) || (int64_t)a.M * std::max(g.ca_tot, g.co_tot) * 2 >= (1ll << 31))
    return false;
  const int nsrc = a.dy2 ? 2 : 1;
  const int64_t out_floats = (int64_t)g.groups * 64 * 576;
  const double out_mb = (double)out_floats * 4.0 / 1e6;
  const bool two_phase = slab != nullptr && slab_floats >= out_floats && ((uintptr_t)a.dW & 15) == 0;
  const double comb_us_per_mb = two_phase ? 0.6 : 1.0 / 1.3;
  int maxsplits = std::max(1, a.M / 256);     // at least four steps per block
  if (two_phase && maxsplits > slab_floats / out_floats) maxsplits = (int)(slab_floats / out_floats);
  // step cost of the model from a sweep on MI355X (tools/micro/wgrad_slab_check.py): 1.2 us suits one operand pair, a
  // tangent launch (two pairs) wants about twice the splits that figure gives
  int splits = 1;
  double best = 1e30;
  for (int sp = 1; sp <= maxsplits; sp = sp < 32 ? sp + 1 : sp + sp / 16) {
    const int chunk = ((a.M + sp - 1) / sp + 63) / 64 * 64;
    const double steps = (double)nsrc * chunk / 64;
    const double waves = (double)(((int64_t)g.groups * sp + 511) / 512);     // two blocks per CU
    const double t = waves * (steps * (nsrc == 2 ? 5.0 : 1.2) + 6.0 * nsrc) + out_mb * sp * comb_us_per_mb;
    if (t < best) { best = t; splits = sp; }
  }
  int mchunk = ((a.M + splits - 1) / splits + 63) / 64 * 64;
  splits = (a.M + mchunk - 1) / mchunk;
  a.mchunk = mchunk;
  a.dbg = 0;
  a.cotiles = a.kptiles = 1;
  a.slab = two_phase ? slab : nullptr;
  a.slab_stride = out_floats;
  k_wgrad_slab<<<dim3(g.groups, splits), 256, 0, st>>>(a);
  if (ev_mid) (void)hipEventRecord(ev_mid, st);
  if (two_phase) {
    const int64_t n4 = out_floats / 4;
    k_wgrad_reduce<<<(unsigned)((n4 + 15) / 16), 256, 0, st>>>(a.dW, slab, n4, out_floats / 4, splits);
  }
  return true;
}

// sum of the split slabs of up to four layers in one launch: segment i covers float4 columns [start4[i], start4[i+1])
struct WRed { float* dW[4]; int64_t start4[5]; int n; };
__global__ __launch_bounds__(256) void k_wgrad_reduce_group(const WRed r, const float* __restrict__ slab, int64_t stride4,
                                                              int splits) {
  __shared__ float4 part[16][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + col;
  const int64_t n4 = r.start4[r.n];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* src = (const float4*)slab + i;
    for (int k = sl; k < splits; k += 16) {
      const float4 v = src[(int64_t)k * stride4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[sl][col] = s;
  __syncthreads();
  if (sl == 0 && i < n4) {
    float4 t = part[0][col];
#pragma unroll
    for (int l = 1; l < 16; ++l) { float4 v = part[l][col]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    int seg = 0;
    while (seg + 1 < r.n && i >= r.start4[seg + 1]) ++seg;
    ((float4*)r.dW[seg])[i - r.start4[seg]] = t;
  }
}

bool pipe_wgrad_takes(const ConvGeom& g, int M) {
  const bool pw = g.k == 1 && g.stride == 1 && g.pad == 0 && g.groups == 1;
  return pipe_kernels_enabled() && pw && g.prec == 0 && g.nc >= MDD_WG_PIPE_MIN && g.kc >= MDD_WG_PIPE_MIN && !(g.kc & 7) && !(g.nc & 7) &&
         !(g.co_tot & 7) && !(g.ca_tot & 7) && M >= 8192;
}

// split policy and launch of k_wgrad_pipe for one to four layers that share the pixel count and the pairing (all with or
// all without a second operand pair); false = not taken (the caller falls through to the one-stage kernel, one layer at
// a time)
struct WItem { ConvGeom g; const void *dy1, *x1, *dy2, *x2; float* dW; float* dbias; };
bool launch_pipe_group(const WItem* it, int n, int M, float* slab, int64_t slab_floats, hipEvent_t ev_mid, hipStream_t st) {
  if (n < 1 || n > 4) return false;
  WGroup G;
  memset(&G, 0, sizeof G);
  int tiles = 0;
  int64_t out_floats = 0;
  bool aligned = true;
  for (int i = 0; i < n; ++i) {
    const ConvGeom& g = it[i].g;
    if (!pipe_wgrad_takes(g, M) || (it[i].dy2 != nullptr) != (it[0].dy2 != nullptr)) return false;
    WProb& q = G.pr[i];
    q.dy1 = it[i].dy1; q.x1 = it[i].x1; q.dy2 = it[i].dy2; q.x2 = it[i].x2; q.dW = it[i].dW; q.dbias = it[i].dbias;
    q.nc = g.nc; q.kc = g.kc; q.co_tot = g.co_tot; q.ca_tot = g.ca_tot;
    q.kptiles = (g.kc + 255) / 256;
    q.tile_start = tiles;
    q.slab_off = out_floats;
    tiles += ((g.nc + 255) / 256) * q.kptiles;
    const int64_t of = (int64_t)g.nc * g.kc;
    aligned = aligned && of % 4 == 0 && ((uintptr_t)it[i].dW & 15) == 0;
    out_floats += of;
  }
  const int nsrc = it[0].dy2 ? 2 : 1;
  const double out_mb = (double)out_floats * 4.0 / 1e6;
  bool two_phase = slab != nullptr && slab_floats >= out_floats && aligned;
  const double comb_us_per_mb = two_phase ? 0.6 : 1.0 / 1.3;
  int maxsplits = M / 256;     // at least four K-tiles per block
  if (two_phase && maxsplits > slab_floats / out_floats) maxsplits = (int)(slab_floats / out_floats);
  if (maxsplits < 1) maxsplits = 1;
  int splits = 1;
  double best = 1e30;
  for (int sp = 1; sp <= maxsplits; ++sp) {
    const int chunk = ((M + sp - 1) / sp + 63) / 64 * 64;
    const double steps = (double)nsrc * chunk / 64;
    const double waves = (double)(((int64_t)tiles * sp + 255) / 256);      // one block per CU
    const double t = waves * (steps * 0.9 + 8.0) + out_mb * sp * comb_us_per_mb;
    if (t < best) { best = t; splits = sp; }
  }
  if (MDD_WGP_GROUP_SPLITS > 0 && n > 1 && tiles * MDD_WGP_GROUP_SPLITS >= 64) splits = MDD_WGP_GROUP_SPLITS;   // small groups: the model
  int mchunk = ((M + splits - 1) / splits + 63) / 64 * 64;
  splits = (M + mchunk - 1) / mchunk;
  G.n = n; G.M = M; G.mchunk = mchunk; G.tiles = tiles;
  G.direct = (splits == 1 && aligned) ? 1 : 0;
  if (G.direct) two_phase = false;
  G.slab = two_phase ? slab : nullptr;
  G.slab_stride = out_floats;
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute((const void*)k_wgrad_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, WGP_NS * 16384) != hipSuccess) return false;
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  k_wgrad_pipe<<<dim3(tiles, splits), 512, WGP_NS * 16384, st>>>(G);
  if (ev_mid) (void)hipEventRecord(ev_mid, st);
  if (two_phase) {
    WRed r;
    memset(&r, 0, sizeof r);
    r.n = n;
    for (int i = 0; i < n; ++i) { r.dW[i] = it[i].dW; r.start4[i] = G.pr[i].slab_off / 4; }
    r.start4[n] = out_floats / 4;
    k_wgrad_reduce_group<<<(unsigned)((out_floats / 4 + 15) / 16), 256, 0, st>>>(r, slab, out_floats / 4, splits);
  }
  return true;
}
bool launch_pipe(WArgs a, float* slab, int64_t slab_floats, hipEvent_t ev_mid, hipStream_t st) {
  WItem it;
  it.g = a.g; it.dy1 = a.dy1; it.x1 = a.x1; it.dy2 = a.dy2; it.x2 = a.x2; it.dW = a.dW; it.dbias = a.dbias;
  return launch_pipe_group(&it, 1, a.M, slab, slab_floats, ev_mid, st);
}

template <class AT, int BCO, int BKP, int BKM>
void launch_cfg(WArgs a, float* slab, int64_t slab_floats, hipEvent_t ev_mid, hipStream_t st) {
  const bool pw = a.g.k == 1 && a.g.stride == 1 && a.g.pad == 0;
  const ConvGeom& g = a.g;
  int ktot = g.k * g.k * g.kc;
  a.cotiles = (g.nc + BCO - 1) / BCO;
  a.kptiles = (ktot + BKP - 1) / BKP;
  int tiles = a.cotiles * a.kptiles * g.groups;
  // Split-M policy from a two-term cost model measured on MI355X (tools/bench_wgrad.py): a block's
  // K-loop is a serial chain of ~1.4 us steps (latency-bound), and every split adds one fp32-atomic
  // pass over the output tile (chip-wide ~1.3 TB/s).  T(S) = waves(S) * (steps(S) * 1.4 us + 5 us)
  //                                                    + out_bytes * S / 1.3 TB/s + 0.05 us * S.
  const int nsrc = a.dy2 ? 2 : 1;
  const int64_t out_floats = (int64_t)g.groups * g.nc * ktot;
  const double out_mb = (double)out_floats * 4.0 / 1e6;
  // the split-M reduce kernel moves float4 columns into dW: it needs a whole number of them and a 16-byte aligned
  // destination, otherwise the combine falls back to fp32 atomics (the caller then provides a zeroed dW: the
  // single-op entry point and the ViT linears, whose dW is a slice of the zeroed flat gradient)
  const bool two_phase = slab != nullptr && slab_floats >= out_floats && out_floats % 4 == 0 &&
                         ((uintptr_t)a.dW & 15) == 0;
  // cost of combining one more split: atomics ~1.3 TB/s of added bytes; slabs: one streamed write + one
  // streamed read of the partial tile (~5 TB/s each)
  const double comb_us_per_mb = two_phase ? 0.6 : 1.0 / 1.3;
  const int slots = 256 * 3;   // resident blocks on the chip at ~3 blocks per CU
  const double tstep = MDD_WG_TSTEP_SCALE * ((g.k == 1 && g.stride == 1) ? 0.8 : 1.4);   // us per K-step (pointwise / gathered)
  int maxsplits = (a.M + 2 * BKM - 1) / (2 * BKM);
  if (two_phase && maxsplits > slab_floats / out_floats) maxsplits = (int)(slab_floats / out_floats);
  if (maxsplits < 1) maxsplits = 1;
  int splits = 1;
  double best = 1e30;
  for (int sp = 1; sp <= maxsplits; sp = sp < 16 ? sp + 1 : sp + sp / 8) {
    int chunk = ((a.M + sp - 1) / sp + BKM - 1) / BKM * BKM;
    double steps = (double)nsrc * chunk / BKM;
    double waves = (double)((int64_t)tiles * sp + slots - 1) / slots;
    if (waves < 1.0) waves = 1.0;
    double t = waves * (steps * tstep + 5.0) + out_mb * sp * comb_us_per_mb + 0.05 * sp;   // + contention per split
    if (t < best) { best = t; splits = sp; }
  }
  int mchunk = (a.M + splits - 1) / splits;
  mchunk = ((mchunk + BKM - 1) / BKM) * BKM;
  splits = (a.M + mchunk - 1) / mchunk;
  a.mchunk = mchunk;
#ifdef MDD_DEBUG_SWITCHES
  static const int dbg = [] { const char* e = getenv("MDD_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
#else
  a.dbg = 0;
#endif
  a.slab = two_phase ? slab : nullptr;
  a.slab_stride = out_floats;
  dim3 grid(tiles, splits);
  auto finish = [&]() {
    if (ev_mid) (void)hipEventRecord(ev_mid, st);    // profiling: end of the contraction kernel proper
    if (!two_phase) return;
    int64_t n4 = out_floats / 4;
    k_wgrad_reduce<<<(unsigned)((n4 + 15) / 16), 256, 0, st>>>(a.dW, slab, n4, out_floats / 4, splits);
  };
  constexpr bool PF2 = MDD_WG_PF2 && sizeof(AT) == 2 && BCO == 128;   // 2 waves/SIMD either way there
  if constexpr (sizeof(AT) == 4) {
    if (g.prec == 3) {
      if (pw) k_conv_wgrad<AT, BCO, BKP, BKM, true, false, true, true><<<grid, 256, 0, st>>>(a);
      else k_conv_wgrad<AT, BCO, BKP, BKM, false, false, true, true><<<grid, 256, 0, st>>>(a);
      finish();
      return;
    }
    if (g.prec != 0) {
      if (pw) k_conv_wgrad<AT, BCO, BKP, BKM, true, false, true><<<grid, 256, 0, st>>>(a);
      else k_conv_wgrad<AT, BCO, BKP, BKM, false, false, true><<<grid, 256, 0, st>>>(a);
      finish();
      return;
    }
  }
  if (pw) k_conv_wgrad<AT, BCO, BKP, BKM, true, PF2, false><<<grid, 256, 0, st>>>(a);
  else k_conv_wgrad<AT, BCO, BKP, BKM, false, false, false><<<grid, 256, 0, st>>>(a);
  finish();
}

}  // namespace

template <class AT>
void launch_conv_wgrad(const ConvGeom& g, const AT* dy1, const AT* x1, const AT* dy2, const AT* x2,
                       float* dW, float* dbias, float* slab, int64_t slab_floats, hipEvent_t ev_mid,
                       hipStream_t st) {
  WArgs a;
  a.dy1 = dy1; a.x1 = x1; a.dy2 = dy2; a.x2 = x2; a.dW = dW; a.dbias = dbias;
  a.g = g;
  a.M = g.nimg * g.ho * g.wo;
  a.cotiles = a.kptiles = a.mchunk = 0;
  if constexpr (sizeof(AT) == 2) {
    if (launch_pipe(a, slab, slab_floats, ev_mid, st)) return;
    if (launch_slab(a, slab, slab_floats, ev_mid, st)) return;
    if (g.nc > 64) launch_cfg<AT, 128, 128, MDD_WG_BKM>(a, slab, slab_floats, ev_mid, st);
    else launch_cfg<AT, 64, 128, MDD_WG_BKM>(a, slab, slab_floats, ev_mid, st);
  } else {
    launch_cfg<AT, 64, 128, 32>(a, slab, slab_floats, ev_mid, st);
  }
}
bool conv_wgrad_group_takes(const ConvGeom& g) { return pipe_wgrad_takes(g, g.nimg * g.ho * g.wo); }
// several wide pointwise bf16 layers with the same pixel count in ONE launch (a ViT layer's four linears)
bool launch_conv_wgrad_group(const WgradItem* items, int n, float* slab, int64_t slab_floats, hipStream_t st) {
  if (n < 1 || n > 4) return false;
  WItem it[4];
  const int M = items[0].g.nimg * items[0].g.ho * items[0].g.wo;
  for (int i = 0; i < n; ++i) {
    if (items[i].g.nimg * items[i].g.ho * items[i].g.wo != M) return false;
    it[i].g = items[i].g; it[i].dy1 = items[i].dy1; it[i].x1 = items[i].x1; it[i].dy2 = items[i].dy2; it[i].x2 = items[i].x2;
    it[i].dW = items[i].dW; it[i].dbias = items[i].dbias;
  }
  return launch_pipe_group(it, n, M, slab, slab_floats, nullptr, st);
}
template void launch_conv_wgrad<float>(const ConvGeom&, const float*, const float*, const float*,
                                       const float*, float*, float*, float*, int64_t, hipEvent_t, hipStream_t);
template void launch_conv_wgrad<bf16>(const ConvGeom&, const bf16*, const bf16*, const bf16*,
                                      const bf16*, float*, float*, float*, int64_t, hipEvent_t, hipStream_t);
