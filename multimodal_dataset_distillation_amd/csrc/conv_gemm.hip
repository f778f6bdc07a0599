// Implicit-GEMM convolution for gfx950 (CDNA4) MFMA -- forward and data-gradient in ONE kernel.
//
//   out[m, g*nc + n] = sum_{tap, kc}  A[src(m, tap), g*kc_ + kc] * B[g][n][tap][kc]
//
//   m   = output pixel (image, oy, ox) on NHWC activations (channels contiguous)
//   src = forward gather (oy*s - p + ty, ...) or transposed/dgrad gather ((oy + p - ty)/s, ...)
//   B   = packed standardised weights from ws.hip: wf [cout][k*k][cin_g] for forward,
//         wt [g][cin_g][k*k][cout_g] for dgrad -- K contiguous per output channel in both.
//   optional second source pair (A2,B2) accumulates into the same MFMA accumulators: that is how
//   the tangent pass computes  conv(a_dot, w_hat) + conv(a, w_hat_dot)  in one launch.
//
// Tiling: 256 threads = 4 waves (64 lanes).  Block tile BM x BN, K-step = one 128-byte row of K
// (64 bf16 / 32 f32), ONE LDS stage: the next tile's global loads are issued into staging registers
// before the MFMA phase and written to LDS after it (the gather needs zero-fill at image borders, and
// 35 KB of LDS per block lets three blocks share a CU -- occupancy beats a second LDS stage here).  LDS rows are 128 B =
// 8 chunks of 16 B; chunk position is XOR-swizzled with (row>>1)&7 so that every 16-lane group
// of a ds_read_b128 hits 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
// bf16: v_mfma_f32_32x32x16_bf16 ; f32 (parity mode): v_mfma_f32_32x32x2_f32 (exact fp32 FMA).
// Fused epilogues (ConvEpi): bias, SiLU*beta, SiLU' / SiLU'' chain rule, residual adds.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

#ifndef MDD_EPI_UNROLL
#define MDD_EPI_UNROLL 4
#endif
#ifndef MDD_BIG_TILE
#define MDD_BIG_TILE 0
#endif
#ifndef MDD_SINGLE_BUF
#define MDD_SINGLE_BUF 1
#endif
#ifndef MDD_FRAG_PREFETCH
#define MDD_FRAG_PREFETCH 1
#endif
#ifndef MDD_MIN_WAVES
#define MDD_MIN_WAVES 3
#endif

#ifndef MDD_PIPE_EPI_HOIST
#define MDD_PIPE_EPI_HOIST 1  // k_gemm_pipe: first group of stash loads of a 32-row block issued before its LDS transposition
#endif
#ifndef MDD_EPI_WAVE_SYNC
#define MDD_EPI_WAVE_SYNC 0   // k_conv_gemm: wave-level ordering instead of block barriers between the 32-row blocks of the epilogue
#endif
#ifndef MDD_SPLIT_3TERM
#define MDD_SPLIT_3TERM 1   // bf16x2 K loop of the 2x2-wave instances: hh + hl + lh on register-regrouped chunk pairs
                            // (0: hh + ll + hl + lh everywhere)
#endif

namespace {

typedef unsigned __attribute__((ext_vector_type(4))) u32x4;   // first-class 16-byte register value

template <class AT> struct Mma;
template <> struct Mma<bf16> {
  static constexpr int KE = 64;  // K elements per 128-byte row
  static constexpr int CE = 8;   // elements per 16-byte chunk
  static DEVI u32x4 stage(const u32x4& raw, bool) { return raw; }
  static DEVI void step(const u32x4& a, const u32x4& b, f32x16& acc) {
    bf16x8 av = __builtin_bit_cast(bf16x8, a), bv = __builtin_bit_cast(bf16x8, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KE = 32;
  static constexpr int CE = 4;
  static DEVI u32x4 stage(const u32x4& raw, bool) { return raw; }
  static DEVI void step(const u32x4& a, const u32x4& b, f32x16& acc) {
    // lane-half h holds k = 4*(2q+h)+j, j=0..3 for BOTH operands: any k-permutation that is the
    // same for A and B leaves the sum unchanged.
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[0]), __uint_as_float(b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[1]), __uint_as_float(b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[2]), __uint_as_float(b[2]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[3]), __uint_as_float(b[3]), acc, 0, 0, 0);
  }
};

// Split-bf16 ("bf16x2") math on fp32 storage: every fp32 operand x is carried as hi = bf16(x) and
// lo = bf16(x - hi) (16 significand bits together, error ~2^-17 per operand instead of 2^-9), and the
// product is formed by bf16 MFMAs with fp32 accumulation:  x*y ~= xh*yh + xh*yl + xl*yh + xl*yl.
// The split happens ONCE per element, between the global load and the LDS store; a 16-byte LDS chunk
// holds [h0 h1 h2 h3 | l0 l1 l2 l3] of four consecutive K positions, so one v_mfma_f32_32x32x16_bf16 on
// the chunks as they stand yields hh + ll and a second one with the halves of B swapped yields hl + lh:
// two MFMAs per four K positions (the exact-fp32 v_mfma_f32_32x32x2_f32 path needs four at twice the
// cycles each) and no operand arithmetic inside the K loop.
// HI: the lo halves are zero (one-bf16-per-operand passes of a mixed mode, ConvGeom::prec == 2): hh is the product
template <bool HI>
struct MmaSplitT {
  static constexpr int KE = 32;
  static constexpr int CE = 4;
  static DEVI u32x4 stage(const u32x4& raw, bool hi_only) {
    float f[4] = {__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3])};
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16 hb = (bf16)f[i];                                   // v_cvt_pk_bf16_f32: RNE, NaN-preserving
      h[i] = (unsigned)__builtin_bit_cast(unsigned short, hb);
      float r = f[i] - __uint_as_float(h[i] << 16);           // exact in fp32
      bf16 lb = (bf16)r;
      l[i] = hi_only ? 0u : (unsigned)__builtin_bit_cast(unsigned short, lb);
    }
    u32x4 o;
    o[0] = h[0] | (h[1] << 16); o[1] = h[2] | (h[3] << 16);
    o[2] = l[0] | (l[1] << 16); o[3] = l[2] | (l[3] << 16);
    return o;
  }
  static DEVI void step(const u32x4& a, const u32x4& b, f32x16& acc) {
    u32x4 bs;
    bs[0] = b[2]; bs[1] = b[3]; bs[2] = b[0]; bs[3] = b[1];
    bf16x8 av = __builtin_bit_cast(bf16x8, a);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    if constexpr (!HI) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, bs), acc, 0, 0, 0);
  }
  // Two chunks of each operand at once (eight K positions per lane): regrouping their registers as
  // [h0..h7] and [l0..l7] costs no instruction and lets the ll term (below 2^-16 relative) be dropped:
  // hh + hl + lh = three MFMAs per eight K positions instead of four.
  static DEVI void step2(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, f32x16& acc) {
    u32x4 ah, al, bh, bl;
    ah[0] = a0[0]; ah[1] = a0[1]; ah[2] = a1[0]; ah[3] = a1[1];
    al[0] = a0[2]; al[1] = a0[3]; al[2] = a1[2]; al[3] = a1[3];
    bh[0] = b0[0]; bh[1] = b0[1]; bh[2] = b1[0]; bh[3] = b1[1];
    bl[0] = b0[2]; bl[1] = b0[3]; bl[2] = b1[2]; bl[3] = b1[3];
    const bf16x8 ahv = __builtin_bit_cast(bf16x8, ah), bhv = __builtin_bit_cast(bf16x8, bh);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahv, bhv, acc, 0, 0, 0);
    if constexpr (!HI) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahv, __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), bhv, acc, 0, 0, 0);
    }
  }
};
// fp32 storage, ONE fp16 per operand (ConvGeom::prec == 3): the precision experiment of DESIGN.md section 5 -- what an
// fp16-storage engine's matrix cores would see (11 significand bits instead of bf16's 8, narrower exponent range)
struct MmaF16Ops {
  static constexpr int KE = 32;
  static constexpr int CE = 4;
  static DEVI u32x4 stage(const u32x4& raw, bool) {
    unsigned h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      _Float16 hh = (_Float16)__uint_as_float(raw[i]);
      h[i] = (unsigned)__builtin_bit_cast(unsigned short, hh);
    }
    u32x4 o;
    o[0] = h[0] | (h[1] << 16); o[1] = h[2] | (h[3] << 16); o[2] = 0u; o[3] = 0u;
    return o;
  }
  static DEVI void step(const u32x4& a, const u32x4& b, f32x16& acc) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  }
  static DEVI void step2(const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, f32x16& acc) {
    u32x4 ah, bh;
    ah[0] = a0[0]; ah[1] = a0[1]; ah[2] = a1[0]; ah[3] = a1[1];
    bh[0] = b0[0]; bh[1] = b0[1]; bh[2] = b1[0]; bh[3] = b1[1];
    step(ah, bh, acc);
  }
};
template <class AT, int PREC> struct MmaSel { typedef Mma<AT> type; };
template <> struct MmaSel<float, 3> { typedef MmaF16Ops type; };
template <> struct MmaSel<float, 1> { typedef MmaSplitT<false> type; };
template <> struct MmaSel<float, 2> { typedef MmaSplitT<true> type; };

__device__ __attribute__((aligned(16))) const unsigned g_zero_block[64] = {0};

DEVI uint4 mask4(uint4 v, bool keep) {
  unsigned m = keep ? 0xffffffffu : 0u;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}
DEVI int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

constexpr size_t MODE4_MAX_LDS = 80 * 1024;   // slab mode only when two blocks still share a CU

struct KArgs {
  const void* A1; const void* B1; const void* A2; const void* B2;
  ConvGeom g;
  ConvEpi ep;
  int M, mtiles, ntiles, dbg;
};

// orders one wave's own LDS accesses (its private epilogue staging): LDS executes a wave's instructions in order, so all
// that is needed is that the compiler neither reorders them nor lets a read's result be used before it has returned
DEVI void epi_wave_sync() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// activation of the fused epilogues: 0 = SiLU (NFNet), 1 = exact GELU (the ViT MLP: fc1's forward and fc2's data gradient)
// (2 = the same GELU through the 1.5e-7 erf approximation: bf16 storage only, k_gemm_pipe)
template <int ACT> DEVI float actf(float x) {
  if constexpr (ACT == 2) return gelu_fast_(x); else if constexpr (ACT == 1) return gelu_(x); else return silu_(x);
}
template <int ACT> DEVI float dactf(float x) {
  if constexpr (ACT == 2) return dgelu_fast_(x); else if constexpr (ACT == 1) return dgelu_(x); else return dsilu_(x);
}
template <int ACT> DEVI Dual dactf(Dual x) {
  if constexpr (ACT == 2) return dgelu_fast_(x); else if constexpr (ACT == 1) return dgelu_(x); else return dsilu_(x);
}

template <class AT, int WGM, int WGN, int TM, int TN, int MODE, bool KFULL, int PREC, bool IBK = false, int ACT = 0>
__global__ __launch_bounds__(256, MDD_MIN_WAVES) void k_conv_gemm(const KArgs p) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  typedef typename MmaSel<AT, PREC>::type MT;
  constexpr int KE = MT::KE, CE = MT::CE;
  constexpr bool EPI_WAVE_SYNC = MDD_EPI_WAVE_SYNC != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NBUF = MDD_SINGLE_BUF ? 1 : 2;   // LDS stages (registers hold the slab in flight)

  const ConvGeom& G = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave - wm * WGN;

  // block -> (m tile, group, n tile); n tile fastest so neighbours share the A panel in L2
  // XCD-aware remap (8 XCDs, private L2s, blocks dealt round-robin): give every XCD a CONTIGUOUS
  // chunk of logical ids so the n-tiles / groups of one m-tile share the A panel in ONE L2 instead
  // of re-fetching it through the fabric once per XCD.  Bijective for any grid size.
  int bid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int nt = bid % p.ntiles; bid /= p.ntiles;
  const int grp = bid % G.groups;
  const int mt = bid / G.groups;
  // MODE 2 (stride-2 data gradient, even output dims): the output pixels are split into the four
  // (oy&1, ox&1) parity classes.  Within a class only taps ty = ty0 + 2a, tx = tx0 + 2b hit an input
  // pixel, at (oy/2 + cy - a, ox/2 + cx - b): a dense stride-1 problem on the half-resolution grid
  // with 1, 2, 2 or 4 taps instead of 9 taps of which 3/4 multiply zeros.
  int e_M = p.M, e_ho = G.ho, e_wo = G.wo, e_kh = G.k, e_kw = G.k;
  int ty0 = 0, tx0 = 0, cy = 0, cx = 0, py = 0, px = 0, mt_ = mt;
  if constexpr (MODE == 2) {
    e_ho = G.ho >> 1; e_wo = G.wo >> 1; e_M = G.nimg * e_ho * e_wo;
    const int mtc = (e_M + BM - 1) / BM;
    const int cls = mt / mtc;
    mt_ = mt - cls * mtc;
    py = cls >> 1; px = cls & 1;
    ty0 = (py + G.pad) & 1; tx0 = (px + G.pad) & 1;
    e_kh = (G.k - ty0 + 1) >> 1; e_kw = (G.k - tx0 + 1) >> 1;
    cy = (py + G.pad - ty0) >> 1; cx = (px + G.pad - tx0) >> 1;
  }
  const int m0 = mt_ * BM, n0 = nt * BN;

  const int ktot_w = G.k * G.k * G.kc;          // K length of one packed weight row
  const int ktot = e_kh * e_kw * G.kc;          // K extent this block reduces over
  constexpr int TS = MODE == 4 ? 3 : 1;          // taps per K-step (MODE 4 stages one ROW of taps of B at a time)
  const int nk1 = (ktot + KE - 1) / KE / TS;
  const int nk = (MDD_DBG_BITS(p) & 1) ? 1 : (p.A2 ? 2 * nk1 : nk1);   // dbg bit0: single K-step (timing only)

  // ---- per-thread staging geometry, hoisted out of the K loop (the loop is instruction-issue
  // bound, not MFMA- or HBM-bound: every VALU op per load counts).
  //   aoff[i] : byte offset of row i's source pixel at tap (0,0) (+ this thread's channel chunk for
  //             pointwise convs); tapm[i]: bit t set <=> tap t of row i is inside the image.
  //   Invalid taps / K tail / rows past M read a 16-byte zero block instead of being masked later.
  constexpr int ESZ = (int)sizeof(AT);
  const int cj = tid & 7, r0 = tid >> 3;
  const char* const Z = (const char*)g_zero_block;
  int aoff[RA];
  unsigned tapm[RA];
  int s2y[MODE == 3 ? RA : 1], s2x[MODE == 3 ? RA : 1], s2b[MODE == 3 ? RA : 1];
  // rows of this thread are m0 + r0 + 32*i: decode the first one with two divisions, advance the
  // others incrementally (the prologue is paid by every block: thousands of blocks per launch)
  int d_ox = 0, d_oy = 0, d_ni = 0;
  if constexpr (MODE != 0) {
    int m = m0 + r0;
    d_ox = m % e_wo;
    int t = m / e_wo;
    d_oy = t % e_ho;
    d_ni = t / e_ho;
  }
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + r0 + 32 * i;
    if constexpr (MODE == 0) {
      int mm = m < e_M ? m : e_M - 1;          // clamp: rows >= M are computed but never stored
      aoff[i] = (mm * G.ca_tot + cj * CE) * ESZ;
      tapm[i] = 1u;
    } else if constexpr (MODE == 4) {
      aoff[i] = 0; tapm[i] = 0u;               // MODE 4 stages no A rows through registers (slab below)
    } else {
      aoff[i] = 0; tapm[i] = 0u;
      if constexpr (MODE == 3) { s2y[i] = -1; s2x[i] = -1; s2b[i] = 0; }
      if (m < e_M) {
        const int ox = d_ox, oy = d_oy, ni = d_ni;
        int y0, x0;
        if constexpr (MODE == 2) { y0 = oy + cy; x0 = ox + cx; }
        else if (G.transposed) { y0 = oy + G.pad; x0 = ox + G.pad; }
        else { y0 = oy * G.stride - G.pad; x0 = ox * G.stride - G.pad; }
        if constexpr (MODE == 3) {
          s2y[i] = y0; s2x[i] = x0; s2b[i] = ni * G.ha * G.wa;
          tapm[i] = 1u;   // validity evaluated per tap in the loop
        } else {
          aoff[i] = ((ni * G.ha + y0) * G.wa + x0) * G.ca_tot * ESZ;   // may be "negative": masked taps never use it
          // valid taps form a rectangle [ylo,yhi] x [xlo,xhi]: build the bit mask from two ranges
          int ylo, yhi, xlo, xhi;
          if (G.transposed) { ylo = y0 - (G.ha - 1); yhi = y0; xlo = x0 - (G.wa - 1); xhi = x0; }
          else { ylo = -y0; yhi = G.ha - 1 - y0; xlo = -x0; xhi = G.wa - 1 - x0; }
          ylo = max(ylo, 0); xlo = max(xlo, 0); yhi = min(yhi, e_kh - 1); xhi = min(xhi, e_kw - 1);
          unsigned xm = (xhi >= xlo) ? (((1u << (xhi - xlo + 1)) - 1u) << xlo) : 0u;
          unsigned tm_ = 0u;
          for (int ty = ylo; ty <= yhi; ++ty) tm_ |= xm << (ty * e_kw);
          tapm[i] = tm_;
        }
      }
      // advance (ox, oy, ni) by 32 pixels
      d_ox += 32;
      while (d_ox >= e_wo) { d_ox -= e_wo; ++d_oy; }
      while (d_oy >= e_ho) { d_oy -= e_ho; ++d_ni; }
    }
  }
  int boff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int n = n0 + r0 + 32 * i;
    if (n >= G.nc) n = G.nc - 1;                 // clamp: columns >= nc are computed but never stored
    boff[i] = ((grp * G.nc + n) * ktot_w + (MODE == 2 ? 0 : cj * CE)) * ESZ;
  }
  // LDS byte addresses.  Rows 32 apart share the swizzle, so the i / j sub-tiles are reached with
  // immediate offsets (i * 4096) from ONE register per q.  (With MDD_SINGLE_BUF=0 a second LDS stage is
  // toggled by XOR-ing one bit into these few registers per K-step: buffer strides are powers of two.)
  constexpr int ABUF = NBUF == 2 ? BM * 128 : 0, BBUF = NBUF == 2 ? BN * 128 : 0;
  // MODE 4 (3x3, stride 1, "same" size, one 128-byte channel row per pixel and group): the A operand of
  // ALL nine taps is one contiguous slab of the NHWC tensor -- source pixel = m + (ty-1)*W + (tx-1) for every
  // tap that lies inside the image -- so the block loads pixels [m0-W-1, m0+BM+W] ONCE into LDS and the
  // K loop (one ROW of three taps per step: 24 KB of B per stage, so the next stage's weight loads have three
  // taps of MFMAs to hide behind) reads its A fragments from the slab at a per-tap pixel offset; taps outside
  // the image (per-row bit mask, as in MODE 1) read a row of zeros.  MODE 1 fetched the same pixels nine
  // times through L1 with one latency-bound round trip per tap.
  const int slab_pad = e_wo + 1;
  const int slab_pix = BM + 2 * e_wo + 2;
  const int BBASE = MODE == 4 ? (slab_pix + 1) * 128 : NBUF * BM * 128;
  const int l31 = lane & 31, lh = lane >> 5;
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    rdA[q] = lds_off(wm * TM * 32 + l31, 2 * q + lh);
    rdB[q] = BBASE + lds_off(wn * TN * 32 + l31, 2 * q + lh);
  }
  int wrA = lds_off(r0, cj), wrB = BBASE + lds_off(r0, cj);   // stage 0 goes to buffer 0
  // MODE 4: slab pixel of this lane's fragment rows at the centre tap, and their tap-validity masks
  int sp0[MODE == 4 ? TM : 1];
  unsigned tapf[MODE == 4 ? TM : 1];
  if constexpr (MODE == 4) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * TM * 32 + i * 32 + l31, m = m0 + row;
      sp0[i] = row + slab_pad;
      tapf[i] = 0u;
      if (m < e_M) {
        const int ox = m % e_wo, oy = (m / e_wo) % e_ho;
        int ylo, yhi, xlo, xhi;
        if (G.transposed) { const int y0 = oy + G.pad, x0 = ox + G.pad; ylo = y0 - (G.ha - 1); yhi = y0; xlo = x0 - (G.wa - 1); xhi = x0; }
        else { const int y0 = oy - G.pad, x0 = ox - G.pad; ylo = -y0; yhi = G.ha - 1 - y0; xlo = -x0; xhi = G.wa - 1 - x0; }
        ylo = max(ylo, 0); xlo = max(xlo, 0); yhi = min(yhi, 2); xhi = min(xhi, 2);
        const unsigned xm = (xhi >= xlo) ? (((1u << (xhi - xlo + 1)) - 1u) << xlo) : 0u;
        unsigned tm_ = 0u;
        for (int ty = ylo; ty <= yhi; ++ty) tm_ |= xm << (ty * 3);
        tapf[i] = tm_;
      }
    }
  }

  // K cursor of this thread's chunk column (generic path): advanced with compare/subtract only
  int c_kidx = cj * CE, c_tap = 0, c_kcq = 0, c_ty = 0, c_tx = 0, c_tapoff = 0;
  const int tap_row = G.wa * G.ca_tot * ESZ, tap_col = G.ca_tot * ESZ;
  auto cursor_init = [&]() __attribute__((always_inline)) {
    c_kidx = cj * CE;
    c_tap = c_kidx / G.kc;
    c_kcq = c_kidx - c_tap * G.kc;
    c_ty = c_tap / e_kw;
    c_tx = c_tap - c_ty * e_kw;
    c_tapoff = c_ty * tap_row + c_tx * tap_col;
  };
  if constexpr (MODE != 0) cursor_init();

  struct Stage { u32x4 ra[RA], rb[RB * TS]; };
  Stage s0;
  auto load_tile = [&](int kt, Stage& S) __attribute__((always_inline)) {
    const bool second = kt >= nk1;
    const char* A = (const char*)(second ? p.A2 : p.A1);
    const char* B = (const char*)(second ? p.B2 : p.B1);
    const int ks = second ? kt - nk1 : kt;          // K-step within the source
    if constexpr (MODE == 4) {
      (void)A;                                      // the A operand lives in the LDS slab
    } else if constexpr (MODE == 0) {
      const char* Ak = A + ks * 128;                // uniform base: zero VALU per load
      const bool kok = KFULL || (ks * KE + cj * CE) < ktot;
#pragma unroll
      for (int i = 0; i < RA; ++i)
        S.ra[i] = *(const u32x4*)(kok ? Ak + (unsigned)aoff[i] : Z);
    } else {
      if (kt == nk1) cursor_init();
      const bool kok = c_kidx < ktot;
      const int koff = (grp * G.kc + c_kcq) * ESZ + (G.transposed ? -c_tapoff : c_tapoff);
      // MODE 2: the class tap (a,b) is weight tap (ty0+2a, tx0+2b)
      const int bt = MODE == 2 ? (((ty0 + 2 * c_ty) * G.k + tx0 + 2 * c_tx) * G.kc + c_kcq) * ESZ : 0;
      const bool bok2 = kok;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        if constexpr (MODE == 1 || MODE == 2) {
          bool ok = kok && ((tapm[i] >> c_tap) & 1u);
          S.ra[i] = *(const u32x4*)(ok ? A + (aoff[i] + koff) : Z);
        } else {   // stride-2 data gradient: only taps of matching parity hit an input pixel
          int ny = s2y[i] - c_ty, nx = s2x[i] - c_tx;
          bool ok = kok && tapm[i] && ny >= 0 && nx >= 0 && ((ny | nx) & 1) == 0 &&
                    (ny >> 1) < G.ha && (nx >> 1) < G.wa;
          size_t off = ((size_t)(s2b[i] + (ny >> 1) * G.wa + (nx >> 1)) * G.ca_tot + grp * G.kc + c_kcq) * ESZ;
          S.ra[i] = *(const u32x4*)(ok ? A + off : Z);
        }
      }
      c_kidx += KE;
      c_kcq += KE;
      while (c_kcq >= G.kc) {
        c_kcq -= G.kc;
        ++c_tap;
        if (++c_tx == e_kw) { c_tx = 0; ++c_ty; }
        c_tapoff = c_ty * tap_row + c_tx * tap_col;
      }
      if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
          S.rb[i] = *(const u32x4*)(bok2 ? B + (boff[i] + bt) : Z);
      }
    }
    if constexpr (MODE == 4) {
      const char* Bk = B + ks * (TS * 128);         // three taps = three consecutive 128-byte K rows
#pragma unroll
      for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int i = 0; i < RB; ++i) S.rb[t * RB + i] = *(const u32x4*)(Bk + (unsigned)boff[i] + t * 128);
    } else if constexpr (MODE != 2) {
      const char* Bk = B + ks * 128;
      const bool kok = KFULL || (ks * KE + cj * CE) < ktot;
#pragma unroll
      for (int i = 0; i < RB; ++i)
        S.rb[i] = *(const u32x4*)(kok ? Bk + (unsigned)boff[i] : Z);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr bool hi_only = PREC == 2;   // fp32 stash, one bf16 per operand (mixed modes / attribution)
  auto store_tile = [&](const Stage& S) __attribute__((always_inline)) {
    if constexpr (MODE != 4) {
#pragma unroll
      for (int i = 0; i < RA; ++i) *(u32x4*)(smem + wrA + i * 4096) = MT::stage(S.ra[i], hi_only);
    }
#pragma unroll
    for (int t = 0; t < TS; ++t)
#pragma unroll
      for (int i = 0; i < RB; ++i)
        *(u32x4*)(smem + wrB + t * (BN * 128) + i * 4096) = MT::stage(S.rb[t * RB + i], hi_only);
    wrA ^= ABUF; wrB ^= BBUF;
  };
  // MODE 4: pixels [m0 - W - 1, m0 + BM + W] of this group's channel row -> LDS, eight 16-byte loads per
  // thread in flight at a time; pixels before / after the tensor read zeros (only masked taps touch them)
  auto load_slab = [&](const void* Asrc) __attribute__((always_inline)) {
    const int p_lo = m0 - slab_pad;
    const int nch = slab_pix * 8;
    const char* Ag = (const char*)Asrc + (size_t)grp * G.kc * ESZ + cj * 16;
    for (int base = 0; base < nch; base += 2048) {
      u32x4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pl = (base >> 3) + 32 * u + r0, pg = p_lo + pl;
        const bool ok = pl < slab_pix && pg >= 0 && pg < p.M;
        r[u] = *(const u32x4*)(ok ? Ag + (size_t)pg * G.ca_tot * ESZ : Z);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pl = (base >> 3) + 32 * u + r0;
        if (pl < slab_pix) *(u32x4*)(smem + lds_off(pl, cj)) = r[u];
      }
    }
  };
  auto compute = [&](int kt) __attribute__((always_inline)) {
    if constexpr (MODE == 4) {
      const int ty = kt >= nk1 ? kt - nk1 : kt;       // this step = the three taps (ty, 0..2)
#pragma unroll
      for (int tx = 0; tx < TS; ++tx) {
        const int tap = ty * 3 + tx;
        const int toff = G.transposed ? (1 - ty) * e_wo + (1 - tx) : (ty - 1) * e_wo + (tx - 1);
        int abase[TM], asw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int sp = ((tapf[i] >> tap) & 1u) ? sp0[i] + toff : slab_pix;   // slab_pix = the row of zeros
          abase[i] = sp * 128; asw[i] = (sp >> 1) & 7;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x4 af[TM], bf[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = *(const u32x4*)(smem + abase[i] + (((2 * q + lh) ^ asw[i]) << 4));
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[j] = *(const u32x4*)(smem + rdB[q] + tx * (BN * 128) + j * 4096);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) MT::step(af[i], bf[j], acc[i][j]);
        }
      }
    } else
    if constexpr (MDD_FRAG_PREFETCH && sizeof(AT) == 2 && MODE == 0) {
    // the LDS fragments of K-slice q+1 are fetched before the MFMAs of slice q are issued (the scheduling
    // barriers keep the compiler from sinking the fetch back next to its use): +11..14 % on the long-K,
    // single-round shapes in tools/micro/gemm_core.hip, neutral elsewhere
    u32x4 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *(const u32x4*)(smem + rdA[0] + i * 4096);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *(const u32x4*)(smem + rdB[0] + j * 4096);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(q + 1) & 1][i] = *(const u32x4*)(smem + rdA[q + 1] + i * 4096);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(q + 1) & 1][j] = *(const u32x4*)(smem + rdB[q + 1] + j * 4096);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) MT::step(af[q & 1][i], bf[q & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    } else if constexpr (PREC >= 1 && MDD_SPLIT_3TERM && WGM == 2) {
    // bf16x2, 2x2-wave instances: the fragments of two K-slices at once, three MFMAs per pair (step2).  The
    // 4x1-wave instances keep one slice at a time: the second fragment set does not fit their 168 registers
    // (measured: 256x64 class 144 -> 191 ms per C2 iteration with it, 128x128 class 190 -> 170 ms).
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
      u32x4 af[2][TM], bf[2][TN];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[h][i] = *(const u32x4*)(smem + rdA[q + h] + i * 4096);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[h][j] = *(const u32x4*)(smem + rdB[q + h] + j * 4096);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) MT::step2(af[0][i], af[1][i], bf[0][j], bf[1][j], acc[i][j]);
    }
    } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const u32x4*)(smem + rdA[q] + i * 4096);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *(const u32x4*)(smem + rdB[q] + j * 4096);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) MT::step(af[i], bf[j], acc[i][j]);
    }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { rdA[q] ^= ABUF; rdB[q] ^= BBUF; }
  };
  if constexpr (MODE == 4) {
    if (tid < 8) *(u32x4*)(smem + slab_pix * 128 + tid * 16) = u32x4{0u, 0u, 0u, 0u};
    load_slab(p.A1);
  }
  load_tile(0, s0);
  store_tile(s0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1, s0);
    if constexpr (MODE == 4) {
      if (kt == nk1) {      // second source of a tangent launch: its slab replaces the first one's
        load_slab(p.A2);    // (every wave is past the barrier that ended the last step on slab 1)
        __syncthreads();
      }
    }
    compute(kt);
    if (NBUF == 1) __syncthreads();      // everyone is done reading the slab before it is replaced
    if (kt + 1 < nk) store_tile(s0);
    __syncthreads();
  }

#define MDD_EPI_HOISTED 0
#include "conv_gemm_epilogue.inc"
#undef MDD_EPI_HOISTED
}


// ======================================================================================================
// Wide pointwise layers in bf16 (the ViT linears: 19,700 token rows, K and N in 768..3072): a 256 x 256 output tile
// per 512-thread block, one block per CU, the schedule of tools/micro/gemm_pipe.hip (and of k_wgrad_pipe):
//   * 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = acc[4][2] of v_mfma_f32_32x32x16_bf16
//   * a K-tile (64 deep) = four HALF-TILES of 128 rows x 128 B (16 KB): A0, B0, B1, A1 -- A_h holds rows [64h, 64h+64) of
//     both wave rows, B_h columns [32h, 32h+32) of all four wave columns, so every wave needs them in the same order --
//     in an 8-slot LDS ring filled by global_load_lds_dwordx4 six phases ahead (swizzle on the source address)
//   * a K-tile = four PHASES: {ds_read_b128 this phase's fragments, issue one half-tile, counted vmcnt} barrier
//     {8 MFMAs: one 64 x 32 quadrant x K 64} barrier; wave row 1 runs ONE barrier behind wave row 0, so of the two
//     waves of a SIMD one issues MFMAs while the other reads LDS.  Hazard rules: see k_wgrad_pipe (conv_wgrad.hip).
// Rows past M and columns past nc are clamped (computed, never stored).  The epilogue is the common one.
#ifndef MDD_PIPE_MIN
#define MDD_PIPE_MIN 768     // narrowest layer (K and output channels) taken by k_gemm_pipe: the ViT linears; NFNet-l0's 512- and 1536-wide layers ran 0.6 % slower per iteration with it
#endif
#ifndef MDD_PIPE_EPI_WAVE_SYNC
#define MDD_PIPE_EPI_WAVE_SYNC 1   // k_gemm_pipe: the waves of a block run their epilogues independently (wave-private staging)
#endif
#ifndef MDD_PIPE_FAST_GELU
#define MDD_PIPE_FAST_GELU 1   // k_gemm_pipe: GELU, GELU', GELU'' from the 1.5e-7 erf approximation with one shared exponential
#endif
#ifndef MDD_PIPE_ACT1
#define MDD_PIPE_ACT1 1      // 1: the GELU-epilogue launches (fc1 forward, fc2 data gradient: 3072-wide outputs, two to five
                             // stashed tensors per output tile) run on k_gemm_pipe too.  Per-launch it looks slower -- with one block
                             // per CU nothing overlaps those HBM-bound epilogues (312 vs 232 us per launch inside configs[4]) -- but the iteration is faster with it (426 vs 445 ms): 0 is kept for the record
#endif
template <int N_> DEVI void wait_vm() {
  __builtin_amdgcn_s_waitcnt((N_ & 0xF) | ((N_ >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}
DEVI void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <int ACT_>
__global__ __launch_bounds__(512, 1) void k_gemm_pipe(const KArgs p) {
  typedef bf16 AT;
  constexpr int ACT = ACT_ == 1 ? (MDD_PIPE_FAST_GELU ? 2 : 1) : 0;
  constexpr int TM = 4, TN = 2, MODE = 0, CE = 8, BM = 256;
  constexpr bool IBK = false, EPI_WAVE_SYNC = MDD_PIPE_EPI_WAVE_SYNC != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ConvGeom& G = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  int bid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  }
  const int nt = bid % p.ntiles, mt = bid / p.ntiles, grp = 0;
  const int e_M = p.M, e_ho = G.ho, e_wo = G.wo, py = 0, px = 0;
  const int m0 = mt * 256, n0 = nt * 256;
  const int l31 = lane & 31, lh = lane >> 5;
  const int K = G.kc;
  const int nk1 = K >> 6;
  const int nk = p.A2 ? 2 * nk1 : nk1, H = 4 * nk;
  // loader: per half-tile two 1-KB wave instructions; instruction j covers LDS rows j*64 + wave*8 + (lane>>3), this
  // lane's 16-byte slot lane&7 receives global chunk slot ^ swizzle(row)
  unsigned aoff[2][2], boff[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 64 + wave * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((r >> 1) & 7);
      int m = m0 + (r >> 6) * 128 + hh * 64 + (r & 63); if (m >= e_M) m = e_M - 1;
      int n = n0 + (r >> 5) * 64 + hh * 32 + (r & 31); if (n >= G.nc) n = G.nc - 1;
      aoff[hh][j] = (unsigned)((m * G.ca_tot + ch * 8) * 2);
      boff[hh][j] = (unsigned)((n * K + ch * 8) * 2);
    }
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ra_ = wm * 64 + l31, rb_ = wn * 32 + l31;
    rdA[q] = ra_ * 128 + (((2 * q + lh) ^ ((ra_ >> 1) & 7)) << 4);
    rdB[q] = rb_ * 128 + (((2 * q + lh) ^ ((rb_ >> 1) & 7)) << 4);
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  // half-tile h = 4*kt + {0: A0, 1: B0, 2: B1, 3: A1} -> ring slot h & 7
  auto issue = [&](int h) __attribute__((always_inline)) {
    const int kt = h >> 2, jj = h & 3;
    const bool second = kt >= nk1;
    const int ktl = second ? kt - nk1 : kt;
    const bool isa = jj == 0 || jj == 3;
    const int hh = jj >= 2 ? 1 : 0;
    const char* src = (const char*)(isa ? (second ? p.A2 : p.A1) : (second ? p.B2 : p.B1)) + (size_t)ktl * 128;
    char* dst = smem + (h & 7) * 16384 + wave * 1024;
    const unsigned o0 = isa ? aoff[hh][0] : boff[hh][0], o1 = isa ? aoff[hh][1] : boff[hh][1];
    __builtin_amdgcn_global_load_lds((gptr_t)(src + o0), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + o1), (lptr_t)(dst + 8192), 16, 0, 0);
  };
  auto mma = [](const u32x4& a, const u32x4& b, f32x16& c) __attribute__((always_inline)) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  };
#pragma unroll
  for (int h = 0; h < 6; ++h) issue(h);       // the launcher guarantees K >= 128: at least eight half-tiles exist
  wait_vm<8>();                                // A0, B0 of K-tile 0 have landed (this wave's share)
  raw_barrier();
  if (wm == 1) raw_barrier();
  u32x4 af[2][4], b0[4], b1[4];
  for (int kt = 0; kt < nk; ++kt) {
    const int g0 = 4 * kt;
    const char* sA0 = smem + ((g0 + 0) & 7) * 16384;
    const char* sB0 = smem + ((g0 + 1) & 7) * 16384;
    const char* sB1 = smem + ((g0 + 2) & 7) * 16384;
    const char* sA1 = smem + ((g0 + 3) & 7) * 16384;
    // ---- phase 0: a0, b0 ; quadrant (0, 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) b0[q] = *(const u32x4*)(sB0 + rdB[q]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = *(const u32x4*)(sA0 + rdA[q] + i * 4096);
    if (g0 + 6 < H) { issue(g0 + 6); wait_vm<8>(); } else wait_vm<0>();
    raw_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b0[q], acc[i][0]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
    // ---- phase 1: b1 ; quadrant (0, 1)
#pragma unroll
    for (int q = 0; q < 4; ++q) b1[q] = *(const u32x4*)(sB1 + rdB[q]);
    if (g0 + 7 < H) { issue(g0 + 7); wait_vm<8>(); } else wait_vm<0>();
    raw_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b1[q], acc[i][1]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
    // ---- phase 2: a1 ; quadrant (1, 1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = *(const u32x4*)(sA1 + rdA[q] + i * 4096);
    if (g0 + 8 < H) { issue(g0 + 8); wait_vm<10>(); } else wait_vm<0>();
    raw_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b1[q], acc[2 + i][1]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
    // ---- phase 3: nothing new ; quadrant (1, 0)
    if (g0 + 9 < H) { issue(g0 + 9); wait_vm<8>(); } else wait_vm<0>();
    raw_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b0[q], acc[2 + i][0]);
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
  }
  if (wm == 0) raw_barrier();
  __syncthreads();
#define MDD_EPI_HOISTED MDD_PIPE_EPI_HOIST
#include "conv_gemm_epilogue.inc"
#undef MDD_EPI_HOISTED
}

// false = shape not taken (the caller goes on to the one-stage kernel)
bool launch_pipe_gemm(const KArgs& a, hipStream_t st) {
  const ConvGeom& g = a.g;
  const bool pw = g.k == 1 && g.stride == 1 && g.pad == 0 && g.groups == 1;
  if (!MDD_PIPE_ACT1 && a.ep.act == 1) return false;
  if (!pipe_kernels_enabled() || !pw || g.prec != 0 || g.kc < MDD_PIPE_MIN || g.nc < MDD_PIPE_MIN || (g.kc & 63) || (g.nc & 7) || (g.ca_tot & 7) ||
      (g.co_tot & 7) || a.M < 8192 || a.ep.ib != nullptr || (int64_t)a.M * g.ca_tot * 2 >= (1ll << 31) ||
      (int64_t)g.nc * g.kc * 2 >= (1ll << 31))
    return false;
  KArgs k = a;
  k.mtiles = (a.M + 255) / 256;
  k.ntiles = (g.nc + 255) / 256;
  // (k.dbg: the timing-only switches of debug builds reach this kernel through the common epilogue)
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute((const void*)k_gemm_pipe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_gemm_pipe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) != hipSuccess)
      return false;
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  const unsigned blocks = (unsigned)(k.mtiles * k.ntiles);
  if (a.ep.act == 1) k_gemm_pipe<1><<<blocks, 512, 131072, st>>>(k);
  else k_gemm_pipe<0><<<blocks, 512, 131072, st>>>(k);
  return true;
}

template <class AT, int WGM, int WGN, int TM, int TN, int MODE, bool KFULL, int PREC, bool IBK = false, int ACT = 0>
void launch_cfg(const KArgs& a, hipStream_t st) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  KArgs k = a;
  k.mtiles = (a.M + BM - 1) / BM;
  if (MODE == 2) k.mtiles = 4 * ((a.g.nimg * (a.g.ho / 2) * (a.g.wo / 2) + BM - 1) / BM);
  k.ntiles = (a.g.nc + BN - 1) / BN;
  size_t shm = (MDD_SINGLE_BUF ? 1 : 2) * (BM + BN) * 128;
  if (MODE == 4) shm = (size_t)(BM + 2 * a.g.wo + 3) * 128 + (size_t)3 * BN * 128;   // slab + zero row + B stage (3 taps)
  size_t shm_epi = 4 * (size_t)32 * (TN * 32 + 4) * sizeof(float);  // per-wave transpose, 32 rows at a time
  if (shm_epi > shm) shm = shm_epi;
  // the dynamic-LDS limit is a per-device function attribute: set it once per (instance, device)
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_conv_gemm<AT, WGM, WGN, TM, TN, MODE, KFULL, PREC, IBK, ACT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       MODE == 4 ? (int)MODE4_MAX_LDS : (int)shm);
    if (e == hipSuccess) attr_devs.fetch_or(bit, std::memory_order_release);
  }
  int64_t blocks = (int64_t)k.mtiles * k.ntiles * a.g.groups;
  k_conv_gemm<AT, WGM, WGN, TM, TN, MODE, KFULL, PREC, IBK, ACT><<<(unsigned)blocks, 256, shm, st>>>(k);
}

}  // namespace

template <class AT>
void launch_conv_gemm(const ConvGeom& g, const AT* A1, const AT* B1, const AT* A2, const AT* B2,
                      const ConvEpi& ep, hipStream_t st) {
  KArgs a;
  a.A1 = A1; a.B1 = B1; a.A2 = A2; a.B2 = B2;
  a.g = g; a.ep = ep;
  a.M = g.nimg * g.ho * g.wo;
  a.mtiles = a.ntiles = 0;
#ifdef MDD_DEBUG_SWITCHES
  static const int dbg = [] { const char* e = getenv("MDD_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
#else
  a.dbg = 0;
#endif
  // tile selection by output-channel width per group
  // MODE 0: pointwise (pure GEMM) ; 1: taps at linear offsets (any forward conv, stride-1 dgrad) ;
  // 2: stride-2 data gradient.  KFULL: K is a whole number of 128-byte steps (no tail select).
  const bool pw = g.k == 1 && g.stride == 1 && g.pad == 0 && g.groups == 1;
  const bool s2t = g.transposed && g.stride == 2;
  const int mode = pw ? 0 : (s2t ? (((g.ho | g.wo) & 1) == 0 && g.k <= 3 ? 2 : 3) : 1);
  const bool kfull = ((g.k * g.k * g.kc) % Mma<AT>::KE) == 0;
  // per-image bias (ConvEpi::ib): pointwise data gradients with an activation epilogue only
  const bool ibk = ep.ib != nullptr && mode == 0 && (ep.mode == EPI_BWD || ep.mode == EPI_BWD_T);
  if constexpr (sizeof(AT) == 2) {
    // MODE 4: 3x3 stride-1 "same" conv (forward or data gradient) whose group is one 128-byte channel row
    const bool slab = g.k == 3 && g.stride == 1 && g.pad == 1 && g.kc == 64 && g.nc <= 64 && g.ha == g.ho &&
                      g.wa == g.wo && (size_t)(256 + 2 * g.wo + 3) * 128 + 3 * 64 * 128 <= MODE4_MAX_LDS;
    if (slab) { launch_cfg<AT, 4, 1, 2, 2, 4, true, 0>(a, st); return; }
  }
#define MDD_DISPATCH_P(WGM, WGN, TM, TN, PREC)                                                \
  do {                                                                                        \
    if (mode == 0 && ibk) { if (kfull) launch_cfg<AT, WGM, WGN, TM, TN, 0, true, PREC, true>(a, st);   \
                            else launch_cfg<AT, WGM, WGN, TM, TN, 0, false, PREC, true>(a, st); }      \
    else if (mode == 0) { if (kfull) launch_cfg<AT, WGM, WGN, TM, TN, 0, true, PREC>(a, st);            \
                     else launch_cfg<AT, WGM, WGN, TM, TN, 0, false, PREC>(a, st); }         \
    else if (mode == 1) { if (kfull) launch_cfg<AT, WGM, WGN, TM, TN, 1, true, PREC>(a, st); \
                          else launch_cfg<AT, WGM, WGN, TM, TN, 1, false, PREC>(a, st); }    \
    else if (mode == 2) launch_cfg<AT, WGM, WGN, TM, TN, 2, false, PREC>(a, st);             \
    else launch_cfg<AT, WGM, WGN, TM, TN, 3, false, PREC>(a, st);                            \
  } while (0)
#define MDD_DISPATCH(WGM, WGN, TM, TN)                                                        \
  do {                                                                                        \
    if constexpr (sizeof(AT) == 4) {                                                          \
      if (g.prec == 3) { MDD_DISPATCH_P(WGM, WGN, TM, TN, 3); break; }                        \
      if (g.prec == 2) { MDD_DISPATCH_P(WGM, WGN, TM, TN, 2); break; }                        \
      if (g.prec != 0) { MDD_DISPATCH_P(WGM, WGN, TM, TN, 1); break; }                        \
    }                                                                                         \
    MDD_DISPATCH_P(WGM, WGN, TM, TN, 0);                                                      \
  } while (0)
  if constexpr (sizeof(AT) == 2) {
    if (launch_pipe_gemm(a, st)) return;
  }
  if (ep.act == 1) {
    // exact-GELU epilogues (ViT MLP): own instances of the 128 x 128 pointwise kernel, so that the erf path's
    // registers do not weigh on the SiLU instances (a run-time switch cost the NFNet iteration +10 %, round 2);
    // conv_gemm_supports_gelu() tells the engine when it may ask for them
    if constexpr (sizeof(AT) == 4) {
      if (g.prec == 2 || g.prec == 3) { if (kfull) launch_cfg<AT, 2, 2, 2, 2, 0, true, 2, false, 1>(a, st); else launch_cfg<AT, 2, 2, 2, 2, 0, false, 2, false, 1>(a, st); return; }
      if (g.prec == 1) { if (kfull) launch_cfg<AT, 2, 2, 2, 2, 0, true, 1, false, 1>(a, st); else launch_cfg<AT, 2, 2, 2, 2, 0, false, 1, false, 1>(a, st); return; }
    }
    if (kfull) launch_cfg<AT, 2, 2, 2, 2, 0, true, 0, false, 1>(a, st);
    else launch_cfg<AT, 2, 2, 2, 2, 0, false, 0, false, 1>(a, st);
    return;
  }
  if (g.nc <= 32) MDD_DISPATCH(4, 1, 1, 1);        // 128 x 32  (stem)
  else if (g.nc <= 64) MDD_DISPATCH(4, 1, 2, 2);   // 256 x 64  (group width 64)
#if MDD_BIG_TILE == 0
  else MDD_DISPATCH(2, 2, 2, 2);                   // 128 x 128
#elif MDD_BIG_TILE == 1
  else MDD_DISPATCH(4, 1, 1, 2);                   // 128 x 64
#else
  else MDD_DISPATCH(2, 2, 1, 2);                   // 64 x 128
#endif
#undef MDD_DISPATCH
#undef MDD_DISPATCH_P
}
// GELU epilogues: pointwise contractions on the 128 x 128 tile (output width per group > 64)
static std::atomic<int> g_pipe_on{1};
bool pipe_kernels_enabled() { return g_pipe_on.load(std::memory_order_relaxed) != 0; }
void set_pipe_kernels(bool on) { g_pipe_on.store(on ? 1 : 0, std::memory_order_relaxed); }
bool conv_gemm_supports_gelu(const ConvGeom& g) {
  return g.k == 1 && g.stride == 1 && g.pad == 0 && g.groups == 1 && g.nc > 64;
}
template void launch_conv_gemm<float>(const ConvGeom&, const float*, const float*, const float*,
                                      const float*, const ConvEpi&, hipStream_t);
template void launch_conv_gemm<bf16>(const ConvGeom&, const bf16*, const bf16*, const bf16*,
                                     const bf16*, const ConvEpi&, hipStream_t);
