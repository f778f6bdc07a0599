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
// (64 bf16 / 32 f32), LDS double-buffered, next tile's global loads issued before the MFMA
// phase (register staging: the gather needs zero-fill at image borders).  LDS rows are 128 B =
// 8 chunks of 16 B; chunk position is XOR-swizzled with (row>>1)&7 so that every 16-lane group
// of a ds_read_b128 hits 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
// bf16: v_mfma_f32_32x32x16_bf16 ; f32 (parity mode): v_mfma_f32_32x32x2_f32 (exact fp32 FMA).
// Fused epilogues (ConvEpi): bias, SiLU*beta, SiLU' / SiLU'' chain rule, residual adds.
#include <cstdlib>

#include "kernels.h"

namespace {

template <class AT> struct Mma;
template <> struct Mma<bf16> {
  static constexpr int KE = 64;  // K elements per 128-byte row
  static constexpr int CE = 8;   // elements per 16-byte chunk
  static DEVI void step(const uint4& a, const uint4& b, f32x16& acc) {
    bf16x8 av = __builtin_bit_cast(bf16x8, a), bv = __builtin_bit_cast(bf16x8, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KE = 32;
  static constexpr int CE = 4;
  static DEVI void step(const uint4& a, const uint4& b, f32x16& acc) {
    // lane-half h holds k = 4*(2q+h)+j, j=0..3 for BOTH operands: any k-permutation that is the
    // same for A and B leaves the sum unchanged.
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

DEVI uint4 mask4(uint4 v, bool keep) {
  unsigned m = keep ? 0xffffffffu : 0u;
  return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}
DEVI int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct KArgs {
  const void* A1; const void* B1; const void* A2; const void* B2;
  ConvGeom g;
  ConvEpi ep;
  int M, mtiles, ntiles, noremap;
};

template <class AT, int WGM, int WGN, int TM, int TN, bool PW, bool PF2>
__global__ __launch_bounds__(256) void k_conv_gemm(const KArgs p) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  constexpr int KE = Mma<AT>::KE, CE = Mma<AT>::CE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                 // 2 * BM * 128
  char* Bs = smem + 2 * BM * 128;  // 2 * BN * 128

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
    if (p.noremap) bid = orig;
  }
  const int nt = bid % p.ntiles; bid /= p.ntiles;
  const int grp = bid % G.groups;
  const int mt = bid / G.groups;
  const int m0 = mt * BM, n0 = nt * BN;

  const int ktot = G.k * G.k * G.kc;
  const int nk1 = (ktot + KE - 1) / KE;
  const int nk = p.A2 ? 2 * nk1 : nk1;

  // ---- per-thread staging geometry (fixed across the K loop)
  const int cj = tid & 7, r0 = tid >> 3;
  int pbase[RA], pby[RA], pbx[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      int ox = m % G.wo;
      int t = m / G.wo;
      int oy = t % G.ho;
      int ni = t / G.ho;
      pbase[i] = PW ? m : ni * G.ha * G.wa;
      if (G.transposed) { pby[i] = oy + G.pad; pbx[i] = ox + G.pad; }
      else { pby[i] = oy * G.stride - G.pad; pbx[i] = ox * G.stride - G.pad; }
    } else {
      pbase[i] = 0; pby[i] = -(1 << 20); pbx[i] = -(1 << 20);
    }
  }
  const int sh = G.stride >> 1;  // stride in {1,2}

  struct Stage { uint4 ra[RA], rb[RB]; bool okA[RA], okB[RB]; };
  Stage s0, s1;   // register stages: s1 only used by the 2-deep prefetch (PF2)
  // K cursor of this thread's chunk column: (tap, channel offset) advance by one K-step per tile
  // with compare/subtract only -- no integer division and no divergent branches in the K loop.
  // Loads are unconditional (invalid taps read offset 0 and are zeroed by a select).
  int c_kidx, c_tap, c_kcq, c_ty, c_tx;
  auto cursor_init = [&]() {
    c_kidx = cj * CE;
    c_tap = c_kidx / G.kc;
    c_kcq = c_kidx - c_tap * G.kc;
    c_ty = c_tap / G.k;
    c_tx = c_tap - c_ty * G.k;
  };
  cursor_init();
  const size_t b_row0 = (size_t)(grp * G.nc + n0 + r0) * ktot;
  auto load_tile = [&](int kt, Stage& S) {
    const AT* A = (const AT*)(kt < nk1 ? p.A1 : p.A2);
    const AT* B = (const AT*)(kt < nk1 ? p.B1 : p.B2);
    if (kt == nk1) cursor_init();
    const bool kok = c_kidx < ktot;
    const int gk = grp * G.kc + c_kcq;
    if constexpr (PW) {
      // pointwise conv = plain GEMM: row m of A is pixel m, K = channels; one add per chunk
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        bool ok = kok && pby[i] >= 0;
        S.okA[i] = ok;
        S.ra[i] = *(const uint4*)(A + (ok ? (size_t)pbase[i] * G.ca_tot + c_kidx : 0));
      }
    } else
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int iy, ix;
      bool ok = kok;
      if (G.transposed) {
        int ny = pby[i] - c_ty, nx = pbx[i] - c_tx;
        ok = ok && ny >= 0 && nx >= 0 && ((ny | nx) & (G.stride - 1)) == 0;
        iy = ny >> sh; ix = nx >> sh;
        ok = ok && iy < G.ha && ix < G.wa;
      } else {
        iy = pby[i] + c_ty; ix = pbx[i] + c_tx;
        ok = ok && (unsigned)iy < (unsigned)G.ha && (unsigned)ix < (unsigned)G.wa;
      }
      size_t off = ok ? ((size_t)(pbase[i] + iy * G.wa + ix)) * G.ca_tot + gk : 0;
      S.ra[i] = *(const uint4*)(A + off);
      S.okA[i] = ok;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      bool ok = kok && (n0 + r0 + 32 * i) < G.nc;
      size_t off = ok ? b_row0 + (size_t)(32 * i) * ktot + c_kidx : 0;
      S.rb[i] = *(const uint4*)(B + off);
      S.okB[i] = ok;
    }
    // advance the cursor by one K-step
    c_kidx += KE;
    c_kcq += KE;
    while (c_kcq >= G.kc) {
      c_kcq -= G.kc;
      ++c_tap;
      if (++c_tx == G.k) { c_tx = 0; ++c_ty; }
    }
  };
  auto store_tile = [&](int buf, const Stage& S) {
    char* a = As + buf * BM * 128;
    char* b = Bs + buf * BN * 128;
#pragma unroll
    for (int i = 0; i < RA; ++i)
      *(uint4*)(a + lds_off(r0 + 32 * i, cj)) = mask4(S.ra[i], S.okA[i]);
#pragma unroll
    for (int i = 0; i < RB; ++i)
      *(uint4*)(b + lds_off(r0 + 32 * i, cj)) = mask4(S.rb[i], S.okB[i]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  auto compute = [&](int buf) {
    const char* a = As + buf * BM * 128;
    const char* b = Bs + buf * BN * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *(const uint4*)(a + lds_off((wm * TM + i) * 32 + l31, 2 * q + lh));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *(const uint4*)(b + lds_off((wn * TN + j) * 32 + l31, 2 * q + lh));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<AT>::step(af[i], bf[j], acc[i][j]);
    }
  };
  if constexpr (!PF2) {
    load_tile(0, s0);
    store_tile(0, s0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_tile(kt + 1, s0);
      compute(buf);
      if (kt + 1 < nk) store_tile(buf ^ 1, s0);
      __syncthreads();
    }
  } else {
    // two tiles in flight: tile kt+2 is requested while tile kt is multiplied and tile kt+1 sits in
    // registers waiting for its LDS slot -- every global load gets a full K-step to land
    load_tile(0, s0);
    if (nk > 1) load_tile(1, s1);
    store_tile(0, s0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 2 < nk) load_tile(kt + 2, s0);
      compute(0);
      if (kt + 1 < nk) store_tile(1, s1);
      __syncthreads();
      if (kt + 1 >= nk) break;
      if (kt + 3 < nk) load_tile(kt + 3, s1);
      compute(1);
      if (kt + 2 < nk) store_tile(0, s0);
      __syncthreads();
    }
  }

  // ---- epilogue.  The accumulators (C/D map of the 32x32 MFMA: col = lane&31,
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5)) are transposed through this wave's LDS region so that
  // every lane owns one 16-byte chunk of consecutive channels: all epilogue reads (bias, stashed
  // c / c_t / a-bar, residual adds) and all stores are 16-byte vector accesses on full NHWC rows.
  constexpr int WROWS = TM * 32, WCOLS = TN * 32, PITCH = WCOLS + 4;
  constexpr int LPR = WCOLS / CE;      // lanes per output row
  constexpr int RPP = 64 / LPR;        // rows per pass
  float* stage = (float*)smem + wave * (WROWS * PITCH);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + j * 32 + l31] = acc[i][j][r];
  __syncthreads();

  const ConvEpi& E = p.ep;
  AT* out_raw = (AT*)E.out_raw;
  AT* out_act = (AT*)E.out_act;
  const AT* Cst = (const AT*)E.c;
  const AT* Ct = (const AT*)E.c_t;
  const AT* Ab = (const AT*)E.abar;
  const AT* add1 = (const AT*)E.add1;
  const AT* add2 = (const AT*)E.add2;
  const int lrow = lane / LPR, lcol = (lane % LPR) * CE;
  const int n = n0 + wn * WCOLS + lcol;
  if (n < G.nc) {
    const int ch = grp * G.nc + n;
    float bias[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) bias[e] = 0.f;
    const float* bp = E.mode == EPI_FWD ? E.bias : (E.mode == EPI_FWD_T ? E.bias_t : nullptr);
    if (bp) {
#pragma unroll
      for (int e = 0; e < CE; e += 4) {
        float4 b4 = *(const float4*)(bp + ch + e);
        bias[e] = b4.x; bias[e + 1] = b4.y; bias[e + 2] = b4.z; bias[e + 3] = b4.w;
      }
    }
    auto ld = [&](const AT* ptr, size_t idx, float* f) { Chunk<AT>::unpack(*(const uint4*)(ptr + idx), f); };
    auto st = [&](AT* ptr, size_t idx, const float* f) { *(uint4*)(ptr + idx) = Chunk<AT>::pack(f); };
#pragma unroll 4
    for (int ps = 0; ps < WROWS / RPP; ++ps) {
      const int row = ps * RPP + lrow;
      const int m = m0 + wm * WROWS + row;
      if (m >= p.M) continue;
      const size_t idx = (size_t)m * G.co_tot + ch;
      float v[CE], t0[CE], t1[CE], t2[CE], o[CE];
#pragma unroll
      for (int e = 0; e < CE; e += 4) {
        float4 s4 = *(const float4*)(stage + row * PITCH + lcol + e);
        v[e] = s4.x + bias[e]; v[e + 1] = s4.y + bias[e + 1];
        v[e + 2] = s4.z + bias[e + 2]; v[e + 3] = s4.w + bias[e + 3];
      }
      if (add1) {
        ld(add1, idx, t0);
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] += t0[e];
      }
      if (out_raw) st(out_raw, idx, v);
      if (!out_act) continue;
      switch (E.mode) {
        case EPI_FWD:
#pragma unroll
          for (int e = 0; e < CE; ++e) o[e] = E.beta * silu_(v[e]);
          break;
        case EPI_FWD_T:
        case EPI_BWD:
          ld(Cst, idx, t0);
#pragma unroll
          for (int e = 0; e < CE; ++e) o[e] = E.beta * dsilu_(t0[e]) * v[e];
          break;
        default:  // EPI_BWD_T
          ld(Cst, idx, t0); ld(Ct, idx, t1); ld(Ab, idx, t2);
#pragma unroll
          for (int e = 0; e < CE; ++e) o[e] = E.beta * (dsilu_(Dual(t0[e], t1[e])) * Dual(t2[e], v[e])).t;
          break;
      }
      if (add2) {
        ld(add2, idx, t0);
#pragma unroll
        for (int e = 0; e < CE; ++e) o[e] += t0[e];
      }
      st(out_act, idx, o);
    }
  }
}


template <class AT, int WGM, int WGN, int TM, int TN, bool PW, bool PF2>
void launch_cfg(const KArgs& a, hipStream_t st) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  KArgs k = a;
  k.mtiles = (a.M + BM - 1) / BM;
  k.ntiles = (a.g.nc + BN - 1) / BN;
  size_t shm = 2 * (BM + BN) * 128;
  size_t shm_epi = 4 * (size_t)(TM * 32) * (TN * 32 + 4) * sizeof(float);  // per-wave transpose
  if (shm_epi > shm) shm = shm_epi;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_conv_gemm<AT, WGM, WGN, TM, TN, PW, PF2>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  int64_t blocks = (int64_t)k.mtiles * k.ntiles * a.g.groups;
  k_conv_gemm<AT, WGM, WGN, TM, TN, PW, PF2><<<(unsigned)blocks, 256, shm, st>>>(k);
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
  static const int noremap = [] { const char* e = getenv("MDD_NOREMAP"); return e && e[0] == '1' ? 1 : 0; }();
  a.noremap = noremap;
  // tile selection by output-channel width per group
  const bool pw = g.k == 1 && g.stride == 1 && g.pad == 0 && g.groups == 1;  // pointwise: pure GEMM
  if (g.nc <= 32) launch_cfg<AT, 4, 1, 1, 1, false, false>(a, st);        // 128 x 32  (stem)
  else if (g.nc <= 64) {                                           // 256 x 64  (group width 64)
    if (pw) launch_cfg<AT, 4, 1, 2, 2, true, false>(a, st); else launch_cfg<AT, 4, 1, 2, 2, false, false>(a, st);
  } else {                                                         // 128 x 128
    if (pw) launch_cfg<AT, 2, 2, 2, 2, true, false>(a, st); else launch_cfg<AT, 2, 2, 2, 2, false, false>(a, st);
  }
}
template void launch_conv_gemm<float>(const ConvGeom&, const float*, const float*, const float*,
                                      const float*, const ConvEpi&, hipStream_t);
template void launch_conv_gemm<bf16>(const ConvGeom&, const bf16*, const bf16*, const bf16*,
                                     const bf16*, const ConvEpi&, hipStream_t);
