// Micro-benchmark for the implicit-GEMM core (pointwise case): C[M,N] = A[M,K] * B[N,K]^T, bf16 in,
// fp32 accumulate, bf16 plain store.  Variants of the K-loop pipeline, same 128x128 block tile / 2x2 waves /
// 64x64 wave tile / v_mfma_f32_32x32x16_bf16 as csrc/conv_gemm.hip:
//   REG  : one LDS stage, next slab staged through registers (what conv_gemm.hip does today)
//   LDS<BK,NS> : global_load_lds_dwordx4 straight into an NS-stage LDS ring (no staging registers, no
//               ds_write), loads issued NS-1 K-steps ahead, ONE barrier per step
// Reference: tools/micro/blas_gemm_ref.py (hipBLASLt): 19600x384x1536 35 us, 19600x1536x384 30 us.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm_core.hip -o tools/micro/gemm_core.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
#define DEVI __device__ __forceinline__

DEVI void mma(const u32x4& a, const u32x4& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
DEVI unsigned pk(float lo, float hi) {
  bf16 a = (bf16)lo, b = (bf16)hi;
  return (unsigned)(*(unsigned short*)&a) | ((unsigned)(*(unsigned short*)&b) << 16);
}

// ---- shared epilogue: 32-row LDS transposition per wave, 16-byte stores (as conv_gemm.hip)
template <int PITCH>
DEVI void store_tile(f32x16 (&acc)[2][2], char* smem, bf16* C, int M, int N, int m0, int n0, int wm, int wn,
                     int lane, int wave) {
  const int l31 = lane & 31, lh = lane >> 5;
  float* stage = (float*)smem + wave * (32 * PITCH);
  const int lrow = lane >> 3, lcol = (lane & 7) * 8;
#pragma unroll
  for (int hi = 0; hi < 2; ++hi) {
    if (hi) __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + j * 32 + l31] = acc[hi][j][r];
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      int row = ps * 8 + lrow;
      int m = m0 + wm * 64 + hi * 32 + row;
      if (m >= M) continue;
      const float* s = stage + row * PITCH + lcol;
      float4 a = *(const float4*)s, b = *(const float4*)(s + 4);
      uint4 o = make_uint4(pk(a.x, a.y), pk(a.z, a.w), pk(b.x, b.y), pk(b.z, b.w));
      *(uint4*)(C + (size_t)m * N + n0 + wn * 64 + lcol) = o;
    }
  }
}

// =============================================================== REG: today's pipeline
__global__ __launch_bounds__(256, 3) void k_reg(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                bf16* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int ntn = N / 128;
  int bid;
  { const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3); }
  const int nt = bid % ntn, mt = bid / ntn, m0 = mt * 128, n0 = nt * 128;
  const int cj = tid & 7, r0 = tid >> 3, l31 = lane & 31, lh = lane >> 5;
  auto off = [](int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); };
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i; if (m >= M) m = M - 1;
    aoff[i] = (unsigned)((m * K + cj * 8) * 2);
    boff[i] = (unsigned)(((n0 + r0 + 32 * i) * K + cj * 8) * 2);
  }
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { rdA[q] = off(wm * 64 + l31, 2 * q + lh); rdB[q] = 128 * 128 + off(wn * 64 + l31, 2 * q + lh); }
  const int wrA = off(r0, cj), wrB = 128 * 128 + off(r0, cj);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u32x4 ra[4], rb[4];
  const int nk = K / 64;
  auto load = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *(const u32x4*)((const char*)A + kt * 128 + aoff[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = *(const u32x4*)((const char*)B + kt * 128 + boff[i]);
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(smem + wrA + i * 4096) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(smem + wrB + i * 4096) = rb[i];
  };
  load(0); store(); __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load(kt + 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *(const u32x4*)(smem + rdA[q] + i * 4096);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *(const u32x4*)(smem + rdB[q] + j * 4096);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma(af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < nk) store();
    __syncthreads();
  }
  store_tile<68>(acc, smem, C, M, N, m0, n0, wm, wn, lane, wave);
}


// =============================================================== REGT<WGM,WGN,NB>: register staging, WGM x WGN waves of
// 64x64, NB LDS stages (1 = two barriers per step, 2 = one barrier per step)
template <int WGM, int WGN, int NB, int LB, int PF = 0>
__global__ __launch_bounds__(64 * WGM * WGN, LB) void k_regt(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                             bf16* __restrict__ C, int M, int N, int K) {
  constexpr int NT = 64 * WGM * WGN, BM = 64 * WGM, BN = 64 * WGN;
  constexpr int RPT = NT / 8;                 // rows covered per loader pass
  constexpr int RA = BM / RPT, RB = BN / RPT;
  constexpr int ABUF = BM * 128, BBUF = BN * 128, STG = ABUF + BBUF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WGN, wn = wave % WGN;
  const int ntn = N / BN;
  int bid;
  { const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3); }
  const int nt = bid % ntn, mt = bid / ntn, m0 = mt * BM, n0 = nt * BN;
  const int cj = tid & 7, r0 = tid >> 3, l31 = lane & 31, lh = lane >> 5;
  auto off = [](int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); };
  unsigned aoff[RA], boff[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) { int m = m0 + r0 + RPT * i; if (m >= M) m = M - 1; aoff[i] = (unsigned)((m * K + cj * 8) * 2); }
#pragma unroll
  for (int i = 0; i < RB; ++i) boff[i] = (unsigned)(((n0 + r0 + RPT * i) * K + cj * 8) * 2);
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { rdA[q] = off(wm * 64 + l31, 2 * q + lh); rdB[q] = ABUF + off(wn * 64 + l31, 2 * q + lh); }
  const int wrA = off(r0, cj), wrB = ABUF + off(r0, cj);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u32x4 ra[RA], rb[RB];
  const int nk = K / 64;
  auto load = [&](int kt) {
#pragma unroll
    for (int i = 0; i < RA; ++i) ra[i] = *(const u32x4*)((const char*)A + kt * 128 + aoff[i]);
#pragma unroll
    for (int i = 0; i < RB; ++i) rb[i] = *(const u32x4*)((const char*)B + kt * 128 + boff[i]);
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) *(u32x4*)(smem + buf * STG + wrA + i * RPT * 128) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *(u32x4*)(smem + buf * STG + wrB + i * RPT * 128) = rb[i];
  };
  load(0); store(0); __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = NB == 2 ? (kt & 1) : 0;
    if (kt + 1 < nk) load(kt + 1);
    const char* base = smem + buf * STG;
    if constexpr (PF == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const u32x4*)(base + rdA[q] + i * 4096);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *(const u32x4*)(base + rdB[q] + j * 4096);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma(af[i], bf[j], acc[i][j]);
      }
    } else {
      // fragments of slice q+1 are fetched from LDS before the MFMAs of slice q are issued
      u32x4 af[2][2], bf[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[0][i] = *(const u32x4*)(base + rdA[0] + i * 4096);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[0][j] = *(const u32x4*)(base + rdB[0] + j * 4096);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[(q + 1) & 1][i] = *(const u32x4*)(base + rdA[q + 1] + i * 4096);
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[(q + 1) & 1][j] = *(const u32x4*)(base + rdB[q + 1] + j * 4096);
        }
        if (PF == 2) __builtin_amdgcn_sched_barrier(0);   // keep the fetch ahead of the MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma(af[q & 1][i], bf[q & 1][j], acc[i][j]);
        if (PF == 2) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (NB == 1) __syncthreads();
    if (kt + 1 < nk) store(NB == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }
  store_tile<68>(acc, smem, C, M, N, m0, n0, wm, wn, lane, wave);
}

// =============================================================== LDS<BK,NS>: direct-to-LDS ring
template <int N_> struct WaitVm {
  static DEVI void go() { __builtin_amdgcn_s_waitcnt((N_ & 0xF) | ((N_ >> 4) << 14) | (0x7 << 4) | (0xF << 8)); }
};

template <int BK, int NS, int LB, int PF = 0>
__global__ __launch_bounds__(256, LB) void k_lds(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                 bf16* __restrict__ C, int M, int N, int K) {
  constexpr int ROWB = BK * 2, CPR = ROWB / 16, RPI = 64 / CPR;   // row bytes, chunks/row, rows per wave-load
  constexpr int OPB = 128 * ROWB, STAGE = 2 * OPB;               // operand tile / stage bytes
  constexpr int LPO = 32 / RPI;                                   // loads per operand per wave per stage
  constexpr int LPS = 2 * LPO;
  constexpr int NQ = BK / 16;                                     // 16-wide K slices per step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int ntn = N / 128;
  int bid;
  { const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3); }
  const int nt = bid % ntn, mt = bid / ntn, m0 = mt * 128, n0 = nt * 128;
  const int l31 = lane & 31, lh = lane >> 5;
  auto swz = [](int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  // loader: wave w brings rows [32w, 32w+32) of both operand tiles; one instruction = RPI rows x CPR chunks,
  // written by the hardware to LDS at (uniform base) + lane*16, i.e. row-major; the XOR swizzle is applied
  // on the GLOBAL side: lane (row, slot) fetches chunk slot ^ swz(row).
  const int rl = lane / CPR, slot = lane % CPR;
  unsigned aoff[LPO], boff[LPO];
#pragma unroll
  for (int j = 0; j < LPO; ++j) {
    int row = wave * 32 + j * RPI + rl;
    int ch = slot ^ swz(row);
    int m = m0 + row; if (m >= M) m = M - 1;
    aoff[j] = (unsigned)((m * K + ch * 8) * 2);
    boff[j] = (unsigned)(((n0 + row) * K + ch * 8) * 2);
  }
  int rdA[NQ], rdB[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int ra_ = wm * 64 + l31, rb_ = wn * 64 + l31;
    rdA[q] = ra_ * ROWB + (((2 * q + lh) ^ swz(ra_)) << 4);
    rdB[q] = OPB + rb_ * ROWB + (((2 * q + lh) ^ swz(rb_)) << 4);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  auto issue = [&](int kt) {
    char* st = smem + (kt % NS) * STAGE + wave * 32 * ROWB;
#pragma unroll
    for (int j = 0; j < LPO; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)((const char*)A + (size_t)kt * ROWB + aoff[j]),
                                       (lptr_t)(st + j * RPI * ROWB), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < LPO; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)((const char*)B + (size_t)kt * ROWB + boff[j]),
                                       (lptr_t)(st + OPB + j * RPI * ROWB), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed: stages kt+1 .. kt+NS-2 may still be in flight
    if (kt + NS - 2 <= nk - 1) WaitVm<(NS - 2) * LPS>::go(); else WaitVm<0>::go();
    asm volatile("s_barrier" ::: "memory");     // NOT __syncthreads(): its fence would wait for the stages still in flight
    if (kt + NS - 1 < nk) issue(kt + NS - 1);     // overwrites the stage computed at step kt-1
    const char* base = smem + (kt % NS) * STAGE;
    if constexpr (PF == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        u32x4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const u32x4*)(base + rdA[q] + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *(const u32x4*)(base + rdB[q] + j * 32 * ROWB);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma(af[i], bf[j], acc[i][j]);
      }
    } else {
      u32x4 af[2][2], bf[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[0][i] = *(const u32x4*)(base + rdA[0] + i * 32 * ROWB);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[0][j] = *(const u32x4*)(base + rdB[0] + j * 32 * ROWB);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (q + 1 < NQ) {
#pragma unroll
          for (int i = 0; i < 2; ++i) af[(q + 1) & 1][i] = *(const u32x4*)(base + rdA[q + 1] + i * 32 * ROWB);
#pragma unroll
          for (int j = 0; j < 2; ++j) bf[(q + 1) & 1][j] = *(const u32x4*)(base + rdB[q + 1] + j * 32 * ROWB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma(af[q & 1][i], bf[q & 1][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();
  store_tile<68>(acc, smem, C, M, N, m0, n0, wm, wn, lane, wave);
}

// =============================================================== host
static void check(hipError_t e, const char* w) { if (e != hipSuccess) { printf("HIP error %s: %s\n", w, hipGetErrorString(e)); exit(1); } }

template <class F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0);
  const int R = 30;
  for (int i = 0; i < R; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  check(hipGetLastError(), "kernel");
  return ms / R * 1e3f;
}

int main() {
  const int shapes[][3] = {{19600, 384, 1536}, {19600, 1536, 384}, {19600, 768, 1536}, {19600, 3072, 384}, {78400, 512, 128}, {313600, 128, 256}, {4900, 1536, 2304}};
  for (auto& sh : shapes) {
    const int M = sh[0], K = sh[1], N = sh[2];
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; float v = ((s >> 9) & 0xffff) / 65536.f - 0.5f; bf16 b = (bf16)v; return *(unsigned short*)&b; };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd();
    bf16 *A, *B, *C, *Cref;
    check(hipMalloc(&A, ha.size() * 2), "malloc"); check(hipMalloc(&B, hb.size() * 2), "malloc");
    check(hipMalloc(&C, (size_t)M * N * 2), "malloc"); check(hipMalloc(&Cref, (size_t)M * N * 2), "malloc");
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    const int grid = ((M + 127) / 128) * (N / 128);
    const double fl = 2.0 * M * K * N;
    auto report = [&](const char* name, float us, bool cmp) {
      double maxd = 0;
      if (cmp) {
        std::vector<unsigned short> x((size_t)M * N), y((size_t)M * N);
        hipMemcpy(x.data(), C, x.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(y.data(), Cref, y.size() * 2, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < x.size(); i += 97) {
          unsigned a = (unsigned)x[i] << 16, b = (unsigned)y[i] << 16;
          float fa = *(float*)&a, fb = *(float*)&b;
          double d = fabs((double)fa - fb); if (d > maxd) maxd = d;
        }
      }
      printf("M %6d K %5d N %5d  %-14s %7.1f us  %6.0f TF/s%s", M, K, N, name, us, fl / us / 1e6, cmp ? "" : "\n");
      if (cmp) printf("   max|diff vs REG| %.3g\n", maxd);
    };
    // clocks drift with load history (a variant measured first looks 10-15 % faster): every variant is timed in
    // three interleaved rounds and the best round is reported
    struct Var { const char* name; std::function<float()> run; float best; bool ok; };
    std::vector<Var> vars;
    hipFuncSetAttribute((const void*)k_reg, hipFuncAttributeMaxDynamicSharedMemorySize, 34816);
    vars.push_back({"REG(1 stage)", [&] { return timeit([&] { k_reg<<<grid, 256, 34816>>>(A, B, Cref, M, N, K); }); }, 1e30f, true});
#define ADD_T(WGM, WGN, NB, LB, NAME) ADD_TP(WGM, WGN, NB, LB, 0, NAME)
#define ADD_TP(WGM, WGN, NB, LB, PF, NAME)                                                                  \
    if (N % (64 * WGN) == 0) {                                                                         \
      int shm = NB * (64 * WGM + 64 * WGN) * 128; int epi = WGM * WGN * 32 * 68 * 4; if (shm < epi) shm = epi; \
      hipFuncSetAttribute((const void*)k_regt<WGM, WGN, NB, LB, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, shm); \
      int g2 = ((M + 64 * WGM - 1) / (64 * WGM)) * (N / (64 * WGN));                                   \
      vars.push_back({NAME, [=] { return timeit([=] { k_regt<WGM, WGN, NB, LB, PF><<<g2, 64 * WGM * WGN, shm>>>(A, B, C, M, N, K); }); }, 1e30f, true}); \
    }
#define ADD_L(BK, NS, LB, NAME) ADD_LP(BK, NS, LB, 0, NAME)
#define ADD_LP(BK, NS, LB, PF, NAME)                                                                        \
    if (K % BK == 0) {                                                                                 \
      int shm = NS * 2 * 128 * BK * 2; if (shm < 34816) shm = 34816;                                   \
      hipFuncSetAttribute((const void*)k_lds<BK, NS, LB, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, shm); \
      vars.push_back({NAME, [=] { return timeit([=] { k_lds<BK, NS, LB, PF><<<grid, 256, shm>>>(A, B, C, M, N, K); }); }, 1e30f, true}); \
    }
    ADD_T(2, 2, 1, 3, "REGT 2x2 nb1") ADD_T(2, 2, 2, 2, "REGT 2x2 nb2") ADD_T(4, 2, 1, 2, "REGT 4x2 nb1")
    ADD_T(4, 2, 1, 4, "REGT 4x2 lb4") ADD_T(2, 4, 1, 4, "REGT 2x4 lb4") ADD_T(4, 2, 1, 3, "REGT 4x2 lb3")
    ADD_TP(2, 2, 1, 3, 2, "REGT 2x2 pf2")  ADD_L(64, 2, 2, "LDS bk64 x2") ADD_LP(64, 2, 2, 1, "LDS bk64 x2 pf") ADD_L(32, 3, 3, "LDS bk32 x3")
    for (int round = 0; round < 3; ++round)
      for (auto& v : vars) { float t = v.run(); if (t < v.best) v.best = t; }
    for (size_t i = 0; i < vars.size(); ++i) {
      if (i) { hipMemset(C, 0, (size_t)M * N * 2); vars[i].run(); }
      report(vars[i].name, vars[i].best, i > 0);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(Cref);
  }
  return 0;
}
