// Micro-benchmark: a 256 x 256 block tile with a ping-pong wave schedule for the long-K, wide-N contractions of the
// ViT topology (19,700 token rows; K, N in {768, 2304, 3072}), against the 128 x 128 one-stage pipeline that
// csrc/conv_gemm.hip runs today.   C[M,N] = A[M,K] * B[N,K]^T, bf16 in, fp32 accumulate, bf16 plain store.
//
//   PIPE: 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 (acc[4][2] of v_mfma_f32_32x32x16_bf16).
//   A K-tile (64 deep) is four HALF-TILES of 128 rows x 128 B (16 KB): A0, B0, B1, A1 -- half-tile A_h holds rows
//   [64h, 64h+64) of BOTH wave rows, B_h columns [32h, 32h+32) of all four wave columns, so every wave needs them in
//   the same order.  They live in an 8-slot LDS ring (128 KB) filled by global_load_lds_dwordx4 (no staging registers)
//   six phases ahead of their use.  A K-tile is four PHASES, each = {read this phase's fragments, issue one half-tile,
//   counted vmcnt} barrier {8 MFMAs = one 64 x 32 quadrant x K 64} barrier.  The waves of wave-row 1 run ONE barrier
//   behind those of wave-row 0: while one wave of a SIMD issues MFMAs the other reads LDS.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm_pipe.hip -o tools/micro/gemm_pipe.bin
#include <hip/hip_runtime.h>
#include <cmath>
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

// ---- epilogue: 32-row LDS transposition per wave, 16-byte stores (as conv_gemm.hip); wave tile TM*32 x 64
template <int TM, int PITCH>
DEVI void store_tile(f32x16 (&acc)[TM][2], char* smem, bf16* C, int M, int N, int m0, int n0, int wm, int wn,
                     int lane, int wave) {
  const int l31 = lane & 31, lh = lane >> 5;
  float* stage = (float*)smem + wave * (32 * PITCH);
  const int lrow = lane >> 3, lcol = (lane & 7) * 8;
#pragma unroll
  for (int hi = 0; hi < TM; ++hi) {
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
      int m = m0 + wm * (TM * 32) + hi * 32 + row;
      int n = n0 + wn * 64 + lcol;
      if (m >= M || n >= N) continue;
      const float* s = stage + row * PITCH + lcol;
      float4 a = *(const float4*)s, b = *(const float4*)(s + 4);
      uint4 o = make_uint4(pk(a.x, a.y), pk(a.z, a.w), pk(b.x, b.y), pk(b.z, b.w));
      *(uint4*)(C + (size_t)m * N + n) = o;
    }
  }
}

// =============================================================== REG: today's 128 x 128 pipeline
__global__ __launch_bounds__(256, 3) void k_reg(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                bf16* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + 127) / 128;
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
    int n = n0 + r0 + 32 * i; if (n >= N) n = N - 1;
    aoff[i] = (unsigned)((m * K + cj * 8) * 2);
    boff[i] = (unsigned)((n * K + cj * 8) * 2);
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
    u32x4 af[2][2], bf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[0][i] = *(const u32x4*)(smem + rdA[0] + i * 4096);
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[0][j] = *(const u32x4*)(smem + rdB[0] + j * 4096);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[(q + 1) & 1][i] = *(const u32x4*)(smem + rdA[q + 1] + i * 4096);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[(q + 1) & 1][j] = *(const u32x4*)(smem + rdB[q + 1] + j * 4096);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma(af[q & 1][i], bf[q & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (kt + 1 < nk) store();
    __syncthreads();
  }
  store_tile<2, 68>(acc, smem, C, M, N, m0, n0, wm, wn, lane, wave);
}

// =============================================================== PIPE: 256 x 256, LDS-DMA ring, ping-pong wave rows
template <int N_> DEVI void wait_vm() {
  __builtin_amdgcn_s_waitcnt((N_ & 0xF) | ((N_ >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}
DEVI void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xc07f); }

template <int STAGGER>
__global__ __launch_bounds__(512, 1) void k_pipe(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                 bf16* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int ntn = (N + 255) / 256;
  int bid;
  { const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3); }
  const int nt = bid % ntn, mt = bid / ntn, m0 = mt * 256, n0 = nt * 256;
  const int l31 = lane & 31, lh = lane >> 5;
  // loader: per half-tile two 1-KB wave instructions; instruction j covers LDS rows j*64 + wave*8 + (lane>>3),
  // this lane's 16-byte slot lane&7 receives global chunk slot ^ swizzle(row)
  unsigned aoff[2][2], boff[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 64 + wave * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((r >> 1) & 7);
      int m = m0 + (r >> 6) * 128 + hh * 64 + (r & 63); if (m >= M) m = M - 1;
      int n = n0 + (r >> 5) * 64 + hh * 32 + (r & 31); if (n >= N) n = N - 1;
      aoff[hh][j] = (unsigned)((m * K + ch * 8) * 2);
      boff[hh][j] = (unsigned)((n * K + ch * 8) * 2);
    }
  // fragment read offsets inside a half-tile slot: A rows wm*64 + i*32 + l31 ; B rows wn*32 + l31
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ra_ = wm * 64 + l31, rb_ = wn * 32 + l31;
    rdA[q] = ra_ * 128 + (((2 * q + lh) ^ ((ra_ >> 1) & 7)) << 4);
    rdB[q] = rb_ * 128 + (((2 * q + lh) ^ ((rb_ >> 1) & 7)) << 4);
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / 64, H = 4 * nk;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  // half-tile h = 4*kt + {0: A0, 1: B0, 2: B1, 3: A1} -> ring slot h & 7
  auto issue = [&](int h) {
    const int kt = h >> 2, jj = h & 3;
    char* dst = smem + (h & 7) * 16384 + wave * 1024;
    const char* src = (jj == 0 || jj == 3) ? (const char*)A : (const char*)B;
    const int hh = (jj >= 2) ? 1 : 0;
    const unsigned o0 = (jj == 0 || jj == 3) ? aoff[hh][0] : boff[hh][0];
    const unsigned o1 = (jj == 0 || jj == 3) ? aoff[hh][1] : boff[hh][1];
    __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kt * 128 + o0), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kt * 128 + o1), (lptr_t)(dst + 8192), 16, 0, 0);
  };
#pragma unroll
  for (int h = 0; h < 6; ++h) issue(h);       // K >= 128: at least eight half-tiles exist
  wait_vm<8>();                                // A0, B0 of K-tile 0 have landed (this wave's share)
  asm volatile("s_barrier" ::: "memory");
  if (STAGGER && wm == 1) asm volatile("s_barrier" ::: "memory");
  u32x4 af[2][4], b0[4], b1[4];
  for (int kt = 0; kt < nk; ++kt) {
    const int g0 = 4 * kt;
    const char* sA0 = smem + ((g0 + 0) & 7) * 16384;
    const char* sB0 = smem + ((g0 + 1) & 7) * 16384;
    const char* sB1 = smem + ((g0 + 2) & 7) * 16384;
    const char* sA1 = smem + ((g0 + 3) & 7) * 16384;
    // ---- phase 0: a0, b0 ; quadrant (0,0)
#pragma unroll
    for (int q = 0; q < 4; ++q) b0[q] = *(const u32x4*)(sB0 + rdB[q]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = *(const u32x4*)(sA0 + rdA[q] + i * 4096);
    if (g0 + 6 < H) { issue(g0 + 6); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b0[q], acc[i][0]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 1: b1 ; quadrant (0,1)
#pragma unroll
    for (int q = 0; q < 4; ++q) b1[q] = *(const u32x4*)(sB1 + rdB[q]);
    if (g0 + 7 < H) { issue(g0 + 7); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b1[q], acc[i][1]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 2: a1 ; quadrant (1,1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = *(const u32x4*)(sA1 + rdA[q] + i * 4096);
    if (g0 + 8 < H) { issue(g0 + 8); wait_vm<10>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b1[q], acc[2 + i][1]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 3: nothing new ; quadrant (1,0)
    if (g0 + 9 < H) { issue(g0 + 9); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b0[q], acc[2 + i][0]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
  }
  if (STAGGER && wm == 0) asm volatile("s_barrier" ::: "memory");
  __syncthreads();
  store_tile<4, 68>(acc, smem, C, M, N, m0, n0, wm, wn, lane, wave);
}

template <int STAGGER>
__global__ __launch_bounds__(512, 1) void k_pipe_ld(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                 bf16* __restrict__ C, int M, int N, int K, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int ntn = (N + 255) / 256;
  int bid;
  { const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3); }
  const int nt = bid % ntn, mt = bid / ntn, m0 = mt * 256, n0 = nt * 256;
  const int l31 = lane & 31, lh = lane >> 5;
  // loader: per half-tile two 1-KB wave instructions; instruction j covers LDS rows j*64 + wave*8 + (lane>>3),
  // this lane's 16-byte slot lane&7 receives global chunk slot ^ swizzle(row)
  unsigned aoff[2][2], boff[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 64 + wave * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((r >> 1) & 7);
      int m = m0 + (r >> 6) * 128 + hh * 64 + (r & 63); if (m >= M) m = M - 1;
      int n = n0 + (r >> 5) * 64 + hh * 32 + (r & 31); if (n >= N) n = N - 1;
      aoff[hh][j] = (unsigned)((m * ld + ch * 8) * 2);
      boff[hh][j] = (unsigned)((n * ld + ch * 8) * 2);
    }
  // fragment read offsets inside a half-tile slot: A rows wm*64 + i*32 + l31 ; B rows wn*32 + l31
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ra_ = wm * 64 + l31, rb_ = wn * 32 + l31;
    rdA[q] = ra_ * 128 + (((2 * q + lh) ^ ((ra_ >> 1) & 7)) << 4);
    rdB[q] = rb_ * 128 + (((2 * q + lh) ^ ((rb_ >> 1) & 7)) << 4);
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / 64, H = 4 * nk;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  // half-tile h = 4*kt + {0: A0, 1: B0, 2: B1, 3: A1} -> ring slot h & 7
  auto issue = [&](int h) {
    const int kt = h >> 2, jj = h & 3;
    char* dst = smem + (h & 7) * 16384 + wave * 1024;
    const char* src = (jj == 0 || jj == 3) ? (const char*)A : (const char*)B;
    const int hh = (jj >= 2) ? 1 : 0;
    const unsigned o0 = (jj == 0 || jj == 3) ? aoff[hh][0] : boff[hh][0];
    const unsigned o1 = (jj == 0 || jj == 3) ? aoff[hh][1] : boff[hh][1];
    __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kt * 128 + o0), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kt * 128 + o1), (lptr_t)(dst + 8192), 16, 0, 0);
  };
#pragma unroll
  for (int h = 0; h < 6; ++h) issue(h);       // K >= 128: at least eight half-tiles exist
  wait_vm<8>();                                // A0, B0 of K-tile 0 have landed (this wave's share)
  asm volatile("s_barrier" ::: "memory");
  if (STAGGER && wm == 1) asm volatile("s_barrier" ::: "memory");
  u32x4 af[2][4], b0[4], b1[4];
  for (int kt = 0; kt < nk; ++kt) {
    const int g0 = 4 * kt;
    const char* sA0 = smem + ((g0 + 0) & 7) * 16384;
    const char* sB0 = smem + ((g0 + 1) & 7) * 16384;
    const char* sB1 = smem + ((g0 + 2) & 7) * 16384;
    const char* sA1 = smem + ((g0 + 3) & 7) * 16384;
    // ---- phase 0: a0, b0 ; quadrant (0,0)
#pragma unroll
    for (int q = 0; q < 4; ++q) b0[q] = *(const u32x4*)(sB0 + rdB[q]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = *(const u32x4*)(sA0 + rdA[q] + i * 4096);
    if (g0 + 6 < H) { issue(g0 + 6); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b0[q], acc[i][0]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 1: b1 ; quadrant (0,1)
#pragma unroll
    for (int q = 0; q < 4; ++q) b1[q] = *(const u32x4*)(sB1 + rdB[q]);
    if (g0 + 7 < H) { issue(g0 + 7); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b1[q], acc[i][1]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 2: a1 ; quadrant (1,1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = *(const u32x4*)(sA1 + rdA[q] + i * 4096);
    if (g0 + 8 < H) { issue(g0 + 8); wait_vm<10>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b1[q], acc[2 + i][1]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 3: nothing new ; quadrant (1,0)
    if (g0 + 9 < H) { issue(g0 + 9); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], b0[q], acc[2 + i][0]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
  }
  if (STAGGER && wm == 0) asm volatile("s_barrier" ::: "memory");
  __syncthreads();
  store_tile<4, 68>(acc, smem, C, M, N, m0, n0, wm, wn, lane, wave);
}

template <int STAGGER>
__global__ __launch_bounds__(512, 1) void k_pipe2(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                 bf16* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int ntn = (N + 255) / 256;
  int bid;
  { const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3); }
  const int nt = bid % ntn, mt = bid / ntn, m0 = mt * 256, n0 = nt * 256;
  const int l31 = lane & 31, lh = lane >> 5;
  // loader: per half-tile two 1-KB wave instructions; instruction j covers LDS rows j*64 + wave*8 + (lane>>3),
  // this lane's 16-byte slot lane&7 receives global chunk slot ^ swizzle(row)
  unsigned aoff[2][2], boff[2][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 64 + wave * 8 + (lane >> 3);
      const int ch = (lane & 7) ^ ((r >> 1) & 7);
      int m = m0 + (r >> 6) * 128 + hh * 64 + (r & 63); if (m >= M) m = M - 1;
      int n = n0 + (r >> 5) * 64 + hh * 32 + (r & 31); if (n >= N) n = N - 1;
      aoff[hh][j] = (unsigned)((m * K + ch * 8) * 2);
      boff[hh][j] = (unsigned)((n * K + ch * 8) * 2);
    }
  // fragment read offsets inside a half-tile slot: A rows wm*64 + i*32 + l31 ; B rows wn*32 + l31
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ra_ = wm * 64 + l31, rb_ = wn * 32 + l31;
    rdA[q] = ra_ * 128 + (((2 * q + lh) ^ ((ra_ >> 1) & 7)) << 4);
    rdB[q] = rb_ * 128 + (((2 * q + lh) ^ ((rb_ >> 1) & 7)) << 4);
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / 64, H = 4 * nk;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  // half-tile h = 4*kt + {0: B0, 1: A0, 2: B1, 3: A1} -> ring slot h & 7; phase g reads half-tile g + 1
  auto issue = [&](int h) {
    const int kt = h >> 2, jj = h & 3;
    char* dst = smem + (h & 7) * 16384 + wave * 1024;
    const bool isa = jj & 1;
    const char* src = isa ? (const char*)A : (const char*)B;
    const int hh = (jj >= 2) ? 1 : 0;
    const unsigned o0 = isa ? aoff[hh][0] : boff[hh][0];
    const unsigned o1 = isa ? aoff[hh][1] : boff[hh][1];
    __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kt * 128 + o0), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kt * 128 + o1), (lptr_t)(dst + 8192), 16, 0, 0);
  };
#pragma unroll
  for (int h = 0; h < 6; ++h) issue(h);       // K >= 128: at least eight half-tiles exist
  wait_vm<10>();                               // B0 of K-tile 0 has landed (this wave's share)
  asm volatile("s_barrier" ::: "memory");
  u32x4 af[2][4], bx[4], by[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bx[q] = *(const u32x4*)(smem + rdB[q]);     // pre-phase: b0 of K-tile 0 (slot 0)
  wait_vm<8>();                                // A0 of K-tile 0
  asm volatile("s_barrier" ::: "memory");
  if (STAGGER && wm == 1) asm volatile("s_barrier" ::: "memory");
  // one K-tile; bcur holds b0 of this K-tile, bnxt receives b1 and then b0 of the next K-tile
  auto ktile = [&](int kt, u32x4 (&bcur)[4], u32x4 (&bnxt)[4]) __attribute__((always_inline)) {
    const int g0 = 4 * kt;
    const char* sA0 = smem + ((g0 + 1) & 7) * 16384;
    const char* sB1 = smem + ((g0 + 2) & 7) * 16384;
    const char* sA1 = smem + ((g0 + 3) & 7) * 16384;
    const char* sBn = smem + ((g0 + 4) & 7) * 16384;
    // ---- phase 0: a0 ; quadrant (0,0)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = *(const u32x4*)(sA0 + rdA[q] + i * 4096);
    if (g0 + 6 < H) { issue(g0 + 6); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], bcur[q], acc[i][0]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 1: b1 ; quadrant (0,1)
#pragma unroll
    for (int q = 0; q < 4; ++q) bnxt[q] = *(const u32x4*)(sB1 + rdB[q]);
    if (g0 + 7 < H) { issue(g0 + 7); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], bnxt[q], acc[i][1]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 2: a1 ; quadrant (1,1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) af[i][q] = *(const u32x4*)(sA1 + rdA[q] + i * 4096);
    if (g0 + 8 < H) { issue(g0 + 8); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    wait_lgkm0();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], bnxt[q], acc[2 + i][1]);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 3: b0 of the next K-tile (into the registers b1 has left) ; quadrant (1,0)
    if (kt + 1 < nk) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bnxt[q] = *(const u32x4*)(sBn + rdB[q]);
    }
    if (g0 + 9 < H) { issue(g0 + 9); wait_vm<8>(); } else wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma(af[i][q], bcur[q], acc[2 + i][0]);
    __builtin_amdgcn_s_setprio(0);
    wait_lgkm0();
    asm volatile("s_barrier" ::: "memory");
  };
  for (int kt = 0; kt < nk; kt += 2) {
    ktile(kt, bx, by);
    if (kt + 1 < nk) ktile(kt + 1, by, bx);
  }
  if (STAGGER && wm == 0) asm volatile("s_barrier" ::: "memory");
  __syncthreads();
  store_tile<4, 68>(acc, smem, C, M, N, m0, n0, wm, wn, lane, wave);
}

// =============================================================== host
static void check(hipError_t e, const char* w) { if (e != hipSuccess) { printf("HIP error %s: %s\n", w, hipGetErrorString(e)); exit(1); } }

template <class F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  check(hipGetLastError(), "kernel");
  return ms / R * 1e3f;
}

int main() {
  const int shapes[][3] = {{19700, 768, 3072}, {19700, 3072, 768}, {19700, 768, 2304}, {19700, 768, 768},
                           {19700, 1536, 3072}, {19700, 6144, 768}, {19600, 384, 1536}, {19600, 1536, 384},
                           {19600, 3072, 384}, {8192, 8192, 8192}};
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
    const double fl = 2.0 * M * K * N;
    const int ld = K + 64;                      // row stride + one 128-byte line: consecutive rows land on different L2 channels
    bf16 *Ap, *Bp;
    check(hipMalloc(&Ap, (size_t)M * ld * 2), "malloc"); check(hipMalloc(&Bp, (size_t)N * ld * 2), "malloc");
    hipMemcpy2D(Ap, (size_t)ld * 2, A, (size_t)K * 2, (size_t)K * 2, M, hipMemcpyDeviceToDevice);
    hipMemcpy2D(Bp, (size_t)ld * 2, B, (size_t)K * 2, (size_t)K * 2, N, hipMemcpyDeviceToDevice);
    struct Var { const char* name; std::function<float()> run; float best; };
    std::vector<Var> vars;
    const int g1 = ((M + 127) / 128) * ((N + 127) / 128), g2 = ((M + 255) / 256) * ((N + 255) / 256);
    hipFuncSetAttribute((const void*)k_reg, hipFuncAttributeMaxDynamicSharedMemorySize, 34816);
    hipFuncSetAttribute((const void*)k_pipe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k_pipe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k_pipe_ld<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k_pipe2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    vars.push_back({"REG 128x128", [&] { return timeit([&] { k_reg<<<g1, 256, 34816>>>(A, B, Cref, M, N, K); }); }, 1e30f});
    vars.push_back({"PIPE 256x256 pingpong", [&] { return timeit([&] { k_pipe<1><<<g2, 512, 131072>>>(A, B, C, M, N, K); }); }, 1e30f});
    vars.push_back({"PIPE2 balanced reads", [&] { return timeit([&] { k_pipe2<1><<<g2, 512, 131072>>>(A, B, C, M, N, K); }); }, 1e30f});
    vars.push_back({"PIPE row stride K+64", [&] { return timeit([&] { k_pipe_ld<1><<<g2, 512, 131072>>>(Ap, Bp, C, M, N, K, ld); }); }, 1e30f});
    vars.push_back({"PIPE 256x256 lockstep", [&] { return timeit([&] { k_pipe<0><<<g2, 512, 131072>>>(A, B, C, M, N, K); }); }, 1e30f});
    for (int round = 0; round < 3; ++round)
      for (auto& v : vars) { float t = v.run(); if (t < v.best) v.best = t; }
    for (size_t i = 0; i < vars.size(); ++i) {
      double maxd = 0; size_t bad = 0;
      if (i) {
        hipMemset(C, 0xff, (size_t)M * N * 2);
        vars[i].run();
        std::vector<unsigned short> x((size_t)M * N), y((size_t)M * N);
        hipMemcpy(x.data(), C, x.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(y.data(), Cref, y.size() * 2, hipMemcpyDeviceToHost);
        for (size_t e = 0; e < x.size(); e += 13) {
          unsigned a = (unsigned)x[e] << 16, b = (unsigned)y[e] << 16;
          float fa = *(float*)&a, fb = *(float*)&b;
          double d = fabs((double)fa - fb); if (!(d <= maxd)) maxd = d;
          if (!(d < 0.05)) ++bad;
        }
      }
      printf("M %6d K %5d N %5d  %-24s %8.1f us  %6.0f TF/s", M, K, N, vars[i].name, vars[i].best, fl / vars[i].best / 1e6);
      if (i) printf("   max|diff vs REG| %.3g  bad %zu", maxd, bad);
      printf("\n");
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(Cref); hipFree(Ap); hipFree(Bp);
  }
  return 0;
}
