// Micro-benchmark: does a WAVE-SPECIALISED block beat the fused-epilogue block of csrc/conv_gemm.hip on the
// epilogue-bound pointwise shapes?   C[M,N] = f(A[M,K] * B[N,K]^T, E0, E1, E2)  (bf16 in/out, fp32 accumulate),
// NOPS stashed [M,N] operands read once in the epilogue -- the traffic pattern of the EPI_FWD_T / EPI_BWD /
// EPI_BWD_T epilogues (1 / 1-2 / 3-4 operands).
//
//   FUSED<NOPS>  : today's structure -- 256 threads, 3 blocks per CU, every wave runs its K loop and then its own
//                  epilogue 32 rows at a time through LDS (8 operand loads in flight per lane at most).
//   WS<NOPS,NPF> : 512 threads, ONE persistent block per CU.  Waves 0-3 run the K loop of tile i (NPF K-steps of
//                  operands in flight in staging registers); waves 4-7 run the epilogue of tile i-1 from an fp32
//                  accumulator buffer in LDS, with ALL of their operand loads (8 passes x NOPS) issued at once --
//                  they hold no accumulators, so their 256-register budget is load queue.  Both roles execute the
//                  same s_barrier sequence (2 per K step + 2 per tile), so the epilogue waves' work is spread over
//                  the K steps and a slow role throttles the other through the barriers.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/ws_gemm.hip -o tools/micro/ws_gemm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
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
DEVI void unpack8(const u32x4& c, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(c[i] << 16); f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u); }
}
DEVI int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct Ops { const bf16* e[3]; };

// the epilogue function of one 8-element chunk: cheap math, the memory pattern is what is measured
template <int NOPS>
DEVI u32x4 epi_math(const float* v, const u32x4* q) {
  float o[8], t0[8], t1[8], t2[8];
  if (NOPS >= 1) unpack8(q[0], t0);
  if (NOPS >= 2) unpack8(q[1], t1);
  if (NOPS >= 3) unpack8(q[2], t2);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float x = v[e];
    if (NOPS >= 1) x = x * (1.f + 0.5f * t0[e]);
    if (NOPS >= 2) x += t1[e];
    if (NOPS >= 3) x += 0.25f * t2[e] * t0[e];
    o[e] = x;
  }
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pk(o[2 * i], o[2 * i + 1]);
  return r;
}

// =============================================================== FUSED: today's structure
template <int NOPS>
__global__ __launch_bounds__(256, 3) void k_fused(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                                  bf16* __restrict__ C, Ops ops, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int ntn = N / 128;
  int bid;
  { const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3); }
  const int nt = bid % ntn, mt = bid / ntn, m0 = mt * 128, n0 = nt * 128;
  const int cj = tid & 7, r0 = tid >> 3, l31 = lane & 31, lh = lane >> 5;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i; if (m >= M) m = M - 1;
    aoff[i] = (unsigned)((m * K + cj * 8) * 2);
    boff[i] = (unsigned)(((n0 + r0 + 32 * i) * K + cj * 8) * 2);
  }
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { rdA[q] = lds_off(wm * 64 + l31, 2 * q + lh); rdB[q] = 128 * 128 + lds_off(wn * 64 + l31, 2 * q + lh); }
  const int wrA = lds_off(r0, cj), wrB = 128 * 128 + lds_off(r0, cj);
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
  // epilogue as conv_gemm.hip: 32-row slab per wave through LDS, UU passes of operand loads in flight
  constexpr int PITCH = 68;
  constexpr int UU = NOPS >= 3 ? 2 : 4;
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
#pragma unroll 1
    for (int ps0 = 0; ps0 < 4; ps0 += UU) {
      u32x4 q[UU][NOPS > 0 ? NOPS : 1];
      size_t idx[UU];
      bool ok[UU];
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const int row = (ps0 + u) * 8 + lrow, m = m0 + wm * 64 + hi * 32 + row;
        ok[u] = m < M;
        idx[u] = (size_t)m * N + n0 + wn * 64 + lcol;
        if (ok[u]) {
#pragma unroll
          for (int o = 0; o < NOPS; ++o) q[u][o] = *(const u32x4*)(ops.e[o] + idx[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        if (!ok[u]) continue;
        const int row = (ps0 + u) * 8 + lrow;
        const float* s = stage + row * PITCH + lcol;
        float v[8];
        float4 a = *(const float4*)s, b = *(const float4*)(s + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        *(u32x4*)(C + idx[u]) = epi_math<NOPS>(v, q[u]);
      }
    }
  }
}

// =============================================================== WS: wave-specialised persistent block
// LDS: [K stage: A 16 KB | B 16 KB][acc buffer: 128 rows x APITCH floats]
template <int NOPS, int NPF>
__global__ __launch_bounds__(512, 2) void k_ws(const bf16* __restrict__ A, const bf16* __restrict__ B,
                                               bf16* __restrict__ C, Ops ops, int M, int N, int K, int ntiles) {
  constexpr int APITCH = 132;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* accbuf = (float*)(smem + 2 * 128 * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_mma = wave < 4;
  const int ntn = N / 128, nk = K / 64;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  // ---- role state
  const int mw = wave & 3, wm = mw >> 1, wn = mw & 1;
  const int t4 = tid & 255, cj = t4 & 7, r0 = t4 >> 3, l31 = lane & 31, lh = lane >> 5;
  int rdA[4], rdB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { rdA[q] = lds_off(wm * 64 + l31, 2 * q + lh); rdB[q] = 128 * 128 + lds_off(wn * 64 + l31, 2 * q + lh); }
  const int wrA = lds_off(r0, cj), wrB = 128 * 128 + lds_off(r0, cj);

  for (int it = 0; it <= my_tiles; ++it) {
    const int tile = blockIdx.x + it * gridDim.x;            // MMA role: tile of this iteration
    const int ptile = tile - (int)gridDim.x;                 // EPI role: the previous one
    if (is_mma) {
      const bool live = it < my_tiles;
      const int nt = tile % ntn, mt = tile / ntn, m0 = mt * 128, n0 = nt * 128;
      unsigned aoff[4], boff[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int m = m0 + r0 + 32 * i; if (m >= M) m = M - 1;
        if (!live) m = 0;
        aoff[i] = (unsigned)((m * K + cj * 8) * 2);
        boff[i] = (unsigned)((((live ? n0 : 0) + r0 + 32 * i) * K + cj * 8) * 2);
      }
      f32x16 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      u32x4 ra[NPF][4], rb[NPF][4];
      auto load = [&](int kt, int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[s][i] = *(const u32x4*)((const char*)A + kt * 128 + aoff[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[s][i] = *(const u32x4*)((const char*)B + kt * 128 + boff[i]);
      };
      auto store = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(smem + wrA + i * 4096) = ra[s][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(smem + wrB + i * 4096) = rb[s][i];
      };
      // prologue: K steps 0..NPF-1 in flight, step 0 into LDS
#pragma unroll
      for (int s = 0; s < NPF; ++s) if (live && s < nk) load(s, s);
      if (live) store(0);
      __syncthreads();
      for (int kt0 = 0; kt0 < nk; kt0 += NPF) {
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
          const int kt = kt0 + s;
          if (kt >= nk) break;
          // slot s held step kt (already in LDS); refill it with step kt+NPF
          if (live && kt + NPF < nk) load(kt + NPF, s);
          if (live) {
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
          }
          __syncthreads();
          if (live && kt + 1 < nk) store((s + 1) % NPF);
          __syncthreads();
        }
      }
      __syncthreads();                                        // B1: the epilogue waves are done with accbuf
      if (live) {
#pragma unroll
        for (int hi = 0; hi < 2; ++hi)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              accbuf[(wm * 64 + hi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * APITCH + wn * 64 + j * 32 + l31] = acc[hi][j][r];
      }
      __syncthreads();                                        // B2: accbuf holds this tile
    } else {
      // ---- epilogue role: rows [32*mw, 32*mw+32) of the previous tile; lane = (row-in-pass, 16-byte column chunk)
      const bool live = it > 0;
      const int nt = ptile % ntn, mt = ptile / ntn, m0 = mt * 128, n0 = nt * 128;
      const int prow = lane >> 4, pcol = (lane & 15) * 8;
      u32x4 q[8][NOPS > 0 ? NOPS : 1];
      size_t idx[8];
      bool ok[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int row = mw * 32 + c * 4 + prow, m = m0 + row;
        ok[c] = live && m < M;
        idx[c] = (size_t)(ok[c] ? m : 0) * N + (live ? n0 : 0) + pcol;
        if (ok[c]) {
#pragma unroll
          for (int o = 0; o < NOPS; ++o) q[c][o] = *(const u32x4*)(ops.e[o] + idx[c]);
        }
      }
      __syncthreads();                                        // pairs with the K-loop prologue barrier
      // 8 passes spread over the 2*nk barrier intervals of the K loop
      int done = 0;
      const int total_bar = 2 * nk;
      for (int b = 0; b < total_bar; ++b) {
        const int want = ((b + 1) * 8 + total_bar - 1) / total_bar;     // passes finished by the end of interval b
        for (; done < want; ++done) {
          // `done` is a loop-carried runtime index into register arrays: resolve with a switch-free unrolled select
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (c != done || !ok[c]) continue;
            const int row = mw * 32 + c * 4 + prow;
            const float* s = accbuf + row * APITCH + pcol;
            float v[8];
            float4 a = *(const float4*)s, bb = *(const float4*)(s + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bb.x; v[5] = bb.y; v[6] = bb.z; v[7] = bb.w;
            *(u32x4*)(C + idx[c]) = epi_math<NOPS>(v, q[c]);
          }
        }
        __syncthreads();
      }
      __syncthreads();                                        // B1
      __syncthreads();                                        // B2
    }
  }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0);
  const int R = 30;
  for (int i = 0; i < R; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  CHECK(hipGetLastError());
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / R * 1e3f;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const int shapes[][3] = {{19600, 384, 1536}, {19600, 1536, 384}, {19600, 768, 1536}, {78400, 128, 512}, {4900, 1536, 2304}};
  for (auto& sh : shapes) {
    const int M = sh[0], K = sh[1], N = sh[2];
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K), he((size_t)M * N);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; float v = ((s >> 9) & 0xffff) / 65536.f - 0.5f; bf16 b = (bf16)v; return *(unsigned short*)&b; };
    for (auto& v : ha) v = rnd();
    for (auto& v : hb) v = rnd();
    bf16 *A, *B, *C, *Cref, *E[3];
    CHECK(hipMalloc(&A, ha.size() * 2)); CHECK(hipMalloc(&B, hb.size() * 2));
    CHECK(hipMalloc(&C, (size_t)M * N * 2)); CHECK(hipMalloc(&Cref, (size_t)M * N * 2));
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    Ops ops;
    for (int o = 0; o < 3; ++o) {
      for (auto& v : he) v = rnd();
      CHECK(hipMalloc(&E[o], he.size() * 2));
      hipMemcpy(E[o], he.data(), he.size() * 2, hipMemcpyHostToDevice);
      ops.e[o] = E[o];
    }
    const int ntiles = ((M + 127) / 128) * (N / 128);
    const double fl = 2.0 * M * K * N;
    auto cmp = [&]() {
      std::vector<unsigned short> x((size_t)M * N), y((size_t)M * N);
      hipMemcpy(x.data(), C, x.size() * 2, hipMemcpyDeviceToHost);
      hipMemcpy(y.data(), Cref, y.size() * 2, hipMemcpyDeviceToHost);
      double maxd = 0;
      for (size_t i = 0; i < x.size(); i += 31) {
        unsigned a = (unsigned)x[i] << 16, b = (unsigned)y[i] << 16;
        double d = fabs((double)*(float*)&a - *(float*)&b); if (d > maxd) maxd = d;
      }
      return maxd;
    };
    struct Var { const char* name; std::function<float()> run; float best; int nops; bool ref; };
    std::vector<Var> vars;
    const int shm_f = 34816, shm_w = 2 * 128 * 128 + 128 * 132 * 4;
    const int gws = ntiles < ncu ? ntiles : ncu;
#define ADD_F(NOPS)                                                                                                   \
    hipFuncSetAttribute((const void*)k_fused<NOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, shm_f);               \
    vars.push_back({"FUSED ops=" #NOPS, [=] { return timeit([=] { k_fused<NOPS><<<ntiles, 256, shm_f>>>(A, B, Cref, ops, M, N, K); }); }, 1e30f, NOPS, true});
#define ADD_W(NOPS, NPF)                                                                                              \
    hipFuncSetAttribute((const void*)k_ws<NOPS, NPF>, hipFuncAttributeMaxDynamicSharedMemorySize, shm_w);             \
    vars.push_back({"WS ops=" #NOPS " pf=" #NPF, [=] { return timeit([=] { k_ws<NOPS, NPF><<<gws, 512, shm_w>>>(A, B, C, ops, M, N, K, ntiles); }); }, 1e30f, NOPS, false});
    ADD_F(0) ADD_W(0, 1) ADD_W(0, 3)
    ADD_F(1) ADD_W(1, 1) ADD_W(1, 3)
    ADD_F(3) ADD_W(3, 1) ADD_W(3, 2) ADD_W(3, 3)
    for (int round = 0; round < 3; ++round)
      for (auto& v : vars) { float t = v.run(); if (t < v.best) v.best = t; }
    for (auto& v : vars) {
      double d = -1;
      if (!v.ref) {   // re-run its reference last, then this variant, and compare
        for (auto& r : vars) if (r.ref && r.nops == v.nops) r.run();
        hipMemset(C, 0, (size_t)M * N * 2);
        v.run();
        d = cmp();
      }
      const double bytes = ((double)M * K + (double)N * K + (double)M * N * (1 + v.nops)) * 2;
      printf("M %6d K %5d N %5d  %-16s %7.1f us  %6.0f TF/s  %5.2f TB/s", M, K, N, v.name, v.best, fl / v.best / 1e6, bytes / v.best / 1e6);
      if (d >= 0) printf("   max|diff vs FUSED| %.3g", d);
      printf("\n");
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(Cref);
    for (int o = 0; o < 3; ++o) hipFree(E[o]);
  }
  return 0;
}
