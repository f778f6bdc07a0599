// Micro-benchmark: HBM bandwidth of tile-structured 16-byte accesses (the conv_gemm epilogue pattern)
// versus a linear stream, on a [M x C] bf16 matrix (M = 19600, C = 1536 by default).
//   mode 0: linear grid-stride copy, 16 B per lane
//   mode 1: block = 128 x 128-channel tile, a wave covers 8 rows x 128 B per instruction (epilogue pattern)
//   mode 2: same tile, a wave covers 4 rows x 256 B per instruction
//   mode 3: block = 32 full rows (3 KB contiguous each), a wave covers one 1 KB segment per instruction
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/tile_copy.hip -o gpurun_out/tile_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

template <int MODE, int NSRC>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                             const u32x4* __restrict__ c, const u32x4* __restrict__ d,
                                             u32x4* __restrict__ out, int M, int C16, int ntn) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto body = [&](size_t i) {
    u32x4 v = a[i];
    if (NSRC > 1) { u32x4 w = b[i]; v += w; }
    if (NSRC > 2) { u32x4 w = c[i]; v += w; }
    if (NSRC > 3) { u32x4 w = d[i]; v += w; }
    out[i] = v;
  };
  if (MODE == 0) {
    for (size_t i = blockIdx.x * 256ull + tid; i < (size_t)M * C16; i += (size_t)gridDim.x * 256) body(i);
  } else if (MODE == 1 || MODE == 2) {
    const int nt = blockIdx.x % ntn, mt = blockIdx.x / ntn;
    const int wm = wave >> 1, wn = wave & 1;            // 2x2 waves, 64 x 64-channel sub-tiles
    constexpr int LPR = MODE == 1 ? 8 : 16;              // lanes per row
    constexpr int RPP = 64 / LPR;
    if (MODE == 1) {
      for (int ps = 0; ps < 64 / RPP; ++ps) {
        int row = mt * 128 + wm * 64 + ps * RPP + lane / LPR;
        int col = nt * 16 + wn * 8 + lane % LPR;
        if (row < M) body((size_t)row * C16 + col);
      }
    } else {
      for (int ps = 0; ps < 128 / (4 * RPP); ++ps) {
        int row = mt * 128 + ps * 4 * RPP + wave * RPP + lane / LPR;
        int col = nt * 16 + lane % LPR;
        if (row < M) body((size_t)row * C16 + col);
      }
    }
  } else {
    // 32 full rows per block
    for (int r = 0; r < 32; r += 4) {
      int row = blockIdx.x * 32 + r + wave;
      if (row >= M) continue;
      for (int cc = lane; cc < C16; cc += 64) body((size_t)row * C16 + cc);
    }
  }
}

template <int MODE, int NSRC>
float run(const u32x4* a, const u32x4* b, const u32x4* c, const u32x4* d, u32x4* out, int M, int C16) {
  int ntn = C16 / 16, grid;
  if (MODE == 0) grid = 256 * 16;
  else if (MODE == 3) grid = (M + 31) / 32;
  else grid = ((M + 127) / 128) * ntn;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k_copy<MODE, NSRC><<<grid, 256>>>(a, b, c, d, out, M, C16, ntn);
  hipEventRecord(e0);
  const int R = 20;
  for (int i = 0; i < R; ++i) k_copy<MODE, NSRC><<<grid, 256>>>(a, b, c, d, out, M, C16, ntn);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / R;
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 19600, C = argc > 2 ? atoi(argv[2]) : 1536;
  int C16 = C / 8;
  size_t bytes = (size_t)M * C * 2;
  u32x4 *buf[5];
  for (int i = 0; i < 5; ++i) { hipMalloc(&buf[i], bytes); hipMemset(buf[i], i, bytes); }
  auto rep = [&](const char* name, int nsrc, float ms) {
    printf("%-28s nsrc %d: %7.1f us  %6.0f GB/s\n", name, nsrc, ms * 1e3, (nsrc + 1) * bytes / ms / 1e6);
  };
  rep("linear", 1, run<0, 1>(buf[0], buf[1], buf[2], buf[3], buf[4], M, C16));
  rep("linear", 4, run<0, 4>(buf[0], buf[1], buf[2], buf[3], buf[4], M, C16));
  rep("tile 8 rows x 128 B", 1, run<1, 1>(buf[0], buf[1], buf[2], buf[3], buf[4], M, C16));
  rep("tile 8 rows x 128 B", 4, run<1, 4>(buf[0], buf[1], buf[2], buf[3], buf[4], M, C16));
  rep("tile 4 rows x 256 B", 1, run<2, 1>(buf[0], buf[1], buf[2], buf[3], buf[4], M, C16));
  rep("tile 4 rows x 256 B", 4, run<2, 4>(buf[0], buf[1], buf[2], buf[3], buf[4], M, C16));
  rep("32 full rows per block", 1, run<3, 1>(buf[0], buf[1], buf[2], buf[3], buf[4], M, C16));
  rep("32 full rows per block", 4, run<3, 4>(buf[0], buf[1], buf[2], buf[3], buf[4], M, C16));
  return 0;
}
