// How much does the shape of a conv epilogue's stores cost?  Writes a [M][256] fp32 tensor (M = 262144: 268 MB, the output of
// the ResNet layer1 64->256 1x1 conv at B=16, 512x512) in tiles of 128 rows x 128 columns per block (512 threads), each
// wave-instruction covering SEG bytes of ROWS_PER_INSTR = 1024 / SEG different rows: SEG = 128 (the per-wave patch epilogue of
// store_tiles), 512 (a block-wide patch), 1024 (full rows, BN = 256).  build: hipcc --offload-arch=gfx950 -O3 -o write_pattern write_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SEG>
__global__ __launch_bounds__(512) void wr(float* out, int M, int N, float v) {
  const int tilesN = N / 128;
  const int tile_m = blockIdx.x / tilesN, tile_n = blockIdx.x % tilesN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int LPR = SEG / 16;          // lanes per row segment
  constexpr int RPI = 64 / LPR;          // rows per wave-instruction
  // the block's 128 x 128 tile = 128 rows x 512 B; wave w owns ...
  if (SEG == 128) {
    // wave tile 64 rows x 32 cols (2 x 4 waves): 8 instructions of 8 rows x 128 B
    const int wm = wave / 4, wn = wave % 4;
    for (int it = 0; it < 8; ++it) {
      const int row = tile_m * 128 + wm * 64 + it * 8 + lane / 8;
      const int col = tile_n * 128 + wn * 32 + (lane % 8) * 4;
      *reinterpret_cast<f32x4*>(out + (long long)row * N + col) = f32x4{v, v, v, v};
    }
  } else {
    // block-wide patch: wave w owns rows [16 w, 16 w + 16), an instruction covers RPI rows x SEG bytes
    constexpr int IT = 16 * 512 / 1024;   // 8 instructions per wave either way
    for (int it = 0; it < IT; ++it) {
      const int idx = it * 64 + lane;            // 16-byte chunk index inside the wave's 16 rows x 512 B
      const int row = tile_m * 128 + wave * 16 + idx / 32;
      const int col = tile_n * 128 + (idx % 32) * 4;
      *reinterpret_cast<f32x4*>(out + (long long)row * N + col) = f32x4{v, v, v, v};
    }
  }
}
__global__ __launch_bounds__(256) void wr_linear(float* out, long long n4, float v) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
    *reinterpret_cast<f32x4*>(out + i * 4) = f32x4{v, v, v, v};
}
int main() {
  const int M = 262144, N = 256;
  float* d;
  hipMalloc(&d, (size_t)M * N * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, (double)M * N * 4 / (ms / 20 * 1e-3) / 1e12);
  };
  const int blocks = (M / 128) * (N / 128);
  time("128 B x 8 rows per instruction", [&] { hipLaunchKernelGGL(wr<128>, dim3(blocks), dim3(512), 0, 0, d, M, N, 1.f); });
  time("512 B x 2 rows per instruction", [&] { hipLaunchKernelGGL(wr<512>, dim3(blocks), dim3(512), 0, 0, d, M, N, 1.f); });
  time("linear grid-stride (2048 blocks)", [&] { hipLaunchKernelGGL(wr_linear, dim3(2048), dim3(256), 0, 0, d, (long long)M * N / 4, 1.f); });
  return 0;
}
