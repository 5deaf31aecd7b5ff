// Practical peak of v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16 on this GPU under sustained load
// (build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  bf16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)a; bh[i] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, double flop_per_mfma, int blocks) {
  float* out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = MODE == 0 ? 2000 : 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(burn<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 10.0 * blocks * 4 * (double)iters * 32 * flop_per_mfma;
    printf("%s blocks=%d: %.3f ms/launch  %.1f TFLOP/s\n", name, blocks, ms / 10, flop / (ms * 1e-3) / 1e12);
  }
  (void)hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 1024}) {
    run<0>("f32 32x32x2 ", 2.0 * 32 * 32 * 2, blocks);
    run<1>("bf16 32x32x16", 2.0 * 32 * 32 * 16, blocks);
  }
  return 0;
}
