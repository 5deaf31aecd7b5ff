// Semantics probe for buffer_load_dwordx4 ... lds on gfx950: per-lane global offset, LDS = M0 base + lane*16,
// out-of-range lanes must deposit zeros.  hipcc --offload-arch=gfx950 -O3 lds_dma_test.hip -o lds_dma_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* src, uint32_t bytes, float* out, const uint32_t* offs) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 4];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, (int)bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 64 * 4; i += blockDim.x) lds[i] = -7.f;
  __syncthreads();
  // each wave fills its own 1 KiB: LDS address = wave-uniform base (+ lane*16 added by the hardware)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, (int)offs[threadIdx.x], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 64 * 4; i += blockDim.x) out[i] = lds[i];
}

int main() {
  const int N = 4096;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *d, *o;
  uint32_t* doffs;
  hipMalloc(&d, N * 4);
  hipMalloc(&o, 512 * 4);
  hipMalloc(&doffs, 128 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<uint32_t> offs(128);
  for (int t = 0; t < 128; ++t) offs[t] = (uint32_t)(((t * 37) % 200) * 16);     // scattered 16-byte chunks
  offs[5] = 0x80000000u;                                                          // out of range -> zeros
  offs[70] = (uint32_t)(N * 4);                                                   // just past the end -> zeros
  hipMemcpy(doffs, offs.data(), 128 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(128), 0, 0, d, (uint32_t)(N * 4), o, doffs);
  std::vector<float> res(512);
  hipMemcpy(res.data(), o, 512 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 128; ++t)
    for (int e = 0; e < 4; ++e) {
      const bool oob = offs[t] >= (uint32_t)(N * 4);
      const float want = oob ? 0.f : (float)(offs[t] / 4 + e);
      if (res[t * 4 + e] != want) {
        if (bad < 8) printf("lane %d elem %d: got %g want %g\n", t, e, res[t * 4 + e], want);
        ++bad;
      }
    }
  printf("lds dma probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
