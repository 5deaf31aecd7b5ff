// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does lane i receive?
// LDS holds a [64 rows][64 cols] int16 matrix, value = row * 100 + col.  Lane 4q+p of each 16-lane group supplies the
// address of (row = r0(group) + q, col = c0(group) + 4p).  hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int r0 = 8 * g, c0 = 16 * g;    // a different block per group
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (r0 + q) * 64 + c0 + 4 * p));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
