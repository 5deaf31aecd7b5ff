// Which CU does block i of a launch land on?  Records (XCC, SE, CU) and start / end clock per block for a launch shaped
// like the 64x128 fp32 gather tiles (256 threads, 48 KB LDS -> up to three blocks per CU), blocks spinning for a while.
// build: hipcc --offload-arch=gfx950 -O2 -o dispatch_probe dispatch_probe.hip ; run: ./dispatch_probe [blocks] [threads] [lds_kb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
struct Rec { unsigned hwid, xcc; unsigned long long t0, t1; };
__global__ void probe(Rec* out, int spin, int lds_dw) {
  extern __shared__ float lds[];
  if (threadIdx.x < (unsigned)lds_dw) lds[threadIdx.x] = 0.f;
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) lds[0] = a;
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = Rec{hwid, xcc, t0, t1};
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 512, threads = argc > 2 ? atoi(argv[2]) : 256, kb = argc > 3 ? atoi(argv[3]) : 48;
  Rec* d;
  hipMalloc(&d, blocks * sizeof(Rec));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), kb * 1024, 0, d, 200000, 64);
    hipDeviceSynchronize();
  }
  std::vector<Rec> h(blocks);
  hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  unsigned long long tmin = ~0ull;
  for (auto& r : h) tmin = r.t0 < tmin ? r.t0 : tmin;
  for (int i = 0; i < blocks; ++i) {
    const unsigned xcc = h[i].xcc & 0xF, se = (h[i].hwid >> 13) & 7, sh = (h[i].hwid >> 12) & 1, cuid = (h[i].hwid >> 8) & 15;
    cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back(i);
    if (i < 40) printf("block %3d: xcc %u se %u sh %u cu %2u  start %6llu end %6llu\n", i, xcc, se, sh, cuid, h[i].t0 - tmin, h[i].t1 - tmin);
  }
  printf("%zu distinct CUs\n", cu.size());
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 24) break;
    printf("cu %04x:", kv.first);
    for (int b : kv.second) printf(" %d(%llu)", b, h[b].t0 - tmin);
    printf("\n");
  }
  std::map<size_t, int> hist;
  for (auto& kv : cu) hist[kv.second.size()]++;
  for (auto& kv : hist) printf("%d CUs hold %zu blocks\n", kv.second, kv.first);
  return 0;
}
