// Do LDS-DMA loads and buffer stores retire IN ORDER on gfx950's vmcnt counter?  (round 6)
//
// Why it matters: VERDICT r5 item 2a / DESIGN.md "Next" asked for a ring whose tile-boundary wait is COUNTED past the epilogue's
// stores (`s_waitcnt vmcnt(#stores)` instead of `vmcnt(0)`): legal only if an OLDER load cannot still be outstanding when the
// counter has fallen to the number of YOUNGER stores, i.e. if the counter retires loads and stores in issue order.  Round 2's lab
// note (profiles/EXPERIMENTS.md section 3, item 1) says they complete out of order with each other; LLVM's waitcnt pass treats
// gfx9 vmcnt as in-order.  This probe decides it on the hardware.
//
// One wave per block.  The wave (1) fills an LDS word per lane with a sentinel, (2) issues ONE `buffer_load_dword ... lds` from a
// COLD address (its own 16 KB-strided slot of a multi-GB buffer: an HBM round trip), (3) issues S `buffer_store_dword` to a HOT,
// L2-resident line (they can complete in a fraction of the load's latency), (4) `s_waitcnt vmcnt(S)`, (5) reads the LDS word at
// once (inline asm: the compiler adds no wait of its own) and writes what it saw.  In-order retirement => every lane sees the
// loaded value.  Out-of-order => the stores drain the counter to <= S while the load is in flight and lanes see the sentinel.
// Control: the same with `vmcnt(S + 1)` (no wait for the load at all) must show sentinels -- otherwise the probe proves nothing.
//   hipcc --offload-arch=gfx950 -O2 vmcnt_order.hip -o vmcnt_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

constexpr uint32_t kSentinel = 0xDEADBEEFu;

template <int S, bool CONTROL>
__global__ __launch_bounds__(64) void probe(const uint32_t* __restrict__ cold, long long stride_dw, uint32_t* __restrict__ hot,
                                            uint32_t* __restrict__ seen) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[64];
  const int lane = threadIdx.x;
  const long long blk = blockIdx.x;
  const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(cold + blk * stride_dw), (short)0, 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(hot + (blk & 63) * 64, (short)0, 256, 0x00020000);
  lds[lane] = kSentinel;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_raw_ptr_buffer_load_lds(cr, (__attribute__((address_space(3))) void*)lds, 4, lane * 4, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < S; ++s) __builtin_amdgcn_raw_buffer_store_b32((int)(blk + s), hr, lane * 4, 0, 0);
  constexpr int N = CONTROL ? S + 1 : S;
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  uint32_t v;
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + lane * 4;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  seen[blk * 64 + lane] = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int S, bool CONTROL>
static int run(const uint32_t* cold, long long stride_dw, uint32_t* hot, uint32_t* seen, int blocks, std::vector<uint32_t>& h) {
  hipLaunchKernelGGL((probe<S, CONTROL>), dim3(blocks), dim3(64), 0, 0, cold, stride_dw, hot, seen);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h.data(), seen, (size_t)blocks * 64 * 4, hipMemcpyDeviceToHost));
  long long stale = 0, wrong = 0;
  for (long long b = 0; b < blocks; ++b)
    for (int l = 0; l < 64; ++l) {
      const uint32_t v = h[b * 64 + l], want = (uint32_t)(b * 64 + l) * 2654435761u;
      if (v == kSentinel) ++stale;
      else if (v != want) ++wrong;
    }
  printf("S = %2d stores, wait vmcnt(%2d)%s: %lld of %lld lanes saw the sentinel (load not landed), %lld wrong values\n", S,
         CONTROL ? S + 1 : S, CONTROL ? " [control: no wait for the load]" : "", stale, (long long)blocks * 64, wrong);
  return stale != 0;
}

__global__ void fill(uint32_t* cold, long long stride_dw, int blocks) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < (long long)blocks * 64) cold[(i / 64) * stride_dw + (i % 64)] = (uint32_t)i * 2654435761u;
}

int main() {
  const int blocks = 1 << 18;                    // 262144 waves
  const long long stride_dw = 4096;              // 16 KB apart: 4 GiB of address range, every load its own DRAM page
  uint32_t *cold, *hot, *seen;
  CHECK(hipMalloc(&cold, (size_t)blocks * stride_dw * 4));
  CHECK(hipMalloc(&hot, 64 * 64 * 4));
  CHECK(hipMalloc(&seen, (size_t)blocks * 64 * 4));
  std::vector<uint32_t> h((size_t)blocks * 64);
  int ooo = 0, control = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(fill, dim3(blocks * 64 / 256), dim3(256), 0, 0, cold, stride_dw, blocks);
    CHECK(hipDeviceSynchronize());
    // evict: the fill left the lines in L2 / Infinity Cache; stream 1 GiB of other lines through
    CHECK(hipMemset(seen, 0, (size_t)blocks * 64 * 4));
    ooo += run<1, false>(cold, stride_dw, hot, seen, blocks, h);
    ooo += run<4, false>(cold, stride_dw, hot, seen, blocks, h);
    ooo += run<16, false>(cold, stride_dw, hot, seen, blocks, h);
    ooo += run<40, false>(cold, stride_dw, hot, seen, blocks, h);
    control += run<4, true>(cold, stride_dw, hot, seen, blocks, h);
  }
  printf("verdict: %s%s\n", ooo ? "OUT OF ORDER: a counted wait past younger stores does NOT cover an older LDS-DMA load"
                               : "in order on every run: vmcnt(#younger stores) covered the older LDS-DMA load",
         control ? "" : "  [control never saw a sentinel: the probe cannot tell -- loads landed before the read anyway]");
  return 0;
}
