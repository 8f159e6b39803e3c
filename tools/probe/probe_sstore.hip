// probe_sstore.hip -- does gfx950 execute scalar stores (s_store_dwordx2 + s_dcache_wb), and what do they cost?  The attention forward has
// every dropout decision of a 32 x 32 score block as sixteen 64-bit lane masks in scalar registers (v_cmp results); the one-kernel backward
// wants exactly those words per key (lane = key there), so the cheapest hand-over is a scalar store of the compare result.
//   part 1 (correctness): every wave stores 16 ballots of a lane-dependent predicate by s_store_dwordx2; the host checks all words.
//   part 2 (cost): per wave a loop of { 16 dependent-free v_cmp + s_store_dwordx2, kFill VALU fillers } against the same loop with
//   v_writelane_b32 x 32 + one global_store_dword, and with no store at all; shader-clock ticks per trip.
// Build: hipcc --offload-arch=gfx950 -O3 probe_sstore.hip -o probe_sstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void sstore_check(unsigned long long* out, unsigned seed) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  unsigned long long* dst = out + static_cast<long>(wave) * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned v = (lane * 2654435761u + i * 40503u + wave * 97u + seed) >> 7;
    const unsigned long long m = __builtin_amdgcn_ballot_w64((v & 0xffffu) >= 6554u);
    asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(m), "s"(dst), "i"(8 * i) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}

template <int kMode, int kFill>      // 0: no store, 1: scalar stores, 2: v_writelane + vector store
__global__ __launch_bounds__(256, 2) void sstore_cost(unsigned long long* out, unsigned* vout, unsigned long long* ticks, float* sink, int trips) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  unsigned long long* dst = out + static_cast<long>(wave) * 16 * 64;
  unsigned* vdst = vout + static_cast<long>(wave) * 32 * 64;
  float f[4] = {1.f + lane, 2.f, 3.f, 4.f};
  unsigned x = lane * 2654435761u + wave;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
  for (int it = 0; it < trips; ++it) {
    unsigned w = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x = x * 1664525u + 1013904223u;
      const unsigned long long m = __builtin_amdgcn_ballot_w64((x >> 16) >= 6554u);
      if (kMode == 1) asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(m), "s"(dst + (it & 63) * 16), "i"(8 * i) : "memory");
      if (kMode == 2) {
        asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(w) : "s"(static_cast<unsigned>(m)), "i"(2 * i));
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(w) : "s"(static_cast<unsigned>(m >> 32)), "i"(2 * i + 1));
      }
      if (kMode == 0) asm volatile("" :: "s"(m));
#pragma unroll
      for (int k = 0; k < kFill; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[k & 3]));
    }
    if (kMode == 2 && lane < 32) vdst[(it & 63) * 32 + lane] = w;
  }
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
  if (kMode == 1) asm volatile("s_dcache_wb" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  sink[blockIdx.x * 256 + threadIdx.x] = f[0] + f[1] + f[2] + f[3] + static_cast<float>(x);
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int kMode, int kFill>
void cost(unsigned long long* d_out, unsigned* d_vout, unsigned long long* d_ticks, float* d_sink) {
  const int trips = 512, grid = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((sstore_cost<kMode, kFill>), dim3(grid), dim3(256), 0, 0, d_out, d_vout, d_ticks, d_sink, trips);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((sstore_cost<kMode, kFill>), dim3(grid), dim3(256), 0, 0, d_out, d_vout, d_ticks, d_sink, trips);
  hipEventRecord(e1, 0);
  if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d: launch failed: %s\n", kMode, hipGetErrorString(hipGetLastError())); exit(1); }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t = 0;
  hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost);
  const char* names[3] = {"no store          ", "16 x s_store_dwx2 ", "32 x writelane+st "};
  printf("%s %2d fillers per compare: %7.1f ticks per block of 16 compares (one wave, 2 waves per SIMD); kernel %.1f us\n", names[kMode], kFill,
         (double)t / trips, ms * 1e3);
}

int main() {
  const int n_waves = 256 * 4 * 4;
  unsigned long long* d_out;
  hipMalloc(&d_out, static_cast<size_t>(512) * 4 * 16 * 64 * 8 + n_waves * 16 * 8);
  hipMemset(d_out, 0xff, n_waves * 16 * 8);
  const unsigned seed = 12345u;
  hipLaunchKernelGGL(sstore_check, dim3(n_waves / 4), dim3(256), 0, 0, d_out, seed);
  if (hipDeviceSynchronize() != hipSuccess) { printf("scalar store kernel FAILED: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<unsigned long long> h(n_waves * 16);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int w = 0; w < n_waves; ++w)
    for (int i = 0; i < 16; ++i) {
      unsigned long long want = 0;
      for (int lane = 0; lane < 64; ++lane) {
        const unsigned v = (lane * 2654435761u + i * 40503u + w * 97u + seed) >> 7;
        if ((v & 0xffffu) >= 6554u) want |= 1ull << lane;
      }
      if (h[w * 16 + i] != want) { if (bad < 4) printf("  wave %d word %d: got %016llx want %016llx\n", w, i, h[w * 16 + i], want); ++bad; }
    }
  printf("scalar stores: %ld of %zu words wrong -> %s\n", bad, h.size(), bad ? "NOT USABLE" : "ok");
  unsigned* d_vout; unsigned long long* d_ticks; float* d_sink;
  hipMalloc(&d_vout, static_cast<size_t>(512) * 4 * 32 * 64 * 4); hipMalloc(&d_ticks, 64); hipMalloc(&d_sink, 512 * 256 * 4);
  cost<0, 4>(d_out, d_vout, d_ticks, d_sink);  cost<1, 4>(d_out, d_vout, d_ticks, d_sink);  cost<2, 4>(d_out, d_vout, d_ticks, d_sink);
  cost<0, 12>(d_out, d_vout, d_ticks, d_sink); cost<1, 12>(d_out, d_vout, d_ticks, d_sink); cost<2, 12>(d_out, d_vout, d_ticks, d_sink);
  return bad ? 2 : 0;
}
