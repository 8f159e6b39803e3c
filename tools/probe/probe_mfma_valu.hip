// probe_mfma_valu.hip -- do the vector ALU and the matrix pipe of a gfx950 SIMD overlap ACROSS the two waves that share it?  Each wave
// runs { one v_mfma_f32_32x32x16_bf16 (dependent chain), kFill vector instructions on four rotating registers } in a loop; one or two
// waves per SIMD (256 or 512 threads, one workgroup per CU).  Prints shader-clock cycles (s_memtime) per loop trip of one wave: perfect
// overlap with two waves is max(64, 2 (8 + 4 kFill)); no overlap is 2 (32 + 4 kFill).  kExp: every second filler is a v_exp_f32.
// Build: hipcc --offload-arch=gfx950 -O3 probe_mfma_valu.hip -o probe_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int kThreads, int kFill, bool kExp>
__global__ __launch_bounds__(kThreads, 1) void mix(unsigned long long* out, float* sink) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + j); }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  float f[4] = {1.f + threadIdx.x, 2.f, 3.f, 4.f};
  asm volatile("s_nop 7" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
  for (int it = 0; it < 128; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < kFill; ++k) {
        if (kExp && (k & 1)) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k & 3]));
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[k & 3]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  sink[blockIdx.x * kThreads + threadIdx.x] = f[0] + f[1] + f[2] + f[3] + c[0] + c[15];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int kThreads, int kFill, bool kExp>
void run(unsigned long long* d_out, float* d_sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mix<kThreads, kFill, kExp>), dim3(256), dim3(kThreads), 0, 0, d_out, d_sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((mix<kThreads, kFill, kExp>), dim3(256), dim3(kThreads), 0, 0, d_out, d_sink);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t = 0;
  hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
  // (the kernel is the loop: its duration over the stamps' difference is the stamp clock; 1024 products x 32768 FLOP per wave)
  printf("%d wave(s) per SIMD, %2d fillers%s: %6.1f ticks per MFMA of one wave; kernel %.1f us = %.2f ticks/ns; %.0f TFLOP/s\n", kThreads / 256, kFill,
         kExp ? " (half v_exp_f32)" : "", (double)t / (128.0 * 8.0), ms * 1e3, (double)t / (ms * 1e6), 256.0 * (kThreads / 64) * 1024 * 65536.0 / (ms * 1e-3) / 1e12);
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 64); hipMalloc(&d_sink, 256 * 512 * 4);
  run<256, 0, false>(d_out, d_sink);  run<512, 0, false>(d_out, d_sink);
  run<256, 4, false>(d_out, d_sink);  run<512, 4, false>(d_out, d_sink);
  run<256, 8, false>(d_out, d_sink);  run<512, 8, false>(d_out, d_sink);
  run<256, 12, false>(d_out, d_sink); run<512, 12, false>(d_out, d_sink);
  run<256, 16, false>(d_out, d_sink); run<512, 16, false>(d_out, d_sink);
  run<256, 8, true>(d_out, d_sink);   run<512, 8, true>(d_out, d_sink);
  run<256, 12, true>(d_out, d_sink);  run<512, 12, true>(d_out, d_sink);
  return 0;
}
