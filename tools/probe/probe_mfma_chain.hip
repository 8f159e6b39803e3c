// probe_mfma_chain.hip -- how fast do DEPENDENT v_mfma_f32_32x32x16_bf16 issue on gfx950?  One wave per SIMD (256-thread workgroup, one
// per CU), N products on 1 / 2 / 4 accumulators in turn, accumulators in arch VGPRs ("v") or in AGPRs ("a"), operands in registers.
// Prints shader-clock cycles (s_memtime) per product.  Build: hipcc --offload-arch=gfx950 -O3 probe_mfma_chain.hip -o probe_mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int kAcc, bool kAgpr, int kFill>
__global__ __launch_bounds__(256, 1) void chain(unsigned long long* out, float* sink) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + threadIdx.x + j); b[j] = (short)(0x3f00 + j); }
  f32x16 c[4];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 16; ++i) c[k][i] = 0.f;
  float f = threadIdx.x;
  asm volatile("s_nop 7" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (kAgpr) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c[u % kAcc]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[u % kAcc]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < kFill; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));   // independent vector work between the products
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = f;
  for (int k = 0; k < kAcc; ++k) s += c[k][0] + c[k][15];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int kAcc, bool kAgpr, int kFill>
void run(const char* name, unsigned long long* d_out, float* d_sink) {
  hipLaunchKernelGGL((chain<kAcc, kAgpr, kFill>), dim3(256), dim3(256), 0, 0, d_out, d_sink);
  hipLaunchKernelGGL((chain<kAcc, kAgpr, kFill>), dim3(256), dim3(256), 0, 0, d_out, d_sink);
  hipDeviceSynchronize();
  unsigned long long t = 0;
  hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
  printf("%-44s %6.1f cycles per MFMA\n", name, (double)t / (64.0 * 16.0));
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 64); hipMalloc(&d_sink, 256 * 256 * 4);
  run<1, false, 0>("VGPR acc, 1 chain", d_out, d_sink);
  run<2, false, 0>("VGPR acc, 2 chains", d_out, d_sink);
  run<4, false, 0>("VGPR acc, 4 chains", d_out, d_sink);
  run<1, true, 0>("AGPR acc, 1 chain", d_out, d_sink);
  run<2, true, 0>("AGPR acc, 2 chains", d_out, d_sink);
  run<4, true, 0>("AGPR acc, 4 chains", d_out, d_sink);
  run<1, false, 4>("VGPR acc, 1 chain, 4 v_fma between", d_out, d_sink);
  run<4, false, 4>("VGPR acc, 4 chains, 4 v_fma between", d_out, d_sink);
  run<1, true, 4>("AGPR acc, 1 chain, 4 v_fma between", d_out, d_sink);
  run<4, true, 8>("AGPR acc, 4 chains, 8 v_fma between", d_out, d_sink);
  return 0;
}
