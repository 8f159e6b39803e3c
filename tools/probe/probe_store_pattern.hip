// probe_store_pattern.hip -- does the SHAPE of the GEMM epilogue's stores bound it?  The 256 x 256 tile kernel stores, per wave
// instruction, 8 rows x 128 B (bf16) at the output's row pitch (6144 B at N = 3072): 16 instructions per 16-row pass pair, 256 KB per tile.
// This probe writes the same 388 MB ([63104 x 3072] bf16) with (a) exactly that pattern, tile by tile, one 512-thread workgroup per
// CU walking tiles; (b) the same tiles with a wave instruction covering 2 rows x 512 B; (c) 1 KiB contiguous per wave instruction
// (a memset-like stream); and prints GB/s.  If (a) is far below (c), the epilogue's store shape is worth changing.
// Build: hipcc --offload-arch=gfx950 -O3 probe_store_pattern.hip -o probe_store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kM = 63104, kN = 3072;

template <int kMode>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* out, int tiles_m, int tiles_n) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
  const uint4 v = make_uint4(tid, tid + 1, tid + 2, tid + 3);
  for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
    const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
    if (kMode == 0) {          // the epilogue's shape: wave (wr, wc) owns rows wr*128.., columns wc*64..; 8 rows x 128 B per instruction
      for (int i = 0; i < 8; ++i)
        for (int pass = 0; pass < 2; ++pass) {
          const int row = m0 + wr * 128 + i * 16 + pass * 8 + (lane >> 3), col = n0 + wc * 64 + (lane & 7) * 8;
          if (row < kM) *reinterpret_cast<uint4*>(out + static_cast<long>(row) * kN + col) = v;
        }
    } else if (kMode == 1) {   // 2 rows x 512 B per instruction: wave w owns rows w*32 .. w*32+31 of the tile, all 256 columns
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + wave * 32 + i * 2 + (lane >> 5), col = n0 + (lane & 31) * 8;
        if (row < kM) *reinterpret_cast<uint4*>(out + static_cast<long>(row) * kN + col) = v;
      }
    } else {                   // 1 KiB contiguous per instruction (the tile's bytes as a linear stream)
      const long base = static_cast<long>(t) * 65536;
      for (int i = 0; i < 16; ++i) {
        const long e = base + (static_cast<long>(i) * 512 + tid) * 8;
        if (e + 8 <= static_cast<long>(kM) * kN) *reinterpret_cast<uint4*>(out + e) = v;
      }
    }
  }
}

template <int kMode>
void run(unsigned short* d, const char* name, int grid = 256) {
  const int tm = (kM + 255) / 256, tn = kN / 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store_kernel<kMode>, dim3(grid), dim3(512), 0, 0, d, tm, tn);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_kernel<kMode>, dim3(grid), dim3(512), 0, 0, d, tm, tn);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 20;
  printf("%-44s %3d workgroups  %.3f ms  %.0f GB/s\n", name, grid, ms, static_cast<double>(kM) * kN * 2 / (ms * 1e-3) / 1e9);
}

int main() {
  unsigned short* d;
  if (hipMalloc(&d, static_cast<size_t>(kM + 256) * kN * 2) != hipSuccess) return 1;
  for (int r = 0; r < 2; ++r) {
    run<0>(d, "epilogue shape: 8 rows x 128 B / instruction");
    run<1>(d, "2 rows x 512 B / instruction");
    run<2>(d, "1 KiB contiguous / instruction");
  }
  // is the limit per CU or shared?  the same bytes from fewer workgroups (one per CU): a per-CU limit doubles the time at half the CUs
  for (int grid : {256, 128, 64, 32}) run<0>(d, "epilogue shape", grid);
  return 0;
}
