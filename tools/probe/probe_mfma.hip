// probe_mfma.hip -- verifies on real gfx950 hardware the MFMA / transposed-LDS-read lane
// maps the GEMM and attention kernels rely on (cdna_hip_programming.md section 3, T10).
// Exact small-integer data, asymmetric operands.  Prints one OK/FAIL line per check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

inline unsigned short f2bf(float f) { unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

// C[16x16] = A[16x32] * B[32x16]; A row-major [16][32], B row-major [32][16]
__global__ void k_16x16x32(const unsigned short* A, const unsigned short* B, float* C) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// C[32x32] = A[32x16] * B[16x32]
__global__ void k_32x32x16(const unsigned short* A, const unsigned short* B, float* C) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j];
    b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// ds_read_b64_tr_b16: LDS holds T[32 rows(k)][16 cols(n)] bf16 row-major (32-byte rows).
// Each 16-lane group g reads rows 8g'..: lane 4q+p of a group supplies the address of row (r0+q), cols 4p..4p+3;
// lane i of the group receives column i of the 4 rows (row q in element q).
__global__ void k_tr(const unsigned short* T, unsigned short* out /*[64][4]*/) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32 * 16];
  const int l = threadIdx.x;
  for (int i = l; i < 32 * 16; i += 64) lds[i] = T[i];
  __syncthreads();
  const int g = l >> 4, t = l & 15, q = t >> 2, p = t & 3;
  const int r0 = 4 * g;                       // group g reads rows 4g..4g+3
  unsigned addr = (unsigned)(((r0 + q) * 16 + 4 * p) * 2);
  addr += (unsigned)(size_t)lds;              // LDS base (usually 0)
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)(v >> (16 * e));
}

// v_cvt_pk_bf16_f32 and permlane32_swap sanity
__global__ void k_misc(float* fin, unsigned* o1, unsigned* o2) {
  const int l = threadIdx.x;
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(fin[2 * l]), "v"(fin[2 * l + 1]));
  o1[l] = r;
  unsigned a = 1000 + l, b = 2000 + l;
  auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o2[2 * l] = sw[0]; o2[2 * l + 1] = sw[1];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, CUs %d, clock %d kHz, LDS/block %zu, L2 %d\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate,
         prop.sharedMemPerBlock, prop.l2CacheSize);
  srand(1);
  {
    std::vector<unsigned short> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16), C(256), R(256, 0.f);
    for (int i = 0; i < 512; ++i) { Af[i] = (float)(rand() % 7 - 3); Bf[i] = (float)(rand() % 5 - 2); A[i] = f2bf(Af[i]); B[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += Af[i * 32 + k] * Bf[k * 16 + j];
    unsigned short *dA, *dB; float* dC;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    k_16x16x32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
    printf("mfma_16x16x32_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  }
  {
    std::vector<unsigned short> A(512), B(512); std::vector<float> Af(512), Bf(512), C(1024), R(1024, 0.f);
    for (int i = 0; i < 512; ++i) { Af[i] = (float)(rand() % 7 - 3); Bf[i] = (float)(rand() % 5 - 2); A[i] = f2bf(Af[i]); B[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
    unsigned short *dA, *dB; float* dC;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    k_32x32x16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != R[i];
    printf("mfma_32x32x16_bf16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  }
  {
    std::vector<unsigned short> T(512), O(256);
    for (int i = 0; i < 512; ++i) T[i] = (unsigned short)i;      // T[row][col] = row*16 + col
    unsigned short *dT, *dO; CK(hipMalloc(&dT, 1024)); CK(hipMalloc(&dO, 512));
    CK(hipMemcpy(dT, T.data(), 1024, hipMemcpyHostToDevice));
    k_tr<<<1, 64>>>(dT, dO); CK(hipDeviceSynchronize());
    CK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
      const int g = l >> 4, i = l & 15;
      const int expect = (4 * g + e) * 16 + i;                   // lane i gets column i, row r0+e in element e
      bad += O[l * 4 + e] != expect;
    }
    printf("ds_read_b64_tr_b16 map: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    if (bad) { for (int l = 0; l < 20; ++l) printf("  lane %d: %d %d %d %d\n", l, O[l*4], O[l*4+1], O[l*4+2], O[l*4+3]); }
  }
  {
    std::vector<float> F(128); std::vector<unsigned> o1(64), o2(128);
    for (int i = 0; i < 128; ++i) F[i] = (float)(i + 1) * 0.5f;
    float* dF; unsigned *d1, *d2; CK(hipMalloc(&dF, 512)); CK(hipMalloc(&d1, 256)); CK(hipMalloc(&d2, 512));
    CK(hipMemcpy(dF, F.data(), 512, hipMemcpyHostToDevice));
    k_misc<<<1, 64>>>(dF, d1, d2); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o1.data(), d1, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), d2, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += o1[l] != ((unsigned)f2bf(F[2 * l]) | ((unsigned)f2bf(F[2 * l + 1]) << 16));
    printf("v_cvt_pk_bf16_f32 (lo=src0, hi=src1): %s\n", bad ? "FAIL" : "OK");
    // permlane32_swap(a,b): lanes 32-63 of vdst(a) swap with lanes 0-31 of src(b)
    bad = 0;
    for (int l = 0; l < 64; ++l) {
      unsigned ea = l < 32 ? 1000 + l : 2000 + (l - 32), eb = l < 32 ? 1000 + (l + 32) : 2000 + l;
      bad += (o2[2 * l] != ea) || (o2[2 * l + 1] != eb);
    }
    printf("permlane32_swap semantics: %s\n", bad ? "FAIL" : "OK");
    if (bad) for (int l = 0; l < 64; l += 9) printf("  lane %d: %u %u\n", l, o2[2*l], o2[2*l+1]);
  }
  return 0;
}
