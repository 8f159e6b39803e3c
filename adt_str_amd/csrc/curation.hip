// curation.hip -- K12: cosine similarity of every sample embedding against the class-mean embeddings
// and the per-sample best class (gfx950).
//
// Stands behind the similarity loop of the reference's curation driver
// (data_modules/augment_data_with_CLAP.py:139-151: F.cosine_similarity per class, 4.8 M Python tuples at
// 100 k samples) and the "first occurrence wins" assignment (:182-193), which is a per-sample argmax with
// the lowest class position winning ties.  HBM-bound: 4*D bytes per sample are read once; the <= 64 class
// vectors live in LDS.  One wave per sample, fp32 throughout.
#include <hip/hip_runtime.h>

#include "adt_common.h"

namespace adt {

constexpr int kCurThreads = 256;
constexpr int kCurMaxRefFloats = 36 * 1024;          // 144 KiB of LDS for the class vectors

// kRow > 0: D <= 64 * kRow, the sample's row stays in registers (one global read per element instead of one per class; same summation
// order as the generic form: element lane + 64 k in turn, then the xor tree).  The class vectors go to LDS 16 bytes at a time when they can.
template <int kRow>
__global__ __launch_bounds__(kCurThreads) void cosine_argmax_kernel(const float* __restrict__ emb, long ld, const float* __restrict__ refs,
                                                                    int N, int D, int C, float eps, int* __restrict__ best_class,
                                                                    float* __restrict__ best_score, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* r = sm;                        // [C][D]
  float* rnorm2 = sm + C * D;           // [C]
  if (((C * D) & 3) == 0 && (reinterpret_cast<uintptr_t>(refs) & 15) == 0) {
    for (int i = threadIdx.x; i < (C * D) / 4; i += kCurThreads) reinterpret_cast<float4*>(r)[i] = reinterpret_cast<const float4*>(refs)[i];
  } else {
    for (int i = threadIdx.x; i < C * D; i += kCurThreads) r[i] = refs[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = wave; c < C; c += kCurThreads / 64) {
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += r[c * D + i] * r[c * D + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) rnorm2[c] = s;
  }
  __syncthreads();
  const int waves_total = gridDim.x * (kCurThreads / 64);
  for (int n = blockIdx.x * (kCurThreads / 64) + wave; n < N; n += waves_total) {
    const float* x = emb + static_cast<long>(n) * ld;
    float xv[kRow > 0 ? kRow : 1];
    float xx = 0.f;
    if (kRow > 0) {
#pragma unroll
      for (int k = 0; k < kRow; ++k) {
        const int i = lane + 64 * k;
        xv[k] = i < D ? x[i] : 0.f;
        if (i < D) xx += xv[k] * xv[k];
      }
    } else {
      for (int i = lane; i < D; i += 64) { const float v = x[i]; xx += v * v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xx += __shfl_xor(xx, o);
    float best = -3.0e38f;
    int bc = 0;
    // four classes at a time: four independent chains of LDS reads and cross-lane sums in flight instead of one (a wave is alone with its row);
    // each class's sum is formed in the same order as before, and the classes are compared in order
    for (int c0 = 0; c0 < C; c0 += 4) {
      float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u < C ? c0 + u : C - 1;
        if (kRow > 0) {
#pragma unroll
          for (int k = 0; k < kRow; ++k) {
            const int i = lane + 64 * k;
            if (i < D) dot[u] += xv[k] * r[c * D + i];
          }
        } else {
          for (int i = lane; i < D; i += 64) dot[u] += x[i] * r[c * D + i];
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int u = 0; u < 4; ++u) dot[u] += __shfl_xor(dot[u], o);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u;
        if (c < C) {
          const float cs = dot[u] / sqrtf(fmaxf(xx * rnorm2[c], eps * eps));
          if (scores && lane == 0) scores[static_cast<long>(n) * C + c] = cs;
          if (cs > best) { best = cs; bc = c; }          // strict: the lowest class position wins ties
        }
      }
    }
    if (lane == 0) { best_class[n] = bc; best_score[n] = best; }
  }
}

}  // namespace adt

extern "C" int adt_cosine_argmax_f32(const float* emb, int64_t ld, const float* refs, int64_t N, int64_t D, int64_t C, float eps,
                                     int32_t* best_class, float* best_score, float* scores, void* stream) {
  using namespace adt;
  if (!emb || !refs || !best_class || !best_score) return set_error(ADT_EINVAL, "adt_cosine_argmax_f32: null pointer");
  if (N < 0 || D <= 0 || C <= 0 || ld < D) return set_error(ADT_EINVAL, "adt_cosine_argmax_f32: bad sizes");
  if (C * D + C > kCurMaxRefFloats) return set_error(ADT_ESHAPE, "adt_cosine_argmax_f32: class vectors exceed the LDS budget (C*D + C <= 36864 floats)");
  if (N == 0) return ADT_OK;
  const int lds = static_cast<int>((C * D + C) * sizeof(float));
  static thread_local int done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (done_for != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cosine_argmax_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kCurMaxRefFloats * 4));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cosine_argmax_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kCurMaxRefFloats * 4));
    done_for = dev;
  }
  int n_cu = 0;
  if (int rc = device_cu_count(&n_cu)) return rc;
  long blocks = (N + 3) / 4;
  if (blocks > n_cu) blocks = n_cu;
  if (D <= 512)
    hipLaunchKernelGGL(cosine_argmax_kernel<8>, dim3(static_cast<unsigned>(blocks)), dim3(kCurThreads), lds, static_cast<hipStream_t>(stream),
                       emb, ld, refs, static_cast<int>(N), static_cast<int>(D), static_cast<int>(C), eps, best_class, best_score, scores);
  else
    hipLaunchKernelGGL(cosine_argmax_kernel<0>, dim3(static_cast<unsigned>(blocks)), dim3(kCurThreads), lds, static_cast<hipStream_t>(stream),
                       emb, ld, refs, static_cast<int>(N), static_cast<int>(D), static_cast<int>(C), eps, best_class, best_score, scores);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
