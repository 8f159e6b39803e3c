// common.hip -- HIP error plumbing, device queries, the persistent kernels' work counters.
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "adt_common.h"

namespace adt {

// set_error / adt_last_error / adt_version live in errors.cpp (host-only translation unit, shared with the sanitizer build)
int set_hip_error(hipError_t e, const char* what) {
  size_t n = 0;
  char* buf = error_buffer(&n);
  std::snprintf(buf, n, "HIP error %d (%s) in %s", static_cast<int>(e), hipGetErrorString(e), what);
  (void)hipGetLastError();   // clear the sticky error so the next call starts clean
  return ADT_EHIP;
}

int device_cu_count(int* n_cu) {
  static std::mutex mu;
  static int cache[64];
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 0 || dev >= 64) return set_error(ADT_EHIP, "device index out of range");
  if (cache[dev] == 0) {
    int v = 0;
    ADT_HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    cache[dev] = v > 0 ? v : 1;
  }
  *n_cu = cache[dev];
  return ADT_OK;
}

// Holds n_wg workgroups (each with lds_bytes of LDS) on the chip for about `micros` microseconds: the test double of a
// communication kernel that occupies CUs while the training kernels run (adt_debug_occupy).
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) smem[0] = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();        // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int sched_counters(void* stream, unsigned** counters) {
  static std::mutex mu;
  static std::map<std::pair<int, void*>, unsigned*> slots;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(dev, stream);
  auto it = slots.find(key);
  if (it == slots.end()) {            // first use of this stream: 8 x 64 bytes of device memory, zeroed before anything can read them
    unsigned* w = nullptr;
    ADT_HIP_TRY(hipMalloc(&w, 512));
    ADT_HIP_TRY(hipMemset(w, 0, 512));
    ADT_HIP_TRY(hipDeviceSynchronize());
    it = slots.emplace(key, w).first;
  }
  *counters = it->second;
  return ADT_OK;
}

}  // namespace adt

extern "C" int adt_debug_occupy(int32_t n_wg, int32_t lds_bytes, int32_t micros, void* stream) {
  using namespace adt;
  if (n_wg <= 0 || lds_bytes < 0 || lds_bytes > 160 * 1024 || micros < 0 || micros > 1000000)
    return set_error(ADT_EINVAL, "adt_debug_occupy: bad arguments");
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(occupy_kernel, dim3(static_cast<unsigned>(n_wg)), dim3(64), static_cast<size_t>(lds_bytes), static_cast<hipStream_t>(stream),
                     static_cast<unsigned long long>(micros) * 100ull);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
