// common.hip -- version, thread-local error string, device queries.
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "adt_common.h"

namespace adt {

static thread_local char g_err[512] = "";

int set_error(int code, const char* msg) {
  std::snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int set_hip_error(hipError_t e, const char* what) {
  std::snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", static_cast<int>(e), hipGetErrorString(e), what);
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

extern "C" int adt_version(void) { return 8; }
extern "C" const char* adt_last_error(void) { return adt::g_err; }
