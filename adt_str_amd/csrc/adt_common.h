// adt_common.h -- error plumbing shared by every translation unit of libadt_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/adt_hip.h"

namespace adt {

// Records `msg` as this thread's last error and returns `code` (see adt_last_error()).
int set_error(int code, const char* msg);
int set_hip_error(hipError_t e, const char* what);
// Number of compute units of the current device (cached per device).
int device_cu_count(int* n_cu);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace adt

#define ADT_HIP_TRY(expr)                                                   \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) return ::adt::set_hip_error(_e, #expr);           \
  } while (0)
