// adt_common.h -- error plumbing shared by every translation unit of libadt_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/adt_hip.h"

namespace adt {

// Records `msg` as this thread's last error and returns `code` (see adt_last_error()).
int set_error(int code, const char* msg);
char* error_buffer(size_t* size);          // this thread's message buffer (errors.cpp)
int set_hip_error(hipError_t e, const char* what);
// Number of compute units of the current device (cached per device).
int device_cu_count(int* n_cu);
// fx.hip: reverb / compressor / limiter over the un-normalised clips flagged in fx[], then the new clip peaks (mixer.hip calls it)
int launch_fx_chain(float* wav, long ld, const int32_t* clip_len, const adt_fx_params* fx, int n_clips, int sample_rate, int width,
                    unsigned* clip_peak, hipStream_t st);
// Work counters of the persistent GEMMs on (current device, stream): eight device words, one per XCD group, 64 bytes apart,
// allocated and zeroed on first use.  They are zero whenever no such kernel runs on the stream: each launch hands out a known
// number of tickets per counter and the workgroup that draws the last one resets it (gemm.hip), so nothing is tracked here.
int sched_counters(void* stream, unsigned** counters);

// gemm.hip: out[m, n] = alpha * sum over s of slabs[s][m, n] in slab order (split-K partials; N % 4 == 0)
void launch_reduce_slabs(const float* slabs, int splits, long mn, int N, float alpha, float* out, long ldc, hipStream_t st);

// nn_ops.hip: out[c] = sum over the n_part rows of partial[n_part][width] (fixed order)
void launch_reduce_partials(const float* partial, int n_part, int width, float* out, hipStream_t st);

// nn_ops.hip, deferred second-stage reductions (adt_reduce_queue_*).  A producer asks for its partial-sum buffer with
// reduce_queue_slice(): non-null while a queue is open on `st` and its arena has room -- the producer then writes its partials
// there and hands the reduction to reduce_queue_push() instead of launching it; null: use the caller's workspace and reduce now.
float* reduce_queue_slice(size_t bytes, hipStream_t st);
// out_k[c] = sum over n_part rows of partial[n_part][width], column k * D + c (out1 / out2 may be null when width == D);
// n_part == 0 stores zeros.  `partial` must come from reduce_queue_slice (or be null with n_part == 0).
int reduce_queue_push(const float* partial, int n_part, int width, float* out0, float* out1, float* out2, int D);
bool reduce_queue_open(hipStream_t st);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace adt

#define ADT_HIP_TRY(expr)                                                   \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) return ::adt::set_hip_error(_e, #expr);           \
  } while (0)
