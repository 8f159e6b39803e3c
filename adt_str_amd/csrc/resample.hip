// resample.hip -- K13: polyphase sinc resampler (gfx950).
//
// Stands behind torchaudio.transforms.Resample(orig, new) as the reference calls it with default arguments
// (utils/audio_utils.py:18-20, inference.py:89-90, data_modules/augment_data_with_CLAP.py:56-59): windowed-sinc
// (Hann, lowpass_filter_width 6, rolloff 0.99) kernel bank [new/g][2*width + orig/g], the input padded by (width,
// width + orig/g) zeros, a strided correlation, the result cut to ceil(new * L / orig) samples.  The host builds the
// kernel bank in float64 exactly as torchaudio's _get_sinc_resample_kernel does and passes, per phase, the range of
// taps that are not exact zeros (the window is 0 outside +-6 periods), so an output sample costs ~2 * 6 * max(1, orig/new)
// multiply-adds instead of the dense row.  HBM-bound: 4 bytes read (through L2: neighbouring outputs share their
// windows) and 4 written per sample.  One thread per output sample, fp32 accumulation in tap order.
#include <hip/hip_runtime.h>

#include "adt_common.h"

namespace adt {

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ in, long ld_in, long L_in, const float* __restrict__ bank,
                                                       const int* __restrict__ tap_range, int K, int width, int orig, int neu,
                                                       float* __restrict__ out, long ld_out, long L_out) {
  const long n = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (n >= L_out) return;
  const long clip = blockIdx.y;
  const long blk = n / neu;
  const int p = static_cast<int>(n - blk * neu);
  const int k0 = tap_range[2 * p], k1 = tap_range[2 * p + 1];          // [k0, k1)
  const float* w = bank + static_cast<long>(p) * K;
  const float* x = in + clip * ld_in;
  const long base = blk * orig - width;                                 // padded index k <-> input index base + k
  float acc = 0.f;
  for (int k = k0; k < k1; ++k) {
    const long i = base + k;
    const float v = (i >= 0 && i < L_in) ? x[i] : 0.f;
    acc = fmaf(w[k], v, acc);
  }
  out[clip * ld_out + n] = acc;
}

}  // namespace adt

using namespace adt;

extern "C" int adt_resample_f32(const float* in, int64_t B, int64_t L_in, int64_t ld_in, const float* bank, const int32_t* tap_range,
                                int32_t K, int32_t width, int32_t orig, int32_t neu, float* out, int64_t L_out, int64_t ld_out, void* stream) {
  if (!in || !bank || !tap_range || !out) return set_error(ADT_EINVAL, "adt_resample_f32: null pointer");
  if (B < 0 || L_in < 0 || L_out < 0 || ld_in < L_in || ld_out < L_out) return set_error(ADT_EINVAL, "adt_resample_f32: bad sizes");
  if (orig <= 0 || neu <= 0 || width < 0 || K != 2 * width + orig) return set_error(ADT_ESHAPE, "adt_resample_f32: kernel bank must be [new][2*width + orig]");
  if (B > 65535) return set_error(ADT_ESHAPE, "adt_resample_f32: at most 65535 clips per call");
  // every output block of `neu` samples must exist in the padded input: L_out <= ceil(neu * L_in / orig) (torchaudio's cut)
  if (L_out > (neu * L_in + orig - 1) / orig) return set_error(ADT_ESHAPE, "adt_resample_f32: L_out exceeds ceil(new * L_in / orig)");
  if (B == 0 || L_out == 0) return ADT_OK;
  hipLaunchKernelGGL(resample_kernel, dim3(static_cast<unsigned>((L_out + 255) / 256), static_cast<unsigned>(B)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, ld_in, L_in, bank, tap_range, K, width, orig, neu, out, ld_out, L_out);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
