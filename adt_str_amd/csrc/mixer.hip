// mixer.hip -- K2: batched one-shot drum mixer (gfx950).
//
// Stands behind SynthDrum.__call__ / drum_rendering / VolumeMixer.instrument_mixer
// (reference modules/synthetiser.py:214-239, 149-156, 255-292), which render one clip
// at a time in a Python loop over notes.  Here a whole batch of clips is rendered from a
// flat, HBM-resident one-shot bank in three launches:
//
//   1. mix_peak_kernel    one workgroup per note: peak = max |main*(1-m) + m*sub| over the
//                         zero-padded pair of one-shots (synthetiser.py:218-225)
//   2. mix_render_kernel  one workgroup per (clip, 1024-sample tile): for every sample, notes are
//                         accumulated per track in note order, tracks are combined with their
//                         class gain in track order (synthetiser.py:226-237, 151-153); the tile's
//                         |max| goes to the clip's peak with one atomic per workgroup
//   3. mix_scale_kernel   out = wav / clip_peak * clip_gain  (synthetiser.py:142-143,156)
//
// Arithmetic is written with explicitly rounded mul/add/div in the reference's operation
// order (no FMA contraction), so results match the fp32 CPU path to the last bit wherever
// the accumulation order is the same.  HBM-bound: bytes = shots read + 3 x output.
#include <hip/hip_runtime.h>

#include "adt_common.h"

namespace adt {

constexpr int kMixThreads = 256;
constexpr int kMixTile = 1024;     // samples per workgroup in the render / scale passes

__device__ __forceinline__ float mixed_sample(const float* __restrict__ bank, long main_off, int main_len,
                                              long sub_off, int sub_len, int idx, float one_minus_m, float m) {
  const float a = idx < main_len ? bank[main_off + idx] : 0.0f;     // pad_sequence zero padding
  const float b = idx < sub_len ? bank[sub_off + idx] : 0.0f;
  return __fadd_rn(__fmul_rn(a, one_minus_m), __fmul_rn(m, b));
}

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  v = red[0];
#pragma unroll
  for (int w = 1; w < kMixThreads / 64; ++w) v = fmaxf(v, red[w]);
  __syncthreads();
  return v;
}

// NaN-propagating |x| max on the raw bits: non-negative floats order like unsigned ints and a NaN
// (0x7fc00000) sorts above every finite value, as torch.max does (it returns NaN).
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

__global__ __launch_bounds__(kMixThreads) void mix_peak_kernel(const float* __restrict__ bank,
                                                               const int64_t* __restrict__ bank_off,
                                                               const adt_note* __restrict__ notes, float* __restrict__ peak) {
  __shared__ unsigned red[kMixThreads / 64];
  const adt_note n = notes[blockIdx.x];
  const long mo = bank_off[n.main_shot], so = bank_off[n.sub_shot];
  const int ml = static_cast<int>(bank_off[n.main_shot + 1] - mo), sl = static_cast<int>(bank_off[n.sub_shot + 1] - so);
  const int len = ml > sl ? ml : sl;
  unsigned best = 0;
  for (int i = threadIdx.x; i < len; i += kMixThreads) {
    const unsigned b = abs_bits(mixed_sample(bank, mo, ml, so, sl, i, n.one_minus_mixup, n.mixup));
    best = b > best ? b : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(best, o); best = t > best ? t : best; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kMixThreads / 64; ++w) best = red[w] > best ? red[w] : best;
    peak[blockIdx.x] = __uint_as_float(best);
  }
}

struct MixArgs {
  const float* bank; const int64_t* bank_off; const adt_note* notes; const int32_t* clip_note_off;
  const int32_t* clip_len; const float* clip_gain; const float* peak; unsigned* clip_peak;
  float* out; long ld_out; int n_clips; int width;
};

__global__ __launch_bounds__(kMixThreads) void mix_render_kernel(MixArgs a) {
  __shared__ unsigned red[kMixThreads / 64];
  const int clip = blockIdx.y;
  const int tile0 = blockIdx.x * kMixTile;
  const int n0 = a.clip_note_off[clip], n1 = a.clip_note_off[clip + 1];
  const int clen = a.clip_len[clip];
  float wav[4] = {0.f, 0.f, 0.f, 0.f};
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int cur_track = -1;
  float cur_gain = 0.f;
  for (int i = n0; i < n1; ++i) {
    const adt_note n = a.notes[i];                         // block-uniform
    if (n.track != cur_track) {                            // notes arrive grouped by track, in track order
      if (cur_track >= 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { wav[j] = __fadd_rn(wav[j], __fmul_rn(acc[j], cur_gain)); acc[j] = 0.f; }
      }
      cur_track = n.track; cur_gain = n.track_gain;
    }
    const long mo = a.bank_off[n.main_shot], so = a.bank_off[n.sub_shot];
    const int ml = static_cast<int>(a.bank_off[n.main_shot + 1] - mo), sl = static_cast<int>(a.bank_off[n.sub_shot + 1] - so);
    const int len = ml > sl ? ml : sl;
    if (n.start >= tile0 + kMixTile || n.start + len <= tile0) continue;
    const float pk = a.peak[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = tile0 + threadIdx.x + kMixThreads * j;
      const int idx = s - n.start;
      if (idx >= 0 && idx < len && s < clen) {
        const float o = mixed_sample(a.bank, mo, ml, so, sl, idx, n.one_minus_mixup, n.mixup);
        acc[j] = __fadd_rn(acc[j], __fmul_rn(__fdiv_rn(o, pk), n.vol));
      }
    }
  }
  unsigned best = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (cur_track >= 0) wav[j] = __fadd_rn(wav[j], __fmul_rn(acc[j], cur_gain));
    const int s = tile0 + threadIdx.x + kMixThreads * j;
    if (s < a.width) a.out[clip * a.ld_out + s] = wav[j];
    const unsigned b = abs_bits(wav[j]);
    best = b > best ? b : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(best, o); best = t > best ? t : best; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kMixThreads / 64; ++w) best = red[w] > best ? red[w] : best;
    if (best) atomicMax(&a.clip_peak[clip], best);
  }
}

__global__ __launch_bounds__(kMixThreads) void mix_scale_kernel(MixArgs a) {
  const int clip = blockIdx.y;
  if (a.clip_note_off[clip] == a.clip_note_off[clip + 1]) return;      // empty clip stays all-zero (synthetiser.py:257-258)
  const float pk = __uint_as_float(a.clip_peak[clip]);
  const float g = a.clip_gain[clip];
  const int clen = a.clip_len[clip];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = blockIdx.x * kMixTile + threadIdx.x + kMixThreads * j;
    if (s < a.width && s < clen) {
      float* p = a.out + clip * a.ld_out + s;
      *p = __fmul_rn(__fdiv_rn(*p, pk), g);
    }
  }
}

}  // namespace adt

extern "C" size_t adt_mix_workspace_bytes(int64_t n_notes, int64_t n_clips) {
  if (n_notes < 0 || n_clips < 0) return 0;
  return static_cast<size_t>(n_notes + n_clips) * 4 + 16;
}

extern "C" int adt_mix_render_f32(const float* bank, const int64_t* bank_off, int64_t n_shots,
                                  const adt_note* notes, int64_t n_notes, const int32_t* clip_note_off,
                                  const int32_t* clip_len, const float* clip_gain, int64_t n_clips, int64_t width,
                                  float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream) {
  return adt_mix_render_fx_f32(bank, bank_off, n_shots, notes, n_notes, clip_note_off, clip_len, clip_gain, n_clips, width, nullptr, 0, out,
                               ld_out, ws, ws_bytes, stream);
}

extern "C" int adt_mix_render_fx_f32(const float* bank, const int64_t* bank_off, int64_t n_shots,
                                     const adt_note* notes, int64_t n_notes, const int32_t* clip_note_off,
                                     const int32_t* clip_len, const float* clip_gain, int64_t n_clips, int64_t width,
                                     const adt_fx_params* fx, int32_t sample_rate,
                                     float* out, int64_t ld_out, void* ws, size_t ws_bytes, void* stream) {
  using namespace adt;
  if (!out || !clip_note_off || !clip_len || !clip_gain) return set_error(ADT_EINVAL, "adt_mix_render_f32: null pointer");
  if (n_notes > 0 && (!bank || !bank_off || !notes)) return set_error(ADT_EINVAL, "adt_mix_render_f32: null bank/notes");
  if (n_notes < 0 || n_clips < 0 || width < 0 || ld_out < width || n_shots < 0)
    return set_error(ADT_EINVAL, "adt_mix_render_f32: negative size or ld_out < width");
  if (n_clips > 65535 || width >= (1ll << 30)) return set_error(ADT_ESHAPE, "adt_mix_render_f32: at most 65535 clips of < 2^30 samples");
  if (ws_bytes < adt_mix_workspace_bytes(n_notes, n_clips) || (!ws && ws_bytes))
    return set_error(ADT_EINVAL, "adt_mix_render_f32: workspace too small (see adt_mix_workspace_bytes)");
  if (n_clips == 0 || width == 0) return ADT_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned* clip_peak = reinterpret_cast<unsigned*>(ws);
  float* peak = reinterpret_cast<float*>(clip_peak + n_clips);
  ADT_HIP_TRY(hipMemsetAsync(clip_peak, 0, static_cast<size_t>(n_clips) * 4, st));
  if (n_notes > 0) {
    hipLaunchKernelGGL(mix_peak_kernel, dim3(static_cast<unsigned>(n_notes)), dim3(kMixThreads), 0, st, bank, bank_off, notes, peak);
  }
  MixArgs a{bank, bank_off, notes, clip_note_off, clip_len, clip_gain, peak, clip_peak, out, ld_out,
            static_cast<int>(n_clips), static_cast<int>(width)};
  const dim3 grid(static_cast<unsigned>((width + kMixTile - 1) / kMixTile), static_cast<unsigned>(n_clips));
  hipLaunchKernelGGL(mix_render_kernel, grid, dim3(kMixThreads), 0, st, a);
  if (fx) {                            // FX works on the un-normalised mix and replaces the peaks of the clips it touched
    if (int rc = launch_fx_chain(out, ld_out, clip_len, fx, static_cast<int>(n_clips), sample_rate, static_cast<int>(width), clip_peak, st)) return rc;
  }
  hipLaunchKernelGGL(mix_scale_kernel, grid, dim3(kMixThreads), 0, st, a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
