// logmel.hip -- K1: fused STFT -> power -> banded mel -> log/clamp/scale -> trim (gfx950).
//
// Stands behind ComputeMelSpectrogram.forward (reference model.py:81-97).
//
// Launch shape: persistent, one 512-thread workgroup (8 waves) per CU.  Each wave owns
// a 16,640-byte LDS buffer and walks (clip, frame-pair) work items with a grid stride;
// the workgroup shares the half-circle twiddle table (8 KiB) and the non-zero mel
// weights (<= 9 KiB).  150 KiB of the CU's 160 KiB LDS are used, so occupancy is
// 2 waves per SIMD, set by LDS.  The waveform is read straight from global memory
// (each sample is touched by ~n_fft/hop = 12.8 overlapping frames, which L2 absorbs);
// the output rows are staged in LDS and written as whole 512-byte rows (float4/lane).
// Phase bodies and index maps: logmel_phases.h.
#include <hip/hip_runtime.h>

#include "adt_common.h"
#include "logmel_phases.h"

namespace adt {

constexpr int kWavesPerBlock = 8;
constexpr int kThreads = 64 * kWavesPerBlock;
constexpr int kMaxMelNnz = 2304;
constexpr size_t kLdsMelw = 1024 * sizeof(cf);
constexpr size_t kLdsBufs = kLdsMelw + kMaxMelNnz * sizeof(float);
constexpr size_t kLdsTotal = kLdsBufs + kWavesPerBlock * kBufElems * sizeof(cf);   // 150,528 B
static_assert(kLdsTotal <= 160 * 1024, "LDS budget");

struct LogmelArgs {
  const float* wave; long n_clips; int n_samples; long ld_wave;
  int hop; int frame_lo; int n_out; int pairs_per_clip;
  const float* window; const int4* mel_meta; const float* mel_w; int n_mels; int mel_nnz;
  float log_eps, clamp_lo, clamp_hi;
  float* out;
  long n_items; int n_iter;
};

// Every LDS exchange below is private to one wave (buf is the wave's own buffer), so a
// workgroup barrier is not needed: the LDS unit executes one wave's DS instructions in issue
// order, and this fence/wave_barrier pair keeps the compiler from moving them across the
// phase boundary.  Waves of a workgroup therefore drift apart and overlap LDS with VALU phases.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kThreads) void logmel_kernel(LogmelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* tw = reinterpret_cast<cf*>(smem);                                  // [1024]
  float* melw = reinterpret_cast<float*>(smem + kLdsMelw);               // [kMaxMelNnz]
  cf* bufs = reinterpret_cast<cf*>(smem + kLdsBufs);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> SGPR addressing
  cf* buf = bufs + wave * kBufElems;

  // shared tables: W_2048^j = exp(-2*pi*i*j/2048), j < 1024; mel weights
  for (int j = tid; j < 1024; j += kThreads) {
    float s, c;
    sincospif(static_cast<float>(j) * (1.0f / 1024.0f), &s, &c);
    tw[j] = cf{c, -s};
  }
  for (int j = tid; j < a.mel_nnz; j += kThreads) melw[j] = a.mel_w[j];

  // per-lane window values of the two pass-1 items (m = lane + 64*it, n1 = 0..15)
  float win[2][16];
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) win[it][n1] = a.window[lane + 64 * it + 128 * n1];

  // per-lane constants: the mel bands of the 8 mel items, packed lo | cnt << 11 | off << 18
  const int g = lane >> 2, s = lane & 3;
  unsigned mband[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = g + 16 * i;
    int4 m = (j < a.n_mels) ? a.mel_meta[j] : make_int4(0, 0, 0, 0);
    mband[i] = static_cast<unsigned>(m.x) | (static_cast<unsigned>(m.y) << 11) | (static_cast<unsigned>(m.z) << 18);
  }
  __syncthreads();

  const long total_waves = static_cast<long>(gridDim.x) * kWavesPerBlock;
  const long first = static_cast<long>(blockIdx.x) * kWavesPerBlock + wave;
  const int halfn = kNfft / 2;

  const int lane_id = lane;
  for (int iter = 0; iter < a.n_iter; ++iter) {
    // Re-derive every per-lane LDS address inside the iteration: left to LICM, the ~150
    // loop-invariant twiddle/buffer addresses are hoisted and then spilled to scratch.
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
    const long item = first + static_cast<long>(iter) * total_waves;
    const bool active = item < a.n_items;             // wave-uniform
    long clip_i = 0; int pair = 0;
    if (active) { clip_i = item / a.pairs_per_clip; pair = static_cast<int>(item - clip_i * a.pairs_per_clip); }
    const int f0 = 2 * pair;                           // first output frame of the pair
    const bool has1 = (f0 + 1) < a.n_out;
    const float* clip = a.wave + clip_i * a.ld_wave;
    const int base0 = (a.frame_lo + f0) * a.hop - halfn;
    const int base1 = base0 + a.hop;
    const bool interior = base0 >= 0 && (base1 + kNfft) <= a.n_samples;

    if (active) {
      if (interior) {
        pass1<true>(lane, 0, clip, a.n_samples, base0, base1, has1, win[0], tw, buf);
        __builtin_amdgcn_sched_barrier(0);
        pass1<true>(lane, 1, clip, a.n_samples, base0, base1, has1, win[1], tw, buf);
      } else {
        pass1<false>(lane, 0, clip, a.n_samples, base0, base1, has1, win[0], tw, buf);
        __builtin_amdgcn_sched_barrier(0);
        pass1<false>(lane, 1, clip, a.n_samples, base0, base1, has1, win[1], tw, buf);
      }
    }
    wave_sync();
    if (active) { pass2(lane, 0, tw, buf); __builtin_amdgcn_sched_barrier(0); pass2(lane, 1, tw, buf); }
    wave_sync();
    cf z[4][8];
    if (active) {
#pragma unroll
      for (int it = 0; it < 4; ++it) pass3_load(lane, it, buf, z[it]);
    }
    wave_sync();
    if (active) {
#pragma unroll
      for (int it = 0; it < 4; ++it) pass3_store(lane, it, z[it], buf);
    }
    wave_sync();
    if (active) untangle(lane, buf);
    wave_sync();
    if (active) {
      float* stage = reinterpret_cast<float*>(buf + kStageBase);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        cf acc = mel_partial(s, mband[i] & 2047u, (mband[i] >> 11) & 127u, mband[i] >> 18, melw, buf);
        acc.x += __shfl_xor(acc.x, 1); acc.y += __shfl_xor(acc.y, 1);
        acc.x += __shfl_xor(acc.x, 2); acc.y += __shfl_xor(acc.y, 2);
        const int j = g + 16 * i;
        if (s == 0 && j < a.n_mels) {
          stage[j] = post(acc.x, a.log_eps, a.clamp_lo, a.clamp_hi);
          stage[a.n_mels + j] = post(acc.y, a.log_eps, a.clamp_lo, a.clamp_hi);
        }
      }
    }
    wave_sync();
    if (active) {
      const float4* stage4 = reinterpret_cast<const float4*>(buf + kStageBase);
      const int quads = (has1 ? 2 : 1) * a.n_mels / 4;   // rows f0, f0+1 are contiguous in out
      float4* dst = reinterpret_cast<float4*>(a.out + (clip_i * a.n_out + f0) * a.n_mels);
      if (lane < quads) dst[lane] = stage4[lane];
    }
    wave_sync();
  }
}

}  // namespace adt

extern "C" int adt_logmel_f32(const float* wave, int64_t n_clips, int64_t n_samples, int64_t ld_wave,
                              int32_t n_fft, int32_t hop, int32_t frame_lo, int32_t n_out,
                              const float* window, const int32_t* mel_meta, const float* mel_w,
                              int32_t n_mels, int32_t mel_nnz, float log_eps, float clamp_lo, float clamp_hi,
                              float* out, void* stream) {
  using namespace adt;
  if (!wave || !window || !mel_meta || !mel_w || !out) return set_error(ADT_EINVAL, "adt_logmel_f32: null pointer");
  if (n_clips < 0 || n_out < 0 || hop <= 0 || frame_lo < 0 || ld_wave < n_samples)
    return set_error(ADT_EINVAL, "adt_logmel_f32: negative size, hop <= 0 or ld_wave < n_samples");
  if (n_fft != kNfft) return set_error(ADT_ESHAPE, "adt_logmel_f32: only n_fft == 2048 is supported");
  if (n_mels <= 0 || n_mels > 128 || (n_mels & 3)) return set_error(ADT_ESHAPE, "adt_logmel_f32: n_mels must be a multiple of 4 in [4,128]");
  if (mel_nnz < 0 || mel_nnz > kMaxMelNnz) return set_error(ADT_ESHAPE, "adt_logmel_f32: filterbank has too many non-zeros");
  if (n_samples >= (1ll << 30)) return set_error(ADT_ESHAPE, "adt_logmel_f32: n_samples must be below 2^30");
  if (n_samples <= n_fft / 2) return set_error(ADT_ESHAPE, "adt_logmel_f32: n_samples must exceed n_fft/2 (reflect padding)");
  // the last frame asked for must exist: frame t needs t*hop <= n_samples (1 + L//hop frames)
  if (n_out > 0 && static_cast<int64_t>(frame_lo + n_out - 1) * hop > n_samples)
    return set_error(ADT_ESHAPE, "adt_logmel_f32: frame_lo + n_out exceeds the 1 + n_samples/hop frames of the clip");
  if (!aligned16(out)) return set_error(ADT_EINVAL, "adt_logmel_f32: out must be 16-byte aligned");
  if (n_clips == 0 || n_out == 0) return ADT_OK;

  LogmelArgs a;
  a.wave = wave; a.n_clips = n_clips; a.n_samples = static_cast<int>(n_samples); a.ld_wave = ld_wave;
  a.hop = hop; a.frame_lo = frame_lo; a.n_out = n_out; a.pairs_per_clip = (n_out + 1) / 2;
  a.window = window; a.mel_meta = reinterpret_cast<const int4*>(mel_meta); a.mel_w = mel_w;
  a.n_mels = n_mels; a.mel_nnz = mel_nnz;
  a.log_eps = log_eps; a.clamp_lo = clamp_lo; a.clamp_hi = clamp_hi; a.out = out;
  a.n_items = n_clips * a.pairs_per_clip;

  int n_cu = 0;
  if (int e = device_cu_count(&n_cu)) return e;
  long blocks = (a.n_items + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > n_cu) blocks = n_cu;
  const long total_waves = blocks * kWavesPerBlock;
  a.n_iter = static_cast<int>((a.n_items + total_waves - 1) / total_waves);

  const size_t lds = kLdsTotal;
  static thread_local int attr_set_for_device = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (attr_set_for_device != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    attr_set_for_device = dev;
  }
  hipLaunchKernelGGL(logmel_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), lds,
                     static_cast<hipStream_t>(stream), a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
