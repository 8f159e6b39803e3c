// clap_logmel.hip -- K9: CLAP feature extractor on the GPU (gfx950).
//
// Stands behind ClapProcessor / ClapFeatureExtractor.__call__ as the reference invokes it
// (modules/clap_encoder.py:22-23; transformers ClapFeatureExtractor: "repeatpad" to 10 s @ 48 kHz, STFT 1024 /
// hop 480 periodic Hann center=True reflect, power, 64 htk mel filters 0-14 kHz without normalisation,
// 10*log10(max(.,1e-10))), which runs per clip in float64 numpy on the host (0.245 s/clip).
// One wave turns two frames into two output rows: packed complex 1024-point FFT (pass 1 of fft1024_phases.h into the L1 layout, then K1's
// second-generation passes 2 and 3 -- logmel2_phases.h: conflict-free 8 KiB buffer, pass-2 twiddles in registers, full-circle table; round 6:
// 1.05 -> 0.97 ms per 512 clips, profiles/r06/clap_logmel_ab.txt), in-place untangling, banded mel reduction (4 lanes per filter), dB.  Ragged input: clips are concatenated, `offsets`
// delimits them; repeat-padding and reflect padding are index arithmetic in the load, never materialised.
// 16 waves per CU (8.3 KB of LDS per wave), persistent grid-stride over (clip, frame pair) items.
#include <hip/hip_runtime.h>

#include "adt_common.h"
#include "fft1024_phases.h"

namespace adt {

constexpr int kClapWaves = 8;
constexpr int kClapThreads = 64 * kClapWaves;
constexpr int kClapMaxNnz = 1536;

struct ClapArgs {
  const float* waves; const float* const* clip_ptrs;      // clip b = clip_ptrs ? clip_ptrs[b] : waves + offsets[b]
  const long* offsets; int n_clips; int target; int hop; int n_frames; int pairs_per_clip;
  const float* window; const int4* mel_meta; const float* mel_w; int n_mels; int mel_nnz; float amin;
  float* out; long n_items; int n_iter;
};

__device__ __forceinline__ void wave_sync1k() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kClapThreads) void clap_logmel_kernel(ClapArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* tw = reinterpret_cast<cf*>(smem);                                    // [1024]  W_1024^j, the whole circle
  float* melw = reinterpret_cast<float*>(smem + 1024 * sizeof(cf));        // [kClapMaxNnz]
  cf* bufs = reinterpret_cast<cf*>(smem + 1024 * sizeof(cf) + kClapMaxNnz * sizeof(float));
  const int tid = threadIdx.x;
  const int lane_id = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cf* buf = bufs + wave * kL2Buf;
  for (int j = tid; j < 1024; j += kClapThreads) {
    float s, c;
    sincospif(static_cast<float>(j) * (1.0f / 512.0f), &s, &c);
    tw[j] = cf{c, -s};
  }
  for (int j = tid; j < a.mel_nnz; j += kClapThreads) melw[j] = a.mel_w[j];
  float win[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) win[n1] = a.window[lane_id + 64 * n1];
  const int g = lane_id >> 2, s4 = lane_id & 3;
  unsigned mband[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = g + 16 * i;
    int4 m = (j < a.n_mels) ? a.mel_meta[j] : make_int4(0, 0, 0, 0);
    mband[i] = static_cast<unsigned>(m.x) | (static_cast<unsigned>(m.y) << 11) | (static_cast<unsigned>(m.z) << 18);
  }
  __syncthreads();
  cf tw2[2][8];                                        // pass 2's twiddles: lane constants, in registers for the whole kernel (logmel.hip)
  l2_pass2_twiddles(lane_id, 0, tw, tw2[0]);
  l2_pass2_twiddles(lane_id, 1, tw, tw2[1]);
  const long total_waves = static_cast<long>(gridDim.x) * kClapWaves;
  const long first = static_cast<long>(blockIdx.x) * kClapWaves + wave;
  // (clip, frame pair) without a 64-bit division per item: quotient and remainder advance by constants and one carry (logmel.hip)
  long item = first;
  int clip_i = static_cast<int>(item / a.pairs_per_clip);
  int pair = static_cast<int>(item - static_cast<long>(clip_i) * a.pairs_per_clip);
  const int step_q = static_cast<int>(total_waves / a.pairs_per_clip);
  const int step_r = static_cast<int>(total_waves - static_cast<long>(step_q) * a.pairs_per_clip);
  for (int iter = 0; iter < a.n_iter; ++iter, item += total_waves, clip_i += step_q + (pair + step_r >= a.pairs_per_clip ? 1 : 0),
           pair = pair + step_r >= a.pairs_per_clip ? pair + step_r - a.pairs_per_clip : pair + step_r) {
    int lane = lane_id;
    asm volatile("" : "+v"(lane));                // keep LDS address arithmetic inside the iteration (see logmel.hip)
#pragma unroll
    for (int k = 0; k < 8; ++k) {                 // (... and the twiddles loop-variant for LICM: it would hoist sixteen products and spill them)
      asm volatile("" : "+v"(tw2[0][k]));
      asm volatile("" : "+v"(tw2[1][k]));
    }
    if (item >= a.n_items) break;                 // wave-uniform; no workgroup barrier inside the loop
    const int f0 = 2 * pair;
    const bool has1 = f0 + 1 < a.n_frames;
    const long o0 = a.offsets[clip_i];
    const int n = static_cast<int>(a.offsets[clip_i + 1] - o0);
    const float* clip = a.clip_ptrs ? a.clip_ptrs[clip_i] : a.waves + o0;
    const int base0 = f0 * a.hop - kN1k / 2, base1 = base0 + a.hop;
    if (base0 >= 0 && base1 + kN1k <= a.target) p1k_pass1_l1<true>(lane, clip, n, a.target, base0, base1, has1, win, tw, buf);
    else p1k_pass1_l1<false>(lane, clip, n, a.target, base0, base1, has1, win, tw, buf);
    wave_sync1k();
    cf z[2][8];
    l2_pass2_load(lane, 0, buf, z[0]);
    l2_pass2_load(lane, 1, buf, z[1]);
    wave_sync1k();
    l2_pass2_store_tw(lane, 0, z[0], tw2[0], buf);
    __builtin_amdgcn_sched_barrier(0);
    l2_pass2_store_tw(lane, 1, z[1], tw2[1], buf);
    wave_sync1k();
    l2_pass3_load(lane, 0, buf, z[0]);
    l2_pass3_load(lane, 1, buf, z[1]);
    wave_sync1k();
    l2_pass3_store(lane, 0, z[0], buf);
    l2_pass3_store(lane, 1, z[1], buf);
    wave_sync1k();
    p1k_untangle(lane, buf);
    wave_sync1k();
    float* stage = reinterpret_cast<float*>(buf + kStage1k);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cf acc = mel_partial(s4, mband[i] & 2047u, (mband[i] >> 11) & 127u, mband[i] >> 18, melw, buf);
      acc.x += __shfl_xor(acc.x, 1); acc.y += __shfl_xor(acc.y, 1);
      acc.x += __shfl_xor(acc.x, 2); acc.y += __shfl_xor(acc.y, 2);
      const int j = g + 16 * i;
      if (s4 == 0 && j < a.n_mels) { stage[j] = to_db(acc.x, a.amin); stage[a.n_mels + j] = to_db(acc.y, a.amin); }
    }
    wave_sync1k();
    const float4* stage4 = reinterpret_cast<const float4*>(buf + kStage1k);
    const int quads = (has1 ? 2 : 1) * a.n_mels / 4;
    float4* dst = reinterpret_cast<float4*>(a.out + (static_cast<long>(clip_i) * a.n_frames + f0) * a.n_mels);
    if (lane < quads) dst[lane] = stage4[lane];
    wave_sync1k();
  }
}

}  // namespace adt

static int clap_logmel_impl(const float* waves, const float* const* clip_ptrs, const int64_t* offsets, int64_t n_clips, int32_t target_len, int32_t n_fft,
                            int32_t hop, int32_t n_frames, const float* window, const int32_t* mel_meta, const float* mel_w,
                            int32_t n_mels, int32_t mel_nnz, float amin, float* out, void* stream) {
  using namespace adt;
  if ((!waves && !clip_ptrs) || !offsets || !window || !mel_meta || !mel_w || !out) return set_error(ADT_EINVAL, "adt_clap_logmel_db_f32: null pointer");
  if (n_fft != kN1k) return set_error(ADT_ESHAPE, "adt_clap_logmel_db_f32: only n_fft == 1024 is supported");
  if (n_mels <= 0 || n_mels > 64 || (n_mels & 3)) return set_error(ADT_ESHAPE, "adt_clap_logmel_db_f32: n_mels must be a multiple of 4 in [4,64]");
  if (mel_nnz < 0 || mel_nnz > kClapMaxNnz) return set_error(ADT_ESHAPE, "adt_clap_logmel_db_f32: filterbank has too many non-zeros");
  if (n_clips < 0 || hop <= 0 || n_frames < 0 || target_len <= n_fft / 2) return set_error(ADT_EINVAL, "adt_clap_logmel_db_f32: bad sizes");
  if (n_frames > 0 && static_cast<int64_t>(n_frames - 1) * hop > target_len)
    return set_error(ADT_ESHAPE, "adt_clap_logmel_db_f32: n_frames exceeds 1 + target_len/hop");
  if (!aligned16(out)) return set_error(ADT_EINVAL, "adt_clap_logmel_db_f32: out must be 16-byte aligned");
  if (n_clips == 0 || n_frames == 0) return ADT_OK;
  ClapArgs a;
  a.waves = waves; a.clip_ptrs = clip_ptrs; a.offsets = reinterpret_cast<const long*>(offsets); a.n_clips = static_cast<int>(n_clips); a.target = target_len;
  a.hop = hop; a.n_frames = n_frames; a.pairs_per_clip = (n_frames + 1) / 2; a.window = window;
  a.mel_meta = reinterpret_cast<const int4*>(mel_meta); a.mel_w = mel_w; a.n_mels = n_mels; a.mel_nnz = mel_nnz; a.amin = amin; a.out = out;
  a.n_items = n_clips * a.pairs_per_clip;
  int n_cu = 0;
  if (int rc = device_cu_count(&n_cu)) return rc;
  long blocks = (a.n_items + kClapWaves - 1) / kClapWaves;
  if (blocks > 2L * n_cu) blocks = 2L * n_cu;
  const long tw_ = blocks * kClapWaves;
  a.n_iter = static_cast<int>((a.n_items + tw_ - 1) / tw_);
  const size_t lds = 1024 * sizeof(cf) + kClapMaxNnz * sizeof(float) + kClapWaves * kL2Buf * sizeof(cf);   // 79,872 B -> 2 workgroups / CU
  static thread_local int done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (done_for != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(clap_logmel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    done_for = dev;
  }
  hipLaunchKernelGGL(clap_logmel_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kClapThreads), lds, static_cast<hipStream_t>(stream), a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
extern "C" int adt_clap_logmel_db_f32(const float* waves, const int64_t* offsets, int64_t n_clips, int32_t target_len, int32_t n_fft,
                                      int32_t hop, int32_t n_frames, const float* window, const int32_t* mel_meta, const float* mel_w,
                                      int32_t n_mels, int32_t mel_nnz, float amin, float* out, void* stream) {
  return clap_logmel_impl(waves, nullptr, offsets, n_clips, target_len, n_fft, hop, n_frames, window, mel_meta, mel_w, n_mels, mel_nnz, amin, out, stream);
}
// The same over clips that already live in device memory one by one (a device array of n_clips pointers): no concatenation pass.
// offsets[b + 1] - offsets[b] is still clip b's length.
extern "C" int adt_clap_logmel_db_ptrs_f32(const float* const* clip_ptrs, const int64_t* offsets, int64_t n_clips, int32_t target_len, int32_t n_fft,
                                           int32_t hop, int32_t n_frames, const float* window, const int32_t* mel_meta, const float* mel_w,
                                           int32_t n_mels, int32_t mel_nnz, float amin, float* out, void* stream) {
  if (!clip_ptrs) return adt::set_error(ADT_EINVAL, "adt_clap_logmel_db_ptrs_f32: null pointer");
  return clap_logmel_impl(nullptr, clip_ptrs, offsets, n_clips, target_len, n_fft, hop, n_frames, window, mel_meta, mel_w, n_mels, mel_nnz, amin, out, stream);
}

// ------------------------------------------------------------------------------------ bilinear resize (fusion "shrink" mel)
// torch.nn.functional.interpolate(x[None, None], size=[H_out, W_out], mode="bilinear", align_corners=False) as the feature
// extractor calls it for clips longer than 10 s (transformers feature_extraction_clap.py, _random_mel_fusion): source index
// max(scale * (dst + 0.5) - 0.5, 0) with scale = in / out in fp32, neighbour clamped at the edge, and the four-term blend in
// torch's operand order.
namespace adt {
__global__ __launch_bounds__(256) void bilinear_resize_kernel(const float* __restrict__ in, int H_in, int W_in, long ld_in,
                                                              float* __restrict__ out, int H_out, int W_out, long ld_out,
                                                              float scale_h, float scale_w) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= static_cast<long>(H_out) * W_out) return;
  const int h2 = static_cast<int>(i / W_out), w2 = static_cast<int>(i - static_cast<long>(h2) * W_out);
  const float h1r = fmaxf(__fsub_rn(__fmul_rn(scale_h, static_cast<float>(h2) + 0.5f), 0.5f), 0.f);
  const float w1r = fmaxf(__fsub_rn(__fmul_rn(scale_w, static_cast<float>(w2) + 0.5f), 0.5f), 0.f);
  const int h1 = static_cast<int>(h1r), w1 = static_cast<int>(w1r);
  const int h1p = h1 < H_in - 1 ? 1 : 0, w1p = w1 < W_in - 1 ? 1 : 0;
  const float h1l = h1r - static_cast<float>(h1), h0l = 1.f - h1l;
  const float w1l = w1r - static_cast<float>(w1), w0l = 1.f - w1l;
  const float* r0 = in + static_cast<long>(h1) * ld_in;
  const float* r1 = in + static_cast<long>(h1 + h1p) * ld_in;
  const float top = __fadd_rn(__fmul_rn(w0l, r0[w1]), __fmul_rn(w1l, r0[w1 + w1p]));
  const float bot = __fadd_rn(__fmul_rn(w0l, r1[w1]), __fmul_rn(w1l, r1[w1 + w1p]));
  out[static_cast<long>(h2) * ld_out + w2] = __fadd_rn(__fmul_rn(h0l, top), __fmul_rn(h1l, bot));
}
}  // namespace adt

extern "C" int adt_bilinear_resize_f32(const float* in, int64_t H_in, int64_t W_in, int64_t ld_in, float* out, int64_t H_out, int64_t W_out,
                                       int64_t ld_out, void* stream) {
  using namespace adt;
  if (!in || !out) return set_error(ADT_EINVAL, "adt_bilinear_resize_f32: null pointer");
  if (H_in <= 0 || W_in <= 0 || H_out < 0 || W_out < 0 || ld_in < W_in || ld_out < W_out || H_in > (1 << 30) || H_out > (1 << 30))
    return set_error(ADT_ESHAPE, "adt_bilinear_resize_f32: bad sizes");
  const long n = static_cast<long>(H_out) * W_out;
  if (n == 0) return ADT_OK;
  const float sh = static_cast<float>(H_in) / static_cast<float>(H_out), sw = static_cast<float>(W_in) / static_cast<float>(W_out);
  hipLaunchKernelGGL(bilinear_resize_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                     static_cast<int>(H_in), static_cast<int>(W_in), static_cast<long>(ld_in), out, static_cast<int>(H_out), static_cast<int>(W_out),
                     static_cast<long>(ld_out), sh, sw);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
