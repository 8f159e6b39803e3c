// fx.hip -- K14: the optional FX chain of the one-shot mixer (gfx950).
//
// Stands behind VolumeMixer._add_fx (reference modules/synthetiser.py:121-137,154-155): with probability use_fx_prob the
// un-normalised mix of a clip goes through a pedalboard chain of Reverb -> Compressor -> Limiter (each present with its own
// probability, parameters drawn on the host: BoardChain, :30-87) before the peak normalisation.  pedalboard wraps JUCE; the
// three effects are restated here from JUCE's published algorithms (juce::Reverb::processMono = Freeverb, juce::dsp::Compressor
// with a peak BallisticsFilter, juce::dsp::Limiter = two compressors + make-up gain + hard clip).  Parity with pedalboard itself
// is unpinned (not installable here); the oracle (oracle/fx.py) is the same restatement, sample by sample.
//
// These are recurrences in time, the opposite of GPU-shaped work, so the two kernels are built around what CAN run in parallel:
//   fx_reverb_kernel    eight waves per clip (one per comb filter), 64 consecutive samples at a time (lane = sample).  Every comb / all-pass delay is
//                       longer than 64 samples (sample_rate >= 12544 Hz), so inside a chunk the delay-line reads and writes of
//                       the 64 samples are independent; the only serial part, the one-pole damping filter in each comb's
//                       feedback path, is a first-order linear recurrence and is evaluated with a 6-step affine prefix scan.
//                       Delay lines live in LDS (18 KB at 16 kHz, 55 KB at 48 kHz).
//   fx_dynamics_kernel  compressor and limiter: envelope followers whose coefficient (attack vs release) depends on their own
//                       previous output.  Also one wave per clip and 64 samples per step: the attack / release pattern of a
//                       chunk is found as a fixed point of (guess pattern -> affine prefix scan -> re-derive pattern), see
//                       Follower::chunk; the gain computer is element-wise.  Records the clip's new peak for the
//                       normalisation that follows.
// Both are latency-bound chains of ~2500 chunk steps per 10 s clip; all FX clips of a batch run side by side (one wave each).
#include <hip/hip_runtime.h>

#include "adt_common.h"

namespace adt {

// ---- 64-lane inclusive scan of affine maps y -> A y + B (lane order = time order) on the DPP network: four row_shr steps inside
// each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals forward.  Lanes without a source keep the
// identity (A = 1, B = 0), so there are no branches.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), kCtrl, kRowMask, 0xf, false));
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ void affine_step(float& A, float& B) {
  const float Ap = dpp_f<kCtrl, kRowMask>(1.0f, A), Bp = dpp_f<kCtrl, kRowMask>(0.0f, B);
  B = fmaf(A, Bp, B);
  A = A * Ap;
}
__device__ __forceinline__ void affine_scan64(float& A, float& B) {
  affine_step<0x111, 0xf>(A, B);       // row_shr:1
  affine_step<0x112, 0xf>(A, B);       // row_shr:2
  affine_step<0x114, 0xf>(A, B);       // row_shr:4
  affine_step<0x118, 0xf>(A, B);       // row_shr:8
  affine_step<0x142, 0xa>(A, B);       // row_bcast:15 into rows 1 and 3
  affine_step<0x143, 0xc>(A, B);       // row_bcast:31 into rows 2 and 3
}
__device__ __forceinline__ float lane_value(float v, int lane_uniform) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_uniform));
}

__constant__ int kCombTuning[8] = {1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617};
__constant__ int kAllpassTuning[4] = {556, 441, 341, 225};

// 8 waves per clip: wave w owns comb filter w (its delay line, its damping state); the comb outputs of a 64-sample chunk meet in a
// double-buffered LDS array, wave 0 adds them up and runs the four all-pass stages and the wet / dry mix while the others are
// already on the next chunk's combs (they only need the input signal).  One barrier per chunk.
__global__ __launch_bounds__(512) void fx_reverb_kernel(float* __restrict__ wav, long ld, const int32_t* __restrict__ clip_len,
                                                        const adt_fx_params* __restrict__ fx, int sample_rate, int width,
                                                        unsigned* __restrict__ clip_peak) {
  extern __shared__ float lines[];
  const int clip = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const adt_fx_params p = fx[clip];
  if (!(p.flags & 1)) return;
  int W = clip_len[clip];
  W = W < width ? W : width;
  int cs = 0, coff = 0, as[4], aoff[4], total = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = (sample_rate * kCombTuning[j]) / 44100;
    if (j == wave) { cs = n; coff = total; }
    total += n;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { as[j] = (sample_rate * kAllpassTuning[j]) / 44100; aoff[j] = total; total += as[j]; }
  float* sums = lines + total;                                   // [2][8][64] comb outputs of the chunk in flight
  for (int i = threadIdx.x; i < total; i += 512) lines[i] = 0.f;
  __syncthreads();
  const float wet = p.wet_level * 3.0f, dry = p.dry_level * 2.0f;
  const float wet1 = 0.5f * wet * (1.0f + p.width);
  const float damp = p.damping * 0.4f, omd = 1.0f - damp, fb = p.room_size * 0.28f + 0.7f;
  int pc = 0, pa[4] = {0, 0, 0, 0};
  float last = 0.f, peak = 0.f;
  float* row = wav + static_cast<long>(clip) * ld;
  int k = 0;
  // The input samples are requested a group of four chunks ahead: a chunk step is a chain of LDS round trips and scans of about a
  // microsecond, and a global load at its head (first touch of the mix: HBM latency) doubled it.
  float xq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xq[i] = 64 * i + lane < W ? row[64 * i + lane] : 0.f;
  for (int g0 = 0; g0 < W; g0 += 256) {
    float xn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int n = g0 + 256 + 64 * i + lane; xn[i] = n < W ? row[n] : 0.f; }
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
    const int n0 = g0 + 64 * c4;
    if (n0 >= W) break;                                          // workgroup-uniform
    const int valid = W - n0 < 64 ? W - n0 : 64;
    const bool on = lane < valid;
    const float x = xq[c4];
    {
      int idx = pc + lane;
      idx = idx >= cs ? idx - cs : idx;
      const float o = lines[coff + idx];
      // last_n = o_n * (1 - damp) + last_{n-1} * damp  over the 64 lanes: inclusive scan of the affine maps (damp, o * (1 - damp))
      float A = damp, B = o * omd;
      affine_scan64(A, B);
      const float lastn = fmaf(A, last, B);
      if (on) lines[coff + idx] = fmaf(lastn, fb, x * 0.015f);
      last = lane_value(lastn, valid - 1);
      sums[((k & 1) * 8 + wave) * 64 + lane] = o;
      pc += 64;
      pc = pc >= cs ? pc - cs : pc;
    }
    __syncthreads();
    if (wave == 0) {
      const float* sb = sums + (k & 1) * 512 + lane;
      float out = ((sb[0] + sb[64]) + (sb[128] + sb[192])) + ((sb[256] + sb[320]) + (sb[384] + sb[448]));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int idx = pa[j] + lane;
        idx = idx >= as[j] ? idx - as[j] : idx;
        const float bv = lines[aoff[j] + idx];
        if (on) lines[aoff[j] + idx] = fmaf(bv, 0.5f, out);
        out = bv - out;
        pa[j] += 64;
        pa[j] = pa[j] >= as[j] ? pa[j] - as[j] : pa[j];
      }
      if (on) {
        const float y = fmaf(out, wet1, x * dry);
        row[n0 + lane] = y;
        peak = fmaxf(peak, fabsf(y));
      }
    }
    ++k;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) xq[i] = xn[i];
  }
  if (wave == 0 && !(p.flags & 6)) {                               // no dynamics stage follows: this is the clip's new peak
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) peak = fmaxf(peak, __shfl_xor(peak, off));
    if (lane == 0) clip_peak[clip] = __float_as_uint(peak);
  }
}

// Envelope follower of juce::dsp::BallisticsFilter (peak): y_n = a_n + c_n (y_{n-1} - a_n) with c_n = attack coefficient when
// a_n > y_{n-1}, release coefficient otherwise -- a recurrence whose coefficient depends on its own previous output.  For a
// chunk of 64 samples (lane = sample) it is solved as a fixed point: guess the attack / release pattern, evaluate the (now
// linear) recurrence with an affine prefix scan, re-derive the pattern from the result, repeat until it no longer changes.
// Every pass makes at least one more leading sample final (sample 0 only depends on the carried-in state), so it terminates
// with the exact pattern; in audio the pattern is piecewise constant and two or three passes are typical.
struct Follower {
  float thr, thr_inv, expo, c_at, c_rl, yold;
  unsigned long long pattern;          // last chunk's attack mask: the first guess for the next one
  __device__ void init(float threshold_db, float ratio, float attack_ms, float release_ms, float sample_rate) {
    thr = threshold_db > -200.0f ? exp2f(threshold_db * 0.16609640474436813f) : 0.0f;          // 10^(dB / 20)
    thr_inv = 1.0f / thr;
    expo = 1.0f / ratio - 1.0f;
    const float ef = -6283.185307179586f / sample_rate;                                          // -2 pi 1000 / sr
    c_at = attack_ms < 1.0e-3f ? 0.0f : expf(ef / attack_ms);
    c_rl = release_ms < 1.0e-3f ? 0.0f : expf(ef / release_ms);
    yold = 0.0f;
    pattern = 0ull;
  }
  // v: this lane's sample; returns gain * v; lanes >= valid are ignored (they only sit behind the valid ones in the scan)
  __device__ __forceinline__ float chunk(float v, int lane, int valid) {
    const float a = fabsf(v);
    float y = 0.f;
    unsigned long long P = pattern;
    for (int pass = 0; pass < 65; ++pass) {
      const float c = ((P >> lane) & 1ull) ? c_at : c_rl;
      float A = c, B = fmaf(-c, a, a);                       // y_n = c y_{n-1} + (1 - c) a_n
      affine_scan64(A, B);
      y = fmaf(A, yold, B);
      const float yp = dpp_f<0x138, 0xf>(yold, y);           // wave_shr:1 -- lane 0 has no source and keeps the carried-in state
      const unsigned long long Pn = __ballot(a > yp);
      if (Pn == P) break;
      P = Pn;
    }
    pattern = P;
    yold = lane_value(y, valid - 1);
    const float g = y < thr ? 1.0f : exp2f(expo * log2f(y * thr_inv));
    return g * v;
  }
};

// one wave per clip, 64 samples at a time: compressor -> limiter stage 1 -> limiter stage 2 -> make-up gain -> clip, then the peak
__global__ __launch_bounds__(64) void fx_dynamics_kernel(float* __restrict__ wav, long ld, const int32_t* __restrict__ clip_len,
                                                         const adt_fx_params* __restrict__ fx, int sample_rate, int width,
                                                         unsigned* __restrict__ clip_peak) {
  const int clip = blockIdx.x, lane = threadIdx.x;
  const adt_fx_params p = fx[clip];
  if (!(p.flags & 6)) return;                                     // (a reverb-only clip got its peak from the reverb kernel)
  int W = clip_len[clip];
  W = W < width ? W : width;
  const bool comp = p.flags & 2, lim = p.flags & 4;
  Follower c, l1, l2;
  c.init(p.c_threshold_db, p.c_ratio, p.c_attack_ms, p.c_release_ms, static_cast<float>(sample_rate));
  l1.init(-10.0f, 4.0f, 2.0f, 200.0f, static_cast<float>(sample_rate));
  l2.init(p.l_threshold_db, 1000.0f, 0.001f, p.l_release_ms, static_cast<float>(sample_rate));
  const float makeup = exp2f((10.0f * 0.75f / 40.0f) * 3.321928094887362f) * exp2f(-p.l_threshold_db * 0.16609640474436813f);
  float* row = wav + static_cast<long>(clip) * ld;
  float peak = 0.f;
  float xq[4];                                                      // inputs a group of four chunks ahead (see fx_reverb_kernel)
#pragma unroll
  for (int i = 0; i < 4; ++i) xq[i] = 64 * i + lane < W ? row[64 * i + lane] : 0.f;
  for (int g0 = 0; g0 < W; g0 += 256) {
    float xn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int n = g0 + 256 + 64 * i + lane; xn[i] = n < W ? row[n] : 0.f; }
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
    const int n0 = g0 + 64 * c4;
    if (n0 >= W) break;
    const int valid = W - n0 < 64 ? W - n0 : 64;
    const bool on = lane < valid;
    float y = xq[c4];
    if (comp) y = c.chunk(y, lane, valid);
    if (lim) {
      y = l1.chunk(y, lane, valid);
      y = l2.chunk(y, lane, valid) * makeup;
      y = fminf(fmaxf(y, -1.0f), 1.0f);
    }
    if (on) {
      row[n0 + lane] = y;
      peak = fmaxf(peak, fabsf(y));
    }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) xq[i] = xn[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) peak = fmaxf(peak, __shfl_xor(peak, off));
  if (lane == 0) clip_peak[clip] = __float_as_uint(peak);
}

int launch_fx_chain(float* wav, long ld, const int32_t* clip_len, const adt_fx_params* fx, int n_clips, int sample_rate, int width,
                    unsigned* clip_peak, hipStream_t st) {
  if (sample_rate < 12544 || sample_rate > 96000)
    return set_error(ADT_ESHAPE, "fx chain: sample_rate must be in [12544, 96000] (the shortest reverb delay has to cover a 64-sample chunk)");
  int total = 0;
  const int comb[8] = {1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617}, ap[4] = {556, 441, 341, 225};
  for (int t : comb) total += (sample_rate * t) / 44100;
  for (int t : ap) total += (sample_rate * t) / 44100;
  const int lds = (total + 2 * 8 * 64) * 4;
  static thread_local int attr_dev = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (attr_dev != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fx_reverb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_dev = dev;
  }
  hipLaunchKernelGGL(fx_reverb_kernel, dim3(static_cast<unsigned>(n_clips)), dim3(512), lds, st, wav, ld, clip_len, fx, sample_rate, width,
                     clip_peak);
  hipLaunchKernelGGL(fx_dynamics_kernel, dim3(static_cast<unsigned>(n_clips)), dim3(64), 0, st, wav, ld, clip_len, fx, sample_rate, width,
                     clip_peak);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

}  // namespace adt
