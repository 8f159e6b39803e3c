// logmel.hip -- K1: fused STFT -> power -> banded mel -> log/clamp/scale -> trim (gfx950).
//
// Stands behind ComputeMelSpectrogram.forward (reference model.py:81-97).
//
// Launch shape: persistent, one 1024-thread workgroup (16 waves, four per SIMD, <= 128 VGPRs) per CU.  Each wave owns an
// 8 KiB LDS buffer and walks (clip, frame) work items with a grid stride: ONE real frame per wave, packed by even / odd samples
// into a 1024-point complex FFT (logmel2_phases.h: the radix passes, the conflict-free LDS layouts, the real-FFT untangling).
// The workgroup shares the half-circle twiddle table (8 KiB), the window as (even, odd) pairs (8 KiB) and the non-zero mel
// weights (<= 9 KiB): 153 KiB of the CU's 160 KiB LDS.  The waveform is read straight from global memory (each sample is
// touched by ~n_fft/hop = 12.8 overlapping frames, which neighbouring waves take from L2 / L1); an output row is staged in
// LDS and written as one 512-byte row (float4 per lane).
// (First generation, round 1: two frames per wave through a 2048-point complex FFT, 16.6 KiB per wave -> 8 waves per CU,
// 252 VGPRs; 1.05 ms per 256 clips with 43 % of its LDS cycles lost to bank conflicts.)
#include <hip/hip_runtime.h>

#include "adt_common.h"
#include "logmel2_phases.h"

namespace adt {

constexpr int kWavesPerBlock = 16;
constexpr int kThreads = 64 * kWavesPerBlock;
constexpr int kMaxMelNnz = 2304;
constexpr int kMaxMelPad = kMaxMelNnz + 4 * 128;      // every band padded with zero weights to a multiple of 4 bins (+ the odd-stride padding of the usual filterbanks)
constexpr size_t kLdsTw = (1024 + 520) * sizeof(cf);     // W_1024^j, j < 1024 | W_2048^k, k <= 512 (padded to 520)
constexpr size_t kLdsWin = 1024 * sizeof(cf);
constexpr size_t kLdsBands = (128 + 8) * sizeof(unsigned);   // per mel: first bin | padded weight offset << 11 | own trips << 24; then the trip count of each mel item
constexpr size_t kLdsBufs = kLdsTw + kLdsWin + kMaxMelPad * sizeof(float) + kLdsBands;
constexpr size_t kLdsTotal = kLdsBufs + kWavesPerBlock * kL2Buf * sizeof(cf);   // 162,912 B
static_assert(kLdsTotal <= 160 * 1024, "LDS budget");
static_assert(kL2Stage + 128 <= 2 * kL2Buf, "staged row fits behind the powers");

struct LogmelArgs {
  const float* wave; long n_clips; int n_samples; long ld_wave;
  int hop; int frame_lo; int n_out;
  const float* window; const int4* mel_meta; const float* mel_w; int n_mels; int mel_nnz;
  float log_eps, clamp_lo, clamp_hi;
  float* out;
  long n_items; int n_iter; int pair_loads;       // pair_loads: every clip row is 8-byte aligned (float2 sample loads for even frame starts)
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

// value of lane (lane ^ 1) / (lane ^ 2): quad_perm [1,0,3,2] = 0xB1 / [2,3,0,1] = 0x4E
template <int kCtrl>
__device__ __forceinline__ float quad_xor(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), kCtrl, 0xf, 0xf, true));
}

// A band as the kernel will use it: first bin inside the spectrum, at most 127 bins that stay inside it, weights inside mel_w.
// MelBands (adt_str_amd/frontend.py) only builds such bands; a malformed table handed to the C entry directly is cut down to
// something in range here (its output is then meaningless, but no LDS or global index leaves its array).
__device__ __forceinline__ int4 sane_band(int4 m, int mel_nnz) {
  const int bins = kNfft / 2 + 1;
  m.x = m.x < 0 ? 0 : (m.x > bins - 1 ? bins - 1 : m.x);
  int n = m.y < 0 ? 0 : (m.y > 127 ? 127 : m.y);
  n = n > bins - m.x ? bins - m.x : n;
  if (m.z < 0 || m.z > mel_nnz - n) n = 0;
  m.y = n;
  return m;
}

__global__ __launch_bounds__(kThreads) void logmel_kernel(LogmelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* t1k = reinterpret_cast<cf*>(smem);                                 // [1024]  W_1024^j, the whole circle
  cf* t2k = t1k + 1024;                                                  // [513]   W_2048^k
  cf* win2 = reinterpret_cast<cf*>(smem + kLdsTw);                       // [1024]  (window[2m], window[2m+1])
  float* melw = reinterpret_cast<float*>(smem + kLdsTw + kLdsWin);       // [kMaxMelPad]  band weights, zero-padded to multiples of 4
  unsigned* bands = reinterpret_cast<unsigned*>(smem + kLdsTw + kLdsWin + kMaxMelPad * sizeof(float));   // [128 + 8]
  cf* bufs = reinterpret_cast<cf*>(smem + kLdsBufs);

  const int tid = threadIdx.x;
  const int lane_id = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> SGPR addressing
  cf* buf = bufs + wave * kL2Buf;

  // shared tables: W_1024^j = exp(-2*pi*i*j/1024), W_2048^k; the window in (even, odd) pairs; mel weights
  for (int j = tid; j < 1024; j += kThreads) {
    float s, c;
    sincospif(static_cast<float>(j) * (1.0f / 512.0f), &s, &c);
    t1k[j] = cf{c, -s};
    win2[j] = cf{a.window[2 * j], a.window[2 * j + 1]};
    if (j <= 512) {
      sincospif(static_cast<float>(j) * (1.0f / 1024.0f), &s, &c);
      t2k[j] = cf{c, -s};
    }
  }
  // Mel bands for the reduction: lane 4 g + s of mel item i sums weight * power over bins lo + s, lo + s + 4, ...  To keep that
  // loop free of per-lane bounds (a divergent trip count costs more VALU than the sums themselves) every band's weights are
  // padded with zeros to a multiple of four bins and all sixteen bands of an item run the item's longest trip count.
  for (int j = tid; j < kMaxMelPad; j += kThreads) melw[j] = 0.f;
  __syncthreads();
  if (tid < 8) {                                        // trip count of item i = widest of its bands g + 16 i
    int trips = 0;
    for (int g = 0; g < 16; ++g) {
      const int j = g + 16 * tid;
      const int c = j < a.n_mels ? (sane_band(a.mel_meta[j], a.mel_nnz).y + 3) >> 2 : 0;
      trips = c > trips ? c : trips;
    }
    bands[128 + tid] = static_cast<unsigned>(trips);
  }
  __syncthreads();
  // Layout of the padded weights: every band of an item as long as the item's trip count when that fits (the loop then needs no
  // per-lane bound at all: the usual 128-mel filterbanks); else every band padded to its own multiple of four and the loop
  // masks the weight past the lane's own trips.
  // Bank conflicts of the weight reads: the eight bands of a half-wave sit 4 * stride floats apart, so with an EVEN stride (in trips)
  // they fall on 4, 2 or 1 of the eight 4-bank groups (trips 2 / 4 / 8: 2-, 4-, 8-way conflicts -- 132 of the 741 LDS cycles of a frame,
  // tools/lds_conflicts_mel.py); an odd stride puts them on eight different groups.  So the band stride of an item is its trip count
  // made odd (the extra trip is never run, only skipped over) when the table still fits.
  int item_total = 0, item_total_odd = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    item_total += 64 * static_cast<int>(bands[128 + i]);
    item_total_odd += 64 * static_cast<int>(bands[128 + i] | 1u);
  }
  const bool uniform_pad = item_total <= kMaxMelPad;   // block-uniform
  const unsigned odd = (uniform_pad && item_total_odd <= kMaxMelPad) ? 1u : 0u;
  if (tid < 128) {
    int poff = 0;                                       // offset of band `tid` in the padded weight array
    for (int j = 0; j < tid && j < a.n_mels; ++j) poff += uniform_pad ? 4 * static_cast<int>(bands[128 + (j >> 4)] | odd) : (sane_band(a.mel_meta[j], a.mel_nnz).y + 3) & ~3;
    const int4 m = (tid < a.n_mels) ? sane_band(a.mel_meta[tid], a.mel_nnz) : make_int4(0, 0, 0, 0);
    bands[tid] = static_cast<unsigned>(m.x) | (static_cast<unsigned>(poff) << 11) | (static_cast<unsigned>((m.y + 3) >> 2) << 24);
    for (int t = 0; t < m.y; ++t) melw[poff + t] = a.mel_w[m.z + t];
  }
  __syncthreads();

  // Work split: the items (clip, frame) are cut into eight contiguous ranges, one per XCD group (workgroups b and b + 8 share an
  // XCD under round-robin placement; only speed depends on that): neighbouring frames overlap by 1 - hop / n_fft = 92 % of their
  // samples, so one XCD's L2 then fetches a stretch of waveform once instead of all eight fetching all of it.
  const int ng = gridDim.x < 8 ? static_cast<int>(gridDim.x) : 8;               // groups (small launches: one workgroup each)
  const int xg = blockIdx.x % ng, n_grp = (static_cast<int>(gridDim.x) + ng - 1 - xg) / ng;      // workgroups in this group
  const long per = (a.n_items + ng - 1) / ng;
  const long lo = per * xg, hi = lo + per < a.n_items ? lo + per : a.n_items;
  const long group_waves = static_cast<long>(n_grp) * kWavesPerBlock;
  const long first = lo + static_cast<long>(blockIdx.x / ng) * kWavesPerBlock + wave;

  // (clip, frame) of the wave's items without a 64-bit division per frame (≈ 250 scalar instructions of the ≈ 260 a frame carried): the
  // item index advances by group_waves per iteration, so quotient and remainder advance by constants and one carry
  long item = first;
  long clip_i = item / a.n_out;
  int f = static_cast<int>(item - clip_i * a.n_out);
  const long step_q = group_waves / a.n_out;
  const int step_r = static_cast<int>(group_waves - step_q * a.n_out);
  cf tw2[2][8];                                        // pass 2's twiddles: lane constants, in registers for the whole kernel
  l2_pass2_twiddles(lane_id, 0, t1k, tw2[0]);
  l2_pass2_twiddles(lane_id, 1, t1k, tw2[1]);
  for (int iter = 0; iter < a.n_iter; ++iter) {
    // Per-lane LDS addresses are re-derived inside the iteration (the empty asm hides that `lane` is loop-invariant): hoisted by LICM
    // they are spilled.  With the complex arithmetic on register pairs (logmel_phases.h) a frame needs 80 registers instead of 116; the 48
    // that are left of the 128 of four waves per SIMD go to pass 2's sixteen twiddles (above) -- hoisting the addresses instead gave the
    // same instruction count and kept the conflict-prone table reads.
    int lane = lane_id;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int k = 0; k < 8; ++k) {                       // (... and that the twiddles are: LICM would hoist sixteen i * w products too, and spill them)
      asm volatile("" : "+v"(tw2[0][k]));
      asm volatile("" : "+v"(tw2[1][k]));
    }
    if (item >= hi) break;                             // wave-uniform; no workgroup barrier inside the loop
    const float* clip = a.wave + clip_i * a.ld_wave;
    const int base = (a.frame_lo + f) * a.hop - kNfft / 2;
    if (base >= 0 && base + kNfft <= a.n_samples) {
      if (a.pair_loads && !(base & 1)) l2_pass1<true, true>(lane, clip, a.n_samples, base, win2, t1k, buf);     // 8-byte sample loads
      else l2_pass1<true, false>(lane, clip, a.n_samples, base, win2, t1k, buf);
    } else {
      l2_pass1<false, false>(lane, clip, a.n_samples, base, win2, t1k, buf);
    }
    wave_sync();
    cf z[2][8];
    l2_pass2_load(lane, 0, buf, z[0]);
    l2_pass2_load(lane, 1, buf, z[1]);
    wave_sync();
    l2_pass2_store_tw(lane, 0, z[0], tw2[0], buf);
    __builtin_amdgcn_sched_barrier(0);
    l2_pass2_store_tw(lane, 1, z[1], tw2[1], buf);
    wave_sync();
    l2_pass3_load(lane, 0, buf, z[0]);
    l2_pass3_load(lane, 1, buf, z[1]);
    wave_sync();
    l2_pass3_store(lane, 0, z[0], buf);
    l2_pass3_store(lane, 1, z[1], buf);
    wave_sync();
    float* pw = reinterpret_cast<float*>(buf);
#ifndef ADT_LM_NOUNTANGLE      // (timing arms, tools/build_variant.sh: the output is meaningless)
    float pk[8], pnk[8], p512;
    l2_untangle_load(lane, t2k, buf, pk, pnk, p512);
    wave_sync();
    l2_untangle_store(lane, pk, pnk, p512, pw);
    wave_sync();
#endif
    const int g = lane >> 2, s = lane & 3;
    float* stage = pw + kL2Stage;
#if defined(ADT_LM_NOMEL) || defined(ADT_LM_NOUNTANGLE)
    if (s == 0) { stage[g] = pw[lane]; stage[g + 64] = pw[lane + 64]; }
#else
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned mb = bands[g + 16 * i];
      const int trips = __builtin_amdgcn_readfirstlane(static_cast<int>(bands[128 + i]));     // wave-uniform
      const float* pp = pw + (mb & 2047u) + s;           // (bins past the band's end carry zero weights; they stay inside the wave's buffer)
      const float* wp = melw + ((mb >> 11) & 8191u) + s;
      float acc = 0.f;
      if (uniform_pad) {
        int t = 0;
        for (; t + 4 <= trips; t += 4) {               // four independent read pairs in flight: the sum is a latency chain otherwise
          const float w0 = wp[4 * t], w1 = wp[4 * t + 4], w2 = wp[4 * t + 8], w3 = wp[4 * t + 12];
          const float p0 = pp[4 * t], p1 = pp[4 * t + 4], p2 = pp[4 * t + 8], p3 = pp[4 * t + 12];
          acc = fmaf(w0, p0, acc); acc = fmaf(w1, p1, acc); acc = fmaf(w2, p2, acc); acc = fmaf(w3, p3, acc);
        }
        for (; t < trips; ++t) acc = fmaf(wp[4 * t], pp[4 * t], acc);
      } else {
        const int own = static_cast<int>(mb >> 24);
        for (int t = 0; t < trips; ++t) acc = fmaf(t < own ? wp[4 * t] : 0.f, pp[4 * t], acc);
      }
      acc += quad_xor<0xB1>(acc);                        // lanes s ^ 1, then s ^ 2: DPP quad permutes (no LDS round trip)
      acc += quad_xor<0x4E>(acc);
      if (s == 0) stage[g + 16 * i] = acc;               // raw band energies; the logarithm is taken once per output element below
    }
#endif
    wave_sync();
    if (lane < a.n_mels / 2) {                         // two mels per lane: log / clamp / scale (model.py:91-93) and the 512-byte row
      const float2 e = reinterpret_cast<const float2*>(stage)[lane];
      float2 y;
      y.x = post_fast(e.x, a.log_eps, a.clamp_lo, a.clamp_hi);
      y.y = post_fast(e.y, a.log_eps, a.clamp_lo, a.clamp_hi);
      reinterpret_cast<float2*>(a.out + (clip_i * a.n_out + f) * a.n_mels)[lane] = y;
    }
    wave_sync();
    item += group_waves;
    clip_i += step_q;
    f += step_r;
    if (f >= a.n_out) { f -= a.n_out; ++clip_i; }
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
  // (mel_meta lives in device memory: the host side validates it when it builds it (MelBands), the kernel clamps every band into
  //  range when it sets up its tables: sane_band)
  if (n_samples >= (1ll << 30)) return set_error(ADT_ESHAPE, "adt_logmel_f32: n_samples must be below 2^30");
  if (n_samples <= n_fft / 2) return set_error(ADT_ESHAPE, "adt_logmel_f32: n_samples must exceed n_fft/2 (reflect padding)");
  // the last frame asked for must exist: frame t needs t*hop <= n_samples (1 + L//hop frames)
  if (n_out > 0 && static_cast<int64_t>(frame_lo + n_out - 1) * hop > n_samples)
    return set_error(ADT_ESHAPE, "adt_logmel_f32: frame_lo + n_out exceeds the 1 + n_samples/hop frames of the clip");
  if (!aligned16(out)) return set_error(ADT_EINVAL, "adt_logmel_f32: out must be 16-byte aligned");
  if (n_clips == 0 || n_out == 0) return ADT_OK;

  LogmelArgs a;
  a.wave = wave; a.n_clips = n_clips; a.n_samples = static_cast<int>(n_samples); a.ld_wave = ld_wave;
  a.hop = hop; a.frame_lo = frame_lo; a.n_out = n_out;
  a.window = window; a.mel_meta = reinterpret_cast<const int4*>(mel_meta); a.mel_w = mel_w;
  a.n_mels = n_mels; a.mel_nnz = mel_nnz;
  a.log_eps = log_eps; a.clamp_lo = clamp_lo; a.clamp_hi = clamp_hi; a.out = out;
  a.n_items = n_clips * n_out;
  a.pair_loads = ((reinterpret_cast<uintptr_t>(wave) & 7u) == 0 && (ld_wave & 1) == 0) ? 1 : 0;

  int n_cu = 0;
  if (int e = device_cu_count(&n_cu)) return e;
  long blocks = (a.n_items + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > n_cu) blocks = n_cu;
  // per XCD group: ceil(n_items / 8) items over floor(blocks / 8) (at least one) workgroups of 16 waves
  const long ng = blocks < 8 ? blocks : 8;
  const long per_group = (a.n_items + ng - 1) / ng, group_waves = (blocks / ng) * kWavesPerBlock;
  a.n_iter = static_cast<int>((per_group + group_waves - 1) / group_waves);

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
