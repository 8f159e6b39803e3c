// htsat.hip -- K10/K11: the Swin-specific kernels of the CLAP HTSAT audio encoder, forward only (gfx950).
//
// Stands behind ClapAudioEncoder.forward as the reference reaches it through ClapWrapper._get_audio_features
// (modules/clap_encoder.py:45-49 -> transformers modeling_clap.py ClapAudioEncoder / ClapAudioLayer /
// ClapAudioSelfAttention / ClapAudioPatchMerging).  The dense layers of the encoder run on adt_gemm_bf16 and
// adt_layernorm_fwd; this file holds what is specific to the audio Swin:
//
//   htsat_front_kernel        BatchNorm2d over the 64 mel bins (eval), bicubic time resize 1001 -> 1024
//                             (align_corners, A = -0.75) and the freq-stacking fold to a 256 x 256 image
//   htsat_patch_embed_kernel  4x4 / stride-4 patch embedding conv (1 -> 96 channels) + LayerNorm
//   window_attn_kernel        8x8-window multi-head attention, head_dim 24, with the learned relative position
//                             bias and the shifted-window mask; window partition, cyclic shift and their inverses
//                             are index arithmetic (tokens are read from / written to their image rows directly)
//   patch_merge_ln_kernel     2x2 neighbourhood gather (Swin patch merging order) + LayerNorm(4C)
//   mean_tokens_kernel        mean over the tokens of a clip (the grouped avg-pool head reduces to it)
//   l2_normalize_kernel       rows / ||row||_2
//
// Window attention uses v_mfma_f32_32x32x16_bf16 with the query on the lane (S^T = K Q^T, d padded 24 -> 32 with
// zero operands), so the softmax is in-register and the probabilities are directly the B operand of
// O^T += V^T P^T; V^T fragments come from a 4 KiB LDS tile through ds_read_b64_tr_b16.  One wave per (window, head).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "adt_common.h"

namespace adt {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned short f2bf_h(float f) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f)); }
__device__ __forceinline__ unsigned pack2_h(float lo, float hi) { return f2bf_h(lo) | (static_cast<unsigned>(f2bf_h(hi)) << 16); }
__device__ __forceinline__ int acc_row_h(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------ front: BN + bicubic + fold
// img[b][F = r*n_mels + f][t'] = resize(bn(x))[t = r*(W) + t'][f],  r = 0..(H/n_mels - 1),  W = out_t / (H/n_mels)
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ __launch_bounds__(256) void htsat_front_kernel(const float* __restrict__ mel, long ld_clip, int in_t, int n_mels, int out_t, int img,
                                                          const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                          float* __restrict__ out, long total) {
  const long idx = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;     // over b * img * img, fastest = f (coalesced reads)
  if (idx >= total) return;
  const int f = static_cast<int>(idx % n_mels);
  const long rest = idx / n_mels;
  const int t = static_cast<int>(rest % out_t);
  const long b = rest / out_t;
  const float scale = static_cast<float>(in_t - 1) / static_cast<float>(out_t - 1);
  const float real = scale * t;
  const int ix = static_cast<int>(floorf(real));
  const float tt = real - ix;
  const float A = -0.75f;
  const float w[4] = {cubic2(tt + 1.f, A), cubic1(tt, A), cubic1(1.f - tt, A), cubic2(2.f - tt, A)};
  const float* src = mel + b * ld_clip;
  const float sc = bn_scale[f], sh = bn_shift[f];
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int j = ix - 1 + k;
    j = j < 0 ? 0 : (j > in_t - 1 ? in_t - 1 : j);
    v += w[k] * (src[static_cast<long>(j) * n_mels + f] * sc + sh);
  }
  const int W = img;                         // time columns per folded row-block
  const int r = t / W, tp = t - r * W;
  out[(b * img + (r * n_mels + f)) * img + tp] = v;
}

// The same for 64 mel bins with the OUTPUT coalesced.  The kernel above walks the mel bins with its lanes: its reads are coalesced but
// every lane's result lands in a different image row, 4 bytes per 1 KiB row -- 33 M scattered stores per 512 clips (0.22 ms for
// 265 MB).  Here a workgroup owns 64 consecutive output times x 64 bins: the <= 68 input frames they interpolate are read once,
// coalesced, batch-normalised and parked in LDS (pitch 65: the column reads of 64 lanes on ~consecutive frames do not conflict);
// then the lane is the output TIME, and every store instruction writes 256 contiguous bytes of one image row.
constexpr int kFrontRows = 68;
__global__ __launch_bounds__(256) void htsat_front64_kernel(const float* __restrict__ mel, long ld_clip, int in_t, int out_t, int img,
                                                            const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                            float* __restrict__ out) {
  __shared__ float tile[kFrontRows][65];
  const int tiles = out_t / 64;
  const long b = blockIdx.x / tiles;
  const int t0 = static_cast<int>(blockIdx.x % tiles) * 64;
  const float scale = static_cast<float>(in_t - 1) / static_cast<float>(out_t - 1);
  const int j_first = static_cast<int>(floorf(scale * t0)) - 1;
  const int n_rows = static_cast<int>(floorf(scale * (t0 + 63))) + 2 - j_first + 1;          // <= 68 while in_t <= out_t (host-checked)
  const float* src = mel + b * ld_clip;
  {
    const int f = threadIdx.x & 63;
    const float sc = bn_scale[f], sh = bn_shift[f];
    for (int rr = threadIdx.x >> 6; rr < n_rows; rr += 4) {
      int j = j_first + rr;
      j = j < 0 ? 0 : (j > in_t - 1 ? in_t - 1 : j);
      tile[rr][f] = src[static_cast<long>(j) * 64 + f] * sc + sh;
    }
  }
  __syncthreads();
  const int tp = threadIdx.x & 63, t = t0 + tp;
  const float real = scale * t;
  const int ix = static_cast<int>(floorf(real));
  const float tt = real - ix;
  const float A = -0.75f;
  const float w[4] = {cubic2(tt + 1.f, A), cubic1(tt, A), cubic1(1.f - tt, A), cubic2(2.f - tt, A)};
  const int rr0 = ix - 1 - j_first;
  const int r = t0 / img, tp0 = t0 - r * img;                 // 64 | img: the 64 times of a workgroup share their row block
  float* dst = out + (b * img + r * 64) * img + tp0 + tp;
  const int f0 = (threadIdx.x >> 6) * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int f = f0 + i;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += w[k] * tile[rr0 + k][f];
    dst[static_cast<long>(f) * img] = v;
  }
}

// ------------------------------------------------------------------------------------ patch embedding + LayerNorm
// token (i, j) of clip b: y[o] = bias[o] + sum_{di,dj} W[o][di*4+dj] * img[b][4i+di][4j+dj], then LayerNorm over o.
// One wave per token; lane o and o+64 hold the channels (C <= 128).
constexpr int kPeTok = 16;       // consecutive tokens of one image row per wave: the lane's 2 x 16 conv weights stay in registers
__global__ __launch_bounds__(256) void htsat_patch_embed_kernel(const float* __restrict__ img, int side, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, int C, float* __restrict__ out32,
                                                                unsigned short* __restrict__ out16, long n_groups) {
  const int lane = threadIdx.x & 63;
  const long grp = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (grp >= n_groups) return;
  const int grid = side / 4, gpr = grid / kPeTok;           // token grid side; groups per token row
  const int j0 = static_cast<int>(grp % gpr) * kPeTok;
  const int i = static_cast<int>((grp / gpr) % grid);
  const long b = grp / (static_cast<long>(gpr) * grid);
  float wr[2][16], bs[2], ga[2], be[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int o = lane + 64 * k;
    const bool on = o < C;
    bs[k] = on ? bias[o] : 0.f; ga[k] = on ? gamma[o] : 0.f; be[k] = on ? beta[o] : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) wr[k][e] = on ? w[o * 16 + e] : 0.f;
  }
  const float* rowp = img + (b * side + 4 * i) * side + 4 * j0;
  const long tok0 = (b * grid + i) * grid + j0;
  for (int t = 0; t < kPeTok; ++t) {
    float px[16];
#pragma unroll
    for (int di = 0; di < 4; ++di) {
      const float4 v = *reinterpret_cast<const float4*>(rowp + di * side + 4 * t);
      px[di * 4] = v.x; px[di * 4 + 1] = v.y; px[di * 4 + 2] = v.z; px[di * 4 + 3] = v.w;
    }
    float y[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float acc = bs[k];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc += wr[k][e] * px[e];
      y[k] = acc;
    }
    const float s = y[0] + (lane + 64 < C ? y[1] : 0.f);           // lanes >= C hold exact zeros in slot 0 (zero weights and bias)
    const float mean = wsum(s) / C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (lane + 64 * k < C) ss += (y[k] - mean) * (y[k] - mean);
    const float rstd = rsqrtf(wsum(ss) / C + eps);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int o = lane + 64 * k;
      if (o < C) {
        const float v = (y[k] - mean) * rstd * ga[k] + be[k];
        if (out32) out32[(tok0 + t) * C + o] = v;
        if (out16) out16[(tok0 + t) * C + o] = f2bf_h(v);
      }
    }
  }
}

// The same with the TOKEN on the lane (C = 96): a wave takes 64 consecutive tokens of one image row, every lane computes all 96
// channels of its own token.  The conv weights, bias, gamma and beta are wave-uniform (scalar loads feeding the FMAs as scalar
// operands), the LayerNorm statistics are per-lane sums (no cross-lane reductions), the four pixel rows of a token row are four
// coalesced KiB loads per wave.  ~33 vector instructions per token against ~60 plus 12 shuffles in the kernel above.
// The wave's 64 output rows are one contiguous 24 KiB block, but a lane holds a whole ROW of it: stored from the registers, every
// store instruction scatters 64 sixteen-byte pieces over 64 rows.  So the rows go through LDS in two halves of 48 channels (pitch 13
// float4: the b128 writes of eight consecutive lanes cover all 32 banks) and leave as 192-byte runs, twelve lanes per run.
template <int C>
__global__ __launch_bounds__(256) void htsat_patch_embed_tok_kernel(const float* __restrict__ img, int side, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float eps, float* __restrict__ out32,
                                                                    unsigned short* __restrict__ out16, long n_waves) {
  static_assert(C == 96, "two halves of twelve float4");
  __shared__ float4 tr[4][64 * 13];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long wv = static_cast<long>(blockIdx.x) * 4 + wave;
  const bool live = wv < n_waves;
  if (!live) wv = n_waves - 1;                                // (keeps the barriers below uniform; its stores are predicated off)
  const int grid = side / 4, wpr = grid / 64;               // token grid side; waves per token row
  const int j0 = static_cast<int>(wv % wpr) * 64;
  const int i = static_cast<int>((wv / wpr) % grid);
  const long b = wv / (static_cast<long>(wpr) * grid);
  float px[16];
  const float* rowp = img + (b * side + 4 * i) * side + 4 * (j0 + lane);
#pragma unroll
  for (int di = 0; di < 4; ++di) {
    const float4 v = *reinterpret_cast<const float4*>(rowp + di * side);
    px[di * 4] = v.x; px[di * 4 + 1] = v.y; px[di * 4 + 2] = v.z; px[di * 4 + 3] = v.w;
  }
  float y[C];
  float sum = 0.f;
#pragma unroll
  for (int o = 0; o < C; ++o) {
    float acc = bias[o];
    // (asm: left to the compiler, channel pairs are SLP-packed into v_pk_fma_f32 whose weight pairs -- 16 floats apart in memory --
    // are then assembled with a v_readlane and two s_mov each: more instructions than the FMAs themselves)
#pragma unroll
    for (int e = 0; e < 16; ++e) asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(w[o * 16 + e]), "v"(px[e]));
    y[o] = acc;
    sum += acc;
  }
  const float mean = sum * (1.0f / C);
  float ss = 0.f;
#pragma unroll
  for (int o = 0; o < C; ++o) { const float d = y[o] - mean; ss = fmaf(d, d, ss); }
  const float rstd = rsqrtf(ss * (1.0f / C) + eps);
  const long tok0 = (b * grid + i) * grid + j0;             // the wave's 64 rows are contiguous from here
  float4* my = tr[wave];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const int o = 48 * half + 4 * q;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf((y[o + e] - mean) * rstd, gamma[o + e], beta[o + e]);
      my[lane * 13 + q] = float4{v[0], v[1], v[2], v[3]};
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int idx = k * 64 + lane, t = idx / 12, q = idx - 12 * t;
      const float4 v = my[t * 13 + q];
      if (live) {
        if (out32) *reinterpret_cast<float4*>(out32 + (tok0 + t) * C + 48 * half + 4 * q) = v;
        if (out16) *reinterpret_cast<uint2*>(out16 + (tok0 + t) * C + 48 * half + 4 * q) = uint2{pack2_h(v.x, v.y), pack2_h(v.z, v.w)};
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ AFF fusion patch embedding
// The `is_longer` branch of ClapAudioPatchEmbed.forward for ONE clip: global = proj(img channel 0) (4x4/4), local =
// mel_conv2d(img channels 1..3) (4x12 / stride 4x12, 21 columns per channel, channel-major along the width, column 63 zero),
// fused by ClapAudioAFFBlock (eval BatchNorms folded into the 1x1 convs by the host) and LayerNorm'd.
// Three launches: conv (per token) -> global attention vector (per clip) -> apply (per token).
struct AffArgs {
  const float *pw, *pb;          // proj [C,16], [C]
  const float *cw, *cb;          // mel_conv2d [C,48], [C]
  const float *lw1, *lb1, *lw2, *lb2;   // local_att  1x1 convs with BN folded: [I,C],[I],[C,I],[C]
  const float *gw1, *gb1, *gw2, *gb2;   // global_att
  const float *gamma, *beta;
  float eps;
  int C, I, side;
};

__global__ __launch_bounds__(256) void aff_conv_kernel(const float* __restrict__ img_g, const float* __restrict__ img_l, AffArgs a,
                                                       float* __restrict__ ws_g, float* __restrict__ ws_l, int n_tokens) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= n_tokens) return;
  const int grid = a.side / 4, lw = a.side / 12;           // 64 tokens per row; 21 local columns per channel
  const int j = tok % grid, i = tok / grid;
  float px[16];
#pragma unroll
  for (int di = 0; di < 4; ++di)
#pragma unroll
    for (int dj = 0; dj < 4; ++dj) px[di * 4 + dj] = img_g[(4 * i + di) * a.side + 4 * j + dj];
  const int ch = j / lw, jj = j - ch * lw;
  const bool has_local = ch < 3;
  float pl[48];
#pragma unroll
  for (int di = 0; di < 4; ++di)
#pragma unroll
    for (int dj = 0; dj < 12; ++dj)
      pl[di * 12 + dj] = has_local ? img_l[(static_cast<long>(ch) * a.side + 4 * i + di) * a.side + 12 * jj + dj] : 0.f;
  for (int o = lane; o < a.C; o += 64) {
    float g = a.pb[o];
#pragma unroll
    for (int e = 0; e < 16; ++e) g += a.pw[o * 16 + e] * px[e];
    float l = 0.f;
    if (has_local) {
      l = a.cb[o];
#pragma unroll
      for (int e = 0; e < 48; ++e) l += a.cw[o * 48 + e] * pl[e];
    }
    ws_g[static_cast<long>(tok) * a.C + o] = g;
    ws_l[static_cast<long>(tok) * a.C + o] = l;
  }
}

// Column sums of (g + l) over the tokens in two stages: kAffParts workgroups each sum a contiguous stretch of tokens (one
// workgroup walking all 4096 tokens x 96 channels took 0.13 ms, a tenth of a 512-clip forward, for the ONE clip of a batch that
// goes through the fusion branch), then aff_global_kernel adds the partial rows in order.
constexpr int kAffParts = 64;
__global__ __launch_bounds__(256) void aff_colsum_partial_kernel(const float* __restrict__ ws_g, const float* __restrict__ ws_l, int C, int n_tokens,
                                                                 float* __restrict__ part) {
  __shared__ float red[256];
  const int per = (n_tokens + kAffParts - 1) / kAffParts;
  const int t0 = blockIdx.x * per, t1 = t0 + per < n_tokens ? t0 + per : n_tokens;
  const int groups = 256 / C;                          // token lanes (C = 96 -> 2)
  const int c = threadIdx.x % C, gidx = threadIdx.x / C;
  float s = 0.f;
  if (gidx < groups)
    for (int t = t0 + gidx; t < t1; t += groups) s += ws_g[static_cast<long>(t) * C + c] + ws_l[static_cast<long>(t) * C + c];
  red[threadIdx.x] = gidx < groups ? s : 0.f;
  __syncthreads();
  if (threadIdx.x < C) {
    float m = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) m += red[g2 * C + threadIdx.x];
    part[blockIdx.x * C + threadIdx.x] = m;
  }
}

// one block: mean over the tokens of (g + l) from the partial sums, then the global_att MLP -> ga[C]
__global__ __launch_bounds__(1024) void aff_global_kernel(const float* __restrict__ partial, AffArgs a, int n_tokens, float* __restrict__ ga) {
  __shared__ float part[1024];
  __shared__ float mean[128];
  __shared__ float hid[64];
  const int groups = 1024 / a.C;                       // row groups (C = 96 -> 10)
  const int c = threadIdx.x % a.C, gidx = threadIdx.x / a.C;
  float s = 0.f;
  if (gidx < groups)
    for (int t = gidx; t < kAffParts; t += groups) s += partial[t * a.C + c];
  part[threadIdx.x] = gidx < groups ? s : 0.f;
  __syncthreads();
  if (threadIdx.x < a.C) {
    float m = 0.f;
    for (int g2 = 0; g2 < groups; ++g2) m += part[g2 * a.C + threadIdx.x];
    mean[threadIdx.x] = m / n_tokens;
  }
  __syncthreads();
  if (threadIdx.x < a.I) {
    float h = a.gb1[threadIdx.x];
    for (int k = 0; k < a.C; ++k) h += a.gw1[threadIdx.x * a.C + k] * mean[k];
    hid[threadIdx.x] = h > 0.f ? h : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < a.C) {
    float z = a.gb2[threadIdx.x];
    for (int k = 0; k < a.I; ++k) z += a.gw2[threadIdx.x * a.I + k] * hid[k];
    ga[threadIdx.x] = z;
  }
}

__global__ __launch_bounds__(256) void aff_apply_kernel(const float* __restrict__ ws_g, const float* __restrict__ ws_l, const float* __restrict__ ga,
                                                        AffArgs a, float* __restrict__ out32, int n_tokens) {
  __shared__ float sa[4][128];
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tok = blockIdx.x * 4 + w;
  if (tok >= n_tokens) return;                         // whole waves leave together; LDS below is wave-private
  float g[2] = {0.f, 0.f}, l[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int o = lane + 64 * k;
    if (o < a.C) {
      g[k] = ws_g[static_cast<long>(tok) * a.C + o];
      l[k] = ws_l[static_cast<long>(tok) * a.C + o];
      sa[w][o] = g[k] + l[k];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < a.I) {
    float h = a.lb1[lane];
    for (int k = 0; k < a.C; ++k) h += a.lw1[lane * a.C + k] * sa[w][k];
    sh[w][lane] = h > 0.f ? h : 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float y[2] = {0.f, 0.f};
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int o = lane + 64 * k;
    if (o < a.C) {
      float z = a.lb2[o] + ga[o];
      for (int q = 0; q < a.I; ++q) z += a.lw2[o * a.I + q] * sh[w][q];
      const float wgt = 1.f / (1.f + expf(-z));
      y[k] = 2.f * g[k] * wgt + 2.f * l[k] * (1.f - wgt);
      s += y[k];
    }
  }
  const float mean = wsum(s) / a.C;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (lane + 64 * k < a.C) ss += (y[k] - mean) * (y[k] - mean);
  const float rstd = rsqrtf(wsum(ss) / a.C + a.eps);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int o = lane + 64 * k;
    if (o < a.C) out32[static_cast<long>(tok) * a.C + o] = (y[k] - mean) * rstd * a.gamma[o] + a.beta[o];
  }
}

// ------------------------------------------------------------------------------------ window attention (head_dim 24)
struct WinAttnArgs {
  const unsigned short* qkv; long ld;          // [B*R*R, 3C] bf16: q | k | v, head h at columns h*24
  unsigned short* ctx; long ldc;               // [B*R*R, C]
  const float* bias; int n_bias_windows;       // [n_bias_windows][heads][qt 2][kt 2][g 4][lane 64][4]: rel-pos bias (+ shift mask per window), see adt_hip.h
  int B, R, C, heads, shift; float scale;
};

__device__ __forceinline__ long token_row(const WinAttnArgs& a, int b, int wy, int wx, int t) {
  const int y = (wy * 8 + (t >> 3) + a.shift) % a.R, x = (wx * 8 + (t & 7) + a.shift) % a.R;
  return (static_cast<long>(b) * a.R + y) * a.R + x;
}
// 8 bf16 of a head slice at d = 8*c .. 8*c+7 (c = 0..3; c == 3 is the zero padding 24..31)
__device__ __forceinline__ bf16x8 head_chunk(const unsigned short* p, int c) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (c < 3) v = *reinterpret_cast<const uint4*>(p + 8 * c);
  return *reinterpret_cast<bf16x8*>(&v);
}

__global__ __launch_bounds__(256) void window_attn_kernel(WinAttnArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char vt_all[4][64 * 64];      // per wave: V tile [64 keys][32 d] bf16
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long item = static_cast<long>(blockIdx.x) * 4 + wave;                     // (b, wy, wx, head)
  const int nw = a.R / 8;
  const long total = static_cast<long>(a.B) * nw * nw * a.heads;
  if (item >= total) return;
  const int head = static_cast<int>(item % a.heads);
  long rest = item / a.heads;
  const int wx = static_cast<int>(rest % nw); rest /= nw;
  const int wy = static_cast<int>(rest % nw);
  const int b = static_cast<int>(rest / nw);
  unsigned char* vt = vt_all[wave];

  // operands: token t0 = r (tile 0) and t1 = r + 32 (tile 1) of the window
  const long row0 = token_row(a, b, wy, wx, r), row1 = token_row(a, b, wy, wx, r + 32);
  const unsigned short* q0 = a.qkv + row0 * a.ld + head * 24;
  const unsigned short* q1 = a.qkv + row1 * a.ld + head * 24;
  bf16x8 kf[2][2], qf[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {                       // k-step s covers d = 16s .. 16s+15; this lane holds 8h..8h+7 of it
    qf[0][s] = head_chunk(q0, 2 * s + h);           qf[1][s] = head_chunk(q1, 2 * s + h);
    kf[0][s] = head_chunk(q0 + a.C, 2 * s + h);     kf[1][s] = head_chunk(q1 + a.C, 2 * s + h);
  }
  // V tile: keys r and r+32, 32 d each (d >= 24 zero); lane half h writes chunks 2h, 2h+1 of both rows
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int ch = 2 * h + c;
    const bf16x8 v0 = head_chunk(q0 + 2 * a.C, ch), v1 = head_chunk(q1 + 2 * a.C, ch);
    *reinterpret_cast<bf16x8*>(vt + r * 64 + ch * 16) = v0;
    *reinterpret_cast<bf16x8*>(vt + (r + 32) * 64 + ch * 16) = v1;
  }
  f32x16 st[2][2];                                   // st[kt][qt] = S^T tile: rows keys, lane = query
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) st[kt][qt][i] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) st[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][s], qf[qt][s], st[kt][qt], 0, 0, 0);
    }
  const int wsel = a.n_bias_windows > 1 ? (wy * nw + wx) : 0;
  // lane-linear bias: the 16 floats of (query tile qt, key tile kt) that this lane adds to its accumulator are four 16-byte
  // pieces, each at lane * 16 inside a contiguous KiB -- one coalesced KiB per wave-load (a row-major [64][64] table costs 32
  // loads of 64 scattered 16-byte pieces per (window, head): the kernel was bound by them)
  const float* bias = a.bias + (static_cast<long>(wsel) * a.heads + head) * 4096 + lane * 4;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  f32x16 o[2];
  const float sl2 = a.scale * 1.4426950408889634f;     // scores in log2 units (the bias table already is)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + ((qt * 2 + kt) * 4 + g) * 256);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(st[kt][qt][4 * g + e], sl2, bb[e]);          // log2 domain: the bias table comes pre-multiplied by log2 e
          st[kt][qt][4 * g + e] = v;
          mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float p = __builtin_amdgcn_exp2f(st[kt][qt][i] - mx); st[kt][qt][i] = p; sum += p; }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[qt][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        // A = V^T: A[row d = lane&31][element j] = V[key = kt*32 + 16*s2 + 8*(j>>2) + 4h + (j&3)][d]
        const int i16 = lane & 15, g4 = (lane >> 4) & 1;
        const int krow = kt * 32 + 16 * s2 + 4 * h + (i16 >> 2);
        const unsigned base = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)vt)) +
                              krow * 64 + (16 * g4 + 4 * (i16 & 3)) * 2;
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi) : "v"(base) : "memory");
        bf16x8 af;
        af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3]; af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
        union { unsigned u[4]; bf16x8 v; } pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) pf.u[e] = pack2_h(st[kt][qt][8 * s2 + 2 * e], st[kt][qt][8 * s2 + 2 * e + 1]);      // un-normalised: O is scaled by 1 / sum below
        o[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf.v, o[qt], 0, 0, 0);
      }
    // O^T[d][query]: this lane owns d = 8g + 4h + (0..3); d >= 24 is padding
    unsigned short* dst = a.ctx + (qt == 0 ? row0 : row1) * a.ldc + head * 24;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      uint2 v;
      v.x = pack2_h(o[qt][4 * g + 0] * inv, o[qt][4 * g + 1] * inv);
      v.y = pack2_h(o[qt][4 * g + 2] * inv, o[qt][4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(dst + 8 * g + 4 * h) = v;
    }
  }
}

// ---- window attention, four heads per workgroup with the window's q | k | v slices staged through LDS.
// The kernel above reads a (window, head)'s operands straight from the packed qkv rows: 16-byte pieces 576+ bytes apart, twelve
// wave-loads of 64 scattered pieces each, and writes its output the same way -- it is bound by the number of memory requests, not
// by bytes.  Here a workgroup = one window x four consecutive heads (wave w = head 4 hg + w): the 64 tokens' three 192-byte runs
// (q, k, v of the four heads) are fetched with 16-byte loads that walk the runs contiguously, parked in LDS (row pitch 592 bytes:
// 37 sixteen-byte chunks, odd, so the row reads of 32 consecutive tokens are conflict-free), read from there as MFMA operands, and
// the four heads' outputs go back through the same buffer to 192-byte coalesced stores.  V^T fragments are read transposed straight
// from the staged rows (ds_read_b64_tr_b16 takes per-lane addresses, any pitch): the padding d = 24..31 of a head then reads the next
// head's first values (the zeroed pad chunk for the last head) -- finite numbers that only reach output rows d >= 24, which are never
// stored; likewise only Q carries explicit zeros for d >= 24 (0 x finite = 0 in S).  Without per-wave V tiles the workgroup needs
// 37 KiB of LDS and four of them share a CU (two before).  A thread fetches chunk cc of the q, the k and the v run of three tokens
// (twelve consecutive lanes walk one 192-byte run), so it computes three row indices, not nine, and keeps them for the output.
constexpr int kWa4Pitch = 592;                          // 3 x 192 + 16
constexpr int kWa4Stage = 64 * kWa4Pitch;               // 37,888 B
constexpr int kWa4Lds = kWa4Stage;

__global__ __launch_bounds__(256) void window_attn4_kernel(WinAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = a.R / 8, hgs = a.heads / 4;
  long item = blockIdx.x;                                                          // (b, wy, wx, head group)
  const int hg = static_cast<int>(item % hgs); item /= hgs;
  const int wx = static_cast<int>(item % nw); item /= nw;
  const int wy = static_cast<int>(item % nw);
  const int b = static_cast<int>(item / nw);
  const int head = 4 * hg + wave;
  unsigned char* stage = smem4;
  // ---- stage: 64 tokens x 36 chunks (q: 0..11, k: 12..23, v: 24..35 = the four heads' 24 values each)
  long st_row[3];
  int st_t[3], st_cc[3];
  {
    int t = tid / 12, cc = tid - 12 * t;                     // pass p covers ids 256 p + tid: 256 = 21 x 12 + 4
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      st_t[p] = t; st_cc[p] = cc;
      st_row[p] = token_row(a, b, wy, wx, t);
      const unsigned short* src = a.qkv + st_row[p] * a.ld + hg * 96 + cc * 8;
      unsigned char* dst = stage + t * kWa4Pitch + cc * 16;
#pragma unroll
      for (int which = 0; which < 3; ++which) *reinterpret_cast<uint4*>(dst + which * 192) = *reinterpret_cast<const uint4*>(src + which * a.C);
      const bool wrap = cc >= 8;
      cc = wrap ? cc - 8 : cc + 4;
      t += wrap ? 22 : 21;
    }
  }
  if (tid < 64) *reinterpret_cast<uint4*>(stage + tid * kWa4Pitch + 576) = make_uint4(0, 0, 0, 0);      // the pad chunk: the last head's V padding
  __syncthreads();
  // ---- operands of this wave's head: token r (tile 0) and r + 32 (tile 1); k-step s covers d = 16s .. 16s+15, the lane holds 8h .. 8h+7
  auto chunk = [&](int t, int which, int c) -> bf16x8 {                             // d = 8c .. 8c+7 of the head; c == 3 is the zero padding 24..31
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < 3) v = *reinterpret_cast<const uint4*>(stage + t * kWa4Pitch + which * 192 + wave * 48 + c * 16);
    return *reinterpret_cast<bf16x8*>(&v);
  };
  auto raw = [&](int t, int which, int c) -> bf16x8 {                               // the same without the padding test: chunk 3 of K is whatever
    const uint4 v = *reinterpret_cast<const uint4*>(stage + t * kWa4Pitch + which * 192 + wave * 48 + c * 16);      // follows the head (finite)
    return *reinterpret_cast<const bf16x8*>(&v);
  };
  bf16x8 kf[2][2], qf[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    qf[0][s] = chunk(r, 0, 2 * s + h);  qf[1][s] = chunk(r + 32, 0, 2 * s + h);
    kf[0][s] = raw(r, 1, 2 * s + h);    kf[1][s] = raw(r + 32, 1, 2 * s + h);
  }
  f32x16 st[2][2];                                   // st[kt][qt] = S^T tile: rows keys, lane = query
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) st[kt][qt][i] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) st[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][s], qf[qt][s], st[kt][qt], 0, 0, 0);
    }
  const int wsel = a.n_bias_windows > 1 ? (wy * nw + wx) : 0;
  const float* bias = a.bias + (static_cast<long>(wsel) * a.heads + head) * 4096;      // wave-uniform; the lane adds 4 * lane floats
  const unsigned bias_lane = static_cast<unsigned>(lane) * 4u;
  __syncthreads();                                    // every wave has its operands: the q part of the stage is free for the outputs
  f32x16 o[2];
  const float sl2 = a.scale * 1.4426950408889634f;     // scores in log2 units (the bias table already is)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + (bias_lane + static_cast<unsigned>(((qt * 2 + kt) * 4 + g) * 256)));
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(st[kt][qt][4 * g + e], sl2, bb[e]);          // log2 domain: the bias table comes pre-multiplied by log2 e
          st[kt][qt][4 * g + e] = v;
          mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float p = __builtin_amdgcn_exp2f(st[kt][qt][i] - mx); st[kt][qt][i] = p; sum += p; }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[qt][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int i16 = lane & 15, g4 = (lane >> 4) & 1;
        const int krow = kt * 32 + 16 * s2 + 4 * h + (i16 >> 2);
        const unsigned base = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)stage)) +
                              krow * kWa4Pitch + 384 + wave * 48 + (16 * g4 + 4 * (i16 & 3)) * 2;
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4736\n\ts_waitcnt lgkmcnt(0)"      // + 8 rows of 592 B
                     : "=&v"(lo), "=&v"(hi) : "v"(base) : "memory");
        bf16x8 af;
        af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3]; af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
        union { unsigned u[4]; bf16x8 v; } pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) pf.u[e] = pack2_h(st[kt][qt][8 * s2 + 2 * e], st[kt][qt][8 * s2 + 2 * e + 1]);      // un-normalised: O is scaled by 1 / sum below
        o[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf.v, o[qt], 0, 0, 0);
      }
    // O^T[d][query]: this lane owns d = 8g + 4h + (0..3), d < 24 -> the output rows in LDS: [token][4 heads x 24] bf16 (192 B, pitch 592)
    unsigned char* dst = stage + (qt * 32 + r) * kWa4Pitch + wave * 48;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      uint2 v;
      v.x = pack2_h(o[qt][4 * g + 0] * inv, o[qt][4 * g + 1] * inv);
      v.y = pack2_h(o[qt][4 * g + 2] * inv, o[qt][4 * g + 3] * inv);
      *reinterpret_cast<uint2*>(dst + (8 * g + 4 * h) * 2) = v;
    }
  }
  __syncthreads();
  // ---- 64 tokens x 12 chunks of 16 bytes -> ctx rows (192 contiguous bytes per token)
#pragma unroll
  for (int p = 0; p < 3; ++p)
    *reinterpret_cast<uint4*>(a.ctx + st_row[p] * a.ldc + hg * 96 + st_cc[p] * 8) = *reinterpret_cast<const uint4*>(stage + st_t[p] * kWa4Pitch + st_cc[p] * 16);
}

// ------------------------------------------------------------------------------------ patch merging gather + LayerNorm(4C)
// out token (b, i, j) = LN(concat[x(2i,2j), x(2i+1,2j), x(2i,2j+1), x(2i+1,2j+1)])   (ClapAudioPatchMerging order)
__global__ __launch_bounds__(256) void patch_merge_ln_kernel(const float* __restrict__ x, int R, int C, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, unsigned short* __restrict__ out,
                                                             long n_out) {
  const int lane = threadIdx.x & 63;
  const long tok = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (tok >= n_out) return;
  const int Ro = R / 2;
  const int j = static_cast<int>(tok % Ro), i = static_cast<int>((tok / Ro) % Ro);
  const long b = tok / (static_cast<long>(Ro) * Ro);
  const int D = 4 * C, nq = D / 4;
  float4 v[6];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int q = lane + 64 * k;
    v[k] = make_float4(0, 0, 0, 0);
    if (q < nq) {
      const int e = 4 * q, part = e / C, c = e - part * C;              // part: 0 (r0,c0) 1 (r1,c0) 2 (r0,c1) 3 (r1,c1)
      const int yy = 2 * i + (part & 1), xx = 2 * j + (part >> 1);
      v[k] = *reinterpret_cast<const float4*>(x + ((b * R + yy) * R + xx) * C + c);
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
  }
  const float mean = wsum(s) / D;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (lane + 64 * k < nq) {
      const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
      ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  const float rstd = rsqrtf(wsum(ss) / D + eps);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int q = lane + 64 * k;
    if (q >= nq) continue;
    const float4 g = reinterpret_cast<const float4*>(gamma)[q], bt = reinterpret_cast<const float4*>(beta)[q];
    uint2 o;
    o.x = pack2_h((v[k].x - mean) * rstd * g.x + bt.x, (v[k].y - mean) * rstd * g.y + bt.y);
    o.y = pack2_h((v[k].z - mean) * rstd * g.z + bt.z, (v[k].w - mean) * rstd * g.w + bt.w);
    reinterpret_cast<uint2*>(out + tok * D)[q] = o;
  }
}

// ------------------------------------------------------------------------------------ mean over tokens, L2 normalise
__global__ __launch_bounds__(256) void mean_tokens_kernel(const float* __restrict__ x, int T, int C, float* __restrict__ out32,
                                                          unsigned short* __restrict__ out16) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += x[(static_cast<long>(b) * T + t) * C + c];
    s /= T;
    if (out32) out32[static_cast<long>(b) * C + c] = s;
    if (out16) out16[static_cast<long>(b) * C + c] = f2bf_h(s);
  }
}
// LayerNorm of every token row followed by the mean over the clip's tokens (the tower's last two steps: ClapAudioEncoder.norm + the avg-pool head)
// in one pass: one workgroup per clip, a wave per token in turn (the row in registers: D <= 1024 floats, the arithmetic of layernorm_fwd_kernel),
// the normalised rows accumulate per lane and are never written.
__global__ __launch_bounds__(256) void ln_mean_tokens_kernel(const float* __restrict__ x, int T, int D, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float* __restrict__ out32,
                                                             unsigned short* __restrict__ out16) {
  __shared__ float4 red[4][256];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  float4 acc[4], g[4], be[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    acc[i] = make_float4(0, 0, 0, 0);
    g[i] = q < nq ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0, 0, 0, 0);
    be[i] = q < nq ? reinterpret_cast<const float4*>(beta)[q] : make_float4(0, 0, 0, 0);
  }
  for (int t = wave; t < T; t += 4) {
    const float* row = x + (static_cast<long>(b) * T + t) * D;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = lane + 64 * i;
      v[i] = q < nq ? reinterpret_cast<const float4*>(row)[q] : make_float4(0, 0, 0, 0);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wsum(s) / D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < nq) {
        const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
        ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    const float rstd = rsqrtf(wsum(ss) / D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i].x += (v[i].x - mean) * rstd * g[i].x + be[i].x; acc[i].y += (v[i].y - mean) * rstd * g[i].y + be[i].y;
      acc[i].z += (v[i].z - mean) * rstd * g[i].z + be[i].z; acc[i].w += (v[i].w - mean) * rstd * g[i].w + be[i].w;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][lane + 64 * i] = acc[i];
  __syncthreads();
  for (int q = threadIdx.x; q < nq; q += 256) {
    const float4 a0 = red[0][q], a1 = red[1][q], a2 = red[2][q], a3 = red[3][q];
    const float inv = 1.0f / T;
    const float4 m = make_float4(((a0.x + a1.x) + (a2.x + a3.x)) * inv, ((a0.y + a1.y) + (a2.y + a3.y)) * inv, ((a0.z + a1.z) + (a2.z + a3.z)) * inv,
                                 ((a0.w + a1.w) + (a2.w + a3.w)) * inv);
    if (out32) reinterpret_cast<float4*>(out32 + static_cast<long>(b) * D)[q] = m;
    if (out16) reinterpret_cast<ushort4*>(out16 + static_cast<long>(b) * D)[q] = make_ushort4(f2bf_h(m.x), f2bf_h(m.y), f2bf_h(m.z), f2bf_h(m.w));
  }
}
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, int D, float* __restrict__ out, long n_rows) {
  const int lane = threadIdx.x & 63;
  const long row = static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = x[row * D + c]; s += v * v; }
  const float inv = 1.0f / sqrtf(wsum(s));
  for (int c = lane; c < D; c += 64) out[row * D + c] = x[row * D + c] * inv;
}

}  // namespace adt

using namespace adt;
#define STR(s) static_cast<hipStream_t>(s)

extern "C" int adt_htsat_front_f32(const float* mel, int64_t ld_clip, int64_t B, int32_t in_frames, int32_t n_mels, int32_t out_frames,
                                   int32_t img_side, const float* bn_scale, const float* bn_shift, float* img, void* stream) {
  if (!mel || !bn_scale || !bn_shift || !img) return set_error(ADT_EINVAL, "adt_htsat_front_f32: null pointer");
  if (B < 0 || in_frames < 2 || n_mels <= 0 || out_frames < 2 || img_side <= 0 || img_side % n_mels || out_frames != img_side * (img_side / n_mels))
    return set_error(ADT_ESHAPE, "adt_htsat_front_f32: need img_side % n_mels == 0 and out_frames == img_side * img_side / n_mels");
  const long total = B * static_cast<long>(out_frames) * n_mels;
  if (total == 0) return ADT_OK;
  static const bool no64 = [] { const char* v = getenv("ADT_HTSAT_NO_FRONT64"); return v && v[0] == '1'; }();
  if (n_mels == 64 && img_side % 64 == 0 && in_frames <= out_frames && !no64)
    hipLaunchKernelGGL(htsat_front64_kernel, dim3(static_cast<unsigned>(B * (out_frames / 64))), dim3(256), 0, STR(stream), mel, ld_clip, in_frames,
                       out_frames, img_side, bn_scale, bn_shift, img);
  else
    hipLaunchKernelGGL(htsat_front_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, STR(stream), mel, ld_clip, in_frames,
                       n_mels, out_frames, img_side, bn_scale, bn_shift, img, total);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_htsat_patch_embed(const float* img, int64_t B, int32_t img_side, const float* w, const float* bias, const float* gamma,
                                     const float* beta, float eps, int32_t C, float* out32, void* out16, void* stream) {
  if (!img || !w || !bias || !gamma || !beta || (!out32 && !out16)) return set_error(ADT_EINVAL, "adt_htsat_patch_embed: null pointer");
  if (B < 0 || img_side <= 0 || (img_side % (4 * kPeTok)) || C <= 0 || C > 128 || !aligned16(img))
    return set_error(ADT_ESHAPE, "adt_htsat_patch_embed: img_side % 64 == 0, C <= 128, 16-byte aligned image");
  if (C == 96 && img_side % 256 == 0 && (!out32 || aligned16(out32)) && (!out16 || (reinterpret_cast<uintptr_t>(out16) & 7) == 0)) {
    const long n_waves = B * static_cast<long>(img_side / 4) * (img_side / 4) / 64;       // 64 tokens of one row per wave
    if (n_waves == 0) return ADT_OK;
    hipLaunchKernelGGL((htsat_patch_embed_tok_kernel<96>), dim3(static_cast<unsigned>((n_waves + 3) / 4)), dim3(256), 0, STR(stream), img, img_side, w,
                       bias, gamma, beta, eps, out32, static_cast<unsigned short*>(out16), n_waves);
    ADT_HIP_TRY(hipGetLastError());
    return ADT_OK;
  }
  const long n = B * static_cast<long>(img_side / 4) * (img_side / 4) / kPeTok;
  if (n == 0) return ADT_OK;
  hipLaunchKernelGGL(htsat_patch_embed_kernel, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0, STR(stream), img, img_side, w, bias, gamma,
                     beta, eps, C, out32, static_cast<unsigned short*>(out16), n);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" size_t adt_htsat_fusion_embed_workspace_bytes(int32_t img_side, int32_t C) {
  if (img_side <= 0 || C <= 0) return 0;
  const size_t n = static_cast<size_t>(img_side / 4) * (img_side / 4);
  return (2 * n * C + 128 + 64 * static_cast<size_t>(C)) * sizeof(float);       // g, l, the gate vector, 64 partial column sums
}

extern "C" int adt_htsat_fusion_embed(const float* img_global, const float* img_local, int32_t img_side, const adt_aff_weights* w, float eps,
                                      int32_t C, int32_t inter, void* ws, size_t ws_bytes, float* out32, void* stream) {
  if (!img_global || !img_local || !w || !ws || !out32) return set_error(ADT_EINVAL, "adt_htsat_fusion_embed: null pointer");
  const float* const* wp = reinterpret_cast<const float* const*>(w);
  for (int i = 0; i < 14; ++i)
    if (!wp[i]) return set_error(ADT_EINVAL, "adt_htsat_fusion_embed: null weight pointer");
  if (img_side <= 0 || (img_side & 3) || img_side / 12 * 3 > img_side / 4 || C <= 0 || C > 128 || inter <= 0 || inter > 64 || C * (1024 / C) > 1024)
    return set_error(ADT_ESHAPE, "adt_htsat_fusion_embed: img_side % 4 == 0, C <= 128, inter <= 64");
  if (ws_bytes < adt_htsat_fusion_embed_workspace_bytes(img_side, C)) return set_error(ADT_EINVAL, "adt_htsat_fusion_embed: workspace too small");
  const int n = (img_side / 4) * (img_side / 4);
  float* ws_g = static_cast<float*>(ws);
  float* ws_l = ws_g + static_cast<size_t>(n) * C;
  float* ga = ws_l + static_cast<size_t>(n) * C;
  AffArgs a{w->proj_w, w->proj_b, w->conv_w, w->conv_b, w->local_w1, w->local_b1, w->local_w2, w->local_b2,
            w->global_w1, w->global_b1, w->global_w2, w->global_b2, w->ln_gamma, w->ln_beta, eps, C, inter, img_side};
  hipLaunchKernelGGL(aff_conv_kernel, dim3((n + 3) / 4), dim3(256), 0, STR(stream), img_global, img_local, a, ws_g, ws_l, n);
  float* partial = ga + 128;
  hipLaunchKernelGGL(aff_colsum_partial_kernel, dim3(kAffParts), dim3(256), 0, STR(stream), ws_g, ws_l, C, n, partial);
  hipLaunchKernelGGL(aff_global_kernel, dim3(1), dim3(1024), 0, STR(stream), partial, a, n, ga);
  hipLaunchKernelGGL(aff_apply_kernel, dim3((n + 3) / 4), dim3(256), 0, STR(stream), ws_g, ws_l, ga, a, out32, n);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_window_attn_fwd(const void* qkv, int64_t ld_qkv, void* ctx, int64_t ld_ctx, const float* bias, int32_t n_bias_windows,
                                   int64_t B, int32_t R, int32_t C, int32_t heads, int32_t shift, float scale, void* stream) {
  if (!qkv || !ctx || !bias) return set_error(ADT_EINVAL, "adt_window_attn_fwd: null pointer");
  if (B < 0 || R <= 0 || (R & 7) || heads <= 0 || C != heads * 24 || shift < 0 || shift >= 8)
    return set_error(ADT_ESHAPE, "adt_window_attn_fwd: window 8, head_dim 24 (C == heads*24), R % 8 == 0");
  if (ld_qkv < 3 * C || ld_ctx < C || (ld_qkv & 7) || (ld_ctx & 3) || !aligned16(qkv))
    return set_error(ADT_ESHAPE, "adt_window_attn_fwd: bad leading dimensions");
  const int nw = R / 8;
  if (n_bias_windows != 1 && n_bias_windows != nw * nw) return set_error(ADT_EINVAL, "adt_window_attn_fwd: n_bias_windows must be 1 or (R/8)^2");
  const long total = B * nw * nw * heads;
  if (total == 0) return ADT_OK;
  WinAttnArgs a{static_cast<const unsigned short*>(qkv), ld_qkv, static_cast<unsigned short*>(ctx), ld_ctx, bias, n_bias_windows,
                static_cast<int>(B), R, C, heads, shift, scale};
  // four heads per workgroup with LDS-staged operands (every HTSAT stage has a multiple of four heads); ADT_WINATTN=1: the
  // one-wave-per-(window, head) kernel, kept as the A/B arm
  static const int variant = [] { const char* v = getenv("ADT_WINATTN"); return v ? atoi(v) : 4; }();
  if (variant == 4 && heads % 4 == 0 && !(ld_ctx & 7) && aligned16(ctx)) {
    static thread_local int done_for = -1;
    int dev = 0;
    ADT_HIP_TRY(hipGetDevice(&dev));
    if (done_for != dev) {
      ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      done_for = dev;
    }
    static const int lds4 = [] { const char* v = getenv("ADT_WA4_LDS"); return v ? atoi(v) : kWa4Lds; }();      // (A/B: 54272 = two workgroups per CU)
    hipLaunchKernelGGL(window_attn4_kernel, dim3(static_cast<unsigned>(total / 4)), dim3(256), lds4, STR(stream), a);
  } else {
    hipLaunchKernelGGL(window_attn_kernel, dim3(static_cast<unsigned>((total + 3) / 4)), dim3(256), 0, STR(stream), a);
  }
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_patch_merge_ln(const float* x, int64_t B, int32_t R, int32_t C, const float* gamma, const float* beta, float eps,
                                  void* out_bf16, void* stream) {
  if (!x || !gamma || !beta || !out_bf16) return set_error(ADT_EINVAL, "adt_patch_merge_ln: null pointer");
  if (B < 0 || R <= 0 || (R & 1) || C <= 0 || (C & 3) || 4 * C > 1536) return set_error(ADT_ESHAPE, "adt_patch_merge_ln: R even, C % 4 == 0, 4C <= 1536");
  const long n = B * static_cast<long>(R / 2) * (R / 2);
  if (n == 0) return ADT_OK;
  hipLaunchKernelGGL(patch_merge_ln_kernel, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0, STR(stream), x, R, C, gamma, beta, eps,
                     static_cast<unsigned short*>(out_bf16), n);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_mean_tokens(const float* x, int64_t B, int32_t T, int32_t C, float* out32, void* out16, void* stream) {
  if (!x || (!out32 && !out16)) return set_error(ADT_EINVAL, "adt_mean_tokens: null pointer");
  if (B < 0 || T <= 0 || C <= 0) return set_error(ADT_EINVAL, "adt_mean_tokens: bad sizes");
  if (B == 0) return ADT_OK;
  hipLaunchKernelGGL(mean_tokens_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, STR(stream), x, T, C, out32, static_cast<unsigned short*>(out16));
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_ln_mean_tokens(const float* x, int64_t B, int32_t T, int32_t D, const float* gamma, const float* beta, float eps, float* out32,
                                  void* out16, void* stream) {
  if (!x || !gamma || !beta || (!out32 && !out16)) return set_error(ADT_EINVAL, "adt_ln_mean_tokens: null pointer");
  if (B < 0 || T <= 0 || D <= 0 || (D & 3) || D > 1024) return set_error(ADT_ESHAPE, "adt_ln_mean_tokens: D must be a multiple of 4, at most 1024");
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta) || (out32 && !aligned16(out32)) || (out16 && (reinterpret_cast<uintptr_t>(out16) & 7)))
    return set_error(ADT_EINVAL, "adt_ln_mean_tokens: misaligned pointer");
  if (B == 0) return ADT_OK;
  hipLaunchKernelGGL(ln_mean_tokens_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, STR(stream), x, T, D, gamma, beta, eps, out32,
                     static_cast<unsigned short*>(out16));
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_l2_normalize(const float* x, int64_t n_rows, int32_t D, float* out, void* stream) {
  if (!x || !out) return set_error(ADT_EINVAL, "adt_l2_normalize: null pointer");
  if (n_rows < 0 || D <= 0) return set_error(ADT_EINVAL, "adt_l2_normalize: bad sizes");
  if (n_rows == 0) return ADT_OK;
  hipLaunchKernelGGL(l2_normalize_kernel, dim3(static_cast<unsigned>((n_rows + 3) / 4)), dim3(256), 0, STR(stream), x, D, out, n_rows);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
