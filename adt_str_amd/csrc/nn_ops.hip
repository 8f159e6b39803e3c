// nn_ops.hip -- K6/K7/K8 and the optimizer: the HBM-bound row kernels of the ADT network (gfx950).
//
//   layernorm fwd/bwd   nn.LayerNorm(768) of the post-norm layers and Encoder.layer_norm
//                       (reference model.py:113-115,118-127,159-168; eps 1e-5, fp32 statistics)
//   embed + PE fwd/bwd  TokenEmbedding_plain * sqrt(d) + PositionalEncoding (model.py:42-65,171)
//   cross-entropy       ADTModel._loss_fn (model.py:228-238): fp32, nan_to_num, ignore_index=1, mean
//   colsum              bias gradients (sum over rows of a bf16 [M,N] gradient)
//   cast / transpose    fp32 master weights -> bf16 W and W^T operands for the GEMMs
//   sumsq + AdamW       torch.optim.AdamW step with global-norm clipping (train.py:219-249:
//                       adamw_torch, weight_decay 1e-5, max_grad_norm 1.0) on flat fp32 buffers;
//                       the clip factor stays on the device (no host sync)
//
// All are one-pass streaming kernels: 16-byte accesses, one wave per row for the row-wise ones,
// two-stage (partials + fixed-order reduce) column/scalar reductions so results are reproducible.
#include <hip/hip_runtime.h>

#include "adt_common.h"
#include "dropout.h"

namespace adt {

__device__ __forceinline__ float bf2f_(unsigned short v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }
__device__ __forceinline__ unsigned short f2bf_(float f) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f)); }
// Wave-wide reductions on the DPP network (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 / :31 carry the row
// totals forward; the total ends up in lane 63 and is broadcast through an SGPR): ~7 VALU instructions instead of six
// ds_bpermute round trips, which dominated the per-row latency of the LayerNorm kernels.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), kCtrl, kRowMask, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0x111, 0xf>(0.f, v);
  v += dpp_mov<0x112, 0xf>(0.f, v);
  v += dpp_mov<0x114, 0xf>(0.f, v);
  v += dpp_mov<0x118, 0xf>(0.f, v);
  v += dpp_mov<0x142, 0xa>(0.f, v);
  v += dpp_mov<0x143, 0xc>(0.f, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {       // (identity for max over the sources a lane does not have: itself)
  v = fmaxf(v, dpp_mov<0x111, 0xf>(v, v));
  v = fmaxf(v, dpp_mov<0x112, 0xf>(v, v));
  v = fmaxf(v, dpp_mov<0x114, 0xf>(v, v));
  v = fmaxf(v, dpp_mov<0x118, 0xf>(v, v));
  v = fmaxf(v, dpp_mov<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_mov<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr int kRowThreads = 256;            // 4 waves = 4 rows per block
constexpr int kMaxD = 1024;                 // row width limit of the one-wave-per-row kernels (4 float4 per lane)

// ------------------------------------------------------------------------------------ LayerNorm
// y = (x - mean) * rstd * gamma + beta ; stats in fp32, biased variance (torch semantics).
struct LnFwdArgs { const float* x; long ldx; const float* gamma; const float* beta; float eps;
                   float* y32; unsigned short* y16; long ldy; float* mean; float* rstd; int M; int D; Drop drop; };

__global__ __launch_bounds__(kRowThreads) void layernorm_fwd_kernel(LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const float* x = a.x + static_cast<long>(row) * a.ldx;
  float4 v[4];
  const int nq = a.D >> 2;                  // float4 per row
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    // non-temporal: the pre-LayerNorm rows are not read again before the backward pass, and kept in the Infinity Cache they displace this
    // kernel's own output, which the next launch reads (the FFN-1 GEMM behind it: 0.42 -> 0.394 ms in the step; -0.13 ms per step)
    typedef float f32x4_nl __attribute__((ext_vector_type(4)));
    if (q < nq) { const f32x4_nl t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nl*>(x) + q); v[i] = make_float4(t.x, t.y, t.z, t.w); }
    else v[i] = make_float4(0, 0, 0, 0);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / a.D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (lane + 64 * i < nq) {
      const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
      ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / a.D + a.eps);
  if (lane == 0) { if (a.mean) a.mean[row] = mean; if (a.rstd) a.rstd[row] = rstd; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    if (q >= nq) continue;
    const float4 g = reinterpret_cast<const float4*>(a.gamma)[q], b = reinterpret_cast<const float4*>(a.beta)[q];
    float4 y;
    y.x = (v[i].x - mean) * rstd * g.x + b.x; y.y = (v[i].y - mean) * rstd * g.y + b.y;
    y.z = (v[i].z - mean) * rstd * g.z + b.z; y.w = (v[i].w - mean) * rstd * g.w + b.w;
    if (a.drop.on()) {
      const uint64_t e0 = static_cast<uint64_t>(row) * a.D + 4 * q;
      float k4[4]; a.drop.scale4(e0, k4); y.x *= k4[0]; y.y *= k4[1]; y.z *= k4[2]; y.w *= k4[3];
    }
    if (a.y32) reinterpret_cast<float4*>(a.y32 + static_cast<long>(row) * a.ldy)[q] = y;
    if (a.y16) {
      ushort4 h = make_ushort4(f2bf_(y.x), f2bf_(y.y), f2bf_(y.z), f2bf_(y.w));
      reinterpret_cast<ushort4*>(a.y16 + static_cast<long>(row) * a.ldy)[q] = h;
    }
  }
}

// Narrow rows (D <= 4 * G, G = 16 or 32 lanes per row): 64 / G rows per wave, one float4 per lane, so the 96- and
// 192-channel stages of the audio Swin keep the wave's lanes and its memory pipeline busy.
template <int G>
__global__ __launch_bounds__(kRowThreads) void layernorm_fwd_narrow_kernel(LnFwdArgs a) {
  constexpr int kRowsPerBlock = kRowThreads / G;
  const int sub = threadIdx.x % G;
  const long row = static_cast<long>(blockIdx.x) * kRowsPerBlock + threadIdx.x / G;
  const bool live = row < a.M;
  const int nq = a.D >> 2;
  const bool has = live && sub < nq;
  float4 v = make_float4(0, 0, 0, 0);
  if (has) v = reinterpret_cast<const float4*>(a.x + row * a.ldx)[sub];
  float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / a.D;
  float ss = 0.f;
  if (has) {
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    ss = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float rstd = rsqrtf(ss / a.D + a.eps);
  if (live && sub == 0) { if (a.mean) a.mean[row] = mean; if (a.rstd) a.rstd[row] = rstd; }
  if (!has) return;
  const float4 g = reinterpret_cast<const float4*>(a.gamma)[sub], b = reinterpret_cast<const float4*>(a.beta)[sub];
  float4 y;
  y.x = (v.x - mean) * rstd * g.x + b.x; y.y = (v.y - mean) * rstd * g.y + b.y;
  y.z = (v.z - mean) * rstd * g.z + b.z; y.w = (v.w - mean) * rstd * g.w + b.w;
  if (a.drop.on()) {
    const uint64_t e0 = static_cast<uint64_t>(row) * a.D + 4 * sub;
    float k4[4]; a.drop.scale4(e0, k4); y.x *= k4[0]; y.y *= k4[1]; y.z *= k4[2]; y.w *= k4[3];
  }
  if (a.y32) reinterpret_cast<float4*>(a.y32 + row * a.ldy)[sub] = y;
  if (a.y16) reinterpret_cast<ushort4*>(a.y16 + row * a.ldy)[sub] = make_ushort4(f2bf_(y.x), f2bf_(y.y), f2bf_(y.z), f2bf_(y.w));
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd
// Per block (kLnBwdRows rows) partial column sums: dgamma += dy * xhat, dbeta += dy, dxsum += dx
// (dxsum = bias gradient of the linear layer that feeds this norm's input).
// rows per block (a quarter per wave): 64 for the long encoder streams, 16 when 64 would leave most CUs without a block
__host__ __device__ inline int ln_bwd_rows(long M) { return M >= 32768 ? 64 : 16; }
// dx16 is the branch-gradient output: bf16 (TB = unsigned short) on the default path, fp32 (TB = float) on the fp32-operand path
struct LnBwdArgs { const float* dy; long lddy; const float* x; long ldx; const float* gamma; const float* mean;
                   const float* rstd; float* dx32; void* dx16; long lddx; float* partial; int M; int D;
                   Drop dy_drop, dx_drop; };
__device__ __forceinline__ void store4_(unsigned short* p, const float4& d) {
  *reinterpret_cast<ushort4*>(p) = make_ushort4(f2bf_(d.x), f2bf_(d.y), f2bf_(d.z), f2bf_(d.w));
}
__device__ __forceinline__ void store4_(float* p, const float4& d) { *reinterpret_cast<float4*>(p) = d; }

template <typename TB>
__global__ __launch_bounds__(kRowThreads, 3) void layernorm_bwd_kernel(LnBwdArgs a) {
  __shared__ float red[3][4][kMaxD];        // [dgamma|dbeta|dxsum][wave][col]  = 48 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = a.D >> 2;
  float4 pg[4], pb[4], px[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pg[i] = pb[i] = px[i] = make_float4(0, 0, 0, 0);
  const int rpb = ln_bwd_rows(a.M);
  // Rows are software-pipelined: the loads of the wave's next row are issued before this row's arithmetic, so two rows' worth of
  // requests are in flight per wave (three waves per SIMD fit; one row each left the memory system under-subscribed at 3.7 TB/s).
  const int row_first = blockIdx.x * rpb + wave * (rpb / 4);
  float4 nxv[4], ndv[4];
  float nmean = 0.f, nrstd = 0.f;
  auto fetch = [&](int row) {
    if (row >= a.M) return;
    nmean = a.mean[row]; nrstd = a.rstd[row];
    const float4* xr = reinterpret_cast<const float4*>(a.x + static_cast<long>(row) * a.ldx);
    const float4* dyr = reinterpret_cast<const float4*>(a.dy + static_cast<long>(row) * a.lddy);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = lane + 64 * i;
      // (non-temporal, as in the forward kernel: both rows are dead after this read; -0.08 ms per step)
      typedef float f32x4_nl __attribute__((ext_vector_type(4)));
      if (q < nq) {
        const f32x4_nl tx = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nl*>(xr) + q), td = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nl*>(dyr) + q);
        nxv[i] = make_float4(tx.x, tx.y, tx.z, tx.w); ndv[i] = make_float4(td.x, td.y, td.z, td.w);
      }
    }
  };
  fetch(row_first);
  for (int rr = 0; rr < rpb / 4; ++rr) {
    const int row = row_first + rr;
    if (row >= a.M) break;
    const float mean = nmean, rstd = nrstd;
    float4 cxv[4], cdv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { cxv[i] = nxv[i]; cdv[i] = ndv[i]; }
    if (rr + 1 < rpb / 4) fetch(row + 1);
    float4 xh[4], g[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = lane + 64 * i;
      xh[i] = g[i] = make_float4(0, 0, 0, 0);
      if (q >= nq) continue;
      const float4 xv = cxv[i], gm = reinterpret_cast<const float4*>(a.gamma)[q];
      float4 dv = cdv[i];
      if (a.dy_drop.on()) {
        const uint64_t e0 = static_cast<uint64_t>(row) * a.D + 4 * q;
        float k4[4]; a.dy_drop.scale4(e0, k4); dv.x *= k4[0]; dv.y *= k4[1]; dv.z *= k4[2]; dv.w *= k4[3];
      }
      xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      g[i] = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
      s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      pg[i].x += dv.x * xh[i].x; pg[i].y += dv.y * xh[i].y; pg[i].z += dv.z * xh[i].z; pg[i].w += dv.w * xh[i].w;
      pb[i].x += dv.x; pb[i].y += dv.y; pb[i].z += dv.z; pb[i].w += dv.w;
    }
    const float m1 = wave_sum(s1) / a.D, m2 = wave_sum(s2) / a.D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = lane + 64 * i;
      if (q >= nq) continue;
      float4 d;
      d.x = rstd * (g[i].x - m1 - xh[i].x * m2); d.y = rstd * (g[i].y - m1 - xh[i].y * m2);
      d.z = rstd * (g[i].z - m1 - xh[i].z * m2); d.w = rstd * (g[i].w - m1 - xh[i].w * m2);
      if (a.dx32) reinterpret_cast<float4*>(a.dx32 + static_cast<long>(row) * a.lddx)[q] = d;
      if (a.dx_drop.on()) {           // gradient of the dropped branch: bf16 dx and the bias gradient see the mask
        const uint64_t e0 = static_cast<uint64_t>(row) * a.D + 4 * q;
        float k4[4]; a.dx_drop.scale4(e0, k4); d.x *= k4[0]; d.y *= k4[1]; d.z *= k4[2]; d.w *= k4[3];
      }
      px[i].x += d.x; px[i].y += d.y; px[i].z += d.z; px[i].w += d.w;
      if (a.dx16) store4_(static_cast<TB*>(a.dx16) + static_cast<long>(row) * a.lddx + 4 * q, d);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = lane + 64 * i;
    if (q >= nq) continue;
    reinterpret_cast<float4*>(red[0][wave])[q] = pg[i];
    reinterpret_cast<float4*>(red[1][wave])[q] = pb[i];
    reinterpret_cast<float4*>(red[2][wave])[q] = px[i];
  }
  __syncthreads();
  float* out = a.partial + static_cast<long>(blockIdx.x) * 3 * a.D;
  for (int c = threadIdx.x; c < 3 * a.D; c += kRowThreads) {
    const int k = c / a.D, col = c - k * a.D;
    out[c] = (red[k][0][col] + red[k][1][col]) + (red[k][2][col] + red[k][3][col]);
  }
}

// out[c] = sum_p partial[p][c] in a fixed order; c < width.  A block owns 32 columns and splits the
// partial rows over 8 row-lanes (coalesced 128-byte reads), then combines the 8 sums through LDS.
constexpr int kRedCols = 16;                 // columns per block (64-byte row pieces), 16 row-lanes each
__device__ __forceinline__ void reduce_partials_block(const float* __restrict__ partial, int n_part, int width,
                                                      float* __restrict__ out0, float* __restrict__ out1,
                                                      float* __restrict__ out2, int D, int block) {
  __shared__ float red[16][kRedCols + 1];
  const int cl = threadIdx.x & (kRedCols - 1), rl = threadIdx.x / kRedCols;
  const int c = block * kRedCols + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                      // four independent chains: the loads of a lane overlap
  if (c < width) {
    const float* p0 = partial + c;
    int p = rl;
    for (; p + 48 < n_part; p += 64) {
      s0 += p0[static_cast<long>(p) * width];
      s1 += p0[static_cast<long>(p + 16) * width];
      s2 += p0[static_cast<long>(p + 32) * width];
      s3 += p0[static_cast<long>(p + 48) * width];
    }
    for (; p < n_part; p += 16) s0 += p0[static_cast<long>(p) * width];
  }
  red[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < width) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    const int k = c / D, col = c - k * D;
    float* o = k == 0 ? out0 : (k == 1 ? out1 : out2);
    if (o) o[col] = t;
  }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int n_part, int width,
                                                              float* __restrict__ out0, float* __restrict__ out1,
                                                              float* __restrict__ out2, int D) {
  reduce_partials_block(partial, n_part, width, out0, out1, out2, D, blockIdx.x);
}

// The same reduction for a batch of independent (partial, outputs) entries in ONE launch: the deferred second stages of a
// gradient segment (adt_reduce_queue_*).  Entry e owns blocks [blk_start[e], blk_start[e + 1]); same summation order as above.
constexpr int kRedBatch = 56;                // 56 * 48 + 57 * 4 bytes of kernel arguments (< 4 KiB)
struct RedEntry { const float* partial; float* out0; float* out1; float* out2; int n_part, width, D, pad; };
struct RedBatchArgs { RedEntry e[kRedBatch]; int blk_start[kRedBatch + 1]; };
__global__ __launch_bounds__(256) void reduce_partials_batch_kernel(RedBatchArgs a, int n_entries) {
  int e = 0;
  while (e + 1 < n_entries && static_cast<int>(blockIdx.x) >= a.blk_start[e + 1]) ++e;     // block-uniform, scalar
  const RedEntry& r = a.e[e];
  reduce_partials_block(r.partial, r.n_part, r.width, r.out0, r.out1, r.out2, r.D, static_cast<int>(blockIdx.x) - a.blk_start[e]);
}

// ------------------------------------------------------------------------------------ column sums of a bf16 matrix
constexpr int kColsumRows = 256;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const unsigned short* __restrict__ x, long ld, int M, int N,
                                                             float* __restrict__ partial) {
  // block = 256 threads: 32 column-groups of 8 columns x 8 row-lanes; grid.x = column blocks of 256, grid.y = row blocks
  __shared__ float red[8][256];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cg * 8;
  const int r0 = blockIdx.y * kColsumRows;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    for (int r = r0 + rl; r < r0 + kColsumRows && r < M; r += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + static_cast<long>(r) * ld + col);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { s[2 * i] += __uint_as_float(w[i] << 16); s[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cg * 8 + i] = s[i];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][c];
    partial[static_cast<long>(blockIdx.y) * N + blockIdx.x * 256 + c] = t;
  }
}

// ------------------------------------------------------------------------------------ embedding + positional encoding
struct EmbedArgs { const long* tokens; const float* table; const float* pe; float scale; float* y32; unsigned short* y16;
                   int n_rows; int T; int D; int vocab; Drop drop; };
__global__ __launch_bounds__(kRowThreads) void embed_pe_fwd_kernel(EmbedArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n_rows) return;
  long tok = a.tokens[row];
  tok = tok < 0 ? 0 : (tok >= a.vocab ? a.vocab - 1 : tok);     // clamp instead of faulting on a bad id
  const float4* e = reinterpret_cast<const float4*>(a.table + tok * a.D);
  const float4* p = reinterpret_cast<const float4*>(a.pe + static_cast<long>(row % a.T) * a.D);
  for (int q = lane; q < (a.D >> 2); q += 64) {
    const float4 ev = e[q], pv = p[q];
    float4 y = make_float4(ev.x * a.scale + pv.x, ev.y * a.scale + pv.y, ev.z * a.scale + pv.z, ev.w * a.scale + pv.w);
    if (a.drop.on()) {
      const uint64_t e0 = static_cast<uint64_t>(row) * a.D + 4 * q;
      float k4[4]; a.drop.scale4(e0, k4); y.x *= k4[0]; y.y *= k4[1]; y.z *= k4[2]; y.w *= k4[3];
    }
    if (a.y32) reinterpret_cast<float4*>(a.y32 + static_cast<long>(row) * a.D)[q] = y;
    if (a.y16) reinterpret_cast<ushort4*>(a.y16 + static_cast<long>(row) * a.D)[q] =
        make_ushort4(f2bf_(y.x), f2bf_(y.y), f2bf_(y.z), f2bf_(y.w));
  }
}
// Operands of the embedding gradient as a TN GEMM, dtable = onehot^T . dy16 (fixed summation order, MFMA rate): one wave writes
// row r of the one-hot matrix [n_rows, vocab] (bf16 1.0 at the row's token) and of dy16 = bf16(scale * keep * dy).
// fp32 operands (the fp32-operand parity path): same rows, one float per element
__global__ __launch_bounds__(kRowThreads) void embed_bwd_operands_f32_kernel(const long* __restrict__ tokens, const float* __restrict__ dy, float scale,
                                                                             float* __restrict__ onehot, long ld_oh, float* __restrict__ dys,
                                                                             int n_rows, int D, int vocab, Drop drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  long tok = tokens[row];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  float4* oh = reinterpret_cast<float4*>(onehot + static_cast<long>(row) * ld_oh);
  const int hot = static_cast<int>(tok >> 2), pos = static_cast<int>(tok & 3);
  for (int q = lane; q < (vocab >> 2); q += 64) {
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q == hot) { if (pos == 0) w.x = 1.f; else if (pos == 1) w.y = 1.f; else if (pos == 2) w.z = 1.f; else w.w = 1.f; }
    oh[q] = w;
  }
  for (int q = lane; q < (D >> 2); q += 64) {
    const float4 v = reinterpret_cast<const float4*>(dy + static_cast<long>(row) * D)[q];
    float k4[4] = {1.f, 1.f, 1.f, 1.f};
    if (drop.on()) drop.scale4(static_cast<uint64_t>(row) * D + 4 * q, k4);
    reinterpret_cast<float4*>(dys + static_cast<long>(row) * D)[q] = make_float4(scale * k4[0] * v.x, scale * k4[1] * v.y, scale * k4[2] * v.z, scale * k4[3] * v.w);
  }
}
__global__ __launch_bounds__(kRowThreads) void embed_bwd_operands_kernel(const long* __restrict__ tokens, const float* __restrict__ dy, float scale,
                                                                         unsigned short* __restrict__ onehot, long ld_oh,
                                                                         unsigned short* __restrict__ dy16, int n_rows, int D, int vocab, Drop drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  long tok = tokens[row];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  uint4* oh = reinterpret_cast<uint4*>(onehot + static_cast<long>(row) * ld_oh);
  const int hot = static_cast<int>(tok >> 3), pos = static_cast<int>(tok & 7);
  for (int q = lane; q < (vocab >> 3); q += 64) {
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (q == hot) w[pos >> 1] = 0x3f80u << (16 * (pos & 1));
    oh[q] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  for (int q = lane; q < (D >> 2); q += 64) {
    float4 v = reinterpret_cast<const float4*>(dy + static_cast<long>(row) * D)[q];
    float k4[4] = {1.f, 1.f, 1.f, 1.f};
    if (drop.on()) drop.scale4(static_cast<uint64_t>(row) * D + 4 * q, k4);
    reinterpret_cast<ushort4*>(dy16 + static_cast<long>(row) * D)[q] =
        make_ushort4(f2bf_(scale * k4[0] * v.x), f2bf_(scale * k4[1] * v.y), f2bf_(scale * k4[2] * v.z), f2bf_(scale * k4[3] * v.w));
  }
}
// dtable[token] += scale * dy[row]   (float atomics: tokens repeat inside a batch)
__global__ __launch_bounds__(kRowThreads) void embed_bwd_kernel(const long* __restrict__ tokens, const float* __restrict__ dy,
                                                                float scale, float* __restrict__ dtable, int n_rows, int D, int vocab, Drop drop) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  long tok = tokens[row];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  for (int c = lane; c < D; c += 64) {
    const float keep = drop.on() ? drop.scale(static_cast<uint64_t>(row) * drop_ld(D) + c) : 1.0f;
    atomicAdd(dtable + tok * D + c, scale * keep * dy[static_cast<long>(row) * D + c]);
  }
}

// ------------------------------------------------------------------------------------ cross-entropy (fwd + bwd in one pass)
// row loss = logsumexp(z) - z[label] for label != ignore; dlogits = (softmax - onehot) * (1 / n_valid).
// n_valid is counted on the device by count_valid_kernel so no host round trip is needed.
__global__ __launch_bounds__(256) void count_valid_kernel(const long* __restrict__ labels, int n, long ignore, float* __restrict__ out) {
  __shared__ int red[4];
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 256) c += labels[i] != ignore;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = static_cast<float>(red[0] + red[1] + red[2] + red[3]);
}

struct CeArgs { const float* logits; long ld; const long* labels; long ignore; const float* n_valid;
                float* row_loss; void* dlogits; long ldd; int M; int V; };
__device__ __forceinline__ void store1_(unsigned short* p, float v) { *p = f2bf_(v); }
__device__ __forceinline__ void store1_(float* p, float v) { *p = v; }
__device__ __forceinline__ float nan_to_num_(float z) {          // model.py:233
  if (z != z) return 0.0f;
  if (z > 3.0e38f) return 1e4f;
  if (z < -3.0e38f) return -1e4f;
  return z;
}
// TG = unsigned short: bf16 gradients, fast exp/log (default path); TG = float: fp32 gradients, libm expf/logf (fp32-operand path)
template <typename TG>
__global__ __launch_bounds__(kRowThreads) void cross_entropy_kernel(CeArgs a) {
  constexpr bool kPrecise = sizeof(TG) == 4;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const float* z = a.logits + static_cast<long>(row) * a.ld;
  const long label = a.labels[row];
  float mx = -3.0e38f;
  for (int c = lane; c < a.V; c += 64) mx = fmaxf(mx, nan_to_num_(z[c]));
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < a.V; c += 64) se += kPrecise ? expf(nan_to_num_(z[c]) - mx) : __expf(nan_to_num_(z[c]) - mx);
  se = wave_sum(se);
  const float lse = mx + (kPrecise ? logf(se) : __logf(se));
  const bool valid = label != a.ignore && label >= 0 && label < a.V;
  if (lane == 0) a.row_loss[row] = valid ? lse - nan_to_num_(z[label]) : 0.0f;
  if (a.dlogits) {
    const float inv = valid ? 1.0f / fmaxf(a.n_valid[0], 1.0f) : 0.0f;
    TG* d = static_cast<TG*>(a.dlogits) + static_cast<long>(row) * a.ldd;
    for (int c = lane; c < a.V; c += 64) {
      const float raw = z[c];
      const bool finite = raw == raw && fabsf(raw) <= 3.0e38f;   // nan_to_num passes no gradient through non-finite inputs
      const float e = kPrecise ? expf(nan_to_num_(raw) - lse) : __expf(nan_to_num_(raw) - lse);
      const float gr = (e - (c == label ? 1.0f : 0.0f)) * inv;
      store1_(d + c, finite ? gr : 0.0f);
    }
  }
}
// loss = sum(row_loss) / n_valid, summed in a fixed order by one block
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ row_loss, int M, const float* __restrict__ n_valid,
                                                          float* __restrict__ loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) s += row_loss[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / n_valid[0];   // 0/0 = NaN like torch when nothing is kept
}

// ------------------------------------------------------------------------------------ fp32 -> bf16 (and transposed copy)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n) {
  const long i = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(x + i + 4);
    uint4 o;
    o.x = f2bf_(a.x) | (static_cast<unsigned>(f2bf_(a.y)) << 16); o.y = f2bf_(a.z) | (static_cast<unsigned>(f2bf_(a.w)) << 16);
    o.z = f2bf_(b.x) | (static_cast<unsigned>(f2bf_(b.y)) << 16); o.w = f2bf_(b.z) | (static_cast<unsigned>(f2bf_(b.w)) << 16);
    *reinterpret_cast<uint4*>(y + i) = o;
  } else {
    for (long j = i; j < n; ++j) y[j] = f2bf_(x[j]);
  }
}
// y[c][r] = bf16(x[r][c]) through a 64x64 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void cast_transpose_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ yt,
                                                                  int R, int Cc) {
  __shared__ unsigned short tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < Cc) ? f2bf_(x[static_cast<long>(r0 + r) * Cc + c0 + c]) : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < Cc && r0 + r < R) yt[static_cast<long>(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

// every weight of the network in ONE launch: block (tile, item) casts a 64x64 tile of item's matrix once and writes it
// both ways (row-major copy and transposed copy); blocks past an item's tile count leave at once
struct CastItem { const float* x; unsigned short* y; unsigned short* y_t; int rows, cols; };
__global__ __launch_bounds__(256) void cast_bf16_batched_kernel(const CastItem* __restrict__ items) {
  __shared__ unsigned short tile[64][66];
  const CastItem it = items[blockIdx.y];
  const int tc = (it.cols + 63) >> 6, tr = (it.rows + 63) >> 6;
  if (static_cast<int>(blockIdx.x) >= tc * tr) return;
  const int r0 = (blockIdx.x / tc) * 64, c0 = (blockIdx.x % tc) * 64;
  // whole tiles of 16-byte-aligned matrices: 64-byte runs per thread (the element-wise loop below moved 4 / 2 bytes per lane and
  // instruction: 140 us for the model's 69 M weights, ~100 us this way)
  if (r0 + 64 <= it.rows && c0 + 64 <= it.cols && (it.cols & 7) == 0 && (it.rows & 7) == 0 && (reinterpret_cast<uintptr_t>(it.x) & 15) == 0 &&
      (!it.y || (reinterpret_cast<uintptr_t>(it.y) & 15) == 0) && (!it.y_t || (reinterpret_cast<uintptr_t>(it.y_t) & 15) == 0)) {
    const int q = threadIdx.x & 3, rr = threadIdx.x >> 2;              // phase 1: row rr, columns 16 q .. 16 q + 15
    const float4* src = reinterpret_cast<const float4*>(it.x + static_cast<long>(r0 + rr) * it.cols + c0 + 16 * q);
    unsigned w[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = src[k];
      w[2 * k] = f2bf_(v.x) | (static_cast<unsigned>(f2bf_(v.y)) << 16);
      w[2 * k + 1] = f2bf_(v.z) | (static_cast<unsigned>(f2bf_(v.w)) << 16);
    }
    if (it.y) {
      uint4* dst = reinterpret_cast<uint4*>(it.y + static_cast<long>(r0 + rr) * it.cols + c0 + 16 * q);
      dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
      dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
    if (!it.y_t) return;
    unsigned* trow = reinterpret_cast<unsigned*>(&tile[rr][16 * q]);     // row pitch 132 B, 16 q columns = 32 q bytes: 4-byte aligned
#pragma unroll
    for (int k = 0; k < 8; ++k) trow[k] = w[k];
    __syncthreads();
    const int cc = threadIdx.x >> 2;                                     // phase 2: column cc, rows 16 q .. 16 q + 15
    unsigned o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = tile[16 * q + 2 * k][cc] | (static_cast<unsigned>(tile[16 * q + 2 * k + 1][cc]) << 16);
    uint4* dst = reinterpret_cast<uint4*>(it.y_t + static_cast<long>(c0 + cc) * it.rows + r0 + 16 * q);
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    return;
  }
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    unsigned short v = 0;
    if (r0 + r < it.rows && c0 + c < it.cols) {
      v = f2bf_(it.x[static_cast<long>(r0 + r) * it.cols + c0 + c]);
      if (it.y) it.y[static_cast<long>(r0 + r) * it.cols + c0 + c] = v;
    }
    tile[r][c] = v;
  }
  if (!it.y_t) return;
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < it.cols && r0 + r < it.rows) it.y_t[static_cast<long>(c0 + c) * it.rows + r0 + r] = tile[r][c];
  }
}

// ------------------------------------------------------------------------------------ grad-norm + AdamW on flat buffers
constexpr int kSumsqBlocks = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<long>(gridDim.x) * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// norm[0] = sqrt(sum partial); norm[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_)
__global__ __launch_bounds__(256) void gradnorm_finish_kernel(const float* __restrict__ partial, int n, float max_norm, float* __restrict__ norm) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    norm[0] = nrm;
    const float c = max_norm / (nrm + 1e-6f);
    norm[1] = max_norm > 0.f ? fminf(c, 1.0f) : 1.0f;
  }
}
struct AdamArgs { float* p; const float* g; float* m; float* v; unsigned short* p16; long n; float lr, beta1, beta2, eps, wd;
                  float bc1, bc2; const float* clip; const long* nodecay; int n_nodecay; };
__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  const long i = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i >= a.n) return;
  const float clip = a.clip ? a.clip[1] : 1.0f;
  // weight decay is skipped inside the sorted, disjoint ranges nodecay[r] = [lo, hi) (biases and LayerNorm weights:
  // HF Trainer.get_decay_parameter_names); range bounds are multiples of 4, so the lane's 4 elements share the answer
  float wd = a.wd;
  if (a.n_nodecay > 0) {
    int lo = 0, hi = a.n_nodecay;                       // first range with hi > i
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.nodecay[2 * mid + 1] > i) hi = mid; else lo = mid + 1; }
    if (lo < a.n_nodecay && a.nodecay[2 * lo] <= i) wd = 0.f;
  }
  float pv[4], gv[4], mv[4], vv[4];
  const int cnt = (a.n - i) >= 4 ? 4 : static_cast<int>(a.n - i);
  if (cnt == 4) {
    *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(a.p + i);
    *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(a.g + i);
    *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(a.m + i);
    *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(a.v + i);
  } else {
    for (int j = 0; j < cnt; ++j) { pv[j] = a.p[i + j]; gv[j] = a.g[i + j]; mv[j] = a.m[i + j]; vv[j] = a.v[i + j]; }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gr = gv[j] * clip;
    pv[j] *= 1.0f - a.lr * wd;                                     // decoupled weight decay (torch AdamW)
    mv[j] = a.beta1 * mv[j] + (1.0f - a.beta1) * gr;
    vv[j] = a.beta2 * vv[j] + (1.0f - a.beta2) * gr * gr;
    const float denom = sqrtf(vv[j]) / sqrtf(a.bc2) + a.eps;
    pv[j] -= (a.lr / a.bc1) * (mv[j] / denom);
  }
  if (cnt == 4) {
    *reinterpret_cast<float4*>(a.p + i) = *reinterpret_cast<float4*>(pv);
    *reinterpret_cast<float4*>(a.m + i) = *reinterpret_cast<float4*>(mv);
    *reinterpret_cast<float4*>(a.v + i) = *reinterpret_cast<float4*>(vv);
    if (a.p16) *reinterpret_cast<ushort4*>(a.p16 + i) = make_ushort4(f2bf_(pv[0]), f2bf_(pv[1]), f2bf_(pv[2]), f2bf_(pv[3]));
  } else {
    for (int j = 0; j < cnt; ++j) { a.p[i + j] = pv[j]; a.m[i + j] = mv[j]; a.v[i + j] = vv[j]; if (a.p16) a.p16[i + j] = f2bf_(pv[j]); }
  }
}

}  // namespace adt

using namespace adt;
#define ST(s) static_cast<hipStream_t>(s)

extern "C" int adt_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                 float* y32, void* y16, int64_t ldy, float* mean, float* rstd, int64_t M, int64_t D,
                                 const adt_dropout* out_drop, void* stream) {
  if (!x || !gamma || !beta || (!y32 && !y16)) return set_error(ADT_EINVAL, "adt_layernorm_fwd: null pointer");
  if (D <= 0 || D > kMaxD || (D & 3) || (ldx & 3) || (ldy & 3)) return set_error(ADT_ESHAPE, "adt_layernorm_fwd: D must be a multiple of 4, <= 1024");
  if (M < 0) return set_error(ADT_EINVAL, "adt_layernorm_fwd: negative M");
  if (M == 0) return ADT_OK;
  LnFwdArgs a{x, ldx, gamma, beta, eps, y32, static_cast<unsigned short*>(y16), ldy, mean, rstd, static_cast<int>(M), static_cast<int>(D),
              out_drop ? make_drop(out_drop->p, out_drop->key) : Drop{0u, 0u, 1.0f}};
  if (D <= 64)
    hipLaunchKernelGGL(layernorm_fwd_narrow_kernel<16>, dim3(static_cast<unsigned>((M + 15) / 16)), dim3(kRowThreads), 0, ST(stream), a);
  else if (D <= 128)
    hipLaunchKernelGGL(layernorm_fwd_narrow_kernel<32>, dim3(static_cast<unsigned>((M + 7) / 8)), dim3(kRowThreads), 0, ST(stream), a);
  else
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(static_cast<unsigned>((M + 3) / 4)), dim3(kRowThreads), 0, ST(stream), a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" size_t adt_layernorm_bwd_workspace_bytes(int64_t M, int64_t D) {
  if (M <= 0 || D <= 0) return 0;
  const int rpb = ln_bwd_rows(M);
  return static_cast<size_t>((M + rpb - 1) / rpb) * 3 * D * 4;
}

template <typename TB>
static int layernorm_bwd_impl(const char* who, const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                              const float* mean, const float* rstd, float* dx32, void* dxb, int64_t lddx,
                              float* dgamma, float* dbeta, float* dxsum, int64_t M, int64_t D,
                              const adt_dropout* dy_drop, const adt_dropout* dxb_drop, void* ws, size_t ws_bytes, void* stream) {
  (void)who;
  if (!dy || !x || !gamma || !mean || !rstd || (!dx32 && !dxb)) return set_error(ADT_EINVAL, "adt_layernorm_bwd: null pointer");
  if (D <= 0 || D > kMaxD || (D & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3)) return set_error(ADT_ESHAPE, "adt_layernorm_bwd: D must be a multiple of 4, <= 1024");
  if (M < 0) return set_error(ADT_EINVAL, "adt_layernorm_bwd: negative M");
  if (M == 0) return ADT_OK;
  if (!ws || ws_bytes < adt_layernorm_bwd_workspace_bytes(M, D)) return set_error(ADT_EINVAL, "adt_layernorm_bwd: workspace too small");
  const int nb = static_cast<int>((M + ln_bwd_rows(M) - 1) / ln_bwd_rows(M));
  float* const slice = reduce_queue_slice(adt_layernorm_bwd_workspace_bytes(M, D), ST(stream));
  if (slice) ws = slice;
  LnBwdArgs a{dy, lddy, x, ldx, gamma, mean, rstd, dx32, dxb, lddx, static_cast<float*>(ws),
              static_cast<int>(M), static_cast<int>(D), dy_drop ? make_drop(dy_drop->p, dy_drop->key) : Drop{0u, 0u, 1.0f},
              dxb_drop ? make_drop(dxb_drop->p, dxb_drop->key) : Drop{0u, 0u, 1.0f}};
  hipLaunchKernelGGL(layernorm_bwd_kernel<TB>, dim3(nb), dim3(kRowThreads), 0, ST(stream), a);
  const int width = 3 * static_cast<int>(D);
  ADT_HIP_TRY(hipGetLastError());
  if (slice) return reduce_queue_push(slice, nb, width, dgamma, dbeta, dxsum, static_cast<int>(D));
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((width + kRedCols - 1) / kRedCols), dim3(256), 0, ST(stream), static_cast<const float*>(ws), nb,
                     width, dgamma, dbeta, dxsum, static_cast<int>(D));
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                 const float* mean, const float* rstd, float* dx32, void* dx16, int64_t lddx,
                                 float* dgamma, float* dbeta, float* dxsum, int64_t M, int64_t D,
                                 const adt_dropout* dy_drop, const adt_dropout* dx16_drop, void* ws,
                                 size_t ws_bytes, void* stream) {
  return layernorm_bwd_impl<unsigned short>("adt_layernorm_bwd", dy, lddy, x, ldx, gamma, mean, rstd, dx32, dx16, lddx, dgamma, dbeta, dxsum, M, D,
                                            dy_drop, dx16_drop, ws, ws_bytes, stream);
}

extern "C" int adt_layernorm_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                     const float* mean, const float* rstd, float* dx32, float* dx_branch, int64_t lddx,
                                     float* dgamma, float* dbeta, float* dxsum, int64_t M, int64_t D,
                                     const adt_dropout* dy_drop, const adt_dropout* branch_drop, void* ws,
                                     size_t ws_bytes, void* stream) {
  return layernorm_bwd_impl<float>("adt_layernorm_bwd_f32", dy, lddy, x, ldx, gamma, mean, rstd, dx32, dx_branch, lddx, dgamma, dbeta, dxsum, M, D,
                                   dy_drop, branch_drop, ws, ws_bytes, stream);
}

namespace adt {
void launch_reduce_partials(const float* partial, int n_part, int width, float* out, hipStream_t st) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((width + kRedCols - 1) / kRedCols)), dim3(256), 0, st, partial, n_part,
                     width, out, static_cast<float*>(nullptr), static_cast<float*>(nullptr), width);
}

// ---- deferred second-stage reductions: this thread's open queue (adt_reduce_queue_begin .. _end)
namespace {
struct ReduceQueue {
  bool open = false;
  hipStream_t st = nullptr;
  char* arena = nullptr;
  size_t bytes = 0, used = 0;
  RedBatchArgs batch;
  int n = 0;
};
thread_local ReduceQueue rq;

int reduce_queue_launch() {              // entries only: the arena is rewound by the explicit flush
  if (rq.n == 0) return ADT_OK;
  hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3(static_cast<unsigned>(rq.batch.blk_start[rq.n])), dim3(256), 0, rq.st, rq.batch, rq.n);
  rq.n = 0;
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
}  // namespace

bool reduce_queue_open(hipStream_t st) { return rq.open && rq.st == st; }

float* reduce_queue_slice(size_t bytes, hipStream_t st) {
  if (!reduce_queue_open(st)) return nullptr;
  const size_t need = (bytes + 255) & ~static_cast<size_t>(255);
  if (need > rq.bytes - rq.used) return nullptr;
  float* p = reinterpret_cast<float*>(rq.arena + rq.used);
  rq.used += need;
  return p;
}

int reduce_queue_push(const float* partial, int n_part, int width, float* out0, float* out1, float* out2, int D) {
  if (!rq.open) return set_error(ADT_EINVAL, "reduce_queue_push: no open queue");
  if (width <= 0) return ADT_OK;
  if (rq.n == kRedBatch)
    if (int rc = reduce_queue_launch()) return rc;
  if (rq.n == 0) rq.batch.blk_start[0] = 0;
  rq.batch.e[rq.n] = RedEntry{partial, out0, out1, out2, n_part, width, D, 0};
  rq.batch.blk_start[rq.n + 1] = rq.batch.blk_start[rq.n] + (width + kRedCols - 1) / kRedCols;
  ++rq.n;
  return ADT_OK;
}
}  // namespace adt

extern "C" int adt_reduce_queue_begin(void* arena, size_t arena_bytes, void* stream) {
  if (rq.open) return set_error(ADT_EINVAL, "adt_reduce_queue_begin: a queue is already open on this thread");
  if (!arena || !aligned16(arena) || arena_bytes < 4096) return set_error(ADT_EINVAL, "adt_reduce_queue_begin: the arena must be a 16-byte aligned device buffer of at least 4 KiB");
  rq.open = true; rq.st = ST(stream); rq.arena = static_cast<char*>(arena); rq.bytes = arena_bytes; rq.used = 0; rq.n = 0;
  return ADT_OK;
}

extern "C" int adt_reduce_queue_flush(void) {
  if (!rq.open) return set_error(ADT_EINVAL, "adt_reduce_queue_flush: no open queue");
  const int rc = reduce_queue_launch();
  rq.used = 0;                  // stream order: whoever writes the arena next runs after the launch that read it
  return rc;
}

extern "C" int adt_reduce_queue_end(int discard) {
  if (!rq.open) return ADT_OK;
  int rc = ADT_OK;
  if (discard) rq.n = 0; else rc = reduce_queue_launch();
  rq = ReduceQueue{};
  return rc;
}

extern "C" size_t adt_colsum_workspace_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  return static_cast<size_t>((M + kColsumRows - 1) / kColsumRows) * N * 4;
}

extern "C" int adt_colsum_bf16(const void* x, int64_t ld, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !out) return set_error(ADT_EINVAL, "adt_colsum_bf16: null pointer");
  if (M < 0 || N <= 0 || (N & 7) || (ld & 7) || !aligned16(x)) return set_error(ADT_ESHAPE, "adt_colsum_bf16: N and ld must be multiples of 8");
  if (M == 0) {
    if (reduce_queue_open(ST(stream))) return reduce_queue_push(nullptr, 0, static_cast<int>(N), out, nullptr, nullptr, static_cast<int>(N));
    ADT_HIP_TRY(hipMemsetAsync(out, 0, N * 4, ST(stream)));
    return ADT_OK;
  }
  float* const slice = reduce_queue_slice(adt_colsum_workspace_bytes(M, N), ST(stream));
  if (slice) ws = slice;
  else if (!ws || ws_bytes < adt_colsum_workspace_bytes(M, N)) return set_error(ADT_EINVAL, "adt_colsum_bf16: workspace too small");
  const int nb = static_cast<int>((M + kColsumRows - 1) / kColsumRows);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(static_cast<unsigned>((N + 255) / 256), nb), dim3(256), 0, ST(stream),
                     static_cast<const unsigned short*>(x), ld, static_cast<int>(M), static_cast<int>(N), static_cast<float*>(ws));
  if (slice) {
    ADT_HIP_TRY(hipGetLastError());
    return reduce_queue_push(slice, nb, static_cast<int>(N), out, nullptr, nullptr, static_cast<int>(N));
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((N + kRedCols - 1) / kRedCols)), dim3(256), 0, ST(stream),
                     static_cast<const float*>(ws), nb, static_cast<int>(N), out, static_cast<float*>(nullptr), static_cast<float*>(nullptr),
                     static_cast<int>(N));
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_embed_pe_fwd(const int64_t* tokens, const float* table, const float* pe, float scale, float* y32, void* y16,
                                int64_t n_rows, int64_t T, int64_t D, int64_t vocab, const adt_dropout* drop, void* stream) {
  if (!tokens || !table || !pe || (!y32 && !y16)) return set_error(ADT_EINVAL, "adt_embed_pe_fwd: null pointer");
  if (D <= 0 || (D & 3) || T <= 0 || vocab <= 0 || n_rows < 0) return set_error(ADT_ESHAPE, "adt_embed_pe_fwd: D must be a multiple of 4; T, vocab > 0");
  if (n_rows == 0) return ADT_OK;
  EmbedArgs a{reinterpret_cast<const long*>(tokens), table, pe, scale, y32, static_cast<unsigned short*>(y16), static_cast<int>(n_rows),
              static_cast<int>(T), static_cast<int>(D), static_cast<int>(vocab), drop ? make_drop(drop->p, drop->key) : Drop{0u, 0u, 1.0f}};
  hipLaunchKernelGGL(embed_pe_fwd_kernel, dim3(static_cast<unsigned>((n_rows + 3) / 4)), dim3(kRowThreads), 0, ST(stream), a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_embed_bwd(const int64_t* tokens, const float* dy, float scale, float* dtable, int64_t n_rows, int64_t D,
                             int64_t vocab, const adt_dropout* drop, void* stream) {
  if (!tokens || !dy || !dtable) return set_error(ADT_EINVAL, "adt_embed_bwd: null pointer");
  if (D <= 0 || vocab <= 0 || n_rows < 0) return set_error(ADT_ESHAPE, "adt_embed_bwd: bad shape");
  if (n_rows == 0) return ADT_OK;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(static_cast<unsigned>((n_rows + 3) / 4)), dim3(kRowThreads), 0, ST(stream),
                     reinterpret_cast<const long*>(tokens), dy, scale, dtable, static_cast<int>(n_rows), static_cast<int>(D), static_cast<int>(vocab),
                     drop ? make_drop(drop->p, drop->key) : Drop{0u, 0u, 1.0f});
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_embed_bwd_operands(const int64_t* tokens, const float* dy, float scale, void* onehot, int64_t ld_onehot, void* dy16,
                                      int64_t n_rows, int64_t D, int64_t vocab, const adt_dropout* drop, void* stream) {
  if (!tokens || !dy || !onehot || !dy16) return set_error(ADT_EINVAL, "adt_embed_bwd_operands: null pointer");
  if (D <= 0 || (D & 3) || vocab <= 0 || (vocab & 7) || ld_onehot < vocab || (ld_onehot & 7) || n_rows < 0 || !aligned16(onehot) || !aligned16(dy) ||
      !aligned16(dy16))
    return set_error(ADT_ESHAPE, "adt_embed_bwd_operands: D % 4, vocab % 8, ld_onehot % 8 must be 0, buffers 16-byte aligned");
  if (n_rows == 0) return ADT_OK;
  hipLaunchKernelGGL(embed_bwd_operands_kernel, dim3(static_cast<unsigned>((n_rows + 3) / 4)), dim3(kRowThreads), 0, ST(stream),
                     reinterpret_cast<const long*>(tokens), dy, scale, static_cast<unsigned short*>(onehot), ld_onehot,
                     static_cast<unsigned short*>(dy16), static_cast<int>(n_rows), static_cast<int>(D), static_cast<int>(vocab),
                     drop ? make_drop(drop->p, drop->key) : Drop{0u, 0u, 1.0f});
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_embed_bwd_operands_f32(const int64_t* tokens, const float* dy, float scale, float* onehot, int64_t ld_onehot, float* dy_scaled,
                                          int64_t n_rows, int64_t D, int64_t vocab, const adt_dropout* drop, void* stream) {
  if (!tokens || !dy || !onehot || !dy_scaled) return set_error(ADT_EINVAL, "adt_embed_bwd_operands_f32: null pointer");
  if (D <= 0 || (D & 3) || vocab <= 0 || (vocab & 3) || ld_onehot < vocab || (ld_onehot & 3) || n_rows < 0 || !aligned16(onehot) || !aligned16(dy) ||
      !aligned16(dy_scaled))
    return set_error(ADT_ESHAPE, "adt_embed_bwd_operands_f32: D % 4, vocab % 4, ld_onehot % 4 must be 0, buffers 16-byte aligned");
  if (n_rows == 0) return ADT_OK;
  hipLaunchKernelGGL(embed_bwd_operands_f32_kernel, dim3(static_cast<unsigned>((n_rows + 3) / 4)), dim3(kRowThreads), 0, ST(stream),
                     reinterpret_cast<const long*>(tokens), dy, scale, onehot, ld_onehot, dy_scaled, static_cast<int>(n_rows), static_cast<int>(D),
                     static_cast<int>(vocab), drop ? make_drop(drop->p, drop->key) : Drop{0u, 0u, 1.0f});
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" size_t adt_cross_entropy_workspace_bytes(int64_t M) { return M > 0 ? static_cast<size_t>(M + 4) * 4 : 16; }

template <typename TG>
static int cross_entropy_impl(const float* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, int64_t M, int64_t V,
                              float* loss, void* dlogits, int64_t ldd, void* ws, size_t ws_bytes, void* stream) {
  if (!logits || !labels || !loss) return set_error(ADT_EINVAL, "adt_cross_entropy: null pointer");
  if (M < 0 || V <= 0 || ld < V || (dlogits && ldd < V)) return set_error(ADT_EINVAL, "adt_cross_entropy: bad shape");
  if (!ws || ws_bytes < adt_cross_entropy_workspace_bytes(M)) return set_error(ADT_EINVAL, "adt_cross_entropy: workspace too small");
  float* n_valid = static_cast<float*>(ws);
  float* row_loss = n_valid + 4;
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(256), 0, ST(stream), reinterpret_cast<const long*>(labels), static_cast<int>(M),
                     static_cast<long>(ignore_index), n_valid);
  if (M > 0) {
    CeArgs a{logits, ld, reinterpret_cast<const long*>(labels), static_cast<long>(ignore_index), n_valid, row_loss,
             dlogits, ldd, static_cast<int>(M), static_cast<int>(V)};
    hipLaunchKernelGGL(cross_entropy_kernel<TG>, dim3(static_cast<unsigned>((M + 3) / 4)), dim3(kRowThreads), 0, ST(stream), a);
  }
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, ST(stream), row_loss, static_cast<int>(M), n_valid, loss);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_cross_entropy(const float* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, int64_t M, int64_t V,
                                 float* loss, void* dlogits, int64_t ldd, void* ws, size_t ws_bytes, void* stream) {
  return cross_entropy_impl<unsigned short>(logits, ld, labels, ignore_index, M, V, loss, dlogits, ldd, ws, ws_bytes, stream);
}

extern "C" int adt_cross_entropy_f32(const float* logits, int64_t ld, const int64_t* labels, int64_t ignore_index, int64_t M, int64_t V,
                                     float* loss, float* dlogits, int64_t ldd, void* ws, size_t ws_bytes, void* stream) {
  return cross_entropy_impl<float>(logits, ld, labels, ignore_index, M, V, loss, dlogits, ldd, ws, ws_bytes, stream);
}

extern "C" int adt_cast_bf16(const float* x, void* y, void* y_t, int64_t rows, int64_t cols, void* stream) {
  if (!x || (!y && !y_t)) return set_error(ADT_EINVAL, "adt_cast_bf16: null pointer");
  if (rows < 0 || cols < 0) return set_error(ADT_EINVAL, "adt_cast_bf16: negative size");
  const long n = rows * cols;
  if (n == 0) return ADT_OK;
  if (y) hipLaunchKernelGGL(cast_bf16_kernel, dim3(static_cast<unsigned>((n + 2047) / 2048)), dim3(256), 0, ST(stream), x,
                            static_cast<unsigned short*>(y), n);
  if (y_t) hipLaunchKernelGGL(cast_transpose_bf16_kernel, dim3(static_cast<unsigned>((cols + 63) / 64), static_cast<unsigned>((rows + 63) / 64)),
                              dim3(256), 0, ST(stream), x, static_cast<unsigned short*>(y_t), static_cast<int>(rows), static_cast<int>(cols));
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

static_assert(sizeof(CastItem) == sizeof(adt_cast_item), "CastItem mirrors adt_cast_item");
extern "C" int adt_cast_bf16_batched(const adt_cast_item* items_dev, int32_t n_items, int32_t max_tiles, void* stream) {
  if (!items_dev) return set_error(ADT_EINVAL, "adt_cast_bf16_batched: null pointer");
  if (n_items < 0 || max_tiles < 0 || n_items > 65535) return set_error(ADT_EINVAL, "adt_cast_bf16_batched: bad counts");
  if (n_items == 0 || max_tiles == 0) return ADT_OK;
  hipLaunchKernelGGL(cast_bf16_batched_kernel, dim3(static_cast<unsigned>(max_tiles), static_cast<unsigned>(n_items)), dim3(256), 0, ST(stream),
                     reinterpret_cast<const CastItem*>(items_dev));
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

// ------------------------------------------------------------------------------------ greedy decode: the step's tail
// argmax over the vocabulary (first index on ties, NaN counts as the maximum: torch.argmax), the finished / end-token logic of
// the reference's sampler (model.py:300-322) and the step's counters, all on device state so that the step replays as a HIP graph.
struct GreedyArgs { const float* logits; long ld; int B, V; unsigned char* finished; long* gen; long ld_gen; long* t; long* tok; int* klen;
                    long* done_at; long eos, Tmax; };
namespace adt {
__device__ __forceinline__ bool argmax_better(float v, int i, float bv, int bi) {
  const bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;
  if (!vn && v != bv) return v > bv;
  return i < bi;
}
__global__ __launch_bounds__(256) void greedy_step_kernel(GreedyArgs a) {
  __shared__ int not_finished;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) not_finished = 0;
  __syncthreads();
  const long t1 = *a.t + 1;                                   // the column this step writes
  for (int b = wave; b < a.B; b += 4) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < a.V; c += 64) {
      const float v = a.logits[static_cast<long>(b) * a.ld + c];
      if (argmax_better(v, c, bv, bi)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (argmax_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
      const bool was = a.finished[b] != 0;
      const long nxt = was ? a.eos : static_cast<long>(bi);
      a.gen[static_cast<long>(b) * a.ld_gen + t1] = nxt;
      const bool fin = was || nxt == a.eos;
      a.finished[b] = fin ? 1 : 0;
      a.tok[b] = nxt;
      a.klen[b] += 1;
      if (!fin) atomicOr(&not_finished, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!not_finished && *a.done_at == a.Tmax) *a.done_at = t1 + 1;     // the number of columns the reference would return
    *a.t = t1;
  }
}
}  // namespace adt

extern "C" int adt_greedy_step(const float* logits, int64_t ld, int64_t B, int64_t V, uint8_t* finished, int64_t* gen, int64_t ld_gen,
                               int64_t* t, int64_t* tok, int32_t* klen, int64_t* done_at, int64_t end_token, int64_t max_length, void* stream) {
  if (!logits || !finished || !gen || !t || !tok || !klen || !done_at) return set_error(ADT_EINVAL, "adt_greedy_step: null pointer");
  if (B < 0 || V <= 0 || V > 0x7fffffff || ld < V || ld_gen < max_length || max_length < 2) return set_error(ADT_ESHAPE, "adt_greedy_step: bad sizes");
  if (B == 0) return ADT_OK;
  GreedyArgs a{logits, static_cast<long>(ld), static_cast<int>(B), static_cast<int>(V), finished, reinterpret_cast<long*>(gen), static_cast<long>(ld_gen),
               reinterpret_cast<long*>(t), reinterpret_cast<long*>(tok), klen, reinterpret_cast<long*>(done_at), static_cast<long>(end_token),
               static_cast<long>(max_length)};
  hipLaunchKernelGGL(greedy_step_kernel, dim3(1), dim3(256), 0, ST(stream), a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" size_t adt_grad_norm_workspace_bytes(void) { return kSumsqBlocks * 4; }

extern "C" int adt_grad_norm(const float* g, int64_t n, float max_norm, float* norm_and_clip, void* ws, size_t ws_bytes, void* stream) {
  if (!g || !norm_and_clip) return set_error(ADT_EINVAL, "adt_grad_norm: null pointer");
  if (n < 0) return set_error(ADT_EINVAL, "adt_grad_norm: negative n");
  if (!ws || ws_bytes < adt_grad_norm_workspace_bytes()) return set_error(ADT_EINVAL, "adt_grad_norm: workspace too small");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(kSumsqBlocks), dim3(256), 0, ST(stream), g, static_cast<long>(n), static_cast<float*>(ws));
  hipLaunchKernelGGL(gradnorm_finish_kernel, dim3(1), dim3(256), 0, ST(stream), static_cast<const float*>(ws), kSumsqBlocks, max_norm, norm_and_clip);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t step, const float* norm_and_clip,
                              const int64_t* nodecay_ranges, int32_t n_nodecay, void* stream) {
  if (!p || !g || !m || !v) return set_error(ADT_EINVAL, "adt_adamw_step: null pointer");
  if (n < 0 || step < 1 || n_nodecay < 0 || (n_nodecay > 0 && !nodecay_ranges)) return set_error(ADT_EINVAL, "adt_adamw_step: n < 0, step < 1 or bad no-decay ranges");
  if (n == 0) return ADT_OK;
  AdamArgs a{p, g, m, v, static_cast<unsigned short*>(p_bf16), static_cast<long>(n), lr, beta1, beta2, eps, weight_decay,
             static_cast<float>(1.0 - pow(static_cast<double>(beta1), static_cast<double>(step))),
             static_cast<float>(1.0 - pow(static_cast<double>(beta2), static_cast<double>(step))), norm_and_clip,
             reinterpret_cast<const long*>(nodecay_ranges), n_nodecay};
  hipLaunchKernelGGL(adamw_kernel, dim3(static_cast<unsigned>((n + 1023) / 1024)), dim3(256), 0, ST(stream), a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
