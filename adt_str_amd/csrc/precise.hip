// precise.hip -- the fp32-operand parity path of the ADT network (gfx950).
//
// The training / serving default computes GEMMs and attention with bf16 operands (what the reference does under its
// bf16 autocast, train.py:233-234).  BASELINE's parity target, "logits within 1e-3 rel-tol of the CPU reference", is a
// statement about the reference's fp32 CPU path (model.py:240-258 run without autocast), so the engine has a second mode
// in which every activation stays fp32 and every contraction runs on the f32-input matrix instructions
// (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate, bit-for-bit an fmaf chain in k order).  That mode is what
// the parity tests run; it is never the bench default (the f32 MFMA rate is 1/16 of the bf16 rate).
//
//   adt_gemm_f32          C = op(A) op(B) with the epilogue of adt_gemm_bf16 (bias, GELU', saved pre-activation, GELU /
//                         ReLU, dropout, residual / PE rows), all operands fp32
//   adt_attn_fwd_f32      flash-style attention, head_dim 128, additive causal / key-padding masks, dropout on P
//   adt_attn_bwd_f32      dQ kernel (+ delta) and dK/dV kernel, probabilities recomputed from lse, no atomics
//   adt_colsum_f32        column sums (bias gradients)
//
// Tiling is deliberately plain (LDS images with a 129-float row stride = conflict-free ds_read_b32 both by row and by
// column, single-buffered with a register prefetch): these kernels have to be right and deterministic, not fast.
#include <hip/hip_runtime.h>

#include "adt_common.h"
#include "dropout.h"

namespace adt {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// row of accumulator register `reg` for lane half h (C/D layout of the 32x32 MFMAs): the column is lane & 31
__device__ __forceinline__ int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// =============================================================================================== GEMM
constexpr int kBM = 128, kBN = 128, kBK = 16, kLdT = kBM + 4;       // LDS tiles are k-major: Xs[k][row], 528-byte rows
constexpr int kXKf = 32;                                             // K granule of a split (= the split-bf16 kernel's K-tile)

struct GemmF32Args {
  const float *A, *B; float* C; long lda, ldb, ldc; int M, N, K;
  const float* bias; const float* gelu_grad_of; long ld_gg; float* pre_act_out; long ld_pa;
  const float* residual; long ld_res; int res_row_mod; int act; float alpha; Drop drop; int drop_after_residual; int act_grad_mode;
  int k_tiles_per_split; float* slabs;     // split-K (blockIdx.z = split): plain fp32 partial tiles into slabs[split][M][N], summed in slab order afterwards
};

// One 128 x 16 operand tile as two float4 per thread.  kKMajor = false: X is [rows][K] (k contiguous);
// true: X is [K][rows] (row index contiguous).
template <bool kKMajor>
__device__ __forceinline__ void tile_fetch(const float* __restrict__ X, long ld, int rows, int K, int r0, int k0, int tid, float4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r, k;
    if (!kKMajor) { r = r0 + (tid >> 2) + 64 * i; k = k0 + (tid & 3) * 4; }
    else { k = k0 + (tid >> 5) + 8 * i; r = r0 + (tid & 31) * 4; }
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows && k < K) v[i] = *reinterpret_cast<const float4*>(kKMajor ? X + static_cast<long>(k) * ld + r : X + static_cast<long>(r) * ld + k);
  }
}
template <bool kKMajor>
__device__ __forceinline__ void tile_store(float* Xs, int tid, const float4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!kKMajor) {
      const int r = (tid >> 2) + 64 * i, k = (tid & 3) * 4;
      Xs[(k + 0) * kLdT + r] = v[i].x; Xs[(k + 1) * kLdT + r] = v[i].y; Xs[(k + 2) * kLdT + r] = v[i].z; Xs[(k + 3) * kLdT + r] = v[i].w;
    } else {
      const int k = (tid >> 5) + 8 * i, r = (tid & 31) * 4;
      *reinterpret_cast<float4*>(Xs + k * kLdT + r) = v[i];
    }
  }
}

__device__ __forceinline__ float gelu_erf(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float u) {
  return 0.5f * (1.0f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * expf(-0.5f * u * u);
}

// The epilogue both GEMM kernels share: acc[mi][ni] is the 32 x 32 block (rows m0 + wm * 64 + mi * 32 + rowmap(e, h), column n0 + wn * 64 + ni * 32 + r).
__device__ __forceinline__ void gemm_f32_epilogue(const GemmF32Args& g, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn, int r, int h) {
  if (g.slabs) {                             // a K split: the raw partial tile
    float* const slab = g.slabs + static_cast<long>(blockIdx.z) * g.M * g.N;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 64 + ni * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 64 + mi * 32 + rowmap(e, h);
          if (row < g.M && col < g.N) slab[static_cast<long>(row) * g.N + col] = acc[mi][ni][e];
        }
      }
    return;
  }
  const uint64_t dld = drop_ld(static_cast<uint64_t>(g.N));
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + r;
      if (col >= g.N) continue;
      const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + mi * 32 + rowmap(e, h);
        if (row >= g.M) continue;
        float z = g.alpha * acc[mi][ni][e] + bias;
        if (g.gelu_grad_of) {
          const float u = g.gelu_grad_of[static_cast<long>(row) * g.ld_gg + col];
          z *= g.act_grad_mode ? u : gelu_erf_grad(u);                 // act_grad_mode: the forward stored gelu'(z) * keep
        }
        const float keep = g.drop.on() ? g.drop.scale(static_cast<uint64_t>(row) * dld + col) : 1.0f;
        if (g.pre_act_out) g.pre_act_out[static_cast<long>(row) * g.ld_pa + col] = (g.act_grad_mode && g.act == 1) ? gelu_erf_grad(z) * keep : z;
        if (g.act == 1) z = gelu_erf(z);
        else if (g.act == 2) z = fmaxf(z, 0.f);
        if (!g.drop_after_residual) z *= keep;
        if (g.residual) {
          const long rr = g.res_row_mod > 0 ? row % g.res_row_mod : row;
          z += g.residual[rr * g.ld_res + col];
        }
        if (g.drop_after_residual) z *= keep;
        g.C[static_cast<long>(row) * g.ldc + col] = z;
      }
    }
}

template <bool kAK, bool kBK_>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args g) {
  __shared__ __attribute__((aligned(16))) float As[kBK * kLdT];
  __shared__ __attribute__((aligned(16))) float Bs[kBK * kLdT];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 av[2], bv[2];
  const int nkt_all = (g.K + kBK - 1) / kBK;
  const int kt0 = g.slabs ? static_cast<int>(blockIdx.z) * g.k_tiles_per_split * (kXKf / kBK) : 0;      // (splits are counted in 32-deep steps for both kernels)
  const int nkt = g.slabs ? min(nkt_all, kt0 + g.k_tiles_per_split * (kXKf / kBK)) : nkt_all;
  tile_fetch<kAK>(g.A, g.lda, g.M, g.K, m0, kt0 * kBK, tid, av);
  tile_fetch<kBK_>(g.B, g.ldb, g.N, g.K, n0, kt0 * kBK, tid, bv);
  for (int kt = kt0; kt < nkt; ++kt) {
    tile_store<kAK>(As, tid, av);
    tile_store<kBK_>(Bs, tid, bv);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_fetch<kAK>(g.A, g.lda, g.M, g.K, m0, (kt + 1) * kBK, tid, av);
      tile_fetch<kBK_>(g.B, g.ldb, g.N, g.K, n0, (kt + 1) * kBK, tid, bv);
    }
#pragma unroll
    for (int kk = 0; kk < kBK / 2; ++kk) {
      const float* ar = As + (2 * kk + h) * kLdT + wm * 64 + r;
      const float* br = Bs + (2 * kk + h) * kLdT + wn * 64 + r;
      const float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
      acc[0][0] = mfma32(a0, b0, acc[0][0]); acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]); acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  gemm_f32_epilogue(g, acc, m0, n0, wm, wn, r, h);
}

// ----------------------------------------------------------------------------------------------- split-bf16 products ("bf16x3")
// The same contraction on the bf16 matrix pipe (16 x the f32-input rate): every fp32 operand x is split once, while its tile is
// staged, into two bf16 planes hi = bf16(x), lo = bf16(x - hi) -- x = hi + lo to 2^-17 relative -- and a product a b is taken as
// a_lo b_hi + a_hi b_lo + a_hi b_hi: three v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator (the dropped a_lo b_lo term is 2^-18 of
// the product).  Error per product ~1e-5 relative against 6e-8 for the exact f32 MFMA and 4e-3 for plain bf16 operands: the parity arm
// that meets BASELINE's "logits within 1e-3 rel-tol of the CPU reference" at several times the f32-MFMA arm's speed.  Everything around
// the products -- operand layouts, the epilogue (gemm_f32_epilogue), fp32 activations, the accumulation order along k in steps of 16 --
// is the f32 path's; selected per call (layout bit 4 of adt_gemm_f32, adt_attn_desc.f32_products).
// Not representable: an operand whose bf16 rounding overflows (|x| > 3.39e38) gives hi = inf, lo = nan.
typedef __attribute__((ext_vector_type(8))) short bf16x8s;
typedef __bf16 bf16x2v_ __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2_(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{a, b}, bf16x2v_)); }
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_bf2_(a, b);
  lo = pack_bf2_(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) { split2(v.x, v.y, hi.x, lo.x); split2(v.z, v.w, hi.y, lo.y); }
__device__ __forceinline__ f32x16 mfma_x3(const bf16x8s& ah, const bf16x8s& al, const bf16x8s& bh, const bf16x8s& bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);        // the small terms first
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

constexpr int kXK = 32;                     // K-tile of the split kernel
constexpr int kXPitch = 80;                 // bytes per LDS row of a plane: 32 bf16 + 16 (16 consecutive rows -> 16 distinct 16-byte bank slots)
constexpr int kXPlane = 128 * kXPitch;      // one plane of one operand tile: 10,240 B; A_hi | A_lo | B_hi | B_lo = 40 KiB

// One 128 x 32 operand tile as four float4 per thread.  kKMajor = false: X is [rows][K] -- float4 i = row (tid >> 3) + 32 i, k (tid & 7) * 4 .. + 3;
// true: X is [K][rows] -- float4 i = k 4 (tid >> 5) + i, rows 4 (tid & 31) .. + 3 (a 4 x 4 block, transposed in registers when it is stored).
template <bool kKMajor>
__device__ __forceinline__ void xtile_fetch(const float* __restrict__ X, long ld, int rows, int K, int r0, int k0, int tid, float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r, k;
    if (!kKMajor) { r = r0 + (tid >> 3) + 32 * i; k = k0 + (tid & 7) * 4; }
    else { k = k0 + 4 * (tid >> 5) + i; r = r0 + (tid & 31) * 4; }
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows && k < K) v[i] = *reinterpret_cast<const float4*>(kKMajor ? X + static_cast<long>(k) * ld + r : X + static_cast<long>(r) * ld + k);
  }
}
template <bool kKMajor>
__device__ __forceinline__ void xtile_store(unsigned char* hi_plane, unsigned char* lo_plane, int tid, const float4 (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 w;
    int row, k;
    if (!kKMajor) { w = v[i]; row = (tid >> 3) + 32 * i; k = (tid & 7) * 4; }
    else {                                   // row 4 (tid & 31) + i of the block: its four consecutive k
      const float e0 = i == 0 ? v[0].x : i == 1 ? v[0].y : i == 2 ? v[0].z : v[0].w, e1 = i == 0 ? v[1].x : i == 1 ? v[1].y : i == 2 ? v[1].z : v[1].w;
      const float e2 = i == 0 ? v[2].x : i == 1 ? v[2].y : i == 2 ? v[2].z : v[2].w, e3 = i == 0 ? v[3].x : i == 1 ? v[3].y : i == 2 ? v[3].z : v[3].w;
      w = make_float4(e0, e1, e2, e3); row = (tid & 31) * 4 + i; k = 4 * (tid >> 5);
    }
    uint2 hi, lo;
    split4(w, hi, lo);
    *reinterpret_cast<uint2*>(hi_plane + row * kXPitch + k * 2) = hi;
    *reinterpret_cast<uint2*>(lo_plane + row * kXPitch + k * 2) = lo;
  }
}

template <bool kAK, bool kBK_>
__global__ __launch_bounds__(256) void gemm_f32x3_kernel(GemmF32Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char planes[4 * kXPlane];
  unsigned char* const Ah = planes; unsigned char* const Al = planes + kXPlane;
  unsigned char* const Bh = planes + 2 * kXPlane; unsigned char* const Bl = planes + 3 * kXPlane;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 av[4], bv[4];
  const int nkt_all = (g.K + kXK - 1) / kXK;
  const int kt0 = g.slabs ? static_cast<int>(blockIdx.z) * g.k_tiles_per_split : 0;
  const int nkt = g.slabs ? min(nkt_all, kt0 + g.k_tiles_per_split) : nkt_all;
  xtile_fetch<kAK>(g.A, g.lda, g.M, g.K, m0, kt0 * kXK, tid, av);
  xtile_fetch<kBK_>(g.B, g.ldb, g.N, g.K, n0, kt0 * kXK, tid, bv);
  const int fa = (wm * 64 + r) * kXPitch + 16 * h, fb = (wn * 64 + r) * kXPitch + 16 * h;      // + 32 rows * pitch per block, + 32 bytes per k-step
  for (int kt = kt0; kt < nkt; ++kt) {
    xtile_store<kAK>(Ah, Al, tid, av);
    xtile_store<kBK_>(Bh, Bl, tid, bv);
    __syncthreads();
    if (kt + 1 < nkt) {
      xtile_fetch<kAK>(g.A, g.lda, g.M, g.K, m0, (kt + 1) * kXK, tid, av);
      xtile_fetch<kBK_>(g.B, g.ldb, g.N, g.K, n0, (kt + 1) * kXK, tid, bv);
    }
#pragma unroll
    for (int ks = 0; ks < kXK / 16; ++ks) {
      bf16x8s ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8s*>(Ah + fa + i * 32 * kXPitch + ks * 32);
        al[i] = *reinterpret_cast<const bf16x8s*>(Al + fa + i * 32 * kXPitch + ks * 32);
        bh[i] = *reinterpret_cast<const bf16x8s*>(Bh + fb + i * 32 * kXPitch + ks * 32);
        bl[i] = *reinterpret_cast<const bf16x8s*>(Bl + fb + i * 32 * kXPitch + ks * 32);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_x3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
    }
    __syncthreads();
  }
  gemm_f32_epilogue(g, acc, m0, n0, wm, wn, r, h);
}

// =============================================================================================== attention
// Templated on the head dimension DH in {16, 32, 64, 128} (the shipped configs use 128; the reference-captured golden
// model of tests/golden/adt_tiny.npz has 16).  LDS block images are 32 rows of 32 * ceil(DH / 32) floats (zero padded)
// with a +1 row stride.
constexpr int kBlk = 32;
template <int DH> struct Geo {
  static constexpr int kNdt = (DH + 31) / 32;          // 32-wide d tiles of the transposed accumulators
  static constexpr int kCols = 32 * kNdt;              // image columns (zero padded past DH)
  static constexpr int kLd = kCols + 1;                // odd row stride: conflict-free by row and by column
  static constexpr int kTile = kBlk * kLd;
  static constexpr int kNs = DH / 2;                   // k steps of a DH-deep 32x32x2 contraction
};

struct AttnF32Args {
  const float *q, *k, *v, *o, *dout; float *out, *dq, *dk, *dv; float* lse; float* delta;
  long ldq, ldk, ldv, ldo; int B, H, Sq, Sk; float scale, mask_value; int causal; const int* key_len; Drop drop;
};

// rows [row0, row0 + 32) of a [n_rows][ld] matrix (columns [0, DH)) into an LDS image; rows past n_rows and columns past DH are zeros
template <int DH, int kThreads>
__device__ __forceinline__ void block_to_lds(const float* __restrict__ base, long ld, int row0, int n_rows, float* img, int tid) {
  constexpr int kQ = Geo<DH>::kCols / 4;               // float4 slots per row
  for (int idx = tid; idx < kBlk * kQ; idx += kThreads) {
    const int j = idx / kQ, c4 = idx % kQ;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + j < n_rows && 4 * c4 < DH) v = *reinterpret_cast<const float4*>(base + static_cast<long>(row0 + j) * ld + 4 * c4);
    float* d = img + j * Geo<DH>::kLd + 4 * c4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}
// the lane's B-operand view of its own row: x[s] = row[2 s + h], s < DH / 2 (zeros when the row does not exist)
template <int DH>
__device__ __forceinline__ void row_operand(const float* __restrict__ row, bool live, int h, float (&x)[DH / 2]) {
#pragma unroll
  for (int u = 0; u < DH / 4; ++u) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) v = reinterpret_cast<const float4*>(row)[u];
    x[2 * u] = h ? v.y : v.x;
    x[2 * u + 1] = h ? v.w : v.z;
  }
}
__device__ __forceinline__ float score_mask(const AttnF32Args& a, int qi, int ki, int klen) {
  float m = 0.f;
  if (a.causal && ki > qi) m += a.mask_value;
  if (ki >= klen) m += a.mask_value;
  return m;
}
// store the transposed accumulator X^T[d][row] (d on registers, `row` = the lane's row) as X[row][0..DH), times `mul`
template <int DH>
__device__ __forceinline__ void store_rows(const f32x16 (&x)[Geo<DH>::kNdt], float mul, float* __restrict__ dst, int h) {
#pragma unroll
  for (int dt = 0; dt < Geo<DH>::kNdt; ++dt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int d0 = 32 * dt + 8 * gq + 4 * h;
      if (d0 < DH)
        *reinterpret_cast<float4*>(dst + d0) = make_float4(x[dt][4 * gq] * mul, x[dt][4 * gq + 1] * mul, x[dt][4 * gq + 2] * mul, x[dt][4 * gq + 3] * mul);
    }
}

// ----------------------------------------------------------------------------------------------- forward
// 4 waves x 32 queries per workgroup; S^T = K Q^T puts the query on the lane and the block's keys in the 16 accumulator
// registers, so the softmax is lane-local (plus one exchange between the lane halves) and P^T is directly the B operand of
// O^T += V^T P^T.
template <int DH>
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(AttnF32Args a) {
  using G = Geo<DH>;
  __shared__ float Ks[G::kTile], Vs[G::kTile];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, wave = tid >> 6;
  const int b = blockIdx.y / a.H, head = blockIdx.y % a.H;
  const int qi = blockIdx.x * 128 + wave * 32 + r;
  const bool live = qi < a.Sq;
  const float* kb = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * DH;
  const float* vb = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * DH;
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  float qreg[G::kNs];
  row_operand<DH>(a.q + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * DH, live, h, qreg);
  f32x16 o[G::kNdt];
#pragma unroll
  for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  const uint64_t drow = ((static_cast<uint64_t>(b) * a.H + head) * a.Sq + qi) * drop_ld(static_cast<uint64_t>(a.Sk));
  for (int j0 = 0; j0 < a.Sk; j0 += kBlk) {
    __syncthreads();
    block_to_lds<DH, 256>(kb, a.ldk, j0, a.Sk, Ks, tid);
    block_to_lds<DH, 256>(vb, a.ldv, j0, a.Sk, Vs, tid);
    __syncthreads();
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int s = 0; s < G::kNs; ++s) st = mfma32(Ks[r * G::kLd + 2 * s + h], qreg[s], st);
    float mx = m;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ki = j0 + rowmap(e, h);
      float x = st[e] * a.scale + score_mask(a, qi, ki, klen);
      if (ki >= a.Sk) x = -INFINITY;
      st[e] = x;
      mx = fmaxf(mx, x);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float alpha = expf(m - mx);
    m = mx;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = expf(st[e] - mx);
      psum += p;
      st[e] = a.drop.on() ? p * a.drop.scale(drow + static_cast<uint64_t>(j0 + rowmap(e, h))) : p;
    }
    l = l * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < G::kNdt; ++dt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
#pragma unroll
      for (int s = 0; s < 16; ++s) o[dt] = mfma32(Vs[rowmap(s, h) * G::kLd + 32 * dt + r], st[s], o[dt]);
    }
  }
  l += __shfl_xor(l, 32);
  if (!live) return;
  store_rows<DH>(o, 1.0f / l, a.out + (static_cast<long>(b) * a.Sq + qi) * a.ldo + head * DH, h);
  if (h == 0) a.lse[(static_cast<long>(b) * a.H + head) * a.Sq + qi] = m + logf(l);
}

// ----------------------------------------------------------------------------------------------- backward: dQ (+ delta)
// Same orientation as the forward: S^T = K Q^T and dP^T = V dO^T with the query on the lane; dS^T is then the B operand
// of dQ^T += K^T dS^T.  delta = rowsum(O o dO) is written for the dK/dV kernel.
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_dq_f32_kernel(AttnF32Args a) {
  using G = Geo<DH>;
  __shared__ float Ks[G::kTile], Vs[G::kTile];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, wave = tid >> 6;
  const int b = blockIdx.y / a.H, head = blockIdx.y % a.H;
  const int qi = blockIdx.x * 128 + wave * 32 + r;
  const bool live = qi < a.Sq;
  const float* kb = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * DH;
  const float* vb = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * DH;
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  const long stat = (static_cast<long>(b) * a.H + head) * a.Sq + qi;
  float qreg[G::kNs], doreg[G::kNs];
  row_operand<DH>(a.q + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * DH, live, h, qreg);
  row_operand<DH>(a.dout + (static_cast<long>(b) * a.Sq + qi) * a.ldo + head * DH, live, h, doreg);
  float delta = 0.f;
  {
    const float* orow = a.o + (static_cast<long>(b) * a.Sq + qi) * a.ldo + head * DH;
#pragma unroll
    for (int u = 0; u < DH / 4; ++u) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) v = reinterpret_cast<const float4*>(orow)[u];
      delta += (h ? v.y : v.x) * doreg[2 * u] + (h ? v.w : v.z) * doreg[2 * u + 1];
    }
    delta += __shfl_xor(delta, 32);
  }
  const float lse = live ? a.lse[stat] : INFINITY;
  if (live && h == 0) a.delta[stat] = delta;
  f32x16 dq[G::kNdt];
#pragma unroll
  for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
  const uint64_t drow = static_cast<uint64_t>(stat) * drop_ld(static_cast<uint64_t>(a.Sk));
  for (int j0 = 0; j0 < a.Sk; j0 += kBlk) {
    __syncthreads();
    block_to_lds<DH, 256>(kb, a.ldk, j0, a.Sk, Ks, tid);
    block_to_lds<DH, 256>(vb, a.ldv, j0, a.Sk, Vs, tid);
    __syncthreads();
    f32x16 st, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
    for (int s = 0; s < G::kNs; ++s) st = mfma32(Ks[r * G::kLd + 2 * s + h], qreg[s], st);
#pragma unroll
    for (int s = 0; s < G::kNs; ++s) dp = mfma32(Vs[r * G::kLd + 2 * s + h], doreg[s], dp);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ki = j0 + rowmap(e, h);
      const float x = st[e] * a.scale + score_mask(a, qi, ki, klen);
      const float p = ki < a.Sk ? expf(x - lse) : 0.f;
      const float keep = a.drop.on() ? a.drop.scale(drow + static_cast<uint64_t>(ki)) : 1.0f;
      st[e] = p * (dp[e] * keep - delta) * a.scale;
    }
#pragma unroll
    for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
      for (int s = 0; s < 16; ++s) dq[dt] = mfma32(Ks[rowmap(s, h) * G::kLd + 32 * dt + r], st[s], dq[dt]);
  }
  if (live) store_rows<DH>(dq, 1.0f, a.dq + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * DH, h);
}

// ----------------------------------------------------------------------------------------------- backward: dK, dV
// 2 waves per workgroup, each owning 32 keys (K, V blocks wave-private in LDS), sweeping the query blocks together
// (Q, dO blocks shared).  S = Q K^T and dP = dO V^T put the key on the lane and the block's queries in the registers, so
// P_drop and dS are the B operands of dV^T += dO^T P_drop and dK^T += Q^T dS.
constexpr int kDkvThreads = 128;
template <int DH> constexpr int dkv_lds_bytes() { return (6 * Geo<DH>::kTile + 64) * 4; }
template <int DH>
__global__ __launch_bounds__(kDkvThreads) void attn_bwd_dkv_f32_kernel(AttnF32Args a) {
  using G = Geo<DH>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Qs = smem; float* Ds = smem + G::kTile;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, wave = tid >> 6;
  float* Kw = smem + (2 + 2 * wave) * G::kTile; float* Vw = Kw + G::kTile;
  float* lse_s = smem + 6 * G::kTile; float* del_s = lse_s + 32;
  const int b = blockIdx.y / a.H, head = blockIdx.y % a.H;
  const int j0 = (blockIdx.x * 2 + wave) * kBlk, ki = j0 + r;
  const float* qb = a.q + static_cast<long>(b) * a.Sq * a.ldq + head * DH;
  const float* dob = a.dout + static_cast<long>(b) * a.Sq * a.ldo + head * DH;
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  const long stat0 = (static_cast<long>(b) * a.H + head) * a.Sq;
  // the wave's own K and V blocks
  block_to_lds<DH, 64>(a.k + static_cast<long>(b) * a.Sk * a.ldk + head * DH, a.ldk, j0, a.Sk, Kw, lane);
  block_to_lds<DH, 64>(a.v + static_cast<long>(b) * a.Sk * a.ldv + head * DH, a.ldv, j0, a.Sk, Vw, lane);
  f32x16 dk[G::kNdt], dv[G::kNdt];
#pragma unroll
  for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[dt][e] = dv[dt][e] = 0.f;
  const uint64_t dld = drop_ld(static_cast<uint64_t>(a.Sk));
  for (int i0 = 0; i0 < a.Sq; i0 += kBlk) {
    __syncthreads();
    block_to_lds<DH, kDkvThreads>(qb, a.ldq, i0, a.Sq, Qs, tid);
    block_to_lds<DH, kDkvThreads>(dob, a.ldo, i0, a.Sq, Ds, tid);
    if (tid < 32) {
      const bool ok = i0 + tid < a.Sq;
      lse_s[tid] = ok ? a.lse[stat0 + i0 + tid] : INFINITY;
      del_s[tid] = ok ? a.delta[stat0 + i0 + tid] : 0.f;
    }
    __syncthreads();
    f32x16 st, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
    for (int s = 0; s < G::kNs; ++s) st = mfma32(Qs[r * G::kLd + 2 * s + h], Kw[r * G::kLd + 2 * s + h], st);
#pragma unroll
    for (int s = 0; s < G::kNs; ++s) dp = mfma32(Ds[r * G::kLd + 2 * s + h], Vw[r * G::kLd + 2 * s + h], dp);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ir = rowmap(e, h), qi = i0 + ir;
      const float x = st[e] * a.scale + score_mask(a, qi, ki, klen);
      const float p = ki < a.Sk ? expf(x - lse_s[ir]) : 0.f;
      const float keep = a.drop.on() ? a.drop.scale((static_cast<uint64_t>(stat0) + qi) * dld + static_cast<uint64_t>(ki)) : 1.0f;
      st[e] = p * keep;                                   // dropped P (what multiplied V in the forward)
      dp[e] = p * (dp[e] * keep - del_s[ir]) * a.scale;   // dS
    }
#pragma unroll
    for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        dv[dt] = mfma32(Ds[rowmap(s, h) * G::kLd + 32 * dt + r], st[s], dv[dt]);
        dk[dt] = mfma32(Qs[rowmap(s, h) * G::kLd + 32 * dt + r], dp[s], dk[dt]);
      }
  }
  if (ki < a.Sk) {
    store_rows<DH>(dk, 1.0f, a.dk + (static_cast<long>(b) * a.Sk + ki) * a.ldk + head * DH, h);
    store_rows<DH>(dv, 1.0f, a.dv + (static_cast<long>(b) * a.Sk + ki) * a.ldv + head * DH, h);
  }
}

// =============================================================================================== attention, split-bf16 products
// The three kernels above with every product taken as three bf16 MFMAs (see "split-bf16 products" at the GEMM): same grids, same
// orientation (softmax axis on the lane), same masks / dropout indices / statistics, fp32 everywhere outside the products.  Operand
// blocks are split into bf16 hi / lo planes while they are staged:
//   row planes   [32 rows][DH] bf16 (pitch DH * 2 + 16 bytes)  -> A / B fragments "row r, 8 consecutive d" by one ds_read_b128
//   transposed   [32 ceil(DH / 32) rows = d][32 positions] bf16 (pitch 80 bytes), position 16 ks + 8 h + j <-> block row rowmap(8 ks + j, h)
//                -> A fragments "d r, the 8 block rows a lane half holds in accumulator registers 8 ks .. 8 ks + 7": the accumulator of
//                one product (P^T, dS^T, P, dS), split into hi / lo, is the B operand of the next without leaving the registers.
template <int DH> struct GeoX {
  static constexpr int kPR = DH * 2 + 16;              // row-plane pitch (bytes): an odd number of 16-byte bank slots
  static constexpr int kRows = 32 * kPR;               // one row plane
  static constexpr int kPT = 80;                       // transposed-plane pitch
  static constexpr int kTrans = Geo<DH>::kCols * kPT;  // one transposed plane (rows past DH stay zero)
  static constexpr int kNK = DH / 16;                  // k-steps of a DH-deep contraction
};
__device__ __forceinline__ int tpos(int j) {           // position of block row j in a transposed plane
  const int e = (j & 3) + 4 * (j >> 3), hh = (j >> 2) & 1;
  return 16 * (e >> 3) + 8 * hh + (e & 7);
}
union Frag8 { bf16x8s v; unsigned u[4]; };
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8s& hi, bf16x8s& lo) {
  Frag8 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) split2(x[2 * i], x[2 * i + 1], a.u[i], b.u[i]);
  hi = a.v; lo = b.v;
}
// rows [row0, row0 + 32) of a [n_rows][ld] fp32 matrix (columns [0, DH)) into row planes and / or transposed planes (hi at the pointer,
// lo one plane behind it); rows past n_rows are zeros
template <int DH, int kThreads, bool kRowP, bool kTransP>
__device__ __forceinline__ void block_to_planes(const float* __restrict__ base, long ld, int row0, int n_rows, unsigned char* rp, unsigned char* tp, int tid) {
  constexpr int kQ = DH / 4;
  for (int idx = tid; idx < kBlk * kQ; idx += kThreads) {
    const int j = idx / kQ, c4 = idx % kQ;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + j < n_rows) v = *reinterpret_cast<const float4*>(base + static_cast<long>(row0 + j) * ld + 4 * c4);
    uint2 hi, lo;
    split4(v, hi, lo);
    if (kRowP) {
      *reinterpret_cast<uint2*>(rp + j * GeoX<DH>::kPR + 8 * c4) = hi;
      *reinterpret_cast<uint2*>(rp + GeoX<DH>::kRows + j * GeoX<DH>::kPR + 8 * c4) = lo;
    }
    if (kTransP) {
      unsigned short* th = reinterpret_cast<unsigned short*>(tp + (4 * c4) * GeoX<DH>::kPT + 2 * tpos(j));
      unsigned short* tl = reinterpret_cast<unsigned short*>(tp + GeoX<DH>::kTrans + (4 * c4) * GeoX<DH>::kPT + 2 * tpos(j));
      constexpr int kStep = GeoX<DH>::kPT / 2;
      th[0] = static_cast<unsigned short>(hi.x); th[kStep] = static_cast<unsigned short>(hi.x >> 16);
      th[2 * kStep] = static_cast<unsigned short>(hi.y); th[3 * kStep] = static_cast<unsigned short>(hi.y >> 16);
      tl[0] = static_cast<unsigned short>(lo.x); tl[kStep] = static_cast<unsigned short>(lo.x >> 16);
      tl[2 * kStep] = static_cast<unsigned short>(lo.y); tl[3 * kStep] = static_cast<unsigned short>(lo.y >> 16);
    }
  }
}
template <int DH, int kThreads>
__device__ __forceinline__ void zero_trans_tail(unsigned char* tp, int tid) {      // d rows [DH, 32 ceil(DH / 32)) of both transposed planes
  if constexpr (Geo<DH>::kCols > DH) {
    constexpr int kWords = (Geo<DH>::kCols - DH) * GeoX<DH>::kPT / 4;
    for (int i = tid; i < kWords; i += kThreads) {
      reinterpret_cast<unsigned*>(tp + DH * GeoX<DH>::kPT)[i] = 0u;
      reinterpret_cast<unsigned*>(tp + GeoX<DH>::kTrans + DH * GeoX<DH>::kPT)[i] = 0u;
    }
  }
}
// the lane's own row as B fragments: x[ks] = row[16 ks + 8 h .. + 7], hi and lo (zeros when the row does not exist)
template <int DH>
__device__ __forceinline__ void row_frags(const float* __restrict__ row, bool live, int h, bf16x8s (&hi)[DH / 16], bf16x8s (&lo)[DH / 16]) {
#pragma unroll
  for (int ks = 0; ks < DH / 16; ++ks) {
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
      const float4 v0 = *reinterpret_cast<const float4*>(row + 16 * ks + 8 * h), v1 = *reinterpret_cast<const float4*>(row + 16 * ks + 8 * h + 4);
      x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
    }
    split8(x, hi[ks], lo[ks]);
  }
}
__device__ __forceinline__ bf16x8s lds_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8s*>(p); }
// accumulator registers 8 ks .. 8 ks + 7 as a B fragment pair
__device__ __forceinline__ void acc_frags(const f32x16& t, int ks, bf16x8s& hi, bf16x8s& lo) {
  const float x[8] = {t[8 * ks], t[8 * ks + 1], t[8 * ks + 2], t[8 * ks + 3], t[8 * ks + 4], t[8 * ks + 5], t[8 * ks + 6], t[8 * ks + 7]};
  split8(x, hi, lo);
}

template <int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd_x3_kernel(AttnF32Args a) {
  using G = Geo<DH>; using X = GeoX<DH>;
  __shared__ __attribute__((aligned(16))) unsigned char Kp[2 * X::kRows];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[2 * X::kTrans];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, wave = tid >> 6;
  const int b = blockIdx.y / a.H, head = blockIdx.y % a.H;
  const int qi = blockIdx.x * 128 + wave * 32 + r;
  const bool live = qi < a.Sq;
  const float* kb = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * DH;
  const float* vb = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * DH;
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  bf16x8s qh[X::kNK], ql[X::kNK];
  row_frags<DH>(a.q + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * DH, live, h, qh, ql);
  zero_trans_tail<DH, 256>(Vt, tid);
  f32x16 o[G::kNdt];
#pragma unroll
  for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m = -INFINITY, l = 0.f;
  const uint64_t drow = ((static_cast<uint64_t>(b) * a.H + head) * a.Sq + qi) * drop_ld(static_cast<uint64_t>(a.Sk));
  for (int j0 = 0; j0 < a.Sk; j0 += kBlk) {
    __syncthreads();
    block_to_planes<DH, 256, true, false>(kb, a.ldk, j0, a.Sk, Kp, nullptr, tid);
    block_to_planes<DH, 256, false, true>(vb, a.ldv, j0, a.Sk, nullptr, Vt, tid);
    __syncthreads();
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < X::kNK; ++ks) {
      const unsigned char* p = Kp + r * X::kPR + (16 * ks + 8 * h) * 2;
      st = mfma_x3(lds_frag(p), lds_frag(p + X::kRows), qh[ks], ql[ks], st);
    }
    float mx = m;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ki = j0 + rowmap(e, h);
      float x = st[e] * a.scale + score_mask(a, qi, ki, klen);
      if (ki >= a.Sk) x = -INFINITY;
      st[e] = x;
      mx = fmaxf(mx, x);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float alpha = expf(m - mx);
    m = mx;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = expf(st[e] - mx);
      psum += p;
      st[e] = a.drop.on() ? p * a.drop.scale(drow + static_cast<uint64_t>(j0 + rowmap(e, h))) : p;
    }
    l = l * alpha + psum;
    bf16x8s ph[2], pl[2];
    acc_frags(st, 0, ph[0], pl[0]);
    acc_frags(st, 1, ph[1], pl[1]);
#pragma unroll
    for (int dt = 0; dt < G::kNdt; ++dt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned char* p = Vt + (32 * dt + r) * X::kPT + (16 * ks + 8 * h) * 2;
        o[dt] = mfma_x3(lds_frag(p), lds_frag(p + X::kTrans), ph[ks], pl[ks], o[dt]);
      }
    }
  }
  l += __shfl_xor(l, 32);
  if (!live) return;
  store_rows<DH>(o, 1.0f / l, a.out + (static_cast<long>(b) * a.Sq + qi) * a.ldo + head * DH, h);
  if (h == 0) a.lse[(static_cast<long>(b) * a.H + head) * a.Sq + qi] = m + logf(l);
}

template <int DH> constexpr int dq_x3_lds_bytes() { return 4 * GeoX<DH>::kRows + 2 * GeoX<DH>::kTrans; }
template <int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq_x3_kernel(AttnF32Args a) {
  using G = Geo<DH>; using X = GeoX<DH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char dq_smem[];
  unsigned char* const Kp = dq_smem; unsigned char* const Vp = Kp + 2 * X::kRows; unsigned char* const Kt = Vp + 2 * X::kRows;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, wave = tid >> 6;
  const int b = blockIdx.y / a.H, head = blockIdx.y % a.H;
  const int qi = blockIdx.x * 128 + wave * 32 + r;
  const bool live = qi < a.Sq;
  const float* kb = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * DH;
  const float* vb = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * DH;
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  const long stat = (static_cast<long>(b) * a.H + head) * a.Sq + qi;
  bf16x8s qh[X::kNK], ql[X::kNK], dh_[X::kNK], dl_[X::kNK];
  const float* qrow = a.q + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * DH;
  const float* dorow = a.dout + (static_cast<long>(b) * a.Sq + qi) * a.ldo + head * DH;
  row_frags<DH>(qrow, live, h, qh, ql);
  row_frags<DH>(dorow, live, h, dh_, dl_);
  float delta = 0.f;                                     // rowsum(O o dO) in fp32 (not a matrix product): this lane half's d = 16 u + 8 h .. + 7
  if (live) {
    const float* orow = a.o + (static_cast<long>(b) * a.Sq + qi) * a.ldo + head * DH;
#pragma unroll
    for (int u = 0; u < DH / 16; ++u) {
      const int d0 = 16 * u + 8 * h;
      const float4 v0 = *reinterpret_cast<const float4*>(orow + d0), v1 = *reinterpret_cast<const float4*>(orow + d0 + 4);
      const float4 w0 = *reinterpret_cast<const float4*>(dorow + d0), w1 = *reinterpret_cast<const float4*>(dorow + d0 + 4);
      delta += v0.x * w0.x + v0.y * w0.y + v0.z * w0.z + v0.w * w0.w + v1.x * w1.x + v1.y * w1.y + v1.z * w1.z + v1.w * w1.w;
    }
  }
  delta += __shfl_xor(delta, 32);
  const float lse = live ? a.lse[stat] : INFINITY;
  if (live && h == 0) a.delta[stat] = delta;
  zero_trans_tail<DH, 256>(Kt, tid);
  f32x16 dq[G::kNdt];
#pragma unroll
  for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
  const uint64_t drow = static_cast<uint64_t>(stat) * drop_ld(static_cast<uint64_t>(a.Sk));
  for (int j0 = 0; j0 < a.Sk; j0 += kBlk) {
    __syncthreads();
    block_to_planes<DH, 256, true, true>(kb, a.ldk, j0, a.Sk, Kp, Kt, tid);
    block_to_planes<DH, 256, true, false>(vb, a.ldv, j0, a.Sk, Vp, nullptr, tid);
    __syncthreads();
    f32x16 st, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < X::kNK; ++ks) {
      const int off = r * X::kPR + (16 * ks + 8 * h) * 2;
      st = mfma_x3(lds_frag(Kp + off), lds_frag(Kp + X::kRows + off), qh[ks], ql[ks], st);
      dp = mfma_x3(lds_frag(Vp + off), lds_frag(Vp + X::kRows + off), dh_[ks], dl_[ks], dp);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ki = j0 + rowmap(e, h);
      const float x = st[e] * a.scale + score_mask(a, qi, ki, klen);
      const float p = ki < a.Sk ? expf(x - lse) : 0.f;
      const float keep = a.drop.on() ? a.drop.scale(drow + static_cast<uint64_t>(ki)) : 1.0f;
      st[e] = p * (dp[e] * keep - delta) * a.scale;
    }
    bf16x8s sh[2], sl[2];
    acc_frags(st, 0, sh[0], sl[0]);
    acc_frags(st, 1, sh[1], sl[1]);
#pragma unroll
    for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned char* p = Kt + (32 * dt + r) * X::kPT + (16 * ks + 8 * h) * 2;
        dq[dt] = mfma_x3(lds_frag(p), lds_frag(p + X::kTrans), sh[ks], sl[ks], dq[dt]);
      }
  }
  if (live) store_rows<DH>(dq, 1.0f, a.dq + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * DH, h);
}

// dK / dV: 4 waves per workgroup, each owning 32 keys whose K and V rows are its B fragments and stay in registers (the lane's own
// key row, split once), sweeping the query blocks together (Q, dO row planes and transposed planes shared through LDS: 76 KiB).  The
// f32 kernel's shape -- two waves, K / V images in LDS, one workgroup per CU by LDS size -- left the CU with two waves.
constexpr int kDkvX3Threads = 256;
template <int DH> constexpr int dkv_x3_lds_bytes() { return 4 * GeoX<DH>::kRows + 4 * GeoX<DH>::kTrans + 256; }   // Q, dO row planes; Q, dO transposed planes; lse, delta
template <int DH>
__global__ __launch_bounds__(kDkvX3Threads) void attn_bwd_dkv_x3_kernel(AttnF32Args a) {
  using G = Geo<DH>; using X = GeoX<DH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char kv_smem[];
  unsigned char* const Qp = kv_smem; unsigned char* const Dp = Qp + 2 * X::kRows;
  unsigned char* const Qt = Dp + 2 * X::kRows; unsigned char* const Dt = Qt + 2 * X::kTrans;
  float* const lse_s = reinterpret_cast<float*>(Dt + 2 * X::kTrans); float* const del_s = lse_s + 32;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, wave = tid >> 6;
  const int b = blockIdx.y / a.H, head = blockIdx.y % a.H;
  const int j0 = (blockIdx.x * 4 + wave) * kBlk, ki = j0 + r;
  const float* qb = a.q + static_cast<long>(b) * a.Sq * a.ldq + head * DH;
  const float* dob = a.dout + static_cast<long>(b) * a.Sq * a.ldo + head * DH;
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  const long stat0 = (static_cast<long>(b) * a.H + head) * a.Sq;
  bf16x8s kh[X::kNK], kl[X::kNK], vh[X::kNK], vl[X::kNK];
  row_frags<DH>(a.k + (static_cast<long>(b) * a.Sk + ki) * a.ldk + head * DH, ki < a.Sk, h, kh, kl);
  row_frags<DH>(a.v + (static_cast<long>(b) * a.Sk + ki) * a.ldv + head * DH, ki < a.Sk, h, vh, vl);
  zero_trans_tail<DH, kDkvX3Threads>(Qt, tid);
  zero_trans_tail<DH, kDkvX3Threads>(Dt, tid);
  f32x16 dk[G::kNdt], dv[G::kNdt];
#pragma unroll
  for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[dt][e] = dv[dt][e] = 0.f;
  const uint64_t dld = drop_ld(static_cast<uint64_t>(a.Sk));
  for (int i0 = 0; i0 < a.Sq; i0 += kBlk) {
    __syncthreads();
    block_to_planes<DH, kDkvX3Threads, true, true>(qb, a.ldq, i0, a.Sq, Qp, Qt, tid);
    block_to_planes<DH, kDkvX3Threads, true, true>(dob, a.ldo, i0, a.Sq, Dp, Dt, tid);
    if (tid < 32) {
      const bool ok = i0 + tid < a.Sq;
      lse_s[tid] = ok ? a.lse[stat0 + i0 + tid] : INFINITY;
      del_s[tid] = ok ? a.delta[stat0 + i0 + tid] : 0.f;
    }
    __syncthreads();
    f32x16 st, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = dp[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < X::kNK; ++ks) {
      const int off = r * X::kPR + (16 * ks + 8 * h) * 2;
      st = mfma_x3(lds_frag(Qp + off), lds_frag(Qp + X::kRows + off), kh[ks], kl[ks], st);
      dp = mfma_x3(lds_frag(Dp + off), lds_frag(Dp + X::kRows + off), vh[ks], vl[ks], dp);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ir = rowmap(e, h), qi = i0 + ir;
      const float x = st[e] * a.scale + score_mask(a, qi, ki, klen);
      const float p = ki < a.Sk ? expf(x - lse_s[ir]) : 0.f;
      const float keep = a.drop.on() ? a.drop.scale((static_cast<uint64_t>(stat0) + qi) * dld + static_cast<uint64_t>(ki)) : 1.0f;
      st[e] = p * keep;                                   // dropped P (what multiplied V in the forward)
      dp[e] = p * (dp[e] * keep - del_s[ir]) * a.scale;   // dS
    }
    bf16x8s ph[2], pl[2], sh[2], sl[2];
    acc_frags(st, 0, ph[0], pl[0]); acc_frags(st, 1, ph[1], pl[1]);
    acc_frags(dp, 0, sh[0], sl[0]); acc_frags(dp, 1, sh[1], sl[1]);
#pragma unroll
    for (int dt = 0; dt < G::kNdt; ++dt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = (32 * dt + r) * X::kPT + (16 * ks + 8 * h) * 2;
        dv[dt] = mfma_x3(lds_frag(Dt + off), lds_frag(Dt + X::kTrans + off), ph[ks], pl[ks], dv[dt]);
        dk[dt] = mfma_x3(lds_frag(Qt + off), lds_frag(Qt + X::kTrans + off), sh[ks], sl[ks], dk[dt]);
      }
  }
  if (ki < a.Sk) {
    store_rows<DH>(dk, 1.0f, a.dk + (static_cast<long>(b) * a.Sk + ki) * a.ldk + head * DH, h);
    store_rows<DH>(dv, 1.0f, a.dv + (static_cast<long>(b) * a.Sk + ki) * a.ldv + head * DH, h);
  }
}

// =============================================================================================== column sums of an fp32 matrix
constexpr int kCsRows = 256;
__global__ __launch_bounds__(256) void colsum_f32_partial_kernel(const float* __restrict__ x, long ld, int M, int N, float* __restrict__ partial) {
  // 64 column lanes x 4 row lanes; grid.x = column blocks of 64, grid.y = row blocks of kCsRows; fixed order
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl, r0 = blockIdx.y * kCsRows;
  float s = 0.f;
  if (col < N)
    for (int row = r0 + rl; row < r0 + kCsRows && row < M; row += 4) s += x[static_cast<long>(row) * ld + col];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && col < N) partial[static_cast<long>(blockIdx.y) * N + col] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

static int check_attn_f32(const adt_attn_desc* d) {
  if (!d) return set_error(ADT_EINVAL, "attention (f32): null descriptor");
  if (d->head_dim != 16 && d->head_dim != 32 && d->head_dim != 64 && d->head_dim != 128)
    return set_error(ADT_ESHAPE, "attention (f32): head_dim must be 16, 32, 64 or 128");
  if (d->batch < 0 || d->heads <= 0 || d->q_len < 0 || d->k_len < 0) return set_error(ADT_EINVAL, "attention (f32): bad sizes");
  const int64_t need = static_cast<int64_t>(d->heads) * d->head_dim;
  if (d->ldq < need || d->ldk < need || d->ldv < need || d->ldo < need || (d->ldq & 3) || (d->ldk & 3) || (d->ldv & 3) || (d->ldo & 3))
    return set_error(ADT_ESHAPE, "attention (f32): row strides must cover heads*128 columns and be multiples of 4");
  if (static_cast<int64_t>(d->batch) * d->heads > 65535) return set_error(ADT_ESHAPE, "attention (f32): batch*heads must be <= 65535");
  if (d->f32_products != 0 && d->f32_products != 1) return set_error(ADT_EINVAL, "attention (f32): f32_products must be 0 (exact f32 products) or 1 (split-bf16)");
  return ADT_OK;
}
static AttnF32Args make_f32_args(const adt_attn_desc* d) {
  AttnF32Args a{};
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
  a.B = d->batch; a.H = d->heads; a.Sq = d->q_len; a.Sk = d->k_len;
  a.scale = d->scale; a.mask_value = d->mask_value; a.causal = d->causal; a.key_len = d->key_len;
  a.drop = make_drop(d->drop.p, d->drop.key);
  return a;
}
template <int DH>
static int launch_attn_fwd_x3(const adt_attn_desc* d, const AttnF32Args& a, hipStream_t st) {
  hipLaunchKernelGGL(attn_fwd_x3_kernel<DH>, dim3((d->q_len + 127) / 128, d->batch * d->heads), dim3(256), 0, st, a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
template <int DH>
static int launch_attn_bwd_x3(const adt_attn_desc* d, const AttnF32Args& a, hipStream_t st) {
  static thread_local int lds_set_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (lds_set_for != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_x3_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, dq_x3_lds_bytes<DH>()));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_x3_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, dkv_x3_lds_bytes<DH>()));
    lds_set_for = dev;
  }
  hipLaunchKernelGGL(attn_bwd_dq_x3_kernel<DH>, dim3((d->q_len + 127) / 128, d->batch * d->heads), dim3(256), dq_x3_lds_bytes<DH>(), st, a);
  hipLaunchKernelGGL(attn_bwd_dkv_x3_kernel<DH>, dim3((d->k_len + 127) / 128, d->batch * d->heads), dim3(kDkvX3Threads), dkv_x3_lds_bytes<DH>(), st, a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

}  // namespace adt

using namespace adt;
#define ST(s) static_cast<hipStream_t>(s)

// K splits of a product whose output has too few 128 x 128 tiles to fill the chip (the weight gradients: K = batch x frames rows against a
// 768 x 3072 output = 144 tiles on 256 CUs): the plain form only (no epilogue arithmetic beyond alpha), N a multiple of 4, partial tiles
// through fp32 slabs summed in slab order (bitwise reproducible).  Returns the number of splits (1: none) and the 32-deep steps per split.
static int plan_f32_splits(int64_t M, int64_t N, int64_t K, const adt_gemm_epilogue* ep, int* per_split) {
  *per_split = 0;
  if (ep && (ep->bias || ep->gelu_grad_of || ep->pre_act_out || ep->residual || ep->act || ep->drop.p > 0.f)) return 1;
  if ((N & 3) || K < 16 * kXKf) return 1;
  int n_cu = 256;
  (void)device_cu_count(&n_cu);
  const int64_t tiles = ((M + kBM - 1) / kBM) * ((N + kBN - 1) / kBN);
  if (tiles >= 2 * n_cu) return 1;
  const int steps = static_cast<int>((K + kXKf - 1) / kXKf);
  int s = static_cast<int>((3 * n_cu + tiles - 1) / tiles);            // ~3 workgroups per CU
  const int max_s = steps / 8 > 1 ? steps / 8 : 1;                      // at least 8 steps (K = 256) per split
  s = s > max_s ? max_s : s;
  s = s > 64 ? 64 : s;
  if (s <= 1) return 1;
  const int per = (steps + s - 1) / s;
  *per_split = per;
  return (steps + per - 1) / per;                                       // no empty trailing split
}

extern "C" size_t adt_gemm_f32_workspace_bytes(int32_t layout, int64_t M, int64_t N, int64_t K) {
  (void)layout;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int per = 0;
  const int s = plan_f32_splits(M, N, K, nullptr, &per);
  return s > 1 ? static_cast<size_t>(s) * M * N * 4 : 0;
}

extern "C" int adt_gemm_f32(int32_t layout, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                            float* C, int64_t ldc, const adt_gemm_epilogue* ep, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !C) return set_error(ADT_EINVAL, "adt_gemm_f32: null pointer");
  if (M < 0 || N < 0 || K < 0 || layout < 0 || layout > 7) return set_error(ADT_EINVAL, "adt_gemm_f32: bad size or layout");
  if (M == 0 || N == 0) return ADT_OK;
  const bool ak = layout & 1, bk = layout & 2, x3 = layout & 4;
  // float4 operand loads: the contiguous extent of each operand and its leading dimension are multiples of 4 floats
  if ((lda & 3) || (ldb & 3) || !aligned16(A) || !aligned16(B) || ((ak ? M : K) & 3) || ((bk ? N : K) & 3))
    return set_error(ADT_ESHAPE, "adt_gemm_f32: contiguous extents and leading dimensions must be multiples of 4 floats, operands 16-byte aligned");
  if (M > 2147483647L || N > 2147483647L || K > 2147483647L || (N + kBN - 1) / kBN > 65535L * 32768L || (M + kBM - 1) / kBM > 65535)
    return set_error(ADT_ESHAPE, "adt_gemm_f32: problem too large");
  GemmF32Args g{};
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = static_cast<int>(M); g.N = static_cast<int>(N); g.K = static_cast<int>(K);
  g.alpha = 1.0f; g.drop = Drop{0u, 0u, 1.0f};
  if (ep) {
    if (ep->aux_bf16_out || ep->colsum_out || ep->res_ln_mean) return set_error(ADT_EINVAL, "adt_gemm_f32: aux_bf16_out / colsum_out / res_ln_* are not part of the fp32 path");
    g.bias = ep->bias; g.gelu_grad_of = static_cast<const float*>(ep->gelu_grad_of); g.ld_gg = ep->ld_gelu_grad;
    g.pre_act_out = static_cast<float*>(ep->pre_act_out); g.ld_pa = ep->ld_pre_act;
    g.residual = static_cast<const float*>(ep->residual); g.ld_res = ep->ld_res; g.res_row_mod = ep->res_row_mod;
    g.act = ep->act; g.alpha = ep->alpha; g.drop = make_drop(ep->drop.p, ep->drop.key); g.drop_after_residual = ep->drop_after_residual;
    g.act_grad_mode = ep->act_grad_mode;
  }
  int per = 0;
  int splits = plan_f32_splits(M, N, K, ep, &per);
  if (splits > 1 && (!ws || !aligned16(ws) || ws_bytes < static_cast<size_t>(splits) * M * N * 4 || (ldc & 3) || !aligned16(C))) splits = 1;   // no workspace: one pass
  g.k_tiles_per_split = splits > 1 ? per : 0;
  g.slabs = splits > 1 ? static_cast<float*>(ws) : nullptr;
  const dim3 grid(static_cast<unsigned>((N + kBN - 1) / kBN), static_cast<unsigned>((M + kBM - 1) / kBM), static_cast<unsigned>(splits));
  if (x3) {
    if (!ak && !bk) hipLaunchKernelGGL((gemm_f32x3_kernel<false, false>), grid, dim3(256), 0, ST(stream), g);
    else if (ak && bk) hipLaunchKernelGGL((gemm_f32x3_kernel<true, true>), grid, dim3(256), 0, ST(stream), g);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_f32x3_kernel<false, true>), grid, dim3(256), 0, ST(stream), g);
    else hipLaunchKernelGGL((gemm_f32x3_kernel<true, false>), grid, dim3(256), 0, ST(stream), g);
  } else {
    if (!ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, ST(stream), g);
    else if (ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, ST(stream), g);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, ST(stream), g);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, ST(stream), g);
  }
  if (splits > 1) launch_reduce_slabs(g.slabs, splits, static_cast<long>(M) * N, g.N, g.alpha, C, ldc, ST(stream));
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_attn_fwd_f32(const adt_attn_desc* d, const float* q, const float* k, const float* v, float* o, float* lse, void* stream) {
  if (int rc = check_attn_f32(d)) return rc;
  if (!q || !k || !v || !o || !lse) return set_error(ADT_EINVAL, "adt_attn_fwd_f32: null pointer");
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o)) return set_error(ADT_EINVAL, "adt_attn_fwd_f32: tensors must be 16-byte aligned");
  if (d->batch == 0 || d->q_len == 0) return ADT_OK;
  if (d->k_len == 0) return set_error(ADT_ESHAPE, "adt_attn_fwd_f32: k_len must be > 0");
  AttnF32Args a = make_f32_args(d);
  a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse;
  if (d->f32_products == 1) {                      // split-bf16 products
    switch (d->head_dim) {
      case 16: return launch_attn_fwd_x3<16>(d, a, ST(stream));
      case 32: return launch_attn_fwd_x3<32>(d, a, ST(stream));
      case 64: return launch_attn_fwd_x3<64>(d, a, ST(stream));
      default: return launch_attn_fwd_x3<128>(d, a, ST(stream));
    }
  }
  const dim3 grid((d->q_len + 127) / 128, d->batch * d->heads);
  switch (d->head_dim) {
    case 16: hipLaunchKernelGGL(attn_fwd_f32_kernel<16>, grid, dim3(256), 0, ST(stream), a); break;
    case 32: hipLaunchKernelGGL(attn_fwd_f32_kernel<32>, grid, dim3(256), 0, ST(stream), a); break;
    case 64: hipLaunchKernelGGL(attn_fwd_f32_kernel<64>, grid, dim3(256), 0, ST(stream), a); break;
    default: hipLaunchKernelGGL(attn_fwd_f32_kernel<128>, grid, dim3(256), 0, ST(stream), a); break;
  }
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

namespace adt {
template <int DH>
static int launch_attn_bwd_f32(const adt_attn_desc* d, const AttnF32Args& a, hipStream_t st) {
  static thread_local int lds_set_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (lds_set_for != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_f32_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    dkv_lds_bytes<DH>()));
    lds_set_for = dev;
  }
  hipLaunchKernelGGL(attn_bwd_dq_f32_kernel<DH>, dim3((d->q_len + 127) / 128, d->batch * d->heads), dim3(256), 0, st, a);
  hipLaunchKernelGGL(attn_bwd_dkv_f32_kernel<DH>, dim3((d->k_len + 63) / 64, d->batch * d->heads), dim3(kDkvThreads), dkv_lds_bytes<DH>(), st, a);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
}  // namespace adt

extern "C" size_t adt_attn_bwd_f32_workspace_bytes(const adt_attn_desc* d) {
  if (!d || d->batch <= 0 || d->heads <= 0 || d->q_len <= 0) return 16;
  return (static_cast<size_t>(d->batch) * d->heads * d->q_len * 4 + 15) & ~static_cast<size_t>(15);
}

extern "C" int adt_attn_bwd_f32(const adt_attn_desc* d, const float* q, const float* k, const float* v, const float* o, const float* dout,
                                const float* lse, float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_attn_f32(d)) return rc;
  if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv) return set_error(ADT_EINVAL, "adt_attn_bwd_f32: null pointer");
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o) || !aligned16(dout) || !aligned16(dq) || !aligned16(dk) || !aligned16(dv))
    return set_error(ADT_EINVAL, "adt_attn_bwd_f32: tensors must be 16-byte aligned");
  if (!ws || ws_bytes < adt_attn_bwd_f32_workspace_bytes(d)) return set_error(ADT_EINVAL, "adt_attn_bwd_f32: workspace too small");
  if (d->dq_colsum || d->dk_colsum || d->dv_colsum) return set_error(ADT_EINVAL, "adt_attn_bwd_f32: column sums are taken with adt_colsum_f32");
  if (d->batch == 0 || d->q_len == 0 || d->k_len == 0) return ADT_OK;
  AttnF32Args a = make_f32_args(d);
  a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout; a.lse = const_cast<float*>(lse); a.delta = static_cast<float*>(ws);
  a.dq = dq; a.dk = dk; a.dv = dv;
  if (d->f32_products == 1) {
    switch (d->head_dim) {
      case 16: return launch_attn_bwd_x3<16>(d, a, ST(stream));
      case 32: return launch_attn_bwd_x3<32>(d, a, ST(stream));
      case 64: return launch_attn_bwd_x3<64>(d, a, ST(stream));
      default: return launch_attn_bwd_x3<128>(d, a, ST(stream));
    }
  }
  switch (d->head_dim) {
    case 16: return launch_attn_bwd_f32<16>(d, a, ST(stream));
    case 32: return launch_attn_bwd_f32<32>(d, a, ST(stream));
    case 64: return launch_attn_bwd_f32<64>(d, a, ST(stream));
    default: return launch_attn_bwd_f32<128>(d, a, ST(stream));
  }
}

extern "C" size_t adt_colsum_f32_workspace_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  return static_cast<size_t>((M + kCsRows - 1) / kCsRows) * N * 4;
}

extern "C" int adt_colsum_f32(const float* x, int64_t ld, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !out) return set_error(ADT_EINVAL, "adt_colsum_f32: null pointer");
  if (M < 0 || N <= 0 || ld < N) return set_error(ADT_ESHAPE, "adt_colsum_f32: bad shape");
  if (M == 0) { ADT_HIP_TRY(hipMemsetAsync(out, 0, N * 4, ST(stream))); return ADT_OK; }
  if (!ws || ws_bytes < adt_colsum_f32_workspace_bytes(M, N)) return set_error(ADT_EINVAL, "adt_colsum_f32: workspace too small");
  const int nb = static_cast<int>((M + kCsRows - 1) / kCsRows);
  hipLaunchKernelGGL(colsum_f32_partial_kernel, dim3(static_cast<unsigned>((N + 63) / 64), nb), dim3(256), 0, ST(stream), x, ld, static_cast<int>(M),
                     static_cast<int>(N), static_cast<float*>(ws));
  launch_reduce_partials(static_cast<const float*>(ws), nb, static_cast<int>(N), out, ST(stream));
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

// ----------------------------------------------------------------------------------------------- split planes for adt_gemm_bf16x3
// fp32 matrix x [rows, cols] -> bf16 planes [hi | lo] side by side: out[r, c] = bf16(x[r, c]), out[r, lo_off + c] = bf16(x[r, c] - hi)
// (the split of split2 above, taken ONCE per tensor instead of once per staged tile), so that the persistent bf16 kernels of gemm.hip
// can take the three products lo x hi + hi x lo + hi x hi as one GEMM over 3 K virtual K-tiles.  transpose: out is [cols, *] with
// out[c, r] / out[c, lo_off + r] (the W^T operand of the data gradients).
namespace adt {
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long ldx, long rows, int cols8, unsigned short* __restrict__ out, long ldo, long lo_off) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= rows * cols8) return;
  const long r = i / cols8;
  const int c = static_cast<int>(i - r * cols8) * 8;
  const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + c), v1 = *reinterpret_cast<const float4*>(x + r * ldx + c + 4);
  uint2 h0, l0, h1, l1;
  split4(v0, h0, l0);
  split4(v1, h1, l1);
  *reinterpret_cast<uint4*>(out + r * ldo + c) = uint4{h0.x, h0.y, h1.x, h1.y};
  *reinterpret_cast<uint4*>(out + r * ldo + lo_off + c) = uint4{l0.x, l0.y, l1.x, l1.y};
}
constexpr int kSpT = 64, kSpPitch = 72;      // 64 x 64 tile; LDS rows of 72 bf16 (144 B: 16-byte aligned pieces)
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ x, long ldx, int rows, int cols, unsigned short* __restrict__ out, long ldo, long lo_off) {
  __shared__ __attribute__((aligned(16))) unsigned short hiT[kSpT * kSpPitch], loT[kSpT * kSpPitch];
  const int tid = threadIdx.x, r0 = blockIdx.y * kSpT, c0 = blockIdx.x * kSpT;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid >> 4) + 16 * i, c = (tid & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < rows && c0 + c < cols) v = *reinterpret_cast<const float4*>(x + static_cast<long>(r0 + r) * ldx + c0 + c);      // cols % 4 == 0
    uint2 h, l;
    split4(v, h, l);
    const unsigned hv[4] = {h.x & 0xffffu, h.x >> 16, h.y & 0xffffu, h.y >> 16}, lv[4] = {l.x & 0xffffu, l.x >> 16, l.y & 0xffffu, l.y >> 16};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hiT[(c + e) * kSpPitch + r] = static_cast<unsigned short>(hv[e]);
      loT[(c + e) * kSpPitch + r] = static_cast<unsigned short>(lv[e]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + 256 * i, c = q >> 3, r8 = (q & 7) * 8;           // out row c0 + c, source rows r0 + r8 .. + 7
    if (c0 + c < cols && r0 + r8 < rows) {                               // rows % 8 == 0
      *reinterpret_cast<uint4*>(out + static_cast<long>(c0 + c) * ldo + r0 + r8) = *reinterpret_cast<const uint4*>(hiT + c * kSpPitch + r8);
      *reinterpret_cast<uint4*>(out + static_cast<long>(c0 + c) * ldo + lo_off + r0 + r8) = *reinterpret_cast<const uint4*>(loT + c * kSpPitch + r8);
    }
  }
}
}  // namespace adt

extern "C" int adt_split_bf16x2(const float* x, int64_t ldx, int64_t rows, int64_t cols, void* planes, int64_t ldp, int64_t lo_off, int32_t transpose,
                                void* stream) {
  using namespace adt;
  if (!x || !planes) return set_error(ADT_EINVAL, "adt_split_bf16x2: null pointer");
  if (rows < 0 || cols < 0 || ldx < cols) return set_error(ADT_EINVAL, "adt_split_bf16x2: bad shape");
  const int64_t w = transpose ? rows : cols;
  if ((rows & 7) || (cols & 7) || (ldx & 3) || (ldp & 7) || (lo_off & 7) || lo_off < w || ldp < lo_off + w || (reinterpret_cast<uintptr_t>(x) & 15) ||
      (reinterpret_cast<uintptr_t>(planes) & 15))
    return set_error(ADT_ESHAPE, "adt_split_bf16x2: rows / cols multiples of 8, 16-byte aligned rows, lo plane behind the hi plane inside the row stride");
  if (rows == 0 || cols == 0) return ADT_OK;
  if (rows >= (1ll << 31) || cols >= (1ll << 31)) return set_error(ADT_ESHAPE, "adt_split_bf16x2: dimension >= 2^31");
  if (!transpose) {
    const long n = rows * (cols / 8);
    hipLaunchKernelGGL(split_planes_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ST(stream), x, static_cast<long>(ldx), static_cast<long>(rows),
                       static_cast<int>(cols / 8), static_cast<unsigned short*>(planes), static_cast<long>(ldp), static_cast<long>(lo_off));
  } else {
    hipLaunchKernelGGL(split_planes_t_kernel, dim3(static_cast<unsigned>((cols + kSpT - 1) / kSpT), static_cast<unsigned>((rows + kSpT - 1) / kSpT)), dim3(256), 0,
                       ST(stream), x, static_cast<long>(ldx), static_cast<int>(rows), static_cast<int>(cols), static_cast<unsigned short*>(planes), static_cast<long>(ldp),
                       static_cast<long>(lo_off));
  }
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
