// gemm.hip -- K3/K5: bf16 MFMA GEMM with fused epilogues (gfx950).
//
// Stands behind every nn.Linear of the ADT network (reference model.py:111,157,224 and the
// in_proj/out_proj/linear1/linear2 of nn.TransformerEncoderLayer / nn.TransformerDecoderLayer
// built at model.py:118-127,159-168) and their backward passes.
//
//   trans = 0 (NT):  C[M,N] = A[M,K] . B[N,K]^T     forward  y = x W^T ; dgrad  dx = dy (W^T)^T
//   trans = 1 (TN):  C[M,N] = A[K,M]^T . B[K,N]     wgrad    dW = dy^T x   (K = rows of dy and x)
//
// Tile 128x128x64, 256 threads = 4 waves in a 2x2 grid, each wave 64x64 = 4x4 MFMA
// v_mfma_f32_16x16x32_bf16 blocks (64 fp32 accumulators).  Operands are staged
// global -> registers -> LDS (16-byte loads, padded rows so fragment reads are bank-conflict
// free), double-buffered with one barrier per K-tile; the next tile's global loads are issued
// before the current tile's MFMAs.
//   NT fragments: ds_read_b128 of 8 consecutive k of one row (row pitch 144 B).
//   TN fragments: two ds_read_b64_tr_b16 per operand from a [k][m] image (row pitch 288 B);
//                 element j of lane group g holds k = 4g + (j&3) + 16*(j>>2) for both operands,
//                 which makes the two lane groups of a half-wave read 8 distinct 32-byte rows.
// TN with split-K writes fp32 partial slabs that reduce_slabs_kernel sums in a fixed order
// (bitwise reproducible, no float atomics).
#include <hip/hip_runtime.h>

#include "adt_common.h"
#include "dropout.h"

namespace adt {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kBM = 128, kBN = 128, kBK = 64;
constexpr int kGemmThreads = 256;
constexpr int kPitchNT = 144;                  // bytes per LDS row: 64 bf16 + 16 B pad
constexpr int kPitchTN = 288;                  // bytes per LDS row: 128 bf16 + 32 B pad
constexpr int kTileBytes = 128 * kPitchNT;     // 18,432 B (== 64 * kPitchTN)
static_assert(kTileBytes == 64 * kPitchTN, "tile images have one size");
constexpr int kStageBytes = 2 * kTileBytes;    // A + B
constexpr int kGemmLds = 2 * kStageBytes;      // double buffered: 73,728 B

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {      // round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {  // byte offset inside the workgroup's LDS
  return static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)p));
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

struct GemmArgs {
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  void* C; long ldc;
  int M, N, K;
  int k_tiles_per_split;
  float* slabs;                 // TN split-K partials [splits][M][N] (null when splits == 1)
  adt_gemm_epilogue ep;
  Drop drop;
};

// ---- global -> register staging (4 x 16 B per thread per operand) -----------------------------
// NT: the tile is 128 rows (m or n) x 64 k.  chunk c = tid + 256*i: row = c >> 3, 16-byte piece = c & 7.
// TN: the tile is 64 rows (k) x 128 cols (m or n).  chunk: row = c >> 4, piece = c & 15.
template <bool kTrans>
__device__ __forceinline__ void stage_load(const unsigned short* __restrict__ src, long ld, int row0, int col0,
                                           int n_rows, int n_cols, int tid, uint4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + kGemmThreads * i;
    const int row = kTrans ? (c >> 4) : (c >> 3);
    const int col = (kTrans ? (c & 15) : (c & 7)) * 8;
    const int gr = row0 + row, gc = col0 + col;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gr < n_rows && gc < n_cols) v = *reinterpret_cast<const uint4*>(src + static_cast<long>(gr) * ld + gc);
    r[i] = v;
  }
}
template <bool kTrans>
__device__ __forceinline__ void stage_store(unsigned char* lds, int tid, const uint4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + kGemmThreads * i;
    const int row = kTrans ? (c >> 4) : (c >> 3);
    const int piece = kTrans ? (c & 15) : (c & 7);
    *reinterpret_cast<uint4*>(lds + row * (kTrans ? kPitchTN : kPitchNT) + piece * 16) = r[i];
  }
}

// ---- LDS -> MFMA fragments ---------------------------------------------------------------------
// NT: 16 rows x 32 k; lane l takes row (l & 15), k = 8*(l >> 4) .. +7.
__device__ __forceinline__ bf16x8 frag_nt(const unsigned char* tile, int row0, int k0, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + (row0 + (lane & 15)) * kPitchNT + (k0 + 8 * (lane >> 4)) * 2);
}
// TN: image is [k][col]; two transposed 8-byte reads give this lane column (col0 + (l & 15)) of
// rows k0 + 4g + {0..3} and k0 + 16 + 4g + {0..3}, g = l >> 4.
__device__ __forceinline__ bf16x8 frag_tn(const unsigned char* tile, int col0, int k0, int lane) {
  const int t = lane & 15, g = lane >> 4;
  const unsigned a0 = lds_addr(tile) + (k0 + 4 * g + (t >> 2)) * kPitchTN + (col0 + 4 * (t & 3)) * 2;
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi) : "v"(a0), "i"(16 * kPitchTN) : "memory");
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// ---- epilogue shared by the kernels: C element (row, col) = acc[i][j][r] with
// row = m0 + wm*64 + i*16 + 4*(lane>>4) + r, col = n0 + wn*64 + j*16 + (lane&15)
template <bool kDrop>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int lane, int split) {
  const adt_gemm_epilogue& ep = g.ep;
  const bool to_slab = g.slabs != nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + (lane & 15);
      if (col >= g.N) continue;
      const float bias = (!to_slab && ep.bias) ? ep.bias[col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + 4 * (lane >> 4) + r;
        if (row >= g.M) continue;
        float z = acc[i][j][r];
        if (to_slab) {
          g.slabs[(static_cast<long>(split) * g.M + row) * g.N + col] = z;
          continue;
        }
        z = z * ep.alpha + bias;
        if (ep.gelu_grad_of) {
          const unsigned short u = reinterpret_cast<const unsigned short*>(ep.gelu_grad_of)[static_cast<long>(row) * ep.ld_gelu_grad + col];
          z *= gelu_erf_grad(bf2f(u));
        }
        if (ep.pre_act_out)
          reinterpret_cast<unsigned short*>(ep.pre_act_out)[static_cast<long>(row) * ep.ld_pre_act + col] = f2bf(z);
        if (ep.act == 1) z = gelu_erf(ep.pre_act_out ? bf2f(f2bf(z)) : z);
        else if (ep.act == 2) z = fmaxf(z, 0.0f);
        const float keep = kDrop ? g.drop.scale(static_cast<uint64_t>(row) * g.N + col) : 1.0f;
        if (kDrop && !ep.drop_after_residual) z *= keep;
        if (ep.residual) {
          const long rr = ep.res_row_mod > 0 ? (row % ep.res_row_mod) : row;
          z += reinterpret_cast<const float*>(ep.residual)[rr * ep.ld_res + col];
        }
        if (kDrop && ep.drop_after_residual) z *= keep;
        if (ep.aux_bf16_out) reinterpret_cast<unsigned short*>(ep.aux_bf16_out)[static_cast<long>(row) * ep.ld_aux + col] = f2bf(z);
        if (ep.out_fp32) reinterpret_cast<float*>(g.C)[static_cast<long>(row) * g.ldc + col] = z;
        else reinterpret_cast<unsigned short*>(g.C)[static_cast<long>(row) * g.ldc + col] = f2bf(z);
      }
    }
  }
}

template <bool kTrans, bool kDrop>
__global__ __launch_bounds__(kGemmThreads) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * kBM, n0 = blockIdx.x * kBN;
  const int split = blockIdx.z;
  const int kt0 = split * g.k_tiles_per_split;
  int kt1 = kt0 + g.k_tiles_per_split;
  const int k_tiles = (g.K + kBK - 1) / kBK;
  if (kt1 > k_tiles) kt1 = k_tiles;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  auto load_tile = [&](int kt) {
    if (kTrans) {
      stage_load<true>(g.A, g.lda, kt * kBK, m0, g.K, g.M, tid, ra);
      stage_load<true>(g.B, g.ldb, kt * kBK, n0, g.K, g.N, tid, rb);
    } else {
      stage_load<false>(g.A, g.lda, m0, kt * kBK, g.M, g.K, tid, ra);
      stage_load<false>(g.B, g.ldb, n0, kt * kBK, g.N, g.K, tid, rb);
    }
  };

  if (kt0 < kt1) {
    load_tile(kt0);
    stage_store<kTrans>(smem, tid, ra);
    stage_store<kTrans>(smem + kTileBytes, tid, rb);
  }
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    const unsigned char* ta = smem + cur * kStageBytes;
    const unsigned char* tb = ta + kTileBytes;
    const bool more = kt + 1 < kt1;
    if (more) load_tile(kt + 1);                       // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < kBK / 32; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = kTrans ? frag_tn(ta, wm * 64 + i * 16, ks * 32, lane) : frag_nt(ta, wm * 64 + i * 16, ks * 32, lane);
        fb[i] = kTrans ? frag_tn(tb, wn * 64 + i * 16, ks * 32, lane) : frag_nt(tb, wn * 64 + i * 16, ks * 32, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      unsigned char* nxt = smem + (cur ^ 1) * kStageBytes;
      stage_store<kTrans>(nxt, tid, ra);
      stage_store<kTrans>(nxt + kTileBytes, tid, rb);
    }
    __syncthreads();
  }

  gemm_epilogue<kDrop>(g, acc, m0, n0, wm, wn, lane, split);
}

// ---- row-vector epilogue (LDS-DMA kernel): the 128x128 fp32 accumulator tile is transposed through
// LDS (row pitch 132 floats: conflict-free 4-byte writes from the MFMA layout), then each thread
// owns 8 consecutive columns of a row, so bias / residual / pre-activation are 16- and 32-byte
// vector accesses and every output row is written as whole 16-byte pieces.
constexpr int kEpiPitch = 132;
constexpr int kEpiLds = 128 * kEpiPitch * 4;          // 67,584 B
template <bool kDrop>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmArgs& g, f32x4 (&acc)[4][4], float* ct, int m0, int n0,
                                                   int wm, int wn, int tid, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ct[(wm * 64 + i * 16 + 4 * (lane >> 4) + r) * kEpiPitch + wn * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
  __syncthreads();
  const adt_gemm_epilogue& ep = g.ep;
  const int c8 = (tid & 15) * 8;
  const int col = n0 + c8;
  const bool full = col + 8 <= g.N;                     // N % 8 == 0 is guaranteed for bf16 C (checked on the host)
  float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ep.bias && full) {
    *reinterpret_cast<float4*>(bias) = *reinterpret_cast<const float4*>(ep.bias + col);
    *reinterpret_cast<float4*>(bias + 4) = *reinterpret_cast<const float4*>(ep.bias + col + 4);
  }
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int lr = pass * 16 + (tid >> 4);
    const int row = m0 + lr;
    if (row >= g.M || !full) continue;
    float z[8];
    *reinterpret_cast<float4*>(z) = *reinterpret_cast<const float4*>(ct + lr * kEpiPitch + c8);
    *reinterpret_cast<float4*>(z + 4) = *reinterpret_cast<const float4*>(ct + lr * kEpiPitch + c8 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = z[e] * ep.alpha + bias[e];
    if (ep.gelu_grad_of) {
      const uint4 uv = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(ep.gelu_grad_of) + static_cast<long>(row) * ep.ld_gelu_grad + col);
      const unsigned w[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        z[2 * e] *= gelu_erf_grad(__uint_as_float(w[e] << 16));
        z[2 * e + 1] *= gelu_erf_grad(__uint_as_float(w[e] & 0xffff0000u));
      }
    }
    if (ep.pre_act_out) {
      uint4 o;
      o.x = f2bf(z[0]) | (static_cast<unsigned>(f2bf(z[1])) << 16); o.y = f2bf(z[2]) | (static_cast<unsigned>(f2bf(z[3])) << 16);
      o.z = f2bf(z[4]) | (static_cast<unsigned>(f2bf(z[5])) << 16); o.w = f2bf(z[6]) | (static_cast<unsigned>(f2bf(z[7])) << 16);
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(ep.pre_act_out) + static_cast<long>(row) * ep.ld_pre_act + col) = o;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = bf2f(f2bf(z[e]));       // the activation sees the value the backward will read
    }
    if (ep.act == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = gelu_erf(z[e]);
    } else if (ep.act == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = fmaxf(z[e], 0.0f);
    }
    float keep[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (kDrop) {
      const uint64_t base = static_cast<uint64_t>(row) * g.N + col;
#pragma unroll
      for (int e = 0; e < 8; ++e) keep[e] = g.drop.scale(base + e);
      if (!ep.drop_after_residual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] *= keep[e];
      }
    }
    if (ep.residual) {
      const long rr = ep.res_row_mod > 0 ? (row % ep.res_row_mod) : row;
      const float* rp = reinterpret_cast<const float*>(ep.residual) + rr * ep.ld_res + col;
      const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
      z[0] += r0.x; z[1] += r0.y; z[2] += r0.z; z[3] += r0.w; z[4] += r1.x; z[5] += r1.y; z[6] += r1.z; z[7] += r1.w;
    }
    if (kDrop && ep.drop_after_residual) {
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] *= keep[e];
    }
    uint4 o16;
    o16.x = f2bf(z[0]) | (static_cast<unsigned>(f2bf(z[1])) << 16); o16.y = f2bf(z[2]) | (static_cast<unsigned>(f2bf(z[3])) << 16);
    o16.z = f2bf(z[4]) | (static_cast<unsigned>(f2bf(z[5])) << 16); o16.w = f2bf(z[6]) | (static_cast<unsigned>(f2bf(z[7])) << 16);
    if (ep.aux_bf16_out)
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(ep.aux_bf16_out) + static_cast<long>(row) * ep.ld_aux + col) = o16;
    if (ep.out_fp32) {
      float* cp = reinterpret_cast<float*>(g.C) + static_cast<long>(row) * g.ldc + col;
      *reinterpret_cast<float4*>(cp) = *reinterpret_cast<float4*>(z);
      *reinterpret_cast<float4*>(cp + 4) = *reinterpret_cast<float4*>(z + 4);
    } else {
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(g.C) + static_cast<long>(row) * g.ldc + col) = o16;
    }
  }
}

// =========================================================================================
// NT kernel, LDS-DMA staging (used whenever K % 32 == 0).
// Both operand tiles are 128 rows x 128 B and are filled by global_load_lds_dwordx4 (no VGPR
// round trip): one wave-instruction writes 1 KiB = 8 rows, lane l -> row (l >> 3), 16-byte
// position (l & 7).  The LDS image must stay lane-linear, so the bank swizzle is applied to the
// SOURCE: position p of row r holds the row's chunk p ^ ((r >> 1) & 7), and fragment reads apply
// the same XOR -- 16 consecutive rows then hit 16 distinct 16-byte bank slots (conflict free).
// Two stages: the next K-tile's DMA is issued before the current tile's MFMAs and retired by the
// vmcnt(0) + barrier that ends the iteration.  Rows past M / N are clamped (read, never stored).
// Blocks are renumbered so that the blocks sharing an XCD (id % 8) work on consecutive tiles of
// one row-panel, which keeps the A panel in that XCD's L2.
constexpr int kTileNT = 128 * 128;               // 16 KiB per operand tile
__device__ __forceinline__ int swz_nt(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void glds_tile(const unsigned short* __restrict__ src, long ld, int row0, int n_rows, int k0, int k_end,
                                          unsigned char* tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + (lane >> 3);
    int gr = row0 + row;
    gr = gr < n_rows ? gr : n_rows - 1;
    const int chunk = (lane & 7) ^ swz_nt(row);
    int kc = k0 + chunk * 8;
    kc = kc < k_end ? kc : k_end - 8;              // half-filled last tile (K % 64 == 32): re-read valid data, never multiplied
    const unsigned short* p = src + static_cast<long>(gr) * ld + kc;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)(tile + (4 * wave + i) * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ bf16x8 frag_glds(const unsigned char* tile, int row0, int ks, int lane) {
  const int row = row0 + (lane & 15);
  const int chunk = (4 * ks + (lane >> 4)) ^ swz_nt(row);
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + chunk * 16);
}

template <bool kDrop>
__global__ __launch_bounds__(kGemmThreads) void gemm_nt_glds_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 stages][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware renumbering (bijective for any grid size)
  const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int m0 = (logical / tiles_n) * kBM, n0 = (logical % tiles_n) * kBN;
  const int k_tiles = (g.K + kBK - 1) / kBK;
  const bool half_tail = (g.K & 32) != 0;          // K % 64 == 32: the last tile carries one 32-deep k-step

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  glds_tile(g.A, g.lda, m0, g.M, 0, g.K, smem, wave, lane);
  glds_tile(g.B, g.ldb, n0, g.N, 0, g.K, smem + kTileNT, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < k_tiles; ++kt) {
    const unsigned char* ta = smem + (kt & 1) * 2 * kTileNT;
    const unsigned char* tb = ta + kTileNT;
    if (kt + 1 < k_tiles) {
      unsigned char* na = smem + ((kt + 1) & 1) * 2 * kTileNT;
      glds_tile(g.A, g.lda, m0, g.M, (kt + 1) * kBK, g.K, na, wave, lane);
      glds_tile(g.B, g.ldb, n0, g.N, (kt + 1) * kBK, g.K, na + kTileNT, wave, lane);
    }
    // both 32-deep k-steps' fragments are requested up front, so the second step's LDS reads
    // are in flight under the first step's MFMAs
    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[ks][i] = frag_glds(ta, wm * 64 + i * 16, ks, lane);
        fb[ks][i] = frag_glds(tb, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 1 && half_tail && kt + 1 == k_tiles) break;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  gemm_epilogue_rows<kDrop>(g, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, tid, lane);
}

// =========================================================================================
// TN kernel (weight gradients), LDS-DMA staging (used when K % 64 == 0).
// Operand tiles are [64 k][128 cols] (256-byte rows); one wave-instruction of global_load_lds
// fills 4 rows.  Position p of row r holds the row's 16-byte chunk p ^ (2*(r & 7)), which puts
// the 8 consecutive rows a half-wave reads with ds_read_b64_tr_b16 on 8 distinct 32-byte bank
// slots.  All 16 transposed reads of a 32-deep k-step are issued back to back and retired by one
// lgkmcnt(0) before the 16 MFMAs.  Columns past M / N are clamped (read, never stored).
__device__ __forceinline__ int swz_tn(int row) { return (row & 7) << 1; }

__device__ __forceinline__ void glds_tile_tn(const unsigned short* __restrict__ src, long ld, int k0, int col0, int n_cols,
                                             unsigned char* tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * (4 * wave + i) + (lane >> 4);
    const int chunk = (lane & 15) ^ swz_tn(row);
    int gc = col0 + chunk * 8;
    gc = gc + 8 <= n_cols ? gc : n_cols - 8;
    const unsigned short* p = src + static_cast<long>(k0 + row) * ld + gc;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)(tile + (4 * wave + i) * 1024), 16, 0, 0);
  }
}
// issue the two transposed reads of one 16-column x 32-k fragment (no wait)
__device__ __forceinline__ void tr_issue(unsigned tile_base, int col0, int ks, int lane, bf16x4& lo, bf16x4& hi) {
  const int t = lane & 15, g = lane >> 4;
  const int row = ks * 32 + 4 * g + (t >> 2);
  const int chunk = ((col0 + 4 * (t & 3)) >> 3) ^ swz_tn(row);
  const unsigned a0 = tile_base + row * 256 + chunk * 16 + 8 * (t & 1);
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4096" : "=&v"(lo), "=&v"(hi) : "v"(a0) : "memory");
}
__device__ __forceinline__ bf16x8 join8(const bf16x4& lo, const bf16x4& hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

__global__ __launch_bounds__(kGemmThreads) void gemm_tn_glds_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 stages][A tile | B tile], reused by the epilogue
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int m0 = (logical / tiles_n) * kBM, n0 = (logical % tiles_n) * kBN;
  const int split = blockIdx.y;
  const int k_tiles = g.K / kBK;
  const int kt0 = split * g.k_tiles_per_split;
  int kt1 = kt0 + g.k_tiles_per_split;
  if (kt1 > k_tiles) kt1 = k_tiles;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) {
    glds_tile_tn(g.A, g.lda, kt0 * kBK, m0, g.M, smem, wave, lane);
    glds_tile_tn(g.B, g.ldb, kt0 * kBK, n0, g.N, smem + kTileNT, wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    const unsigned ta = lds_addr(smem + cur * 2 * kTileNT), tb = ta + kTileNT;
    if (kt + 1 < kt1) {
      unsigned char* na = smem + (cur ^ 1) * 2 * kTileNT;
      glds_tile_tn(g.A, g.lda, (kt + 1) * kBK, m0, g.M, na, wave, lane);
      glds_tile_tn(g.B, g.ldb, (kt + 1) * kBK, n0, g.N, na + kTileNT, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < kBK / 32; ++ks) {
      bf16x4 alo[4], ahi[4], blo[4], bhi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tr_issue(ta, wm * 64 + i * 16, ks, lane, alo[i], ahi[i]);
        tr_issue(tb, wn * 64 + i * 16, ks, lane, blo[i], bhi[i]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { fa[i] = join8(alo[i], ahi[i]); fb[i] = join8(blo[i], bhi[i]); }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  GemmArgs o = g;
  if (g.slabs) {                       // split-K partial: plain fp32 tile into this split's slab
    o.C = g.slabs + static_cast<long>(split) * g.M * g.N;
    o.ldc = g.N;
    o.ep = adt_gemm_epilogue{};
    o.ep.alpha = 1.0f;
    o.ep.out_fp32 = 1;
    o.drop = Drop{0u, 0u, 1.0f};
  }
  gemm_epilogue_rows<false>(o, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, tid, lane);
}

// sums split-K slabs in slab order: out[m,n] = alpha * sum_s slab[s][m,n]   (fp32 out)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int splits, long mn, int N,
                                                           float alpha, float* __restrict__ out, long ldc) {
  const long i4 = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i4 >= mn) return;
  float4 s = *reinterpret_cast<const float4*>(slabs + i4);
  for (int k = 1; k < splits; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(slabs + k * mn + i4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const long row = i4 / N; const int col = static_cast<int>(i4 - row * N);   // N % 4 == 0: a float4 never straddles rows
  float* o = out + row * ldc + col;
  o[0] = s.x * alpha; o[1] = s.y * alpha; o[2] = s.z * alpha; o[3] = s.w * alpha;
}

// the row-vector epilogue moves 8 columns at a time: everything it touches must be 16-byte aligned
static bool vector_epilogue_ok(const GemmArgs& g, const adt_gemm_epilogue& e) {
  auto ok = [](const void* p, long ld, int elem) { return !p || (aligned16(p) && (ld * elem) % 16 == 0); };
  return (g.N % 8) == 0 && ok(g.C, g.ldc, e.out_fp32 ? 4 : 2) && ok(e.bias, 4, 4) && ok(e.residual, e.ld_res, 4) &&
         ok(e.pre_act_out, e.ld_pre_act, 2) && ok(e.gelu_grad_of, e.ld_gelu_grad, 2) && ok(e.aux_bf16_out, e.ld_aux, 2);
}

static int pick_splits(int M, int N, int K, int n_cu) {
  const int tiles = ((M + kBM - 1) / kBM) * ((N + kBN - 1) / kBN);
  const int k_tiles = (K + kBK - 1) / kBK;
  int s = 1;
  while (tiles * s < 2 * n_cu && s * 2 <= k_tiles / 4 && s < 64) s *= 2;
  return s;
}

}  // namespace adt

extern "C" size_t adt_gemm_workspace_bytes(int32_t trans, int64_t M, int64_t N, int64_t K) {
  if (!trans || M <= 0 || N <= 0 || K <= 0) return 0;
  int n_cu = 256;
  (void)adt::device_cu_count(&n_cu);
  const int s = adt::pick_splits(static_cast<int>(M), static_cast<int>(N), static_cast<int>(K), n_cu);
  return s > 1 ? static_cast<size_t>(s) * M * N * 4 : 0;
}

extern "C" int adt_gemm_bf16(int32_t trans, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                             const void* B, int64_t ldb, void* C, int64_t ldc, const adt_gemm_epilogue* ep,
                             void* ws, size_t ws_bytes, void* stream) {
  using namespace adt;
  if (!A || !B || !C) return set_error(ADT_EINVAL, "adt_gemm_bf16: null pointer");
  if (M < 0 || N < 0 || K < 0) return set_error(ADT_EINVAL, "adt_gemm_bf16: negative size");
  if (M >= (1ll << 31) || N >= (1ll << 31) || K >= (1ll << 31)) return set_error(ADT_ESHAPE, "adt_gemm_bf16: dimension >= 2^31");
  const int64_t a_cols = trans ? M : K, b_cols = trans ? N : K;
  if (lda < a_cols || ldb < b_cols || ldc < N) return set_error(ADT_EINVAL, "adt_gemm_bf16: leading dimension too small");
  if ((a_cols & 7) || (b_cols & 7) || (lda & 7) || (ldb & 7) || !aligned16(A) || !aligned16(B))
    return set_error(ADT_ESHAPE, "adt_gemm_bf16: operand rows must be 16-byte aligned multiples of 8 elements");
  if (M == 0 || N == 0) return ADT_OK;
  GemmArgs g;
  g.A = static_cast<const unsigned short*>(A); g.lda = lda;
  g.B = static_cast<const unsigned short*>(B); g.ldb = ldb;
  g.C = C; g.ldc = ldc; g.M = static_cast<int>(M); g.N = static_cast<int>(N); g.K = static_cast<int>(K);
  adt_gemm_epilogue e;
  if (ep) e = *ep; else { e = adt_gemm_epilogue{}; e.alpha = 1.0f; }
  g.ep = e;
  g.drop = make_drop(e.drop.p, e.drop.key);
  const int k_tiles = static_cast<int>((K + kBK - 1) / kBK);
  int splits = 1;
  if (trans) {
    int n_cu = 0;
    if (int rc = device_cu_count(&n_cu)) return rc;
    splits = pick_splits(g.M, g.N, g.K, n_cu);
    if (splits > 1) {
      if (!e.out_fp32 || e.bias || e.residual || e.act || e.pre_act_out || e.gelu_grad_of || e.aux_bf16_out || e.drop.p > 0.f || (N & 3))
        splits = 1;                                  // split-K only for the plain fp32 weight-gradient form
      else if (!ws || ws_bytes < static_cast<size_t>(splits) * M * N * 4)
        return set_error(ADT_EINVAL, "adt_gemm_bf16: workspace too small (see adt_gemm_workspace_bytes)");
    }
  }
  g.k_tiles_per_split = (k_tiles + splits - 1) / splits;
  g.slabs = splits > 1 ? static_cast<float*>(ws) : nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>((N + kBN - 1) / kBN), static_cast<unsigned>((M + kBM - 1) / kBM), splits);
  static thread_local int attr_dev = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (attr_dev != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_glds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kEpiLds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_glds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kEpiLds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_glds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kEpiLds));
    attr_dev = dev;
  }
  if (trans && (K % kBK) == 0 && K > 0 && vector_epilogue_ok(g, e) && M >= 8 && (splits == 1 || aligned16(ws))) {
    const int tm = static_cast<int>((M + kBM - 1) / kBM), tn = static_cast<int>((N + kBN - 1) / kBN);
    hipLaunchKernelGGL(gemm_tn_glds_kernel, dim3(static_cast<unsigned>(tm) * tn, splits), dim3(kGemmThreads), kEpiLds, st, g, tm, tn);
  } else if (trans) {
    if (g.drop.on()) return set_error(ADT_EINVAL, "adt_gemm_bf16: dropout is not supported with trans = 1");
    hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, dim3(kGemmThreads), kGemmLds, st, g);
  } else if ((K % 32) == 0 && K > 0 && vector_epilogue_ok(g, e)) {
    const int tm = static_cast<int>((M + kBM - 1) / kBM), tn = static_cast<int>((N + kBN - 1) / kBN);
    const dim3 g1(static_cast<unsigned>(tm) * tn);
    if (g.drop.on()) hipLaunchKernelGGL(gemm_nt_glds_kernel<true>, g1, dim3(kGemmThreads), kEpiLds, st, g, tm, tn);
    else hipLaunchKernelGGL(gemm_nt_glds_kernel<false>, g1, dim3(kGemmThreads), kEpiLds, st, g, tm, tn);
  } else {
    if (g.drop.on()) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, dim3(kGemmThreads), kGemmLds, st, g);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, dim3(kGemmThreads), kGemmLds, st, g);
  }
  if (splits > 1) {
    const long mn = static_cast<long>(M) * N;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(static_cast<unsigned>((mn / 4 + 255) / 256)), dim3(256), 0, st,
                       g.slabs, splits, mn, g.N, e.alpha, static_cast<float*>(C), ldc);
  }
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
