// gemm.hip -- K3/K5: bf16 MFMA GEMM with fused epilogues (gfx950).
//
// Stands behind every nn.Linear of the ADT network (reference model.py:111,157,224 and the
// in_proj/out_proj/linear1/linear2 of nn.TransformerEncoderLayer / nn.TransformerDecoderLayer
// built at model.py:118-127,159-168) and their backward passes.
//
//   trans = 0 (NT):  C[M,N] = A[M,K] . B[N,K]^T     forward  y = x W^T ; dgrad  dx = dy (W^T)^T
//   trans = 1 (TN):  C[M,N] = A[K,M]^T . B[K,N]     wgrad    dW = dy^T x   (K = rows of dy and x)
//
// Three kernel structures, picked per call by adt_gemm_bf16 (all v_mfma_f32_16x16x32_bf16, fp32 accumulate):
//   * persistent 256x256x64 kernels (gemm_nt_256_kernel, gemm_tn_256_kernel) for the large GEMMs of the training step:
//     8 waves, 4 phases of 16 MFMAs per K-tile, LDS-DMA (global_load_lds_dwordx4) half-tiles in flight across raw
//     barriers, staggered wave rows, per-XCD-slice work counters, next tile's prologue under the epilogue;
//   * 128x128x64 LDS-DMA kernels (gemm_nt_glds_kernel: any K, the last K-tile zero-filled; gemm_tn_glds_kernel: split-K through fp32 slabs)
//     for everything smaller: 4 waves in a 2x2 grid, each 64x64 = 4x4 MFMA blocks, 2-stage pipeline, XCD-aware tile order;
//   * a register-staged 128x128x64 kernel (gemm_bf16_kernel) for the remaining shapes (TN with K % 64 != 0, outputs the row-vector epilogue cannot address).
//   NT fragments: ds_read_b128 of 8 consecutive k of one row.
//   TN fragments: two ds_read_b64_tr_b16 per operand from a [k][m] image.
// Epilogues transpose the accumulators through LDS so that bias / GELU / GELU' / ReLU / dropout / residual / dual-dtype
// stores work on 8 consecutive columns of a row (16- and 32-byte accesses).
// TN with split-K writes fp32 partial slabs that reduce_slabs_kernel sums in a fixed order
// (bitwise reproducible, no float atomics).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "adt_common.h"
#include "dropout.h"

namespace adt {

// Cache-policy bits of the persistent NT kernel's operand DMAs (aux of global_load_lds: 1 = sc0, 2 = nt, 16 = sc1) and whether the main
// output leaves by non-temporal stores (A/B arms through tools/build_variant.sh).  Measured in round 6 (profiles/r06/gemm_cache_policy_ab.txt):
// nt on either operand costs the K = 768 shapes +10...15 % (the panels' reuse in L2 goes), nt on the main output is +-0 on the step; the
// saved-factor output stays non-temporal (round 4).
#ifndef ADT_NT_A_AUX
#define ADT_NT_A_AUX 0
#endif
#ifndef ADT_NT_B_AUX
#define ADT_NT_B_AUX 0
#endif
#ifndef ADT_NT_C_NT
#define ADT_NT_C_NT 0
#endif
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
#include "gelu.h"
#define ADT_DS_READ_B128_ADDR(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))

struct GemmArgs {
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  void* C; long ldc;
  int M, N, K;
  int k_tiles_per_split;
  float* slabs;                 // TN split-K partials [splits][M][N] (null when splits == 1)
  float* colsum_ws;             // 256^2 NT kernel: column sums of each 128-row band of the output, [ceil(M/256)*2][N]
  adt_gemm_epilogue ep;
  Drop drop; unsigned drop_key2;      // drop_key2 = mix32(drop.key)
  int tail_stores;                         // gemm_nt_256_kernel: leave an interior tile's last stores in flight across the tile boundary
  int group_n;                             // gemm_nt_256_kernel: tile columns per column group of the tile order (>= tiles_n: row-major)
  int stagger;                             // gemm_nt_256_kernel, experiment builds only (-DADT_GEMM_EXPERIMENT, ADT_GEMM_STAGGER=<s_memtime ticks>): every second workgroup of an XCD group starts late
  unsigned* sched; unsigned sched_total[8];  // persistent kernels: per-XCD-group work counters (16 words apart, zero between launches) and the number of tickets each hands out in this launch
  // split-bf16 products on the persistent kernels (adt_gemm_bf16x3; kX3 instantiations only): the operands are [hi | lo] bf16 plane pairs
  // of fp32 matrices, K counts VIRTUAL K-tiles -- three segments of x3_kt tiles: lo x hi, hi x lo, hi x hi -- and a segment's tile r is read
  // from the plane its segment names (lo plane = x3_a_lo / x3_b_lo elements to the right of the hi plane)
  int x3_kt; long x3_a_lo, x3_b_lo;
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
          z *= ep.act_grad_mode ? bf2f(u) : gelu_erf_grad(bf2f(u));
        }
        const float keep = kDrop ? g.drop.scale(static_cast<uint64_t>(row) * drop_ld(g.N) + col) : 1.0f;
        if (ep.act_grad_mode && ep.pre_act_out && ep.act == 1) {
          reinterpret_cast<unsigned short*>(ep.pre_act_out)[static_cast<long>(row) * ep.ld_pre_act + col] = f2bf(gelu_erf_grad(z) * keep);
          z = gelu_erf(z);
        } else {
          if (ep.pre_act_out)
            reinterpret_cast<unsigned short*>(ep.pre_act_out)[static_cast<long>(row) * ep.ld_pre_act + col] = f2bf(z);
          if (ep.act == 1) z = gelu_erf(ep.pre_act_out ? bf2f(f2bf(z)) : z);
          else if (ep.act == 2) z = fmaxf(z, 0.0f);
        }
        if (kDrop && !ep.drop_after_residual) z *= keep;
        if (ep.residual) {
          const long rr = ep.res_row_mod > 0 ? (row % ep.res_row_mod) : row;
          const float rv = reinterpret_cast<const float*>(ep.residual)[rr * ep.ld_res + col];
          z += ep.res_ln_mean ? (rv - ep.res_ln_mean[row]) * ep.res_ln_rstd[row] * ep.res_ln_gamma[col] + ep.res_ln_beta[col] : rv;
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
// One row piece of 8 consecutive columns: z = acc * alpha + bias -> [gelu'] -> [pre-act out] -> act -> dropout / residual -> stores.
// Epilogue form as a compile-time mask (the persistent kernel is instantiated for the forms the training step launches, so an
// output piece costs its arithmetic and not a dozen uniform branches, selects and register shuffles); kEpiGeneric reads the
// flags of adt_gemm_epilogue at run time.
enum : unsigned {
  kEfBias = 1u, kEfGeluGrad = 2u, kEfFactor = 4u, kEfPreAct = 8u, kEfGelu = 16u, kEfRelu = 32u, kEfResidual = 64u, kEfRowMod = 128u,
  kEfDropAfterRes = 256u, kEfAux = 512u, kEfFp32 = 1024u, kEfAlpha = 2048u, kEfResLn = 4096u, kEpiGeneric = 0x80000000u
};
static unsigned epilogue_mask(const adt_gemm_epilogue& e) {
  if (e.side_fp32) return kEpiGeneric;                  // fp32 saved-factor / pre-activation arrays (adt_gemm_bf16x3): run-time flags only
  return (e.bias ? kEfBias : 0u) | (e.gelu_grad_of ? kEfGeluGrad : 0u) | (e.act_grad_mode ? kEfFactor : 0u) | (e.pre_act_out ? kEfPreAct : 0u) |
         (e.act == 1 ? kEfGelu : 0u) | (e.act == 2 ? kEfRelu : 0u) | (e.residual ? kEfResidual : 0u) | (e.residual && e.res_row_mod > 0 ? kEfRowMod : 0u) |
         (e.drop.p > 0.f && e.drop_after_residual ? kEfDropAfterRes : 0u) | (e.aux_bf16_out ? kEfAux : 0u) | (e.out_fp32 ? kEfFp32 : 0u) | (e.alpha != 1.0f ? kEfAlpha : 0u) | (e.residual && e.res_ln_mean ? kEfResLn : 0u);
}
template <unsigned kMask, unsigned kBit>
__device__ __forceinline__ bool ef(bool at_run_time) {
  if constexpr ((kMask & kEpiGeneric) != 0u) return at_run_time;
  else return (kMask & kBit) != 0u;
}
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {             // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2v));
}
__device__ __forceinline__ uint4 pack_bf8(const float (&z)[8]) {
  return uint4{pack_bf2(z[0], z[1]), pack_bf2(z[2], z[3]), pack_bf2(z[4], z[5]), pack_bf2(z[6], z[7])};
}
// One row piece of 8 consecutive columns: z = acc * alpha + bias -> [gelu'] -> [pre-act out] -> act -> dropout / residual -> stores.
// Element offsets of one lane's (row0, col) in every array the epilogue touches: formed once per tile, outside the per-piece
// conditionals, so that a piece at row0 + irow (irow a compile-time constant) costs one 64-bit add of a scalar per array.
struct EpiAddr { long c, pre, gg, res, aux, hash; };
template <bool kDrop, unsigned kMask>
__device__ __forceinline__ EpiAddr epilogue_addr(const GemmArgs& g, int row0, int col) {
  const adt_gemm_epilogue& ep = g.ep;
  const long r = row0;
  EpiAddr a;
  a.c = r * g.ldc + col;
  a.pre = ef<kMask, kEfPreAct>(true) ? r * ep.ld_pre_act + col : 0;
  a.gg = ef<kMask, kEfGeluGrad>(true) ? r * ep.ld_gelu_grad + col : 0;
  a.res = ef<kMask, kEfResidual>(true) ? r * ep.ld_res + col : 0;
  a.aux = ef<kMask, kEfAux>(true) ? r * ep.ld_aux + col : 0;
  a.hash = kDrop ? r * g.N + col : 0;
  return a;
}
template <bool kDrop, unsigned kMask = kEpiGeneric>
__device__ __forceinline__ void epilogue_apply8(const GemmArgs& g, float (&z)[8], const float (&bias)[8], const EpiAddr& ea, int row0, int irow, int col) {
  // The piece is row (row0 + irow), columns col .. col + 7; ea = epilogue_addr(row0, col).
  const adt_gemm_epilogue& ep = g.ep;
  const int row = row0 + irow;
  if (ef<kMask, kEfAlpha>(true)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] *= ep.alpha;
  }
  if (ef<kMask, kEfBias>(true)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] += bias[e];
  }
  if ((kMask & kEpiGeneric) != 0u && ep.side_fp32 && ep.gelu_grad_of != nullptr) {        // fp32 factor / pre-activation array (the split-bf16 parity arm)
    const float* gp = reinterpret_cast<const float*>(ep.gelu_grad_of) + (ea.gg + static_cast<long>(irow) * ep.ld_gelu_grad);
    const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
    const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] *= ep.act_grad_mode ? gv[e] : gelu_erf_grad(gv[e]);
  } else if (ef<kMask, kEfGeluGrad>(ep.gelu_grad_of != nullptr)) {
    const uint4 uv = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(ep.gelu_grad_of) + (ea.gg + static_cast<long>(irow) * ep.ld_gelu_grad));
    const unsigned w[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x2 gd = f32x2{__uint_as_float(w[e] << 16), __uint_as_float(w[e] & 0xffff0000u)};
      if (!ef<kMask, kEfFactor>(ep.act_grad_mode != 0)) gd = gelu_erf_grad2(gd);        // act_grad_mode: the forward already stored gelu'(z) * keep
      z[2 * e] *= gd[0];
      z[2 * e + 1] *= gd[1];
    }
  }
  float keep[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if (kDrop) {
    const uint64_t pair0 = static_cast<uint64_t>(ea.hash + static_cast<long>(irow) * g.N) >> 1;   // multiple of 4: the 4 pairs share the index's high word
    // inner hash of the index's high word: zero for every output below 2^33 elements, where it is the launch constant drop_key2
    const uint32_t hi = static_cast<uint32_t>(pair0 >> 32), lo = static_cast<uint32_t>(pair0);
    uint32_t key2 = g.drop_key2;
    if (__builtin_amdgcn_ballot_w64(hi != 0u) != 0ull) key2 = mix32(hi ^ g.drop.key);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t hh = g.drop.pair_hash32(lo + e, key2);
      keep[2 * e] = g.drop.lo(hh);
      keep[2 * e + 1] = g.drop.hi(hh);
    }
  }
  const bool has_pre = ef<kMask, kEfPreAct>(ep.pre_act_out != nullptr), is_gelu = ef<kMask, kEfGelu>(ep.act == 1);
  const bool save_factor = ef<kMask, kEfFactor>(ep.act_grad_mode != 0) && has_pre && is_gelu;
  if (save_factor) {                                    // h = gelu(z), saved: gelu'(z) * keep (what the backward multiplies by)
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x2 gv, gd;
      gelu_and_grad2(f32x2{z[2 * e], z[2 * e + 1]}, gv, gd);
      z[2 * e] = gv[0]; z[2 * e + 1] = gv[1];
      f[2 * e] = gd[0] * keep[2 * e]; f[2 * e + 1] = gd[1] * keep[2 * e + 1];
    }
    if ((kMask & kEpiGeneric) != 0u && ep.side_fp32) {
      typedef float f32x4_nt __attribute__((ext_vector_type(4)));
      float* fp = reinterpret_cast<float*>(ep.pre_act_out) + (ea.pre + static_cast<long>(irow) * ep.ld_pre_act);
      __builtin_nontemporal_store(f32x4_nt{f[0], f[1], f[2], f[3]}, reinterpret_cast<f32x4_nt*>(fp));
      __builtin_nontemporal_store(f32x4_nt{f[4], f[5], f[6], f[7]}, reinterpret_cast<f32x4_nt*>(fp + 4));
    } else {
      // the saved factor is not read before the backward pass: a non-temporal store keeps it from displacing h (which the next launch
      // reads) from the Infinity Cache -- the launch itself does not change, the step gains 0.09 ms (same-box A/B, twice)
      typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
      const uint4 pf = pack_bf8(f);
      __builtin_nontemporal_store(u32x4_nt{pf.x, pf.y, pf.z, pf.w}, reinterpret_cast<u32x4_nt*>(reinterpret_cast<unsigned short*>(ep.pre_act_out) + (ea.pre + static_cast<long>(irow) * ep.ld_pre_act)));
    }
  } else {
    if (has_pre && (kMask & kEpiGeneric) != 0u && ep.side_fp32) {        // fp32 pre-activation, kept exactly (nothing is rounded on this arm)
      float* fp = reinterpret_cast<float*>(ep.pre_act_out) + (ea.pre + static_cast<long>(irow) * ep.ld_pre_act);
      *reinterpret_cast<float4*>(fp) = float4{z[0], z[1], z[2], z[3]};
      *reinterpret_cast<float4*>(fp + 4) = float4{z[4], z[5], z[6], z[7]};
    } else if (has_pre) {
      const uint4 o = pack_bf8(z);
      *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(ep.pre_act_out) + (ea.pre + static_cast<long>(irow) * ep.ld_pre_act)) = o;
      const unsigned w[4] = {o.x, o.y, o.z, o.w};       // the activation sees the value the backward will read
#pragma unroll
      for (int e = 0; e < 4; ++e) { z[2 * e] = __uint_as_float(w[e] << 16); z[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    }
    if (is_gelu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 gv = gelu_erf2(f32x2{z[2 * e], z[2 * e + 1]});
        z[2 * e] = gv[0];
        z[2 * e + 1] = gv[1];
      }
    } else if (ef<kMask, kEfRelu>(ep.act == 2)) {
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = fmaxf(z[e], 0.0f);
    }
  }
  const bool drop_late = ef<kMask, kEfDropAfterRes>(ep.drop_after_residual != 0);
  if (kDrop && !drop_late) {
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] *= keep[e];
  }
  if (ef<kMask, kEfResidual>(ep.residual != nullptr)) {
    const float* rp = reinterpret_cast<const float*>(ep.residual) +
                      (ef<kMask, kEfRowMod>(ep.res_row_mod > 0) ? static_cast<long>(row % ep.res_row_mod) * ep.ld_res + col : (ea.res + static_cast<long>(irow) * ep.ld_res));
    const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
    if (ef<kMask, kEfResLn>(ep.res_ln_mean != nullptr)) {
      // the residual is LayerNorm(y), rebuilt from the pre-LayerNorm row: adt_layernorm_fwd's (y - mean) * rstd * gamma + beta
      const float mu = ep.res_ln_mean[row], rs = ep.res_ln_rstd[row];
      const float4 g0 = *reinterpret_cast<const float4*>(ep.res_ln_gamma + col), g1 = *reinterpret_cast<const float4*>(ep.res_ln_gamma + col + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(ep.res_ln_beta + col), b1 = *reinterpret_cast<const float4*>(ep.res_ln_beta + col + 4);
      z[0] += (r0.x - mu) * rs * g0.x + b0.x; z[1] += (r0.y - mu) * rs * g0.y + b0.y; z[2] += (r0.z - mu) * rs * g0.z + b0.z; z[3] += (r0.w - mu) * rs * g0.w + b0.w;
      z[4] += (r1.x - mu) * rs * g1.x + b1.x; z[5] += (r1.y - mu) * rs * g1.y + b1.y; z[6] += (r1.z - mu) * rs * g1.z + b1.z; z[7] += (r1.w - mu) * rs * g1.w + b1.w;
    } else {
      z[0] += r0.x; z[1] += r0.y; z[2] += r0.z; z[3] += r0.w; z[4] += r1.x; z[5] += r1.y; z[6] += r1.z; z[7] += r1.w;
    }
  }
  if (kDrop && drop_late) {
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] *= keep[e];
  }
  const bool fp32_out = ef<kMask, kEfFp32>(ep.out_fp32 != 0), aux = ef<kMask, kEfAux>(ep.aux_bf16_out != nullptr);
  typedef unsigned u32x4_st __attribute__((ext_vector_type(4)));
  typedef float f32x4_st __attribute__((ext_vector_type(4)));
  if (aux || !fp32_out) {
    const uint4 o16 = pack_bf8(z);
    if (aux) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(ep.aux_bf16_out) + (ea.aux + static_cast<long>(irow) * ep.ld_aux)) = o16;
    if (!fp32_out) {
      unsigned short* cp = reinterpret_cast<unsigned short*>(g.C) + (ea.c + static_cast<long>(irow) * g.ldc);
      if (ADT_NT_C_NT) __builtin_nontemporal_store(u32x4_st{o16.x, o16.y, o16.z, o16.w}, reinterpret_cast<u32x4_st*>(cp));
      else *reinterpret_cast<uint4*>(cp) = o16;
    }
  }
  if (fp32_out) {
    float* cp = reinterpret_cast<float*>(g.C) + (ea.c + static_cast<long>(irow) * g.ldc);
    if (ADT_NT_C_NT) {
      __builtin_nontemporal_store(f32x4_st{z[0], z[1], z[2], z[3]}, reinterpret_cast<f32x4_st*>(cp));
      __builtin_nontemporal_store(f32x4_st{z[4], z[5], z[6], z[7]}, reinterpret_cast<f32x4_st*>(cp + 4));
    } else {
      *reinterpret_cast<float4*>(cp) = float4{z[0], z[1], z[2], z[3]};
      *reinterpret_cast<float4*>(cp + 4) = float4{z[4], z[5], z[6], z[7]};
    }
  }
}

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
  const int c8 = (tid & 15) * 8;
  const int col = n0 + c8;
  const bool full = col + 8 <= g.N;                     // N % 8 == 0 is guaranteed for bf16 C (checked on the host)
  float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (g.ep.bias && full) {
    *reinterpret_cast<float4*>(bias) = *reinterpret_cast<const float4*>(g.ep.bias + col);
    *reinterpret_cast<float4*>(bias + 4) = *reinterpret_cast<const float4*>(g.ep.bias + col + 4);
  }
  const EpiAddr ea = epilogue_addr<kDrop, kEpiGeneric>(g, m0 + (tid >> 4), col);
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int lr = pass * 16 + (tid >> 4);
    const int row = m0 + lr;
    if (row >= g.M || !full) continue;
    float z[8];
    *reinterpret_cast<float4*>(z) = *reinterpret_cast<const float4*>(ct + lr * kEpiPitch + c8);
    *reinterpret_cast<float4*>(z + 4) = *reinterpret_cast<const float4*>(ct + lr * kEpiPitch + c8 + 4);
    epilogue_apply8<kDrop>(g, z, bias, ea, m0 + (tid >> 4), pass * 16, col);
  }
}

// =========================================================================================
// NT kernel, LDS-DMA staging (any K that is a multiple of 8).
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

// K is any multiple of 8: the 16-byte pieces of the last K-tile that lie past K come from a zero piece for the A operand (kZeroFill)
// and re-read the row's last valid piece for B -- finite data times zero -- so every K-tile is processed as a full one.
__device__ __attribute__((aligned(16))) unsigned short g_zero_piece[8];
template <bool kZeroFill>
__device__ __forceinline__ void glds_tile(const unsigned short* __restrict__ src, long ld, int row0, int n_rows, int k0, int k_end,
                                          unsigned char* tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + (lane >> 3);
    int gr = row0 + row;
    gr = gr < n_rows ? gr : n_rows - 1;
    const int chunk = (lane & 7) ^ swz_nt(row);
    const int kc = k0 + chunk * 8;
    const unsigned short* p = src + static_cast<long>(gr) * ld + (kc < k_end ? kc : k_end - 8);
    if (kZeroFill && kc >= k_end) p = g_zero_piece;
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
  const int k_tiles = (g.K + kBK - 1) / kBK;       // the last one zero-filled past K (glds_tile)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  glds_tile<true>(g.A, g.lda, m0, g.M, 0, g.K, smem, wave, lane);
  glds_tile<false>(g.B, g.ldb, n0, g.N, 0, g.K, smem + kTileNT, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < k_tiles; ++kt) {
    const unsigned char* ta = smem + (kt & 1) * 2 * kTileNT;
    const unsigned char* tb = ta + kTileNT;
    if (kt + 1 < k_tiles) {
      unsigned char* na = smem + ((kt + 1) & 1) * 2 * kTileNT;
      glds_tile<true>(g.A, g.lda, m0, g.M, (kt + 1) * kBK, g.K, na, wave, lane);
      glds_tile<false>(g.B, g.ldb, n0, g.N, (kt + 1) * kBK, g.K, na + kTileNT, wave, lane);
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

#ifdef ADT_GEMM_RING      // experiment build (make EXTRA=-DADT_GEMM_RING): measured slower than gemm_nt_glds_kernel, see below
// =========================================================================================
// MEASURED AND NOT TAKEN (profiles/r06/gemm_ring4_ab.txt: bit for bit the two-stage kernel on every form, 1.2-1.3x SLOWER on the decoder's
// M = 8192 shapes -- those launches are not latency-bound the way their in-step durations suggested: alone they run at 0.5-0.86 PFLOP/s --
// and +-0 where the output dominates).
// NT kernel for the latency-bound launches (round 6): the 128 x 128 tile of gemm_nt_glds_kernel with K-steps of 32 through a FOUR-slot LDS
// ring, three steps in flight across raw barriers (counted vmcnt, one barrier per step).  The two-stage kernel above waits for a DMA it
// issued one 32-MFMA step earlier: at a few hundred tiles per launch (the decoder's M = 8192 products, the CLAP tower's small stages: one
// or two workgroups per CU, nothing else to run meanwhile) a K-tile costs ~1.5 us of mostly memory latency.  Here a step's operands were
// requested three steps before they are read.
//   slot = [A: 128 rows x 64 B | B: 128 rows x 64 B] = 16 KiB; position p (16 bytes) of row r holds the row's chunk p ^ ((r >> 1) & 3): the 16 rows
//   of a fragment read fall on 8 distinct 16-byte slots of the 128-byte bank line, twice (the two cycles 256 bytes take anyway).
//   DMA: wave w, instruction j fills rows 16 (2w + j) + (lane >> 2) of A and of B (4 instructions per wave and step).
//   step s: s_waitcnt vmcnt(8) (this wave's part of step s has landed; s + 1, s + 2 stay in flight) -> s_barrier (everybody's has, and
//   everybody has read step s - 1) -> DMA of step s + 3 into the slot of step s - 1 -> 8 ds_read_b128 -> 16 MFMAs.
//   Steps past the end re-fetch the last one into slots nobody reads again, so the counts stay exact.  K % 32 == 0.
constexpr int kR4BK = 32;
constexpr int kR4Slot = 2 * 128 * 64;            // 16 KiB
constexpr int kR4Slots = 4;
constexpr int kR4Lds = kR4Slots * kR4Slot > kEpiLds ? kR4Slots * kR4Slot : kEpiLds;      // the epilogue's transposition tile re-uses the ring
template <bool kDrop>
__global__ __launch_bounds__(kGemmThreads) void gemm_nt_ring4_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int m0 = (logical / tiles_n) * kBM, n0 = (logical % tiles_n) * kBN;
  const int k_steps = g.K / kR4BK;

  const unsigned short* pa[2];
  const unsigned short* pb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 16 * (2 * wave + j) + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 1) & 3);
    int ar = m0 + row, br = n0 + row;
    ar = ar < g.M ? ar : g.M - 1;
    br = br < g.N ? br : g.N - 1;
    pa[j] = g.A + static_cast<long>(ar) * g.lda + chunk * 8;
    pb[j] = g.B + static_cast<long>(br) * g.ldb + chunk * 8;
  }
  auto dma = [&](int step) {
    const int st = step < k_steps ? step : k_steps - 1;
    unsigned char* slot = smem + (step & (kR4Slots - 1)) * kR4Slot;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[j] + static_cast<long>(st) * kR4BK),
                                       (__attribute__((address_space(3))) void*)(slot + (2 * wave + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[j] + static_cast<long>(st) * kR4BK),
                                       (__attribute__((address_space(3))) void*)(slot + 8192 + (2 * wave + j) * 1024), 16, 0, 0);
    }
  };
  // fragment addresses inside a slot: row = base + 16 i + (lane & 15), chunk lane >> 4
  unsigned a_ad[4], b_ad[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + (lane & 15), rb = wn * 64 + i * 16 + (lane & 15);
    a_ad[i] = lds_addr(smem) + static_cast<unsigned>(ra * 64 + (((lane >> 4) ^ ((ra >> 1) & 3)) << 4));
    b_ad[i] = lds_addr(smem) + 8192u + static_cast<unsigned>(rb * 64 + (((lane >> 4) ^ ((rb >> 1) & 3)) << 4));
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  dma(0); dma(1); dma(2);
  for (int s = 0; s < k_steps; ++s) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    dma(s + 3);
    const unsigned so = static_cast<unsigned>((s & (kR4Slots - 1)) * kR4Slot);
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ADT_DS_READ_B128_ADDR(fa[i], a_ad[i] + so);
      ADT_DS_READ_B128_ADDR(fb[i], b_ad[i] + so);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the three re-fetches past the end: landed before the epilogue re-uses the ring
  __syncthreads();
  gemm_epilogue_rows<kDrop>(g, acc, reinterpret_cast<float*>(smem), m0, n0, wm, wn, tid, lane);
}
#endif  // ADT_GEMM_RING (ring4)

// =========================================================================================
// NT kernel, 256 x 256 x 64 tile, 8 waves (2 x 4), LDS-DMA staging with the DMA in flight across barriers.
//
// Used for the large forward / dgrad GEMMs (>= 2 blocks per CU, K % 64 == 0).  Each wave owns a 128 x 64 output
// (8 x 4 MFMA blocks, 128 fp32 accumulators) and walks it in four 64 x 32 quadrants per K-tile; one quadrant = one
// "phase" = 16 MFMAs.  LDS (128 KiB) = 2 buffers x 4 half-tiles of 128 rows x 128 B:
//     A_h : rows {wr * 128 + h * 64 + 0..63} of the block tile for wr = 0, 1        (the A rows of quadrant half h)
//     B_h : rows {wc * 64 + h * 32 + 0..31} of the B tile for wc = 0..3             (the B rows of quadrant half h)
// so every half-tile is consumed in exactly one phase by all waves:
//     phase 1: read B_0 (4 x ds_read_b128) + A_0 (8)   MFMA (a0, b0)      DMA issue: A_1 of tile t+1
//     phase 2: read B_1 (4)                            MFMA (a0, b1)      DMA issue: B_0 of tile t+2
//     phase 3: read A_1 (8)                            MFMA (a1, b1)      DMA issue: A_0 of tile t+2
//     phase 4: --                                      MFMA (a1, b0)      DMA issue: B_1 of tile t+2, then s_waitcnt vmcnt(6)
// Three half-tiles (6 DMA instructions per wave) stay in flight across the phase-4 wait, which retires tile t+1 (read
// from phase 1 of the next tile on).  The two wave rows run one barrier apart (wr == 1 takes an extra s_barrier up
// front, wr == 0 one at the end): while one group issues LDS reads / DMA the other one runs its MFMAs on the same SIMDs.
// Hazards under that stagger (G0 = ahead): a DMA-filled half is read >= 1 phase after the counted vmcnt that retires it;
// a half is refilled >= 2 phases after its last ds_read, or 1 phase after when the reads were retired before the reading
// phase's first barrier (B_0: lgkmcnt(8) in phase 1).  All barriers are raw s_barrier (a __syncthreads would drain
// the DMA queue).  Past-the-end tiles re-fetch the last tile into halves nobody reads again, so the counts stay exact.
// Swizzle as in the 128^2 kernel: position p of LDS row r holds the row's 16-byte chunk p ^ ((r >> 1) & 7).
constexpr int kBig = 256;
constexpr int kBigThreads = 512;
constexpr int kHalfTile = 128 * 128;                 // 16 KiB
constexpr int kBigBuf = 4 * kHalfTile;               // A_0 | A_1 | B_0 | B_1
constexpr int kBigStage = 2 * kBigBuf;               // 131,072 B of operand staging
constexpr int kEpi2Bytes = 16 * 64 * 4;              // per wave: 16 x 64 fp32 transposition tile (XOR-swizzled, no padding)
constexpr int kBigLds = kBigStage + 8 * kEpi2Bytes;  // 163,840 B = the whole LDS of a CU

// Work-counter tickets of the persistent kernels (protocol: comment of gemm_nt_256_kernel).
__device__ __forceinline__ void ticket_drawn(unsigned* counter, unsigned ticket, unsigned total) {
  if (ticket + 1u == total) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the launch's last draw: ready for the next launch
}
__device__ __forceinline__ unsigned take_ticket(unsigned* counter, unsigned total) {
  const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket_drawn(counter, t, total);
  return t;
}

#define ADT_DS_READ_B128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))

// Persistent: one workgroup per CU.  The tiles are cut into eight contiguous slices (row-major over the tile grid), slice x
// is worked off by the workgroups with blockIdx.x % 8 == x -- the ones that share an XCD and its L2 -- which take tiles
// from the slice's device work counter: dynamic, so a CU held by another stream's kernel (an RCCL all-reduce under DDP)
// delays no tile.  A block ends on its first ticket >= its slice's size, so a launch draws a known number of tickets from
// each counter (tiles of the slice + one ending ticket per workgroup of the slice: sched_total); whoever draws the LAST one
// puts the counter back to zero -- every other draw has happened by then, and the next launch on the stream starts after this
// one has finished.  The protocol is therefore self-contained: nothing on the host tracks the counters, a failed launch leaves
// them untouched, and tickets are compared unsigned, so a counter that is off for any reason ends workgroups instead of sending
// them to tiles outside the matrix.  The fetch for the next tile is issued by thread 0 before the K loop and handed
// over through the one staging slot the prologue DMAs do not touch (A_1 of buffer 1).
// After a tile's K loop the staging buffers are free, so the NEXT tile's first seven half-tile
// DMAs are issued before this tile's epilogue, which works from a separate 32 KiB of wave-private LDS: the epilogue's
// LDS transposes, activation math and global stores hide the next tile's DMA latency (and there is no workgroup
// turn-around between tiles).
#ifdef ADT_GEMM_EXPERIMENT      // tile timeline of four workgroups of XCD group 0 (tools/probe/gemm_tile_stamps.py): s_memtime at K-loop start / end, epilogue end, tile end
__device__ unsigned long long g_gemm_stamps[4][16][4];
#define ADT_GSTAMP(K)                                                                                   \
  do {                                                                                                  \
    if ((blockIdx.x & 7) == 0 && blockIdx.x < 32 && stamp_tile < 16 && tid == 0)                          \
      g_gemm_stamps[blockIdx.x >> 3][stamp_tile][K] = __builtin_amdgcn_s_memtime();                     \
  } while (0)
#else
#define ADT_GSTAMP(K) do { } while (0)
#endif
template <bool kDrop, bool kColsum, unsigned kMask, bool kX3 = false>
__global__ __launch_bounds__(kBigThreads) void gemm_nt_256_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nwg = tiles_m * tiles_n;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int k_tiles = g.K / kBK;

  // ---- DMA source pointers: wave w, instruction j fills LDS rows 8 * (2w + j) + (lane >> 3) of a half-tile
  const unsigned short* pa[2][2];      // [h][j]
  const unsigned short* pb[2][2];
  const int xg = blockIdx.x & 7;
  const int slice0 = xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8, slice_n = q8 + (xg < r8 ? 1 : 0);
  unsigned* const counter = g.sched + xg * 16;
  const unsigned ctotal = g.sched_total[xg];
  auto set_tile = [&](int v, int& m0, int& n0) {
    const int logical = slice0 + v;
    // column-group-major tile order: groups of group_n tile columns, all row panels of a group before the next group, so
    // that the tiles in flight on an XCD share few B panels AND few A panels (both sets fit its 4 MiB L2)
    const int per_group = tiles_m * g.group_n;
    const int cg = logical / per_group, rem = logical - cg * per_group;
    const int gw = min(g.group_n, tiles_n - cg * g.group_n);
    m0 = (rem / gw) * kBig;
    n0 = (cg * g.group_n + rem % gw) * kBig;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 8 * (2 * wave + j) + (lane >> 3);
      const int chunk = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int ar = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
        ar = ar < g.M ? ar : g.M - 1;
        pa[h][j] = g.A + static_cast<long>(ar) * g.lda + chunk * 8;
        int br = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
        br = br < g.N ? br : g.N - 1;
        pb[h][j] = g.B + static_cast<long>(br) * g.ldb + chunk * 8;
      }
    }
  };
  auto dma = [&](const unsigned short* const (&p)[2], int tile, int buf, int half_slot) {
    const int tt = tile < k_tiles ? tile : k_tiles - 1;
    long koff = static_cast<long>(tt) * kBK;
    if (kX3) {                                             // virtual K-tile -> (segment, tile of the plane): scalar arithmetic
      const int seg = (tt >= g.x3_kt ? 1 : 0) + (tt >= 2 * g.x3_kt ? 1 : 0);
      koff = static_cast<long>(tt - seg * g.x3_kt) * kBK + (half_slot < 2 ? (seg == 0 ? g.x3_a_lo : 0) : (seg == 1 ? g.x3_b_lo : 0));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const auto src = (const __attribute__((address_space(1))) void*)(p[j] + koff);
      const auto dst = (__attribute__((address_space(3))) void*)(smem + buf * kBigBuf + half_slot * kHalfTile + (2 * wave + j) * 1024);
      if (half_slot < 2) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, ADT_NT_A_AUX);      // (half_slot is a literal at every call site)
      else __builtin_amdgcn_global_load_lds(src, dst, 16, 0, ADT_NT_B_AUX);
    }
  };
  auto prologue_dma = [&]() {          // tile 0 complete + B_0 / A_0 / B_1 of tile 1: what the K loop expects to be in flight
    dma(pa[0], 0, 0, 0); dma(pb[0], 0, 0, 2); dma(pa[1], 0, 0, 1); dma(pb[1], 0, 0, 3);
    dma(pb[0], 1, 1, 2); dma(pa[0], 1, 1, 0); dma(pb[1], 1, 1, 3);
  };

  // ---- fragment read addresses (bytes inside a half-tile): row = base + 16 i + (lane & 15), chunk (4 ks + (lane >> 4)) ^ swz
  const unsigned base0 = lds_addr(smem);
  const int sw = ((lane & 15) >> 1) & 7;
  const unsigned c0 = static_cast<unsigned>(((lane >> 4) ^ sw) * 16), c1 = c0 ^ 64u;
  const unsigned a_row = static_cast<unsigned>((wr * 64 + (lane & 15)) * 128), b_row = static_cast<unsigned>((wc * 32 + (lane & 15)) * 128);
  const unsigned a_k0 = base0 + a_row + c0, a_k1 = base0 + a_row + c1;            // + h * kHalfTile + i * 2048 (+ buffer)
  const unsigned b_k0 = base0 + 2 * kHalfTile + b_row + c0, b_k1 = base0 + 2 * kHalfTile + b_row + c1;

  unsigned* const flag = reinterpret_cast<unsigned*>(smem + kBigBuf + kHalfTile);
#ifdef ADT_GEMM_EXPERIMENT
  if (g.stagger > 0 && ((blockIdx.x >> 3) & 1) && tid == 0) {          // experiment: half of an XCD's workgroups half a tile behind the others
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < static_cast<unsigned long long>(g.stagger)) __builtin_amdgcn_s_sleep(16);
  }
#endif
  if (tid == 0) *flag = take_ticket(counter, ctotal);
  __syncthreads();
  int v = static_cast<int>(*flag), m0, n0;
  if (static_cast<unsigned>(v) >= static_cast<unsigned>(slice_n)) return;   // block-uniform (the slice is already handed out)
  __syncthreads();
  set_tile(v, m0, n0);
  prologue_dma();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
#define ADT_MFMA_QUAD(I0, FB, J0)                                                                                  \
  do {                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
          acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], FB[j][ks], acc[I0 + i][J0 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    asm volatile("s_barrier" ::: "memory");                                                                        \
  } while (0)

  float* ct = reinterpret_cast<float*>(smem + kBigStage + wave * kEpi2Bytes);
  unsigned ct_w[4], ct_r[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) ct_w[j] = lds_addr(ct) + static_cast<unsigned>(((4 * (lane >> 4)) * 64 + ((j ^ (lane >> 4)) << 4) + (lane & 15)) * 4);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int lr = pass * 8 + (lane >> 3);
    ct_r[pass] = lds_addr(ct) + static_cast<unsigned>((lr * 64 + ((((lane & 7) >> 1) ^ ((lr >> 2) & 3)) << 4) + (lane & 1) * 8) * 4);
  }

#ifdef ADT_GEMM_EXPERIMENT
  int stamp_tile = 0;
#endif
  while (true) {
    ADT_GSTAMP(0);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // This tile's bias piece and the next tile's index are requested before the K loop and first used after the
    // s_waitcnt vmcnt(0) that ends it.  Inline asm: the compiler would wait for its own loads right here (a loop with
    // VMEM traffic follows), which puts their latency in front of every tile's first MFMA.
    const int ecolL = n0 + wc * 64 + (lane & 7) * 8;
    const bool efullL = ecolL + 8 <= g.N;
    const bool has_bias = g.ep.bias != nullptr && efullL;
    const float* bptr = has_bias ? g.ep.bias + ecolL : reinterpret_cast<const float*>(g.A);     // always a readable 32 bytes
    f32x4 braw0, braw1;
    if constexpr ((kMask & (kEpiGeneric | kEfBias)) != 0u) {   // (an asm load whose result is never read would land in a register the allocator has reused)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(braw0) : "v"(bptr) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(braw1) : "v"(bptr) : "memory");
    }
    unsigned v_next;
    // (address in VGPRs: an SGPR pair the compiler has just re-read from a spill lane would need wait states before a VMEM
    //  instruction, which it cannot know this asm is)
    if (tid == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(v_next) : "v"(counter), "v"(1u) : "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");

    for (int t = 0; t < k_tiles; ++t) {
      const int buf = t & 1;
      const unsigned bo = static_cast<unsigned>(buf) * kBigBuf;
      const unsigned ak0 = a_k0 + bo, ak1 = a_k1 + bo, bk0 = b_k0 + bo, bk1 = b_k1 + bo;
      // ---------------- phase 1: B_0 then A_0; DMA A_1(t+1)
      ADT_DS_READ_B128(fb0[0][0], bk0, 0);    ADT_DS_READ_B128(fb0[0][1], bk1, 0);
      ADT_DS_READ_B128(fb0[1][0], bk0, 2048); ADT_DS_READ_B128(fb0[1][1], bk1, 2048);
      __builtin_amdgcn_sched_barrier(0);
      ADT_DS_READ_B128(fa[0][0], ak0, 0);     ADT_DS_READ_B128(fa[0][1], ak1, 0);
      ADT_DS_READ_B128(fa[1][0], ak0, 2048);  ADT_DS_READ_B128(fa[1][1], ak1, 2048);
      ADT_DS_READ_B128(fa[2][0], ak0, 4096);  ADT_DS_READ_B128(fa[2][1], ak1, 4096);
      ADT_DS_READ_B128(fa[3][0], ak0, 6144);  ADT_DS_READ_B128(fa[3][1], ak1, 6144);
      dma(pa[1], t + 1, buf ^ 1, 1);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");           // B_0 reads retired before the barrier: B_0 may be refilled in phase 2
      asm volatile("s_barrier" ::: "memory");
      ADT_MFMA_QUAD(0, fb0, 0);
      // ---------------- phase 2: B_1; DMA B_0(t+2)
      ADT_DS_READ_B128(fb1[0][0], bk0, kHalfTile);        ADT_DS_READ_B128(fb1[0][1], bk1, kHalfTile);
      ADT_DS_READ_B128(fb1[1][0], bk0, kHalfTile + 2048); ADT_DS_READ_B128(fb1[1][1], bk1, kHalfTile + 2048);
      dma(pb[0], t + 2, buf, 2);
      asm volatile("s_barrier" ::: "memory");
      ADT_MFMA_QUAD(0, fb1, 2);
      // ---------------- phase 3: A_1; DMA A_0(t+2)
      ADT_DS_READ_B128(fa[0][0], ak0, kHalfTile);         ADT_DS_READ_B128(fa[0][1], ak1, kHalfTile);
      ADT_DS_READ_B128(fa[1][0], ak0, kHalfTile + 2048);  ADT_DS_READ_B128(fa[1][1], ak1, kHalfTile + 2048);
      ADT_DS_READ_B128(fa[2][0], ak0, kHalfTile + 4096);  ADT_DS_READ_B128(fa[2][1], ak1, kHalfTile + 4096);
      ADT_DS_READ_B128(fa[3][0], ak0, kHalfTile + 6144);  ADT_DS_READ_B128(fa[3][1], ak1, kHalfTile + 6144);
      dma(pa[0], t + 2, buf, 0);
      asm volatile("s_barrier" ::: "memory");
      ADT_MFMA_QUAD(4, fb1, 2);
      // ---------------- phase 4: DMA B_1(t+2); retire tile t+1
      dma(pb[1], t + 2, buf, 3);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      ADT_MFMA_QUAD(4, fb0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_barrier" ::: "memory");            // every DMA has landed and every fragment read is done: staging is free
    ADT_GSTAMP(1);
    const int em0 = m0, en0 = n0;
    if (tid == 0) { ticket_drawn(counter, v_next, ctotal); *flag = v_next; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    v = static_cast<int>(*flag);
    const bool more = static_cast<unsigned>(v) < static_cast<unsigned>(slice_n);   // block-uniform
    if (more) {
      set_tile(v, m0, n0);
      prologue_dma();                                   // flies under the epilogue below
    }

    float biasL[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr ((kMask & (kEpiGeneric | kEfBias)) != 0u) { biasL[e] = has_bias ? braw0[e] : 0.f; biasL[4 + e] = has_bias ? braw1[e] : 0.f; }
      else { biasL[e] = 0.f; biasL[4 + e] = 0.f; }
    }
    float csL[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // kColsum: this lane's 16 rows of its 8 columns, as stored
    const EpiAddr eaL = epilogue_addr<kDrop, kMask>(g, em0 + wr * 128 + (lane >> 3), ecolL);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("ds_write_b32 %0, %1" :: "v"(ct_w[j]), "v"(acc[i][j][0]));
        asm volatile("ds_write_b32 %0, %1 offset:256" :: "v"(ct_w[j]), "v"(acc[i][j][1]));
        asm volatile("ds_write_b32 %0, %1 offset:512" :: "v"(ct_w[j]), "v"(acc[i][j][2]));
        asm volatile("ds_write_b32 %0, %1 offset:768" :: "v"(ct_w[j]), "v"(acc[i][j][3]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      f32x4 zz[2][2];
      asm volatile("ds_read_b128 %0, %1" : "=v"(zz[0][0]) : "v"(ct_r[0]));
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(zz[0][1]) : "v"(ct_r[0]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(zz[1][0]) : "v"(ct_r[1]));
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(zz[1][1]) : "v"(ct_r[1]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_wave_barrier();                  // every lane's reads are done before the next pass overwrites the tile
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = em0 + wr * 128 + i * 16 + pass * 8 + (lane >> 3);
        float z[8] = {zz[pass][0][0], zz[pass][0][1], zz[pass][0][2], zz[pass][0][3], zz[pass][1][0], zz[pass][1][1], zz[pass][1][2], zz[pass][1][3]};
        if (row < g.M && efullL) {
          epilogue_apply8<kDrop, kMask>(g, z, biasL, eaL, em0 + wr * 128 + (lane >> 3), i * 16 + pass * 8, ecolL);
          if (kColsum) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csL[e] += g.ep.out_fp32 ? z[e] : bf2f(f2bf(z[e]));
          }
        }
      }
    }
    if (kColsum) {                                      // lanes l, l + 8, ..., l + 56 hold the same 8 columns
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        csL[e] += __shfl_xor(csL[e], 8);
        csL[e] += __shfl_xor(csL[e], 16);
        csL[e] += __shfl_xor(csL[e], 32);
      }
      if (lane < 8 && efullL) {
        float* cp = g.colsum_ws + static_cast<long>(em0 / 128 + wr) * g.N + ecolL;
        *reinterpret_cast<float4*>(cp) = float4{csL[0], csL[1], csL[2], csL[3]};
        *reinterpret_cast<float4*>(cp + 4) = float4{csL[4], csL[5], csL[6], csL[7]};
      }
    }
    ADT_GSTAMP(2);
    if (!more) break;
    // The next tile's first k-tiles have landed.  vmcnt retires in issue order and the prologue DMAs are older than everything the
    // epilogue issued, so on an interior tile -- where each of the 16 epilogue_apply8 calls above issued at least one store -- the 16
    // youngest operations are stores: they may still be in flight (they drain under the next tile's first K-step, whose vmcnt(6)
    // retires them).  An edge tile skips stores by predicate, so it waits for everything.
    if (g.tail_stores && em0 + kBig <= g.M && en0 + kBig <= g.N) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    ADT_GSTAMP(3);
#ifdef ADT_GEMM_EXPERIMENT
    ++stamp_tile;
#endif
  }
#undef ADT_MFMA_QUAD
}

#ifdef ADT_GEMM_RING      // experiment build (make EXTRA=-DADT_GEMM_RING): measured slower than gemm_nt_256_kernel on every shape, see the header below
// =========================================================================================
// NT "ring" kernel (round 6): 256 x (64 kWN) tile, 2 x kWN waves, K-steps of 32 through a kStages-slot LDS ring.
//
// MEASURED AND NOT TAKEN (profiles/r06/gemm_ring_ab.txt, gemm_ring_pmc.txt; DESIGN 8.1 of round 6).  Correct -- bit for bit
// gemm_nt_256_kernel on every epilogue form -- and slower everywhere: two workgroups per CU 1.07 x (FFN-1 form) ... 1.5 x (K = 3072), the
// 256 x 256 form with the deeper ring 1.03 ... 1.2 x.  Counters: a ring slot's rows are 64 bytes (32 k), so every 128-byte line is asked
// for as two 64-byte L2 requests in different K-steps (TCP_TCC_READ_REQ 36.2 M against 18.9 M per launch at M 63104, N 768, K 3072), and a
// 256 x 128 tile needs 1.5 x the operand bytes per FLOP (54.4 M requests, L2 misses 10.1 M against 4.9 M): what the second workgroup
// hides of the epilogue is less than what its K loop loses.  Kept compilable (-DADT_GEMM_RING, ADT_GEMM_NT=2wg / ring, tools/exp_gemm_2wg.py)
// so that the measurement can be repeated.
//
//   kWN = 2, kStages = 3: 256 x 128 tile, four waves (one per SIMD), 72 KiB of LDS -> TWO workgroups per CU, so that one's epilogue /
//                         store drain / next-tile prologue runs beside the other's K loop (the second half of the grid starts `stagger`
//                         ticks late, so the two start out of phase).
//   kWN = 4, kStages = 4: 256 x 256 tile, eight waves, one workgroup per CU as gemm_nt_256_kernel, but with up to ~2.5 K-steps
//                         (80 KiB) of operand in flight instead of three half-tiles (48 KiB).
// gemm_nt_256_kernel's K loop is bound by how many operand bytes a CU keeps in flight against a ~1.3 us loaded L2 / Infinity-Cache
// latency (2.3 GB through 12.6 MB in flight); a 256 x 128 tile needs 1.5 x the bytes per FLOP, which is what the two-workgroup
// form pays for its overlap (profiles/r06/gemm_ring_ab.txt).
// Staging: one ring slot = A slab 256 rows x 64 B + B slab (64 kWN) rows x 64 B, filled by global_load_lds_dwordx4 (one wave-instruction
// = 16 rows x 64 B), kStages - 1 steps ahead.  Position p of LDS row r holds the row's 16-byte chunk p ^ f((r >> 2) & 3), f = {0, 3, 2, 1}:
// with ds_read_b128's lane groups ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md, LDS table) the 16 rows x 2 chunks of a group then fall
// on 16 distinct 16-byte bank slots.  One K-step per wave = 12 ds_read_b128 + 32 MFMAs in two halves of 16 (rows 0-63 / 64-127 of the
// 128 x 64 wave tile -- the wave tile and the epilogue are gemm_nt_256_kernel's); the reads of each half are issued before the other
// half's MFMAs (next step's B and low-A fragments go into a second register set), so the matrix pipe does not wait for LDS; ONE
// s_barrier per K-step, between the halves, publishes the step that landed (counted vmcnt: the DMAs of the younger steps stay in flight
// across it).  Between K loops the last slot is free: it holds the epilogue's transposition tiles, and the next tile's first
// kStages - 1 steps are requested into the other slots before the epilogue starts.  Same sums in the same order as gemm_nt_256_kernel:
// the two agree bit for bit (tools/exp_gemm_2wg.py, tests/test_gemm_gpu.py).
constexpr int kRingK = 32;
constexpr int kRingSlabA = 256 * 64;               // 16 KiB
template <int kWN> struct RingShape {
  static constexpr int kTileN = 64 * kWN;
  static constexpr int kWaves = 2 * kWN;
  static constexpr int kThreads = 64 * kWaves;
  static constexpr int kSlabB = kTileN * 64;        // 8 / 16 KiB
  static constexpr int kStage = kRingSlabA + kSlabB;
  static constexpr int kDmaA = 8 / kWN;             // A-slab DMA instructions per wave and step (16 in all)
  static constexpr int kDmaOps = kDmaA + 2;         // + the wave's two B-slab instructions
};
template <int kWN, int kStages> constexpr int ring_lds_bytes() { return kStages * RingShape<kWN>::kStage + 64; }
template <int N> __device__ __forceinline__ void ring_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int kWN, int kStages, bool kDrop, bool kColsum, unsigned kMask>
__global__ __launch_bounds__(RingShape<kWN>::kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_ring_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  using RS = RingShape<kWN>;
  constexpr int kTileN = RS::kTileN, kStage = RS::kStage, kDmaA = RS::kDmaA, kOps = RS::kDmaOps, kDist = kStages - 1;
  constexpr int kEpiOff = (kStages - 1) * kStage, kFlagOff = kStages * kStage;
  static_assert(RS::kWaves * kEpi2Bytes <= kStage, "epilogue tiles fit one ring slot");
  static_assert((kDist - 1) * kOps <= 63, "counted vmcnt fits its field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / kWN, wc = wave % kWN;
  const int nwg = tiles_m * tiles_n;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int k_steps = g.K / kRingK;                // even and >= 8 (the host takes this kernel for K % 64 == 0, K >= 256)

  const unsigned short* pa[kDmaA];                 // wave w, instruction j fills LDS rows (16 kDmaA) w + 16 j + (lane >> 2) of the A slab
  const unsigned short* pb[2];                     //                                  rows 32 w + 16 j + (lane >> 2) of the B slab
  const int xg = blockIdx.x & 7;
  const int slice0 = xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8, slice_n = q8 + (xg < r8 ? 1 : 0);
  unsigned* const counter = g.sched + xg * 16;
  const unsigned ctotal = g.sched_total[xg];
  const int src_chunk = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);      // the chunk of its row this lane's LDS position holds
  auto set_tile = [&](int v, int& m0, int& n0) {
    const int logical = slice0 + v;
    const int per_group = tiles_m * g.group_n;     // column-group-major order, as gemm_nt_256_kernel (group_n counts this kernel's tile columns)
    const int cg = logical / per_group, rem = logical - cg * per_group;
    const int gw = min(g.group_n, tiles_n - cg * g.group_n);
    m0 = (rem / gw) * kBig;
    n0 = (cg * g.group_n + rem % gw) * kTileN;
#pragma unroll
    for (int j = 0; j < kDmaA; ++j) {
      int ar = m0 + 16 * kDmaA * wave + 16 * j + (lane >> 2);
      ar = ar < g.M ? ar : g.M - 1;
      pa[j] = g.A + static_cast<long>(ar) * g.lda + src_chunk * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int br = n0 + 32 * wave + 16 * j + (lane >> 2);
      br = br < g.N ? br : g.N - 1;
      pb[j] = g.B + static_cast<long>(br) * g.ldb + src_chunk * 8;
    }
  };
  auto dma_step = [&](int step, int slot) {
    const long kk = static_cast<long>(step) * kRingK;
    unsigned char* const dst = smem + slot * kStage;
#pragma unroll
    for (int j = 0; j < kDmaA; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[j] + kk),
                                       (__attribute__((address_space(3))) void*)(dst + (kDmaA * wave + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[j] + kk),
                                       (__attribute__((address_space(3))) void*)(dst + kRingSlabA + (2 * wave + j) * 1024), 16, 0, 0);
  };
  auto prologue_dma = [&]() {
#pragma unroll
    for (int d = 0; d < kDist; ++d) dma_step(d, d);
  };

  // fragment read addresses inside a slot: row (16 i + (lane & 15)) of the wave's rows, chunk (lane >> 4) at its swizzled position
  const unsigned base0 = lds_addr(smem);
  const unsigned fchunk = static_cast<unsigned>(((lane >> 4) ^ ((4 - ((lane >> 2) & 3)) & 3)) * 16);
  const unsigned a_fr = base0 + static_cast<unsigned>((wr * 128 + (lane & 15)) * 64) + fchunk;                 // + slot * kStage + i * 1024
  const unsigned b_fr = base0 + kRingSlabA + static_cast<unsigned>((wc * 64 + (lane & 15)) * 64) + fchunk;     // + slot * kStage + j * 1024

  unsigned* const flag = reinterpret_cast<unsigned*>(smem + kFlagOff);
  if (kWN == 2 && g.stagger > 0 && blockIdx.x >= (gridDim.x >> 1) && tid == 0) {   // the CU's second workgroup starts about half a tile behind the first
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < static_cast<unsigned long long>(g.stagger)) __builtin_amdgcn_s_sleep(16);
  }
  if (tid == 0) *flag = take_ticket(counter, ctotal);
  __syncthreads();
  int v = static_cast<int>(*flag), m0, n0;
  if (static_cast<unsigned>(v) >= static_cast<unsigned>(slice_n)) return;   // block-uniform
  __syncthreads();
  set_tile(v, m0, n0);
  prologue_dma();

  float* ct = reinterpret_cast<float*>(smem + kEpiOff + wave * kEpi2Bytes);
  unsigned ct_w[4], ct_r[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) ct_w[j] = lds_addr(ct) + static_cast<unsigned>(((4 * (lane >> 4)) * 64 + ((j ^ (lane >> 4)) << 4) + (lane & 15)) * 4);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int lr = pass * 8 + (lane >> 3);
    ct_r[pass] = lds_addr(ct) + static_cast<unsigned>((lr * 64 + ((((lane & 7) >> 1) ^ ((lr >> 2) & 3)) << 4) + (lane & 1) * 8) * 4);
  }

  bf16x8 fal[4], fah[4], fb[4], fb2[4];
#define ADT_RING_READ_B(FB, ADDR)                                                                        \
  do { ADT_DS_READ_B128(FB[0], ADDR, 0); ADT_DS_READ_B128(FB[1], ADDR, 1024); ADT_DS_READ_B128(FB[2], ADDR, 2048); ADT_DS_READ_B128(FB[3], ADDR, 3072); } while (0)
#define ADT_RING_READ_A(FA, ADDR, OFF)                                                                   \
  do { ADT_DS_READ_B128(FA[0], ADDR, OFF); ADT_DS_READ_B128(FA[1], ADDR, (OFF) + 1024); ADT_DS_READ_B128(FA[2], ADDR, (OFF) + 2048); ADT_DS_READ_B128(FA[3], ADDR, (OFF) + 3072); } while (0)
#define ADT_RING_MFMA(I0, FA, FB)                                                                        \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
        acc[I0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[i], FB[j], acc[I0 + i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
  // One K-step S whose B / low-A fragments are in FB / fal: high-A reads and the DMA of step S + kDist in front of the low half's MFMAs;
  // then the barrier that publishes step S + 1 (every step younger than it may still be in flight: `younger` of them were issued),
  // its B / low-A reads (into FBN / fal) in front of the high half's MFMAs.
#define ADT_RING_STEP(S, FB, FBN)                                                                        \
  do {                                                                                                   \
    const unsigned so = static_cast<unsigned>(slot) * kStage;                                            \
    const int slot1 = slot == kStages - 1 ? 0 : slot + 1, slotd = slot == 0 ? kStages - 1 : slot - 1;    \
    ADT_RING_READ_A(fah, a_fr + so, 4096);                                                               \
    if ((S) + kDist < k_steps) dma_step((S) + kDist, slotd);                                             \
    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                                   \
    ADT_RING_MFMA(0, fal, FB);                                                                           \
    if ((S) + 1 < k_steps) {                                                                             \
      const int younger = min(kDist - 1, k_steps - 2 - (S));                                             \
      if (younger >= 3) ring_wait_vmcnt<3 * kOps>();                                                     \
      else if (younger == 2) ring_wait_vmcnt<2 * kOps>();                                                \
      else if (younger == 1) ring_wait_vmcnt<kOps>();                                                    \
      else ring_wait_vmcnt<0>();                                                                         \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
      asm volatile("s_barrier" ::: "memory");                                                            \
      const unsigned sn = static_cast<unsigned>(slot1) * kStage;                                         \
      ADT_RING_READ_B(FBN, b_fr + sn);                                                                   \
      ADT_RING_READ_A(fal, a_fr + sn, 0);                                                                \
    } else {                                                                                             \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
    }                                                                                                    \
    ADT_RING_MFMA(4, fah, FB);                                                                           \
    slot = slot1;                                                                                        \
  } while (0)
  static_assert(kDist - 1 <= 3, "ADT_RING_STEP's counted waits cover up to three younger steps");

#ifdef ADT_GEMM_EXPERIMENT
  int stamp_tile = 0;
#endif
  while (true) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this tile's bias piece and the next tile's ticket: requested here, first used after the K loop (inline asm, as in gemm_nt_256_kernel)
    const int ecolL = n0 + wc * 64 + (lane & 7) * 8;
    const bool efullL = ecolL + 8 <= g.N;
    const bool has_bias = g.ep.bias != nullptr && efullL;
    const float* bptr = has_bias ? g.ep.bias + ecolL : reinterpret_cast<const float*>(g.A);
    f32x4 braw0, braw1;
    if constexpr ((kMask & (kEpiGeneric | kEfBias)) != 0u) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(braw0) : "v"(bptr) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(braw1) : "v"(bptr) : "memory");
    }
    unsigned v_next;
    if (tid == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(v_next) : "v"(counter), "v"(1u) : "memory");
    // the first steps were requested before the previous tile's epilogue: everything older than the three operations above has to be done
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");          // ... for every wave; and every wave has left the previous epilogue (the last slot is a stage again)
    ADT_GSTAMP(0);
    int slot = 0;
    ADT_RING_READ_B(fb, b_fr);
    ADT_RING_READ_A(fal, a_fr, 0);
    for (int s = 0; s < k_steps; s += 2) {
      ADT_RING_STEP(s, fb, fb2);
      ADT_RING_STEP(s + 1, fb2, fb);
    }
    // every DMA has landed (vmcnt(0) at step k_steps - 2) and this wave's fragment reads are done
    if (tid == 0) { ticket_drawn(counter, v_next, ctotal); *flag = v_next; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");            // all waves: the ring is free, the flag is visible
    ADT_GSTAMP(1);
    const int em0 = m0;
    v = static_cast<int>(*flag);
    const bool more = static_cast<unsigned>(v) < static_cast<unsigned>(slice_n);   // block-uniform
    if (more) {
      set_tile(v, m0, n0);
      prologue_dma();                                  // flies under the epilogue below
    }

    float biasL[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr ((kMask & (kEpiGeneric | kEfBias)) != 0u) { biasL[e] = has_bias ? braw0[e] : 0.f; biasL[4 + e] = has_bias ? braw1[e] : 0.f; }
      else { biasL[e] = 0.f; biasL[4 + e] = 0.f; }
    }
    float csL[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const EpiAddr eaL = epilogue_addr<kDrop, kMask>(g, em0 + wr * 128 + (lane >> 3), ecolL);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("ds_write_b32 %0, %1" :: "v"(ct_w[j]), "v"(acc[i][j][0]));
        asm volatile("ds_write_b32 %0, %1 offset:256" :: "v"(ct_w[j]), "v"(acc[i][j][1]));
        asm volatile("ds_write_b32 %0, %1 offset:512" :: "v"(ct_w[j]), "v"(acc[i][j][2]));
        asm volatile("ds_write_b32 %0, %1 offset:768" :: "v"(ct_w[j]), "v"(acc[i][j][3]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      f32x4 zz[2][2];
      asm volatile("ds_read_b128 %0, %1" : "=v"(zz[0][0]) : "v"(ct_r[0]));
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(zz[0][1]) : "v"(ct_r[0]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(zz[1][0]) : "v"(ct_r[1]));
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(zz[1][1]) : "v"(ct_r[1]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = em0 + wr * 128 + i * 16 + pass * 8 + (lane >> 3);
        float z[8] = {zz[pass][0][0], zz[pass][0][1], zz[pass][0][2], zz[pass][0][3], zz[pass][1][0], zz[pass][1][1], zz[pass][1][2], zz[pass][1][3]};
        if (row < g.M && efullL) {
          epilogue_apply8<kDrop, kMask>(g, z, biasL, eaL, em0 + wr * 128 + (lane >> 3), i * 16 + pass * 8, ecolL);
          if (kColsum) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csL[e] += g.ep.out_fp32 ? z[e] : bf2f(f2bf(z[e]));
          }
        }
      }
    }
    if (kColsum) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        csL[e] += __shfl_xor(csL[e], 8);
        csL[e] += __shfl_xor(csL[e], 16);
        csL[e] += __shfl_xor(csL[e], 32);
      }
      if (lane < 8 && efullL) {
        float* cp = g.colsum_ws + static_cast<long>(em0 / 128 + wr) * g.N + ecolL;
        *reinterpret_cast<float4*>(cp) = float4{csL[0], csL[1], csL[2], csL[3]};
        *reinterpret_cast<float4*>(cp + 4) = float4{csL[4], csL[5], csL[6], csL[7]};
      }
    }
    ADT_GSTAMP(2);
    if (!more) break;
#ifdef ADT_GEMM_EXPERIMENT
    ++stamp_tile;
#endif
  }
#undef ADT_RING_STEP
#undef ADT_RING_MFMA
#undef ADT_RING_READ_A
#undef ADT_RING_READ_B
}

#endif  // ADT_GEMM_RING

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

// One 128 x 128 output tile (rows m0.., columns n0..) of K-split `split`.
__device__ __forceinline__ void gemm_tn_glds_tile(const GemmArgs& g, int m0, int n0, int split, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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
// =========================================================================================
// Skinny NT kernel (M <= 64): the projections of the KV-cached decode step (network.py: greedy_decode_cached -- 25 GEMMs of B rows
// per decoded token; B = the 10 s windows of the audio being transcribed).  The 128^2 tile kernel gives such a product ONE row of
// workgroups that walk K serially (10-30 us per launch); here a workgroup owns 16 output columns, its four waves each take a
// quarter of K, eight 32-deep steps of operand loads are issued before their MFMAs (one memory round trip per launch at K = 768),
// the W fragments are shared by the kMT 16-row tiles of the activations, and the four partial tiles are summed through LDS in a
// fixed order.  The epilogue is the generic one (bias / GELU / residual / dropout / fp32 or bf16 output), one 8-column piece per lane.
constexpr int kSkinnyThreads = 256, kSkinnyDepth = 8;
template <bool kDrop, int kMT>
__global__ __launch_bounds__(kSkinnyThreads) void gemm_nt_skinny_kernel(GemmArgs g) {
  __shared__ float part[4][16 * kMT][17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * 16;
  const int kq = g.K >> 2;                                     // K % 128 == 0: a whole number of 32-deep steps per quarter
  int brow = n0 + (lane & 15);
  brow = brow < g.N ? brow : g.N - 1;                          // rows >= M / columns >= N are computed from clamped operands, never stored
  const unsigned short* pb = g.B + static_cast<long>(brow) * g.ldb + wave * kq + 8 * (lane >> 4);
  const unsigned short* pa[kMT];
#pragma unroll
  for (int t = 0; t < kMT; ++t) {
    int arow = 16 * t + (lane & 15);
    arow = arow < g.M ? arow : g.M - 1;
    pa[t] = g.A + static_cast<long>(arow) * g.lda + wave * kq + 8 * (lane >> 4);
  }
  f32x4 acc[kMT];
#pragma unroll
  for (int t = 0; t < kMT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < kq; k += 32 * kSkinnyDepth) {
    bf16x8 fb[kSkinnyDepth], fa[kMT][kSkinnyDepth];
#pragma unroll
    for (int i = 0; i < kSkinnyDepth; ++i)
      if (k + 32 * i < kq) {
        fb[i] = *reinterpret_cast<const bf16x8*>(pb + k + 32 * i);
#pragma unroll
        for (int t = 0; t < kMT; ++t) fa[t][i] = *reinterpret_cast<const bf16x8*>(pa[t] + k + 32 * i);
      }
#pragma unroll
    for (int i = 0; i < kSkinnyDepth; ++i)
      if (k + 32 * i < kq) {
#pragma unroll
        for (int t = 0; t < kMT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t][i], fb[i], acc[t], 0, 0, 0);
      }
  }
#pragma unroll
  for (int t = 0; t < kMT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wave][16 * t + 4 * (lane >> 4) + j][lane & 15] = acc[t][j];      // D[row 4 (l >> 4) + j][col l & 15]
  __syncthreads();
  for (int p = tid; p < 32 * kMT; p += kSkinnyThreads) {
    const int r = p >> 1, col = n0 + 8 * (p & 1);
    if (r < g.M && col + 8 <= g.N) {
      float z[8], bias[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 8 * (p & 1) + e;
        z[e] = (part[0][r][c] + part[1][r][c]) + (part[2][r][c] + part[3][r][c]);
        bias[e] = g.ep.bias ? g.ep.bias[col + e] : 0.f;
      }
      const EpiAddr ea = epilogue_addr<kDrop, kEpiGeneric>(g, r, col);
      epilogue_apply8<kDrop, kEpiGeneric>(g, z, bias, ea, r, 0, col);
    }
  }
}
// The same kernel with a LayerNorm in front of its A operand: C = LN(y) W^T (+ epilogue), y fp32 [M, K].  In the decode step every
// LayerNorm output has exactly one GEMM that consumes it as an operand and one later GEMM that adds it as the residual, so the
// twelve LayerNorm launches of a step fold into their consumers: every workgroup recomputes the row statistics of its (at most 64)
// rows -- 8 x 3 KB from L2 -- normalises its A fragments on the fly, and writes its share of the columns of x32 = LN(y) for the
// later residual.  Arithmetic of adt_layernorm_fwd: two-pass statistics in fp32, (y - mean) * rstd * gamma + beta, bf16 operand.
struct LnPrologue { const float* y; long ldy; const float* gamma; const float* beta; float eps; float* x32; long ldx; };
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float skinny_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, kRowMask, 0xf, false));
}
__device__ __forceinline__ float skinny_wave_sum(float v) {   // DPP network: row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31; lane 63 holds the total
  v = skinny_dpp_add<0x111, 0xf>(v); v = skinny_dpp_add<0x112, 0xf>(v); v = skinny_dpp_add<0x114, 0xf>(v); v = skinny_dpp_add<0x118, 0xf>(v);
  v = skinny_dpp_add<0x142, 0xa>(v); v = skinny_dpp_add<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <bool kDrop, int kMT>
__global__ __launch_bounds__(kSkinnyThreads) void gemm_nt_skinny_ln_kernel(GemmArgs g, LnPrologue p) {
  __shared__ float part[4][16 * kMT][17];
  __shared__ float mean_s[16 * kMT], rstd_s[16 * kMT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.x * 16;
  const int kq = g.K >> 2;
  // ---- row statistics: wave w takes rows w, w + 4, ... two at a time; a lane holds K / 64 <= 16 elements of a row in registers, so a
  // row costs one memory round trip (shared by the pair) and two DPP reductions
  for (int r0 = wave; r0 < g.M; r0 += 8) {
    float v[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = r0 + 4 * h < g.M ? r0 + 4 * h : r0;
      const float* yr = p.y + static_cast<long>(r) * p.ldy;
#pragma unroll
      for (int i = 0; i < 16; ++i) v[h][i] = lane + 64 * i < g.K ? yr[lane + 64 * i] : 0.f;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += v[h][i];
      const float mean = skinny_wave_sum(s) / g.K;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float d = lane + 64 * i < g.K ? v[h][i] - mean : 0.f; ss = fmaf(d, d, ss); }
      const float rstd = rsqrtf(skinny_wave_sum(ss) / g.K + p.eps);
      if (lane == 0 && r0 + 4 * h < g.M) { mean_s[r0 + 4 * h] = mean; rstd_s[r0 + 4 * h] = rstd; }
    }
  }
  __syncthreads();
  // ---- this workgroup's columns of x32 = LN(y) (the residual of a later GEMM)
  if (p.x32) {
    const int cpw = (g.K + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
    const int c0 = static_cast<int>(blockIdx.x) * cpw, c1 = c0 + cpw < g.K ? c0 + cpw : g.K;
    for (int i = tid; i < g.M * cpw; i += kSkinnyThreads) {
      const int r = i / cpw, c = c0 + i - r * cpw;
      if (c < c1) p.x32[static_cast<long>(r) * p.ldx + c] = (p.y[static_cast<long>(r) * p.ldy + c] - mean_s[r]) * rstd_s[r] * p.gamma[c] + p.beta[c];
    }
  }
  int brow = n0 + (lane & 15);
  brow = brow < g.N ? brow : g.N - 1;
  const unsigned short* pb = g.B + static_cast<long>(brow) * g.ldb + wave * kq + 8 * (lane >> 4);
  const int kl = wave * kq + 8 * (lane >> 4);                 // this lane's first k of a step
  const float* py[kMT];
  float mu[kMT], rs[kMT];
#pragma unroll
  for (int t = 0; t < kMT; ++t) {
    int arow = 16 * t + (lane & 15);
    arow = arow < g.M ? arow : g.M - 1;
    py[t] = p.y + static_cast<long>(arow) * p.ldy + kl;
    mu[t] = mean_s[arow]; rs[t] = rstd_s[arow];
  }
  f32x4 acc[kMT];
#pragma unroll
  for (int t = 0; t < kMT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int kDepth = 4;                                    // fp32 operands: twice the registers per step of the bf16 kernel
  for (int k = 0; k < kq; k += 32 * kDepth) {
    bf16x8 fb[kDepth];
    float4 ya[kMT][kDepth][2], ga[kDepth][2], be[kDepth][2];
#pragma unroll
    for (int i = 0; i < kDepth; ++i)
      if (k + 32 * i < kq) {
        fb[i] = *reinterpret_cast<const bf16x8*>(pb + k + 32 * i);
        ga[i][0] = *reinterpret_cast<const float4*>(p.gamma + kl + k + 32 * i); ga[i][1] = *reinterpret_cast<const float4*>(p.gamma + kl + k + 32 * i + 4);
        be[i][0] = *reinterpret_cast<const float4*>(p.beta + kl + k + 32 * i); be[i][1] = *reinterpret_cast<const float4*>(p.beta + kl + k + 32 * i + 4);
#pragma unroll
        for (int t = 0; t < kMT; ++t) {
          ya[t][i][0] = *reinterpret_cast<const float4*>(py[t] + k + 32 * i);
          ya[t][i][1] = *reinterpret_cast<const float4*>(py[t] + k + 32 * i + 4);
        }
      }
#pragma unroll
    for (int i = 0; i < kDepth; ++i)
      if (k + 32 * i < kq) {
#pragma unroll
        for (int t = 0; t < kMT; ++t) {
          union { bf16x8 v; unsigned u[4]; } fa;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 y4 = ya[t][i][h], g4 = ga[i][h], b4 = be[i][h];
            fa.u[2 * h] = pack_bf2((y4.x - mu[t]) * rs[t] * g4.x + b4.x, (y4.y - mu[t]) * rs[t] * g4.y + b4.y);
            fa.u[2 * h + 1] = pack_bf2((y4.z - mu[t]) * rs[t] * g4.z + b4.z, (y4.w - mu[t]) * rs[t] * g4.w + b4.w);
          }
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb[i], acc[t], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int t = 0; t < kMT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wave][16 * t + 4 * (lane >> 4) + j][lane & 15] = acc[t][j];
  __syncthreads();
  for (int q = tid; q < 32 * kMT; q += kSkinnyThreads) {
    const int r = q >> 1, col = n0 + 8 * (q & 1);
    if (r < g.M && col + 8 <= g.N) {
      float z[8], bias[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 8 * (q & 1) + e;
        z[e] = (part[0][r][c] + part[1][r][c]) + (part[2][r][c] + part[3][r][c]);
        bias[e] = g.ep.bias ? g.ep.bias[col + e] : 0.f;
      }
      const EpiAddr ea = epilogue_addr<kDrop, kEpiGeneric>(g, r, col);
      epilogue_apply8<kDrop, kEpiGeneric>(g, z, bias, ea, r, 0, col);
    }
  }
}
template <bool kDrop>
static void launch_skinny(const GemmArgs& g, hipStream_t st) {
  const dim3 gs(static_cast<unsigned>((g.N + 15) / 16));
  if (g.M <= 16) hipLaunchKernelGGL((gemm_nt_skinny_kernel<kDrop, 1>), gs, dim3(kSkinnyThreads), 0, st, g);
  else if (g.M <= 32) hipLaunchKernelGGL((gemm_nt_skinny_kernel<kDrop, 2>), gs, dim3(kSkinnyThreads), 0, st, g);
  else hipLaunchKernelGGL((gemm_nt_skinny_kernel<kDrop, 4>), gs, dim3(kSkinnyThreads), 0, st, g);
}

// XCD-aware renumbering of a 1-D grid (workgroups b and b + 8 share an XCD): every XCD gets a contiguous range of logical tiles
__device__ __forceinline__ int xcd_logical(int nwg, int bid) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__global__ __launch_bounds__(kGemmThreads) void gemm_tn_glds_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 stages][A tile | B tile], reused by the epilogue
  const int logical = xcd_logical(tiles_m * tiles_n, blockIdx.x);
  gemm_tn_glds_tile(g, (logical / tiles_n) * kBM, (logical % tiles_n) * kBN, blockIdx.y, smem);
}

// Grouped form: several independent weight-gradient products C_i[M_i, N_i] = A_i[K_i, M_i]^T B_i[K_i, N_i] (fp32 out) in ONE
// launch.  The decoder's weight gradients are 25 products with K = B * T = 8192 and 36-144 tiles each: alone, each needs a K
// split over fp32 slabs plus a reduction launch to fill 256 CUs; together their ~2000 tiles fill the chip with whole-K tiles.
constexpr int kMaxGroup = 32;
struct TnGroupArgs {
  int n;
  int tile_start[kMaxGroup + 1];     // first logical tile of item i; tile_start[n] = total
  int tiles_n[kMaxGroup];
  int M[kMaxGroup], N[kMaxGroup], K[kMaxGroup];
  const unsigned short* A[kMaxGroup]; const unsigned short* B[kMaxGroup]; float* C[kMaxGroup];
  long lda[kMaxGroup], ldb[kMaxGroup], ldc[kMaxGroup];
};
__global__ __launch_bounds__(kGemmThreads) void gemm_tn_grouped_kernel(TnGroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int logical = xcd_logical(ga.tile_start[ga.n], blockIdx.x);
  int it = 0;
  while (it + 1 < ga.n && logical >= ga.tile_start[it + 1]) ++it;      // block-uniform
  const int local = logical - ga.tile_start[it];
  GemmArgs g;
  g.A = ga.A[it]; g.lda = ga.lda[it];
  g.B = ga.B[it]; g.ldb = ga.ldb[it];
  g.C = ga.C[it]; g.ldc = ga.ldc[it];
  g.M = ga.M[it]; g.N = ga.N[it]; g.K = ga.K[it];
  g.k_tiles_per_split = g.K / kBK;
  g.slabs = nullptr; g.colsum_ws = nullptr;
  g.ep = adt_gemm_epilogue{};
  g.ep.alpha = 1.0f;
  g.ep.out_fp32 = 1;
  g.drop = Drop{0u, 0u, 1.0f}; g.drop_key2 = 0u;
  g.sched = nullptr;
  gemm_tn_glds_tile(g, (local / ga.tiles_n[it]) * kBM, (local % ga.tiles_n[it]) * kBN, 0, smem);
}

// =========================================================================================
// TN kernel (weight gradients), persistent 256 x 256 x 64 tiles: the schedule of gemm_nt_256_kernel (8 waves, 4 phases of
// 16 MFMAs per K-tile, three half-tile DMAs in flight across raw barriers, staggered wave rows, per-XCD-slice work counters,
// next item's prologue under the epilogue) with the TN operand path: half-tiles are [64 k][128 columns] images (256-byte
// rows, chunk p of row r at position p ^ (2 * (r & 7))) of
//     A_h : columns {wr * 128 + h * 64 + 0..63} of the A tile for wr = 0, 1
//     B_h : columns {wc * 64 + h * 32 + 0..31} of the B tile for wc = 0..3
// read as MFMA fragments by pairs of ds_read_b64_tr_b16.  A work item is (tile, K split); splits write fp32 slabs that
// reduce_slabs_kernel sums in slab order.
#define ADT_TR_PAIR(lo, hi, addr, imm)                                                                  \
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"            \
               : "=&v"(lo), "=&v"(hi) : "v"(addr), "i"(imm), "i"((imm) + 4096))

template <bool kX3>
__global__ __launch_bounds__(kBigThreads) void gemm_tn_256_kernel(GemmArgs g, int tiles_m, int tiles_n, int splits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int nwg = tiles_m * tiles_n * splits;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int k_tiles_all = g.K / kBK;

  const unsigned short* pa[2][2];      // [h][j]: wave w, instruction j fills LDS rows 4 * (2w + j) + (lane >> 4) of a half-tile
  const unsigned short* pb[2][2];
  const int xg = blockIdx.x & 7;
  const int slice0 = xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8, slice_n = q8 + (xg < r8 ? 1 : 0);
  unsigned* const counter = g.sched + xg * 16;
  const unsigned ctotal = g.sched_total[xg];
  int k_tiles = 0;                     // K-tiles of the current item
  int kt_first = 0;                    // kX3: the item's first VIRTUAL K-tile (the pointers then carry no K offset; dma() maps every tile)
  auto set_item = [&](int v, int& m0, int& n0, int& split) {
    const int logical = slice0 + v;              // split-major: an XCD slice holds neighbouring tiles of ONE K range (they share
    const int n_tiles = tiles_m * tiles_n;       // A / B panels through that XCD's L2)
    split = logical / n_tiles;
    const int tile = logical - split * n_tiles;
    m0 = (tile / tiles_n) * kBig;
    n0 = (tile % tiles_n) * kBig;
    const int kt0 = split * g.k_tiles_per_split;
    int kt1 = kt0 + g.k_tiles_per_split;
    kt1 = kt1 < k_tiles_all ? kt1 : k_tiles_all;
    k_tiles = kt1 - kt0;               // >= 1: the host sizes the splits so that the last one is not empty
    kt_first = kt0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 4 * (2 * wave + j) + (lane >> 4);
      const int c = (((lane & 15) ^ ((r & 7) << 1))) * 8;                 // first LDS column of this lane's 16-byte chunk
      const long krow = (kX3 ? 0l : static_cast<long>(kt0) * kBK) + r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int ac = m0 + (c >> 6) * 128 + h * 64 + (c & 63);
        ac = ac + 8 <= g.M ? ac : g.M - 8;
        pa[h][j] = g.A + krow * g.lda + ac;
        int bc = n0 + (c >> 5) * 64 + h * 32 + (c & 31);
        bc = bc + 8 <= g.N ? bc : g.N - 8;
        pb[h][j] = g.B + krow * g.ldb + bc;
      }
    }
  };
  auto dma = [&](const unsigned short* const (&p)[2], long ld, int tile, int buf, int half_slot) {
    const int tt = tile < k_tiles ? tile : k_tiles - 1;
    long off = static_cast<long>(tt) * kBK * ld;
    if (kX3) {                                             // virtual K-tile -> (segment, row tile of the planes): lo x hi, hi x lo, hi x hi
      const int vt = kt_first + tt;
      const int seg = (vt >= g.x3_kt ? 1 : 0) + (vt >= 2 * g.x3_kt ? 1 : 0);
      off = static_cast<long>(vt - seg * g.x3_kt) * kBK * ld + (half_slot < 2 ? (seg == 0 ? g.x3_a_lo : 0) : (seg == 1 ? g.x3_b_lo : 0));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p[j] + off),
                                       (__attribute__((address_space(3))) void*)(smem + buf * kBigBuf + half_slot * kHalfTile + (2 * wave + j) * 1024),
                                       16, 0, 0);
  };
  auto prologue_dma = [&]() {
    dma(pa[0], g.lda, 0, 0, 0); dma(pb[0], g.ldb, 0, 0, 2); dma(pa[1], g.lda, 0, 0, 1); dma(pb[1], g.ldb, 0, 0, 3);
    dma(pb[0], g.ldb, 1, 1, 2); dma(pa[0], g.lda, 1, 1, 0); dma(pb[1], g.ldb, 1, 1, 3);
  };

  // ---- transposed fragment reads: 16 columns x 32 k per pair; lane (t = lane & 15, q = lane >> 4) reads row ks*32 + 4q + (t >> 2)
  // (+16 for the second half), 8 bytes at column chunk ((col0 + 4 (t & 3)) >> 3) ^ swz
  const unsigned base0 = lds_addr(smem);
  const int tq = lane & 15, gq = lane >> 4;
  const int frow = 4 * gq + (tq >> 2);
  const unsigned fsw = static_cast<unsigned>((frow & 7) << 1);
  unsigned a_ad[4], b_ad[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    a_ad[i] = base0 + static_cast<unsigned>(frow * 256) + ((static_cast<unsigned>(wr * 8 + 2 * i + ((tq & 3) >> 1)) ^ fsw) << 4) + 8u * (tq & 1);
#pragma unroll
  for (int n = 0; n < 2; ++n)
    b_ad[n] = base0 + 2 * kHalfTile + static_cast<unsigned>(frow * 256) + ((static_cast<unsigned>(wc * 4 + 2 * n + ((tq & 3) >> 1)) ^ fsw) << 4) + 8u * (tq & 1);
  float* ct = reinterpret_cast<float*>(smem + kBigStage + wave * kEpi2Bytes);
  unsigned ct_w[4], ct_r[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) ct_w[j] = lds_addr(ct) + static_cast<unsigned>(((4 * (lane >> 4)) * 64 + ((j ^ (lane >> 4)) << 4) + (lane & 15)) * 4);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int lr = pass * 8 + (lane >> 3);
    ct_r[pass] = lds_addr(ct) + static_cast<unsigned>((lr * 64 + ((((lane & 7) >> 1) ^ ((lr >> 2) & 3)) << 4) + (lane & 1) * 8) * 4);
  }

  unsigned* const flag = reinterpret_cast<unsigned*>(smem + kBigBuf + kHalfTile);
  if (tid == 0) *flag = take_ticket(counter, ctotal);
  __syncthreads();
  int v = static_cast<int>(*flag), m0, n0, split;
  if (static_cast<unsigned>(v) >= static_cast<unsigned>(slice_n)) return;
  __syncthreads();
  set_item(v, m0, n0, split);
  prologue_dma();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");

  bf16x4 alo[4][2], ahi[4][2], b0lo[2][2], b0hi[2][2], b1lo[2][2], b1hi[2][2];
#define ADT_TN_QUAD(I0, BLO, BHI, J0)                                                                              \
  do {                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
          acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(alo[i][ks], ahi[i][ks]), join8(BLO[j][ks], BHI[j][ks]), \
                                                                        acc[I0 + i][J0 + j], 0, 0, 0);            \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    asm volatile("s_barrier" ::: "memory");                                                                        \
  } while (0)

  while (true) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned v_next;
    if (tid == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(v_next) : "v"(counter), "v"(1u) : "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");

    for (int t = 0; t < k_tiles; ++t) {
      const int buf = t & 1;
      const unsigned bo = static_cast<unsigned>(buf) * kBigBuf;
      const unsigned a0 = a_ad[0] + bo, a1 = a_ad[1] + bo, a2 = a_ad[2] + bo, a3 = a_ad[3] + bo, bb0 = b_ad[0] + bo, bb1 = b_ad[1] + bo;
      // ---------------- phase 1: B_0, A_0; DMA A_1(t+1)
      ADT_TR_PAIR(b0lo[0][0], b0hi[0][0], bb0, 0);    ADT_TR_PAIR(b0lo[0][1], b0hi[0][1], bb0, 8192);
      ADT_TR_PAIR(b0lo[1][0], b0hi[1][0], bb1, 0);    ADT_TR_PAIR(b0lo[1][1], b0hi[1][1], bb1, 8192);
      __builtin_amdgcn_sched_barrier(0);
      ADT_TR_PAIR(alo[0][0], ahi[0][0], a0, 0);       ADT_TR_PAIR(alo[0][1], ahi[0][1], a0, 8192);
      ADT_TR_PAIR(alo[1][0], ahi[1][0], a1, 0);       ADT_TR_PAIR(alo[1][1], ahi[1][1], a1, 8192);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");           // the 8 B_0 reads are retired: B_0 may be refilled in phase 2
      ADT_TR_PAIR(alo[2][0], ahi[2][0], a2, 0);       ADT_TR_PAIR(alo[2][1], ahi[2][1], a2, 8192);
      ADT_TR_PAIR(alo[3][0], ahi[3][0], a3, 0);       ADT_TR_PAIR(alo[3][1], ahi[3][1], a3, 8192);
      dma(pa[1], g.lda, t + 1, buf ^ 1, 1);
      asm volatile("s_barrier" ::: "memory");
      ADT_TN_QUAD(0, b0lo, b0hi, 0);
      // ---------------- phase 2: B_1; DMA B_0(t+2)
      ADT_TR_PAIR(b1lo[0][0], b1hi[0][0], bb0, kHalfTile);    ADT_TR_PAIR(b1lo[0][1], b1hi[0][1], bb0, kHalfTile + 8192);
      ADT_TR_PAIR(b1lo[1][0], b1hi[1][0], bb1, kHalfTile);    ADT_TR_PAIR(b1lo[1][1], b1hi[1][1], bb1, kHalfTile + 8192);
      dma(pb[0], g.ldb, t + 2, buf, 2);
      asm volatile("s_barrier" ::: "memory");
      ADT_TN_QUAD(0, b1lo, b1hi, 2);
      // ---------------- phase 3: A_1; DMA A_0(t+2)
      ADT_TR_PAIR(alo[0][0], ahi[0][0], a0, kHalfTile);       ADT_TR_PAIR(alo[0][1], ahi[0][1], a0, kHalfTile + 8192);
      ADT_TR_PAIR(alo[1][0], ahi[1][0], a1, kHalfTile);       ADT_TR_PAIR(alo[1][1], ahi[1][1], a1, kHalfTile + 8192);
      ADT_TR_PAIR(alo[2][0], ahi[2][0], a2, kHalfTile);       ADT_TR_PAIR(alo[2][1], ahi[2][1], a2, kHalfTile + 8192);
      ADT_TR_PAIR(alo[3][0], ahi[3][0], a3, kHalfTile);       ADT_TR_PAIR(alo[3][1], ahi[3][1], a3, kHalfTile + 8192);
      dma(pa[0], g.lda, t + 2, buf, 0);
      asm volatile("s_barrier" ::: "memory");
      ADT_TN_QUAD(4, b1lo, b1hi, 2);
      // ---------------- phase 4: DMA B_1(t+2); retire tile t+1
      dma(pb[1], g.ldb, t + 2, buf, 3);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      ADT_TN_QUAD(4, b0lo, b0hi, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    const int em0 = m0, en0 = n0, esplit = split;
    if (tid == 0) { ticket_drawn(counter, v_next, ctotal); *flag = v_next; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    v = static_cast<int>(*flag);
    const bool more = static_cast<unsigned>(v) < static_cast<unsigned>(slice_n);
    if (more) {
      set_item(v, m0, n0, split);
      prologue_dma();
    }

    // ---- epilogue (as gemm_nt_256_kernel): split-K items write the plain fp32 tile into their slab
    GemmArgs o = g;
    if (g.slabs) {
      o.C = g.slabs + static_cast<long>(esplit) * g.M * g.N;
      o.ldc = g.N;
      o.ep = adt_gemm_epilogue{};
      o.ep.alpha = 1.0f;
      o.ep.out_fp32 = 1;
    }
    const int ecol = en0 + wc * 64 + (lane & 7) * 8;
    const bool efull = ecol + 8 <= g.N;
    const EpiAddr ea = epilogue_addr<false, kEpiGeneric>(o, em0 + wr * 128 + (lane >> 3), ecol);
    float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (o.ep.bias && efull) {            // (weight gradients carry no bias: this load is not on the hot path)
      *reinterpret_cast<float4*>(bias) = *reinterpret_cast<const float4*>(o.ep.bias + ecol);
      *reinterpret_cast<float4*>(bias + 4) = *reinterpret_cast<const float4*>(o.ep.bias + ecol + 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("ds_write_b32 %0, %1" :: "v"(ct_w[j]), "v"(acc[i][j][0]));
        asm volatile("ds_write_b32 %0, %1 offset:256" :: "v"(ct_w[j]), "v"(acc[i][j][1]));
        asm volatile("ds_write_b32 %0, %1 offset:512" :: "v"(ct_w[j]), "v"(acc[i][j][2]));
        asm volatile("ds_write_b32 %0, %1 offset:768" :: "v"(ct_w[j]), "v"(acc[i][j][3]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      f32x4 zz[2][2];
      asm volatile("ds_read_b128 %0, %1" : "=v"(zz[0][0]) : "v"(ct_r[0]));
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(zz[0][1]) : "v"(ct_r[0]));
      asm volatile("ds_read_b128 %0, %1" : "=v"(zz[1][0]) : "v"(ct_r[1]));
      asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(zz[1][1]) : "v"(ct_r[1]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = em0 + wr * 128 + i * 16 + pass * 8 + (lane >> 3);
        float z[8] = {zz[pass][0][0], zz[pass][0][1], zz[pass][0][2], zz[pass][0][3], zz[pass][1][0], zz[pass][1][1], zz[pass][1][2], zz[pass][1][3]};
        if (row < g.M && efull) epilogue_apply8<false>(o, z, bias, ea, em0 + wr * 128 + (lane >> 3), i * 16 + pass * 8, ecol);
      }
    }
    if (!more) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
  }
#undef ADT_TN_QUAD
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
  if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) {
    *reinterpret_cast<float4*>(o) = float4{s.x * alpha, s.y * alpha, s.z * alpha, s.w * alpha};
    return;
  }
  o[0] = s.x * alpha; o[1] = s.y * alpha; o[2] = s.z * alpha; o[3] = s.w * alpha;
}

void launch_reduce_slabs(const float* slabs, int splits, long mn, int N, float alpha, float* out, long ldc, hipStream_t st) {
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(static_cast<unsigned>((mn / 4 + 255) / 256)), dim3(256), 0, st, slabs, splits, mn, N, alpha, out, ldc);
}

// One instantiation of the persistent NT kernel: (dropout, column sums, epilogue form).
template <bool kDrop, bool kColsum, unsigned kMask, bool kX3 = false>
static int launch_nt_256(const GemmArgs& g, dim3 grid, int tm, int tn, hipStream_t st) {
  static thread_local int attr_dev = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (attr_dev != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_256_kernel<kDrop, kColsum, kMask, kX3>), hipFuncAttributeMaxDynamicSharedMemorySize, kBigLds));
    attr_dev = dev;
  }
  hipLaunchKernelGGL((gemm_nt_256_kernel<kDrop, kColsum, kMask, kX3>), grid, dim3(kBigThreads), kBigLds, st, g, tm, tn);
  return ADT_OK;
}
// The forms the training step launches (adt_str_amd/network.py; ADT_GEMM_LOG_FORMS=1 lists what a workload launches) get their own
// instantiation: 8-28 KB of straight-line epilogue instead of 63-75 KB of run-time branches, -7 % over the eight NT GEMMs of an
// encoder layer timed inside the layer's launch sequence (tools/exp_gemm_instep.py).  Anything else runs the generic kernel, which
// reads the flags at run time (ADT_GEMM_GENERIC=1 forces it, for A/B runs).
//   X(dropout, colsum, mask)
#define ADT_NT256_FORMS(X)                                                                                       \
  X(true, false, kEfBias | kEfFactor | kEfPreAct | kEfGelu)   /* FFN linear1: bias, GELU, dropout, saved factor */      \
  X(false, false, kEfBias | kEfFactor | kEfPreAct | kEfGelu)  /* ... with dropout off */                                \
  X(false, false, 0u)                                         /* plain data gradient */                                 \
  X(false, false, kEfBias)                                    /* in-projections */                                      \
  X(true, false, kEfBias | kEfResidual | kEfFp32)             /* out-proj / FFN linear2: bias, dropout, residual */     \
  X(false, false, kEfBias | kEfResidual | kEfFp32)            /* ... with dropout off */                                \
  X(true, false, kEfBias | kEfResidual | kEfFp32 | kEfResLn)  /* ... the residual rebuilt from the pre-LayerNorm tensor */ \
  X(false, false, kEfBias | kEfResidual | kEfFp32 | kEfResLn)                                                            \
  X(false, false, kEfResidual | kEfFp32)                      /* data gradient added to the residual stream's */        \
  X(false, true, kEfGeluGrad | kEfFactor)                     /* FFN data gradient x saved factor, + bias gradient */   \
  X(false, false, kEfGeluGrad | kEfFactor)                                                                              \
  X(true, false, kEfResidual | kEfRowMod | kEfDropAfterRes | kEfAux | kEfFp32)   /* encoder input: + positional rows, dropout, fp32 + bf16 */ \
  X(true, false, kEfResidual | kEfDropAfterRes | kEfAux | kEfFp32)                                                      \
  X(false, false, kEfBias | kEfFp32)                          /* logits */                                              \
  X(false, false, kEfFp32)                                    /* the memory's data gradient (K = all decoder layers' K / V columns) */ \
  X(false, false, kEfBias | kEfGelu)                          /* inference MLP linear1 (the CLAP tower's last stage): bias, GELU */
static int dispatch_nt_256(const GemmArgs& g, bool colsum, unsigned mask, dim3 grid, int tm, int tn, hipStream_t st) {
  const bool drop = g.drop.on();
  if (g.x3_kt > 0) {                                      // split-bf16 products: the generic epilogue with the virtual-K-tile operand map
    if (colsum) return drop ? launch_nt_256<true, true, kEpiGeneric, true>(g, grid, tm, tn, st) : launch_nt_256<false, true, kEpiGeneric, true>(g, grid, tm, tn, st);
    return drop ? launch_nt_256<true, false, kEpiGeneric, true>(g, grid, tm, tn, st) : launch_nt_256<false, false, kEpiGeneric, true>(g, grid, tm, tn, st);
  }
  static const bool generic_only = getenv("ADT_GEMM_GENERIC") != nullptr;
  if (!generic_only) {
#define ADT_NT256_CASE(D, C, MK) if (drop == D && colsum == C && mask == (MK)) return launch_nt_256<D, C, (MK)>(g, grid, tm, tn, st);
    ADT_NT256_FORMS(ADT_NT256_CASE)
#undef ADT_NT256_CASE
  }
  if (colsum) return drop ? launch_nt_256<true, true, kEpiGeneric>(g, grid, tm, tn, st) : launch_nt_256<false, true, kEpiGeneric>(g, grid, tm, tn, st);
  return drop ? launch_nt_256<true, false, kEpiGeneric>(g, grid, tm, tn, st) : launch_nt_256<false, false, kEpiGeneric>(g, grid, tm, tn, st);
}

#ifdef ADT_GEMM_RING
// The ring kernels (gemm_nt_ring_kernel: 256 x 128 tiles, two workgroups per CU; 256 x 256 tiles with a four-slot ring), same instantiation list.
template <int kWN, int kStages, bool kDrop, bool kColsum, unsigned kMask>
static int launch_nt_ring(const GemmArgs& g, dim3 grid, int tm, int tn, hipStream_t st) {
  static thread_local int attr_dev = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  constexpr int lds = ring_lds_bytes<kWN, kStages>();
  if (attr_dev != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring_kernel<kWN, kStages, kDrop, kColsum, kMask>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_dev = dev;
  }
  hipLaunchKernelGGL((gemm_nt_ring_kernel<kWN, kStages, kDrop, kColsum, kMask>), grid, dim3(RingShape<kWN>::kThreads), lds, st, g, tm, tn);
  return ADT_OK;
}
template <int kWN, int kStages>
static int dispatch_nt_ring(const GemmArgs& g, bool colsum, unsigned mask, dim3 grid, int tm, int tn, hipStream_t st) {
  const bool drop = g.drop.on();
  static const bool generic_only = getenv("ADT_GEMM_GENERIC") != nullptr;
  if (!generic_only) {
#define ADT_NTRING_CASE(D, C, MK) if (drop == D && colsum == C && mask == (MK)) return launch_nt_ring<kWN, kStages, D, C, (MK)>(g, grid, tm, tn, st);
    ADT_NT256_FORMS(ADT_NTRING_CASE)
#undef ADT_NTRING_CASE
  }
  if (colsum) return drop ? launch_nt_ring<kWN, kStages, true, true, kEpiGeneric>(g, grid, tm, tn, st) : launch_nt_ring<kWN, kStages, false, true, kEpiGeneric>(g, grid, tm, tn, st);
  return drop ? launch_nt_ring<kWN, kStages, true, false, kEpiGeneric>(g, grid, tm, tn, st) : launch_nt_ring<kWN, kStages, false, false, kEpiGeneric>(g, grid, tm, tn, st);
}

#endif  // ADT_GEMM_RING

// Which persistent NT kernel a large product takes: 1 = gemm_nt_256_kernel (256 x 256 tiles, half-tile phases); 2 = ring kernel, 256 x 128
// tiles, two workgroups per CU; 3 = ring kernel, 256 x 256 tiles, four-slot ring.
// ADT_GEMM_NT=256 / 2wg / ring forces one (A/B runs, tests); read once -- except under ADT_GEMM_ENV_DYNAMIC=1 (tools that switch inside one process).
static int nt_persistent_form(int64_t M, int64_t N, int64_t K, unsigned mask) {
  static const bool dynamic = getenv("ADT_GEMM_ENV_DYNAMIC") != nullptr;
  auto read = [] { const char* v = getenv("ADT_GEMM_NT"); return !v ? 0 : (v[0] == '2' && v[1] == 'w') ? 2 : v[0] == 'r' ? 3 : 1; };
  static const int forced_once = read();
  const int forced = dynamic ? read() : forced_once;
  (void)M; (void)N; (void)K; (void)mask;
#ifdef ADT_GEMM_RING
  if (forced) return forced;
#else
  (void)forced;            // the ring kernels are not in this build: everything takes gemm_nt_256_kernel
#endif
  return 1;
}

#ifdef ADT_GEMM_RING
// The four-slot-ring 128^2 kernel instead of the two-stage one (experiment build only): ADT_GEMM_RING4=1 (read once; under
// ADT_GEMM_ENV_DYNAMIC=1 on every call, for tools that alternate inside one process).
static bool use_ring4(long tiles, int64_t K) {
  static const bool dynamic = getenv("ADT_GEMM_ENV_DYNAMIC") != nullptr;
  auto read = [] { const char* v = getenv("ADT_GEMM_RING4"); return !v ? -1 : (v[0] == '0' ? 0 : 1); };
  static const int forced_once = read();
  const int forced = dynamic ? read() : forced_once;
  if ((K % kR4BK) != 0 || K < 4 * kR4BK) return false;
  (void)tiles;
  return forced == 1;
}
#endif

static int set_big_lds_once() {      // the persistent kernels use the CU's whole LDS
  static thread_local int done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (done_for == dev) return ADT_OK;
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kBigLds));
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kBigLds));
  done_for = dev;
  return ADT_OK;
}

// the row-vector epilogue moves 8 columns at a time: everything it touches must be 16-byte aligned
static bool vector_epilogue_ok(const GemmArgs& g, const adt_gemm_epilogue& e) {
  auto ok = [](const void* p, long ld, int elem) { return !p || (aligned16(p) && (ld * elem) % 16 == 0); };
  return (g.N % 8) == 0 && ok(g.C, g.ldc, e.out_fp32 ? 4 : 2) && ok(e.bias, 4, 4) && ok(e.residual, e.ld_res, 4) &&
         ok(e.pre_act_out, e.ld_pre_act, e.side_fp32 ? 4 : 2) && ok(e.gelu_grad_of, e.ld_gelu_grad, e.side_fp32 ? 4 : 2) && ok(e.aux_bf16_out, e.ld_aux, 2);
}

// 256^2 tile: when there are at least two blocks per CU of it and the K loop is long enough to amortise its prologue.
// ADT_GEMM_TILE=128 / 256 in the environment forces one structure (A/B measurements, tests).
static bool use_big_tile(int64_t M, int64_t N, int64_t K) {
  static const int forced = [] { const char* v = getenv("ADT_GEMM_TILE"); return v ? atoi(v) : 0; }();
  if (K <= 0 || (K % kBK) != 0 || K < 2 * kBK || forced == 128) return false;
  if (forced == 256) return true;
  const int64_t tiles = ((M + kBig - 1) / kBig) * ((N + kBig - 1) / kBig);
  if (K < 256) return false;
  // a 256-wide tile over a narrower output is mostly padding; below 512 tiles (the decoder's M = 8192) the 256^2 kernel still
  // wins from N = 1400 up -- 192 tiles on 256 CUs included -- and loses at N = 768 (tools/exp_small_gemm.py)
  // (and with a long K loop from 384 tiles: the CLAP tower's last-stage fc2, M = 32768, N = 768, K = 3072, 182 vs 206 us -- tools/probe/clap_fc2_tile.py)
  return (tiles >= 512 && N >= 160) || (tiles >= 192 && N >= 1024) || (tiles >= 384 && K >= 2048 && N >= 512);
}

static int pick_splits(int M, int N, int K, int n_cu) {
  const int tiles = ((M + kBM - 1) / kBM) * ((N + kBN - 1) / kBN);
  const int k_tiles = (K + kBK - 1) / kBK;
  int s = 1;
  while (tiles * s < 2 * n_cu && s * 2 <= k_tiles / 4 && s < 64) s *= 2;
  return s;
}


// Weight-gradient form on the persistent 256^2 kernel: worth it when the output has enough 256^2 tiles that few K splits fill the
// chip (each split costs a full fp32 slab round trip).  Returns the number of splits (0: use the 128^2 kernel) and K-tiles per split.
static int plan_tn_big(int64_t M, int64_t N, int64_t K, int n_cu, bool may_split, int* per_split) {
  static const int forced = [] { const char* v = getenv("ADT_GEMM_TILE"); return v ? atoi(v) : 0; }();
  if (K <= 0 || (K % kBK) != 0 || forced == 128) return 0;
  const int k_tiles = static_cast<int>(K / kBK);
  const int64_t tiles = ((M + kBig - 1) / kBig) * ((N + kBig - 1) / kBig);
  if (forced != 256 && (tiles < 8 || k_tiles < 64)) return 0;
  if (k_tiles < 2) return 0;
  int s = 1;
  if (may_split) {
    s = static_cast<int>(n_cu / tiles);          // one item per CU, as many CUs as possible: every extra split is a slab round trip
    const int max_s = k_tiles / 16 > 1 ? k_tiles / 16 : 1;
    s = s < 1 ? 1 : (s > max_s ? max_s : s);
  }
  int per = (k_tiles + s - 1) / s;
  per = per < 2 ? 2 : per;
  s = (k_tiles + per - 1) / per;          // no empty trailing split
  *per_split = per;
  return s;
}
}  // namespace adt

static size_t tn_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  int n_cu = 256;
  (void)adt::device_cu_count(&n_cu);
  int s = adt::pick_splits(static_cast<int>(M), static_cast<int>(N), static_cast<int>(K), n_cu), per = 0;
  const int sb = adt::plan_tn_big(M, N, K, n_cu, true, &per);
  s = sb > s ? sb : s;
  return s > 1 ? static_cast<size_t>(s) * M * N * 4 : 0;
}
extern "C" size_t adt_gemm_workspace_bytes(int32_t trans, int64_t M, int64_t N, int64_t K) {
  if (!trans || M <= 0 || N <= 0 || K <= 0) return 0;
  const size_t whole = tn_workspace_bytes(M, N, K);
  if ((K % adt::kBK) == 0 || K < adt::kBK) return whole;
  const size_t head = tn_workspace_bytes(M, N, K / adt::kBK * adt::kBK);       // K % 64 != 0: the product is taken in two pieces (adt_gemm_bf16)
  return head > whole ? head : whole;
}

extern "C" size_t adt_gemm_colsum_workspace_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  const size_t fused = static_cast<size_t>((M + adt::kBig - 1) / adt::kBig) * 2 * N * 4, alone = adt_colsum_workspace_bytes(M, N);
  return fused > alone ? fused : alone;
}

// x3_plane > 0 (adt_gemm_bf16x3): A / B are [hi | lo] plane pairs, K is the VIRTUAL depth 3 * x3_plane, a_lo / b_lo the element offset of
// the lo plane from the hi plane; only the persistent kernels know that operand map, every other path is refused.
static int gemm_bf16_impl(int32_t trans, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                          const void* B, int64_t ldb, void* C, int64_t ldc, const adt_gemm_epilogue* ep,
                          void* ws, size_t ws_bytes, void* stream, int64_t x3_plane, int64_t a_lo, int64_t b_lo) {
  using namespace adt;
  if (!A || !B || !C) return set_error(ADT_EINVAL, "adt_gemm_bf16: null pointer");
  if (M < 0 || N < 0 || K < 0) return set_error(ADT_EINVAL, "adt_gemm_bf16: negative size");
  if (M >= (1ll << 31) || N >= (1ll << 31) || K >= (1ll << 31)) return set_error(ADT_ESHAPE, "adt_gemm_bf16: dimension >= 2^31");
  if (ep && ep->side_fp32 && x3_plane == 0) return set_error(ADT_EINVAL, "adt_gemm_bf16: side_fp32 belongs to adt_gemm_bf16x3");
  const int64_t a_cols = x3_plane ? (trans ? M : x3_plane) : (trans ? M : K), b_cols = x3_plane ? (trans ? N : x3_plane) : (trans ? N : K);
  if (lda < a_cols || ldb < b_cols || ldc < N) return set_error(ADT_EINVAL, "adt_gemm_bf16: leading dimension too small");
  if ((a_cols & 7) || (b_cols & 7) || (lda & 7) || (ldb & 7) || !aligned16(A) || !aligned16(B))
    return set_error(ADT_ESHAPE, "adt_gemm_bf16: operand rows must be 16-byte aligned multiples of 8 elements");
  if (M == 0 || N == 0) return ADT_OK;
  // Weight gradients whose K (= batch x frames rows) is not a multiple of the 64-deep K-tile: the LDS-DMA kernels want whole tiles,
  // and the register-staged fallback is several times slower at K ~ 50 000.  Take the first floor(K / 64) * 64 rows on the fast
  // path and add the last < 64 rows with the fallback kernel (C is both its residual and its output: every element is read and
  // written by one thread).  Same products, fp32 sums in a fixed order.
  if (!x3_plane && trans && K > kBK && (K % kBK) != 0 && ep && ep->out_fp32 && !ep->bias && !ep->residual && !ep->act && !ep->pre_act_out && !ep->gelu_grad_of &&
      !ep->aux_bf16_out && !ep->colsum_out && ep->drop.p <= 0.f) {
    const int64_t k0 = K / kBK * kBK;
    if (int rc = adt_gemm_bf16(1, M, N, k0, A, lda, B, ldb, C, ldc, ep, ws, ws_bytes, stream)) return rc;
    adt_gemm_epilogue tail = *ep;
    tail.residual = C; tail.ld_res = ldc; tail.res_row_mod = 0;
    return adt_gemm_bf16(1, M, N, K - k0, static_cast<const unsigned short*>(A) + k0 * lda, lda, static_cast<const unsigned short*>(B) + k0 * ldb, ldb,
                         C, ldc, &tail, ws, ws_bytes, stream);
  }
  GemmArgs g;
  g.A = static_cast<const unsigned short*>(A); g.lda = lda;
  g.B = static_cast<const unsigned short*>(B); g.ldb = ldb;
  g.C = C; g.ldc = ldc; g.M = static_cast<int>(M); g.N = static_cast<int>(N); g.K = static_cast<int>(K);
  adt_gemm_epilogue e;
  if (ep) e = *ep; else { e = adt_gemm_epilogue{}; e.alpha = 1.0f; }
  g.ep = e;
  g.drop = make_drop(e.drop.p, e.drop.key);
  g.drop_key2 = mix32(g.drop.key);
  g.x3_kt = static_cast<int>(x3_plane / kBK); g.x3_a_lo = a_lo; g.x3_b_lo = b_lo;
  const int k_tiles = static_cast<int>((K + kBK - 1) / kBK);
  int splits = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  g.colsum_ws = nullptr;
  bool colsum_done = false;
  if (e.res_ln_mean) {
    if (!e.residual || !e.res_ln_rstd || !e.res_ln_gamma || !e.res_ln_beta || e.res_row_mod > 0 || trans || !aligned16(e.res_ln_gamma) || !aligned16(e.res_ln_beta))
      return set_error(ADT_EINVAL, "adt_gemm_bf16: res_ln_* needs trans = 0, a residual (the pre-LayerNorm tensor), rstd / gamma / beta (16-byte aligned) and res_row_mod = 0");
  }
  if (e.colsum_out) {
    if (trans || e.out_fp32) return set_error(ADT_EINVAL, "adt_gemm_bf16: colsum_out needs trans = 0 and a bf16 output");
    if (!ws || !aligned16(ws) || ws_bytes < adt_gemm_colsum_workspace_bytes(M, N))
      return set_error(ADT_EINVAL, "adt_gemm_bf16: workspace too small (see adt_gemm_colsum_workspace_bytes)");
  }
  // The persistent kernels share one set of work counters per (device, stream) and allocate it on first use: under stream
  // capture (a graph may be replayed on any stream, concurrently with itself) they are not used; the tiled kernels are.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st != nullptr && hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  const bool persistent_ok = cap == hipStreamCaptureStatusNone;
  if (trans && persistent_ok && vector_epilogue_ok(g, e) && M >= 8 && N >= 8) {
    int n_cu = 0, per = 0;
    if (int rc = device_cu_count(&n_cu)) return rc;
    const bool plain = e.out_fp32 && !e.bias && !e.residual && !e.act && !e.pre_act_out && !e.gelu_grad_of && !e.aux_bf16_out && e.drop.p <= 0.f && !(N & 3);
    const int sb = plan_tn_big(M, N, K, n_cu, plain, &per);
    if (sb >= 1 && !g.drop.on() && (sb == 1 || (ws && aligned16(ws)))) {
      if (sb > 1 && ws_bytes < static_cast<size_t>(sb) * M * N * 4)
        return set_error(ADT_EINVAL, "adt_gemm_bf16: workspace too small (see adt_gemm_workspace_bytes)");
      if (int rc = set_big_lds_once()) return rc;
      g.k_tiles_per_split = per;
      g.slabs = sb > 1 ? static_cast<float*>(ws) : nullptr;
      const int tm = static_cast<int>((M + kBig - 1) / kBig), tn = static_cast<int>((N + kBig - 1) / kBig);
      const long items = static_cast<long>(tm) * tn * sb;
      const dim3 g1(static_cast<unsigned>(items < n_cu ? items : n_cu));
      for (int x = 0; x < 8; ++x)        // per slice: one ticket per work item + the ending ticket of each of its workgroups
        g.sched_total[x] = static_cast<unsigned>(items / 8 + (x < items % 8 ? 1 : 0)) + g1.x / 8 + (static_cast<unsigned>(x) < g1.x % 8 ? 1u : 0u);
      if (int rc = sched_counters(stream, &g.sched)) return rc;
      if (g.x3_kt > 0) hipLaunchKernelGGL(gemm_tn_256_kernel<true>, g1, dim3(kBigThreads), kBigLds, st, g, tm, tn, sb);
      else hipLaunchKernelGGL(gemm_tn_256_kernel<false>, g1, dim3(kBigThreads), kBigLds, st, g, tm, tn, sb);
      if (sb > 1) {
        const long mn = static_cast<long>(M) * N;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(static_cast<unsigned>((mn / 4 + 255) / 256)), dim3(256), 0, st,
                           g.slabs, sb, mn, g.N, e.alpha, static_cast<float*>(C), ldc);
      }
      ADT_HIP_TRY(hipGetLastError());
      return ADT_OK;
    }
  }
  if (x3_plane && (trans || !(persistent_ok && use_big_tile(M, N, K) && vector_epilogue_ok(g, e))))
    return set_error(ADT_ESHAPE, "adt_gemm_bf16x3: this shape does not take a persistent kernel (ask adt_gemm_bf16x3_supported; use adt_gemm_f32)");
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
  // up to 64 rows (the decode step's projections): one 16-column workgroup per output slice, K split over its waves
  static const bool no_skinny = getenv("ADT_GEMM_NO_SKINNY") != nullptr;
  if (!trans && M >= 1 && M <= 64 && K > 0 && (K % 128) == 0 && !e.colsum_out && vector_epilogue_ok(g, e) && aligned16(A) && aligned16(B) &&
      (lda % 8) == 0 && (ldb % 8) == 0 && !no_skinny) {
    if (g.drop.on()) launch_skinny<true>(g, st); else launch_skinny<false>(g, st);
    ADT_HIP_TRY(hipGetLastError());
    return ADT_OK;
  }
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
#ifdef ADT_GEMM_RING
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kR4Lds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kR4Lds));
#endif
    if (int rc = set_big_lds_once()) return rc;
    attr_dev = dev;
  }
  if (trans && (K % kBK) == 0 && K > 0 && vector_epilogue_ok(g, e) && M >= 8 && (splits == 1 || aligned16(ws))) {
    const int tm = static_cast<int>((M + kBM - 1) / kBM), tn = static_cast<int>((N + kBN - 1) / kBN);
    hipLaunchKernelGGL(gemm_tn_glds_kernel, dim3(static_cast<unsigned>(tm) * tn, splits), dim3(kGemmThreads), kEpiLds, st, g, tm, tn);
  } else if (trans) {
    if (g.drop.on()) return set_error(ADT_EINVAL, "adt_gemm_bf16: dropout is not supported with trans = 1");
    hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, dim3(kGemmThreads), kGemmLds, st, g);
  } else if (persistent_ok && use_big_tile(M, N, K) && vector_epilogue_ok(g, e)) {
    const unsigned mask = epilogue_mask(e);
    const int nt_form = nt_persistent_form(M, N, K, mask);
    const bool two_wg = nt_form == 2;
    const int tile_n = two_wg ? 128 : kBig;
    const int tm = static_cast<int>((M + kBig - 1) / kBig), tn = static_cast<int>((N + tile_n - 1) / tile_n);
    int n_cu = 0;
    if (int rc = device_cu_count(&n_cu)) return rc;
    const long nt = static_cast<long>(tm) * tn;
    const long wg_max = two_wg ? 2l * n_cu : n_cu;
    const dim3 g1(static_cast<unsigned>(nt < wg_max ? nt : wg_max));      // persistent: one workgroup per CU (two of the 256 x 128 form)
    for (int x = 0; x < 8; ++x)          // per slice: one ticket per tile + the ending ticket of each of its workgroups
      g.sched_total[x] = static_cast<unsigned>(nt / 8 + (x < nt % 8 ? 1 : 0)) + g1.x / 8 + (static_cast<unsigned>(x) < g1.x % 8 ? 1u : 0u);
    if (int rc = sched_counters(stream, &g.sched)) return rc;
    static const bool tail_off = getenv("ADT_GEMM_NO_TAIL_STORES") != nullptr;
    g.tail_stores = tail_off ? 0 : 1;
    static const int group_env = getenv("ADT_GEMM_GROUP_N") ? atoi(getenv("ADT_GEMM_GROUP_N")) : 0;
    // tile columns per column group: measured per number of tile columns (tools/exp_gemm_group.py, profiles/r05/gemm_group_n.txt: M = 63104,
    // bare products): 12 columns (N = 3072) 4 -> 305 us against 323 with 3 (308 / 319 / 325 with 2 / 6 / 12); 9 columns (N = 2304) 3 -> 195
    // (198 with 4, 228 with 5); 24 columns (N = 6144) 6 -> 538 (547 / 556 / 583 with 3 / 4 / 8); three columns or fewer: all of them
    const int tn256 = static_cast<int>((N + kBig - 1) / kBig);
    const int group_auto = tn256 >= 24 && tn256 % 6 == 0 ? 6 : (tn256 % 4 == 0 && tn256 % 3 != 0) || tn256 == 12 ? 4 : 3;
    const int group_n = (group_env > 0 ? group_env : group_auto) * (two_wg ? 2 : 1);      // (counted in this kernel's tile columns)
    g.group_n = group_n < tn ? group_n : tn;
    g.stagger = 0;
#ifdef ADT_GEMM_EXPERIMENT
    { const char* sg = getenv("ADT_GEMM_STAGGER"); g.stagger = sg ? atoi(sg) : 0; }
    if (g.stagger) fprintf(stderr, "adt_gemm_bf16: ADT_GEMM_STAGGER=%d (experiment build)\n", g.stagger);
#endif
    if (two_wg) {
      // the CU's second workgroup starts about half a tile late, so that one's epilogue meets the other's K loop (ticks of s_memtime)
      static const int stagger2_env = getenv("ADT_GEMM_STAGGER2") ? atoi(getenv("ADT_GEMM_STAGGER2")) : -1;
      g.stagger = stagger2_env >= 0 ? stagger2_env : static_cast<int>(K / 32) * 300 + 5000;
    }
    static const bool log_forms = getenv("ADT_GEMM_LOG_FORMS") != nullptr;      // debugging aid: which forms does a workload launch?
    if (log_forms) fprintf(stderr, "adt_gemm nt256 form: drop=%d colsum=%d mask=0x%x M=%ld N=%ld K=%ld\n", g.drop.on() ? 1 : 0, e.colsum_out ? 1 : 0, mask, (long)M, (long)N, (long)K);
    float* cs_slice = nullptr;
    if (e.colsum_out) {
      cs_slice = reduce_queue_slice(static_cast<size_t>(2 * tm) * g.N * 4, st);          // open reduction queue: the second stage is deferred
      g.colsum_ws = cs_slice ? cs_slice : static_cast<float*>(ws);
    }
#ifdef ADT_GEMM_RING
    const int rc = two_wg ? dispatch_nt_ring<2, 3>(g, e.colsum_out != nullptr, mask, g1, tm, tn, st)
                   : nt_form == 3 ? dispatch_nt_ring<4, 4>(g, e.colsum_out != nullptr, mask, g1, tm, tn, st)
                                  : dispatch_nt_256(g, e.colsum_out != nullptr, mask, g1, tm, tn, st);
#else
    const int rc = dispatch_nt_256(g, e.colsum_out != nullptr, mask, g1, tm, tn, st);
#endif
    if (rc) return rc;
#ifdef ADT_GEMM_EXPERIMENT
    if (getenv("ADT_GEMM_STAMPS")) {
      unsigned long long h[4][16][4];
      ADT_HIP_TRY(hipStreamSynchronize(st));
      ADT_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_stamps), sizeof(h)));
      for (int b = 0; b < 4; ++b) {
        fprintf(stderr, "gemm stamps wg %d (ticks from wg 0's first K loop; per tile: K-loop start, +K loop, +epilogue, +tile-end wait):", 8 * b);
        for (int t = 0; t < 14; ++t)
          fprintf(stderr, " [%lld %lld %lld %lld]", static_cast<long long>(h[b][t][0] - h[0][0][0]), static_cast<long long>(h[b][t][1] - h[b][t][0]),
                  static_cast<long long>(h[b][t][2] - h[b][t][1]), static_cast<long long>(h[b][t][3] - h[b][t][2]));
        fprintf(stderr, "\n");
      }
    }
#endif
    if (e.colsum_out) {
      if (cs_slice) {
        if (int rc2 = reduce_queue_push(cs_slice, 2 * tm, g.N, e.colsum_out, nullptr, nullptr, g.N)) return rc2;
      } else {
        launch_reduce_partials(g.colsum_ws, 2 * tm, g.N, e.colsum_out, st);
      }
      colsum_done = true;
    }
  } else if (K > 0 && vector_epilogue_ok(g, e)) {       // any K (a multiple of 8, checked above): the last K-tile is zero-filled
    const int tm = static_cast<int>((M + kBM - 1) / kBM), tn = static_cast<int>((N + kBN - 1) / kBN);
    const dim3 g1(static_cast<unsigned>(tm) * tn);
#ifdef ADT_GEMM_RING
    if (use_ring4(tm * static_cast<long>(tn), K)) {         // experiment build, ADT_GEMM_RING4=1: the four-slot ring (three K-steps in flight)
      if (g.drop.on()) hipLaunchKernelGGL(gemm_nt_ring4_kernel<true>, g1, dim3(kGemmThreads), kR4Lds, st, g, tm, tn);
      else hipLaunchKernelGGL(gemm_nt_ring4_kernel<false>, g1, dim3(kGemmThreads), kR4Lds, st, g, tm, tn);
    } else
#endif
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
  if (e.colsum_out && !colsum_done)                  // the smaller tilings leave the sums to the stand-alone kernel
    return adt_colsum_bf16(C, ldc, M, N, e.colsum_out, ws, ws_bytes, stream);
  return ADT_OK;
}

extern "C" int adt_gemm_bf16(int32_t trans, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                             const void* B, int64_t ldb, void* C, int64_t ldc, const adt_gemm_epilogue* ep,
                             void* ws, size_t ws_bytes, void* stream) {
  return gemm_bf16_impl(trans, M, N, K, A, lda, B, ldb, C, ldc, ep, ws, ws_bytes, stream, 0, 0, 0);
}

// Does (trans, M, N, K) -- K the depth of ONE plane -- take a persistent kernel, i.e. can adt_gemm_bf16x3 run it?
extern "C" int32_t adt_gemm_bf16x3_supported(int32_t trans, int64_t M, int64_t N, int64_t K) {
  using namespace adt;
  if (M < 8 || N < 8 || K < kBK || (K % kBK) != 0 || (N % 8) != 0 || (M % 8) != 0 || 3 * K >= (1ll << 31)) return 0;
  if (!trans) return use_big_tile(M, N, 3 * K) ? 1 : 0;
  int n_cu = 256, per = 0;
  (void)device_cu_count(&n_cu);
  return plan_tn_big(M, N, 3 * K, n_cu, true, &per) >= 1 ? 1 : 0;
}
extern "C" size_t adt_gemm_bf16x3_workspace_bytes(int32_t trans, int64_t M, int64_t N, int64_t K) {
  return adt_gemm_workspace_bytes(trans, M, N, 3 * K);
}
extern "C" int adt_gemm_bf16x3(int32_t trans, int64_t M, int64_t N, int64_t K, const void* A2, int64_t lda, int64_t a_lo,
                               const void* B2, int64_t ldb, int64_t b_lo, float* C, int64_t ldc, const adt_gemm_epilogue* ep,
                               void* ws, size_t ws_bytes, void* stream) {
  using namespace adt;
  if (K <= 0 || (K % kBK) != 0) return set_error(ADT_ESHAPE, "adt_gemm_bf16x3: K must be a positive multiple of 64");
  if (!ep || !ep->out_fp32 || ep->aux_bf16_out || ep->colsum_out || ep->res_ln_mean)
    return set_error(ADT_EINVAL, "adt_gemm_bf16x3: fp32 output only, no bf16 side output, column sums or rebuilt-LayerNorm residual");
  if (a_lo <= 0 || b_lo <= 0 || (a_lo & 7) || (b_lo & 7)) return set_error(ADT_EINVAL, "adt_gemm_bf16x3: plane offsets must be positive multiples of 8 elements");
  const int64_t a_w = trans ? M : K, b_w = trans ? N : K;
  if (a_lo < a_w || b_lo < b_w || lda < a_lo + a_w || ldb < b_lo + b_w) return set_error(ADT_EINVAL, "adt_gemm_bf16x3: planes overlap or exceed the row stride");
  if (!adt_gemm_bf16x3_supported(trans, M, N, K)) return set_error(ADT_ESHAPE, "adt_gemm_bf16x3: shape not supported (adt_gemm_bf16x3_supported)");
  return gemm_bf16_impl(trans, M, N, 3 * K, A2, lda, B2, ldb, C, ldc, ep, ws, ws_bytes, stream, K, a_lo, b_lo);
}

extern "C" int adt_ln_gemm_bf16(int64_t M, int64_t N, int64_t K, const float* y, int64_t ldy, const float* gamma, const float* beta, float eps,
                               const void* B, int64_t ldb, void* C, int64_t ldc, const adt_gemm_epilogue* ep, float* x32, int64_t ldx, void* stream) {
  using namespace adt;
  if (!y || !gamma || !beta || !B || !C) return set_error(ADT_EINVAL, "adt_ln_gemm_bf16: null pointer");
  if (M < 1 || M > 64 || N <= 0 || K <= 0 || (K % 128) != 0 || K > 1024) return set_error(ADT_ESHAPE, "adt_ln_gemm_bf16: 1 <= M <= 64, K a multiple of 128 up to 1024");
  if ((ldy & 3) || (ldb & 7) || !aligned16(y) || !aligned16(B) || !aligned16(gamma) || !aligned16(beta) || (x32 && ldx < K))
    return set_error(ADT_EINVAL, "adt_ln_gemm_bf16: y / gamma / beta / B must be 16-byte aligned with aligned row strides");
  GemmArgs g{};
  g.A = nullptr; g.lda = 0;
  g.B = static_cast<const unsigned short*>(B); g.ldb = ldb;
  g.C = C; g.ldc = ldc; g.M = static_cast<int>(M); g.N = static_cast<int>(N); g.K = static_cast<int>(K);
  adt_gemm_epilogue e;
  if (ep) e = *ep; else { e = adt_gemm_epilogue{}; e.alpha = 1.0f; }
  if (e.colsum_out) return set_error(ADT_EINVAL, "adt_ln_gemm_bf16: colsum_out is not supported");
  g.ep = e;
  g.drop = make_drop(e.drop.p, e.drop.key);
  g.drop_key2 = mix32(g.drop.key);
  if (!vector_epilogue_ok(g, e)) return set_error(ADT_EINVAL, "adt_ln_gemm_bf16: outputs must be 16-byte aligned, N a multiple of 8");
  const LnPrologue p{y, static_cast<long>(ldy), gamma, beta, eps, x32, static_cast<long>(ldx)};
  const dim3 gs(static_cast<unsigned>((N + 15) / 16));
  hipStream_t st = static_cast<hipStream_t>(stream);
#define ADT_LN_SKINNY(D, T) hipLaunchKernelGGL((gemm_nt_skinny_ln_kernel<D, T>), gs, dim3(kSkinnyThreads), 0, st, g, p)
  if (g.drop.on()) { if (M <= 16) ADT_LN_SKINNY(true, 1); else if (M <= 32) ADT_LN_SKINNY(true, 2); else ADT_LN_SKINNY(true, 4); }
  else { if (M <= 16) ADT_LN_SKINNY(false, 1); else if (M <= 32) ADT_LN_SKINNY(false, 2); else ADT_LN_SKINNY(false, 4); }
#undef ADT_LN_SKINNY
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

extern "C" int adt_gemm_bf16_tn_grouped(const adt_gemm_tn_item* items, int32_t n, void* stream) {
  using namespace adt;
  if (n < 0 || (n > 0 && !items)) return set_error(ADT_EINVAL, "adt_gemm_bf16_tn_grouped: bad item list");
  if (n == 0) return ADT_OK;
  if (n > kMaxGroup) return set_error(ADT_EINVAL, "adt_gemm_bf16_tn_grouped: at most 32 items per call");
  TnGroupArgs ga;
  ga.n = n;
  long total = 0;
  for (int i = 0; i < n; ++i) {
    const adt_gemm_tn_item& it = items[i];
    if (it.M < 8 || it.N < 8 || it.K <= 0 || (it.K % kBK) != 0 || (it.N % 8) != 0 || (it.M % 8) != 0)
      return set_error(ADT_ESHAPE, "adt_gemm_bf16_tn_grouped: M, N must be multiples of 8 (>= 8) and K a multiple of 64");
    if (!it.A || !it.B || !it.C) return set_error(ADT_EINVAL, "adt_gemm_bf16_tn_grouped: null pointer");
    if (!aligned16(it.A) || !aligned16(it.B) || !aligned16(it.C) || (it.lda % 8) != 0 || (it.ldb % 8) != 0 || (it.ldc % 4) != 0 ||
        it.lda < it.M || it.ldb < it.N || it.ldc < it.N)
      return set_error(ADT_EINVAL, "adt_gemm_bf16_tn_grouped: operands must be 16-byte aligned with 16-byte-multiple row strides");
    const long tm = (it.M + kBM - 1) / kBM, tn = (it.N + kBN - 1) / kBN;
    ga.tile_start[i] = static_cast<int>(total);
    ga.tiles_n[i] = static_cast<int>(tn);
    total += tm * tn;
    if (total > (1l << 30)) return set_error(ADT_ESHAPE, "adt_gemm_bf16_tn_grouped: too many tiles");
    ga.M[i] = static_cast<int>(it.M); ga.N[i] = static_cast<int>(it.N); ga.K[i] = static_cast<int>(it.K);
    ga.A[i] = static_cast<const unsigned short*>(it.A); ga.B[i] = static_cast<const unsigned short*>(it.B); ga.C[i] = it.C;
    ga.lda[i] = it.lda; ga.ldb[i] = it.ldb; ga.ldc[i] = it.ldc;
  }
  ga.tile_start[n] = static_cast<int>(total);
  static thread_local int attr_dev = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (attr_dev != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_grouped_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kEpiLds));
    attr_dev = dev;
  }
  hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(static_cast<unsigned>(total)), dim3(kGemmThreads), kEpiLds, static_cast<hipStream_t>(stream), ga);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
