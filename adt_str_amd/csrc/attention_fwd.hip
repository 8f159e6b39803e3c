// attention_fwd.hip -- K4 forward, software-pipelined: the flash-style forward of attention.hip's header (lane <-> query, S^T = K Q^T,
// O^T += V^T P^T, v_mfma_f32_32x32x16_bf16, 128 queries per workgroup = 4 waves x 32, two workgroups per CU) with the two products
// of neighbouring 32-key blocks overlapped inside each wave.  Stands behind nn.MultiheadAttention of the reference's encoder / decoder
// layers (model.py:118-127, 159-168) exactly as the kernel it replaces.
//
// A wave's own chain per 32-key block is   K reads -> 8 MFMAs (S) -> max -> exp / sum / dropout / pack -> V^T reads -> 8 MFMAs (O):
// run in that order (the first forward kernel) it costs ~1800 cycles for 512 cycles of MFMA, and the second wave of the SIMD only
// hides part of it.  Here block g + 1's score product runs UNDER block g's exponentials, and block g's value product under block
// g + 1's row maximum and dropout hashes:
//     phase A(g):  S(g+1) = K(g+1) Q^T          ||  finish(g): p = exp2(s * scale * log2e - m), row sum, dropout, bf16 pack
//     phase B(g):  O^T += V(g)^T P(g)^T         ||  start(g+1): mask, row maximum over the 32 keys;  hashes of block g + 1
//     then the lazy rescale of (m, l, O) if some row's maximum rose by more than kRescaleThr (after block g's product went into O).
// Each phase is cut into eight slices of one MFMA + one eighth of the vector work, separated by scheduling fences, so that the
// interleaving in the source is the interleaving in the binary; the compiler still inserts every wait and hazard no-op (builtins only).
// Tiles of 64 keys arrive by LDS-DMA into two K and two V buffers; ONE barrier per tile, between A(2t) and B(2t): K(t) is dead there
// (its rows 32..63 were just used) and V(t - 1) has been dead for two phases, so K(t + 2) and V(t + 1) are issued right behind it,
// and K(t + 1), V(t) -- issued one tile earlier -- are made visible by it, one phase before their first use.
// Dropout keeps the element iff its 16-bit hash half >= thr (dropout.h); the 1 / (1 - p) factor is folded into the exponent
// (exp2(x + log2(1 / (1 - p)))), so a dropped element costs a compare and a select, and the row sum is rescaled once at the end.
#include "attn_common.h"
#include <cstdio>

namespace adt {

typedef short v4s __attribute__((ext_vector_type(4)));
#define ADT_AS3 __attribute__((address_space(3)))
#define ADT_FENCE() __builtin_amdgcn_sched_barrier(0)

// (the builtin, not attn_common.h's asm pack2: its inputs come straight from v_exp_f32 here, and the hazard recogniser does not see inside asm)
typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_c(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2p));
}
// One LDS-DMA instruction (1 KiB: lane l's 16 bytes land at lds_off + 16 l) as asm: through the builtin the compiler treats the transfer as
// a store that may alias every later LDS read and puts s_waitcnt vmcnt(0) in front of the next phase's reads -- the whole HBM / L2 latency
// of a transfer that is not needed for another tile.  Waits for these are the explicit vmcnt(0) in front of the tile's barrier.
__device__ __forceinline__ void dma1k(const unsigned short* src, unsigned lds_off_) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_off_) : "memory");
}
// Pins a value's computation in front of this point: the vector work of a slice is only consumed a phase later, and left alone the
// optimiser sinks it across the fences into the block that uses it (the fences order machine instructions inside a block, not IR).
__device__ __forceinline__ void pin(unsigned& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ float cross_half_max(float v) {      // max over lanes l and l ^ 32, in both
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

#ifdef ADT_FWD_EXPERIMENT      // cycle stamps of one wave's phases over two tiles (tools/probe/attn_fwd_stamps.py)
__device__ unsigned long long g_fwd_stamps[32];
#define ADT_STAMP(K)                                                                                          \
  do {                                                                                                        \
    if (stamp_on && (t == 6 || t == 7)) {                                                                     \
      const unsigned long long tnow = __builtin_amdgcn_s_memtime();                                           \
      if (lane == 0) g_fwd_stamps[(t - 6) * 8 + (K)] = tnow;                                                  \
    }                                                                                                         \
  } while (0)
#else
#define ADT_STAMP(K) do { } while (0)
#endif

// kWaves = 8: one workgroup of 256 queries per CU (two waves per SIMD): a K / V tile is fetched once for eight waves -- the LDS-DMA path
// delivers ~33 B/clk per CU (the same bound as the GEMM's operand delivery), and a wave's DMA instruction costs it ~80 issue cycles.
// kWaves = 4: 128 queries, two workgroups per CU: the shapes with at most 128 queries (decoder self- and cross-attention).
template <bool kDrop, int kWaves>
__global__ __launch_bounds__(64 * kWaves, 8 / kWaves) void attn_fwd2_kernel(AttnArgs a) {
  constexpr int kQ = 32 * kWaves;          // queries per workgroup
  constexpr int kPc = 16 / kWaves;         // 1-KiB DMA pieces of a 16-KiB tile per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [2][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ADT_FWD_EXPERIMENT
  const bool stamp_on = blockIdx.x == gridDim.x / 2 + 3 && wave == 1;
#endif
  const TileXY tc = tile_coords((a.Sq + kQ - 1) / kQ);
  const int b = tc.y / a.H, head = tc.y % a.H;
  const int qi = tc.x * kQ + wave * 32 + r;
  const unsigned short* qb = a.q + static_cast<long>(b) * a.Sq * a.ldq + head * kDh;
  const unsigned short* kg = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * kDh;
  const unsigned short* vg = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * kDh;
  const int klen = __builtin_amdgcn_readfirstlane(a.key_len ? a.key_len[b] : a.Sk);
  const float sl2 = a.scale * kLog2e;
  const float mvs = a.mask_value / a.scale;                  // the additive mask in raw-score units
  const float cfold = kDrop ? __log2f(a.drop.inv_keep) : 0.f;
  const unsigned thr = a.drop.thr;

  bf16x8 qf[8];
  frags_from_global(qb, a.ldq, qi, a.Sq, lane, qf);
  f32x16 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
  float m = kNegBig, mneg = cfold - kNegBig, l = 0.f;
  // dropout index of (row, key) is row * Sk2 + key (Sk2 = Sk rounded up to even; dropout.h): pair = row * Sk2 / 2 + key / 2
  const unsigned pb2 = static_cast<unsigned>(((static_cast<uint64_t>(b) * a.H + head) * a.Sq + qi) * ((a.Sk + 1) >> 1)) + 2u * h;
  const unsigned key2 = mix32(a.drop.key);

  const int n_tiles = (a.Sk + kRowsPerTile - 1) / kRowsPerTile;
  // DMA piece j of this wave = rows 4 (kPc wave + j) .. + 3 of a tile (one row per quarter-wave, 16 bytes per lane); the LDS image is
  // lane-linear, so the swizzle sits on the source address (attn_common.h tile_dma)
  unsigned koff[kPc], voff[kPc];
#pragma unroll
  for (int j = 0; j < kPc; ++j) {
    const int row = 4 * (kPc * wave + j) + (lane >> 4);
    const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    koff[j] = static_cast<unsigned>(row * a.ldk + chunk * 8);
    voff[j] = static_cast<unsigned>(row * a.ldv + chunk * 8);
  }
  const unsigned lds_base = lds_off(smem);
  // edge tile: element offset of piece j's source with rows past the end clamped to the last one (their scores are masked)
  auto clamp_off = [&](int row0, int j, long ld) __attribute__((always_inline)) {
    const int row = 4 * (kPc * wave + j) + (lane >> 4);
    const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    int gr = row0 + row;
    gr = gr < a.Sk ? gr : a.Sk - 1;
    return static_cast<long>(gr) * ld + chunk * 8;
  };
  // one piece of tile `row0 / 64` of K (ld = ldk) or V into the 16-KiB buffer at LDS byte offset buf
  auto dma_piece_of = [&](const unsigned short* g, long ld, const unsigned (&off)[kPc], int row0, int j, int buf) __attribute__((always_inline)) {
    const unsigned dst = lds_base + buf + (kPc * wave + j) * 1024;
    if (row0 + kRowsPerTile <= a.Sk) dma1k(g + static_cast<long>(row0) * ld + off[j], dst);          // block-uniform
    else if (row0 < a.Sk) dma1k(g + clamp_off(row0, j, ld), dst);
  };
#pragma unroll
  for (int j = 0; j < kPc; ++j) {
    dma_piece_of(kg, a.ldk, koff, 0, j, 0);
    dma_piece_of(vg, a.ldv, voff, 0, j, kAttnTileBytes);
    dma_piece_of(kg, a.ldk, koff, kRowsPerTile, j, 2 * kAttnTileBytes);
  }
  // lane constants of the LDS reads: K rows (16-byte pieces, chunk 2s + h) and V^T (transposed 8-byte pieces)
  unsigned kfo[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) kfo[s] = static_cast<unsigned>(swz(r, 2 * s + h));
  unsigned troff[4][2];
  tr_offsets(lane, troff);
  dma_wait_and_sync();

  f32x16 s0, s1;
  float mloc = kNegBig;
  unsigned hh[8];
  union PF { unsigned u[4]; bf16x8 v; } pf[2];

  auto masked_tile = [&](int t) __attribute__((always_inline)) {      // block-uniform: does a block of tile t need the per-element mask arithmetic?
    const int t0 = t * kRowsPerTile;
    return t < n_tiles && (a.causal || t0 + kRowsPerTile > klen || t0 + kRowsPerTile > a.Sk);
  };
  // ---- vector work, in slices -------------------------------------------------------------------------------------------
  // start(g): mask (raw-score units) and the row maximum of block g's scores, scaled to the log2 domain
  auto start_mask = [&](f32x16& st, int key0, int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < i0 || i >= i1) continue;
      const int ki = key0 + acc_row(i, h);
      float add = 0.f;
      if (a.causal && ki > qi) add += mvs;
      if (ki >= klen) add += mvs;
      float tt = st[i] + add;
      if (ki >= a.Sk) tt = kNegBig;
      st[i] = tt;
    }
  };
  auto start_max = [&](const f32x16& st, float mx, int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i >= i0 && i < i1) mx = fmaxf(mx, st[i]);
    return mx;
  };
  auto hash_one = [&](int key0, int i) __attribute__((always_inline)) {
    if (kDrop) {
      hh[i] = mix32((pb2 + static_cast<unsigned>(key0 >> 1) + 4u * (i >> 1) + (i & 1)) ^ key2);
      pin(hh[i]);
    }
  };
  // finish(g), pair s: two probabilities, their share of the row sum, dropout, one packed bf16 pair of the P^T operand
  float psum = 0.f;
  auto finish_pair = [&](const f32x16& st, int s) __attribute__((always_inline)) {
    float p0 = __builtin_amdgcn_exp2f(fmaf(st[2 * s], sl2, mneg)), p1 = __builtin_amdgcn_exp2f(fmaf(st[2 * s + 1], sl2, mneg));
    psum += p0;
    psum += p1;
    if (kDrop) {
      p0 = (hh[s] & 0xffffu) >= thr ? p0 : 0.f;
      p1 = (hh[s] >> 16) >= thr ? p1 : 0.f;
    }
    pf[s >> 2].u[s & 3] = pack2_c(p0, p1);
    pin(pf[s >> 2].u[s & 3]);
  };
  auto rescale = [&]() __attribute__((always_inline)) {               // lazy: only when some row's block maximum exceeds the running one by more than kRescaleThr
    if (__any(mloc > m + kRescaleThr)) {
      const float m_new = fmaxf(m, mloc);
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      m = m_new;
      mneg = cfold - m_new;
      l *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
    }
  };
  // LDS reads: lane-constant address registers (kfo, troff) + compile-time offsets (buffer, 32-row block, 16-row step), so a read costs
  // no address arithmetic and nothing lane-constant is hoisted into extra registers around the loop
  const ADT_AS3 unsigned char* const lds = (const ADT_AS3 unsigned char*)smem;
  auto k_frag = [&](int s, int off) __attribute__((always_inline)) { return *reinterpret_cast<const ADT_AS3 bf16x8*>(lds + kfo[s] + off); };
  auto v_frag = [&](int db, int off) __attribute__((always_inline)) {
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ADT_AS3 v4s*)(lds + troff[db][0] + off));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ADT_AS3 v4s*)(lds + troff[db][1] + off));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  // ---- phase A: sn = K(32 rows at LDS byte offset koff_) Q^T   ||   finish of the block whose scores are in sc
  auto phase_a = [&](f32x16& sn, int koff_, const f32x16& sc) __attribute__((always_inline)) {
    bf16x8 kf[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[s] = k_frag(s, koff_);
#pragma unroll
    for (int i = 0; i < 16; ++i) sn[i] = 0.f;
    psum = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < 4) kf[s + 4] = k_frag(s + 4, koff_);
      finish_pair(sc, s);
      sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sn, 0, 0, 0);
      ADT_FENCE();
    }
    l += psum;
    pin(l);
  };
  // ---- phase B: O^T += V(32 rows at LDS byte offset voff_)^T P^T   ||   row maximum of the block whose raw scores are in sn, its hashes
  // (a tile that needs the per-element mask gets it in a pass of its own in front of the phase: one tile in sixteen in the encoder)
  auto phase_b = [&](int voff_, f32x16& sn, int key0, bool masked, auto&& dma_piece) __attribute__((always_inline)) {
    if (masked) start_mask(sn, key0, 0, 16);
    bf16x8 vt[2][4];
#pragma unroll
    for (int db = 0; db < 4; ++db) vt[0][db] = v_frag(db, voff_);
    float mx = kNegBig;
#pragma unroll
    for (int i8 = 0; i8 < 8; ++i8) {
      const int s2 = i8 >> 2, db = i8 & 3;
      if (i8 < 4) {
        vt[1][i8] = v_frag(i8, voff_ + 16 * 256);
        mx = start_max(sn, mx, 4 * i8, 4 * i8 + 4);
        pin(mx);
      }
      if (i8 == 4) { mloc = cross_half_max(mx * sl2); pin(mloc); }
      hash_one(key0, i8);
      o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[s2][db], pf[s2].v, o[db], 0, 0, 0);
      dma_piece(i8);
      ADT_FENCE();
    }
  };

  // ---- before the loop: block 0's scores, maximum and hashes
  {
    bf16x8 kf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) kf[s] = k_frag(s, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) s0[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], s0, 0, 0, 0);
    if (masked_tile(0)) start_mask(s0, 0, 0, 16);
    mloc = cross_half_max(start_max(s0, kNegBig, 0, 16) * sl2);
#pragma unroll
    for (int i = 0; i < 8; ++i) hash_one(0, i);
    rescale();
  }
  // one 64-key tile; kP = t & 1 (compile time: the tile's K / V buffers)
  auto tile_body = [&](int t, auto parity) __attribute__((always_inline)) {
    constexpr int kP = decltype(parity)::value;
    constexpr int kK = kP * 2 * kAttnTileBytes, kV = kK + kAttnTileBytes, kKn = (kP ^ 1) * 2 * kAttnTileBytes;
    const int tile0 = t * kRowsPerTile;
    ADT_STAMP(0);
    phase_a(s1, kK + 32 * 256, s0);
    ADT_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ADT_STAMP(2);
    // K(t + 2) -> this tile's K buffer, V(t + 1) -> the other V buffer: one 1-KiB piece behind an MFMA of phase B(2t)
    auto piece = [&](int i8) __attribute__((always_inline)) {
      const int j = i8 & 3;
      if (j >= kPc) return;
      if (i8 < 4) dma_piece_of(kg, a.ldk, koff, tile0 + 2 * kRowsPerTile, j, kK);
      else dma_piece_of(vg, a.ldv, voff, tile0 + kRowsPerTile, j, kKn + kAttnTileBytes);
    };
    ADT_STAMP(3);
    phase_b(kV, s1, tile0 + 32, masked_tile(t), piece);
    ADT_STAMP(4);
    rescale();
    ADT_STAMP(5);
    phase_a(s0, kKn, s1);
    ADT_STAMP(6);
    phase_b(kV + 32 * 256, s0, tile0 + kRowsPerTile, masked_tile(t + 1), [](int) {});
    ADT_STAMP(7);
    if (t + 1 < n_tiles) rescale();
  };
  for (int t = 0; t < n_tiles; t += 2) {
    tile_body(t, std::integral_constant<int, 0>{});
    if (t + 1 < n_tiles) tile_body(t + 1, std::integral_constant<int, 1>{});
  }
  const float lt = l + __shfl_xor(l, 32);                  // (the sum carries the folded 1 / (1 - p))
  const float inv = (kDrop ? a.drop.inv_keep : 1.0f) / lt;
  store_transposed(o, inv, a.out + static_cast<long>(b) * a.Sq * a.ldo + head * kDh, a.ldo, qi, a.Sq, lane);
  if (h == 0 && qi < a.Sq && a.lse) a.lse[(static_cast<long>(b) * a.H + head) * a.Sq + qi] = (m + log2f(lt) - cfold) * kLn2;
}

int launch_attn_fwd2(const AttnArgs& a, hipStream_t st) {
  static thread_local int done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  const int lds = 4 * kAttnTileBytes;
  if (done_for != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done_for = dev;
  }
  const char* wenv = getenv("ADT_ATTN_FWD_WAVES");        // A/B: 4 or 8 waves per workgroup whatever the shape; read on every call
  const bool eight = wenv ? atoi(wenv) == 8 : a.Sq > 128;
  const int kq = eight ? 256 : 128;
  const dim3 grid(static_cast<unsigned>((a.Sq + kq - 1) / kq) * a.B * a.H);      // 1-D: tile_coords() renumbers it
  if (eight) {
    if (a.drop.on()) hipLaunchKernelGGL((attn_fwd2_kernel<true, 8>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((attn_fwd2_kernel<false, 8>), grid, dim3(512), lds, st, a);
  } else {
    if (a.drop.on()) hipLaunchKernelGGL((attn_fwd2_kernel<true, 4>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((attn_fwd2_kernel<false, 4>), grid, dim3(256), lds, st, a);
  }
  ADT_HIP_TRY(hipGetLastError());
#ifdef ADT_FWD_EXPERIMENT
  if (getenv("ADT_FWD_STAMPS")) {
    unsigned long long h[32];
    ADT_HIP_TRY(hipStreamSynchronize(st));
    ADT_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fwd_stamps), sizeof(h)));
    fprintf(stderr, "fwd stamps (cycles from the first):");
    for (int i = 1; i < 16; ++i) fprintf(stderr, " %lld", static_cast<long long>(h[i] - h[0]));
    fprintf(stderr, "\n");
  }
#endif
  return ADT_OK;
}

}  // namespace adt
