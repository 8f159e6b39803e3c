// attention_fwd.hip -- K4 forward, software-pipelined and (for more than 128 queries) persistent: the flash-style forward of attention.hip's
// header (lane <-> query, S^T = K Q^T, O^T += V^T P^T, v_mfma_f32_32x32x16_bf16, a wave = 32 queries) with the two products of neighbouring
// 32-key blocks overlapped inside each wave.  Stands behind nn.MultiheadAttention of the reference's encoder / decoder layers
// (model.py:118-127, 159-168).  Two workgroup forms, same bits: 4 waves x 2 workgroups per CU (at most 128 queries: the decoder), and ONE
// 8-wave workgroup per CU that walks its (batch, head, 256-query block) items, the next item's operands arriving under the current one's tiles.
//
// A wave's own chain per 32-key block is   K reads -> 8 MFMAs (S) -> max -> exp / sum / dropout / pack -> V^T reads -> 8 MFMAs (O):
// run in that order (the first forward kernel, rounds 1-3) it costs ~1800 cycles for 512 cycles of MFMA, and the second wave of the SIMD
// only hides part of it.  Here block g + 1's score product runs UNDER block g's exponentials, and block g's value product under block
// g + 1's row maximum, block g's row sum and block g + 1's dropout hashes:
//     phase A(g):  S(g+1) = K(g+1) Q^T          ||  finish(g): p = exp2(s * scale * log2e - m) (kept in place of s), dropout, bf16 pack
//     phase B(g):  O^T += V(g)^T P(g)^T         ||  row sum of block g; start(g+1): row maximum over the 32 keys; hashes of block g + 1
//     then the lazy rescale of (m, l, O) if some row's maximum rose by more than kRescaleThr (after block g's product went into O).
// Each phase is cut into eight slices of one MFMA + one eighth of the vector work, separated by scheduling fences, so that the
// interleaving in the source is the interleaving in the binary; the compiler still inserts every wait and hazard no-op (builtin MFMAs,
// LDS reads and conversions; the only asm with an instruction in it is the LDS-DMA issue, see dma1k).  Every slice's result is pinned
// (pin()): it is consumed a phase later, and the optimiser would otherwise sink the whole of finish(g) into the block that reads P.
// Tiles of 64 keys arrive by LDS-DMA into two K and two V buffers; ONE barrier per tile, between A(2t) and B(2t): K(t) is dead there
// (its rows 32..63 were just used) and V(t - 1) has been dead for two phases, so K(t + 2) and V(t + 1) are issued right behind it
// (one 1-KiB piece behind each MFMA of phase B(2t)), and K(t + 1), V(t) -- issued one tile earlier -- are made visible by it, one phase
// before their first use.  A tile that needs the per-element mask (causal, key padding, the key range's end) gets it in a pass of its own.
// Dropout keeps the element iff its 16-bit hash half >= thr (dropout.h); the 1 / (1 - p) factor is folded into the exponent
// (exp2(x + log2(1 / (1 - p)))), so a dropped element costs a compare and a select, and the row sum is rescaled once at the end.
// Measurements and the order in which the pieces paid: DESIGN.md section 8; profiles/r04/attn_fwd*.txt.
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
#define ADT_STAMP_AT(K)                                                                                       \
  do {                                                                                                        \
    if (stamp_on) {                                                                                           \
      const unsigned long long tnow = __builtin_amdgcn_s_memtime();                                           \
      if (lane == 0) g_fwd_stamps[K] = tnow;                                                                  \
    }                                                                                                         \
  } while (0)
#else
#define ADT_STAMP(K) do { } while (0)
#define ADT_STAMP_AT(K) do { } while (0)
#endif

// kWaves = 8: one workgroup of 256 queries per CU (two waves per SIMD): a K / V tile is fetched once for eight waves -- the LDS-DMA path
// delivers ~33 B/clk per CU (the same bound as the GEMM's operand delivery), and a wave's DMA instruction costs it ~80 issue cycles.
// kWaves = 4: 128 queries, two workgroups per CU: the shapes with at most 128 queries (decoder self- and cross-attention).
// kBits (dropout only): every keep decision also goes to AttnArgs::keep_bits for the backward (attn_common.h keep_bits_*).  The compare that
// feeds the select IS the word the backward wants (a 64-bit lane mask in a scalar register pair), so it leaves by a scalar store:
// s_store_dwordx4 of the two compares of a finish pair (gfx950 executes scalar stores -- tools/probe/probe_sstore.hip checks that and prices
// one at ~7 instruction slots of its wave; the dirty lines leave the scalar cache at s_dcache_wb, before the wave ends).
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
template <bool kDrop, int kWaves, bool kBits = false>
__global__ __launch_bounds__(64 * kWaves, 8 / kWaves) void attn_fwd2_kernel(AttnArgs a, int n_items) {
  static_assert(kDrop || !kBits, "keep bits exist with dropout only");
  constexpr bool kPersist = kWaves == 8;   // one workgroup per CU walks its share of the (batch, head, query block) items
  constexpr int kQ = 32 * kWaves;          // queries per workgroup
  constexpr int kPc = 16 / kWaves;         // 1-KiB DMA pieces of a 16-KiB tile per wave
  constexpr int kQNext = 4 * kAttnTileBytes;   // kPersist: LDS byte offset of the next item's Q rows (256 x 256 B, the tiles' swizzled image)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [2][K tile | V tile] (+ kPersist: next Q)
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef ADT_FWD_EXPERIMENT
  const bool stamp_on = blockIdx.x == gridDim.x / 2 + 3 && wave == 1;
  ADT_STAMP_AT(16);
#endif
  const int nx = (a.Sq + kQ - 1) / kQ;
  const float sl2 = a.scale * kLog2e;
  const float mvs = a.mask_value / a.scale;                  // the additive mask in raw-score units
  const float cfold = kDrop ? __log2f(a.drop.inv_keep) : 0.f;
  const unsigned thr = a.drop.thr;
  const unsigned key2 = mix32(a.drop.key);
  // ---- the item in hand: (batch, head, query block) of virtual block id v (the grid's renumbering of attn_common.h tile_coords, taken
  // over n_items ids: ids v, v + 8, ... of one XCD are consecutive query blocks / heads, so an XCD's L2 fetches a head's K / V once)
  struct Item { int b, head, q0; };
  auto item_of = [&](int v) __attribute__((always_inline)) {
    const int q8 = n_items >> 3, r8 = n_items & 7, xcd = v & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
    const int y = logical / nx;
    return Item{y / a.H, y % a.H, (logical % nx) * kQ};
  };
  int b, head, qi, klen;
  const unsigned short *kg, *vg;
  unsigned pb2;
  const unsigned* bits_row = nullptr;      // kBits: keep words of (this item's head, this wave's 32-query slice, key block 0)
  auto set_item = [&](const Item& it) __attribute__((always_inline)) {
    b = it.b; head = it.head;
    qi = it.q0 + wave * 32 + r;
    kg = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * kDh;
    vg = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * kDh;
    klen = __builtin_amdgcn_readfirstlane(a.key_len ? a.key_len[b] : a.Sk);
    // dropout index of (row, key) is row * Sk2 + key (Sk2 = Sk rounded up to even; dropout.h): pair = row * Sk2 / 2 + key / 2
    pb2 = static_cast<unsigned>(((static_cast<uint64_t>(b) * a.H + head) * a.Sq + qi) * ((a.Sk + 1) >> 1)) + 2u * h;
    if (kBits) bits_row = a.keep_bits + ((static_cast<long>(b) * a.H + head) * a.bits_nq + ((it.q0 >> 5) + wave)) * a.bits_nk * 32;
  };
  int item = blockIdx.x;
  set_item(item_of(item));

  bf16x8 qf[8];
  f32x16 o[4];
  float m, mneg, l;

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
    int row = 4 * (kPc * wave + j) + (lane >> 4);
    asm volatile("" : "+v"(row));          // (formed at the use: see dma_piece_of)
    const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    int gr = row0 + row;
    gr = gr < a.Sk ? gr : a.Sk - 1;
    return static_cast<long>(gr) * ld + chunk * 8;
  };
  // one piece of tile `row0 / 64` of K (ld = ldk) or V into the 16-KiB buffer at LDS byte offset buf
  auto dma_piece_of = [&](const unsigned short* g, long ld, const unsigned (&off)[kPc], int row0, int j, int buf) __attribute__((always_inline)) {
    const unsigned dst = lds_base + buf + (kPc * wave + j) * 1024;
    unsigned oj = off[j];
    asm volatile("" : "+v"(oj));           // (the 64-bit source address is formed here, not hoisted out of the tile loop into registers that then spill)
    if (row0 + kRowsPerTile <= a.Sk) dma1k(g + static_cast<long>(row0) * ld + oj, dst);          // block-uniform
    else if (row0 < a.Sk) dma1k(g + clamp_off(row0, j, ld), dst);
  };
  // the first three tiles of an item: K(0), V(0) into buffers 0, K(1) into K buffer 1
  auto first_tiles = [&](const unsigned short* kgi, const unsigned short* vgi, int j) __attribute__((always_inline)) {
    dma_piece_of(kgi, a.ldk, koff, 0, j, 0);
    dma_piece_of(vgi, a.ldv, voff, 0, j, kAttnTileBytes);
    dma_piece_of(kgi, a.ldk, koff, kRowsPerTile, j, 2 * kAttnTileBytes);
  };
  // kPersist: piece j (0..7) of this wave's OWN 32 query rows of item `it` into the next-Q region (rows past Sq clamp to the last one:
  // never stored).  A wave only ever reads its own rows back, so its own vmcnt(0) is all the synchronisation they need.
  auto q_piece = [&](const Item& it, int j) __attribute__((always_inline)) {
    int row = 4 * (8 * wave + j) + (lane >> 4);
    asm volatile("" : "+v"(row));          // (as above: eight pieces' addresses of the NEXT item are loop invariants)
    const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    int gr = it.q0 + row;
    gr = gr < a.Sq ? gr : a.Sq - 1;
    dma1k(a.q + (static_cast<long>(it.b) * a.Sq + gr) * a.ldq + it.head * kDh + chunk * 8, lds_base + kQNext + (8 * wave + j) * 1024);
  };
  // lane constants of the LDS reads: K rows (16-byte pieces, chunk 2s + h) and V^T (transposed 8-byte pieces)
  unsigned kfo[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) kfo[s] = static_cast<unsigned>(swz(r, 2 * s + h));
  unsigned troff[4][2];
  tr_offsets(lane, troff);
  const ADT_AS3 unsigned char* const lds = (const ADT_AS3 unsigned char*)smem;
  auto q_from_lds = [&]() __attribute__((always_inline)) {
    int base = kQNext + wave * 32 * 256;
    asm volatile("" : "+s"(base));         // (eight adds here instead of eight more address registers across the tile loop)
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const ADT_AS3 bf16x8*>(lds + base + kfo[s]);
  };
  {
    const Item it0 = item_of(item);
    if (kPersist) {
#pragma unroll
      for (int j = 0; j < 8; ++j) q_piece(it0, j);
    } else {
      frags_from_global(a.q + static_cast<long>(b) * a.Sq * a.ldq + head * kDh, a.ldq, qi, a.Sq, lane, qf);
    }
#pragma unroll
    for (int j = 0; j < kPc; ++j) first_tiles(kg, vg, j);
    dma_wait_and_sync();
    if (kPersist) q_from_lds();
  }

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
  // finish(g), pair s: two probabilities (left in place of the scores: phase B adds them up, it has the issue slots), dropout, one packed
  // bf16 pair of the P^T operand
  auto finish_pair = [&](f32x16& st, int s, const unsigned* bits_blk) __attribute__((always_inline)) {      // bits_blk: the block's 32 keep words (kBits)
    float p0 = __builtin_amdgcn_exp2f(fmaf(st[2 * s], sl2, mneg)), p1 = __builtin_amdgcn_exp2f(fmaf(st[2 * s + 1], sl2, mneg));
    st[2 * s] = p0;
    st[2 * s + 1] = p1;
    if (kDrop) {
      const bool k0 = (hh[s] & 0xffffu) >= thr, k1 = (hh[s] >> 16) >= thr;
      p0 = k0 ? p0 : 0.f;
      p1 = k1 ? p1 : 0.f;
      if (kBits) {                           // registers 2 s, 2 s + 1 of the block: words 4 s .. 4 s + 3
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(k0), m1 = __builtin_amdgcn_ballot_w64(k1);
        const u32x4s w = {static_cast<unsigned>(m0), static_cast<unsigned>(m0 >> 32), static_cast<unsigned>(m1), static_cast<unsigned>(m1 >> 32)};
        asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(w), "s"(bits_blk), "i"(16 * s));
      }
    }
    pf[s >> 2].u[s & 3] = pack2_c(p0, p1);
    pin(pf[s >> 2].u[s & 3]);
  };
  unsigned long long grow = 0;         // lanes whose block maximum exceeds the running one by more than kRescaleThr (taken inside phase B)
  auto rescale = [&]() __attribute__((always_inline)) {               // lazy: only when some row's maximum rose by more than kRescaleThr
    if (grow != 0) {
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
  auto k_frag = [&](int s, int off) __attribute__((always_inline)) { return *reinterpret_cast<const ADT_AS3 bf16x8*>(lds + kfo[s] + off); };
  auto v_frag = [&](int db, int off) __attribute__((always_inline)) {
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ADT_AS3 v4s*)(lds + troff[db][0] + off));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ADT_AS3 v4s*)(lds + troff[db][1] + off));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  // ---- phase A: sn = K(32 rows at LDS byte offset koff_) Q^T   ||   finish of the block whose scores are in sc
  // The first operands of a phase are read during the last slices of the phase before it (kf[0..3] by phase B / block0, vpre by phase A
  // when no barrier lies between): the LDS round trip is not at the head of the MFMA chain.
  bf16x8 kf[8], vpre[4];
  auto phase_a = [&](f32x16& sn, int koff_, f32x16& sc, int vnext, int kblk) __attribute__((always_inline)) {      // vnext: LDS offset of the next phase B's V rows, -1: behind a barrier; kblk: 32-key block index of sc
    const unsigned* bits_blk = kBits ? bits_row + kblk * 32 : nullptr;
#pragma unroll
    for (int i = 0; i < 16; ++i) sn[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < 4) kf[s + 4] = k_frag(s + 4, koff_);
      if (s >= 6 && vnext >= 0) { vpre[2 * (s - 6)] = v_frag(2 * (s - 6), vnext); vpre[2 * (s - 6) + 1] = v_frag(2 * (s - 6) + 1, vnext); }
      finish_pair(sc, s, bits_blk);
      sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sn, 0, 0, 0);
      ADT_FENCE();
    }
  };
  // ---- phase B: O^T += V(32 rows at LDS byte offset voff_)^T P^T   ||   row maximum of the block whose raw scores are in sn, its hashes
  // (a tile that needs the per-element mask gets it in a pass of its own in front of the phase: one tile in sixteen in the encoder)
  auto phase_b = [&](int voff_, bool v_ready, int knext, const f32x16& sp, f32x16& sn, int key0, bool masked, auto&& dma_piece) __attribute__((always_inline)) {
    // v_ready: the first four V fragments were read by the phase A before; knext: LDS offset of the next phase A's K rows, -1: none
    if (masked) start_mask(sn, key0, 0, 16);
    bf16x8 vt[2][4];
#pragma unroll
    for (int db = 0; db < 4; ++db) vt[0][db] = v_ready ? vpre[db] : v_frag(db, voff_);
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
      if (i8 == 5) { grow = __builtin_amdgcn_ballot_w64(mloc > m + kRescaleThr); asm volatile("" : "+s"(grow)); }
      if (i8 >= 6 && knext >= 0) { kf[2 * (i8 - 6)] = k_frag(2 * (i8 - 6), knext); kf[2 * (i8 - 6) + 1] = k_frag(2 * (i8 - 6) + 1, knext); }
      hash_one(key0, i8);
      l += sp[2 * i8];                 // block g's row sum (un-dropped probabilities)
      l += sp[2 * i8 + 1];
      pin(l);
      o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[s2][db], pf[s2].v, o[db], 0, 0, 0);
      dma_piece(i8);
      ADT_FENCE();
    }
  };

  auto block0 = [&]() __attribute__((always_inline)) {      // an item's first block: scores, maximum, hashes (before the pipeline starts)
#pragma unroll
    for (int s = 0; s < 8; ++s) kf[s] = k_frag(s, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) s0[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], s0, 0, 0, 0);
    if (masked_tile(0)) start_mask(s0, 0, 0, 16);
    mloc = cross_half_max(start_max(s0, kNegBig, 0, 16) * sl2);
    grow = __builtin_amdgcn_ballot_w64(mloc > m + kRescaleThr);
#pragma unroll
    for (int i = 0; i < 8; ++i) hash_one(0, i);
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[s] = k_frag(s, 32 * 256);      // phase A(0)'s first operands
    rescale();
  };
  // kPersist: the next item's operands are fetched under this item's tiles
  Item nxt{0, 0, 0};
  bool has_next = false;
  int qn = 0;                              // next-Q pieces issued so far (8 per wave)
  auto q_hook = [&](int i8) __attribute__((always_inline)) {      // behind MFMAs 0 and 4 of a tile's second phase B
    if (kPersist && (i8 & 3) == 0 && has_next && qn < 8) { q_piece(nxt, qn); ++qn; }
  };
  // one 64-key tile; kP = t & 1 (compile time: the tile's K / V buffers)
  auto tile_body = [&](int t, auto parity) __attribute__((always_inline)) {
    constexpr int kP = decltype(parity)::value;
    constexpr int kK = kP * 2 * kAttnTileBytes, kV = kK + kAttnTileBytes, kKn = (kP ^ 1) * 2 * kAttnTileBytes;
    const int tile0 = t * kRowsPerTile;
    ADT_STAMP(0);
    phase_a(s1, kK + 32 * 256, s0, -1, 2 * t);
    ADT_STAMP(1);
    // (tile 0: K(1) and V(0) were waited for before the item started; what is still in flight is the previous item's output)
    if (t > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    phase_b(kV, false, kKn, s0, s1, tile0 + 32, masked_tile(t), piece);
    ADT_STAMP(4);
    rescale();
    ADT_STAMP(5);
    phase_a(s0, kKn, s1, kV + 32 * 256, 2 * t + 1);
    ADT_STAMP(6);
    phase_b(kV + 32 * 256, true, kKn + 32 * 256, s1, s0, tile0 + kRowsPerTile, masked_tile(t + 1), q_hook);
    ADT_STAMP(7);
    rescale();
  };
  // the last tile: no later tile to fetch or to start, and no second block when at most 32 of its keys exist (986 keys: block 31 of 32).
  // Run-time buffer offsets (an add per LDS read), once per workgroup.
  auto finish_only = [&](f32x16& sc, int kblk) __attribute__((always_inline)) {
    const unsigned* bits_blk = kBits ? bits_row + kblk * 32 : nullptr;
#pragma unroll
    for (int s = 0; s < 8; ++s) finish_pair(sc, s, bits_blk);
  };
  auto last_tile = [&](int t) __attribute__((always_inline)) {
    int kK = (t & 1) * 2 * kAttnTileBytes;
    asm volatile("" : "+s"(kK));           // (as in q_from_lds)
    const int kV = kK + kAttnTileBytes;
    const int tile0 = t * kRowsPerTile;
    const bool second = a.Sk - tile0 > 32;               // block-uniform
    if (second) phase_a(s1, kK + 32 * 256, s0, -1, 2 * t);
    else finish_only(s0, 2 * t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // With an even number of tiles this one sits in buffers 1: K(t) is dead behind the barrier and buffers 0 have been for a tile, so the
    // next item's K(0), V(0), K(1) go where its own pipeline expects them (an odd number: they are fetched after the tile, see below)
    const unsigned short* kgn = a.k + static_cast<long>(nxt.b) * a.Sk * a.ldk + nxt.head * kDh;
    const unsigned short* vgn = a.v + static_cast<long>(nxt.b) * a.Sk * a.ldv + nxt.head * kDh;
    auto next_hook = [&](int i8) __attribute__((always_inline)) {
      if (kPersist && has_next && (t & 1) && i8 < kPc) first_tiles(kgn, vgn, i8);
    };
    phase_b(kV, false, -1, s0, s1, tile0 + 32, masked_tile(t), next_hook);      // (without a second block: its maximum of s1 is not used)
    if (second) {
      rescale();
      finish_only(s1, 2 * t + 1);
      phase_b(kV + 32 * 256, false, -1, s1, s0, tile0 + kRowsPerTile, false, [](int) {});
    }
  };
  while (true) {
    const int next_id = item + static_cast<int>(gridDim.x);
    has_next = kPersist && next_id < n_items;
    if (has_next) nxt = item_of(next_id);
    qn = 0;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    m = kNegBig; mneg = cfold - kNegBig; l = 0.f;
    ADT_STAMP_AT(17);
    block0();
    ADT_STAMP_AT(18);
    for (int t = 0; t + 1 < n_tiles; t += 2) {
      tile_body(t, std::integral_constant<int, 0>{});
      if (t + 2 < n_tiles) tile_body(t + 1, std::integral_constant<int, 1>{});
    }
    ADT_STAMP_AT(19);
    last_tile(n_tiles - 1);
    ADT_STAMP_AT(20);
    if (has_next) {
      if (!((n_tiles - 1) & 1)) {          // odd number of tiles: the buffers were in use until now
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPc; ++j)
          first_tiles(a.k + static_cast<long>(nxt.b) * a.Sk * a.ldk + nxt.head * kDh, a.v + static_cast<long>(nxt.b) * a.Sk * a.ldv + nxt.head * kDh, j);
      }
      for (; qn < 8; ++qn) q_piece(nxt, qn);          // (short key ranges: what the tiles did not get to)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      q_from_lds();                        // Q is dead after the last score product: the next item's rows, before this item's output goes out
    }
    const float lt = l + __shfl_xor(l, 32);                  // (the sum carries the folded 1 / (1 - p))
    const float inv = (kDrop ? a.drop.inv_keep : 1.0f) / lt;
    store_transposed(o, inv, a.out + static_cast<long>(b) * a.Sq * a.ldo + head * kDh, a.ldo, qi, a.Sq, lane);
    if (h == 0 && qi < a.Sq && a.lse) a.lse[(static_cast<long>(b) * a.H + head) * a.Sq + qi] = (m + log2f(lt) - cfold) * kLn2;
    ADT_STAMP_AT(21);
    if (!has_next) break;
    item = next_id;
    set_item(nxt);
  }
  if (kBits) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");      // the keep words leave the scalar cache
}

int launch_attn_fwd2(const AttnArgs& a, hipStream_t st) {
  static thread_local int done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  const int lds = 4 * kAttnTileBytes, lds8 = 8 * kAttnTileBytes;          // the persistent form adds the next item's 256 Q rows
  if (done_for != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<true, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd2_kernel<true, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
    done_for = dev;
  }
  const char* wenv = getenv("ADT_ATTN_FWD_WAVES");        // A/B: 4 or 8 waves per workgroup whatever the shape; read on every call
  const bool eight = wenv ? atoi(wenv) == 8 : a.Sq > 128;
  const int kq = eight ? 256 : 128;
  const int n_items = ((a.Sq + kq - 1) / kq) * a.B * a.H;      // 1-D ids, renumbered per XCD inside the kernel
  if (eight) {
    int n_cu = 0;
    if (int rc = device_cu_count(&n_cu)) return rc;
    const dim3 grid(static_cast<unsigned>(n_items < n_cu ? n_items : n_cu));          // persistent: one workgroup per CU
    if (a.drop.on() && a.keep_bits) hipLaunchKernelGGL((attn_fwd2_kernel<true, 8, true>), grid, dim3(512), lds8, st, a, n_items);
    else if (a.drop.on()) hipLaunchKernelGGL((attn_fwd2_kernel<true, 8>), grid, dim3(512), lds8, st, a, n_items);
    else hipLaunchKernelGGL((attn_fwd2_kernel<false, 8>), grid, dim3(512), lds8, st, a, n_items);
  } else {
    const dim3 grid(static_cast<unsigned>(n_items));
    if (a.drop.on() && a.keep_bits) hipLaunchKernelGGL((attn_fwd2_kernel<true, 4, true>), grid, dim3(256), lds, st, a, n_items);
    else if (a.drop.on()) hipLaunchKernelGGL((attn_fwd2_kernel<true, 4>), grid, dim3(256), lds, st, a, n_items);
    else hipLaunchKernelGGL((attn_fwd2_kernel<false, 4>), grid, dim3(256), lds, st, a, n_items);
  }
  ADT_HIP_TRY(hipGetLastError());
#ifdef ADT_FWD_EXPERIMENT
  if (getenv("ADT_FWD_STAMPS")) {
    unsigned long long h[32];
    ADT_HIP_TRY(hipStreamSynchronize(st));
    ADT_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fwd_stamps), sizeof(h)));
    fprintf(stderr, "fwd stamps (cycles from the first):");
    for (int i = 1; i < 16; ++i) fprintf(stderr, " %lld", static_cast<long long>(h[i] - h[0]));
    fprintf(stderr, "\n   workgroup: entry 0, operands landed %lld, loop starts %lld, last tile starts %lld, last tile done %lld, stored %lld\n",
            static_cast<long long>(h[17] - h[16]), static_cast<long long>(h[18] - h[16]), static_cast<long long>(h[19] - h[16]),
            static_cast<long long>(h[20] - h[16]), static_cast<long long>(h[21] - h[16]));
  }
#endif
  return ADT_OK;
}

}  // namespace adt
