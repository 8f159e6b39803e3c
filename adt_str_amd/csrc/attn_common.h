// attn_common.h -- types, LDS layouts and operand helpers shared by the attention kernels (attention.hip, attention_bwd_fused.hip).
//
// K/V (or Q/dO) tiles of rows x 256 B live in LDS under the XOR swizzle
//   off(row, chunk) = 256*row + 16*(chunk ^ (((row&3)<<2) | ((row>>2)&3)))
// which makes both the 16-byte row reads and the transposed 8-byte reads (ds_read_b64_tr_b16) bank-conflict free.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "adt_common.h"
#include "dropout.h"

namespace adt {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kDh = 128;
constexpr int kAttnThreads = 256;
constexpr int kRowsPerTile = 64;
constexpr int kAttnTileBytes = kRowsPerTile * 256;   // 16 KiB
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -1.0e30f;
constexpr float kRescaleThr = 5.0f;          // log2 units: lazy online-softmax rescale threshold

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)p));
}
__device__ __forceinline__ int swz(int row, int chunk) {
  return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
// (asm: left to itself the compiler pairs the conversions of elements 0,2 / 1,3 and re-interleaves with four more instructions)
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// accumulator registers 8s .. 8s+7 of a 32x32 tile -> the bf16 B operand of k-step s (rows 16s..16s+15 of the tile)
__device__ __forceinline__ bf16x8 acc_to_b(const f32x16& x, int s) {
  union { unsigned u[4]; bf16x8 v; } r;
  r.u[0] = pack2(x[8 * s + 0], x[8 * s + 1]); r.u[1] = pack2(x[8 * s + 2], x[8 * s + 3]);
  r.u[2] = pack2(x[8 * s + 4], x[8 * s + 5]); r.u[3] = pack2(x[8 * s + 6], x[8 * s + 7]);
  return r.v;
}
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- tile staging by LDS-DMA: 64 rows x 128 bf16 of one head; one wave-instruction (1 KiB) fills 4 rows.
// The LDS image must be lane-linear, so the swizzle sits on the SOURCE address: position p of row r
// receives the row's chunk p ^ f(r) (an involution, so reads use swz()).  Rows past n_rows are clamped
// to the last valid row: their scores are masked / their probabilities are zero, and 0 * finite = 0.
__device__ __forceinline__ void tile_dma(const unsigned short* __restrict__ base, long row_stride, int row0, int n_rows,
                                         unsigned char* tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * (4 * wave + i) + (lane >> 4);
    const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    int gr = row0 + row;
    gr = gr < n_rows ? gr : n_rows - 1;
    const unsigned short* p = base + static_cast<long>(gr) * row_stride + chunk * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)(tile + (4 * wave + i) * 1024), 16, 0, 0);
  }
}
// The same staging with the per-lane part of the source address (row inside the tile, swizzled chunk) computed once per kernel:
// off[i] = row_i * row_stride + chunk_i * 8 elements; a full tile then costs one 64-bit add per instruction.
__device__ __forceinline__ void tile_dma_offsets(long row_stride, int wave, int lane, unsigned (&off)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * (4 * wave + i) + (lane >> 4);
    const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    off[i] = static_cast<unsigned>(row * row_stride + chunk * 8);
  }
}
__device__ __forceinline__ void tile_dma_pre(const unsigned short* __restrict__ base, long row_stride, int row0, int n_rows,
                                             unsigned char* tile, int wave, int lane, const unsigned (&off)[4]) {
  if (row0 + kRowsPerTile <= n_rows) {                  // block-uniform
    const unsigned short* b0 = base + static_cast<long>(row0) * row_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b0 + off[i]),
                                       (__attribute__((address_space(3))) void*)(tile + (4 * wave + i) * 1024), 16, 0, 0);
  } else {
    tile_dma(base, row_stride, row0, n_rows, tile, wave, lane);
  }
}
__device__ __forceinline__ void dma_wait_and_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
// A operand, rows of the tile: tile row (rb*32 + lane&31), d = 16s + 8h .. +7
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* tile, int rb, int s, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + swz(rb * 32 + (lane & 31), 2 * s + (lane >> 5)));
}
// A operand, transposed tile: A[row = d = 32db + lane&31][element j] = tile[R0 + 8(j>>2) + 4h + (j&3)][d].
// tr4_issue starts the 8 transposed reads of the four 32-d blocks of one 16-row k-step; tr_wait retires
// them (one lgkmcnt(0) for all, then a scheduling fence so no MFMA is hoisted above the wait).
struct TrFrag { bf16x4 lo, hi; };
__device__ __forceinline__ void tr4_issue(const unsigned char* tile, int R0, int lane, TrFrag (&f)[4]) {
  const int i = lane & 15, g4 = (lane >> 4) & 1, h = lane >> 5;
  const int row = R0 + 4 * h + (i >> 2);
  const unsigned base = lds_off(tile) + 8 * (i & 1);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const int chunk = 4 * db + 2 * g4 + ((i & 3) >> 1);
    const unsigned a0 = base + swz(row, chunk), a1 = base + swz(row + 8, chunk);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(f[db].lo), "=&v"(f[db].hi) : "v"(a0), "v"(a1) : "memory");
  }
}
// The same reads from per-lane offsets computed once: the swizzle only involves row bits 0..3, so a lane's eight offsets (4 d-blocks
// x rows r, r + 8) inside a 16-row group are loop constants; the tile base goes in with one add, the 16-row group as an immediate.
__device__ __forceinline__ void tr_offsets(int lane, unsigned (&o)[4][2]) {
  const int i = lane & 15, g4 = (lane >> 4) & 1, h = lane >> 5, row = 4 * h + (i >> 2);
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const int chunk = 4 * db + 2 * g4 + ((i & 3) >> 1);
    o[db][0] = static_cast<unsigned>(8 * (i & 1) + swz(row, chunk));
    o[db][1] = static_cast<unsigned>(8 * (i & 1) + swz(row + 8, chunk));
  }
}
template <int kR0>
__device__ __forceinline__ void tr4_issue_at(const unsigned (&a)[4][2], TrFrag (&f)[4]) {      // a = tr_offsets + LDS address of the tile
#pragma unroll
  for (int db = 0; db < 4; ++db)
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                 : "=&v"(f[db].lo), "=&v"(f[db].hi) : "v"(a[db][0]), "v"(a[db][1]), "i"(kR0 * 256) : "memory");
}
__device__ __forceinline__ void tr_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 tr_get(const TrFrag& f) {
  bf16x8 r;
  r[0] = f.lo[0]; r[1] = f.lo[1]; r[2] = f.lo[2]; r[3] = f.lo[3];
  r[4] = f.hi[0]; r[5] = f.hi[1]; r[6] = f.hi[2]; r[7] = f.hi[3];
  return r;
}
// B operand held in registers: row `row` of a global [*, stride] matrix, d = 16s + 8h .. +7, s = 0..7
__device__ __forceinline__ void frags_from_global(const unsigned short* __restrict__ base, long row_stride, int row, int n_rows,
                                                  int lane, bf16x8 (&f)[8]) {
  const int h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < n_rows) v = *reinterpret_cast<const uint4*>(base + static_cast<long>(row) * row_stride + 16 * s + 8 * h);
    f[s] = *reinterpret_cast<bf16x8*>(&v);
  }
}
// transposed accumulator [d][x] (4 blocks of 32 d) -> bf16 rows out[x][d]; lane owns x = lane&31.
// A row is split across the half-waves (lane x: columns 8k .. 8k+3, lane x + 32: columns 8k+4 .. 8k+7 of column group k), so the
// natural store is 16 x 8 bytes per lane, and that tail is bound by the number of store instructions, not by bytes.  One
// v_permlane32_swap per dword and pair of groups (k, k + 1) hands the upper half's group k to the lower lanes and the lower half's
// group k + 1 to the upper lanes: 8 x 16-byte stores of the same bytes to the same addresses (cdna_hip_programming.md T21).
__device__ __forceinline__ void store_transposed(const f32x16 (&acc)[4], float mul, unsigned short* __restrict__ base, long row_stride,
                                                 int row, int n_rows, int lane) {
  if (row >= n_rows) return;                           // lanes x and x + 32 own the same row: they leave together
  const int h = lane >> 5;
  unsigned short* p = base + static_cast<long>(row) * row_stride + 8 * h;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      unsigned ax = pack2(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul), ay = pack2(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
      unsigned bx = pack2(acc[db][4 * g + 4] * mul, acc[db][4 * g + 5] * mul), by = pack2(acc[db][4 * g + 6] * mul, acc[db][4 * g + 7] * mul);
      const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
      const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      // lower lanes: [own k | upper's k] = columns 8k .. 8k+7; upper lanes: [lower's k+1 | own k+1] = columns 8k+8 .. 8k+15
      *reinterpret_cast<uint4*>(p + 32 * db + 8 * g) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
    }
}

// Sum over the 32 lanes of each half-wave on the DPP network (row_shr 1/2/4/8, then row_bcast:15): lanes 31 and 63 hold the sums.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, kRowMask, 0xf, false));
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);
  return dpp_add<0x142, 0xa>(v);
}
// Column sums of a transposed accumulator as store_transposed writes it (bf16-rounded acc * mul; rows >= n_rows count as 0)
// over this wave's 32 rows: lane 31 / 63 leaves them in red[32 db + 8 g + 4 h + e].  The bias gradient of the in-projection
// is the column sum of dQ | dK | dV; taking it here saves a pass over those matrices.
__device__ __forceinline__ void colsum_transposed(const f32x16 (&acc)[4], float mul, bool row_ok, int lane, float* __restrict__ red) {
  const int h = lane >> 5;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = row_ok ? __uint_as_float(static_cast<unsigned>(__builtin_bit_cast(unsigned short, static_cast<__bf16>(acc[db][i] * mul))) << 16) : 0.f;
      v = half_wave_sum(v);
      if ((lane & 31) == 31) red[32 * db + 8 * (i >> 2) + 4 * h + (i & 3)] = v;
    }
}

// sums inside the 16 lanes of a quarter-wave on the DPP network
template <int kCtrl>
__device__ __forceinline__ float dpp_row(float v) {          // v + (v of the lane kCtrl names inside the 16-lane DPP row)
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, 0xf, 0xf, false));
}
__device__ __forceinline__ float quarter_sum(float v) {      // sum over the 16 lanes of a quarter-wave, in every one of them
  v = dpp_row<0xB1>(v);                                      // quad_perm [1, 0, 3, 2]
  v = dpp_row<0x4E>(v);                                      // quad_perm [2, 3, 0, 1]
  v = dpp_row<0x141>(v);                                     // row_half_mirror
  return dpp_row<0x140>(v);                                  // row_mirror
}
struct AttnArgs {
  const unsigned short *q, *k, *v, *o, *dout;
  unsigned short *out, *dq, *dk, *dv;
  float* lse; const float* delta;
  long ldq, ldk, ldv, ldo;          // row strides (elements); gradients share the strides of their tensors
  int B, H, Sq, Sk;
  float scale, mask_value; int causal; const int* key_len;
  Drop drop;
  float* cs_dq;        // column sums of dQ per (batch, 128-query block): [B * ceil(Sq/128)][H*128], null: off
  unsigned* keep_bits; // dropout: the forward's keep decisions as bits (see keep_bits_* below), null: off
  int bits_nq, bits_nk;   // 32-query slices / 32-key blocks per (batch, head) in keep_bits
};

// ---- keep bits: the dropout decisions of the forward, handed to the one-kernel backward instead of being hashed again there.
// The forward holds S^T with the QUERY on the lane, so the compare that decides accumulator register i of a 32-key block is, as a
// 64-bit lane mask, exactly { bits 0..31: queries 0..31 of the wave's slice for key acc_row(i, 0); bits 32..63: the same queries for key
// acc_row(i, 1) } -- one 32-bit word per key with a bit per query, which is what the backward (KEY on the lane) wants in a vector register.
// Layout: [batch * head][32-query slice][32-key block][32 words], word 2 i + h <-> key acc_row(i, h) of the block; the forward stores the
// compare results from its scalar registers (s_store_dwordx4: 16 bytes = registers 2 s, 2 s + 1), the backward's lane for key r loads word
// keep_bits_word(r) and tests bit q.  Slices / blocks are padded to whole 256s of queries / keys so that neither kernel needs a bounds test;
// blocks past the key range and slices past the query range hold garbage that is never used (their probabilities are zero).
__host__ __device__ __forceinline__ int keep_bits_word(int key) { return 2 * ((key & 3) + 4 * (key >> 3)) + ((key >> 2) & 1); }
inline int keep_bits_nq(int q_len) { return ((q_len + 255) / 256) * 8; }
inline int keep_bits_nk(int k_len) { return ((k_len + 255) / 256) * 8; }

// Tile coordinates of this workgroup.  The grids are 1-D: nx tiles (query or key blocks) per (batch, head) times ny = B * H.
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one), so ids are renumbered to give every XCD a
// CONTIGUOUS range of logical tiles, x fastest: all blocks of one (batch, head) then sit on one XCD at about the same time, and
// that XCD's L2 fetches the head's K / V (or Q / dO) once instead of eight L2s fetching it once each.  Placement only changes
// speed, never results; the map is a bijection for any grid size (same renumbering as the GEMMs).
struct TileXY { int x, y; };
__device__ __forceinline__ TileXY tile_coords(int nx) {
  const int n = gridDim.x, id = blockIdx.x;
  const int q = n >> 3, r = n & 7, xcd = id & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  return TileXY{logical % nx, logical / nx};
}

// additive mask of the reference (model.py:173-181): causal and key-padding contributions add up
__device__ __forceinline__ float mask_add(const AttnArgs& a, int qi, int ki, int klen) {
  float m = 0.f;
  if (a.causal && ki > qi) m += a.mask_value;
  if (ki >= klen) m += a.mask_value;
  return m;
}

// attention_fwd.hip: the software-pipelined forward (tiled shapes; the one-query decode kernel stays in attention.hip)
int launch_attn_fwd2(const AttnArgs& a, hipStream_t st);
// attention_bwd_fused.hip: the one-kernel backward (5 products, ordered dQ hand-off); its workspace region and launcher
// arguments of the one-kernel backward forms (attention_bwd_fused.hip, attention_bwd_fused8.hip)
struct FusedArgs {
  AttnArgs a;
  const float* stats;                 // [B*H][ns][2][32]: nl of the slice's queries, then nd (attn_bwd_stats_kernel)
  float* part;                        // [B*H][ns][nkb][4 waves][1024]: every key block's dQ^T tile of every slice (nkb > 1)
  unsigned* flags;                    // [B*H][ns][nkb][4 waves] + 4 words: [0] of the tail = number of waves that gave up waiting
  unsigned* sched; unsigned sched_total[8];
  unsigned* giveup_host;              // pinned host word of this device (attn_bwd_giveup_word): every give-up is also counted there, where the host sees it without a sync
  int nkb, ns;
  unsigned long long* stamps;         // experiment build, kDbg & 32: cycle stamps of one wave's phases in one slice
  int dbg;                            // timing experiments only (ADT_FB_DBG): 1 no hand-off, 2 no dQ product, 4 no dV / dK products, 8 no S / dP chains
};

size_t attn_bwd_fused_workspace_bytes(const adt_attn_desc* d);
int attn_bwd_fused_prepare(const adt_attn_desc* d, const AttnArgs& a, void* ws, size_t ws_bytes, hipStream_t st, FusedArgs* out);
int attn_bwd_fused_check(const FusedArgs& fa, hipStream_t st);
// A wave of the one-kernel backward that gives up waiting for a dQ tile (kFbSpinLimit) leaves an incomplete dQ.  Besides the per-launch
// counter in the workspace it bumps a pinned, device-mapped HOST word (one per device, allocated on first use), which
// attn_bwd_fused_prepare reads before every later launch and adt_attn_bwd_giveups reads on request: the failure surfaces as ADT_EHIP at
// the next adt_attn_bwd call / at the end of the training step without an environment variable and without a device synchronisation.
int attn_bwd_giveup_word(unsigned** host_word, unsigned** device_alias);
__device__ __forceinline__ void attn_bwd_report_giveup(const FusedArgs& fa, long tail_word) {
  atomicAdd(fa.flags + tail_word, 1u);
  if (fa.giveup_host) __hip_atomic_fetch_add(fa.giveup_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int launch_attn_bwd_fused8(const adt_attn_desc* d, const AttnArgs& a, void* ws, size_t ws_bytes, hipStream_t st);
int launch_attn_bwd_fused(const adt_attn_desc* d, const AttnArgs& a, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace adt
