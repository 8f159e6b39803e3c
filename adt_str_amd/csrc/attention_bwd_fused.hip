// attention_bwd_fused.hip -- K4 backward as ONE kernel with the five algorithmic products (S, dP, dV, dK, dQ).
//
// Stands behind the backward of nn.MultiheadAttention inside the reference's transformer layers (model.py:118-127 encoder
// self-attention, :159-181 decoder self- / cross-attention with the additive -1e4 masks).  The two-kernel backward in attention.hip
// recomputes S = Q K^T and dP = dO V^T in both of its kernels (7 products executed for 5 algorithmic, and every dropout decision
// hashed twice); here they are computed once:
//
//   workgroup = 4 waves (one per SIMD, the whole 512-register file each) = 256 keys of one (batch, head);
//   wave w owns keys 64 w .. 64 w + 63 and keeps dK^T and dV^T of them in 256 accumulator registers while the workgroup sweeps
//   the head's queries in slices of 32 rows.  Per slice and 32-key block, with the KEY ON THE LANE:
//       S' = Q K^T - lse / scale   (A = Q rows of the slice from LDS, B = K rows of the workgroup's resident K image; the row
//                                    constant is the chain's initial accumulator, so P = 2^(scale log2e S') needs no subtraction)
//       dP = dO V^T                (A = dO rows from LDS, B = V fragments held in registers for the whole kernel)
//       P, dS = P (dP keep - delta) in the accumulator registers, which are exactly the B operands of
//       dV^T += dO^T P,  dK^T += Q^T dS   (A = transposed LDS reads of the same dO / Q images)
//   dS^T goes to LDS once (bf16 [key][32 q]); after the slice's barrier wave w computes the d-block w of
//       dQ^T[d][q] = K^T dS^T over the workgroup's 256 keys (A = transposed reads of the K image, B = transposed reads of dS^T).
//   dQ is summed over the ceil(Sk / 256) key-block workgroups of the (batch, head) by an ORDERED HAND-OFF: key block k adds its tile
//   to the running sum block k - 1 published (fp32, write-through stores + one flag per wave; cdna_hip_programming.md Guideline 16
//   R1), the last one scales, rounds and stores dQ.  The order is fixed, so the result is bitwise reproducible; no float atomics.
//
// Workgroups take their (batch, head, key block) from the XCD group's ticket counter (the persistent GEMMs' counters), in logical
// order, so the nkb key blocks of a head hold CONSECUTIVE tickets of one counter.  The fan-in makes key block jr % nkb wait for the tiles of
// all the head's other key blocks -- lower AND higher tickets -- so progress needs the head's nkb workgroups resident together: with the
// lowest unfinished head's workgroups always the first to be dispatched, that holds whenever nkb <= the workgroups one XCD group can hold
// (one per CU).  The launcher's caller (attention.hip fused_can_run) only takes this path for nkb <= half of that, and never under stream
// capture; a wave that still waits 2^24 polls gives up and counts itself in the word behind the flags AND in a pinned host word of the
// device: the next adt_attn_bwd call (or adt_attn_bwd_giveups at the end of a training step) turns that into ADT_EHIP instead of a silently
// incomplete dQ; ADT_ATTN_BWD_CHECK=1 (the tests) checks synchronously after every launch.
// delta = rowsum(O * dO) and -lse / scale are prepared per query by a small kernel in front (attn_bwd_stats_kernel).
#include <mutex>
#include <type_traits>

#include "attn_common.h"

namespace adt {

constexpr int kFbThreads = 256;
constexpr int kFbKeys = 256;                        // keys per workgroup
constexpr int kFbSlice = 32;                        // queries per step
constexpr int kFbKimg = kFbKeys * 256;              // K rows of the workgroup's keys (swizzled 256-byte rows)
constexpr int kFbX = kFbKeys * 64;                  // dS^T of one slice: [key][32 q] bf16
constexpr int kFbTile = 2 * kFbSlice * 256;         // Q rows | dO rows of one slice
constexpr int kFbOffX = kFbKimg;
constexpr int kFbOffT = kFbOffX + kFbX;
constexpr int kFbOffS = kFbOffT + 2 * kFbTile;      // per slice -lse / scale [32] | -delta [32]
constexpr int kFbOffStash = kFbOffS + 2 * 256;      // running dQ^T sum of the slice this workgroup is reducing (4 KiB per wave)
constexpr int kFbOffLand = kFbOffStash + 4 * 4096;  // landing zone of the key block tile that is added next (4 KiB per wave, LDS-DMA)
constexpr int kFbOffFlag = kFbOffLand + 4 * 4096;
constexpr int kFbOffBits = kFbOffFlag + 16;         // keep bits of the slice (kDrop == 2): per ring slot 4 waves x 256 B (lane l < 32: the word of key l of the
                                                    // wave's block 0, l >= 32: of key l - 32 of block 1)
constexpr int kFbLds = kFbOffBits + 2 * 1024;       // 150,032 B
constexpr unsigned kFbSpinLimit = 1u << 24;         // polls of ~0.3 us each before a wave gives up (and reports it)


// Per (batch, head, query), queries padded to whole slices ({-1e30, 0}: P = 0 there):
//   nl = -lse / scale [+ log2(1 / (1 - p)) / (scale log2 e) with dropout]: the INITIAL ACCUMULATOR of the score chain, so that
//        P [/ (1 - p)] = 2^((S + nl) scale log2 e) needs no per-element subtraction (and no per-element keep scale);
//   nd = -delta [x (1 - p) with dropout]:                 dS = P_dropped dP + (P / (1 - p)) nd  (= P (dP keep / (1 - p) - delta)).
// The same launch zeroes the fan-in's flags (one item per (batch, head, slice) clears that slice's nkb x 4 words, the first item the give-up
// counter behind them): the backward kernel follows it in stream order, so no separate memset launch is needed.
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(AttnArgs a, float* __restrict__ stats, int ns, unsigned* __restrict__ flags, int nkb) {
  const int tid = threadIdx.x, c = tid & 15;
  const int sqp = ns * kFbSlice;
  const long item = static_cast<long>(blockIdx.x) * 16 + (tid >> 4);          // (b, q, head), head fastest: one row's heads are neighbours
  if (item >= static_cast<long>(a.B) * sqp * a.H) return;                     // whole quarter-waves leave together
  const int head = static_cast<int>(item % a.H);
  const long bq = item / a.H;
  const int q = static_cast<int>(bq % sqp), b = static_cast<int>(bq / sqp);
  float nl = -1.0e30f, nd = 0.f;
  if (q < a.Sq) {
    const long row = static_cast<long>(b) * a.Sq + q;
    const uint4 ov = *reinterpret_cast<const uint4*>(a.o + row * a.ldo + head * kDh + 8 * c);
    const uint4 gv = *reinterpret_cast<const uint4*>(a.dout + row * a.ldo + head * kDh + 8 * c);
    const unsigned ow[4] = {ov.x, ov.y, ov.z, ov.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s = fmaf(__uint_as_float(ow[e] << 16), __uint_as_float(gw[e] << 16), s);
      s = fmaf(__uint_as_float(ow[e] & 0xffff0000u), __uint_as_float(gw[e] & 0xffff0000u), s);
    }
    s = quarter_sum(s);
    nl = (-a.lse[(static_cast<long>(b) * a.H + head) * a.Sq + q] * kLog2e + (a.drop.on() ? __log2f(a.drop.inv_keep) : 0.f)) / (a.scale * kLog2e);
    nd = a.drop.on() ? -s / a.drop.inv_keep : -s;
  }
  if (c == 0) {                                                    // per slice of 32 queries: [nl x 32][nd x 32]
    float* sl = stats + ((static_cast<long>(b) * a.H + head) * sqp + (q & ~31)) * 2 + (q & 31);
    sl[0] = nl;
    sl[32] = nd;
  }
  if ((q & 31) == 0 && c < 4 * nkb) {                              // (nkb <= 16 on this path: at most 64 words per slice, 16 lanes x up to 4)
    unsigned* fl = flags + ((static_cast<long>(b) * a.H + head) * ns + (q >> 5)) * nkb * 4;
    for (int w = c; w < 4 * nkb; w += 16) fl[w] = 0u;
  }
  if (item == 0 && c < 4) flags[static_cast<long>(a.B) * a.H * ns * nkb * 4 + c] = 0u;
}

// one 32-d block of a transposed-read A (or B) operand: rows R0 + 8 (j >> 2) + 4 h + (j & 3) of a swizzled 256-byte-row image
__device__ __forceinline__ void tr1_issue(const unsigned char* img, int R0, int db, int lane, TrFrag& f) {
  const int i = lane & 15, g4 = (lane >> 4) & 1, h = lane >> 5;
  const int row = R0 + 4 * h + (i >> 2), chunk = 4 * db + 2 * g4 + ((i & 3) >> 1);
  const unsigned base = lds_off(img) + 8 * (i & 1);
  const unsigned a0 = base + swz(row, chunk), a1 = base + swz(row + 8, chunk);
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(f.lo), "=&v"(f.hi) : "v"(a0), "v"(a1) : "memory");
}
// the dS^T image: [key][32 q] bf16 = 64-byte rows of four 16-byte chunks, chunk c of row k at 16 (c ^ ((k >> 2) & 3)): the writes (a lane =
// a key, 8 bytes per run of four queries) and the transposed reads below both run at the two cycles their 512 bytes need
__device__ __forceinline__ unsigned x_off(int key, int chunk) { return static_cast<unsigned>(64 * key + 16 * (chunk ^ ((key >> 2) & 3))); }
__device__ __forceinline__ void trx_issue(const unsigned char* x, int K0, int lane, TrFrag& f) {
  const int i = lane & 15, g4 = (lane >> 4) & 1, h = lane >> 5;
  const int row = K0 + 4 * h + (i >> 2), chunk = 2 * g4 + ((i & 3) >> 1);
  const unsigned base = lds_off(x) + 8 * (i & 1);
  const unsigned a0 = base + x_off(row, chunk), a1 = base + x_off(row + 8, chunk);
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(f.lo), "=&v"(f.hi) : "v"(a0), "v"(a1) : "memory");
}

// dK^T / dV^T accumulate in the 256 AGPRs for the whole kernel.  Written as inline asm with the "a" constraint: with the builtin the
// register allocator also puts the short-lived S / dP / dQ accumulators into AGPRs and then shuttles dK / dV tiles between the two files
// around every product (sixteen v_accvgpr_read + sixteen v_accvgpr_write + an s_nop 11 per pair of MFMAs).  Operands are compiler-visible
// registers; the results are only read after the loop (compiler-generated v_accvgpr_read, hundreds of cycles behind the last product).
__device__ __forceinline__ void mfma_acc(f32x16& acc, const bf16x8& x, const bf16x8& y) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y));
}
// ... and S', dP, dQ^T accumulate in arch VGPRs.  The compiler does not know these are MFMAs, so the wait states it would insert are
// placed by hand: mfma_settle() between the last product of a chain and the first vector instruction that reads its result
// (a 32x32x16 product is 8 passes; 16 wait states cover it), mfma_srcc_ready() between vector writes of an accumulator and its first product.
__device__ __forceinline__ void mfma_vgpr(f32x16& acc, const bf16x8& x, const bf16x8& y) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
}
// first product of a chain: the accumulator starts from the inline constant 0 -- no sixteen v_mov per chain, no write -> SrcC wait states
__device__ __forceinline__ void mfma_vgpr0(f32x16& acc, const bf16x8& x, const bf16x8& y) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(x), "v"(y));
}
// (the accumulators go through the statements as operands: that is what keeps the compiler's own reads / writes of them on the right side)
__device__ __forceinline__ void mfma_settle(f32x16& x, f32x16& y) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ void mfma_settle(f32x16& x) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(x)); }
__device__ __forceinline__ void mfma_srcc_ready(f32x16& x, f32x16& y) { asm volatile("s_nop 3" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ void mfma_srcc_ready(f32x16& x) { asm volatile("s_nop 3" : "+v"(x)); }

// kDrop: 0 no dropout; 1 the keep decisions are re-made from the hash (dropout.h); 2 they are read back as the bits the forward left
// (AttnArgs::keep_bits: per 32-query slice and 32-key block one 32-bit word per key, bit q = keep(query q of the slice, key) -- the forward's
// compare results as they stand in its scalar registers, see attn_common.h keep_bits_*): two vector instructions per element instead of
// a hash per two elements plus the scalar-register mask traffic, which is what dropout cost this kernel (+65 % VALU, +100 % SALU).
typedef __bf16 bf16x2c __attribute__((ext_vector_type(2)));
template <int kDrop, int kDbg>
__global__ __launch_bounds__(kFbThreads, 1) void attn_bwd_fused_kernel(FusedArgs fa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnArgs& a = fa.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this workgroup's (batch, head, key block): ticket v of the XCD group's counter = logical tile slice0 + v
  const int n_tiles = fa.nkb * a.B * a.H;
  // (kDbg & 16, tests only: ONE counter for the whole grid, so that the key blocks of a head land on different XCDs and every hand-off
  // crosses XCDs)
  const int q8 = n_tiles >> 3, r8 = n_tiles & 7, xg = (kDbg & 16) ? 0 : (blockIdx.x & 7);
  const int slice0 = (kDbg & 16) ? 0 : (xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8);
  const int slice_n = (kDbg & 16) ? n_tiles : q8 + (xg < r8 ? 1 : 0);
  unsigned* const tflag = reinterpret_cast<unsigned*>(smem + kFbOffFlag);
  if (tid == 0) {
    unsigned* const counter = fa.sched + xg * 16;
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1u == fa.sched_total[xg]) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the launch's last draw
    *tflag = t;
  }
  __syncthreads();
  const unsigned v = *tflag;
  if (v >= static_cast<unsigned>(slice_n)) return;                 // block-uniform; cannot happen with grid == n_tiles and clean counters
  const int logical = slice0 + static_cast<int>(v);
  const int kb = logical % fa.nkb, bh = logical / fa.nkb;
  const int b = bh / a.H, head = bh % a.H;
  const int key0 = kb * kFbKeys, ns = fa.ns;

  const unsigned short* qb = a.q + static_cast<long>(b) * a.Sq * a.ldq + head * kDh;
  const unsigned short* dob = a.dout + static_cast<long>(b) * a.Sq * a.ldo + head * kDh;
  const unsigned short* kb_ = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * kDh;
  const unsigned short* vb = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * kDh;
  const float* stat_g = fa.stats + static_cast<long>(bh) * ns * kFbSlice * 2;
  // keep words of (this head, slice 0, this wave's first 32-key block)
  const unsigned* bits_g = kDrop == 2 ? a.keep_bits + ((static_cast<long>(bh) * a.bits_nq) * a.bits_nk + (kb * (kFbKeys / 32) + 2 * wave)) * 32 : nullptr;
  const int klen = a.key_len ? __builtin_amdgcn_readfirstlane(a.key_len[b]) : a.Sk;      // (a loaded value is "divergent" to the compiler: make it scalar)
  const float sl2 = a.scale * kLog2e;
  const bool key_mask = a.causal || key0 + kFbKeys > klen || key0 + kFbKeys > a.Sk;      // block-uniform, in a scalar register

  // ---- staging: the K image once; Q | dO | statistics of slice j into ring slot j & 1
  const int lrow = lane >> 4, lchunk = lane & 15;
  const int ldq_i = static_cast<int>(a.ldq), ldo_i = static_cast<int>(a.ldo), ldk_i = static_cast<int>(a.ldk);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int g = 16 * wave + i, row = 4 * g + lrow;
    const int chunk = lchunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
    int gr = key0 + row;
    gr = gr < a.Sk ? gr : a.Sk - 1;                               // keys past the end: P is forced to 0 for them, their rows are never stored
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb_ + static_cast<unsigned>(gr * ldk_i + chunk * 8)),
                                     (__attribute__((address_space(3))) void*)(smem + g * 1024), 16, 0, 0);
  }
  auto issue_slice = [&](int j) {
    unsigned char* slot = smem + kFbOffT + (j & 1) * kFbTile;
    // the lane's part of the source addresses is recomputed per slice: hoisted out of the loop it would sit in (or be spilled from) registers
    // the whole time (cdna_hip_programming.md, attention prefill pitfalls)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int lrow = lane_o >> 4, lchunk = lane_o & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = 4 * wave + i, rg = g & 7, row = 4 * rg + lrow;            // g 0..7: Q, 8..15: dO (wave-uniform)
      const int chunk = lchunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
      int gr = j * kFbSlice + row;
      gr = gr < a.Sq ? gr : a.Sq - 1;                             // rows past the end repeat the last valid row (their P is 0 by the statistics)
      const unsigned short* src = g < 8 ? qb + static_cast<unsigned>(gr * ldq_i + chunk * 8) : dob + static_cast<unsigned>(gr * ldo_i + chunk * 8);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + g * 1024), 16, 0, 0);
    }
    if (wave == 3)                                                // 64 lanes x 4 bytes = the slice's 32 x {nl, nd}
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stat_g + j * kFbSlice * 2 + lane_o),
                                       (__attribute__((address_space(3))) void*)(smem + kFbOffS + (j & 1) * 256), 4, 0, 0);
    if (kDrop == 2) {                                             // the keep words of this wave's two key blocks: lane = (block, key)
      const int key = lane_o & 31;
      const unsigned* src = bits_g + (static_cast<long>(j) * a.bits_nk + (lane_o >> 5)) * 32 + keep_bits_word(key);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + kFbOffBits + (j & 1) * 1024 + wave * 256), 4, 0, 0);
    }
  };
  issue_slice(0);

  // V of this wave's 64 keys: the B operand of dP, in registers for the whole kernel
  bf16x8 vf[2][8];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) frags_from_global(vb, a.ldv, key0 + 64 * wave + 32 * blk + r, a.Sk, lane, vf[blk]);

  f32x16 dk[2][4], dv[2][4];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dk[blk][db][i] = 0.f; dv[blk][db][i] = 0.f; }

  const unsigned sk_pairs = static_cast<unsigned>((a.Sk + 1) >> 1);
  const unsigned key2 = mix32(a.drop.key);
  const unsigned par = static_cast<unsigned>(lane) & 1u;
  const unsigned thr16 = a.drop.thr << 16;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // All LDS reads of the loop are inline asm with hand-placed lgkmcnt waits and a small ring of operand registers: left to the compiler the
  // reads of a whole chain are hoisted in front of it and the kernel spills (a reload of a spilled value also drains the DMA in flight).
  // Addresses come from two lane constants by XOR: the swizzle is an XOR of address bits 4..7, so the k-step / d-block enters as
  // `^ 32 s` / `^ 64 db` (attention.hip, dK/dV kernel).
  const unsigned smem_base = lds_off(smem);
  unsigned rowbase = smem_base + static_cast<unsigned>(256 * r + 16 * (h ^ (((r & 3) << 2) | ((r >> 2) & 3))));
  unsigned trbase, xbase;
  {
    const int i = lane & 15, g4 = (lane >> 4) & 1, row = 4 * h + (i >> 2);
    trbase = smem_base + static_cast<unsigned>(8 * (i & 1) + swz(row, 2 * g4 + ((i & 3) >> 1)));
    xbase = smem_base + static_cast<unsigned>(8 * (i & 1) + 64 * row + 16 * ((2 * g4 + ((i & 3) >> 1)) ^ h));
  }
#define ADT_TR2(F, ADDR, IMM)                                                                                                   \
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%5"                                     \
               : "=&v"((F).lo), "=&v"((F).hi) : "v"(ADDR), "v"((ADDR) ^ 32u), "i"(IMM), "i"((IMM) + 2048) : "memory")
#define ADT_TRX(F, ADDR, IMM)                                                                                                   \
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%5"                                     \
               : "=&v"((F).lo), "=&v"((F).hi) : "v"(ADDR), "v"((ADDR) ^ 32u), "i"(IMM), "i"((IMM) + 512) : "memory")

  // ---- dQ across the key-block workgroups of the (batch, head): a scheduled fan-in that never sits in the critical path.
  // Every key block stores its dQ^T tile of slice j at the end of iteration j (write-through stores) and counts it in the tile's flag after
  // the drain that iteration j + 1 performs anyway.  Slice jr is reduced by key block jr % nkb in nkb STEPS, one per iteration, starting at
  // iteration jr + 3: step n brings key block n's tile into a landing zone in LDS by LDS-DMA at the top of the iteration (no registers, a
  // whole iteration of latency hiding; its flag was read one iteration earlier, a slice barrier in between) and adds it to the running sum
  // parked in LDS at the end: ((p0 + p1) + p2) + ... in key-block order, a fixed order, so dQ is bitwise reproducible.  The schedule is a
  // pure function of the iteration number; what is still open after the last slice is finished behind the loop.
  const int nkb = fa.nkb;
  const bool handoff = nkb > 1 && !(kDbg & 1);
  const long fl_bh = static_cast<long>(bh) * ns;
  int pub_pending = -1;
  unsigned fl = 0;                                                 // flag of the step of the coming iteration (wave-uniform)
  auto flag_of = [&](int jj, int src) { return fa.flags + ((fl_bh + jj) * nkb + src) * 4 + wave; };
  auto part_of = [&](int jj, int src) { return fa.part + (((fl_bh + jj) * nkb + src) * 4 + wave) * 1024; };
  // the reduction step of (virtual) iteration i is slice jr = kb + nkb * floor((i - 4 - kb) / nkb), key block n = (i - 4 - kb) mod nkb (none while
  // i < 4 + kb or once jr >= ns); kept as a pair that advances by one step per iteration -- no divisions in the loop
  int cur_jr = kb, cur_n = -(4 + kb);                              // cur_n < 0: the schedule has not started yet
  auto advance = [&](int& jr, int& n) {
    if (++n == nkb) { n = 0; jr += nkb; }
  };
  auto valid = [&](int jr, int n) { return handoff && n >= 0 && jr < ns; };
  auto land_to = [&](int jr, int n, int lds_byte) {                // this wave's quarter of key block n's tile -> 4 KiB of LDS (L1 bypassed)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const float* src = part_of(jr, n) + lane_o * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + g * 256),
                                       (__attribute__((address_space(3))) void*)(smem + lds_byte + g * 1024), 16, 0, 16);
  };
  auto land = [&](int jr, int n) { land_to(jr, n, kFbOffLand + wave * 4096); };
  auto wait_flag = [&](int jr, int n) {                            // a tile that had not been published when its flag was prefetched (rare)
    unsigned spins = 0, f = 0;
    for (;;) {
      if (lane == 0) f = __hip_atomic_load(flag_of(jr, n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f = __builtin_amdgcn_readfirstlane(f);
      if (f != 0u) break;
      if (++spins > kFbSpinLimit) {                               // never in a healthy launch: report and carry on instead of hanging the GPU
        if (lane == 0) attn_bwd_report_giveup(fa, static_cast<long>(a.B) * a.H * ns * nkb * 4);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  };
  auto store_dq = [&](const f32x16& t, int jj) {                   // scale, round, store this wave's 32 d of slice jj's rows
    int lane_o = lane;                                            // (per-lane address parts recomputed here, not carried through the loop)
    asm volatile("" : "+v"(lane_o));
    const int qi = jj * kFbSlice + (lane_o & 31);
    if (qi < a.Sq) {                                              // lanes q and q + 32 own the same row: they skip together
      unsigned short* p = a.dq + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * kDh + 32 * wave + 8 * (lane_o >> 5);
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        unsigned ax = pack2(t[4 * g + 0] * a.scale, t[4 * g + 1] * a.scale), ay = pack2(t[4 * g + 2] * a.scale, t[4 * g + 3] * a.scale);
        unsigned bx = pack2(t[4 * g + 4] * a.scale, t[4 * g + 5] * a.scale), by = pack2(t[4 * g + 6] * a.scale, t[4 * g + 7] * a.scale);
        const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        *reinterpret_cast<uint4*>(p + 8 * g) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
      }
    }
  };
  const unsigned stash_a = smem_base + static_cast<unsigned>(kFbOffStash + wave * 4096 + lane * 16);
  auto add_step_from = [&](int jr, int n, unsigned land_a) {       // running sum (+)= landed tile (at LDS address land_a + 1024 g); the last step stores dQ
    f32x4 o[4], q4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o[g]) : "v"(land_a), "i"(1024 * g) : "memory");
    if (n > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q4[g]) : "v"(stash_a), "i"(1024 * g) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (n > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) o[g] = q4[g] + o[g];
    }
    if (n == nkb - 1) {
      f32x16 t;
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = o[i >> 2][i & 3];
      store_dq(t, jr);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(stash_a), "v"(o[g]), "i"(1024 * g) : "memory");
    }
  };
  auto add_step = [&](int jr, int n) { add_step_from(jr, n, stash_a + 4u * 4096u); };

#define ADT_STAMP(K)                                                                                                      \
  if ((kDbg & 32) && j == 12 && wave == 0 && blockIdx.x == 600) {                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    const unsigned long long tnow = __builtin_amdgcn_s_memtime();                                                         \
    if (lane == 0) fa.stamps[K] = tnow;                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
  auto hand_on = [&](const f32x16& t, int jj) {                    // slice jj's dQ^T tile of this key block: store dQ (one key block) or publish the tile
    if (!handoff) {
      store_dq(t, jj);
      return;
    }
    // write-through (sc1) 16-byte stores as compiler-visible buffer stores: an inline-asm store gets no hazard wait states before the
    // next instruction that overwrites its data registers
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(part_of(jj, kb), 0, 4096, 0x00020000);
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 o = {__float_as_uint(t[4 * g]), __float_as_uint(t[4 * g + 1]), __float_as_uint(t[4 * g + 2]), __float_as_uint(t[4 * g + 3])};
      __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (g * 64 + lane_s) * 16, 0, 16);      // aux 16 = sc1
    }
    pub_pending = jj;
  };
  // dQ^T, d-block `wave`, over the workgroup's 256 keys: 16 k-steps of 16 keys through a ring of four operand units (three k-steps ahead
  // of the product: one wave per SIMD, nobody else hides the LDS latency), two accumulation chains (even / odd k-steps: a product waits
  // ~100 cycles for the previous one on the same VGPR accumulator).  ADT_DQ_BEGIN / ADT_DQ_STEP(0..15) / ADT_DQ_END.
#define ADT_DQ_BEGIN                                                                              \
  f32x16 dq2;                                                                                     \
  const unsigned ka_a = trbase ^ static_cast<unsigned>(64 * wave), xb_a = xbase + static_cast<unsigned>(kFbOffX); \
  TrFrag ka[4], xb[4];                                                                            \
  ADT_TR2(ka[0], ka_a, 0);                                                                        \
  ADT_TRX(xb[0], xb_a, 0);                                                                        \
  ADT_TR2(ka[1], ka_a, 4096);                                                                     \
  ADT_TRX(xb[1], xb_a, 1024);                                                                     \
  ADT_TR2(ka[2], ka_a, 2 * 4096);                                                                 \
  ADT_TRX(xb[2], xb_a, 2 * 1024);
#define ADT_DQ_STEP(KK)                                                                           \
  if ((KK) + 3 < 16) {                                                                            \
    ADT_TR2(ka[((KK) + 3) & 3], ka_a, ((KK) + 3) * 4096);                                         \
    ADT_TRX(xb[((KK) + 3) & 3], xb_a, ((KK) + 3) * 1024);                                         \
    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");                                           \
  } else if ((KK) + 2 < 16) {                                                                     \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                            \
  } else if ((KK) + 1 < 16) {                                                                     \
    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                            \
  } else {                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
  }                                                                                               \
  __builtin_amdgcn_sched_barrier(0);                                                              \
  if ((KK) == 0) mfma_vgpr0(dq, tr_get(ka[0]), tr_get(xb[0]));                                    \
  else if ((KK) == 1) mfma_vgpr0(dq2, tr_get(ka[1]), tr_get(xb[1]));                              \
  else if ((KK) & 1) mfma_vgpr(dq2, tr_get(ka[(KK) & 3]), tr_get(xb[(KK) & 3]));                  \
  else mfma_vgpr(dq, tr_get(ka[(KK) & 3]), tr_get(xb[(KK) & 3]));                                 \
  __builtin_amdgcn_sched_barrier(0);
#define ADT_DQ_END                                                                                \
  mfma_settle(dq, dq2);                                                                           \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) dq[i] += dq2[i];

  // The slice loop exists twice: key blocks that need the masked form of the arithmetic (causal, padded or partial key block) and interior
  // ones; the choice is per workgroup, and a run-time flag inside the loop costs two taken branches per pair of elements.
  auto run_slices = [&](auto masked_tag) {
  constexpr bool kMasked = decltype(masked_tag)::value;
  // dropout: block 0's keep masks of the coming slice are made during phase 4 of the running one (pure matrix work, the vector pipe idles)
  // and carried in scalar registers; block 1's hashes are made during phase 1 (hw) and turned into masks where phase 3 uses them
  uint64_t km0[16];
  auto masks_from_hash = [&](int e, unsigned hh, uint64_t (&km)[16]) {      // hash e of a block gives the masks of its accumulator registers e and e + 8
    const uint64_t even = 0x5555555555555555ull, odd = 0xaaaaaaaaaaaaaaaaull;
    const uint64_t c_lo = __builtin_amdgcn_ballot_w64((hh << 16) >= thr16);       // decision of the pair's even key
    const uint64_t c_hi = __builtin_amdgcn_ballot_w64(hh >= thr16);               // ... of its odd key
    km[e] = (c_lo & even) | ((c_hi & even) << 1);
    km[e + 8] = (c_hi & odd) | ((c_lo & odd) >> 1);
  };
  const unsigned headpair0 = static_cast<unsigned>(static_cast<uint64_t>(bh) * a.Sq * sk_pairs) + static_cast<unsigned>((key0 + 64 * wave + r) >> 1);
  auto slice_hash = [&](int jj, int blk, int e) {                  // pair (query row (e & 3) + 8 (e >> 2) + 16 par + 4 h of slice jj, this lane's key pair of block blk)
    return mix32((headpair0 + (blk ? 16u : 0u) + static_cast<unsigned>(jj * kFbSlice + 16 * static_cast<int>(par) + 4 * h + (e & 3) + 8 * (e >> 2)) * sk_pairs) ^ key2);
  };
#pragma unroll
  for (int e = 0; e < 16; ++e) km0[e] = 0;
  if (kDrop == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) masks_from_hash(e, slice_hash(0, 0, e), km0);
  }
  for (int j = 0; j < ns; ++j) {
    asm volatile("" : "+v"(rowbase), "+v"(trbase), "+v"(xbase));    // keep the per-k-step / per-d-block addresses derived from these out of loop-invariant registers
    const int sjr = cur_jr, sn = cur_n;
    const bool step = valid(sjr, sn);                             // block-uniform
    int njr = cur_jr, nn = cur_n;
    advance(njr, nn);
    bool late = false;
    if (step) {
      if (sn == kb || fl != 0u) land(sjr, sn);                    // own tile: drained two iterations ago; others: flag seen last iteration
      else late = true;
    }
    const bool nstep = valid(njr, nn) && nn != kb;
    unsigned fv = 0;
    if (nstep && lane == 0) fv = __hip_atomic_load(flag_of(njr, nn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j + 1 < ns) issue_slice(j + 1);
    const unsigned slot = static_cast<unsigned>(kFbOffT + (j & 1) * kFbTile);
    const unsigned stat_a = smem_base + static_cast<unsigned>(kFbOffS + (j & 1) * 256 + 16 * h);       // + 32 g: queries 8 g + 4 h .. + 3
    unsigned char* xs = smem + kFbOffX;
    // ---- the slice as four phases, software-pipelined across the wave's two 32-key blocks so that the matrix pipe and the vector ALUs
    // work side by side (one wave per SIMD: nothing else would fill the other pipe):
    //   1  S', dP of block 0                       beside  the dropout hashes of both blocks
    //   2  S', dP of block 1                       beside  the softmax / dS arithmetic of block 0
    //   3  dV^T, dK^T of block 0                   beside  the arithmetic of block 1
    //   4  dV^T, dK^T of block 1
    // sched_barrier(0) pins the interleave: one pair of MFMAs, then one pair of elements' arithmetic (or one hash).
    f32x16 st0, dp0, st1, dp1;
    unsigned hp0[8], hs0[8], hp1[8], hs1[8];
    f32x4 ndv[4];                                                 // the slice's row constants nd of this lane's 16 query rows
    bf16x8 fq[3], fd[3], fk[3];                                   // operand ring of the chains: two k-steps ahead
    const unsigned tq_a = rowbase + slot;
    const unsigned kr_a0 = rowbase + static_cast<unsigned>((64 * wave) * 256);
    const int krow0 = 64 * wave + r, ki0 = key0 + krow0, ki1 = ki0 + 32;
#define ADT_UNIT(U, S, KR)                                                                                                      \
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:8192\n\tds_read_b128 %2, %4"                                \
                 : "=&v"(fq[U]), "=&v"(fd[U]), "=&v"(fk[U])                                                                      \
                 : "v"(tq_a ^ static_cast<unsigned>(32 * (S))), "v"((KR) ^ static_cast<unsigned>(32 * (S))) : "memory")
    // chain prologue: the row constants as the initial accumulator, the first two operand units
#define ADT_CHAIN_BEGIN(ST, DP, KR)                                                                                             \
    {                                                                                                                           \
      f32x4 c[4];                                                                                                               \
      _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                                             \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c[g]) : "v"(stat_a), "i"(32 * g) : "memory");                       \
      ADT_UNIT(0, 0, KR); ADT_UNIT(1, 1, KR); ADT_UNIT(2, 2, KR);                                                               \
      asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");                                                                        \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) { ST[i] = c[i >> 2][i & 3]; DP[i] = 0.f; }                                 \
      mfma_srcc_ready(ST, DP);                                                                                                  \
    }
#define ADT_CHAIN_STEP(S, ST, DP, KR, VF)                                                                                       \
    if ((S) < 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");                                                             \
    else if ((S) < 7) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");                                                        \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                          \
    mfma_vgpr(ST, fq[(S) % 3], fk[(S) % 3]);                                                                                    \
    mfma_vgpr(DP, fd[(S) % 3], VF[S]);                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                          \
    if ((S) + 3 < 8) {                                                                                                          \
      if ((S) % 3 == 0) ADT_UNIT(0, (S) + 3, KR);                                                                               \
      else if ((S) % 3 == 1) ADT_UNIT(1, (S) + 3, KR);                                                                          \
      else ADT_UNIT(2, (S) + 3, KR);                                                                                            \
    }
    // keep masks: one 64-bit lane mask per accumulator register (the two lanes of a key pair share every hash: dropout.h)
    unsigned hw[8];                                               // block 1's hash words of this slice (made in phase 1)
    // softmax / dropout / dS arithmetic of elements 2 m, 2 m + 1 of a block: P (dropped) and dS, packed to bf16 for the second products
    // lane constants of the masked form: the key-padding term, the causal term, the validity of this lane's keys, and the first query row
    // of the slice that may see the key (causal: row q of the slice is masked iff q < key - slice start)
    const float cau2 = a.causal ? a.mask_value * kLog2e : 0.f;
    const float pad2_0 = ki0 >= klen ? a.mask_value * kLog2e : 0.f, pad2_1 = ki1 >= klen ? a.mask_value * kLog2e : 0.f;
    const float kval0 = ki0 < a.Sk ? 1.f : 0.f, kval1 = ki1 < a.Sk ? 1.f : 0.f;
    const int qrel0 = ki0 - j * kFbSlice, qrel1 = qrel0 + 32;
    // kDrop == 2: wb = the lane's keep word of the block, shifted right by 4 h (so that the bit of accumulator register i sits at
    // position acc_row(i, 0)); v_bfe_i32 spreads the bit over the register, an AND with 1 / (1 - p) gives the keep scale
    unsigned wb0 = 0, wb1 = 0;
    // Two elements per call, on register PAIRS (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: accumulator registers 2 m, 2 m + 1 are an
    // aligned pair, and so are their row constants): one wave per SIMD issues a vector instruction every ~5 cycles whatever it is
    // (tools/probe/probe_mfma_valu), so what counts is the NUMBER of instructions beside the products -- per pair 7 without dropout
    // (10 before), 11 with the keep bits.  P arrives scaled by 1 / (1 - p) (folded into the chain's initial accumulator nl), the keep decision is an AND (bits) or a
    // select (hash) on it, and dS = P_dropped dP + P' nd' with nd' = -delta (1 - p) -- no separate keep-scale product.
    auto arith_pair = [&](int m, const f32x16& st, const f32x16& dp, const uint64_t (&km)[16], unsigned wb, float pad2, float kval, int qrel,
                          unsigned (&hp)[8], unsigned (&hs)[8]) {
      const int i0 = 2 * m, g = i0 >> 2, c0 = i0 & 3;
      const f32x2 s2 = {st[i0], st[i0 + 1]}, d2 = {dp[i0], dp[i0 + 1]};
      const f32x2 nd2 = {ndv[g][c0], ndv[g][c0 + 1]};
      f32x2 x;
      if (kMasked) {                                             // compile-time: interior key blocks take the short form
        const f32x2 madd = {pad2 + (acc_row(i0, h) < qrel ? cau2 : 0.f), pad2 + (acc_row(i0 + 1, h) < qrel ? cau2 : 0.f)};
        x = s2 * sl2 + madd;
      } else {
        x = s2 * sl2;
      }
      f32x2 pv = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
      if (kMasked) pv = pv * kval;
      f32x2 pd, ds;
      if (kDrop == 0) {
        pd = pv;
        ds = pv * (d2 + nd2);
      } else {
        if (kDrop == 1) {
          f32x2 k01;                                             // (selects of a constant: asm must not consume v_exp_f32 results directly, see below)
          asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(k01[0]) : "s"(km[i0]));
          asm("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(k01[1]) : "s"(km[i0 + 1]));
          pd = pv * k01;
        } else {                                                 // bit acc_row(i, 0) of the lane's (shifted) keep word, spread over the register
          // (v_bfe_i32 as asm: given the builtin the compiler turns bit test + AND into v_and + v_cmp + v_cndmask through scalar register
          // pairs -- three instructions and thirty-two masks to keep; the AND stays compiler-visible because its other input comes
          // straight from v_exp_f32, a hazard the recogniser must see)
          unsigned t0, t1;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t0) : "v"(wb), "n"(c0 + 8 * g));
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t1) : "v"(wb), "n"(c0 + 8 * g + 1));
          pd[0] = __uint_as_float(__float_as_uint(pv[0]) & t0);
          pd[1] = __uint_as_float(__float_as_uint(pv[1]) & t1);
        }
        ds = pd * d2 + pv * nd2;
      }
      // (without dropout P comes straight from v_exp_f32: the compiler-visible conversion there, not attn_common.h's asm pack2 -- an asm
      // instruction gets none of the wait states a transcendental's consumer needs, and it read the exponent instead of the power)
      if (kDrop == 0) hp[m] = __builtin_bit_cast(unsigned, __builtin_convertvector(pd, bf16x2c));
      else hp[m] = pack2(pd[0], pd[1]);
      hs[m] = pack2(ds[0], ds[1]);
    };
    // block 1, elements 2 t, 2 t + 1, 2 t + 8, 2 t + 9: their keep masks come from two hashes made right here (four scalar register pairs
    // live at a time instead of thirty-two), then two arithmetic pairs
    auto arith_quad1 = [&](int t) {
      uint64_t km1[16];                                          // (only entries 2 t, 2 t + 1, 2 t + 8, 2 t + 9 are made and read)
      if (kDrop == 1) {
        masks_from_hash(2 * t, hw[2 * t], km1);
        masks_from_hash(2 * t + 1, hw[2 * t + 1], km1);
      }
      arith_pair(t, st1, dp1, km1, wb1, pad2_1, kval1, qrel1, hp1, hs1);
      arith_pair(t + 4, st1, dp1, km1, wb1, pad2_1, kval1, qrel1, hp1, hs1);
    };
    auto write_ds = [&](int krow, const unsigned (&hs)[8]) {    // dS^T of a block to LDS for the dQ product
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2 w2 = make_uint2(hs[2 * g], hs[2 * g + 1]);
        asm volatile("ds_write_b64 %0, %1" :: "v"(lds_off(xs) + x_off(krow, g) + 8 * h), "v"(w2) : "memory");
      }
    };

    ADT_STAMP(0)
    // ---- phase 1: S', dP of BOTH blocks in one interleaved chain: four accumulators take turns (a product waits ~100 cycles for the
    // previous one on the same VGPR accumulator, so two chains alone run at half rate), the Q / dO fragments are read once for both blocks
    if (!(kDbg & 8)) {
      bf16x8 fk1[3];
#define ADT_UNIT2(U, S)                                                                                                         \
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:8192\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:8192" \
                   : "=&v"(fq[U]), "=&v"(fd[U]), "=&v"(fk[U]), "=&v"(fk1[U])                                                     \
                   : "v"(tq_a ^ static_cast<unsigned>(32 * (S))), "v"(kr_a0 ^ static_cast<unsigned>(32 * (S))) : "memory")
      if (kDrop == 2) {                                           // (first: every counted wait below then covers them too)
        const unsigned bw_a = smem_base + static_cast<unsigned>(kFbOffBits + (j & 1) * 1024 + wave * 256 + 4 * r);
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:128" : "=&v"(wb0), "=&v"(wb1) : "v"(bw_a) : "memory");
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ndv[g]) : "v"(stat_a), "i"(128 + 32 * g) : "memory");
      {
        // the score chains start from the row constant nl: read from LDS STRAIGHT INTO the accumulators' quarters, once per block (a
        // ds_read_b128 per four registers instead of four v_mov); the dP chains start from the inline constant 0
        f32x4 q0[4], q1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(q0[g]), "=&v"(q1[g]) : "v"(stat_a), "i"(32 * g) : "memory");
        ADT_UNIT2(0, 0); ADT_UNIT2(1, 1); ADT_UNIT2(2, 2);
        st0 = __builtin_shufflevector(__builtin_shufflevector(q0[0], q0[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(q0[2], q0[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                      0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        st1 = __builtin_shufflevector(__builtin_shufflevector(q1[0], q1[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(q1[2], q1[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                      0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s < 6) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // (in order: the statistics and keep words, read first, are back too)
        else if (s < 7) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0) {
          mfma_vgpr(st0, fq[0], fk[0]);
          mfma_vgpr(st1, fq[0], fk1[0]);
          mfma_vgpr0(dp0, fd[0], vf[0][0]);
          mfma_vgpr0(dp1, fd[0], vf[1][0]);
        } else {
          mfma_vgpr(st0, fq[s % 3], fk[s % 3]);
          mfma_vgpr(st1, fq[s % 3], fk1[s % 3]);
          mfma_vgpr(dp0, fd[s % 3], vf[0][s]);
          mfma_vgpr(dp1, fd[s % 3], vf[1][s]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < 8) {
          if (s % 3 == 0) ADT_UNIT2(0, s + 3);
          else if (s % 3 == 1) ADT_UNIT2(1, s + 3);
          else ADT_UNIT2(2, s + 3);
        }
        if (kDrop == 1) {
          hw[s] = slice_hash(j, 1, s);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#undef ADT_UNIT2
      mfma_settle(st0, dp0);
      mfma_settle(st1, dp1);
      __builtin_amdgcn_sched_barrier(0);
      if (kDrop == 2) {                                           // (behind the chain's lgkmcnt(0): the words have arrived)
        wb0 >>= 4 * h;
        wb1 >>= 4 * h;
      }
      ADT_STAMP(1)
    } else {                                                      // (experiment build: no S / dP chains)
#pragma unroll
      for (int g = 0; g < 4; ++g) ndv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 16; ++i) { st0[i] = 0.f; dp0[i] = 0.f; st1[i] = 0.f; dp1[i] = 0.f; }
    }
    // ---- phase 2: the arithmetic of block 0 beside the dQ product of the PREVIOUS slice (its dS^T image is complete since that slice's
    // barrier, and is overwritten only behind the barrier below)
    f32x16 dq;
    if (j > 0 && !(kDbg & 2)) {
      // (here a ring of two units and ONE accumulation chain: the arithmetic between the products hides both latencies, and registers are scarce)
      const unsigned ka_a = trbase ^ static_cast<unsigned>(64 * wave), xb_a = xbase + static_cast<unsigned>(kFbOffX);
      TrFrag ka[2], xb[2];
      ADT_TR2(ka[0], ka_a, 0);
      ADT_TRX(xb[0], xb_a, 0);
      ADT_TR2(ka[1], ka_a, 4096);
      ADT_TRX(xb[1], xb_a, 1024);
#define ADT_DQ2_STEP(KK)                                                                          \
      if ((KK) + 1 < 16) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                       \
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      if ((KK) == 0) mfma_vgpr0(dq, tr_get(ka[0]), tr_get(xb[0]));                                \
      else mfma_vgpr(dq, tr_get(ka[(KK) & 1]), tr_get(xb[(KK) & 1]));                             \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      if ((KK) + 2 < 16) {                                                                        \
        ADT_TR2(ka[(KK) & 1], ka_a, ((KK) + 2) * 4096);                                           \
        ADT_TRX(xb[(KK) & 1], xb_a, ((KK) + 2) * 1024);                                           \
      }
#define ADT_PH2(M)                                                                                \
      ADT_DQ2_STEP(2 * (M))                                                                       \
      ADT_DQ2_STEP(2 * (M) + 1)                                                                   \
      arith_pair(M, st0, dp0, km0, wb0, pad2_0, kval0, qrel0, hp0, hs0);                          \
      __builtin_amdgcn_sched_barrier(0);
      ADT_PH2(0) ADT_PH2(1) ADT_PH2(2) ADT_PH2(3) ADT_PH2(4) ADT_PH2(5) ADT_PH2(6) ADT_PH2(7)
#undef ADT_PH2
#undef ADT_DQ2_STEP
      mfma_settle(dq);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) dq[i] = 0.f;
#pragma unroll
      for (int m = 0; m < 8; ++m) arith_pair(m, st0, dp0, km0, wb0, pad2_0, kval0, qrel0, hp0, hs0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the dS^T image is single-buffered: every wave must have finished the previous slice's dQ product (they have, long ago: this barrier
    // does not wait in practice)
    ADT_STAMP(2)
    asm volatile("s_barrier" ::: "memory");
    write_ds(krow0, hs0);
    ADT_STAMP(3)
    // ---- phase 3 and 4: dV^T += dO^T P, dK^T += Q^T dS per k-step of 16 queries and d-block; block 1's arithmetic rides on block 0's products
    const unsigned trb = trbase + slot;
#define ADT_DVDK(BLK, HP, HS, WITH_ARITH)                                                                                       \
    {                                                                                                                           \
      union { unsigned u[4]; bf16x8 v; } pf0, pf1, dsf0, dsf1;                                                                  \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) { pf0.u[e] = HP[e]; pf1.u[e] = HP[4 + e]; dsf0.u[e] = HS[e]; dsf1.u[e] = HS[4 + e]; } \
      TrFrag fo[4], fqq[4];                                                                                                     \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                                                        \
        ADT_TR2(fo[db], trb ^ static_cast<unsigned>(64 * db), 8192);                                                            \
        ADT_TR2(fqq[db], trb ^ static_cast<unsigned>(64 * db), 0);                                                              \
      }                                                                                                                         \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                        \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) {      /* k-step 0's fragment registers take k-step 1's once their MFMA has issued */ \
        mfma_acc(dv[BLK][db], tr_get(fo[db]), pf0.v);                                                                          \
        mfma_acc(dk[BLK][db], tr_get(fqq[db]), dsf0.v);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        ADT_TR2(fo[db], trb ^ static_cast<unsigned>(64 * db), 8192 + 16 * 256);                                                 \
        ADT_TR2(fqq[db], trb ^ static_cast<unsigned>(64 * db), 16 * 256);                                                       \
        if ((WITH_ARITH) == 1) { arith_quad1(db); __builtin_amdgcn_sched_barrier(0); }                                          \
        if ((WITH_ARITH) == 2 && kDrop == 1) { masks_from_hash(db, slice_hash(j + 1, 0, db), km0); __builtin_amdgcn_sched_barrier(0); } \
      }                                                                                                                         \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                        \
      _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                                                        \
        mfma_acc(dv[BLK][db], tr_get(fo[db]), pf1.v);                                                                          \
        mfma_acc(dk[BLK][db], tr_get(fqq[db]), dsf1.v);                                                                        \
        if ((WITH_ARITH) == 2 && kDrop == 1) { __builtin_amdgcn_sched_barrier(0); masks_from_hash(4 + db, slice_hash(j + 1, 0, 4 + db), km0); __builtin_amdgcn_sched_barrier(0); } \
      }                                                                                                                         \
    }
    if (!(kDbg & 4)) {
      ADT_DVDK(0, hp0, hs0, 1)
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) arith_quad1(t);
    }
    __builtin_amdgcn_sched_barrier(0);
    ADT_STAMP(4)
    write_ds(krow0 + 32, hs1);
    if (!(kDbg & 4)) {
      ADT_DVDK(1, hp1, hs1, 2)
    }
    __builtin_amdgcn_sched_barrier(0);
#undef ADT_DVDK
#undef ADT_CHAIN_STEP
#undef ADT_CHAIN_BEGIN
#undef ADT_UNIT
    ADT_STAMP(5)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next slice's tiles have landed; this wave's dS^T writes are done
    ADT_STAMP(6)
    asm volatile("s_barrier" ::: "memory");                       // ... and every wave's dS^T of this slice is in LDS
    ADT_STAMP(7)
    if (pub_pending >= 0) {                                       // the tile stored at the end of the last iteration has left (drained above)
      if (lane == 0) __hip_atomic_fetch_add(flag_of(pub_pending, kb), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pub_pending = -1;
    }
    fl = nstep ? static_cast<unsigned>(__builtin_amdgcn_readfirstlane(fv)) : 0u;

    ADT_STAMP(8)
    // ---- this iteration's reduction step, then this slice's own tile
    if (step) {
      if (late) {
        wait_flag(sjr, sn);
        land(sjr, sn);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      add_step(sjr, sn);
    }
    if (j > 0) hand_on(dq, j - 1);
    cur_jr = njr;
    cur_n = nn;
    ADT_STAMP(9)
  }
  };
  if (key_mask) run_slices(std::true_type{});
  else run_slices(std::false_type{});
#undef ADT_STAMP
  {                                                               // the last slice's dQ product (its dS^T image is complete: the loop ends on a barrier)
    f32x16 dq;
    if (!(kDbg & 2)) {
      ADT_DQ_BEGIN
      ADT_DQ_STEP(0) ADT_DQ_STEP(1) ADT_DQ_STEP(2) ADT_DQ_STEP(3) ADT_DQ_STEP(4) ADT_DQ_STEP(5) ADT_DQ_STEP(6) ADT_DQ_STEP(7)
      ADT_DQ_STEP(8) ADT_DQ_STEP(9) ADT_DQ_STEP(10) ADT_DQ_STEP(11) ADT_DQ_STEP(12) ADT_DQ_STEP(13) ADT_DQ_STEP(14) ADT_DQ_STEP(15)
      ADT_DQ_END
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) dq[i] = 0.f;
    }
    const int prev_pending = pub_pending;
    hand_on(dq, ns - 1);
    if (handoff) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        if (prev_pending >= 0) __hip_atomic_fetch_add(flag_of(prev_pending, kb), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(flag_of(ns - 1, kb), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      pub_pending = -1;
    }
  }
#undef ADT_DQ_BEGIN
#undef ADT_DQ_STEP
#undef ADT_DQ_END
#undef ADT_TR2
#undef ADT_TRX

#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int ki = key0 + 64 * wave + 32 * blk + r;
    store_transposed(dk[blk], a.scale, a.dk + static_cast<long>(b) * a.Sk * a.ldk + head * kDh, a.ldk, ki, a.Sk, lane);
    store_transposed(dv[blk], 1.0f, a.dv + static_cast<long>(b) * a.Sk * a.ldv + head * kDh, a.ldv, ki, a.Sk, lane);
  }
  // ---- behind the last slice: the reduction steps that were still to come, in batches: the K / dS^T / tile images are dead now and give
  // seven landing slots per wave, so the open steps' flags are polled together (a lane each), their tiles land together, and only the
  // additions run in sequence -- one flag latency and one DMA latency for the whole tail instead of one of each per step
  if (handoff) {
    __syncthreads();                                               // every wave has finished its last dQ product: the images may be overwritten
    constexpr int kSlots = kFbOffS / (4 * 4096);
    while (cur_n < 0) advance(cur_jr, cur_n);
    while (cur_jr < ns) {
      // lane i < kSlots: the i-th open step
      int ljr = cur_jr, ln = cur_n;
      for (int i = 0; i < kSlots; ++i)
        if (i < lane) advance(ljr, ln);
      const bool need = lane < kSlots && ljr < ns && ln != kb;
      unsigned spins = 0;
      for (;;) {
        unsigned f = 1u;
        if (need) f = __hip_atomic_load(flag_of(ljr, ln), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64(f == 0u) == 0ull) break;
        if (++spins > kFbSpinLimit) {                             // never in a healthy launch: report and carry on instead of hanging the GPU
          if (lane == 0) attn_bwd_report_giveup(fa, static_cast<long>(a.B) * a.H * ns * nkb * 4);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      int tjr = cur_jr, tn = cur_n, cnt = 0;
      for (; cnt < kSlots && tjr < ns; ++cnt, advance(tjr, tn)) land_to(tjr, tn, (cnt * 4 + wave) * 4096);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int i = 0; i < cnt; ++i, advance(cur_jr, cur_n)) {
        add_step_from(cur_jr, cur_n, smem_base + static_cast<unsigned>((i * 4 + wave) * 4096 + lane * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
}

static size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

// ---- the sticky give-up word (see attn_common.h)
namespace {
constexpr int kMaxDevices = 64;
struct GiveupWords { std::mutex mu; unsigned* host[kMaxDevices] = {}; unsigned* dev[kMaxDevices] = {}; };
GiveupWords& giveup_words() { static GiveupWords w; return w; }
}
int attn_bwd_giveup_word(unsigned** host_word, unsigned** device_alias) {
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return set_error(ADT_EHIP, "adt_attn_bwd: device index out of range");
  GiveupWords& w = giveup_words();
  std::lock_guard<std::mutex> lock(w.mu);
  if (!w.host[dev]) {
    void* h = nullptr; void* d = nullptr;
    ADT_HIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable));
    *static_cast<volatile unsigned*>(h) = 0u;
    ADT_HIP_TRY(hipHostGetDevicePointer(&d, h, 0));
    w.host[dev] = static_cast<unsigned*>(h);
    w.dev[dev] = static_cast<unsigned*>(d);
  }
  if (host_word) *host_word = w.host[dev];
  if (device_alias) *device_alias = w.dev[dev];
  return ADT_OK;
}
static int fused_ns(const adt_attn_desc* d) { return (d->q_len + kFbSlice - 1) / kFbSlice; }
static int fused_nkb(const adt_attn_desc* d) { return (d->k_len + kFbKeys - 1) / kFbKeys; }

size_t attn_bwd_fused_workspace_bytes(const adt_attn_desc* d) {
  const size_t bh = static_cast<size_t>(d->batch) * d->heads, ns = static_cast<size_t>(fused_ns(d));
  const size_t nkb = static_cast<size_t>(fused_nkb(d));
  size_t bytes = align256(bh * ns * nkb * 4 * 4 + 16) + 256;                      // flags (+ the give-up counter): zeroed every launch; + the experiment build's stamps at the very end
  bytes += align256(bh * ns * kFbSlice * 2 * 4);                                  // statistics
  if (nkb > 1) bytes += align256(bh * ns * nkb * 4096 * 4);                       // every key block's dQ^T tiles
  return bytes;
}

// Carves the workspace, launches the statistics kernel (which also zeroes the fan-in's flags) and fills the arguments both one-kernel
// backward forms share (this file's 4-wave kernel, attention_bwd_fused8.hip's 8-wave one).
int attn_bwd_fused_prepare(const adt_attn_desc* d, const AttnArgs& a, void* ws, size_t ws_bytes, hipStream_t st, FusedArgs* out) {
  if (ws_bytes < attn_bwd_fused_workspace_bytes(d) || !aligned16(ws)) return set_error(ADT_EINVAL, "adt_attn_bwd: workspace too small");
  FusedArgs fa{};
  {
    // an EARLIER launch on this device gave up (its dQ is incomplete): say so now, once, instead of training on -- the word is host
    // memory, so this read costs nothing and needs no synchronisation
    unsigned* hw = nullptr;
    if (int rc = attn_bwd_giveup_word(&hw, &fa.giveup_host)) return rc;
    const unsigned n = __atomic_exchange_n(hw, 0u, __ATOMIC_RELAXED);
    if (n) return set_error(ADT_EHIP, "adt_attn_bwd: waves of an earlier one-kernel backward on this device gave up waiting for a dQ tile (its dQ is incomplete)");
  }
  fa.a = a;
  fa.ns = fused_ns(d);
  fa.nkb = fused_nkb(d);
  { const char* e = getenv("ADT_FB_DBG"); fa.dbg = e ? atoi(e) : 0; }
  const size_t bh = static_cast<size_t>(d->batch) * d->heads, ns = static_cast<size_t>(fa.ns);
  unsigned char* p = static_cast<unsigned char*>(ws);
  const size_t flag_bytes = align256(bh * ns * static_cast<size_t>(fa.nkb) * 4 * 4 + 16);
  fa.stamps = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(ws) + attn_bwd_fused_workspace_bytes(d) - 128);
  fa.flags = reinterpret_cast<unsigned*>(p);
  p += flag_bytes;
  float* stats = reinterpret_cast<float*>(p);
  fa.stats = stats;
  p += align256(bh * ns * kFbSlice * 2 * 4);
  fa.part = reinterpret_cast<float*>(p);
  const long items = static_cast<long>(bh) * ns * kFbSlice;
  hipLaunchKernelGGL(attn_bwd_stats_kernel, dim3(static_cast<unsigned>((items + 15) / 16)), dim3(256), 0, st, a, stats, fa.ns, fa.flags, fa.nkb);
  const long n_tiles = static_cast<long>(fa.nkb) * static_cast<long>(bh);
  for (int x = 0; x < 8; ++x) fa.sched_total[x] = static_cast<unsigned>(n_tiles / 8 + (x < n_tiles % 8 ? 1 : 0));
  if (fa.dbg & 16) fa.sched_total[0] = static_cast<unsigned>(n_tiles);
  if (int rc = sched_counters(st, &fa.sched)) return rc;
  *out = fa;
  return ADT_OK;
}
// the give-up check both forms end with (tests / debugging: ADT_ATTN_BWD_CHECK=1)
int attn_bwd_fused_check(const FusedArgs& fa, hipStream_t st) {
  if (!getenv("ADT_ATTN_BWD_CHECK")) return ADT_OK;
  const size_t bh = static_cast<size_t>(fa.a.B) * fa.a.H, ns = static_cast<size_t>(fa.ns);
  unsigned gave_up = 0;
  ADT_HIP_TRY(hipStreamSynchronize(st));
  ADT_HIP_TRY(hipMemcpy(&gave_up, fa.flags + bh * ns * static_cast<size_t>(fa.nkb) * 4, sizeof(gave_up), hipMemcpyDeviceToHost));
  if (gave_up) return set_error(ADT_EHIP, "adt_attn_bwd: waves of the one-kernel backward gave up waiting for a dQ tile (result incomplete)");
  return ADT_OK;
}

int launch_attn_bwd_fused(const adt_attn_desc* d, const AttnArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
  static thread_local int lds_done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
#ifdef ADT_FB_EXPERIMENT
#define ADT_FB_DBGS(X) X(0) X(1) X(2) X(3) X(4) X(7) X(8) X(15) X(16) X(32)
#else
#define ADT_FB_DBGS(X) X(0)
#endif
  if (lds_done_for != dev) {
#define ADT_FB_ATTR(N)                                                                                                                      \
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<0, N>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds)); \
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<1, N>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds)); \
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<2, N>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
    ADT_FB_DBGS(ADT_FB_ATTR)
#undef ADT_FB_ATTR
    lds_done_for = dev;
  }
  FusedArgs fa{};
  if (int rc = attn_bwd_fused_prepare(d, a, ws, ws_bytes, st, &fa)) return rc;
  const size_t bh = static_cast<size_t>(d->batch) * d->heads;
  const long n_tiles = static_cast<long>(fa.nkb) * static_cast<long>(bh);
  bool launched = false;
#define ADT_FB_LAUNCH(N)                                                                                                                    \
  if (!launched && fa.dbg == N) {                                                                                                           \
    launched = true;                                                                                                                        \
    if (a.drop.on() && a.keep_bits) hipLaunchKernelGGL((attn_bwd_fused_kernel<2, N>), dim3(static_cast<unsigned>(n_tiles)), dim3(kFbThreads), kFbLds, st, fa); \
    else if (a.drop.on()) hipLaunchKernelGGL((attn_bwd_fused_kernel<1, N>), dim3(static_cast<unsigned>(n_tiles)), dim3(kFbThreads), kFbLds, st, fa);   \
    else hipLaunchKernelGGL((attn_bwd_fused_kernel<0, N>), dim3(static_cast<unsigned>(n_tiles)), dim3(kFbThreads), kFbLds, st, fa);         \
  }
  ADT_FB_DBGS(ADT_FB_LAUNCH)
#undef ADT_FB_LAUNCH
  if (!launched) return set_error(ADT_EINVAL, "adt_attn_bwd: ADT_FB_DBG value not built (experiment build only)");
  ADT_HIP_TRY(hipGetLastError());
  if (int rc = attn_bwd_fused_check(fa, st)) return rc;
  return ADT_OK;
}

}  // namespace adt
