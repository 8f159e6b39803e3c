// attention_bwd_fused8.hip -- the one-kernel attention backward (attention_bwd_fused.hip: five products, dQ summed over the key-block
// workgroups by the scheduled fan-in) with EIGHT waves per workgroup: two per SIMD.
//
// The 4-wave form keeps dK^T / dV^T of 64 keys per wave (256 accumulators: one wave per SIMD) and software-pipelines its two 32-key blocks
// against each other inside the wave; a lone wave issues a vector instruction every ~5 cycles, so the softmax / dS arithmetic (~700
// instructions per slice and wave) is what the slice costs.  Here a wave owns ONE 32-key block (128 accumulators, 256 registers in all): waves
// w and w + 4 share a SIMD and a key group (keys 64 (w & 3) .. + 63: block 0 on wave w, block 1 on wave w + 4), the SIMD's two instruction
// streams interleave in hardware -- one wave's products under the other's arithmetic -- and the vector pipe takes an instruction every two
// cycles.  Everything the 4-wave form shares between waves is unchanged (K image, dS^T image, Q | dO tile ring, statistics, keep bits, the
// fan-in's flags / tiles / schedule -- attn_bwd_fused_prepare), so the two forms are interchangeable launch by launch.  The extra work of a
// slice is split by ROLE: waves 0-3 (the block-0 waves) run the dQ product of the previous slice beside their arithmetic and publish the
// tile; waves 4-7 run the fan-in (landing tiles, additions, the final dQ store) of quarter w - 4.  Dropout: keep bits or none (a launch with
// hashed masks takes the 4-wave form).
#include <type_traits>

#include "attn_common.h"

namespace adt {

constexpr int kFbThreads = 512;
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
constexpr int kFbOffBits = kFbOffFlag + 16;         // keep bits of the slice (kDrop == 2): per ring slot 4 key groups x 256 B (lane l < 32: the word of key l of the
                                                    // group's block 0, l >= 32: of key l - 32 of block 1)
constexpr int kFbLds = kFbOffBits + 2 * 1024;       // 150,032 B
constexpr unsigned kFbSpinLimit = 1u << 24;         // polls of ~0.3 us each before a wave gives up (and reports it)


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
__global__ __launch_bounds__(kFbThreads, 1) void attn_bwd_fused8_kernel(FusedArgs fa) {
  static_assert(kDrop != 1, "hashed masks: the 4-wave form");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnArgs& a = fa.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave & 3, blk = wave >> 2;          // key group (64 keys; also the dQ d-block / fan-in quarter) and the wave's 32-key block of it
  const bool reducer = blk == 1;                      // waves 4-7: the fan-in; waves 0-3: the dQ product and its publication
#define ADT_ITEM_STAMP(K)                                                                                                 \
  if ((kDbg & 96) && wave == ((kDbg & 64) ? 4 : 0) && blockIdx.x == 600) {                                               \
    const unsigned long long tnow = __builtin_amdgcn_s_memtime();                                                        \
    if (lane == 0) fa.stamps[K] = tnow;                                                                                  \
  }
  ADT_ITEM_STAMP(12)

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
  const unsigned* bits_g = kDrop == 2 ? a.keep_bits + ((static_cast<long>(bh) * a.bits_nq) * a.bits_nk + (kb * (kFbKeys / 32) + 2 * grp)) * 32 : nullptr;
  const int klen = a.key_len ? __builtin_amdgcn_readfirstlane(a.key_len[b]) : a.Sk;      // (a loaded value is "divergent" to the compiler: make it scalar)
  const float sl2 = a.scale * kLog2e;
  const bool key_mask = a.causal || key0 + kFbKeys > klen || key0 + kFbKeys > a.Sk;      // block-uniform, in a scalar register

  // ---- staging: the K image once; Q | dO | statistics of slice j into ring slot j & 1
  const int lrow = lane >> 4, lchunk = lane & 15;
  const int ldq_i = static_cast<int>(a.ldq), ldo_i = static_cast<int>(a.ldo), ldk_i = static_cast<int>(a.ldk);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int g = 8 * wave + i, row = 4 * g + lrow;
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
      const int g = 4 * grp + i, rg = g & 7, row = 4 * rg + lrow;             // g 0..7: Q, 8..15: dO (wave-uniform)
      const int chunk = lchunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
      int gr = j * kFbSlice + row;
      gr = gr < a.Sq ? gr : a.Sq - 1;                             // rows past the end repeat the last valid row (their P is 0 by the statistics)
      const unsigned short* src = g < 8 ? qb + static_cast<unsigned>(gr * ldq_i + chunk * 8) : dob + static_cast<unsigned>(gr * ldo_i + chunk * 8);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(slot + g * 1024), 16, 0, 0);
    }
    if (grp == 3)                                                 // 64 lanes x 4 bytes = the slice's 32 x {nl, nd}
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stat_g + j * kFbSlice * 2 + lane_o),
                                       (__attribute__((address_space(3))) void*)(smem + kFbOffS + (j & 1) * 256), 4, 0, 0);
    if (kDrop == 2) {                                             // the keep words of the key group's two blocks: lane = (block, key)
      const int key = lane_o & 31;
      const unsigned* src = bits_g + (static_cast<long>(j) * a.bits_nk + (lane_o >> 5)) * 32 + keep_bits_word(key);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + kFbOffBits + (j & 1) * 1024 + grp * 256), 4, 0, 0);
    }
  };
  // (the slices' staging is the fan-in waves' job: waves 0-3 carry the dQ product and are the longer instruction stream of every SIMD)
  if (reducer) issue_slice(0);

  // V of this wave's 64 keys: the B operand of dP, in registers for the whole kernel
  bf16x8 vf[8];
  frags_from_global(vb, a.ldv, key0 + 64 * grp + 32 * blk + r, a.Sk, lane, vf);

  f32x16 dk[4], dv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }


  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // All LDS reads of the loop are inline asm with hand-placed lgkmcnt waits and a small ring of operand registers: left to the compiler the
  // reads of a whole chain are hoisted in front of it and the kernel spills (a reload of a spilled value also drains the DMA in flight).
  // Addresses come from two lane constants by XOR: the swizzle is an XOR of address bits 4..7, so the k-step / d-block enters as
  // `^ 32 s` / `^ 64 db` (attention.hip, dK/dV kernel).
  const unsigned smem_base = lds_off(smem);
  unsigned rowbase = smem_base + static_cast<unsigned>(256 * r + 16 * (h ^ (((r & 3) << 2) | ((r >> 2) & 3))));
  // Lane constants that only one phase uses are RE-MADE there from the lane number (a few integer instructions per slice) instead of
  // living in registers across the S' / dP chain, which has none to spare (128 + 128 at two waves per SIMD).
  auto lane_now = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  auto make_trbase = [&](int l) {         // transposed reads of the Q | dO tiles and of the K image (dV / dK products, dQ product)
    const int i = l & 15, g4 = (l >> 4) & 1, row = 4 * (l >> 5) + (i >> 2);
    return smem_base + static_cast<unsigned>(8 * (i & 1) + swz(row, 2 * g4 + ((i & 3) >> 1)));
  };
  auto make_xbase = [&](int l) {          // transposed reads of the dS^T image (dQ product)
    const int i = l & 15, g4 = (l >> 4) & 1, hh = l >> 5, row = 4 * hh + (i >> 2);
    return smem_base + static_cast<unsigned>(8 * (i & 1) + 64 * row + 16 * ((2 * g4 + ((i & 3) >> 1)) ^ hh));
  };
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
  auto flag_of = [&](int jj, int src) { return fa.flags + ((fl_bh + jj) * nkb + src) * 4 + grp; };
  auto part_of = [&](int jj, int src) { return fa.part + (((fl_bh + jj) * nkb + src) * 4 + grp) * 1024; };
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
  auto land = [&](int jr, int n) { land_to(jr, n, kFbOffLand + grp * 4096); };
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
      unsigned short* p = a.dq + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * kDh + 32 * grp + 8 * (lane_o >> 5);
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
  auto stash_of = [&](int l) { return smem_base + static_cast<unsigned>(kFbOffStash + grp * 4096 + l * 16); };
  auto add_step_from = [&](int jr, int n, unsigned land_a) {       // running sum (+)= landed tile (at LDS address land_a + 1024 g); the last step stores dQ
    const unsigned stash_a = stash_of(lane_now());
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
  auto add_step = [&](int jr, int n) { add_step_from(jr, n, stash_of(lane_now()) + 4u * 4096u); };

#define ADT_STAMP(K)                                                                                                      \
  if ((kDbg & 96) && j == 12 && wave == ((kDbg & 64) ? 4 : 0) && blockIdx.x == 600) {                                                        \
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
  const unsigned ka_a = trbase ^ static_cast<unsigned>(64 * grp), xb_a = xbase + static_cast<unsigned>(kFbOffX); \
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
  const uint64_t km_none[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // (arith_pair's hashed-mask operand: not used here)
  for (int j = 0; j < ns; ++j) {
    asm volatile("" : "+v"(rowbase));                             // keep the per-k-step addresses derived from it out of loop-invariant registers
    // ---- the fan-in's step of this iteration (waves 4-7): its tile is brought in now, added at the end of the iteration
    const int sjr = cur_jr, sn = cur_n;
    const bool step = reducer && valid(sjr, sn);                  // wave-uniform
    int njr = cur_jr, nn = cur_n;
    advance(njr, nn);
    bool late = false;
    if (step) {
      if (sn == kb || fl != 0u) land(sjr, sn);                    // own tile: drained two iterations ago; others: flag seen last iteration
      else late = true;
    }
    const bool nstep = reducer && valid(njr, nn) && nn != kb;
    unsigned fv = 0;
    if (nstep && lane == 0) fv = __hip_atomic_load(flag_of(njr, nn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (reducer && j + 1 < ns) issue_slice(j + 1);
    const unsigned slot = static_cast<unsigned>(kFbOffT + (j & 1) * kFbTile);
    const unsigned stat_a = smem_base + static_cast<unsigned>(kFbOffS + (j & 1) * 256 + 16 * h);       // + 32 g: queries 8 g + 4 h .. + 3
    // ---- a wave's slice: the S' / dP chains of its block, the arithmetic (on waves 0-3 beside the previous slice's dQ product), the dS^T
    // write, the dV^T / dK^T products.  The SIMD's other wave runs the same sequence on the group's other block: the hardware interleaves
    // the two streams (one wave's products under the other's arithmetic).
    f32x16 st, dp;
    unsigned hp[8], hs[8];
    f32x4 ndv[4];                                                 // the slice's row constants nd of this lane's 16 query rows
    bf16x8 fq[3], fd[3], fk[3];                                   // operand ring of the chains: two k-steps ahead
    const unsigned tq_a = rowbase + slot;
    const unsigned kr_a = rowbase + static_cast<unsigned>((64 * grp + 32 * blk) * 256);
    const int krow = 64 * grp + 32 * blk + r, ki = key0 + krow;
#define ADT_UNIT(U, S)                                                                                                          \
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:8192\n\tds_read_b128 %2, %4"                                \
                 : "=&v"(fq[U]), "=&v"(fd[U]), "=&v"(fk[U])                                                                      \
                 : "v"(tq_a ^ static_cast<unsigned>(32 * (S))), "v"(kr_a ^ static_cast<unsigned>(32 * (S))) : "memory")
    // lane constants of the masked form: the key-padding term, the causal term, the validity of this lane's key, and the first query row
    // of the slice that may see the key (causal: row q of the slice is masked iff q < key - slice start)
    const float cau2 = a.causal ? a.mask_value * kLog2e : 0.f;
    const float pad2 = ki >= klen ? a.mask_value * kLog2e : 0.f;
    const float kval = ki < a.Sk ? 1.f : 0.f;
    const int qrel = ki - j * kFbSlice;
    // kDrop == 2: wb = the lane's keep word of the block, shifted right by 4 h (so that the bit of accumulator register i sits at
    // position acc_row(i, 0)); v_bfe_i32 spreads the bit over the register, an AND with 1 / (1 - p) gives the keep scale
    unsigned wb = 0;
    // Two elements per call, on register PAIRS (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: accumulator registers 2 m, 2 m + 1 are an
    // aligned pair, and so are their row constants); attention_bwd_fused.hip has the derivation.
    auto arith_pair = [&](int m) {
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
      } else {                                                   // bit acc_row(i, 0) of the lane's (shifted) keep word, spread over the register
        unsigned t0, t1;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t0) : "v"(wb), "n"(c0 + 8 * g));
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t1) : "v"(wb), "n"(c0 + 8 * g + 1));
        pd[0] = __uint_as_float(__float_as_uint(pv[0]) & t0);
        pd[1] = __uint_as_float(__float_as_uint(pv[1]) & t1);
        ds = pd * d2 + pv * nd2;
      }
      // (P comes straight from v_exp_f32 / a compiler-visible AND: the conversion of P stays compiler-visible too -- an asm instruction gets
      // none of the wait states a transcendental's consumer needs)
      if (kDrop == 0) hp[m] = __builtin_bit_cast(unsigned, __builtin_convertvector(pd, bf16x2c));
      else hp[m] = pack2(pd[0], pd[1]);
      hs[m] = pack2(ds[0], ds[1]);
    };
    (void)km_none;

    ADT_STAMP(0)
    // ---- phase 1: S', dP of the wave's block: two accumulators take turns; the other wave of the SIMD fills what the ~100-cycle
    // accumulator latency leaves open
    {
      if (kDrop == 2) {                                           // (first: every counted wait below then covers it too)
        const unsigned bw_a = smem_base + static_cast<unsigned>(kFbOffBits + (j & 1) * 1024 + grp * 256 + 128 * blk + 4 * r);
        asm volatile("ds_read_b32 %0, %1" : "=&v"(wb) : "v"(bw_a) : "memory");
      }
      {
        // the score chain starts from the row constant nl: read from LDS STRAIGHT INTO the accumulator's quarters; dP starts from 0
        f32x4 q0[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(q0[g]) : "v"(stat_a), "i"(32 * g) : "memory");
        ADT_UNIT(0, 0); ADT_UNIT(1, 1); ADT_UNIT(2, 2);
        st = __builtin_shufflevector(__builtin_shufflevector(q0[0], q0[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(q0[2], q0[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                     0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s < 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");      // (in order: the row constant and the keep word, read first, are back too)
        else if (s < 7) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_vgpr(st, fq[s % 3], fk[s % 3]);
        if (s == 0) mfma_vgpr0(dp, fd[0], vf[0]);
        else mfma_vgpr(dp, fd[s % 3], vf[s]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < 8) {
          if (s % 3 == 0) ADT_UNIT(0, s + 3);
          else if (s % 3 == 1) ADT_UNIT(1, s + 3);
          else ADT_UNIT(2, s + 3);
        }
      }
      // (the row constants nd only now: during the chain their sixteen registers belong to the operand ring)
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ndv[g]) : "v"(stat_a), "i"(128 + 32 * g) : "memory");
      mfma_settle(st, dp);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (kDrop == 2) wb >>= 4 * h;                               // (behind the chain's lgkmcnt(0): the word has arrived)
      ADT_STAMP(1)
    }
    // ---- phase 2: the arithmetic (the SIMD's other wave has products to run beside it)
#pragma unroll
    for (int m = 0; m < 8; ++m) arith_pair(m);
    __builtin_amdgcn_sched_barrier(0);
    ADT_STAMP(10)
    // ---- phase 2b (waves 0-3): the PREVIOUS slice's dQ product over the workgroup's 256 keys (its dS^T image is complete since that slice's
    // barrier B and is overwritten only behind barrier A below), published right away.  Placed here so that the SIMD's two waves run
    // complementary work through the whole stretch: chain | fan-in addition, arithmetic | chain, dQ product | arithmetic.
    if (!reducer && j > 0) {
      f32x16 dq;
      if (!(kDbg & 2)) {
        // (a ring of two units and ONE accumulation chain -- the 4-wave form's in-loop order, so dQ comes out bit for bit the same; the
        // P / dS words of this slice are live here, registers are scarce, and the SIMD's other wave has the vector pipe busy meanwhile)
        const int l = lane_now();
        const unsigned ka_a = make_trbase(l) ^ static_cast<unsigned>(64 * grp), xb_a = make_xbase(l) + static_cast<unsigned>(kFbOffX);
        TrFrag ka[2], xb[2];
        ADT_TR2(ka[0], ka_a, 0);
        ADT_TRX(xb[0], xb_a, 0);
        ADT_TR2(ka[1], ka_a, 4096);
        ADT_TRX(xb[1], xb_a, 1024);
#define ADT_DQ1_STEP(KK)                                                                          \
        if ((KK) + 1 < 16) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                     \
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                   \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if ((KK) == 0) mfma_vgpr0(dq, tr_get(ka[0]), tr_get(xb[0]));                              \
        else mfma_vgpr(dq, tr_get(ka[(KK) & 1]), tr_get(xb[(KK) & 1]));                           \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if ((KK) + 2 < 16) {                                                                      \
          ADT_TR2(ka[(KK) & 1], ka_a, ((KK) + 2) * 4096);                                         \
          ADT_TRX(xb[(KK) & 1], xb_a, ((KK) + 2) * 1024);                                         \
        }
        ADT_DQ1_STEP(0) ADT_DQ1_STEP(1) ADT_DQ1_STEP(2) ADT_DQ1_STEP(3) ADT_DQ1_STEP(4) ADT_DQ1_STEP(5) ADT_DQ1_STEP(6) ADT_DQ1_STEP(7)
        ADT_DQ1_STEP(8) ADT_DQ1_STEP(9) ADT_DQ1_STEP(10) ADT_DQ1_STEP(11) ADT_DQ1_STEP(12) ADT_DQ1_STEP(13) ADT_DQ1_STEP(14) ADT_DQ1_STEP(15)
#undef ADT_DQ1_STEP
        mfma_settle(dq);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[i] = 0.f;
      }
      hand_on(dq, j - 1);
    }
    // the dS^T image is single-buffered: waves 0-3 must have finished the previous slice's dQ product before anybody overwrites it
    ADT_STAMP(2)
    asm volatile("s_barrier" ::: "memory");
    // dS^T of the block to LDS for the dQ product: x_off(key, g) + 8 h = 64 key + ((16 s + 8 h) ^ 16 g), s = (key >> 2) & 3 -- one lane
    // constant (re-made per slice: four hoisted address registers are four too many here) and an XOR per write
    {
      const int l = lane_now(), rr = l & 31, hh = l >> 5, key = 64 * grp + 32 * blk + rr;
      const unsigned xw_lo = static_cast<unsigned>(16 * ((key >> 2) & 3) + 8 * hh), xw_hi = smem_base + static_cast<unsigned>(kFbOffX + 64 * key);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2 w2 = make_uint2(hs[2 * g], hs[2 * g + 1]);
        asm volatile("ds_write_b64 %0, %1" :: "v"(xw_hi + (xw_lo ^ static_cast<unsigned>(16 * g))), "v"(w2) : "memory");
      }
    }
    ADT_STAMP(3)
    // ---- phase 3: dV^T += dO^T P, dK^T += Q^T dS per k-step of 16 queries and d-block
    const unsigned trb = make_trbase(lane_now()) + slot;
    if (!(kDbg & 4)) {
      union { unsigned u[4]; bf16x8 v; } pf0, pf1, dsf0, dsf1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { pf0.u[e] = hp[e]; pf1.u[e] = hp[4 + e]; dsf0.u[e] = hs[e]; dsf1.u[e] = hs[4 + e]; }
      TrFrag fo[4], fqq[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        ADT_TR2(fo[db], trb ^ static_cast<unsigned>(64 * db), 8192);
        ADT_TR2(fqq[db], trb ^ static_cast<unsigned>(64 * db), 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < 4; ++db) {      /* k-step 0's fragment registers take k-step 1's once their MFMA has issued */
        mfma_acc(dv[db], tr_get(fo[db]), pf0.v);
        mfma_acc(dk[db], tr_get(fqq[db]), dsf0.v);
        __builtin_amdgcn_sched_barrier(0);
        ADT_TR2(fo[db], trb ^ static_cast<unsigned>(64 * db), 8192 + 16 * 256);
        ADT_TR2(fqq[db], trb ^ static_cast<unsigned>(64 * db), 16 * 256);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        mfma_acc(dv[db], tr_get(fo[db]), pf1.v);
        mfma_acc(dk[db], tr_get(fqq[db]), dsf1.v);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#undef ADT_UNIT
    ADT_STAMP(5)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next slice's tiles have landed; this wave's dS^T writes are done
    ADT_STAMP(6)
    asm volatile("s_barrier" ::: "memory");                       // ... and every wave's dS^T of this slice is in LDS
    ADT_STAMP(7)
    if (pub_pending >= 0) {                                       // (waves 0-3) the tile stored at the end of the last iteration has left (drained above)
      if (lane == 0) __hip_atomic_fetch_add(flag_of(pub_pending, kb), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pub_pending = -1;
    }
    fl = nstep ? static_cast<unsigned>(__builtin_amdgcn_readfirstlane(fv)) : 0u;

    ADT_STAMP(8)
    // ---- this iteration's reduction step (waves 4-7), the previous slice's own tile (waves 0-3)
    if (step) {
      if (late) {
        wait_flag(sjr, sn);
        land(sjr, sn);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      add_step(sjr, sn);
    }
    cur_jr = njr;
    cur_n = nn;
    ADT_STAMP(11)
  }
  };
  ADT_ITEM_STAMP(13)
  if (key_mask) run_slices(std::true_type{});
  else run_slices(std::false_type{});
  ADT_ITEM_STAMP(14)
#undef ADT_STAMP
  if (!reducer) {                                                 // (waves 0-3) the last slice's dQ product (its dS^T image is complete: the loop ends on a barrier)
    f32x16 dq;
    if (!(kDbg & 2)) {
      const int l = lane_now();
      const unsigned trbase = make_trbase(l), xbase = make_xbase(l);
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

  // (dK / dV in FRONT of the tail: behind it -- a wave may end with stores in flight -- the kernel was 2-4 % slower: here their drain
  // passes under the wait for the head's other key blocks)
  {
    const int ki = key0 + 64 * grp + 32 * blk + r;
    store_transposed(dk, a.scale, a.dk + static_cast<long>(b) * a.Sk * a.ldk + head * kDh, a.ldk, ki, a.Sk, lane);
    store_transposed(dv, 1.0f, a.dv + static_cast<long>(b) * a.Sk * a.ldv + head * kDh, a.ldv, ki, a.Sk, lane);
  }
  // ---- behind the last slice: the reduction steps that were still to come, in batches: the K / dS^T / tile images are dead now and give
  // seven landing slots per wave, so the open steps' flags are polled together (a lane each), their tiles land together, and only the
  // additions run in sequence -- one flag latency and one DMA latency for the whole tail instead of one of each per step
  if (handoff) __syncthreads();                                    // every dQ wave has finished its last product: the images may be overwritten
  if (handoff && reducer) {
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
      for (; cnt < kSlots && tjr < ns; ++cnt, advance(tjr, tn)) land_to(tjr, tn, (cnt * 4 + grp) * 4096);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int i = 0; i < cnt; ++i, advance(cur_jr, cur_n)) {
        add_step_from(cur_jr, cur_n, smem_base + static_cast<unsigned>((i * 4 + grp) * 4096 + lane * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
  ADT_ITEM_STAMP(15)
#undef ADT_ITEM_STAMP
}

int launch_attn_bwd_fused8(const adt_attn_desc* d, const AttnArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
  static thread_local int lds_done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (lds_done_for != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused8_kernel<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused8_kernel<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
    lds_done_for = dev;
  }
  if (a.drop.on() && !a.keep_bits) return set_error(ADT_EINVAL, "adt_attn_bwd: the 8-wave backward takes keep bits or no dropout");
  FusedArgs fa{};
  if (int rc = attn_bwd_fused_prepare(d, a, ws, ws_bytes, st, &fa)) return rc;
  const long n_tiles = static_cast<long>(fa.nkb) * static_cast<long>(d->batch) * d->heads;
#ifdef ADT_FB_EXPERIMENT      // ADT_FB_DBG=32 / 64: cycle stamps of wave 0's / wave 4's phases in one slice (tools/probe/attn_bwd_stamps.py)
  if (fa.dbg == 32 || fa.dbg == 64) {
    static thread_local int exp_done_for = -1;
    if (exp_done_for != dev) {
      ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused8_kernel<0, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
      ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused8_kernel<2, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
      ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused8_kernel<0, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
      ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused8_kernel<2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
      exp_done_for = dev;
    }
    const dim3 grid(static_cast<unsigned>(n_tiles));
    if (fa.dbg == 32) {
      if (a.drop.on()) hipLaunchKernelGGL((attn_bwd_fused8_kernel<2, 32>), grid, dim3(kFbThreads), kFbLds, st, fa);
      else hipLaunchKernelGGL((attn_bwd_fused8_kernel<0, 32>), grid, dim3(kFbThreads), kFbLds, st, fa);
    } else {
      if (a.drop.on()) hipLaunchKernelGGL((attn_bwd_fused8_kernel<2, 64>), grid, dim3(kFbThreads), kFbLds, st, fa);
      else hipLaunchKernelGGL((attn_bwd_fused8_kernel<0, 64>), grid, dim3(kFbThreads), kFbLds, st, fa);
    }
    ADT_HIP_TRY(hipGetLastError());
    return attn_bwd_fused_check(fa, st);
  }
#endif
  if (fa.dbg != 0) return set_error(ADT_EINVAL, "adt_attn_bwd: ADT_FB_DBG value not built for the 8-wave form");
  if (a.drop.on()) hipLaunchKernelGGL((attn_bwd_fused8_kernel<2, 0>), dim3(static_cast<unsigned>(n_tiles)), dim3(kFbThreads), kFbLds, st, fa);
  else hipLaunchKernelGGL((attn_bwd_fused8_kernel<0, 0>), dim3(static_cast<unsigned>(n_tiles)), dim3(kFbThreads), kFbLds, st, fa);
  ADT_HIP_TRY(hipGetLastError());
  return attn_bwd_fused_check(fa, st);
}

}  // namespace adt
