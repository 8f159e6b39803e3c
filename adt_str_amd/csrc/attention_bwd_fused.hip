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
// order: a workgroup only ever waits for a LOWER ticket of its own counter (or for the end of the previous counter's range), i.e.
// for a workgroup that is already running or finished -- no assumption about dispatch order, and spins are bounded.
// delta = rowsum(O * dO) and -lse / scale are prepared per query by a small kernel in front (attn_bwd_stats_kernel).
#include "attn_common.h"

namespace adt {

constexpr int kFbThreads = 256;
constexpr int kFbKeys = 256;                        // keys per workgroup
constexpr int kFbSlice = 32;                        // queries per step
constexpr int kFbKimg = kFbKeys * 256;              // K rows of the workgroup's keys (swizzled 256-byte rows)
constexpr int kFbX = kFbKeys * 64;                  // dS^T of one slice: [key][32 q] bf16
constexpr int kFbTile = 2 * kFbSlice * 256;         // Q rows | dO rows of one slice
constexpr int kFbOffX = kFbKimg;
constexpr int kFbOffT = kFbOffX + 2 * kFbX;
constexpr int kFbOffS = kFbOffT + 2 * kFbTile;      // per slice -lse / scale [32] | -delta [32]
constexpr int kFbOffFlag = kFbOffS + 2 * 256;
constexpr int kFbLds = kFbOffFlag + 16;             // 131,600 B
constexpr unsigned kFbSpinLimit = 1u << 24;         // polls of ~0.3 us each before a wave gives up (and reports it)

struct FusedArgs {
  AttnArgs a;
  const float* stats;                 // [B*H][ns][2][32]: -lse / scale of the slice's queries, then -delta
  float* part;                        // [B*H][ns][4 waves][1024] running dQ^T sums (nkb > 1)
  unsigned* flags;                    // [B*H][ns][4] + 4 words: [0] of the tail = number of waves that gave up waiting
  unsigned* sched; unsigned sched_total[8];
  int nkb, ns;
  int dbg;                            // timing experiments only (ADT_FB_DBG): 1 no hand-off, 2 no dQ product, 4 no dV / dK products, 8 no S / dP chains
};

// {-lse / scale, -delta} per (batch, head, query), queries padded to whole slices ({-1e30, 0}: P = 0 there)
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(AttnArgs a, float* __restrict__ stats, int ns) {
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
    nl = -a.lse[(static_cast<long>(b) * a.H + head) * a.Sq + q] / a.scale;
    nd = -s;
  }
  if (c == 0) {                                                    // per slice of 32 queries: [nl x 32][nd x 32]
    float* sl = stats + ((static_cast<long>(b) * a.H + head) * sqp + (q & ~31)) * 2 + (q & 31);
    sl[0] = nl;
    sl[32] = nd;
  }
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

template <bool kDrop>
__global__ __launch_bounds__(kFbThreads, 1) void attn_bwd_fused_kernel(FusedArgs fa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnArgs& a = fa.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this workgroup's (batch, head, key block): ticket v of the XCD group's counter = logical tile slice0 + v
  const int n_tiles = fa.nkb * a.B * a.H;
  const int q8 = n_tiles >> 3, r8 = n_tiles & 7, xg = blockIdx.x & 7;
  const int slice0 = xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8, slice_n = q8 + (xg < r8 ? 1 : 0);
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
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  const float sl2 = a.scale * kLog2e;
  const bool key_mask = a.causal || key0 + kFbKeys > klen || key0 + kFbKeys > a.Sk;      // block-uniform

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
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stat_g + j * kFbSlice * 2 + lane),
                                       (__attribute__((address_space(3))) void*)(smem + kFbOffS + (j & 1) * 256), 4, 0, 0);
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
  const unsigned rowbase = smem_base + static_cast<unsigned>(256 * r + 16 * (h ^ (((r & 3) << 2) | ((r >> 2) & 3))));
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

  for (int j = 0; j < ns; ++j) {
    if (j + 1 < ns) issue_slice(j + 1);
    const unsigned slot = static_cast<unsigned>(kFbOffT + (j & 1) * kFbTile);
    const unsigned stat_a = smem_base + static_cast<unsigned>(kFbOffS + (j & 1) * 256 + 16 * h);       // + 32 g: queries 8 g + 4 h .. + 3
    unsigned char* xs = smem + kFbOffX + (j & 1) * kFbX;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int krow = 64 * wave + 32 * blk + r;                 // row of the K image = key of this lane inside the workgroup
      const int ki = key0 + krow;
      // ---- S' = Q K^T - lse / scale, dP = dO V^T: operands through a ring of two k-step units
      f32x16 st, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
      if (!(fa.dbg & 8)) {
        f32x4 c[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c[g]) : "v"(stat_a), "i"(32 * g) : "memory");
        const unsigned tq_a = rowbase + slot, kr_a = rowbase + static_cast<unsigned>((64 * wave + 32 * blk) * 256);
        bf16x8 fq[2], fd[2], fk[2];
#define ADT_UNIT(U, S)                                                                                                          \
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:8192\n\tds_read_b128 %2, %4"                            \
                     : "=&v"(fq[U]), "=&v"(fd[U]), "=&v"(fk[U])                                                                  \
                     : "v"(tq_a ^ static_cast<unsigned>(32 * (S))), "v"(kr_a ^ static_cast<unsigned>(32 * (S))) : "memory")
        ADT_UNIT(0, 0); ADT_UNIT(1, 1);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");       // the four statistics reads are back (in order)
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = c[i >> 2][i & 3]; dp[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          if (s < 7) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[s & 1], fk[s & 1], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[s & 1], vf[blk][s], dp, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (s + 2 < 8) {
            if (s & 1) ADT_UNIT(1, s + 2);
            else ADT_UNIT(0, s + 2);
          }
        }
#undef ADT_UNIT
      }
      __builtin_amdgcn_sched_barrier(0);
      // -delta of the lane's 16 query rows, and the first k-step's transposed dO / Q fragments, while the arithmetic runs
      f32x4 ndv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ndv[g]) : "v"(stat_a), "i"(128 + 32 * g) : "memory");
      // keep masks: one 64-bit lane mask per accumulator register (the two lanes of a key pair share every hash: dropout.h)
      uint64_t km[16];
      if (kDrop) {
        const uint64_t even = 0x5555555555555555ull, odd = 0xaaaaaaaaaaaaaaaaull;
        const unsigned headpair = static_cast<unsigned>(static_cast<uint64_t>(bh) * a.Sq * sk_pairs) + static_cast<unsigned>(ki >> 1);
        const unsigned vbq = headpair + static_cast<unsigned>(j * kFbSlice + 16 * static_cast<int>(par) + 4 * h) * sk_pairs;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned hh = mix32((vbq + static_cast<unsigned>((e & 3) + 8 * (e >> 2)) * sk_pairs) ^ key2);
          const uint64_t c_lo = __builtin_amdgcn_ballot_w64((hh << 16) >= thr16);       // decision of the pair's even key
          const uint64_t c_hi = __builtin_amdgcn_ballot_w64(hh >= thr16);               // ... of its odd key
          km[e] = (c_lo & even) | ((c_hi & even) << 1);
          km[e + 8] = (c_hi & odd) | ((c_lo & odd) >> 1);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      unsigned hp[8], hs[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        float pv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int i = 2 * m + e;
          if (key_mask) {
            const int qi = j * kFbSlice + acc_row(i, h);
            const float tt = fmaf(st[i], sl2, mask_add(a, qi, ki, klen) * kLog2e);
            pv[e] = ki < a.Sk ? __builtin_amdgcn_exp2f(tt) : 0.f;
          } else {
            pv[e] = __builtin_amdgcn_exp2f(st[i] * sl2);
          }
        }
        float pd[2], ds[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int i = 2 * m + e;
          const float ndi = ndv[i >> 2][i & 3];
          if (kDrop) {
            float ks;
            asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(ks) : "v"(a.drop.inv_keep), "s"(km[i]));
            pd[e] = pv[e] * ks;
            ds[e] = pv[e] * fmaf(dp[i], ks, ndi);
          } else {
            pd[e] = pv[e];
            ds[e] = pv[e] * (dp[i] + ndi);
          }
        }
        hp[m] = pack2(pd[0], pd[1]);
        hs[m] = pack2(ds[0], ds[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // dS^T of this block to LDS for the dQ product (read by every wave after the slice's barrier)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2 w2 = make_uint2(hs[2 * g], hs[2 * g + 1]);
        asm volatile("ds_write_b64 %0, %1" :: "v"(lds_off(xs) + x_off(krow, g) + 8 * h), "v"(w2) : "memory");
      }
      // ---- dV^T += dO^T P, dK^T += Q^T dS: per k-step of 16 queries, four d-blocks
      if (!(fa.dbg & 4)) {
        const unsigned trb = trbase + slot;
        union { unsigned u[4]; bf16x8 v; } pf0, pf1, dsf0, dsf1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pf0.u[e] = hp[e]; pf1.u[e] = hp[4 + e]; dsf0.u[e] = hs[e]; dsf1.u[e] = hs[4 + e]; }
        TrFrag fo[4], fq[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          ADT_TR2(fo[db], trb ^ static_cast<unsigned>(64 * db), 8192);
          ADT_TR2(fq[db], trb ^ static_cast<unsigned>(64 * db), 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 4; ++db) {                    // the registers of k-step 0's fragments take k-step 1's as soon as their MFMA has issued
          dv[blk][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(fo[db]), pf0.v, dv[blk][db], 0, 0, 0);
          dk[blk][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(fq[db]), dsf0.v, dk[blk][db], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          ADT_TR2(fo[db], trb ^ static_cast<unsigned>(64 * db), 8192 + 16 * 256);
          ADT_TR2(fq[db], trb ^ static_cast<unsigned>(64 * db), 16 * 256);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          dv[blk][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(fo[db]), pf1.v, dv[blk][db], 0, 0, 0);
          dk[blk][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(fq[db]), dsf1.v, dk[blk][db], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next slice's tiles have landed; this wave's dS^T writes are done
    asm volatile("s_barrier" ::: "memory");                       // ... and every wave's dS^T of this slice is in LDS

    // ---- dQ^T, d-block `wave`, over the workgroup's 256 keys (16 k-steps of 16 keys, two per wait)
    f32x16 dq;
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[i] = 0.f;
    if (!(fa.dbg & 2)) {
      const unsigned ka_a = trbase ^ static_cast<unsigned>(64 * wave), xb_a = xbase + static_cast<unsigned>(kFbOffX + (j & 1) * kFbX);
      TrFrag ka[2], xb[2];
      ADT_TR2(ka[0], ka_a, 0);
      ADT_TRX(xb[0], xb_a, 0);
#define ADT_DQ_STEP(KK)                                                                           \
      if ((KK) + 1 < 16) {                                                                        \
        ADT_TR2(ka[((KK) + 1) & 1], ka_a, ((KK) + 1) * 4096);                                     \
        ADT_TRX(xb[((KK) + 1) & 1], xb_a, ((KK) + 1) * 1024);                                     \
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                        \
      } else {                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(ka[(KK) & 1]), tr_get(xb[(KK) & 1]), dq, 0, 0, 0);   \
      __builtin_amdgcn_sched_barrier(0);
      ADT_DQ_STEP(0) ADT_DQ_STEP(1) ADT_DQ_STEP(2) ADT_DQ_STEP(3) ADT_DQ_STEP(4) ADT_DQ_STEP(5) ADT_DQ_STEP(6) ADT_DQ_STEP(7)
      ADT_DQ_STEP(8) ADT_DQ_STEP(9) ADT_DQ_STEP(10) ADT_DQ_STEP(11) ADT_DQ_STEP(12) ADT_DQ_STEP(13) ADT_DQ_STEP(14) ADT_DQ_STEP(15)
#undef ADT_DQ_STEP
    }

    // ---- ordered hand-off of the running sum (this wave's 32 d x 32 q quarter of the slice's tile)
    float* const mypart = fa.part + ((static_cast<long>(bh) * ns + j) * 4 + wave) * 1024;
    unsigned* const myflag = fa.flags + (static_cast<long>(bh) * ns + j) * 4 + wave;
    const bool handoff = !(fa.dbg & 1);
    if (kb > 0 && handoff) {                                      // block-uniform
      unsigned spins = 0;
      for (;;) {
        unsigned f = 0;
        if (lane == 0) f = __hip_atomic_load(myflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f = __builtin_amdgcn_readfirstlane(f);
        if (f == static_cast<unsigned>(kb)) break;
        if (++spins > kFbSpinLimit) {                             // never in a healthy launch: report and carry on instead of hanging the GPU
          if (lane == 0) atomicAdd(fa.flags + static_cast<long>(a.B) * a.H * ns * 4, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 pvv = *reinterpret_cast<const f32x4*>(mypart + (g * 64 + lane) * 4);
        dq[4 * g] += pvv[0]; dq[4 * g + 1] += pvv[1]; dq[4 * g + 2] += pvv[2]; dq[4 * g + 3] += pvv[3];
      }
    }
    if (kb + 1 < fa.nkb && handoff) {
      // write-through (sc1) 16-byte stores as compiler-visible buffer stores: an inline-asm store gets no hazard wait states before the
      // next instruction that overwrites its data registers
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(mypart, 0, 4096, 0x00020000);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x4 o = {__float_as_uint(dq[4 * g]), __float_as_uint(dq[4 * g + 1]), __float_as_uint(dq[4 * g + 2]), __float_as_uint(dq[4 * g + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (g * 64 + lane) * 16, 0, 16);      // aux 16 = sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(myflag, static_cast<unsigned>(kb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const int qi = j * kFbSlice + r;
      if (qi < a.Sq) {                                            // lanes q and q + 32 own the same row: they skip together
        unsigned short* p = a.dq + (static_cast<long>(b) * a.Sq + qi) * a.ldq + head * kDh + 32 * wave + 8 * h;
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          unsigned ax = pack2(dq[4 * g + 0] * a.scale, dq[4 * g + 1] * a.scale), ay = pack2(dq[4 * g + 2] * a.scale, dq[4 * g + 3] * a.scale);
          unsigned bx = pack2(dq[4 * g + 4] * a.scale, dq[4 * g + 5] * a.scale), by = pack2(dq[4 * g + 6] * a.scale, dq[4 * g + 7] * a.scale);
          const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
          const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
          *reinterpret_cast<uint4*>(p + 8 * g) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
      }
    }
  }
#undef ADT_TR2
#undef ADT_TRX

#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int ki = key0 + 64 * wave + 32 * blk + r;
    store_transposed(dk[blk], a.scale, a.dk + static_cast<long>(b) * a.Sk * a.ldk + head * kDh, a.ldk, ki, a.Sk, lane);
    store_transposed(dv[blk], 1.0f, a.dv + static_cast<long>(b) * a.Sk * a.ldv + head * kDh, a.ldv, ki, a.Sk, lane);
  }
}

static size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }
static int fused_ns(const adt_attn_desc* d) { return (d->q_len + kFbSlice - 1) / kFbSlice; }
static int fused_nkb(const adt_attn_desc* d) { return (d->k_len + kFbKeys - 1) / kFbKeys; }

size_t attn_bwd_fused_workspace_bytes(const adt_attn_desc* d) {
  const size_t bh = static_cast<size_t>(d->batch) * d->heads, ns = static_cast<size_t>(fused_ns(d));
  size_t bytes = align256(bh * ns * 4 * 4 + 16);                                  // flags (+ the give-up counter): zeroed every launch
  bytes += align256(bh * ns * kFbSlice * 2 * 4);                                  // statistics
  if (fused_nkb(d) > 1) bytes += align256(bh * ns * 4096 * 4);                    // running dQ sums
  return bytes;
}

int launch_attn_bwd_fused(const adt_attn_desc* d, const AttnArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < attn_bwd_fused_workspace_bytes(d) || !aligned16(ws)) return set_error(ADT_EINVAL, "adt_attn_bwd: workspace too small");
  static thread_local int lds_done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (lds_done_for != dev) {
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
    ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kFbLds));
    lds_done_for = dev;
  }
  FusedArgs fa{};
  fa.a = a;
  fa.ns = fused_ns(d);
  fa.nkb = fused_nkb(d);
  { const char* e = getenv("ADT_FB_DBG"); fa.dbg = e ? atoi(e) : 0; }
  const size_t bh = static_cast<size_t>(d->batch) * d->heads, ns = static_cast<size_t>(fa.ns);
  unsigned char* p = static_cast<unsigned char*>(ws);
  const size_t flag_bytes = align256(bh * ns * 4 * 4 + 16);
  fa.flags = reinterpret_cast<unsigned*>(p);
  p += flag_bytes;
  float* stats = reinterpret_cast<float*>(p);
  fa.stats = stats;
  p += align256(bh * ns * kFbSlice * 2 * 4);
  fa.part = reinterpret_cast<float*>(p);
  ADT_HIP_TRY(hipMemsetAsync(fa.flags, 0, flag_bytes, st));
  const long items = static_cast<long>(bh) * ns * kFbSlice;
  hipLaunchKernelGGL(attn_bwd_stats_kernel, dim3(static_cast<unsigned>((items + 15) / 16)), dim3(256), 0, st, a, stats, fa.ns);
  const long n_tiles = static_cast<long>(fa.nkb) * static_cast<long>(bh);
  for (int x = 0; x < 8; ++x) fa.sched_total[x] = static_cast<unsigned>(n_tiles / 8 + (x < n_tiles % 8 ? 1 : 0));
  if (int rc = sched_counters(st, &fa.sched)) return rc;
  if (a.drop.on()) hipLaunchKernelGGL(attn_bwd_fused_kernel<true>, dim3(static_cast<unsigned>(n_tiles)), dim3(kFbThreads), kFbLds, st, fa);
  else hipLaunchKernelGGL(attn_bwd_fused_kernel<false>, dim3(static_cast<unsigned>(n_tiles)), dim3(kFbThreads), kFbLds, st, fa);
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}

}  // namespace adt
