// attention.hip -- K4: flash-style multi-head attention forward / backward on MFMA (gfx950).
//
// Stands behind nn.MultiheadAttention inside the reference's nn.TransformerEncoderLayer /
// nn.TransformerDecoderLayer (model.py:118-127 encoder self-attention, no mask; :159-168 decoder
// self-attention with the additive causal + key-padding masks built at :173-181, and
// cross-attention onto the encoder memory, no mask).  head_dim = 128.
//
// Layout: Q, K, V, O, dO are bf16 matrices [B*S, ld] whose head h occupies columns
// [h*128, h*128+128) -- i.e. the packed in_proj output is consumed in place, no head permute.
//
// Every product uses v_mfma_f32_32x32x16_bf16 with the *softmax axis on the lane*:
//   forward / dQ kernel : lane <-> query.  S^T = K Q^T  (A = K rows from LDS, B = Q^T held in
//       registers), so one lane owns its query's scores, the row max/sum are in-register plus one
//       cross-half shuffle, and the accumulator (keys on rows) is directly the B operand of
//       O^T += V^T P^T  (A = V^T through ds_read_b64_tr_b16).  dQ^T += K^T dS^T the same way.
//   dK/dV kernels       : lane <-> key.  S = Q K^T, dP = dO V^T (A = Q / dO rows from LDS,
//       B = K^T / V^T in registers); P and dS (queries on rows) are the B operands of
//       dV^T += dO^T P and dK^T += Q^T dS (A through transposed LDS reads).
// K/V (or Q/dO) tiles of 64 rows x 256 B live in LDS under the XOR swizzle
//   off(row, chunk) = 256*row + 16*(chunk ^ (((row&3)<<2) | ((row>>2)&3)))
// which makes both the 16-byte row reads and the transposed 8-byte reads bank-conflict free.
// Tiles are double-buffered: the next tile's global loads are issued before the MFMAs of the
// current one and written to the other buffer afterwards (one barrier per tile).
// The tiled forward lives in attention_fwd.hip (software-pipelined, persistent); this file holds the one-query decode kernel, the
// two-kernel backward and the C entry points.
// The backward recomputes P from the saved log-sum-exp; dQ (which also produces delta = rowsum(O * dO) for its queries)
// and dK/dV are separate kernels so no float atomics are needed (bitwise reproducible).
#include "attn_common.h"

namespace adt {

// =============================================================================== forward: attention_fwd.hip (software-pipelined, persistent)

// =============================================================================== forward, ONE query per (batch, head): decode
// The KV-cached greedy decode (reference model.py:260-324 recomputes the whole prefix; network.py: greedy_decode_cached) asks for
// the attention of a single new position over the cache (self-attention) or over the encoder memory (cross-attention).  The tiled
// kernel above would run one 128-query tile with one live row and walk the keys serially in one workgroup per (batch, head):
// 27 us per launch at 986 keys, eight launches per decoded token.  Here the keys are spread over sixteen waves, a wave takes 64
// keys at a time, and there is no matrix unit:
//   loads    one wave-instruction reads FOUR key (or value) rows: the 16 lanes of a quarter-wave read one row's 256 bytes as
//            16-byte pieces (fully coalesced; a lane-per-key layout issues 64 separate 16-byte requests per instruction and
//            was bound by the address path at 17 us);
//   scores   a lane dots its 8 head dimensions with the matching piece of the query row (four v_dot2c_f32_bf16), the quarter-wave sums on the DPP
//            network (quad_perm x2, row_half_mirror, row_mirror): sixteen scores per lane, one per row group;
//   softmax  online across a wave's blocks; max / sum over the lane's sixteen values, then across the four quarter-waves;
//   P V      the same four-rows-per-instruction reads: a lane multiplies its row's probability -- already in its own register --
//            into its 8 dimensions; the four quarter-waves' partial outputs are added at the end;
//   combine  the sixteen (max, sum, partial output) triples through LDS.
// Keys behind the key-padding length are skipped when the additive mask is the reference's -1e4 (any mask <= -1000): their
// weight exp(score - 1e4 - max) is exactly 0 in fp32 next to one unmasked key; with a milder mask every key is visited and masked.
constexpr int kDecWaves = 8;
typedef __bf16 bf16x2d __attribute__((ext_vector_type(2)));
typedef float f32x2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64 * kDecWaves) void attn_decode_kernel(AttnArgs a) {
  __shared__ float red_m[kDecWaves], red_l[kDecWaves];
  __shared__ float red_o[kDecWaves][kDh];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, head = blockIdx.x % a.H;
  int klen = a.key_len ? a.key_len[b] : a.Sk;
  klen = klen < 0 ? 0 : (klen > a.Sk ? a.Sk : klen);
  const bool skip_masked = a.mask_value <= -1000.f && klen > 0;
  const int n = skip_masked ? klen : a.Sk;
  const float sl2 = a.scale * kLog2e, mask2 = a.mask_value * kLog2e;
  // this lane's 8 dimensions of the query row, as four bf16 pairs (v_dot2c_f32_bf16 multiplies pairs and adds in fp32)
  const uint4 qw = *reinterpret_cast<const uint4*>(a.q + static_cast<long>(b) * a.ldq + head * kDh + 8 * c);
  const bf16x2d qp[4] = {__builtin_bit_cast(bf16x2d, qw.x), __builtin_bit_cast(bf16x2d, qw.y), __builtin_bit_cast(bf16x2d, qw.z), __builtin_bit_cast(bf16x2d, qw.w)};
  const unsigned short* kb = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * kDh + 8 * c;
  const unsigned short* vb = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * kDh + 8 * c;
  float m = -INFINITY, l = 0.f;
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j0 = wave * 64; j0 < n; j0 += 64 * kDecWaves) {
    // rows j0 + 4 i + g, i = 0 .. 15 (clamped to the last live row: its probability is forced to 0 below)
    // addresses = a uniform base (scalar registers) + a 32-bit lane offset: per-row 64-bit pointers would not fit 128 registers
    const char* kbase = reinterpret_cast<const char*>(kb + static_cast<long>(j0) * a.ldk);
    const char* vbase = reinterpret_cast<const char*>(vb + static_cast<long>(j0) * a.ldv);
    const unsigned kstride = static_cast<unsigned>(a.ldk) * 2u, vstride = static_cast<unsigned>(a.ldv) * 2u;
    const int last = n - 1 - j0;                              // rows of the block past the live keys re-read the last live row
    uint4 kr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = 4 * i + g;
      kr[i] = *reinterpret_cast<const uint4*>(kbase + static_cast<unsigned>(r < last ? r : last) * kstride);
    }
    float s2[16], bm = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2d, kr[i].x), qp[0], 0.f, false);
      acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2d, kr[i].y), qp[1], acc, false);
      acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2d, kr[i].z), qp[2], acc, false);
      acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2d, kr[i].w), qp[3], acc, false);
      const int j = j0 + 4 * i + g;
      s2[i] = j < n ? fmaf(quarter_sum(acc), sl2, j >= klen ? mask2 : 0.f) : -INFINITY;
      bm = fmaxf(bm, s2[i]);
    }
    uint4 vr[8];                                              // V rows in two halves of eight: 1024 threads leave 128 registers per lane
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 4 * i + g;
      vr[i] = *reinterpret_cast<const uint4*>(vbase + static_cast<unsigned>(r < last ? r : last) * vstride);
    }
    bm = fmaxf(bm, __shfl_xor(bm, 16));                       // across the four quarter-waves (lanes of a quarter agree)
    bm = fmaxf(bm, __shfl_xor(bm, 32));
    const float m_new = fmaxf(m, bm);                         // finite: the block has a live row
    const float corr = exp2f(m - m_new);                      // first block: exp2(-inf) = 0
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= corr;
    // two rows at a time: (v_i[d], v_i+1[d]) . (p_i, p_i+1) per dimension d -- v_perm_b32 pairs the rows' halves, the
    // probabilities are rounded to bf16 like the tiled kernel rounds P for its P V product
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 vn[8];
      if (half == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = 4 * (i + 8) + g;
          vn[i] = *reinterpret_cast<const uint4*>(vbase + static_cast<unsigned>(r < last ? r : last) * vstride);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const float p0 = exp2f(s2[8 * half + i] - m_new), p1 = exp2f(s2[8 * half + i + 1] - m_new);      // rows past the block: exp2(-inf) = 0
        ps += p0 + p1;
        const bf16x2d pp = __builtin_convertvector(f32x2d{p0, p1}, bf16x2d);
        const unsigned w0[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w}, w1[4] = {vr[i + 1].x, vr[i + 1].y, vr[i + 1].z, vr[i + 1].w};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned lo = __builtin_amdgcn_perm(w1[w], w0[w], 0x05040100u);    // (w0 low half, w1 low half)
          const unsigned hi = __builtin_amdgcn_perm(w1[w], w0[w], 0x07060302u);    // (w0 high half, w1 high half)
          o[2 * w] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2d, lo), pp, o[2 * w], false);
          o[2 * w + 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2d, hi), pp, o[2 * w + 1], false);
        }
      }
      if (half == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) vr[i] = vn[i];
      }
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l = fmaf(l, corr, ps);
    m = m_new;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {                               // the four quarter-waves hold partial sums of the same 8 dimensions
    o[e] += __shfl_xor(o[e], 16);
    o[e] += __shfl_xor(o[e], 32);
  }
  if (lane == 0) { red_m[wave] = m; red_l[wave] = l; }
  if (g == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red_o[wave][8 * c + e] = o[e];
  }
  __syncthreads();
  if (tid < kDh) {
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < kDecWaves; ++w) M = fmaxf(M, red_m[w]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < kDecWaves; ++w) {
      const float cw = exp2f(red_m[w] - M);                   // a wave without keys: exp2(-inf) = 0
      L = fmaf(red_l[w], cw, L);
      acc = fmaf(red_o[w][tid], cw, acc);
    }
    a.out[static_cast<long>(b) * a.ldo + head * kDh + tid] = __builtin_bit_cast(unsigned short, static_cast<__bf16>(acc / L));
    if (tid == 0 && a.lse) a.lse[static_cast<long>(b) * a.H + head] = (M + log2f(L)) * kLn2;
  }
}

// =============================================================================== backward: dQ  (lane <-> query)
template <bool kDrop>
__global__ __launch_bounds__(kAttnThreads, 2) void attn_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TileXY tc = tile_coords((a.Sq + 127) / 128);
  const int b = tc.y / a.H, head = tc.y % a.H;
  const int qi = tc.x * 128 + wave * 32 + r;
  const unsigned short* qb = a.q + static_cast<long>(b) * a.Sq * a.ldq + head * kDh;
  const unsigned short* dob = a.dout + static_cast<long>(b) * a.Sq * a.ldo + head * kDh;
  const unsigned short* kb_ = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * kDh;
  const unsigned short* vb = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * kDh;
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  const float sl2 = a.scale * kLog2e;
  const long stat = (static_cast<long>(b) * a.H + head) * a.Sq + (qi < a.Sq ? qi : 0);
  const float lse2 = a.lse[stat] * kLog2e;
  // dropout index of (row, key) is row * Sk2 + key (Sk2 = Sk rounded up to even; dropout.h): pair = row * Sk2 / 2 + key / 2
  const unsigned pairbase = static_cast<unsigned>(((static_cast<uint64_t>(b) * a.H + head) * a.Sq + qi) * ((a.Sk + 1) >> 1));
  const unsigned key2 = mix32(a.drop.key);

  bf16x8 qf[8], dof[8];
  frags_from_global(qb, a.ldq, qi, a.Sq, lane, qf);
  frags_from_global(dob, a.ldo, qi, a.Sq, lane, dof);
  // delta = rowsum(O * dO) of this lane's query: the lane already holds half of its dO row (d = 16s + 8h .. +7), the partner
  // lane (lane ^ 32) the other half.  Written out for the dK/dV kernel, which runs after this one on the same stream.
  float dlt = 0.f;
  {
    bf16x8 of[8];
    frags_from_global(a.o + static_cast<long>(b) * a.Sq * a.ldo + head * kDh, a.ldo, qi, a.Sq, lane, of);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        dlt = fmaf(__uint_as_float(static_cast<unsigned>(static_cast<unsigned short>(of[s][e])) << 16),
                   __uint_as_float(static_cast<unsigned>(static_cast<unsigned short>(dof[s][e])) << 16), dlt);
    dlt += __shfl_xor(dlt, 32);
    if (h == 0 && qi < a.Sq) const_cast<float*>(a.delta)[stat] = dlt;
  }
  f32x16 dq[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;

  const int n_tiles = (a.Sk + kRowsPerTile - 1) / kRowsPerTile;
  const bool wave_has_rows = tc.x * 128 + wave * 32 < a.Sq;      // wave-uniform (986 queries: wave 3 of the eighth query block has none)
  unsigned koff[4], voff[4];
  tile_dma_offsets(a.ldk, wave, lane, koff);
  tile_dma_offsets(a.ldv, wave, lane, voff);
  tile_dma(kb_, a.ldk, 0, a.Sk, smem, wave, lane);
  tile_dma(vb, a.ldv, 0, a.Sk, smem + kAttnTileBytes, wave, lane);
  dma_wait_and_sync();
  unsigned troff[4][2];
  tr_offsets(lane, troff);
  for (int t = 0; t < n_tiles; ++t) {
    const unsigned char* tk = smem + (t & 1) * 2 * kAttnTileBytes;
    const unsigned char* tv = tk + kAttnTileBytes;
    unsigned tka[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db) { tka[db][0] = troff[db][0] + lds_off(tk); tka[db][1] = troff[db][1] + lds_off(tk); }
    if (t + 1 < n_tiles) {
      unsigned char* nk = smem + ((t + 1) & 1) * 2 * kAttnTileBytes;
      tile_dma_pre(kb_, a.ldk, (t + 1) * kRowsPerTile, a.Sk, nk, wave, lane, koff);
      tile_dma_pre(vb, a.ldv, (t + 1) * kRowsPerTile, a.Sk, nk + kAttnTileBytes, wave, lane, voff);
    }
    const int tile0 = t * kRowsPerTile;
    const bool need_mask = a.causal || tile0 + kRowsPerTile > klen || tile0 + kRowsPerTile > a.Sk;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb == 1 && tile0 + 32 >= a.Sk) break;          // the last tile's second block holds no key (986 keys: block 31 of 32): its dS is exactly 0
      if (!wave_has_rows) break;                         // a wave whose 32 rows all lie past Sq only stages tiles and meets the barriers
      f32x16 st, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tk, kb, s, lane), qf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tv, kb, s, lane), dof[s], dp, 0, 0, 0);
      }
      if (need_mask) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ki = tile0 + kb * 32 + acc_row(i, h);
          const float tt = fmaf(st[i], sl2, mask_add(a, qi, ki, klen) * kLog2e);
          st[i] = ki < a.Sk ? __builtin_amdgcn_exp2f(tt - lse2) : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = __builtin_amdgcn_exp2f(fmaf(st[i], sl2, -lse2));
      }
      float keep[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)         // the lane's keys come in runs of four: two hashes per run (dropout.h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned hh = kDrop ? a.drop.pair_hash32(pairbase + static_cast<unsigned>((tile0 + kb * 32 + 8 * g4 + 4 * h) >> 1) + u, key2) : 0u;
          keep[4 * g4 + 2 * u] = kDrop ? a.drop.lo(hh) : 1.0f;
          keep[4 * g4 + 2 * u + 1] = kDrop ? a.drop.hi(hh) : 1.0f;
        }
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] *= fmaf(dp[i], keep[i], -dlt);   // dS^T / scale (the softmax scale is applied once, when dQ is stored)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        TrFrag kt[4];
        if (kb == 0 && s2 == 0) tr4_issue_at<0>(tka, kt);
        else if (kb == 0) tr4_issue_at<16>(tka, kt);
        else if (s2 == 0) tr4_issue_at<32>(tka, kt);
        else tr4_issue_at<48>(tka, kt);
        const bf16x8 dsf = acc_to_b(st, s2);
        tr_wait();
#pragma unroll
        for (int db = 0; db < 4; ++db)
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(kt[db]), dsf, dq[db], 0, 0, 0);
      }
    }
    dma_wait_and_sync();
  }
  store_transposed(dq, a.scale, a.dq + static_cast<long>(b) * a.Sq * a.ldq + head * kDh, a.ldq, qi, a.Sq, lane);
  if (a.cs_dq) {                                            // block-uniform
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][128]; the tiles are dead after the loop's last barrier
    colsum_transposed(dq, a.scale, qi < a.Sq, lane, red + wave * kDh);
    __syncthreads();
    if (tid < kDh)
      a.cs_dq[(static_cast<long>(b) * ((a.Sq + 127) / 128) + tc.x) * (a.H * kDh) + head * kDh + tid] =
          (red[tid] + red[kDh + tid]) + (red[2 * kDh + tid] + red[3 * kDh + tid]);
  }
}

// =============================================================================== backward: dK, dV, eight symmetric waves
// (The single-wave kernel and the producer / consumer wave-pair kernel of rounds 1 and 2 lost every A/B for two rounds and are gone;
// a pair kernel is bound by its S-wave's serial chain: 16 MFMAs -> softmax / dS arithmetic -> hand-over, ~1900 cycles per 32 x 32
// block for 1024 cycles of MFMA on the SIMD.)  Here every wave does the whole block in its
// own registers -- S = Q K^T and dP = dO V^T (16 MFMAs), the arithmetic, dV^T += dO^T P and dK^T += Q^T dS (16 MFMAs) -- so nothing
// crosses LDS between roles.  What makes it fit 256 registers (two waves per SIMD) is that only the accumulators live in
// registers (dK^T, dV^T: 128; S, dP: 32): K and V of the workgroup's 128 keys are LDS images (32 KiB each) read as MFMA operands,
// the Q / dO / K / V fragments go through a two-deep ring of k-step units, the per-query statistics and the transposed Q / dO
// fragments arrive in halves.
//   * 8 waves = 4 key groups x 2: waves w and w + 4 own the SAME 32 keys and split the query blocks (the 64-query tile's first /
//     second 32 rows); their partial dK / dV are added through LDS once, after the loop (fixed order: bitwise reproducible).
//   * The two waves of a SIMD run the same program, so left alone they would want the matrix pipe together and the vector pipe
//     together.  Waves 4-7 therefore run ONE PHASE behind (MI355X_MICROARCH.md, "two waves that run the same program": stagger
//     by wave number >= 4): in iteration t they first finish block t - 1 (arithmetic, dV / dK MFMAs; S and dP were left in their
//     accumulators across the barrier) and then run the S / dP chain of block t, while waves 0-3 go chain -> arithmetic -> MFMAs
//     inside the iteration.  So the second half of a tile is read in two consecutive iterations: half-tiles are staged separately,
//     two slots for the first halves, three for the second (80 KiB instead of three whole tiles).
//   * Dropout: the two lanes of an even / odd key pair share every hash (dropout.h pairs keys), so each computes 8 of the block's
//     16, both halves are compared against the threshold, and the per-element keep masks are assembled from the compare results
//     with scalar mask arithmetic (shift by one lane) -- 8 hashes per lane and block instead of 16, no cross-lane data movement.
constexpr int kDkv3Threads = 512;
constexpr int kDkv3Img = 128 * 256;                           // K (and V) rows of the workgroup's 128 keys
constexpr int kDkv3Half = 2 * 32 * 256;                       // one half-tile slot: 32 Q rows | 32 dO rows
constexpr int kDkv3H0 = 2 * kDkv3Img, kDkv3H1 = kDkv3H0 + 2 * kDkv3Half, kDkv3Stats = kDkv3H1 + 3 * kDkv3Half;
constexpr int kDkv3Lds = kDkv3Stats + 3 * 512;                // 148,992 B: K | V | 2 first-half slots | 3 second-half slots | 3 x (-lse2[64], -delta[64])

template <bool kDrop, bool kStagger>
__global__ __launch_bounds__(kDkv3Threads) void attn_bwd_dkv3_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3, half = wave >> 2;
  const bool skew = kStagger && half == 1;                        // wave-uniform
  const TileXY tc = tile_coords((a.Sk + 127) / 128);
  const int b = tc.y / a.H, head = tc.y % a.H;
  const int key0 = tc.x * 128;
  const int ki = key0 + pair * 32 + r;
  const unsigned short* qb = a.q + static_cast<long>(b) * a.Sq * a.ldq + head * kDh;
  const unsigned short* dob = a.dout + static_cast<long>(b) * a.Sq * a.ldo + head * kDh;
  const unsigned short* kb_ = a.k + static_cast<long>(b) * a.Sk * a.ldk + head * kDh;
  const unsigned short* vb = a.v + static_cast<long>(b) * a.Sk * a.ldv + head * kDh;
  const float* lse_b = a.lse + (static_cast<long>(b) * a.H + head) * a.Sq;
  const float* dl_b = a.delta + (static_cast<long>(b) * a.H + head) * a.Sq;
  const int klen = a.key_len ? a.key_len[b] : a.Sk;
  const float sl2 = a.scale * kLog2e;
  const bool key_mask = a.causal || key0 + 128 > klen || key0 + 128 > a.Sk;
  const int n_tiles = (a.Sq + kRowsPerTile - 1) / kRowsPerTile;
  const unsigned smem_base = lds_off(smem);
  // slot of (tile t, half): byte offset from smem
  auto slot = [&](int t, int hf) { return hf ? kDkv3H1 + (t % 3) * kDkv3Half : kDkv3H0 + (t & 1) * kDkv3Half; };

  // ---- staging (source addresses as a wave-uniform base + a 32-bit per-lane element offset: the saddr form of the load).
  // K and V images once: 32 row groups of 4 rows each, 4 + 4 per wave; a Q / dO tile per iteration: 16 + 16 groups, 2 + 2 per wave.
  const int lrow = lane >> 4, lchunk = lane & 15;
  const int ldq_i = static_cast<int>(a.ldq), ldo_i = static_cast<int>(a.ldo), ldk_i = static_cast<int>(a.ldk), ldv_i = static_cast<int>(a.ldv);
  auto issue_tile = [&](int t) {
    const int row0 = t * kRowsPerTile;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = 2 * wave + i, row = 4 * g + lrow;              // g 0..7: first half, 8..15: second half (wave-uniform)
      const int chunk = lchunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
      int gr = row0 + row;
      gr = gr < a.Sq ? gr : a.Sq - 1;                              // rows past the end repeat the last valid row (their P is masked to 0)
      const unsigned oq = static_cast<unsigned>(gr * ldq_i + chunk * 8), od = static_cast<unsigned>(gr * ldo_i + chunk * 8);
      unsigned char* dst = smem + slot(t, g >> 3) + (g & 7) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qb + oq), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dob + od),
                                       (__attribute__((address_space(3))) void*)(dst + 32 * 256), 16, 0, 0);
    }
  };
  float rs = 0.f;
  auto load_stats = [&](int t) {                                   // wave 0: lse, wave 1: delta of tile t's 64 queries (raw)
    const int qq = t * kRowsPerTile + lane;
    rs = 0.f;
    if (wave == 0 && qq < a.Sq) rs = lse_b[qq];
    if (wave == 1 && qq < a.Sq) rs = dl_b[qq];
  };
  auto store_stats = [&](int t) {                                  // both enter the arithmetic negated (fma-ready)
    if (wave < 2) reinterpret_cast<float*>(smem + kDkv3Stats + (t % 3) * 512)[wave * 64 + lane] = wave == 0 ? -rs * kLog2e : -rs;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = 4 * wave + i, row = 4 * g + lrow;
    const int chunk = lchunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
    int gr = key0 + row;
    gr = gr < a.Sk ? gr : a.Sk - 1;                                // keys past the end: masked in the arithmetic, their rows never stored
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb_ + static_cast<unsigned>(gr * ldk_i + chunk * 8)),
                                     (__attribute__((address_space(3))) void*)(smem + g * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + static_cast<unsigned>(gr * ldv_i + chunk * 8)),
                                     (__attribute__((address_space(3))) void*)(smem + kDkv3Img + g * 1024), 16, 0, 0);
  }
  load_stats(0);
  issue_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_stats(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");

  f32x16 dk[4], dv[4], st, dp;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }
#pragma unroll
  for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }

  // All LDS reads of the loop are inline asm with hand-placed lgkmcnt waits: a compiler-generated LDS access would be ordered behind
  // the tile DMA in flight with a vmcnt(0).  Addresses are re-derived from two lane constants by XOR (the swizzle is an XOR of
  // address bits 4..7, so the k-step / d-block enters as `^ 32 s` / `^ 64 db`): nothing per-lane but those two stays in registers.
  // row-fragment address: chunk (2s + h) of row r = (256 r + 16 (h ^ x)) ^ 32 s,  x = swizzle bits of r
  const unsigned rowbase = smem_base + static_cast<unsigned>(256 * r + 16 * (h ^ (((r & 3) << 2) | ((r >> 2) & 3))));
  // transposed-read address of d-block db, rows (4h + i/4) and + 8 of a 16-row group: (trbase ^ 64 db) and ((trbase ^ 64 db) ^ 32) + 2048
  unsigned trbase;
  {
    const int i = lane & 15, g4 = (lane >> 4) & 1, row = 4 * h + (i >> 2);
    trbase = smem_base + static_cast<unsigned>(8 * (i & 1) + swz(row, 2 * g4 + ((i & 3) >> 1)));
  }
  const unsigned sk_pairs = static_cast<unsigned>((a.Sk + 1) >> 1);
  const unsigned headpair = static_cast<unsigned>((static_cast<uint64_t>(b) * a.H + head) * a.Sq * sk_pairs) + static_cast<unsigned>(ki >> 1);
  const unsigned key2 = mix32(a.drop.key);
  const unsigned par = static_cast<unsigned>(lane) & 1u;
  const unsigned thr16 = a.drop.thr << 16;

  // ---- phase C: S = Q K^T, dP = dO V^T of block (tile t, this wave's 32 rows); operands through a ring of two k-step units
  auto chain = [&](int t) {
    const unsigned tq_a = rowbase + static_cast<unsigned>(slot(t, half));
    const unsigned krow = rowbase + static_cast<unsigned>(pair * 32 * 256);
    bf16x8 fq[2], fd[2], fk[2], fv[2];
#define ADT_UNIT(U, S)                                                                                                              \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %5 offset:32768"  \
                 : "=&v"(fq[U]), "=&v"(fk[U]), "=&v"(fd[U]), "=&v"(fv[U])                                                           \
                 : "v"(tq_a ^ static_cast<unsigned>(32 * (S))), "v"(krow ^ static_cast<unsigned>(32 * (S))) : "memory")
    ADT_UNIT(0, 0); ADT_UNIT(1, 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < 7) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[s & 1], fk[s & 1], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[s & 1], fv[s & 1], dp, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < 8) {
        if (s & 1) ADT_UNIT(1, s + 2);
        else ADT_UNIT(0, s + 2);
      }
    }
#undef ADT_UNIT
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- phases E + D of block (tile t): softmax / dropout / dS arithmetic on st, dp (in halves of 8 elements = one k-step of the
  // second products each, so that at most 16 statistics are live), then dV^T += dO^T P, dK^T += Q^T dS
  auto finish = [&](int t) {
    const unsigned st_a = smem_base + static_cast<unsigned>(kDkv3Stats + (t % 3) * 512 + (half * 32 + 4 * h) * 4);
    const unsigned trb = trbase + static_cast<unsigned>(slot(t, half));
    f32x4 la[2], da[2], lb[2], db_[2];
    asm volatile("ds_read_b128 %0, %1" : "=v"(la[0]) : "v"(st_a) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(da[0]) : "v"(st_a) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(la[1]) : "v"(st_a) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:288" : "=v"(da[1]) : "v"(st_a) : "memory");
    // keep masks (one 64-bit lane mask per accumulator register) while the statistics are in flight
    uint64_t km[16];
    if (kDrop) {
      const uint64_t even = 0x5555555555555555ull, odd = 0xaaaaaaaaaaaaaaaaull;
      const unsigned vbq = headpair + static_cast<unsigned>(t * kRowsPerTile + half * 32 + 16 * static_cast<int>(par) + 4 * h) * sk_pairs;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned hh = mix32((vbq + static_cast<unsigned>((j & 3) + 8 * (j >> 2)) * sk_pairs) ^ key2);
        const uint64_t c_lo = __builtin_amdgcn_ballot_w64((hh << 16) >= thr16);       // decision of the pair's even key
        const uint64_t c_hi = __builtin_amdgcn_ballot_w64(hh >= thr16);               // ... of its odd key
        // even lanes hashed query rows 0..15 of the block, odd lanes rows 16..31: element j comes from the even lanes, j + 8 from the odd
        km[j] = (c_lo & even) | ((c_hi & even) << 1);
        km[j + 8] = (c_hi & odd) | ((c_lo & odd) >> 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool need_mask = key_mask || (t + 1) * kRowsPerTile > a.Sq;     // block-uniform
    unsigned hp[8], hs[8];
    auto arith = [&](int m, const f32x2 nl, const f32x2 nd) {
      f32x2 pv;
      if (need_mask) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int i = 2 * m + e;
          const int qi = t * kRowsPerTile + half * 32 + acc_row(i, h);
          const float tt = fmaf(st[i], sl2, mask_add(a, qi, ki, klen) * kLog2e) + nl[e];
          pv[e] = __builtin_amdgcn_exp2f((qi < a.Sq && ki < a.Sk) ? tt : kNegBig);
        }
      } else {
        const f32x2 sv = {st[2 * m], st[2 * m + 1]};
        const f32x2 arg = sv * sl2 + nl;
        pv = f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
      }
      const f32x2 dpv = {dp[2 * m], dp[2 * m + 1]};
      f32x2 pd = pv, ds;
      if (kDrop) {
        f32x2 ks;
        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(ks[0]) : "v"(a.drop.inv_keep), "s"(km[2 * m]));
        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(ks[1]) : "v"(a.drop.inv_keep), "s"(km[2 * m + 1]));
        pd = pv * ks;
        ds = pv * (dpv * ks + nd);
      } else {
        ds = pv * (dpv + nd);
      }
      hp[m] = pack2(pd[0], pd[1]);
      hs[m] = pack2(ds[0], ds[1]);
    };
#define ADT_TR2(F, DB, IMM)                                                                                    \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%5"                  \
                 : "=&v"((F).lo), "=&v"((F).hi) : "v"(trb ^ static_cast<unsigned>(64 * (DB))), "v"((trb ^ static_cast<unsigned>(64 * (DB))) ^ 32u), "i"(IMM), "i"((IMM) + 2048) : "memory")
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(lb[0]) : "v"(st_a) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:320" : "=v"(db_[0]) : "v"(st_a) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(lb[1]) : "v"(st_a) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:352" : "=v"(db_[1]) : "v"(st_a) : "memory");
#pragma unroll
    for (int m = 0; m < 4; ++m)
      arith(m, f32x2{la[m >> 1][2 * (m & 1)], la[m >> 1][2 * (m & 1) + 1]}, f32x2{da[m >> 1][2 * (m & 1)], da[m >> 1][2 * (m & 1) + 1]});
    __builtin_amdgcn_sched_barrier(0);
    TrFrag dot[4], qt[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      ADT_TR2(dot[db], db, 32 * 256);
      ADT_TR2(qt[db], db, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");           // 4 + 16 reads queued, in order: the four statistics reads are back
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 4; m < 8; ++m)
      arith(m, f32x2{lb[(m - 4) >> 1][2 * (m & 1)], lb[(m - 4) >> 1][2 * (m & 1) + 1]}, f32x2{db_[(m - 4) >> 1][2 * (m & 1)], db_[(m - 4) >> 1][2 * (m & 1) + 1]});
    union { unsigned u[4]; bf16x8 v; } pf0, pf1, dsf0, dsf1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { pf0.u[e] = hp[e]; pf1.u[e] = hp[4 + e]; dsf0.u[e] = hs[e]; dsf1.u[e] = hs[4 + e]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < 4; ++db) {                    // the registers of k-step 0's fragments take k-step 1's as soon as their MFMA has issued
      dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(dot[db]), pf0.v, dv[db], 0, 0, 0);
      dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(qt[db]), dsf0.v, dk[db], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ADT_TR2(dot[db], db, 32 * 256 + 16 * 256);
      ADT_TR2(qt[db], db, 16 * 256);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(dot[db]), pf1.v, dv[db], 0, 0, 0);
      dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_get(qt[db]), dsf1.v, dk[db], 0, 0, 0);
    }
#undef ADT_TR2
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- main loop: n_tiles + 1 iterations with one barrier each.  Tile t + 1 is requested at the start of iteration t (its first-half
  // slot held tile t - 1's, last read in iteration t - 1; its second-half slot tile t - 2's, last read by waves 4-7 in iteration
  // t - 1 as well) and waited for at its end.  One loop per role so that each has a single path.
  auto end_iter = [&](int it, bool more) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (more) store_stats(it + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
  };
  // A wave whose 32 query rows of a tile all lie past Sq (986 queries: the second half of tile 15) has nothing to add -- its P is exactly 0 --
  // and sits the block out (wave-uniform; it still issues its share of the next tile and meets the barrier): its SIMD's other wave runs alone.
  auto has_rows = [&](int t) { return t * kRowsPerTile + half * 32 < a.Sq; };
  if (!skew) {
    for (int it = 0; it <= n_tiles; ++it) {
      const bool more = it + 1 < n_tiles;
      if (more) { load_stats(it + 1); issue_tile(it + 1); }
      if (it < n_tiles && has_rows(it)) { chain(it); finish(it); }
      end_iter(it, more);
    }
  } else {
    for (int it = 0; it <= n_tiles; ++it) {
      const bool more = it + 1 < n_tiles;
      if (more) { load_stats(it + 1); issue_tile(it + 1); }
      if (it >= 1 && has_rows(it - 1)) finish(it - 1);
      if (it < n_tiles && has_rows(it)) chain(it);
      end_iter(it, more);
    }
  }

  // ---- the two waves of a key group add their partial sums (waves 4-7 hand theirs over through LDS, dK then dV), waves 0-3 store
  float* xch = reinterpret_cast<float*>(smem) + pair * (4 * 16 * 64);         // 16 KiB per key group; every image and tile is dead by now
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    f32x16 (&acc)[4] = which == 0 ? dk : dv;
    if (half == 1) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(xch + ((db * 4 + g) * 64 + lane) * 4) = f32x4{acc[db][4 * g], acc[db][4 * g + 1], acc[db][4 * g + 2], acc[db][4 * g + 3]};
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(xch + ((db * 4 + g) * 64 + lane) * 4);
          acc[db][4 * g] += o[0]; acc[db][4 * g + 1] += o[1]; acc[db][4 * g + 2] += o[2]; acc[db][4 * g + 3] += o[3];
        }
    }
    __syncthreads();
  }
  if (half == 0) {
    store_transposed(dk, a.scale, a.dk + static_cast<long>(b) * a.Sk * a.ldk + head * kDh, a.ldk, ki, a.Sk, lane);
    store_transposed(dv, 1.0f, a.dv + static_cast<long>(b) * a.Sk * a.ldv + head * kDh, a.ldv, ki, a.Sk, lane);
  }
}

static int check_desc(const adt_attn_desc* d, const char* who) {
  if (!d) return set_error(ADT_EINVAL, "attention: null descriptor");
  if (d->head_dim != kDh) return set_error(ADT_ESHAPE, "attention: head_dim must be 128");
  if (d->batch < 0 || d->heads <= 0 || d->q_len < 0 || d->k_len < 0) return set_error(ADT_EINVAL, "attention: bad sizes");
  const int64_t need = static_cast<int64_t>(d->heads) * kDh;
  if (d->ldq < need || d->ldk < need || d->ldv < need || d->ldo < need || (d->ldq & 7) || (d->ldk & 7) || (d->ldv & 7) || (d->ldo & 7))
    return set_error(ADT_ESHAPE, "attention: row strides must cover heads*128 columns and be multiples of 8");
  if (static_cast<int64_t>(d->batch) * d->heads > 65535) return set_error(ADT_ESHAPE, "attention: batch*heads must be <= 65535");
  if (d->drop.p > 0.f && static_cast<double>(d->batch) * d->heads * d->q_len * (d->k_len + 1) >= 4294967296.0)
    return set_error(ADT_ESHAPE, "attention: dropout needs batch*heads*q_len*(k_len+1) < 2^32");
  (void)who;
  return ADT_OK;
}
static AttnArgs make_args(const adt_attn_desc* d) {
  AttnArgs a{};
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
  a.B = d->batch; a.H = d->heads; a.Sq = d->q_len; a.Sk = d->k_len;
  a.scale = d->scale; a.mask_value = d->mask_value; a.causal = d->causal; a.key_len = d->key_len;
  a.drop = make_drop(d->drop.p, d->drop.key);
  a.cs_dq = nullptr;
  a.keep_bits = a.drop.on() ? static_cast<unsigned*>(d->keep_bits) : nullptr;
  a.bits_nq = keep_bits_nq(d->q_len); a.bits_nk = keep_bits_nk(d->k_len);
  return a;
}
extern "C" size_t adt_attn_keep_bits_bytes(const adt_attn_desc* d) {
  if (!d || d->batch <= 0 || d->heads <= 0 || d->q_len <= 0 || d->k_len <= 0) return 16;
  return static_cast<size_t>(d->batch) * d->heads * keep_bits_nq(d->q_len) * keep_bits_nk(d->k_len) * 128;
}
static int set_lds_once() {      // raise the dynamic-LDS limit of the backward kernels once per device and thread
  static thread_local int done_for = -1;
  int dev = 0;
  ADT_HIP_TRY(hipGetDevice(&dev));
  if (done_for == dev) return ADT_OK;
  const int l4 = 4 * kAttnTileBytes;
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, l4));
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, l4));
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv3_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkv3Lds));
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv3_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkv3Lds));
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv3_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkv3Lds));
  ADT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv3_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kDkv3Lds));
  done_for = dev;
  return ADT_OK;
}

}  // namespace adt

using namespace adt;

extern "C" int adt_attn_fwd(const adt_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse, void* stream) {
  if (int rc = check_desc(d, "adt_attn_fwd")) return rc;
  if (!q || !k || !v || !o) return set_error(ADT_EINVAL, "adt_attn_fwd: null pointer");
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o)) return set_error(ADT_EINVAL, "adt_attn_fwd: tensors must be 16-byte aligned");
  if (d->batch == 0 || d->q_len == 0) return ADT_OK;
  if (d->k_len == 0) return set_error(ADT_ESHAPE, "adt_attn_fwd: k_len must be > 0");
  AttnArgs a = make_args(d);
  a.q = static_cast<const unsigned short*>(q); a.k = static_cast<const unsigned short*>(k); a.v = static_cast<const unsigned short*>(v);
  a.out = static_cast<unsigned short*>(o); a.lse = lse;
  // one query per (batch, head) without dropout or a causal mask: the decode kernel (ADT_ATTN_NO_DECODE=1: the tiled kernel, for A/B)
  if (d->q_len == 1 && !a.drop.on() && !a.causal && (d->ldq % 8) == 0 && (d->ldk % 8) == 0 && (d->ldv % 8) == 0 && getenv("ADT_ATTN_NO_DECODE") == nullptr) {
    hipLaunchKernelGGL(attn_decode_kernel, dim3(static_cast<unsigned>(d->batch) * d->heads), dim3(64 * kDecWaves), 0, static_cast<hipStream_t>(stream), a);
    ADT_HIP_TRY(hipGetLastError());
    return ADT_OK;
  }
  return launch_attn_fwd2(a, static_cast<hipStream_t>(stream));
}

static size_t delta_floats(const adt_attn_desc* d) { return (static_cast<size_t>(d->batch) * d->heads * d->q_len + 4 + 3) & ~static_cast<size_t>(3); }
// bytes of the two-kernel path's workspace (delta, column-sum partials and scratch); the fused path's region follows it, 256-aligned
static size_t split_workspace_bytes(const adt_attn_desc* d) {
  size_t bytes = delta_floats(d) * 4;
  if (d->dq_colsum || d->dk_colsum || d->dv_colsum)        // dQ partial sums per (batch, 128-query block) + the dV column-sum scratch
    bytes += static_cast<size_t>(d->batch) * d->heads * kDh * ((d->q_len + 127) / 128) * 4 +
             adt_colsum_workspace_bytes(static_cast<int64_t>(d->batch) * (d->k_len > d->q_len ? d->k_len : d->q_len), static_cast<int64_t>(d->heads) * kDh);
  return (bytes + 255) & ~static_cast<size_t>(255);
}
// Which backward runs.  The one-kernel path (attention_bwd_fused.hip / attention_bwd_fused8.hip -- the 8-wave form wherever it applies: no
// dropout or keep bits; same bits, 0.77 vs 0.86 ms -- : the five algorithmic products, dQ summed over the key-block workgroups
// by a scheduled fan-in) is the default without dropout (MI355X, encoder shape 0.83 vs 0.87 ms) and with dropout WHEN THE FORWARD LEFT ITS
// KEEP BITS (adt_attn_desc.keep_bits: no mask is hashed again; profiles/r05/attn_bwd_paths.txt).  With dropout and no bits the two-kernel path
// (dQ kernel + dK/dV kernel, 7 products, every mask hashed twice) is still the faster one (0.95 vs 0.98 ms): at one wave per SIMD the mask
// generation sits on the fused kernel's vector pipe with nothing to hide it.  ADT_ATTN_BWD=fused / split forces a path (read on every call:
// the tests and tools/exp_attn_bwd.py switch it) -- except where the fused path cannot be trusted to make progress: its fan-in has key block
// jr % nkb wait for the tiles of ALL the head's other key blocks, which draw consecutive tickets of one XCD group's counter, so every one of
// the head's nkb workgroups must be resident on that XCD group at once (one workgroup per CU, n_cu / 8 CUs per group).  Half of that is
// the bound here (CUs held by another stream's kernel -- an RCCL collective -- are not ours): 16 key blocks = 4096 keys on 256 CUs;
// beyond it, and under stream capture (the ticket counters are refused there, as for the persistent GEMMs), the two-kernel path runs.
static bool fused_can_run(const adt_attn_desc* d, hipStream_t st, bool* ok) {
  int n_cu = 0;
  if (device_cu_count(&n_cu)) return false;
  const int nkb = (d->k_len + 255) / 256;
  *ok = 2 * nkb <= n_cu / 8 || nkb == 1;
  if (*ok && st != nullptr) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) *ok = false;
  }
  return true;
}
static bool wants_fused(const adt_attn_desc* d) {
  const char* bwd_env = getenv("ADT_ATTN_BWD");
  if (bwd_env) return bwd_env[0] == 'f';
  return !(d->drop.p > 0.0f) || d->keep_bits != nullptr;
}
extern "C" int32_t adt_attn_bwd_giveups(int32_t clear) {
  unsigned* hw = nullptr;
  if (adt::attn_bwd_giveup_word(&hw, nullptr) != ADT_OK) return -1;
  const unsigned n = clear ? __atomic_exchange_n(hw, 0u, __ATOMIC_RELAXED) : __atomic_load_n(hw, __ATOMIC_RELAXED);
  return n > 0x7fffffffu ? 0x7fffffff : static_cast<int32_t>(n);
}

extern "C" size_t adt_attn_bwd_workspace_bytes(const adt_attn_desc* d) {
  if (!d || d->batch <= 0 || d->heads <= 0 || d->q_len <= 0 || d->k_len <= 0) return 16;
  // sized by the path that will run (the fused path's region is nkb fp32 copies of dQ: 780 MB at the encoder shape); the capture state of
  // the stream is not known here, so a shape the fused path may take is sized for it
  bool ok = false;
  if (!fused_can_run(d, nullptr, &ok)) ok = true;
  return split_workspace_bytes(d) + (wants_fused(d) && ok ? attn_bwd_fused_workspace_bytes(d) : 0);
}

extern "C" int adt_attn_bwd(const adt_attn_desc* d, const void* q, const void* k, const void* v, const void* o, const void* dout,
                            const float* lse, void* dq, void* dk, void* dv, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_desc(d, "adt_attn_bwd")) return rc;
  if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv) return set_error(ADT_EINVAL, "adt_attn_bwd: null pointer");
  if (!ws || ws_bytes < adt_attn_bwd_workspace_bytes(d)) return set_error(ADT_EINVAL, "adt_attn_bwd: workspace too small");
  if (d->batch == 0 || d->q_len == 0 || d->k_len == 0) return ADT_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  AttnArgs a = make_args(d);
  a.q = static_cast<const unsigned short*>(q); a.k = static_cast<const unsigned short*>(k); a.v = static_cast<const unsigned short*>(v);
  a.o = static_cast<const unsigned short*>(o); a.dout = static_cast<const unsigned short*>(dout);
  a.lse = const_cast<float*>(lse); a.delta = static_cast<const float*>(ws);
  a.dq = static_cast<unsigned short*>(dq); a.dk = static_cast<unsigned short*>(dk); a.dv = static_cast<unsigned short*>(dv);
  const bool want_cs = d->dq_colsum || d->dk_colsum || d->dv_colsum;
  if (want_cs && !(d->dq_colsum && d->dk_colsum && d->dv_colsum)) return set_error(ADT_EINVAL, "adt_attn_bwd: give all three column-sum outputs or none");
  const int nqb = (d->q_len + 127) / 128, hd = d->heads * kDh;
  bool fused_ok = false;
  if (!fused_can_run(d, st, &fused_ok)) return set_error(ADT_EHIP, "adt_attn_bwd: device query failed");
  const bool use_fused = wants_fused(d) && fused_ok;          // (see wants_fused / fused_can_run above)
  if (use_fused) {
    const size_t off = split_workspace_bytes(d);
    // the 8-wave form (attention_bwd_fused8.hip: two waves per SIMD, one 32-key block per wave) takes keep bits or no dropout;
    // ADT_ATTN_BWD_WAVES=4 / 8 forces a form (A/B runs, tests)
    const char* waves_env = getenv("ADT_ATTN_BWD_WAVES");
    const bool can8 = !(d->drop.p > 0.0f) || d->keep_bits != nullptr;
    const bool use8 = can8 && (waves_env ? waves_env[0] == '8' : true);
    if (int rc = use8 ? launch_attn_bwd_fused8(d, a, static_cast<unsigned char*>(ws) + off, ws_bytes - off, st)
                      : launch_attn_bwd_fused(d, a, static_cast<unsigned char*>(ws) + off, ws_bytes - off, st)) return rc;
    if (want_cs) {
      // bias gradient of the in-projection: column sums of the stored dQ and dV (one pass each); dK's vanish identically (the rows of dS
      // sum to zero), so exact zeros are written instead of rounding noise
      float* cws = static_cast<float*>(ws) + delta_floats(d) + static_cast<size_t>(d->batch) * nqb * hd;
      const int64_t rq = static_cast<int64_t>(d->batch) * d->q_len, rk = static_cast<int64_t>(d->batch) * d->k_len;
      if (int rc = adt_colsum_bf16(dq, d->ldq, rq, hd, d->dq_colsum, cws, adt_colsum_workspace_bytes(rq, hd), stream)) return rc;
      if (reduce_queue_open(st)) {
        if (int rc = reduce_queue_push(nullptr, 0, hd, d->dk_colsum, nullptr, nullptr, hd)) return rc;
      } else {
        ADT_HIP_TRY(hipMemsetAsync(d->dk_colsum, 0, static_cast<size_t>(hd) * 4, st));
      }
      if (int rc = adt_colsum_bf16(dv, d->ldv, rk, hd, d->dv_colsum, cws, adt_colsum_workspace_bytes(rk, hd), stream)) return rc;
    }
    ADT_HIP_TRY(hipGetLastError());
    return ADT_OK;
  }
  float* cs_slice = nullptr;
  if (want_cs) {
    cs_slice = reduce_queue_slice(static_cast<size_t>(d->batch) * nqb * hd * 4, st);     // open reduction queue: the second stage is deferred
    a.cs_dq = cs_slice ? cs_slice : static_cast<float*>(ws) + delta_floats(d);
  }
  const int lds_dq = 4 * kAttnTileBytes;
  if (int rc = set_lds_once()) return rc;
  const dim3 gq(static_cast<unsigned>((d->q_len + 127) / 128) * d->batch * d->heads), gk(static_cast<unsigned>((d->k_len + 127) / 128) * d->batch * d->heads);
  // dK / dV kernel: eight symmetric waves; ADT_ATTN_DKV=3 runs it with waves 4-7 staggered by one phase (A/B arm: it pays without
  // dropout only -- with it the staggered build spills seven registers inside the loop).  Read on every call: the tests switch it.
  const char* dkv_env = getenv("ADT_ATTN_DKV");
  const bool stagger = dkv_env && atoi(dkv_env) == 3;
  if (a.drop.on()) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, gq, dim3(kAttnThreads), lds_dq, st, a);
    if (stagger) hipLaunchKernelGGL((attn_bwd_dkv3_kernel<true, true>), gk, dim3(kDkv3Threads), kDkv3Lds, st, a);
    else hipLaunchKernelGGL((attn_bwd_dkv3_kernel<true, false>), gk, dim3(kDkv3Threads), kDkv3Lds, st, a);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, gq, dim3(kAttnThreads), lds_dq, st, a);
    if (stagger) hipLaunchKernelGGL((attn_bwd_dkv3_kernel<false, true>), gk, dim3(kDkv3Threads), kDkv3Lds, st, a);
    else hipLaunchKernelGGL((attn_bwd_dkv3_kernel<false, false>), gk, dim3(kDkv3Threads), kDkv3Lds, st, a);
  }
  if (want_cs) {
    // dQ: partial sums out of the dQ kernel's epilogue.  dK: every row of dS sums to zero (softmax), so the column sums of
    // dK = dS^T Q vanish identically -- the key bias does not change the attention output; what a sum over the stored dK would
    // return is rounding noise.  dV: a pass over the dV columns (taking it in the dK/dV kernel's epilogue costs more than that
    // pass: its one workgroup per CU has nothing to hide the cross-lane sums under).
    if (cs_slice) {
      if (int rc = reduce_queue_push(cs_slice, d->batch * nqb, hd, d->dq_colsum, nullptr, nullptr, hd)) return rc;
      if (int rc = reduce_queue_push(nullptr, 0, hd, d->dk_colsum, nullptr, nullptr, hd)) return rc;
    } else {
      launch_reduce_partials(a.cs_dq, d->batch * nqb, hd, d->dq_colsum, st);
      ADT_HIP_TRY(hipMemsetAsync(d->dk_colsum, 0, static_cast<size_t>(hd) * 4, st));
    }
    float* cws = static_cast<float*>(ws) + delta_floats(d) + static_cast<size_t>(d->batch) * nqb * hd;
    const int64_t rows = static_cast<int64_t>(d->batch) * d->k_len;
    if (int rc = adt_colsum_bf16(dv, d->ldv, rows, hd, d->dv_colsum, cws, adt_colsum_workspace_bytes(rows, hd), stream)) return rc;
  }
  ADT_HIP_TRY(hipGetLastError());
  return ADT_OK;
}
