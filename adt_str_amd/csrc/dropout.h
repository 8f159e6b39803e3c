// dropout.h -- counter-based dropout masks shared by every kernel that applies or re-applies one.
//
// The reference uses torch's Philox stream inside nn.Dropout / SDPA (model.py:116,132,134,156,172 and the
// transformer layers); its draws cannot be reproduced, so dropout here only has to be a sound
// Bernoulli(1-p) mask that the backward pass can regenerate without storing it.
//
// Element index: idx = row * L2 + col for a tensor whose last dimension is L (L2 = L rounded up to even; rows = all
// leading dimensions flattened).  Two neighbouring elements share one 32-bit hash: pair = idx >> 1,
//   h = mix32(pair_lo ^ mix32(pair_hi ^ key)),  element idx is kept iff its 16-bit half (idx & 1 ? h >> 16 : h & 0xffff)
//   >= round(p * 65536).
// mix32 = the "lowbias32" integer hash (2 multiplies, 3 xor-shifts): the multiplies are quarter-rate instructions and
// the hash is the largest single piece of VALU work in the attention kernels, hence one hash for two decisions (p is
// resolved to 1.5e-5).  Kept elements are scaled by 1/(1-p).
#pragma once
#include <cstdint>

namespace adt {

#if defined(__HIPCC__)
#define ADT_DROP_HD __host__ __device__ __forceinline__
#else
#define ADT_DROP_HD inline
#endif

ADT_DROP_HD uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
struct Drop {
  uint32_t thr, key; float inv_keep;       // thr: 16-bit threshold; 0: disabled
  ADT_DROP_HD uint32_t pair_hash(uint64_t pair) const {
    return mix32(static_cast<uint32_t>(pair) ^ mix32(static_cast<uint32_t>(pair >> 32) ^ key));
  }
  // pairs below 2^32 (element indices below 2^33): the inner hash is the constant key2 = mix32(key)
  ADT_DROP_HD uint32_t pair_hash32(uint32_t pair, uint32_t key2) const { return mix32(pair ^ key2); }
  ADT_DROP_HD float pick(uint32_t h, uint32_t half) const { return ((half ? (h >> 16) : (h & 0xffffu)) >= thr) ? inv_keep : 0.0f; }
  ADT_DROP_HD float lo(uint32_t h) const { return (h & 0xffffu) >= thr ? inv_keep : 0.0f; }
  ADT_DROP_HD float hi(uint32_t h) const { return (h >> 16) >= thr ? inv_keep : 0.0f; }
  ADT_DROP_HD float scale(uint64_t idx) const { return pick(pair_hash(idx >> 1), static_cast<uint32_t>(idx) & 1u); }
  // four consecutive elements starting at a multiple of 4: two hashes
  ADT_DROP_HD void scale4(uint64_t idx0, float (&k)[4]) const {
    const uint32_t h0 = pair_hash(idx0 >> 1), h1 = pair_hash((idx0 >> 1) + 1);
    k[0] = lo(h0); k[1] = hi(h0); k[2] = lo(h1); k[3] = hi(h1);
  }
  ADT_DROP_HD bool on() const { return thr != 0; }
};
// last-dimension stride of the element index
ADT_DROP_HD uint64_t drop_ld(uint64_t L) { return L + (L & 1u); }
inline Drop make_drop(float p, uint32_t key) {
  Drop d{0u, key, 1.0f};
  if (p > 0.0f) {
    const double pp = p < 0.999999 ? p : 0.999999;
    long t = static_cast<long>(pp * 65536.0 + 0.5);
    t = t < 1 ? 1 : (t > 65535 ? 65535 : t);
    d.thr = static_cast<uint32_t>(t);
    d.inv_keep = static_cast<float>(1.0 / (1.0 - pp));
  }
  return d;
}

}  // namespace adt
