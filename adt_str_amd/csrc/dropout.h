// dropout.h -- counter-based dropout masks shared by every kernel that applies or re-applies one.
//
// The reference uses torch's Philox stream inside nn.Dropout / SDPA (model.py:116,132,134,156,172 and the
// transformer layers); its draws cannot be reproduced, so dropout here only has to be a sound
// Bernoulli(1-p) mask that the backward pass can regenerate without storing it: element `idx` of dropout
// site `key` is kept iff  mix32(idx_lo ^ mix32(idx_hi ^ key)) >= p * 2^32.   mix32 = the "lowbias32" integer
// hash (2 multiplies, 3 xor-shifts).  Kept elements are scaled by 1/(1-p).
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
  uint32_t thr, key; float inv_keep;       // thr == 0: disabled
  ADT_DROP_HD float scale(uint64_t idx) const {
    const uint32_t h = mix32(static_cast<uint32_t>(idx) ^ mix32(static_cast<uint32_t>(idx >> 32) ^ key));
    return h >= thr ? inv_keep : 0.0f;
  }
  // same function for indices below 2^32 (idx_hi == 0): the inner hash is the constant key2 = mix32(key)
  ADT_DROP_HD float scale32(uint32_t idx, uint32_t key2) const { return mix32(idx ^ key2) >= thr ? inv_keep : 0.0f; }
  ADT_DROP_HD bool on() const { return thr != 0; }
};
inline Drop make_drop(float p, uint32_t key) {
  Drop d{0u, key, 1.0f};
  if (p > 0.0f) {
    const double pp = p < 0.999999 ? p : 0.999999;
    d.thr = static_cast<uint32_t>(pp * 4294967296.0);
    d.inv_keep = static_cast<float>(1.0 / (1.0 - pp));
  }
  return d;
}

}  // namespace adt
