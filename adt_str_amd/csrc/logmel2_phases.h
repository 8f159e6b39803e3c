// logmel2_phases.h -- per-lane phase bodies of the fused log-mel kernel K1 (second generation).
//
// One wave (64 lanes) turns ONE real 2048-sample frame into one row of normalised log-mel:
//   z[m] = w[2m] x[2m] + i w[2m+1] x[2m+1], m < 1024      (even / odd samples packed into one complex sequence)
//   Z    = FFT_1024(z)        1024 = 16 x 8 x 8, three in-register radix passes, two exchanges through a wave-private LDS buffer
//   E[k] = (Z[k] + conj Z[1024-k]) / 2,  O[k] = (Z[k] - conj Z[1024-k]) / (2i),  T = W_2048^k O[k]
//   |X[k]|^2 = |E + T|^2,  |X[1024-k]|^2 = |E - T|^2,  k = 0..512                (the 1025 bins of the real 2048-point FFT)
// then the banded mel reduction and the reference's post-processing (model.py:91-93).
//
// Why one frame per wave (the first generation packed two frames into a 2048-point complex FFT): the wave's data is 16
// complex values per lane instead of 32 and its LDS buffer 8 KB instead of 16.6 KB, so 16 waves fit a CU (four per SIMD,
// <= 128 VGPRs) instead of 8 -- the kernel is bound by VALU issue and LDS latency, and two waves per SIMD cannot cover either.
//
// LDS layouts (cf = 8-byte units inside the wave's 1024-cf buffer), each checked with the bank model of the LDS
// (tests/test_logmel_emu.py::test_lds_layouts_are_conflict_free: every wave-wide access below is conflict-free):
//   L1 (pass 1 -> pass 2)  (n2, k1, n3) at n2*128 + 8*(k1 ^ (n2 & 1)) + n3        written by n2-pairs, read by k1 runs
//   L2 (pass 2 -> pass 3)  (k2, k1, n3) at k2*128 + 8*k1 + 2*(((n3 >> 1) + (k1 >> 2)) & 3) + (n3 & 1)
//                          (the four 16-byte pieces of a 64-byte run are rotated by k1 >> 2: sixteen lanes reading the runs
//                          of sixteen k1 with ds_read_b128 then hit sixteen different 16-byte slots)
//   L3 (pass 3 -> untangle) Z[k] at k, k = k1 + 16*k2 + 128*k3; pass 3 maps lane -> (k1 = lane & 15, k2 = lane >> 4 (+ 4)),
//                          so every 16-lane store group covers 16 consecutive k
//   P  (untangle -> mel)   the 1025 powers as floats at the start of the buffer, the staged output row behind them
// Index maps: n = 64*n1 + 8*n2 + n3, k = k1 + 16*k2 + 128*k3 (fft1024_phases.h).  Like the other phase headers the
// bodies have no cross-lane intrinsics and compile for the host (tests/emu/logmel2_emu.cpp).
#pragma once
#include "logmel_phases.h"

namespace adt {

constexpr int kL2Half = 1024;                    // complex FFT length
constexpr int kL2Buf = 1024;                     // cf per wave (8 KiB)
constexpr int kL2Stage = 1040;                   // float slot of the staged output row (powers use floats 0..1024)

ADT_HD int l1_index(int n2, int k1, int n3) { return n2 * 128 + 8 * (k1 ^ (n2 & 1)) + n3; }
// Twiddle tables of this generation (no sign logic in the hot loops): t1k[j] = W_1024^j for the whole circle, j < 1024
// (the radix passes), t2k[k] = W_2048^k, k <= 512 (the untangling).
ADT_HD cf tw1k(const cf* t1k, int j) { return t1k[j & 1023]; }
ADT_HD int l2_index(int k2, int k1, int n3) { return k2 * 128 + 8 * k1 + 2 * (((n3 >> 1) + (k1 >> 2)) & 3) + (n3 & 1); }

// ---- pass 1: window, pack even / odd samples, radix-16 over n1, twiddle W_128^(n2 k1), store to L1 ----------------
// base: first sample of the frame (may be negative or run past L when kInterior is false); win2: the window as 1024 (even, odd) pairs.
template <bool kInterior, bool kPairLoads>
ADT_HD void l2_pass1(int lane, const float* clip, int L, int base, const cf* win2, const cf* t1k, cf* buf) {
  const int m = lane;                    // 8*n2 + n3
  const int n2 = m >> 3;
  cf z[16];
  const float* p = clip + base + 2 * m;  // interior: sample pair n1 sits at p[128 n1], p[128 n1 + 1] (constant offsets)
  const cf* wp = win2 + m;
  _Pragma("unroll")
  for (int n1 = 0; n1 < 16; ++n1) {
    const cf w = wp[64 * n1];
    float x0, x1;
    if (kInterior) {
      if (kPairLoads) { const cf x = *reinterpret_cast<const cf*>(p + 128 * n1); x0 = x.x; x1 = x.y; }
      else { x0 = p[128 * n1]; x1 = p[128 * n1 + 1]; }
    } else {
      const int s0 = base + 2 * (m + 64 * n1);
      x0 = clip[reflect_index(s0, L)];
      x1 = clip[reflect_index(s0 + 1, L)];
    }
    z[n1].x = w.x * x0;
    z[n1].y = w.y * x1;
  }
  dft16(z);
  // W_128^(n2*k1) = W_1024^(8*n2*k1), k1 = 4q + r, as table[W^(32*n2*q)] * table[W^(8*n2*r)]: 6 table reads, not 15
  cf sr[4], bq[4];
  _Pragma("unroll")
  for (int r = 1; r < 4; ++r) { sr[r] = tw1k(t1k, 8 * n2 * r); bq[r] = tw1k(t1k, 32 * n2 * r); }
  // L1 address of (n2, k1, n3): n2*128 + 8*(k1 ^ b) + n3 with b = n2 & 1 = two lane-constant bases + 8*k1 (immediate offsets)
  const int b = n2 & 1;
  cf* even = buf + n2 * 128 + (m & 7) + 8 * b;       // even k1: k1 ^ b = k1 + b
  cf* odd = buf + n2 * 128 + (m & 7) - 8 * b;        // odd k1:  k1 ^ b = k1 - b
  _Pragma("unroll")
  for (int k1 = 0; k1 < 16; ++k1) {
    const int q = k1 >> 2, r = k1 & 3;
    cf v = z[k1];
    if (q != 0 && r != 0) v = cmul(v, cmul(bq[q], sr[r]));
    else if (q != 0) v = cmul(v, bq[q]);
    else if (r != 0) v = cmul(v, sr[r]);
    ((k1 & 1) ? odd : even)[8 * k1] = v;
  }
}

// ---- pass 2: radix-8 over n2, twiddle W_1024^(n3 (k1 + 16 k2)), L1 -> L2 (two items per lane; not in place: load both, then store both)
ADT_HD void l2_pass2_load(int lane, int it, const cf* buf, cf* z /*8*/) {
  const int c = lane + 64 * it, k1 = c >> 3, n3 = c & 7;
  const cf* even = buf + 8 * k1 + n3;                 // even n2: k1 ^ 0
  const cf* odd = buf + 8 * (k1 ^ 1) + n3;            // odd n2:  k1 ^ 1
  _Pragma("unroll")
  for (int n2 = 0; n2 < 8; ++n2) z[n2] = ((n2 & 1) ? odd : even)[128 * n2];
}
// The eight twiddles W_1024^(n3 (k1 + 16 k2)) of item `it` depend on the lane only: the kernel reads them ONCE, before its frame loop, and
// keeps them in registers (l2_pass2_twiddles) -- their table reads were the conflict-prone ones (per-lane scattered 8-byte reads) of a
// kernel that round 5 left LDS-bound.
ADT_HD void l2_pass2_twiddles(int lane, int it, const cf* t1k, cf* tw /*8*/) {
  const int c = lane + 64 * it, k1 = c >> 3, n3 = c & 7;
  const int j0 = n3 * k1, step = 16 * n3;
  _Pragma("unroll")
  for (int k2 = 0; k2 < 8; ++k2) tw[k2] = tw1k(t1k, j0 + step * k2);
}
ADT_HD void l2_pass2_store_tw(int lane, int it, cf* z /*8*/, const cf* tw /*8*/, cf* buf) {
  const int c = lane + 64 * it, k1 = c >> 3, n3 = c & 7;
  dft8(z);
  cf* dst = buf + l2_index(0, k1, n3);                // + 128 * k2
  _Pragma("unroll")
  for (int k2 = 0; k2 < 8; ++k2) dst[128 * k2] = cmul(z[k2], tw[k2]);
}
ADT_HD void l2_pass2_store(int lane, int it, cf* z /*8*/, const cf* t1k, cf* buf) {
  cf tw[8];
  l2_pass2_twiddles(lane, it, t1k, tw);
  l2_pass2_store_tw(lane, it, z, tw, buf);
}

// ---- pass 3: radix-8 over n3, L2 -> L3 (two items per lane) ---------------------------------------------------------
ADT_HD void l2_pass3_load(int lane, int it, const cf* buf, cf* z /*8*/) {
  const int k1 = lane & 15, k2 = (lane >> 4) + 4 * it;
  const cf* row = buf + k2 * 128 + 8 * k1;
  _Pragma("unroll")
  for (int j = 0; j < 4; ++j) {          // piece j of the run sits at position (j + (k1 >> 2)) & 3: one 16-byte read
    const cf* p = row + 2 * ((j + (k1 >> 2)) & 3);
    z[2 * j] = p[0];
    z[2 * j + 1] = p[1];
  }
}
ADT_HD void l2_pass3_store(int lane, int it, cf* z /*8*/, cf* buf) {
  const int k1 = lane & 15, k2 = (lane >> 4) + 4 * it;
  dft8(z);
  _Pragma("unroll")
  for (int k3 = 0; k3 < 8; ++k3) buf[k1 + 16 * k2 + 128 * k3] = z[k3];
}

// ---- untangle: the two power bins that Z[k] and Z[1024 - k] determine ------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
ADT_HD void real_power_pair(cf a, cf b, cf w, float& pk, float& pnk) {      // the same on register pairs (logmel_phases.h): 14 packed / scalar instructions
  const cf cb = b * cf{0.5f, -0.5f};                  // conj(b) / 2
  const cf e = a * 0.5f + cb, d = a * 0.5f - cb;      // E = (a + conj b) / 2;  d = (a - conj b) / 2,  O = -i d
  const cf t = cmul(w, mul_mi(d));
  const cf u = e + t, v = e - t;
  const cf uu = u * u, vv = v * v;
  pk = uu.x + uu.y;
  pnk = vv.x + vv.y;
}
#else
ADT_HD void real_power_pair(cf a, cf b, cf w, float& pk, float& pnk) {      // a = Z[k], b = Z[1024 - k], w = W_2048^k
  const cf e = {0.5f * (a.x + b.x), 0.5f * (a.y - b.y)};
  const cf o = {0.5f * (a.y + b.y), -0.5f * (a.x - b.x)};
  const cf t = cmul(w, o);
  const cf u = cadd(e, t), v = csub(e, t);
  pk = u.x * u.x + u.y * u.y;
  pnk = v.x * v.x + v.y * v.y;
}
#endif
// lane -> bins k = lane + 64 i (i < 8) and 1024 - k; lane 0 also bin 512.  Reads L3, returns the powers in registers.
ADT_HD void l2_untangle_load(int lane, const cf* t2k, const cf* buf, float* pk /*8*/, float* pnk /*8*/, float& p512) {
  const cf* up = buf + lane;                                    // Z[k], k = lane + 64 i
  const cf* dn = buf + ((kL2Half - lane) & (kL2Half - 1));      // Z[1024 - k] = dn[-64 i] (i = 0, lane 0: Z[0] itself)
  _Pragma("unroll")
  for (int i = 0; i < 8; ++i) {
    const cf b = (i == 0) ? dn[0] : (lane == 0 ? buf[kL2Half - 64 * i] : dn[-64 * i]);
    real_power_pair(up[64 * i], b, t2k[lane + 64 * i], pk[i], pnk[i]);
  }
  p512 = 0.f;
  if (lane == 0) { const cf a = buf[512]; p512 = a.x * a.x + a.y * a.y; }
}
ADT_HD void l2_untangle_store(int lane, const float* pk, const float* pnk, float p512, float* pw) {
  _Pragma("unroll")
  for (int i = 0; i < 8; ++i) {
    const int k = lane + 64 * i;
    pw[k] = pk[i];
    pw[kL2Half - k] = pnk[i];
  }
  if (lane == 0) pw[512] = p512;
}

// ---- banded mel reduction over the float powers: lane = 4*g + s handles mels j = g + 16*i, bins lo+s, lo+s+4, ... ----
ADT_HD float l2_mel_partial(int s, int lo, int cnt, int off, const float* melw, const float* pw) {
  float acc = 0.f;
  for (int idx = s; idx < cnt; idx += 4) acc += melw[off + idx] * pw[lo + idx];
  return acc;
}

// post-processing of model.py:91-93; on the device the logarithm is the hardware v_log_f32 (|error| ~ 1 ulp of log2: 4e-6 on
// the value, 1e-7 on the [0, 1] output -- the kernel's tolerance is 2e-5).
ADT_HD float post_fast(float mel, float eps, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  float v = __logf(mel + eps);
#else
  float v = std::log(mel + eps);
#endif
  v = v < lo ? lo : (v > hi ? hi : v);
  return (v - lo) / (hi - lo);
}

}  // namespace adt
