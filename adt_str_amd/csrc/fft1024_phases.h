// fft1024_phases.h -- per-lane phase bodies of a 1024-point complex FFT done by ONE wave (64 lanes x 16 points),
// used by the CLAP log-mel front end (K9): two real 1024-sample frames are packed as z = x0 + i*x1.
//
//   N = 1024 = 16 x 8 x 8:   n = 64*n1 + 8*n2 + n3,   k = k1 + 16*k2 + 128*k3
//   pass 1 (1 item/lane):  A[k1,n2,n3] = W_128^(n2*k1)          * sum_n1 z[n] W_16^(n1*k1)   -> L1 layout (logmel2_phases.h)
//   pass 2 (2 items/lane): B[k1,k2,n3] = W_1024^(n3*(k1+16*k2)) * sum_n2 A    W_8^(n2*k2)    -> L2 layout   } K1's second-generation
//   pass 3 (2 items/lane): Z[k]        =                          sum_n3 B    W_8^(n3*k3)    -> buf[k]      } passes, shared
// Like logmel_phases.h the bodies have no cross-lane intrinsics and compile for the host
// (tests/emu/clap_logmel_emu.cpp) as well as for gfx950.
#pragma once
#include "logmel2_phases.h"      // (brings logmel_phases.h; passes 2 / 3 and the L1 / L2 / L3 layouts of the second generation)

namespace adt {

constexpr int kN1k = 1024;
constexpr int kStage1k = 560;                   // cf slot where the 2 x n_mels output rows are staged (bins use 0..512)

// sample `s` of a clip of `n` samples under CLAP's "repeatpad" (tile floor(target/n) times, then zeros) and the
// reflect padding of spectrogram(center=True) at both ends of the target-length signal
ADT_HD float repeatpad_sample(const float* clip, int n, int target, int s) {
  if (s < 0) s = -s;
  if (s >= target) s = 2 * (target - 1) - s;
  const int filled = (target / n) * n;
  return s < filled ? clip[s % n] : 0.0f;
}

// Pass 1 -- two real frames as one complex sequence, window, radix-16 over n1, twiddle W_128^(n2 k1) -- stored into the L1 layout of
// logmel2_phases.h, so that K1's conflict-free passes 2 and 3 (l2_pass2_*, l2_pass3_*: twiddles in registers, 16-byte runs rotated by
// k1 >> 2) finish the 1024-point FFT (round 6; the first generation kept its own pitch-130 passes: 1.05 -> 0.97 ms per 512 clips);
// t1k = W_1024^j for the whole circle.
// kInterior: both frames lie inside [0, target) -> no reflection, and the position inside the repeated clip is advanced incrementally
// (one integer modulo per frame instead of one per sample).
template <bool kInterior>
ADT_HD void p1k_pass1_l1(int lane, const float* clip, int n, int target, int base0, int base1, bool has1,
                         const float* win16, const cf* t1k, cf* buf) {
  const int m = lane;                    // 8*n2 + n3
  const int n2 = m >> 3;
  cf z[16];
  if (kInterior) {
    const int filled = (target / n) * n;
    int s0 = base0 + m, s1 = base1 + m;
    int p0 = s0 % n, p1 = s1 % n;
    _Pragma("unroll")
    for (int n1 = 0; n1 < 16; ++n1) {
      const float w = win16[n1];
      z[n1].x = s0 < filled ? w * clip[p0] : 0.0f;
      z[n1].y = (has1 && s1 < filled) ? w * clip[p1] : 0.0f;
      s0 += 64; s1 += 64; p0 += 64; p1 += 64;
      if (p0 >= n) p0 = (p0 - n < n) ? p0 - n : p0 % n;
      if (p1 >= n) p1 = (p1 - n < n) ? p1 - n : p1 % n;
    }
  } else {
    _Pragma("unroll")
    for (int n1 = 0; n1 < 16; ++n1) {
      const int o = m + 64 * n1;
      const float w = win16[n1];
      z[n1].x = w * repeatpad_sample(clip, n, target, base0 + o);
      z[n1].y = has1 ? w * repeatpad_sample(clip, n, target, base1 + o) : 0.0f;
    }
  }
  dft16(z);
  cf sr[4], bq[4];                       // W_128^(n2*k1), k1 = 4q + r, from 6 table reads
  _Pragma("unroll")
  for (int r = 1; r < 4; ++r) { sr[r] = tw1k(t1k, 8 * n2 * r); bq[r] = tw1k(t1k, 32 * n2 * r); }
  const int b = n2 & 1;                  // L1 address of (n2, k1, n3): n2*128 + 8*(k1 ^ b) + n3 = two lane-constant bases + 8*k1
  cf* even = buf + n2 * 128 + (m & 7) + 8 * b;
  cf* odd = buf + n2 * 128 + (m & 7) - 8 * b;
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

// buf[k] <- (|X0[k]|^2, |X1[k]|^2), k = 0..512, in place
ADT_HD void p1k_untangle(int lane, cf* buf) {
  cf p[8];
  _Pragma("unroll")
  for (int i = 0; i < 8; ++i) {
    const int k = lane + 64 * i;
    p[i] = power_pair(buf[k], buf[(kN1k - k) & (kN1k - 1)]);
  }
  cf pn = {0.f, 0.f};
  if (lane == 0) pn = power_pair(buf[512], buf[512]);
  _Pragma("unroll")
  for (int i = 0; i < 8; ++i) buf[lane + 64 * i] = p[i];
  if (lane == 0) buf[512] = pn;
}

// 10 * log10(max(mel, amin))  (transformers.audio_utils.power_to_db, reference 1.0, no db range)
ADT_HD float to_db(float mel, float amin) {
  const float v = mel > amin ? mel : amin;
#if defined(__HIP_DEVICE_COMPILE__)
  return 10.0f * log10f(v);
#else
  return 10.0f * std::log10(v);
#endif
}

}  // namespace adt
