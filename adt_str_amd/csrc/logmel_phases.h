// logmel_phases.h -- per-lane phase bodies of the fused log-mel kernel (K1).
//
// One wave (64 lanes) turns TWO audio frames into two rows of normalised log-mel:
// the frames are packed as z = x0 + i*x1 and pushed through one complex 2048-point
// FFT (2048 = 16 x 16 x 8, three in-register radix passes with two exchanges through
// a wave-private LDS buffer), untangled into the two power spectra in place, reduced
// against the banded mel filterbank and post-processed.
//
// Every phase is a plain function of (lane, buffers) with no cross-lane intrinsics,
// so the same source is compiled for gfx950 (logmel.hip puts barriers between the
// phases) and for the host (tests/emu/logmel_emu.cpp runs each phase for lane 0..63),
// which lets the index maps and twiddles be checked on a machine without a GPU.
//
// Index maps (N = 2048, N1 = 16, N2 = 16, N3 = 8):
//   n = 128*n1 + 8*n2 + n3,  k = k1 + 16*k2 + 256*k3
//   pass 1: A[k1,n2,n3] = W_256^(n2*k1) * sum_n1 z[n] W_16^(n1*k1)       -> buf[n2*P + 8*k1 + n3]
//   pass 2: B[k1,k2,n3] = W_2048^(n3*(k1+16*k2)) * sum_n2 A W_16^(n2*k2) -> buf[k2*P + 8*k1 + n3] (in place)
//   pass 3: Z[k]        = sum_n3 B W_8^(n3*k3)                            -> buf[k]
// P = 130 keeps the 8-element runs 16-byte aligned and spreads rows over LDS banks.
#pragma once

#if defined(__HIPCC__)
#define ADT_HD __host__ __device__ __forceinline__
#else
#define ADT_HD inline
#include <cmath>
#endif

namespace adt {

struct cf { float x, y; };

constexpr int kNfft = 2048;
#ifndef ADT_LOGMEL_PITCH
#define ADT_LOGMEL_PITCH 130
#endif
constexpr int kRowPitch = ADT_LOGMEL_PITCH;    // P, in cf units (even: 8-element runs stay 16-byte aligned)
constexpr int kBufElems = 16 * kRowPitch;      // 2080 cf = 16,640 B per wave
constexpr int kStageBase = 1100;               // cf slot where the 2 x n_mels output rows are staged

ADT_HD cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
ADT_HD cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
ADT_HD cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
ADT_HD cf mul_mi(cf a) { return {a.y, -a.x}; }   // a * (-i)

// W_2048^j from the half-circle table tw[0..1023] (W^(j+1024) = -W^j).
ADT_HD cf twiddle(const cf* tw, int j) {
  j &= 2047;
  cf v = tw[j & 1023];
  if (j & 1024) { v.x = -v.x; v.y = -v.y; }
  return v;
}

ADT_HD void dft4(cf& a, cf& b, cf& c, cf& d) {   // forward (e^-i) 4-point, in place, natural order
  cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = mul_mi(csub(b, d));
  a = cadd(t0, t2); b = cadd(t1, t3); c = csub(t0, t2); d = csub(t1, t3);
}

// forward 16-point DFT, in place, natural order.  n = 4a + b, k = c + 4d.
ADT_HD void dft16(cf* x) {
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  _Pragma("unroll")
  for (int b = 0; b < 4; ++b) dft4(x[b], x[4 + b], x[8 + b], x[12 + b]);   // over a; result index c at x[4c + b]
  // twiddle W_16^(b*c) on element x[4c + b]
  x[4 * 1 + 1] = cmul(x[4 * 1 + 1], cf{c1, -s1});      // bc = 1
  x[4 * 1 + 2] = cmul(x[4 * 1 + 2], cf{h, -h});        // 2
  x[4 * 1 + 3] = cmul(x[4 * 1 + 3], cf{s1, -c1});      // 3
  x[4 * 2 + 1] = cmul(x[4 * 2 + 1], cf{h, -h});        // 2
  x[4 * 2 + 2] = mul_mi(x[4 * 2 + 2]);                 // 4
  x[4 * 2 + 3] = cmul(x[4 * 2 + 3], cf{-h, -h});       // 6
  x[4 * 3 + 1] = cmul(x[4 * 3 + 1], cf{s1, -c1});      // 3
  x[4 * 3 + 2] = cmul(x[4 * 3 + 2], cf{-h, -h});       // 6
  x[4 * 3 + 3] = cmul(x[4 * 3 + 3], cf{-c1, s1});      // 9
  _Pragma("unroll")
  for (int c = 0; c < 4; ++c) dft4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);  // over b; result d at x[4c + d]
  // x[4c + d] holds X[c + 4d]: transpose the 4x4 to natural order
  _Pragma("unroll")
  for (int c = 0; c < 4; ++c)
    _Pragma("unroll")
    for (int d = c + 1; d < 4; ++d) { cf t = x[4 * c + d]; x[4 * c + d] = x[4 * d + c]; x[4 * d + c] = t; }
}

// forward 8-point DFT, in place, natural order.
ADT_HD void dft8(cf* x) {
  const float h = 0.70710678118654752f;
  cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
  cf o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  o1 = cmul(o1, cf{h, -h});
  o2 = mul_mi(o2);
  o3 = cmul(o3, cf{-h, -h});
  x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
  x[1] = cadd(e1, o1); x[5] = csub(e1, o1);
  x[2] = cadd(e2, o2); x[6] = csub(e2, o2);
  x[3] = cadd(e3, o3); x[7] = csub(e3, o3);
}

// Reflect an index into [0, L) the way torch.stft(center=True, pad_mode="reflect") pads.
ADT_HD int reflect_index(int s, int L) {
  if (s < 0) s = -s;
  if (s >= L) s = 2 * (L - 1) - s;
  return s;
}

// ---- pass 1: window, pack two frames, radix-16 over n1, twiddle, store ----------------
// clip: the clip's samples; base0/base1: first sample of each frame (may be negative or run
// past L when kInterior is false).  win: the 2048-entry window table.
template <bool kInterior>
ADT_HD void pass1(int lane, int it, const float* clip, int L, int base0, int base1, bool has1,
                  const float* win16 /*window at m + 128*n1, n1 = 0..15*/, const cf* tw, cf* buf) {
  const int m = lane + 64 * it;          // 8*n2 + n3
  const int n2 = m >> 3;
  cf z[16];
  _Pragma("unroll")
  for (int n1 = 0; n1 < 16; ++n1) {
    const int o = m + 128 * n1;
    int s0 = base0 + o, s1 = base1 + o;
    if (!kInterior) { s0 = reflect_index(s0, L); s1 = reflect_index(s1, L); }
    const float w = win16[n1];
    z[n1].x = w * clip[s0];
    z[n1].y = has1 ? w * clip[s1] : 0.0f;
  }
  dft16(z);
  // twiddle W_256^(n2*k1), k1 = 4q + r, as table[W^(n2*4q)] * table[W^(n2*r)]: 6 table reads, not 15
  cf sr[4], bq[4];
  _Pragma("unroll")
  for (int r = 1; r < 4; ++r) { sr[r] = twiddle(tw, 8 * n2 * r); bq[r] = twiddle(tw, 32 * n2 * r); }
  _Pragma("unroll")
  for (int k1 = 0; k1 < 16; ++k1) {
    const int q = k1 >> 2, r = k1 & 3;
    cf v = z[k1];
    if (q != 0 && r != 0) v = cmul(v, cmul(bq[q], sr[r]));
    else if (q != 0) v = cmul(v, bq[q]);
    else if (r != 0) v = cmul(v, sr[r]);
    buf[n2 * kRowPitch + 8 * k1 + (m & 7)] = v;
  }
}

// ---- pass 2: radix-16 over n2, twiddle, store in place -------------------------------
ADT_HD void pass2(int lane, int it, const cf* tw, cf* buf) {
  const int c = lane + 64 * it;          // 8*k1 + n3
  const int k1 = c >> 3, n3 = c & 7;
  cf z[16];
  _Pragma("unroll")
  for (int n2 = 0; n2 < 16; ++n2) z[n2] = buf[n2 * kRowPitch + c];
  dft16(z);
  // twiddle W_2048^(n3*(k1 + 16*k2)), k2 = 4q + r, as table[W^(n3*(k1 + 64q))] * table[W^(16*n3*r)]
  cf sr[4], bq[4];
  _Pragma("unroll")
  for (int r = 1; r < 4; ++r) sr[r] = twiddle(tw, 16 * n3 * r);
  _Pragma("unroll")
  for (int q = 0; q < 4; ++q) bq[q] = twiddle(tw, n3 * (k1 + 64 * q));
  _Pragma("unroll")
  for (int k2 = 0; k2 < 16; ++k2) {
    const int q = k2 >> 2, r = k2 & 3;
    const cf t = (r == 0) ? bq[q] : cmul(bq[q], sr[r]);
    buf[k2 * kRowPitch + c] = cmul(z[k2], t);
  }
}

// ---- pass 3: lane -> (k1, k2) map chosen so 16-lane groups hit distinct 16-byte slots ----
ADT_HD void pass3_index(int lane, int it, int& k1, int& k2) {
  const int t = lane & 15;
  k1 = (t & 3) + 4 * (lane >> 4);
  k2 = (t >> 2) + 4 * it;
}
ADT_HD void pass3_load(int lane, int it, const cf* buf, cf* z /*8*/) {
  int k1, k2; pass3_index(lane, it, k1, k2);
  const cf* p = buf + k2 * kRowPitch + 8 * k1;
  _Pragma("unroll")
  for (int n3 = 0; n3 < 8; ++n3) z[n3] = p[n3];
}
ADT_HD void pass3_store(int lane, int it, cf* z /*8*/, cf* buf) {
  int k1, k2; pass3_index(lane, it, k1, k2);
  dft8(z);
  _Pragma("unroll")
  for (int k3 = 0; k3 < 8; ++k3) buf[k1 + 16 * k2 + 256 * k3] = z[k3];
}

// ---- untangle the packed spectrum into the two power spectra, in place ----------------
// buf[k] <- (|X0[k]|^2, |X1[k]|^2), k = 0..1024.
ADT_HD cf power_pair(cf a, cf b) {          // a = Z[k], b = Z[N-k]
  const float ar = a.x + b.x, ai = a.y - b.y;   // 2*X0
  const float br = a.y + b.y, bi = b.x - a.x;   // 2*X1
  return {0.25f * (ar * ar + ai * ai), 0.25f * (br * br + bi * bi)};
}
ADT_HD void untangle(int lane, cf* buf) {
  cf p[16];
  _Pragma("unroll")
  for (int i = 0; i < 16; ++i) {
    const int k = lane + 64 * i;
    p[i] = power_pair(buf[k], buf[(kNfft - k) & (kNfft - 1)]);
  }
  cf pn = {0.f, 0.f};
  if (lane == 0) pn = power_pair(buf[1024], buf[1024]);
  _Pragma("unroll")
  for (int i = 0; i < 16; ++i) buf[lane + 64 * i] = p[i];
  if (lane == 0) buf[1024] = pn;
}

// ---- banded mel reduction: lane = 4*g + s handles mels j = g + 16*i, bins lo+s, lo+s+4, ... ----
// Returns the partial sums of this lane (to be xor-reduced over s by the caller).
ADT_HD cf mel_partial(int s, int lo, int cnt, int off, const float* melw, const cf* buf) {
  cf acc = {0.f, 0.f};
  for (int idx = s; idx < cnt; idx += 4) {
    const float w = melw[off + idx];
    const cf p = buf[lo + idx];
    acc.x += w * p.x;
    acc.y += w * p.y;
  }
  return acc;
}

// post-processing of model.py:91-93 with (eps, lo, hi) = (1e-10, -23, 12).
ADT_HD float post(float mel, float eps, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  float v = logf(mel + eps);
#else
  float v = std::log(mel + eps);
#endif
  v = v < lo ? lo : (v > hi ? hi : v);
  return (v - lo) / (hi - lo);
}

}  // namespace adt
