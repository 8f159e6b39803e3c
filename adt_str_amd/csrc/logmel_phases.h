// logmel_phases.h -- building blocks shared by the per-lane phase bodies of the FFT front ends: K1 (logmel2_phases.h, one
// real 2048-sample frame per wave) and K9 (fft1024_phases.h, two packed 1024-sample frames per wave).
//
// Complex helpers, the in-register radix-4 / 8 / 16 DFTs, the half-circle twiddle lookup, torch's reflect padding index,
// the packed-pair power untangling, the banded mel partial sum and the reference's log / clamp / scale post-processing.
//
// Every function is a plain function of (lane, buffers) with no cross-lane intrinsics, so the same source is compiled for
// gfx950 and for the host (tests/emu/*.cpp run each phase for lane 0..63), which lets index maps and twiddles be checked
// on a machine without a GPU.
#pragma once

#if defined(__HIPCC__)
#define ADT_HD __host__ __device__ __forceinline__
#else
#define ADT_HD inline
#include <cmath>
#endif

namespace adt {

constexpr int kNfft = 2048;

// Complex arithmetic.  On the device a complex number is a register PAIR (ext_vector_type(2)) and every operation below is written as
// whole-pair arithmetic the compiler turns into ONE packed instruction each (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel
// swizzles): a rotation by -i is a swapped operand times the constant pair (1, -1) inside an FMA, a complex product is a packed
// multiply and a packed FMA (plus one packed multiply for i b when b is not a compile-time constant).  Written element by element
// (the struct form the host emulator keeps) the same kernels carried ~200 v_mov / v_pk_mov register shuffles per frame -- a sixth of
// their instructions -- because the compiler folds a swizzle into a packed operand but not a swizzle with a per-element sign
// (round 5: K1 and K9 are bound by vector instruction issue, so instructions are what counts).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float cf __attribute__((ext_vector_type(2)));
ADT_HD cf cswap_(cf a) { return __builtin_shufflevector(a, a, 1, 0); }
ADT_HD cf cre_(cf a) { return __builtin_shufflevector(a, a, 0, 0); }
ADT_HD cf cim_(cf a) { return __builtin_shufflevector(a, a, 1, 1); }
ADT_HD cf cadd(cf a, cf b) { return a + b; }
ADT_HD cf csub(cf a, cf b) { return a - b; }
ADT_HD cf mul_mi(cf a) { return cswap_(a) * cf{1.f, -1.f}; }                       // a * (-i)
ADT_HD cf cadd_mi(cf a, cf b) { return cswap_(b) * cf{1.f, -1.f} + a; }            // a + (-i) b
ADT_HD cf csub_mi(cf a, cf b) { return cswap_(b) * cf{-1.f, 1.f} + a; }            // a - (-i) b
ADT_HD cf cmul_k(cf a, float cr, float ci) { return cre_(a) * cf{cr, ci} + cim_(a) * cf{-ci, cr}; }   // a * (cr + i ci), constants
ADT_HD cf cmul(cf a, cf b) { const cf ib = cswap_(b) * cf{-1.f, 1.f}; return cre_(a) * b + cim_(a) * ib; }
#else
struct cf { float x, y; };
ADT_HD cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
ADT_HD cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
ADT_HD cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
ADT_HD cf mul_mi(cf a) { return {a.y, -a.x}; }   // a * (-i)
ADT_HD cf cadd_mi(cf a, cf b) { return {a.x + b.y, a.y - b.x}; }
ADT_HD cf csub_mi(cf a, cf b) { return {a.x - b.y, a.y + b.x}; }
ADT_HD cf cmul_k(cf a, float cr, float ci) { return cmul(a, cf{cr, ci}); }
#endif

// W_2048^j from the half-circle table tw[0..1023] (W^(j+1024) = -W^j).
ADT_HD cf twiddle(const cf* tw, int j) {
  j &= 2047;
  cf v = tw[j & 1023];
  if (j & 1024) { v.x = -v.x; v.y = -v.y; }
  return v;
}

ADT_HD void dft4(cf& a, cf& b, cf& c, cf& d) {   // forward (e^-i) 4-point, in place, natural order
  cf t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
  a = cadd(t0, t2); b = cadd_mi(t1, t3); c = csub(t0, t2); d = csub_mi(t1, t3);
}

// forward 16-point DFT, in place, natural order.  n = 4a + b, k = c + 4d.
ADT_HD void dft16(cf* x) {
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  _Pragma("unroll")
  for (int b = 0; b < 4; ++b) dft4(x[b], x[4 + b], x[8 + b], x[12 + b]);   // over a; result index c at x[4c + b]
  // twiddle W_16^(b*c) on element x[4c + b]
  x[4 * 1 + 1] = cmul_k(x[4 * 1 + 1], c1, -s1);        // bc = 1
  x[4 * 1 + 2] = cmul_k(x[4 * 1 + 2], h, -h);          // 2
  x[4 * 1 + 3] = cmul_k(x[4 * 1 + 3], s1, -c1);        // 3
  x[4 * 2 + 1] = cmul_k(x[4 * 2 + 1], h, -h);          // 2
  x[4 * 2 + 2] = mul_mi(x[4 * 2 + 2]);                 // 4
  x[4 * 2 + 3] = cmul_k(x[4 * 2 + 3], -h, -h);         // 6
  x[4 * 3 + 1] = cmul_k(x[4 * 3 + 1], s1, -c1);        // 3
  x[4 * 3 + 2] = cmul_k(x[4 * 3 + 2], -h, -h);         // 6
  x[4 * 3 + 3] = cmul_k(x[4 * 3 + 3], -c1, s1);        // 9
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
  o1 = cmul_k(o1, h, -h);
  o3 = cmul_k(o3, -h, -h);
  x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
  x[1] = cadd(e1, o1); x[5] = csub(e1, o1);
  x[2] = cadd_mi(e2, o2); x[6] = csub_mi(e2, o2);      // o2 * (-i)
  x[3] = cadd(e3, o3); x[7] = csub(e3, o3);
}

// Reflect an index into [0, L) the way torch.stft(center=True, pad_mode="reflect") pads.
ADT_HD int reflect_index(int s, int L) {
  if (s < 0) s = -s;
  if (s >= L) s = 2 * (L - 1) - s;
  return s;
}

// ---- untangle the packed spectrum into the two power spectra, in place ----------------
// buf[k] <- (|X0[k]|^2, |X1[k]|^2), k = 0..1024.
ADT_HD cf power_pair(cf a, cf b) {          // a = Z[k], b = Z[N-k]
  const float ar = a.x + b.x, ai = a.y - b.y;   // 2*X0
  const float br = a.y + b.y, bi = b.x - a.x;   // 2*X1
  return {0.25f * (ar * ar + ai * ai), 0.25f * (br * br + bi * bi)};
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
