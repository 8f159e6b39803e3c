// gelu.h -- exact-erf GELU (and its derivative) on scalars and register pairs, shared by the GEMM epilogues (gemm.hip) and the
// fused HTSAT row-block kernels (htsat_fused.hip).  Included INSIDE namespace adt.
#pragma once

// Exact-erf GELU (torch default) through erfc(a) ~= t * P5(t) * exp(-a^2), t = 1 / (1 + 0.3275911 a) (Abramowitz-Stegun 7.1.26,
// |error| <= 1.5e-7 on erf): one v_exp, one v_rcp, straight-line; the same exp(-x^2/2) is the Gaussian of the derivative.
// Phi(x) is formed from erfc on the side where it does not cancel.
__device__ __forceinline__ float gauss_cdf(float x, float& e) {
  const float a = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  e = __expf(-0.5f * x * x);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float hq = 0.5f * poly * e;
  return x < 0.f ? hq : 1.0f - hq;
}
__device__ __forceinline__ float gelu_erf(float x) { float e; return x * gauss_cdf(x, e); }
__device__ __forceinline__ float gelu_erf_grad(float x) { float e; const float c = gauss_cdf(x, e); return fmaf(x * 0.3989422804014327f, e, c); }
// The same arithmetic on register pairs (v_pk_fma_f32 / v_pk_mul_f32: two elements per instruction; only rcp and exp stay scalar):
// hq = 0.5 erfc(|x| / sqrt 2) and e = exp(-x^2 / 2) of both elements.  GELU is then max(x, 0) - |x| hq on either side of 0.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 half_erfc2(f32x2 x, f32x2 ax, f32x2& e) {
  const f32x2 den = ax * 0.23164190f + 1.0f;                               // 0.3275911 / sqrt 2
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 arg = (x * x) * -0.72134752f;                                // -0.5 log2 e
  e = f32x2{__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  const f32x2 poly = t * ((((t * 0.5307027145f + -0.7265760135f) * t + 0.7107068705f) * t + -0.142248368f) * t + 0.127414796f);   // 0.5 P5
  return poly * e;
}
__device__ __forceinline__ f32x2 abs2(f32x2 x) { return f32x2{fabsf(x[0]), fabsf(x[1])}; }
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  f32x2 e;
  const f32x2 ax = abs2(x), hq = half_erfc2(x, ax, e);
  return f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)} - ax * hq;
}
__device__ __forceinline__ f32x2 gelu_erf_grad2(f32x2 x) {               // Phi(x) + x phi(x)
  f32x2 e;
  const f32x2 ax = abs2(x), hq = half_erfc2(x, ax, e);
  const f32x2 d = 0.5f - hq;                                              // Phi = 0.5 + sign(x) (0.5 - hq)
  const f32x2 c = f32x2{copysignf(d[0], x[0]), copysignf(d[1], x[1])} + 0.5f;
  return (x * 0.3989422804014327f) * e + c;
}

// GELU and its derivative from one erfc / exp evaluation (the saved-factor form of the FFN backward, adt_gemm_epilogue.act_grad_mode)
__device__ __forceinline__ void gelu_and_grad2(f32x2 x, f32x2& gv, f32x2& gd) {
  f32x2 e;
  const f32x2 ax = abs2(x), hq = half_erfc2(x, ax, e);
  gv = f32x2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)} - ax * hq;
  const f32x2 d = 0.5f - hq;
  gd = (x * 0.3989422804014327f) * e + (f32x2{copysignf(d[0], x[0]), copysignf(d[1], x[1])} + 0.5f);
}



// A cheaper GELU for activations that are rounded to bf16 right away (the fused HTSAT MLP, where the vector pipe is the bound):
// Q(|x|) = 0.5 erfc(|x| / sqrt 2) = 2^P6(|x|) on [0, 5.5] (|x| clamped there: Q(5.5) = 2e-8), one v_exp and six packed FMAs per pair,
// no reciprocal; GELU(x) = max(x, 0) - |x| Q.  Against the exact-erf value: |error| <= 4e-6 absolute, <= 1.5e-3 relative wherever
// |GELU| > 1e-4 -- below half a bf16 ulp (2e-3) everywhere (fit and bounds: Chebyshev fit of log2 Q, evaluated in fp32).
__device__ __forceinline__ float absmin_(float x, float cap) {
  float r;
  asm("v_min_f32 %0, |%1|, %2" : "=v"(r) : "v"(x), "v"(cap));
  return r;
}
__device__ __forceinline__ float max0_(float x) {            // (asm: fmaxf on an MFMA result gets a canonicalising v_max in front)
  float r;
  asm("v_max_f32 %0, %1, 0" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ f32x2 gelu_bf16_2(f32x2 x) {
  const f32x2 ax = {absmin_(x[0], 5.5f), absmin_(x[1], 5.5f)};
  f32x2 p = ax * 2.6412943043396808e-05f + -0.00066360057098791f;
  p = p * ax + 0.007492306642234325f;
  p = p * ax + -0.051936905831098557f;
  p = p * ax + -0.46045857667922974f;
  p = p * ax + -1.150443434715271f;
  p = p * ax + -1.0000735521316528f;
  // GELU = max(x, 0) - |x| Q on either side of 0
  return f32x2{fmaf(-fabsf(x[0]), __builtin_amdgcn_exp2f(p[0]), max0_(x[0])), fmaf(-fabsf(x[1]), __builtin_amdgcn_exp2f(p[1]), max0_(x[1]))};
}
// two pairs at once, the two polynomial chains written interleaved (one chain of dependent packed FMAs pays a wait state per instruction)
__device__ __forceinline__ void gelu_bf16_4(f32x2 xa, f32x2 xb, f32x2& ga, f32x2& gb) {
  const float ma0 = max0_(xa[0]), ma1 = max0_(xa[1]), mb0 = max0_(xb[0]), mb1 = max0_(xb[1]);       // (early: far from their use)
  const f32x2 aa = {absmin_(xa[0], 5.5f), absmin_(xa[1], 5.5f)}, ab = {absmin_(xb[0], 5.5f), absmin_(xb[1], 5.5f)};
  f32x2 pa = aa * 2.6412943043396808e-05f + -0.00066360057098791f;
  f32x2 pb = ab * 2.6412943043396808e-05f + -0.00066360057098791f;
  pa = pa * aa + 0.007492306642234325f;   pb = pb * ab + 0.007492306642234325f;
  pa = pa * aa + -0.051936905831098557f;  pb = pb * ab + -0.051936905831098557f;
  pa = pa * aa + -0.46045857667922974f;   pb = pb * ab + -0.46045857667922974f;
  pa = pa * aa + -1.150443434715271f;     pb = pb * ab + -1.150443434715271f;
  pa = pa * aa + -1.0000735521316528f;    pb = pb * ab + -1.0000735521316528f;
  const float ea0 = __builtin_amdgcn_exp2f(pa[0]), eb0 = __builtin_amdgcn_exp2f(pb[0]), ea1 = __builtin_amdgcn_exp2f(pa[1]), eb1 = __builtin_amdgcn_exp2f(pb[1]);
  ga = f32x2{fmaf(-fabsf(xa[0]), ea0, ma0), fmaf(-fabsf(xa[1]), ea1, ma1)};
  gb = f32x2{fmaf(-fabsf(xb[0]), eb0, mb0), fmaf(-fabsf(xb[1]), eb1, mb1)};
}
