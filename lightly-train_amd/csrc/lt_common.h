// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the DINOv2 hot path.
// wave = 64 lanes; all kernels are written for gfx950 only (no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lt_amd.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits (storage type in HBM / LDS)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LT_WAVE 64

// ---- error plumbing (host) -----------------------------------------------------------
void lt_set_error(const char* fmt, ...);
#define LT_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      lt_set_error(__VA_ARGS__);            \
      return LT_ERR_INVALID;                \
    }                                       \
  } while (0)
#define LT_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      lt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return LT_ERR_HIP;                                                        \
    }                                                                           \
    return LT_OK;                                                               \
  } while (0)

// ---- bf16 <-> f32 (device) -----------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// f32 -> bf16 round-to-nearest-even: the native fptrunc lowers to v_cvt_pk_bf16_f32 on gfx950 (one instruction per
// PAIR, NaN-safe) instead of ~10 integer ops + a NaN branch per element.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  bf16x2_t v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

// ---- wave / block reductions ---------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block reductions; `red` = LDS scratch of >= 32 floats; result broadcast to all threads.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = red[0];
  for (int i = 1; i < nw; ++i) s = fmaxf(s, red[i]);
  return s;
}

// erf-based GELU (nn.GELU default, layers/mlp.py:36-42) and its derivative.  erf by Abramowitz-Stegun 7.1.26
// (|abs error| <= 1.5e-7, below fp32 epilogue noise and far below the bf16 store) with ONE v_exp_f32:
// exp(-(x/sqrt2)^2) = exp(-x^2/2) is also the Gaussian pdf the derivative needs.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf_unnorm) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));  // v_rcp_f32 (1 ulp) instead of the IEEE division sequence
  const float ex = __expf(-z * z);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erfc_abs = poly * t * ex;               // erfc(|x|/sqrt2)
  cdf = x >= 0.f ? 1.f - 0.5f * erfc_abs : 0.5f * erfc_abs;
  pdf_unnorm = ex;
}
__device__ __forceinline__ float gelu_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return fmaf(x * 0.3989422804014327f, e, cdf);
}

// The same evaluation on a PAIR of values, written on 2-vectors so that the rational / polynomial part compiles to
// v_pk_fma_f32 / v_pk_mul_f32 (two fp32 operations per lane and issue slot); constants folded (1/sqrt2 into p, 1/2 into the
// polynomial, -log2(e)/2 into the exponent: v_exp_f32 is a base-2 exponential).  The GELU epilogues of the token GEMMs are
// VALU-bound, not HBM-bound -- storing the pre-activations as a second output costs nothing, the GELU arithmetic costs 72 us
// of a 391-us fc1 launch (tools/gelu_probe.py) -- so issue slots are what this buys: 33 per pair instead of 2 x 22.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& h, f32x2& ex) {   // h = erfc(|x| / sqrt2) / 2,  ex = exp(-x^2 / 2)
  const f32x2 one = {1.f, 1.f};
  const f32x2 den = __builtin_elementwise_abs(x) * 0.2316418882f + one;
  f32x2 t, arg = x * x * -0.7213475204f;
  t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
  ex.x = __builtin_amdgcn_exp2f(arg.x); ex.y = __builtin_amdgcn_exp2f(arg.y);
  f32x2 poly = t * 0.5307027145f + (-0.7265760135f);
  poly = poly * t + 0.7107068705f;
  poly = poly * t + (-0.142248368f);
  poly = poly * t + 0.127414796f;
  h = poly * t * ex;
}
// Division- and exponential-free evaluation for the GEMM epilogues (v_rcp_f32 / v_exp_f32 issue at quarter rate: 4 of them per pair
// cost as much as 16 packed FMAs).  Phi(x) - 1/2 and GELU'(x) - 1/2 are odd: x * P(t), t = 2 x^2 / X0^2 - 1, one degree-11 polynomial
// each on |x| <= X0 = 4.5 (least-squares fit on Chebyshev nodes, tools/gelu_fit.py), x clamped beyond.  fp32 Horner against float64:
// |Phi| error 2.4e-7 (A&S 7.1.26: 1.5e-7), |GELU| 1.1e-6, |GELU'| 2.9e-6 inside; outside, Phi stays at Phi(+-4.5) = 1 - 3.4e-6 | 3.4e-6
// and GELU' at 1 + 6.9e-5 | -6.9e-5 -- all far below the bf16 rounding (2^-9 relative) of the values these multiply.
// 17 issue slots per pair, no transcendental, against 33 + 4 quarter-rate ones.
#define LT_GELU_X0 4.5f
template <bool GRAD>
__device__ __forceinline__ f32x2 gelu_half_plus_odd2(float a, float b) {     // Phi(x) | GELU'(x) on the clamped pair
  f32x2 xc;
  xc.x = __builtin_amdgcn_fmed3f(a, -LT_GELU_X0, LT_GELU_X0);
  xc.y = __builtin_amdgcn_fmed3f(b, -LT_GELU_X0, LT_GELU_X0);
  const f32x2 t = xc * xc * 0.09876543f + (-1.f);
  f32x2 q;
  if (GRAD) {
    q = t * -5.077220500e-03f + 1.393363997e-02f;
    q = q * t + -1.640383340e-02f;
    q = q * t + 2.396630123e-02f;
    q = q * t + -4.816723987e-02f;
    q = q * t + 7.325470448e-02f;
    q = q * t + -8.904542774e-02f;
    q = q * t + 9.680890292e-02f;
    q = q * t + -9.469939023e-02f;
    q = q * t + 8.710412681e-02f;
    q = q * t + -8.997824788e-02f;
    q = q * t + 1.594295949e-01f;
  } else {
    q = t * -3.126069787e-04f + 9.459542343e-04f;
    q = q * t + -1.462977496e-03f;
    q = q * t + 2.794438740e-03f;
    q = q * t + -6.145955529e-03f;
    q = q * t + 1.137843449e-02f;
    q = q * t + -1.862203889e-02f;
    q = q * t + 2.830292843e-02f;
    q = q * t + -4.018180072e-02f;
    q = q * t + 5.469915643e-02f;
    q = q * t + -7.719016075e-02f;
    q = q * t + 1.569049656e-01f;
  }
  return xc * q + 0.5f;
}
__device__ __forceinline__ void gelu2(float& a, float& b) {
#if defined(LT_GELU_SCALAR)
  a = gelu_f(a); b = gelu_f(b);
#elif defined(LT_GELU_AS)
  const f32x2 x = {a, b};
  f32x2 h, ex;
  gelu_parts2(x, h, ex);
  const f32x2 xh = x * h;
  a = a >= 0.f ? a - xh.x : xh.x;      // x * cdf, cdf = 1 - h | h
  b = b >= 0.f ? b - xh.y : xh.y;
#else
  const f32x2 x = {a, b};
  const f32x2 y = x * gelu_half_plus_odd2<false>(a, b);
  a = y.x; b = y.y;
#endif
}
// (a, b) *= GELU'(pa, pb)
__device__ __forceinline__ void mul_gelu_grad2(float& a, float& b, float pa, float pb) {
#if defined(LT_GELU_SCALAR)
  a *= gelu_grad_f(pa); b *= gelu_grad_f(pb);
#elif defined(LT_GELU_AS)
  const f32x2 x = {pa, pb};
  f32x2 h, ex;
  gelu_parts2(x, h, ex);
  const f32x2 xpdf = x * ex * 0.3989422804014327f;   // x * pdf(x)
  const f32x2 one = {1.f, 1.f};
  const f32x2 up = one - h + xpdf, dn = h + xpdf;     // cdf + x pdf for x >= 0 | x < 0
  a *= pa >= 0.f ? up.x : dn.x;
  b *= pb >= 0.f ? up.y : dn.y;
#else
  const f32x2 d = {a, b};
  const f32x2 y = d * gelu_half_plus_odd2<true>(pa, pb);
  a = y.x; b = y.y;
#endif
}

// A region of `floats` device floats out of a ring of 8 library-owned buffers (reduce.hip): scratch for the ordered second stage of a
// reduction, valid until 7 further calls have been made (kernels of one step never hold more than a few at a time).  nullptr on failure.
float* lt_scratch_ring(size_t floats);

static inline int lt_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
