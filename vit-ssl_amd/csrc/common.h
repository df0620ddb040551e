// Shared device/host helpers for libvitssl_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/vitssl_hip.h"

// ---------------------------------------------------------------- error plumbing
void vitssl_set_error(const char* fmt, ...);
// CUs the persistent one-workgroup-per-CU GEMM grids may occupy: the device's CU count minus VITSSL_RESERVE_CUS
// (default 0).  A collective library's kernels cannot co-reside with those workgroups (they take the whole register
// file and 129 KiB of LDS of their CU), so under data parallelism a few CUs can be left to the all-reduce.
int vitssl_persistent_cus(void);

#define VS_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      vitssl_set_error(__VA_ARGS__);       \
      return VITSSL_ERR_ARG;               \
    }                                      \
  } while (0)

#define VS_CHECK_LAUNCH(what)                                                        \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      vitssl_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e_));    \
      return VITSSL_ERR_LAUNCH;                                                      \
    }                                                                                \
  } while (0)

// ---------------------------------------------------------------- vector types
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef unsigned short bf16_t;  // raw storage

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// plain cast -> v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }

// four fp32 -> four OCP e4m3fn bytes (round to nearest even, saturating at +-448: the
// conversion instruction is given clamped inputs), element 0 in the low byte
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
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

// ---------------------------------------------------------------- dropout stream
// Threefry-2x32-style ARX (add / rotate / xor only: all full-rate VALU), 8 rounds,
// keyed by (seed, site); counter = index of the aligned 4-element group.  One call
// yields 4 x 16-bit uniforms for elements 4*grp .. 4*grp+3.
struct DropKey {
  unsigned k0, k1;
  unsigned thr;   // keep iff u16 >= thr ; thr = round(p * 65536)
  float scale;    // 1 / (1 - p)
};

__host__ __device__ inline DropKey make_drop_key(vitssl_dropout_t d) {
  DropKey k;
  unsigned long long s = d.seed * 0x9E3779B97F4A7C15ull + ((unsigned long long)d.site + 1) * 0xD1B54A32D192ED03ull;
  s ^= s >> 29;
  s *= 0xBF58476D1CE4E5B9ull;
  s ^= s >> 32;
  k.k0 = (unsigned)s;
  k.k1 = (unsigned)(s >> 32);
  float p = d.p;
  if (!(p > 0.f)) p = 0.f;
  unsigned t = (unsigned)(p * 65536.0f + 0.5f);
  if (t > 65535u) t = 65535u;
  k.thr = t;
  k.scale = t ? 65536.0f / (float)(65536u - t) : 1.0f;
  return k;
}

__device__ __forceinline__ unsigned rotl32(unsigned x, int r) { return __builtin_rotateleft32(x, r); }

__device__ __forceinline__ u32x2 drop_bits(const DropKey& k, unsigned grp_lo, unsigned grp_hi) {
  unsigned x0 = grp_lo + k.k0, x1 = grp_hi + k.k1;
  const unsigned k2 = k.k0 ^ k.k1 ^ 0x1BD11BDAu;
#define VS_RND(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  VS_RND(13) VS_RND(15) VS_RND(26) VS_RND(6)
  x0 += k.k1; x1 += k2 + 1u;
  VS_RND(17) VS_RND(29) VS_RND(16) VS_RND(24)
  x0 += k2; x1 += k.k0 + 2u;
#undef VS_RND
  u32x2 r = {x0, x1};
  return r;
}

// keep-multipliers (0 or scale) for the 4 elements of group `grp` (64-bit group index)
__device__ __forceinline__ void drop_mult4(const DropKey& k, unsigned long long grp, float m[4]) {
  u32x2 b = drop_bits(k, (unsigned)grp, (unsigned)(grp >> 32));
  m[0] = ((b[0] & 0xffffu) >= k.thr) ? k.scale : 0.f;
  m[1] = ((b[0] >> 16) >= k.thr) ? k.scale : 0.f;
  m[2] = ((b[1] & 0xffffu) >= k.thr) ? k.scale : 0.f;
  m[3] = ((b[1] >> 16) >= k.thr) ? k.scale : 0.f;
}

// ---------------------------------------------------------------- GELU (exact-erf form)
// erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution):
// the reference evaluates F.gelu (erf form) and stores bf16 under autocast.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __expf(-ax * ax);
  const float r = fmaf(-poly, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
// gelu(x) and gelu'(x) = Phi(x) + x phi(x) together: the exp(-x^2/2) inside the erf
// approximation is exactly the Gaussian factor of phi(x).
__device__ __forceinline__ void gelu_both(float x, float& y, float& dy) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __expf(-ax * ax);                 // = exp(-x^2/2)
  const float erfv = copysignf(fmaf(-poly, e, 1.0f), x);
  const float cdf = 0.5f * (1.0f + erfv);
  y = x * cdf;
  dy = fmaf(x * 0.3989422804014327f, e, cdf);
}
// d/dx gelu(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return fmaf(x, pdf, cdf);
}

// ---------------------------------------------------------------- misc
__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
