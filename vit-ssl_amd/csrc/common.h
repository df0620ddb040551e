// Shared device/host helpers for libvitssl_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "../../include/vitssl_hip.h"

// ---------------------------------------------------------------- error plumbing
void vitssl_set_error(const char* fmt, ...);
// CUs the persistent one-workgroup-per-CU GEMM grids may occupy: the device's CU count minus VITSSL_RESERVE_CUS
// (default 0).  A collective library's kernels cannot co-reside with those workgroups (they take the whole register
// file and 129 KiB of LDS of their CU), so under data parallelism a few CUs can be left to the all-reduce.
int vitssl_persistent_cus(void);

// ---------------------------------------------------------------- host-side knob caches
// Entry points run on the Python main thread AND on autograd's backward thread (include/vitssl_hip.h promises re-entrancy): the
// per-process caches of environment knobs and of "attribute already set" flags are atomics, never plain statics.  A knob is
// read from the environment once; two threads racing through the first read store the same value.
struct VsEnvInt {
  std::atomic<int> v{INT_MIN};
  int get(const char* name, int dflt) {
    int x = v.load(std::memory_order_relaxed);
    if (x == INT_MIN) {
      const char* e = getenv(name);
      x = e ? (int)strtol(e, nullptr, 0) : dflt;
      if (x == INT_MIN) x = dflt;
      v.store(x, std::memory_order_relaxed);
    }
    return x;
  }
};
struct VsEnvMilli {   // a float knob, cached as thousandths
  std::atomic<int> v{INT_MIN};
  float get(const char* name, float dflt) {
    int x = v.load(std::memory_order_relaxed);
    if (x == INT_MIN) {
      const char* e = getenv(name);
      x = (int)((e ? atof(e) : (double)dflt) * 1000.0 + 0.5);
      v.store(x, std::memory_order_relaxed);
    }
    return (float)x * 1e-3f;
  }
};
typedef std::atomic<bool> VsOnce;   // "hipFuncSetAttribute already done" flags: idempotent work, so relaxed loads / stores suffice

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
// Counter-based, keyed by (seed, site); counter g = index of the aligned 4-element group, (row * cols + col) >> 2, as
// 32 bits (callers check rows * cols < 2^34).  One call yields two words = 4 x 16-bit uniforms for elements 4g .. 4g+3
// (element 0 = low half of a, 1 = high half of a, 2 / 3 = halves of b); an element is KEPT iff its 16 bits, read as a
// SIGNED number, are >= thr - 32768, i.e. with probability 1 - thr / 65536, thr = round(p * 65536).
//
// Round 3: a multiply / xor-shift mixer instead of the 8-round add-rotate-xor network of rounds 1-2.  Measured on MI355X
// (tools/probes/valu_rates.hip, two waves per SIMD): v_mul_lo_u32 issues every 4.5 cycles, exactly like v_alignbit_b32,
// so one multiply round (multiply, shift, xor: 9.6 cycles) costs what one ARX round costs and mixes far better; the old
// network was ~110 cycles per group and, fed with consecutive counters, still showed lag correlations (z ~ 20 over 12 M
// elements with 8 rounds on a 2-D counter).  This one is ~39 cycles per group; tests/test_dropout_stream.py holds its
// statistics (keep rate per row / column, 27 lag correlations, chi-square of the 16-bit values, 2-D spectrum, rectangle
// and within-group products, cross-key correlation) on the NumPy restatement, and the GPU test compares the kernels'
// masks with that restatement bit for bit.
struct DropKey {
  unsigned k0, k1, k2;
  unsigned m1;         // first multiplier of word a: the odd number 0x2c1b3c6d ^ (k1 << 1)
  unsigned thr;        // keep iff u16 (unsigned view) >= thr ... kept for the exported-mask helper and p == 0 tests
  int thr_s;           // thr - 32768: keep iff (signed 16-bit view) >= thr_s
  unsigned tm1_pair;   // (thr_s - 1) & 0xffff in both halves: operand of the packed compare
  float scale;         // 1 / (1 - p_effective)
};

__host__ __device__ inline DropKey make_drop_key(vitssl_dropout_t d) {
  DropKey k;
  unsigned long long s = d.seed * 0x9E3779B97F4A7C15ull + ((unsigned long long)d.site + 1) * 0xD1B54A32D192ED03ull;
  s ^= s >> 29;
  s *= 0xBF58476D1CE4E5B9ull;
  s ^= s >> 32;
  k.k0 = (unsigned)s;
  k.k1 = (unsigned)(s >> 32);
  unsigned long long t2 = s * 0x94D049BB133111EBull;
  t2 ^= t2 >> 31;
  k.k2 = (unsigned)(t2 >> 16);
  k.m1 = (0x2c1b3c6du ^ (k.k1 << 1)) | 1u;
  float p = d.p;
  if (!(p > 0.f)) p = 0.f;
  unsigned t = (unsigned)(p * 65536.0f + 0.5f);
  if (t > 65535u) t = 65535u;
  k.thr = t;
  k.thr_s = (int)t - 32768;
  const unsigned tm1 = (unsigned)(k.thr_s - 1) & 0xffffu;     // (never used when t == 0: dropout is off then)
  k.tm1_pair = tm1 | (tm1 << 16);
  k.scale = t ? 65536.0f / (float)(65536u - t) : 1.0f;
  return k;
}

constexpr unsigned DROP_C0 = 0x9E3779B1u;   // counter multiplier (so that the mixer never sees consecutive integers)

// first word of the stream's state for group g: a0 = g * C0 + k0.  Linear in g: kernels that walk rows / columns of a
// tile form it with one add per group from per-row and per-column terms (drop_a0_row / drop_a0_col).
__device__ __forceinline__ unsigned drop_a0(const DropKey& k, unsigned g) { return g * DROP_C0 + k.k0; }

// Round 4: the key enters word a as well (its first multiplier is key-dependent and k2 is xored in between the two multiplies;
// same instruction count: the multiplier is a scalar operand either way, the xor fuses into v_xor3_b32).  Before, word a was a
// fixed bijection of g * C0 + k0, so the a-streams of any two (seed, site) keys were ONE 2^32-long sequence read at different
// offsets (advisor finding; tests/test_dropout_stream.py::test_streams_of_different_keys_are_not_shifted_copies).
__device__ __forceinline__ u32x2 drop_words_a0(const DropKey& k, unsigned a) {
  a ^= a >> 15;
  a *= k.m1;
  a = a ^ (a >> 12) ^ k.k2;
  a *= 0x297a2d39u;
  a ^= a >> 15;
  unsigned b = (a ^ k.k1) * 0xc2b2ae35u;
  b ^= b >> 15;
  u32x2 r = {a, b};
  return r;
}
__device__ __forceinline__ u32x2 drop_words(const DropKey& k, unsigned g) { return drop_words_a0(k, drop_a0(k, g)); }

// 0xffff in every 16-bit half of w whose element is kept (two instructions: saturating packed subtract, packed
// arithmetic shift): AND it onto a packed bf16 pair.
__device__ __forceinline__ unsigned drop_keep_pair(const DropKey& k, unsigned w) {
  unsigned d;
  asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(k.tm1_pair), "v"(w));   // sat(thr_s - 1 - u) < 0  <=>  u >= thr_s
  asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(d));   // (op_sel_hi: the inline constant's low half shifts BOTH halves)
  return d;
}
// keep decisions of the 4 elements as booleans (fp32 consumers: select, then one fma with the scale)
__device__ __forceinline__ void drop_keep4(const DropKey& k, const u32x2& w, bool keep[4]) {
  const int t = (int)((unsigned)k.thr_s << 16);
  keep[0] = (int)(w[0] << 16) >= t;
  keep[1] = (int)w[0] >= t;
  keep[2] = (int)(w[1] << 16) >= t;
  keep[3] = (int)w[1] >= t;
}
// keep-multipliers (0 or scale) for the 4 elements of group g
__device__ __forceinline__ void drop_mult4(const DropKey& k, unsigned g, float m[4]) {
  bool keep[4];
  drop_keep4(k, drop_words(k, g), keep);
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = keep[r] ? k.scale : 0.f;
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
// the same with a common factor s folded into the constants: y = s gelu(x), dy = s gelu'(x); hs = 0.5 s, cs = s / sqrt(2 pi)
// (the GEMM epilogue's dropout scale: no separate multiplies)
__device__ __forceinline__ void gelu_both_scaled(float x, float hs, float cs, float& y, float& dy) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __expf(-ax * ax);                 // = exp(-x^2/2)
  const float erfv = copysignf(fmaf(-poly, e, 1.0f), x);
  const float cdf = fmaf(hs, erfv, hs);             // s Phi(x)
  y = x * cdf;
  dy = fmaf(x * cs, e, cdf);
}
// d/dx gelu(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return fmaf(x, pdf, cdf);
}

// ---------------------------------------------------------------- misc
__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
