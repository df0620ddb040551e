// OCP e4m3fn operand images for the fp8 forward GEMMs (gfx950; BASELINE.json configs[4]).
//
// Activations are quantised at unit scale (LayerNorm / attention / GELU outputs are O(1); +-448 saturates),
// weights per tensor with a power-of-two scale 2^k, k = floor(log2(448 / max|w|)): multiplying by a power
// of two is exact in fp32, so e4m3(w * 2^k) * 2^-k is w rounded to 4 significant bits and nothing else.
// All kernels are HBM streams: 16-byte loads, 4- or 8-byte stores per lane.
#include "common.h"

namespace {

constexpr int Q_THREADS = 256;

// y = e4m3(x * qs); optional running max |x| into *qamax (one atomic per workgroup at most)
__global__ __launch_bounds__(Q_THREADS) void quantize_fp8_kernel(const bf16_t* __restrict__ x, unsigned char* __restrict__ y,
                                                                 long long n, const float* __restrict__ qscale, float* qamax) {
  __shared__ float red[Q_THREADS / 64];
  const float qs = qscale ? *qscale : 1.0f;
  float m = 0.f;
  const long long n8 = n >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const u32x4 v = *(const u32x4*)(x + 8 * i);
    float f[8] = {bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1]), bf_lo(v[2]), bf_hi(v[2]), bf_lo(v[3]), bf_hi(v[3])};
    if (qamax) {
#pragma unroll
      for (int r = 0; r < 8; ++r) m = fmaxf(m, fabsf(f[r]));
    }
    u32x2 o;
    o[0] = pack_fp8x4(f[0] * qs, f[1] * qs, f[2] * qs, f[3] * qs);
    o[1] = pack_fp8x4(f[4] * qs, f[5] * qs, f[6] * qs, f[7] * qs);
    *(u32x2*)(y + 8 * i) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long long i = (n8 << 3) + threadIdx.x;
    const float f = bf2f(x[i]);
    m = fmaxf(m, fabsf(f));
    y[i] = (unsigned char)(pack_fp8x4(f * qs, 0.f, 0.f, 0.f) & 0xffu);
  }
  if (qamax) {               // kernel-argument uniform
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      unsigned* slot = (unsigned*)qamax;
      if (__float_as_uint(m) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(m));
    }
  }
}

__device__ __forceinline__ int find_job(const int* __restrict__ tile_start, int njobs, int b) {
  int lo = 0, hi = njobs;              // tile_start[lo] <= b < tile_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

// Weights: one workgroup per 64x64 tile of one [R, C] tensor (the job table of cast_transpose_batch's kind).
// max |w| per tensor: non-negative floats order like their bit patterns, so the per-tensor reduction is one
// atomicMax on the bits per workgroup
__global__ __launch_bounds__(Q_THREADS) void fp8_weight_amax_kernel(const vitssl_fp8_weight_job_t* __restrict__ jobs,
                                                                    const int* __restrict__ tile_start, int njobs,
                                                                    float* __restrict__ amax) {
  __shared__ float red[Q_THREADS / 64];
  const int j = find_job(tile_start, njobs, blockIdx.x);
  const vitssl_fp8_weight_job_t job = jobs[j];
  // the tensor is contiguous: workgroup t of the job's ceil(R/64) * ceil(C/64) >= R*C / 4096 workgroups scans the
  // linear chunk [4096 t, 4096 (t+1)) (whole cache lines, unlike a 64 x 64 tile of 256-byte row pieces)
  const long long n = (long long)job.R * job.C;
  const long long base = (long long)(blockIdx.x - tile_start[j]) * 4096;
  float m = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long long i = base + 4ll * (it * Q_THREADS + threadIdx.x);
    if (i + 3 < n) {
      const f32x4 v = *(const f32x4*)(job.src + i);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    } else {
      for (long long e = i; e < n; ++e) m = fmaxf(m, fabsf(job.src[e]));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // same-address atomics serialise in L2 (~5 ns each; a 4 M-element weight is 1024 workgroups): skip the atomic when
    // the value already there is not smaller.  The plain read may be stale (smaller): then the atomic is merely redundant.
    unsigned* slot = (unsigned*)amax + j;
    if (__float_as_uint(m) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(m));
  }
}

// k = floor(log2(448 / amax)), clamped so that 2^k and 2^-k are normal floats
__device__ __forceinline__ int fp8_scale_exp(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 0;
  int e;
  const float fr = frexpf(amax, &e);           // amax = fr * 2^e, fr in [0.5, 1)
  // 448 = 0.875 * 2^9: 448 / amax = (0.875 / fr) * 2^(9 - e), and 0.875 / fr is in (0.875, 1.75]
  int k = 9 - e + (fr <= 0.875f ? 0 : -1);
  k = k > 120 ? 120 : (k < -120 ? -120 : k);
  return k;
}

// dst [R, C] and / or dst_t [C, R] = e4m3(src * 2^k): the tile is quantised once, kept as bytes in LDS and written
// in both orientations with 4-byte stores
__global__ __launch_bounds__(Q_THREADS) void fp8_weight_quant_kernel(const vitssl_fp8_weight_job_t* __restrict__ jobs,
                                                                     const int* __restrict__ tile_start, int njobs,
                                                                     const float* __restrict__ amax, float* __restrict__ alpha) {
  __shared__ __attribute__((aligned(4))) unsigned char tile[64][68];
  const int j = find_job(tile_start, njobs, blockIdx.x);
  const vitssl_fp8_weight_job_t job = jobs[j];
  const int k = fp8_scale_exp(amax[j]);
  const float sc = ldexpf(1.0f, k);
  if (blockIdx.x == tile_start[j] && threadIdx.x == 0) alpha[j] = ldexpf(1.0f, -k);
  const int t = blockIdx.x - tile_start[j];
  const int tx_n = (job.C + 63) >> 6;
  const int r0 = (t / tx_n) * 64, c0 = (t % tx_n) * 64;
  const int R = job.R, C = job.C;
  unsigned char* dst = (unsigned char*)job.dst_fp8;
  unsigned char* dst_t = (unsigned char*)job.dst_t_fp8;
  const int tx = (threadIdx.x & 15) * 4, ty = threadIdx.x >> 4;
  const bool vec_c = (C & 3) == 0, vec_r = (R & 3) == 0;
#pragma unroll
  for (int rr = 0; rr < 64; rr += 16) {
    const int r = r0 + rr + ty, c = c0 + tx;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      const float* src = job.src + (long long)r * C + c;
      if (vec_c && c + 3 < C) {
        const f32x4 q = *(const f32x4*)src;
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
      } else {
        for (int e = 0; e < 4 && c + e < C; ++e) v[e] = src[e];
      }
    }
    const unsigned w = pack_fp8x4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
    *(unsigned*)&tile[rr + ty][tx] = w;
    if (dst && r < R) {
      if (vec_c && c + 3 < C) *(unsigned*)(dst + (long long)r * C + c) = w;
      else for (int e = 0; e < 4 && c + e < C; ++e) dst[(long long)r * C + c + e] = (unsigned char)(w >> (8 * e));
    }
  }
  __syncthreads();
  if (dst_t) {
#pragma unroll
    for (int cc = 0; cc < 64; cc += 16) {
      const int c = c0 + cc + ty, r = r0 + tx;          // this thread writes dst_t[c][r .. r+3]
      if (c < C) {
        const unsigned w = (unsigned)tile[tx][cc + ty] | ((unsigned)tile[tx + 1][cc + ty] << 8) |
                           ((unsigned)tile[tx + 2][cc + ty] << 16) | ((unsigned)tile[tx + 3][cc + ty] << 24);
        if (vec_r && r + 3 < R) *(unsigned*)(dst_t + (long long)c * R + r) = w;
        else for (int e = 0; e < 4 && r + e < R; ++e) dst_t[(long long)c * R + r + e] = (unsigned char)(w >> (8 * e));
      }
    }
  }
}

}  // namespace

extern "C" int vitssl_quantize_fp8(const void* x_bf16, void* y_fp8, int64_t n, void* stream) {
  return vitssl_quantize_fp8_scaled(x_bf16, y_fp8, n, nullptr, nullptr, stream);
}

extern "C" int vitssl_quantize_fp8_scaled(const void* x_bf16, void* y_fp8, int64_t n, const float* qscale, float* qamax,
                                          void* stream) {
  VS_CHECK_ARG(x_bf16 && y_fp8 && n > 0, "quantize_fp8: bad args");
  VS_CHECK_ARG(((uintptr_t)x_bf16 & 15) == 0 && ((uintptr_t)y_fp8 & 7) == 0, "quantize_fp8: pointers must be 16- / 8-byte aligned");
  long long blocks = (n / 8 + Q_THREADS - 1) / Q_THREADS;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(quantize_fp8_kernel, dim3((unsigned)blocks), dim3(Q_THREADS), 0, (hipStream_t)stream, (const bf16_t*)x_bf16,
                     (unsigned char*)y_fp8, (long long)n, qscale, qamax);
  VS_CHECK_LAUNCH("quantize_fp8");
  return VITSSL_OK;
}

extern "C" int vitssl_fp8_quantize_weights(const vitssl_fp8_weight_job_t* jobs, const int* tile_start, int njobs, int total_tiles,
                                           float* amax_ws, float* alpha, void* stream) {
  VS_CHECK_ARG(jobs && tile_start && amax_ws && alpha && njobs > 0 && total_tiles > 0, "fp8_quantize_weights: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(amax_ws, 0, sizeof(float) * (size_t)njobs, s);
  if (e != hipSuccess) {
    vitssl_set_error("fp8_quantize_weights: memset failed: %s", hipGetErrorString(e));
    return VITSSL_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(fp8_weight_amax_kernel, dim3(total_tiles), dim3(Q_THREADS), 0, s, jobs, tile_start, njobs, amax_ws);
  VS_CHECK_LAUNCH("fp8_weight_amax");
  hipLaunchKernelGGL(fp8_weight_quant_kernel, dim3(total_tiles), dim3(Q_THREADS), 0, s, jobs, tile_start, njobs, amax_ws, alpha);
  VS_CHECK_LAUNCH("fp8_weight_quant");
  return VITSSL_OK;
}
