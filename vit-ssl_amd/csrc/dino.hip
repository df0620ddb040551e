// DINO head / loss kernels (gfx950, HBM-bound): row L2 normalisation, weight-norm fold,
// softmax / log-softmax cross-entropy over K = 65536 outputs with the algebraic
// reduction  loss = -(1/(G B K)) sum_{b,k} (sum_g t_g)(sum_v s_v)   (the
// reference's DINOLoss includes same-view pairs and divides by K), centre update.
#include "common.h"

namespace {

constexpr int DN_THREADS = 256;

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < nw; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// ---- F.normalize(dim=1, eps=1e-12): wave per row ---------------------------------
__global__ void rownorm_fwd_kernel(const float* __restrict__ z, bf16_t* __restrict__ zn, float* __restrict__ inv_norm,
                                   long long rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* zr = z + row * cols;
  float ss = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 v = *(const f32x4*)(zr + c);
    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  if (lane == 0) inv_norm[row] = inv;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 v = *(const f32x4*)(zr + c) * inv;
    u32x2 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *(u32x2*)(zn + row * cols + c) = w;
  }
}

// dz = inv_norm * (dzn - zn * <zn, dzn>)
__global__ void rownorm_bwd_kernel(const float* __restrict__ dzn, const bf16_t* __restrict__ zn, const float* __restrict__ inv_norm,
                                   bf16_t* __restrict__ dz, long long rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 d = *(const f32x4*)(dzn + row * cols + c);
    const u32x2 w = *(const u32x2*)(zn + row * cols + c);
    dot += d[0] * bf_lo(w[0]) + d[1] * bf_hi(w[0]) + d[2] * bf_lo(w[1]) + d[3] * bf_hi(w[1]);
  }
  dot = wave_sum(dot);
  const float inv = inv_norm[row];
  for (int c = lane * 4; c < cols; c += 256) {
    const f32x4 d = *(const f32x4*)(dzn + row * cols + c);
    const u32x2 w = *(const u32x2*)(zn + row * cols + c);
    const f32x4 n = {bf_lo(w[0]), bf_hi(w[0]), bf_lo(w[1]), bf_hi(w[1])};
    const f32x4 o = (d - n * dot) * inv;
    u32x2 ow = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
    *(u32x2*)(dz + row * cols + c) = ow;
  }
}

// ---- weight_norm(dim=0): W[k,:] = g[k] * v[k,:] / ||v[k,:]||; wave per output row ------
__global__ void weightnorm_fold_kernel(const float* __restrict__ g, const float* __restrict__ v, float* __restrict__ w,
                                       float* __restrict__ inv_vnorm, int K, int D) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (k >= K) return;
  const float* vr = v + (long long)k * D;
  float ss = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 x = *(const f32x4*)(vr + c);
    ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
  }
  ss = wave_sum(ss);
  const float inv = rsqrtf(ss);
  if (lane == 0) inv_vnorm[k] = inv;
  const float s = g[k] * inv;
  for (int c = lane * 4; c < D; c += 256) {
    *(f32x4*)(w + (long long)k * D + c) = *(const f32x4*)(vr + c) * s;
  }
}

// dg[k] += <dW[k,:], vhat> ; dv[k,:] += g*inv*(dW[k,:] - vhat*<dW[k,:], vhat>)
__global__ void weightnorm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ g, const float* __restrict__ v,
                                      const float* __restrict__ inv_vnorm, float* __restrict__ dg, float* __restrict__ dv, int K, int D) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (k >= K) return;
  const float inv = inv_vnorm[k];
  float dot = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 d = *(const f32x4*)(dw + (long long)k * D + c);
    const f32x4 x = *(const f32x4*)(v + (long long)k * D + c);
    dot += d[0] * x[0] + d[1] * x[1] + d[2] * x[2] + d[3] * x[3];
  }
  dot = wave_sum(dot) * inv;   // <dW, vhat>
  if (lane == 0) dg[k] += dot;
  const float s = g[k] * inv;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 d = *(const f32x4*)(dw + (long long)k * D + c);
    const f32x4 x = *(const f32x4*)(v + (long long)k * D + c) * inv;
    f32x4 o = *(const f32x4*)(dv + (long long)k * D + c);
    o += (d - x * dot) * s;
    *(f32x4*)(dv + (long long)k * D + c) = o;
  }
}

// ---- DINO loss -----------------------------------------------------------------------
// phase 1: T[b,:] = sum_g softmax((teacher[g,b,:] - center) / tau_t); one block per b
__global__ void dino_teacher_kernel(const float* __restrict__ teacher, const float* __restrict__ center, float* __restrict__ T,
                                    int G, int B, int K, float inv_tau) {
  __shared__ float red[DN_THREADS / 64];
  const int b = blockIdx.x;
  for (int g = 0; g < G; ++g) {
    const float* tr = teacher + ((long long)g * B + b) * K;
    float m = -INFINITY;
    for (int k = threadIdx.x * 4; k < K; k += DN_THREADS * 4) {
      const f32x4 x = (*(const f32x4*)(tr + k) - *(const f32x4*)(center + k)) * inv_tau;
      m = fmaxf(fmaxf(m, fmaxf(x[0], x[1])), fmaxf(x[2], x[3]));
    }
    m = block_reduce(m, red, true);
    float s = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += DN_THREADS * 4) {
      const f32x4 x = (*(const f32x4*)(tr + k) - *(const f32x4*)(center + k)) * inv_tau;
      s += __expf(x[0] - m) + __expf(x[1] - m) + __expf(x[2] - m) + __expf(x[3] - m);
    }
    s = block_reduce(s, red, false);
    const float inv = 1.0f / s;
    for (int k = threadIdx.x * 4; k < K; k += DN_THREADS * 4) {
      const f32x4 x = (*(const f32x4*)(tr + k) - *(const f32x4*)(center + k)) * inv_tau;
      f32x4 p = {__expf(x[0] - m) * inv, __expf(x[1] - m) * inv, __expf(x[2] - m) * inv, __expf(x[3] - m) * inv};
      if (g > 0) p += *(const f32x4*)(T + (long long)b * K + k);
      *(f32x4*)(T + (long long)b * K + k) = p;
    }
  }
}

// phase 2: one block per student row (v, b)
__global__ void dino_student_kernel(const float* __restrict__ student, const float* __restrict__ T, float* __restrict__ loss_sum,
                                    bf16_t* __restrict__ dstudent, int G, int B, int K, float inv_tau, float gscale) {
  __shared__ float red[DN_THREADS / 64];
  const long long row = blockIdx.x;
  const int b = (int)(row % B);
  const float* sr = student + row * K;
  const float* tb = T + (long long)b * K;
  float m = -INFINITY;
  for (int k = threadIdx.x * 4; k < K; k += DN_THREADS * 4) {
    const f32x4 x = *(const f32x4*)(sr + k) * inv_tau;
    m = fmaxf(fmaxf(m, fmaxf(x[0], x[1])), fmaxf(x[2], x[3]));
  }
  m = block_reduce(m, red, true);
  float s = 0.f;
  for (int k = threadIdx.x * 4; k < K; k += DN_THREADS * 4) {
    const f32x4 x = *(const f32x4*)(sr + k) * inv_tau;
    s += __expf(x[0] - m) + __expf(x[1] - m) + __expf(x[2] - m) + __expf(x[3] - m);
  }
  s = block_reduce(s, red, false);
  const float lse = m + __logf(s);
  const float norm = 1.0f / ((float)G * (float)B * (float)K);
  const float gs = gscale * inv_tau * norm;   // d loss / d logit = gs * (G * p - T)
  float acc = 0.f;
  for (int k = threadIdx.x * 4; k < K; k += DN_THREADS * 4) {
    const f32x4 x = *(const f32x4*)(sr + k) * inv_tau - lse;   // log-softmax
    const f32x4 t = *(const f32x4*)(tb + k);
    acc += t[0] * x[0] + t[1] * x[1] + t[2] * x[2] + t[3] * x[3];
    if (dstudent) {
      float d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) d[r] = gs * ((float)G * __expf(x[r]) - t[r]);
      u32x2 w = {pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])};
      *(u32x2*)(dstudent + row * K + k) = w;
    }
  }
  acc = block_reduce(acc, red, false);
  if (threadIdx.x == 0) atomicAdd(loss_sum, -acc * norm);
}

// Register-resident forms for K = 4096 * NV <= 65536 (the reference's K = 65536: NV = 16): 1024 threads hold one
// whole row (NV float4 per thread), so every logit row is read from HBM ONCE instead of three times (the three-pass
// kernels above re-read 256 KiB rows that 32 workgroups per XCD push out of the 4 MiB L2 in between).
constexpr int DR_THREADS = 1024;

__device__ __forceinline__ float block_reduce16(float v, float* red, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < DR_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

template <int NV>
__global__ __launch_bounds__(DR_THREADS) void dino_teacher_reg_kernel(const float* __restrict__ teacher, const float* __restrict__ center,
                                                                      float* __restrict__ T, int G, int B, int K, float inv_tau) {
  __shared__ float red[DR_THREADS / 64];
  const int b = blockIdx.x;
#pragma unroll 1
  for (int g = 0; g < G; ++g) {
    const float* tr = teacher + ((long long)g * B + b) * K;
    f32x4 x[NV];
    float m = -INFINITY;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int k = 4 * (threadIdx.x + DR_THREADS * v);
      x[v] = (*(const f32x4*)(tr + k) - *(const f32x4*)(center + k)) * inv_tau;
      m = fmaxf(fmaxf(m, fmaxf(x[v][0], x[v][1])), fmaxf(x[v][2], x[v][3]));
      if ((v & 3) == 3) asm volatile("" ::: "memory");   // at most 8 loads in flight: the row itself fills half the registers
    }
    m = block_reduce16(m, red, true);
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int r = 0; r < 4; ++r) x[v][r] = __expf(x[v][r] - m);
      s += (x[v][0] + x[v][1]) + (x[v][2] + x[v][3]);
    }
    s = block_reduce16(s, red, false);
    const float inv = 1.0f / s;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int k = 4 * (threadIdx.x + DR_THREADS * v);
      f32x4 p = x[v] * inv;
      if (g > 0) p += *(const f32x4*)(T + (long long)b * K + k);
      *(f32x4*)(T + (long long)b * K + k) = p;
      if ((v & 3) == 3) asm volatile("" ::: "memory");   // keep at most 4 of the T loads in flight: the row already fills the registers
    }
  }
}

// ---- teacher half on the whole chip ---------------------------------------------------------------------------------
// dino_teacher_reg_kernel runs ONE workgroup per batch element (64 of 256 CUs at the bench's batch, 63 us per step, round-2
// profile).  Two small launches over row SLICES instead: (1) every (teacher row, slice) workgroup reduces its K / DT_SLICES
// columns to (max, sum of exp), (2) every (batch element, slice) workgroup combines the slices' statistics of each global
// view's row, recomputes its columns and writes T = sum_g softmax once.  The logits are read twice (2 x G B K floats,
// L2 / Infinity-Cache resident the second time) for 4 x the workgroups.
constexpr int DT_SLICES = 4;

template <int NVS>     // float4 per thread per slice: K / (4 * 1024 * DT_SLICES)
__global__ __launch_bounds__(DR_THREADS) void dino_teacher_stats_kernel(const float* __restrict__ teacher, const float* __restrict__ center,
                                                                        float* __restrict__ stats, int K, float inv_tau) {
  __shared__ float red[DR_THREADS / 64];
  const int row = blockIdx.x / DT_SLICES, sl = blockIdx.x - row * DT_SLICES;
  const float* tr = teacher + (long long)row * K + (long long)sl * (K / DT_SLICES);
  const float* cr = center + (long long)sl * (K / DT_SLICES);
  f32x4 x[NVS];
  float m = -INFINITY;
#pragma unroll
  for (int v = 0; v < NVS; ++v) {
    const int k = 4 * (threadIdx.x + DR_THREADS * v);
    x[v] = (*(const f32x4*)(tr + k) - *(const f32x4*)(cr + k)) * inv_tau;
    m = fmaxf(fmaxf(m, fmaxf(x[v][0], x[v][1])), fmaxf(x[v][2], x[v][3]));
  }
  m = block_reduce16(m, red, true);
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < NVS; ++v) s += (__expf(x[v][0] - m) + __expf(x[v][1] - m)) + (__expf(x[v][2] - m) + __expf(x[v][3] - m));
  s = block_reduce16(s, red, false);
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x] = m;
    stats[2 * blockIdx.x + 1] = s;
  }
}

template <int NVS>
__global__ __launch_bounds__(DR_THREADS) void dino_teacher_norm_kernel(const float* __restrict__ teacher, const float* __restrict__ center,
                                                                       const float* __restrict__ stats, float* __restrict__ T, int G,
                                                                       int B, int K, float inv_tau) {
  const int b = blockIdx.x / DT_SLICES, sl = blockIdx.x - b * DT_SLICES;
  const long long col0 = (long long)sl * (K / DT_SLICES);
  f32x4 acc[NVS];
#pragma unroll
  for (int v = 0; v < NVS; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int g = 0; g < G; ++g) {
    const float* st = stats + 2 * ((long long)g * B + b) * DT_SLICES;
    float m = st[0];
#pragma unroll
    for (int j = 1; j < DT_SLICES; ++j) m = fmaxf(m, st[2 * j]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < DT_SLICES; ++j) s += st[2 * j + 1] * __expf(st[2 * j] - m);
    const float inv = 1.0f / s;
    const float* tr = teacher + ((long long)g * B + b) * K + col0;
#pragma unroll
    for (int v = 0; v < NVS; ++v) {
      const int k = 4 * (threadIdx.x + DR_THREADS * v);
      const f32x4 x = (*(const f32x4*)(tr + k) - *(const f32x4*)(center + col0 + k)) * inv_tau;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[v][r] += __expf(x[r] - m) * inv;
    }
  }
#pragma unroll
  for (int v = 0; v < NVS; ++v) *(f32x4*)(T + (long long)b * K + col0 + 4 * (threadIdx.x + DR_THREADS * v)) = acc[v];
}

// Round 4: every access goes through a raw buffer instruction with the lane's offset in ONE register and the chunk's offset
// (16 KiB per chunk of 1024 lanes x 16 bytes) in the scalar operand.  With plain pointers hipcc kept 16 64-bit store addresses
// alive (20 spilled VGPRs at the 128-register budget of a 1024-thread workgroup, 84 bytes of scratch per lane) and issued the
// teacher-probability loads of the last pass one at a time, each waited for before its four exponentials; now four are in flight.
template <int NV, bool GRAD>
__global__ __launch_bounds__(DR_THREADS) void dino_student_reg_kernel(const float* __restrict__ student, const float* __restrict__ T,
                                                                      float* __restrict__ loss_sum, bf16_t* __restrict__ dstudent, int G,
                                                                      int B, int K, float inv_tau, float gscale) {
  __shared__ float red[DR_THREADS / 64];
  const long long row = blockIdx.x;
  const int b = (int)(row % B);
  const unsigned row_bytes = (unsigned)K * 4u;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(student + row * K), 0, (int)row_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)(T + (long long)b * K), 0, (int)row_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)threadIdx.x * 16u;
  f32x4 x[NV];
  float m = -INFINITY;
#pragma unroll
  for (int v = 0; v < NV; ++v)
    x[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, (unsigned)(v * DR_THREADS * 16), 0));
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    x[v] *= inv_tau;
    m = fmaxf(fmaxf(m, fmaxf(x[v][0], x[v][1])), fmaxf(x[v][2], x[v][3]));
  }
  m = block_reduce16(m, red, true);
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    s += (__expf(x[v][0] - m) + __expf(x[v][1] - m)) + (__expf(x[v][2] - m) + __expf(x[v][3] - m));
    if (NV >= 16 && (v & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // (64 row registers: no room for 64 exponentials in flight)
  }
  s = block_reduce16(s, red, false);
  const float lse = m + __logf(s);
  const float norm = 1.0f / ((float)G * (float)B * (float)K);
  const float gs = gscale * inv_tau * norm;   // d loss / d logit = gs * (G * p - T)
  float acc = 0.f;
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(GRAD ? dstudent + row * K : (bf16_t*)nullptr), 0,
                                                                 GRAD ? (int)(row_bytes / 2) : 0, 0x00020000);
  constexpr int CH = NV >= 4 ? 4 : NV;        // teacher-probability loads in flight
#pragma unroll
  for (int v0 = 0; v0 < NV; v0 += CH) {
    f32x4 t[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
      t[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, lane_off, (unsigned)((v0 + c) * DR_THREADS * 16), 0));
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const f32x4 ls = x[v0 + c] - lse;       // log-softmax
      acc += t[c][0] * ls[0] + t[c][1] * ls[1] + t[c][2] * ls[2] + t[c][3] * ls[3];
      if constexpr (GRAD) {                   // (a template flag: with a run-time test hipcc postponed the dot products above and spilled their operands)
        float d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = gs * ((float)G * __expf(ls[r]) - t[c][r]);
        const u32x2 w = {pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])};
        __builtin_amdgcn_raw_buffer_store_b64(w, rd, lane_off >> 1, (unsigned)((v0 + c) * DR_THREADS * 8), 0);
      }
    }
    if (NV >= 16) __builtin_amdgcn_sched_barrier(0);
  }
  acc = block_reduce16(acc, red, false);
  if (threadIdx.x == 0) atomicAdd(loss_sum, -acc * norm);
}

// ---- centre --------------------------------------------------------------------------
__global__ void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long long rows, int cols) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= cols) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  long long r = 0;
  for (; r + 8 <= rows; r += 8) {               // eight row loads in flight per thread; added in row order (same sum as a plain loop)
    f32x4 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = *(const f32x4*)(x + (r + u) * cols + c);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += w[u];
  }
  for (; r < rows; ++r) acc += *(const f32x4*)(x + r * cols + c);
  *(f32x4*)(out + c) = acc;
}

__global__ void center_ema_kernel(float* __restrict__ center, const float* __restrict__ colsum, int K, float mom, float inv_rows) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) center[k] = mom * center[k] + (1.f - mom) * colsum[k] * inv_rows;
}

// ------------------------------------------------------------------ positional-table resize
// F.interpolate(mode="bicubic", align_corners=False) of the patch part of the positional
// embedding (vit_core/patch_embedding.py:26-48), channel-last: src [gh0*gw0, D] -> dst [gh*gw, D].
// ATen's upsample_bicubic2d: source coordinate (dst + 0.5) * in/out - 0.5 (not clamped), Keys
// kernel A = -0.75, taps clamped to the border, rows interpolated along x first, then along y.
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__device__ __forceinline__ void cubic_site(int dst, int in_n, int out_n, int idx[4], float c[4]) {
  const float scale = (float)in_n / (float)out_n;
  const float real = scale * ((float)dst + 0.5f) - 0.5f;
  const float fl = floorf(real);
  cubic_coeffs(real - fl, c);
  const int i0 = (int)fl;
#pragma unroll
  for (int k = 0; k < 4; ++k) idx[k] = min(max(i0 - 1 + k, 0), in_n - 1);
}

// grid = gh*gw output tokens, threads over D
__global__ void bicubic_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst, int gh0, int gw0, int gh, int gw, int D) {
  const int y = blockIdx.x / gw, x = blockIdx.x - y * gw;
  int iy[4], ix[4];
  float cy[4], cx[4];
  cubic_site(y, gh0, gh, iy, cy);
  cubic_site(x, gw0, gw, ix, cx);
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float rows[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* r = src + (long long)iy[i] * gw0 * D + d;
      rows[i] = r[(long long)ix[0] * D] * cx[0] + r[(long long)ix[1] * D] * cx[1] + r[(long long)ix[2] * D] * cx[2] + r[(long long)ix[3] * D] * cx[3];
    }
    dst[(long long)blockIdx.x * D + d] = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
  }
}

// dsrc[tap] += wy * wx * ddst (atomics: 16 taps per output token, a few hundred tokens)
__global__ void bicubic_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc, int gh0, int gw0, int gh, int gw, int D) {
  const int y = blockIdx.x / gw, x = blockIdx.x - y * gw;
  int iy[4], ix[4];
  float cy[4], cx[4];
  cubic_site(y, gh0, gh, iy, cy);
  cubic_site(x, gw0, gw, ix, cx);
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float g = ddst[(long long)blockIdx.x * D + d];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(dsrc + ((long long)iy[i] * gw0 + ix[j]) * D + d, g * cy[i] * cx[j]);
  }
}

}  // namespace

extern "C" int vitssl_rownorm_fwd(const float* z, void* zn_bf16, float* inv_norm, int64_t rows, int cols, void* stream) {
  VS_CHECK_ARG(z && zn_bf16 && inv_norm && rows > 0 && cols > 0 && cols % 4 == 0, "rownorm_fwd: bad args");
  hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, (bf16_t*)zn_bf16,
                     inv_norm, (long long)rows, cols);
  VS_CHECK_LAUNCH("rownorm_fwd");
  return VITSSL_OK;
}

extern "C" int vitssl_rownorm_bwd(const float* dzn, const void* zn_bf16, const float* inv_norm, void* dz_bf16, int64_t rows,
                                  int cols, void* stream) {
  VS_CHECK_ARG(dzn && zn_bf16 && inv_norm && dz_bf16 && rows > 0 && cols > 0 && cols % 4 == 0, "rownorm_bwd: bad args");
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dzn,
                     (const bf16_t*)zn_bf16, inv_norm, (bf16_t*)dz_bf16, (long long)rows, cols);
  VS_CHECK_LAUNCH("rownorm_bwd");
  return VITSSL_OK;
}

extern "C" int vitssl_weightnorm_fold(const float* g, const float* v, float* w_f32, float* inv_vnorm, int K, int D, void* stream) {
  VS_CHECK_ARG(g && v && w_f32 && inv_vnorm && K > 0 && D > 0 && D % 4 == 0, "weightnorm_fold: bad args");
  hipLaunchKernelGGL(weightnorm_fold_kernel, dim3((K + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, v, w_f32, inv_vnorm, K, D);
  VS_CHECK_LAUNCH("weightnorm_fold");
  return VITSSL_OK;
}

extern "C" int vitssl_weightnorm_bwd(const float* dw, const float* g, const float* v, const float* inv_vnorm, float* dg, float* dv,
                                     int K, int D, void* stream) {
  VS_CHECK_ARG(dw && g && v && inv_vnorm && dg && dv && K > 0 && D > 0 && D % 4 == 0, "weightnorm_bwd: bad args");
  hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3((K + 3) / 4), dim3(256), 0, (hipStream_t)stream, dw, g, v, inv_vnorm, dg, dv, K, D);
  VS_CHECK_LAUNCH("weightnorm_bwd");
  return VITSSL_OK;
}

extern "C" int64_t vitssl_dino_loss_workspace_floats(int G, int B, int K) {
  if (G <= 0 || B <= 0 || K <= 0) return 0;
  return (int64_t)B * K + VITSSL_DINO_TWS_EXTRA((int64_t)G, (int64_t)B);
}

extern "C" int vitssl_dino_loss(const float* teacher, const float* student, const float* center, float* t_ws, int64_t t_ws_floats,
                                float* loss_sum, void* dstudent_bf16, int G, int V, int B, int K, float teacher_temp,
                                float student_temp, float gscale, void* stream) {
  VS_CHECK_ARG(teacher && student && center && t_ws && loss_sum, "dino_loss: null pointer");
  VS_CHECK_ARG(G > 0 && V > 0 && B > 0 && K > 0 && K % 4 == 0, "dino_loss: K=%d must be a positive multiple of 4", K);
  VS_CHECK_ARG(t_ws_floats >= vitssl_dino_loss_workspace_floats(G, B, K), "dino_loss: scratch of %lld floats, %lld needed (vitssl_dino_loss_workspace_floats)",
               (long long)t_ws_floats, (long long)vitssl_dino_loss_workspace_floats(G, B, K));
  VS_CHECK_ARG(teacher_temp > 0.f && student_temp > 0.f, "dino_loss: temperatures must be positive");
  hipStream_t s = (hipStream_t)stream;
  const int nv = (K % 4096 == 0) ? K / 4096 : 0;
#define VS_DINO_REG(NV)                                                                                                            \
  do {                                                                                                                             \
    if constexpr (NV % DT_SLICES == 0) {                                                                                           \
      float* stats = t_ws + (long long)B * K;     /* [G, B, DT_SLICES, 2] behind T: see the header */                               \
      hipLaunchKernelGGL(dino_teacher_stats_kernel<NV / DT_SLICES>, dim3(G * B * DT_SLICES), dim3(DR_THREADS), 0, s, teacher, center,  \
                         stats, K, 1.0f / teacher_temp);                                                                           \
      VS_CHECK_LAUNCH("dino_teacher_stats");                                                                                       \
      hipLaunchKernelGGL(dino_teacher_norm_kernel<NV / DT_SLICES>, dim3(B * DT_SLICES), dim3(DR_THREADS), 0, s, teacher, center,      \
                         stats, t_ws, G, B, K, 1.0f / teacher_temp);                                                               \
      VS_CHECK_LAUNCH("dino_teacher_norm");                                                                                        \
    } else {                                                                                                                       \
      hipLaunchKernelGGL(dino_teacher_reg_kernel<NV>, dim3(B), dim3(DR_THREADS), 0, s, teacher, center, t_ws, G, B, K, 1.0f / teacher_temp); \
      VS_CHECK_LAUNCH("dino_teacher");                                                                                             \
    }                                                                                                                              \
    if (dstudent_bf16)                                                                                                             \
      hipLaunchKernelGGL((dino_student_reg_kernel<NV, true>), dim3(V * B), dim3(DR_THREADS), 0, s, student, t_ws, loss_sum,           \
                         (bf16_t*)dstudent_bf16, G, B, K, 1.0f / student_temp, gscale);                                            \
    else                                                                                                                           \
      hipLaunchKernelGGL((dino_student_reg_kernel<NV, false>), dim3(V * B), dim3(DR_THREADS), 0, s, student, t_ws, loss_sum,          \
                         (bf16_t*)nullptr, G, B, K, 1.0f / student_temp, gscale);                                                  \
    VS_CHECK_LAUNCH("dino_student");                                                                                               \
    return VITSSL_OK;                                                                                                              \
  } while (0)
  if (nv == 16) VS_DINO_REG(16);
  if (nv == 8) VS_DINO_REG(8);
  if (nv == 4) VS_DINO_REG(4);
  if (nv == 2) VS_DINO_REG(2);
  if (nv == 1) VS_DINO_REG(1);
#undef VS_DINO_REG
  hipLaunchKernelGGL(dino_teacher_kernel, dim3(B), dim3(DN_THREADS), 0, s, teacher, center, t_ws, G, B, K, 1.0f / teacher_temp);
  VS_CHECK_LAUNCH("dino_teacher");
  hipLaunchKernelGGL(dino_student_kernel, dim3(V * B), dim3(DN_THREADS), 0, s, student, t_ws, loss_sum, (bf16_t*)dstudent_bf16, G, B,
                     K, 1.0f / student_temp, gscale);
  VS_CHECK_LAUNCH("dino_student");
  return VITSSL_OK;
}

extern "C" int vitssl_colsum_f32(const float* x, float* out, int64_t rows, int cols, void* stream) {
  VS_CHECK_ARG(x && out && rows > 0 && cols > 0 && cols % 4 == 0, "colsum_f32: bad args");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((cols / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, (long long)rows, cols);
  VS_CHECK_LAUNCH("colsum_f32");
  return VITSSL_OK;
}

extern "C" int vitssl_center_ema(float* center, const float* colsum, int K, float momentum, float inv_rows, void* stream) {
  VS_CHECK_ARG(center && colsum && K > 0, "center_ema: bad args");
  hipLaunchKernelGGL(center_ema_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, center, colsum, K, momentum, inv_rows);
  VS_CHECK_LAUNCH("center_ema");
  return VITSSL_OK;
}

extern "C" int vitssl_bicubic_resize_fwd(const float* src, float* dst, int gh0, int gw0, int gh, int gw, int D, void* stream) {
  VS_CHECK_ARG(src && dst && gh0 > 0 && gw0 > 0 && gh > 0 && gw > 0 && D > 0, "bicubic_resize_fwd: bad args");
  hipLaunchKernelGGL(bicubic_fwd_kernel, dim3(gh * gw), dim3(256), 0, (hipStream_t)stream, src, dst, gh0, gw0, gh, gw, D);
  VS_CHECK_LAUNCH("bicubic_resize_fwd");
  return VITSSL_OK;
}

extern "C" int vitssl_bicubic_resize_bwd(const float* ddst, float* dsrc, int gh0, int gw0, int gh, int gw, int D, void* stream) {
  VS_CHECK_ARG(ddst && dsrc && gh0 > 0 && gw0 > 0 && gh > 0 && gw > 0 && D > 0, "bicubic_resize_bwd: bad args");
  hipLaunchKernelGGL(bicubic_bwd_kernel, dim3(gh * gw), dim3(256), 0, (hipStream_t)stream, ddst, dsrc, gh0, gw0, gh, gw, D);
  VS_CHECK_LAUNCH("bicubic_resize_bwd");
  return VITSSL_OK;
}
