// LayerNorm forward / backward for the fp32 residual stream (gfx950, HBM-bound).
//
// One 64-lane wave per row; the row lives in registers as V float4 per lane
// (cols <= 256*V), statistics are two-pass in fp32 via wavefront shuffles, the
// output is emitted as bf16 (the operand of the next MFMA GEMM).
//
// Backward fuses, per row: LN input-gradient, + residual-branch gradient, fp32 store
// of the new residual gradient, and the dropout-masked bf16 copy the next
// (reverse-order) GEMM consumes; per column: dgamma, dbeta and the bias-gradient
// column sum are accumulated in registers across the rows a wave visits, reduced
// across the block's 4 waves in LDS and added to global with one atomic per column
// per block.
#include "common.h"

namespace {

constexpr int LN_THREADS = 256;  // 4 waves = 4 rows in flight per block

// 1: the residual-branch gradient of a row is requested together with dy and x (one memory round trip per row
// instead of two: the second used to start only after the two wave reductions)
#ifndef LN_HOIST_GRES
#define LN_HOIST_GRES 1
#endif
// 1: backward requests the NEXT row's dy / x / residual gradient before it reduces the current row (the loads fly
// under the two wave reductions and the stores); costs 7.5 V registers per lane
#ifndef LN_PREFETCH
#define LN_PREFETCH 1
#endif
#ifndef LN_FWD_PREFETCH
#define LN_FWD_PREFETCH 1
#endif
// 1: forward stores 16 bytes per lane (adjacent lanes trade their v = 0 / v = 1 pieces with a DPP quad
// swap, so an even lane writes 8 consecutive columns of the first 256, an odd lane of the second 256)
#ifndef LN_FWD_PAIR
#define LN_FWD_PAIR 0
#endif
// 1: the backward processes rows of 384 columns two per wave (ln_bwd_kernel's PAIR form); 0: one row per wave (A/B builds)
#ifndef LN_ROW_PAIRS
#define LN_ROW_PAIRS 1
#endif

// Q8: also emit the e4m3 image of the output (fp8 operand path, unit scale)
template <int V, bool Q8 = false>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long long rows, int cols, float eps, unsigned char* __restrict__ y8 = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c4n = cols >> 2;
  f32x4 g[V], b[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c4 = lane + 64 * v;
    if (c4 < c4n) {
      g[v] = *(const f32x4*)(gamma + 4 * c4);
      b[v] = *(const f32x4*)(beta + 4 * c4);
    }
  }
  const float inv = 1.0f / (float)cols;
  const long long row_step = (long long)gridDim.x * 4;
  f32x4 nxt[V];
  auto fetch = [&](long long row) {
    if (!LN_FWD_PREFETCH || row >= rows) return;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c4 = lane + 64 * v;
      if (c4 < c4n) nxt[v] = *(const f32x4*)(x + row * cols + 4 * c4);
    }
  };
  fetch((long long)blockIdx.x * 4 + wave);
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += row_step) {
    const float* xr = x + row * cols;
    f32x4 xv[V];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c4 = lane + 64 * v;
      if (c4 < c4n) {
        xv[v] = LN_FWD_PREFETCH ? nxt[v] : *(const f32x4*)(xr + 4 * c4);
        s += xv[v][0] + xv[v][1] + xv[v][2] + xv[v][3];
      }
    }
    fetch(row + row_step);                        // the next row's load flies under the reductions and the stores
    const float mu = wave_sum(s) * inv;
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c4 = lane + 64 * v;
      if (c4 < c4n) {
        xv[v] -= mu;
        ss += xv[v][0] * xv[v][0] + xv[v][1] * xv[v][1] + xv[v][2] * xv[v][2] + xv[v][3] * xv[v][3];
      }
    }
    const float rs = rsqrtf(wave_sum(ss) * inv + eps);
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    bf16_t* yr = y + row * cols;
    if (LN_FWD_PAIR && !Q8 && V >= 2 && c4n == 64 * V) {
      u32x2 w[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const f32x4 o = xv[v] * rs * g[v] + b[v];
        w[v] = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
      }
      const bool odd = lane & 1;
#pragma unroll
      for (int v = 0; v + 1 < V; v += 2) {
        const u32x2 t = odd ? w[v] : w[v + 1];                        // what the partner lane needs
        const u32x2 r = {(unsigned)__builtin_amdgcn_update_dpp(0, (int)t[0], 0xB1, 0xF, 0xF, false),
                         (unsigned)__builtin_amdgcn_update_dpp(0, (int)t[1], 0xB1, 0xF, 0xF, false)};   // quad_perm [1,0,3,2]
        const u32x4 q = odd ? u32x4{r[0], r[1], w[v + 1][0], w[v + 1][1]} : u32x4{w[v][0], w[v][1], r[0], r[1]};
        *(u32x4*)(yr + 256 * (v + (odd ? 1 : 0)) + 4 * (lane - (odd ? 1 : 0))) = q;
      }
      if (V & 1) *(u32x2*)(yr + 4 * (lane + 64 * (V - 1))) = w[V - 1];
      continue;
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c4 = lane + 64 * v;
      if (c4 < c4n) {
        const f32x4 o = xv[v] * rs * g[v] + b[v];
        if (!Q8 || y) {                         // fp8 path: the bf16 image is optional (kernel-argument uniform)
          u32x2 w = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
          *(u32x2*)(yr + 4 * c4) = w;
        }
        if constexpr (Q8) *(unsigned*)(y8 + row * cols + 4 * c4) = pack_fp8x4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// workgroups per CU the register budget is sized for.  The prefetching form holds two rows of operands; where that
// spills, fewer resident waves win (MI355X, interleaved A/B, M = 50176, D = 768: with column sums + dropout
// 134 us without prefetch, 154 at 3 workgroups (spills), 125 at 2; without column sums 132 -> 118 at 3)
constexpr int ln_bwd_blocks(int V, bool has_ln, bool has_cs) {
  if (!(LN_PREFETCH && has_ln)) return (V <= 2 || (V == 3 && !has_cs)) ? 4 : (V <= 4 ? 3 : 1);
  return V <= 1 ? 4 : (V == 2 ? (has_cs ? 3 : 4) : (V == 3 ? (has_cs ? 2 : 3) : (V == 4 ? 2 : 1)));
}

// HAS_LN = false turns the kernel into the plain "mask + cast + column-sum" of g_res.
// Q8: also write the e4m3 image gm8 = e4m3(gm * *qscale) and record max |gm| in *qamax (fp8 dgrad operand).
// PAIR (rows of 384 columns, ViT-S): a wave takes two consecutive rows = one contiguous run of 192 float4 = three full
// 64-lane vectors, the access shape of a 768-column row.  One row per wave leaves half the lanes of the second vector idle
// (96 float4 = 64 + 32) and half as many bytes per wave in flight: 57.8 us against 52.2 for M = 50176 (5.3 -> 5.9 TB/s; 33.8 ->
// 28.9 us at M = 25088; interleaved A/B, MI355X).  The caller passes rows = row pairs and cols = 2 x the row length; a slot
// (lane, v) belongs to the pair's second row where lane + 64 v >= cols / 8, and the row statistics are reduced per row.  (The
// same form of the FORWARD kernel was 17 % slower than one row per wave, 24.5 against 21.0 us, and is not built.)
template <int V, bool HAS_LN, bool HAS_CS, bool Q8 = false, bool PAIR = false>
__global__ __launch_bounds__(LN_THREADS, ln_bwd_blocks(V, HAS_LN, HAS_CS)) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* g_res,
                                                            float* g_out, bf16_t* __restrict__ gm, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ gm_colsum,
                                                            DropKey dk, int drop_on, long long rows, int cols,
                                                            unsigned char* __restrict__ gm8 = nullptr,
                                                            const float* __restrict__ qscale = nullptr, float* qamax = nullptr) {
  __shared__ __attribute__((aligned(16))) float red[4][V * 256];   // [wave][col]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c4n = cols >> 2;
  const int c4h = PAIR ? c4n >> 1 : c4n;            // float4 per row
  static_assert(!PAIR || (HAS_LN && LN_PREFETCH && !Q8), "row pairs are built for the prefetching LayerNorm form, bf16 image");
  f32x4 g[V], acc_dg[V], acc_db[V], acc_cs[V];
  bool hi[V];                                       // PAIR: the slot holds columns of the pair's second row
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c4 = lane + 64 * v;
    hi[v] = PAIR && c4 >= c4h;
    acc_dg[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc_db[v] = acc_dg[v];
    acc_cs[v] = acc_dg[v];
    if (HAS_LN && c4 < c4n) g[v] = *(const f32x4*)(gamma + 4 * (c4 - (hi[v] ? c4h : 0)));
  }
  const float inv = 1.0f / (float)(PAIR ? cols >> 1 : cols);
  const float qs = (Q8 && qscale) ? *qscale : 1.0f;
  float qmax = 0.f;
  // raw operands of one row (HAS_LN): x, residual gradient, packed dy, statistics
  f32x4 nx[V], ngr[V];
  u32x2 ndy[V];
  float nmu = 0.f, nrs = 0.f, nmuB = 0.f, nrsB = 0.f;
  auto fetch = [&](long long row) {
    if (!(HAS_LN && LN_PREFETCH) || row >= rows) return;
    if constexpr (PAIR) {
      nmu = mean[2 * row];
      nrs = rstd[2 * row];
      nmuB = mean[2 * row + 1];
      nrsB = rstd[2 * row + 1];
    } else {
      nmu = mean[row];
      nrs = rstd[row];
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c4 = lane + 64 * v;
      if (c4 < c4n) {
        if (g_res) ngr[v] = *(const f32x4*)(g_res + row * cols + 4 * c4);
        ndy[v] = *(const u32x2*)(dy + row * cols + 4 * c4);
        nx[v] = *(const f32x4*)(x + row * cols + 4 * c4);
      }
    }
  };
  const long long row_step = (long long)gridDim.x * 4;
  fetch((long long)blockIdx.x * 4 + wave);
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += row_step) {
    f32x4 dx[V];
    if constexpr (HAS_LN && LN_PREFETCH) {
      const float mu = nmu, rs = nrs, muB = nmuB, rsB = nrsB;
      f32x4 xh[V], gr[V];
      u32x2 dyp[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        xh[v] = nx[v];
        gr[v] = ngr[v];
        dyp[v] = ndy[v];
      }
      fetch(row + row_step);                      // in flight during everything below
      float s1 = 0.f, s2 = 0.f, s1B = 0.f, s2B = 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c4 = lane + 64 * v;
        if (c4 < c4n) {
          const f32x4 d = {bf_lo(dyp[v][0]), bf_hi(dyp[v][0]), bf_lo(dyp[v][1]), bf_hi(dyp[v][1])};
          xh[v] = (xh[v] - (hi[v] ? muB : mu)) * (hi[v] ? rsB : rs);
          const f32x4 dyg = d * g[v];
          acc_dg[v] += d * xh[v];
          acc_db[v] += d;
          const float t1 = dyg[0] + dyg[1] + dyg[2] + dyg[3];
          const f32x4 t = dyg * xh[v];
          const float t2 = t[0] + t[1] + t[2] + t[3];
          if (hi[v]) {
            s1B += t1;
            s2B += t2;
          } else {
            s1 += t1;
            s2 += t2;
          }
        }
      }
      const float m1 = wave_sum(s1) * inv;
      const float m2 = wave_sum(s2) * inv;
      const float m1B = PAIR ? wave_sum(s1B) * inv : 0.f;
      const float m2B = PAIR ? wave_sum(s2B) * inv : 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c4 = lane + 64 * v;
        if (c4 < c4n) {
          const f32x4 d = {bf_lo(dyp[v][0]), bf_hi(dyp[v][0]), bf_lo(dyp[v][1]), bf_hi(dyp[v][1])};
          dx[v] = (d * g[v] - (hi[v] ? m1B : m1) - xh[v] * (hi[v] ? m2B : m2)) * (hi[v] ? rsB : rs);
          if (g_res) dx[v] += gr[v];
          *(f32x4*)(g_out + row * cols + 4 * c4) = dx[v];
        }
      }
    } else if constexpr (HAS_LN) {
      const float mu = mean[row], rs = rstd[row];
      f32x4 xh[V];
      f32x4 gr[V];
      u32x2 dyp[V];   // dy kept packed (bf16) between the two passes: fewer live registers
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c4 = lane + 64 * v;
        if (c4 < c4n) {
          if (LN_HOIST_GRES && g_res) gr[v] = *(const f32x4*)(g_res + row * cols + 4 * c4);
          dyp[v] = *(const u32x2*)(dy + row * cols + 4 * c4);
          const f32x4 d = {bf_lo(dyp[v][0]), bf_hi(dyp[v][0]), bf_lo(dyp[v][1]), bf_hi(dyp[v][1])};
          xh[v] = (*(const f32x4*)(x + row * cols + 4 * c4) - mu) * rs;
          const f32x4 dyg = d * g[v];
          acc_dg[v] += d * xh[v];
          acc_db[v] += d;
          s1 += dyg[0] + dyg[1] + dyg[2] + dyg[3];
          const f32x4 t = dyg * xh[v];
          s2 += t[0] + t[1] + t[2] + t[3];
        }
      }
      const float m1 = wave_sum(s1) * inv;
      const float m2 = wave_sum(s2) * inv;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c4 = lane + 64 * v;
        if (c4 < c4n) {
          const f32x4 d = {bf_lo(dyp[v][0]), bf_hi(dyp[v][0]), bf_lo(dyp[v][1]), bf_hi(dyp[v][1])};
          dx[v] = (d * g[v] - m1 - xh[v] * m2) * rs;
          if (g_res) dx[v] += LN_HOIST_GRES ? gr[v] : *(const f32x4*)(g_res + row * cols + 4 * c4);
          *(f32x4*)(g_out + row * cols + 4 * c4) = dx[v];
        }
      }
    } else {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c4 = lane + 64 * v;
        if (c4 < c4n) dx[v] = *(const f32x4*)(g_res + row * cols + 4 * c4);
      }
    }
    if (gm || (Q8 && gm8)) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c4 = lane + 64 * v;
        if (c4 < c4n) {
          f32x4 o = dx[v];
          if (drop_on) {
            bool keep[4];
            drop_keep4(dk, drop_words(dk, (unsigned)row * (unsigned)c4n + (unsigned)c4), keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = keep[r] ? o[r] * dk.scale : 0.f;
          }
          if (gm) {                              // fp8 path: the bf16 image is optional
            u32x2 w = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            *(u32x2*)(gm + row * cols + 4 * c4) = w;
          }
          if constexpr (Q8) {
            *(unsigned*)(gm8 + row * cols + 4 * c4) = pack_fp8x4(o[0] * qs, o[1] * qs, o[2] * qs, o[3] * qs);
            qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
          }
          if (HAS_CS) acc_cs[v] += o;
        }
      }
    }
  }
  if constexpr (Q8) {
    if (qamax) {
      qmax = wave_max(qmax);
      unsigned* slot = (unsigned*)qamax;
      if (lane == 0 && __float_as_uint(qmax) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(qmax));
    }
  }
  // cross-wave column reduction, one atomic per column per block (LDS buffer reused)
#pragma unroll
  for (int qty = 0; qty < 3; ++qty) {
    if (qty < 2 && !HAS_LN) continue;
    if (qty == 2 && !HAS_CS) continue;
    float* target = qty == 0 ? dgamma : (qty == 1 ? dbeta : gm_colsum);
    if (!target) continue;   // kernel-argument uniform
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c4 = lane + 64 * v;
      if (c4 < c4n) {
        const f32x4 a = qty == 0 ? acc_dg[v] : (qty == 1 ? acc_db[v] : acc_cs[v]);
        *(f32x4*)&red[wave][4 * c4] = a;
      }
    }
    __syncthreads();
    if constexpr (PAIR) {                            // a column's two slots: columns c and c + cols / 2 of the pair
      const int ch = cols >> 1;
      for (int c = threadIdx.x; c < ch; c += LN_THREADS)
        atomicAdd(target + c, (red[0][c] + red[1][c] + red[2][c] + red[3][c]) +
                                  (red[0][c + ch] + red[1][c + ch] + red[2][c + ch] + red[3][c + ch]));
    } else {
      for (int c = threadIdx.x; c < cols; c += LN_THREADS)
        atomicAdd(target + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    }
    __syncthreads();
  }
}

// Blocks of 4 rows, grid-stride.  Backward: every block ends with one atomic per column per accumulated vector, and
// the register budget admits 2-4 blocks per CU, so a grid larger than what is resident at once only adds such tails:
// one block per CU (two for rows of <= 512 columns) is fastest now that the next row is prefetched (MI355X, interleaved
// A/B against the former cap of 2048: M = 50176, D = 768: 123 -> 110 us, with column sums 126 -> 111; D = 1024,
// M = 25088: 89 -> 75; D = 384: 79 -> 58; DINO local crops, M = 18944, D = 768: 65 -> 37).  Forward has no such tail:
// 4096 blocks give 43.5-43.6 ms per ViT-B step against 43.8 with 2048 (uncapped: 43.7).
#ifndef LN_BWD_GRID_CAP
#define LN_BWD_GRID_CAP 0     // 0 = the rule above; a positive value forces the cap (developer A/B builds)
#endif
inline int ln_grid(long long rows, bool fwd = false, int cols = 1024) {
  long long g = (rows + 3) / 4;
#ifndef LN_FWD_GRID_CAP
#define LN_FWD_GRID_CAP 4096
#endif
  long long cap = fwd ? LN_FWD_GRID_CAP : (long long)vitssl_persistent_cus() * (cols <= 512 ? 2 : 1);
  if (!fwd && LN_BWD_GRID_CAP > 0) cap = LN_BWD_GRID_CAP;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

template <bool HAS_LN, bool Q8 = false>
int launch_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                  const float* g_res, float* g_out, void* gm, float* dgamma, float* dbeta, float* gm_colsum,
                  vitssl_dropout_t drop, int64_t rows, int cols, hipStream_t s, void* gm8 = nullptr,
                  const float* qscale = nullptr, float* qamax = nullptr) {
  DropKey dk = make_drop_key(drop);
  const int on = dk.thr != 0;
  VS_CHECK_ARG(!on || (unsigned long long)rows * (unsigned long long)cols < (1ull << 34),
               "layernorm_bwd / grad_mask_cast: the dropout stream's group counter is 32 bits (rows * cols < 2^34)");
  if constexpr (HAS_LN && !Q8 && LN_PREFETCH && LN_ROW_PAIRS) {
    if (cols == 384 && (rows & 1) == 0) {            // two rows per wave (PAIR)
      const int pgrid = ln_grid(rows / 2, false, 768);
      if (gm_colsum)
        hipLaunchKernelGGL((ln_bwd_kernel<3, true, true, false, true>), dim3(pgrid), dim3(LN_THREADS), 0, s, (const bf16_t*)dy, x, mean,
                           rstd, gamma, g_res, g_out, (bf16_t*)gm, dgamma, dbeta, gm_colsum, dk, on, (long long)(rows / 2), 768,
                           (unsigned char*)nullptr, (const float*)nullptr, (float*)nullptr);
      else
        hipLaunchKernelGGL((ln_bwd_kernel<3, true, false, false, true>), dim3(pgrid), dim3(LN_THREADS), 0, s, (const bf16_t*)dy, x, mean,
                           rstd, gamma, g_res, g_out, (bf16_t*)gm, dgamma, dbeta, gm_colsum, dk, on, (long long)(rows / 2), 768,
                           (unsigned char*)nullptr, (const float*)nullptr, (float*)nullptr);
      VS_CHECK_LAUNCH("layernorm_bwd");
      return VITSSL_OK;
    }
  }
  const int grid = ln_grid(rows, false, cols);
#define VS_LNB(V)                                                                                                        \
  do {                                                                                                                   \
    if (gm_colsum)                                                                                                       \
      hipLaunchKernelGGL((ln_bwd_kernel<V, HAS_LN, true, Q8>), dim3(grid), dim3(LN_THREADS), 0, s, (const bf16_t*)dy, x, mean, \
                         rstd, gamma, g_res, g_out, (bf16_t*)gm, dgamma, dbeta, gm_colsum, dk, on, (long long)rows, cols,  \
                         (unsigned char*)gm8, qscale, qamax);                                                            \
    else                                                                                                                 \
      hipLaunchKernelGGL((ln_bwd_kernel<V, HAS_LN, false, Q8>), dim3(grid), dim3(LN_THREADS), 0, s, (const bf16_t*)dy, x, mean, \
                         rstd, gamma, g_res, g_out, (bf16_t*)gm, dgamma, dbeta, gm_colsum, dk, on, (long long)rows, cols,  \
                         (unsigned char*)gm8, qscale, qamax);                                                            \
  } while (0)
  if (cols <= 256) VS_LNB(1);
  else if (cols <= 512) VS_LNB(2);
  else if (cols <= 768) VS_LNB(3);
  else if (cols <= 1024) VS_LNB(4);
  else VS_LNB(8);
#undef VS_LNB
  VS_CHECK_LAUNCH("layernorm_bwd");
  return VITSSL_OK;
}

}  // namespace

extern "C" int vitssl_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                                    float* rstd, int64_t rows, int cols, float eps, void* stream) {
  VS_CHECK_ARG(x && gamma && beta && y_bf16 && mean && rstd, "layernorm_fwd: null pointer");
  VS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "layernorm_fwd: cols=%d must be a multiple of 4 and <= 2048", cols);
  hipStream_t s = (hipStream_t)stream;
  const int grid = ln_grid(rows, true);
#define VS_LNF(V)                                                                                                   \
  hipLaunchKernelGGL((ln_fwd_kernel<V, false>), dim3(grid), dim3(LN_THREADS), 0, s, x, gamma, beta, (bf16_t*)y_bf16, mean, rstd, \
                     (long long)rows, cols, eps, (unsigned char*)nullptr)
  if (cols <= 256) VS_LNF(1);
  else if (cols <= 512) VS_LNF(2);
  else if (cols <= 768) VS_LNF(3);
  else if (cols <= 1024) VS_LNF(4);
  else VS_LNF(8);
#undef VS_LNF
  VS_CHECK_LAUNCH("layernorm_fwd");
  return VITSSL_OK;
}

extern "C" int vitssl_layernorm_fwd_fp8(const float* x, const float* gamma, const float* beta, void* y_bf16, void* y_fp8,
                                        float* mean, float* rstd, int64_t rows, int cols, float eps, void* stream) {
  VS_CHECK_ARG(x && gamma && beta && y_fp8 && mean && rstd, "layernorm_fwd_fp8: null pointer");   // y_bf16 may be NULL
  VS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "layernorm_fwd_fp8: cols=%d must be a multiple of 4 and <= 2048", cols);
  const int grid = ln_grid(rows, true);
  hipStream_t s = (hipStream_t)stream;
#define VS_LNF(V)                                                                                                         \
  hipLaunchKernelGGL((ln_fwd_kernel<V, true>), dim3(grid), dim3(LN_THREADS), 0, s, x, gamma, beta, (bf16_t*)y_bf16, mean, rstd, \
                     (long long)rows, cols, eps, (unsigned char*)y_fp8)
  if (cols <= 256) VS_LNF(1);
  else if (cols <= 512) VS_LNF(2);
  else if (cols <= 768) VS_LNF(3);
  else if (cols <= 1024) VS_LNF(4);
  else VS_LNF(8);
#undef VS_LNF
  VS_CHECK_LAUNCH("layernorm_fwd_fp8");
  return VITSSL_OK;
}

extern "C" int vitssl_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                    const float* gamma, const float* g_res, float* g_out, void* gm_bf16, float* dgamma,
                                    float* dbeta, float* gm_colsum, vitssl_dropout_t drop, int64_t rows, int cols,
                                    void* stream) {
  VS_CHECK_ARG(dy_bf16 && x && mean && rstd && gamma && g_out && dgamma && dbeta, "layernorm_bwd: null pointer");
  VS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "layernorm_bwd: cols=%d must be a multiple of 4 and <= 2048", cols);
  VS_CHECK_ARG(!gm_colsum || gm_bf16, "layernorm_bwd: gm_colsum without gm_bf16");
  return launch_ln_bwd<true>(dy_bf16, x, mean, rstd, gamma, g_res, g_out, gm_bf16, dgamma, dbeta, gm_colsum, drop, rows,
                             cols, (hipStream_t)stream);
}

extern "C" int vitssl_layernorm_bwd_fp8(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                                        const float* gamma, const float* g_res, float* g_out, void* gm_bf16, void* gm_fp8,
                                        const float* qscale, float* qamax, float* dgamma, float* dbeta, float* gm_colsum,
                                        vitssl_dropout_t drop, int64_t rows, int cols, void* stream) {
  VS_CHECK_ARG(dy_bf16 && x && mean && rstd && gamma && g_out && dgamma && dbeta && gm_fp8, "layernorm_bwd_fp8: null pointer");   // gm_bf16 may be NULL
  VS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "layernorm_bwd_fp8: cols=%d must be a multiple of 4 and <= 2048", cols);
  return launch_ln_bwd<true, true>(dy_bf16, x, mean, rstd, gamma, g_res, g_out, gm_bf16, dgamma, dbeta, gm_colsum, drop, rows,
                                   cols, (hipStream_t)stream, gm_fp8, qscale, qamax);
}

extern "C" int vitssl_grad_mask_cast_fp8(const float* g, void* gm_bf16, void* gm_fp8, const float* qscale, float* qamax,
                                         float* gm_colsum, vitssl_dropout_t drop, int64_t rows, int cols, void* stream) {
  VS_CHECK_ARG(g && gm_fp8, "grad_mask_cast_fp8: null pointer");   // gm_bf16 may be NULL
  VS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "grad_mask_cast_fp8: cols=%d must be a multiple of 4 and <= 2048", cols);
  return launch_ln_bwd<false, true>(nullptr, nullptr, nullptr, nullptr, nullptr, g, nullptr, gm_bf16, nullptr, nullptr,
                                    gm_colsum, drop, rows, cols, (hipStream_t)stream, gm_fp8, qscale, qamax);
}

extern "C" int vitssl_grad_mask_cast(const float* g, void* gm_bf16, float* gm_colsum, vitssl_dropout_t drop,
                                     int64_t rows, int cols, void* stream) {
  VS_CHECK_ARG(g && gm_bf16, "grad_mask_cast: null pointer");
  VS_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 2048, "grad_mask_cast: cols=%d must be a multiple of 4 and <= 2048", cols);
  return launch_ln_bwd<false>(nullptr, nullptr, nullptr, nullptr, nullptr, g, nullptr, gm_bf16, nullptr, nullptr,
                              gm_colsum, drop, rows, cols, (hipStream_t)stream);
}
