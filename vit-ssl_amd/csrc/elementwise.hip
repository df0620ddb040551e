// HBM-bound glue kernels of the hot path (gfx950): patch gather, row gather/scatter,
// embedding backward, losses, parameter casts, AdamW, EMA.  All accesses are 8/16 bytes
// per lane and coalesced along the contiguous axis.
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;

inline unsigned ew_grid(long long work_items) {
  long long g = (work_items + EW_THREADS - 1) / EW_THREADS;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ------------------------------------------------------------------ dropout mask export
__global__ void dropout_mask_kernel(unsigned char* keep, long long ngroups, DropKey dk) {
  const long long grp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (grp >= ngroups) return;
  float m[4];
  drop_mult4(dk, (unsigned)grp, m);
  uchar4 o;
  o.x = m[0] != 0.f; o.y = m[1] != 0.f; o.z = m[2] != 0.f; o.w = m[3] != 0.f;
  if (dk.thr == 0) o = make_uchar4(1, 1, 1, 1);
  *(uchar4*)(keep + 4 * grp) = o;
}

// ------------------------------------------------------------------ patches
// out-ordered: one thread = 4 consecutive kw of one (patch row, c, kh)
template <bool GATHER, typename OutT>
__global__ void patchify_kernel(const float* __restrict__ img, const int* __restrict__ idx, OutT* __restrict__ out,
                                long long nrows, int C, int H, int W, int P) {
  const int gw = W / P, gh = H / P;
  const int pd4 = C * P * P / 4;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nrows * pd4) return;
  const long long r = t / pd4;
  const int f = (int)(t - r * pd4) * 4;
  const long long m = GATHER ? (long long)idx[r] : r;
  const long long b = m / (gh * gw);
  const int pi = (int)(m - b * gh * gw);
  const int py = pi / gw, px = pi - py * gw;
  const int c = f / (P * P);
  const int kh = (f - c * P * P) / P;
  const int kw = f - c * P * P - kh * P;
  const f32x4 v = *(const f32x4*)(img + ((b * C + c) * H + (py * P + kh)) * (long long)W + px * P + kw);
  if constexpr (sizeof(OutT) == 2) {
    u32x2 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *(u32x2*)((bf16_t*)out + r * (4LL * pd4) + f) = w;
  } else {
    *(f32x4*)((float*)out + r * (4LL * pd4) + f) = v;
  }
}

// ------------------------------------------------------------------ row gather / scatter
__global__ void gather_rows_bf16_kernel(const float* __restrict__ x, const int* __restrict__ idx, bf16_t* __restrict__ out,
                                        int n_idx, int cols) {
  const int c4n = cols >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_idx * c4n) return;
  const int i = (int)(t / c4n);
  const int c = (int)(t - (long long)i * c4n) * 4;
  const f32x4 v = *(const f32x4*)(x + (long long)idx[i] * cols + c);
  u32x2 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
  *(u32x2*)(out + (long long)i * cols + c) = w;
}

__global__ void scatter_rows_f32_kernel(const bf16_t* __restrict__ src, const int* __restrict__ inv, float* __restrict__ g,
                                        long long rows, int cols) {
  const int c4n = cols >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * c4n) return;
  const long long r = t / c4n;
  const int c = (int)(t - r * c4n) * 4;
  const int i = inv[r];
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (i >= 0) {
    const u32x2 w = *(const u32x2*)(src + (long long)i * cols + c);
    v = f32x4{bf_lo(w[0]), bf_hi(w[0]), bf_lo(w[1]), bf_hi(w[1])};
  }
  *(f32x4*)(g + r * cols + c) = v;
}

__global__ void gather_cls_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int T, int D) {
  const int d4n = D >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * d4n) return;
  const int b = (int)(t / d4n);
  const int c = (int)(t - (long long)b * d4n) * 4;
  *(f32x4*)(out + (long long)b * D + c) = *(const f32x4*)(x + (long long)b * T * D + c);
}

// g[b, 0, :] = gcls[b, :], every other row zero
__global__ void scatter_cls_kernel(const float* __restrict__ gcls, float* __restrict__ g, int B, int T, int D) {
  const int d4n = D >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * T * d4n) return;
  const long long row = t / d4n;
  const int c = (int)(t - row * d4n) * 4;
  const long long b = row / T;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row - b * T == 0) v = *(const f32x4*)(gcls + b * D + c);
  *(f32x4*)(g + row * D + c) = v;
}

// ------------------------------------------------------------------ embedding backward
// grid = (T_out, bsplits); thread = 4 columns
__global__ void embed_bwd_kernel(const float* __restrict__ dtok, const unsigned char* __restrict__ mask,
                                 bf16_t* __restrict__ dproj, float* __restrict__ dpos, float* __restrict__ dmask_token,
                                 float* __restrict__ dbias, float* __restrict__ dcls, int B, int tokens, int tok_offset, int D,
                                 float* __restrict__ ws) {
  const int t = blockIdx.x;
  const int T_out = tokens + tok_offset;
  const int per = (B + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per;
  const int b1 = min(B, b0 + per);
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    f32x4 pos = {0.f, 0.f, 0.f, 0.f}, mt = pos, bias = pos;
    for (int bb = b0; bb < b1; bb += 4) {
      // 4 images per trip: independent loads in flight (the kernel is latency-bound)
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (bb + u < b1) v[u] = *(const f32x4*)(dtok + ((long long)(bb + u) * T_out + t) * D + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = bb + u;
        if (b >= b1) break;
        pos += v[u];
        if (t >= tok_offset) {
          const long long m = (long long)b * tokens + (t - tok_offset);
          const bool masked = mask && mask[m];
          u32x2 w = {0u, 0u};
          if (masked) {
            mt += v[u];
          } else {
            bias += v[u];
            w = u32x2{pack_bf2(v[u][0], v[u][1]), pack_bf2(v[u][2], v[u][3])};
          }
          *(u32x2*)(dproj + m * D + c) = w;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (dpos) atomicAdd(dpos + (long long)t * D + c + r, pos[r]);
      if (t >= tok_offset) {
        if (ws) continue;   // single-row accumulators go through the workspace (below)
        if (dmask_token) atomicAdd(dmask_token + c + r, mt[r]);
        if (dbias) atomicAdd(dbias + c + r, bias[r]);
      } else if (dcls) {
        atomicAdd(dcls + c + r, pos[r]);
      }
    }
    if (ws && t >= tok_offset) {
      // d(mask_token) and d(bias) are single rows fed by EVERY block (~2000 adders per
      // address: the atomic path ran 250-420 us); each block parks its partials instead and
      // embed_bwd_reduce_kernel sums them
      float* slot = ws + ((long long)(blockIdx.y * gridDim.x + t) * 2) * D + c;
      *(f32x4*)slot = mt;
      *(f32x4*)(slot + D) = bias;
    }
  }
}

// column sums of the parked partials: ws [nblk][2][D] -> dmask_token[D], dbias[D]; rows of
// blocks with t < tok_offset were never written and are skipped
__global__ void embed_bwd_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dmask_token, float* __restrict__ dbias,
                                        int T_out, int tok_offset, int nblk, int D) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= D) return;
  f32x4 mt = {0.f, 0.f, 0.f, 0.f}, bias = mt;
#pragma unroll 4
  for (int i = blockIdx.y; i < nblk; i += gridDim.y) {
    if (i % T_out < tok_offset) continue;
    const float* slot = ws + (long long)i * 2 * D + c;
    mt += *(const f32x4*)slot;
    bias += *(const f32x4*)(slot + D);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (dmask_token) atomicAdd(dmask_token + c + r, mt[r]);
    if (dbias) atomicAdd(dbias + c + r, bias[r]);
  }
}

// ------------------------------------------------------------------ losses
__global__ void l1_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ loss_sum,
                               bf16_t* __restrict__ dpred, float gscale, long long n4) {
  __shared__ float part[EW_THREADS / 64];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 d = *(const f32x4*)(pred + 4 * i) - *(const f32x4*)(target + 4 * i);
    acc += fabsf(d[0]) + fabsf(d[1]) + fabsf(d[2]) + fabsf(d[3]);
    if (dpred) {
      float s[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = d[r] > 0.f ? gscale : (d[r] < 0.f ? -gscale : 0.f);
      u32x2 w = {pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3])};
      *(u32x2*)(dpred + 4 * i) = w;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < EW_THREADS / 64; ++w) s += part[w];
    atomicAdd(loss_sum, s);
  }
}

// one wave per row
__global__ void cross_entropy_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                     float* __restrict__ loss_sum, bf16_t* __restrict__ dlogits, float gscale, int B, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* lr = logits + (long long)row * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, lr[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += __expf(lr[c] - m);
  s = wave_sum(s);
  const float lse = m + __logf(s);
  const int y = (int)labels[row];
  if (lane == 0) atomicAdd(loss_sum, lse - lr[y]);
  if (dlogits) {
    for (int c = lane; c < C; c += 64) {
      const float p = __expf(lr[c] - lse);
      dlogits[(long long)row * C + c] = f2bf((p - (c == y ? 1.f : 0.f)) * gscale);
    }
  }
}

// ------------------------------------------------------------------ casts
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(src + 4 * i);
    u32x2 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *(u32x2*)(dst + 4 * i) = w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = f2bf(src[(n4 << 2) + threadIdx.x]);
}

// 64x64 tile through LDS: dst[R,C] = bf16(src), dst_t[C,R] = bf16(src)^T
__global__ void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, bf16_t* __restrict__ dst_t,
                                      int R, int C) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows per pass
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    bf16_t h = 0;
    if (r < R && c < C) {
      h = f2bf(src[(long long)r * C + c]);
      if (dst) dst[(long long)r * C + c] = h;
    }
    tile[rr][tx] = h;
  }
  __syncthreads();
  if (dst_t) {
    for (int cc = ty; cc < 64; cc += 4) {
      const int c = c0 + cc, r = r0 + tx;
      if (c < C && r < R) dst_t[(long long)c * R + r] = tile[tx][cc];
    }
  }
}

// every weight of a parameter store in one launch: block -> (job, 64x64 tile) by binary
// search in the tile prefix sum
__global__ void cast_transpose_batch_kernel(const vitssl_cast_job_t* __restrict__ jobs, const int* __restrict__ tile_start,
                                            int njobs) {
  __shared__ bf16_t tile[64][66];
  const int b = blockIdx.x;
  int lo = 0, hi = njobs;              // tile_start[lo] <= b < tile_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= b) lo = mid; else hi = mid;
  }
  const vitssl_cast_job_t j = jobs[lo];
  const int t = b - tile_start[lo];
  const int tx_n = (j.C + 63) >> 6;
  const int r0 = (t / tx_n) * 64, c0 = (t % tx_n) * 64;
  const int R = j.R, C = j.C;
  bf16_t* dst = (bf16_t*)j.dst;
  bf16_t* dst_t = (bf16_t*)j.dst_t;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, c = c0 + tx;
    bf16_t h = 0;
    if (r < R && c < C) {
      h = f2bf(j.src[(long long)r * C + c]);
      if (dst) dst[(long long)r * C + c] = h;
    }
    tile[rr][tx] = h;
  }
  __syncthreads();
  if (dst_t) {
    for (int cc = ty; cc < 64; cc += 4) {
      const int c = c0 + cc, r = r0 + tx;
      if (c < C && r < R) dst_t[(long long)c * R + r] = tile[tx][cc];
    }
  }
}

// ------------------------------------------------------------------ optimizer / EMA
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                             float gscale) {
  const long long n4 = n >> 2;
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pv = *(const f32x4*)(p + 4 * i);
    const f32x4 gv = *(const f32x4*)(g + 4 * i) * gscale;
    f32x4 mv = *(const f32x4*)(m + 4 * i);
    f32x4 vv = *(const f32x4*)(v + 4 * i);
    pv *= (1.f - lr * wd);
    mv = mv * b1 + gv * (1.f - b1);
    vv = vv * b2 + gv * gv * (1.f - b2);
#pragma unroll
    for (int r = 0; r < 4; ++r) pv[r] -= step_size * mv[r] / (sqrtf(vv[r]) / bc2_sqrt + eps);
    *(f32x4*)(p + 4 * i) = pv;
    *(f32x4*)(m + 4 * i) = mv;
    *(f32x4*)(v + 4 * i) = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    float pv = p[i] * (1.f - lr * wd);
    const float gv = g[i] * gscale;
    const float mv = m[i] * b1 + gv * (1.f - b1);
    const float vv = v[i] * b2 + gv * gv * (1.f - b2);
    pv -= step_size * mv / (sqrtf(vv) / bc2_sqrt + eps);
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}

__global__ void ema_kernel(float* __restrict__ t, const float* __restrict__ s, long long n, float mom) {
  const long long n4 = n >> 2;
  const float om = 1.f - mom;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 tv = *(const f32x4*)(t + 4 * i);
    const f32x4 sv = *(const f32x4*)(s + 4 * i);
    *(f32x4*)(t + 4 * i) = tv * mom + sv * om;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    t[i] = t[i] * mom + s[i] * om;
  }
}

// out[c] += sum_rows x[r][c]; grid = (col chunks of 256*4, row splits)
__global__ void colsum_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, long long rows, int cols) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= cols) return;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = blockIdx.y * per;
  const long long r1 = r0 + per < rows ? r0 + per : rows;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  long long r = r0;
  for (; r + 8 <= r1; r += 8) {                 // eight independent row loads in flight per thread
    u32x2 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = *(const u32x2*)(x + (r + u) * cols + c);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += f32x4{bf_lo(w[u][0]), bf_hi(w[u][0]), bf_lo(w[u][1]), bf_hi(w[u][1])};
  }
  for (; r < r1; ++r) {
    const u32x2 w = *(const u32x2*)(x + r * cols + c);
    acc += f32x4{bf_lo(w[0]), bf_hi(w[0]), bf_lo(w[1]), bf_hi(w[1])};
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) atomicAdd(out + c + k, acc[k]);
}

inline unsigned stream_grid(long long n4) {
  long long g = (n4 + EW_THREADS - 1) / EW_THREADS;
  if (g > 2048) g = 2048;  // 256 CUs x 8 blocks; grid-stride the rest
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int vitssl_dropout_mask(uint8_t* keep, int64_t rows, int64_t cols, vitssl_dropout_t d, void* stream) {
  VS_CHECK_ARG(keep && rows > 0 && cols > 0 && cols % 4 == 0, "dropout_mask: cols must be a multiple of 4");
  VS_CHECK_ARG((unsigned long long)rows * (unsigned long long)cols < (1ull << 34), "dropout_mask: the stream's group counter is 32 bits (rows * cols < 2^34)");
  const long long ng = rows * cols / 4;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid(ng)), dim3(EW_THREADS), 0, (hipStream_t)stream, keep, ng, make_drop_key(d));
  VS_CHECK_LAUNCH("dropout_mask");
  return VITSSL_OK;
}

static int check_patch_args(const char* who, int B, int C, int H, int W, int P) {
  VS_CHECK_ARG(B > 0 && C > 0 && P > 0 && P % 4 == 0, "%s: patch size %d must be a positive multiple of 4", who, P);
  VS_CHECK_ARG(H % P == 0 && W % P == 0, "%s: image %dx%d not divisible by patch %d", who, H, W, P);
  return VITSSL_OK;
}

extern "C" int vitssl_patchify_bf16(const float* img, void* patches, int B, int C, int H, int W, int P, void* stream) {
  VS_CHECK_ARG(img && patches, "patchify: null pointer");
  if (int rc = check_patch_args("patchify", B, C, H, W, P)) return rc;
  const long long nrows = (long long)B * (H / P) * (W / P);
  const long long work = nrows * (C * P * P / 4);
  hipLaunchKernelGGL((patchify_kernel<false, bf16_t>), dim3(ew_grid(work)), dim3(EW_THREADS), 0, (hipStream_t)stream, img,
                     (const int*)nullptr, (bf16_t*)patches, nrows, C, H, W, P);
  VS_CHECK_LAUNCH("patchify");
  return VITSSL_OK;
}

extern "C" int vitssl_gather_patches_f32(const float* img, const int32_t* idx, float* out, int n_idx, int C, int H, int W,
                                         int P, void* stream) {
  VS_CHECK_ARG(img && idx && out && n_idx > 0, "gather_patches: null pointer / empty");
  if (int rc = check_patch_args("gather_patches", 1, C, H, W, P)) return rc;
  const long long work = (long long)n_idx * (C * P * P / 4);
  hipLaunchKernelGGL((patchify_kernel<true, float>), dim3(ew_grid(work)), dim3(EW_THREADS), 0, (hipStream_t)stream, img, idx,
                     out, (long long)n_idx, C, H, W, P);
  VS_CHECK_LAUNCH("gather_patches");
  return VITSSL_OK;
}

extern "C" int vitssl_gather_rows_bf16(const float* x, const int32_t* idx, void* out, int n_idx, int cols, void* stream) {
  VS_CHECK_ARG(x && idx && out && n_idx > 0 && cols > 0 && cols % 4 == 0, "gather_rows: bad args");
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(ew_grid((long long)n_idx * (cols / 4))), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, x, idx, (bf16_t*)out, n_idx, cols);
  VS_CHECK_LAUNCH("gather_rows");
  return VITSSL_OK;
}

extern "C" int vitssl_scatter_rows_f32(const void* src_bf16, const int32_t* inv, float* g, int64_t rows, int cols, void* stream) {
  VS_CHECK_ARG(src_bf16 && inv && g && rows > 0 && cols > 0 && cols % 4 == 0, "scatter_rows: bad args");
  hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3(ew_grid(rows * (cols / 4))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)src_bf16, inv, g, (long long)rows, cols);
  VS_CHECK_LAUNCH("scatter_rows");
  return VITSSL_OK;
}

extern "C" int vitssl_gather_cls_f32(const float* x, float* out, int B, int T, int D, void* stream) {
  VS_CHECK_ARG(x && out && B > 0 && T > 0 && D > 0 && D % 4 == 0, "gather_cls: bad args");
  hipLaunchKernelGGL(gather_cls_kernel, dim3(ew_grid((long long)B * (D / 4))), dim3(EW_THREADS), 0, (hipStream_t)stream, x, out, B, T, D);
  VS_CHECK_LAUNCH("gather_cls");
  return VITSSL_OK;
}

extern "C" int vitssl_scatter_cls_f32(const float* gcls, float* g, int B, int T, int D, void* stream) {
  VS_CHECK_ARG(gcls && g && B > 0 && T > 0 && D > 0 && D % 4 == 0, "scatter_cls: bad args");
  hipLaunchKernelGGL(scatter_cls_kernel, dim3(ew_grid((long long)B * T * (D / 4))), dim3(EW_THREADS), 0, (hipStream_t)stream, gcls, g, B, T, D);
  VS_CHECK_LAUNCH("scatter_cls");
  return VITSSL_OK;
}

static int embed_bwd_splits(int B, int T_out) {
  int bs = 2048 / T_out;
  if (bs < 1) bs = 1;
  if (bs > B) bs = B;
  if (bs > 16) bs = 16;
  return bs;
}

extern "C" int64_t vitssl_embed_bwd_workspace_floats(int B, int tokens, int tok_offset, int D) {
  if (B <= 0 || tokens <= 0 || D <= 0) return 0;
  const int T_out = tokens + tok_offset;
  return (int64_t)embed_bwd_splits(B, T_out) * T_out * 2 * D;
}

extern "C" int vitssl_embed_bwd(const float* dtok, const uint8_t* mask, void* dproj_bf16, float* dpos, float* dmask_token,
                                float* dbias, float* dcls, int B, int tokens, int tok_offset, int D, float* workspace,
                                int64_t workspace_floats, void* stream) {
  VS_CHECK_ARG(dtok && dproj_bf16 && B > 0 && tokens > 0 && D > 0 && D % 4 == 0, "embed_bwd: bad args");
  VS_CHECK_ARG(tok_offset == 0 || tok_offset == 1, "embed_bwd: tok_offset must be 0 or 1");
  VS_CHECK_ARG(!mask || dmask_token, "embed_bwd: mask without dmask_token");
  const int T_out = tokens + tok_offset;
  const int bs = embed_bwd_splits(B, T_out);
  const int64_t need = (int64_t)bs * T_out * 2 * D;
  VS_CHECK_ARG(!workspace || workspace_floats >= need, "embed_bwd: workspace of %lld floats, need %lld", (long long)workspace_floats,
               (long long)need);
  float* ws = (dmask_token || dbias) ? workspace : nullptr;   // NULL workspace = atomics fallback
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(T_out, bs), dim3(EW_THREADS), 0, (hipStream_t)stream, dtok, mask, (bf16_t*)dproj_bf16,
                     dpos, dmask_token, dbias, dcls, B, tokens, tok_offset, D, ws);
  VS_CHECK_LAUNCH("embed_bwd");
  if (ws) {
    const int nblk = bs * T_out;
    hipLaunchKernelGGL(embed_bwd_reduce_kernel, dim3((D / 4 + 63) / 64, 128), dim3(64), 0, (hipStream_t)stream, ws, dmask_token, dbias,
                       T_out, tok_offset, nblk, D);
    VS_CHECK_LAUNCH("embed_bwd_reduce");
  }
  return VITSSL_OK;
}

extern "C" int vitssl_l1_loss(const float* pred, const float* target, float* loss_sum, void* dpred_bf16, float gscale,
                              int64_t n, void* stream) {
  VS_CHECK_ARG(pred && target && loss_sum && n > 0 && n % 4 == 0, "l1_loss: n=%lld must be a positive multiple of 4", (long long)n);
  hipLaunchKernelGGL(l1_loss_kernel, dim3(stream_grid(n / 4)), dim3(EW_THREADS), 0, (hipStream_t)stream, pred, target, loss_sum,
                     (bf16_t*)dpred_bf16, gscale, (long long)(n / 4));
  VS_CHECK_LAUNCH("l1_loss");
  return VITSSL_OK;
}

extern "C" int vitssl_cross_entropy(const float* logits, const int64_t* labels, float* loss_sum, void* dlogits_bf16,
                                    float gscale, int B, int C, void* stream) {
  VS_CHECK_ARG(logits && labels && loss_sum && B > 0 && C > 0, "cross_entropy: bad args");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3((B + 3) / 4), dim3(EW_THREADS), 0, (hipStream_t)stream, logits,
                     (const long long*)labels, loss_sum, (bf16_t*)dlogits_bf16, gscale, B, C);
  VS_CHECK_LAUNCH("cross_entropy");
  return VITSSL_OK;
}

extern "C" int vitssl_colsum_bf16(const void* x_bf16, float* out, int64_t rows, int cols, void* stream) {
  VS_CHECK_ARG(x_bf16 && out && rows > 0 && cols > 0 && cols % 4 == 0, "colsum_bf16: bad args");
  int splits = (int)((rows + 127) / 128);
  if (splits > 512) splits = 512;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((cols / 4 + EW_THREADS - 1) / EW_THREADS, splits), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, (const bf16_t*)x_bf16, out, (long long)rows, cols);
  VS_CHECK_LAUNCH("colsum_bf16");
  return VITSSL_OK;
}

extern "C" int vitssl_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
  VS_CHECK_ARG(src && dst && n > 0, "cast_bf16: bad args");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(stream_grid(n / 4)), dim3(EW_THREADS), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long long)n);
  VS_CHECK_LAUNCH("cast_bf16");
  return VITSSL_OK;
}

extern "C" int vitssl_cast_transpose_bf16(const float* src, void* dst, void* dst_t, int R, int C, void* stream) {
  VS_CHECK_ARG(src && (dst || dst_t) && R > 0 && C > 0, "cast_transpose: bad args");
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(EW_THREADS), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, (bf16_t*)dst_t, R, C);
  VS_CHECK_LAUNCH("cast_transpose");
  return VITSSL_OK;
}

extern "C" int vitssl_cast_transpose_batch(const vitssl_cast_job_t* jobs, const int* tile_start, int njobs, int total_tiles,
                                           void* stream) {
  VS_CHECK_ARG(jobs && tile_start && njobs > 0 && total_tiles > 0, "cast_transpose_batch: bad args");
  hipLaunchKernelGGL(cast_transpose_batch_kernel, dim3(total_tiles), dim3(EW_THREADS), 0, (hipStream_t)stream, jobs, tile_start,
                     njobs);
  VS_CHECK_LAUNCH("cast_transpose_batch");
  return VITSSL_OK;
}

extern "C" int vitssl_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, float wd, int step, float gscale, void* stream) {
  VS_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(n / 4)), dim3(EW_THREADS), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr,
                     beta1, beta2, eps, wd, (float)bc1, (float)sqrt(bc2), gscale);
  VS_CHECK_LAUNCH("adamw");
  return VITSSL_OK;
}

extern "C" int vitssl_ema(float* teacher, const float* student, int64_t n, float m, void* stream) {
  VS_CHECK_ARG(teacher && student && n > 0, "ema: bad args");
  hipLaunchKernelGGL(ema_kernel, dim3(stream_grid(n / 4)), dim3(EW_THREADS), 0, (hipStream_t)stream, teacher, student, (long long)n, m);
  VS_CHECK_LAUNCH("ema");
  return VITSSL_OK;
}
