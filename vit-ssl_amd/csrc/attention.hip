// Fused multi-head self-attention for ViT-sized sequences (N <= 256, dh = 64), gfx950.
//
// The reference materialises S = QK^T / sqrt(dk), softmax, P.V as four passes over a
// [B,H,N,N] tensor (vit_core/attention.py:20-23).  Here one workgroup owns one
// (batch, head): K and V (forward) live in LDS for the whole head, the score row of
// every query is held in registers (N <= 256 keys = 16 accumulators), so the softmax
// is a plain in-register row reduction (no online rescaling needed).
//
// MFMA orientation (v_mfma_f32_16x16x32_bf16; D[i][j], lane holds D[4*(lane>>4)+r][lane&15]):
//   forward / dQ : S^T[key][q] = K.Q^T   -> a lane holds one query column, its keys in
//                  registers; the accumulators are directly the B operand of the next
//                  product that contracts over keys (O^T = V^T.P^T, dQ^T = K^T.dS^T).
//   dK/dV        : S[q][key]   = Q.K^T   -> key on the lane, queries in registers; the
//                  accumulators are the B operand of dV^T = dO^T.P and dK^T = Q^T.dS.
// The contraction index inside a 32-deep MFMA step is permuted consistently on both
// operands (element jj of lane group g <-> row 16*(jj>>2) + 4g + (jj&3) of the step), so
// an accumulator tile never moves between lanes.
//
// One LDS image per tile serves row reads (ds_read_b128) and transposed reads
// (ds_read_b64_tr_b16): 128-byte rows, 16-byte chunk index XORed with ((row>>1)&3)<<1;
// both read kinds are bank-conflict free.
//
// Backward = ONE launch (attn_bwd_fused_kernel): wave w owns keys 32w .. 32w+31 (dK / dV accumulators),
// sweeps the queries, drops every dS tile into an LDS exchange image from which the waves contract dQ;
// delta = rowsum(dO*O) is computed in its prologue.  P is recomputed from the saved log-sum-exp; nothing
// of size N^2 ever reaches HBM.  The older two-launch form (attn_bwd_dq_kernel + attn_bwd_dkv_kernel)
// stays behind VITSSL_ATTN_BWD=split for A/B timing.
#include <stdlib.h>
#include "common.h"

#ifdef VITSSL_ATTN_STAMPS
__device__ unsigned long long* g_attn_stamps = nullptr;   // [workgroup][4] x 100 MHz ticks (tools/attn_stamps.py)
#define ATTN_STAMP(i)                                                                                   \
  do {                                                                                                  \
    if (g_attn_stamps && threadIdx.x == 0) g_attn_stamps[(size_t)blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define ATTN_STAMP(i) \
  do {                \
  } while (0)
#endif

// 1: the fused backward's 8-wave form runs waves 4-7 one barrier behind waves 0-3 (two barriers per query step);
// 0 (default): one barrier per step, all waves in phase.  Measured (B = 256, H = 12, N = 196, alternating runs): staggered
// 276 us, in phase 242 us -- the second barrier per step costs more than the MFMA / VALU overlap between SIMD partners
// returns; kept as an A/B build.  (Moving dV / dK behind the step's barrier, which both forms share, was worth 253 -> 242 us:
// the dS columns are published before the 16 MFMAs + 32 transposed reads instead of after them.)
// 1: the pipelined backward keeps its K^T fragments for the dQ contraction in registers for the whole item (223 -> 218 us at B256 H12 N196; 248 VGPRs)
#ifndef ATTN_BWD_KT_REGS
#define ATTN_BWD_KT_REGS 1
#endif
#ifndef ATTN_BWD_STAGGER
#define ATTN_BWD_STAGGER 0
#endif
// diagnostic builds only (tools/build_variant.sh): persistent forward without 1 = the arithmetic, 2 = the tile prefetch
#ifndef ATTN_FWD_ABLATE
#define ATTN_FWD_ABLATE 0
#endif

namespace {

template <int V>
struct IC2 {
  static constexpr int value = V;
};
// f(IC2<0>{}), f(IC2<1>{}), ... f(IC2<N-1>{}): a loop whose index is a compile-time constant (asm immediates)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IC2<I>{});
    static_for<N, I + 1>(f);
  }
}

constexpr int DH = 64;
constexpr int ROWB = DH * 2;  // bytes per tile row

// Start-up stagger (developer knob VITSSL_ATTN_STAGGER, in 10 ns ticks; 0 = off).  Every workgroup
// is load -> compute -> store on equal work, so the chip runs the three phases in lock-step: the load phase
// of all CUs at once is HBM-bound (~21 GB/s per CU) while HBM idles during the compute phase.  Delaying the
// FIRST workgroup of every CU by a different fraction of the window spreads the phases; later workgroups
// inherit the phase of the CU they land on.
__device__ __forceinline__ void startup_stagger(int first_round_wgs, int window_ticks) {
  if (window_ticks <= 0 || (int)blockIdx.x >= first_round_wgs) return;
  if (threadIdx.x < 64) {
    const unsigned h = (blockIdx.x * 2654435761u) >> 20;               // 12 well-mixed bits
    const unsigned long long delay = (unsigned long long)window_ticks * h >> 12;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < delay) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
}

__device__ __forceinline__ int tile_off(int row, int ch16) { return row * ROWB + ((ch16 ^ (((row >> 1) & 3) << 1)) << 4); }

// cooperative load of X[n][0..63] (n < N, row stride `stride` elements) into a swizzled LDS tile of Np rows
__device__ __forceinline__ void load_tile(char* lds, const bf16_t* g, long long stride, int N, int Np, int tid, int nthreads) {
  for (int idx = tid; idx < Np * 8; idx += nthreads) {
    const int row = idx >> 3, ch = idx & 7;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < N) v = *(const u32x4*)(g + (long long)row * stride + ch * 8);
    *(u32x4*)(lds + tile_off(row, ch)) = v;
  }
}

// asynchronous version: LDS-DMA (buffer_load ... lds), 1 KiB = 8 tile rows per wave
// instruction, the swizzle applied on the SOURCE chunk (same involution as tile_off);
// rows >= N fall outside the buffer descriptor and are zero-filled by the hardware.
// Completion = this wave's vmcnt, then a workgroup barrier.  Np/8 instructions in total,
// spread over NWAVES waves; returns how many this wave issued.
template <int NWAVES>
__device__ __forceinline__ int dma_tile(char* lds, const bf16_t* g, long long stride, int N, int Np, int wave, int lane) {
  const unsigned bytes = (unsigned)((long long)(N - 1) * stride * 2 + ROWB);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)bytes, 0x00020000);
  int n = 0;
  for (int i = wave; i < Np / 8; i += NWAVES) {
    const int row = i * 8 + (lane >> 3);
    const int c = lane & 7;
    const int sc = c ^ (((row >> 1) & 3) << 1);
    const unsigned voff = (unsigned)((long long)row * stride * 2 + sc * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + i * 1024), 16, voff, 0, 0, 0);
    ++n;
  }
  return n;
}

// natural fragment (8 consecutive d) of row `row`, k-step kk, from an LDS tile
__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int kk, int lane) {
  return *(const bf16x8*)(tile + tile_off(row, 4 * kk + (lane >> 4)));
}
// natural fragment straight from global (zero past N)
__device__ __forceinline__ bf16x8 glb_frag(const bf16_t* g, long long stride, int row, int kk, int N, int lane) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (row < N) v = *(const u32x4*)(g + (long long)row * stride + 32 * kk + 8 * (lane >> 4));
  return __builtin_bit_cast(bf16x8, v);
}
// transposed fragment for contraction step s (rows 32s..32s+31, permuted order), columns d0..d0+15
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int s, int dt, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int r0 = 32 * s + 4 * g + q;
  const int r1 = r0 + 16;
  const int ch = 2 * dt;  // 16-B chunk of column d0 = 16*dt; lanes pp>=2 use the odd chunk
  const char* a0 = tile + tile_off(r0, ch + (pp >> 1)) + 8 * (pp & 1);
  const char* a1 = tile + tile_off(r1, ch + (pp >> 1)) + 8 * (pp & 1);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// accumulator tiles (2s, 2s+1) -> B-operand fragment of contraction step s
__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
  u32x4 w = {pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, w);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// One 128-byte row of 64 bf16 outputs.  A lane holds columns 16 dt + 4g .. +3 of the tiles
// dt = 2p, 2p+1; exchanging halves between lane rows (g, g^1) with v_permlane16_swap gives every
// lane 8 contiguous columns, i.e. one 16-byte store instead of two 8-byte ones (64-byte
// segments per row per instruction; HBM writes are sensitive to this).
// Both lanes of an exchanging pair share lane&15, so a row predicate on lane&15 is safe.
__device__ __forceinline__ void store_row_pair16(bf16_t* row, int p, int g, const u32x2& w0, const u32x2& w1) {
  auto lo = __builtin_amdgcn_permlane16_swap(w0[0], w1[0], false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(w0[1], w1[1], false, false);
  const int odd = g & 1;
  const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
  *(u32x4*)(row + (2 * p + odd) * 16 + 4 * (g - odd)) = v;
}
// d[t] of lane row g = dword t*4 + g of a 16-dword row  ->  lane row g gets dwords 4g .. 4g+3 (v_permlane32_swap trades
// the wave halves, v_permlane16_swap the odd / even lane rows)
__device__ __forceinline__ u32x4 lane_rows_transpose4(const unsigned (&d)[4]) {
  auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
  auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
  auto e = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
  auto f = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
  return u32x4{e[0], e[1], f[0], f[1]};
}
__device__ __forceinline__ void store_row_pair16_if(bool ok, bf16_t* row, int p, int g, const u32x2& w0, const u32x2& w1) {
  auto lo = __builtin_amdgcn_permlane16_swap(w0[0], w1[0], false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(w0[1], w1[1], false, false);
  const int odd = g & 1;
  const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
  if (ok) *(u32x4*)(row + (2 * p + odd) * 16 + 4 * (g - odd)) = v;
}
__device__ __forceinline__ u32x2 pack4(const f32x4& o) { return u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])}; }

// ------------------------------------------------------------------ forward
// Softmax VALU budget.  The kernels are VALU-bound, not MFMA-bound (rocprofv3 PMC, forward:
// 13.9 VALU instructions per score element against 1/14 MFMA), so every per-element
// instruction counts:
//   * the key mask costs nothing in the loops: the accumulators of the last two 16-key
//     tiles (the only ones that can hold keys >= N, since 32*NS - 32 < N) START at -inf
//     for those keys, and exp2(-inf) = 0;
//   * 1/sqrt(dh) and log2(e) are folded into one fused multiply-add in front of v_exp_f32
//     (packed: v_pk_fma_f32 handles two scores), the row maximum is taken on raw scores;
//   * probabilities stay unnormalised (in (0, 1]) through the P.V product and the 64 outputs
//     of a query are scaled by 1/sum instead of its up to 256 probabilities.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float SCALE = 0.125f;                  // 1/sqrt(64)
constexpr float SCALE_LOG2E = SCALE * LOG2E;

// -inf for the keys >= N of 16-key tile kt (lane holds keys 16kt + 4g + r), else 0
__device__ __forceinline__ f32x4 key_mask_init(int kt, int g, int N) {
  f32x4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (kt * 16 + 4 * g + r < N) ? 0.f : -INFINITY;
  return v;
}

// in-register row softmax of one 16-query tile on RAW scores: s holds
// S^T[key = 16kt + 4g + r][q = lane&15] (masked keys = -inf); on return s = exp(scale*(s - max))
// (unnormalised), m_out = raw row maximum, sum_out = row sum of s.
template <int NKT>
__device__ __forceinline__ void softmax_tile(f32x4 (&s)[NKT], float& m_out, float& sum_out) {
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float mc = -m * SCALE_LOG2E;
  const f32x4 c4 = {SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E}, m4 = {mc, mc, mc, mc};
  f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const f32x4 e = __builtin_elementwise_fma(s[kt], c4, m4);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[kt][r] = __builtin_amdgcn_exp2f(e[r]);
    part += s[kt];
  }
  float sum = (part[0] + part[1]) + (part[2] + part[3]);
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  m_out = m;
  sum_out = sum;
}

// grid = B*H, NW waves per workgroup: 4 (two workgroups per CU) for long sequences, 2 when
// NS <= 2 (a 37-token DINO local crop has two query pairs: two of four waves would idle and
// more, smaller workgroups share a CU).  NS = number of 32-key steps (Np = 32*NS >= N).
// Each wave processes PAIRS of 16-query tiles so that every K / V fragment read from LDS
// feeds two MFMAs.
template <int NS, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          float* __restrict__ lse, float* __restrict__ probs, int N, int H,
                                                          int stagger_wgs, int stagger_ticks, unsigned char* __restrict__ out8) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Np = 32 * NS;
  constexpr int NKT = 2 * NS;
  startup_stagger(stagger_wgs, stagger_ticks);
  char* Kt = smem;
  char* Vt = smem + Np * ROWB;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long long stride = 3LL * H * DH;
  const bf16_t* qg = qkv + (long long)b * N * stride + h * DH;
  const bf16_t* kg = qg + (long long)H * DH;
  const bf16_t* vg = kg + (long long)H * DH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int nqp = (N + 31) >> 5;   // pairs of 16-query tiles
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 tail0 = key_mask_init(NKT - 2, g, N), tail1 = key_mask_init(NKT - 1, g, N);

  // Q fragments of this wave's first pair (ordinary loads) and the K tile (LDS-DMA) are
  // awaited together; the V tile's DMA is issued only then, so it lands underneath the
  // first S = K.Q^T + softmax.  (No ordinary-load result is consumed while a DMA is in
  // flight: hipcc would answer with a full vmcnt(0) drain.)
  ATTN_STAMP(0);
  bf16x8 qn[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qn[t][0] = glb_frag(qg, stride, wave * 32 + t * 16 + li, 0, N, lane);
    qn[t][1] = glb_frag(qg, stride, wave * 32 + t * 16 + li, 1, N, lane);
  }
  dma_tile<NW>(Kt, kg, stride, N, Np, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  dma_tile<NW>(Vt, vg, stride, N, Np, wave, lane);
  ATTN_STAMP(1);

  const int iters = (nqp + NW - 1) / NW;   // same trip count for every wave (uniform barriers)
  for (int it = 0; it < iters; ++it) {
    const int qp = wave + NW * it;
    const bool active = qp < nqp;      // wave-uniform
    const int q[2] = {qp * 32 + li, qp * 32 + 16 + li};
    bf16x8 qf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      qf[t][0] = qn[t][0];
      qf[t][1] = qn[t][1];
    }
    if (it + 1 < iters) {              // software prefetch of the next pair's Q fragments
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        qn[t][0] = glb_frag(qg, stride, q[t] + 32 * NW, 0, N, lane);
        qn[t][1] = glb_frag(qg, stride, q[t] + 32 * NW, 1, N, lane);
      }
    }
    f32x4 s0[NKT], s1[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const bf16x8 k0 = lds_frag(Kt, kt * 16 + li, 0, lane);
      const bf16x8 k1 = lds_frag(Kt, kt * 16 + li, 1, lane);
      f32x4 a = kt == NKT - 2 ? tail0 : (kt == NKT - 1 ? tail1 : zero4), c = a;
      a = MFMA16(k0, qf[0][0], a);
      c = MFMA16(k0, qf[1][0], c);
      a = MFMA16(k1, qf[0][1], a);
      c = MFMA16(k1, qf[1][1], c);
      s0[kt] = a;
      s1[kt] = c;
    }
    float m[2], sum[2], inv[2];
    softmax_tile<NKT>(s0, m[0], sum[0]);
    softmax_tile<NKT>(s1, m[1], sum[1]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      inv[t] = 1.0f / sum[t];
      if (q[t] < N) {
        if (g == 0) lse[((long long)b * H + h) * N + q[t]] = m[t] * SCALE + __logf(sum[t]);
        if (probs) {
          float* pr = probs + (((long long)b * H + h) * N + q[t]) * N;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 16 + 4 * g + r;
              if (key < N) pr[key] = (t == 0 ? s0[kt][r] : s1[kt][r]) * inv[t];
            }
        }
      }
    }
    // O^T[d][q] = sum_key V[key][d] P[q][key]
    if (it == 0) {   // uniform: every wave's share of the V tile has landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (!active) continue;
    f32x4 o0[4], o1[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      o0[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      o1[dt] = o0[dt];
    }
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const bf16x8 p0 = pack_frag(s0[2 * st], s0[2 * st + 1]);
      const bf16x8 p1 = pack_frag(s1[2 * st], s1[2 * st + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = tr_frag(Vt, st, dt, lane);
        o0[dt] = MFMA16(vf, p0, o0[dt]);
        o1[dt] = MFMA16(vf, p1, o1[dt]);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (q[t] < N) {
        bf16_t* og = out + ((long long)b * N + q[t]) * (H * DH) + h * DH;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
          store_row_pair16(og, pr, g, pack4((t == 0 ? o0[2 * pr] : o1[2 * pr]) * inv[t]),
                           pack4((t == 0 ? o0[2 * pr + 1] : o1[2 * pr + 1]) * inv[t]));
        if (out8) {   // e4m3 image of the same fp32 values (operand of the fp8 out-projection GEMM)
          unsigned char* o8 = out8 + ((long long)b * N + q[t]) * (H * DH) + h * DH + 4 * g;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const f32x4 v = (t == 0 ? o0[dt] : o1[dt]) * inv[t];
            *(unsigned*)(o8 + 16 * dt) = pack_fp8x4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
    if (it == 0) ATTN_STAMP(2);
  }
  ATTN_STAMP(3);
}

// ------------------------------------------------------------------ forward, persistent (long sequences)
// One 8-wave workgroup per CU walks the (batch, head) items b, b + G, ...; the K / V tiles of the NEXT item arrive by
// LDS-DMA in the second tile buffer while the current item is computed, and its Q fragments are requested early too.
// Why (round-2 stamps, B = 256, H = 12, N = 196, two 4-wave workgroups per CU): a workgroup waited 3.8 us for its K tile,
// then ran two query passes of 6.8 + 4.8 us; the two co-resident workgroups hide only part of each other's waits
// (7.7 us per item against 5.8 us of compute).  Here nqp <= 8 query pairs are one pass of the 8 waves and no wave ever
// waits for a tile except in the first item.
//
// The transposed V reads are inline asm: for the ds_read_tr builtin (no memory operand) hipcc assumes a dependency on
// every LDS-DMA in flight and would drain vmcnt(0) -- the next item's prefetch -- in front of each read (same finding as
// csrc/gemm_tn.hip).  The item loop is unrolled by two so that every tile address is a compile-time offset.
// the eight transposed reads of contraction step ST (rows 32 ST .. + 31) of a V tile: four column slices x (rows, rows + 16)
template <int ST>
__device__ __forceinline__ void tr_read_v4(s16x4 (&lo)[4], s16x4 (&hi)[4], const unsigned (&va)[4]) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[dt]) : "v"(va[dt]), "n"(4096 * ST));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dt]) : "v"(va[dt]), "n"(4096 * ST + 2048));
  }
}

// QT = 16-query tiles per wave, 16 / QT waves per workgroup.  QT = 2: a K / V fragment read feeds two MFMAs, 8 waves of up
// to 256 VGPRs; QT = 1: 16 waves of <= 128 VGPRs (four per SIMD) -- twice the LDS fragment traffic, but the kernel is bound
// by waits, not by LDS or issue (rocprofv3 PMC of the QT = 2 form: waves parked in s_waitcnt / s_barrier 41 % of their
// lifetime, issue stalls 29 %, issuing 30 %; matrix pipe 18 % busy), and four waves per SIMD cover each other's waits.
template <int NS, int QT>
__global__ __launch_bounds__(1024 / QT, QT == 1 ? 4 : 2) void attn_fwd_pers_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                                float* __restrict__ lse, float* __restrict__ probs,
                                                                                int N, int H, int nitems,
                                                                                unsigned char* __restrict__ out8) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Np = 32 * NS;
  constexpr int NKT = 2 * NS;
  constexpr int NW = 16 / QT;
  constexpr int TILE = Np * ROWB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const bool active = wave * 16 * QT < N;         // wave-uniform: this wave has at least one query
  const long long stride = 3LL * H * DH;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 tail0 = key_mask_init(NKT - 2, g, N), tail1 = key_mask_init(NKT - 1, g, N);
  int q[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) q[t] = (wave * QT + t) * 16 + li;
  int item = blockIdx.x;
  if (item >= nitems) return;

  auto q_base = [&](int it) { return qkv + (long long)(it / H) * N * stride + (it % H) * DH; };
  auto issue_tiles = [&](int it, char* kt, char* vt) {
    const bf16_t* qg = q_base(it);
    dma_tile<NW>(kt, qg + (long long)H * DH, stride, N, Np, wave, lane);
    dma_tile<NW>(vt, qg + 2LL * H * DH, stride, N, Np, wave, lane);
  };
  // Every vector-memory instruction of an item is issued by every wave that has queries, unconditionally (rows >= N fall
  // outside the buffer descriptors: loads return 0, stores are dropped), so the number of operations behind the prefetch
  // is a constant and the wait at the top of the next item can leave the output stores in flight (counted vmcnt).
  const unsigned q_bytes = (unsigned)((long long)(N - 1) * stride * 2 + ROWB);
  auto load_q = [&](int it, bf16x8 (&qq)[QT][2]) {
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)q_base(it), 0, (int)q_bytes, 0x00020000);
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        qq[t][kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rq, (unsigned)((long long)q[t] * stride * 2 + 64 * kk + 16 * g), 0, 0));
  };
  constexpr unsigned OOB = 0x80000000u;
  const unsigned out_bytes = (unsigned)((long long)N * H * DH * 2);
  // transposed V fragment addresses of this lane (tile-relative), as in tr_frag: row 32 st + 4 g + q (+ 16), column slice
  // dt.  The swizzle of a row depends on (row >> 1) & 3 only, which 32 st and + 16 leave alone: four per-lane bases (one
  // per dt) and the immediates 4096 st (+ 2048) address every read.
  unsigned vrel[4];
  {
    const int tq = (lane >> 2) & 3, tpp = lane & 3;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vrel[dt] = (unsigned)(tile_off(4 * g + tq, 2 * dt + (tpp >> 1)) + 8 * (tpp & 1));
  }

  bf16x8 qn[QT][2];
  issue_tiles(item, smem, smem + TILE);
  load_q(item, qn);
  bool first = true;
#ifdef VITSSL_ATTN_STAMPS
  // diagnostic build: time (10 ns ticks) wave 0 spends in each segment, summed over the workgroup's items:
  // 0 wait + barrier, 1 prefetch issue + S, 2 softmax, 3 P.V, 4 stores; 5 = items
  unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memrealtime();
#define PSTAMP(i)                                                   \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime(); \
    seg[i] += tn_ - tprev;                                          \
    tprev = tn_;                                                    \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
#else
#define PSTAMP(i) \
  do {            \
  } while (0)
#endif

  auto one_item = [&](auto cur_c) {
    constexpr int cur = decltype(cur_c)::value;
    const char* Kt = smem + cur * 2 * TILE;
    const char* Vt = Kt + TILE;
    char* Kn = smem + (cur ^ 1) * 2 * TILE;
    const int b = item / H, h = item - b * H;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < nitems;            // workgroup-uniform
    // this item's tiles (every wave's share) have landed; every wave is done with the other buffer.  The BUILTIN wait (not
    // inline asm): hipcc then knows that the Q loads of this item are complete too, and does not drain the prefetch issued
    // below in front of their first use (s_waitcnt vmcnt(n) alone: expcnt 7, lgkmcnt 15 = 0x0F70 | n)
    if (probs || first || !active) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0) (a wave without queries issues no stores)
    else if (out8) __builtin_amdgcn_s_waitcnt(0x0F70 | (QT * 7));  // the previous item's QT x (1 + 2 + 4) stores may still be in flight
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (QT * 3));            // ... QT x (1 lse + 2 output rows halves)
    first = false;
    __builtin_amdgcn_s_barrier();
    PSTAMP(0);
    bf16x8 qf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      qf[t][0] = qn[t][0];
      qf[t][1] = qn[t][1];
    }
    if (has_next && ATTN_FWD_ABLATE != 2) {
      issue_tiles(next, Kn, Kn + TILE);
      load_q(next, qn);
    }
    if (active && ATTN_FWD_ABLATE != 1) {
      f32x4 sc[QT][NKT];
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const bf16x8 k0 = lds_frag(Kt, kt * 16 + li, 0, lane);
        const bf16x8 k1 = lds_frag(Kt, kt * 16 + li, 1, lane);
#pragma unroll
        for (int t = 0; t < QT; ++t) sc[t][kt] = MFMA16(k0, qf[t][0], kt == NKT - 2 ? tail0 : (kt == NKT - 1 ? tail1 : zero4));
#pragma unroll
        for (int t = 0; t < QT; ++t) sc[t][kt] = MFMA16(k1, qf[t][1], sc[t][kt]);
      }
      PSTAMP(1);
      float m[QT], sum[QT], inv[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) softmax_tile<NKT>(sc[t], m[t], sum[t]);
      PSTAMP(2);
      __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(lse + ((long long)b * H + h) * N), 0, N * 4, 0x00020000);
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        inv[t] = 1.0f / sum[t];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[t] * SCALE + __logf(sum[t])), rl, g == 0 ? (unsigned)q[t] * 4u : OOB, 0, 0);
        if (probs && q[t] < N) {
          // (the row term is formed HERE: hoisted out of the item loop, its 64-bit lane value was the one spilled register pair of
          // the NS = 8 instantiation -- in a path only `return_attn` callers take)
          int qrow = q[t];
          asm volatile("" : "+v"(qrow));
          float* pr = probs + (((long long)b * H + h) * N + qrow) * N;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 16 + 4 * g + r;
              if (key < N) pr[key] = sc[t][kt][r] * inv[t];
            }
        }
      }
      f32x4 o[QT][4];
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned vbase = (unsigned)(size_t)LDS_PTR(Vt);
      unsigned va[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) va[dt] = vbase + vrel[dt];
      // transposed V reads one contraction step ahead of the MFMAs that consume them (two register sets, counted lgkmcnt)
      // (QT = 2 only: with four waves per SIMD the other waves cover the read latency and the registers are not there)
      constexpr int SETS = QT == 2 ? 2 : 1;
      s16x4 lo[SETS][4], hi[SETS][4];
      if constexpr (SETS == 2) tr_read_v4<0>(lo[0], hi[0], va);
      static_for<NS>([&](auto st_c) {
        constexpr int st = decltype(st_c)::value;
        constexpr int set = SETS == 2 ? (st & 1) : 0;
        bf16x8 pf[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) pf[t] = pack_frag(sc[t][2 * st], sc[t][2 * st + 1]);
        if constexpr (SETS == 1) {
          tr_read_v4<st>(lo[0], hi[0], va);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (st + 1 < NS) {
          tr_read_v4<st + 1>(lo[set ^ 1], hi[set ^ 1], va);
          asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the 8 reads just issued may stay in flight
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 v = {lo[set][dt][0], lo[set][dt][1], lo[set][dt][2], lo[set][dt][3], hi[set][dt][0], hi[set][dt][1], hi[set][dt][2], hi[set][dt][3]};
          const bf16x8 vf = __builtin_bit_cast(bf16x8, v);
#pragma unroll
          for (int t = 0; t < QT; ++t) o[t][dt] = MFMA16(vf, pf[t], o[t][dt]);
        }
      });
      PSTAMP(3);
      __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long long)b * N * (H * DH)), 0, (int)out_bytes, 0x00020000);
      // line-shaped stores (csrc/gemm_nt.hip, NT_LDS_T): the tile's 16 rows x 128 bytes pass through the wave's private 2 KiB LDS
      // window behind the four K / V tiles and leave as two instructions of 8 rows x 128 contiguous bytes (same count as before)
      char* xw = smem + 4 * TILE + wave * 2048;
      const int rho = lane >> 3, kap = lane & 7;
      const unsigned tw = (unsigned)(li * 128 + ((g ^ (2 * (li >> 1))) << 3));
      const unsigned tr = (unsigned)(rho * 128 + ((kap ^ (rho >> 1)) << 4));
#pragma unroll
      for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(u32x2*)(xw + (tw ^ (unsigned)(32 * dt))) = pack4(o[t][dt] * inv[t]);
        const u32x4 s1 = *(const u32x4*)(xw + tr), s2 = *(const u32x4*)(xw + ((tr + 1024u) ^ 64u));
        const unsigned rowoff = (unsigned)((wave * QT + t) * 16 + rho) * (unsigned)(H * DH * 2) + (unsigned)(h * DH * 2 + 16 * kap);   // beyond out_bytes for rows >= N
        __builtin_amdgcn_raw_buffer_store_b128(s1, ro, rowoff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(s2, ro, rowoff + 8u * (unsigned)(H * DH * 2), 0, 0);
      }
      if (out8) {
        __amdgpu_buffer_rsrc_t r8 = __builtin_amdgcn_make_buffer_rsrc((void*)(out8 + (long long)b * N * (H * DH)), 0, (int)(out_bytes / 2), 0x00020000);
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const f32x4 v = o[t][dt] * inv[t];
            __builtin_amdgcn_raw_buffer_store_b32(pack_fp8x4(v[0], v[1], v[2], v[3]), r8,
                                                  (unsigned)q[t] * (unsigned)(H * DH) + (unsigned)(h * DH + 4 * g + 16 * dt), 0, 0);
          }
      }
    }
    PSTAMP(4);
#ifdef VITSSL_ATTN_STAMPS
    seg[5] += 1;
#endif
    item = next;
    return has_next;
  };
  for (;;) {
    if (!one_item(IC2<0>{})) break;
    if (!one_item(IC2<1>{})) break;
  }
#ifdef VITSSL_ATTN_STAMPS
  if (g_attn_stamps && threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) g_attn_stamps[(size_t)blockIdx.x * 8 + i] = seg[i];
#endif
#undef PSTAMP
}

// ------------------------------------------------------------------ backward, fused (one launch)
// grid = B*H, 512 threads.  Wave w < NS owns keys [32w, 32w+32): it holds their K / V row
// fragments in registers, sweeps the queries in steps of 32 and accumulates dK / dV exactly
// like a key-stationary dK / dV kernel.  The dS tile of every step is also dropped (bf16) into a
// double-buffered LDS exchange image [32 queries][Np keys]; after one barrier the eight
// waves each take one (16-query tile, 16-column slice) of dQ for those 32 queries and
// contract over ALL keys from that image -- S and dP are computed once instead of twice,
// Q / K / V / dO are read from HBM once, there are no atomics and no N^2 traffic.
// delta = rowsum(dO * O) is computed in the prologue.
// NW = waves per workgroup: 8 for long sequences; 4 when NS <= 4 (every key owner still gets
// a wave, each wave then takes two dQ tiles per step) so that two workgroups share a CU --
// DINO's 96x96 local crops are 37 tokens, and an 8-wave workgroup with six idle waves per
// (batch, head) was launch-bound (5.8 us per workgroup, as long as a 197-token one).
// (launch bound: two waves per SIMD.  Without it hipcc spread the 2- and 4-wave forms over 258-284 registers (VGPRs + AGPRs), which
// admits ONE wave per SIMD: the 37-token launches of DINO's local crops ran two workgroups per CU where the LDS has room for four.)
template <int NS, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ outp,
                                                             const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                             bf16_t* __restrict__ dqkv, int N, int H, int stagger_wgs,
                                                             int stagger_ticks, unsigned char* __restrict__ dq8,
                                                             const float* __restrict__ qscale, float* qamax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int Np = 32 * NS;
  constexpr int SROW = Np * 2 + 16;
  // fp8 path: dq8 = e4m3(dqkv * *qscale) written next to the bf16 image, max |dqkv| recorded in *qamax
  const float qsc = (dq8 && qscale) ? *qscale : 1.0f;
  float qmax = 0.f;
  startup_stagger(stagger_wgs, stagger_ticks);     // exchange-image row (bytes); +16 makes the 8-byte column reads conflict-free
  char* Qt = smem;
  char* Dt = Qt + Np * ROWB;
  char* Kt = Dt + Np * ROWB;
  char* Sx = Kt + Np * ROWB;            // 2 x [32][SROW]
  float* lse_s = (float*)(Sx + 2 * 32 * SROW);
  float* del_s = lse_s + Np;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long long stride = 3LL * H * DH, ostride = (long long)H * DH;
  const bf16_t* qg = qkv + (long long)b * N * stride + h * DH;
  const bf16_t* kg = qg + (long long)H * DH;
  const bf16_t* vg = kg + (long long)H * DH;
  const bf16_t* dog = dout + (long long)b * N * ostride + h * DH;
  const bf16_t* og = outp + (long long)b * N * ostride + h * DH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const bool has_keys = wave < NS;      // wave-uniform; such a wave has at least one key < N

  // ---- prologue.  Per-CU load bandwidth bounds it (all CUs load at once: ~27 GB/s each), so every
  // operand is fetched ONCE: Q, dO, K and O arrive as LDS tiles by LDS-DMA (O borrows the dS exchange
  // area, which is idle until the first step), V fragments come straight from global; the wave's K
  // fragments and delta = rowsum(dO * O) are then read from the LDS tiles.  (Round 1 also fetched K and dO
  // a second time from global: 190 KB per workgroup instead of 140, 7.6 us of a 22 us workgroup.)
  ATTN_STAMP(0);
  char* Ot = Sx;
  dma_tile<NW>(Qt, qg, stride, N, Np, wave, lane);
  dma_tile<NW>(Dt, dog, ostride, N, Np, wave, lane);
  dma_tile<NW>(Kt, kg, stride, N, Np, wave, lane);
  dma_tile<NW>(Ot, og, ostride, N, Np, wave, lane);
  bf16x8 kf[2][2], vf[2][2];
  f32x4 kinit[2];
  if (has_keys) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = wave * 32 + kt * 16 + li;
      const float mi = key < N ? 0.f : -INFINITY;   // masked keys: scores start at -inf -> p = dS = 0
      kinit[kt] = f32x4{mi, mi, mi, mi};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) vf[kt][kk] = glb_frag(vg, stride, key, kk, N, lane);
    }
  }
  float lrow[(Np + 32 * NW - 1) / (32 * NW)];
#pragma unroll
  for (int it = 0; it < (Np + 32 * NW - 1) / (32 * NW); ++it) {
    const int row = (threadIdx.x >> 1) + it * 32 * NW;
    lrow[it] = row < N ? lse[((long long)b * H + h) * N + row] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (has_keys) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) kf[kt][kk] = lds_frag(Kt, wave * 32 + kt * 16 + li, kk, lane);
  }
#pragma unroll
  for (int it = 0; it < (Np + 32 * NW - 1) / (32 * NW); ++it) {   // 2 threads per query row
    const int row = (threadIdx.x >> 1) + it * 32 * NW;
    const int half = threadIdx.x & 1;
    float part = 0.f;
    if (row < N) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 ov = *(const u32x4*)(Ot + tile_off(row, half * 4 + j));
        const u32x4 dv = *(const u32x4*)(Dt + tile_off(row, half * 4 + j));
#pragma unroll
        for (int w = 0; w < 4; ++w) part += bf_lo(ov[w]) * bf_lo(dv[w]) + bf_hi(ov[w]) * bf_hi(dv[w]);
      }
    }
    part += __shfl_xor(part, 1, 64);
    if (half == 0 && row < Np) {
      // pre-multiplied (log2(e), 1/sqrt(dh)); padded queries: +inf makes p = exp2(x - inf) = 0
      lse_s[row] = row < N ? lrow[it] * LOG2E : INFINITY;
      del_s[row] = row < N ? part * SCALE : 0.f;
    }
  }
  __syncthreads();                                     // delta / lse visible; the O tile may now be overwritten by dS
  ATTN_STAMP(1);

  const f32x4 c4 = {SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E}, sc4 = {SCALE, SCALE, SCALE, SCALE};
  f32x4 dv[4][2], dk[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      dv[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dk[dt][kt] = dv[dt][kt];
    }
  // dQ^T[d][q] = sum_key K[key][d] dS[q][key]: the 8 tiles (2 query tiles x 4 column slices) of
  // a 32-query step are dealt round-robin to the NW waves (NW = 4: a wave gets both query
  // tiles of one column slice).  Two accumulators halve the dependent-MFMA chain.
  auto dq_one = [&](const char* sx, int qs, int qt_w, int dt_w) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const char* rowp = sx + (16 * qt_w + li) * SROW + 8 * g;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const u32x2 lo = *(const u32x2*)(rowp + 64 * st);        // keys 32st + 4g .. +3
      const u32x2 hi = *(const u32x2*)(rowp + 64 * st + 32);   // keys 32st + 16 + 4g .. +3
      const u32x4 w = {lo[0], lo[1], hi[0], hi[1]};
      if (st & 1) acc1 = MFMA16(tr_frag(Kt, st, dt_w, lane), __builtin_bit_cast(bf16x8, w), acc1);
      else acc0 = MFMA16(tr_frag(Kt, st, dt_w, lane), __builtin_bit_cast(bf16x8, w), acc0);
    }
    const f32x4 acc = acc0 + acc1;
    const int q = qs * 32 + 16 * qt_w + li;
    if (q < N) {
      if (dqkv) {                                 // fp8 path: the bf16 image is optional (kernel-argument uniform)
        const u32x2 w = {pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])};
        *(u32x2*)(dqkv + ((long long)b * N + q) * stride + h * DH + dt_w * 16 + 4 * g) = w;
      }
      if (dq8) {
        *(unsigned*)(dq8 + ((long long)b * N + q) * stride + h * DH + dt_w * 16 + 4 * g) =
            pack_fp8x4(acc[0] * qsc, acc[1] * qsc, acc[2] * qsc, acc[3] * qsc);
        qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3]))));
      }
    }
  };
  auto dq_tile = [&](const char* sx, int qs) {
#pragma unroll
    for (int tile = 0; tile < 8; tile += NW) dq_one(sx, qs, (wave + tile) >> 2, (wave + tile) & 3);
  };

  // Software pipeline: the dQ tile of step qs-1 (independent MFMAs and LDS reads) is issued between the score MFMAs of step
  // qs and their exp / dS arithmetic, so the matrix pipe has work while the VALU runs the softmax recompute.
  //
  // A step is {scores + dQ(qs-1) + exp / dS / exchange writes} X {dV / dK}: the barrier X publishes the dS columns BEFORE the
  // dV / dK products (round 3; they need nothing from other waves).
  // ATTN_BWD_STAGGER = 1 (experiment, slower: see the macro): a second barrier Y per step, and waves 4-7 (the second wave of
  // every SIMD) run ONE barrier behind waves 0-3, so that the VALU-heavy first half of one wave runs beside the MFMA /
  // LDS-heavy second half of its SIMD partner.
  // Events e0, e1, ...: waves 0-3 pass X(qs) = e(2qs), Y(qs) = e(2qs+1); waves 4-7 pass an extra barrier e0 first, then
  // X(qs) = e(2qs+1), Y(qs) = e(2qs+2).  dS(qs) of every wave is published by e(2qs+1); dQ(qs-1) is read after e(2qs-1)
  // (waves 0-3) / e(2qs) (waves 4-7): every writer is past its X(qs-1).  Buffer (qs+1)&1 is rewritten after e(2qs+1)
  // / e(2qs+2): its last readers (dQ(qs-1)) finished before e(2qs) / e(2qs+1).  Waves 0-3 add one barrier after the loop.
  const bool lag = ATTN_BWD_STAGGER && NW == 8 && wave >= 4;     // wave-uniform
  if (lag) __syncthreads();
#pragma unroll 1
  for (int qs = 0; qs < NS; ++qs) {
    char* sx = Sx + (qs & 1) * 32 * SROW;
    f32x4 p[2][2], ds[2][2];  // [query tile in step][key tile]
    if (has_keys) {
      f32x4 a[2][2], c[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qrow = qs * 32 + t * 16 + li;
        const bf16x8 qa0 = lds_frag(Qt, qrow, 0, lane), qa1 = lds_frag(Qt, qrow, 1, lane);
        const bf16x8 da0 = lds_frag(Dt, qrow, 0, lane), da1 = lds_frag(Dt, qrow, 1, lane);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          a[t][kt] = kinit[kt];
          c[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          a[t][kt] = MFMA16(qa0, kf[kt][0], a[t][kt]);
          a[t][kt] = MFMA16(qa1, kf[kt][1], a[t][kt]);
          c[t][kt] = MFMA16(da0, vf[kt][0], c[t][kt]);
          c[t][kt] = MFMA16(da1, vf[kt][1], c[t][kt]);
        }
      }
      if (qs > 0) dq_tile(Sx + ((qs - 1) & 1) * 32 * SROW, qs - 1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 nl = -*(const f32x4*)(lse_s + qs * 32 + t * 16 + 4 * g);
        const f32x4 nd = -*(const f32x4*)(del_s + qs * 32 + t * 16 + 4 * g);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const f32x4 e = __builtin_elementwise_fma(a[t][kt], c4, nl);
#pragma unroll
          for (int r = 0; r < 4; ++r) p[t][kt][r] = __builtin_amdgcn_exp2f(e[r]);
          ds[t][kt] = p[t][kt] * __builtin_elementwise_fma(c[t][kt], sc4, nd);
          // exchange image: row = query 16t + 4g + r of the step, column = key 32w + 16kt + li
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *(bf16_t*)(sx + (16 * t + 4 * g + r) * SROW + (32 * wave + 16 * kt + li) * 2) = f2bf(ds[t][kt][r]);
        }
      }
    } else if (qs > 0) {
      dq_tile(Sx + ((qs - 1) & 1) * 32 * SROW, qs - 1);
    }
    __syncthreads();   // X: this wave's dS columns of the step are in the image
    if (has_keys) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const bf16x8 pf = pack_frag(p[0][kt], p[1][kt]);
        const bf16x8 sf = pack_frag(ds[0][kt], ds[1][kt]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt][kt] = MFMA16(tr_frag(Dt, qs, dt, lane), pf, dv[dt][kt]);
          dk[dt][kt] = MFMA16(tr_frag(Qt, qs, dt, lane), sf, dk[dt][kt]);
        }
      }
    }
    if (ATTN_BWD_STAGGER && NW == 8) __syncthreads();   // Y
  }
  if (ATTN_BWD_STAGGER && NW == 8 && !lag) __syncthreads();     // waves 0-3: the event waves 4-7 pass as their last Y
  dq_tile(Sx + ((NS - 1) & 1) * 32 * SROW, NS - 1);
  ATTN_STAMP(2);
  if (has_keys) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = wave * 32 + kt * 16 + li;
      if (dqkv) {                                 // (wave-uniform; the lane predicate sits inside: the row exchange needs all lanes)
        bf16_t* dkg = dqkv + ((long long)b * N + key) * stride + (long long)H * DH + h * DH;
        bf16_t* dvg = dkg + (long long)H * DH;
        // each tensor's 128-byte row leaves in two back-to-back 16-byte-per-lane stores
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) store_row_pair16_if(key < N, dkg, pr, g, pack4(dk[2 * pr][kt]), pack4(dk[2 * pr + 1][kt]));
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) store_row_pair16_if(key < N, dvg, pr, g, pack4(dv[2 * pr][kt]), pack4(dv[2 * pr + 1][kt]));
      }
      if (dq8) {    // wave-uniform.  The row's 64 bytes: a 4x4 transpose over the four lane rows gives every lane 16 contiguous bytes
        unsigned qk[4], qv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const f32x4 a = dk[dt][kt], c = dv[dt][kt];
          qk[dt] = pack_fp8x4(a[0] * qsc, a[1] * qsc, a[2] * qsc, a[3] * qsc);
          qv[dt] = pack_fp8x4(c[0] * qsc, c[1] * qsc, c[2] * qsc, c[3] * qsc);
          if (key < N) {
            qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
            qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3]))));
          }
        }
        const u32x4 rk = lane_rows_transpose4(qk), rv = lane_rows_transpose4(qv);
        if (key < N) {
          unsigned char* k8 = dq8 + ((long long)b * N + key) * stride + (long long)H * DH + h * DH + 16 * g;
          *(u32x4*)k8 = rk;
          *(u32x4*)(k8 + (long long)H * DH) = rv;
        }
      }
    }
  }
  if (dq8 && qamax) {
    qmax = wave_max(qmax);
    unsigned* slot = (unsigned*)qamax;
    if (lane == 0 && __float_as_uint(qmax) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(qmax));
  }
#ifdef VITSSL_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  ATTN_STAMP(3);
}


// ------------------------------------------------------------------ backward, fused, pipelined prologue (N > 128)
// attn_bwd_fused_kernel with the operand loads spread over the query sweep (round 3).  Stamps of the fused kernel (B = 256,
// H = 12, N = 196): prologue 4.9 us of a 19.6 us workgroup -- all four tiles (Q, dO, K, O: 112 KB) land before the first MFMA,
// although step qs of the sweep touches only rows 32 qs .. 32 qs + 31 of Q / dO (and, through delta, of O).  Here the
// prologue waits for K, V and the first TWO 32-row slices; slice qs + 2 is requested by LDS-DMA at the top of step qs and is
// covered by the wait + barrier that ends the step; delta = rowsum(dO * O) of slice qs + 1 is computed during step qs by the
// last wave (which owns no keys at N <= 224) from a three-slot ring of O slices.
// What that needs: nothing in the sweep may make hipcc drain the DMA queue -- the transposed reads are inline asm (for the
// ds_read_tr builtin hipcc waits vmcnt(0) while an LDS-DMA is in flight), the step's barrier is a raw s_barrier behind a
// BUILTIN s_waitcnt (vmcnt(0) of this wave's own, by then one step old, requests + lgkmcnt(0)), never __syncthreads().
__device__ __forceinline__ void dma_rows8(__amdgpu_buffer_rsrc_t rs, char* lds_rows, int lrow0, long long stride, int grow0, int lane) {
  const int rl = lane >> 3, c = lane & 7;
  const int sc = c ^ ((((lrow0 + rl) >> 1) & 3) << 1);      // the tile's swizzle, by the row's position in its tile / ring slot
  const unsigned voff = (unsigned)((long long)(grow0 + rl) * stride * 2 + sc * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds_rows), 16, voff, 0, 0, 0);
}
template <int IMM>
__device__ __forceinline__ void ds_tr16(s16x4& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}
__device__ __forceinline__ bf16x8 join_tr(const s16x4& lo, const s16x4& hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int NS>
__global__ __launch_bounds__(512) void attn_bwd_pipe_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ outp,
                                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dqkv, int N, int H, int stagger_wgs,
                                                            int stagger_ticks, unsigned char* __restrict__ dq8,
                                                            const float* __restrict__ qscale, float* qamax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 8;
  constexpr int Np = 32 * NS;
  constexpr int TILE = Np * ROWB;
  constexpr int SROW = Np * 2 + 16;
  const float qsc = (dq8 && qscale) ? *qscale : 1.0f;
  float qmax = 0.f;
  startup_stagger(stagger_wgs, stagger_ticks);
  char* Qt = smem;
  char* Dt = Qt + TILE;
  char* Kt = Dt + TILE;
  char* Sx = Kt + TILE;                 // 2 x [32][SROW]
  float* lse_s = (float*)(Sx + 2 * 32 * SROW);
  float* del_s = lse_s + Np;
  char* Or = (char*)(del_s + Np);       // ring of three 32-row slices of O
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const long long stride = 3LL * H * DH, ostride = (long long)H * DH;
  const bf16_t* qg = qkv + (long long)b * N * stride + h * DH;
  const bf16_t* kg = qg + (long long)H * DH;
  const bf16_t* vg = kg + (long long)H * DH;
  const bf16_t* dog = dout + (long long)b * N * ostride + h * DH;
  const bf16_t* og = outp + (long long)b * N * ostride + h * DH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const bool has_keys = wave < NS;      // wave-uniform
  const bool delta_wave = wave == NW - 1;

  __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)qg, 0, (int)((long long)(N - 1) * stride * 2 + ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)dog, 0, (int)((long long)(N - 1) * ostride * 2 + ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)og, 0, (int)((long long)(N - 1) * ostride * 2 + ROWB), 0x00020000);
  // the dQ rows of this (batch, head): [N, 3 H dh] bf16 (and its e4m3 image), row stride `stride`; only the q part is written here
  __amdgpu_buffer_rsrc_t rsDQ = __builtin_amdgcn_make_buffer_rsrc((void*)(dqkv ? dqkv + (long long)b * N * stride : nullptr), 0,
                                                                  dqkv ? (int)((long long)N * stride * 2) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rsDQ8 = __builtin_amdgcn_make_buffer_rsrc((void*)(dq8 ? dq8 + (long long)b * N * stride : nullptr), 0,
                                                                   dq8 ? (int)((long long)N * stride) : 0, 0x00020000);
  const int dq_stores = (dqkv ? 1 : 0) + (dq8 ? 1 : 0);      // vector-memory stores of one dq_tile call (launch-uniform)
  // slice s = rows 32 s .. 32 s + 31 of Q, dO (into their tiles) and O (into ring slot s % 3): 12 requests of 8 rows,
  // waves 0-3: one of Q and one of O, waves 4-7: one of dO
  auto issue_slice = [&](int s) {
    const int j = wave & 3;
    if (wave < 4) {
      dma_rows8(rsQ, Qt + (32 * s + 8 * j) * ROWB, 32 * s + 8 * j, stride, 32 * s + 8 * j, lane);
      dma_rows8(rsO, Or + ((s % 3) * 32 + 8 * j) * ROWB, 8 * j, ostride, 32 * s + 8 * j, lane);
    } else {
      dma_rows8(rsD, Dt + (32 * s + 8 * j) * ROWB, 32 * s + 8 * j, ostride, 32 * s + 8 * j, lane);
    }
  };
  // delta (pre-multiplied by 1/sqrt(dh)) of slice s, by ONE wave: 2 lanes per row
  auto delta_slice = [&](int s) {
    const int rl = lane >> 1, half = lane & 1;
    const int row = 32 * s + rl;
    const char* orow = Or + (s % 3) * 32 * ROWB;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 ov = *(const u32x4*)(orow + tile_off(rl, half * 4 + j));
      const u32x4 dv = *(const u32x4*)(Dt + tile_off(row, half * 4 + j));
#pragma unroll
      for (int w = 0; w < 4; ++w) part += bf_lo(ov[w]) * bf_lo(dv[w]) + bf_hi(ov[w]) * bf_hi(dv[w]);
    }
    part += __shfl_xor(part, 1, 64);
    if (half == 0) del_s[row] = row < N ? part * SCALE : 0.f;
  };

  // ---- prologue: K tile, slices 0 and 1, V fragments, lse
  ATTN_STAMP(0);
  dma_tile<NW>(Kt, kg, stride, N, Np, wave, lane);
  issue_slice(0);
  if (NS > 1) issue_slice(1);
  bf16x8 kf[2][2], vf[2][2];
  f32x4 kinit[2];
  if (has_keys) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = wave * 32 + kt * 16 + li;
      const float mi = key < N ? 0.f : -INFINITY;
      kinit[kt] = f32x4{mi, mi, mi, mi};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) vf[kt][kk] = glb_frag(vg, stride, key, kk, N, lane);
    }
  }
  {
    const int row = threadIdx.x >> 1;                 // 256 >= Np rows, two threads per row
    const float l = row < N ? lse[((long long)b * H + h) * N + row] : 0.f;
    if ((threadIdx.x & 1) == 0 && row < Np) lse_s[row] = row < N ? l * LOG2E : INFINITY;   // padded queries: p = exp2(x - inf) = 0
  }
  __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0) lgkmcnt(0)
  __syncthreads();
  if (has_keys) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) kf[kt][kk] = lds_frag(Kt, wave * 32 + kt * 16 + li, kk, lane);
  }
  if (delta_wave) delta_slice(0);
  __syncthreads();
  ATTN_STAMP(1);

  const f32x4 c4 = {SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E}, sc4 = {SCALE, SCALE, SCALE, SCALE};
  f32x4 dv[4][2], dk[4][2];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      dv[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dk[dt][kt] = dv[dt][kt];
    }
  // transposed-read addresses (as tr_frag): tile base + row 4 g + q, chunk of column slice dt; + 4096 per 32-row step, + 2048
  // for the second 16 rows, + TILE from the Q tile to the dO tile
  const int tq = (lane >> 2) & 3, tpp = lane & 3;
  const unsigned qbase = (unsigned)(size_t)LDS_PTR(Qt);
  unsigned vrel[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vrel[dt] = qbase + (unsigned)(tile_off(4 * g + tq, 2 * dt + (tpp >> 1)) + 8 * (tpp & 1));
  const int dt_w = wave & 3, qt_w = wave >> 2;        // this wave's dQ tile of every step: (query tile, column slice)
  const unsigned kaddr = vrel[0] - qbase + (unsigned)(size_t)LDS_PTR(Kt) +
                         (unsigned)(tile_off(4 * g + tq, 2 * dt_w + (tpp >> 1)) - tile_off(4 * g + tq, (tpp >> 1)));

  // dQ^T[d][q] = sum_key K[key][d] dS[q][key] for this wave's tile of step qs, from exchange image sx
#if ATTN_BWD_KT_REGS
  // this wave's K^T fragments (16 head dimensions x all keys) are the same in every step: read once, 4 NS registers
  s16x4 klo[NS], khi[NS];
  auto load_kt = [&]() {
    static_for<NS>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      ds_tr16<4096 * st>(klo[st], kaddr);
      ds_tr16<4096 * st + 2048>(khi[st], kaddr);
    });
  };
#endif
  auto dq_tile = [&](const char* sx, int qs) {
#if !ATTN_BWD_KT_REGS
    s16x4 klo[NS], khi[NS];
    static_for<NS>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      ds_tr16<4096 * st>(klo[st], kaddr);
      ds_tr16<4096 * st + 2048>(khi[st], kaddr);
    });
#endif
    const char* rowp = sx + (16 * qt_w + li) * SROW + 8 * g;
    u32x2 dlo[NS], dhi[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      dlo[st] = *(const u32x2*)(rowp + 64 * st);        // keys 32st + 4g .. +3
      dhi[st] = *(const u32x2*)(rowp + 64 * st + 32);   // keys 32st + 16 + 4g .. +3
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const u32x4 w = {dlo[st][0], dlo[st][1], dhi[st][0], dhi[st][1]};
      if (st & 1) acc1 = MFMA16(join_tr(klo[st], khi[st]), __builtin_bit_cast(bf16x8, w), acc1);
      else acc0 = MFMA16(join_tr(klo[st], khi[st]), __builtin_bit_cast(bf16x8, w), acc0);
    }
    const f32x4 acc = acc0 + acc1;
    const int q = qs * 32 + 16 * qt_w + li;
    // unconditional stores through descriptors (rows >= N are out of range and dropped): every wave issues exactly one
    // store per image and call, so the step's wait can be a counted one that leaves them in flight
    const unsigned qoff = (unsigned)q * (unsigned)(stride * 2) + (unsigned)((h * DH + dt_w * 16 + 4 * g) * 2);
    if (dqkv) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])}, rsDQ, qoff, 0, 0);
    if (dq8) {
      __builtin_amdgcn_raw_buffer_store_b32(pack_fp8x4(acc[0] * qsc, acc[1] * qsc, acc[2] * qsc, acc[3] * qsc), rsDQ8, qoff >> 1, 0, 0);
      if (q < N) qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3]))));
    }
  };

#if ATTN_BWD_KT_REGS
  load_kt();                                       // (the K tile landed with the prologue's wait; the step's lgkmcnt(0) covers the reads)
#endif
#pragma unroll 1
  for (int qs = 0; qs < NS; ++qs) {
    char* sx = Sx + (qs & 1) * 32 * SROW;
    if (qs + 2 < NS) issue_slice(qs + 2);
    if (delta_wave && qs + 1 < NS) delta_slice(qs + 1);       // slice qs + 1 landed with the wait that ended step qs - 1
    f32x4 p[2][2], ds[2][2];  // [query tile in step][key tile]
    if (has_keys) {
      f32x4 a[2][2], c[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qrow = qs * 32 + t * 16 + li;
        const bf16x8 qa0 = lds_frag(Qt, qrow, 0, lane), qa1 = lds_frag(Qt, qrow, 1, lane);
        const bf16x8 da0 = lds_frag(Dt, qrow, 0, lane), da1 = lds_frag(Dt, qrow, 1, lane);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          a[t][kt] = kinit[kt];
          c[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          a[t][kt] = MFMA16(qa0, kf[kt][0], a[t][kt]);
          a[t][kt] = MFMA16(qa1, kf[kt][1], a[t][kt]);
          c[t][kt] = MFMA16(da0, vf[kt][0], c[t][kt]);
          c[t][kt] = MFMA16(da1, vf[kt][1], c[t][kt]);
        }
      }
      if (qs > 0) dq_tile(Sx + ((qs - 1) & 1) * 32 * SROW, qs - 1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 nl = -*(const f32x4*)(lse_s + qs * 32 + t * 16 + 4 * g);
        const f32x4 nd = -*(const f32x4*)(del_s + qs * 32 + t * 16 + 4 * g);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const f32x4 e = __builtin_elementwise_fma(a[t][kt], c4, nl);
#pragma unroll
          for (int r = 0; r < 4; ++r) p[t][kt][r] = __builtin_amdgcn_exp2f(e[r]);
          ds[t][kt] = p[t][kt] * __builtin_elementwise_fma(c[t][kt], sc4, nd);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *(bf16_t*)(sx + (16 * t + 4 * g + r) * SROW + (32 * wave + 16 * kt + li) * 2) = f2bf(ds[t][kt][r]);
        }
      }
    } else if (qs > 0) {
      dq_tile(Sx + ((qs - 1) & 1) * 32 * SROW, qs - 1);
    }
    // end of the step's first half: this wave's dS columns, its delta rows and its share of slice qs + 2 are complete.
    // The dQ store(s) of this step were issued AFTER the slice's requests: they may stay in flight (counted vmcnt; a store
    // acknowledgement is ~1 us away), lgkmcnt(0) covers the LDS writes.
    if (qs == 0 || dq_stores == 0) __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0)
    else if (dq_stores == 1) __builtin_amdgcn_s_waitcnt(0x0071);           // vmcnt(1)
    else __builtin_amdgcn_s_waitcnt(0x0072);                               // vmcnt(2)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if (has_keys) {
      // dV^T += dO^T P, dK^T += Q^T dS: the transposed fragments of the step's 32 rows serve both key tiles
      s16x4 qlo[4], qhi[4], dlo[4], dhi[4];
      const unsigned soff = (unsigned)qs * 4096u;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        ds_tr16<TILE>(dlo[dt], vrel[dt] + soff);
        ds_tr16<TILE + 2048>(dhi[dt], vrel[dt] + soff);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        ds_tr16<0>(qlo[dt], vrel[dt] + soff);
        ds_tr16<2048>(qhi[dt], vrel[dt] + soff);
      }
      bf16x8 pf[2], sf[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        pf[kt] = pack_frag(p[0][kt], p[1][kt]);
        sf[kt] = pack_frag(ds[0][kt], ds[1][kt]);
      }
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the dO fragments are in; the Q fragments land under the dV MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 dfr = join_tr(dlo[dt], dhi[dt]);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) dv[dt][kt] = MFMA16(dfr, pf[kt], dv[dt][kt]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 qfr = join_tr(qlo[dt], qhi[dt]);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) dk[dt][kt] = MFMA16(qfr, sf[kt], dk[dt][kt]);
      }
    }
  }
  dq_tile(Sx + ((NS - 1) & 1) * 32 * SROW, NS - 1);
  ATTN_STAMP(2);
  if (has_keys) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = wave * 32 + kt * 16 + li;
      if (dqkv) {
        // Line-shaped stores (as the NT GEMM epilogue, csrc/gemm_nt.hip NT_LDS_T): the 16 x 128-byte tile goes through a private
        // 2 KiB LDS window and leaves as 8 rows x 128 contiguous bytes per instruction.  The window is rows 16 w .. 16 w + 15 of
        // the Q tile: slices <= NS - 2 (rows < 32 (NS - 1), NS >= 5) were last read before the final step's barrier.
        char* xw = Qt + wave * 2048;
        const int rho = lane >> 3, kap = lane & 7;
        const unsigned tw = (unsigned)(li * 128 + ((g ^ (2 * (li >> 1))) << 3));
        const unsigned tr = (unsigned)(rho * 128 + ((kap ^ (rho >> 1)) << 4));
        const int key_a = wave * 32 + kt * 16 + rho;
        bf16_t* dka = dqkv + ((long long)b * N + key_a) * stride + (long long)H * DH + h * DH + 8 * kap;
#pragma unroll
        for (int tsel = 0; tsel < 2; ++tsel) {        // dK, then dV
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) *(u32x2*)(xw + (tw ^ (unsigned)(32 * dt))) = pack4(tsel == 0 ? dk[dt][kt] : dv[dt][kt]);
          const u32x4 s1 = *(const u32x4*)(xw + tr), s2 = *(const u32x4*)(xw + ((tr + 1024u) ^ 64u));
          bf16_t* dst = dka + (long long)tsel * H * DH;
          if (key_a < N) *(u32x4*)dst = s1;
          if (key_a + 8 < N) *(u32x4*)(dst + 8 * stride) = s2;
        }
      }
      if (dq8) {
        unsigned qk[4], qv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const f32x4 a = dk[dt][kt], c = dv[dt][kt];
          qk[dt] = pack_fp8x4(a[0] * qsc, a[1] * qsc, a[2] * qsc, a[3] * qsc);
          qv[dt] = pack_fp8x4(c[0] * qsc, c[1] * qsc, c[2] * qsc, c[3] * qsc);
          if (key < N) {
            qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
            qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3]))));
          }
        }
        const u32x4 rk = lane_rows_transpose4(qk), rv = lane_rows_transpose4(qv);
        if (key < N) {
          unsigned char* k8 = dq8 + ((long long)b * N + key) * stride + (long long)H * DH + h * DH + 16 * g;
          *(u32x4*)k8 = rk;
          *(u32x4*)(k8 + (long long)H * DH) = rv;
        }
      }
    }
  }
  if (dq8 && qamax) {
    qmax = wave_max(qmax);
    unsigned* slot = (unsigned*)qamax;
    if (lane == 0 && __float_as_uint(qmax) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(qmax));
  }
#ifdef VITSSL_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  ATTN_STAMP(3);
}


// ------------------------------------------------------------------ backward, fused, persistent (128 < N <= 224; round 4)
// attn_bwd_pipe_kernel inside an item loop (one workgroup per CU, items b, b + G, ...), with the next item's operands requested
// while the current one is swept, so that the 3.2 us prologue of the pipelined kernel (K tile, V fragments, two slices, lse: nothing
// else runs on the CU meanwhile, 12 times per CU at B = 256, H = 12) shrinks to fragment reads from LDS.  What moves where:
//   * K': the K tile is dead once its fragments sit in registers (kf, and the K^T fragments of the dQ products: ATTN_BWD_KT_REGS), so
//     K' lands in it during the sweep; V' lands in a tile of its own (the pipelined kernel reads V with ordinary loads: +28 KiB of LDS,
//     which is why N > 224 stays on that kernel).  Both are requested right BEHIND the dQ stores of step 1, NS instructions per wave,
//     and that step's wait leaves them in flight next to the stores (counted vmcnt); step 2's wait covers them.
//   * slices 0 and 1 of Q', dO', O' are requested at the top of the LAST step: rows 0-63 of the Q / dO tiles were last read in steps
//     0 / 1, the O ring's slots 0 / 1 before the barrier of step NS - 2; the last step's wait + barrier covers them.
//   * lse' goes to the other half of a two-row lse buffer (ordinary loads at the top of the last step).
//   * the dK / dV store windows move from rows 0-127 of the Q tile to rows 64-127 of the Q and dO tiles (slices 2, 3: dead since
//     steps 2 / 3, NS >= 5), away from the prefetched slices.
// The item's first barrier (behind the fragment reads) also separates the last dQ products' reads of the exchange image and the
// store windows of the slower waves from the next item's first writes / requests.
// (Q8 = the fp8 operand path's extras -- e4m3 image of dqkv, its scale and maximum -- as a template flag: as run-time arguments they
// cost a dozen scalar registers the bf16 launches never use, in a kernel that is short of them.)
template <int NS, bool Q8>
__global__ __launch_bounds__(512) void attn_bwd_pers_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ outp,
                                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dqkv_arg, int N, int H, int nitems,
                                                            unsigned char* __restrict__ dq8_arg, const float* __restrict__ qscale,
                                                            float* qamax) {
  // bf16 launches: dqkv is there, no e4m3 image; fp8 launches: the e4m3 image is there, the bf16 one optional
  bf16_t* const dqkv = dqkv_arg;
  unsigned char* const dq8 = Q8 ? dq8_arg : nullptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(NS >= 5 && NS <= 7, "the persistent backward is built for 129-224 tokens");
  constexpr int NW = 8;
  constexpr int Np = 32 * NS;
  constexpr int TILE = Np * ROWB;
  constexpr int SROW = Np * 2 + 16;
  const float qsc = (Q8 && qscale) ? *qscale : 1.0f;
  float qmax = 0.f;
  char* Qt = smem;
  char* Dt = Qt + TILE;
  char* Kt = Dt + TILE;
  char* Vt = Kt + TILE;
  char* Sx = Vt + TILE;                 // 2 x [32][SROW]
  float* lse_s = (float*)(Sx + 2 * 32 * SROW);      // [2][Np]: this item's / the next item's
  float* del_s = lse_s + 2 * Np;
  char* Or = (char*)(del_s + Np);       // ring of three 32-row slices of O
  const long long stride = 3LL * H * DH, ostride = (long long)H * DH;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, li = lane & 15;
  const bool has_keys = wave < NS;      // wave-uniform
  const bool delta_wave = wave == NW - 1;
  const int dq_stores = ((!Q8 || dqkv) ? 1 : 0) + (Q8 ? 1 : 0);      // vector-memory stores of one dq_tile call (launch-uniform)
  const int G = gridDim.x;
  int item = blockIdx.x;
  if (item >= nitems) return;

  // An item is carried as two element offsets (its q rows in qkv, its rows in out / dout): 4 scalar registers.  The buffer
  // descriptors are rebuilt where they are used (scalar arithmetic); kept alive for two items they cost 40 scalar registers, the
  // kernel spilled scalars into vector registers and those into scratch (30-41 VGPRs).
  struct Item {
    long long qoff, ooff;
  };
  auto item_of = [&](int it) {
    const int b = it / H, h = it - b * H;
    Item r;
    r.qoff = (long long)b * N * stride + h * DH;
    r.ooff = (long long)b * N * ostride + h * DH;
    return r;
  };
  const int qbytes = (int)((long long)(N - 1) * stride * 2 + ROWB), obytes = (int)((long long)(N - 1) * ostride * 2 + ROWB);
  // One request = 8 rows x 128 bytes of an operand whose first row is `base`.  The lane's part of the source offset (its row inside
  // the group, its swizzled chunk: row groups start at multiples of 8 rows, so the chunk XOR depends on the lane only) is ONE
  // register per row stride; the row group's part goes into the DESCRIPTOR (window = the operand from row grow0 on, scalar
  // arithmetic, exact range check: rows >= N are zero-filled as before).  Per-call lane offsets, as in dma_rows8, were hoisted out
  // of the loops by hipcc and cost 40 spilled VGPRs.
  const int rl8 = lane >> 3;
  const unsigned lane_q = (unsigned)(rl8 * stride * 2 + (((lane & 7) ^ (((rl8 >> 1) & 3) << 1)) << 4));
  const unsigned lane_o = (unsigned)(rl8 * ostride * 2 + (((lane & 7) ^ (((rl8 >> 1) & 3) << 1)) << 4));
  auto dma8 = [&](const bf16_t* base, int total_bytes, long long row_elems, char* lds_rows, unsigned lane_off, int grow0) {
    const long long skip = (long long)grow0 * row_elems * 2;
    const int left = total_bytes > skip ? (int)(total_bytes - skip) : 0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long long)grow0 * row_elems), 0, left, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds_rows), 16, lane_off, 0, 0, 0);
  };
  // slice s = rows 32 s .. 32 s + 31 of Q, dO (into their tiles) and O (into ring slot s % 3): 12 requests of 8 rows,
  // waves 0-3: one of Q and one of O, waves 4-7: one of dO
  auto issue_slice = [&](const Item& it, int s) {
    const int j = wave & 3;
    const int r0 = 32 * s + 8 * j;
    if (wave < 4) {
      dma8(qkv + it.qoff, qbytes, stride, Qt + r0 * ROWB, lane_q, r0);
      dma8(outp + it.ooff, obytes, ostride, Or + ((s % 3) * 32 + 8 * j) * ROWB, lane_o, r0);
    } else {
      dma8(dout + it.ooff, obytes, ostride, Dt + r0 * ROWB, lane_o, r0);
    }
  };
  // the K and V tiles of an item: 2 x 4 NS requests of 8 rows, exactly NS per wave (row group i of K for even, of V for odd turns)
  auto issue_kv = [&](const Item& it) {
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const int idx = n * NW + wave;                 // 0 .. 8 NS - 1
      const int i = idx >> 1;                        // row group of 8 rows, 0 .. 4 NS - 1
      if (idx & 1) dma8(qkv + it.qoff + 2LL * H * DH, qbytes, stride, Vt + i * 8 * ROWB, lane_q, 8 * i);
      else dma8(qkv + it.qoff + (long long)H * DH, qbytes, stride, Kt + i * 8 * ROWB, lane_q, 8 * i);
    }
  };
  auto load_lse = [&](int it, int buf) {              // padded queries: p = exp2(x - inf) = 0
    const int row = threadIdx.x >> 1;                 // 256 >= Np rows, two threads per row
    const float l = row < N ? lse[(long long)it * N + row] : 0.f;
    if ((threadIdx.x & 1) == 0 && row < Np) lse_s[buf * Np + row] = row < N ? l * LOG2E : INFINITY;
  };
  // delta (pre-multiplied by 1/sqrt(dh)) of slice s, by ONE wave: 2 lanes per row
  auto delta_slice = [&](int s) {
    const int rl = lane >> 1, half = lane & 1;
    const int row = 32 * s + rl;
    const char* orow = Or + (s % 3) * 32 * ROWB;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 ov = *(const u32x4*)(orow + tile_off(rl, half * 4 + j));
      const u32x4 dv = *(const u32x4*)(Dt + tile_off(row, half * 4 + j));
#pragma unroll
      for (int w = 0; w < 4; ++w) part += bf_lo(ov[w]) * bf_lo(dv[w]) + bf_hi(ov[w]) * bf_hi(dv[w]);
    }
    part += __shfl_xor(part, 1, 64);
    if (half == 0) del_s[row] = row < N ? part * SCALE : 0.f;
  };

  // ---- the first item's operands
  Item cur = item_of(item);
  issue_kv(cur);
  issue_slice(cur, 0);
  issue_slice(cur, 1);
  load_lse(item, 0);
  __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0) lgkmcnt(0)
  __syncthreads();

  const f32x4 c4 = {SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E, SCALE_LOG2E}, sc4 = {SCALE, SCALE, SCALE, SCALE};
  // transposed-read addresses (as tr_frag): tile base + row 4 g + q, chunk of column slice dt; + 4096 per 32-row step, + 2048
  // for the second 16 rows, + TILE from the Q tile to the dO tile
  const int tq = (lane >> 2) & 3, tpp = lane & 3;
  const unsigned qbase = (unsigned)(size_t)LDS_PTR(Qt);
  unsigned vrel[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vrel[dt] = qbase + (unsigned)(tile_off(4 * g + tq, 2 * dt + (tpp >> 1)) + 8 * (tpp & 1));
  const int dt_w = wave & 3, qt_w = wave >> 2;        // this wave's dQ tile of every step: (query tile, column slice)
  const unsigned kaddr = vrel[0] - qbase + (unsigned)(size_t)LDS_PTR(Kt) +
                         (unsigned)(tile_off(4 * g + tq, 2 * dt_w + (tpp >> 1)) - tile_off(4 * g + tq, (tpp >> 1)));
  f32x4 kinit[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const float mi = wave * 32 + kt * 16 + li < N ? 0.f : -INFINITY;
    kinit[kt] = f32x4{mi, mi, mi, mi};
  }

  for (int par = 0;; par ^= 1) {
    const int next = item + G;
    const bool has_next = next < nitems;             // workgroup-uniform
    const float* lse_c = lse_s + par * Np;

    // ---- item start: K / V fragments and the K^T fragments of the dQ products from the tiles (they landed during the previous item)
    bf16x8 kf[2][2], vf[2][2];
    if (has_keys) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          kf[kt][kk] = lds_frag(Kt, wave * 32 + kt * 16 + li, kk, lane);
          vf[kt][kk] = lds_frag(Vt, wave * 32 + kt * 16 + li, kk, lane);
        }
    }
    s16x4 klo[NS], khi[NS];
    static_for<NS>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      ds_tr16<4096 * st>(klo[st], kaddr);
      ds_tr16<4096 * st + 2048>(khi[st], kaddr);
    });
    if (delta_wave) delta_slice(0);
    // lgkmcnt(0): the K / V tiles are free behind the barrier.  A raw s_barrier: __syncthreads() would also wait for the previous
    // item's last stores (vmcnt(0)), which step 0's own wait absorbs half a step later.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");

    auto dq_tile = [&](const char* sx, int qs) {
      const char* rowp = sx + (16 * qt_w + li) * SROW + 8 * g;
      u32x2 dlo[NS], dhi[NS];
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        dlo[st] = *(const u32x2*)(rowp + 64 * st);        // keys 32st + 4g .. +3
        dhi[st] = *(const u32x2*)(rowp + 64 * st + 32);   // keys 32st + 16 + 4g .. +3
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const u32x4 w = {dlo[st][0], dlo[st][1], dhi[st][0], dhi[st][1]};
        if (st & 1) acc1 = MFMA16(join_tr(klo[st], khi[st]), __builtin_bit_cast(bf16x8, w), acc1);
        else acc0 = MFMA16(join_tr(klo[st], khi[st]), __builtin_bit_cast(bf16x8, w), acc0);
      }
      const f32x4 acc = acc0 + acc1;
      const int q = qs * 32 + 16 * qt_w + li;
      // unconditional stores through descriptors (rows >= N are out of range and dropped): every wave issues exactly one
      // store per image and call, so the step's wait can be a counted one that leaves them in flight
      // (descriptors of this item's dq rows, [N, 3 H dh] with row stride `stride`, built here from the item's q offset: scalar work)
      const unsigned qoff = (unsigned)q * (unsigned)(stride * 2) + (unsigned)((dt_w * 16 + 4 * g) * 2);
      if (!Q8 || dqkv) {
        __amdgpu_buffer_rsrc_t rsDQ = __builtin_amdgcn_make_buffer_rsrc((void*)(dqkv + cur.qoff), 0, (int)((long long)(N - 1) * stride * 2 + ROWB), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])}, rsDQ, qoff, 0, 0);
      }
      if constexpr (Q8) {
        __amdgpu_buffer_rsrc_t rsDQ8 = __builtin_amdgcn_make_buffer_rsrc((void*)(dq8 + cur.qoff), 0, (int)((long long)(N - 1) * stride + DH), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(pack_fp8x4(acc[0] * qsc, acc[1] * qsc, acc[2] * qsc, acc[3] * qsc), rsDQ8, qoff >> 1, 0, 0);
        if (q < N) qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3]))));
      }
    };

    f32x4 dv[4][2], dk[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        dv[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dk[dt][kt] = dv[dt][kt];
      }

#pragma unroll 1
    for (int qs = 0; qs < NS; ++qs) {
      char* sx = Sx + (qs & 1) * 32 * SROW;
      // (the next item's offsets are worked out inside the two branches that use them, from an opaque copy of its index: hoisted
      // in front of the sweep they -- and the descriptors built from them -- stay alive through every step; this kernel is short of
      // scalar registers, spilled scalars are read back through v_readlane)
      int nz = next;
      asm volatile("" : "+s"(nz));
      if (qs + 2 < NS) issue_slice(cur, qs + 2);
      if (qs == NS - 1 && has_next) {                  // the next item's first two slices and its lse row
        const Item nx = item_of(nz);
        issue_slice(nx, 0);
        issue_slice(nx, 1);
        load_lse(nz, par ^ 1);
      }
      if (delta_wave && qs + 1 < NS) delta_slice(qs + 1);       // slice qs + 1 landed with the wait that ended step qs - 1
      f32x4 p[2][2], ds[2][2];  // [query tile in step][key tile]
      if (has_keys) {
        f32x4 a[2][2], c[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int qrow = qs * 32 + t * 16 + li;
          const bf16x8 qa0 = lds_frag(Qt, qrow, 0, lane), qa1 = lds_frag(Qt, qrow, 1, lane);
          const bf16x8 da0 = lds_frag(Dt, qrow, 0, lane), da1 = lds_frag(Dt, qrow, 1, lane);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            a[t][kt] = kinit[kt];
            c[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            a[t][kt] = MFMA16(qa0, kf[kt][0], a[t][kt]);
            a[t][kt] = MFMA16(qa1, kf[kt][1], a[t][kt]);
            c[t][kt] = MFMA16(da0, vf[kt][0], c[t][kt]);
            c[t][kt] = MFMA16(da1, vf[kt][1], c[t][kt]);
          }
        }
        if (qs > 0) dq_tile(Sx + ((qs - 1) & 1) * 32 * SROW, qs - 1);
        if (qs == 1 && has_next) issue_kv(item_of(nz));   // behind this step's dQ stores: both may stay in flight at its wait
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 nl = -*(const f32x4*)(lse_c + qs * 32 + t * 16 + 4 * g);
          const f32x4 nd = -*(const f32x4*)(del_s + qs * 32 + t * 16 + 4 * g);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            const f32x4 e = __builtin_elementwise_fma(a[t][kt], c4, nl);
#pragma unroll
            for (int r = 0; r < 4; ++r) p[t][kt][r] = __builtin_amdgcn_exp2f(e[r]);
            ds[t][kt] = p[t][kt] * __builtin_elementwise_fma(c[t][kt], sc4, nd);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *(bf16_t*)(sx + (16 * t + 4 * g + r) * SROW + (32 * wave + 16 * kt + li) * 2) = f2bf(ds[t][kt][r]);
          }
        }
      } else {
        if (qs > 0) dq_tile(Sx + ((qs - 1) & 1) * 32 * SROW, qs - 1);
        if (qs == 1 && has_next) issue_kv(item_of(nz));
      }
      // end of the step's first half: this wave's dS columns, its delta rows and its share of slice qs + 2 are complete.  The dQ
      // store(s) of this step -- and, in step 1, the NS requests of the next item's K / V tiles -- were issued AFTER the slice's
      // requests and may stay in flight (counted vmcnt); lgkmcnt(0) covers the LDS writes.  Step 0 waits for everything: the
      // previous item's last stores are older than slice 2's requests and have had the step's first half to be acknowledged.
      const int fly = (qs == 0 ? 0 : dq_stores) + ((qs == 1 && has_next) ? NS : 0);
      switch (fly) {                                    // (immediates: lgkmcnt(0), expcnt untouched, vmcnt = fly)
        case 0: __builtin_amdgcn_s_waitcnt(0x0070); break;
        case 1: __builtin_amdgcn_s_waitcnt(0x0071); break;
        case 2: __builtin_amdgcn_s_waitcnt(0x0072); break;
        case 5: __builtin_amdgcn_s_waitcnt(0x0075); break;
        case 6: __builtin_amdgcn_s_waitcnt(0x0076); break;
        case 7: __builtin_amdgcn_s_waitcnt(0x0077); break;
        case 8: __builtin_amdgcn_s_waitcnt(0x0078); break;
        case 9: __builtin_amdgcn_s_waitcnt(0x0079); break;
        default: __builtin_amdgcn_s_waitcnt(0x0070); break;
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      if (has_keys) {
        // dV^T += dO^T P, dK^T += Q^T dS: the transposed fragments of the step's 32 rows serve both key tiles
        s16x4 qlo[4], qhi[4], dlo[4], dhi[4];
        const unsigned soff = (unsigned)qs * 4096u;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          ds_tr16<TILE>(dlo[dt], vrel[dt] + soff);
          ds_tr16<TILE + 2048>(dhi[dt], vrel[dt] + soff);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          ds_tr16<0>(qlo[dt], vrel[dt] + soff);
          ds_tr16<2048>(qhi[dt], vrel[dt] + soff);
        }
        bf16x8 pf[2], sf[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          pf[kt] = pack_frag(p[0][kt], p[1][kt]);
          sf[kt] = pack_frag(ds[0][kt], ds[1][kt]);
        }
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");      // the dO fragments are in; the Q fragments land under the dV MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 dfr = join_tr(dlo[dt], dhi[dt]);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) dv[dt][kt] = MFMA16(dfr, pf[kt], dv[dt][kt]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 qfr = join_tr(qlo[dt], qhi[dt]);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) dk[dt][kt] = MFMA16(qfr, sf[kt], dk[dt][kt]);
        }
      }
    }
    dq_tile(Sx + ((NS - 1) & 1) * 32 * SROW, NS - 1);
    if (has_keys) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const int key = wave * 32 + kt * 16 + li;
        if (!Q8 || dqkv) {
          // Line-shaped stores (as the NT GEMM epilogue): the 16 x 128-byte tile goes through a private 2 KiB LDS window and leaves
          // as 8 rows x 128 contiguous bytes per instruction.  Windows: rows 64-127 (slices 2 and 3, dead since steps 2 / 3) of the
          // Q tile (waves 0-3) and of the dO tile (waves 4-7); rows 0-63 hold the next item's first slices by now.
          char* xw = (wave < 4 ? Qt : Dt) + 64 * ROWB + (wave & 3) * 2048;
          int lz = lane;                                  // (opaque copy: the store addressing below is formed here, not in front of the sweep)
          asm volatile("" : "+v"(lz));
          const int rho = lz >> 3, kap = lz & 7, gz = lz >> 4, liz = lz & 15;
          const unsigned tw = (unsigned)(liz * 128 + ((gz ^ (2 * (liz >> 1))) << 3));
          const unsigned tr = (unsigned)(rho * 128 + ((kap ^ (rho >> 1)) << 4));
          const int key_a = wave * 32 + kt * 16 + rho;
          bf16_t* dka = dqkv + cur.qoff + (long long)key_a * stride + (long long)H * DH + 8 * kap;
#pragma unroll
          for (int tsel = 0; tsel < 2; ++tsel) {        // dK, then dV
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(u32x2*)(xw + (tw ^ (unsigned)(32 * dt))) = pack4(tsel == 0 ? dk[dt][kt] : dv[dt][kt]);
            const u32x4 s1 = *(const u32x4*)(xw + tr), s2 = *(const u32x4*)(xw + ((tr + 1024u) ^ 64u));
            bf16_t* dst = dka + (long long)tsel * H * DH;
            if (key_a < N) *(u32x4*)dst = s1;
            if (key_a + 8 < N) *(u32x4*)(dst + 8 * stride) = s2;
          }
        }
        if constexpr (Q8) {
          unsigned qk[4], qv[4];
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const f32x4 a = dk[dt][kt], c = dv[dt][kt];
            qk[dt] = pack_fp8x4(a[0] * qsc, a[1] * qsc, a[2] * qsc, a[3] * qsc);
            qv[dt] = pack_fp8x4(c[0] * qsc, c[1] * qsc, c[2] * qsc, c[3] * qsc);
            if (key < N) {
              qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
              qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fmaxf(fabsf(c[2]), fabsf(c[3]))));
            }
          }
          const u32x4 rk = lane_rows_transpose4(qk), rv = lane_rows_transpose4(qv);
          if (key < N) {
            unsigned char* k8 = dq8 + cur.qoff + (long long)key * stride + (long long)H * DH + 16 * g;
            *(u32x4*)k8 = rk;
            *(u32x4*)(k8 + (long long)H * DH) = rv;
          }
        }
      }
    }
    if (!has_next) break;
    item = next;
    cur = item_of(next);
  }
  if (Q8 && qamax) {
    qmax = wave_max(qmax);
    unsigned* slot = (unsigned*)qamax;
    if (lane == 0 && __float_as_uint(qmax) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(qmax));
  }
}

// CUs the launches of this file size themselves for: the device's count minus the reserve in force (vitssl_set_reserved_cus).  The
// persistent forward holds 147-160 KiB of LDS and 8 waves x 256 VGPRs per workgroup, i.e. a whole CU for the whole launch, so its grid
// must leave the reserved CUs to the collective library like the GEMM grids do (round-3 advisor finding: it used the raw count).
int attn_cu_count() { return vitssl_persistent_cus(); }
std::atomic<int> g_last_attn_fwd_grid{0};

// stagger window in 10 ns ticks: VITSSL_ATTN_STAGGER_FWD / _BWD (developer knobs; 0 disables)
int attn_stagger_ticks(int which) {
  static VsEnvInt fwd, bwd;
  // forward: two workgroups per CU already overlap each other's loads (stagger measured -2 %)
  // fused backward (one workgroup per CU): prologue 6.5 -> 4.9 us per workgroup, launch 269 -> 265 us
  return which == 0 ? fwd.get("VITSSL_ATTN_STAGGER_FWD", 0) : bwd.get("VITSSL_ATTN_STAGGER_BWD", 1400);
}

template <typename K>
int ensure_lds(K kernel, int bytes, VsOnce* done, const char* who) {
  if (done->load(std::memory_order_relaxed) || bytes <= 48 * 1024) return VITSSL_OK;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) {
    vitssl_set_error("%s: cannot raise dynamic LDS: %s", who, hipGetErrorString(e));
    return VITSSL_ERR_LAUNCH;
  }
  done->store(true, std::memory_order_relaxed);
  return VITSSL_OK;
}

// 1 (default): sequences of more than 128 tokens run the persistent forward (8 waves x two query tiles, next item's K / V prefetched);
// 0: the two-workgroups-per-CU kernel that shorter sequences always use (VITSSL_ATTN_FWD_PERSIST; exercised by tests/test_gpu_knobs.py)
bool attn_fwd_persist() {
  static VsEnvInt knob;
  return knob.get("VITSSL_ATTN_FWD_PERSIST", 1) != 0;
}

template <int NS>
int launch_fwd(const bf16_t* qkv, bf16_t* out, float* lse, float* probs, int B, int N, int H, hipStream_t s,
               unsigned char* out8 = nullptr) {
  if constexpr (NS >= 5) {
    if (attn_fwd_persist()) {
      static VsOnce done_p{false};
      const int lds_p = 4 * NS * 32 * ROWB + 8 * 2048;   // two (K, V) tile pairs + a 2 KiB output window per wave
      const int nitems = B * H;
      const int cus = attn_cu_count();
      const int grid = nitems < cus ? nitems : cus;
      if (int rc = ensure_lds(attn_fwd_pers_kernel<NS, 2>, lds_p, &done_p, "attn_fwd_pers")) return rc;
      g_last_attn_fwd_grid.store(grid, std::memory_order_relaxed);
      hipLaunchKernelGGL((attn_fwd_pers_kernel<NS, 2>), dim3(grid), dim3(512), lds_p, s, qkv, out, lse, probs, N, H, nitems, out8);
      VS_CHECK_LAUNCH("attn_fwd_pers");
      return VITSSL_OK;
    }
  }
  static VsOnce done{false};
  const int lds = 2 * NS * 32 * ROWB;
  constexpr int NW = NS <= 2 ? 2 : 4;
  if (int rc = ensure_lds(attn_fwd_kernel<NS, NW>, lds, &done, "attn_fwd")) return rc;
  const int per_cu = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3);
  const int cus = attn_cu_count();
  g_last_attn_fwd_grid.store(B * H, std::memory_order_relaxed);
  hipLaunchKernelGGL((attn_fwd_kernel<NS, NW>), dim3(B * H), dim3(64 * NW), lds, s, qkv, out, lse, probs, N, H,
                     cus * per_cu, B * H > 2 * cus * per_cu ? attn_stagger_ticks(0) : 0, out8);
  VS_CHECK_LAUNCH("attn_fwd");
  return VITSSL_OK;
}

// 1 (default): sequences of more than 128 tokens run the fused backward with the pipelined prologue; 0: attn_bwd_fused_kernel, which
// shorter sequences always use (VITSSL_ATTN_BWD_PIPE; exercised by tests/test_gpu_knobs.py)
bool attn_bwd_pipe() {
  static VsEnvInt knob;
  return knob.get("VITSSL_ATTN_BWD_PIPE", 1) != 0;
}

// 1 (default): sequences of 129-224 tokens run the persistent backward (next item's operands prefetched); 0: one workgroup per
// (batch, head) with the pipelined prologue (VITSSL_ATTN_BWD_PERSIST; exercised by tests/test_gpu_knobs.py)
bool attn_bwd_persist() {
  static VsEnvInt knob;
  return knob.get("VITSSL_ATTN_BWD_PERSIST", 1) != 0;
}

template <int NS>
int launch_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, int B,
               int N, int H, hipStream_t s, unsigned char* dq8 = nullptr, const float* qscale = nullptr, float* qamax = nullptr) {
  const int cus = attn_cu_count();
  if constexpr (NS >= 5 && NS <= 7) {
    if (attn_bwd_pipe() && attn_bwd_persist()) {
      static VsOnce done_ps{false}, done_ps8{false};
      constexpr int LDS_PS = 4 * NS * 32 * ROWB + 2 * 32 * (NS * 64 + 16) + 3 * NS * 32 * 4 + 3 * 32 * ROWB;
      static_assert(LDS_PS <= 160 * 1024, "the persistent backward needs four operand tiles in LDS");
      const int nitems = B * H;
      const int grid = nitems < cus ? nitems : cus;
      if (dq8) {
        if (int rc = ensure_lds(attn_bwd_pers_kernel<NS, true>, LDS_PS, &done_ps8, "attn_bwd_pers")) return rc;
        hipLaunchKernelGGL((attn_bwd_pers_kernel<NS, true>), dim3(grid), dim3(512), LDS_PS, s, qkv, out, dout, lse, dqkv, N, H, nitems, dq8,
                           qscale, qamax);
      } else {
        if (int rc = ensure_lds(attn_bwd_pers_kernel<NS, false>, LDS_PS, &done_ps, "attn_bwd_pers")) return rc;
        hipLaunchKernelGGL((attn_bwd_pers_kernel<NS, false>), dim3(grid), dim3(512), LDS_PS, s, qkv, out, dout, lse, dqkv, N, H, nitems, dq8,
                           qscale, qamax);
      }
      VS_CHECK_LAUNCH("attn_bwd_pers");
      return VITSSL_OK;
    }
  }
  if constexpr (NS >= 5) {
    if (attn_bwd_pipe()) {
      static VsOnce done_p{false};
      const int lds_p = 3 * NS * 32 * ROWB + 2 * 32 * (NS * 64 + 16) + 2 * NS * 32 * 4 + 3 * 32 * ROWB;
      if (int rc = ensure_lds(attn_bwd_pipe_kernel<NS>, lds_p, &done_p, "attn_bwd_pipe")) return rc;
      hipLaunchKernelGGL((attn_bwd_pipe_kernel<NS>), dim3(B * H), dim3(512), lds_p, s, qkv, out, dout, lse, dqkv, N, H,
                         cus, B * H > 2 * cus ? attn_stagger_ticks(1) : 0, dq8, qscale, qamax);
      VS_CHECK_LAUNCH("attn_bwd_pipe");
      return VITSSL_OK;
    }
  }
  static VsOnce done_f{false};
  constexpr int NW = NS <= 2 ? 2 : (NS <= 4 ? 4 : 8);
  const int lds_f = 3 * NS * 32 * ROWB + 2 * 32 * (NS * 64 + 16) + 2 * NS * 32 * 4;
  if (int rc = ensure_lds(attn_bwd_fused_kernel<NS, NW>, lds_f, &done_f, "attn_bwd_fused")) return rc;
  const int per_cu = lds_f > 80 * 1024 ? 1 : 2;
  hipLaunchKernelGGL((attn_bwd_fused_kernel<NS, NW>), dim3(B * H), dim3(64 * NW), lds_f, s, qkv, out, dout, lse, dqkv, N, H,
                     cus * per_cu, B * H > 2 * cus * per_cu ? attn_stagger_ticks(1) : 0, dq8, qscale, qamax);
  VS_CHECK_LAUNCH("attn_bwd_fused");
  return VITSSL_OK;
}

#define VS_NS_SWITCH(NSV, CALL)        \
  switch (NSV) {                       \
    case 1: return CALL(1);            \
    case 2: return CALL(2);            \
    case 3: return CALL(3);            \
    case 4: return CALL(4);            \
    case 5: return CALL(5);            \
    case 6: return CALL(6);            \
    case 7: return CALL(7);            \
    default: return CALL(8);           \
  }

int check_attn_shape(const char* who, int B, int N, int H, int dh) {
  VS_CHECK_ARG(B > 0 && N > 0 && H > 0, "%s: empty problem B=%d N=%d H=%d", who, B, N, H);
  VS_CHECK_ARG(dh == DH, "%s: head dim %d unsupported (this build handles dh=64: ViT-Tiny/S/B/L)", who, dh);
  VS_CHECK_ARG(N <= 256, "%s: sequence length %d > 256 unsupported", who, N);
  return VITSSL_OK;
}

}  // namespace

#ifdef VITSSL_ATTN_STAMPS
extern "C" int vitssl_debug_attn_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &p, sizeof(p));
}
#endif

extern "C" int vitssl_debug_last_attn_fwd_grid(void) { return g_last_attn_fwd_grid.load(std::memory_order_relaxed); }

extern "C" int vitssl_attn_fwd(const void* qkv, void* out, float* lse, float* probs, int B, int N, int H, int dh,
                               void* stream) {
  VS_CHECK_ARG(qkv && out && lse, "attn_fwd: null pointer");
  if (int rc = check_attn_shape("attn_fwd", B, N, H, dh)) return rc;
#define VS_CALL(NS) launch_fwd<NS>((const bf16_t*)qkv, (bf16_t*)out, lse, probs, B, N, H, (hipStream_t)stream)
  VS_NS_SWITCH((N + 31) / 32, VS_CALL)
#undef VS_CALL
}

extern "C" int vitssl_attn_fwd_fp8(const void* qkv, void* out, void* out_fp8, float* lse, float* probs, int B, int N, int H,
                                   int dh, void* stream) {
  VS_CHECK_ARG(qkv && out && out_fp8 && lse, "attn_fwd_fp8: null pointer");
  if (int rc = check_attn_shape("attn_fwd_fp8", B, N, H, dh)) return rc;
#define VS_CALL(NS) launch_fwd<NS>((const bf16_t*)qkv, (bf16_t*)out, lse, probs, B, N, H, (hipStream_t)stream, (unsigned char*)out_fp8)
  VS_NS_SWITCH((N + 31) / 32, VS_CALL)
#undef VS_CALL
}

extern "C" int vitssl_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                               float* delta_ws, int B, int N, int H, int dh, void* stream) {
  (void)delta_ws;   // kept in the signature for ABI stability: the fused kernels compute rowsum(dO * O) in their prologue
  VS_CHECK_ARG(qkv && out && dout && lse && dqkv, "attn_bwd: null pointer");
  if (int rc = check_attn_shape("attn_bwd", B, N, H, dh)) return rc;
  hipStream_t s = (hipStream_t)stream;
#define VS_CALL(NS) launch_bwd<NS>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, B, N, H, s)
  VS_NS_SWITCH((N + 31) / 32, VS_CALL)
#undef VS_CALL
}

extern "C" int vitssl_attn_bwd_fp8(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, void* dqkv_fp8,
                                   const float* qscale, float* qamax, int B, int N, int H, int dh, void* stream) {
  VS_CHECK_ARG(qkv && out && dout && lse && dqkv_fp8, "attn_bwd_fp8: null pointer");   // dqkv (bf16) may be NULL
  if (int rc = check_attn_shape("attn_bwd_fp8", B, N, H, dh)) return rc;
  hipStream_t s = (hipStream_t)stream;
#define VS_CALL(NS) launch_bwd<NS>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, B, N, H, s, \
                                   (unsigned char*)dqkv_fp8, qscale, qamax)
  VS_NS_SWITCH((N + 31) / 32, VS_CALL)
#undef VS_CALL
}
