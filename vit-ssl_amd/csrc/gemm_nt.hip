// bf16 MFMA GEMM  C[M,N] = A[M,K] . B[N,K]^T  with fused epilogues (gfx950).
//
// Tile 256x256x64, 512 threads = 8 waves as 2(M) x 4(N), each wave owns a 128x64
// sub-tile = 8x4 accumulators of v_mfma_f32_16x16x32_bf16.  The MFMA is issued with
// the B-tile fragment as its A operand and the A-tile fragment as its B operand, so a
// lane ends up holding 4 CONSECUTIVE output columns of one output row
// (D[n = 4*(lane>>4)+r][m = lane&15]) -> 8-byte bf16 / 16-byte fp32 row-contiguous
// epilogue accesses with no LDS transpose.
//
// Operand staging is LDS-DMA (buffer_load_dwordx4 ... lds): each wave-instruction lands
// 1 KiB = 8 tile rows x 128 B linearly in LDS; the 16-byte chunk a lane FETCHES is
// XOR-swizzled on the SOURCE side (chunk ^ ((row>>1)&7)) and the same XOR is applied
// when fragments are read, which makes every 16-lane ds_read_b128 group hit 16
// distinct 16-byte slots of the 256-byte bank row (conflict-free).  The hardware
// bounds check of the buffer descriptor zero-fills rows past the end of A / B, so
// ragged M and N need no special staging code.
//
// Pipeline: 2 LDS buffers; the stage of k-tile t+1 is issued before the MFMAs of
// k-tile t and retired (vmcnt(0) + barrier) after them.
#include "common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NT_THREADS = 512;
constexpr int TILE_BYTES = 256 * BK * 2;          // one operand tile
constexpr int NT_LDS_BYTES = 4 * TILE_BYTES;      // 2 buffers x (A, B) = 128 KiB

struct NtParams {
  const bf16_t* A;
  const bf16_t* B;
  long long M;
  int N, K;
  const float* bias;
  const void* aux;
  void* out0;
  void* out1;
  float* colsum;
  DropKey dk;
  int drop_on;
  vitssl_embed_t embed;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, long long row0, int k0,
                                           int K, int wave, int lane) {
  // 32 wave-instructions cover the 256-row tile; this wave issues 4 of them.
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wave * 4 + j;                 // wave-uniform instruction slot
    const int r = i * 8 + (lane >> 3);          // tile row this lane fetches for
    const int c = lane & 7;                     // 16-B chunk position in the LDS row
    const int sc = c ^ ((r >> 1) & 7);          // chunk fetched from global
    const unsigned voff = (unsigned)(((row0 + r) * (long long)K + k0) * 2 + sc * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + i * 1024), 16, voff, 0, 0, 0);
  }
}

template <int EPI>
__global__ __launch_bounds__(NT_THREADS) void gemm_nt_kernel(NtParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (round-robin dispatch);
  // give each XCD a contiguous run of tiles so neighbouring tiles (same A row-panel,
  // weights) hit the same L2.  Speed only, never correctness.
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  const int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  const int tile_m = wgid / p.tiles_n;
  const int tile_n = wgid - tile_m * p.tiles_n;
  const long long m0 = (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const unsigned long long a_bytes = (unsigned long long)p.M * p.K * 2ull;
  const unsigned long long b_bytes = (unsigned long long)p.N * p.K * 2ull;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)b_bytes, 0x00020000);

  f32x4 acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_tile(rsA, smem, m0, 0, p.K, wave, lane);
  stage_tile(rsB, smem + TILE_BYTES, n0, 0, p.K, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int swz = (lane >> 1) & 7;                 // (row>>1)&7 with row = 16*x + (lane&15)
  const int frag_row = lane & 15;
  const int kq = lane >> 4;

  for (int t = 0; t < nk; ++t) {
    char* bufA = smem + (t & 1) * 2 * TILE_BYTES;
    char* bufB = bufA + TILE_BYTES;
    if (t + 1 < nk) {
      char* nA = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
      stage_tile(rsA, nA, m0, (t + 1) * BK, p.K, wave, lane);
      stage_tile(rsB, nA + TILE_BYTES, n0, (t + 1) * BK, p.K, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + kq) ^ swz) << 4;
      bf16x8 fb[4], fa[8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fb[j] = *(const bf16x8*)(bufB + (wn * 64 + j * 16 + frag_row) * 128 + coff);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        fa[i] = *(const bf16x8*)(bufA + (wm * 128 + i * 16 + frag_row) * 128 + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  float csum[4][4];
  if (p.colsum) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) csum[j][r] = 0.f;
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
    if (n >= p.N) continue;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *(const f32x4*)(p.bias + n);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long m = m0 + wm * 128 + i * 16 + (lane & 15);
      if (m >= p.M) continue;
      f32x4 v = acc[j][i] + bias4;
      const long long e = m * p.N + n;

      if constexpr (EPI == VITSSL_EPI_BF16) {
        u32x2 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *(u32x2*)((bf16_t*)p.out0 + e) = w;
      } else if constexpr (EPI == VITSSL_EPI_F32) {
        *(f32x4*)((float*)p.out0 + e) = v;
      } else if constexpr (EPI == VITSSL_EPI_GELU) {
        u32x2 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *(u32x2*)((bf16_t*)p.out0 + e) = w;
        float mult[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop_on) drop_mult4(p.dk, (unsigned long long)e >> 2, mult);
        float g0 = gelu_f(bf_lo(w[0])) * mult[0], g1 = gelu_f(bf_hi(w[0])) * mult[1];
        float g2 = gelu_f(bf_lo(w[1])) * mult[2], g3 = gelu_f(bf_hi(w[1])) * mult[3];
        u32x2 a = {pack_bf2(g0, g1), pack_bf2(g2, g3)};
        *(u32x2*)((bf16_t*)p.out1 + e) = a;
      } else if constexpr (EPI == VITSSL_EPI_RESID) {
        if (p.drop_on) {
          float mult[4];
          drop_mult4(p.dk, (unsigned long long)e >> 2, mult);
          v[0] *= mult[0]; v[1] *= mult[1]; v[2] *= mult[2]; v[3] *= mult[3];
        }
        const f32x4 res = *(const f32x4*)((const float*)p.aux + e);
        v += res;
        *(f32x4*)((float*)p.out0 + e) = v;
      } else if constexpr (EPI == VITSSL_EPI_DGELU) {
        const u32x2 u = *(const u32x2*)((const bf16_t*)p.aux + e);
        float mult[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop_on) drop_mult4(p.dk, (unsigned long long)e >> 2, mult);
        v[0] *= mult[0] * dgelu_f(bf_lo(u[0]));
        v[1] *= mult[1] * dgelu_f(bf_hi(u[0]));
        v[2] *= mult[2] * dgelu_f(bf_lo(u[1]));
        v[3] *= mult[3] * dgelu_f(bf_hi(u[1]));
        u32x2 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *(u32x2*)((bf16_t*)p.out0 + e) = w;
      } else if constexpr (EPI == VITSSL_EPI_EMBED) {
        const long long img = m / p.embed.tokens;
        const int rin = (int)(m - img * p.embed.tokens);
        if (p.embed.mask && p.embed.mask[m]) v = *(const f32x4*)(p.embed.mask_token + n);
        const f32x4 pos = *(const f32x4*)(p.embed.pos + (long long)(p.embed.tok_offset + rin) * p.N + n);
        v += pos;
        const long long orow = img * p.embed.out_tokens + p.embed.tok_offset + rin;
        *(f32x4*)((float*)p.out0 + orow * p.N + n) = v;
      }
      if (p.colsum) {
#pragma unroll
        for (int r = 0; r < 4; ++r) csum[j][r] += v[r];
      }
    }
  }

  if (p.colsum) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = csum[j][r];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4) + r;
        if ((lane & 15) == 0 && n < p.N) atomicAdd(p.colsum + n, s);
      }
    }
  }
}

template <int EPI>
int launch_nt(const NtParams& p, hipStream_t s) {
  static bool attr_done = false;  // idempotent; a benign race sets the same value
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_nt: cannot raise dynamic LDS to %d: %s", NT_LDS_BYTES, hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(p.tiles_m * p.tiles_n), dim3(NT_THREADS), NT_LDS_BYTES, s, p);
  VS_CHECK_LAUNCH("gemm_nt");
  return VITSSL_OK;
}

}  // namespace

extern "C" int vitssl_gemm_bf16_nt(const vitssl_gemm_t* g, void* stream) {
  VS_CHECK_ARG(g && g->A && g->B && g->out0, "gemm_nt: null operand");
  VS_CHECK_ARG(g->M > 0 && g->N > 0 && g->K > 0, "gemm_nt: empty problem M=%lld N=%d K=%d", (long long)g->M, g->N, g->K);
  VS_CHECK_ARG(g->K % BK == 0, "gemm_nt: K=%d must be a multiple of %d", g->K, BK);
  VS_CHECK_ARG(g->N % 4 == 0, "gemm_nt: N=%d must be a multiple of 4", g->N);
  VS_CHECK_ARG((unsigned long long)g->M * g->K * 2ull < (1ull << 31) && (unsigned long long)g->N * g->K * 2ull < (1ull << 31),
               "gemm_nt: operand larger than 2 GiB (M=%lld N=%d K=%d)", (long long)g->M, g->N, g->K);
  NtParams p;
  p.A = (const bf16_t*)g->A;
  p.B = (const bf16_t*)g->B;
  p.M = g->M;
  p.N = g->N;
  p.K = g->K;
  p.bias = g->bias;
  p.aux = g->aux;
  p.out0 = g->out0;
  p.out1 = g->out1;
  p.colsum = g->colsum;
  p.dk = make_drop_key(g->drop);
  p.drop_on = p.dk.thr != 0;
  p.embed = g->embed;
  p.tiles_m = (int)ceil_div64(g->M, BM);
  p.tiles_n = (int)ceil_div64(g->N, BN);
  hipStream_t s = (hipStream_t)stream;
  switch (g->epilogue) {
    case VITSSL_EPI_BF16: return launch_nt<VITSSL_EPI_BF16>(p, s);
    case VITSSL_EPI_F32: return launch_nt<VITSSL_EPI_F32>(p, s);
    case VITSSL_EPI_GELU:
      VS_CHECK_ARG(g->out1, "gemm_nt: EPI_GELU needs out1");
      return launch_nt<VITSSL_EPI_GELU>(p, s);
    case VITSSL_EPI_RESID:
      VS_CHECK_ARG(g->aux, "gemm_nt: EPI_RESID needs aux (residual)");
      return launch_nt<VITSSL_EPI_RESID>(p, s);
    case VITSSL_EPI_DGELU:
      VS_CHECK_ARG(g->aux, "gemm_nt: EPI_DGELU needs aux (pre-activation)");
      return launch_nt<VITSSL_EPI_DGELU>(p, s);
    case VITSSL_EPI_EMBED:
      VS_CHECK_ARG(g->embed.pos && g->embed.tokens > 0 && g->embed.out_tokens >= g->embed.tokens + g->embed.tok_offset,
                   "gemm_nt: EPI_EMBED needs pos/tokens");
      VS_CHECK_ARG(!g->embed.mask || g->embed.mask_token, "gemm_nt: EPI_EMBED mask without mask_token");
      return launch_nt<VITSSL_EPI_EMBED>(p, s);
    default:
      vitssl_set_error("gemm_nt: unknown epilogue %d", g->epilogue);
      return VITSSL_ERR_ARG;
  }
}
