// Weight-gradient GEMM  C[N1,N2] (fp32) += A[M,N1]^T . B[M,N2]   (gfx950, bf16 MFMA).
//
// Both operands have the contraction index m as their ROW index, so the MFMA fragments
// (8 consecutive k per lane) are columns of the staged tiles.  Tiles are staged
// row-major exactly as they sit in HBM (coalesced 512-B rows via LDS-DMA) and read
// back column-major with ds_read_b64_tr_b16 (hardware transpose read, 4 rows x 16
// columns per 16-lane group).  A source-side XOR of the 32-byte chunk index with
// f(row) = (row&3) | ((row>>3)&1)<<2 makes every 32-lane half of a transposed read
// touch 8 rows x 32 B = one full 256-B bank row (conflict-free).
//
// Grid = tiles(N1/256) x tiles(N2/256) x splits(M), sized to ONE round of the 256 CUs;
// each workgroup accumulates its M-range into a 256x256 fp32 tile in registers.  The
// MFMA operands are swapped (D[n2][n1]) so a lane owns 4 consecutive n2: partial tiles go
// to a per-split slab with plain 16-byte stores and a streaming reduce kernel adds the
// slabs into C (fp32 atomics cost ~50 us per workgroup at the chip's 1.3 TB/s atomic
// rate; they remain as the fallback when no workspace is given).  Rows past M are
// zero-filled by the buffer descriptor's bounds check, so ragged M needs no tail code.
#include "common.h"

namespace {

constexpr int TN_T = 256;          // output tile edge
constexpr int TN_KM = 64;          // contraction rows per stage
constexpr int TN_THREADS = 512;
constexpr int TN_TILE_BYTES = TN_KM * TN_T * 2;   // 32 KiB
constexpr int TN_LDS_BYTES = 4 * TN_TILE_BYTES;   // 128 KiB

struct TnParams {
  const bf16_t* A;
  const bf16_t* B;
  float* C;
  long long M;
  int N1, N2;
  int tiles1, tiles2, splits;
  int chunks_per_split;   // in units of TN_KM rows
  float* slabs;           // [splits][N1][N2] fp32 or nullptr (atomic mode)
};

__device__ __forceinline__ int tn_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// stage rows [mrow0, mrow0+64) x cols [col0, col0+256) of X[M, ld] into lds_tile
__device__ __forceinline__ void tn_stage(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, long long mrow0, int col0,
                                         int ld, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wave * 4 + j;                  // instruction slot: rows 2i, 2i+1
    const int row = i * 2 + (lane >> 5);
    const int ch16 = lane & 31;                  // 16-B chunk position in the LDS row
    const int sc32 = (ch16 >> 1) ^ tn_f(row);    // 32-B chunk fetched from global
    const int sch16 = (sc32 << 1) | (ch16 & 1);
    const unsigned voff = (unsigned)(((mrow0 + row) * (long long)ld + col0) * 2 + sch16 * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + i * 1024), 16, voff, 0, 0, 0);
  }
}

// transposed fragment: 8 consecutive m (rows ks*32 + 8g + 0..7) of column c0 + (lane&15)
__device__ __forceinline__ bf16x8 tn_frag(const char* tile, int ks, int c0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int c32 = c0 >> 4;
  const int r0 = ks * 32 + 8 * g + q;
  const int r1 = r0 + 4;
  const char* a0 = tile + r0 * 512 + ((c32 ^ tn_f(r0)) << 5) + 8 * pp;
  const char* a1 = tile + r1 * 512 + ((c32 ^ tn_f(r1)) << 5) + 8 * pp;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w1 = wave >> 2, w2 = wave & 3;

  // block -> (split, tile1, tile2); splits of one tile are spread over XCDs (they share
  // nothing), tiles sharing an operand panel are adjacent.
  // XCD-aware bijective remap (blocks b, b+8, .. share an XCD): the tiles of one split,
  // which share the split's dY / X row slabs, become neighbours on one XCD's L2.
  const int ntiles = p.tiles1 * p.tiles2;
  const int nwg = ntiles * p.splits;
  const int xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
  const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
  const int split = bid / ntiles;
  const int tile = bid - split * ntiles;
  const int t1 = tile / p.tiles2, t2 = tile - t1 * p.tiles2;
  const int c1 = t1 * TN_T, c2 = t2 * TN_T;

  const long long total_chunks = (p.M + TN_KM - 1) / TN_KM;
  const long long ch_begin = (long long)split * p.chunks_per_split;
  long long ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  const int nk = ch_begin < ch_end ? (int)(ch_end - ch_begin) : 0;   // empty split: writes a zero slab

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((unsigned long long)p.M * p.N1 * 2ull), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((unsigned long long)p.M * p.N2 * 2ull), 0x00020000);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    tn_stage(rsA, smem, ch_begin * TN_KM, c1, p.N1, wave, lane);
    tn_stage(rsB, smem + TN_TILE_BYTES, ch_begin * TN_KM, c2, p.N2, wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const char* bufA = smem + (t & 1) * 2 * TN_TILE_BYTES;
    const char* bufB = bufA + TN_TILE_BYTES;
    if (t + 1 < nk) {
      char* nA = smem + ((t + 1) & 1) * 2 * TN_TILE_BYTES;
      tn_stage(rsA, nA, (ch_begin + t + 1) * TN_KM, c1, p.N1, wave, lane);
      tn_stage(rsB, nA + TN_TILE_BYTES, (ch_begin + t + 1) * TN_KM, c2, p.N2, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[8], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = tn_frag(bufB, ks, w2 * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = tn_frag(bufA, ks, w1 * 128 + i * 16, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // D[n2 = 4*(lane>>4)+r][n1 = lane&15]: 4 consecutive n2 of one C row per lane
  float* dst = p.slabs ? p.slabs + (long long)split * p.N1 * p.N2 : p.C;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n1 = c1 + w1 * 128 + i * 16 + (lane & 15);
    if (n1 >= p.N1) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n2 = c2 + w2 * 64 + j * 16 + 4 * (lane >> 4);
      if (n2 >= p.N2) continue;
      float* q = dst + (long long)n1 * p.N2 + n2;
      if (p.slabs) {
        *(f32x4*)q = acc[i][j];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(q + r, acc[i][j][r]);
      }
    }
  }
}

// C[e] += sum_s slabs[s][e]
__global__ void tn_reduce_kernel(float* __restrict__ C, const float* __restrict__ slabs, long long n4, long long stride, int splits) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 a = *(const f32x4*)(C + 4 * i);
    for (int s = 0; s < splits; ++s) a += *(const f32x4*)(slabs + s * stride + 4 * i);
    *(f32x4*)(C + 4 * i) = a;
  }
}

void tn_plan(long long M, int N1, int N2, int* tiles1, int* tiles2, int* splits, int* chunks_per_split) {
  *tiles1 = (N1 + TN_T - 1) / TN_T;
  *tiles2 = (N2 + TN_T - 1) / TN_T;
  const long long total_chunks = (M + TN_KM - 1) / TN_KM;
  const int ntiles = *tiles1 * *tiles2;
  long long sp = 256 / ntiles;                 // one round of the 256 CUs
  if (sp < 1) sp = 1;
  if (sp > total_chunks) sp = total_chunks;
  *chunks_per_split = (int)((total_chunks + sp - 1) / sp);
  *splits = (int)((total_chunks + *chunks_per_split - 1) / *chunks_per_split);
}

}  // namespace

extern "C" int64_t vitssl_gemm_tn_workspace_floats(int64_t M, int N1, int N2) {
  if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
  int t1, t2, sp, cps;
  tn_plan(M, N1, N2, &t1, &t2, &sp, &cps);
  return (int64_t)sp * N1 * N2;
}

extern "C" int vitssl_gemm_bf16_tn(const void* A, const void* B, float* C, int64_t M, int N1, int N2, float* workspace,
                                   int64_t workspace_floats, void* stream) {
  VS_CHECK_ARG(A && B && C, "gemm_tn: null operand");
  VS_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0, "gemm_tn: empty problem");
  VS_CHECK_ARG(N1 % 8 == 0 && N2 % 8 == 0, "gemm_tn: N1=%d N2=%d must be multiples of 8", N1, N2);
  VS_CHECK_ARG((unsigned long long)M * N1 * 2ull < (1ull << 31) && (unsigned long long)M * N2 * 2ull < (1ull << 31),
               "gemm_tn: operand larger than 2 GiB");
  TnParams p;
  p.A = (const bf16_t*)A;
  p.B = (const bf16_t*)B;
  p.C = C;
  p.M = M;
  p.N1 = N1;
  p.N2 = N2;
  tn_plan(M, N1, N2, &p.tiles1, &p.tiles2, &p.splits, &p.chunks_per_split);
  const long long need = (long long)p.splits * N1 * N2;
  p.slabs = (workspace && workspace_floats >= need) ? workspace : nullptr;
  VS_CHECK_ARG(!workspace || p.slabs, "gemm_tn: workspace too small (%lld < %lld floats)", (long long)workspace_floats, need);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_tn: cannot raise dynamic LDS: %s", hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(p.tiles1 * p.tiles2 * p.splits), dim3(TN_THREADS), TN_LDS_BYTES, s, p);
  VS_CHECK_LAUNCH("gemm_tn");
  if (p.slabs) {
    const long long n4 = (long long)N1 * N2 / 4;
    long long grid = (n4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, s, C, p.slabs, n4, (long long)N1 * N2, p.splits);
    VS_CHECK_LAUNCH("gemm_tn_reduce");
  }
  return VITSSL_OK;
}
