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
// Grid = tiles(N1/256) x tiles(N2/256) x splits(M); each workgroup accumulates its
// M-range into a 256x256 fp32 tile in registers and combines with fp32 atomics
// (64-byte contiguous segments per 16-lane group).  Rows past M are zero-filled by the
// buffer descriptor's bounds check, so ragged M needs no tail code.
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
  int bid = blockIdx.x;
  const int ntiles = p.tiles1 * p.tiles2;
  const int split = bid / ntiles;
  const int tile = bid - split * ntiles;
  const int t1 = tile / p.tiles2, t2 = tile - t1 * p.tiles2;
  const int c1 = t1 * TN_T, c2 = t2 * TN_T;

  const long long total_chunks = (p.M + TN_KM - 1) / TN_KM;
  const long long ch_begin = (long long)split * p.chunks_per_split;
  long long ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  if (ch_begin >= ch_end) return;
  const int nk = (int)(ch_end - ch_begin);

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((unsigned long long)p.M * p.N1 * 2ull), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((unsigned long long)p.M * p.N2 * 2ull), 0x00020000);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  tn_stage(rsA, smem, ch_begin * TN_KM, c1, p.N1, wave, lane);
  tn_stage(rsB, smem + TN_TILE_BYTES, ch_begin * TN_KM, c2, p.N2, wave, lane);
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // D[n1 = 4*(lane>>4)+r][n2 = lane&15]
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n1 = c1 + w1 * 128 + i * 16 + 4 * (lane >> 4) + r;
      if (n1 >= p.N1) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n2 = c2 + w2 * 64 + j * 16 + (lane & 15);
        if (n2 < p.N2) unsafeAtomicAdd(p.C + (long long)n1 * p.N2 + n2, acc[i][j][r]);
      }
    }
  }
}

}  // namespace

extern "C" int vitssl_gemm_bf16_tn(const void* A, const void* B, float* C, int64_t M, int N1, int N2, void* stream) {
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
  p.tiles1 = (N1 + TN_T - 1) / TN_T;
  p.tiles2 = (N2 + TN_T - 1) / TN_T;
  const long long total_chunks = (M + TN_KM - 1) / TN_KM;
  const int ntiles = p.tiles1 * p.tiles2;
  long long splits = (512 + ntiles - 1) / ntiles;      // aim at ~2 workgroups per CU in flight
  if (splits > total_chunks) splits = total_chunks;
  if (splits < 1) splits = 1;
  p.chunks_per_split = (int)((total_chunks + splits - 1) / splits);
  p.splits = (int)((total_chunks + p.chunks_per_split - 1) / p.chunks_per_split);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_tn: cannot raise dynamic LDS: %s", hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(ntiles * p.splits), dim3(TN_THREADS), TN_LDS_BYTES, (hipStream_t)stream, p);
  VS_CHECK_LAUNCH("gemm_tn");
  return VITSSL_OK;
}
