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
#include <stdlib.h>
#include <type_traits>
#include "common.h"

#ifdef VITSSL_TN_STAMPS
__device__ unsigned long long* g_tn_stamps = nullptr;     // [workgroup][wave group][8]: six 10 ns segment sums, K-tiles, shader clocks of the loop
extern "C" int vitssl_debug_set_tn_stamps(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tn_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

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
  float* slabs;           // [splits][N1][N2] fp32 or nullptr (atomic mode, or direct mode)
  int direct;             // splits == 1: every C tile has exactly one owner, which adds its result into C itself (no slab, no reduce pass)
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
      } else if (p.direct) {
        *(f32x4*)q = *(const f32x4*)q + acc[i][j];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(q + r, acc[i][j][r]);
      }
    }
  }
}

// ====================================================================================
// Ping-pong main loop for the weight-gradient GEMM (same scheme as gemm_nt_pp_kernel, see there).
// A K-tile (64 contraction rows) is multiplied in 4 phases of 16 MFMAs: (ks, n1-half) =
// (0,0) (0,1) (1,0) (1,1); waves 4-7 run one barrier behind waves 0-3, so each SIMD always has one
// wave in its MFMA cluster and one reading fragments / issuing DMA.  Staging units are the 32-row
// halves of the two operand tiles (16 KiB = 2 wave-instructions per wave):
//     p0: Bk1(t+1)   p1: Ak1(t+1) + vmcnt(8)   p2: Bk0(t+2)   p3: Ak0(t+2) + vmcnt(8)
// (a half is refilled two phases after its last fragment read: Bk0 is read in p0, Ak0 in p0-p1, Bk1 in
// p2, Ak1 in p2-p3; the wait covering a half sits in the phase before its first read).
typedef __attribute__((ext_vector_type(8))) short tn_s16x8;
// diagnostic builds only (tools/build_variant.sh), bit mask: 1 = no DMA inside the K loop, 2 = no fragment reads, 4 = no MFMA
#ifndef TN_ABLATE
#define TN_ABLATE 0
#endif
// (A K-tile is two phases of 32 MFMAs per wave; round 2's four phases of 16 -- twice the barriers -- were 2-4 % slower and are gone.)
// 1 = s_setprio 1 around the MFMA clusters, 2 = raised priority for the LOAD parts; 0 (default): none -- interleaved A/B on the four
// ViT-B weight-gradient shapes: 0 is 1.0-1.9 % faster than 1, 2 is 0.6-1.3 % faster than 1 (round 2's loop gained 10 % from 1)
#ifndef TN_SETPRIO
#define TN_SETPRIO 0
#endif
// 1: the operand DMA of a partial tile skips the columns past the matrix (A/B builds: 0)
#ifndef TN_MASK_COLS
#define TN_MASK_COLS 1
#endif


template <int N>
__device__ __forceinline__ void tn_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void tn_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_wave_base), 16, voffset, 0, 0, 0);
}

template <int IMM>
__device__ __forceinline__ void tn_ds_tr(s16x4& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

__device__ __forceinline__ void tn_section() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}

// Tried in round 3 and dropped (git history: "single-stream" kernel): all 8 waves in step, every wave reading the NEXT phase's
// fragments into a second register set in the shadow of its own MFMAs (inline-asm MFMAs interleaved 1:1 with the 24 reads + DMA,
// one barrier per phase, 230 VGPRs): correct, but 5-8 % SLOWER than this loop on the four ViT-B shapes -- with nothing else to run
// at the phase's wait + barrier the matrix pipe drains twice per K-tile.
// One unit of work of the ping-pong loop: the K-tiles [ch_begin, ch_begin + nk) of the output tile at (c1, c2).
struct TnUnit {
  const bf16_t* A;
  const bf16_t* B;
  long long M;
  int N1, N2;
  int c1, c2;
  long long ch_begin;
  int nk;
  float* dst;      // mode 0 / 1 / 2: the [N1, N2] matrix (a split's slab, or C); mode 3: a compact 256 x 256 slot
  int mode;        // 0 store into a slab, 1 add into C (sole owner of the tile), 2 atomic add into C, 3 store the whole tile into a slot
  int stamp_wg;    // diagnostic build: workgroup index of the stamp record, or -1
};

__device__ __forceinline__ void tn_pp_unit(const TnUnit& p, char* smem) {
  // LDS: [A tile of buffer 0][A of buffer 1][B of buffer 0][B of buffer 1]: both buffers of an operand are within the 64 KiB reach
  // of a ds_read immediate, so the fragment reads of either buffer use ONE address register per 16-column tile (the K loop is
  // unrolled by two, the buffer is a compile-time constant)
  constexpr int B_BASE = 2 * TN_TILE_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w1 = wave >> 2, w2 = wave & 3;

  const int c1 = p.c1, c2 = p.c2;
  const long long ch_begin = p.ch_begin;
  const int nk = p.nk;

  // staging slots of this wave: unit X_kh = rows 32 h + [0, 32) of the operand tile; slot e covers rows
  // 32 h + 4 wave + 2 e + (lane >> 5)
  unsigned voffA[2][2], voffB[2][2];
  int ldsA[2][2], ldsB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r0 = 32 * h + 4 * wave + 2 * e;
      const int row = r0 + (lane >> 5);
      const int ch16 = lane & 31;
      const int sch16 = (((ch16 >> 1) ^ tn_f(row)) << 1) | (ch16 & 1);
      // columns past the operand's width get the out-of-range offset (zero fill, no traffic): without it a partial tile
      // (ViT-S: 384 = 256 + 128) staged the first columns of the NEXT row in their place -- finite junk that the stores
      // skip, but 16 KiB per K-tile the CU took in for nothing
      const bool okA = !TN_MASK_COLS || c1 + sch16 * 8 < p.N1, okB = !TN_MASK_COLS || c2 + sch16 * 8 < p.N2;
      voffA[h][e] = okA ? (unsigned)((row * (long long)p.N1 + c1) * 2 + sch16 * 16) : 0x80000000u;
      voffB[h][e] = okB ? (unsigned)((row * (long long)p.N2 + c2) * 2 + sch16 * 16) : 0x80000000u;
      ldsA[h][e] = r0 * 512;
      ldsB[h][e] = B_BASE + r0 * 512;
    }
  // The K-tile's position lives in the buffer DESCRIPTOR (scalar arithmetic), not in the lanes' offsets: window = rows of K-tile t
  // .. M - 1 of the operand (rows past M read as zero; a K-tile past this workgroup's range gets an empty window: zero fill, no
  // traffic).  The K loop's LOAD parts share their SIMD with the partner wave's MFMA cluster and get about one issue slot per MFMA
  // (tools/tn_stamps.py: 42 instructions took 720 cycles), so every vector instruction removed from them counts.
  auto window = [&](const bf16_t* base, int ncols, int t) {
    const long long row0 = (ch_begin + t) * (long long)TN_KM;
    const long long rows = t < nk ? p.M - row0 : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + row0 * ncols), 0, (int)(rows * ncols * 2), 0x00020000);
  };
  auto stage_a = [&](int t, int bufsel, auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if ((TN_ABLATE & 1) && t >= 2) return;
    __amdgpu_buffer_rsrc_t rs = window(p.A, p.N1, t);
#pragma unroll
    for (int e = 0; e < 2; ++e) tn_dma16(rs, smem + bufsel * TN_TILE_BYTES + ldsA[h][e], voffA[h][e]);
  };
  auto stage_b = [&](int t, int bufsel, auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if ((TN_ABLATE & 1) && t >= 2) return;
    __amdgpu_buffer_rsrc_t rs = window(p.B, p.N2, t);
#pragma unroll
    for (int e = 0; e < 2; ++e) tn_dma16(rs, smem + bufsel * TN_TILE_BYTES + ldsB[h][e], voffB[h][e]);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragment reads are inline asm: for the ds_read_tr builtin (no memory operand) hipcc assumes a
  // dependency on every LDS-DMA in flight and drains vmcnt(0) before each read, which serialises the
  // staging with the arithmetic (that is what held the two-phase loop at ~0.85 PF).  The asm reads land
  // in 64-bit temporaries; the 128-bit MFMA operands are assembled only after the explicit
  // s_waitcnt lgkmcnt(0) that follows the phase's first barrier, so no instruction touches a destination
  // register before its data has arrived.
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  unsigned fragA[8], fragB[4];                         // per-lane byte address of (ks 0, rows r0) of every 16-column tile
  {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp_ = lane & 3;
    const int r0 = 8 * g + q;
#pragma unroll
    for (int i = 0; i < 8; ++i) fragA[i] = lds0 + r0 * 512 + (((w1 * 8 + i) ^ tn_f(r0)) << 5) + 8 * pp_;
#pragma unroll
    for (int j = 0; j < 4; ++j) fragB[j] = lds0 + B_BASE + r0 * 512 + (((w2 * 4 + j) ^ tn_f(r0)) << 5) + 8 * pp_;
  }
  s16x4 tb[4][2];                                      // raw halves (rows r0.., rows r0 + 4..) of the fragments being read
  bf16x8 fb[4];
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

#ifdef VITSSL_TN_STAMPS
  // diagnostic build (tools/tn_stamps.py): time (10 ns ticks) wave 0 / wave 4 spend in each part of a phase, summed over the K loop:
  // 0 LOAD issue (reads + DMA), 1 vmcnt wait, 2 barrier, 3 lgkmcnt wait, 4 MFMA issue, 5 barrier
  unsigned long long tseg[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define TSTAMP(i)                                                    \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime();  \
    tseg[i] += tn_ - tprev;                                          \
    tprev = tn_;                                                     \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)
#else
#define TSTAMP(i) \
  do {            \
  } while (0)
#endif
  s16x4 ta2[8][2];
  bf16x8 fa2[8];
  // the 24 transposed reads of one phase: buffer and contraction half are compile-time (immediate offsets), no address arithmetic
  auto read_frags = [&](auto buf_c, auto ks_c) {
    constexpr int IMM = decltype(buf_c)::value * TN_TILE_BYTES + decltype(ks_c)::value * 16384;
    if (TN_ABLATE & 2) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tn_ds_tr<IMM>(tb[j][0], fragB[j]);
      tn_ds_tr<IMM + 2048>(tb[j][1], fragB[j]);
    }
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
      tn_ds_tr<IMM>(ta2[ii][0], fragA[ii]);
      tn_ds_tr<IMM + 2048>(ta2[ii][1], fragA[ii]);
    }
  };
  auto landed8 = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const tn_s16x8 v = {tb[j][0][0], tb[j][0][1], tb[j][0][2], tb[j][0][3], tb[j][1][0], tb[j][1][1], tb[j][1][2], tb[j][1][3]};
      fb[j] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
      const tn_s16x8 v = {ta2[ii][0][0], ta2[ii][0][1], ta2[ii][0][2], ta2[ii][0][3], ta2[ii][1][0], ta2[ii][1][1], ta2[ii][1][2], ta2[ii][1][3]};
      fa2[ii] = __builtin_bit_cast(bf16x8, v);
    }
  };
  auto mma32 = [&]() {
    if (TN_ABLATE & 4) {
      asm volatile("" ::"v"(fa2[0]), "v"(fa2[1]), "v"(fa2[2]), "v"(fa2[3]), "v"(fa2[4]), "v"(fa2[5]), "v"(fa2[6]), "v"(fa2[7]), "v"(fb[0]), "v"(fb[1]),
                   "v"(fb[2]), "v"(fb[3]));
      return;
    }
    if (TN_SETPRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ii = 0; ii < 8; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa2[ii], acc[ii][j], 0, 0, 0);
    if (TN_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
  };
  // one phase = contraction half ks (32 rows) of K-tile t in buffer b: all 32 MFMAs of the wave; the ks half of K-tile t + 1 goes
  // into the other buffer
  auto phase = [&](auto buf_c, auto ks_c, int t, bool skip_section) {
    constexpr int bsel = decltype(buf_c)::value;
    if (TN_SETPRIO == 2) __builtin_amdgcn_s_setprio(2);
    read_frags(buf_c, ks_c);
    stage_b(t + 1, bsel ^ 1, ks_c);
    stage_a(t + 1, bsel ^ 1, ks_c);
    TSTAMP(0);
    tn_wait_vmcnt<4>();
    TSTAMP(1);
    tn_section();
    TSTAMP(2);
    landed8();
    TSTAMP(3);
    mma32();
    TSTAMP(4);
    if (!skip_section) tn_section();
    TSTAMP(5);
  };
  if (nk > 0) {
    stage_b(0, 0, I0{});
    stage_a(0, 0, I0{});
    stage_b(0, 0, I1{});
    stage_a(0, 0, I1{});
    tn_wait_vmcnt<4>();                                // the ks-0 halves of K-tile 0 have landed
    tn_section();
    if (w1 == 1) tn_section();                         // waves 4-7 run one barrier behind waves 0-3
#ifdef VITSSL_TN_STAMPS
    tprev = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tclk0 = __builtin_amdgcn_s_memtime();     // shader clocks over the same loop: slot 7
#endif
    for (int t = 0; t < nk; t += 2) {
      phase(I0{}, I0{}, t, false);
      phase(I0{}, I1{}, t, t + 1 == nk && w1 == 1);    // (waves 4-7 leave the stagger at the end)
      if (t + 1 >= nk) break;
      phase(I1{}, I0{}, t + 1, false);
      phase(I1{}, I1{}, t + 1, t + 2 == nk && w1 == 1);
    }
#ifdef VITSSL_TN_STAMPS
    if (g_tn_stamps && p.stamp_wg >= 0 && lane == 0 && (wave & 3) == 0) {
      for (int i = 0; i < 6; ++i) g_tn_stamps[((size_t)p.stamp_wg * 2 + w1) * 8 + i] = tseg[i];
      g_tn_stamps[((size_t)p.stamp_wg * 2 + w1) * 8 + 6] = (unsigned long long)nk;
      g_tn_stamps[((size_t)p.stamp_wg * 2 + w1) * 8 + 7] = __builtin_amdgcn_s_memtime() - tclk0;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-fill DMA retired before the LDS is released
  }

  if (p.mode == 3) {                                    // the whole tile, compact: slot[n1 - c1][n2 - c2] (no bounds: the slot is private)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *(f32x4*)(p.dst + (w1 * 128 + i * 16 + (lane & 15)) * TN_T + w2 * 64 + j * 16 + 4 * (lane >> 4)) = acc[i][j];
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n1 = c1 + w1 * 128 + i * 16 + (lane & 15);
    if (n1 >= p.N1) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n2 = c2 + w2 * 64 + j * 16 + 4 * (lane >> 4);
      if (n2 >= p.N2) continue;
      float* q = p.dst + (long long)n1 * p.N2 + n2;
      if (p.mode == 0) {
        *(f32x4*)q = acc[i][j];
      } else if (p.mode == 1) {
        *(f32x4*)q = *(const f32x4*)q + acc[i][j];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(q + r, acc[i][j][r]);
      }
    }
  }
}

__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_pp_kernel(TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntiles = p.tiles1 * p.tiles2;
  const int nwg = ntiles * p.splits;
  const int xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
  const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
  const int split = bid / ntiles;
  const int tile = bid - split * ntiles;
  const int t1 = tile / p.tiles2, t2 = tile - t1 * p.tiles2;
  const long long total_chunks = (p.M + TN_KM - 1) / TN_KM;
  const long long ch_begin = (long long)split * p.chunks_per_split;
  long long ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  TnUnit u;
  u.A = p.A;
  u.B = p.B;
  u.M = p.M;
  u.N1 = p.N1;
  u.N2 = p.N2;
  u.c1 = t1 * TN_T;
  u.c2 = t2 * TN_T;
  u.ch_begin = ch_begin;
  u.nk = ch_begin < ch_end ? (int)(ch_end - ch_begin) : 0;   // empty split: writes a zero slab
  u.dst = p.slabs ? p.slabs + (long long)split * p.N1 * p.N2 : p.C;
  u.mode = p.slabs ? 0 : (p.direct ? 1 : 2);
  u.stamp_wg = (int)blockIdx.x;
  tn_pp_unit(u, smem);
}

// ---- several weight gradients over the same rows in ONE launch (vitssl_gemm_bf16_tn_batch).
// Why: a launch of the kernel above is one round of the CUs, splits = CUs / tiles, and writes CUs x 256 KiB of partial tiles that
// tn_reduce_kernel reads back -- 66 MB each way whatever the shape, ~25 us per launch, for the four gradients of a transformer
// block four times.  Here the tiles of all jobs form one list, unit u = split * T + tile (split-major: the workgroups running at the
// same time work on the same rows of the operands, as before), and workgroup w takes units w, w + G, ...: the split count is
// chosen for the whole list (tn_batch_plan), so fewer, longer units, one launch and one reduce.
constexpr int TN_MAX_JOBS = 8;
struct TnBatchParams {
  const bf16_t* A[TN_MAX_JOBS];
  const bf16_t* B[TN_MAX_JOBS];
  float* C[TN_MAX_JOBS];
  int N1[TN_MAX_JOBS], N2[TN_MAX_JOBS];
  int tiles2[TN_MAX_JOBS];
  int tile0[TN_MAX_JOBS + 1];    // first tile of every job in the list (tile0[njobs] = T)
  int njobs;
  long long M;
  int splits, chunks_per_split;
  int rem_chunks;                // > 0: one round; the last rem_chunks K-tiles of EVERY tile are taken by the workgroups beyond T x splits
                                 // (round-robin over the tiles) and land in the slots of "split" number `splits`
  float* slots;                  // [(splits + (rem_chunks > 0)) * T][256][256] partial tiles, or nullptr when every tile has one owner
};

// Units of the batch kernels.  Without a remainder range: unit u = split * T + tile, workgroup w takes u = w, w + G, ...
// With one (p.rem_chunks > 0; then T * splits <= G): workgroup w < T * splits takes unit w alone, the others share the
// remainder units (K-tiles [splits * chunks_per_split, total) of tile t, slot splits * T + t) round-robin.
struct TnBatchWalk {
  int first, step, end;          // unit indices first, first + step, ... < end
};
__device__ __forceinline__ TnBatchWalk tn_batch_walk(const TnBatchParams& p, int bid, int G) {
  const int T = p.tile0[p.njobs];
  const int mainu = T * p.splits;
  TnBatchWalk w;
  if (p.rem_chunks > 0) {
    if (bid < mainu) {
      w.first = bid;
      w.step = mainu + T;
      w.end = mainu;
    } else {
      w.first = mainu + (bid - mainu);
      w.step = G - mainu;
      w.end = mainu + T;
    }
  } else {
    w.first = bid;
    w.step = G;
    w.end = mainu;
  }
  return w;
}
// (job, tile origin, K range) of unit un
__device__ __forceinline__ void tn_batch_unit(const TnBatchParams& p, int un, int km, int& j, int& c1, int& c2, long long& ch_begin, int& nk) {
  const int T = p.tile0[p.njobs];
  const int split = un / T;
  const int gt = un - split * T;
  j = 0;
#pragma unroll
  for (int q = 1; q < TN_MAX_JOBS; ++q)
    if (q < p.njobs && gt >= p.tile0[q]) j = q;
  const int lt = gt - p.tile0[j];
  const int t1 = lt / p.tiles2[j], t2 = lt - t1 * p.tiles2[j];
  c1 = t1 * TN_T;
  c2 = t2 * TN_T;
  const long long total_chunks = (p.M + km - 1) / km;
  ch_begin = (long long)split * p.chunks_per_split;
  long long ch_end = split < p.splits ? ch_begin + p.chunks_per_split : total_chunks;     // "split" == splits: the remainder range
  if (ch_end > total_chunks) ch_end = total_chunks;
  nk = ch_begin < ch_end ? (int)(ch_end - ch_begin) : 0;
}

__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_batch_kernel(TnBatchParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, qq = G >> 3, rr = G & 7;
  const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
  const TnBatchWalk w = tn_batch_walk(p, bid, G);
  for (int un = w.first; un < w.end; un += w.step) {
    TnUnit u;
    int j;
    tn_batch_unit(p, un, TN_KM, j, u.c1, u.c2, u.ch_begin, u.nk);
    u.A = p.A[j];
    u.B = p.B[j];
    u.M = p.M;
    u.N1 = p.N1[j];
    u.N2 = p.N2[j];
    u.dst = p.slots ? p.slots + (long long)un * (TN_T * TN_T) : p.C[j];
    u.mode = p.slots ? 3 : 1;
    u.stamp_wg = -1;
    tn_pp_unit(u, smem);
    tn_section();                                      // every wave is done with the LDS before the next unit's first DMA
  }
}

// C_j[tile] += sum over splits of the tile's slots; one block = 32 rows of one tile
__global__ __launch_bounds__(256) void tn_batch_reduce_kernel(TnBatchParams p) {
  const int T = p.tile0[p.njobs];
  const int gt = blockIdx.x >> 3, rblk = blockIdx.x & 7;
  int j = 0;
  for (int q = 1; q < p.njobs; ++q)
    if (gt >= p.tile0[q]) j = q;
  const int lt = gt - p.tile0[j];
  const int t1 = lt / p.tiles2[j], t2 = lt - t1 * p.tiles2[j];
  const int col = 4 * (threadIdx.x & 63);
  const int n2 = t2 * TN_T + col;
  if (n2 >= p.N2[j]) return;
#pragma unroll 2
  for (int r = threadIdx.x >> 6; r < 32; r += 4) {
    const int row = rblk * 32 + r;
    const int n1 = t1 * TN_T + row;
    if (n1 >= p.N1[j]) break;
    float* q = p.C[j] + (long long)n1 * p.N2[j] + n2;
    f32x4 a = *(const f32x4*)q;
    const float* sl = p.slots + (long long)gt * (TN_T * TN_T) + row * TN_T + col;
    const long long sstride = (long long)T * (TN_T * TN_T);
    const int nsl = p.splits + (p.rem_chunks > 0 ? 1 : 0);
    int s = 0;
    for (; s + 4 <= nsl; s += 4) {                    // four independent loads in flight per row
      const f32x4 v0 = *(const f32x4*)(sl + (s + 0) * sstride), v1 = *(const f32x4*)(sl + (s + 1) * sstride);
      const f32x4 v2 = *(const f32x4*)(sl + (s + 2) * sstride), v3 = *(const f32x4*)(sl + (s + 3) * sstride);
      a += (v0 + v1) + (v2 + v3);
    }
    for (; s < nsl; ++s) a += *(const f32x4*)(sl + s * sstride);
    *(f32x4*)q = a;
  }
}

// ====================================================================================
// Weight gradient on e4m3 operands (fp8 path):  C[N1,N2] += alpha * A8[M,N1]^T . B8[M,N2].
// A K-tile is 128 contraction rows of 256 one-byte columns (32 KiB per operand, the bf16 tile's bytes); a lane's
// operand of v_mfma_f32_16x16x128_f8f6f4 -- 32 k-values of one column -- is four ds_read_b64_tr_b8 (each hands a lane
// column (lane & 15) of 8 rows; lane i of a 16-lane group addresses the 8-byte half i & 1 of row i >> 1).  Lane group g
// takes rows 32 g + 8 q + 0..7 in read q: the same k order for both operands, which is all a contraction needs.
// The 16-byte chunk index is XORed with f(row) = (row & 7) | ((row >> 5) & 1) << 3 on the DMA source side and on the
// reads: the two lane groups of a half-wave then touch 16 rows x 16 B in 16 distinct chunk positions (conflict-free).
// Plain double buffering, all 8 waves in step: one K = 128 MFMA step consumes the whole tile, so the half-tile units of
// the ping-pong schedule (refilled two phases after their last read) do not exist here; the next tile's DMA is in
// flight under the current tile's 48 reads and 32 MFMAs per wave.
struct Tn8Params {
  const unsigned char* A;
  const unsigned char* B;
  float* C;
  long long M;
  int N1, N2;
  int tiles1, tiles2, splits;
  int chunks_per_split;   // in units of TN8_KM rows
  float* slabs;
  int direct;             // as TnParams::direct
  const float* alpha;     // device scalars multiplied into the result (dequantisation of the two operands), or NULL
  const float* alpha2;
};
constexpr int TN8_KM = 128;
typedef __attribute__((ext_vector_type(2))) int tn_i32x2;

__device__ __forceinline__ int tn8_f(int row) { return (row & 7) | (((row >> 5) & 1) << 3); }

template <int IMM>
__device__ __forceinline__ void tn8_ds_tr(tn_i32x2& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM));
}

// One unit of work of the e4m3 weight-gradient loop (as TnUnit; K-tiles of TN8_KM rows)
struct Tn8Unit {
  const unsigned char* A;
  const unsigned char* B;
  long long M;
  int N1, N2;
  int c1, c2;
  long long ch_begin;
  int nk;
  float* dst;
  int mode;               // 0 slab, 1 add into C, 2 atomic add into C, 3 compact 256 x 256 slot
  const float* alpha;
  const float* alpha2;
};

__device__ __forceinline__ void tn8_unit(const Tn8Unit& p, char* smem) {
  constexpr int BUF = 2 * TN_TILE_BYTES;               // one K-tile: A tile then B tile
  constexpr unsigned OOBV = 0x80000000u;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w1 = wave >> 2, w2 = wave & 3;

  const int c1 = p.c1, c2 = p.c2;
  const long long ch_begin = p.ch_begin;
  const int nk = p.nk;

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((unsigned long long)p.M * p.N1), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((unsigned long long)p.M * p.N2), 0x00020000);

  // staging: instruction slot e of this wave covers tile rows 16 wave + 4 e + (lane >> 4), one 16-byte chunk per lane
  unsigned voffA[4], voffB[4];
  int ldsoff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r0 = 16 * wave + 4 * e;
    const int row = r0 + (lane >> 4);
    const int sch = (lane & 15) ^ tn8_f(row);
    const bool ok1 = c1 + sch * 16 < p.N1, ok2 = c2 + sch * 16 < p.N2;     // columns past the matrix: zero fill
    voffA[e] = ok1 ? (unsigned)(row * (long long)p.N1 + c1 + sch * 16) : OOBV;
    voffB[e] = ok2 ? (unsigned)(row * (long long)p.N2 + c2 + sch * 16) : OOBV;
    ldsoff[e] = r0 * 256;
  }
  const unsigned stepA = (unsigned)TN8_KM * (unsigned)p.N1, stepB = (unsigned)TN8_KM * (unsigned)p.N2;
  auto stage = [&](int t, int bufsel) {
    const unsigned ba = (unsigned)(ch_begin + t) * stepA, bb = (unsigned)(ch_begin + t) * stepB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      tn_dma16(rsA, smem + bufsel * BUF + ldsoff[e], voffA[e] == OOBV ? OOBV : voffA[e] + ba);
      tn_dma16(rsB, smem + bufsel * BUF + TN_TILE_BYTES + ldsoff[e], voffB[e] == OOBV ? OOBV : voffB[e] + bb);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane byte address of read 0 (rows 32 g + (i >> 1)) of every 16-column tile; read q adds 8 rows = 2048 bytes
  // (f(row + 8 q) = f(row) for q < 4: bits 3 and 4 of the row do not enter f)
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  unsigned fragA[8], fragB[4];
  {
    const int g = lane >> 4, i16 = lane & 15;
    const int r0 = 32 * g + (i16 >> 1);
    const int fx = tn8_f(r0);
#pragma unroll
    for (int i = 0; i < 8; ++i) fragA[i] = lds0 + r0 * 256 + (((w1 * 8 + i) ^ fx) << 4) + 8 * (i16 & 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) fragB[j] = lds0 + TN_TILE_BYTES + r0 * 256 + (((w2 * 4 + j) ^ fx) << 4) + 8 * (i16 & 1);
  }

  if (nk > 0) stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const unsigned cur = (unsigned)((t & 1) * BUF);
    if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
    tn_i32x2 rb[4][4], ra[8][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tn8_ds_tr<0>(rb[j][0], fragB[j] + cur);
      tn8_ds_tr<2048>(rb[j][1], fragB[j] + cur);
      tn8_ds_tr<4096>(rb[j][2], fragB[j] + cur);
      tn8_ds_tr<6144>(rb[j][3], fragB[j] + cur);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tn8_ds_tr<0>(ra[i][0], fragA[i] + cur);
      tn8_ds_tr<2048>(ra[i][1], fragA[i] + cur);
      tn8_ds_tr<4096>(ra[i][2], fragA[i] + cur);
      tn8_ds_tr<6144>(ra[i][3], fragA[i] + cur);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    i32x8 fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fb[j] = i32x8{rb[j][0][0], rb[j][0][1], rb[j][1][0], rb[j][1][1], rb[j][2][0], rb[j][2][1], rb[j][3][0], rb[j][3][1]};
    if (TN_SETPRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const i32x8 fa = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1], ra[i][2][0], ra[i][2][1], ra[i][3][0], ra[i][3][1]};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa, acc[i][j], 0, 0, 0, 0, 0, 0);
    }
    if (TN_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const float al = (p.alpha ? *p.alpha : 1.0f) * (p.alpha2 ? *p.alpha2 : 1.0f);
  if (p.mode == 3) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *(f32x4*)(p.dst + (w1 * 128 + i * 16 + (lane & 15)) * TN_T + w2 * 64 + j * 16 + 4 * (lane >> 4)) = acc[i][j] * al;
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n1 = c1 + w1 * 128 + i * 16 + (lane & 15);
    if (n1 >= p.N1) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n2 = c2 + w2 * 64 + j * 16 + 4 * (lane >> 4);
      if (n2 >= p.N2) continue;
      float* q = p.dst + (long long)n1 * p.N2 + n2;
      const f32x4 v = acc[i][j] * al;
      if (p.mode == 0) {
        *(f32x4*)q = v;
      } else if (p.mode == 1) {
        *(f32x4*)q = *(const f32x4*)q + v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(q + r, v[r]);
      }
    }
  }
}


// ---- ping-pong form of the e4m3 weight-gradient loop (round 4) ----------------------------------------------------------------
// Same roles as tn_pp_unit: waves 4-7 (the n1 half w1 = 1 of the tile) run one barrier behind waves 0-3, so on every SIMD one wave is
// in its 32-MFMA cluster while its partner reads fragments and issues DMA.  One K = 128 MFMA consumes all 128 rows of a K-tile, so
// the refill units cannot be row halves as in the bf16 loop; they are COLUMN ranges, made separable in LDS:
//     A_g = columns 128 g .. 128 g + 127 of the A tile (what wave group g multiplies): 128 rows x 128 B, its own 16 KiB region,
//     B   = the whole B tile (both groups read it): 128 rows x 256 B.
// LDS: [A_0 buf 0][A_0 buf 1][A_1 buf 0][A_1 buf 1][B buf 0][B buf 1] (both buffers of a region within a ds_read immediate).
// A_g rows are 128 B, so rows r and r + 1 share a 256-byte bank row; 16-byte chunk c of row r sits at position c ^ fa(r),
// fa(r) = ((r >> 1) & 3) | ((r >> 5) & 1) << 2: the 32 lanes of a transposed read (rows 8 q + 0..7 and 32 + 8 q + 0..7) then touch
// 16 distinct 16-byte slots of the bank row (conflict-free), and a DMA instruction fetches 8 rows x one whole 128-byte line.
// Schedule (group 0 = waves 0-3, group 1 = waves 4-7; K-tile t lives in buffer t & 1):
//     group 0, LOAD(t): 48 reads of (A_0, B)(t); DMA B(t+1);                lgkmcnt(0); barrier; 32 MFMAs; vmcnt(0); barrier
//     group 1, LOAD(t): 48 reads of (A_1, B)(t); DMA A_1(t+1), A_0(t+2); vmcnt(8); lgkmcnt(0); barrier; 32 MFMAs; vmcnt(4); barrier
// A region is refilled by a wave that has passed a barrier BEHIND the lgkmcnt(0) of its last readers (hence the wait in front of the
// barrier, unlike the bf16 loop), and every wave waits for its own DMA in front of the barrier that precedes the first read of it:
// B(t+1) (issued before barrier 2t) by the vmcnt(0) in front of barrier 2t+1; A_1(t+1) (issued between barriers 2t and 2t+1) by the
// vmcnt(4) in front of barrier 2t+2; A_0(t+2) (same place) by the vmcnt(8) of LOAD(t+1), in front of barrier 2t+3 -- group 0 reads
// it after that barrier.  K-tiles past the unit's range get an empty buffer window (zero fill, no traffic), so counts stay uniform.
__device__ __forceinline__ int tn8_fa(int row) { return ((row >> 1) & 3) | (((row >> 5) & 1) << 2); }

__device__ __forceinline__ void tn8_pp_unit(const Tn8Unit& p, char* smem) {
  constexpr int AG = TN8_KM * 128;                     // 16 KiB: one A_g region
  constexpr int BT = TN8_KM * 256;                     // 32 KiB: the B tile
  constexpr int B_BASE = 4 * AG;
  constexpr unsigned OOBV = 0x80000000u;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w1 = wave >> 2, w2 = wave & 3;
  const int c1 = p.c1, c2 = p.c2;
  const long long ch_begin = p.ch_begin;
  const int nk = p.nk;

  // K-tile t of an operand as a buffer window: rows (ch_begin + t) * 128 .. M - 1 (rows past M read as zero); empty past the range
  auto window = [&](const unsigned char* base, int ncols, int t) {
    const long long row0 = (ch_begin + t) * (long long)TN8_KM;
    const long long rows = t < nk ? p.M - row0 : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + row0 * ncols), 0, (int)(rows * ncols), 0x00020000);
  };
  // staging slots: 8 DMA instructions per wave and K-tile.  Group 0: the B tile, slot e = rows 32 w2 + 4 e (+ lane >> 4), chunk
  // lane & 15 (as tn8_unit).  Group 1: slots 0-3 = A_1, slots 4-7 = A_0, rows 32 w2 + 8 (e & 3) (+ lane >> 3), position lane & 7.
  unsigned voff[8];
  int ldsoff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (w1 == 0) {
      const int r0 = 32 * w2 + 4 * e;
      const int row = r0 + (lane >> 4);
      const int sch = (lane & 15) ^ tn8_f(row);
      voff[e] = c2 + sch * 16 < p.N2 ? (unsigned)(row * (long long)p.N2 + c2 + sch * 16) : OOBV;
      ldsoff[e] = B_BASE + r0 * 256;
    } else {
      const int gsel = e < 4 ? 1 : 0;
      const int r0 = 32 * w2 + 8 * (e & 3);
      const int row = r0 + (lane >> 3);
      const int sc = (lane & 7) ^ tn8_fa(row);
      const int col = c1 + 128 * gsel + sc * 16;
      voff[e] = col < p.N1 ? (unsigned)(row * (long long)p.N1 + col) : OOBV;
      ldsoff[e] = gsel * 2 * AG + r0 * 128;
    }
  }
  auto issue_b = [&](int t, int bufsel) {               // group 0
    __amdgpu_buffer_rsrc_t rs = window(p.B, p.N2, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) tn_dma16(rs, smem + ldsoff[e] + bufsel * BT, voff[e]);
  };
  auto issue_a = [&](int t, int bufsel, auto first_c) {  // group 1: slots first .. first + 3
    constexpr int first = decltype(first_c)::value;
    __amdgpu_buffer_rsrc_t rs = window(p.A, p.N1, t);
#pragma unroll
    for (int e = first; e < first + 4; ++e) tn_dma16(rs, smem + ldsoff[e] + bufsel * AG, voff[e]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I4 = std::integral_constant<int, 4>;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses (buffer 0, read 0): A_g rows are 128 B (read q adds 8 rows = 1024 B), B rows 256 B (2048 B per read)
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  unsigned fragA[8], fragB[4];
  {
    const int g = lane >> 4, i16 = lane & 15;
    const int r0 = 32 * g + (i16 >> 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) fragA[i] = lds0 + w1 * 2 * AG + r0 * 128 + ((i ^ tn8_fa(r0)) << 4) + 8 * (i16 & 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) fragB[j] = lds0 + B_BASE + r0 * 256 + (((w2 * 4 + j) ^ tn8_f(r0)) << 4) + 8 * (i16 & 1);
  }
  tn_i32x2 rb[4][4], ra[8][4];
  auto read_frags = [&](auto buf_c) {
    constexpr int IA = decltype(buf_c)::value * AG, IB = decltype(buf_c)::value * BT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tn8_ds_tr<IB>(rb[j][0], fragB[j]);
      tn8_ds_tr<IB + 2048>(rb[j][1], fragB[j]);
      tn8_ds_tr<IB + 4096>(rb[j][2], fragB[j]);
      tn8_ds_tr<IB + 6144>(rb[j][3], fragB[j]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tn8_ds_tr<IA>(ra[i][0], fragA[i]);
      tn8_ds_tr<IA + 1024>(ra[i][1], fragA[i]);
      tn8_ds_tr<IA + 2048>(ra[i][2], fragA[i]);
      tn8_ds_tr<IA + 3072>(ra[i][3], fragA[i]);
    }
  };
  auto mma = [&]() {
    i32x8 fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fb[j] = i32x8{rb[j][0][0], rb[j][0][1], rb[j][1][0], rb[j][1][1], rb[j][2][0], rb[j][2][1], rb[j][3][0], rb[j][3][1]};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const i32x8 fa = {ra[i][0][0], ra[i][0][1], ra[i][1][0], ra[i][1][1], ra[i][2][0], ra[i][2][1], ra[i][3][0], ra[i][3][1]};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa, acc[i][j], 0, 0, 0, 0, 0, 0);
    }
  };
  auto ktile = [&](auto buf_c, int t, bool skip_section) {
    constexpr int bsel = decltype(buf_c)::value;
    read_frags(buf_c);
    if (w1 == 0) {
      issue_b(t + 1, bsel ^ 1);
    } else {
      issue_a(t + 1, bsel ^ 1, I0{});                   // A_1(t+1)
      issue_a(t + 2, bsel, I4{});                       // A_0(t+2)
      tn_wait_vmcnt<8>();                               // everything older than those two: A_0(t+1) among it
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are DONE before the barrier: the partner group refills behind it
    tn_section();
    mma();
    if (w1 == 0) tn_wait_vmcnt<0>();                    // B(t+1)
    else tn_wait_vmcnt<4>();                            // A_1(t+1)
    if (!skip_section) tn_section();
  };

  if (nk > 0) {
    if (w1 == 0) {
      issue_b(0, 0);
    } else {
      issue_a(0, 0, I0{});
      issue_a(0, 0, I4{});
      issue_a(1, 1, I4{});
    }
    tn_wait_vmcnt<0>();
    tn_section();
    if (w1 == 1) tn_section();                         // waves 4-7 run one barrier behind waves 0-3
    for (int t = 0; t < nk; t += 2) {
      ktile(I0{}, t, t + 1 == nk && w1 == 1);          // (waves 4-7 leave the stagger at the end)
      if (t + 1 >= nk) break;
      ktile(I1{}, t + 1, t + 2 == nk && w1 == 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing zero-fill DMA retired before the LDS is released
  }

  const float al = (p.alpha ? *p.alpha : 1.0f) * (p.alpha2 ? *p.alpha2 : 1.0f);
  if (p.mode == 3) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *(f32x4*)(p.dst + (w1 * 128 + i * 16 + (lane & 15)) * TN_T + w2 * 64 + j * 16 + 4 * (lane >> 4)) = acc[i][j] * al;
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n1 = c1 + w1 * 128 + i * 16 + (lane & 15);
    if (n1 >= p.N1) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n2 = c2 + w2 * 64 + j * 16 + 4 * (lane >> 4);
      if (n2 >= p.N2) continue;
      float* q = p.dst + (long long)n1 * p.N2 + n2;
      const f32x4 v = acc[i][j] * al;
      if (p.mode == 0) {
        *(f32x4*)q = v;
      } else if (p.mode == 1) {
        *(f32x4*)q = *(const f32x4*)q + v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(q + r, v[r]);
      }
    }
  }
}

template <bool PP>
__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_fp8_kernel(Tn8Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ntiles = p.tiles1 * p.tiles2;
  const int nwg = ntiles * p.splits;
  const int xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
  const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
  const int split = bid / ntiles;
  const int tile = bid - split * ntiles;
  const int t1 = tile / p.tiles2, t2 = tile - t1 * p.tiles2;
  const long long total_chunks = (p.M + TN8_KM - 1) / TN8_KM;
  const long long ch_begin = (long long)split * p.chunks_per_split;
  long long ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  Tn8Unit u;
  u.A = p.A;
  u.B = p.B;
  u.M = p.M;
  u.N1 = p.N1;
  u.N2 = p.N2;
  u.c1 = t1 * TN_T;
  u.c2 = t2 * TN_T;
  u.ch_begin = ch_begin;
  u.nk = ch_begin < ch_end ? (int)(ch_end - ch_begin) : 0;   // empty split: writes a zero slab
  u.dst = p.slabs ? p.slabs + (long long)split * p.N1 * p.N2 : p.C;
  u.mode = p.slabs ? 0 : (p.direct ? 1 : 2);
  u.alpha = p.alpha;
  u.alpha2 = p.alpha2;
  if constexpr (PP) tn8_pp_unit(u, smem);
  else tn8_unit(u, smem);
}

// several e4m3 weight gradients over the same rows in one launch (vitssl_gemm_fp8_tn_batch; see gemm_tn_batch_kernel)
struct Tn8BatchParams {
  TnBatchParams b;               // A / B hold the e4m3 images
  const float* alpha[TN_MAX_JOBS];
  const float* alpha2[TN_MAX_JOBS];
};

template <bool PP>
__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_fp8_batch_kernel(Tn8BatchParams pp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TnBatchParams& p = pp.b;
  const int G = gridDim.x;
  const int xcd = blockIdx.x & 7, qq = G >> 3, rr = G & 7;
  const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
  const TnBatchWalk w = tn_batch_walk(p, bid, G);
  for (int un = w.first; un < w.end; un += w.step) {
    Tn8Unit u;
    int j;
    tn_batch_unit(p, un, TN8_KM, j, u.c1, u.c2, u.ch_begin, u.nk);
    u.A = (const unsigned char*)p.A[j];
    u.B = (const unsigned char*)p.B[j];
    u.M = p.M;
    u.N1 = p.N1[j];
    u.N2 = p.N2[j];
    u.dst = p.slots ? p.slots + (long long)un * (TN_T * TN_T) : p.C[j];
    u.mode = p.slots ? 3 : 1;
    u.alpha = pp.alpha[j];
    u.alpha2 = pp.alpha2[j];
    if constexpr (PP) tn8_pp_unit(u, smem);
    else tn8_unit(u, smem);
    __syncthreads();                                   // every wave is done with the LDS before the next unit's first DMA
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

void tn_plan(long long M, int N1, int N2, int* tiles1, int* tiles2, int* splits, int* chunks_per_split, int km = TN_KM) {
  *tiles1 = (N1 + TN_T - 1) / TN_T;
  *tiles2 = (N2 + TN_T - 1) / TN_T;
  const long long total_chunks = (M + km - 1) / km;
  const int ntiles = *tiles1 * *tiles2;
  long long sp = vitssl_persistent_cus() / ntiles;   // one round of the CUs this library may occupy
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
  // One split (more C tiles than CUs: the DINO head's [65536, 768] weight gradient): the slab would be written, read back and
  // added to C by a second kernel -- 3 x 201 MB for nothing, since every tile has a single owner (tn_reduce was 1.6 ms of a
  // DINO step, 114 us per such call).  The owner adds into C itself.
  p.direct = p.splits == 1;
  if (p.direct) p.slabs = nullptr;
  static VsOnce attr_done{false};
  static VsEnvInt pp_env;                              // VITSSL_TN_PP=0: the two-phase loop (developer knob; tests/test_gpu_knobs.py)
  const int use_pp = pp_env.get("VITSSL_TN_PP", 1);
  if (!attr_done.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_tn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_tn: cannot raise dynamic LDS: %s", hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done.store(true, std::memory_order_relaxed);
  }
  hipStream_t s = (hipStream_t)stream;
  if (use_pp)
    hipLaunchKernelGGL(gemm_tn_pp_kernel, dim3(p.tiles1 * p.tiles2 * p.splits), dim3(TN_THREADS), TN_LDS_BYTES, s, p);
  else
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

// ---- batch of weight gradients over the same M rows
namespace {
// Split count for a list of T tiles: minimises (rounds of the CUs) x (K-tiles per unit x 1.46 us + 9 us per unit: pipeline fill,
// hand-over, slot store) plus the partial-tile traffic (0.15 us per 256 KiB partial written and read back).  The two constants are
// fitted to whole-step measurements with forced split counts (VITSSL_TN_BATCH_SPLITS): ViT-B (108 tiles) 2 splits 33.27 ms,
// 7: 33.49, 9: 33.91, 3: 35.69; ViT-S (38 tiles) 6 splits 14.35 ms, 13: 14.78, 4: 14.95, 20: 15.21, 8: 15.55.
// 1 split = every tile has one owner: no partials.
// G = workgroups of the launch = CUs this library may occupy, read ONCE per entry call by the caller (the reserve is mutable process
// state: vitssl_set_reserved_cus), so that the plan, the workspace requirement and the grid of one launch all follow the same value.
void tn_batch_plan(long long M, int T, int G, int* splits, int* chunks_per_split, int* rem_chunks, int km = TN_KM) {
  const long long total_chunks = (M + km - 1) / km;
  const double c_kt = 1.46, c_unit = 9.0, c_part = 0.15;        // us per K-tile of a unit, per unit, per partial tile (see above)
  double best = 1e30;
  int best_s = 1;
  const long long smax = total_chunks < 64 ? total_chunks : 64;
  for (long long sp = 1; sp <= smax; ++sp) {
    const long long cps = (total_chunks + sp - 1) / sp;
    const long long s_eff = (total_chunks + cps - 1) / cps;
    if (s_eff != sp) continue;                                  // same partition as a smaller count
    const long long units = (long long)T * sp;
    const long long rounds = (units + G - 1) / G;
    const double cost = (double)rounds * ((double)cps * c_kt + c_unit) + (sp > 1 ? (double)units * c_part : 0.0);
    if (cost < best) {
      best = cost;
      best_s = (int)sp;
    }
  }
  static VsEnvInt forced_env;                                   // VITSSL_TN_BATCH_SPLITS: force the split count (developer knob)
  const int forced = forced_env.get("VITSSL_TN_BATCH_SPLITS", 0);
  if (forced > 0 && forced <= total_chunks) best_s = forced;
  *chunks_per_split = (int)((total_chunks + best_s - 1) / best_s);
  *splits = (int)((total_chunks + *chunks_per_split - 1) / *chunks_per_split);
  *rem_chunks = 0;
  // One round with idle CUs (ViT-B: 108 tiles x 2 splits = 216 of 256): the R idle workgroups take the LAST part of every tile's
  // rows, ceil(T / R) tiles each, and the main units shrink until both kinds finish together:
  //   L c_kt + c_unit = n_r ((total - S L) c_rem + c_unit).
  // c_rem > c_kt: the few helper workgroups of an XCD work on different tiles and share little in L2.  Measured (whole step, same
  // box, alternating; helpers sized with c_rem = c_kt): ViT-B 32.90 -> 32.67 ms, but ViT-S 14.17 -> 14.24 and DINO 39.9 -> 40.2
  // where the model promised 7 %: with c_rem = 2.2 us and a 5 % threshold only the ViT-B-like lists use helpers.
  // VITSSL_TN_BATCH_REM=0 turns them off.
  static VsEnvInt rem_env;
  const int use_rem = rem_env.get("VITSSL_TN_BATCH_REM", 1);
  const long long mainu = (long long)T * *splits;
  const long long R = G - mainu;
  if (use_rem && mainu <= G && R >= 8) {
    const double c_rem = 2.2;
    const long long S = *splits;
    const double nr = (double)((T + R - 1) / R);
    const double L = (nr * ((double)total_chunks * c_rem + c_unit) - c_unit) / (c_kt + nr * (double)S * c_rem);
    long long Li = (long long)(L + 0.999);
    const long long rem = total_chunks - S * Li;
    const double t_main = (double)Li * c_kt + c_unit, t_rem = nr * ((double)rem * c_rem + c_unit);
    const double t_old = (double)*chunks_per_split * c_kt + c_unit;
    const double tax = (double)T * c_part * (S == 1 ? 2.0 : 1.0);             // T more partials (with one split there were none)
    if (rem >= 4 && Li >= 4 && (t_main > t_rem ? t_main : t_rem) + tax < 0.95 * t_old) {
      *chunks_per_split = (int)Li;
      *rem_chunks = (int)rem;
    }
  }
}
int tn_batch_tiles(const vitssl_tn_job_t* jobs, int njobs, TnBatchParams* p) {
  int t = 0;
  for (int j = 0; j < njobs; ++j) {
    p->tile0[j] = t;
    p->tiles2[j] = (jobs[j].N2 + TN_T - 1) / TN_T;
    t += ((jobs[j].N1 + TN_T - 1) / TN_T) * p->tiles2[j];
  }
  p->tile0[njobs] = t;
  return t;
}
}  // namespace

extern "C" int64_t vitssl_gemm_tn_batch_workspace_floats(const vitssl_tn_job_t* jobs, int njobs, int64_t M) {
  if (!jobs || njobs <= 0 || njobs > TN_MAX_JOBS || M <= 0) return 0;
  TnBatchParams p;
  const int T = tn_batch_tiles(jobs, njobs, &p);
  int sp, cps, rem;
  tn_batch_plan(M, T, vitssl_persistent_cus(), &sp, &cps, &rem);
  return (sp > 1 || rem > 0) ? (int64_t)(sp + (rem > 0)) * T * TN_T * TN_T : 0;
}

extern "C" int vitssl_gemm_bf16_tn_batch(const vitssl_tn_job_t* jobs, int njobs, int64_t M, float* workspace, int64_t workspace_floats,
                                         void* stream) {
  VS_CHECK_ARG(jobs && njobs > 0 && njobs <= TN_MAX_JOBS, "gemm_tn_batch: 1..%d jobs", TN_MAX_JOBS);
  VS_CHECK_ARG(M > 0, "gemm_tn_batch: empty problem");
  TnBatchParams p;
  for (int j = 0; j < njobs; ++j) {
    const vitssl_tn_job_t& q = jobs[j];
    VS_CHECK_ARG(q.A && q.B && q.C, "gemm_tn_batch: job %d: null operand", j);
    VS_CHECK_ARG(q.N1 > 0 && q.N2 > 0 && q.N1 % 8 == 0 && q.N2 % 8 == 0, "gemm_tn_batch: job %d: N1=%d N2=%d must be positive multiples of 8", j,
                 q.N1, q.N2);
    VS_CHECK_ARG((unsigned long long)M * q.N1 * 2ull < (1ull << 31) && (unsigned long long)M * q.N2 * 2ull < (1ull << 31),
                 "gemm_tn_batch: job %d: operand larger than 2 GiB", j);
    p.A[j] = (const bf16_t*)q.A;
    p.B[j] = (const bf16_t*)q.B;
    p.C[j] = q.C;
    p.N1[j] = q.N1;
    p.N2[j] = q.N2;
  }
  for (int j = njobs; j < TN_MAX_JOBS; ++j) {
    p.A[j] = p.B[j] = nullptr;
    p.C[j] = nullptr;
    p.N1[j] = p.N2[j] = p.tiles2[j] = 0;
  }
  const int T = tn_batch_tiles(jobs, njobs, &p);
  for (int j = njobs + 1; j <= TN_MAX_JOBS; ++j) p.tile0[j] = T;
  p.njobs = njobs;
  p.M = M;
  const long long cus = vitssl_persistent_cus();               // once per launch: plan, workspace check and grid use this value
  tn_batch_plan(M, T, (int)cus, &p.splits, &p.chunks_per_split, &p.rem_chunks);
  const long long need = (p.splits > 1 || p.rem_chunks > 0) ? (long long)(p.splits + (p.rem_chunks > 0)) * T * TN_T * TN_T : 0;
  VS_CHECK_ARG(need == 0 || (workspace && workspace_floats >= need), "gemm_tn_batch: workspace of %lld floats needed with %lld CUs available (vitssl_gemm_tn_batch_workspace_floats; the answer\n"
               "changes with vitssl_set_reserved_cus: query again after changing the reserve)", need, cus);
  p.slots = need ? workspace : nullptr;
  static VsOnce attr_done{false};
  if (!attr_done.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_tn_batch: cannot raise dynamic LDS: %s", hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done.store(true, std::memory_order_relaxed);
  }
  hipStream_t s = (hipStream_t)stream;
  const long long units = (long long)T * p.splits;
  const unsigned grid = (unsigned)((units < cus && p.rem_chunks == 0) ? units : cus);      // (with a remainder range every CU has work)
  hipLaunchKernelGGL(gemm_tn_batch_kernel, dim3(grid), dim3(TN_THREADS), TN_LDS_BYTES, s, p);
  VS_CHECK_LAUNCH("gemm_tn_batch");
  if (p.slots) {
    hipLaunchKernelGGL(tn_batch_reduce_kernel, dim3((unsigned)T * 8u), dim3(256), 0, s, p);
    VS_CHECK_LAUNCH("gemm_tn_batch_reduce");
  }
  return VITSSL_OK;
}

// 1 (default): the e4m3 weight gradients run the ping-pong loop (tn8_pp_unit); 0: all eight waves in step (tn8_unit)
// (VITSSL_TN8_PP, developer knob; tests/test_gpu_knobs.py)
static int tn8_pp_enabled() {
  static VsEnvInt knob;
  return knob.get("VITSSL_TN8_PP", 1);
}

extern "C" int64_t vitssl_gemm_fp8_tn_workspace_floats(int64_t M, int N1, int N2) {
  if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
  int t1, t2, sp, cps;
  tn_plan(M, N1, N2, &t1, &t2, &sp, &cps, TN8_KM);
  return (int64_t)sp * N1 * N2;
}

extern "C" int vitssl_gemm_fp8_tn(const void* A8, const void* B8, float* C, int64_t M, int N1, int N2, const float* alpha,
                                  const float* alpha2, float* workspace, int64_t workspace_floats, void* stream) {
  VS_CHECK_ARG(A8 && B8 && C, "gemm_fp8_tn: null operand");
  VS_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0, "gemm_fp8_tn: empty problem");
  VS_CHECK_ARG(N1 % 16 == 0 && N2 % 16 == 0, "gemm_fp8_tn: N1=%d N2=%d must be multiples of 16", N1, N2);
  VS_CHECK_ARG((unsigned long long)M * N1 < (1ull << 31) && (unsigned long long)M * N2 < (1ull << 31),
               "gemm_fp8_tn: operand larger than 2 GiB");
  Tn8Params p;
  p.A = (const unsigned char*)A8;
  p.B = (const unsigned char*)B8;
  p.C = C;
  p.M = M;
  p.N1 = N1;
  p.N2 = N2;
  p.alpha = alpha;
  p.alpha2 = alpha2;
  tn_plan(M, N1, N2, &p.tiles1, &p.tiles2, &p.splits, &p.chunks_per_split, TN8_KM);
  const long long need = (long long)p.splits * N1 * N2;
  p.slabs = (workspace && workspace_floats >= need) ? workspace : nullptr;
  VS_CHECK_ARG(!workspace || p.slabs, "gemm_fp8_tn: workspace too small (%lld < %lld floats)", (long long)workspace_floats, need);
  p.direct = p.splits == 1;
  if (p.direct) p.slabs = nullptr;
  static VsOnce attr_done{false};
  if (!attr_done.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_fp8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_tn_fp8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_fp8_tn: cannot raise dynamic LDS: %s", hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done.store(true, std::memory_order_relaxed);
  }
  hipStream_t s = (hipStream_t)stream;
  if (tn8_pp_enabled())
    hipLaunchKernelGGL(gemm_tn_fp8_kernel<true>, dim3(p.tiles1 * p.tiles2 * p.splits), dim3(TN_THREADS), TN_LDS_BYTES, s, p);
  else
    hipLaunchKernelGGL(gemm_tn_fp8_kernel<false>, dim3(p.tiles1 * p.tiles2 * p.splits), dim3(TN_THREADS), TN_LDS_BYTES, s, p);
  VS_CHECK_LAUNCH("gemm_fp8_tn");
  if (p.slabs) {
    const long long n4 = (long long)N1 * N2 / 4;
    long long grid = (n4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, s, C, p.slabs, n4, (long long)N1 * N2, p.splits);
    VS_CHECK_LAUNCH("gemm_fp8_tn_reduce");
  }
  return VITSSL_OK;
}

namespace {
int tn8_batch_fill(const vitssl_fp8_tn_job_t* jobs, int njobs, Tn8BatchParams* pp) {
  TnBatchParams& p = pp->b;
  int t = 0;
  for (int j = 0; j < TN_MAX_JOBS; ++j) {
    const bool live = j < njobs;
    p.A[j] = live ? (const bf16_t*)jobs[j].A8 : nullptr;
    p.B[j] = live ? (const bf16_t*)jobs[j].B8 : nullptr;
    p.C[j] = live ? jobs[j].C : nullptr;
    p.N1[j] = live ? jobs[j].N1 : 0;
    p.N2[j] = live ? jobs[j].N2 : 0;
    pp->alpha[j] = live ? jobs[j].alpha : nullptr;
    pp->alpha2[j] = live ? jobs[j].alpha2 : nullptr;
    p.tile0[j] = t;
    p.tiles2[j] = live ? (jobs[j].N2 + TN_T - 1) / TN_T : 0;
    if (live) t += ((jobs[j].N1 + TN_T - 1) / TN_T) * p.tiles2[j];
  }
  p.tile0[TN_MAX_JOBS] = t;
  for (int j = njobs; j <= TN_MAX_JOBS; ++j) p.tile0[j] = t;
  p.njobs = njobs;
  return t;
}
}  // namespace

extern "C" int64_t vitssl_gemm_fp8_tn_batch_workspace_floats(const vitssl_fp8_tn_job_t* jobs, int njobs, int64_t M) {
  if (!jobs || njobs <= 0 || njobs > TN_MAX_JOBS || M <= 0) return 0;
  Tn8BatchParams pp;
  const int T = tn8_batch_fill(jobs, njobs, &pp);
  int sp, cps, rem;
  tn_batch_plan(M, T, vitssl_persistent_cus(), &sp, &cps, &rem, TN8_KM);
  return (sp > 1 || rem > 0) ? (int64_t)(sp + (rem > 0)) * T * TN_T * TN_T : 0;
}

extern "C" int vitssl_gemm_fp8_tn_batch(const vitssl_fp8_tn_job_t* jobs, int njobs, int64_t M, float* workspace, int64_t workspace_floats,
                                        void* stream) {
  VS_CHECK_ARG(jobs && njobs > 0 && njobs <= TN_MAX_JOBS, "gemm_fp8_tn_batch: 1..%d jobs", TN_MAX_JOBS);
  VS_CHECK_ARG(M > 0, "gemm_fp8_tn_batch: empty problem");
  for (int j = 0; j < njobs; ++j) {
    const vitssl_fp8_tn_job_t& q = jobs[j];
    VS_CHECK_ARG(q.A8 && q.B8 && q.C, "gemm_fp8_tn_batch: job %d: null operand", j);
    VS_CHECK_ARG(q.N1 > 0 && q.N2 > 0 && q.N1 % 16 == 0 && q.N2 % 16 == 0, "gemm_fp8_tn_batch: job %d: N1=%d N2=%d must be positive multiples of 16",
                 j, q.N1, q.N2);
    VS_CHECK_ARG((unsigned long long)M * q.N1 < (1ull << 31) && (unsigned long long)M * q.N2 < (1ull << 31),
                 "gemm_fp8_tn_batch: job %d: operand larger than 2 GiB", j);
  }
  Tn8BatchParams pp;
  TnBatchParams& p = pp.b;
  const int T = tn8_batch_fill(jobs, njobs, &pp);
  p.M = M;
  const long long cus = vitssl_persistent_cus();               // once per launch: plan, workspace check and grid use this value
  tn_batch_plan(M, T, (int)cus, &p.splits, &p.chunks_per_split, &p.rem_chunks, TN8_KM);
  const long long need = (p.splits > 1 || p.rem_chunks > 0) ? (long long)(p.splits + (p.rem_chunks > 0)) * T * TN_T * TN_T : 0;
  VS_CHECK_ARG(need == 0 || (workspace && workspace_floats >= need),
               "gemm_fp8_tn_batch: workspace of %lld floats needed (vitssl_gemm_fp8_tn_batch_workspace_floats)", need);
  p.slots = need ? workspace : nullptr;
  static VsOnce attr_done{false};
  if (!attr_done.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_fp8_batch_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_tn_fp8_batch_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_fp8_tn_batch: cannot raise dynamic LDS: %s", hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done.store(true, std::memory_order_relaxed);
  }
  hipStream_t s = (hipStream_t)stream;
  const long long units = (long long)T * p.splits;
  const unsigned grid = (unsigned)((units < cus && p.rem_chunks == 0) ? units : cus);
  if (tn8_pp_enabled())
    hipLaunchKernelGGL(gemm_tn_fp8_batch_kernel<true>, dim3(grid), dim3(TN_THREADS), TN_LDS_BYTES, s, pp);
  else
    hipLaunchKernelGGL(gemm_tn_fp8_batch_kernel<false>, dim3(grid), dim3(TN_THREADS), TN_LDS_BYTES, s, pp);
  VS_CHECK_LAUNCH("gemm_fp8_tn_batch");
  if (p.slots) {
    hipLaunchKernelGGL(tn_batch_reduce_kernel, dim3((unsigned)T * 8u), dim3(256), 0, s, p);
    VS_CHECK_LAUNCH("gemm_fp8_tn_batch_reduce");
  }
  return VITSSL_OK;
}
