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
//
// Two tile configurations share this code (every wave always owns 128x64 outputs):
//   BIG   256x256x64, 8 waves, 128 KiB LDS, one workgroup per CU: best main loop;
//   SMALL 256x128x32, 4 waves,  48 KiB LDS, up to three workgroups per CU: the VALU- and
//         HBM-heavy epilogue of one workgroup (GELU / dGELU / residual) overlaps the
//         MFMAs of its neighbours, and the finer tiles fill the 256 CUs better when the
//         256x256 grid is a poor multiple of 256 (e.g. N = 768: 588 tiles).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

template <int V>
using IC = std::integral_constant<int, V>;

template <int BK_, int WM_, int WN_, int NBUF_ = 2>
struct NtCfg {
  static constexpr int BK = BK_, WM = WM_, WN = WN_, NBUF = NBUF_;
  static constexpr int BM = 128 * WM_, BN = 64 * WN_;
  static constexpr int WAVES = WM_ * WN_, THREADS = 64 * WM_ * WN_;
  static constexpr int ROWB = BK_ * 2;                       // bytes per tile row
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  static constexpr int BUF_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = NBUF_ * BUF_BYTES;
  static constexpr int MIN_WAVES_PER_SIMD = 2;
};
using NtBig = NtCfg<64, 2, 4>;
using NtSmall = NtCfg<32, 2, 2>;
// 256x256 tile, 32-deep stages in a 4-slot LDS ring: three stages (96 KiB per CU) stay in
// flight behind a counted vmcnt and a raw s_barrier, instead of draining to zero each step.
using NtDeep = NtCfg<32, 2, 4, 4>;

struct NtParams {
  const bf16_t* A;
  const bf16_t* B;
  long long M;        // one past the last row this launch may touch
  long long m_begin;  // first row of this launch (row window [m_begin, M): tail peeling)
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
  int group_n;   // tile columns per raster group (their B panels stay L2-resident)
  int stagger;   // start-up delay units (x ~2.7 us) per phase group, 0 = off
  int phases;    // MFMA clusters per K-step in the BIG loop: 2 or 4
  int l2_prefetch;
};

// XOR applied to the 16-byte chunk index of tile row r (source side for the DMA, and on
// the fragment reads): BK=64 (128-B rows) chunk ^ ((r>>1)&7); BK=32 (64-B rows)
// chunk ^ 3*((r>>3)&1).  Both make every 16-lane ds_read_b128 group conflict-free.
template <int BK_>
__device__ __forceinline__ int nt_swz(int r) {
  return BK_ == 64 ? ((r >> 1) & 7) : 3 * ((r >> 3) & 1);
}

// PART / NPARTS: issue only that share of this wave's DMA instructions (spreads the stage
// of one tile over several MFMA clusters)
template <int BK_, int ROWS, int WAVES, int PART = 0, int NPARTS = 1>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, long long row0, int k0,
                                           int K, int wave, int lane, const bf16_t* gbase = nullptr, long long nrows = 0) {
  constexpr int ROWB = BK_ * 2;
  constexpr int RPI = 1024 / ROWB;              // tile rows per 1-KiB wave-instruction
  constexpr int LPR = ROWB / 16;                // lanes per row
  constexpr int SLOTS = ROWS / RPI;
  static_assert(SLOTS % WAVES == 0, "tile rows must split evenly over the waves");
  constexpr int PER = SLOTS / WAVES;
  static_assert(PER % NPARTS == 0, "DMA instructions must split evenly over the parts");
#pragma unroll
  for (int j = PART * (PER / NPARTS); j < (PART + 1) * (PER / NPARTS); ++j) {
    const int i = wave * PER + j;               // wave-uniform instruction slot
    const int r = i * RPI + lane / LPR;         // tile row this lane fetches for
    const int c = lane % LPR;                   // 16-B chunk position in the LDS row
    const int sc = c ^ nt_swz<BK_>(r);          // chunk fetched from global
#ifdef VS_NT_GLOBAL_LDS
    long long gr = row0 + r;
    if (gr > nrows - 1) gr = nrows - 1;   // clamp: duplicated rows only feed outputs that are never stored
    const char* gp = (const char*)gbase + (gr * (long long)K + k0) * 2 + sc * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, LDS_PTR(lds_tile + i * 1024), 16, 0, 0);
#else
    const unsigned voff = (unsigned)(((row0 + r) * (long long)K + k0) * 2 + sc * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + i * 1024), 16, voff, 0, 0, 0);
#endif
  }
}

// Half-tile staging for the split-half pipeline (BK = 64, 8 waves, 256-row tiles).  A
// tile row belongs to the LO half when (row % PERIOD) < PERIOD/2 -- PERIOD = 128 for the A
// tile (per-wave m-reps 0..3), 64 for the B tile (per-wave n-reps 0..1).  Each half is 16
// one-KiB DMA instructions = 2 per wave.
template <int PERIOD, int HI>
__device__ __forceinline__ void stage_half(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, long long row0, int k0,
                                           int K, int wave, int lane) {
  constexpr int SPP = PERIOD / 8;               // instruction slots per period
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int h = wave * 2 + j;                 // 0..15: index inside this half
    const int i = (h / (SPP / 2)) * SPP + HI * (SPP / 2) + (h % (SPP / 2));   // slot (8 rows each)
    const int r = i * 8 + (lane >> 3);
    const int c = lane & 7;
    const int sc = c ^ ((r >> 1) & 7);
    const unsigned voff = (unsigned)(((row0 + r) * (long long)K + k0) * 2 + sc * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + i * 1024), 16, voff, 0, 0, 0);
  }
}

template <int EPI, typename CFG, int PH>
__global__ __launch_bounds__(CFG::THREADS, CFG::MIN_WAVES_PER_SIMD) void gemm_nt_kernel(NtParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK;

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (round-robin dispatch);
  // give each XCD a contiguous run of tiles so neighbouring tiles (same A row-panel,
  // weights) hit the same L2.  Speed only, never correctness.
  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  const int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
  // column-group-major order: all tile rows of a group of `group_n` tile columns, then
  // the next group -> a group's weight panels are re-used from L2 by every tile row.
  const int full = p.tiles_m * p.group_n;
  const int cg = wgid / full;
  const int rem = wgid - cg * full;
  const int gw = min(p.group_n, p.tiles_n - cg * p.group_n);
  const int tile_m = rem / gw;
  const int tile_n = cg * p.group_n + (rem - tile_m * gw);
  const long long m0 = p.m_begin + (long long)tile_m * BM;
  const int n0 = tile_n * BN;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / CFG::WN, wn = wave % CFG::WN;

  // Phase stagger: every tile costs the same, so the 256 CUs would run their
  // HBM-heavy epilogues in lock-step (a chip-wide burst, then an idle HBM during the
  // main loops).  Delaying the first wave of workgroups by 0..3 quarter tile periods
  // spreads the epilogues of different CUs over the others' main loops.
  if (p.stagger > 0 && bid < 256) {
    const int grp = (bid >> 3) & 3;
    for (int i = 0; i < grp * p.stagger; ++i) __builtin_amdgcn_s_sleep(100);
  }

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
  const int swz = nt_swz<BK>(lane & 15);           // rows are 16*x + (lane&15)
  const int frag_row = lane & 15;
  const int kq = lane >> 4;

  auto compute_stage = [&](const char* bufA, const char* bufB) {
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int coff = ((kk * 4 + kq) ^ swz) << 4;
      bf16x8 fb[4], fa[8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fb[j] = *(const bf16x8*)(bufB + (wn * 64 + j * 16 + frag_row) * CFG::ROWB + coff);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        fa[i] = *(const bf16x8*)(bufA + (wm * 128 + i * 16 + frag_row) * CFG::ROWB + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    }
  };

  auto compute_half = [&](const char* bufA, const char* bufB, auto kk_c) {
    constexpr int kk = decltype(kk_c)::value;
    const int coff = ((kk * 4 + kq) ^ swz) << 4;
    bf16x8 fb[4], fa[8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fb[j] = *(const bf16x8*)(bufB + (wn * 64 + j * 16 + frag_row) * CFG::ROWB + coff);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      fa[i] = *(const bf16x8*)(bufA + (wm * 128 + i * 16 + frag_row) * CFG::ROWB + coff);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  // 16-MFMA cluster: k-step kk, accumulator rows ih*4 .. ih*4+3
  // (kk and ih are compile-time constants: a runtime index would push acc[] to scratch)
  auto compute_quarter = [&](const char* bufA, const char* bufB, auto kk_c, auto ih_c) {
    constexpr int kk = decltype(kk_c)::value, ih = decltype(ih_c)::value;
    const int coff = ((kk * 4 + kq) ^ swz) << 4;
    bf16x8 fb[4], fa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fb[j] = *(const bf16x8*)(bufB + (wn * 64 + j * 16 + frag_row) * CFG::ROWB + coff);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      fa[i] = *(const bf16x8*)(bufA + (wm * 128 + (ih * 4 + i) * 16 + frag_row) * CFG::ROWB + coff);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[j][ih * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][ih * 4 + i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  if constexpr (CFG::NBUF == 2 && PH == 3 && BK == 64 && CFG::WAVES == 8) {
    // ---- split-half pipeline ------------------------------------------------------
    // step t:  [B0] issue LO(t+1) | Q_ll(t) | vmcnt(4): HI(t) landed [B1] issue HI(t+1) |
    //          Q_lh, Q_hl, Q_hh(t) | vmcnt(4): LO(t+1) landed [B0 of t+1] ...
    // Each half has a full step to land and the DMA queue never drains to zero.
    auto frA = [&](const char* bufA, int kk, int i) {
      return *(const bf16x8*)(bufA + (wm * 128 + i * 16 + frag_row) * CFG::ROWB + (((kk * 4 + kq) ^ swz) << 4));
    };
    auto frB = [&](const char* bufB, int kk, int j) {
      return *(const bf16x8*)(bufB + (wn * 64 + j * 16 + frag_row) * CFG::ROWB + (((kk * 4 + kq) ^ swz) << 4));
    };
    // prologue: tile 0 (both halves), wait, then LO(1) is issued inside step 0
    stage_half<128, 0>(rsA, smem, m0, 0, p.K, wave, lane);
    stage_half<64, 0>(rsB, smem + CFG::A_BYTES, n0, 0, p.K, wave, lane);
    stage_half<128, 1>(rsA, smem, m0, 0, p.K, wave, lane);
    stage_half<64, 1>(rsB, smem + CFG::A_BYTES, n0, 0, p.K, wave, lane);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // LO(0) landed
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nk; ++t) {
      const char* bufA = smem + (t & 1) * CFG::BUF_BYTES;
      const char* bufB = bufA + CFG::A_BYTES;
      char* nA = smem + ((t + 1) & 1) * CFG::BUF_BYTES;
      const bool more = t + 1 < nk;
      if (more) {
        stage_half<128, 0>(rsA, nA, m0, (t + 1) * BK, p.K, wave, lane);
        stage_half<64, 0>(rsB, nA + CFG::A_BYTES, n0, (t + 1) * BK, p.K, wave, lane);
      }
      bf16x8 alo[2][4], blo[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < 2; ++j) blo[kk][j] = frB(bufB, kk, j);
#pragma unroll
        for (int i = 0; i < 4; ++i) alo[kk][i] = frA(bufA, kk, i);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo[kk][j], alo[kk][i], acc[j][i], 0, 0, 0);
      // HI(t) must have landed (every wave's share): allow only the LO(t+1) DMAs in flight
      if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (more) {
        stage_half<128, 1>(rsA, nA, m0, (t + 1) * BK, p.K, wave, lane);
        stage_half<64, 1>(rsB, nA + CFG::A_BYTES, n0, (t + 1) * BK, p.K, wave, lane);
      }
      bf16x8 bhi[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) bhi[kk][j] = frB(bufB, kk, 2 + j);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[2 + j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi[kk][j], alo[kk][i], acc[2 + j][i], 0, 0, 0);
      bf16x8 ahi[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) ahi[kk][i] = frA(bufA, kk, 4 + i);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[j][4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo[kk][j], ahi[kk][i], acc[j][4 + i], 0, 0, 0);
            acc[2 + j][4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi[kk][j], ahi[kk][i], acc[2 + j][4 + i], 0, 0, 0);
          }
      // LO(t+1) must have landed before the next step reads it; HI(t+1) may stay in flight
      if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else if constexpr (CFG::NBUF == 2) {
    stage_tile<BK, BM, CFG::WAVES>(rsA, smem, m0, 0, p.K, wave, lane, p.A, p.M);
    stage_tile<BK, BN, CFG::WAVES>(rsB, smem + CFG::A_BYTES, n0, 0, p.K, wave, lane, p.B, p.N);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      char* bufA = smem + (t & 1) * CFG::BUF_BYTES;
      char* nA = smem + ((t + 1) & 1) * CFG::BUF_BYTES;
      const bool more = t + 1 < nk;
      if constexpr (BK == 64 && CFG::WAVES == 8) {
        // L2 warm-up of the streamed A panel two K-steps ahead: one throw-away dword per
        // 64-byte segment (256 rows x 2 segments = 512 threads).  The first workgroup to
        // touch a line pays the HBM latency here instead of in the LDS-DMA of step t+1.
        // (asm: the result register is never read; the load is retired by the manual
        // vmcnt(0) at the end of this step.)
        if (p.l2_prefetch && t + 2 < nk) {
          const unsigned poff = (unsigned)(((m0 + (threadIdx.x >> 1)) * (long long)p.K + (t + 2) * BK) * 2 + (threadIdx.x & 1) * 64);
          unsigned dummy;
          asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dummy) : "v"(poff), "s"(rsA) : "memory");
        }
      }
      if constexpr (BK == 64) {
        // spread the next tile's DMA over the step: A before the first MFMA block, B between
        // the two blocks (a single 64-KiB burst right after the barrier queues in the TA)
        if constexpr (PH == 4) {
          if (more) stage_tile<BK, BM, CFG::WAVES, 0, 2>(rsA, nA, m0, (t + 1) * BK, p.K, wave, lane);
          compute_quarter(bufA, bufA + CFG::A_BYTES, IC<0>{}, IC<0>{});
          if (more) stage_tile<BK, BM, CFG::WAVES, 1, 2>(rsA, nA, m0, (t + 1) * BK, p.K, wave, lane);
          compute_quarter(bufA, bufA + CFG::A_BYTES, IC<0>{}, IC<1>{});
          if (more) stage_tile<BK, BN, CFG::WAVES, 0, 2>(rsB, nA + CFG::A_BYTES, n0, (t + 1) * BK, p.K, wave, lane);
          compute_quarter(bufA, bufA + CFG::A_BYTES, IC<1>{}, IC<0>{});
          if (more) stage_tile<BK, BN, CFG::WAVES, 1, 2>(rsB, nA + CFG::A_BYTES, n0, (t + 1) * BK, p.K, wave, lane);
          compute_quarter(bufA, bufA + CFG::A_BYTES, IC<1>{}, IC<1>{});
        } else {
          if (more) stage_tile<BK, BM, CFG::WAVES>(rsA, nA, m0, (t + 1) * BK, p.K, wave, lane, p.A, p.M);
          compute_half(bufA, bufA + CFG::A_BYTES, IC<0>{});
          if (more) stage_tile<BK, BN, CFG::WAVES>(rsB, nA + CFG::A_BYTES, n0, (t + 1) * BK, p.K, wave, lane, p.B, p.N);
          compute_half(bufA, bufA + CFG::A_BYTES, IC<1>{});
        }
      } else {
        if (more) {
          stage_tile<BK, BM, CFG::WAVES>(rsA, nA, m0, (t + 1) * BK, p.K, wave, lane);
          stage_tile<BK, BN, CFG::WAVES>(rsB, nA + CFG::A_BYTES, n0, (t + 1) * BK, p.K, wave, lane);
        }
        compute_stage(bufA, bufA + CFG::A_BYTES);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // LDS ring of NBUF stages; stage t lives in slot t % NBUF.  Per wave a stage is
    // GL = 4 + 2*... DMA instructions (A then B), so "all but the 2 youngest stages
    // have landed" is vmcnt(2*GL).
    constexpr int GL = (BM * CFG::ROWB / 1024 + BN * CFG::ROWB / 1024) / CFG::WAVES;
    static_assert(GL == 4, "counted vmcnt immediates below assume 4 DMA instructions per stage per wave");
    auto issue = [&](int t) {
      char* dst = smem + (t % CFG::NBUF) * CFG::BUF_BYTES;
      stage_tile<BK, BM, CFG::WAVES>(rsA, dst, m0, t * BK, p.K, wave, lane);
      stage_tile<BK, BN, CFG::WAVES>(rsB, dst + CFG::A_BYTES, n0, t * BK, p.K, wave, lane);
    };
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    for (int t = 0; t < nk; ++t) {
      // stage t must have landed; stages t+1, t+2 (if they exist) may stay in flight
      const int younger = nk - 1 - t;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // every wave's share of stage t has landed AND every wave has finished reading
      // slot (t-1) % NBUF, which the next DMA overwrites
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 3 < nk) issue(t + 3);
      const char* bufA = smem + (t % CFG::NBUF) * CFG::BUF_BYTES;
      compute_stage(bufA, bufA + CFG::A_BYTES);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ------------------------------------------------------------------ epilogue
  // All global traffic of the epilogue goes through raw buffer instructions on a window
  // that starts at the tile's first row: rows past M fall outside num_records and columns
  // past N get the out-of-range offset, so loads return 0 and stores are dropped WITHOUT a
  // branch.  That lets every residual / g' load of a 64-column half be issued back to back
  // before the first use (the branchy form waited for each 16-byte load in turn: 32
  // dependent HBM round trips per wave, measured +65 us on the N = K = 768 projection).
  float csum[4][4];
  if (p.colsum) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) csum[j][r] = 0.f;
  }
  const int g4 = lane >> 4;                       // lane group = 16-lane row of the wave
  // bf16 images: tile columns (j, j+1) exchange halves between lane rows (g, g^1) with
  // v_permlane16_swap so that every lane moves 16 contiguous bytes (8 columns).
  const bool wide = (p.N & 7) == 0;
  constexpr unsigned OOB = 0x80000000u;
  const long long rows_left = p.M - m0;
  auto window = [&](const void* base, int elt) {
    const unsigned long long bytes = (unsigned long long)rows_left * (unsigned long long)p.N * (unsigned)elt;
    const unsigned rec = bytes > 0x80000000ull ? 0x80000000u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + m0 * p.N * elt), 0, (int)rec, 0x00020000);
  };
  const unsigned row_l = (unsigned)(wm * 128 + (lane & 15));      // + 16 i : row inside the tile
  const unsigned un = (unsigned)p.N;
  const int odd = g4 & 1;

  // byte offset of this lane's 16-byte piece of the bf16 image (wide form), row i, pair jp
  auto off_bf16_wide = [&](int i, int jp) -> unsigned {
    const int n = n0 + wn * 64 + (2 * jp + odd) * 16 + 4 * (g4 - odd);
    return n < p.N ? ((row_l + 16u * i) * un + (unsigned)n) * 2u : OOB;
  };
  auto off_elem = [&](int i, int n, unsigned elt) -> unsigned {
    return n < p.N ? ((row_l + 16u * i) * un + (unsigned)n) * elt : OOB;
  };
  auto store_bf16_pair = [&](__amdgpu_buffer_rsrc_t rs, int i, int jp, const u32x2& w0, const u32x2& w1) {
    if (wide) {
      // after the swap: even rows hold tile 2jp  cols 4g .. 4g+7, odd rows tile 2jp+1 cols 4(g-1) .. 4(g-1)+7
      auto lo = __builtin_amdgcn_permlane16_swap(w0[0], w1[0], false, false);
      auto hi = __builtin_amdgcn_permlane16_swap(w0[1], w1[1], false, false);
      const u32x4 v = {lo[0], hi[0], lo[1], hi[1]};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, off_bf16_wide(i, jp), 0, 0);
    } else {
      const int na = n0 + wn * 64 + (2 * jp) * 16 + 4 * g4;
      __builtin_amdgcn_raw_buffer_store_b64(w0, rs, off_elem(i, na, 2u), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(w1, rs, off_elem(i, na + 16, 2u), 0, 0);
    }
  };

  __amdgpu_buffer_rsrc_t rsOut0, rsOut1, rsAux;
  if constexpr (EPI == VITSSL_EPI_BF16 || EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_DGELU) rsOut0 = window(p.out0, 2);
  if constexpr (EPI == VITSSL_EPI_F32 || EPI == VITSSL_EPI_RESID) rsOut0 = window(p.out0, 4);
  if constexpr (EPI == VITSSL_EPI_GELU) rsOut1 = window(p.out1, 2);
  if constexpr (EPI == VITSSL_EPI_RESID) rsAux = window(p.aux, 4);
  if constexpr (EPI == VITSSL_EPI_DGELU) rsAux = window(p.aux, 2);

#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const int nA = n0 + wn * 64 + (2 * jp) * 16 + 4 * g4;
    const int nB = nA + 16;
    const bool okA = nA < p.N, okB = nB < p.N;
    f32x4 biasA = {0.f, 0.f, 0.f, 0.f}, biasB = biasA;
    if (p.bias) {
      if (okA) biasA = *(const f32x4*)(p.bias + nA);
      if (okB) biasB = *(const f32x4*)(p.bias + nB);
    }
    const int nn[2] = {nA, nB};
    const bool okn[2] = {okA, okB};

    // ---- operand prefetch for this half: every load in flight before the first use
    f32x4 res[8][2];     // RESID: residual stream
    u32x2 gpre[8][2];    // DGELU: g' in accumulator layout
    if constexpr (EPI == VITSSL_EPI_RESID) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          res[i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsAux, off_elem(i, nn[h], 4u), 0, 0));
    }
    if constexpr (EPI == VITSSL_EPI_DGELU) {
      if (wide) {
        u32x4 raw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsAux, off_bf16_wide(i, jp), 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          // inverse of the store shuffle (the swap is an involution)
          auto a = __builtin_amdgcn_permlane16_swap(raw[i][0], raw[i][2], false, false);
          auto b = __builtin_amdgcn_permlane16_swap(raw[i][1], raw[i][3], false, false);
          gpre[i][0] = u32x2{a[0], b[0]};
          gpre[i][1] = u32x2{a[1], b[1]};
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) gpre[i][h] = __builtin_amdgcn_raw_buffer_load_b64(rsAux, off_elem(i, nn[h], 2u), 0, 0);
      }
    }

#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long m = m0 + wm * 128 + i * 16 + (lane & 15);
      const bool okm = m < p.M;
      f32x4 v[2] = {acc[2 * jp][i] + biasA, acc[2 * jp + 1][i] + biasB};

      if constexpr (EPI == VITSSL_EPI_BF16) {
        const u32x2 w0 = {pack_bf2(v[0][0], v[0][1]), pack_bf2(v[0][2], v[0][3])};
        const u32x2 w1 = {pack_bf2(v[1][0], v[1][1]), pack_bf2(v[1][2], v[1][3])};
        store_bf16_pair(rsOut0, i, jp, w0, w1);
      } else if constexpr (EPI == VITSSL_EPI_GELU) {
        // u = bf16(acc + bias) (never stored); out1 = a = keep*scale*gelu(u) feeds the next
        // GEMM; out0 = g' = keep*scale*gelu'(u) is what the backward dGELU epilogue needs.
        u32x2 gp[2], a[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float mult[4] = {1.f, 1.f, 1.f, 1.f};
          if (p.drop_on) drop_mult4(p.dk, (unsigned long long)(m * p.N + nn[h]) >> 2, mult);
          float y[4], d[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gelu_both(round_bf(v[h][r]), y[r], d[r]);
            y[r] *= mult[r];
            d[r] *= mult[r];
          }
          a[h] = u32x2{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
          gp[h] = u32x2{pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])};
        }
        store_bf16_pair(rsOut0, i, jp, gp[0], gp[1]);
        store_bf16_pair(rsOut1, i, jp, a[0], a[1]);
      } else if constexpr (EPI == VITSSL_EPI_DGELU) {
        // du = acc * g'  (g' already carries the dropout mask and its scale)
        u32x2 w[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x2 gpv = gpre[i][h];
          v[h][0] *= bf_lo(gpv[0]);
          v[h][1] *= bf_hi(gpv[0]);
          v[h][2] *= bf_lo(gpv[1]);
          v[h][3] *= bf_hi(gpv[1]);
          w[h] = u32x2{pack_bf2(v[h][0], v[h][1]), pack_bf2(v[h][2], v[h][3])};
        }
        store_bf16_pair(rsOut0, i, jp, w[0], w[1]);
      } else if constexpr (EPI == VITSSL_EPI_F32) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[h]), rsOut0, off_elem(i, nn[h], 4u), 0, 0);
      } else if constexpr (EPI == VITSSL_EPI_RESID) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (p.drop_on) {
            float mult[4];
            drop_mult4(p.dk, (unsigned long long)(m * p.N + nn[h]) >> 2, mult);
            v[h][0] *= mult[0]; v[h][1] *= mult[1]; v[h][2] *= mult[2]; v[h][3] *= mult[3];
          }
          v[h] += res[i][h];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[h]), rsOut0, off_elem(i, nn[h], 4u), 0, 0);
        }
      } else {   // VITSSL_EPI_EMBED (one launch per step: plain addressing)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (!(okm && okn[h])) continue;
          const long long img = m / p.embed.tokens;
          const int rin = (int)(m - img * p.embed.tokens);
          if (p.embed.mask && p.embed.mask[m]) v[h] = *(const f32x4*)(p.embed.mask_token + nn[h]);
          v[h] += *(const f32x4*)(p.embed.pos + (long long)(p.embed.tok_offset + rin) * p.N + nn[h]);
          const long long orow = img * p.embed.out_tokens + p.embed.tok_offset + rin;
          *(f32x4*)((float*)p.out0 + orow * p.N + nn[h]) = v[h];
        }
      }
      if (p.colsum) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (okm && okn[h]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) csum[2 * jp + h][r] += v[h][r];
          }
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

template <int EPI, typename CFG, int PH = 2>
int launch_cfg(NtParams p, hipStream_t s) {
  static bool attr_done = false;  // idempotent; a benign race sets the same value
  if (!attr_done && CFG::LDS_BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, CFG, PH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       CFG::LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_nt: cannot raise dynamic LDS to %d: %s", CFG::LDS_BYTES, hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done = true;
  }
  p.tiles_m = (int)ceil_div64(p.M - p.m_begin, CFG::BM);
  p.tiles_n = (int)ceil_div64(p.N, CFG::BN);
  const int want = (int)((2 * 1024 * 1024) / ((long long)CFG::BN * p.K * 2));   // panels of a group <= ~2 MiB of L2
  if (p.tiles_n <= 4) p.group_n = p.tiles_n;
  else if (want <= 2) p.group_n = 2;
  else if (p.tiles_n % 6 == 0 && want >= 6) p.group_n = 6;
  else if (p.tiles_n % 4 == 0) p.group_n = 4;
  else if (p.tiles_n % 3 == 0) p.group_n = 3;
  else p.group_n = want < 4 ? want : 4;
  {
    static int knob = -1;
    if (knob < 0) {
      const char* e = getenv("VITSSL_NT_STAGGER");
      knob = e ? atoi(e) : 0;   // measured: no gain (per-CU memory throughput, not lock-step bursts, bounds the epilogue)
    }
    const bool heavy = EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_DGELU || EPI == VITSSL_EPI_RESID || EPI == VITSSL_EPI_F32;
    const int nk = p.K / CFG::BK;
    p.stagger = (knob && heavy && p.tiles_m * p.tiles_n > 256) ? (nk * knob + 5) / 6 : 0;
    p.phases = PH;
    static int pf = -1;
    if (pf < 0) {
      const char* e = getenv("VITSSL_NT_L2PF");
      pf = e ? atoi(e) : 0;
    }
    p.l2_prefetch = pf;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, CFG, PH>), dim3(p.tiles_m * p.tiles_n), dim3(CFG::THREADS), CFG::LDS_BYTES, s, p);
  VS_CHECK_LAUNCH("gemm_nt");
  return VITSSL_OK;
}

// 0 = auto, 1 = always BIG, 2 = always SMALL, 3 = always DEEP (VITSSL_NT_TILE, developer knob)
int nt_tile_override() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VITSSL_NT_TILE");
    v = e ? atoi(e) : 0;
  }
  return v;
}

template <int EPI>
int launch_big(const NtParams& p, hipStream_t s) {
  static int ph = -1;   // MFMA clusters per K-step in the BIG loop (VITSSL_NT_PHASES, developer knob)
  if (ph < 0) {
    const char* e = getenv("VITSSL_NT_PHASES");
    ph = e ? atoi(e) : 2;
  }
  if (ph == 3) return launch_cfg<EPI, NtBig, 3>(p, s);
  return ph == 4 ? launch_cfg<EPI, NtBig, 4>(p, s) : launch_cfg<EPI, NtBig, 2>(p, s);
}

// Tail peeling.  Every 256x256 tile costs the same and one workgroup owns a CU, so a grid of
// T tiles takes ceil(T / 256) rounds: 588 tiles (N = 768, M = 50176) pay for 3 rounds and use
// 2.3.  When the last round is badly filled, the BIG kernel takes the tile rows that make
// whole rounds and the remaining rows go to a second launch with the 256x128 SMALL tiles
// (half the work per tile, co-resident workgroups), which costs about two thirds of a round.
// Row indices stay global in both launches (dropout counters, embedding rows, bounds).
// Measured (tools/bench_gemm.py, M = 50176): N=768 K=768 75.4 -> 78.2 us, N=3072 K=768
// 256 -> 264 us, N=768 K=3072 232 -> 244 us: the "rounds" model is wrong for this kernel --
// it is bound by the L2 -> LDS request path, so the tiles of an under-filled last round run
// faster, and the second launch adds a drain + ramp.  Kept as a developer knob, default off.
// Returns the number of rows for the BIG launch, or M when peeling does not pay.
long long peel_rows(long long M, int N) {
  static int knob = -1;
  if (knob < 0) {
    const char* e = getenv("VITSSL_NT_PEEL");
    knob = e ? atoi(e) : 0;   // measured: 2-9 % SLOWER on every ViT-B shape (see below)
  }
  if (!knob) return M;
  const long long tm = ceil_div64(M, 256), tn = ceil_div64(N, 256);
  const long long tiles = tm * tn;
  const long long full = tiles / 256;
  if (full < 1 || tiles % 256 == 0) return M;
  const long long tm_big = full * 256 / tn;
  if (tm_big < 1 || tm_big >= tm) return M;
  const long long rest_rows = M - tm_big * 256;
  const long long small_tiles = ceil_div64(rest_rows, 256) * ceil_div64(N, 128);
  const double tail = 0.68 * (double)ceil_div64(small_tiles, 256);
  return ((double)full + tail < (double)(full + 1) - 0.15) ? tm_big * 256 : M;
}

template <int EPI>
int launch_nt(const NtParams& p, hipStream_t s) {
  int mode = nt_tile_override();
  if (mode == 1) return launch_big<EPI>(p, s);
  if (mode == 2) return launch_cfg<EPI, NtSmall>(p, s);
  if (mode == 3) return launch_cfg<EPI, NtDeep>(p, s);
  // Measured on MI355X (tools/bench_gemm.py, round 1): as the ONLY tile, SMALL loses 10-25 %
  // on every ViT-B shape, heavy epilogues included.  It pays for tiny grids and for tails.
  const long long big_tiles = ceil_div64(p.M, 256) * ceil_div64(p.N, 256);
  if (big_tiles < 64) return launch_cfg<EPI, NtSmall>(p, s);
  const long long rows_big = peel_rows(p.M, p.N);
  if (rows_big >= p.M) return launch_big<EPI>(p, s);
  NtParams head = p, tail = p;
  head.M = rows_big;
  tail.m_begin = rows_big;
  int rc = launch_big<EPI>(head, s);
  if (rc != VITSSL_OK) return rc;
  return launch_cfg<EPI, NtSmall>(tail, s);
}

}  // namespace

extern "C" int vitssl_gemm_bf16_nt(const vitssl_gemm_t* g, void* stream) {
  VS_CHECK_ARG(g && g->A && g->B && g->out0, "gemm_nt: null operand");
  VS_CHECK_ARG(g->M > 0 && g->N > 0 && g->K > 0, "gemm_nt: empty problem M=%lld N=%d K=%d", (long long)g->M, g->N, g->K);
  VS_CHECK_ARG(g->K % 64 == 0, "gemm_nt: K=%d must be a multiple of 64", g->K);
  VS_CHECK_ARG(g->N % 4 == 0, "gemm_nt: N=%d must be a multiple of 4", g->N);
  VS_CHECK_ARG(g->N <= (1 << 20), "gemm_nt: N=%d exceeds 2^20 (epilogue windows use 32-bit byte offsets)", g->N);
  VS_CHECK_ARG((unsigned long long)g->M * g->K * 2ull < (1ull << 31) && (unsigned long long)g->N * g->K * 2ull < (1ull << 31),
               "gemm_nt: operand larger than 2 GiB (M=%lld N=%d K=%d)", (long long)g->M, g->N, g->K);
  NtParams p;
  p.A = (const bf16_t*)g->A;
  p.B = (const bf16_t*)g->B;
  p.M = g->M;
  p.m_begin = 0;
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
  p.tiles_m = p.tiles_n = p.group_n = 0;   // set per tile configuration in launch_cfg
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
