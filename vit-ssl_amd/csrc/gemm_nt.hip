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
// k-tile t (A before the first 32-MFMA cluster, B before the second) and retired
// (vmcnt(0) + barrier) after them.
//
// Persistent mode (short contractions only by default -- see launch_cfg): the grid
// is one workgroup per CU slot, each workgroup walks tiles b, b + G, ... and the first stage
// of the NEXT tile is issued during the last K-step of the current one, so it lands behind
// the epilogue.  With one tile per workgroup the same loop simply runs once.
//
// Two tile configurations share this code (every wave always owns 128x64 outputs):
//   BIG   256x256x64, 8 waves, 128 KiB LDS, one workgroup per CU: best main loop;
//   SMALL 256x128x32, 4 waves,  48 KiB LDS, two workgroups per CU: tiny grids only.
// Tried and dropped in round 1: 4-slot 32-deep LDS ring, split-half
// and quarter-step DMA schedules, start-up stagger, L2 warm-up loads, global_load_lds,
// peeling the under-filled last round into a SMALL-tile launch.
#include <stdlib.h>
#include <type_traits>
#include <atomic>
#include "common.h"

// cache policy of the epilogue stores (buffer aux bits: 1 = sc0, 2 = nt, 16 = sc1); developer experiments only
#ifndef NT_STORE_AUX
#define NT_STORE_AUX 0
#endif
// residual-stream stores (fp32, read next by a different kernel) are issued non-temporal: interleaved A/B
// on MI355X, out-projection shape M = 50176, N = K = 768: 116 -> 97 us; neutral at K = 3072
#ifndef NT_RESID_AUX
#define NT_RESID_AUX 2
#endif
// bf16 images (activations / gradients consumed by the NEXT kernel): 16 = sc1, write-through without keeping
// the line in this XCD's L2, so the 30-60 MB of output per tile round do not evict the weight panels the
// following rounds re-use (FETCH_SIZE of the FC1 GELU launch 362 -> 297 MB at group 4, 223 MB at group 6)
#ifndef NT_BF16_AUX
#define NT_BF16_AUX 0
#endif
// g' = keep*scale*gelu'(u), the second image of the FC1 epilogue, is read again only in backward: its stores
// are non-temporal (2) so that `a`, which FC2 reads next, keeps its lines (in-step A/B: -0.25 ms per ViT-B step; sc1 = 16: slower)
#ifndef NT_GPRIME_AUX
#define NT_GPRIME_AUX 2
#endif
// 1 (default): the GELU epilogue of the ping-pong kernel reads gelu / gelu' from an LDS table; 0: arithmetic only (A/B builds)
#ifndef NT_GELU_LUT
#define NT_GELU_LUT 1
#endif
// rows (16-row MFMA tiles) whose residual / g' operands are requested together in the RESID / DGELU epilogues
#ifndef NT_RG_RESID
#define NT_RG_RESID 1
#endif
#ifndef NT_RG_DGELU
#define NT_RG_DGELU 2
#endif
// ... with the operand prefetch (two groups alive at once)
#ifndef NT_RG_RESID_PF
#define NT_RG_RESID_PF 1
#endif
#ifndef NT_RG_DGELU_PF
#define NT_RG_DGELU_PF 2
#endif
// 1: the RESID / DGELU epilogues of the ping-pong kernel request the operand lines of the next row group before they store the
// current one (see nt_epilogue)
#ifndef NT_EPI_PREFETCH
#define NT_EPI_PREFETCH 1
#endif
// 1 (default): the ping-pong kernel's epilogue hands every 16-row tile of an output image through a private 2 KiB LDS window
// of the wave, so that a store instruction's 64 lanes cover 8 rows x 128 CONTIGUOUS bytes with adjacent lanes on adjacent
// addresses.  tools/probes/store_patterns.hip: a CU stores 55 GB/s in the accumulator layout (adjacent lanes = adjacent ROWS,
// 16 bytes each: every lane is its own request) and 183-190 GB/s once four or more adjacent lanes are contiguous.
#ifndef NT_LDS_T
#define NT_LDS_T 1
#endif
// start-up stagger window of the ping-pong kernel in tile times (VITSSL_NT_STAGGER overrides)
// 1: the K-tile position of the ping-pong loop's operand DMA lives in the buffer descriptor instead of the lanes' offsets
#ifndef NT_DESC_WINDOW
#define NT_DESC_WINDOW 1
#endif
// 1: s_setprio 1 around the MFMA clusters of the ping-pong loop.  Round 2 measured it neutral; with the leaner loops of round 3 it
// costs 0.5-1.5 % on 15 of 16 shape x epilogue pairs (interleaved A/B, profiles/r03_tls_ab.txt): 8 more scalar instructions per
// K-tile in loops whose LOAD parts are bound by issue slots (see gemm_tn.hip).  Default off.
#ifndef NT_SETPRIO
#define NT_SETPRIO 0
#endif
// epilogues whose launches stagger (bit = VITSSL_EPI_* value).  Spreading the workgroups in time costs L2 sharing (the workgroups
// of a raster group are no longer at the same k): FETCH_SIZE per launch with / without stagger (tools/fetch_ab.sh): plain bf16,
// 224-row tiles 415 / 282 MB, residual 547 / 412, dGELU 692 / 645, GELU 347 / 344.  Whole step (same box, alternating, ms): all
// 33.12, residual + dGELU 33.13, dGELU only 33.26, none 33.42 -> only the two epilogues that are bound by their own traffic stagger.
#ifndef NT_STAGGER_EPIS_DEFAULT
#define NT_STAGGER_EPIS_DEFAULT 0x18
#endif
#ifndef NT_STAGGER_DEFAULT
#define NT_STAGGER_DEFAULT 1.0f
#endif
// diagnostic builds only (tools/build_variant.sh): 1 = epilogue arithmetic and loads but NO stores,
// 2 = stores but no GELU / dropout arithmetic and no residual / g' loads
#ifndef NT_ABLATE
#define NT_ABLATE 0
#endif
// diagnostic builds only: ping-pong K loop without 1 = DMA, 2 = fragment reads, 3 = MFMA
#ifndef NT_LOOP_ABLATE
#define NT_LOOP_ABLATE 0
#endif

namespace {

template <int V>
using IC = std::integral_constant<int, V>;

template <int BK_, int WM_, int WN_, int MI_ = 8>
struct NtCfg {
  static constexpr int BK = BK_, WM = WM_, WN = WN_;
  static constexpr int MI = MI_;                              // 16-row MFMA tiles per wave along M
  static constexpr int WROWS = 16 * MI_;                      // rows per wave
  static constexpr int BM = WROWS * WM_, BN = 64 * WN_;
  static constexpr int WAVES = WM_ * WN_, THREADS = 64 * WM_ * WN_;
  static constexpr int ROWB = BK_ * 2;                       // bytes per tile row
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  static constexpr int BUF_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
  static constexpr int MIN_WAVES_PER_SIMD = 2;
  static constexpr int WG_PER_CU = (WAVES == 8) ? 1 : 2;     // what registers + LDS allow
};
using NtBig = NtCfg<64, 2, 4>;
using NtSmall = NtCfg<32, 2, 2>;
// 224x256x64 and 192x256x64: fewer rows per tile when that needs fewer tile-rounds of the 256
// CUs than the 256-row tiling (see launch_nt)
using NtBig224 = NtCfg<64, 2, 4, 7>;
using NtBig192 = NtCfg<64, 2, 4, 6>;

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
  int group_n;   // tile columns per raster group (their B panels stay L2-resident)
  int k_chunk;   // split-K: K elements per blockIdx.y slice (0 = no split)
  int stagger;   // ping-pong kernel: estimated time of one output tile in 100 MHz ticks (0 = no start-up stagger)
  int esz;       // operand element size in bytes: 2 = bf16, 1 = fp8 e4m3 (ping-pong kernel only)
  const float* alpha;   // fp8 operands: device scalar multiplied into the accumulators (product of the dequantisation scales), or NULL
  void* out2;    // fp8 operands: optional e4m3 image of out1 (EPI_GELU) / of out0 (EPI_DGELU) = the next GEMM's A operand, or NULL
  const float* alpha2;  // second device scalar multiplied into the accumulators (1 / scale of a scaled gradient operand), or NULL
  const float* qscale;  // device scalar the values are multiplied by before they are quantised into out2 (NULL = 1)
  float* qamax;         // device slot that receives max |value| written to out2, before scaling (atomic max; NULL = none)
#ifdef VITSSL_NT_STAMPS
  unsigned long long* stamps;   // diagnostic build only (tools/nt_stamps.py): [2: 100 MHz ticks, shader clocks][wg][2 wave groups][16 rounds][4]
#endif
};

// internal epilogue: fp32 output accumulated with atomics by the split-K slices (out0 is
// zeroed by the launcher; slice 0 adds the bias)
constexpr int EPI_F32_SPLITK = 100;

// XOR applied to the 16-byte chunk index of tile row r (source side for the DMA, and on
// the fragment reads): BK=64 (128-B rows) chunk ^ ((r>>1)&7); BK=32 (64-B rows)
// chunk ^ 3*((r>>3)&1).  Both make every 16-lane ds_read_b128 group conflict-free.
template <int BK_>
__device__ __forceinline__ int nt_swz(int r) {
  return BK_ == 64 ? ((r >> 1) & 7) : 3 * ((r >> 3) & 1);
}

template <int BK_, int ROWS, int WAVES>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, long long row0, int k0,
                                           int K, int wave, int lane) {
  constexpr int ROWB = BK_ * 2;
  constexpr int RPI = 1024 / ROWB;              // tile rows per 1-KiB wave-instruction
  constexpr int LPR = ROWB / 16;                // lanes per row
  constexpr int SLOTS = ROWS / RPI;
  constexpr int PER = (SLOTS + WAVES - 1) / WAVES;   // uneven for 224-row tiles: the last wave issues fewer
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = wave * PER + j;               // wave-uniform instruction slot
    if (SLOTS % WAVES != 0 && i >= SLOTS) break;
    const int r = i * RPI + lane / LPR;         // tile row this lane fetches for
    const int c = lane % LPR;                   // 16-B chunk position in the LDS row
    const int sc = c ^ nt_swz<BK_>(r);          // chunk fetched from global
    const unsigned voff = (unsigned)(((row0 + r) * (long long)K + k0) * 2 + sc * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + i * 1024), 16, voff, 0, 0, 0);
  }
}

// s_waitcnt vmcnt(N) with a compile-time N
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- GELU by table (EPI_GELU of the ping-pong kernel) ---------------------------------------------------------
// The epilogue evaluates gelu / gelu' on the pre-activation ROUNDED TO bf16 (the reference's autocast stores that tensor
// in bf16, feed_forward.py:26-27), i.e. on one of 65536 inputs, and writes two bf16 results.  For 2^-14 <= |u| < 16
// (18 binades x 128 mantissas x 2 signs = 4608 inputs, all but ~1e-4 of the elements) the pair
// (bf16(s gelu(u)), bf16(s gelu'(u))), s = dropout scale, is read from an 18 KiB table in the LDS the staging buffers
// leave free; the workgroup fills it once per launch with gelu_both_scaled, so every entry is bit-identical to what the
// arithmetic path produces for that input.  A 4-element group with any lane outside the range (wave-uniform test) is
// redone with the arithmetic.  Why: the arithmetic is ~60 VALU cycles per element of an epilogue that is VALU-issue
// bound (tools/probes/valu_rates.hip: v_exp / v_rcp 8.2 cycles, packed fp32 4.8 per pair); the lookup is ~25 cycles of
// index arithmetic plus one ds_read_b32 on the otherwise idle LDS pipe (8 cycles per wave-instruction per CU with random
// addresses).
constexpr int GLUT_E_LO = 127 - 14;                       // bf16 exponent field of 2^-14
constexpr int GLUT_BINADES = 18;                          // ... up to [8, 16)
constexpr int GLUT_ENTRIES = GLUT_BINADES * 128 * 2;      // (magnitude index, sign)
constexpr int GLUT_BYTES = GLUT_ENTRIES * 4;
constexpr unsigned GLUT_LO8 = (unsigned)GLUT_E_LO << 10;  // magnitude part of the byte address: (h & 0x7fff) << 3
constexpr unsigned GLUT_HI8 = (unsigned)(GLUT_E_LO + GLUT_BINADES) << 10;

// Table addressing.  Entry (magnitude index m, sign s) of bf16 pattern h sits at byte (2 m + s) * 4 = rot16(h, 1) * 4 (relative to
// table base - GLUT_LO8): both halves of a packed pair are rotated by two packed 16-bit instructions, an SDWA shift per element
// turns a half into its byte address, and the range test of a 4-element group is three packed 16-bit instructions on the
// rotated words (16 instructions per group less than shift / and / min / or per element).  A lane outside the table's range
// reads staging bytes or beyond the workgroup's LDS (reads return 0 there): its group is redone arithmetically.
__device__ __forceinline__ unsigned glut_rot2(unsigned w) {
  unsigned t, r;
  asm("v_pk_lshrrev_b16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(t) : "v"(w));            // sign of each half (the inline constant serves both)
  asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(w), "v"(t));     // (h << 1) + sign, modulo 2^16
  return r;
}
__device__ __forceinline__ void glut_addr2(unsigned r, unsigned& a0, unsigned& a1) {
  asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a0) : "v"(r));
  asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a1) : "v"(r));
}
// true when any of the four rotated patterns lies outside [2 LO, 2 HI): d = r - 2 LO (modulo 2^16) must be <= 2 (HI - LO) - 1
__device__ __forceinline__ bool glut_out_of_range(unsigned r01, unsigned r23) {
  const unsigned lo2 = ((unsigned)GLUT_E_LO << 8) * 0x10001u, span = ((unsigned)(GLUT_BINADES << 8) - 1u) * 0x10001u;
  unsigned d01, d23, dm, e;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d01) : "v"(r01), "s"(lo2));
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d23) : "v"(r23), "s"(lo2));
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(dm) : "v"(d01), "v"(d23));
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(e) : "v"(dm), "s"(span));
  return e != 0;
}
// plain LDS loads (the compiler counts them): checked in the .s that no vmcnt wait is attached to them -- the operand DMA in
// flight during the epilogue writes other LDS bytes, but hipcc cannot always prove that (cdna guide, "three .s-level traps")
// (Addressed as a raw LDS offset: through the `smem` symbol hipcc emits one `v_add_u32 v, 0, v` per lookup that it does not fold.
// The kernel has no static LDS, so its dynamic LDS starts at offset 0; the kernel traps at entry if that ever stops being true.)
template <int OFF>
__device__ __forceinline__ unsigned glut_read(const char* smem, unsigned a) {
  (void)smem;
  return *(const __attribute__((address_space(3))) unsigned*)(unsigned long long)(a + (unsigned)OFF);
}

// Fused epilogue of one 128x64 wave tile (shared by both main-loop variants).
template <int EPI, typename CFG, bool Q8 = false, int GLUT_OFF = -1, bool TLS = false>
__device__ __forceinline__ void nt_epilogue(const NtParams& p, f32x4 (&acc)[4][CFG::MI], const long long m0, const int n0,
                                            const int wm, const int wn, const int lane_in, const char* lds = nullptr, char* xs = nullptr) {
  constexpr int MI = CFG::MI;
  // the lane-only address constants below are re-derived in every epilogue: hoisted out of the tile loop they stay live through
  // the K loop, and the 256-row kernels (128 accumulator registers) then spill K-loop state into scratch
  int lane = lane_in;
  if (TLS) asm volatile("" : "+v"(lane));
  // TLS: xs = this wave's private 2 KiB of LDS (ping-pong kernel: its own B1 staging slots of the buffer that is not being
  // refilled, see the kernel) and N is a multiple of 8; otherwise the stores leave in the accumulator layout
  // GLUT_OFF >= 0: LDS byte offset of the GELU table minus GLUT_LO8 (the ds_read's immediate)
  constexpr bool GLUT = EPI == VITSSL_EPI_GELU && GLUT_OFF >= 0 && NT_ABLATE == 0;
  // fp8 operands: the table's entries are (bf16 s gelu'(u)) | (e4m3(s gelu(u) qs) << 16) -- the e4m3 byte is quantised from
  // the fp32 value when the table is built, exactly as the arithmetic path does per element -- and serve launches that
  // write the g' image and the e4m3 image only (no bf16 `a`, no amax: what engine.EncoderStack asks for); launch-uniform
  const bool glut_on = GLUT && (!Q8 || (p.out1 == nullptr && p.qamax == nullptr));
  // ------------------------------------------------------------------ epilogue
  // All global traffic of the epilogue goes through raw buffer instructions on a window
  // that starts at the tile's first row: rows past M fall outside num_records and columns
  // past N get the out-of-range offset, so loads return 0 and stores are dropped WITHOUT a
  // branch.  That lets every residual / g' load of a 64-column half be issued back to back
  // before the first use (the branchy form waited for each 16-byte load in turn: 32
  // dependent HBM round trips per wave, measured +65 us on the N = K = 768 projection).
  // column sums exist for the epilogues that produce a gradient operand (bias gradients); the entry point rejects them elsewhere
  constexpr bool CS = EPI == VITSSL_EPI_BF16 || EPI == VITSSL_EPI_F32 || EPI == VITSSL_EPI_DGELU;
  float csum[4][4];
  if (CS && p.colsum) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) csum[j][r] = 0.f;
  }
  const int g4 = lane >> 4;                       // lane group = 16-lane row of the wave
  // bf16 images: tile columns (j, j+1) exchange halves between lane rows (g, g^1) with
  // v_permlane16_swap so that every lane moves 16 contiguous bytes (8 columns).
  const bool wide = TLS || (p.N & 7) == 0;
  constexpr unsigned OOB = 0x80000000u;
  const long long rows_left = p.M - m0;
  auto window = [&](const void* base, int elt) {
    const unsigned long long bytes = (unsigned long long)rows_left * (unsigned long long)p.N * (unsigned)elt;
    const unsigned rec = bytes > 0x80000000ull ? 0x80000000u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + m0 * p.N * elt), 0, (int)rec, 0x00020000);
  };
  const unsigned row_l = (unsigned)(wm * CFG::WROWS + (lane & 15));      // + 16 i : row inside the tile
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
  auto pack_pair = [&](const u32x2& w0, const u32x2& w1) -> u32x4 {
    // after the swap: even lane rows hold tile 2jp cols 4g .. 4g+7, odd rows tile 2jp+1 cols 4(g-1) .. 4(g-1)+7
    auto lo = __builtin_amdgcn_permlane16_swap(w0[0], w1[0], false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(w0[1], w1[1], false, false);
    return u32x4{lo[0], hi[0], lo[1], hi[1]};
  };
  // ---- line-shaped stores through the wave's LDS window (NT_LDS_T).  A 16-row tile of an image is 16 rows x 128 bytes
  // (64 bf16 columns, or the 32 fp32 columns of a pair).  It is written in the accumulator layout (lane (m, c) = row m, 8- or
  // 16-byte chunk), with the chunk position XORed by the row so the writes spread over the banks, and read back as lane
  // (rho, kappa) = row rho (+8 for the second read), 16-byte chunk kappa: 8 lanes = one 128-byte line.
  constexpr bool tls = TLS;
  const int tm = lane & 15, trho = lane >> 3, tkap = lane & 7;
  // bf16: this lane's 8-byte chunk of tile j is chunk (4j + c) ^ 2 (m >> 1) of row m: tile 0's offset, tile j's = that ^ 32 j
  const unsigned tw16 = (unsigned)(tm * 128 + ((g4 ^ (2 * (tm >> 1))) << 3));
  const unsigned tr16a = (unsigned)(trho * 128 + ((tkap ^ (trho >> 1)) << 4));      // rows 0-7: 16-byte chunk kappa; rows 8-15: (+1024) ^ 64
  // fp32: 16-byte chunk (4h + c) ^ (m & 7) of the pair's 128-byte row: tile 0's offset, tile 1's = that ^ 64
  const unsigned tw32 = (unsigned)(tm * 128 + ((g4 ^ (tm & 7)) << 4));
  const unsigned tr32a = (unsigned)(trho * 128 + ((tkap ^ trho) << 4));              // rows 8-15: + 1024 ((rho + 8) & 7 == rho & 7)
  const unsigned trow = (unsigned)(wm * CFG::WROWS + trho);                // + 16 i (+ 8): row inside the tile
  // lane part of the offsets (row trho of tile 0; out-of-range columns get the sentinel, which stays out of range when the
  // wave-uniform row term below is added: that term is < 2^31); the row term is scalar arithmetic
  const int tn16 = n0 + wn * 64 + 8 * tkap;
  const unsigned tb16 = tn16 < p.N ? (trow * un + (unsigned)tn16) * 2u : OOB;
  auto off_line16 = [&](int i, int half) -> unsigned { return tb16 + (unsigned)(16 * i + 8 * half) * (un * 2u); };   // bf16 image: 8 columns per lane
  auto off_line32 = [&](int i, int jp, int half) -> unsigned {              // fp32 image: 4 columns per lane
    const int n = n0 + wn * 64 + 32 * jp + 4 * tkap;
    const unsigned b = n < p.N ? (trow * un + (unsigned)n) * 4u : OOB;
    return b + (unsigned)(16 * i + 8 * half) * (un * 4u);
  };
  // one row tile of a bf16 image: w[jp][h]
  constexpr int BF16_AUX = NT_STORE_AUX != 0 ? NT_STORE_AUX : NT_BF16_AUX;
  auto store_bf16_row = [&](__amdgpu_buffer_rsrc_t rs, int i, const u32x2 (&w)[2][2], auto aux_c) {
    constexpr int AUXV = decltype(aux_c)::value;
    if (NT_ABLATE == 1) {
      asm volatile("" ::"v"(w[0][0]), "v"(w[0][1]), "v"(w[1][0]), "v"(w[1][1]));
      return;
    }
    if (tls) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int h = 0; h < 2; ++h) *(u32x2*)(xs + (tw16 ^ (unsigned)(32 * (2 * jp + h)))) = w[jp][h];
      const u32x4 s1 = *(const u32x4*)(xs + tr16a), s2 = *(const u32x4*)(xs + ((tr16a + 1024u) ^ 64u));
      __builtin_amdgcn_raw_buffer_store_b128(s1, rs, off_line16(i, 0), 0, AUXV);
      __builtin_amdgcn_raw_buffer_store_b128(s2, rs, off_line16(i, 1), 0, AUXV);
    } else if (wide) {
      const u32x4 a = pack_pair(w[0][0], w[0][1]), b = pack_pair(w[1][0], w[1][1]);
      __builtin_amdgcn_raw_buffer_store_b128(a, rs, off_bf16_wide(i, 0), 0, AUXV);
      __builtin_amdgcn_raw_buffer_store_b128(b, rs, off_bf16_wide(i, 1), 0, AUXV);
    } else {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int na = n0 + wn * 64 + (2 * jp) * 16 + 4 * g4;
        __builtin_amdgcn_raw_buffer_store_b64(w[jp][0], rs, off_elem(i, na, 2u), 0, AUXV);
        __builtin_amdgcn_raw_buffer_store_b64(w[jp][1], rs, off_elem(i, na + 16, 2u), 0, AUXV);
      }
    }
  };
  // one pair (32 columns) of a row tile of an fp32 image
  constexpr int F32_AUX = NT_STORE_AUX != 0 ? NT_STORE_AUX : (EPI == VITSSL_EPI_RESID ? NT_RESID_AUX : 0);
  auto store_f32_pair = [&](__amdgpu_buffer_rsrc_t rs, int i, int jp, const f32x4& v0, const f32x4& v1, const int (&nnp)[2]) {
    if (NT_ABLATE == 1) {
      asm volatile("" ::"v"(v0), "v"(v1));
      return;
    }
    if (tls) {
      *(f32x4*)(xs + tw32) = v0;
      *(f32x4*)(xs + (tw32 ^ 64u)) = v1;
      const u32x4 s1 = *(const u32x4*)(xs + tr32a), s2 = *(const u32x4*)(xs + tr32a + 1024);
      __builtin_amdgcn_raw_buffer_store_b128(s1, rs, off_line32(i, jp, 0), 0, F32_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(s2, rs, off_line32(i, jp, 1), 0, F32_AUX);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rs, off_elem(i, nnp[0], 4u), 0, F32_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), rs, off_elem(i, nnp[1], 4u), 0, F32_AUX);
    }
  };
  __amdgpu_buffer_rsrc_t rsOut0, rsOut1, rsOut2, rsAux;
  constexpr bool Q8EPI = Q8 && (EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_DGELU);
  const bool q8 = Q8EPI && p.out2 != nullptr;
  if (q8) rsOut2 = window(p.out2, 1);
  const float qs = (Q8EPI && p.qscale) ? *p.qscale : 1.0f;
  float qmax = 0.f;                               // running max |value| of this lane's share of the e4m3 image
  if constexpr (EPI == VITSSL_EPI_BF16 || EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_DGELU) rsOut0 = window(p.out0, 2);
  if constexpr (EPI == VITSSL_EPI_F32 || EPI == VITSSL_EPI_RESID) rsOut0 = window(p.out0, 4);
  if constexpr (EPI == VITSSL_EPI_GELU) rsOut1 = window(p.out1, 2);
  if constexpr (EPI == VITSSL_EPI_RESID) rsAux = window(p.aux, 4);
  if constexpr (EPI == VITSSL_EPI_DGELU) rsAux = window(p.aux, 2);

  // Column bookkeeping of this wave's 64 columns: pair jp covers tiles (2jp, 2jp+1), half h
  // of a pair is one 16-column tile; a lane owns 4 consecutive columns of each.
  int nn[2][2];
  bool okn[2][2];
  f32x4 bias4[2][2];
  // dropout stream (common.h): the state word of group g = row * N/4 + col/4 is g * C0 + k0 = (row term) + (column term);
  // the row term advances by a launch constant per 16-row tile, the column terms are four lane constants
  constexpr bool DROPS = EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_RESID;
  unsigned a0col[2][2];
  const unsigned a0rowstep = 16u * (unsigned)(p.N >> 2) * DROP_C0;
  unsigned a0row = 0;
  if (DROPS && p.drop_on)
    a0row = drop_a0(p.dk, (unsigned)(m0 + wm * CFG::WROWS + (lane & 15)) * (unsigned)(p.N >> 2));
  // GELU: the dropout scale is folded into the two constants of gelu_both_scaled
  const float gelu_hs = (DROPS && p.drop_on) ? 0.5f * p.dk.scale : 0.5f;
  const float gelu_cs = (DROPS && p.drop_on) ? 0.3989422804014327f * p.dk.scale : 0.3989422804014327f;
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      nn[jp][h] = n0 + wn * 64 + (2 * jp + h) * 16 + 4 * g4;
      a0col[jp][h] = (unsigned)(nn[jp][h] >> 2) * DROP_C0;
      okn[jp][h] = nn[jp][h] < p.N;
      bias4[jp][h] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias && (EPI != EPI_F32_SPLITK || blockIdx.y == 0) && okn[jp][h]) bias4[jp][h] = *(const f32x4*)(p.bias + nn[jp][h]);
    }

  // Rows are walked in groups; inside a group the loop order is row -> pair, so the two
  // 64-byte halves of every 128-byte output line are stored by consecutive instructions
  // (pair-major order left them half an epilogue apart and the second bf16 image of the GELU
  // epilogue was written at ~3 TB/s).  The residual / g' operands of a whole group are
  // loaded before its first use.
  // Rows per group: the group's loads are one memory round trip.  The 224-row tiles (MI = 7: every N = 768 launch of ViT-B,
  // both residual epilogues among them) take ONE row per group.  Round 3 measured uneven groups (4 + 3) against that,
  // interleaved: N = K = 768 residual 104 -> 109 us, K = 3072 residual 251 -> 247 us, dGELU unchanged -- the phase is
  // bound by bytes, not by round trips (as round 1 found for MI = 8); NT_RG_RESID / NT_RG_DGELU keep the experiment.
  // Operand prefetch (NT_EPI_PREFETCH, line-shaped form only): the residual / g' lines of group g+1 are requested BEFORE group g
  // is computed and stored.  The vector-memory counter retires in issue order, so in the plain order (loads of g+1 behind the
  // stores of g) the wait for a group's operands also waited for the previous group's stores to be acknowledged by the L2 --
  // one load round trip plus one store round trip per group, 7 times per tile for the 224-row residual epilogue.
  constexpr bool PREF = NT_EPI_PREFETCH != 0 && TLS && (EPI == VITSSL_EPI_RESID || EPI == VITSSL_EPI_DGELU);
  // (with the prefetch two groups of operands are alive at once: 1 row per group for the residual lines, 2 for g')
  constexpr int RG = EPI == VITSSL_EPI_RESID ? (PREF ? NT_RG_RESID_PF : (MI % 2 == 0 ? 2 : NT_RG_RESID))
                     : EPI == VITSSL_EPI_DGELU ? (PREF ? NT_RG_DGELU_PF : (MI % 4 == 0 ? 4 : NT_RG_DGELU))
                                               : 4;
  constexpr int NB = PREF ? 2 : 1;
  f32x4 res[NB][RG][2][2];   // RESID: residual stream (line layout until used)
  u32x4 raw[NB][RG][2];      // DGELU, 16-byte form: g' as loaded
  auto load_group = [&](const int ig, const int b) {
    const int cnt = MI - ig < RG ? MI - ig : RG;
    if constexpr (EPI == VITSSL_EPI_RESID) {
#pragma unroll
      for (int ii = 0; ii < RG; ++ii)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          if (ii >= cnt) continue;
          if (NT_ABLATE == 2) {
            res[b][ii][jp][0] = res[b][ii][jp][1] = f32x4{0.f, 0.f, 0.f, 0.f};
          } else if (tls) {
            // whole lines (8 rows x 128 bytes per instruction); turned into the accumulator layout through the LDS window at use
            res[b][ii][jp][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsAux, off_line32(ig + ii, jp, 0), 0, 0));
            res[b][ii][jp][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsAux, off_line32(ig + ii, jp, 1), 0, 0));
          } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
              res[b][ii][jp][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsAux, off_elem(ig + ii, nn[jp][h], 4u), 0, 0));
          }
        }
    }
    if constexpr (EPI == VITSSL_EPI_DGELU) {
      if (wide) {
#pragma unroll
        for (int ii = 0; ii < RG; ++ii) {
          if (ii >= cnt) continue;
          if (tls) {
            raw[b][ii][0] = __builtin_amdgcn_raw_buffer_load_b128(rsAux, off_line16(ig + ii, 0), 0, 0);    // rows 0-7 of the row tile, whole lines
            raw[b][ii][1] = __builtin_amdgcn_raw_buffer_load_b128(rsAux, off_line16(ig + ii, 1), 0, 0);    // rows 8-15
          } else {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) raw[b][ii][jp] = __builtin_amdgcn_raw_buffer_load_b128(rsAux, off_bf16_wide(ig + ii, jp), 0, 0);
          }
        }
      }
    }
  };
  if constexpr (PREF) load_group(0, 0);
#pragma unroll
  for (int ig = 0; ig < MI; ig += RG) {
    const int cnt = MI - ig < RG ? MI - ig : RG;      // (a constant once the loop is unrolled)
    const int gb = PREF ? (ig / RG) & 1 : 0;
    if constexpr (PREF) {
      if (ig + RG < MI) load_group(ig + RG, gb ^ 1);
    } else {
      load_group(ig, 0);
    }
    u32x2 gpre[RG][2][2];    // DGELU: g' in accumulator layout
    if constexpr (EPI == VITSSL_EPI_DGELU) {
      if (wide) {
#pragma unroll
        for (int ii = 0; ii < RG; ++ii)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            if (ii >= cnt) continue;
            if (tls) {
              if (jp == 0) {                        // (both pairs at once: the window holds the whole row tile)
                *(u32x4*)(xs + tr16a) = raw[gb][ii][0];
                *(u32x4*)(xs + ((tr16a + 1024u) ^ 64u)) = raw[gb][ii][1];
#pragma unroll
                for (int j = 0; j < 4; ++j) gpre[ii][j >> 1][j & 1] = *(const u32x2*)(xs + (tw16 ^ (unsigned)(32 * j)));
              }
              continue;
            }
            // inverse of the store shuffle (the swap is an involution)
            auto sa = __builtin_amdgcn_permlane16_swap(raw[gb][ii][jp][0], raw[gb][ii][jp][2], false, false);
            auto sb = __builtin_amdgcn_permlane16_swap(raw[gb][ii][jp][1], raw[gb][ii][jp][3], false, false);
            gpre[ii][jp][0] = u32x2{sa[0], sb[0]};
            gpre[ii][jp][1] = u32x2{sa[1], sb[1]};
          }
      } else {
#pragma unroll
        for (int ii = 0; ii < RG; ++ii)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              if (ii < cnt) gpre[ii][jp][h] = __builtin_amdgcn_raw_buffer_load_b64(rsAux, off_elem(ig + ii, nn[jp][h], 2u), 0, 0);
      }
    }

#pragma unroll
    for (int ii = 0; ii < RG; ++ii) {
      if (ii >= cnt) continue;
      const int i = ig + ii;
      const long long m = m0 + wm * CFG::WROWS + i * 16 + (lane & 15);
      const bool okm = m < p.M;
      u32x2 out_a[2][2], out_b[2][2];   // bf16 images of this row, both pairs: stored together below
      unsigned out_q[2][2];             // e4m3 image of out_b (fp8 operand path)
      // GELU by table: the row's 16 lookups are issued first and fly under the dropout-stream arithmetic below
      unsigned gl[16];                  // (bf16 s gelu(u)) | (bf16 s gelu'(u)) << 16 per element: [jp][h][r]
      bool gslow[2][2];                 // wave-uniform: some lane of this group is outside the table's range
      if (GLUT && glut_on) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 vv = acc[2 * jp + h][i] + bias4[jp][h];
            const unsigned w0 = pack_bf2(vv[0], vv[1]), w1 = pack_bf2(vv[2], vv[3]);
            const unsigned r01 = glut_rot2(w0), r23 = glut_rot2(w1);
            gslow[jp][h] = __builtin_amdgcn_ballot_w64(glut_out_of_range(r01, r23)) != 0;
            unsigned ad[4];
            glut_addr2(r01, ad[0], ad[1]);
            glut_addr2(r23, ad[2], ad[3]);
            const int q = 8 * jp + 4 * h;
#pragma unroll
            for (int r = 0; r < 4; ++r) gl[q + r] = glut_read<GLUT_OFF>(lds, ad[r]);
          }
      }
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        f32x4 v[2] = {acc[2 * jp][i] + bias4[jp][0], acc[2 * jp + 1][i] + bias4[jp][1]};

        if constexpr (EPI == VITSSL_EPI_BF16) {
          out_a[jp][0] = u32x2{pack_bf2(v[0][0], v[0][1]), pack_bf2(v[0][2], v[0][3])};
          out_a[jp][1] = u32x2{pack_bf2(v[1][0], v[1][1]), pack_bf2(v[1][2], v[1][3])};
        } else if constexpr (EPI == VITSSL_EPI_GELU) {
          // u = bf16(acc + bias) (never stored); out1 = a = keep*scale*gelu(u) feeds the next
          // GEMM; out0 = g' = keep*scale*gelu'(u) is what the backward dGELU epilogue needs.
          // keep masks of each group's two bf16 pairs (0xffff per kept element), ANDed onto the packed outputs
          unsigned km[2][2] = {{0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu}};
          u32x2 dw[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};
          const bool dropping = p.drop_on && NT_ABLATE != 2;
          if (dropping) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              dw[h] = drop_words_a0(p.dk, a0row + (unsigned)i * a0rowstep + a0col[jp][h]);
              km[h][0] = drop_keep_pair(p.dk, dw[h][0]);
              km[h][1] = drop_keep_pair(p.dk, dw[h][1]);
            }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (GLUT && glut_on && !gslow[jp][h]) {
              const int q = 8 * jp + 4 * h;
              if constexpr (!Q8) {
                out_b[jp][h] = u32x2{__builtin_amdgcn_perm(gl[q + 1], gl[q], 0x05040100u) & km[h][0],
                                     __builtin_amdgcn_perm(gl[q + 3], gl[q + 2], 0x05040100u) & km[h][1]};
                out_a[jp][h] = u32x2{__builtin_amdgcn_perm(gl[q + 1], gl[q], 0x07060302u) & km[h][0],
                                     __builtin_amdgcn_perm(gl[q + 3], gl[q + 2], 0x07060302u) & km[h][1]};
              } else {
                out_a[jp][h] = u32x2{__builtin_amdgcn_perm(gl[q + 1], gl[q], 0x05040100u) & km[h][0],
                                     __builtin_amdgcn_perm(gl[q + 3], gl[q + 2], 0x05040100u) & km[h][1]};
                // the four e4m3 bytes (byte 2 of every entry) and the byte mask of the kept elements
                const unsigned b01 = __builtin_amdgcn_perm(gl[q + 1], gl[q], 0x0c0c0602u), b23 = __builtin_amdgcn_perm(gl[q + 3], gl[q + 2], 0x06020c0cu);
                out_q[jp][h] = (b01 | b23) & __builtin_amdgcn_perm(km[h][1], km[h][0], 0x06040200u);
              }
              continue;
            }
            float y[4], d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (NT_ABLATE == 2) {
                y[r] = v[h][r];
                d[r] = v[h][r] * 0.5f;
                continue;
              }
              gelu_both_scaled(round_bf(v[h][r]), gelu_hs, gelu_cs, y[r], d[r]);
            }
            out_b[jp][h] = u32x2{pack_bf2(y[0], y[1]) & km[h][0], pack_bf2(y[2], y[3]) & km[h][1]};
            out_a[jp][h] = u32x2{pack_bf2(d[0], d[1]) & km[h][0], pack_bf2(d[2], d[3]) & km[h][1]};
            if constexpr (Q8) {
              if (dropping) {
                bool keep[4];
                drop_keep4(p.dk, dw[h], keep);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = keep[r] ? y[r] : 0.f;
              }
              out_q[jp][h] = pack_fp8x4(y[0] * qs, y[1] * qs, y[2] * qs, y[3] * qs);
              if (okm && okn[jp][h]) qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
            }
          }
        } else if constexpr (EPI == VITSSL_EPI_DGELU) {
          // du = acc * g'  (g' already carries the dropout mask and its scale)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const u32x2 gpv = gpre[ii][jp][h];
            v[h][0] *= bf_lo(gpv[0]);
            v[h][1] *= bf_hi(gpv[0]);
            v[h][2] *= bf_lo(gpv[1]);
            v[h][3] *= bf_hi(gpv[1]);
            out_a[jp][h] = u32x2{pack_bf2(v[h][0], v[h][1]), pack_bf2(v[h][2], v[h][3])};
            if constexpr (Q8) {
              out_q[jp][h] = pack_fp8x4(v[h][0] * qs, v[h][1] * qs, v[h][2] * qs, v[h][3] * qs);
              if (okm && okn[jp][h])
                qmax = fmaxf(qmax, fmaxf(fmaxf(fabsf(v[h][0]), fabsf(v[h][1])), fmaxf(fabsf(v[h][2]), fabsf(v[h][3]))));
            }
          }
        } else if constexpr (EPI == EPI_F32_SPLITK) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (!(okm && okn[jp][h])) continue;
            float* o = (float*)p.out0 + m * p.N + nn[jp][h];
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(o + r, v[h][r]);
          }
        } else if constexpr (EPI == VITSSL_EPI_F32) {
          store_f32_pair(rsOut0, i, jp, v[0], v[1], nn[jp]);
        } else if constexpr (EPI == VITSSL_EPI_RESID) {
          if (tls && NT_ABLATE != 2) {            // the pair's residual lines -> accumulator layout (inverse of store_f32_pair's path)
            *(f32x4*)(xs + tr32a) = res[gb][ii][jp][0];
            *(f32x4*)(xs + tr32a + 1024) = res[gb][ii][jp][1];
            res[gb][ii][jp][0] = *(const f32x4*)(xs + tw32);
            res[gb][ii][jp][1] = *(const f32x4*)(xs + (tw32 ^ 64u));
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (p.drop_on && NT_ABLATE != 2) {
              bool keep[4];
              drop_keep4(p.dk, drop_words_a0(p.dk, a0row + (unsigned)i * a0rowstep + a0col[jp][h]), keep);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[h][r] = fmaf(keep[r] ? v[h][r] : 0.f, p.dk.scale, res[gb][ii][jp][h][r]);
            } else {
              v[h] += res[gb][ii][jp][h];
            }
          }
          store_f32_pair(rsOut0, i, jp, v[0], v[1], nn[jp]);
        } else {   // VITSSL_EPI_EMBED (one launch per step: plain addressing)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (!(okm && okn[jp][h])) continue;
            const long long img = m / p.embed.tokens;
            const int rin = (int)(m - img * p.embed.tokens);
            if (p.embed.mask && p.embed.mask[m]) v[h] = *(const f32x4*)(p.embed.mask_token + nn[jp][h]);
            v[h] += *(const f32x4*)(p.embed.pos + (long long)(p.embed.tok_offset + rin) * p.N + nn[jp][h]);
            const long long orow = img * p.embed.out_tokens + p.embed.tok_offset + rin;
            *(f32x4*)((float*)p.out0 + orow * p.N + nn[jp][h]) = v[h];
          }
        }
        if (CS && p.colsum) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (okm && okn[jp][h]) {
#pragma unroll
              for (int r = 0; r < 4; ++r) csum[2 * jp + h][r] += v[h][r];
            }
        }
      }
      // the row's 128 bytes of every bf16 image leave in back-to-back instructions
      if constexpr (EPI == VITSSL_EPI_BF16 || EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_DGELU) {
        // fp8 path: the bf16 image of a dGELU output is optional once its e4m3 image is written (launch-uniform)
        if (!(Q8 && EPI == VITSSL_EPI_DGELU) || p.out0)
          store_bf16_row(rsOut0, i, out_a, IC<(EPI == VITSSL_EPI_GELU && NT_STORE_AUX == 0) ? NT_GPRIME_AUX : BF16_AUX>{});
      }
      if constexpr (EPI == VITSSL_EPI_GELU) {
        if (!Q8 || p.out1) store_bf16_row(rsOut1, i, out_b, IC<BF16_AUX>{});   // same for the GELU output
      }
      if constexpr (Q8EPI) {
        if (q8) {
          if ((p.N & 15) == 0) {
            // The row's 64 bytes: dword t*4 + g lives in lane row g as out_q of tile t.  A 4x4 transpose over the
            // four lane rows (permlane32_swap exchanges the wave halves, permlane16_swap odd / even rows) leaves
            // lane row g with dwords 4g .. 4g+3 = 16 contiguous bytes: one store instruction per row tile.
            auto s02 = __builtin_amdgcn_permlane32_swap(out_q[0][0], out_q[1][0], false, false);
            auto s13 = __builtin_amdgcn_permlane32_swap(out_q[0][1], out_q[1][1], false, false);
            auto e = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
            auto f = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
            if (tls) {
              // 16 rows x 64 bytes through the LDS window: written as lane (m, c) -> row m, chunk c ^ ((m >> 1) & 3), read as
              // lane (rho = lane >> 2, kappa = lane & 3): the four lanes of a quad cover one row's 64 contiguous bytes
              *(u32x4*)(xs + tm * 64 + ((g4 ^ ((tm >> 1) & 3)) << 4)) = u32x4{e[0], e[1], f[0], f[1]};
              const int rho4 = lane >> 2, kap4 = lane & 3;
              const u32x4 s8 = *(const u32x4*)(xs + rho4 * 64 + ((kap4 ^ ((rho4 >> 1) & 3)) << 4));
              const int n8 = n0 + wn * 64 + 16 * kap4;
              const unsigned o8 = n8 < p.N ? ((unsigned)(wm * CFG::WROWS + 16 * i + rho4) * un + (unsigned)n8) : OOB;
              __builtin_amdgcn_raw_buffer_store_b128(s8, rsOut2, o8, 0, BF16_AUX);
            } else {
              const int n = n0 + wn * 64 + 16 * g4;
              __builtin_amdgcn_raw_buffer_store_b128(u32x4{e[0], e[1], f[0], f[1]}, rsOut2, off_elem(i, n, 1u), 0, BF16_AUX);
            }
          } else {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
              for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_raw_buffer_store_b32(out_q[jp][h], rsOut2, off_elem(i, nn[jp][h], 1u), 0, BF16_AUX);
          }
        }
      }
    }
  }

  if (CS && p.colsum) {
    f32x4 red[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = csum[j][r];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        red[j][r] = s;
      }
    }
    if (tls) {
      // One atomic instruction per wave and tile, 64 lanes on 256 contiguous bytes (the float-atomic path's full-rate shape),
      // instead of sixteen with four active lanes each: the 64 column sums of the wave pass through its LDS window (column
      // 16 j + 4 g + r of lane group g -> float slot of that column).  Round 4: the sixteen-instruction form cost ~10 us of a
      // 266 us dGELU launch (the atomics are paced per instruction, ~50 ns per CU).
      if ((lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(xs + (16 * j + 4 * g4) * 4) = red[j];
      }
      asm volatile("" ::: "memory");                  // (the wave's LDS operations execute in order; this keeps hipcc from reordering them)
      const float v = *(const volatile float*)(xs + lane * 4);
      const int n = n0 + wn * 64 + lane;
      if (n < p.N) atomicAdd(p.colsum + n, v);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4) + r;
          if ((lane & 15) == 0 && n < p.N) atomicAdd(p.colsum + n, red[j][r]);
        }
    }
  }
  if constexpr (Q8EPI) {
    if (p.qamax) {          // one atomic per wave and tile at most; skipped when the slot already holds a larger value
      qmax = wave_max(qmax);
      unsigned* slot = (unsigned*)p.qamax;
      if (lane == 0 && __float_as_uint(qmax) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(qmax));
    }
  }

}

template <int EPI, typename CFG>
__global__ __launch_bounds__(CFG::THREADS, CFG::MIN_WAVES_PER_SIMD) void gemm_nt_kernel(NtParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = CFG::BM, BN = CFG::BN, BK = CFG::BK;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / CFG::WN, wn = wave % CFG::WN;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int G = gridDim.x;
  const int bid = blockIdx.x;

  // Tile of this workgroup in round r, or false.  Within a round the `cnt` live workgroups
  // get an XCD-aware bijective remap: blocks b, b+8, ... share an XCD (round-robin
  // dispatch), so each XCD takes a contiguous run of tiles and neighbouring tiles (same A
  // row-panel, same weight panels) hit the same L2.  Tiles are ordered column-group-major:
  // all tile rows of a group of `group_n` tile columns, then the next group, so a group's
  // weight panels are re-used from L2 by every tile row.  Speed only, never correctness.
  auto tile_of = [&](int r, long long& m0, int& n0) -> bool {
    const int base = r * G;
    const int cnt = min(G, ntiles - base);
    if (bid >= cnt) return false;
    const int xcd = bid & 7, q = cnt >> 3, rr = cnt & 7;
    const int wgid = base + (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int full = p.tiles_m * p.group_n;
    const int cg = wgid / full;
    const int rem = wgid - cg * full;
    const int gw = min(p.group_n, p.tiles_n - cg * p.group_n);
    const int tile_m = rem / gw;
    m0 = (long long)tile_m * BM;
    n0 = (cg * p.group_n + (rem - tile_m * gw)) * BN;
    return true;
  };

  const unsigned long long a_bytes = (unsigned long long)p.M * p.K * 2ull;
  const unsigned long long b_bytes = (unsigned long long)p.N * p.K * 2ull;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)b_bytes, 0x00020000);

  const int kb = p.k_chunk ? (int)blockIdx.y * p.k_chunk : 0;                       // this slice's K range
  const int nk = (p.k_chunk ? min(p.K - kb, p.k_chunk) : p.K) / BK;
  const int swz = nt_swz<BK>(lane & 15);           // rows are 16*x + (lane&15)
  const int frag_row = lane & 15;
  const int kq = lane >> 4;

  constexpr int MI = CFG::MI;
  f32x4 acc[4][MI];

  auto compute_stage = [&](const char* bufA, const char* bufB) {
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int coff = ((kk * 4 + kq) ^ swz) << 4;
      bf16x8 fb[4], fa[MI];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fb[j] = *(const bf16x8*)(bufB + (wn * 64 + j * 16 + frag_row) * CFG::ROWB + coff);
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = *(const bf16x8*)(bufA + (wm * CFG::WROWS + i * 16 + frag_row) * CFG::ROWB + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    }
  };

  // 32-MFMA cluster of k-step kk (a compile-time constant: a runtime index pushes acc[] to scratch)
  auto compute_half = [&](const char* bufA, const char* bufB, auto kk_c) {
    constexpr int kk = decltype(kk_c)::value;
    const int coff = ((kk * 4 + kq) ^ swz) << 4;
    bf16x8 fb[4], fa[MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      fb[j] = *(const bf16x8*)(bufB + (wn * 64 + j * 16 + frag_row) * CFG::ROWB + coff);
#pragma unroll
    for (int i = 0; i < MI; ++i)
      fa[i] = *(const bf16x8*)(bufA + (wm * CFG::WROWS + i * 16 + frag_row) * CFG::ROWB + coff);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[j][i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  long long m0 = 0, m0n = 0;
  int n0 = 0, n0n = 0;
  if (!tile_of(0, m0, n0)) return;                 // workgroup-uniform
  int par = 0;                                     // LDS buffer of the stage consumed next
  stage_tile<BK, BM, CFG::WAVES>(rsA, smem, m0, kb, p.K, wave, lane);
  stage_tile<BK, BN, CFG::WAVES>(rsB, smem + CFG::A_BYTES, n0, kb, p.K, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int round = 0;; ++round) {
    const bool has_next = tile_of(round + 1, m0n, n0n);
    // stage 0 of this tile has landed for this wave; the barrier makes every wave's share
    // visible and fences the previous tile's last LDS reads from this tile's first DMA
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nk; ++t) {
      char* bufA = smem + par * CFG::BUF_BYTES;
      char* nA = smem + (par ^ 1) * CFG::BUF_BYTES;
      const bool more = t + 1 < nk;
      // what to prefetch into the other buffer: this tile's next K-step, or the first
      // K-step of the next tile of this workgroup (lands behind the epilogue)
      const bool fetch = more || has_next;
      const long long sm = more ? m0 : m0n;
      const int sn = more ? n0 : n0n;
      const int sk = kb + (more ? (t + 1) * BK : 0);
      if constexpr (BK == 64) {
        // spread the DMA over the step: A before the first MFMA cluster, B between the
        // two (a single 64-KiB burst right after the barrier queues in the TA)
        if (fetch) stage_tile<BK, BM, CFG::WAVES>(rsA, nA, sm, sk, p.K, wave, lane);
        compute_half(bufA, bufA + CFG::A_BYTES, IC<0>{});
        if (fetch) stage_tile<BK, BN, CFG::WAVES>(rsB, nA + CFG::A_BYTES, sn, sk, p.K, wave, lane);
        compute_half(bufA, bufA + CFG::A_BYTES, IC<1>{});
      } else {
        if (fetch) {
          stage_tile<BK, BM, CFG::WAVES>(rsA, nA, sm, sk, p.K, wave, lane);
          stage_tile<BK, BN, CFG::WAVES>(rsB, nA + CFG::A_BYTES, sn, sk, p.K, wave, lane);
        }
        compute_stage(bufA, bufA + CFG::A_BYTES);
      }
      par ^= 1;
      if (more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }

    nt_epilogue<EPI, CFG>(p, acc, m0, n0, wm, wn, lane);

    if (!has_next) break;
    // The next tile's first stage was issued before every store above; vector-memory
    // operations retire in issue order, so "at most <the stores of the last half> still in
    // flight" implies that DMA has landed, without waiting for the stores themselves.
    constexpr int TAIL = (EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_F32) ? 16
                         : ((EPI == VITSSL_EPI_EMBED || EPI == EPI_F32_SPLITK) ? 0 : 8);   // stores of the last row group
    wait_vmcnt<TAIL>();
    m0 = m0n;
    n0 = n0n;
  }
}

// ====================================================================================
// Ping-pong main loop (BK = 64 tiles, 8 waves): the K-step is cut into 4 phases of one
// 64x32 accumulator quadrant each (16 MFMAs); every phase is {LOAD: fragment ds_reads + the
// LDS-DMA of one half-tile "unit" ; s_barrier ; COMPUTE: 16 MFMAs ; s_barrier}.  Waves 4-7
// (the second wave of every SIMD: tile rows 128..255) run ONE barrier behind waves 0-3, so on
// each SIMD one wave is in its MFMA cluster while its partner reads LDS / issues DMA -- the
// matrix pipe never waits for a fragment read and the two waves never contend for it.
//
// Staging never drains: operands are DMA'd in units of 128 tile rows x 64 k (16 KiB = 2
// wave-instructions per wave): A0/A1 = m-half 0/1 of BOTH wave rows, B0/B1 = n-half 0/1 of all
// four wave columns.  The unit issued in phase p of K-tile t is
//     p0: B1(t+1)   p1: A1(t+1)   p2: B0(t+2)   p3: A0(t+2)
// i.e. a region is refilled two phases after its last fragment read (B0/A0 are read in p0, B1 in
// p1, A1 in p2) and every unit is issued 5-6 phases before its first read.  Waits are counted:
// p0, p1 and p3 end their LOAD part with s_waitcnt vmcnt(8) = "everything but the 4 newest units
// has landed", which is exactly what the NEXT phase reads; p2 needs no wait.  Rules this obeys
// (MI355X guide, "Read a staged buffer one phase AFTER the wait that retires it"): the wait that
// covers a unit sits before the first barrier of the phase preceding its first read (one extra
// barrier because the two wave groups are staggered), and a region is re-issued no earlier
// than two phases after its last read.
//
// The K-tile stream is continuous across OUTPUT tiles (persistent workgroups): while the last
// K-tiles of one output tile are multiplied, the first units of the workgroup's next output tile
// are already in flight, so there is no prologue bubble after the first tile; the epilogue runs
// with ~80 KiB of the next tile staged.  The vector-memory operations of the epilogue sit in the
// same in-order counter, so the first K-tile after an epilogue waits with vmcnt(8 + S), S = a
// lower bound of the epilogue's operations per wave (a smaller count only waits longer).
// Past the last tile the stream issues out-of-range DMA (zero fill, no memory traffic) so every
// count stays uniform.
template <int EPI, int MI, bool F8>
constexpr int nt_epi_vmem_ops() {
  // LOWER bound of the buffer loads + stores per wave in nt_epilogue (a smaller count only waits longer; a larger one would let
  // the first K-tile after the epilogue read a staging unit that has not landed).  bf16 operands: GELU = g' + a images (2 + 2 per
  // row tile), DGELU = g' loads + the bf16 image (2 + 2).  fp8 operands: the bf16 `a` image (GELU) / the bf16 dGELU image are
  // optional at launch time, so only g' stores (GELU: 2) or g' loads (DGELU: 2) plus the one e4m3 store are certain: 3 per row tile
  // (round-3 advisor finding: 4 was counted).
  if (NT_ABLATE != 0) return 0;
  return EPI == VITSSL_EPI_BF16 ? 2 * MI
         : (EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_DGELU) ? (F8 ? 3 * MI : 4 * MI)
         : EPI == VITSSL_EPI_F32 ? 4 * MI
         : EPI == VITSSL_EPI_RESID ? 8 * MI
         : 0;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_exact() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// two 16-byte fragments of one tile row -> the 32-byte operand of the K = 128 fp8 MFMA
__device__ __forceinline__ i32x8 join_frags(const bf16x8& lo, const bf16x8& hi) {
  const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
  return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
}

// (target builtins with immediate operands are kept out of the kernel's lambdas: on the host pass a
// lambda body is checked eagerly and the kernel would silently lose its stub)
// cache policy of the operand DMA (buffer aux bits: 1 = sc0, 2 = nt, 16 = sc1); developer experiments only
#ifndef NT_DMA_AUX
#define NT_DMA_AUX 0
#endif
__device__ __forceinline__ void dma16_to_lds(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_wave_base), 16, voffset, 0, 0, NT_DMA_AUX);
}

template <int EPI, typename CFG, bool F8 = false>
__global__ __launch_bounds__(CFG::THREADS, 2) void gemm_nt_pp_kernel(NtParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(CFG::BK == 64 && CFG::WM == 2 && CFG::WN == 4, "ping-pong loop is written for 8 waves, BK = 64");
  constexpr int BM = CFG::BM, BN = CFG::BN, MI = CFG::MI;
  constexpr int MH1 = MI - 4;                          // 16-row tiles in the second m-half (first has 4)
  constexpr int BUF = CFG::BUF_BYTES;
  constexpr int DUMMY = 2 * BUF;                       // 1 KiB sink for the slots a short A1 unit does not need
  constexpr unsigned OOBV = 0x80000000u;
  // EPI_GELU (bf16 operands): the 18 KiB GELU table sits behind the sink (see "GELU by table" above nt_epilogue)
  constexpr bool USE_GLUT = EPI == VITSSL_EPI_GELU && NT_ABLATE == 0 && NT_GELU_LUT;
  constexpr int GLUT_BASE = 2 * BUF + 1024;
  constexpr int GLUT_IMM = USE_GLUT ? GLUT_BASE - (int)GLUT_LO8 : -1;
  static_assert(!USE_GLUT || (GLUT_IMM >= 0 && GLUT_IMM < 65536), "the table's offset must fit the ds_read immediate");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int G = gridDim.x;
  const int bid = blockIdx.x;

  auto tile_of = [&](int r, long long& m0, int& n0) -> bool {      // same raster as gemm_nt_kernel
    const int base = r * G;
    const int cnt = min(G, ntiles - base);
    if (bid >= cnt) return false;
    const int xcd = bid & 7, q = cnt >> 3, rr = cnt & 7;
    const int wgid = base + (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int full = p.tiles_m * p.group_n;
    const int cg = wgid / full;
    const int rem = wgid - cg * full;
    const int gw = min(p.group_n, p.tiles_n - cg * p.group_n);
    const int tile_m = rem / gw;
    m0 = (long long)tile_m * BM;
    n0 = (cg * p.group_n + (rem - tile_m * gw)) * BN;
    return true;
  };

  long long m0 = 0;
  int n0 = 0;
  if (!tile_of(0, m0, n0)) return;                     // workgroup-uniform, before any barrier

  // Start-up stagger.  Every workgroup runs the same program on equal tiles, so left alone the
  // whole chip alternates between "all CUs in the K loop" (HBM nearly idle) and "all CUs in the
  // epilogue" (a 32-128 MB burst at the HBM rate with the matrix pipes idle): measured 9-23 us of
  // epilogue per tile for the two-image / residual epilogues against 2-4 us of issue time.  With
  // the static tile assignment the workgroups b >= ntiles % G own one tile fewer than the others:
  // they can start up to one tile time late for free.  Spreading their start over that window
  // puts their epilogues beside other CUs' K loops.
  if (p.stagger > 0) {
    const int rem = ntiles % G;
    if (rem != 0 && bid >= rem && wave == 0) {
      const unsigned long long delay = (unsigned long long)p.stagger * (unsigned)(bid - rem) / (unsigned)(G - rem);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < delay) __builtin_amdgcn_s_sleep(8);
    }   // the other waves are held by the prologue's barrier
  }

  // A K-tile is 128 BYTES of every operand row: 64 bf16 or 128 e4m3 elements.  Staging, LDS layout and
  // fragment reads are byte-identical for the two operand types; the fp8 form hands the two 16-byte
  // fragments of a row to ONE v_mfma_f32_16x16x128_f8f6f4 (its k order inside the tile is a permutation
  // applied to both operands alike, which a contraction does not see).
  constexpr unsigned ESZ = F8 ? 1u : 2u;
  const unsigned long long a_bytes = (unsigned long long)p.M * p.K * ESZ;
  const unsigned long long b_bytes = (unsigned long long)p.N * p.K * ESZ;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)b_bytes, 0x00020000);
  const int nk = F8 ? p.K / 128 : p.K / 64;
  const unsigned rowb = (unsigned)p.K * ESZ;           // bytes per operand row

  // ---- staging slots: every wave issues exactly 2 DMA instructions per unit (8 rows x 128 B each)
  // unit A_h: rows g*WROWS + 64 h + 8 q (+ lane/8); unit B_h: rows 64 c + 32 h + 8 q
  unsigned voffA[2][2], voffB[2][2];                   // tile-relative byte offsets of this lane's 16-byte chunk
  int ldsA[2][2], ldsB[2][2];                          // wave-uniform LDS byte offsets inside a buffer
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int s = 2 * wave + e;
      {
        const int per_g = h == 0 ? 8 : 2 * MH1;        // slots per wave row in this unit
        const bool ok = s < 2 * per_g;
        const int g = s / per_g, q = s - g * per_g;
        const int r = g * CFG::WROWS + 64 * h + 8 * q;
        const int rl = r + (lane >> 3);
        const int sc = (lane & 7) ^ ((rl >> 1) & 7);
        voffA[h][e] = ok ? (unsigned)rl * rowb + (unsigned)sc * 16u : OOBV;
        ldsA[h][e] = ok ? r * 128 : DUMMY;
      }
      {
        const int c = s >> 2, q = s & 3;
        const int r = 64 * c + 32 * h + 8 * q;
        const int rl = r + (lane >> 3);
        const int sc = (lane & 7) ^ ((rl >> 1) & 7);
        voffB[h][e] = (unsigned)rl * rowb + (unsigned)sc * 16u;
        ldsB[h][e] = CFG::A_BYTES + r * 128;
      }
    }

  // ---- staging stream: K-tiles in the order they will be multiplied, across output tiles
  struct Cur {
    int round, kt;
    unsigned a, b;          // byte offset of (tile row 0, k0) in A / B; OOBV once the stream has ended
  };
  auto cur_at = [&](int round) {
    Cur c;
    c.round = round;
    c.kt = 0;
    long long mm;
    int nn;
    if (tile_of(round, mm, nn)) {
      c.a = (unsigned)((unsigned long long)mm * rowb);
      c.b = (unsigned)((unsigned long long)nn * rowb);
    } else {
      c.a = c.b = OOBV;
    }
    return c;
  };
  auto cur_next = [&](const Cur& c) {
    if (c.a == OOBV) return c;
    if (c.kt + 1 < nk) {
      Cur n = c;
      n.kt += 1;
      n.a += 128u;
      n.b += 128u;
      return n;
    }
    return cur_at(c.round + 1);
  };
  // DMA of one unit: 2 wave-instructions.  bufsel = LDS buffer (0/1) of the unit's K-tile.
  auto stage_a = [&](const Cur& c, int bufsel, auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if (NT_LOOP_ABLATE == 1 && bufsel >= 0) return;
    // The K-tile's position goes into the DESCRIPTOR (scalar arithmetic: window = operand bytes from (tile row 0, k0) on; an ended
    // stream gets an empty window -> zero fill, no traffic), not into the lanes' offsets: the LOAD parts of this loop share their SIMD
    // with the partner wave's MFMA cluster and vector-ALU instructions there cost issue slots (csrc/gemm_tn.hip).
    __amdgpu_buffer_rsrc_t rs = NT_DESC_WINDOW ? __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + c.a), 0,
                                                                                    c.a == OOBV ? 0 : (int)(a_bytes - c.a), 0x00020000)
                                               : rsA;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool live = ldsA[h][e] != DUMMY;
      // (an ended stream has c.a = OOBV: out of range for every live slot -> zero fill)
      dma16_to_lds(rs, smem + (live ? bufsel * BUF : 0) + ldsA[h][e], NT_DESC_WINDOW ? voffA[h][e] : voffA[h][e] + c.a);
    }
  };
  auto stage_b = [&](const Cur& c, int bufsel, auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if (NT_LOOP_ABLATE == 1 && bufsel >= 0) return;
    __amdgpu_buffer_rsrc_t rs = NT_DESC_WINDOW ? __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + c.b), 0,
                                                                                    c.b == OOBV ? 0 : (int)(b_bytes - c.b), 0x00020000)
                                               : rsB;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      dma16_to_lds(rs, smem + bufsel * BUF + ldsB[h][e], NT_DESC_WINDOW ? voffB[h][e] : voffB[h][e] + c.b);
  };

  // ---- fragment addressing (as in gemm_nt_kernel)
  const int frag_row = lane & 15;
  const int kq = lane >> 4;
  const int swz = nt_swz<64>(frag_row);
  const int offA0 = (wm * CFG::WROWS + frag_row) * 128 + (((0 + kq) ^ swz) << 4);
  const int offA1 = (wm * CFG::WROWS + frag_row) * 128 + (((4 + kq) ^ swz) << 4);
  const int offB0 = CFG::A_BYTES + (wn * 64 + frag_row) * 128 + (((0 + kq) ^ swz) << 4);
  const int offB1 = CFG::A_BYTES + (wn * 64 + frag_row) * 128 + (((4 + kq) ^ swz) << 4);

  f32x4 acc[4][MI];
  bf16x8 fa[2][4];          // A fragments of the current m-half: [kk][tile]
  bf16x8 fb[2][2][2];       // B fragments: [n-half][kk][tile]

  auto read_a = [&](const char* buf, auto mh_c) {
    constexpr int mh = decltype(mh_c)::value;
    constexpr int cnt = mh == 0 ? 4 : MH1;
    if (NT_LOOP_ABLATE == 2 && buf != nullptr) return;
#pragma unroll
    for (int ii = 0; ii < cnt; ++ii) {
      fa[0][ii] = *(const bf16x8*)(buf + offA0 + (4 * mh + ii) * 2048);
      fa[1][ii] = *(const bf16x8*)(buf + offA1 + (4 * mh + ii) * 2048);
    }
  };
  auto read_b = [&](const char* buf, auto nh_c) {
    constexpr int nh = decltype(nh_c)::value;
    if (NT_LOOP_ABLATE == 2 && buf != nullptr) return;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      fb[nh][0][jj] = *(const bf16x8*)(buf + offB0 + (2 * nh + jj) * 2048);
      fb[nh][1][jj] = *(const bf16x8*)(buf + offB1 + (2 * nh + jj) * 2048);
    }
  };
  auto mma_quad = [&](auto mh_c, auto nh_c) {
    constexpr int mh = decltype(mh_c)::value, nh = decltype(nh_c)::value;
    constexpr int cnt = mh == 0 ? 4 : MH1;
    if (NT_LOOP_ABLATE == 3) {
      asm volatile("" ::"v"(fa[0][0]), "v"(fa[1][0]), "v"(fa[0][1]), "v"(fa[1][1]), "v"(fa[0][2]), "v"(fa[1][2]), "v"(fa[0][3]), "v"(fa[1][3]),
                   "v"(fb[nh][0][0]), "v"(fb[nh][1][0]), "v"(fb[nh][0][1]), "v"(fb[nh][1][1]));
      return;
    }
    if (NT_SETPRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr (F8) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const i32x8 bq = join_frags(fb[nh][0][jj], fb[nh][1][jj]);
#pragma unroll
        for (int ii = 0; ii < cnt; ++ii)
          acc[2 * nh + jj][4 * mh + ii] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              bq, join_frags(fa[0][ii], fa[1][ii]), acc[2 * nh + jj][4 * mh + ii], 0, 0, 0, 0, 0, 0);   // e4m3 x e4m3, scales 0 = unscaled form
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int ii = 0; ii < cnt; ++ii)
            acc[2 * nh + jj][4 * mh + ii] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nh][kk][jj], fa[kk][ii], acc[2 * nh + jj][4 * mh + ii], 0, 0, 0);
    }
    if (NT_SETPRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto section = [&]() {                               // end of a LOAD or COMPUTE part
    // the raw s_barrier carries no fence: the empty asm statements keep the optimiser from moving
    // fragment loads across it, sched_barrier keeps the machine scheduler from moving MFMAs
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };

  // One K-tile = 4 phases.  NW = counted wait of p0 / p1 / p3 (8, or 8 + S right after an epilogue).
  auto ktile = [&](int buf, const Cur& c1, const Cur& c2, bool last_of_round, auto nw_c) {
    constexpr int NW = decltype(nw_c)::value;
    const char* cur = smem + buf * BUF;
    // p0: quadrant (m0, n0)
    read_b(cur, IC<0>{});
    read_a(cur, IC<0>{});
    stage_b(c1, buf ^ 1, IC<1>{});
    wait_vmcnt_exact<NW>();
    section();
    mma_quad(IC<0>{}, IC<0>{});
    section();
    // p1: quadrant (m0, n1)
    read_b(cur, IC<1>{});
    stage_a(c1, buf ^ 1, IC<1>{});
    wait_vmcnt_exact<NW>();
    section();
    mma_quad(IC<0>{}, IC<1>{});
    section();
    // p2: quadrant (m1, n1)
    read_a(cur, IC<1>{});
    stage_b(c2, buf, IC<0>{});
    section();
    mma_quad(IC<1>{}, IC<1>{});
    section();
    // p3: quadrant (m1, n0)
    stage_a(c2, buf, IC<0>{});
    wait_vmcnt_exact<NW>();
    section();
    mma_quad(IC<1>{}, IC<0>{});
    // waves 4-7 leave the stagger at the end of an output tile (they re-enter it with the
    // barrier at the top of the next one): both wave groups then run their epilogues together
    if (!(last_of_round && wm == 1)) section();
  };

  // ---- prologue: K-tiles 0 and (B0, A0 of) 1 of the stream
  Cur c0 = cur_at(0);
  Cur c1 = cur_next(c0);
  stage_b(c0, 0, IC<0>{});
  stage_a(c0, 0, IC<0>{});
  stage_b(c0, 0, IC<1>{});
  stage_a(c0, 0, IC<1>{});
  stage_b(c1, 1, IC<0>{});
  stage_a(c1, 1, IC<0>{});
  Cur c2 = cur_next(c1);
  if constexpr (USE_GLUT) {
    if ((unsigned)(unsigned long long)LDS_PTR(smem) != 0u) __builtin_trap();      // glut_read addresses the table by raw LDS offsets
    // fill the table while the first K-tiles are in flight: entry (magnitude index, sign) = the two bf16 results for that
    // bf16 input, from the SAME function the arithmetic path evaluates.  Every wave passes the K loop's barriers before
    // the first epilogue reads it.
    const float hs = p.drop_on ? 0.5f * p.dk.scale : 0.5f;
    const float cs = p.drop_on ? 0.3989422804014327f * p.dk.scale : 0.3989422804014327f;
    unsigned* tab = (unsigned*)(smem + GLUT_BASE);
    for (int e = threadIdx.x; e < GLUT_ENTRIES; e += CFG::THREADS) {
      const unsigned h = (unsigned)((e >> 1) + (GLUT_E_LO << 7)) | ((unsigned)(e & 1) << 15);
      float y, dy;
      gelu_both_scaled(bf2f((bf16_t)h), hs, cs, y, dy);
      if constexpr (F8) {
        const float qs = p.qscale ? *p.qscale : 1.0f;
        tab[e] = (unsigned)f2bf(dy) | ((pack_fp8x4(y * qs, 0.f, 0.f, 0.f) & 0xffu) << 16);
      } else {
        tab[e] = (unsigned)f2bf(y) | ((unsigned)f2bf(dy) << 16);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  wait_vmcnt_exact<8>();                               // B0, A0 of K-tile 0 have landed
  section();

  constexpr int S = nt_epi_vmem_ops<EPI, MI, F8>();
#ifndef NT_PROBE_NW
#define NT_PROBE_NW 8      // timing probe only: any other value reads staging units that may not have landed
#endif
  constexpr int NW_POST = (NT_PROBE_NW + S) > 63 ? 63 : (NT_PROBE_NW + S);
  int buf = 0;
#ifdef VITSSL_NT_STAMPS
  auto stamp = [&](int round, int which) {
    if (p.stamps && (wave & 3) == 0 && lane == 0 && round < 16)
      p.stamps[(((size_t)bid * 2 + wm) * 16 + round) * 4 + which] = __builtin_amdgcn_s_memrealtime();
    // second half of the buffer: the same points in shader clocks (in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz)
    if (p.stamps && (wave & 3) == 0 && lane == 0 && round < 16)
      p.stamps[(size_t)256 * 2 * 16 * 4 + (((size_t)bid * 2 + wm) * 16 + round) * 4 + which] = __builtin_amdgcn_s_memtime();
  };
#else
  auto stamp = [&](int, int) {};
#endif
  for (int round = 0;; ++round) {
    stamp(round, 0);
    if (wm == 1) section();                            // waves 4-7 run one barrier behind waves 0-3
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nk; ++t) {
      const bool last = t + 1 == nk;
      if (t == 0 && round > 0) ktile(buf, c1, c2, last, IC<NW_POST>{});
      else ktile(buf, c1, c2, last, IC<NT_PROBE_NW>{});
      c1 = c2;
      c2 = cur_next(c2);
      buf ^= 1;
    }
    stamp(round, 1);
    if constexpr (F8) {
      if (p.alpha || p.alpha2) {
        const float al = (p.alpha ? *p.alpha : 1.0f) * (p.alpha2 ? *p.alpha2 : 1.0f);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[j][i] *= al;
      }
    }
    // the wave's LDS window for line-shaped stores: its own two B1 staging slots (2 KiB, contiguous) of the buffer whose B1 / A1
    // units are not in flight -- K-tile (last) read them two barriers ago, and this wave itself re-issues them in p0 of the
    // next K-tile, after its epilogue in program order; no other wave ever writes there
    nt_epilogue<EPI, CFG, F8, GLUT_IMM, NT_LDS_T != 0>(p, acc, m0, n0, wm, wn, lane, smem, smem + (buf ^ 1) * BUF + ldsB[1][0]);
    stamp(round, 2);
#ifdef VITSSL_NT_STAMPS
    if (p.stamps) {                                    // diagnostic: when have this wave's stores been acknowledged?
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(round, 3);
    }
#endif
    if (!tile_of(round + 1, m0, n0)) break;
  }
  // the stream's trailing (out-of-range, zero-filling) DMA must have retired before the LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int cu_count() { return vitssl_persistent_cus(); }
void nt_note_grid(int grid);   // remembers the workgroup count of the last ping-pong launch (vitssl_debug_last_nt_grid)

// 1 (default): 8-wave BK = 64 tiles run the ping-pong kernel; 0: the two-phase loop (VITSSL_NT_PP, developer knob)
int nt_pp_enabled() {
  static VsEnvInt knob;
  return knob.get("VITSSL_NT_PP", 1);
}

template <int EPI, typename CFG, bool F8 = false>
int launch_pp(NtParams p, hipStream_t s) {
  constexpr int LDS = 2 * CFG::BUF_BYTES + 1024 + ((EPI == VITSSL_EPI_GELU && NT_ABLATE == 0 && NT_GELU_LUT) ? GLUT_BYTES : 0);
  static VsOnce attr_done{false};
  if (!attr_done.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, CFG, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_nt: cannot raise dynamic LDS to %d: %s", LDS, hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done.store(true, std::memory_order_relaxed);
  }
  const long long ntiles = (long long)p.tiles_m * p.tiles_n;
  const long long slots = cu_count();
  const long long grid = ntiles < slots ? ntiles : slots;
  // tile-time estimate for the start-up stagger (us): K loop ~1.45 us per 256-row K-tile + an uncontended epilogue
  // VITSSL_NT_STAGGER (developer knob): scale of the window, 0 = off
  static VsEnvMilli stagger_knob;
  {
    // Round 2 measured this time-neutral: in the accumulator layout a CU's stores were bound inside the CU (55 GB/s) whether or not
    // the other CUs stored at the same moment.  With line-shaped epilogue accesses (NT_LDS_T) a CU alone stores 180+ GB/s but only
    // ~55 when all 256 burst together, so spreading the epilogues now pays: interleaved A/B, M = 50176: dGELU 281 -> 262 us
    // (N = 3072, K = 768), residual 241 -> 231 (K = 3072), plain stores -1.5..-3 %, never slower; whole step 34.29 -> 34.05 ms.
  }
  const float stagger_scale = stagger_knob.get("VITSSL_NT_STAGGER", NT_STAGGER_DEFAULT);
  // which epilogues stagger (bit = VITSSL_EPI_* value; VITSSL_NT_STAGGER_EPIS overrides): see NT_STAGGER_EPIS_DEFAULT
  static VsEnvInt stagger_mask_knob;
  const int stagger_mask = stagger_mask_knob.get("VITSSL_NT_STAGGER_EPIS", NT_STAGGER_EPIS_DEFAULT);
  const float epi_us = EPI == VITSSL_EPI_BF16 ? 2.f : EPI == VITSSL_EPI_GELU ? 7.f : EPI == VITSSL_EPI_DGELU ? 5.f
                       : EPI == VITSSL_EPI_RESID ? 8.f : 4.f;
  const float tile_us = (float)(p.K * p.esz / 128) * 1.45f * (float)CFG::MI / 8.f + epi_us;
  p.stagger = ((stagger_mask >> EPI) & 1) ? (int)(stagger_scale * tile_us * 100.f) : 0;
  nt_note_grid((int)grid);
  hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, CFG, F8>), dim3((unsigned)grid), dim3(CFG::THREADS), LDS, s, p);
  VS_CHECK_LAUNCH("gemm_nt_pp");
  return VITSSL_OK;
}

template <int EPI, typename CFG>
int launch_cfg(NtParams p, hipStream_t s) {
  static VsOnce attr_done{false};
  if (!attr_done.load(std::memory_order_relaxed) && CFG::LDS_BYTES > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, CFG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       CFG::LDS_BYTES);
    if (e != hipSuccess) {
      vitssl_set_error("gemm_nt: cannot raise dynamic LDS to %d: %s", CFG::LDS_BYTES, hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done.store(true, std::memory_order_relaxed);
  }
  p.tiles_m = (int)ceil_div64(p.M, CFG::BM);
  p.tiles_n = (int)ceil_div64(p.N, CFG::BN);
  const int want = (int)((2 * 1024 * 1024) / ((long long)CFG::BN * p.K * p.esz));   // panels of a group <= ~2 MiB of L2
  static VsEnvInt group_env;                           // VITSSL_NT_GROUPN: force the raster group width (developer knob)
  const int group_knob = group_env.get("VITSSL_NT_GROUPN", 0);
  if (group_knob > 0) p.group_n = group_knob < p.tiles_n ? group_knob : p.tiles_n;
  else if (p.tiles_n <= 4) p.group_n = p.tiles_n;
  else if (want <= 2) p.group_n = 2;
  else if (p.tiles_n % 6 == 0 && want >= 6) p.group_n = 6;
  else if (p.tiles_n % 4 == 0) p.group_n = 4;
  else if (p.tiles_n % 3 == 0) p.group_n = 3;
  else p.group_n = want < 4 ? want : 4;
  if (p.esz == 1) {
    if ((p.N & 7) != 0) {
      vitssl_set_error("gemm_fp8_nt: N=%d must be a multiple of 8", p.N);
      return VITSSL_ERR_ARG;
    }
    // fp8 operands exist for the ping-pong loop and the epilogues of the transformer-block forward only
    if constexpr (CFG::WAVES == 8 && CFG::BK == 64 &&
                  (EPI == VITSSL_EPI_BF16 || EPI == VITSSL_EPI_F32 || EPI == VITSSL_EPI_GELU || EPI == VITSSL_EPI_RESID ||
                   EPI == VITSSL_EPI_DGELU)) {
      return launch_pp<EPI, CFG, true>(p, s);
    } else {
      vitssl_set_error("gemm_fp8_nt: epilogue %d / tile configuration not built for fp8 operands", EPI);
      return VITSSL_ERR_ARG;
    }
  }
  if constexpr (CFG::WAVES == 8 && CFG::BK == 64 && EPI != EPI_F32_SPLITK) {
    if (nt_pp_enabled() && p.k_chunk == 0 && (p.N & 7) == 0) return launch_pp<EPI, CFG>(p, s);   // (its line-shaped stores move 8 columns per lane)
  }
  // Persistent workgroups.  Measured on MI355X (bench.py, same box, alternating runs): with
  // K = 768 / 3072 (ViT-B) the NT family takes 23.12 ms per step either way and the whole step
  // is 0.2-0.5 ms slower persistent; with K = 384 (ViT-S: 6 K-steps per tile, the fixed
  // per-tile cost dominates) the step drops 21.75 -> 20.83 ms.  Default: persistent for
  // K <= 512 only; VITSSL_NT_PERSIST=0/1 forces it (developer knob).
  static VsEnvInt persist_env;
  const int knob = persist_env.get("VITSSL_NT_PERSIST", -1);
  const bool persist = knob >= 0 ? knob != 0 : p.K <= 512;
  const long long ntiles = (long long)p.tiles_m * p.tiles_n;
  long long grid = ntiles;
  if (persist) {
    const long long slots = (long long)cu_count() * CFG::WG_PER_CU;
    if (grid > slots) grid = slots;
  }
  const unsigned slices = p.k_chunk ? (unsigned)ceil_div64(p.K, p.k_chunk) : 1u;
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, CFG>), dim3((unsigned)grid, slices), dim3(CFG::THREADS), CFG::LDS_BYTES, s, p);
  VS_CHECK_LAUNCH("gemm_nt");
  return VITSSL_OK;
}

// 0 = auto, 1 = always BIG, 2 = always SMALL, 3 = always 192x256 (VITSSL_NT_TILE, developer knob)
int nt_tile_override() {
  static VsEnvInt knob;
  return knob.get("VITSSL_NT_TILE", 0);
}

// Split-K for fp32 outputs whose grid cannot fill the chip but whose contraction is long
// (DINO head: dX[640,256] = dY[640,65536] . W[65536,256] is 6 SMALL tiles x 2048 K-steps:
// 483 us on 6 CUs).  The K range is cut into slices that accumulate with fp32 atomics into a
// zeroed output.
int launch_splitk_f32(NtParams p, hipStream_t s, bool* done) {
  *done = false;
  const long long tiles = ceil_div64(p.M, NtSmall::BM) * ceil_div64(p.N, NtSmall::BN);
  if (tiles >= 64 || p.K < 4096) return VITSSL_OK;
  // about half a workgroup per CU: more slices mean more colliding fp32 atomics on the small
  // output (M = 512: 128 slices 253 us, 64: 147 us, 32: 103 us, 16: 114 us, 8: 178 us)
  long long slices = (cu_count() / 2 + tiles - 1) / tiles;
  const long long max_slices = p.K / 512;            // at least 16 K-steps of 32 per slice
  if (slices > max_slices) slices = max_slices;
  if (slices < 2) return VITSSL_OK;
  long long chunk = ceil_div64(ceil_div64(p.K, slices), 64) * 64;
  p.k_chunk = (int)chunk;
  hipError_t e = hipMemsetAsync(p.out0, 0, (size_t)p.M * p.N * sizeof(float), s);
  if (e != hipSuccess) {
    vitssl_set_error("gemm_nt: split-K memset failed: %s", hipGetErrorString(e));
    return VITSSL_ERR_LAUNCH;
  }
  *done = true;
  return launch_cfg<EPI_F32_SPLITK, NtSmall>(p, s);
}

template <int EPI>
int launch_nt(const NtParams& p, hipStream_t s) {
  if constexpr (EPI == VITSSL_EPI_F32) {
    if (!p.colsum && p.esz == 2) {
      bool done = false;
      const int rc = launch_splitk_f32(p, s, &done);
      if (rc != VITSSL_OK || done) return rc;
    }
  }
  const int mode = p.esz == 1 ? 0 : nt_tile_override();
  if (mode == 1) return launch_cfg<EPI, NtBig>(p, s);
  if (mode == 2) return launch_cfg<EPI, NtSmall>(p, s);
  if (mode == 3) return launch_cfg<EPI, NtBig192>(p, s);
  if (mode == 4) return launch_cfg<EPI, NtBig224>(p, s);
  // Measured on MI355X (tools/bench_gemm.py, round 1): SMALL loses 10-25 % on every ViT-B
  // shape, heavy epilogues included; it only pays for grids too small to fill the chip.
  const long long big_tiles = ceil_div64(p.M, 256) * ceil_div64(p.N, 256);
  if (big_tiles < 64 && p.esz == 2) return launch_cfg<EPI, NtSmall>(p, s);
  // (N = 384, ViT-S: three exact 128-wide SMALL columns instead of two 256-wide ones with the
  // second half empty were measured too: 21.1 vs 20.9 ms per step, not worth it.)
  //
  // Row counts whose 256-row tiling is "one round and a bit" (DINO global crops: M = 25216,
  // N = 768 -> 297 tiles = 2 rounds for 1.16 rounds of work) are re-tiled with 192-row tiles
  // when that needs fewer tile-rounds: cost model = rounds x (1 for 256x256, 0.78 for 192x256,
  // measured ratio of the two main loops).
  const long long slots = cu_count();
  const long long tn = ceil_div64(p.N, 256);
  const double c256 = (double)ceil_div64(ceil_div64(p.M, 256) * tn, slots);
  const double c224 = 0.90 * (double)ceil_div64(ceil_div64(p.M, 224) * tn, slots);
  const double c192 = 0.78 * (double)ceil_div64(ceil_div64(p.M, 192) * tn, slots);
  double best = c256;
  if (c192 < 0.9 * c256 && c192 <= c224) best = c192;
  else if (c224 < 0.95 * c256) best = c224;
  // (Two launches over disjoint row ranges -- whole rounds of one tile height, then one round of another -- were built and
  // measured in round 2: 0.3 % of a ViT-B step, because the ping-pong loop is bound by the bytes it stages, (rows + 256) per
  // K-tile, so the smaller tiles give the tail back; removed in round 4, DESIGN.md section 12b keeps the numbers.)
  if (best == c192 && best != c256) return launch_cfg<EPI, NtBig192>(p, s);
  if (best == c224 && best != c256) return launch_cfg<EPI, NtBig224>(p, s);
  return launch_cfg<EPI, NtBig>(p, s);
}

}  // namespace

// ---- CUs the persistent grids may occupy --------------------------------------------------------------
// Explicit library state (round 2 read an environment variable ONCE, at the first GEMM launch, so a reducer built after any
// forward was silently ignored): vitssl_set_reserved_cus() takes effect at the next launch of every persistent grid (forward /
// input-gradient / weight-gradient GEMMs, LayerNorm backward).  VITSSL_RESERVE_CUS only provides the initial value.
static std::atomic<int> g_reserved_cus{-1};     // -1 = not initialised
static std::atomic<int> g_device_cus{0};
static std::atomic<int> g_last_nt_grid{0};

static int device_cus() {
  int c = g_device_cus.load(std::memory_order_relaxed);
  if (c > 0) return c;
  int dev = 0, cus = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  if (cus <= 0) cus = 256;
  g_device_cus.store(cus, std::memory_order_relaxed);
  return cus;
}

static int clamp_reserve(int n) {
  const int cus = device_cus();
  if (n < 0) n = 0;
  if (n > cus - 8) n = cus - 8;
  return n;
}

extern "C" int vitssl_get_reserved_cus(void) {
  int r = g_reserved_cus.load(std::memory_order_relaxed);
  if (r < 0) {
    const char* e = getenv("VITSSL_RESERVE_CUS");
    r = clamp_reserve(e ? atoi(e) : 0);
    g_reserved_cus.store(r, std::memory_order_relaxed);
  }
  return r;
}

extern "C" int vitssl_set_reserved_cus(int n) {
  if (n < 0) {
    vitssl_set_error("set_reserved_cus: negative count %d", n);
    return VITSSL_ERR_ARG;
  }
  g_reserved_cus.store(clamp_reserve(n), std::memory_order_relaxed);
  return VITSSL_OK;
}

int vitssl_persistent_cus(void) { return device_cus() - vitssl_get_reserved_cus(); }

extern "C" int vitssl_debug_last_nt_grid(void) { return g_last_nt_grid.load(std::memory_order_relaxed); }
namespace {
void nt_note_grid(int grid) { g_last_nt_grid.store(grid, std::memory_order_relaxed); }
}

#ifdef VITSSL_NT_STAMPS
static unsigned long long* g_nt_stamps = nullptr;
extern "C" void vitssl_debug_nt_stamps(void* buf) { g_nt_stamps = (unsigned long long*)buf; }
#endif

static int gemm_nt_entry(const vitssl_gemm_t* g, int esz, const vitssl_fp8_gemm_t* q, void* stream) {
  const char* who = esz == 1 ? "gemm_fp8_nt" : "gemm_nt";
  VS_CHECK_ARG(g && g->A && g->B && (g->out0 || (esz == 1 && g->epilogue == VITSSL_EPI_DGELU && q && q->out_fp8)), "%s: null operand", who);
  VS_CHECK_ARG(g->M > 0 && g->N > 0 && g->K > 0, "%s: empty problem M=%lld N=%d K=%d", who, (long long)g->M, g->N, g->K);
  VS_CHECK_ARG(g->K % (128 / esz) == 0, "%s: K=%d must be a multiple of %d", who, g->K, 128 / esz);
  VS_CHECK_ARG(g->N % 4 == 0, "%s: N=%d must be a multiple of 4", who, g->N);
  VS_CHECK_ARG(g->N <= (1 << 20), "%s: N=%d exceeds 2^20 (epilogue windows use 32-bit byte offsets)", who, g->N);
  VS_CHECK_ARG((unsigned long long)g->M * g->K * (unsigned)esz < (1ull << 31) && (unsigned long long)g->N * g->K * (unsigned)esz < (1ull << 31),
               "%s: operand larger than 2 GiB (M=%lld N=%d K=%d)", who, (long long)g->M, g->N, g->K);
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
  p.tiles_m = p.tiles_n = p.group_n = 0;   // set per tile configuration in launch_cfg
  p.k_chunk = 0;
  p.stagger = 0;
  p.esz = esz;
  p.alpha = q ? q->alpha : nullptr;
  p.alpha2 = q ? q->alpha2 : nullptr;
  p.out2 = q ? q->out_fp8 : nullptr;
  p.qscale = q ? q->out_scale : nullptr;
  p.qamax = q ? q->out_amax : nullptr;
#ifdef VITSSL_NT_STAMPS
  p.stamps = g_nt_stamps;
#endif
  hipStream_t s = (hipStream_t)stream;
  VS_CHECK_ARG(!g->colsum || g->epilogue == VITSSL_EPI_BF16 || g->epilogue == VITSSL_EPI_F32 || g->epilogue == VITSSL_EPI_DGELU,
               "%s: column sums are built for the BF16 / F32 / DGELU epilogues only", who);
  VS_CHECK_ARG(!p.drop_on || (unsigned long long)g->M * (unsigned long long)g->N < (1ull << 34),
               "%s: dropout over %lld x %d elements: the stream's group counter is 32 bits (M * N < 2^34)", who, (long long)g->M, g->N);
  switch (g->epilogue) {
    case VITSSL_EPI_BF16: return launch_nt<VITSSL_EPI_BF16>(p, s);
    case VITSSL_EPI_F32: return launch_nt<VITSSL_EPI_F32>(p, s);
    case VITSSL_EPI_GELU:
      VS_CHECK_ARG(g->out1 || p.out2, "%s: EPI_GELU needs out1 (or, with fp8 operands, out_fp8)", who);
      return launch_nt<VITSSL_EPI_GELU>(p, s);
    case VITSSL_EPI_RESID:
      VS_CHECK_ARG(g->aux, "%s: EPI_RESID needs aux (residual)", who);
      return launch_nt<VITSSL_EPI_RESID>(p, s);
    case VITSSL_EPI_DGELU:
      VS_CHECK_ARG(g->aux, "%s: EPI_DGELU needs aux (pre-activation)", who);
      return launch_nt<VITSSL_EPI_DGELU>(p, s);
    case VITSSL_EPI_EMBED:
      VS_CHECK_ARG(g->embed.pos && g->embed.tokens > 0 && g->embed.out_tokens >= g->embed.tokens + g->embed.tok_offset,
                   "%s: EPI_EMBED needs pos/tokens", who);
      VS_CHECK_ARG(!g->embed.mask || g->embed.mask_token, "%s: EPI_EMBED mask without mask_token", who);
      return launch_nt<VITSSL_EPI_EMBED>(p, s);
    default:
      vitssl_set_error("%s: unknown epilogue %d", who, g->epilogue);
      return VITSSL_ERR_ARG;
  }
}

extern "C" int vitssl_gemm_bf16_nt(const vitssl_gemm_t* g, void* stream) { return gemm_nt_entry(g, 2, nullptr, stream); }

extern "C" int vitssl_gemm_fp8_nt(const vitssl_gemm_t* g, const vitssl_fp8_gemm_t* q, void* stream) {
  VS_CHECK_ARG(g && (g->epilogue == VITSSL_EPI_BF16 || g->epilogue == VITSSL_EPI_F32 || g->epilogue == VITSSL_EPI_GELU ||
                     g->epilogue == VITSSL_EPI_RESID || g->epilogue == VITSSL_EPI_DGELU),
               "gemm_fp8_nt: fp8 operands are built for the BF16 / F32 / GELU / RESID / DGELU epilogues");
  VS_CHECK_ARG(!(q && q->out_fp8) || g->epilogue == VITSSL_EPI_GELU || g->epilogue == VITSSL_EPI_DGELU,
               "gemm_fp8_nt: out_fp8 belongs to EPI_GELU / EPI_DGELU");
  return gemm_nt_entry(g, 1, q, stream);
}
