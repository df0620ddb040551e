// Issue-rate probe for the VALU instructions the GEMM epilogues are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
// Every kernel runs ITER x 32 independent copies of one instruction per wave; grids of 256 CUs x {4, 8} waves
// (one / two waves per SIMD).  Reported: shader cycles per wave-instruction per SIMD (s_memtime ticks around the loop,
// median over waves), i.e. 2 = full rate on a SIMD-32 for a wave64 instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define ITER 512

#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
                 X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)

typedef __attribute__((ext_vector_type(2))) float f32x2;

#define KERNEL(NAME, DECL, BODY, SINK)                                                              \
  __global__ __launch_bounds__(512) void NAME(unsigned long long* out, float seedf, unsigned seedu) { \
    DECL                                                                                            \
    unsigned long long t0, t1;                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                      \
    for (int it = 0; it < ITER; ++it) {                                                             \
      BODY                                                                                          \
    }                                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                      \
    SINK                                                                                            \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;  \
  }

// ---- float
#define F_DECL float a[32]; _Pragma("unroll") for (int i = 0; i < 32; ++i) a[i] = seedf + i;
#define F_SINK float s = 0; _Pragma("unroll") for (int i = 0; i < 32; ++i) s += a[i]; if (s == 12345.678f) out[0] = 1;
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seedf));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seedf));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define CVTPK(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seedf));
KERNEL(k_fma, F_DECL, REP32(FMA), F_SINK)
KERNEL(k_mul, F_DECL, REP32(MUL), F_SINK)
KERNEL(k_exp, F_DECL, REP32(EXP), F_SINK)
KERNEL(k_rcp, F_DECL, REP32(RCP), F_SINK)
KERNEL(k_cvtpk, F_DECL, REP32(CVTPK), F_SINK)

// ---- packed float (16 independent pairs x 2 per iteration = 32 instructions)
#define P_DECL f32x2 a[16]; _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = f32x2{seedf + i, seedf - i}; f32x2 c = {seedf, seedf};
#define P_SINK float s = 0; _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i][0] + a[i][1]; if (s == 12345.678f) out[0] = 1;
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[(i) & 15]) : "v"(c));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[(i) & 15]) : "v"(c));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[(i) & 15]) : "v"(c));
KERNEL(k_pkfma, P_DECL, REP32(PKFMA), P_SINK)
KERNEL(k_pkmul, P_DECL, REP32(PKMUL), P_SINK)
KERNEL(k_pkadd, P_DECL, REP32(PKADD), P_SINK)

// ---- integer
#define U_DECL unsigned a[32]; _Pragma("unroll") for (int i = 0; i < 32; ++i) a[i] = seedu + 977u * i;
#define U_SINK unsigned s = 0; _Pragma("unroll") for (int i = 0; i < 32; ++i) s ^= a[i]; if (s == 0x12345678u) out[0] = 1;
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seedu));
#define XORU(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(seedu));
#define ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(a[i]));
#define XAD(i) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seedu));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seedu));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seedu));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seedu));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(seedu));
#define MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(seedu));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seedu));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(seedu));
#define PKSUBI16(i) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(seedu));
#define PKASHR(i) asm volatile("v_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "+v"(a[i]));
#define PKMAXU16(i) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(seedu));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seedu));
#define CMPSEL(i) asm volatile("v_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seedu) : "vcc");
#define CMPSDWASEL(i) asm volatile("v_cmp_ge_u32_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:DWORD\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seedu) : "vcc");
KERNEL(k_addu, U_DECL, REP32(ADDU), U_SINK)
KERNEL(k_xor, U_DECL, REP32(XORU), U_SINK)
KERNEL(k_alignbit, U_DECL, REP32(ALIGNBIT), U_SINK)
KERNEL(k_xad, U_DECL, REP32(XAD), U_SINK)
KERNEL(k_add3, U_DECL, REP32(ADD3), U_SINK)
KERNEL(k_mullo, U_DECL, REP32(MULLO), U_SINK)
KERNEL(k_mulhi, U_DECL, REP32(MULHI), U_SINK)
KERNEL(k_mul24, U_DECL, REP32(MUL24), U_SINK)
KERNEL(k_mulhi24, U_DECL, REP32(MULHI24), U_SINK)
KERNEL(k_mad24, U_DECL, REP32(MAD24), U_SINK)
KERNEL(k_bfi, U_DECL, REP32(BFI), U_SINK)
KERNEL(k_pksubi16, U_DECL, REP32(PKSUBI16), U_SINK)
KERNEL(k_pkashr, U_DECL, REP32(PKASHR), U_SINK)
KERNEL(k_pkmaxu16, U_DECL, REP32(PKMAXU16), U_SINK)
KERNEL(k_perm, U_DECL, REP32(PERM), U_SINK)
KERNEL(k_cmpsel, U_DECL, REP32(CMPSEL), U_SINK)          // 2 instructions per slot
KERNEL(k_cmpsdwasel, U_DECL, REP32(CMPSDWASEL), U_SINK)  // 2 instructions + s_nop 1 per slot

// ---- LDS table lookups: random 8-byte reads from a 22 KiB table (the GELU LUT idea), 32 per iteration
__global__ __launch_bounds__(512) void k_lut64(unsigned long long* out, float seedf, unsigned seedu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 22528 / 4; i += blockDim.x) ((unsigned*)lds)[i] = i * 2654435761u;
  __syncthreads();
  unsigned idx[32];
  unsigned h = seedu + threadIdx.x * 747796405u + blockIdx.x;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    h = h * 1664525u + 1013904223u;
    idx[i] = ((h >> 9) % 2816u) * 8u;
  }
  unsigned acc0 = 0, acc1 = 0;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const uint2 v = *(const uint2*)(lds + idx[i]);
      acc0 += v.x;
      acc1 ^= v.y;
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) idx[i] = (idx[i] + 8u * ((acc0 >> 7) & 3u)) % 22528u;   // keeps the addresses data-dependent and spread
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (acc0 + acc1 == 0x12345678u) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long*, float, unsigned);
struct Entry { const char* name; kern_t k; int instr_per_slot; int lds; };

int main() {
  Entry es[] = {
      {"v_fma_f32", k_fma, 1, 0}, {"v_mul_f32", k_mul, 1, 0}, {"v_exp_f32", k_exp, 1, 0}, {"v_rcp_f32", k_rcp, 1, 0},
      {"v_cvt_pk_bf16_f32", k_cvtpk, 1, 0}, {"v_pk_fma_f32", k_pkfma, 1, 0}, {"v_pk_mul_f32", k_pkmul, 1, 0}, {"v_pk_add_f32", k_pkadd, 1, 0},
      {"v_add_u32", k_addu, 1, 0}, {"v_xor_b32", k_xor, 1, 0}, {"v_alignbit_b32", k_alignbit, 1, 0}, {"v_xad_u32", k_xad, 1, 0},
      {"v_add3_u32", k_add3, 1, 0}, {"v_mul_lo_u32", k_mullo, 1, 0}, {"v_mul_hi_u32", k_mulhi, 1, 0}, {"v_mul_u32_u24", k_mul24, 1, 0},
      {"v_mul_hi_u32_u24", k_mulhi24, 1, 0}, {"v_mad_u32_u24", k_mad24, 1, 0}, {"v_bfi_b32", k_bfi, 1, 0},
      {"v_pk_sub_i16 clamp", k_pksubi16, 1, 0}, {"v_pk_ashrrev_i16", k_pkashr, 1, 0}, {"v_pk_max_u16", k_pkmaxu16, 1, 0},
      {"v_perm_b32", k_perm, 1, 0}, {"v_cmp_ge_u32+v_cndmask (pair)", k_cmpsel, 1, 0},
      {"v_cmp_sdwa+s_nop1+v_cndmask (triple)", k_cmpsdwasel, 1, 0}, {"ds_read_b64 random LUT (22 KiB)", k_lut64, 1, 22528},
  };
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * 256 * 8);
  std::vector<unsigned long long> h(256 * 8);
  printf("%-42s %12s %12s\n", "instruction (cycles per wave-instr per SIMD)", "1 wave/SIMD", "2 waves/SIMD");
  for (auto& e : es) {
    double res[2];
    for (int w = 0; w < 2; ++w) {
      const int threads = w == 0 ? 256 : 512;
      for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), e.lds, 0, d, 1.0001f, 12345u);
        hipDeviceSynchronize();
      }
      const int nw = 256 * threads / 64;
      hipMemcpy(h.data(), d, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.begin() + nw);
      const double ticks = (double)h[nw / 2];
      // per-wave cycles per slot; per SIMD throughput = that / waves per SIMD
      res[w] = ticks / (double)(ITER * 32) / (w == 0 ? 1.0 : 2.0);
    }
    printf("%-42s %12.2f %12.2f\n", e.name, res[0], res[1]);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(err)); return 1; }
  return 0;
}
