// Probe: how fast can ONE compute unit write an output tile, alone and beside the others?  (MI355X, gfx950)
// Every workgroup (512 threads, one per CU) writes 256 x 256 bf16 tiles of a [M, 3072] bf16 matrix the way the GEMM epilogue does
// (one wave-instruction = 16 rows x 64 bytes), `images` tiles per round (1 = plain store, 2 = the GELU pair), optionally reads a
// tile of the same shape first (the dGELU / residual operands), and spins `gap_us` between rounds (the K loop: no memory traffic).
// Phases of the workgroups: aligned (all burst together) or spread uniformly over one period.
//   hipcc --offload-arch=gfx950 -O3 -o store_rates store_rates.hip && ./store_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(char* out, const char* in, int rounds, int images, int loads, int gap_ticks, int spread_ticks,
                                             unsigned long long* stamps, int active) {
  const int bid = blockIdx.x;
  if (bid >= active) return;
  const int G = active;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const long long rowb = 3072 * 2;
  const long long img = 50176ll * rowb;
  if (spread_ticks > 0 && wave == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long d = (unsigned long long)spread_ticks * ((bid * 37) % G) / G;
    while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  unsigned long long t_issue = 0, t_done = 0;
  u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  for (int r = 0; r < rounds; ++r) {
    const int tile = bid + r * G;                     // 12 tile columns, row-major raster
    const int tm = tile / 12, tn = tile % 12;
    const long long base = ((long long)tm * 256 + wm * 128) * rowb + (tn * 256 + wn * 64) * 2;
    if (gap_ticks > 0) {
      if (wave == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)gap_ticks) __builtin_amdgcn_s_sleep(8);
      }
      __syncthreads();
    }
    const unsigned long long ta = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 8; ++i) {                     // 8 row tiles of 16 rows
      const long long ro = base + (long long)(16 * i + (lane & 15)) * rowb;
      if (loads) {
        for (int l = 0; l < loads; ++l) {
          const u32x4 a = *(const u32x4*)(in + l * img + ro + (lane >> 4) * 16);
          const u32x4 b = *(const u32x4*)(in + l * img + ro + 64 + (lane >> 4) * 16);
          v ^= a + b;
        }
      }
      for (int im = 0; im < images; ++im) {
        *(u32x4*)(out + im * img + ro + (lane >> 4) * 16) = v;            // columns 0..31 of the wave's 64
        *(u32x4*)(out + im * img + ro + 64 + (lane >> 4) * 16) = v;       // columns 32..63
      }
    }
    const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tc = __builtin_amdgcn_s_memrealtime();
    t_issue += tb - ta;
    t_done += tc - ta;
  }
  if (lane == 0) {
    stamps[(bid * 8 + wave) * 2 + 0] = t_issue;
    stamps[(bid * 8 + wave) * 2 + 1] = t_done;
  }
}

int main() {
  const long long img = 50176ll * 3072 * 2;
  char *out, *in;
  unsigned long long* st;
  hipMalloc(&out, 2 * img);
  hipMalloc(&in, 2 * img);
  hipMemset(in, 1, 2 * img);
  hipMalloc(&st, 256 * 8 * 2 * 8);
  std::vector<unsigned long long> h(256 * 8 * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-8s %-7s %-6s %-6s %-7s | %-10s %-10s %-10s %-12s %-10s\n", "active", "images", "loads", "gap", "spread", "issue us", "done us", "span us", "GB/s per CU", "chip TB/s");
  const int actives[] = {8, 32, 64, 128, 256};
  struct Case { int images, loads, gap, spread; };
  const Case cases[] = {{1, 0, 0, 0}, {2, 0, 0, 0}, {1, 1, 0, 0}, {2, 0, 1800, 0}, {2, 0, 1800, 2800}, {1, 1, 1800, 0}, {1, 1, 1800, 2800}, {1, 2, 1800, 0}, {1, 2, 1800, 2800}};
  for (const Case& c : cases)
    for (int a : actives) {
      const int rounds = 9;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, in, rounds, c.images, c.loads, c.gap, c.spread, st, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
      }
      hipMemcpy(h.data(), st, a * 8 * 2 * 8, hipMemcpyDeviceToHost);
      double iss = 0, don = 0;
      for (int i = 0; i < a * 8; ++i) {
        iss += h[2 * i];
        don += h[2 * i + 1];
      }
      iss = iss / (a * 8) / rounds / 100.0;      // 100 MHz ticks -> us per round
      don = don / (a * 8) / rounds / 100.0;
      // words per lane here are 16 B (twice the bf16 epilogue's 8 B): a "tile" of this probe is 256 rows x 512 B = 128 KiB per image
      const double bytes = (double)(c.images + c.loads) * 256 * 512;
      printf("%-8d %-7d %-6d %-6d %-7d | %-10.2f %-10.2f %-10.1f %-12.1f %-10.2f\n", a, c.images, c.loads, c.gap / 100, c.spread / 100, iss, don, best * 1e3,
             bytes / (don * 1e-6) / 1e9, bytes * a / (don * 1e-6) / 1e12);
    }
  return 0;
}
