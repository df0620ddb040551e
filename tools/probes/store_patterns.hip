// Probe: per-CU store rate by access pattern (MI355X).  One workgroup of 512 threads per CU writes 256-row x 512-byte tiles of a
// [50176, 6144-byte] matrix; one wave owns 128 rows x 128 bytes.  Patterns (one wave-instruction):
//   0: 16 rows x 64 B  (16 B per lane; the GEMM epilogue's "wide" form)      1: 8 rows x 128 B (16 B per lane)
//   2: 16 rows x 32 B  (8 B per lane)                                         3: 32 rows x 32 B... (not used)
//   4: 4 rows x 256 B  (two waves' columns; needs an exchange in a real kernel)  5: 1 row x 1 KiB (contiguous)
//   6: pattern 0 with nt stores    7: pattern 1 with nt stores    8: pattern 0, 4 B per lane (16 rows x 16 B)
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int PAT>
__global__ __launch_bounds__(512) void probe(char* out, int rounds, unsigned long long* stamps, int active) {
  const int bid = blockIdx.x;
  if (bid >= active) return;
  const int G = active;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const long long rowb = 3072 * 2;
  unsigned long long t_issue = 0, t_done = 0;
  u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)(50176ll * 3072 * 2), 0x00020000);   // the allocation: out-of-range stores are dropped
  for (int r = 0; r < rounds; ++r) {
    const int tile = bid + r * G;
    const int tm = tile / 12, tn = tile % 12;
    const long long tbase = (long long)tm * 256 * rowb + tn * 512;          // tile origin
    const long long base = tbase + (long long)(wm * 128) * rowb + wn * 128;  // wave origin: 128 rows x 128 B
    __syncthreads();
    const unsigned long long ta = __builtin_amdgcn_s_memrealtime();
    if (PAT == 0 || PAT == 6) {
      for (int i = 0; i < 8; ++i)
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned off = (unsigned)(base + (long long)(16 * i + (lane & 15)) * rowb + hh * 64 + (lane >> 4) * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, PAT == 6 ? 2 : 0);
        }
    } else if (PAT == 1 || PAT == 7) {
      for (int i = 0; i < 16; ++i) {
        const unsigned off = (unsigned)(base + (long long)(8 * i + (lane >> 3)) * rowb + (lane & 7) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, PAT == 7 ? 2 : 0);
      }
    } else if (PAT == 2) {
      for (int i = 0; i < 8; ++i)
        for (int q = 0; q < 4; ++q) {
          const unsigned off = (unsigned)(base + (long long)(16 * i + (lane & 15)) * rowb + q * 32 + (lane >> 4) * 8);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[0], v[1]}, rs, off, 0, 0);
        }
    } else if (PAT == 8) {
      for (int i = 0; i < 8; ++i)
        for (int q = 0; q < 8; ++q) {
          const unsigned off = (unsigned)(base + (long long)(16 * i + (lane & 15)) * rowb + q * 16 + (lane >> 4) * 4);
          __builtin_amdgcn_raw_buffer_store_b32(v[0], rs, off, 0, 0);
        }
    } else if (PAT == 4) {
      // the workgroup's 256 rows x 512 B as 4-row x 256-B instructions: wave w writes rows 32 w .. 32 w + 31, both 256-B halves
      for (int i = 0; i < 8; ++i)
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned off = (unsigned)(tbase + (long long)(32 * wave + 4 * i + (lane >> 4)) * rowb + hh * 256 + (lane & 15) * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
        }
    } else if (PAT == 9) {
      // 16 rows x 64 B, but the 4 lanes of a quad are contiguous (row = lane >> 2)
      for (int i = 0; i < 8; ++i)
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned off = (unsigned)(base + (long long)(16 * i + (lane >> 2)) * rowb + hh * 64 + (lane & 3) * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
        }
    } else if (PAT == 10) {
      // 8 rows x 128 B with adjacent lanes on DIFFERENT rows (row = lane & 7)
      for (int i = 0; i < 16; ++i) {
        const unsigned off = (unsigned)(base + (long long)(8 * i + (lane & 7)) * rowb + (lane >> 3) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
      }
    } else if (PAT == 11) {
      // 16 rows x 64 B, lane pairs contiguous (32 B): row = (lane >> 1) & 15, chunk = (lane & 1) + 2 (lane >> 5)
      for (int i = 0; i < 8; ++i)
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned off = (unsigned)(base + (long long)(16 * i + ((lane >> 1) & 15)) * rowb + hh * 64 + ((lane & 1) + 2 * (lane >> 5)) * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
        }
    } else if (PAT == 12 || PAT == 13) {
      // LOADS of the same footprint: 12 = 16 rows x 64 B with row = lane & 15 (the epilogue's operand loads), 13 = 8 rows x 128 B, row = lane >> 3
      u32x4 a[16];
      for (int i = 0; i < 16; ++i) {
        const unsigned off = PAT == 12 ? (unsigned)(base + (long long)(16 * (i >> 1) + (lane & 15)) * rowb + (i & 1) * 64 + (lane >> 4) * 16)
                                       : (unsigned)(base + (long long)(8 * i + (lane >> 3)) * rowb + (lane & 7) * 16);
        a[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      }
      for (int i = 0; i < 16; ++i) v ^= a[i];
    } else if (PAT == 5) {
      // fully contiguous 1 KiB per instruction (a different matrix shape: what the path can do at best)
      for (int i = 0; i < 16; ++i) {
        const unsigned off = (unsigned)((long long)(tile % 2304) * 131072 + (wave * 16 + i) * 1024 + lane * 16);   // < 2304 x 128 KiB = 302 MB
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
      }
    }
    const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tc = __builtin_amdgcn_s_memrealtime();
    t_issue += tb - ta;
    t_done += tc - ta;
  }
  if (v[0] == 0x12345u) out[0] = 1;     // keeps the loads of patterns 12 / 13 alive
  if (lane == 0) {
    stamps[(bid * 8 + wave) * 2 + 0] = t_issue;
    stamps[(bid * 8 + wave) * 2 + 1] = t_done;
  }
}

template <int PAT>
void run(const char* name, char* out, unsigned long long* st) {
  std::vector<unsigned long long> h(256 * 8 * 2);
  const int actives[] = {8, 64, 256};
  for (int a : actives) {
    const int rounds = 9;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(512), 0, 0, out, rounds, st, a);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), st, a * 8 * 2 * 8, hipMemcpyDeviceToHost);
    double iss = 0, don = 0;
    for (int i = 0; i < a * 8; ++i) {
      iss += h[2 * i];
      don += h[2 * i + 1];
    }
    iss = iss / (a * 8) / rounds / 100.0;
    don = don / (a * 8) / rounds / 100.0;
    const double bytes = 256.0 * 512;
    printf("%-34s active %-4d | issue %6.2f us  done %6.2f us  %6.1f GB/s per CU  %5.2f B/clk @2.4GHz\n", name, a, iss, don, bytes / (don * 1e-6) / 1e9,
           bytes / (don * 1e-6) / 2.4e9);
  }
}

int main() {
  const long long img = 50176ll * 3072 * 2;
  char* out;
  unsigned long long* st;
  if (hipMalloc(&out, img + (64 << 20)) != hipSuccess) return 1;
  if (hipMalloc(&st, 256 * 8 * 2 * 8) != hipSuccess) return 1;
  run<0>("16 rows x 64 B (b128)", out, st);
  run<1>("8 rows x 128 B (b128)", out, st);
  run<2>("16 rows x 32 B (b64)", out, st);
  run<8>("16 rows x 16 B (b32)", out, st);
  run<4>("4 rows x 256 B (b128)", out, st);
  run<5>("1 KiB contiguous (b128)", out, st);
  run<9>("16 rows x 64 B, quad-contiguous", out, st);
  run<11>("16 rows x 64 B, pair-contiguous", out, st);
  run<10>("8 rows x 128 B, row = lane & 7", out, st);
  run<12>("LOAD 16 rows x 64 B, row = lane&15", out, st);
  run<13>("LOAD 8 rows x 128 B, row = lane>>3", out, st);
  run<6>("16 rows x 64 B (b128, nt)", out, st);
  run<7>("8 rows x 128 B (b128, nt)", out, st);
  return 0;
}
