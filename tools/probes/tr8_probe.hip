// Probe of ds_read_b64_tr_b8 (gfx950): which byte does lane l / result byte e come from?
// LDS holds byte(o) = o & 0xff; lane l of a 16-lane group reads address 128*group + 16*(l>>1) + 8*(l&1)
// (8 rows of 16 bytes per group).   hipcc --offload-arch=gfx950 tr8_probe.hip -o tr8_probe && ./tr8_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;
__global__ void k(unsigned char* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned char)(i & 0xff);
  __syncthreads();
  const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
  const unsigned char* a = lds + 128 * g + 16 * (l >> 1) + 8 * (l & 1);
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)a);
  *(i32x2*)(out + 8 * threadIdx.x) = v;
}
int main() {
  unsigned char* d;
  hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned char h[512];
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int t = 0; t < 64; ++t) {
    printf("lane %2d:", t);
    for (int e = 0; e < 8; ++e) printf(" %3d", h[8 * t + e]);
    printf("\n");
  }
  return 0;
}
