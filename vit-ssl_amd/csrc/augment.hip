// DINO multi-crop input pipeline on the GPU (gfx950): per-view image arithmetic of
// data/datasets.py:80-123 with the transform lists of configs/dino/globals.yaml / locals.yaml
// (RandomResizedCrop -> flip -> ColorJitter -> [grayscale] -> GaussianBlur -> ToTensor).
//
// The arithmetic is Pillow's (uint8 images, fixed-point resampling, float blends with
// truncation, float HSV conversion) and torchvision's float32 blur; every kernel reproduces
// it operation for operation (oracle/augment_oracle.py, pinned against Pillow):
//   * no FMA contraction anywhere (Pillow / NumPy round the product and the sum separately);
//   * doubles exactly where the C sources promote to double.
// All kernels are HBM / latency-bound byte work: coalesced channel-interleaved rows, one
// workgroup per image for the colour chain (the contrast step needs the image's mean).
//
// Layouts: src uint8 [B, H, W, 3]; intermediates uint8 [B, S, S, 3]; output float32 [B, 3, S, S].
// Per-image parameters (device arrays):
//   iparams int32 [B, 11] = top, left, h, w, flip, order0..3 (0 brightness, 1 contrast,
//                           2 saturation, 3 hue), gray, hue_shift (uint8 added to H)
//   fparams f32   [B, 10] = brightness, contrast, saturation, k1d[0..6] (normalised Gaussian)
#include "common.h"

// Pillow / NumPy round every product and every sum separately; hipcc's default
// (-ffp-contract=fast) fuses `a + b * c` into one FMA even through __fmul_rn / __fadd_rn, and a
// file-scope `#pragma clang fp contract(off)` did not stop it (162 v_fma in the blur kernel):
// this file is compiled with -ffp-contract=off (__graft_entry__.py, PER_FILE_FLAGS).

namespace {

constexpr int AUG_IP = 11, AUG_FP = 10;
constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow Resample.c

__device__ __forceinline__ double bilinear_filter(double x) {
  x = x < 0.0 ? -x : x;
  return x < 1.0 ? 1.0 - x : 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output position xx, evaluated on the
// fly (BILINEAR, support 1).  Returns the first source index; writes up to `cap` fixed-point
// taps to k[] and their number to *count.
__device__ __forceinline__ int resample_taps(int in_size, int out_size, int xx, int* k, int cap, int* count) {
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = filterscale;
  const double ss = 1.0 / filterscale;
  const double center = ((double)xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > cap) xmax = cap;          // cannot happen for cap >= 2*ceil(scale)+1 (checked on the host)
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bilinear_filter(((double)(x + xmin) - center + 0.5) * ss);
  for (int x = 0; x < xmax; ++x) {
    double w = bilinear_filter(((double)(x + xmin) - center + 0.5) * ss);
    if (ww != 0.0) w /= ww;
    k[x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
  }
  *count = xmax;
  return xmin;
}

__device__ __forceinline__ unsigned char clip8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

constexpr int MAX_TAPS = 64;   // source / output size ratio up to 31

// horizontal pass: tmp[b][y][xx][c] for crop rows y < h;  grid = (H, B), threads over xx (the
// taps are shared by the three channels; mapping threads over xx*3 + c for contiguous byte
// stores was measured slower: 166 vs 129 us, the redundant double-precision taps cost more)
__global__ void aug_resize_h_kernel(const unsigned char* __restrict__ src, const int* __restrict__ ip, unsigned char* __restrict__ tmp,
                                    int H, int W, int S) {
  const int b = blockIdx.y, y = blockIdx.x;
  const int* p = ip + b * AUG_IP;
  const int top = p[0], left = p[1], h = p[2], w = p[3];
  if (y >= h) return;
  const unsigned char* row = src + (((long long)b * H + top + y) * W + left) * 3;
  unsigned char* orow = tmp + (((long long)b * H + y) * S) * 3;
  for (int xx = threadIdx.x; xx < S; xx += blockDim.x) {
    int k[MAX_TAPS], n;
    const int xmin = resample_taps(w, S, xx, k, MAX_TAPS, &n);
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
      const unsigned char* px = row + (xmin + x) * 3;
      s0 += px[0] * k[x];
      s1 += px[1] * k[x];
      s2 += px[2] * k[x];
    }
    orow[xx * 3 + 0] = clip8(s0 >> PRECISION_BITS);
    orow[xx * 3 + 1] = clip8(s1 >> PRECISION_BITS);
    orow[xx * 3 + 2] = clip8(s2 >> PRECISION_BITS);
  }
}

// vertical pass + horizontal flip on store: dst[b][yy][x'][c];  grid = (S, B)
__global__ void aug_resize_v_kernel(const unsigned char* __restrict__ tmp, const int* __restrict__ ip, unsigned char* __restrict__ dst,
                                    int H, int S) {
  const int b = blockIdx.y, yy = blockIdx.x;
  const int* p = ip + b * AUG_IP;
  const int h = p[2], flip = p[4];
  __shared__ int k[MAX_TAPS];
  __shared__ int meta[2];
  if (threadIdx.x == 0) {
    int n;
    meta[0] = resample_taps(h, S, yy, k, MAX_TAPS, &n);
    meta[1] = n;
  }
  __syncthreads();
  const int ymin = meta[0], n = meta[1];
  const unsigned char* base = tmp + ((long long)b * H * S) * 3;
  unsigned char* orow = dst + (((long long)b * S + yy) * S) * 3;
  for (int i = threadIdx.x; i < S * 3; i += blockDim.x) {
    int s = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < n; ++y) s += base[(long long)(ymin + y) * S * 3 + i] * k[y];
    const int x = i / 3, c = i - 3 * x;
    const int xo = flip ? S - 1 - x : x;
    orow[xo * 3 + c] = clip8(s >> PRECISION_BITS);
  }
}

// ---------------------------------------------------------------- colour chain
__device__ __forceinline__ unsigned char lum8(int r, int g, int b) { return (unsigned char)((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16); }

// Pillow ImagingBlend(deg, img, f): float, product and sum rounded separately, truncation
__device__ __forceinline__ unsigned char blend8(int deg, int v, float f, bool interp) {
  const float t = __fadd_rn((float)deg, __fmul_rn(f, (float)(v - deg)));
  if (interp) return (unsigned char)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)t);
}

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) {
    uh = 0;
    us = 0;
    return;
  }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float h;
  if (r == maxc) h = __fsub_rn(bc, gc);
  else if (g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
  else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
  // fmod(h / 6.0 + 1.0, 1.0): the argument lies in (0.8, 1.9), so the remainder is an exact
  // subtraction of its integer part (the library fmod is a long software loop)
  const double hx = __dadd_rn(__ddiv_rn((double)h, 6.0), 1.0);
  h = (float)(hx - floor(hx));
  const int ih = (int)__dmul_rn((double)h, 255.0), is = (int)__dmul_rn((double)s, 255.0);
  uh = ih < 0 ? 0 : (ih > 255 ? 255 : ih);
  us = is < 0 ? 0 : (is > 255 ? 255 : is);
}

__device__ __forceinline__ int round_clip8(float x) {   // C round() on the double-promoted value, then CLIP8
  const double d = floor((double)x + 0.5);
  return d < 0.0 ? 0 : (d > 255.0 ? 255 : (int)d);
}

__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) {
    r = g = b = v;
    return;
  }
  const float hf = __fdiv_rn(__fmul_rn((float)h, 6.0f), 255.0f);
  const float fi = floorf(hf);
  const float f = __fsub_rn(hf, fi);
  const float fs = __fdiv_rn((float)s, 255.0f);
  const float vf = (float)v;
  const int p = round_clip8(__fmul_rn(vf, __fsub_rn(1.0f, fs)));
  const int q = round_clip8(__fmul_rn(vf, __fsub_rn(1.0f, __fmul_rn(fs, f))));
  const int t = round_clip8(__fmul_rn(vf, __fsub_rn(1.0f, __fmul_rn(fs, __fsub_rn(1.0f, f)))));
  switch (((int)fi) % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// one workgroup per image, the image lives in LDS for the whole chain (S*S*3 <= 150 KiB)
__global__ __launch_bounds__(1024) void aug_color_kernel(unsigned char* __restrict__ img, const int* __restrict__ ip,
                                                        const float* __restrict__ fp, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_img[];
  __shared__ unsigned long long wsum[16];
  __shared__ int mean_s;
  const int b = blockIdx.x;
  const int* p = ip + b * AUG_IP;
  const float* f = fp + b * AUG_FP;
  const int npix = S * S;
  unsigned char* g = img + (long long)b * npix * 3;
  for (int i = threadIdx.x; i < npix * 3 / 4; i += blockDim.x) ((unsigned*)lds_img)[i] = ((const unsigned*)g)[i];
  __syncthreads();
  for (int step = 0; step < 4; ++step) {
    const int fn = p[5 + step];
    if (fn == 0) {                                   // brightness: blend with black
      const float fac = f[0];
      const bool interp = fac >= 0.f && fac <= 1.f;
      for (int i = threadIdx.x; i < npix * 3; i += blockDim.x) lds_img[i] = blend8(0, lds_img[i], fac, interp);
    } else if (fn == 1) {                            // contrast: blend with the mean luminance
      unsigned long long part = 0;
      for (int i = threadIdx.x; i < npix; i += blockDim.x) part += lum8(lds_img[3 * i], lds_img[3 * i + 1], lds_img[3 * i + 2]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
      if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += wsum[w];
        mean_s = (int)((double)tot / (double)npix + 0.5);     // int(ImageStat mean + 0.5)
      }
      __syncthreads();
      const int m = mean_s;
      const float fac = f[1];
      const bool interp = fac >= 0.f && fac <= 1.f;
      for (int i = threadIdx.x; i < npix * 3; i += blockDim.x) lds_img[i] = blend8(m, lds_img[i], fac, interp);
    } else if (fn == 2) {                            // saturation: blend with the pixel's luminance
      const float fac = f[2];
      const bool interp = fac >= 0.f && fac <= 1.f;
      for (int i = threadIdx.x; i < npix; i += blockDim.x) {
        const int r = lds_img[3 * i], gg = lds_img[3 * i + 1], bb = lds_img[3 * i + 2];
        const int l = lum8(r, gg, bb);
        lds_img[3 * i] = blend8(l, r, fac, interp);
        lds_img[3 * i + 1] = blend8(l, gg, fac, interp);
        lds_img[3 * i + 2] = blend8(l, bb, fac, interp);
      }
    } else {                                         // hue: uint8 wrap-around shift of H
      const int shift = p[10];
      for (int i = threadIdx.x; i < npix; i += blockDim.x) {
        int h, s, v, r, gg, bb;
        rgb2hsv(lds_img[3 * i], lds_img[3 * i + 1], lds_img[3 * i + 2], h, s, v);
        h = (h + shift) & 255;
        hsv2rgb(h, s, v, r, gg, bb);
        lds_img[3 * i] = (unsigned char)r;
        lds_img[3 * i + 1] = (unsigned char)gg;
        lds_img[3 * i + 2] = (unsigned char)bb;
      }
    }
    __syncthreads();
  }
  if (p[9]) {                                        // RandomGrayscale hit: L replicated
    for (int i = threadIdx.x; i < npix; i += blockDim.x) {
      const unsigned char l = lum8(lds_img[3 * i], lds_img[3 * i + 1], lds_img[3 * i + 2]);
      lds_img[3 * i] = l;
      lds_img[3 * i + 1] = l;
      lds_img[3 * i + 2] = l;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < npix * 3 / 4; i += blockDim.x) ((unsigned*)g)[i] = ((const unsigned*)lds_img)[i];
}

// ---------------------------------------------------------------- blur + ToTensor
// float32 KxK convolution (k2 = k1[dy]*k1[dx]), reflect padding, taps added in (dy, dx) order
// with separate multiply and add, round-half-even, /255, planar store.
// grid = (ceil(S / ROWS), B): a workgroup stages ROWS + K - 1 source rows (reflect-resolved,
// converted to float once, with the horizontal reflect padding materialised) in LDS and every
// thread then walks its 49 taps with conflict-free ds_read_b32.  (The first version read 147
// bytes per pixel straight from global memory: 594 us for 256 views of 224^2.)
template <int K, int ROWS>
__global__ __launch_bounds__(256) void aug_blur_tensor_kernel(const unsigned char* __restrict__ img, const float* __restrict__ fp,
                                                             float* __restrict__ out, int S) {
  extern __shared__ __attribute__((aligned(16))) float rows_f[];   // [ROWS + K - 1][(S + K - 1) * 3]
  constexpr int P = K / 2;
  const int b = blockIdx.y, y0 = blockIdx.x * ROWS;
  const float* k1 = fp + b * AUG_FP + 3;
  float k2[K][K];
#pragma unroll
  for (int dy = 0; dy < K; ++dy)
#pragma unroll
    for (int dx = 0; dx < K; ++dx) k2[dy][dx] = k1[dy] * k1[dx];
  const unsigned char* base = img + (long long)b * S * S * 3;
  const int pw = (S + K - 1) * 3;                       // padded row length in floats
  for (int i = threadIdx.x; i < (ROWS + K - 1) * pw; i += blockDim.x) {
    const int r = i / pw, j = i - r * pw;
    const int xp = j / 3, c = j - 3 * xp;
    int yy = y0 + r - P;
    yy = yy < 0 ? -yy : (yy >= S ? 2 * S - 2 - yy : yy);
    if (yy < 0) yy = 0;                                 // rows past the image bottom of a ragged last block (never used)
    if (yy >= S) yy = S - 1;
    int xx = xp - P;
    xx = xx < 0 ? -xx : (xx >= S ? 2 * S - 2 - xx : xx);
    rows_f[i] = (float)base[((long long)yy * S + xx) * 3 + c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ROWS * S * 3; i += blockDim.x) {
    const int ry = i / (S * 3), j = i - ry * (S * 3);    // j = x * 3 + c
    const int y = y0 + ry;
    if (y >= S) continue;
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < K; ++dy) {
      const float* row = rows_f + (ry + dy) * pw + j;
#pragma unroll
      for (int dx = 0; dx < K; ++dx) acc = acc + k2[dy][dx] * row[dx * 3];
    }
    float r = rintf(acc);
    r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
    const int x = j / 3, c = j - 3 * x;
    out[(((long long)b * 3 + c) * S + y) * S + x] = r / 255.0f;
  }
}

}  // namespace

extern "C" int vitssl_aug_resized_crop_u8(const uint8_t* src, const int32_t* iparams, uint8_t* tmp, uint8_t* dst, int B, int H,
                                          int W, int S, void* stream) {
  VS_CHECK_ARG(src && iparams && tmp && dst && B > 0 && H > 0 && W > 0 && S > 0, "aug_resized_crop: bad args");
  VS_CHECK_ARG((H + S - 1) / S * 2 + 1 <= MAX_TAPS && (W + S - 1) / S * 2 + 1 <= MAX_TAPS,
               "aug_resized_crop: source %dx%d is more than %dx the output size %d", H, W, (MAX_TAPS - 1) / 2, S);
  hipLaunchKernelGGL(aug_resize_h_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, src, iparams, tmp, H, W, S);
  VS_CHECK_LAUNCH("aug_resize_h");
  hipLaunchKernelGGL(aug_resize_v_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, tmp, iparams, dst, H, S);
  VS_CHECK_LAUNCH("aug_resize_v");
  return VITSSL_OK;
}

extern "C" int vitssl_aug_color_u8(uint8_t* img, const int32_t* iparams, const float* fparams, int B, int S, void* stream) {
  VS_CHECK_ARG(img && iparams && fparams && B > 0 && S > 0, "aug_color: bad args");
  const int bytes = S * S * 3;
  VS_CHECK_ARG(bytes % 4 == 0 && bytes <= 150 * 1024, "aug_color: view %dx%d does not fit the 150 KiB LDS image (S <= 224, S even)", S, S);
  static VsOnce attr_done{false};
  if (!attr_done.load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute((const void*)aug_color_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) {
      vitssl_set_error("aug_color: cannot raise dynamic LDS: %s", hipGetErrorString(e));
      return VITSSL_ERR_LAUNCH;
    }
    attr_done.store(true, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(aug_color_kernel, dim3(B), dim3(1024), bytes, (hipStream_t)stream, img, iparams, fparams, S);
  VS_CHECK_LAUNCH("aug_color");
  return VITSSL_OK;
}

extern "C" int vitssl_aug_blur_to_tensor(const uint8_t* img, const float* fparams, float* out, int B, int S, int ksize,
                                         void* stream) {
  VS_CHECK_ARG(img && fparams && out && B > 0 && S > 0, "aug_blur_to_tensor: bad args");
  VS_CHECK_ARG(ksize == 7, "aug_blur_to_tensor: kernel size %d unsupported (the reference configs use 7)", ksize);
  VS_CHECK_ARG(S > ksize / 2, "aug_blur_to_tensor: view smaller than the reflect padding");
  constexpr int ROWS = 8;
  const int lds = (ROWS + 6) * (S + 6) * 3 * (int)sizeof(float);
  VS_CHECK_ARG(lds <= 64 * 1024, "aug_blur_to_tensor: view width %d too large for the LDS row cache", S);
  hipLaunchKernelGGL((aug_blur_tensor_kernel<7, ROWS>), dim3((S + ROWS - 1) / ROWS, B), dim3(256), lds, (hipStream_t)stream, img,
                     fparams, out, S);
  VS_CHECK_LAUNCH("aug_blur_to_tensor");
  return VITSSL_OK;
}
