// Device-side input transforms (reference utils/data_utils.py:21-81): the torchvision pipelines
//   cifar10 train : RandomCrop(32, padding=4) -> RandomHorizontalFlip -> Resize(S) -> ToTensor -> Normalize
//   imagenet train: RandomResizedCrop(S) -> RandomHorizontalFlip -> ToTensor -> Normalize
//   test / default: Resize(r) [-> CenterCrop(S)] -> ToTensor -> Normalize
// run as two kernels on raw uint8 HWC images that were copied to the device asynchronously, instead of per-image
// PIL work in DataLoader worker processes followed by a blocking .to(device).
//
// torchvision's Resize / RandomResizedCrop on PIL images ARE Pillow's Image.resize(BILINEAR): an 8-bit, two-pass
// (horizontal, then vertical) resampling with double-precision triangle-filter coefficients rounded to 22-bit fixed
// point and a uint8 intermediate image.  These kernels restate exactly that arithmetic (Pillow's
// precompute_coeffs / normalize_coeffs_8bpc / ImagingResample{Horizontal,Vertical}_8bpc), so the resized bytes are
// BIT-IDENTICAL to Pillow's; ToTensor (/255) and Normalize ((x - mean) / std) follow in fp32 as torchvision does.
// Pillow is importable in this image and is the checker in tests/test_data_pipeline.py (torchvision is not).
//
// Per-image parameters (int32 x 12): crop top, left, height, width in the zero-padded source; pad; resized
// height, width of the crop; output window origin oy, ox inside the resized crop; flip_src (flip the crop before
// resizing), flip_out (flip the output window), reserved.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;        // Pillow src/libImaging/Resample.c
constexpr int MAX_TAPS = 64;

struct Taps {
  int xmin, n;
  int kk[MAX_TAPS];
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc for ONE output position xx (bilinear = triangle filter, support 1)
__device__ __forceinline__ void pil_taps(int in_size, int out_size, int xx, Taps& t) {
#pragma clang fp contract(off)        // Pillow's C is compiled without fused multiply-add
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > MAX_TAPS) xmax = MAX_TAPS;
  double k[MAX_TAPS];
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = (x + xmin - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    const double w = a < 1.0 ? 1.0 - a : 0.0;
    k[x] = w;
    ww += w;
  }
  for (int x = 0; x < xmax; ++x) {
    double v = k[x];
    if (ww != 0.0) v /= ww;
    t.kk[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << PRECISION_BITS)) : (int)(0.5 + v * (double)(1 << PRECISION_BITS));
  }
  t.xmin = xmin;
  t.n = xmax;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;                              // arithmetic shift, then Pillow's clip8 lookup clamps
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[b][y][x][c] for crop rows y < ch and resized columns x < rw that the output window needs
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                                                         const int* __restrict__ prm, int Hs, int Ws, int C, int ch_max,
                                                         int S) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;       // output-window column
  const int* p = prm + b * 12;
  const int top = p[0], left = p[1], ch = p[2], cw = p[3], pad = p[4], rw = p[6], ox = p[8], flip_src = p[9], flip_out = p[10];
  if (x >= S || y >= ch) return;
  const int xr = ox + (flip_out ? S - 1 - x : x);     // column of the resized crop
  uint8_t* o = tmp + (((long)b * ch_max + y) * S + x) * C;
  if (xr < 0 || xr >= rw) { for (int c = 0; c < C; ++c) o[c] = 0; return; }
  Taps t;
  pil_taps(cw, rw, xr, t);
  const int sy = top + y - pad;
  const uint8_t* img = src + (long)b * Hs * Ws * C;
  int acc[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
  for (int k = 0; k < t.n; ++k) {
    const int xc = t.xmin + k;                        // column inside the crop
    const int sx = left + (flip_src ? cw - 1 - xc : xc) - pad;
    if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) {   // zero padding elsewhere (RandomCrop fill = 0)
      const uint8_t* q = img + ((long)sy * Ws + sx) * C;
      for (int c = 0; c < C; ++c) acc[c] += (int)q[c] * t.kk[k];
    }
  }
  for (int c = 0; c < C; ++c) o[c] = (uint8_t)clip8(acc[c]);
}

// vertical pass + ToTensor + Normalize: out[b][c][y][x] fp32; also the resized bytes (optional, tests)
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ out,
                                                         uint8_t* __restrict__ out_u8, const int* __restrict__ prm, int C,
                                                         int ch_max, int S, float3 mean, float3 stdv) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= S) return;
  const int* p = prm + b * 12;
  const int ch = p[2], rh = p[5], oy = p[7];
  const int yr = oy + y;
  int v[4] = {0, 0, 0, 0};
  if (yr >= 0 && yr < rh) {
    Taps t;
    pil_taps(ch, rh, yr, t);
    int acc[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int k = 0; k < t.n; ++k) {
      const uint8_t* q = tmp + (((long)b * ch_max + (t.xmin + k)) * S + x) * C;
      for (int c = 0; c < C; ++c) acc[c] += (int)q[c] * t.kk[k];
    }
    for (int c = 0; c < C; ++c) v[c] = clip8(acc[c]);
  }
  const float m[3] = {mean.x, mean.y, mean.z}, sd[3] = {stdv.x, stdv.y, stdv.z};
  for (int c = 0; c < C; ++c) {
    if (out_u8) out_u8[(((long)b * S + y) * S + x) * C + c] = (uint8_t)v[c];
    // ToTensor: byte / 255 (fp32 division, as torch's .div(255)); Normalize: (t - mean) / std
    const float tv = (float)v[c] / 255.0f;
    out[(((long)b * C + c) * S + y) * S + x] = (tv - m[c < 3 ? c : 2]) / sd[c < 3 ? c : 2];
  }
}

}  // namespace

extern "C" int favit_image_transform(const uint8_t* src, uint8_t* tmp, float* out, uint8_t* out_u8, const int32_t* params,
                                     int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t ch_max, int32_t S,
                                     const float* mean, const float* std, void* stream) {
  if (!src || !tmp || !out || !params || !mean || !std || B <= 0 || Hs <= 0 || Ws <= 0 || S <= 0 || ch_max <= 0)
    return FAVIT_ERR_INVALID;
  if (C < 1 || C > 4) return FAVIT_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  const dim3 gh((unsigned)((S + 255) / 256), (unsigned)ch_max, (unsigned)B);
  hipLaunchKernelGGL(resample_h_kernel, gh, dim3(256), 0, st, src, tmp, params, Hs, Ws, C, ch_max, S);
  FAVIT_CHECK_LAUNCH();
  const dim3 gv((unsigned)((S + 255) / 256), (unsigned)S, (unsigned)B);
  const float3 m = make_float3(mean[0], mean[C > 1 ? 1 : 0], mean[C > 2 ? 2 : 0]);
  const float3 sd = make_float3(std[0], std[C > 1 ? 1 : 0], std[C > 2 ? 2 : 0]);
  hipLaunchKernelGGL(resample_v_kernel, gv, dim3(256), 0, st, tmp, out, out_u8, params, C, ch_max, S, m, sd);
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}
