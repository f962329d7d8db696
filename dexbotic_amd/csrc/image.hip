// Device-side image preprocessing: uint8 RGB frames -> [expand to square] -> bicubic
// resize -> center crop -> rescale / normalise -> planar fp32 / bf16, bit-exact with what the reference gets from
// Pillow's 8-bit resampler under the CLIP image processor:
//   dexbotic/data/dataset/rgb_preprocess.py:13-44  PreprocessRGB.__call__ / expand2square
//   dexbotic/model/dexbotic_arch.py:498-529        process_images
//   Pillow 12.2.0 src/libImaging/Resample.c        precompute_coeffs, normalize_coeffs_8bpc, ImagingResample*_8bpc
// Byte work, HBM/latency bound: two launches per batch of frames (horizontal pass into a uint8 scratch image that
// holds only the rows and columns the crop needs, then vertical pass fused with crop + normalise + HWC->CHW).
#include <math.h>

#include <vector>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Resample.c

double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

int ksize_for(int in_size, int out_size) {
  double filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(2.0 * filterscale) * 2 + 1;
}

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

struct ImgP {
  const uint8_t* src;
  int n, h, w;            // source frames
  int ph, pw, off_y, off_x;   // padded (virtual) frame and where the source sits in it
  uint32_t bg;            // r | g << 8 | b << 16
  int row0, rows;         // padded rows the vertical pass reads: [row0, row0 + rows)
  int crop_top, crop_left, out_h, out_w;
  const int32_t *hb, *hk;
  int hks;
  const int32_t *vb, *vk;
  int vks;
  uint8_t* tmp;           // [n, rows, out_w, 3]
  void* out;              // [n, 3, out_h, out_w]
  uint8_t* out_u8;        // optional [n, out_h, out_w, 3]
  double scale;
  float mean[3], stdv[3];
};

// One workgroup per (needed padded row, frame): the padded row is staged in LDS, then out_w * 3 taps-sums.
__global__ __launch_bounds__(256) void image_hpass_k(const ImgP p) {
  extern __shared__ uint8_t row[];
  const int r = p.row0 + blockIdx.x, img = blockIdx.y;
  const int sy = r - p.off_y;
  const bool in_y = sy >= 0 && sy < p.h;
  const uint8_t* s = p.src + ((size_t)img * p.h + (in_y ? sy : 0)) * p.w * 3;
  const int nb = p.pw * 3;
  for (int e = threadIdx.x; e < nb; e += 256) {
    const int x = e / 3, c = e - 3 * x, sx = x - p.off_x;
    row[e] = (in_y && sx >= 0 && sx < p.w) ? s[sx * 3 + c] : (uint8_t)(p.bg >> (8 * c));
  }
  __syncthreads();
  uint8_t* d = p.tmp + ((size_t)img * p.rows + blockIdx.x) * p.out_w * 3;
  for (int e = threadIdx.x; e < p.out_w * 3; e += 256) {
    const int xo = e / 3, c = e - 3 * xo, x = xo + p.crop_left;
    if (p.hb == nullptr) {
      d[e] = row[x * 3 + c];
      continue;
    }
    const int xmin = p.hb[2 * x], cnt = p.hb[2 * x + 1];
    const int32_t* k = p.hk + (size_t)x * p.hks;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int i = 0; i < cnt; ++i) acc += (int)row[(xmin + i) * 3 + c] * k[i];
    d[e] = clip8(acc);
  }
}

// One workgroup per (output row, frame); threads run over (channel, x) so the planar stores are contiguous.
template <typename TO>
__global__ __launch_bounds__(256) void image_vpass_k(const ImgP p) {
  const int yo = blockIdx.x, img = blockIdx.y, y = yo + p.crop_top;
  const uint8_t* t = p.tmp + (size_t)img * p.rows * p.out_w * 3;
  int ymin = y, cnt = 1;
  const int32_t* k = nullptr;
  if (p.vb) {
    ymin = p.vb[2 * y];
    cnt = p.vb[2 * y + 1];
    k = p.vk + (size_t)y * p.vks;
  }
  for (int e = threadIdx.x; e < 3 * p.out_w; e += 256) {
    const int c = e / p.out_w, x = e - c * p.out_w;
    const uint8_t* col = t + (size_t)(ymin - p.row0) * p.out_w * 3 + x * 3 + c;
    uint8_t v;
    if (k) {
      int acc = 1 << (PRECISION_BITS - 1);
      for (int i = 0; i < cnt; ++i) acc += (int)col[(size_t)i * p.out_w * 3] * k[i];
      v = clip8(acc);
    } else {
      v = col[0];
    }
    if (p.out_u8) p.out_u8[(((size_t)img * p.out_h + yo) * p.out_w + x) * 3 + c] = v;
    // transformers rescale(): uint8 -> float64 * scale -> float32; normalize(): (x - mean) / std in float32
    const float f = (float)((double)v * p.scale);
    const float o = __fdiv_rn(__fsub_rn(f, p.mean[c]), p.stdv[c]);
    stf<TO>(reinterpret_cast<TO*>(p.out) + (((size_t)img * 3 + c) * p.out_h + yo) * p.out_w + x, o);
  }
}

}  // namespace

extern "C" int dxa_resample_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  return ksize_for(in_size, out_size);
}

extern "C" int dxa_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk) {
  DXA_CHECK_ARG(in_size > 0 && out_size > 0, "dxa_resample_coeffs: sizes must be positive");
  DXA_CHECK_ARG(filter == DXA_FILTER_BICUBIC, "dxa_resample_coeffs: only DXA_FILTER_BICUBIC is implemented");
  DXA_CHECK_ARG(bounds && kk, "dxa_resample_coeffs: null table");
  // Resample.c precompute_coeffs (in0 = 0, in1 = in_size), then normalize_coeffs_8bpc
  double scale = (double)in_size / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    int32_t* o = kk + (size_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
      const double v = x < xmax ? k[x] : 0.0;
      o[x] = v < 0 ? (int32_t)(-0.5 + v * (1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return DXA_OK;
}

extern "C" int dxa_image_preprocess(const dxa_image_desc* d, dxa_stream_t stream) {
  DXA_CHECK_ARG(d != nullptr, "dxa_image_preprocess: null desc");
  DXA_CHECK_ARG(d->n >= 0 && d->h > 0 && d->w > 0, "dxa_image_preprocess: bad frame size");
  if (d->n == 0) return DXA_OK;
  DXA_CHECK_ARG(d->src && d->tmp && d->out, "dxa_image_preprocess: null buffer");
  DXA_CHECK_ARG(d->out_dtype == DXA_F32 || d->out_dtype == DXA_BF16, "dxa_image_preprocess: bad out_dtype");
  ImgP p;
  memset(&p, 0, sizeof(p));
  p.src = (const uint8_t*)d->src; p.n = d->n; p.h = d->h; p.w = d->w;
  p.ph = d->h; p.pw = d->w;
  if (d->pad) {   // expand2square: the frame is centred along the short side (integer division as the reference)
    p.ph = p.pw = d->h > d->w ? d->h : d->w;
    p.off_y = (p.ph - d->h) / 2;
    p.off_x = (p.pw - d->w) / 2;
  }
  p.bg = (uint32_t)d->bg[0] | ((uint32_t)d->bg[1] << 8) | ((uint32_t)d->bg[2] << 16);
  DXA_CHECK_ARG(d->res_h > 0 && d->res_w > 0, "dxa_image_preprocess: bad resize target");
  DXA_CHECK_ARG((d->hb != nullptr) == (d->res_w != p.pw) && (d->vb != nullptr) == (d->res_h != p.ph),
                "dxa_image_preprocess: a pass has tables exactly when it changes the size (%dx%d -> %dx%d)", p.ph, p.pw,
                d->res_h, d->res_w);
  DXA_CHECK_ARG(!d->hb || d->hks == ksize_for(p.pw, d->res_w), "dxa_image_preprocess: horizontal table stride");
  DXA_CHECK_ARG(!d->vb || d->vks == ksize_for(p.ph, d->res_h), "dxa_image_preprocess: vertical table stride");
  DXA_CHECK_ARG(d->crop_top >= 0 && d->crop_left >= 0 && d->out_h > 0 && d->out_w > 0 &&
                    d->crop_top + d->out_h <= d->res_h && d->crop_left + d->out_w <= d->res_w,
                "dxa_image_preprocess: crop window outside the resized frame");
  DXA_CHECK_ARG(d->row0 >= 0 && d->rows > 0 && d->row0 + d->rows <= p.ph, "dxa_image_preprocess: bad scratch row range");
  DXA_CHECK_ARG((size_t)p.pw * 3 <= 64 * 1024, "dxa_image_preprocess: frames wider than 21845 px are not supported");
  DXA_CHECK_ARG(d->n <= 65535, "dxa_image_preprocess: too many frames per call");
  p.row0 = d->row0; p.rows = d->rows;
  p.crop_top = d->crop_top; p.crop_left = d->crop_left; p.out_h = d->out_h; p.out_w = d->out_w;
  p.hb = d->hb; p.hk = d->hk; p.hks = d->hks;
  p.vb = d->vb; p.vk = d->vk; p.vks = d->vks;
  p.tmp = (uint8_t*)d->tmp; p.out = d->out; p.out_u8 = (uint8_t*)d->out_u8;
  p.scale = d->rescale;
  for (int c = 0; c < 3; ++c) { p.mean[c] = d->mean[c]; p.stdv[c] = d->std[c]; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(image_hpass_k, dim3(p.rows, p.n), dim3(256), (size_t)p.pw * 3, st, p);
  DXA_CHECK_LAUNCH();
  if (d->out_dtype == DXA_F32) hipLaunchKernelGGL(image_vpass_k<float>, dim3(p.out_h, p.n), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(image_vpass_k<bf16_t>, dim3(p.out_h, p.n), dim3(256), 0, st, p);
  DXA_CHECK_LAUNCH();
  return DXA_OK;
}
