// GPU multi-crop augmentation for gfx950: the DINO view pipeline of the reference
//   RandomResizedCrop(INTER_AREA) -> HorizontalFlip -> ColorJitter -> ToGray -> GaussianBlur -> Solarize -> Normalize -> ToTensor
// (LT/_transforms/view_transform.py:133-215 `ViewTransform`, LT/_methods/dino/dino_transform.py:129-202 `DINOTransform`: 2 global
// views + N local views per image) produced directly in HBM from decoded uint8 HWC images.  The reference runs these ops per image
// in albumentations / OpenCV on CPU dataloader workers; at >1000 img/s/GPU x 10 views that is the bottleneck of the whole pipeline.
// albumentations and cv2 are third-party dependencies that are not vendored in the reference tree: the op DEFINITIONS below are
// restated from their public documentation (oracle/augment_oracle.py holds the same definitions in plain torch; parity unpinned
// against the libraries themselves), the random PARAMETERS are drawn on the host (lightly_train_amd/augment.py).
//
// Three HBM-bound passes per view group (all views of one output size in one launch each):
//   A  crop + area-resample + flip : uint8 HWC source -> f32 planar view in [0,1]          (one thread per output pixel)
//   B  colour jitter + grayscale   : in place; ONE block per view because the contrast op blends with the view-wide mean
//                                    luminance (two sweeps over a view that is L2-resident: 224^2 x 3 floats = 602 KB)
//   C  gaussian blur + solarize + normalize : f32 view -> f32 output [n, 3, S, S]         (separable, LDS tile with halo)
#include "lt_common.h"

namespace {

struct CropItem {       // one (image, view) pair
  long src_off;         // byte offset of the image in the packed uint8 HWC source buffer
  int H, W;             // source image size
  float x0, y0, cw, ch; // crop box in source pixels
  int flip;             // horizontal flip
};

// ---- A: crop + area resampling ("pixel area relation": each output pixel is the mean of the source over its footprint) --------
__global__ __launch_bounds__(256) void crop_resize_area_kernel(const uint8_t* __restrict__ src, const CropItem* __restrict__ items,
                                                               float* __restrict__ out, int S) {
  const CropItem it = items[blockIdx.y];
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S * S) return;
  const int oy = p / S, oxo = p % S;
  const int ox = it.flip ? (S - 1 - oxo) : oxo;
  const float sx = it.cw / (float)S, sy = it.ch / (float)S;
  const float fx0 = it.x0 + ox * sx, fx1 = fx0 + sx, fy0 = it.y0 + oy * sy, fy1 = fy0 + sy;
  const int ix0 = max(0, (int)floorf(fx0)), ix1 = min(it.W - 1, (int)ceilf(fx1) - 1);
  const int iy0 = max(0, (int)floorf(fy0)), iy1 = min(it.H - 1, (int)ceilf(fy1) - 1);
  const uint8_t* img = src + it.src_off;
  float r = 0.f, g = 0.f, b = 0.f, wsum = 0.f;
  for (int iy = iy0; iy <= iy1; ++iy) {
    const float wy = fminf(fy1, (float)(iy + 1)) - fmaxf(fy0, (float)iy);
    if (wy <= 0.f) continue;
    for (int ix = ix0; ix <= ix1; ++ix) {
      const float wx = fminf(fx1, (float)(ix + 1)) - fmaxf(fx0, (float)ix);
      if (wx <= 0.f) continue;
      const float w = wx * wy;
      const uint8_t* px = img + ((long)iy * it.W + ix) * 3;
      r += w * px[0]; g += w * px[1]; b += w * px[2];
      wsum += w;
    }
  }
  const float inv = wsum > 0.f ? 1.f / (wsum * 255.f) : 0.f;
  float* o = out + (long)blockIdx.y * 3 * S * S + p;
  o[0] = r * inv; o[(long)S * S] = g * inv; o[2L * S * S] = b * inv;
}

// ---- B: colour jitter (torchvision / albumentations ColorJitter semantics on floats in [0,1]) + grayscale -----------------------
struct ColorItem {
  int apply;            // ColorJitter fires
  int order;            // permutation code of the four ops: digits base 4, op ids 0 brightness 1 contrast 2 saturation 3 hue
  float fb, fc, fs, fh; // factors: brightness / contrast / saturation in [max(0, 1 - x), 1 + x], hue shift in [-h, h] (turns)
  int gray;             // ToGray fires (after the jitter)
};
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float lum(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }
__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float dh) {
  const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
  const float v = mx, c = mx - mn;
  const float s = mx > 0.f ? c / mx : 0.f;
  float h = 0.f;
  if (c > 0.f) {
    if (mx == r) h = (g - b) / c;
    else if (mx == g) h = 2.f + (b - r) / c;
    else h = 4.f + (r - g) / c;
    h *= (1.f / 6.f);
    h -= floorf(h);
  }
  h += dh;
  h -= floorf(h);
  const float h6 = h * 6.f;
  const int i = (int)floorf(h6) % 6;
  const float f = h6 - floorf(h6);
  const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}
// apply ops [first, last) of the item's order to one pixel; `mean_l` = view-wide mean luminance for the contrast op
__device__ __forceinline__ void jitter_ops(const ColorItem& it, int first, int last, float mean_l, float& r, float& g, float& b) {
  for (int k = first; k < last; ++k) {
    const int op = (it.order >> (2 * k)) & 3;
    if (op == 0) { r = clamp01(r * it.fb); g = clamp01(g * it.fb); b = clamp01(b * it.fb); }
    else if (op == 1) { const float m = (1.f - it.fc) * mean_l; r = clamp01(it.fc * r + m); g = clamp01(it.fc * g + m); b = clamp01(it.fc * b + m); }
    else if (op == 2) { const float l = (1.f - it.fs) * lum(r, g, b); r = clamp01(it.fs * r + l); g = clamp01(it.fs * g + l); b = clamp01(it.fs * b + l); }
    else hue_shift(r, g, b, it.fh);
  }
}
__global__ __launch_bounds__(1024) void color_jitter_kernel(float* __restrict__ views, const ColorItem* __restrict__ items, int S) {
  __shared__ float red[32];
  const ColorItem it = items[blockIdx.x];
  if (!it.apply && !it.gray) return;
  float* v = views + (long)blockIdx.x * 3 * S * S;
  const int n = S * S;
  int cpos = 4;   // position of the contrast op in the order (4 = jitter not applied)
  if (it.apply)
    for (int k = 0; k < 4; ++k)
      if (((it.order >> (2 * k)) & 3) == 1) cpos = k;
  float mean_l = 0.f;
  if (it.apply) {
    // sweep 1: mean luminance of the view as the contrast op will see it (after the ops that precede it)
    float acc = 0.f;
    for (int p = threadIdx.x; p < n; p += 1024) {
      float r = v[p], g = v[n + p], b = v[2 * n + p];
      jitter_ops(it, 0, cpos, 0.f, r, g, b);
      acc += lum(r, g, b);
    }
    mean_l = block_sum(acc, red) / (float)n;
  }
  // sweep 2: all ops (+ grayscale)
  for (int p = threadIdx.x; p < n; p += 1024) {
    float r = v[p], g = v[n + p], b = v[2 * n + p];
    if (it.apply) jitter_ops(it, 0, 4, mean_l, r, g, b);
    if (it.gray) { const float l = lum(r, g, b); r = g = b = l; }
    v[p] = r; v[n + p] = g; v[2 * n + p] = b;
  }
}

// ---- C: gaussian blur (separable, reflect-101 border) + solarize + normalize ----------------------------------------------------
struct FinishItem {
  float sigma;          // 0 = no blur; else radius = ceil(3 sigma) <= LT_AUG_MAX_RADIUS
  int solarize;         // x >= threshold -> 1 - x
  float threshold;
};
constexpr int TB = 32, RMAX = LT_AUG_MAX_RADIUS, TW = TB + 2 * RMAX;
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}
__global__ __launch_bounds__(256) void blur_finish_kernel(const float* __restrict__ views, const FinishItem* __restrict__ items,
                                                          float* __restrict__ out, int S, float m0, float m1, float m2, float s0, float s1,
                                                          float s2) {
  __shared__ float tile[TW][TW + 1];
  __shared__ float hz[TW][TB + 1];
  __shared__ float wgt[2 * RMAX + 1];
  const FinishItem it = items[blockIdx.z / 3];
  const int ch = blockIdx.z % 3;
  const float* v = views + ((long)(blockIdx.z / 3) * 3 + ch) * S * S;
  float* o = out + ((long)(blockIdx.z / 3) * 3 + ch) * S * S;
  const int x0 = blockIdx.x * TB, y0 = blockIdx.y * TB;
  const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2), stdv = ch == 0 ? s0 : (ch == 1 ? s1 : s2);
  const int R = it.sigma > 0.f ? min(RMAX, (int)ceilf(3.f * it.sigma)) : 0;
  if (threadIdx.x <= 2 * R) {
    const float d = (float)((int)threadIdx.x - R);
    wgt[threadIdx.x] = R > 0 ? __expf(-0.5f * d * d / (it.sigma * it.sigma)) : 1.f;
  }
  const int span = TB + 2 * R;
  for (int i = threadIdx.x; i < span * span; i += 256) {
    const int ty = i / span, tx = i % span;
    tile[ty][tx] = v[(long)reflect101(y0 - R + ty, S) * S + reflect101(x0 - R + tx, S)];
  }
  __syncthreads();
  float norm = 0.f;
  for (int k = 0; k <= 2 * R; ++k) norm += wgt[k];
  const float inv = 1.f / norm;
  for (int i = threadIdx.x; i < span * TB; i += 256) {   // horizontal pass
    const int ty = i / TB, tx = i % TB;
    float a = 0.f;
    for (int k = 0; k <= 2 * R; ++k) a += wgt[k] * tile[ty][tx + k];
    hz[ty][tx] = a * inv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TB * TB; i += 256) {     // vertical pass + solarize + normalize
    const int ty = i / TB, tx = i % TB;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= S || x >= S) continue;
    float a = 0.f;
    for (int k = 0; k <= 2 * R; ++k) a += wgt[k] * hz[ty + k][tx];
    a *= inv;
    if (it.solarize && a >= it.threshold) a = 1.f - a;
    o[(long)y * S + x] = (a - mean) / stdv;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

static_assert(sizeof(CropItem) == sizeof(lt_aug_crop_item), "lt_aug_crop_item layout");
static_assert(sizeof(ColorItem) == sizeof(lt_aug_color_item), "lt_aug_color_item layout");
static_assert(sizeof(FinishItem) == sizeof(lt_aug_finish_item), "lt_aug_finish_item layout");

extern "C" int lt_aug_crop_resize(const uint8_t* src, const lt_aug_crop_item* items, float* views, int n, int S, void* stream) {
  LT_CHECK_ARG(src && items && views && n >= 0 && S > 0, "lt_aug_crop_resize: bad arguments");
  if (n == 0) return LT_OK;
  hipLaunchKernelGGL(crop_resize_area_kernel, dim3(lt_cdiv(S * S, 256), n), dim3(256), 0, ST, src, (const CropItem*)items, views, S);
  LT_CHECK_LAUNCH("lt_aug_crop_resize");
}
extern "C" int lt_aug_color(float* views, const lt_aug_color_item* items, int n, int S, void* stream) {
  LT_CHECK_ARG(views && items && n >= 0 && S > 0, "lt_aug_color: bad arguments");
  if (n == 0) return LT_OK;
  hipLaunchKernelGGL(color_jitter_kernel, dim3(n), dim3(1024), 0, ST, views, (const ColorItem*)items, S);
  LT_CHECK_LAUNCH("lt_aug_color");
}
extern "C" int lt_aug_finish(const float* views, const lt_aug_finish_item* items, float* out, int n, int S, const float* mean3, const float* std3,
                             void* stream) {
  LT_CHECK_ARG(views && items && out && mean3 && std3 && n >= 0 && S > 0 && views != out, "lt_aug_finish: bad arguments (out of place)");
  if (n == 0) return LT_OK;
  hipLaunchKernelGGL(blur_finish_kernel, dim3(lt_cdiv(S, TB), lt_cdiv(S, TB), n * 3), dim3(256), 0, ST, views, (const FinishItem*)items, out, S,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  LT_CHECK_LAUNCH("lt_aug_finish");
}
