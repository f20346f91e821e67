// PaKA pieces of the DINOv31 method (LT/_methods/dinov31/dinov31.py:258-437: cross-view patch kernel alignment between clean-teacher
// global crops and high-overlap student local crops).  HBM-bound glue around the MFMA GEMMs the step already has (projection heads,
// per-image token Gram matrices): RoI resampling of token maps with per-image 4-tap tables and its backward, centring of the projected
// tokens over an image's grid, and the centred-kernel-alignment loss with its gradient with respect to the student Gram matrix.
#include "lt_common.h"

namespace {

// out[o, :] = sum_a w[o, a] * in[src_image[b] * img_stride + idx[o, a] * D ...], o = b * n_out + j: the bilinear RoI sampling of one
// image's token map (tables built on the host from the crop geometry, dinov31.py:338-437; flips are folded into the indices).
// `in` points at the first PATCH token of image 0 (cls / register rows are skipped by the caller's offset), images `img_stride` floats apart.
__global__ __launch_bounds__(256) void roi_resample_kernel(const float* __restrict__ in, const int32_t* __restrict__ src_image,
                                                           const int32_t* __restrict__ idx, const float* __restrict__ w,
                                                           bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32, long img_stride,
                                                           int n_out, int D) {
  const long o = blockIdx.x;                 // b * n_out + j
  const long b = o / n_out;
  const float* src = in + (long)(src_image ? src_image[b] : b) * img_stride;
  const int32_t* io = idx + o * 4;
  const float* wo = w + o * 4;
  const float w0 = wo[0], w1 = wo[1], w2 = wo[2], w3 = wo[3];
  const float* p0 = src + (long)io[0] * D;
  const float* p1 = src + (long)io[1] * D;
  const float* p2 = src + (long)io[2] * D;
  const float* p3 = src + (long)io[3] * D;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    const float4 a = *reinterpret_cast<const float4*>(p0 + d), c = *reinterpret_cast<const float4*>(p1 + d);
    const float4 e = *reinterpret_cast<const float4*>(p2 + d), f = *reinterpret_cast<const float4*>(p3 + d);
    const float4 r = make_float4(w0 * a.x + w1 * c.x + w2 * e.x + w3 * f.x, w0 * a.y + w1 * c.y + w2 * e.y + w3 * f.y,
                                 w0 * a.z + w1 * c.z + w2 * e.z + w3 * f.z, w0 * a.w + w1 * c.w + w2 * e.w + w3 * f.w);
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + o * D + d) = r;
    if (out_bf16) *reinterpret_cast<uint2*>(out_bf16 + o * D + d) = make_uint2(pack_bf2(r.x, r.y), pack_bf2(r.z, r.w));
  }
}

// backward, gather form (no atomics: order-fixed): din[b, i, :] = sum over the (j, a) of image b with idx[b, j, a] == i of w * dout[b, j, :].
// One workgroup per (input cell i, image b); the image's table (n_out * 4 entries) is scanned from LDS.
__global__ __launch_bounds__(256) void roi_resample_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                               const float* __restrict__ w, float* __restrict__ din, long img_stride,
                                                               int n_in, int n_out, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int32_t* s_idx = reinterpret_cast<int32_t*>(smem_raw);
  float* s_w = reinterpret_cast<float*>(smem_raw + (size_t)n_out * 4 * sizeof(int32_t));
  const int i = blockIdx.x;
  const long b = blockIdx.y;
  for (int t = threadIdx.x; t < n_out * 4; t += 256) { s_idx[t] = idx[b * n_out * 4 + t]; s_w[t] = w[b * n_out * 4 + t]; }
  __syncthreads();
  float* dst = din + b * img_stride + (long)i * D;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < n_out * 4; ++t) {
      if (s_idx[t] != i) continue;           // (wave-uniform: the table is shared by the block)
      const float wa = s_w[t];
      if (wa == 0.f) continue;
      const float4 v = *reinterpret_cast<const float4*>(dout + (b * n_out + (t >> 2)) * D + d);
      acc.x = fmaf(wa, v.x, acc.x); acc.y = fmaf(wa, v.y, acc.y); acc.z = fmaf(wa, v.z, acc.z); acc.w = fmaf(wa, v.w, acc.w);
    }
    *reinterpret_cast<float4*>(dst + d) = acc;
  }
}

// zc[b, j, :] = z[b, j, :] - mean_j z[b, :, :]   (H Z: centring the token kernel = centring the features over the image's tokens)
// grid (C / 256 column groups, B); n tokens are walked by the block's threads column-wise.
__global__ __launch_bounds__(256) void center_tokens_kernel(const float* __restrict__ z, bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32,
                                                            int n, int C) {
  const long b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float* zb = z + b * n * C;
  float s = 0.f;
  for (int j = 0; j < n; ++j) s += zb[(long)j * C + c];
  const float mean = s / (float)n;
  for (int j = 0; j < n; ++j) {
    const float v = zb[(long)j * C + c] - mean;
    if (out_f32) out_f32[(b * n + j) * C + c] = v;
    if (out_bf16) out_bf16[(b * n + j) * C + c] = f2bf(v);
  }
}

// Per image: hst = <Ks, Kt>, hss = <Ks, Ks>, htt = <Kt, Kt> over the n x n Gram matrices (rows `ld` floats apart), cka = hst / (sqrt(hss)
// sqrt(htt) + eps), term[b] = coef[b] * (1 - cka), G = d term / d Ks = -coef * (Kt / den - hst * sqrt(htt) * Ks / (sqrt(hss) * den^2)),
// den = sqrt(hss) sqrt(htt) + eps; pad columns of G are zeroed.  coef[b] = 0 marks an image without overlap (its G is zero).
__global__ __launch_bounds__(256) void cka_kernel(const float* __restrict__ Ks, const float* __restrict__ Kt, const float* __restrict__ coef,
                                                  float* __restrict__ terms, bf16_t* __restrict__ G, int n, int ld, float eps) {
  __shared__ float red[32];
  const long b = blockIdx.x;
  const float* ks = Ks + b * n * ld;
  const float* kt = Kt + b * n * ld;
  float hst = 0.f, hss = 0.f, htt = 0.f;
  for (int t = threadIdx.x; t < n * n; t += 256) {
    const int i = t / n, j = t - i * n;
    const float a = ks[(long)i * ld + j], c = kt[(long)i * ld + j];
    hst = fmaf(a, c, hst); hss = fmaf(a, a, hss); htt = fmaf(c, c, htt);
  }
  hst = block_sum(hst, red);
  hss = block_sum(hss, red);
  htt = block_sum(htt, red);
  const float ns = sqrtf(hss), nt = sqrtf(htt);
  const float den = ns * nt + eps;
  const float cf = coef[b];
  if (threadIdx.x == 0) terms[b] = cf * (1.f - hst / den);
  if (G) {
    const float g_t = -cf / den;                                           // multiplies Kt
    const float g_s = (ns > 0.f) ? cf * hst * nt / (ns * den * den) : 0.f;  // multiplies Ks
    bf16_t* g = G + b * n * ld;
    for (int t = threadIdx.x; t < n * ld; t += 256) {
      const int i = t / ld, j = t - i * ld;
      g[t] = (j < n) ? f2bf(g_t * kt[(long)i * ld + j] + g_s * ks[(long)i * ld + j]) : (bf16_t)0;
    }
  }
}

__global__ __launch_bounds__(256) void terms_sum_kernel(const float* __restrict__ terms, int rows, float* __restrict__ loss) {
  __shared__ float red[32];
  float a = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) a += terms[r];
  a = block_sum(a, red);
  if (threadIdx.x == 0) loss[0] += a;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int lt_roi_resample_tokens(const float* in, const int32_t* src_image, const int32_t* idx, const float* w, void* out_bf16,
                                      float* out_f32, int B, int64_t img_stride, int n_out, int D, void* stream) {
  LT_CHECK_ARG(in && idx && w && (out_bf16 || out_f32) && B >= 0 && n_out > 0 && D > 0 && D % 4 == 0 && img_stride % 4 == 0,
               "lt_roi_resample_tokens: bad arguments (D and the image stride must be multiples of 4)");
  if (B == 0) return LT_OK;
  hipLaunchKernelGGL(roi_resample_kernel, dim3((unsigned)((long)B * n_out)), dim3(256), 0, ST, in, src_image, idx, w, (bf16_t*)out_bf16, out_f32,
                     (long)img_stride, n_out, D);
  LT_CHECK_LAUNCH("lt_roi_resample_tokens");
}

extern "C" int lt_roi_resample_tokens_bwd(const float* dout, const int32_t* idx, const float* w, float* din, int B, int64_t img_stride,
                                          int n_in, int n_out, int D, void* stream) {
  LT_CHECK_ARG(dout && idx && w && din && B >= 0 && n_in > 0 && n_out > 0 && D > 0 && D % 4 == 0 && img_stride % 4 == 0 && n_out <= 2048,
               "lt_roi_resample_tokens_bwd: bad arguments");
  if (B == 0) return LT_OK;
  hipLaunchKernelGGL(roi_resample_bwd_kernel, dim3(n_in, B), dim3(256), (size_t)n_out * 4 * (sizeof(int32_t) + sizeof(float)), ST, dout, idx, w, din,
                     (long)img_stride, n_in, n_out, D);
  LT_CHECK_LAUNCH("lt_roi_resample_tokens_bwd");
}

extern "C" int lt_center_tokens(const float* z, void* out_bf16, float* out_f32, int B, int n, int C, void* stream) {
  LT_CHECK_ARG(z && (out_bf16 || out_f32) && B >= 0 && n > 0 && C > 0, "lt_center_tokens: bad arguments");
  if (B == 0) return LT_OK;
  hipLaunchKernelGGL(center_tokens_kernel, dim3(lt_cdiv(C, 256), B), dim3(256), 0, ST, z, (bf16_t*)out_bf16, out_f32, n, C);
  LT_CHECK_LAUNCH("lt_center_tokens");
}

extern "C" int lt_cka_fwd_bwd(const float* Ks, const float* Kt, const float* coef, float* loss, void* G_bf16, int B, int n, int ld, float eps,
                              void* stream) {
  LT_CHECK_ARG(Ks && Kt && coef && loss && B >= 0 && n > 0 && ld >= n, "lt_cka_fwd_bwd: bad arguments");
  if (B == 0) return LT_OK;
  float* terms = lt_scratch_ring((size_t)B);
  if (!terms) { lt_set_error("lt_cka_fwd_bwd: scratch allocation failed"); return LT_ERR_HIP; }
  hipLaunchKernelGGL(cka_kernel, dim3(B), dim3(256), 0, ST, Ks, Kt, coef, terms, (bf16_t*)G_bf16, n, ld, eps);
  hipLaunchKernelGGL(terms_sum_kernel, dim3(1), dim3(256), 0, ST, terms, B, loss);
  LT_CHECK_LAUNCH("lt_cka_fwd_bwd");
}
