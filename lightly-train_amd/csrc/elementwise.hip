// HBM-bound kernels of the ViT token path: im2col, token assembly, LayerNorm fwd/bwd, LayerScale bwd,
// column sums, row gather/scatter, casts, L2-normalise, weight-norm.  All are coalesced 16-byte
// (4 x f32 / 8 x bf16) streams, one wave (64 lanes) per row where a row reduction is needed.
#include <algorithm>

#include "lt_common.h"
#include "reduce_ledger.h"

namespace {

constexpr int MAXV = 8;  // float4 per lane cached in registers: rows up to 2048 wide

// ------------------------------------------------------------------------------------ im2col
__global__ void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ cols, int B, int C, int H, int W, int p,
                              int kpad) {
  const int gh = H / p, gw = W / p;
  const long row = blockIdx.x;  // b*gh*gw + gy*gw + gx
  const int b = row / (gh * gw), rem = row % (gh * gw), gy = rem / gw, gx = rem % gw;
  const int kreal = C * p * p;
  for (int k = threadIdx.x; k < kpad; k += blockDim.x) {
    float v = 0.f;
    if (k < kreal) {
      const int c = k / (p * p), r = k % (p * p), py = r / p, px = r % p;
      v = img[(((long)b * C + c) * H + gy * p + py) * W + gx * p + px];
    }
    cols[row * kpad + k] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------ bicubic pad-resize (patch_embed.py:90-99)
// separable 4-tap resize with host-precomputed taps: out[p,y,x] = sum_a sum_b wy[y,a]*wx[x,b]*in[p, iy[y,a], ix[x,b]]
// grid (ceil(Ho*Wo / 256), planes): 32-bit index math only, the taps of an output pixel are read once and shared by its 16 gathers
__global__ __launch_bounds__(256) void resize4tap_kernel(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ iy,
                                                         const float* __restrict__ wy, const int32_t* __restrict__ ix,
                                                         const float* __restrict__ wx, int planes, int H, int W, int Ho, int Wo) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= Ho * Wo) return;
  const int y = o / Wo, x = o - y * Wo;
  const int4 jx = *reinterpret_cast<const int4*>(ix + x * 4), jy = *reinterpret_cast<const int4*>(iy + y * 4);
  const float4 cx = *reinterpret_cast<const float4*>(wx + x * 4), cy = *reinterpret_cast<const float4*>(wy + y * 4);
  const int jya[4] = {jy.x, jy.y, jy.z, jy.w};
  const float cya[4] = {cy.x, cy.y, cy.z, cy.w};
  for (int p = blockIdx.y; p < planes; p += gridDim.y) {
    const float* src = in + (size_t)p * H * W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float* row = src + jya[a] * W;
      float r = 0.f;
      r = fmaf(cx.x, row[jx.x], r); r = fmaf(cx.y, row[jx.y], r); r = fmaf(cx.z, row[jx.z], r); r = fmaf(cx.w, row[jx.w], r);
      acc = fmaf(cya[a], r, acc);
    }
    out[(size_t)p * Ho * Wo + o] = acc;
  }
}

// Linear resampling of token grids with a sparse tap table: out[b, o, :] = sum_a w[o, a] * in[b, idx[o, a], :]
// (the bilinear F.interpolate of the student's spatial features onto the teacher's grid, distillationv3.py:338-345; its
// backward is the same kernel with the transposed table).  grid (n_out, B).
__global__ __launch_bounds__(256) void resample_tokens_kernel(const float* __restrict__ in, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ w, float* __restrict__ out, int n_in, int n_out,
                                                              int D, int taps) {
  const int o = blockIdx.x;
  const long b = blockIdx.y;
  const float* src = in + b * n_in * D;
  float* dst = out + (b * n_out + o) * D;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < taps; ++a) {
      const float wa = w[o * taps + a];
      if (wa == 0.f) continue;
      const float4 v = *reinterpret_cast<const float4*>(src + (long)idx[o * taps + a] * D + d);
      acc.x = fmaf(wa, v.x, acc.x); acc.y = fmaf(wa, v.y, acc.y); acc.z = fmaf(wa, v.z, acc.z); acc.w = fmaf(wa, v.w, acc.w);
    }
    *reinterpret_cast<float4*>(dst + d) = acc;
  }
}
extern "C" int lt_resample_tokens(const float* in, const int32_t* idx, const float* w, float* out, int B, int n_in, int n_out, int D,
                                  int taps, void* stream) {
  LT_CHECK_ARG(in && idx && w && out && in != out && B > 0 && n_in > 0 && n_out > 0 && D > 0 && D % 4 == 0 && taps > 0,
               "lt_resample_tokens: bad arguments (D must be a multiple of 4)");
  hipLaunchKernelGGL(resample_tokens_kernel, dim3(n_out, B), dim3(256), 0, (hipStream_t)stream, in, idx, w, out, n_in, n_out, D, taps);
  LT_CHECK_LAUNCH("lt_resample_tokens");
}

// ------------------------------------------------------------------------------------ tokens
__global__ void assemble_kernel(const float* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                const float* __restrict__ mask_token, const uint8_t* __restrict__ masks, const float* __restrict__ reg,
                                float* __restrict__ x, int B, int n_p, int n_reg, int D) {
  const long tok = blockIdx.x;  // b*N + t,  token order [cls | registers | patches]
  const int N = n_p + 1 + n_reg;
  const int b = tok / N, t = tok % N;
  const float* src;
  const float* pe = nullptr;    // registers get no positional embedding (vision_transformer.py:318-327)
  if (t == 0) { src = cls; pe = pos; }
  else if (t <= n_reg) src = reg + (long)(t - 1) * D;
  else {
    const int i = t - 1 - n_reg;
    src = (masks && masks[(long)b * n_p + i]) ? mask_token : patch + ((long)b * n_p + i) * D;
    pe = pos + (long)(1 + i) * D;
  }
  for (int d = threadIdx.x; d < D; d += blockDim.x) x[tok * D + d] = src[d] + (pe ? pe[d] : 0.f);
}

// block = 64 lanes x V columns x 4 batch-lanes for one token position t; the batch reduction goes through LDS.
// V = 4 (D % 4 == 0): 16-byte loads, 8-byte bf16 stores (the scalar form ran at 1.5 TB/s).
template <int V>
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const float* __restrict__ dx, const uint8_t* __restrict__ masks,
                                                           bf16_t* __restrict__ dpatch, float* __restrict__ dcls, float* __restrict__ dpos,
                                                           float* __restrict__ dmask, float* __restrict__ dreg, int B, int n_p, int n_reg,
                                                           int D, float* __restrict__ pmask) {
  __shared__ float red[2][4][64 * V];
  const int N = n_p + 1 + n_reg;
  const int t = blockIdx.x;
  const int i = t - 1 - n_reg;   // patch index (>= 0 for patch tokens)
  const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int d = (blockIdx.y * 64 + cl) * V;
  float sum[V], msum[V];
#pragma unroll
  for (int v = 0; v < V; ++v) { sum[v] = 0.f; msum[v] = 0.f; }
  if (d < D) {
    // four images per trip with their loads (token rows N * D floats apart) issued together: one load in flight per thread made the
    // kernel latency-bound (64 dependent trips for batch 256: 153 us for 235 MB, 1.5 TB/s)
    constexpr int U = 4;
    for (int b0 = bl; b0 < B; b0 += 4 * U) {
      float g[U][V];
      bool mk[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int b = b0 + 4 * u;
#pragma unroll
        for (int v = 0; v < V; ++v) g[u][v] = 0.f;
        mk[u] = false;
        if (b < B) {
          if (V == 4) {
            const float4 g4 = *reinterpret_cast<const float4*>(dx + ((long)b * N + t) * D + d);
            g[u][0] = g4.x; g[u][1 % V] = g4.y; g[u][2 % V] = g4.z; g[u][3 % V] = g4.w;
          } else {
            g[u][0] = dx[((long)b * N + t) * D + d];
          }
          if (i >= 0) mk[u] = masks && masks[(long)b * n_p + i];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {   // (accumulation order: ascending image index within a batch lane, as before)
        const int b = b0 + 4 * u;
        if (b >= B) continue;
#pragma unroll
        for (int v = 0; v < V; ++v) sum[v] += g[u][v];
        if (i >= 0) {
          if (mk[u]) {
#pragma unroll
            for (int v = 0; v < V; ++v) msum[v] += g[u][v];
          }
          bf16_t* o = dpatch + ((long)b * n_p + i) * D + d;
          if (V == 4) *reinterpret_cast<uint2*>(o) = mk[u] ? make_uint2(0, 0) : make_uint2(pack_bf2(g[u][0], g[u][1 % V]), pack_bf2(g[u][2 % V], g[u][3 % V]));
          else o[0] = mk[u] ? (bf16_t)0 : f2bf(g[u][0]);
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) { red[0][bl][cl * V + v] = sum[v]; red[1][bl][cl * V + v] = msum[v]; }
  __syncthreads();
  if (bl == 0 && d < D) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int k = cl * V + v;
      const float s2 = red[0][0][k] + red[0][1][k] + red[0][2][k] + red[0][3][k];
      const float m2 = red[1][0][k] + red[1][1][k] + red[1][2][k] + red[1][3][k];
      if (t == 0) { dcls[d + v] += s2; dpos[d + v] += s2; }
      else if (i < 0) dreg[(long)(t - 1) * D + d + v] += s2;
      else {
        dpos[(long)(1 + i) * D + d + v] += s2;
        if (pmask) pmask[(size_t)i * D + d + v] = m2;   // [n_p][D] partial rows (reduction ledger)
        else if (masks && m2 != 0.f) atomicAdd(&dmask[d + v], m2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ RoPE (DINOv3)
// In place on the packed qkv activation [B, N, 3, H, dh] (bf16): q and k of the tokens >= prefix (cls / storage tokens are not
// rotated) become x*cos + rotate_half(x)*sin with rotate_half([x1 | x2]) = [-x2 | x1]  (dinov3 layers/attention.py:23-34,79-103).
// sin / cos: f32 [N - prefix, dh].  inverse = 1 applies the transposed rotation (backward of the same op).
// One thread per (token, q|k, head, 4-wide group of the first half).
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ sin_t, const float* __restrict__ cos_t,
                                                   long total, int N, int H, int dh, int prefix, int inverse) {
  const int hq = dh >> 3;                 // groups of 4 in the first half
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int g = (int)(idx % hq);
  long r = idx / hq;
  const int h = (int)(r % H); r /= H;
  const int which = (int)(r % 2); r /= 2;   // 0 = q, 1 = k
  const int n = (int)(r % N);
  const long b = r / N;
  if (n < prefix) return;
  const int half = dh >> 1, d = g * 4;
  bf16_t* p = qkv + (((b * N + n) * 3 + which) * H + h) * (long)dh;
  const float* sn = sin_t + (long)(n - prefix) * dh;
  const float* cs = cos_t + (long)(n - prefix) * dh;
  const uint2 u1 = *reinterpret_cast<const uint2*>(p + d);
  const uint2 u2 = *reinterpret_cast<const uint2*>(p + half + d);
  const float x1[4] = {bf2f((bf16_t)(u1.x & 0xffff)), bf2f((bf16_t)(u1.x >> 16)), bf2f((bf16_t)(u1.y & 0xffff)), bf2f((bf16_t)(u1.y >> 16))};
  const float x2[4] = {bf2f((bf16_t)(u2.x & 0xffff)), bf2f((bf16_t)(u2.x >> 16)), bf2f((bf16_t)(u2.y & 0xffff)), bf2f((bf16_t)(u2.y >> 16))};
  float y1[4], y2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float c1 = cs[d + j], s1 = sn[d + j], c2 = cs[half + d + j], s2 = sn[half + d + j];
    if (!inverse) { y1[j] = x1[j] * c1 - x2[j] * s1; y2[j] = x2[j] * c2 + x1[j] * s2; }
    else { y1[j] = x1[j] * c1 + x2[j] * s2; y2[j] = x2[j] * c2 - x1[j] * s1; }
  }
  *reinterpret_cast<uint2*>(p + d) = make_uint2(pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3]));
  *reinterpret_cast<uint2*>(p + half + d) = make_uint2(pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3]));
}
extern "C" int lt_rope_apply(void* qkv_bf16, const float* sin_t, const float* cos_t, int B, int N, int H, int dh, int prefix,
                             int inverse, void* stream) {
  LT_CHECK_ARG(qkv_bf16 && sin_t && cos_t && B > 0 && N > 0 && H > 0 && dh > 0 && dh % 8 == 0 && prefix >= 0 && prefix <= N,
               "lt_rope_apply: bad arguments (head_dim must be a multiple of 8)");
  const long total = (long)B * N * 2 * H * (dh >> 3);
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv_bf16, sin_t, cos_t,
                     total, N, H, dh, prefix, inverse);
  LT_CHECK_LAUNCH("lt_rope_apply");
}

// ------------------------------------------------------------------------------------ SwiGLU
// x12 [rows, 2H] bf16 = [x1 | x2];  hidden = silu(x1) * x2   (swiglu_ffn.py:31-35).  8 columns per thread.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ x12, bf16_t* __restrict__ out, long rows, int H) {
  const int hv = H >> 3;
  const long n = rows * hv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / hv;
    const int c = (int)(i - r * hv) << 3;
    const uint4 a = *reinterpret_cast<const uint4*>(x12 + r * 2 * H + c);
    const uint4 b = *reinterpret_cast<const uint4*>(x12 + r * 2 * H + H + c);
    const bf16_t* pa = reinterpret_cast<const bf16_t*>(&a);
    const bf16_t* pb = reinterpret_cast<const bf16_t*>(&b);
    uint4 o;
    bf16_t* po = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x1 = bf2f(pa[j]), x2 = bf2f(pb[j]);
      po[j] = f2bf(x1 * sigmoid_f(x1) * x2);
    }
    *reinterpret_cast<uint4*>(out + r * H + c) = o;
  }
}

// nn.GELU() (erf form) as its own pass: the BatchNorm projection heads put a normalisation between the Linear and the activation, so
// the GEMM epilogue cannot carry it.  bwd: dx = dy * gelu'(x) with x the saved pre-activation.
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long nvec) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const uint4 a = *reinterpret_cast<const uint4*>(x + i * 8);
    const bf16_t* pa = reinterpret_cast<const bf16_t*>(&a);
    uint4 o;
    bf16_t* po = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
    for (int j = 0; j < 8; ++j) po[j] = f2bf(gelu_f(bf2f(pa[j])));
    *reinterpret_cast<uint4*>(y + i * 8) = o;
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx, long nvec) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const uint4 a = *reinterpret_cast<const uint4*>(x + i * 8);
    const uint4 g = *reinterpret_cast<const uint4*>(dy + i * 8);
    const bf16_t* pa = reinterpret_cast<const bf16_t*>(&a);
    const bf16_t* pg = reinterpret_cast<const bf16_t*>(&g);
    uint4 o;
    bf16_t* po = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
    for (int j = 0; j < 8; ++j) po[j] = f2bf(bf2f(pg[j]) * gelu_grad_f(bf2f(pa[j])));
    *reinterpret_cast<uint4*>(dx + i * 8) = o;
  }
}

// d12 = [dh * x2 * silu'(x1) | dh * silu(x1)],  silu'(x) = s(x) * (1 + x * (1 - s(x)))
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ x12, const bf16_t* __restrict__ dh,
                                                         bf16_t* __restrict__ d12, long rows, int H) {
  const int hv = H >> 3;
  const long n = rows * hv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / hv;
    const int c = (int)(i - r * hv) << 3;
    const uint4 a = *reinterpret_cast<const uint4*>(x12 + r * 2 * H + c);
    const uint4 b = *reinterpret_cast<const uint4*>(x12 + r * 2 * H + H + c);
    const uint4 g = *reinterpret_cast<const uint4*>(dh + r * H + c);
    const bf16_t* pa = reinterpret_cast<const bf16_t*>(&a);
    const bf16_t* pb = reinterpret_cast<const bf16_t*>(&b);
    const bf16_t* pg = reinterpret_cast<const bf16_t*>(&g);
    uint4 o1, o2;
    bf16_t* p1 = reinterpret_cast<bf16_t*>(&o1);
    bf16_t* p2 = reinterpret_cast<bf16_t*>(&o2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x1 = bf2f(pa[j]), x2 = bf2f(pb[j]), gg = bf2f(pg[j]);
      const float sg = sigmoid_f(x1);
      p1[j] = f2bf(gg * x2 * sg * (1.f + x1 * (1.f - sg)));
      p2[j] = f2bf(gg * x1 * sg);
    }
    *reinterpret_cast<uint4*>(d12 + r * 2 * H + c) = o1;
    *reinterpret_cast<uint4*>(d12 + r * 2 * H + H + c) = o2;
  }
}

extern "C" int lt_gelu_fwd_bf16(const void* x, void* y, int64_t n, void* stream) {
  LT_CHECK_ARG(x && y && n >= 0 && n % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0,
               "lt_gelu_fwd_bf16: bad arguments (n must be a multiple of 8, tensors 16-byte aligned)");
  if (n == 0) return LT_OK;
  const long nv = n >> 3;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)min((long)8192, (nv + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y,
                     nv);
  LT_CHECK_LAUNCH("lt_gelu_fwd_bf16");
}

extern "C" int lt_gelu_bwd_bf16(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
  LT_CHECK_ARG(dy && x && dx && n >= 0 && n % 8 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dx & 15) == 0,
               "lt_gelu_bwd_bf16: bad arguments (n must be a multiple of 8, tensors 16-byte aligned)");
  if (n == 0) return LT_OK;
  const long nv = n >> 3;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)min((long)8192, (nv + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)x, (bf16_t*)dx, nv);
  LT_CHECK_LAUNCH("lt_gelu_bwd_bf16");
}

extern "C" int lt_swiglu_fwd(const void* x12_bf16, void* out_bf16, int64_t rows, int H, void* stream) {
  LT_CHECK_ARG(x12_bf16 && out_bf16 && rows >= 0 && H > 0 && H % 8 == 0, "lt_swiglu_fwd: bad arguments (H must be a multiple of 8)");
  if (rows == 0) return LT_OK;
  const long n = rows * (H >> 3);
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3((unsigned)min((long)8192, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x12_bf16,
                     (bf16_t*)out_bf16, rows, H);
  LT_CHECK_LAUNCH("lt_swiglu_fwd");
}

extern "C" int lt_swiglu_bwd(const void* x12_bf16, const void* dh_bf16, void* d12_bf16, int64_t rows, int H, void* stream) {
  LT_CHECK_ARG(x12_bf16 && dh_bf16 && d12_bf16 && rows >= 0 && H > 0 && H % 8 == 0, "lt_swiglu_bwd: bad arguments (H must be a multiple of 8)");
  if (rows == 0) return LT_OK;
  const long n = rows * (H >> 3);
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((unsigned)min((long)8192, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x12_bf16,
                     (const bf16_t*)dh_bf16, (bf16_t*)d12_bf16, rows, H);
  LT_CHECK_LAUNCH("lt_swiglu_bwd");
}

// ------------------------------------------------------------------------------------ LayerNorm
__device__ __forceinline__ void load_row(float4 (&v)[MAXV], const float* __restrict__ p, int D, int lane) {
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = (c < D) ? *reinterpret_cast<const float4*>(p + c) : make_float4(0, 0, 0, 0);
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, bf16_t* __restrict__ yb,
                                                            float* __restrict__ yf, float* __restrict__ mean_o,
                                                            float* __restrict__ rstd_o, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * D;
  if (VEC) {
    float4 v[MAXV];
    load_row(v, xr, D, lane);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
        q += a * a + bb * bb + cc * cc + dd * dd;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) {
      if (mean_o) mean_o[row] = mean;
      if (rstd_o) rstd_o[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        const float4 ww = *reinterpret_cast<const float4*>(w + c);
        const float4 bb = *reinterpret_cast<const float4*>(b + c);
        float4 o;
        o.x = (v[i].x - mean) * rstd * ww.x + bb.x;
        o.y = (v[i].y - mean) * rstd * ww.y + bb.y;
        o.z = (v[i].z - mean) * rstd * ww.z + bb.z;
        o.w = (v[i].w - mean) * rstd * ww.w + bb.w;
        if (yf) *reinterpret_cast<float4*>(yf + row * D + c) = o;
        if (yb) *reinterpret_cast<uint2*>(yb + row * D + c) = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
      }
    }
  } else {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float a = xr[c] - mean; q += a * a; }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) {
      if (mean_o) mean_o[row] = mean;
      if (rstd_o) rstd_o[row] = rstd;
    }
    for (int c = lane; c < D; c += 64) {
      const float o = (xr[c] - mean) * rstd * w[c] + b[c];
      if (yf) yf[row * D + c] = o;
      if (yb) yb[row * D + c] = f2bf(o);
    }
  }
}

// R rows per wave, every load of all R rows (and of the affine parameters) issued before the first reduction: a wave that owns one row
// holds 3 KB in flight and then idles through two dependent wave reductions; with R rows the reductions of one row run under the
// loads of the others.  Rows past the end are clamped to the last row and not stored (branch-free loads keep the waitcnt counted).
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ b, bf16_t* __restrict__ yb,
                                                                 float* __restrict__ yf, float* __restrict__ mean_o,
                                                                 float* __restrict__ rstd_o, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float4 v[R][NV], ww[NV], bb[NV];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long row = row0 + r < rows ? row0 + r : (long)rows - 1;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      v[r][i] = *reinterpret_cast<const float4*>(x + row * D + (c < D ? c : 0));
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    ww[i] = *reinterpret_cast<const float4*>(w + (c < D ? c : 0));
    bb[i] = *reinterpret_cast<const float4*>(b + (c < D ? c : 0));
  }
  const float invD = 1.f / (float)D;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if ((i * 64 + lane) * 4 < D) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if ((i * 64 + lane) * 4 < D) {
        const float a0 = v[r][i].x - mean, a1 = v[r][i].y - mean, a2 = v[r][i].z - mean, a3 = v[r][i].w - mean;
        q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invD + eps);
    const long row = row0 + r;
    if (row < rows) {
      if (lane == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
          float4 o;
          o.x = (v[r][i].x - mean) * rstd * ww[i].x + bb[i].x;
          o.y = (v[r][i].y - mean) * rstd * ww[i].y + bb[i].y;
          o.z = (v[r][i].z - mean) * rstd * ww[i].z + bb[i].z;
          o.w = (v[r][i].w - mean) * rstd * ww[i].w + bb[i].w;
          if (yf) *reinterpret_cast<float4*>(yf + row * D + c) = o;
          if (yb) *reinterpret_cast<uint2*>(yb + row * D + c) = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
        }
      }
    }
  }
}

// Vector backward for D % 4 == 0, D <= NV*256: lane owns float4 column groups c = (i*64 + lane)*4.  Each wave walks its rows
// two at a time (both rows' loads are issued before either row's reductions, hiding the HBM latency a single dependent
// row chain would expose); per-lane dw/db partials stay in registers, are reduced over the block's 4 waves in LDS, then
// one atomicAdd per column per block.
// One row of LayerNorm-backward operands held by a wave (lane owns NV float4 column groups)
template <bool DYF32, int NV>
struct LnbRow {
  float4 xv[NV], rv[NV];
  float4 dvf[DYF32 ? NV : 1];
  uint2 dvh[DYF32 ? 1 : NV];
  float mu, rs, rsn;   // rsn: row scale of the fused next-branch output (loaded with the row, not at its use)
  long rr;             // IDX: the row of dres / dx this row's gradient is added to (ridx[row]); otherwise the row itself
  bool ok;
};

template <bool DYF32, int NV, bool IDX = false>
__device__ __forceinline__ void lnb_load(LnbRow<DYF32, NV>& R, long row, int rows, int D, int lane, const float* __restrict__ x,
                                         const float* __restrict__ mean, const float* __restrict__ rstd, const void* __restrict__ dyv,
                                         const float* __restrict__ dres, const float* __restrict__ rowscale_next,
                                         const int64_t* __restrict__ ridx = nullptr) {
  // branch-free on purpose: rows / columns out of range load from a clamped (valid) address and are ignored by lnb_compute.
  // With one basic block per guarded load the waitcnt pass cannot count, and the wait for THIS row's operands became
  // vmcnt(0) -- i.e. it also waited for the next row's loads issued just before, which is the whole point of the pipeline.
  R.ok = row < rows;
  const long r = R.ok ? row : (long)rows - 1;
  R.rr = IDX ? (long)ridx[r] : r;
  R.mu = mean[r];
  R.rs = rstd[r];
  R.rsn = rowscale_next ? rowscale_next[r] : 1.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = (i * 64 + lane) * 4;
    const int c = c0 < D ? c0 : 0;
    R.xv[i] = *reinterpret_cast<const float4*>(x + r * D + c);
    R.rv[i] = dres ? *reinterpret_cast<const float4*>(dres + (IDX ? R.rr : r) * D + c) : make_float4(0, 0, 0, 0);
    if (DYF32) R.dvf[i] = *reinterpret_cast<const float4*>((const float*)dyv + r * D + c);
    else R.dvh[i] = *reinterpret_cast<const uint2*>((const bf16_t*)dyv + r * D + c);
  }
}

// optional fused producer of the NEXT branch's upstream gradient: dnext(bf16) = dx * gamma_next * rowscale_next, column sums -> abn
struct LnbNext {
  bf16_t* dnext; const float* rowscale; float scale;
};

template <bool DYF32, int NV, bool IDX = false>
__device__ __forceinline__ void lnb_compute(LnbRow<DYF32, NV>& R, long row, int D, int lane, const float4 (&wv4)[NV], float4 (&aw)[NV],
                                            float4 (&ab)[NV], float* __restrict__ dx, const LnbNext& nx, const float4 (&gn4)[NV],
                                            float4 (&abn)[NV]) {
  if (!R.ok) return;
  float c1 = 0.f, c2 = 0.f;
  float4 gyv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    gyv[i] = make_float4(0, 0, 0, 0);
    if (c < D) {
      float4 d;
      if (DYF32) d = R.dvf[i];
      else {
        const uint2 u = R.dvh[i];
        d = make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16)));
      }
      const float4 xh = make_float4((R.xv[i].x - R.mu) * R.rs, (R.xv[i].y - R.mu) * R.rs, (R.xv[i].z - R.mu) * R.rs, (R.xv[i].w - R.mu) * R.rs);
      const float4 gy = make_float4(d.x * wv4[i].x, d.y * wv4[i].y, d.z * wv4[i].z, d.w * wv4[i].w);
      c1 += gy.x + gy.y + gy.z + gy.w;
      c2 += gy.x * xh.x + gy.y * xh.y + gy.z * xh.z + gy.w * xh.w;
      aw[i].x += d.x * xh.x; aw[i].y += d.y * xh.y; aw[i].z += d.z * xh.z; aw[i].w += d.w * xh.w;
      ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
      R.xv[i] = xh; gyv[i] = gy;
    }
  }
  c1 = wave_sum(c1) / D;
  c2 = wave_sum(c2) / D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const float4 xh = R.xv[i], gy = gyv[i];
      float4 o = make_float4(R.rs * (gy.x - c1 - xh.x * c2), R.rs * (gy.y - c1 - xh.y * c2),
                             R.rs * (gy.z - c1 - xh.z * c2), R.rs * (gy.w - c1 - xh.w * c2));
      const float4 rr = R.rv[i];
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
      *reinterpret_cast<float4*>(dx + (IDX ? R.rr : row) * D + c) = o;
      if (nx.dnext) {
        const float m_r = nx.scale * R.rsn;
        const float4 dn = make_float4(o.x * m_r * gn4[i].x, o.y * m_r * gn4[i].y, o.z * m_r * gn4[i].z, o.w * m_r * gn4[i].w);
        *reinterpret_cast<uint2*>(nx.dnext + row * D + c) = make_uint2(pack_bf2(dn.x, dn.y), pack_bf2(dn.z, dn.w));
        abn[i].x += dn.x; abn[i].y += dn.y; abn[i].z += dn.z; abn[i].w += dn.w;
      }
    }
  }
}

// dx = dres + LN'(dy);  dw/db column sums.  Persistent waves (grid <= 512 blocks) stream rows with a two-deep software
// pipeline: the loads of the NEXT row are issued before the current row is reduced and stored, so a wave always has a row
// (7.5 KiB at D = 768) in flight -- without it every iteration paid one full HBM round trip (3.5 TB/s at 512 blocks).
// IDX: row r of (x, mean, rstd, dy) is row ridx[r] of the gradient stream: dx[ridx[r]] = dres[ridx[r]] + LN'(dy[r]) -- the LayerNorm backward
// of a batch-subset stochastic-depth branch (block.py:118-141) added in place to the rows it belongs to, instead of a compact result
// plus a scatter-add pass.  A separate instantiation: the default kernels' code is unchanged.
template <bool DYF32, int NV, bool IDX = false>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const void* __restrict__ dyv, const float* __restrict__ dres,
                                                                float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                                float* __restrict__ partial, int prows, int rows, int D, LnbNext nx,
                                                                const float* __restrict__ gamma_next, float* __restrict__ dbias_next,
                                                                const int64_t* __restrict__ ridx = nullptr) {
  __shared__ float4 red[2][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float4 aw[NV], ab[NV], wv4[NV], gn4[NV], abn[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    aw[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); abn[i] = make_float4(0, 0, 0, 0);
    const int c = (i * 64 + lane) * 4;
    wv4[i] = c < D ? *reinterpret_cast<const float4*>(w + c) : make_float4(0, 0, 0, 0);
    gn4[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (nx.dnext && gamma_next && c < D) gn4[i] = *reinterpret_cast<const float4*>(gamma_next + c);
  }
  const long stride = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wv;
  LnbRow<DYF32, NV> A, B;
  lnb_load<DYF32, NV, IDX>(A, row, rows, D, lane, x, mean, rstd, dyv, dres, nx.rowscale, ridx);
  while (row < rows) {
    lnb_load<DYF32, NV, IDX>(B, row + stride, rows, D, lane, x, mean, rstd, dyv, dres, nx.rowscale, ridx);
    lnb_compute<DYF32, NV, IDX>(A, row, D, lane, wv4, aw, ab, dx, nx, gn4, abn);
    row += stride;
    if (row >= rows) break;
    lnb_load<DYF32, NV, IDX>(A, row + stride, rows, D, lane, x, mean, rstd, dyv, dres, nx.rowscale, ridx);
    lnb_compute<DYF32, NV, IDX>(B, row, D, lane, wv4, aw, ab, dx, nx, gn4, abn);
    row += stride;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (i * 256 >= D) break;
    __syncthreads();
    red[0][wv][lane] = aw[i];
    red[1][wv][lane] = ab[i];
    __syncthreads();
    if (wv == 0) {
      const int c = (i * 64 + lane) * 4;
      if (c < D) {
        const float4 a0 = red[0][0][lane], a1 = red[0][1][lane], a2 = red[0][2][lane], a3 = red[0][3][lane];
        const float4 b0 = red[1][0][lane], b1 = red[1][1][lane], b2 = red[1][2][lane], b3 = red[1][3][lane];
        const float4 sa = make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z, a0.w + a1.w + a2.w + a3.w);
        const float4 sb = make_float4(b0.x + b1.x + b2.x + b3.x, b0.y + b1.y + b2.y + b3.y, b0.z + b1.z + b2.z + b3.z, b0.w + b1.w + b2.w + b3.w);
        if (partial) {  // per-block partial rows [gridDim.x][prows][D], summed in a fixed order later (deterministic, no atomics)
          *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * prows) * D + c) = sa;
          *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * prows + 1) * D + c) = sb;
        } else {
          atomicAdd(&dw[c], sa.x); atomicAdd(&dw[c + 1], sa.y); atomicAdd(&dw[c + 2], sa.z); atomicAdd(&dw[c + 3], sa.w);
          atomicAdd(&db[c], sb.x); atomicAdd(&db[c + 1], sb.y); atomicAdd(&db[c + 2], sb.z); atomicAdd(&db[c + 3], sb.w);
        }
      }
    }
    if (nx.dnext && dbias_next) {  // column sums of the fused next-branch gradient (bias of the Linear feeding that LayerScale)
      __syncthreads();
      red[0][wv][lane] = abn[i];
      __syncthreads();
      if (wv == 0) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
          const float4 a0 = red[0][0][lane], a1 = red[0][1][lane], a2 = red[0][2][lane], a3 = red[0][3][lane];
          if (partial && prows == 3) {
            *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 3 + 2) * D + c) =
                make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z, a0.w + a1.w + a2.w + a3.w);
          } else {
            atomicAdd(&dbias_next[c], a0.x + a1.x + a2.x + a3.x); atomicAdd(&dbias_next[c + 1], a0.y + a1.y + a2.y + a3.y);
            atomicAdd(&dbias_next[c + 2], a0.z + a1.z + a2.z + a3.z); atomicAdd(&dbias_next[c + 3], a0.w + a1.w + a2.w + a3.w);
          }
        }
      }
    }
  }
}
// dw[c] += sum_b partial[b][0][c]; db[c] += sum_b partial[b][1][c]
__global__ __launch_bounds__(256) void ln_partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                float* __restrict__ db, int nblk, int D) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, which = blockIdx.y;
  float a = 0.f;
  if (c < D)
    for (int b = rl; b < nblk; b += 4) a += partial[((size_t)b * 2 + which) * D + c];
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && c < D) {
    float* o = which ? db : dw;
    o[c] += red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
  }
}

// generic (scalar-column) backward; each wave walks rows with stride, keeps dw/db partials per lane-column
// in registers for up to 32 columns per lane (D <= 2048), reduces over the block's 4 waves in LDS,
// then one atomicAdd per column per block.
template <bool DYF32>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const void* __restrict__ dyv, const float* __restrict__ dres,
                                                            float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                            int rows, int D) {
  constexpr int MAXC = 32;
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float aw[MAXC], ab[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) { aw[i] = 0.f; ab[i] = 0.f; }
  for (long row = (long)blockIdx.x * 4 + wv; row < rows; row += (long)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + row * D;
    float xh[MAXC], gy[MAXC];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      xh[i] = 0.f; gy[i] = 0.f;
      if (c < D) {
        const float dyv_ = DYF32 ? ((const float*)dyv)[row * D + c] : bf2f(((const bf16_t*)dyv)[row * D + c]);
        xh[i] = (xr[c] - mu) * rs;
        gy[i] = dyv_ * w[c];
        c1 += gy[i];
        c2 += gy[i] * xh[i];
        aw[i] += dyv_ * xh[i];
        ab[i] += dyv_;
      }
    }
    c1 = wave_sum(c1) / D;
    c2 = wave_sum(c2) / D;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < D) {
        float o = rs * (gy[i] - c1 - xh[i] * c2);
        if (dres) o += dres[row * D + c];
        dx[row * D + c] = o;
      }
    }
  }
  // cross-wave reduction, one column group (64 columns) at a time
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    if (i * 64 >= D) break;
    __syncthreads();
    red[0][wv][lane] = aw[i];
    red[1][wv][lane] = ab[i];
    __syncthreads();
    if (wv == 0) {
      const int c = i * 64 + lane;
      if (c < D) {
        atomicAdd(&dw[c], red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]);
        atomicAdd(&db[c], red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------ LayerScale bwd
// block = 32 column-chunks (4 columns each = 128 columns) x 8 row-lanes; grid.x over 128-column groups, grid.y row slabs.
// dy(bf16) = dout*gamma; dgamma += sum_rows dout*y; dbias += sum_rows dout*gamma (bias of the Linear feeding LayerScale).
// ridx (optional): row r of the branch reads its upstream gradient at row ridx[r] of `dout` (the rows of a batch-subset branch or of the last
// block's loss rows inside the full gradient stream: no gathered copy in between); y / dy / rowscale stay indexed by r.
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const float* __restrict__ dout, const bf16_t* __restrict__ y,
                                                             const float* __restrict__ gamma, bf16_t* __restrict__ dy,
                                                             float* __restrict__ dgamma, float* __restrict__ dbias,
                                                             const float* __restrict__ rowscale, float scale, int rows, int D,
                                                             float* __restrict__ partial, const int64_t* __restrict__ ridx) {
  __shared__ float4 red[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cl) * 4;
  float4 ag = make_float4(0, 0, 0, 0), abias = make_float4(0, 0, 0, 0);
  if (c < D) {
    const float4 gm = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    for (long r = (long)blockIdx.y * 8 + rl; r < rows; r += (long)gridDim.y * 8) {
      const float m_r = scale * (rowscale ? rowscale[r] : 1.f);
      float4 g = *reinterpret_cast<const float4*>(dout + (ridx ? (long)ridx[r] : r) * D + c);
      g.x *= m_r; g.y *= m_r; g.z *= m_r; g.w *= m_r;
      const float4 o = make_float4(g.x * gm.x, g.y * gm.y, g.z * gm.z, g.w * gm.w);
      *reinterpret_cast<uint2*>(dy + r * D + c) = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
      abias.x += o.x; abias.y += o.y; abias.z += o.z; abias.w += o.w;
      if (gamma && y) {
        const uint2 u = *reinterpret_cast<const uint2*>(y + r * D + c);
        ag.x += g.x * bf2f((bf16_t)(u.x & 0xffff)); ag.y += g.y * bf2f((bf16_t)(u.x >> 16));
        ag.z += g.z * bf2f((bf16_t)(u.y & 0xffff)); ag.w += g.w * bf2f((bf16_t)(u.y >> 16));
      }
    }
  }
  red[0][rl][cl] = ag;
  red[1][rl][cl] = abias;
  __syncthreads();
  if (rl == 0 && c < D) {
    float4 a = red[0][0][cl], b = red[1][0][cl];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      const float4 a2 = red[0][i][cl], b2 = red[1][i][cl];
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
    }
    if (partial) {   // [gridDim.y][2][D] partial rows for the ordered sum of the reduction ledger
      *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.y * 2) * D + c) = a;
      *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.y * 2 + 1) * D + c) = b;
    } else {
      if (gamma && y) { atomicAdd(&dgamma[c], a.x); atomicAdd(&dgamma[c + 1], a.y); atomicAdd(&dgamma[c + 2], a.z); atomicAdd(&dgamma[c + 3], a.w); }
      if (dbias) { atomicAdd(&dbias[c], b.x); atomicAdd(&dbias[c + 1], b.y); atomicAdd(&dbias[c + 2], b.z); atomicAdd(&dbias[c + 3], b.w); }
    }
  }
}
// scalar fallback for D % 4 != 0
__global__ __launch_bounds__(256) void layerscale_bwd_scalar_kernel(const float* __restrict__ dout, const bf16_t* __restrict__ y,
                                                                    const float* __restrict__ gamma, bf16_t* __restrict__ dy,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbias,
                                                                    const float* __restrict__ rowscale, float scale, int rows, int D) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float acc = 0.f, accb = 0.f;
  if (c < D) {
    const float gm = gamma ? gamma[c] : 1.f;
    for (long r = (long)blockIdx.y * 4 + rl; r < rows; r += (long)gridDim.y * 4) {
      const float g = dout[r * D + c] * scale * (rowscale ? rowscale[r] : 1.f);
      dy[r * D + c] = f2bf(g * gm);
      accb += g * gm;
      if (gamma && y) acc += g * bf2f(y[r * D + c]);
    }
  }
  red[0][rl][cl] = acc; red[1][rl][cl] = accb;
  __syncthreads();
  if (rl == 0 && c < D) {
    if (gamma && y) atomicAdd(&dgamma[c], red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
    if (dbias) atomicAdd(&dbias[c], red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
  }
}

// column sums: bf16 rows read as 16-byte vectors (8 columns per thread); block = 32 column-chunks x 8 row-lanes
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int rows, int N,
                                                              float* __restrict__ partial) {
  __shared__ float red[8][32][9];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cl) * 8;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
  if (c < N) {
    for (long r = (long)blockIdx.y * 8 + rl; r < rows; r += (long)gridDim.y * 8) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + r * N + c);
      a[0] += bf2f((bf16_t)(u.x & 0xffff)); a[1] += bf2f((bf16_t)(u.x >> 16));
      a[2] += bf2f((bf16_t)(u.y & 0xffff)); a[3] += bf2f((bf16_t)(u.y >> 16));
      a[4] += bf2f((bf16_t)(u.z & 0xffff)); a[5] += bf2f((bf16_t)(u.z >> 16));
      a[6] += bf2f((bf16_t)(u.w & 0xffff)); a[7] += bf2f((bf16_t)(u.w >> 16));
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cl][j] = a[j];
  __syncthreads();
  // 256 threads reduce 32 chunks x 8 columns over the 8 row-lanes
  const int cc = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int col = (blockIdx.x * 32 + cc) * 8 + j;
  if (col < N) {
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s2 += red[i][cc][j];
    if (partial) partial[(size_t)blockIdx.y * N + col] = s2;   // ordered sum at the reduction ledger's next flush
    else atomicAdd(&out[col], s2);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int N) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float acc = 0.f;
  if (c < N) {
    for (long r = (long)blockIdx.y * 4 + rl; r < rows; r += (long)gridDim.y * 4) {
      if (sizeof(T) == 2) acc += bf2f(((const bf16_t*)x)[r * N + c]);
      else acc += ((const float*)x)[r * N + c];
    }
  }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < N) atomicAdd(&out[c], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

// fp32 column sums, 16 bytes per lane and four independent row loads in flight per thread (the teacher-logit center sums over
// [rows, 65536]: the scalar form above ran at 3.7 TB/s)
__global__ __launch_bounds__(256) void colsum_f32_vec_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int N,
                                                             float* __restrict__ partial) {
  __shared__ float4 red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (c < N) {
    const long step = (long)gridDim.y * 4;
    long r = (long)blockIdx.y * 4 + rl;
    for (; r + 3 * step < rows; r += 4 * step) {
      const float4 v0 = *reinterpret_cast<const float4*>(x + r * N + c);
      const float4 v1 = *reinterpret_cast<const float4*>(x + (r + step) * N + c);
      const float4 v2 = *reinterpret_cast<const float4*>(x + (r + 2 * step) * N + c);
      const float4 v3 = *reinterpret_cast<const float4*>(x + (r + 3 * step) * N + c);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; r < rows; r += step) {
      const float4 v0 = *reinterpret_cast<const float4*>(x + r * N + c);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
  }
  red[rl][cl] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (rl == 0 && c < N) {
    const float4 s0 = red[0][cl], s1 = red[1][cl], s2 = red[2][cl], s3 = red[3][cl];
    const float4 t = make_float4(s0.x + s1.x + s2.x + s3.x, s0.y + s1.y + s2.y + s3.y, s0.z + s1.z + s2.z + s3.z, s0.w + s1.w + s2.w + s3.w);
    if (partial) *reinterpret_cast<float4*>(partial + (size_t)blockIdx.y * N + c) = t;   // row slabs added in order by colsum_slabs_kernel
    else { float* o = out + c; o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w; }       // a single row slab: plain add
  }
}
// out[c] += partial[0][c] + partial[1][c] + ...  (fixed order: the sums feed the loss centers, which must not depend on scheduling)
__global__ __launch_bounds__(256) void colsum_slabs_kernel(const float* __restrict__ partial, float* __restrict__ out, int slabs, int N) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= N) return;
  float4 a = *reinterpret_cast<const float4*>(partial + c);
  for (int s = 1; s < slabs; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)s * N + c);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  float* o = out + c;
  o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
}

// ------------------------------------------------------------------------------------ gather / scatter / cast
__global__ void gather_rows_kernel(const float* __restrict__ src, int ld, const int64_t* __restrict__ idx, bf16_t* __restrict__ ob,
                                   float* __restrict__ of, int M, int D) {
  const long m = blockIdx.x;
  const float* s = src + idx[m] * ld;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float v = s[d];
    if (of) of[m * D + d] = v;
    if (ob) ob[m * D + d] = f2bf(v);
  }
}
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ dst,
                                        int ld, int M, int D) {
  const long m = blockIdx.x;
  float* o = dst + idx[m] * ld;
  for (int d = threadIdx.x; d < D; d += blockDim.x) o[d] += src[m * D + d];
}
__global__ void cast_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(s + i);
    *reinterpret_cast<uint2*>(d + i) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
  }
  if (i < n) for (long j = i; j < n && j < i + 4; ++j) d[j] = f2bf(s[j]);
}
// dst bf16 [R, Cpad] = cast(src f32 [R, C]) zero padded (patch-embed weights when 3*p*p is not a multiple of 8, e.g. /14)
__global__ void cast_pad_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int Cc, int Cpad) {
  const long n = (long)R * Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = i % Cpad; const long r = i / Cpad;
    dst[i] = c < Cc ? f2bf(src[r * Cc + c]) : (bf16_t)0;
  }
}
// dst f32 [R, C] += src f32 [R, Cpad][:, :C]
__global__ void unpad_accumulate_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc, int Cpad) {
  const long n = (long)R * Cc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = i % Cc; const long r = i / Cc;
    dst[i] += src[r * Cpad + c];
  }
}
__global__ void scale_kernel(float* d, float a, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d[i] *= a;
}
__global__ void fill_kernel(float* d, float v, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d[i] = v;
}

// ------------------------------------------------------------------------------------ L2 normalise / weight-norm
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, bf16_t* __restrict__ y,
                                                         float* __restrict__ inv, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * D;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
  const float nrm = sqrtf(wave_sum(s));
  const float iv = 1.f / fmaxf(nrm, eps);
  if (lane == 0) inv[row] = iv;
  for (int c = lane; c < D; c += 64) y[row * D + c] = f2bf(xr[c] * iv);
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ inv, bf16_t* __restrict__ dx, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float iv = inv[row];
  float dot = 0.f;
  for (int c = lane; c < D; c += 64) dot += dy[row * D + c] * x[row * D + c] * iv;
  dot = wave_sum(dot);
  for (int c = lane; c < D; c += 64) {
    const float yv = x[row * D + c] * iv;
    dx[row * D + c] = f2bf((dy[row * D + c] - yv * dot) * iv);
  }
}
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             bf16_t* __restrict__ w, int K, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= K) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float a = v[row * D + c]; s += a * a; }
  const float sc = g[row] / sqrtf(wave_sum(s));
  for (int c = lane; c < D; c += 64) w[row * D + c] = f2bf(v[row * D + c] * sc);
}
__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                             const float* __restrict__ g, float* __restrict__ dv,
                                                             float* __restrict__ dg, int K, int D) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= K) return;
  float s = 0.f, dot = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float a = v[row * D + c];
    s += a * a;
    dot += a * dw[row * D + c];
  }
  s = wave_sum(s);
  dot = wave_sum(dot);
  const float nrm = sqrtf(s), gg = g[row];
  if (lane == 0) dg[row] += dot / nrm;
  for (int c = lane; c < D; c += 64)
    dv[row * D + c] += gg / nrm * (dw[row * D + c] - v[row * D + c] * dot / s);
}

// tiny fp32 matmul (pos-embed interpolation map): C[M,N] (+)= op(A)[M,K] B[K,N]
__global__ void matmul_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
                                  int K, int ta, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(ta ? A[(long)k * M + m] : A[(long)m * K + k], B[(long)k * N + n], s);
  if (accumulate) C[(long)m * N + n] += s;
  else C[(long)m * N + n] = s;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int lt_im2col_bf16(const float* img, void* cols, int B, int C, int H, int W, int p, int kpad, void* stream) {
  LT_CHECK_ARG(img && cols && B > 0 && p > 0 && H % p == 0 && W % p == 0 && kpad >= C * p * p,
               "lt_im2col_bf16: bad arguments (H=%d W=%d p=%d kpad=%d)", H, W, p, kpad);
  hipLaunchKernelGGL(im2col_kernel, dim3(B * (H / p) * (W / p)), dim3(256), 0, ST, img, (bf16_t*)cols, B, C, H, W, p, kpad);
  LT_CHECK_LAUNCH("lt_im2col_bf16");
}
extern "C" int lt_resize_4tap(const float* in, float* out, const int32_t* iy, const float* wy, const int32_t* ix, const float* wx,
                              int planes, int H, int W, int Ho, int Wo, void* stream) {
  LT_CHECK_ARG(in && out && iy && wy && ix && wx && planes > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "lt_resize_4tap: bad arguments");
  LT_CHECK_ARG((((uintptr_t)iy | (uintptr_t)wy | (uintptr_t)ix | (uintptr_t)wx) & 15) == 0, "lt_resize_4tap: tap tables must be 16-byte aligned");
  hipLaunchKernelGGL(resize4tap_kernel, dim3((unsigned)lt_cdiv(Ho * Wo, 256), (unsigned)min(planes, 2048)), dim3(256), 0, ST, in, out, iy, wy, ix, wx,
                     planes, H, W, Ho, Wo);
  LT_CHECK_LAUNCH("lt_resize_4tap");
}
extern "C" int lt_assemble_tokens(const float* patch, const float* cls, const float* pos, const float* mask_token,
                                  const uint8_t* masks, const float* reg, float* x, int B, int n_p, int n_reg, int D, void* stream) {
  LT_CHECK_ARG(patch && cls && pos && x && (!masks || mask_token) && n_reg >= 0 && (n_reg == 0 || reg), "lt_assemble_tokens: bad arguments");
  hipLaunchKernelGGL(assemble_kernel, dim3(B * (n_p + 1 + n_reg)), dim3(256), 0, ST, patch, cls, pos, mask_token, masks, reg, x, B, n_p, n_reg, D);
  LT_CHECK_LAUNCH("lt_assemble_tokens");
}
extern "C" int lt_assemble_tokens_bwd(const float* dx, const uint8_t* masks, void* dpatch_bf16, float* dcls, float* dpos,
                                      float* dmask_token, float* dreg, int B, int n_p, int n_reg, int D, void* stream) {
  LT_CHECK_ARG(dx && dpatch_bf16 && dcls && dpos && (!masks || dmask_token) && n_reg >= 0 && (n_reg == 0 || dreg),
               "lt_assemble_tokens_bwd: bad arguments");
  float* pmask = (masks && lt_ledger::active()) ? lt_ledger::reserve((size_t)n_p * D) : nullptr;
  if (D % 4 == 0 && (((uintptr_t)dx | (uintptr_t)dpatch_bf16) & 15) == 0)
    hipLaunchKernelGGL(assemble_bwd_kernel<4>, dim3(n_p + 1 + n_reg, lt_cdiv(D, 256)), dim3(256), 0, ST, dx, masks, (bf16_t*)dpatch_bf16, dcls,
                       dpos, dmask_token, dreg, B, n_p, n_reg, D, pmask);
  else
    hipLaunchKernelGGL(assemble_bwd_kernel<1>, dim3(n_p + 1 + n_reg, lt_cdiv(D, 64)), dim3(256), 0, ST, dx, masks, (bf16_t*)dpatch_bf16, dcls,
                       dpos, dmask_token, dreg, B, n_p, n_reg, D, pmask);
  if (pmask) lt_ledger::record(dmask_token, pmask, n_p, (long)D, D);
  LT_CHECK_LAUNCH("lt_assemble_tokens_bwd");
}
extern "C" int lt_layernorm_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32, float* mean,
                                float* rstd, int rows, int D, float eps, void* stream) {
  LT_CHECK_ARG(x && w && b && (y_bf16 || y_f32) && rows >= 0 && D > 0 && D <= MAXV * 256, "lt_layernorm_fwd: bad arguments (D=%d)", D);
  if (rows == 0) return LT_OK;
  // rows per wave (read per call: tools/ln_bench.py sweeps it).  1, 2 and 4 rows per wave all run at the rate of a plain fp32 -> bf16 cast of
  // the same size (profiles/r03f_ln_stream_limit.log: 3.1 TB/s from HBM, 5.5 from the Infinity Cache): the kernel is at the streaming limit
  const char* env_r = getenv("LT_LN_FWD_ROWS");
  const int R = env_r ? atoi(env_r) : 1;
  const int nv = (D + 255) / 256;
  const bool al = D % 4 == 0 && (((uintptr_t)x | (uintptr_t)w | (uintptr_t)b | (uintptr_t)y_f32) & 15) == 0 && ((uintptr_t)y_bf16 & 7) == 0;
#define LT_LNF(NV, RR) hipLaunchKernelGGL((layernorm_fwd_rows_kernel<NV, RR>), dim3(lt_cdiv(rows, 4 * RR)), dim3(256), 0, ST, x, w, b, (bf16_t*)y_bf16, y_f32, mean, rstd, rows, D, eps)
  if (al && R >= 1 && nv <= 4 && rows >= 1024) {
    if (R >= 4 && nv <= 3) { if (nv <= 2) LT_LNF(2, 4); else LT_LNF(3, 4); }
    else if (R >= 2) { if (nv <= 2) LT_LNF(2, 2); else if (nv == 3) LT_LNF(3, 2); else LT_LNF(4, 2); }
    else { if (nv <= 2) LT_LNF(2, 1); else if (nv == 3) LT_LNF(3, 1); else LT_LNF(4, 1); }
  } else
#undef LT_LNF
  if (D % 4 == 0)
    hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3(lt_cdiv(rows, 4)), dim3(256), 0, ST, x, w, b, (bf16_t*)y_bf16, y_f32,
                       mean, rstd, rows, D, eps);
  else
    hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3(lt_cdiv(rows, 4)), dim3(256), 0, ST, x, w, b, (bf16_t*)y_bf16, y_f32,
                       mean, rstd, rows, D, eps);
  LT_CHECK_LAUNCH("lt_layernorm_fwd");
}
extern "C" int lt_layernorm_bwd_fused(const float* x, const float* w, const float* mean, const float* rstd, const void* dy,
                                      int dy_is_f32, const float* dres, float* dx, float* dw, float* db, float* ws, int64_t ws_floats,
                                      void* dnext_bf16, const float* gamma_next, const float* rowscale_next, float scale_next,
                                      float* dbias_next, int rows, int D, void* stream);
extern "C" int lt_layernorm_bwd(const float* x, const float* w, const float* mean, const float* rstd, const void* dy,
                                int dy_is_f32, const float* dres, float* dx, float* dw, float* db, float* ws, int64_t ws_floats,
                                int rows, int D, void* stream) {
  return lt_layernorm_bwd_fused(x, w, mean, rstd, dy, dy_is_f32, dres, dx, dw, db, ws, ws_floats, nullptr, nullptr, nullptr, 1.f,
                                nullptr, rows, D, stream);
}
static int layernorm_bwd_impl(const float* x, const float* w, const float* mean, const float* rstd, const void* dy,
                              int dy_is_f32, const float* dres, float* dx, float* dw, float* db, float* ws, int64_t ws_floats,
                              void* dnext_bf16, const float* gamma_next, const float* rowscale_next, float scale_next,
                              float* dbias_next, int rows, int D, const int64_t* ridx, void* stream);
extern "C" int lt_layernorm_bwd_fused(const float* x, const float* w, const float* mean, const float* rstd, const void* dy,
                                      int dy_is_f32, const float* dres, float* dx, float* dw, float* db, float* ws, int64_t ws_floats,
                                      void* dnext_bf16, const float* gamma_next, const float* rowscale_next, float scale_next,
                                      float* dbias_next, int rows, int D, void* stream) {
  return layernorm_bwd_impl(x, w, mean, rstd, dy, dy_is_f32, dres, dx, dw, db, ws, ws_floats, dnext_bf16, gamma_next, rowscale_next, scale_next,
                            dbias_next, rows, D, nullptr, stream);
}
// dx[ridx[r]] = dres[ridx[r]] + LN'(dy[r]) for r < rows (dres may be dx: each target row is read, then written, by one wave; ridx without repeats)
extern "C" int lt_layernorm_bwd_rows(const float* x, const float* w, const float* mean, const float* rstd, const void* dy, int dy_is_f32,
                                     const float* dres, float* dx, const int64_t* ridx, float* dw, float* db, int rows, int D, void* stream) {
  LT_CHECK_ARG(ridx && D % 4 == 0, "lt_layernorm_bwd_rows: needs the row index and D %% 4 == 0");
  return layernorm_bwd_impl(x, w, mean, rstd, dy, dy_is_f32, dres, dx, dw, db, nullptr, 0, nullptr, nullptr, nullptr, 1.f, nullptr, rows, D, ridx, stream);
}
static int layernorm_bwd_impl(const float* x, const float* w, const float* mean, const float* rstd, const void* dy,
                              int dy_is_f32, const float* dres, float* dx, float* dw, float* db, float* ws, int64_t ws_floats,
                              void* dnext_bf16, const float* gamma_next, const float* rowscale_next, float scale_next,
                              float* dbias_next, int rows, int D, const int64_t* ridx, void* stream) {
  LT_CHECK_ARG(x && w && mean && rstd && dy && dx && dw && db && D > 0 && D <= 2048, "lt_layernorm_bwd: bad arguments (D=%d)", D);
  LnbNext nx{(bf16_t*)dnext_bf16, rowscale_next, scale_next};
  if (rows == 0) return LT_OK;
  static const int grid_cap = [] { const char* e = getenv("LT_LN_BWD_GRID"); return e ? atoi(e) : 256; }();  // one 4-wave block per CU: best measured (115 us vs 133 at 512)
  int grid = min(lt_cdiv(rows, 4), grid_cap);
  float* partial = nullptr;
  int prows = 2;
  bool ledger = false;
  const bool vec = D % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0) &&
                   (!dres || (uintptr_t)dres % 16 == 0) && ((uintptr_t)w % 16 == 0);
  LT_CHECK_ARG(!dnext_bf16 || (vec && (uintptr_t)dnext_bf16 % 8 == 0 && (!gamma_next || (uintptr_t)gamma_next % 16 == 0)),
               "lt_layernorm_bwd_fused: the fused next-branch output needs D %% 4 == 0 and 16-byte aligned rows");
  if (vec && ws && ws_floats >= (int64_t)2 * D * 64) {  // workspace given: no same-address atomics
    grid = (int)std::min<int64_t>(grid, ws_floats / (2 * D));
    partial = ws;
  } else if (vec && lt_ledger::active()) {               // deferred: partial rows now, one ordered sum at the next lt_reduce_flush
    prows = (dnext_bf16 && dbias_next) ? 3 : 2;
    partial = lt_ledger::reserve((size_t)grid * prows * D);
    ledger = partial != nullptr;
    if (!ledger) prows = 2;
  }
#define LT_LNB(F32, NV) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<F32, NV>), dim3(grid), dim3(256), 0, ST, x, w, mean, rstd, dy, dres, dx, dw, db, partial, prows, rows, D, nx, gamma_next, dbias_next)
  if (ridx) {   // indexed rows: the bf16-dy, D <= 1024 instantiations the ViT step uses
    LT_CHECK_ARG(vec && !dy_is_f32 && D <= 1024, "lt_layernorm_bwd_rows: vector path only (bf16 dy, D %% 4 == 0, D <= 1024, 16-byte aligned rows)");
    const int nv = (D + 255) / 256;
#define LT_LNBI(NV) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<false, NV, true>), dim3(grid), dim3(256), 0, ST, x, w, mean, rstd, dy, dres, dx, dw, db, partial, prows, rows, D, nx, gamma_next, dbias_next, ridx)
    if (nv <= 2) LT_LNBI(2); else if (nv <= 3) LT_LNBI(3); else LT_LNBI(4);
#undef LT_LNBI
  } else
  if (vec) {
    const int nv = (D + 255) / 256;
    if (dy_is_f32) { if (nv <= 2) LT_LNB(true, 2); else if (nv <= 3) LT_LNB(true, 3); else if (nv <= 4) LT_LNB(true, 4); else LT_LNB(true, 8); }
    else { if (nv <= 2) LT_LNB(false, 2); else if (nv <= 3) LT_LNB(false, 3); else if (nv <= 4) LT_LNB(false, 4); else LT_LNB(false, 8); }
  }
#undef LT_LNB
  if (vec && ledger) {
    lt_ledger::record(dw, partial, grid, (long)prows * D, D);
    lt_ledger::record(db, partial + D, grid, (long)prows * D, D);
    if (prows == 3) lt_ledger::record(dbias_next, partial + 2 * D, grid, (long)prows * D, D);
  } else if (vec && partial)
    hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3(lt_cdiv(D, 64), 2), dim3(256), 0, ST, partial, dw, db, grid, D);
  if (vec) { LT_CHECK_LAUNCH("lt_layernorm_bwd"); }
  if (dy_is_f32)
    hipLaunchKernelGGL(layernorm_bwd_kernel<true>, dim3(grid), dim3(256), 0, ST, x, w, mean, rstd, dy, dres, dx, dw, db, rows, D);
  else
    hipLaunchKernelGGL(layernorm_bwd_kernel<false>, dim3(grid), dim3(256), 0, ST, x, w, mean, rstd, dy, dres, dx, dw, db, rows, D);
  LT_CHECK_LAUNCH("lt_layernorm_bwd");
}
extern "C" int lt_layerscale_bwd_rows(const float* dout, const int64_t* ridx, const void* y_bf16, const float* gamma, void* dy_bf16, float* dgamma,
                                      float* dbias, const float* rowscale, float scale, int rows, int D, void* stream);
extern "C" int lt_layerscale_bwd(const float* dout, const void* y_bf16, const float* gamma, void* dy_bf16, float* dgamma,
                                 float* dbias, const float* rowscale, float scale, int rows, int D, void* stream) {
  return lt_layerscale_bwd_rows(dout, nullptr, y_bf16, gamma, dy_bf16, dgamma, dbias, rowscale, scale, rows, D, stream);
}
extern "C" int lt_layerscale_bwd_rows(const float* dout, const int64_t* ridx, const void* y_bf16, const float* gamma, void* dy_bf16, float* dgamma,
                                      float* dbias, const float* rowscale, float scale, int rows, int D, void* stream) {
  LT_CHECK_ARG(dout && dy_bf16 && (!y_bf16 || (gamma && dgamma)), "lt_layerscale_bwd: null pointer");
  if (rows == 0) return LT_OK;
  if (D % 4 == 0 && (uintptr_t)dout % 16 == 0 && (uintptr_t)dy_bf16 % 8 == 0 && (!y_bf16 || (uintptr_t)y_bf16 % 8 == 0) &&
      (!gamma || (uintptr_t)gamma % 16 == 0)) {
    const bool want = (gamma && y_bf16) || dbias;
    const bool defer = want && lt_ledger::active();
    dim3 grid(lt_cdiv(D, 128), min(lt_cdiv(rows, 8), defer ? 64 : 256));
    float* partial = defer ? lt_ledger::reserve((size_t)grid.y * 2 * D) : nullptr;
    hipLaunchKernelGGL(layerscale_bwd_kernel, grid, dim3(256), 0, ST, dout, (const bf16_t*)y_bf16, gamma, (bf16_t*)dy_bf16, dgamma,
                       dbias, rowscale, scale, rows, D, partial, ridx);
    if (partial) {
      if (gamma && y_bf16) lt_ledger::record(dgamma, partial, (int)grid.y, 2L * D, D);
      if (dbias) lt_ledger::record(dbias, partial + D, (int)grid.y, 2L * D, D);
    }
  } else {
    LT_CHECK_ARG(!ridx, "lt_layerscale_bwd_rows: indexed rows need D %% 4 == 0 and 16-byte aligned rows");
    dim3 grid(lt_cdiv(D, 64), min(lt_cdiv(rows, 4), 256));
    hipLaunchKernelGGL(layerscale_bwd_scalar_kernel, grid, dim3(256), 0, ST, dout, (const bf16_t*)y_bf16, gamma, (bf16_t*)dy_bf16,
                       dgamma, dbias, rowscale, scale, rows, D);
  }
  LT_CHECK_LAUNCH("lt_layerscale_bwd");
}
// LayerScale gradient without the saved branch output y:  sum_r dD[r,c] * y[r,c]  with y = A W^T + b and dW = dD^T A is
// sum_k W[c,k] dW[c,k] + b[c] db[c];  dD = dx * gamma  =>  dgamma[c] += (rowdot(W, dW)[c] + b[c] db[c]) / gamma[c].
// One wave per output channel c.
// blockIdx.y: one of `batch` identically shaped layers whose tensors lie `stride` elements apart in the flat parameter / gradient storage
__global__ __launch_bounds__(256) void layerscale_dgamma_kernel(const bf16_t* __restrict__ W, const float* __restrict__ dW,
                                                                const float* __restrict__ bias, const float* __restrict__ dbias,
                                                                const float* __restrict__ gamma, float* __restrict__ dgamma, int N, int K,
                                                                long stride) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= N) return;
  const long o = (long)blockIdx.y * stride;
  W += o; dW += o; gamma += o; dgamma += o;
  if (bias) bias += o;
  if (dbias) dbias += o;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += bf2f(W[(size_t)c * K + k]) * dW[(size_t)c * K + k];
  acc = wave_sum(acc);
  if (lane == 0) {
    const float g = gamma[c];
    if (bias && dbias) acc += bias[c] * dbias[c];
    if (fabsf(g) > 1e-30f) dgamma[c] += acc / g;
  }
}
extern "C" int lt_layerscale_dgamma(const void* W_bf16, const float* dW, const float* bias, const float* dbias, const float* gamma,
                                    float* dgamma, int N, int K, void* stream) {
  LT_CHECK_ARG(W_bf16 && dW && gamma && dgamma && N > 0 && K > 0, "lt_layerscale_dgamma: bad arguments");
  hipLaunchKernelGGL(layerscale_dgamma_kernel, dim3(lt_cdiv(N, 4)), dim3(256), 0, ST, (const bf16_t*)W_bf16, dW, bias, dbias, gamma,
                     dgamma, N, K, 0L);
  LT_CHECK_LAUNCH("lt_layerscale_dgamma");
}
extern "C" int lt_layerscale_dgamma_batched(const void* W_bf16, const float* dW, const float* bias, const float* dbias, const float* gamma,
                                            float* dgamma, int N, int K, int batch, int64_t stride, void* stream) {
  LT_CHECK_ARG(W_bf16 && dW && gamma && dgamma && N > 0 && K > 0 && batch > 0 && stride >= 0, "lt_layerscale_dgamma_batched: bad arguments");
  hipLaunchKernelGGL(layerscale_dgamma_kernel, dim3(lt_cdiv(N, 4), batch), dim3(256), 0, ST, (const bf16_t*)W_bf16, dW, bias, dbias, gamma,
                     dgamma, N, K, (long)stride);
  LT_CHECK_LAUNCH("lt_layerscale_dgamma_batched");
}
extern "C" int lt_colsum_bf16(const void* x, float* out, int rows, int N, void* stream) {
  LT_CHECK_ARG(x && out, "lt_colsum_bf16: null pointer");
  if (rows == 0) return LT_OK;
  if (N % 8 == 0 && (uintptr_t)x % 16 == 0) {
    const bool defer = lt_ledger::active();
    // deferred: fewer, longer row slabs keep the partial rows small (128 x N floats), still >= 1000 workgroups for N >= 2048
    dim3 grid(lt_cdiv(N, 256), min(lt_cdiv(rows, 8), defer ? 128 : 256));
    float* partial = defer ? lt_ledger::reserve((size_t)grid.y * N) : nullptr;
    hipLaunchKernelGGL(colsum_bf16_vec_kernel, grid, dim3(256), 0, ST, (const bf16_t*)x, out, rows, N, partial);
    if (partial) lt_ledger::record(out, partial, (int)grid.y, (long)N, N);
  } else {
    dim3 grid(lt_cdiv(N, 64), min(lt_cdiv(rows, 4), 128));
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, ST, (const bf16_t*)x, out, rows, N);
  }
  LT_CHECK_LAUNCH("lt_colsum_bf16");
}
extern "C" int lt_colsum_f32(const float* x, float* out, int rows, int N, int accumulate, void* stream) {
  LT_CHECK_ARG(x && out, "lt_colsum_f32: null pointer");
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * N, ST);
    if (e != hipSuccess) { lt_set_error("lt_colsum_f32: memset failed"); return LT_ERR_HIP; }
  }
  if (rows == 0) return LT_OK;
  if (N % 4 == 0 && ((uintptr_t)x & 15) == 0) {
    const int gx = lt_cdiv(N / 4, 64);
    dim3 grid(gx, max(1, min(lt_cdiv(rows, 16), lt_cdiv(2048, gx))));   // ~2048 blocks, >= 4 rows per thread where possible
    float* partial = nullptr;
    if (grid.y > 1) {
      partial = lt_scratch_ring((size_t)grid.y * N);
      if (!partial) { lt_set_error("lt_colsum_f32: scratch allocation failed"); return LT_ERR_HIP; }
    }
    hipLaunchKernelGGL(colsum_f32_vec_kernel, grid, dim3(256), 0, ST, x, out, rows, N, partial);
    if (partial) hipLaunchKernelGGL(colsum_slabs_kernel, dim3(lt_cdiv(N, 1024)), dim3(256), 0, ST, partial, out, (int)grid.y, N);
  } else {
    dim3 grid(lt_cdiv(N, 64), min(lt_cdiv(rows, 4), 128));
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, ST, x, out, rows, N);
  }
  LT_CHECK_LAUNCH("lt_colsum_f32");
}
extern "C" int lt_gather_rows(const float* src, int ld_src, const int64_t* idx, void* out_bf16, float* out_f32, int M, int D,
                              void* stream) {
  LT_CHECK_ARG(src && idx && (out_bf16 || out_f32), "lt_gather_rows: null pointer");
  if (M == 0) return LT_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(M), dim3(256), 0, ST, src, ld_src, idx, (bf16_t*)out_bf16, out_f32, M, D);
  LT_CHECK_LAUNCH("lt_gather_rows");
}
extern "C" int lt_scatter_add_rows(const float* src, const int64_t* idx, float* dst, int ld_dst, int M, int D, void* stream) {
  LT_CHECK_ARG(src && idx && dst, "lt_scatter_add_rows: null pointer");
  if (M == 0) return LT_OK;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(M), dim3(256), 0, ST, src, idx, dst, ld_dst, M, D);
  LT_CHECK_LAUNCH("lt_scatter_add_rows");
}
extern "C" int lt_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  LT_CHECK_ARG(src && dst && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "lt_cast_f32_to_bf16: bad pointer/alignment");
  if (n == 0) return LT_OK;
  const int grid = (int)min((long)2048, (long)lt_cdiv(n, 1024));
  hipLaunchKernelGGL(cast_kernel, dim3(grid), dim3(256), 0, ST, src, (bf16_t*)dst, (long)n);
  LT_CHECK_LAUNCH("lt_cast_f32_to_bf16");
}
extern "C" int lt_fill_f32(float* dst, float value, int64_t n, void* stream) {
  LT_CHECK_ARG(dst, "lt_fill_f32: null pointer");
  if (n == 0) return LT_OK;
  const int grid = (int)min((long)2048, (long)lt_cdiv(n, 256));
  hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, ST, dst, value, (long)n);
  LT_CHECK_LAUNCH("lt_fill_f32");
}
extern "C" int lt_cast_pad_rows(const float* src, void* dst_bf16, int R, int Cc, int Cpad, void* stream) {
  LT_CHECK_ARG(src && dst_bf16 && R > 0 && Cc > 0 && Cpad >= Cc, "lt_cast_pad_rows: bad arguments");
  hipLaunchKernelGGL(cast_pad_rows_kernel, dim3((unsigned)min((long)2048, ((long)R * Cpad + 255) / 256)), dim3(256), 0, ST, src, (bf16_t*)dst_bf16, R, Cc, Cpad);
  LT_CHECK_LAUNCH("lt_cast_pad_rows");
}
extern "C" int lt_unpad_accumulate(const float* src, float* dst, int R, int Cc, int Cpad, void* stream) {
  LT_CHECK_ARG(src && dst && R > 0 && Cc > 0 && Cpad >= Cc, "lt_unpad_accumulate: bad arguments");
  hipLaunchKernelGGL(unpad_accumulate_kernel, dim3((unsigned)min((long)2048, ((long)R * Cc + 255) / 256)), dim3(256), 0, ST, src, dst, R, Cc, Cpad);
  LT_CHECK_LAUNCH("lt_unpad_accumulate");
}
extern "C" int lt_scale_f32(float* dst, float alpha, int64_t n, void* stream) {
  LT_CHECK_ARG(dst, "lt_scale_f32: null pointer");
  if (n == 0) return LT_OK;
  const int grid = (int)min((long)2048, (long)lt_cdiv(n, 256));
  hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(256), 0, ST, dst, alpha, (long)n);
  LT_CHECK_LAUNCH("lt_scale_f32");
}
extern "C" int lt_l2norm_fwd(const float* x, void* y_bf16, float* inv_norm, int rows, int D, float eps, void* stream) {
  LT_CHECK_ARG(x && y_bf16 && inv_norm, "lt_l2norm_fwd: null pointer");
  if (rows == 0) return LT_OK;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(lt_cdiv(rows, 4)), dim3(256), 0, ST, x, (bf16_t*)y_bf16, inv_norm, rows, D, eps);
  LT_CHECK_LAUNCH("lt_l2norm_fwd");
}
extern "C" int lt_l2norm_bwd(const float* dy, const float* x, const float* inv_norm, void* dx_bf16, int rows, int D, void* stream) {
  LT_CHECK_ARG(dy && x && inv_norm && dx_bf16, "lt_l2norm_bwd: null pointer");
  if (rows == 0) return LT_OK;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(lt_cdiv(rows, 4)), dim3(256), 0, ST, dy, x, inv_norm, (bf16_t*)dx_bf16, rows, D);
  LT_CHECK_LAUNCH("lt_l2norm_bwd");
}
extern "C" int lt_weightnorm_fwd(const float* v, const float* g, void* w_bf16, int K, int D, void* stream) {
  LT_CHECK_ARG(v && g && w_bf16, "lt_weightnorm_fwd: null pointer");
  hipLaunchKernelGGL(weightnorm_fwd_kernel, dim3(lt_cdiv(K, 4)), dim3(256), 0, ST, v, g, (bf16_t*)w_bf16, K, D);
  LT_CHECK_LAUNCH("lt_weightnorm_fwd");
}
extern "C" int lt_weightnorm_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int K, int D, void* stream) {
  LT_CHECK_ARG(dw && v && g && dv && dg, "lt_weightnorm_bwd: null pointer");
  hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3(lt_cdiv(K, 4)), dim3(256), 0, ST, dw, v, g, dv, dg, K, D);
  LT_CHECK_LAUNCH("lt_weightnorm_bwd");
}
extern "C" int lt_matmul_f32(const float* A, const float* B, float* C, int M, int N, int K, int trans_a, int accumulate, void* stream) {
  LT_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "lt_matmul_f32: bad arguments");
  hipLaunchKernelGGL(matmul_f32_kernel, dim3(lt_cdiv(N, 128), M), dim3(128), 0, ST, A, B, C, M, N, K, trans_a, accumulate);
  LT_CHECK_LAUNCH("lt_matmul_f32");
}
