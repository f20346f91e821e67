// Convolutional-student kernels for gfx950 (MI355X): the torchvision ResNet-50 student of DistillationV3
// (BASELINE.json configs[3]; reference LT/_models/torchvision/resnet.py:21-47 wraps torchvision.models.resnet50, driven from
// LT/_methods/distillationv3/distillationv3.py:324-354).
//
// Layout: activations NHWC bf16, i.e. a [B*H*W, C] row-major matrix -- a 1x1 convolution IS the token GEMM of gemm.hip, a
// k x k convolution is that GEMM on an im2col matrix [B*Ho*Wo, k*k*C] (taps outer, channels inner: every copy is a 16-byte
// vector of 8 channels).  Convolution weights are kept [Cout][kh][kw][Cin] in the flat parameter storage (the host permutes
// at state_dict import / export), so forward, dgrad and wgrad all run on the unmodified MFMA GEMM.  Everything in this file
// is HBM-bound glue around those GEMMs: im2col / col2im, training-mode BatchNorm (batch statistics, fp32) with fused
// ReLU / residual add, 3x3 max-pooling with saved arg-max, global average pooling.
// All reductions are two-level with a fixed summation order (deterministic: data-parallel replicas stay bit-identical).
#include "lt_common.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
  f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
  f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
  f[4] = bf2f((bf16_t)(u.z & 0xffff)); f[5] = bf2f((bf16_t)(u.z >> 16));
  f[6] = bf2f((bf16_t)(u.w & 0xffff)); f[7] = bf2f((bf16_t)(u.w >> 16));
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

// ---- im2col / col2im (NHWC, channels a multiple of 8) --------------------------------------------------------------------
// cols[(b,oy,ox)][(ky*KW + kx)*C + c] = x[b][oy*s - pad + ky][ox*s - pad + kx][c]   (0 outside the image)
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ cols, int B, int H, int W, int C,
                                                          int KH, int KW, int stride, int pad, int Ho, int Wo, int ld) {
  const int cv = C >> 3;
  const long total = (long)B * Ho * Wo * KH * KW * cv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % cv);
    long t = i / cv;
    const int tap = (int)(t % (KH * KW));
    const long row = t / (KH * KW);
    const int ox = (int)(row % Wo);
    const int oy = (int)((row / Wo) % Ho);
    const int b = (int)(row / ((long)Wo * Ho));
    const int iy = oy * stride - pad + tap / KW, ix = ox * stride - pad + tap % KW;
    uint4 u = make_uint4(0, 0, 0, 0);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) u = *reinterpret_cast<const uint4*>(x + (((long)b * H + iy) * W + ix) * C + v * 8);
    *reinterpret_cast<uint4*>(cols + row * ld + (long)tap * C + v * 8) = u;
  }
}
// transpose of the gather above in gather form (no atomics): dx[pixel] = sum over the windows that read it (+ add[pixel])
__global__ __launch_bounds__(256) void col2im_nhwc_kernel(const bf16_t* __restrict__ dcols, const bf16_t* __restrict__ add, bf16_t* __restrict__ dx,
                                                          int B, int H, int W, int C, int KH, int KW, int stride, int pad, int Ho, int Wo, int ld) {
  const int cv = C >> 3;
  const long total = (long)B * H * W * cv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % cv);
    const long pix = i / cv;
    const int ix = (int)(pix % W);
    const int iy = (int)((pix / W) % H);
    const int b = (int)(pix / ((long)W * H));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (add) { unpack8(*reinterpret_cast<const uint4*>(add + pix * C + v * 8), acc); }
    for (int ky = 0; ky < KH; ++ky) {
      const int ny = iy + pad - ky;
      if (ny < 0 || ny % stride) continue;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < KW; ++kx) {
        const int nx = ix + pad - kx;
        if (nx < 0 || nx % stride) continue;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(dcols + (((long)b * Ho + oy) * Wo + ox) * ld + (long)(ky * KW + kx) * C + v * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += f[j];
      }
    }
    *reinterpret_cast<uint4*>(dx + pix * C + v * 8) = pack8(acc);
  }
}
// stem: NCHW f32 image -> cols[(b,oy,ox)][c*KH*KW + ky*KW + kx] bf16 (torch's weight.flatten(1) order), zero-padded to ld columns.
// One thread = 8 consecutive k of one row = one 16-byte store (ld % 8 == 0); the scattered 4-byte reads hit L1 / L2.
__global__ __launch_bounds__(256) void im2col_nchw_f32_kernel(const float* __restrict__ x, bf16_t* __restrict__ cols, int B, int Cin, int H, int W,
                                                              int KH, int KW, int stride, int pad, int Ho, int Wo, int ld) {
  const int kv = ld >> 3;
  const long total = (long)B * Ho * Wo * kv;
  const int kk = KH * KW, kreal = Cin * kk;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k0 = (int)(i % kv) * 8;
    const long row = i / kv;
    const int ox = (int)(row % Wo);
    const int oy = (int)((row / Wo) % Ho);
    const int b = (int)(row / ((long)Wo * Ho));
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float val = 0.f;
      if (k < kreal) {
        const int c = k / kk, ky = (k % kk) / KW, kx = k % KW;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = x[(((long)b * Cin + c) * H + iy) * W + ix];
      }
      f[j] = val;
    }
    *reinterpret_cast<uint4*>(cols + row * ld + k0) = pack8(f);
  }
}

// ---- training-mode BatchNorm over the rows of a [rows, C] matrix ------------------------------------------------------------
// pass 1: per-block partial column sums.  MODE 0: (sum x, sum x^2).  MODE 1 (backward): dz = dy * (y > 0 if relu), optionally
// stored, and (sum dz, sum dz * xhat).  Block = VEC channel-vectors x LANES row-lanes (VEC * LANES = 256); grid.y row chunks.
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd, bf16_t* __restrict__ dz_out,
                                                         float* __restrict__ partial, long rows, int C, int vecs_per_block) {
  __shared__ float red[256][17];
  const int lanes = 256 / vecs_per_block;
  const int vl = threadIdx.x % vecs_per_block, lane = threadIdx.x / vecs_per_block;
  const int v = blockIdx.x * vecs_per_block + vl;
  const int G = gridDim.y;
  const long per = (rows + G - 1) / G;
  const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
  float a[8], q[8], mu[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = 0.f; q[j] = 0.f; mu[j] = 0.f; rs[j] = 1.f; }
  const bool active = v * 8 < C;
  if (MODE == 1 && active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = mean[v * 8 + j]; rs[j] = rstd[v * 8 + j]; }
  }
  if (active) {
    for (long r = r0 + lane; r < r1; r += lanes) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(x + r * C + v * 8), f);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] += f[j]; q[j] = fmaf(f[j], f[j], q[j]); }
      } else {
        float d[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + r * C + v * 8), d);
        if (y) {
          float o[8];
          unpack8(*reinterpret_cast<const uint4*>(y + r * C + v * 8), o);
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
        }
        if (dz_out) *reinterpret_cast<uint4*>(dz_out + r * C + v * 8) = pack8(d);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] += d[j]; q[j] = fmaf(d[j], (f[j] - mu[j]) * rs[j], q[j]); }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = a[j]; red[threadIdx.x][8 + j] = q[j]; }
  __syncthreads();
  // thread (vl, j16) sums the row-lanes of its channel-vector in lane order (fixed order: deterministic)
  for (int o = threadIdx.x; o < vecs_per_block * 16; o += 256) {
    const int vv = o / 16, j = o % 16;
    const int vg = blockIdx.x * vecs_per_block + vv;
    if (vg * 8 >= C) continue;
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * vecs_per_block + vv][j];
    // partial layout [2][G][C]
    partial[((long)(j >> 3) * G + blockIdx.y) * C + vg * 8 + (j & 7)] = s;
  }
}
// pass 2: block = BN_FC channels x BN_FL partial-lanes.  Every thread adds its share of the G per-block partials (g = lane, lane + BN_FL, ...)
// in double, the lanes of a channel are then combined in lane order: a fixed summation tree (deterministic).  8 channels x 32 lanes: a
// thread walks G / 32 partials (32 dependent double adds at G = 1024) -- with 32 channels x 8 lanes it was 128, and the kernel took 29 us
// per layer for a few KB of data (one thread walking all G: 140-170 us).
constexpr int BN_FC = 8, BN_FL = 32;
__device__ __forceinline__ void bn_sum_partials(const float* __restrict__ partial, int G, int C, int c, int lane, double& s0, double& s1,
                                                double (*red)[BN_FC][2]) {
  // every load of the thread's share is issued before the first add: as a loop of dependent load -> add pairs the 16 round trips to the
  // partial rows ran one after the other (29 us per layer for a few hundred KB); the adds keep their order (bit-identical sums)
  constexpr int PER = (LT_BN_MAX_CHUNKS + BN_FL - 1) / BN_FL;
  float pa[PER], pb[PER];
  const int cc = c < C ? c : C - 1;
#pragma unroll
  for (int i = 0; i < PER; ++i) {          // branch-free (clamped) addresses: a guarded load gets a basic block and a wait of its own
    const int g = lane + i * BN_FL;
    const int gc = g < G ? g : G - 1;
    pa[i] = partial[(long)gc * C + cc];
    pb[i] = partial[((long)G + gc) * C + cc];
  }
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const bool ok = c < C && lane + i * BN_FL < G;
    a += ok ? (double)pa[i] : 0.0;
    b += ok ? (double)pb[i] : 0.0;
  }
  red[lane][threadIdx.x % BN_FC][0] = a;
  red[lane][threadIdx.x % BN_FC][1] = b;
  __syncthreads();
  s0 = 0.0; s1 = 0.0;
#pragma unroll
  for (int l = 0; l < BN_FL; ++l) { s0 += red[l][threadIdx.x % BN_FC][0]; s1 += red[l][threadIdx.x % BN_FC][1]; }
}
// forward: batch statistics (biased variance for normalisation, unbiased for the running estimate, torch semantics)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int G, long rows, int C, float eps, float momentum,
                                                          float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var) {
  __shared__ double red[BN_FL][BN_FC][2];
  const int c = blockIdx.x * BN_FC + (threadIdx.x % BN_FC), lane = threadIdx.x / BN_FC;
  double s, ss;
  bn_sum_partials(partial, G, C, c, lane, s, ss, red);
  if (lane != 0 || c >= C) return;
  const double n = (double)rows;
  const double m = s / n;
  double var = ss / n - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(rows > 1 ? var * n / (n - 1.0) : var);
}
// backward: dgamma += sum dz*xhat, dbeta += sum dz, and the two means the input gradient needs
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int G, long rows, int C, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ c1, float* __restrict__ c2) {
  __shared__ double red[BN_FL][BN_FC][2];
  const int c = blockIdx.x * BN_FC + (threadIdx.x % BN_FC), lane = threadIdx.x / BN_FC;
  double s, sx;
  bn_sum_partials(partial, G, C, c, lane, s, sx, red);
  if (lane != 0 || c >= C) return;
  if (dbeta) dbeta[c] += (float)s;
  if (dgamma) dgamma[c] += (float)sx;
  c1[c] = (float)(s / (double)rows);
  c2[c] = (float)(sx / (double)rows);
}
// ---- SyncBatchNorm pieces: the per-rank raw sums leave the device-side reduction as doubles (one all-reduce over [2C + 1]: the last
// element carries the row count), the statistics / input-gradient coefficients are then formed from the global sums.
__global__ __launch_bounds__(256) void bn_sums_kernel(const float* __restrict__ partial, int G, long rows, int C, double* __restrict__ sums) {
  __shared__ double red[BN_FL][BN_FC][2];
  const int c = blockIdx.x * BN_FC + (threadIdx.x % BN_FC), lane = threadIdx.x / BN_FC;
  double s, ss;
  bn_sum_partials(partial, G, C, c, lane, s, ss, red);
  if (blockIdx.x == 0 && threadIdx.x == 0) sums[2L * C] = (double)rows;
  if (lane != 0 || c >= C) return;
  sums[c] = s;
  sums[(long)C + c] = ss;
}
__global__ __launch_bounds__(256) void bn_finalize_sums_kernel(const double* __restrict__ sums, int C, float eps, float momentum, float* __restrict__ mean,
                                                               float* __restrict__ rstd, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = sums[2L * C];
  const double m = sums[c] / n;
  double var = sums[(long)C + c] / n - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
}
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ partial, int G, long rows, int C, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, double* __restrict__ sums) {
  __shared__ double red[BN_FL][BN_FC][2];
  const int c = blockIdx.x * BN_FC + (threadIdx.x % BN_FC), lane = threadIdx.x / BN_FC;
  double s, sx;
  bn_sum_partials(partial, G, C, c, lane, s, sx, red);
  if (blockIdx.x == 0 && threadIdx.x == 0) sums[2L * C] = (double)rows;
  if (lane != 0 || c >= C) return;
  if (dbeta) dbeta[c] += (float)s;         // parameter gradients stay local: the data-parallel gradient mean treats them like every other one
  if (dgamma) dgamma[c] += (float)sx;
  sums[c] = s;
  sums[(long)C + c] = sx;
}
__global__ __launch_bounds__(256) void bn_bwd_coefs_kernel(const double* __restrict__ sums, int C, float* __restrict__ c1, float* __restrict__ c2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = sums[2L * C];
  c1[c] = (float)(sums[c] / n);
  c2[c] = (float)(sums[(long)C + c] / n);
}
// y = act( gamma * (x - mean) * rstd + beta (+ resid) ), act = ReLU or identity
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, const bf16_t* __restrict__ resid,
                                                       bf16_t* __restrict__ y, long rows, int C, int relu) {
  const int cv = C >> 3;
  const long total = rows * cv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % cv);
    float f[8], r[8];
    unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
    if (resid) unpack8(*reinterpret_cast<const uint4*>(resid + i * 8), r);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = v * 8 + j;
      float o = fmaf((f[j] - mean[c]) * rstd[c], gamma[c], beta[c]);
      if (resid) o += r[j];
      f[j] = relu ? fmaxf(o, 0.f) : o;
    }
    *reinterpret_cast<uint4*>(y + i * 8) = pack8(f);
  }
}
// dx = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ c1,
                                                           const float* __restrict__ c2, bf16_t* __restrict__ dx, long rows, int C) {
  const int cv = C >> 3;
  const long total = rows * cv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % cv);
    float f[8], d[8];
    unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
    unpack8(*reinterpret_cast<const uint4*>(dz + i * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = v * 8 + j;
      const float xh = (f[j] - mean[c]) * rstd[c];
      f[j] = gamma[c] * rstd[c] * (d[j] - c1[c] - xh * c2[c]);
    }
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8(f);
  }
}

// ---- 3x3 / stride 2 / pad 1 max pooling (NHWC) with saved arg-max tap (torch: first maximum in scan order wins, padding skipped) ----
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, uint8_t* __restrict__ idx, int B,
                                                          int H, int W, int C, int Ho, int Wo) {
  const int cv = C >> 3;
  const long total = (long)B * Ho * Wo * cv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % cv);
    const long o = i / cv;
    const int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((long)Wo * Ho));
    float best[8];
    int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = 0; }
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + (((long)b * H + iy) * W + ix) * C + v * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (f[j] > best[j] || f[j] != f[j]) { best[j] = f[j]; arg[j] = ky * 3 + kx; }
      }
    }
    *reinterpret_cast<uint4*>(y + o * C + v * 8) = pack8(best);
    uint2 pk;
    pk.x = (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
    pk.y = (uint32_t)arg[4] | ((uint32_t)arg[5] << 8) | ((uint32_t)arg[6] << 16) | ((uint32_t)arg[7] << 24);
    *reinterpret_cast<uint2*>(idx + o * C + v * 8) = pk;
  }
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ dy, const uint8_t* __restrict__ idx, bf16_t* __restrict__ dx, int B,
                                                          int H, int W, int C, int Ho, int Wo) {
  const int cv = C >> 3;
  const long total = (long)B * H * W * cv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % cv);
    const long pix = i / cv;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int ny = iy + 1 - ky;
      if (ny < 0 || (ny & 1)) continue;
      const int oy = ny >> 1;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int nx = ix + 1 - kx;
        if (nx < 0 || (nx & 1)) continue;
        const int ox = nx >> 1;
        if (ox >= Wo) continue;
        const long o = ((long)b * Ho + oy) * Wo + ox;
        const uint2 pk = *reinterpret_cast<const uint2*>(idx + o * C + v * 8);
        float d[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + o * C + v * 8), d);
        const int tap = ky * 3 + kx;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int a = (int)(((j < 4 ? pk.x : pk.y) >> (8 * (j & 3))) & 0xff);
          if (a == tap) acc[j] += d[j];
        }
      }
    }
    *reinterpret_cast<uint4*>(dx + pix * C + v * 8) = pack8(acc);
  }
}

// ---- global average pooling over the n positions of every image ([B, n, C] bf16 -> [B, C] bf16) and its backward, fused with
// the sum of the two gradient paths that reach the feature map: dfeat[b,p,:] = d_tok[b,p,:] + d_pool[b,:] / n
__global__ __launch_bounds__(256) void token_mean_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int n, int C) {
  const int cv = C >> 3;
  const long total = (long)B * cv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % cv);
    const long b = i / cv;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int p = 0; p < n; ++p) {
      float f[8];
      unpack8(*reinterpret_cast<const uint4*>(x + (b * n + p) * C + v * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
    const float inv = 1.f / (float)n;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= inv;
    *reinterpret_cast<uint4*>(out + b * C + v * 8) = pack8(acc);
  }
}
__global__ __launch_bounds__(256) void pool_bwd_add_kernel(const float* __restrict__ d_tok, const float* __restrict__ d_pool, bf16_t* __restrict__ out,
                                                           int B, int n, int C) {
  const int cv = C >> 2;
  const long total = (long)B * n * cv;
  const float inv = 1.f / (float)n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int v = (int)(i % cv);
    const long row = i / cv;
    const long b = row / n;
    float4 t = d_tok ? *reinterpret_cast<const float4*>(d_tok + row * C + v * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (d_pool) {
      const float4 p = *reinterpret_cast<const float4*>(d_pool + b * C + v * 4);
      t.x = fmaf(p.x, inv, t.x); t.y = fmaf(p.y, inv, t.y); t.z = fmaf(p.z, inv, t.z); t.w = fmaf(p.w, inv, t.w);
    }
    *reinterpret_cast<uint2*>(out + row * C + v * 4) = make_uint2(pack_bf2(t.x, t.y), pack_bf2(t.z, t.w));
  }
}
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, long nvec) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    float f[8], g[8];
    unpack8(*reinterpret_cast<const uint4*>(a + i * 8), f);
    unpack8(*reinterpret_cast<const uint4*>(b + i * 8), g);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] += g[j];
    *reinterpret_cast<uint4*>(out + i * 8) = pack8(f);
  }
}

inline int grid_for(long work_items) { return (int)min((long)8192, max((long)1, (work_items + 255) / 256)); }
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int lt_im2col_nhwc_bf16(const void* x, void* cols, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int ld, void* stream) {
  LT_CHECK_ARG(x && cols && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               "lt_im2col_nhwc_bf16: bad arguments (C=%d must be a multiple of 8)", C);
  LT_CHECK_ARG(ld >= KH * KW * C && ld % 8 == 0 && al16(x) && al16(cols), "lt_im2col_nhwc_bf16: ld=%d too small / misaligned", ld);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  LT_CHECK_ARG(Ho > 0 && Wo > 0, "lt_im2col_nhwc_bf16: empty output");
  hipLaunchKernelGGL(im2col_nhwc_kernel, dim3(grid_for((long)B * Ho * Wo * KH * KW * (C / 8))), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)cols, B, H,
                     W, C, KH, KW, stride, pad, Ho, Wo, ld);
  LT_CHECK_LAUNCH("lt_im2col_nhwc_bf16");
}
extern "C" int lt_col2im_nhwc_bf16(const void* dcols, const void* add, void* dx, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
                                   int ld, void* stream) {
  LT_CHECK_ARG(dcols && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
               "lt_col2im_nhwc_bf16: bad arguments (C=%d must be a multiple of 8)", C);
  LT_CHECK_ARG(ld >= KH * KW * C && ld % 8 == 0 && al16(dcols) && al16(dx) && al16(add), "lt_col2im_nhwc_bf16: ld=%d too small / misaligned", ld);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  hipLaunchKernelGGL(col2im_nhwc_kernel, dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0, ST, (const bf16_t*)dcols, (const bf16_t*)add,
                     (bf16_t*)dx, B, H, W, C, KH, KW, stride, pad, Ho, Wo, ld);
  LT_CHECK_LAUNCH("lt_col2im_nhwc_bf16");
}
extern "C" int lt_im2col_nchw_f32(const float* x, void* cols, int B, int Cin, int H, int W, int KH, int KW, int stride, int pad, int ld, void* stream) {
  LT_CHECK_ARG(x && cols && B > 0 && Cin > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && ld >= Cin * KH * KW && ld % 8 == 0 &&
                   al16(cols),
               "lt_im2col_nchw_f32: bad arguments (ld=%d must be >= Cin*KH*KW and a multiple of 8, cols 16-byte aligned)", ld);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  LT_CHECK_ARG(Ho > 0 && Wo > 0, "lt_im2col_nchw_f32: empty output");
  hipLaunchKernelGGL(im2col_nchw_f32_kernel, dim3(grid_for((long)B * Ho * Wo * (ld / 8))), dim3(256), 0, ST, x, (bf16_t*)cols, B, Cin, H, W, KH, KW, stride,
                     pad, Ho, Wo, ld);
  LT_CHECK_LAUNCH("lt_im2col_nchw_f32");
}

namespace {
struct BnGeom { int vpb, gx, G; };
inline BnGeom bn_geom(long rows, int C) {
  BnGeom g;
  const int vecs = C / 8;
  g.vpb = 1;
  while (g.vpb < vecs && g.vpb < 256) g.vpb <<= 1;          // power of two <= 256 (256 % vpb == 0)
  g.gx = (vecs + g.vpb - 1) / g.vpb;
  const int lanes = 256 / g.vpb;
  g.G = (int)max((long)1, min((long)LT_BN_MAX_CHUNKS, rows / ((long)lanes * 4)));
  return g;
}
}  // namespace

extern "C" int64_t lt_batchnorm_ws_floats(int C) { return 2LL * LT_BN_MAX_CHUNKS * C + 2LL * C; }

extern "C" int lt_batchnorm_fwd(const void* x, const float* gamma, const float* beta, const void* resid, void* y, float* mean, float* rstd,
                                float* running_mean, float* running_var, int64_t rows, int C, float eps, float momentum, int relu, float* ws,
                                void* stream) {
  LT_CHECK_ARG(x && gamma && beta && y && mean && rstd && ws && rows > 0 && C > 0 && C % 8 == 0, "lt_batchnorm_fwd: bad arguments (C=%d)", C);
  LT_CHECK_ARG(al16(x) && al16(y) && al16(resid), "lt_batchnorm_fwd: x / y / resid must be 16-byte aligned");
  const BnGeom g = bn_geom(rows, C);
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(g.gx, g.G), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, ws, (long)rows, C, g.vpb);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(lt_cdiv(C, BN_FC)), dim3(256), 0, ST, ws, g.G, (long)rows, C, eps, momentum, mean, rstd, running_mean,
                     running_var);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, ST, (const bf16_t*)x, mean, rstd, gamma, beta, (const bf16_t*)resid,
                     (bf16_t*)y, (long)rows, C, relu);
  LT_CHECK_LAUNCH("lt_batchnorm_fwd");
}

/* eval-mode BatchNorm: y = act(gamma * (x - mean) * rstd + beta (+ resid)) with caller-provided statistics (the running estimates) */
extern "C" int lt_batchnorm_apply(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const void* resid, void* y,
                                  int64_t rows, int C, int relu, void* stream) {
  LT_CHECK_ARG(x && mean && rstd && gamma && beta && y && rows > 0 && C > 0 && C % 8 == 0 && al16(x) && al16(y) && al16(resid),
               "lt_batchnorm_apply: bad arguments (C=%d)", C);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, ST, (const bf16_t*)x, mean, rstd, gamma, beta, (const bf16_t*)resid,
                     (bf16_t*)y, (long)rows, C, relu);
  LT_CHECK_LAUNCH("lt_batchnorm_apply");
}

extern "C" int lt_batchnorm_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* mean, const float* rstd, void* dz,
                                void* dx, float* dgamma, float* dbeta, int64_t rows, int C, float* ws, void* stream) {
  LT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && ws && rows > 0 && C > 0 && C % 8 == 0, "lt_batchnorm_bwd: bad arguments (C=%d)", C);
  LT_CHECK_ARG(!y || dz, "lt_batchnorm_bwd: a ReLU mask (y) needs a dz buffer for the masked upstream gradient");
  LT_CHECK_ARG(al16(dy) && al16(y) && al16(x) && al16(dz) && al16(dx), "lt_batchnorm_bwd: tensors must be 16-byte aligned");
  const BnGeom g = bn_geom(rows, C);
  float* c1 = ws + 2LL * LT_BN_MAX_CHUNKS * C;
  float* c2 = c1 + C;
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(g.gx, g.G), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y, mean, rstd,
                     (bf16_t*)dz, ws, (long)rows, C, g.vpb);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(lt_cdiv(C, BN_FC)), dim3(256), 0, ST, ws, g.G, (long)rows, C, dgamma, dbeta, c1, c2);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, ST, (const bf16_t*)(dz ? dz : dy), (const bf16_t*)x, mean, rstd,
                     gamma, c1, c2, (bf16_t*)dx, (long)rows, C);
  LT_CHECK_LAUNCH("lt_batchnorm_bwd");
}

/* SyncBatchNorm (torch.nn.SyncBatchNorm, which Lightning's sync_batchnorm=True puts in place of every BatchNorm layer: train_helpers.py:223,
 * 335-342) in two halves around the caller's all-reduce.  sums: doubles [2C + 1] = (sum x, sum x^2, rows) resp. (sum dz, sum dz*xhat, rows). */
extern "C" int lt_batchnorm_stats(const void* x, int64_t rows, int C, float* ws, double* sums, void* stream) {
  LT_CHECK_ARG(x && ws && sums && rows > 0 && C > 0 && C % 8 == 0 && al16(x), "lt_batchnorm_stats: bad arguments (C=%d)", C);
  const BnGeom g = bn_geom(rows, C);
  hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(g.gx, g.G), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (bf16_t*)nullptr, ws, (long)rows, C, g.vpb);
  hipLaunchKernelGGL(bn_sums_kernel, dim3(lt_cdiv(C, BN_FC)), dim3(256), 0, ST, ws, g.G, (long)rows, C, sums);
  LT_CHECK_LAUNCH("lt_batchnorm_stats");
}
extern "C" int lt_batchnorm_fwd_from_sums(const void* x, const double* sums, const float* gamma, const float* beta, const void* resid, void* y,
                                          float* mean, float* rstd, float* running_mean, float* running_var, int64_t rows, int C, float eps,
                                          float momentum, int relu, void* stream) {
  LT_CHECK_ARG(x && sums && gamma && beta && y && mean && rstd && rows > 0 && C > 0 && C % 8 == 0 && al16(x) && al16(y) && al16(resid),
               "lt_batchnorm_fwd_from_sums: bad arguments (C=%d)", C);
  hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3(lt_cdiv(C, 256)), dim3(256), 0, ST, sums, C, eps, momentum, mean, rstd, running_mean, running_var);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, ST, (const bf16_t*)x, mean, rstd, gamma, beta, (const bf16_t*)resid,
                     (bf16_t*)y, (long)rows, C, relu);
  LT_CHECK_LAUNCH("lt_batchnorm_fwd_from_sums");
}
extern "C" int lt_batchnorm_bwd_sums(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, void* dz, float* dgamma,
                                     float* dbeta, int64_t rows, int C, float* ws, double* sums, void* stream) {
  LT_CHECK_ARG(dy && x && mean && rstd && ws && sums && rows > 0 && C > 0 && C % 8 == 0, "lt_batchnorm_bwd_sums: bad arguments (C=%d)", C);
  LT_CHECK_ARG(!y || dz, "lt_batchnorm_bwd_sums: a ReLU mask (y) needs a dz buffer for the masked upstream gradient");
  LT_CHECK_ARG(al16(dy) && al16(y) && al16(x) && al16(dz), "lt_batchnorm_bwd_sums: tensors must be 16-byte aligned");
  const BnGeom g = bn_geom(rows, C);
  hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(g.gx, g.G), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y, mean, rstd,
                     (bf16_t*)dz, ws, (long)rows, C, g.vpb);
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(lt_cdiv(C, BN_FC)), dim3(256), 0, ST, ws, g.G, (long)rows, C, dgamma, dbeta, sums);
  LT_CHECK_LAUNCH("lt_batchnorm_bwd_sums");
}
extern "C" int lt_batchnorm_bwd_from_sums(const void* dz, const void* x, const float* gamma, const float* mean, const float* rstd, const double* sums,
                                          void* dx, int64_t rows, int C, float* ws, void* stream) {
  LT_CHECK_ARG(dz && x && gamma && mean && rstd && sums && dx && ws && rows > 0 && C > 0 && C % 8 == 0 && al16(dz) && al16(x) && al16(dx),
               "lt_batchnorm_bwd_from_sums: bad arguments (C=%d)", C);
  float* c1 = ws + 2LL * LT_BN_MAX_CHUNKS * C;
  float* c2 = c1 + C;
  hipLaunchKernelGGL(bn_bwd_coefs_kernel, dim3(lt_cdiv(C, 256)), dim3(256), 0, ST, sums, C, c1, c2);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, ST, (const bf16_t*)dz, (const bf16_t*)x, mean, rstd, gamma, c1, c2,
                     (bf16_t*)dx, (long)rows, C);
  LT_CHECK_LAUNCH("lt_batchnorm_bwd_from_sums");
}

extern "C" int lt_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, void* stream) {
  LT_CHECK_ARG(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && al16(x) && al16(y) && ((uintptr_t)idx & 7) == 0,
               "lt_maxpool3x3s2_fwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long)B * Ho * Wo * (C / 8))), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, idx, B, H, W, C, Ho,
                     Wo);
  LT_CHECK_LAUNCH("lt_maxpool3x3s2_fwd");
}
extern "C" int lt_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int C, void* stream) {
  LT_CHECK_ARG(dy && idx && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && al16(dy) && al16(dx), "lt_maxpool3x3s2_bwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0, ST, (const bf16_t*)dy, idx, (bf16_t*)dx, B, H, W, C, Ho,
                     Wo);
  LT_CHECK_LAUNCH("lt_maxpool3x3s2_bwd");
}
extern "C" int lt_token_mean_bf16(const void* x, void* out, int B, int n, int C, void* stream) {
  LT_CHECK_ARG(x && out && B > 0 && n > 0 && C > 0 && C % 8 == 0 && al16(x) && al16(out), "lt_token_mean_bf16: bad arguments");
  hipLaunchKernelGGL(token_mean_kernel, dim3(grid_for((long)B * (C / 8))), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)out, B, n, C);
  LT_CHECK_LAUNCH("lt_token_mean_bf16");
}
extern "C" int lt_pool_bwd_add(const float* d_tok, const float* d_pool, void* out_bf16, int B, int n, int C, void* stream) {
  LT_CHECK_ARG((d_tok || d_pool) && out_bf16 && B > 0 && n > 0 && C > 0 && C % 4 == 0 && al16(d_tok) && al16(d_pool) && ((uintptr_t)out_bf16 & 7) == 0,
               "lt_pool_bwd_add: bad arguments");
  hipLaunchKernelGGL(pool_bwd_add_kernel, dim3(grid_for((long)B * n * (C / 4))), dim3(256), 0, ST, d_tok, d_pool, (bf16_t*)out_bf16, B, n, C);
  LT_CHECK_LAUNCH("lt_pool_bwd_add");
}
extern "C" int lt_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  LT_CHECK_ARG(a && b && out && n >= 0 && n % 8 == 0 && al16(a) && al16(b) && al16(out), "lt_add_bf16: bad arguments (n must be a multiple of 8)");
  if (n == 0) return LT_OK;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, ST, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (long)(n / 8));
  LT_CHECK_LAUNCH("lt_add_bf16");
}
