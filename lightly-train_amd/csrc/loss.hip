// DINO / iBOT prototype losses over K = 65 536 prototypes (LT/_methods/dinov2/dinov2_loss.py) and KoLeo.
// One 256-thread block per logits row; rows are streamed as float4 with an online (max, sum-exp) so each
// row is read twice (second pass from L2: one row = 256 KiB) instead of three times.
#include "lt_common.h"
#include <cstdlib>

namespace {

struct MaxSum { float m, s; };
__device__ __forceinline__ MaxSum ms_combine(MaxSum a, MaxSum b) {
  const float m = fmaxf(a.m, b.m);
  MaxSum r;
  r.m = m;
  r.s = (a.m == -INFINITY ? 0.f : a.s * __expf(a.m - m)) + (b.m == -INFINITY ? 0.f : b.s * __expf(b.m - m));
  return r;
}
__device__ __forceinline__ void ms_push(MaxSum& a, float z) {
  if (z > a.m) { a.s = a.s * __expf(a.m - z) + 1.f; a.m = z; }
  else a.s += __expf(z - a.m);
}
__device__ __forceinline__ MaxSum block_ms(MaxSum v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxSum other;
    other.m = __shfl_xor(v.m, o, 64);
    other.s = __shfl_xor(v.s, o, 64);
    v = ms_combine(v, other);
  }
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[2 * w] = v.m; red[2 * w + 1] = v.s; }
  __syncthreads();
  MaxSum r; r.m = red[0]; r.s = red[1];
  for (int i = 1; i < nw; ++i) { MaxSum o; o.m = red[2 * i]; o.s = red[2 * i + 1]; r = ms_combine(r, o); }
  return r;
}

// Logit rows in fp32 (the default) or bf16 (round 6, `bf16_logits`: what the reference's own bf16-mixed path holds -- the prototype Linear
// runs under autocast, LT/_methods/dinov2/dinov2_head.py:66-71, and the losses cast back with .float(), dinov2_loss.py:37-38,88): every
// kernel below that streams logits takes the element type as a template parameter and does its arithmetic in fp32 either way.
__device__ __forceinline__ float4 load_logits4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load_logits4(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float load_logit1(const float* p) { return *p; }
__device__ __forceinline__ float load_logit1(const bf16_t* p) { return bf2f(*p); }

// probs = softmax((logits - center) * inv_temp)
__global__ __launch_bounds__(256) void softmax_center_kernel(const float* __restrict__ logits, const float* __restrict__ center,
                                                             float* __restrict__ probs, int K, float inv_temp) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const float* x = logits + row * K;
  MaxSum a; a.m = -INFINITY; a.s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) ms_push(a, (x[k] - (center ? center[k] : 0.f)) * inv_temp);
  a = block_ms(a, red);
  const float inv = 1.f / a.s;
  for (int k = threadIdx.x; k < K; k += 256)
    probs[row * K + k] = __expf((x[k] - (center ? center[k] : 0.f)) * inv_temp - a.m) * inv;
}

// Register-resident variants for K <= 65 536 (K % 4 == 0): 1024 threads hold the whole row (16 float4 each), so the logits
// are read from HBM once instead of twice (the 256-thread kernels re-read the 256-KiB row for their second pass; 32 rows in
// flight per XCD do not fit its 4-MiB L2).
constexpr int ROW_NV = 16;
__global__ __launch_bounds__(1024) void softmax_center_reg_kernel(const float* __restrict__ logits, const float* __restrict__ center,
                                                                  float* __restrict__ probs, int K, float inv_temp) {
  __shared__ float red[32];
  const long row = blockIdx.x;
  const float* x = logits + row * K;
  float4 v[ROW_NV];
  MaxSum a; a.m = -INFINITY; a.s = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int k = (i * 1024 + threadIdx.x) * 4;
    if (k < K) {
      float4 u = *reinterpret_cast<const float4*>(x + k);
      if (center) { const float4 c = *reinterpret_cast<const float4*>(center + k); u.x -= c.x; u.y -= c.y; u.z -= c.z; u.w -= c.w; }
      u.x *= inv_temp; u.y *= inv_temp; u.z *= inv_temp; u.w *= inv_temp;
      v[i] = u;
      ms_push(a, u.x); ms_push(a, u.y); ms_push(a, u.z); ms_push(a, u.w);
    }
  }
  a = block_ms(a, red);
  const float inv = 1.f / a.s;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int k = (i * 1024 + threadIdx.x) * 4;
    if (k < K)
      *reinterpret_cast<float4*>(probs + row * K + k) = make_float4(__expf(v[i].x - a.m) * inv, __expf(v[i].y - a.m) * inv,
                                                                    __expf(v[i].z - a.m) * inv, __expf(v[i].w - a.m) * inv);
  }
}

// Per-row loss terms are stored, then added by ONE workgroup in a fixed order (thread t takes rows t, t + 256, ...; LDS tree): the loss
// scalars are bitwise reproducible (one atomicAdd per row was not).  Up to 8 slots.
constexpr int LOSS_SLOTS = 8;
__global__ __launch_bounds__(256) void rowloss_sum_kernel(const float* __restrict__ terms, const int32_t* __restrict__ slot, int rows,
                                                          float* __restrict__ loss) {
  __shared__ float red[LOSS_SLOTS][256];
  float acc[LOSS_SLOTS];
#pragma unroll
  for (int k = 0; k < LOSS_SLOTS; ++k) acc[k] = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const int sl = slot ? slot[r] : 0;
    const float v = terms[r];
#pragma unroll
    for (int k = 0; k < LOSS_SLOTS; ++k) acc[k] += (sl == k) ? v : 0.f;
  }
#pragma unroll
  for (int k = 0; k < LOSS_SLOTS; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
#pragma unroll
      for (int k = 0; k < LOSS_SLOTS; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < LOSS_SLOTS && red[threadIdx.x][0] != 0.f) loss[threadIdx.x] += red[threadIdx.x][0];
}

__global__ __launch_bounds__(1024) void ce_reg_kernel(const float* __restrict__ s, const float* __restrict__ teacher,
                                                      const int32_t* __restrict__ ta, const int32_t* __restrict__ tb,
                                                      const float* __restrict__ row_weight, const int32_t* __restrict__ slot,
                                                      float scale, float inv_temp, float* __restrict__ terms,
                                                      bf16_t* __restrict__ dlogits, int K) {
  __shared__ float red[32];
  const long row = blockIdx.x;
  const float* z = s + row * K;
  const float* t0 = teacher + (long)ta[row] * K;
  const float* t1 = (tb && tb[row] >= 0) ? teacher + (long)tb[row] * K : nullptr;
  float4 v[ROW_NV];
  MaxSum a; a.m = -INFINITY; a.s = 0.f;
  float dot = 0.f, tsum = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int k = (i * 1024 + threadIdx.x) * 4;
    if (k < K) {
      float4 u = *reinterpret_cast<const float4*>(z + k);
      u.x *= inv_temp; u.y *= inv_temp; u.z *= inv_temp; u.w *= inv_temp;
      v[i] = u;
      float4 t = *reinterpret_cast<const float4*>(t0 + k);
      if (t1) { const float4 t2 = *reinterpret_cast<const float4*>(t1 + k); t.x += t2.x; t.y += t2.y; t.z += t2.z; t.w += t2.w; }
      ms_push(a, u.x); ms_push(a, u.y); ms_push(a, u.z); ms_push(a, u.w);
      dot += t.x * u.x + t.y * u.y + t.z * u.z + t.w * u.w;
      tsum += t.x + t.y + t.z + t.w;
    }
  }
  a = block_ms(a, red);
  dot = block_sum(dot, red);
  tsum = block_sum(tsum, red);
  const float lse = a.m + __logf(a.s);
  const float coef = scale * (row_weight ? row_weight[row] : 1.f);
  if (threadIdx.x == 0) terms[row] = -coef * (dot - lse * tsum);
  if (dlogits) {
    const float c2 = coef * inv_temp;
#pragma unroll
    for (int i = 0; i < ROW_NV; ++i) {
      const int k = (i * 1024 + threadIdx.x) * 4;
      if (k < K) {
        float4 t = *reinterpret_cast<const float4*>(t0 + k);   // second read of the teacher row(s): L2 / Infinity Cache
        if (t1) { const float4 t2 = *reinterpret_cast<const float4*>(t1 + k); t.x += t2.x; t.y += t2.y; t.z += t2.z; t.w += t2.w; }
        const float4 u = v[i];
        *reinterpret_cast<uint2*>(dlogits + row * K + k) =
            make_uint2(pack_bf2(c2 * (__expf(u.x - lse) * tsum - t.x), c2 * (__expf(u.y - lse) * tsum - t.y)),
                       pack_bf2(c2 * (__expf(u.z - lse) * tsum - t.z), c2 * (__expf(u.w - lse) * tsum - t.w)));
      }
    }
  }
}

// ---- softmax centering without the probability matrix ------------------------------------------------------------------------------
// The teacher side of DINOLoss / IBOTPatchLoss with center_method="softmax" (dinov2_loss.py:76-82,139-160 / :178-186,274-297) needs, per
// teacher row, softmax((x - center) / temp) inside the student's cross-entropy, and the column sums of the raw logits for the center
// update.  Written as three passes (probabilities out, column sums, CE in) that is 2 GB of fp32 written and read twice more for
// [7.8 k, 65 536] rows; here ONE pass over the teacher logits leaves the row statistics (max, 1 / sum-exp) and the column sums, and the
// cross-entropy kernel below rebuilds the probabilities from the logits it reads anyway.
//
// 1024 threads own the 65 536 columns (thread t: columns (i * 1024 + t) * 4, i < 16) and walk rows blockIdx.x, + gridDim.x, ...: the
// column sums stay in registers over the block's rows (partial[blockIdx.x][K] at the end, added in block order by colsum_slabs_kernel),
// the row statistics take one wave reduction + one barrier per row (combined by wave 0 from a double-buffered LDS slot).
constexpr int SC_CHUNK = 4;   // float4 loads in flight per thread and chunk
constexpr int SC_GROUP = 8;   // rows between two barriers
template <typename TL>
__global__ __launch_bounds__(1024) void softmax_stats_colsum_kernel(const TL* __restrict__ logits, const float* __restrict__ center,
                                                                    float* __restrict__ stats, float* __restrict__ partial, int rows, int K,
                                                                    float inv_temp) {
  // wave partials (max, sum-exp) of SC_GROUP rows, double-buffered: ONE barrier per SC_GROUP rows (a barrier per row drained the
  // memory pipeline at every row boundary: 4.2 TB/s); wave 0 combines group g while every wave already streams group g + 1
  __shared__ float red[2][SC_GROUP][32];
  float4 cs[ROW_NV];
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) cs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const long nmine = rows > (long)blockIdx.x ? (rows - 1 - blockIdx.x) / gridDim.x + 1 : 0;   // rows this workgroup walks
  for (long it = 0; it < nmine; ++it) {
    const long row = blockIdx.x + it * gridDim.x;
    const int slot = (int)(it % SC_GROUP), par = (int)((it / SC_GROUP) & 1);
    const TL* x = logits + row * K;
    MaxSum a; a.m = -INFINITY; a.s = 0.f;
#pragma unroll
    for (int j = 0; j < ROW_NV; j += SC_CHUNK) {
      float4 u[SC_CHUNK], c[SC_CHUNK];
#pragma unroll
      for (int i = 0; i < SC_CHUNK; ++i) {
        const int k = ((j + i) * 1024 + threadIdx.x) * 4;
        u[i] = make_float4(0.f, 0.f, 0.f, 0.f); c[i] = u[i];
        if (k < K) {
          u[i] = load_logits4(x + k);
          if (center) c[i] = *reinterpret_cast<const float4*>(center + k);
        }
      }
      // chunk maximum first, then ONE exponential per element against it (ms_push would run both of its branches -- an exponential each --
      // for nearly every element: with 64 elements per thread some lane of the wave meets a new maximum almost every time)
      float cm = -INFINITY;
#pragma unroll
      for (int i = 0; i < SC_CHUNK; ++i) {
        const int k = ((j + i) * 1024 + threadIdx.x) * 4;
        if (k < K) {
          cs[j + i].x += u[i].x; cs[j + i].y += u[i].y; cs[j + i].z += u[i].z; cs[j + i].w += u[i].w;
          u[i].x = (u[i].x - c[i].x) * inv_temp; u[i].y = (u[i].y - c[i].y) * inv_temp;
          u[i].z = (u[i].z - c[i].z) * inv_temp; u[i].w = (u[i].w - c[i].w) * inv_temp;
          cm = fmaxf(cm, fmaxf(fmaxf(u[i].x, u[i].y), fmaxf(u[i].z, u[i].w)));
        }
      }
      if (cm > -INFINITY) {
        float csum_e = 0.f;
#pragma unroll
        for (int i = 0; i < SC_CHUNK; ++i) {
          const int k = ((j + i) * 1024 + threadIdx.x) * 4;
          if (k < K) csum_e += (__expf(u[i].x - cm) + __expf(u[i].y - cm)) + (__expf(u[i].z - cm) + __expf(u[i].w - cm));
        }
        MaxSum b; b.m = cm; b.s = csum_e;
        a = ms_combine(a, b);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      MaxSum other;
      other.m = __shfl_xor(a.m, o, 64);
      other.s = __shfl_xor(a.s, o, 64);
      a = ms_combine(a, other);
    }
    if (l == 0) { red[par][slot][2 * w] = a.m; red[par][slot][2 * w + 1] = a.s; }
    if (slot == SC_GROUP - 1 || it == nmine - 1) {
      __syncthreads();   // the group's partials are visible; also fences buffer `par` against its re-use two groups on
      if (w == 0) {      // lanes: 4 rows x 16 wave partials per pass
        const long first = it - slot;
        for (int r0 = 0; r0 <= slot; r0 += 4) {
          const int r = r0 + (l >> 4), ww = l & 15;
          MaxSum rr; rr.m = -INFINITY; rr.s = 0.f;
          if (r <= slot) { rr.m = red[par][r][2 * ww]; rr.s = red[par][r][2 * ww + 1]; }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            MaxSum other;
            other.m = __shfl_xor(rr.m, o, 64);
            other.s = __shfl_xor(rr.s, o, 64);
            rr = ms_combine(rr, other);
          }
          if (ww == 0 && r <= slot) {
            const long orow = blockIdx.x + (first + r) * gridDim.x;
            stats[2 * orow] = rr.m; stats[2 * orow + 1] = 1.f / rr.s;
          }
        }
      }
    }
  }
  float* p = partial + (size_t)blockIdx.x * K;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int k = (i * 1024 + threadIdx.x) * 4;
    if (k < K) *reinterpret_cast<float4*>(p + k) = cs[i];
  }
}
// out[c] = sum over slabs of partial[s][c] in a fixed order (the sums feed the loss centers, which must not depend on scheduling): 64 column
// lanes x 4 slab lanes per workgroup, slab lane q adds slabs q, q + 4, ... (four loads in flight), the lanes combine as ((0 + 1) + 2) + 3
__global__ __launch_bounds__(256) void colsum_slabs_f32_kernel(const float* __restrict__ partial, float* __restrict__ out, int slabs, int N) {
  __shared__ float4 red[4][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (c < N) {
    int s = q;
    for (; s + 12 < slabs; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(partial + (size_t)s * N + c);
      const float4 v1 = *reinterpret_cast<const float4*>(partial + (size_t)(s + 4) * N + c);
      const float4 v2 = *reinterpret_cast<const float4*>(partial + (size_t)(s + 8) * N + c);
      const float4 v3 = *reinterpret_cast<const float4*>(partial + (size_t)(s + 12) * N + c);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; s < slabs; s += 4) {
      const float4 v0 = *reinterpret_cast<const float4*>(partial + (size_t)s * N + c);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
  }
  red[q][cl] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (q == 0 && c < N) {
    const float4 s0 = red[0][cl], s1 = red[1][cl], s2 = red[2][cl], s3 = red[3][cl];
    *reinterpret_cast<float4*>(out + c) = make_float4(((s0.x + s1.x) + s2.x) + s3.x, ((s0.y + s1.y) + s2.y) + s3.y,
                                                      ((s0.z + s1.z) + s2.z) + s3.z, ((s0.w + s1.w) + s2.w) + s3.w);
  }
}
// generic widths: one 256-thread block per row (statistics only; the column sums come from lt_colsum_f32)
template <typename TL>
__global__ __launch_bounds__(256) void softmax_stats_kernel(const TL* __restrict__ logits, const float* __restrict__ center,
                                                            float* __restrict__ stats, int K, float inv_temp) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const TL* x = logits + row * K;
  MaxSum a; a.m = -INFINITY; a.s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) ms_push(a, (load_logit1(x + k) - (center ? center[k] : 0.f)) * inv_temp);
  a = block_ms(a, red);
  if (threadIdx.x == 0) { stats[2 * row] = a.m; stats[2 * row + 1] = 1.f / a.s; }
}

// teacher probability of column k of teacher row `tr`, rebuilt from its logits: exp((x - center) * inv_temp_t - max) / sum-exp
template <typename TL> struct TRow { const TL* x; const float* c; float m, is; };
template <typename TL>
__device__ __forceinline__ TRow<TL> trow(const TL* __restrict__ t_logits, const float* __restrict__ stats, const float* ca, const float* cb,
                                         int split, int tr, int K) {
  TRow<TL> r;
  r.x = t_logits + (long)tr * K;
  r.c = tr < split ? ca : cb;
  r.m = stats[2 * tr];
  r.is = stats[2 * tr + 1];
  return r;
}
template <typename TL>
__device__ __forceinline__ float4 tprob4(const TRow<TL>& r, int k, float itt) {
  const float4 x = load_logits4(r.x + k);
  float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r.c) c = *reinterpret_cast<const float4*>(r.c + k);
  return make_float4(__expf((x.x - c.x) * itt - r.m) * r.is, __expf((x.y - c.y) * itt - r.m) * r.is,
                     __expf((x.z - c.z) * itt - r.m) * r.is, __expf((x.w - c.w) * itt - r.m) * r.is);
}
template <typename TL>
__device__ __forceinline__ float tprob1(const TRow<TL>& r, int k, float itt) {
  return __expf((load_logit1(r.x + k) - (r.c ? r.c[k] : 0.f)) * itt - r.m) * r.is;
}

// ce_reg_kernel with the teacher probabilities rebuilt from (teacher logits, row statistics, center): same arithmetic per element as
// softmax_center_reg_kernel followed by ce_reg_kernel, without the probability matrix in between
template <typename TL>
__global__ __launch_bounds__(1024) void ce_logits_reg_kernel(const TL* __restrict__ s, const TL* __restrict__ t_logits,
                                                             const float* __restrict__ t_stats, const float* __restrict__ center_a,
                                                             const float* __restrict__ center_b, int split, const int32_t* __restrict__ ta,
                                                             const int32_t* __restrict__ tb, const float* __restrict__ row_weight,
                                                             float scale, float inv_temp, float inv_temp_t, float* __restrict__ terms,
                                                             bf16_t* __restrict__ dlogits, int K) {
  __shared__ float red[32];
  const long row = blockIdx.x;
  const TL* z = s + row * K;
  const TRow<TL> r0 = trow(t_logits, t_stats, center_a, center_b, split, ta[row], K);
  const bool two = tb && tb[row] >= 0;
  const TRow<TL> r1 = trow(t_logits, t_stats, center_a, center_b, split, two ? tb[row] : ta[row], K);
  // The student row stays in registers, so its softmax takes ONE exponential per element: row maximum first (64 v_max per thread + one
  // block reduction), then e = exp(u - max) overwrites u -- its sum gives the log-sum-exp, and e / sum is the probability the gradient
  // needs.  (An online max / sum-exp pushes nearly every element through both of its branches, an exponential each, and the gradient
  // pass then pays a third.)
  float4 v[ROW_NV];
  float mx = -INFINITY, dot = 0.f, tsum = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int k = (i * 1024 + threadIdx.x) * 4;
    if (k < K) {
      float4 u = load_logits4(z + k);
      u.x *= inv_temp; u.y *= inv_temp; u.z *= inv_temp; u.w *= inv_temp;
      v[i] = u;
      float4 t = tprob4(r0, k, inv_temp_t);
      if (two) { const float4 t2 = tprob4(r1, k, inv_temp_t); t.x += t2.x; t.y += t2.y; t.z += t2.z; t.w += t2.w; }
      mx = fmaxf(mx, fmaxf(fmaxf(u.x, u.y), fmaxf(u.z, u.w)));
      dot += t.x * u.x + t.y * u.y + t.z * u.z + t.w * u.w;
      tsum += t.x + t.y + t.z + t.w;
    }
  }
  mx = block_max(mx, red);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int k = (i * 1024 + threadIdx.x) * 4;
    if (k < K) {
      float4 e = v[i];
      e.x = __expf(e.x - mx); e.y = __expf(e.y - mx); e.z = __expf(e.z - mx); e.w = __expf(e.w - mx);
      v[i] = e;
      se += (e.x + e.y) + (e.z + e.w);
    }
  }
  se = block_sum(se, red);
  dot = block_sum(dot, red);
  tsum = block_sum(tsum, red);
  const float lse = mx + __logf(se);
  const float coef = scale * (row_weight ? row_weight[row] : 1.f);
  if (threadIdx.x == 0) terms[row] = -coef * (dot - lse * tsum);
  if (dlogits) {
    const float c2 = coef * inv_temp;
    const float pn = tsum / se;             // softmax(u) * sum(t) = e * (sum(t) / sum(e))
#pragma unroll
    for (int i = 0; i < ROW_NV; ++i) {
      const int k = (i * 1024 + threadIdx.x) * 4;
      if (k < K) {
        float4 t = tprob4(r0, k, inv_temp_t);   // second read of the teacher row(s): L2 / Infinity Cache
        if (two) { const float4 t2 = tprob4(r1, k, inv_temp_t); t.x += t2.x; t.y += t2.y; t.z += t2.z; t.w += t2.w; }
        const float4 e = v[i];
        *reinterpret_cast<uint2*>(dlogits + row * K + k) =
            make_uint2(pack_bf2(c2 * (e.x * pn - t.x), c2 * (e.y * pn - t.y)), pack_bf2(c2 * (e.z * pn - t.z), c2 * (e.w * pn - t.w)));
      }
    }
  }
}
template <typename TL>
__global__ __launch_bounds__(256) void ce_logits_kernel(const TL* __restrict__ s, const TL* __restrict__ t_logits,
                                                        const float* __restrict__ t_stats, const float* __restrict__ center_a,
                                                        const float* __restrict__ center_b, int split, const int32_t* __restrict__ ta,
                                                        const int32_t* __restrict__ tb, const float* __restrict__ row_weight, float scale,
                                                        float inv_temp, float inv_temp_t, float* __restrict__ terms,
                                                        bf16_t* __restrict__ dlogits, int K) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const TL* z = s + row * K;
  const TRow<TL> r0 = trow(t_logits, t_stats, center_a, center_b, split, ta[row], K);
  const bool two = tb && tb[row] >= 0;
  const TRow<TL> r1 = trow(t_logits, t_stats, center_a, center_b, split, two ? tb[row] : ta[row], K);
  MaxSum a; a.m = -INFINITY; a.s = 0.f;
  float dot = 0.f, tsum = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float zk = load_logit1(z + k) * inv_temp;
    const float tk = tprob1(r0, k, inv_temp_t) + (two ? tprob1(r1, k, inv_temp_t) : 0.f);
    ms_push(a, zk);
    dot += tk * zk;
    tsum += tk;
  }
  a = block_ms(a, red);
  dot = block_sum(dot, red);
  tsum = block_sum(tsum, red);
  const float lse = a.m + __logf(a.s);
  const float coef = scale * (row_weight ? row_weight[row] : 1.f);
  if (threadIdx.x == 0) terms[row] = -coef * (dot - lse * tsum);
  if (dlogits) {
    const float c2 = coef * inv_temp;
    for (int k = threadIdx.x; k < K; k += 256) {
      const float zk = load_logit1(z + k) * inv_temp;
      const float tk = tprob1(r0, k, inv_temp_t) + (two ? tprob1(r1, k, inv_temp_t) : 0.f);
      dlogits[row * K + k] = f2bf(c2 * (__expf(zk - lse) * tsum - tk));
    }
  }
}

__global__ void center_ema_kernel(float* center, const float* colsum, float scale, float momentum, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) center[k] = center[k] * momentum + colsum[k] * scale * (1.f - momentum);
}

// fused CE forward + d(logits)
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ s, const float* __restrict__ teacher,
                                                 const int32_t* __restrict__ ta, const int32_t* __restrict__ tb,
                                                 const float* __restrict__ row_weight, const int32_t* __restrict__ slot,
                                                 float scale, float inv_temp, float* __restrict__ terms,
                                                 bf16_t* __restrict__ dlogits, int K) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const float* z = s + row * K;
  const float* t0 = teacher + (long)ta[row] * K;
  const float* t1 = (tb && tb[row] >= 0) ? teacher + (long)tb[row] * K : nullptr;
  MaxSum a; a.m = -INFINITY; a.s = 0.f;
  float dot = 0.f, tsum = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float zk = z[k] * inv_temp;
    const float tk = t0[k] + (t1 ? t1[k] : 0.f);
    ms_push(a, zk);
    dot += tk * zk;
    tsum += tk;
  }
  a = block_ms(a, red);
  dot = block_sum(dot, red);
  tsum = block_sum(tsum, red);
  const float lse = a.m + __logf(a.s);
  const float coef = scale * (row_weight ? row_weight[row] : 1.f);
  if (threadIdx.x == 0) terms[row] = -coef * (dot - lse * tsum);
  if (dlogits) {
    const float c2 = coef * inv_temp;
    for (int k = threadIdx.x; k < K; k += 256) {
      const float zk = z[k] * inv_temp;
      const float tk = t0[k] + (t1 ? t1[k] : 0.f);
      dlogits[row * K + k] = f2bf(c2 * (__expf(zk - lse) * tsum - tk));
    }
  }
}

// ------------------------------------------------------------------------------------ distillation (KL on similarity logits)
// One block per row: KL(softmax(t/T) || softmax(s/T)) = sum_k t_k (log t_k - log p_k), d(s_logit) = coef/T * (p_k - t_k)
// (DistillationV3Loss, LT/_methods/distillationv3/distillationv3_loss.py:60-115: KLDivLoss(batchmean) of a log_softmax
// student against a softmax teacher).  Rows are `ld` floats apart (>= K; the similarity matrices are padded to 8 columns).
__global__ __launch_bounds__(256) void kl_kernel(const float* __restrict__ s, const float* __restrict__ t, int ld, float inv_temp,
                                                 float coef, float* __restrict__ terms, bf16_t* __restrict__ dlogits, int ldd, int K) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const float* zs = s + row * ld;
  const float* zt = t + row * ld;
  MaxSum a, b; a.m = b.m = -INFINITY; a.s = b.s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) { ms_push(a, zs[k] * inv_temp); ms_push(b, zt[k] * inv_temp); }
  a = block_ms(a, red);
  b = block_ms(b, red);
  const float lse_s = a.m + __logf(a.s), lse_t = b.m + __logf(b.s);
  float kl = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float ls = zs[k] * inv_temp - lse_s, lt = zt[k] * inv_temp - lse_t;
    const float tk = __expf(lt);
    kl += tk * (lt - ls);
    if (dlogits) dlogits[row * ldd + k] = f2bf(coef * inv_temp * (__expf(ls) - tk));
  }
  kl = block_sum(kl, red);
  if (threadIdx.x == 0) terms[row] = coef * kl;
}
extern "C" int lt_kl_fwd_bwd(const float* s_logits, const float* t_logits, int ld, float inv_temp, float coef, float* loss,
                             void* dlogits_bf16, int ldd, int rows, int K, void* stream) {
  LT_CHECK_ARG(s_logits && t_logits && loss && K > 0 && ld >= K && (!dlogits_bf16 || ldd >= K), "lt_kl_fwd_bwd: bad arguments");
  if (rows == 0) return LT_OK;
  float* terms = lt_scratch_ring((size_t)rows);
  if (!terms) { lt_set_error("lt_kl_fwd_bwd: scratch allocation failed"); return LT_ERR_HIP; }
  hipLaunchKernelGGL(kl_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, s_logits, t_logits, ld, inv_temp, coef, terms, (bf16_t*)dlogits_bf16, ldd, K);
  hipLaunchKernelGGL(rowloss_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, terms, (const int32_t*)nullptr, rows, loss);
  LT_CHECK_LAUNCH("lt_kl_fwd_bwd");
}

// g[b][i][j] = d[b][i][j] + d[b][j][i]  (gradient of X X^T w.r.t. X is (dS + dS^T) X); square n x n matrices, row stride ld
__global__ __launch_bounds__(256) void symmetrize_kernel(const bf16_t* __restrict__ d, bf16_t* __restrict__ g, int n, int ld, long per) {
  const long base = (long)blockIdx.y * per;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n * n; idx += gridDim.x * 256) {
    const int i = idx / n, j = idx - i * n;
    g[base + (long)i * ld + j] = f2bf(bf2f(d[base + (long)i * ld + j]) + bf2f(d[base + (long)j * ld + i]));
  }
}
extern "C" int lt_symmetrize_bf16(const void* d, void* g, int batch, int n, int ld, void* stream) {
  LT_CHECK_ARG(d && g && d != g && batch > 0 && n > 0 && ld >= n, "lt_symmetrize_bf16: bad arguments (in-place is not supported)");
  hipLaunchKernelGGL(symmetrize_kernel, dim3(min(lt_cdiv(n * n, 256), 64), batch), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d, (bf16_t*)g, n, ld,
                     (long)n * ld);
  LT_CHECK_LAUNCH("lt_symmetrize_bf16");
}

// mixup (DistillationV3._mixup_data, distillationv3.py:356-368): out[b] = lam * x[b] + (1 - lam) * x[index[b]]
__global__ __launch_bounds__(256) void mixup_kernel(const float* __restrict__ x, const int64_t* __restrict__ index, float lam,
                                                    float* __restrict__ out, long per) {
  const long b = blockIdx.y;
  const float* a = x + b * per;
  const float* c = x + index[b] * per;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < per; i += (long)gridDim.x * 1024) {
    if (i + 3 < per) {
      const float4 u = *reinterpret_cast<const float4*>(a + i), v = *reinterpret_cast<const float4*>(c + i);
      *reinterpret_cast<float4*>(out + b * per + i) = make_float4(lam * u.x + (1.f - lam) * v.x, lam * u.y + (1.f - lam) * v.y,
                                                                  lam * u.z + (1.f - lam) * v.z, lam * u.w + (1.f - lam) * v.w);
    } else {
      for (long j = i; j < per; ++j) out[b * per + j] = lam * a[j] + (1.f - lam) * c[j];
    }
  }
}
extern "C" int lt_mixup(const float* x, const int64_t* index, float lam, float* out, int B, int64_t per_image, void* stream) {
  LT_CHECK_ARG(x && index && out && x != out && B > 0 && per_image > 0 && per_image % 4 == 0, "lt_mixup: bad arguments (per-image size must be a multiple of 4)");
  hipLaunchKernelGGL(mixup_kernel, dim3((unsigned)min((long)256, (long)lt_cdiv(per_image, 1024)), B), dim3(256), 0, (hipStream_t)stream, x, index, lam, out,
                     (long)per_image);
  LT_CHECK_LAUNCH("lt_mixup");
}

__global__ void sk_exp_kernel(const float* __restrict__ x, float* __restrict__ q, long n, float inv_temp) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) q[i] = __expf(x[i] * inv_temp);
}
__global__ __launch_bounds__(256) void sk_iter_kernel(float* __restrict__ Q, const float* __restrict__ colsum, int K, float n_total,
                                                      float final_mul) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  float* q = Q + row * K;
  const float invK = 1.f / (float)K;
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) s += q[k] / colsum[k] * invK;
  s = block_sum(s, red);
  const float mul = final_mul / (s * n_total);
  for (int k = threadIdx.x; k < K; k += 256) q[k] = q[k] / colsum[k] * invK * mul;
}

// ------------------------------------------------------------------------------------ KoLeo
// ws layout: xn [n*D], inv_norm [n], coef [n]
__global__ __launch_bounds__(256) void koleo_normalize_kernel(const float* __restrict__ x, int ld, float* __restrict__ xn,
                                                              float* __restrict__ inv, int D, float eps) {
  __shared__ float red[16];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) { const float a = x[(long)i * ld + d]; s += a * a; }
  s = block_sum(s, red);
  const float iv = 1.f / fmaxf(sqrtf(s), eps);
  if (threadIdx.x == 0) inv[i] = iv;
  for (int d = threadIdx.x; d < D; d += 256) xn[(long)i * D + d] = x[(long)i * ld + d] * iv;
}
// one block per row i: nearest neighbour by max cosine (diag excluded), distance, loss, per-row coefficient
__global__ __launch_bounds__(256) void koleo_nn_kernel(const float* __restrict__ xn, int32_t* __restrict__ nn, float* __restrict__ coef,
                                                       float* __restrict__ term, int n, int D, float eps, float weight) {
  __shared__ float sval[256];
  __shared__ int sidx[256];
  __shared__ float red[16];
  const int i = blockIdx.x;
  float best = -INFINITY; int bj = 0x7fffffff;
  for (int j = threadIdx.x; j < n; j += 256) {
    if (j == i) continue;
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot = fmaf(xn[(long)i * D + d], xn[(long)j * D + d], dot);
    if (dot > best) { best = dot; bj = j; }
  }
  sval[threadIdx.x] = best; sidx[threadIdx.x] = bj;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float v2 = sval[threadIdx.x + o]; const int i2 = sidx[threadIdx.x + o];
      if (v2 > sval[threadIdx.x] || (v2 == sval[threadIdx.x] && i2 < sidx[threadIdx.x])) { sval[threadIdx.x] = v2; sidx[threadIdx.x] = i2; }
    }
    __syncthreads();
  }
  const int j = sidx[0];
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) { const float u = xn[(long)i * D + d] - xn[(long)j * D + d] + eps; s += u * u; }
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float dist = sqrtf(s);
    nn[i] = j;
    // L = -(1/n) sum log(dist + eps);  dL/du = -(1/n) * 1/(dist+eps) * u/dist
    coef[i] = -weight / ((float)n * (dist + eps) * fmaxf(dist, 1e-30f));
    term[i] = -(weight == 0.f ? 1.f : weight) * __logf(dist + eps) / (float)n;   // weight 0: report the unweighted term
  }
}
// the n row terms added in row order by one thread (n = rows of one crop chunk: tens to a few hundred)
__global__ void koleo_loss_kernel(const float* __restrict__ term, int n, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += term[i];
    *loss += s;
  }
}
// dxn[r] = c_r u_r - sum_{i: nn(i) = r} c_i u_i, u_i = xn_i - xn_nn(i) + eps.  One block per row r gathers its contributions in
// ascending i (several rows may share a neighbour: scattering them needed atomics and left the order to chance).
__global__ __launch_bounds__(256) void koleo_dxn_kernel(const float* __restrict__ xn, const int32_t* __restrict__ nn,
                                                        const float* __restrict__ coef, float* __restrict__ dxn, int n, int D, float eps) {
  const int r = blockIdx.x, j = nn[r];
  const float c = coef[r];
  for (int d = threadIdx.x; d < D; d += 256) {
    const float xr = xn[(long)r * D + d];
    float acc = c * (xr - xn[(long)j * D + d] + eps);
    for (int i = 0; i < n; ++i)       // nn / coef: n small, broadcast reads out of L1
      if (nn[i] == r) acc -= coef[i] * (xn[(long)i * D + d] - xr + eps);
    dxn[(long)r * D + d] = acc;
  }
}
// through x/max(||x||,eps): dx += (dxn - xn*(xn.dxn)) * inv
__global__ __launch_bounds__(256) void koleo_dx_kernel(const float* __restrict__ xn, const float* __restrict__ dxn,
                                                       const float* __restrict__ inv, float* __restrict__ dx, int ld_dx, int D) {
  __shared__ float red[16];
  const int i = blockIdx.x;
  float dot = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) dot += xn[(long)i * D + d] * dxn[(long)i * D + d];
  dot = block_sum(dot, red);
  const float iv = inv[i];
  for (int d = threadIdx.x; d < D; d += 256)
    dx[(long)i * ld_dx + d] += (dxn[(long)i * D + d] - xn[(long)i * D + d] * dot) * iv;
}

// ------------------------------------------------------------------------------------ MSE (DistillationV2Loss = nn.MSELoss(), mean over all elements)
// loss += scale * sum (s - t)^2,  ds = 2 * scale * (s - t); two-level deterministic sum (per-block partials, last block adds them in order)
constexpr int MSE_MAX_GRID = 1024;
// per-block partial sums (already scaled) into caller-provided scratch; rowloss_sum_kernel adds them in a fixed order.  The scratch comes
// from the library's ring, so launches in flight on different streams do not share it.
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ s, const float* __restrict__ t, float* __restrict__ ds, long n, float scale,
                                                  float* __restrict__ partials) {
  __shared__ float red[16];
  float acc = 0.f;
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    const float4 a = *reinterpret_cast<const float4*>(s + i), b = *reinterpret_cast<const float4*>(t + i);
    const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    acc += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    if (ds) *reinterpret_cast<float4*>(ds + i) = make_float4(2.f * scale * d.x, 2.f * scale * d.y, 2.f * scale * d.z, 2.f * scale * d.w);
  }
  for (long j = i; j < n && j < i + 4; ++j) {
    const float d = s[j] - t[j];
    acc += d * d;
    if (ds) ds[j] = 2.f * scale * d;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = scale * acc;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int lt_softmax_center(const float* logits, const float* center, float* probs, int rows, int K, float inv_temp, void* stream) {
  LT_CHECK_ARG(logits && probs && K > 0, "lt_softmax_center: bad arguments");
  if (rows == 0) return LT_OK;
  static const int reg_rows = [] { const char* e = getenv("LT_LOSS_REG"); return e ? atoi(e) : 1; }();
  if (reg_rows && K % 4 == 0 && K <= 4096 * ROW_NV && K >= 8192 && ((uintptr_t)logits % 16 == 0) && ((uintptr_t)probs % 16 == 0) &&
      (!center || (uintptr_t)center % 16 == 0))
    hipLaunchKernelGGL(softmax_center_reg_kernel, dim3(rows), dim3(1024), 0, ST, logits, center, probs, K, inv_temp);
  else
    hipLaunchKernelGGL(softmax_center_kernel, dim3(rows), dim3(256), 0, ST, logits, center, probs, K, inv_temp);
  LT_CHECK_LAUNCH("lt_softmax_center");
}
extern "C" int lt_center_ema(float* center, const float* colsum, float scale, float momentum, int K, void* stream) {
  LT_CHECK_ARG(center && colsum, "lt_center_ema: null pointer");
  hipLaunchKernelGGL(center_ema_kernel, dim3(lt_cdiv(K, 256)), dim3(256), 0, ST, center, colsum, scale, momentum, K);
  LT_CHECK_LAUNCH("lt_center_ema");
}
extern "C" int lt_ce_fwd_bwd(const float* s, const float* teacher, const int32_t* ta, const int32_t* tb, const float* row_weight,
                             const int32_t* slot, float scale, float inv_temp, float* loss, void* dlogits_bf16, int rows, int K,
                             void* stream) {
  LT_CHECK_ARG(s && teacher && ta && loss && K > 0, "lt_ce_fwd_bwd: bad arguments");
  if (rows == 0) return LT_OK;
  float* terms = lt_scratch_ring((size_t)rows);
  if (!terms) { lt_set_error("lt_ce_fwd_bwd: scratch allocation failed"); return LT_ERR_HIP; }
  static const int reg_rows = [] { const char* e = getenv("LT_LOSS_REG"); return e ? atoi(e) : 1; }();
  if (reg_rows && K % 4 == 0 && K <= 4096 * ROW_NV && K >= 8192 && ((uintptr_t)s % 16 == 0) && ((uintptr_t)teacher % 16 == 0) &&
      (!dlogits_bf16 || (uintptr_t)dlogits_bf16 % 8 == 0))
    hipLaunchKernelGGL(ce_reg_kernel, dim3(rows), dim3(1024), 0, ST, s, teacher, ta, tb, row_weight, slot, scale, inv_temp, terms,
                       (bf16_t*)dlogits_bf16, K);
  else
    hipLaunchKernelGGL(ce_kernel, dim3(rows), dim3(256), 0, ST, s, teacher, ta, tb, row_weight, slot, scale, inv_temp, terms,
                       (bf16_t*)dlogits_bf16, K);
  hipLaunchKernelGGL(rowloss_sum_kernel, dim3(1), dim3(256), 0, ST, terms, slot, rows, loss);
  LT_CHECK_LAUNCH("lt_ce_fwd_bwd");
}
extern "C" int lt_colsum_f32(const float* x, float* out, int rows, int N, int accumulate, void* stream);
extern "C" int lt_colsum_bf16(const void* x_bf16, float* out, int rows, int N, void* stream);
namespace {
template <typename TL>
int softmax_stats_colsum_impl(const TL* logits, const float* center, float* stats, float* colsum, int rows, int K, float inv_temp, float* scratch,
                              int64_t scratch_floats, void* stream) {
  LT_CHECK_ARG(colsum && K > 0 && rows >= 0 && (rows == 0 || (logits && stats)), "lt_softmax_stats_colsum: bad arguments");
  if (rows == 0) {
    if (hipMemsetAsync(colsum, 0, sizeof(float) * K, ST) != hipSuccess) { lt_set_error("lt_softmax_stats_colsum: memset failed"); return LT_ERR_HIP; }
    return LT_OK;
  }
  static const int reg_rows = [] { const char* e = getenv("LT_LOSS_REG"); return e ? atoi(e) : 1; }();
  if (reg_rows && K % 4 == 0 && K <= 4096 * ROW_NV && K >= 8192 && ((uintptr_t)logits % 16 == 0) && (!center || (uintptr_t)center % 16 == 0) &&
      ((uintptr_t)colsum % 16 == 0)) {
    int grid = rows < 256 ? rows : 256;           // one workgroup per CU; each keeps its rows' column sums in registers
    LT_CHECK_ARG(scratch && scratch_floats >= K && ((uintptr_t)scratch & 15) == 0, "lt_softmax_stats_colsum: needs >= K floats of 16-byte aligned scratch");
    if ((int64_t)grid * K > scratch_floats) grid = (int)(scratch_floats / K);   // fewer, longer row walks when the caller's scratch is small
    float* partial = scratch;
    hipLaunchKernelGGL(softmax_stats_colsum_kernel<TL>, dim3(grid), dim3(1024), 0, ST, logits, center, stats, partial, rows, K, inv_temp);
    hipLaunchKernelGGL(colsum_slabs_f32_kernel, dim3(lt_cdiv(K, 256)), dim3(256), 0, ST, partial, colsum, grid, K);
    LT_CHECK_LAUNCH("lt_softmax_stats_colsum");
  }
  hipLaunchKernelGGL(softmax_stats_kernel<TL>, dim3(rows), dim3(256), 0, ST, logits, center, stats, K, inv_temp);
  if constexpr (sizeof(TL) == 4) return lt_colsum_f32((const float*)logits, colsum, rows, K, 0, stream);
  else {
    LT_CHECK_ARG(K % 8 == 0, "lt_softmax_stats_colsum_bf16: K must be a multiple of 8 on the generic path");
    if (hipMemsetAsync(colsum, 0, sizeof(float) * K, ST) != hipSuccess) { lt_set_error("lt_softmax_stats_colsum_bf16: memset failed"); return LT_ERR_HIP; }
    return lt_colsum_bf16(logits, colsum, rows, K, stream);
  }
}
template <typename TL>
int ce_fwd_bwd_logits_impl(const TL* s, const TL* t_logits, const float* t_stats, const float* center_a, const float* center_b, int split_row,
                           const int32_t* ta, const int32_t* tb, const float* row_weight, const int32_t* slot, float scale, float inv_temp,
                           float inv_temp_t, float* loss, void* dlogits_bf16, int rows, int K, void* stream) {
  LT_CHECK_ARG(s && t_logits && t_stats && ta && loss && K > 0, "lt_ce_fwd_bwd_logits: bad arguments");
  if (rows == 0) return LT_OK;
  float* terms = lt_scratch_ring((size_t)rows);
  if (!terms) { lt_set_error("lt_ce_fwd_bwd_logits: scratch allocation failed"); return LT_ERR_HIP; }
  static const int reg_rows = [] { const char* e = getenv("LT_LOSS_REG"); return e ? atoi(e) : 1; }();
  auto al16 = [](const void* p) { return p == nullptr || (uintptr_t)p % 16 == 0; };
  if (reg_rows && K % 4 == 0 && K <= 4096 * ROW_NV && K >= 8192 && al16(s) && al16(t_logits) && al16(center_a) && al16(center_b) &&
      (!dlogits_bf16 || (uintptr_t)dlogits_bf16 % 8 == 0))
    hipLaunchKernelGGL(ce_logits_reg_kernel<TL>, dim3(rows), dim3(1024), 0, ST, s, t_logits, t_stats, center_a, center_b, split_row, ta, tb,
                       row_weight, scale, inv_temp, inv_temp_t, terms, (bf16_t*)dlogits_bf16, K);
  else
    hipLaunchKernelGGL(ce_logits_kernel<TL>, dim3(rows), dim3(256), 0, ST, s, t_logits, t_stats, center_a, center_b, split_row, ta, tb, row_weight,
                       scale, inv_temp, inv_temp_t, terms, (bf16_t*)dlogits_bf16, K);
  hipLaunchKernelGGL(rowloss_sum_kernel, dim3(1), dim3(256), 0, ST, terms, slot, rows, loss);
  LT_CHECK_LAUNCH("lt_ce_fwd_bwd_logits");
}
}  // namespace
extern "C" int lt_softmax_stats_colsum(const float* logits, const float* center, float* stats, float* colsum, int rows, int K, float inv_temp,
                                       float* scratch, int64_t scratch_floats, void* stream) {
  return softmax_stats_colsum_impl<float>(logits, center, stats, colsum, rows, K, inv_temp, scratch, scratch_floats, stream);
}
extern "C" int lt_softmax_stats_colsum_bf16(const void* logits_bf16, const float* center, float* stats, float* colsum, int rows, int K, float inv_temp,
                                            float* scratch, int64_t scratch_floats, void* stream) {
  return softmax_stats_colsum_impl<bf16_t>((const bf16_t*)logits_bf16, center, stats, colsum, rows, K, inv_temp, scratch, scratch_floats, stream);
}
extern "C" int lt_ce_fwd_bwd_logits(const float* s, const float* t_logits, const float* t_stats, const float* center_a, const float* center_b,
                                    int split_row, const int32_t* ta, const int32_t* tb, const float* row_weight, const int32_t* slot,
                                    float scale, float inv_temp, float inv_temp_t, float* loss, void* dlogits_bf16, int rows, int K,
                                    void* stream) {
  return ce_fwd_bwd_logits_impl<float>(s, t_logits, t_stats, center_a, center_b, split_row, ta, tb, row_weight, slot, scale, inv_temp, inv_temp_t, loss,
                                       dlogits_bf16, rows, K, stream);
}
extern "C" int lt_ce_fwd_bwd_logits_bf16(const void* s_bf16, const void* t_logits_bf16, const float* t_stats, const float* center_a, const float* center_b,
                                         int split_row, const int32_t* ta, const int32_t* tb, const float* row_weight, const int32_t* slot,
                                         float scale, float inv_temp, float inv_temp_t, float* loss, void* dlogits_bf16, int rows, int K,
                                         void* stream) {
  return ce_fwd_bwd_logits_impl<bf16_t>((const bf16_t*)s_bf16, (const bf16_t*)t_logits_bf16, t_stats, center_a, center_b, split_row, ta, tb, row_weight,
                                        slot, scale, inv_temp, inv_temp_t, loss, dlogits_bf16, rows, K, stream);
}
extern "C" int lt_sk_exp(const float* logits, float* Q, int64_t n, float inv_temp, void* stream) {
  LT_CHECK_ARG(logits && Q, "lt_sk_exp: null pointer");
  if (n == 0) return LT_OK;
  const int grid = (int)min((long)4096, (long)lt_cdiv(n, 256));
  hipLaunchKernelGGL(sk_exp_kernel, dim3(grid), dim3(256), 0, ST, logits, Q, (long)n, inv_temp);
  LT_CHECK_LAUNCH("lt_sk_exp");
}
extern "C" int lt_sk_iter(float* Q, const float* colsum, int rows, int K, float n_total, float final_mul, void* stream) {
  LT_CHECK_ARG(Q && colsum && K > 0, "lt_sk_iter: bad arguments");
  if (rows == 0) return LT_OK;
  hipLaunchKernelGGL(sk_iter_kernel, dim3(rows), dim3(256), 0, ST, Q, colsum, K, n_total, final_mul);
  LT_CHECK_LAUNCH("lt_sk_iter");
}
extern "C" int lt_koleo_fwd_bwd(const float* x, int ld, float* loss, float* dx, int ld_dx, int n, int D, float eps, float weight,
                                float* ws, int32_t* nn, void* stream) {
  LT_CHECK_ARG(x && loss && dx && ws && nn && n > 1 && D > 0, "lt_koleo_fwd_bwd: bad arguments (n=%d)", n);
  float* xn = ws;
  float* inv = ws + (size_t)n * D;
  float* coef = inv + n;
  float* dxn = coef + n;    // [n, D]; its first n floats hold the per-row loss terms until koleo_loss_kernel has added them
  hipLaunchKernelGGL(koleo_normalize_kernel, dim3(n), dim3(256), 0, ST, x, ld, xn, inv, D, eps);
  hipLaunchKernelGGL(koleo_nn_kernel, dim3(n), dim3(256), 0, ST, xn, nn, coef, dxn, n, D, eps, weight);
  hipLaunchKernelGGL(koleo_loss_kernel, dim3(1), dim3(64), 0, ST, dxn, n, loss);
  if (weight == 0.f) { LT_CHECK_LAUNCH("lt_koleo_fwd_bwd"); }   // value only (the reference logs the term even when it is not trained on)
  hipLaunchKernelGGL(koleo_dxn_kernel, dim3(n), dim3(256), 0, ST, xn, nn, coef, dxn, n, D, eps);
  hipLaunchKernelGGL(koleo_dx_kernel, dim3(n), dim3(256), 0, ST, xn, dxn, inv, dx, ld_dx, D);
  LT_CHECK_LAUNCH("lt_koleo_fwd_bwd");
}
extern "C" int lt_mse_fwd_bwd(const float* s, const float* t, float* ds, int64_t n, float scale, float* loss, void* stream) {
  LT_CHECK_ARG(s && t && loss && n >= 0 && ((uintptr_t)s & 15) == 0 && ((uintptr_t)t & 15) == 0 && ((uintptr_t)ds & 15) == 0,
               "lt_mse_fwd_bwd: bad arguments / alignment");
  if (n == 0) return LT_OK;
  const int grid = (int)min((long)MSE_MAX_GRID, (long)lt_cdiv(n, 1024));
  float* partials = lt_scratch_ring((size_t)grid);
  if (!partials) { lt_set_error("lt_mse_fwd_bwd: scratch allocation failed"); return LT_ERR_HIP; }
  hipLaunchKernelGGL(mse_kernel, dim3(grid), dim3(256), 0, ST, s, t, ds, (long)n, scale, partials);
  hipLaunchKernelGGL(rowloss_sum_kernel, dim3(1), dim3(256), 0, ST, partials, (const int32_t*)nullptr, grid, loss);
  LT_CHECK_LAUNCH("lt_mse_fwd_bwd");
}
