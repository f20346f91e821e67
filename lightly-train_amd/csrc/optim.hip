// Multi-tensor optimizer kernels on flat parameter storage: global grad-norm, AdamW (+clip, +bf16 shadow),
// EMA teacher update (+bf16 shadow).  Pure HBM streams: AdamW moves 28 B/param, EMA 14 B/param.
// Replaces torch.optim.AdamW(foreach) + clip_grad_norm_ + _foreach_mul_/_foreach_add_
// (LT/_methods/dinov2/utils.py:191-250, dinov2.py:588-660, LT/_torch_helpers.py:75-96).
#include "lt_common.h"
#include <atomic>

namespace {

// Deterministic global sum of squares: every block leaves its partial in a scratch slot, the last block to arrive adds
// the partials in a fixed order.  The clip coefficient derived from it multiplies every gradient, so an atomic (order-
// dependent) sum would let data-parallel replicas drift apart by an ulp per step although their gradients are identical.
constexpr int SUMSQ_SLOTS = 16, SUMSQ_MAX_GRID = 1024;
__device__ float sumsq_partials[SUMSQ_SLOTS][SUMSQ_MAX_GRID];
__device__ unsigned sumsq_tickets[SUMSQ_SLOTS];

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, float* __restrict__ out, long n, int slot) {
  __shared__ float red[16];
  __shared__ bool last;
  float s = 0.f;
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(g + i);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (i < n) for (long j = i; j < n; ++j) s += g[j] * g[j];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    sumsq_partials[slot][blockIdx.x] = s;
    __threadfence();
    last = atomicAdd(&sumsq_tickets[slot], 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += __builtin_nontemporal_load(&sumsq_partials[slot][b]);
  t = block_sum(t, red);
  if (threadIdx.x == 0) {
    *out += t;
    sumsq_tickets[slot] = 0;
  }
}

// one block per 1024-element chunk (a chunk never straddles two parameter tensors)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ pb, long n,
                                                    const int32_t* __restrict__ seg_of_chunk, const float* __restrict__ seg_lr,
                                                    const uint8_t* __restrict__ seg_wd_on, const uint8_t* __restrict__ seg_frozen,
                                                    int freeze, float lr_factor, float wd, float beta1, float beta2, float om1, float om2, float eps,
                                                    float bc1, float bc2_sqrt, const float* __restrict__ sumsq, float max_norm) {
  const long chunk = blockIdx.x;
  const int seg = seg_of_chunk[chunk];
  float lr = seg_lr[seg] * lr_factor;
  if (freeze & seg_frozen[seg]) lr = 0.f;   // bit 0: last-layer freeze, bit 1: backbone freeze (dinov2.py:619-635)
  const float wdv = seg_wd_on[seg] ? wd : 0.f;
  float clip = 1.f;
  if (max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
  const long i = chunk * 1024 + threadIdx.x * 4;
  if (i >= n) return;
  float4 pp = *reinterpret_cast<float4*>(p + i);
  float4 gg = *reinterpret_cast<const float4*>(g + i);
  float4 mm = *reinterpret_cast<float4*>(m + i);
  float4 vv = *reinterpret_cast<float4*>(v + i);
  float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gr = G[e] * clip;
    P[e] *= 1.f - lr * wdv;                      // decoupled weight decay (torch AdamW order)
    M[e] = M[e] + (gr - M[e]) * om1;             // lerp form used by torch; om = 1 - beta formed in double on the host
    V[e] = V[e] * beta2 + gr * gr * om2;
    const float denom = sqrtf(V[e]) / bc2_sqrt + eps;
    P[e] -= (lr / bc1) * (M[e] / denom);
  }
  *reinterpret_cast<float4*>(p + i) = pp;
  *reinterpret_cast<float4*>(m + i) = mm;
  *reinterpret_cast<float4*>(v + i) = vv;
  if (pb) *reinterpret_cast<uint2*>(pb + i) = make_uint2(pack_bf2(pp.x, pp.y), pack_bf2(pp.z, pp.w));
}

// mom and om = 1 - mom are both formed in double on the host: near the end of the cosine schedule 1 - m ~ 1e-6 and `1.f - mom` would be
// quantised to 2^-24 steps (percent-level relative error); the reference computes 1 - m in Python floats (_torch_helpers.py:75-96).
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ t, const float* __restrict__ s, bf16_t* __restrict__ tb, long n,
                                                  float mom, float om) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 256 * 4;
  for (; i + 3 < n; i += stride) {
    float4 a = *reinterpret_cast<float4*>(t + i);
    const float4 b = *reinterpret_cast<const float4*>(s + i);
    a.x = a.x * mom + b.x * om; a.y = a.y * mom + b.y * om; a.z = a.z * mom + b.z * om; a.w = a.w * mom + b.w * om;
    *reinterpret_cast<float4*>(t + i) = a;
    if (tb) *reinterpret_cast<uint2*>(tb + i) = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w));
  }
}

// ---- LARS (lightly.utils.lars.LARS as LT/_optim/lars_args.py builds it; the rule is stated in include/lt_amd.h) ----------------
// Per parameter tensor: ||p||, ||g|| -> trust ratio -> momentum-SGD step.  Norms in two deterministic stages: one block per 1024-element
// chunk leaves (sum p^2, sum g^2) in scratch, one block per segment adds its chunks in a fixed order (double accumulators).
__global__ __launch_bounds__(256) void lars_chunk_norms_kernel(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ws) {
  __shared__ float red[16];
  const long i = (long)blockIdx.x * 1024 + threadIdx.x * 4;
  const float4 a = *reinterpret_cast<const float4*>(p + i), b = *reinterpret_cast<const float4*>(g + i);
  const float sp = block_sum(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w, red);
  __syncthreads();
  const float sg = block_sum(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w, red);
  if (threadIdx.x == 0) { ws[2 * blockIdx.x] = sp; ws[2 * blockIdx.x + 1] = sg; }
}
__global__ __launch_bounds__(64) void lars_seg_norms_kernel(const float* __restrict__ ws, const int32_t* __restrict__ seg_chunk_begin,
                                                            float* __restrict__ seg_norms) {
  const int seg = blockIdx.x, c0 = seg_chunk_begin[seg], c1 = seg_chunk_begin[seg + 1];
  double sp = 0.0, sg = 0.0;
  for (int c = c0 + (int)threadIdx.x; c < c1; c += 64) { sp += (double)ws[2 * c]; sg += (double)ws[2 * c + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sp += __shfl_down(sp, o, 64); sg += __shfl_down(sg, o, 64); }
  if (threadIdx.x == 0) { seg_norms[2 * seg] = (float)sqrt(sp); seg_norms[2 * seg + 1] = (float)sqrt(sg); }
}
// ADAPT = false: torch.optim.SGD (LT/_optim/sgd_args.py:19-31) -- the same momentum rule without the trust ratio, coupled weight decay on
// every tensor of a decayed group: d = g + wd * p.
template <bool ADAPT>
__global__ __launch_bounds__(256) void lars_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, bf16_t* __restrict__ pb,
                                                   const int32_t* __restrict__ seg_of_chunk, const float* __restrict__ seg_lr,
                                                   const uint8_t* __restrict__ seg_wd_on, const float* __restrict__ seg_norms, float lr_factor, float wd,
                                                   float momentum, float dampening, int nesterov, float trust, float eps, int first_step,
                                                   const float* __restrict__ sumsq, float max_norm) {
  const long chunk = blockIdx.x;
  const int seg = seg_of_chunk[chunk];
  const float lr = seg_lr[seg] * lr_factor;
  float clip = 1.f;
  if (max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));   // the optimizer sees the clipped gradient
  const float wdv = seg_wd_on[seg] ? wd : 0.f;
  float q = 1.f, wadd = wdv;
  if constexpr (ADAPT) {
    const float p_norm = seg_norms[2 * seg], g_norm = seg_norms[2 * seg + 1] * clip;
    const bool adapt = wdv != 0.f && p_norm != 0.f && g_norm != 0.f;
    q = adapt ? p_norm / (g_norm + p_norm * wdv + eps) * trust : 1.f;
    wadd = adapt ? wdv : 0.f;
  }
  const long i = chunk * 1024 + threadIdx.x * 4;
  float4 pp = *reinterpret_cast<float4*>(p + i);
  const float4 gg = *reinterpret_cast<const float4*>(g + i);
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (momentum != 0.f && !first_step) bb = *reinterpret_cast<float4*>(buf + i);
  float* P = &pp.x; const float* G = &gg.x; float* B = &bb.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float d = (G[e] * clip + wadd * P[e]) * q;
    if (momentum != 0.f) {
      B[e] = first_step ? d : B[e] * momentum + (1.f - dampening) * d;     // torch SGD: the buffer starts as a copy of the first step
      d = nesterov ? d + momentum * B[e] : B[e];
    }
    P[e] -= lr * d;
  }
  *reinterpret_cast<float4*>(p + i) = pp;
  if (momentum != 0.f) *reinterpret_cast<float4*>(buf + i) = bb;
  if (pb) *reinterpret_cast<uint2*>(pb + i) = make_uint2(pack_bf2(pp.x, pp.y), pack_bf2(pp.z, pp.w));
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int lt_sumsq_f32(const float* g, float* out, int64_t n, void* stream) {
  LT_CHECK_ARG(g && out && ((uintptr_t)g & 15) == 0, "lt_sumsq_f32: bad pointer/alignment");
  if (n == 0) return LT_OK;
  static std::atomic<unsigned> next_slot{0};   // launches in flight on different streams use different scratch slots
  const int slot = (int)(next_slot.fetch_add(1) % SUMSQ_SLOTS);
  const int grid = (int)min((long)SUMSQ_MAX_GRID, (long)lt_cdiv(n, 1024));
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, ST, g, out, (long)n, slot);
  LT_CHECK_LAUNCH("lt_sumsq_f32");
}

extern "C" int lt_adamw_flat(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, const int32_t* seg_of_chunk,
                             const float* seg_lr, const uint8_t* seg_wd_on, const uint8_t* seg_frozen, int freeze,
                             float lr_factor, float wd, double beta1, double beta2, float eps, int step, const float* sumsq,
                             float max_norm, void* stream) {
  LT_CHECK_ARG(p && g && m && v && seg_of_chunk && seg_lr && seg_wd_on && seg_frozen, "lt_adamw_flat: null pointer");
  LT_CHECK_ARG(n % 1024 == 0, "lt_adamw_flat: n must be a multiple of the 1024-element chunk (n=%ld)", (long)n);
  LT_CHECK_ARG(step >= 1 && (max_norm <= 0.f || sumsq), "lt_adamw_flat: step must be >= 1 and sumsq given when clipping");
  if (n == 0) return LT_OK;
  // bias corrections in double like torch.optim.AdamW (1 - beta ** step on Python floats): powf loses ~3e-5 relative at step 1
  const float bc1 = (float)(1.0 - pow(beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(n / 1024)), dim3(256), 0, ST, p, g, m, v, (bf16_t*)p_bf16, (long)n, seg_of_chunk,
                     seg_lr, seg_wd_on, seg_frozen, freeze, lr_factor, wd, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, bc1, bc2_sqrt, sumsq, max_norm);
  LT_CHECK_LAUNCH("lt_adamw_flat");
}

extern "C" int lt_ema_flat(float* teacher, const float* student, void* teacher_bf16, int64_t n, double m, void* stream) {
  LT_CHECK_ARG(teacher && student && n % 4 == 0, "lt_ema_flat: bad arguments");
  if (n == 0) return LT_OK;
  const int grid = (int)min((long)2048, (long)lt_cdiv(n, 1024));
  hipLaunchKernelGGL(ema_kernel, dim3(grid), dim3(256), 0, ST, teacher, student, (bf16_t*)teacher_bf16, (long)n, (float)m, (float)(1.0 - m));
  LT_CHECK_LAUNCH("lt_ema_flat");
}

extern "C" int lt_lars_norms(const float* p, const float* g, int64_t n, const int32_t* seg_chunk_begin, int nseg, float* ws, float* seg_norms,
                             void* stream) {
  LT_CHECK_ARG(p && g && seg_chunk_begin && ws && seg_norms && nseg > 0 && n % 1024 == 0, "lt_lars_norms: bad arguments (n=%ld)", (long)n);
  if (n == 0) return LT_OK;
  hipLaunchKernelGGL(lars_chunk_norms_kernel, dim3((unsigned)(n / 1024)), dim3(256), 0, ST, p, g, ws);
  hipLaunchKernelGGL(lars_seg_norms_kernel, dim3((unsigned)nseg), dim3(64), 0, ST, ws, seg_chunk_begin, seg_norms);
  LT_CHECK_LAUNCH("lt_lars_norms");
}

extern "C" int lt_lars_flat(float* p, const float* g, float* buf, void* p_bf16, int64_t n, const int32_t* seg_of_chunk, const float* seg_lr,
                            const uint8_t* seg_wd_on, const float* seg_norms, float lr_factor, float wd, float momentum, float dampening,
                            int nesterov, float trust, float eps, int first_step, const float* sumsq, float max_norm, void* stream) {
  LT_CHECK_ARG(p && g && seg_of_chunk && seg_lr && seg_wd_on && seg_norms && (momentum == 0.f || buf), "lt_lars_flat: null pointer");
  LT_CHECK_ARG(n % 1024 == 0, "lt_lars_flat: n must be a multiple of the 1024-element chunk (n=%ld)", (long)n);
  LT_CHECK_ARG(max_norm <= 0.f || sumsq, "lt_lars_flat: sumsq must be given when clipping");
  LT_CHECK_ARG(!nesterov || (momentum > 0.f && dampening == 0.f), "lt_lars_flat: Nesterov momentum requires a momentum and zero dampening");
  if (n == 0) return LT_OK;
  hipLaunchKernelGGL(lars_kernel<true>, dim3((unsigned)(n / 1024)), dim3(256), 0, ST, p, g, buf, (bf16_t*)p_bf16, seg_of_chunk, seg_lr, seg_wd_on, seg_norms,
                     lr_factor, wd, momentum, dampening, nesterov, trust, eps, first_step, sumsq, max_norm);
  LT_CHECK_LAUNCH("lt_lars_flat");
}

extern "C" int lt_sgd_flat(float* p, const float* g, float* buf, void* p_bf16, int64_t n, const int32_t* seg_of_chunk, const float* seg_lr,
                           const uint8_t* seg_wd_on, float lr_factor, float wd, float momentum, float dampening, int nesterov, int first_step,
                           const float* sumsq, float max_norm, void* stream) {
  LT_CHECK_ARG(p && g && seg_of_chunk && seg_lr && seg_wd_on && (momentum == 0.f || buf), "lt_sgd_flat: null pointer");
  LT_CHECK_ARG(n % 1024 == 0, "lt_sgd_flat: n must be a multiple of the 1024-element chunk (n=%ld)", (long)n);
  LT_CHECK_ARG(max_norm <= 0.f || sumsq, "lt_sgd_flat: sumsq must be given when clipping");
  LT_CHECK_ARG(!nesterov || (momentum > 0.f && dampening == 0.f), "lt_sgd_flat: Nesterov momentum requires a momentum and zero dampening");
  if (n == 0) return LT_OK;
  hipLaunchKernelGGL(lars_kernel<false>, dim3((unsigned)(n / 1024)), dim3(256), 0, ST, p, g, buf, (bf16_t*)p_bf16, seg_of_chunk, seg_lr, seg_wd_on,
                     (const float*)nullptr, lr_factor, wd, momentum, dampening, nesterov, 0.f, 0.f, first_step, sumsq, max_norm);
  LT_CHECK_LAUNCH("lt_sgd_flat");
}
