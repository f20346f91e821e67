// Row-owning residual GEMM with the NEXT LayerNorm fused into its epilogue (gfx950):
//
//   x'[M, 768]  = resid + gamma * (A[M, K] . W[768, K]^T + bias)          (attention projection / fc2 + LayerScale + residual,
//                                                                           LT/_models/dinov2_vit/dinov2_vit_src/layers/block.py:90-115)
//   y[M, 768]   = LayerNorm(x') * ln_w + ln_b   (bf16),  mean[M], rstd[M]   (the norm in front of the next branch, block.py:60,74)
//
// A workgroup owns 128 WHOLE rows (tile 128 x 768, 8 waves as 2 x 4, wave tile 64 x 192 = 2 x 6 MFMA blocks of 32 x 32: 192 accumulator
// registers per lane), so the row statistics are complete inside the workgroup and the normalised bf16 operand of the next GEMM leaves from
// registers: the separate LayerNorm pass re-read the fp32 row from HBM (4 of its 6 bytes per element).  The 256 x 256 tile of gemm.hip
// splits a 768-wide row over three workgroups and cannot do this.
//
// K-loop: BK = 32, operands brought in by LDS-DMA (global_load_lds_dwordx4) into [rows][64 B] images, 16-byte chunk c of row r stored at
// c ^ ((r >> 2) & 3) (conflict-free for the 16-lane groups of ds_read_b128: four consecutive rows cover the 64 banks, the chunk rotation
// separates rows r and r + 4); two stages of 8 KiB (A) + 48 KiB (W) = 112 KiB, one barrier per K-tile, fragments of the second k16 step
// requested while the first feeds its 12 MFMAs.
// Epilogue (everything from registers; the 128 KiB of stage buffers become eight wave-private 64 x 64 fp32 transposition tiles):
//   1. per 64-column sub-tile: accumulators -> LDS rows, x' = resid + gamma * (acc + bias) with 16-byte loads / stores, x' back into the
//      accumulator registers;
//   2. row mean, then the centred sum of squares (the arithmetic of layernorm_fwd_rows_kernel): in-lane over the wave's 6 column blocks,
//      shuffles over the 32 column lanes, LDS over the 4 column waves;
//   3. y = (x' - mean) * rstd * ln_w + ln_b through the same transposition tiles, 8-byte bf16 stores.
#include "lt_common.h"

namespace {

constexpr int RM = 128, RN = 768, RK = 32, RT = 512;
constexpr int RA_BYTES = RM * 64, RW_BYTES = RN * 64, RSTAGE = RA_BYTES + RW_BYTES;   // 8 KiB + 48 KiB
constexpr int R_SCRATCH = 8 * 16384;                                                   // epilogue: one 64 x 64 fp32 tile per wave
constexpr int R_LDS = R_SCRATCH + 2 * 4 * RM * (int)sizeof(float);                     // + row partials [2 passes][4 column waves][128 rows]... see below

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

struct RowsArgs {
  const bf16_t* A; const bf16_t* W;
  const float* bias; const float* gamma; const float* resid;
  float* out;
  const float* ln_w; const float* ln_b; bf16_t* y; float* mean; float* rstd;
  int M, K;
  float eps;
};

// DMA one [ROWS x 32 k] operand tile into its LDS image: a wave instruction moves 16 rows x 64 B (lane -> row blk * 16 + lane / 4, slot lane % 4)
template <int ROWS>
__device__ __forceinline__ void stage32(char* lds, const bf16_t* __restrict__ P, int ld, int rows, int row0, int k0) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  constexpr int PER_WAVE = ROWS / 16 / 8;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int blk = w * PER_WAVE + j;
    const int row = blk * 16 + (l >> 2), slot = l & 3;
    const int c = slot ^ ((row >> 2) & 3);
    const int gr = min(row0 + row, rows - 1);
    const bf16_t* src = P + (size_t)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(lds + blk * 1024), 16, 0, 0);
  }
}
// MFMA fragment: 32 rows (block rb) x 16 k (step ks of the 32-k tile)
__device__ __forceinline__ bf16x8 rfrag(const char* lds, int rb, int ks) {
  const int l = threadIdx.x & 63;
  const int row = rb * 32 + (l & 31), c = ks * 2 + (l >> 5);
  return *reinterpret_cast<const bf16x8*>(lds + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
}
__device__ __forceinline__ int crow_(int e, int hi) { return (e & 3) + 8 * (e >> 2) + 4 * hi; }

__global__ __launch_bounds__(RT) void gemm_rows768_kernel(const RowsArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* part = reinterpret_cast<float*>(smem + R_SCRATCH);   // [4 column waves][128 rows] partial row sums
  float* rowv = part + 4 * RM;                                  // [2][128]: mean, rstd of the tile's rows
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, hi = l >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = blockIdx.x * RM;
  const int nk = g.K / RK;

  f32x16 acc[2][6];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  stage32<RM>(smem, g.A, g.K, g.M, m0, 0);
  stage32<RN>(smem + RA_BYTES, g.W, g.K, RN, 0, 0);
  __syncthreads();   // (hipcc drains the outstanding LDS-DMA in front of the barrier)
  for (int kt = 0; kt < nk; ++kt) {
    const char* la = smem + (kt & 1) * RSTAGE;
    const char* lb = la + RA_BYTES;
    if (kt + 1 < nk) {
      char* na = smem + ((kt + 1) & 1) * RSTAGE;
      stage32<RM>(na, g.A, g.K, g.M, m0, (kt + 1) * RK);
      stage32<RN>(na + RA_BYTES, g.W, g.K, RN, 0, (kt + 1) * RK);
    }
    // (16 fragments of both k16 steps would be 64 registers beside 192 accumulators: one step's 8 at a time; the SIMD's second wave covers
    // the read latency)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[2], fb[6];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = rfrag(la, wm * 2 + i, ks);
#pragma unroll
      for (int j = 0; j < 6; ++j) fb[j] = rfrag(lb, wn * 6 + j, ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  // 1. x' = resid + gamma * (acc + bias) in the accumulators' own layout (lane -> column, register -> row: a wave instruction touches two rows
  //    x 128 contiguous bytes), written back into the accumulator registers -- no LDS round trip for the fp32 stream
  float* wl = reinterpret_cast<float*>(smem + wave * 16384);   // wave-private 64 x 64 fp32 tile (over the dead stage buffers), step 3 only
  const int cc = (l & 15) * 4, rs = l >> 4;
  const int row_base = m0 + wm * 64;
  const bool has_resid = g.resid != nullptr;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int col = wn * 192 + j * 32 + (l & 31);
    const float bj = g.bias ? g.bias[col] : 0.f, gj = g.gamma ? g.gamma[col] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float r[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = min(row_base + i * 32 + crow_(e, hi), g.M - 1);   // branch-free: rows past the end re-read the last row, never stored
        r[e] = has_resid ? g.resid[(size_t)row * RN + col] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row_base + i * 32 + crow_(e, hi);
        const float v = r[e] + gj * (acc[i][j][e] + bj);
        acc[i][j][e] = v;
        if (row < g.M) g.out[(size_t)row * RN + col] = v;
      }
    }
  }
  if (!g.y) return;

  // ---- row statistics: mean, then the centred sum of squares
  const float invN = 1.f / (float)RN;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = wm * 64 + i * 32 + crow_(e, hi);
        float p = 0.f;
        if (pass == 0) {
#pragma unroll
          for (int j = 0; j < 6; ++j) p += acc[i][j][e];
        } else {
          const float mu = rowv[r];
#pragma unroll
          for (int j = 0; j < 6; ++j) { const float d = acc[i][j][e] - mu; p += d * d; }
        }
        p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 8, 64); p += __shfl_xor(p, 16, 64);
        if ((l & 31) == 0) part[wn * RM + r] = p;
      }
    __syncthreads();
    if (threadIdx.x < RM) {
      const int r = threadIdx.x;
      const float t = ((part[r] + part[RM + r]) + part[2 * RM + r]) + part[3 * RM + r];
      if (pass == 0) {
        rowv[r] = t * invN;
        if (g.mean && m0 + r < g.M) g.mean[m0 + r] = t * invN;
      } else {
        const float rstd = rsqrtf(t * invN + g.eps);
        rowv[RM + r] = rstd;
        if (g.rstd && m0 + r < g.M) g.rstd[m0 + r] = rstd;
      }
    }
    __syncthreads();
  }

  // ---- y = (x' - mean) * rstd * ln_w + ln_b: the row factors once per accumulator row (two LDS reads per row, not per element), then the
  // column factors while the sub-tiles go through the transposition tiles to 8-byte bf16 stores
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = wm * 64 + i * 32 + crow_(e, hi);
      const float mu = rowv[r], rstd = rowv[RM + r];
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j][e] = (acc[i][j][e] - mu) * rstd;
    }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int c = wn * 192 + (2 * s + jj) * 32 + (l & 31);
      const float lw = g.ln_w[c], lbv = g.ln_b[c];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) wl[(i * 32 + crow_(e, hi)) * 64 + jj * 32 + (l & 31)] = acc[i][2 * s + jj][e] * lw + lbv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int col = wn * 192 + s * 64 + cc;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int rl = t * 4 + rs, row = row_base + rl;
      const float4 v = *reinterpret_cast<const float4*>(wl + rl * 64 + cc);
      if (row < g.M) *reinterpret_cast<uint2*>(g.y + (size_t)row * RN + col) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

// out f32 [M, 768] = resid + gamma * (A bf16 [M, K] . W bf16 [768, K]^T + bias); with ln_out: ln_out bf16 [M, 768] = LayerNorm(out) * ln_w + ln_b,
// mean / rstd f32 [M] (either may be NULL).  resid / gamma / bias may be NULL (0 / 1 / 0).  K % 32 == 0, 16-byte aligned operands.
extern "C" int lt_gemm_resid_ln768(const void* a_bf16, const void* w_bf16, const float* bias, const float* gamma, const float* resid, float* out,
                                   const float* ln_w, const float* ln_b, float eps, void* ln_out_bf16, float* mean, float* rstd, int M, int K,
                                   void* stream) {
  LT_CHECK_ARG(a_bf16 && w_bf16 && out && M > 0 && K >= RK && K % RK == 0, "lt_gemm_resid_ln768: bad arguments (M=%d K=%d)", M, K);
  LT_CHECK_ARG(!ln_out_bf16 || (ln_w && ln_b), "lt_gemm_resid_ln768: the fused LayerNorm needs its weight and bias");
  auto al16 = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
  LT_CHECK_ARG(al16(a_bf16) && al16(w_bf16) && al16(bias) && al16(gamma) && al16(resid) && al16(out) && ((uintptr_t)ln_out_bf16 & 7) == 0,
               "lt_gemm_resid_ln768: operands must be 16-byte aligned");
  static bool configured = false;
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows768_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
    if (e != hipSuccess) { lt_set_error("lt_gemm_resid_ln768: cannot enable %d B of LDS: %s", R_LDS, hipGetErrorString(e)); return LT_ERR_HIP; }
    configured = true;
  }
  RowsArgs g{(const bf16_t*)a_bf16, (const bf16_t*)w_bf16, bias, gamma, resid, out, ln_w, ln_b, (bf16_t*)ln_out_bf16, mean, rstd, M, K, eps};
  hipLaunchKernelGGL(gemm_rows768_kernel, dim3(lt_cdiv(M, RM)), dim3(RT), R_LDS, ST, g);
  LT_CHECK_LAUNCH("lt_gemm_resid_ln768");
}
