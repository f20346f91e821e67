// Argument block of the MFMA GEMM kernels (gemm.hip: 128^2 / 256^2 tiles).
#pragma once
#include "lt_common.h"

namespace lt_gemm {

enum Epi : int {
  EPI_BF16 = LT_EPI_BF16,
  EPI_BF16_GELU = LT_EPI_BF16_GELU,
  EPI_RESID = LT_EPI_RESID,
  EPI_F32 = LT_EPI_F32,
  EPI_BF16_GELUGRAD = LT_EPI_BF16_GELUGRAD,
  EPI_F32_ACCUM = LT_EPI_F32_ACCUM,
};

struct GemmArgs {
  const bf16_t* A; const bf16_t* B;
  int M, N, K, lda, ldb;
  void* C; int ldc;
  void* C2; int ldc2;
  const float* bias; const float* gamma;
  const float* resid; int ldr;
  const bf16_t* aux; int ldaux;
  const float* rowscale; float branch_scale;
  float alpha;
  int tiles_m, tiles_n, k_per_split;
  int band;         // four-phase kernel: column tiles per band of the tile order (0 / >= tiles_n: plain row-major order)
  long sa, sb, sc;  // batched launch (gridDim.z > 1) of the 128x128 kernel: element strides of A, B, C between batch entries
  float* cs;        // four-phase slab kernel with CS: per-(slice, column tile, wave column) partial sums over k of the transposed A operand,
                    // [gridDim.y * tiles_n * 4][M] floats (the bias gradient of the Linear whose weight gradient this GEMM forms)
};

}  // namespace lt_gemm
