// bf16 MFMA GEMM for gfx950 (MI355X) with fused epilogues -- the matmul core of the DINOv2 step
// (replaces the ATen Linear/Conv2d calls of LT/_models/dinov2_vit/dinov2_vit_src/layers/{attention,mlp,
// patch_embed}.py and LT/_methods/dinov2/dinov2_head.py, forward AND backward).
//
//   C[M,N] = opA(A) . opB(B)        fp32 accumulate on v_mfma_f32_32x32x16_bf16
//   TA=0: A stored [M][K] (K contiguous)      TA=1: A stored [K][M] (M contiguous)
//   TB=0: B stored [N][K] (K contiguous)      TB=1: B stored [K][N] (N contiguous)
//   forward  y = x W^T        : TA=0 TB=0        (x [M,K], W [N,K])
//   dgrad    dx = dy W        : TA=0 TB=1        (dy [M,N'], W [N',K'] read as [k=N'][n=K'])
//   wgrad    dW = dy^T x      : TA=1 TB=1        (dy [Mtok,N] read as [k][m], x [Mtok,K] read as [k][n])
//
// Block tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles.
// HBM -> VGPR (16-B loads, issued one k-tile ahead, T14 split) -> LDS (double buffered, one barrier
// per k-tile) -> MFMA fragments:
//   * K-contiguous operands: LDS image [128 rows][64 k], 16-B chunk index XOR ((row>>1)&7):
//     conflict-free for ds_read_b128's 16-lane groups and for the 8-lane ds_write_b128 groups.
//   * transposed operands: LDS image of [4 k][16 rows] 128-B pieces, piece(q=k/4, b=row/16) at
//     (q*8+b)*128 with its k-rows rotated by b; fragments come from ds_read_b64_tr_b16 (hardware
//     4x16 transpose; semantics pinned by tools/probe_hw.hip).
// Tails: M, N arbitrary (predicated loads/stores); contiguous dims must be multiples of 8 elements.
// XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so the id is
// remapped to give each XCD's L2 a contiguous band of tiles sharing A-rows.
#include "lt_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

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
  float alpha;
  int tiles_m, tiles_n, k_per_split;
};

__device__ __forceinline__ uint4 ldg16(const bf16_t* p, bool ok) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (ok) v = *reinterpret_cast<const uint4*>(p);
  return v;
}

// ---- global -> registers (one 128 x 64 operand tile = 4 x 16 B per thread) ---------------------
template <bool TR>
__device__ __forceinline__ void load_tile(uint4 (&r)[4], const bf16_t* __restrict__ P, int ld, int rows, int kdim,
                                          int row0, int k0, int kend) {
  const int t = threadIdx.x;
  if (!TR) {
    const int c = t & 7, rr = t >> 3;
    const int k = k0 + c * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + rr + 32 * i;
      r[i] = ldg16(P + (size_t)row * ld + k, row < rows && k < kend);
    }
  } else {
    const int rc = t & 15, kk = t >> 4;
    const int row = row0 + rc * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + kk + 16 * i;
      r[i] = ldg16(P + (size_t)k * ld + row, k < kend && row < rows);
    }
  }
}

// ---- registers -> LDS -----------------------------------------------------------------------
template <bool TR>
__device__ __forceinline__ void store_tile(const uint4 (&r)[4], char* lds) {
  const int t = threadIdx.x;
  if (!TR) {
    const int c = t & 7, rr = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rr + 32 * i;
      *reinterpret_cast<uint4*>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = r[i];
    }
  } else {
    const int rc = t & 15, kk = t >> 4;
    const int b = rc >> 1, half = rc & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kk + 16 * i;
      const int q = k >> 2, kr = k & 3;
      *reinterpret_cast<uint4*>(lds + (q * 8 + b) * 128 + (((kr + b) & 3) << 5) + (half << 4)) = r[i];
    }
  }
}

// ---- LDS -> MFMA fragment (32 rows x 16 k) -----------------------------------------------------
// rb: 32-row block (0..3) inside the 128-row tile, ks: k16 step (0..3) inside the 64-k tile
template <bool TR>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int rb, int ks) {
  const int l = threadIdx.x & 63;
  if (!TR) {
    const int row = rb * 32 + (l & 31);
    const int c = ks * 2 + (l >> 5);
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  } else {
    const int i = l & 15, cb = (l >> 4) & 1, kh = l >> 5;
    const int b = rb * 2 + cb;
    const int inner = ((((i >> 2) + b) & 3) << 5) + ((i & 3) << 3);
    const int q0 = ks * 4 + kh * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (q0 * 8 + b) * 128 + inner));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + ((q0 + 1) * 8 + b) * 128 + inner));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
  }
}

template <bool TA, bool TB, int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // stage s: A image at smem + 2*s*TILE_BYTES, B image right behind it
#define LDS_A(s) (smem + (2 * (s)) * TILE_BYTES)
#define LDS_B(s) (smem + (2 * (s) + 1) * TILE_BYTES)

  // XCD-aware remap of the linear tile id (bijective; guide T1)
  const int ntiles = g.tiles_m * g.tiles_n;
  int id = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = id & 7, j = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tm = id / g.tiles_n, tn = id % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.y * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  uint4 ra[4], rb[4];
  if (nk > 0) {
    load_tile<TA>(ra, g.A, g.lda, g.M, g.K, m0, kbeg, kend);
    load_tile<TB>(rb, g.B, g.ldb, g.N, g.K, n0, kbeg, kend);
    store_tile<TA>(ra, LDS_A(0));
    store_tile<TB>(rb, LDS_B(0));
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      load_tile<TA>(ra, g.A, g.lda, g.M, g.K, m0, kbeg + (kt + 1) * BK, kend);
      load_tile<TB>(rb, g.B, g.ldb, g.N, g.K, n0, kbeg + (kt + 1) * BK, kend);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[2], fb[2];
      fa[0] = read_frag<TA>(LDS_A(cur), wm * 2 + 0, ks);
      fa[1] = read_frag<TA>(LDS_A(cur), wm * 2 + 1, ks);
      fb[0] = read_frag<TB>(LDS_B(cur), wn * 2 + 0, ks);
      fb[1] = read_frag<TB>(LDS_B(cur), wn * 2 + 1, ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_tile<TA>(ra, LDS_A(cur ^ 1));
      store_tile<TB>(rb, LDS_B(cur ^ 1));
    }
    __syncthreads();
  }

  // ---- epilogue: C layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + (l & 31);
    if (col >= g.N) continue;
    float bias = 0.f, gam = 1.f;
    if (EPI != EPI_F32_ACCUM && EPI != EPI_BF16_GELUGRAD && g.bias) bias = g.bias[col];
    if (EPI == EPI_RESID && g.gamma) gam = g.gamma[col];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        if (row >= g.M) continue;
        float v = acc[i][j][e] * g.alpha + bias;
        const size_t o = (size_t)row * g.ldc + col;
        if (EPI == EPI_BF16) {
          ((bf16_t*)g.C)[o] = f2bf(v);
        } else if (EPI == EPI_BF16_GELU) {
          if (g.C2) ((bf16_t*)g.C2)[(size_t)row * g.ldc2 + col] = f2bf(v);
          ((bf16_t*)g.C)[o] = f2bf(gelu_f(v));
        } else if (EPI == EPI_RESID) {
          if (g.C2) ((bf16_t*)g.C2)[(size_t)row * g.ldc2 + col] = f2bf(v);
          const float rs = g.resid ? g.resid[(size_t)row * g.ldr + col] : 0.f;
          ((float*)g.C)[o] = rs + gam * v;
        } else if (EPI == EPI_F32) {
          ((float*)g.C)[o] = v;
        } else if (EPI == EPI_BF16_GELUGRAD) {
          const float pre = bf2f(g.aux[(size_t)row * g.ldaux + col]);
          ((bf16_t*)g.C)[o] = f2bf(v * gelu_grad_f(pre));
        } else if (EPI == EPI_F32_ACCUM) {
          if (gridDim.y > 1) atomicAdd(&((float*)g.C)[o], v);
          else ((float*)g.C)[o] += v;
        }
      }
    }
  }
}

// ---- plain reference-grade GEMM (one thread per output; cross-check for the MFMA kernel) -------
__global__ void gemm_naive_kernel(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K, int lda, int ldb,
                                  int ldc, int ta, int tb) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (col >= N || row >= M) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = bf2f(ta ? A[(size_t)k * lda + row] : A[(size_t)row * lda + k]);
    const float b = bf2f(tb ? B[(size_t)k * ldb + col] : B[(size_t)col * ldb + k]);
    s = fmaf(a, b, s);
  }
  C[(size_t)row * ldc + col] = s;
}

template <bool TA, bool TB>
int launch_epi(const GemmArgs& g, int epi, dim3 grid, hipStream_t st) {
  const size_t smem = 4 * TILE_BYTES;
#define LT_CASE(E)                                                              \
  case E:                                                                        \
    hipLaunchKernelGGL((gemm_kernel<TA, TB, E>), grid, dim3(NTHREADS), smem, st, g); \
    break;
  switch (epi) {
    LT_CASE(EPI_BF16)
    LT_CASE(EPI_BF16_GELU)
    LT_CASE(EPI_RESID)
    LT_CASE(EPI_F32)
    LT_CASE(EPI_BF16_GELUGRAD)
    LT_CASE(EPI_F32_ACCUM)
    default:
      lt_set_error("lt_gemm_bf16: unknown epilogue %d", epi);
      return LT_ERR_INVALID;
  }
#undef LT_CASE
  return LT_OK;
}

}  // namespace

extern "C" int lt_gemm_bf16(const lt_gemm_desc* d, void* stream) {
  LT_CHECK_ARG(d != nullptr, "lt_gemm_bf16: null descriptor");
  LT_CHECK_ARG(d->M >= 0 && d->N > 0 && d->K > 0, "lt_gemm_bf16: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  if (d->M == 0) return LT_OK;
  LT_CHECK_ARG(d->A && d->B && d->C, "lt_gemm_bf16: null operand");
  // contiguous dims must be multiples of 8 elements (16-byte vector loads)
  LT_CHECK_ARG(d->lda % 8 == 0 && d->ldb % 8 == 0, "lt_gemm_bf16: lda/ldb must be multiples of 8 (lda=%d ldb=%d)", d->lda, d->ldb);
  LT_CHECK_ARG(d->trans_a ? (d->M % 8 == 0) : (d->K % 8 == 0), "lt_gemm_bf16: A contiguous dim not a multiple of 8");
  LT_CHECK_ARG(d->trans_b ? (d->N % 8 == 0) : (d->K % 8 == 0), "lt_gemm_bf16: B contiguous dim not a multiple of 8");
  LT_CHECK_ARG(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, "lt_gemm_bf16: A/B must be 16-byte aligned");
  LT_CHECK_ARG(d->epilogue != LT_EPI_BF16_GELUGRAD || d->aux, "lt_gemm_bf16: GELUGRAD epilogue needs aux");
  GemmArgs g;
  g.A = (const bf16_t*)d->A; g.B = (const bf16_t*)d->B;
  g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb;
  g.C = d->C; g.ldc = d->ldc; g.C2 = d->C2; g.ldc2 = d->ldc2;
  g.bias = d->bias; g.gamma = d->gamma; g.resid = d->resid; g.ldr = d->ldr;
  g.aux = (const bf16_t*)d->aux; g.ldaux = d->ldaux;
  g.alpha = d->alpha;
  g.tiles_m = lt_cdiv(d->M, BM); g.tiles_n = lt_cdiv(d->N, BN);
  int split = d->split_k > 0 ? d->split_k : 1;
  if (d->epilogue != LT_EPI_F32_ACCUM) split = 1;
  const int ktiles = lt_cdiv(d->K, BK);
  if (split > ktiles) split = ktiles;
  g.k_per_split = lt_cdiv(ktiles, split) * BK;
  split = lt_cdiv(d->K, g.k_per_split);
  dim3 grid(g.tiles_m * g.tiles_n, split);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (!d->trans_a && !d->trans_b) rc = launch_epi<false, false>(g, d->epilogue, grid, st);
  else if (!d->trans_a && d->trans_b) rc = launch_epi<false, true>(g, d->epilogue, grid, st);
  else if (d->trans_a && d->trans_b) rc = launch_epi<true, true>(g, d->epilogue, grid, st);
  else rc = launch_epi<true, false>(g, d->epilogue, grid, st);
  if (rc != LT_OK) return rc;
  LT_CHECK_LAUNCH("lt_gemm_bf16");
}

extern "C" int lt_gemm_bf16_naive(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                  int trans_a, int trans_b, void* stream) {
  LT_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "lt_gemm_bf16_naive: bad arguments");
  dim3 grid(lt_cdiv(N, 128), M);
  hipLaunchKernelGGL(gemm_naive_kernel, grid, dim3(128), 0, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)B, C, M, N,
                     K, lda, ldb, ldc, trans_a, trans_b);
  LT_CHECK_LAUNCH("lt_gemm_bf16_naive");
}
