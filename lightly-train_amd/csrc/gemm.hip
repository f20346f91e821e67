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
#include <algorithm>

#include "gemm_args.h"
#include "reduce_ledger.h"
#include <cstdlib>

using lt_gemm::GemmArgs;
using namespace lt_gemm;

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

#ifdef LT_GEMM_TIMING
// diagnostic build only (tools/gemm_timeline.py): per-workgroup timestamps of the four-phase kernel (100 MHz wall clock) and the CU it ran on
__device__ unsigned long long lt_gemm_timing_buf[8 * 16384];
#define LT_TSTAMP(slot)                                                                                      \
  do {                                                                                                       \
    if (threadIdx.x == 0) {                                                                                  \
      const int L_ = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;                                     \
      if (L_ < 16384) lt_gemm_timing_buf[L_ * 8 + (slot)] = wall_clock64();                                  \
    }                                                                                                        \
  } while (0)
#else
#define LT_TSTAMP(slot) do { } while (0)
#endif


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

// ---- vector epilogue: emit a [64 x 64] fp32 sub-tile staged in wave-private LDS `wl` (row-major, 64 floats/row) --------
// FULL: all 64 rows are inside the matrix.  That path has no per-row branch on purpose: with one basic block per row the
// waitcnt pass has to assume the bias / residual loads may still be pending at every join and puts `s_waitcnt vmcnt(0)` in
// front of every row -- and since stores count in vmcnt too, each row then waits for the previous row's store to complete
// (measured: 3.6-4.5 TB/s epilogues).  Straight-line code gets counted waits and keeps all 16 stores of a lane in flight.
template <int EPI, bool FULL>
__device__ __forceinline__ void emit_rows(const GemmArgs& g, const float* wl, int row_base, int col, int cc, int rs, bool atomic) {
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), gam4 = make_float4(1.f, 1.f, 1.f, 1.f);
  if (EPI != EPI_F32_ACCUM && EPI != EPI_BF16_GELUGRAD && g.bias) bias4 = *reinterpret_cast<const float4*>(g.bias + col);
  if (EPI == EPI_RESID && g.gamma) gam4 = *reinterpret_cast<const float4*>(g.gamma + col);
  // operands the epilogue has to fetch from HBM (residual rows / GELU pre-activations / stochastic-depth row scales): issue
  // all loads up front so their latency overlaps instead of being paid once per row group
  float4 r4[16];
  uint2 a2[16];
  float rsc[16];
  if (EPI == EPI_RESID || EPI == EPI_BF16_GELUGRAD) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = row_base + it * 4 + rs;
      r4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      a2[it] = make_uint2(0, 0);
      rsc[it] = 1.f;
      if (FULL || row < g.M) {
        if (EPI == EPI_RESID && g.resid) r4[it] = *reinterpret_cast<const float4*>(g.resid + (size_t)row * g.ldr + col);
        if (EPI == EPI_BF16_GELUGRAD) a2[it] = *reinterpret_cast<const uint2*>(g.aux + (size_t)row * g.ldaux + col);
      }
    }
    if (EPI == EPI_RESID && g.rowscale) {   // stochastic depth: subset b/s or per-sample mask/keep
#pragma unroll
      for (int it = 0; it < 16; ++it) rsc[it] = g.rowscale[min(row_base + it * 4 + rs, g.M - 1)];
    }
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int rl = it * 4 + rs;
    const int row = row_base + rl;
    if (!FULL && row >= g.M) continue;
    float4 v = *reinterpret_cast<const float4*>(wl + rl * 64 + cc);
    v.x = v.x * g.alpha + bias4.x; v.y = v.y * g.alpha + bias4.y; v.z = v.z * g.alpha + bias4.z; v.w = v.w * g.alpha + bias4.w;
    const size_t o = (size_t)row * g.ldc + col;
    if (EPI == EPI_BF16) {
      *reinterpret_cast<uint2*>((bf16_t*)g.C + o) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    } else if (EPI == EPI_BF16_GELU) {
      if (g.C2) *reinterpret_cast<uint2*>((bf16_t*)g.C2 + (size_t)row * g.ldc2 + col) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
      gelu2(v.x, v.y); gelu2(v.z, v.w);
      *reinterpret_cast<uint2*>((bf16_t*)g.C + o) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    } else if (EPI == EPI_RESID) {
      if (g.C2) *reinterpret_cast<uint2*>((bf16_t*)g.C2 + (size_t)row * g.ldc2 + col) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
      const float sc = g.branch_scale * rsc[it];
      *reinterpret_cast<float4*>((float*)g.C + o) = make_float4(r4[it].x + sc * gam4.x * v.x, r4[it].y + sc * gam4.y * v.y,
                                                                r4[it].z + sc * gam4.z * v.z, r4[it].w + sc * gam4.w * v.w);
    } else if (EPI == EPI_F32) {
      *reinterpret_cast<float4*>((float*)g.C + o) = v;
    } else if (EPI == EPI_BF16_GELUGRAD) {
      const uint2 a = a2[it];
      const float p0 = bf2f((bf16_t)(a.x & 0xffff)), p1 = bf2f((bf16_t)(a.x >> 16)), p2 = bf2f((bf16_t)(a.y & 0xffff)), p3 = bf2f((bf16_t)(a.y >> 16));
      mul_gelu_grad2(v.x, v.y, p0, p1); mul_gelu_grad2(v.z, v.w, p2, p3);
      *reinterpret_cast<uint2*>((bf16_t*)g.C + o) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    } else if (EPI == EPI_F32_ACCUM) {
      float* c = (float*)g.C + o;
      if (atomic) { atomicAdd(c, v.x); atomicAdd(c + 1, v.y); atomicAdd(c + 2, v.z); atomicAdd(c + 3, v.w); }
      else { float4 old = *reinterpret_cast<float4*>(c); *reinterpret_cast<float4*>(c) = make_float4(old.x + v.x, old.y + v.y, old.z + v.z, old.w + v.w); }
    }
  }
}

// LT_NT_STORE (default 2): the bf16 outputs of the wide epilogue (1: the saved pre-activations, 2: the activations too) leave with
// non-temporal stores.  A [256 x 256] output tile written through L2 evicts the operand panels the next workgroups of the same XCD are
// about to read: with streaming stores the pipeline fill of a workgroup drops from 3.0-3.5 us to 1.3 us (tools/gemm_timeline.py: fc1
// 337.7 -> 324.2 us, qkv 199.9 -> 193.4 us), -0.5 ms per step; the fp32 residual outputs gained nothing and stay cached.
#ifndef LT_NT_STORE
#define LT_NT_STORE 2
#endif
// bf16-output epilogues on a full sub-tile with 16-byte-aligned rows: 8 columns per lane, i.e. 16-byte stores (and 16-byte loads
// of the saved pre-activations) -- half as many store instructions in the queue as the 8-byte form for the same bytes.
template <int EPI>
__device__ __forceinline__ void emit_rows_wide(const GemmArgs& g, const float* wl, int row_base, int col_base, int l) {
  const int cc = (l & 7) * 8, rs = l >> 3;
  const int col = col_base + cc;
  if (col >= g.N) return;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (EPI != EPI_BF16_GELUGRAD && g.bias) { b0 = *reinterpret_cast<const float4*>(g.bias + col); b1 = *reinterpret_cast<const float4*>(g.bias + col + 4); }
  uint4 a4[8];
  if (EPI == EPI_BF16_GELUGRAD) {
#pragma unroll
    for (int it = 0; it < 8; ++it) a4[it] = *reinterpret_cast<const uint4*>(g.aux + (size_t)(row_base + it * 8 + rs) * g.ldaux + col);
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int rl = it * 8 + rs;
    const int row = row_base + rl;
    const float4 w0 = *reinterpret_cast<const float4*>(wl + rl * 64 + cc), w1 = *reinterpret_cast<const float4*>(wl + rl * 64 + cc + 4);
    float v[8] = {w0.x * g.alpha + b0.x, w0.y * g.alpha + b0.y, w0.z * g.alpha + b0.z, w0.w * g.alpha + b0.w,
                  w1.x * g.alpha + b1.x, w1.y * g.alpha + b1.y, w1.z * g.alpha + b1.z, w1.w * g.alpha + b1.w};
    const size_t o = (size_t)row * g.ldc + col;
    if (EPI == EPI_BF16_GELU) {
      if (g.C2) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
#if LT_NT_STORE >= 1
        __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>((bf16_t*)g.C2 + (size_t)row * g.ldc2 + col));
#else
        *reinterpret_cast<u32x4*>((bf16_t*)g.C2 + (size_t)row * g.ldc2 + col) = pk;
#endif
      }
#pragma unroll
      for (int e = 0; e < 8; e += 2) gelu2(v[e], v[e + 1]);
    } else if (EPI == EPI_BF16_GELUGRAD) {
      const unsigned a[4] = {a4[it].x, a4[it].y, a4[it].z, a4[it].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mul_gelu_grad2(v[2 * e], v[2 * e + 1], bf2f((bf16_t)(a[e] & 0xffff)), bf2f((bf16_t)(a[e] >> 16)));
      }
    }
    {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
#if LT_NT_STORE >= 2
      __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>((bf16_t*)g.C + o));
#else
      *reinterpret_cast<u32x4*>((bf16_t*)g.C + o) = pk;
#endif
    }
  }
}

template <int EPI>
__device__ __forceinline__ void emit_subtile(const GemmArgs& g, const float* wl, int row_base, int col_base, int l, bool atomic) {
  if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_GELUGRAD) {
    const bool wide = row_base + 64 <= g.M && col_base + 64 <= g.N && (g.ldc & 7) == 0 && (((uintptr_t)g.C) & 15) == 0 &&
                      (EPI != EPI_BF16_GELU || !g.C2 || ((g.ldc2 & 7) == 0 && (((uintptr_t)g.C2) & 15) == 0)) &&
                      (EPI != EPI_BF16_GELUGRAD || ((g.ldaux & 7) == 0 && (((uintptr_t)g.aux) & 15) == 0)) &&
                      (EPI == EPI_BF16_GELUGRAD || !g.bias || (((uintptr_t)g.bias) & 15) == 0);
    if (wide) { emit_rows_wide<EPI>(g, wl, row_base, col_base, l); return; }
  }
  const int cc = (l & 15) * 4, rs = l >> 4;
  const int col = col_base + cc;
  if (col >= g.N) return;
  if (row_base + 64 <= g.M) emit_rows<EPI, true>(g, wl, row_base, col, cc, rs, atomic);
  else emit_rows<EPI, false>(g, wl, row_base, col, cc, rs, atomic);
}

template <bool TA, bool TB, int EPI, bool VEC>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs g0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GemmArgs g = g0;
  if (gridDim.z > 1) {  // batched: independent problems of one shape (per-image token-similarity matrices of the distillation loss)
    constexpr bool F32OUT = EPI == EPI_RESID || EPI == EPI_F32 || EPI == EPI_F32_ACCUM;
    g.A = g0.A + (size_t)blockIdx.z * g0.sa;
    g.B = g0.B + (size_t)blockIdx.z * g0.sb;
    g.C = F32OUT ? (void*)((float*)g0.C + (size_t)blockIdx.z * g0.sc) : (void*)((bf16_t*)g0.C + (size_t)blockIdx.z * g0.sc);
  }
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

  // ---- epilogue
  if (VEC) {
    // Stage the wave's 64x64 fp32 sub-tile through its private 16 KiB of LDS (the operand images are dead after the
    // loop's last barrier) so that every global access of the epilogue is a 16-byte, row-contiguous vector:
    // the C layout (lane -> column, regs -> rows) would otherwise issue one 4-byte store per element.
    float* wl = reinterpret_cast<float*>(smem + wave * 16384);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          wl[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 64 + j * 32 + (l & 31)] = acc[i][j][e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    emit_subtile<EPI>(g, wl, m0 + wm * 64, n0 + wn * 64, l, gridDim.y > 1);
    return;
  }
  // scalar fallback (N or a leading dimension not a multiple of 4): C layout of the 32x32 MFMA is
  // col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
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
          ((float*)g.C)[o] = rs + g.branch_scale * (g.rowscale ? g.rowscale[row] : 1.f) * gam * v;
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

// =====================================================================================================================
// 256 x 256 x 64 tile, 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA 32x32 tiles), one workgroup per CU, operands brought in
// by LDS-DMA (global_load_lds_dwordx4: HBM -> LDS without the VGPR round trip or the ds_write pass, which at
// ~80 B/clk/CU costs as many LDS cycles as all fragment reads of the 128^2 kernel).  The DMA destination is
// wave-linear (M0 base + lane*16), so both LDS images are produced by permuting the per-lane SOURCE address:
//   row image  : lane -> (row = blk*8 + lane/8, slot = lane%8), source chunk = slot ^ ((row>>1)&7)
//   T image    : lane -> (piece = blk*8 + lane/8, slot), source k-row = (slot/2 - piece%16) & 3, 8-column half = slot&1
// Two LDS stages (128 KiB): tile k+1 streams in while tile k feeds 256 MFMAs; one barrier per k-tile.
// Used for the large token-major GEMMs (forward and dgrad); requires K % 64 == 0 and the vector epilogue;
// out-of-range rows/columns are clamped at the source (their products are never stored).
namespace g256 {
constexpr int BM2 = 256, NT2 = 512;
constexpr int LDS_BYTES = 131072;  // 2 stages x (A 32 KiB + B <=32 KiB); also 8 x 16 KiB epilogue scratch

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// DMA one [ROWS x 64] operand tile into its LDS image (ROWS/8 one-KiB wave-instructions, ROWS/64 per wave)
// KT (K not a multiple of 64: the last K-tile is partial): k indices past the end are clamped to the last valid chunk / row, so
// the tile holds duplicates there instead of whatever follows the operand in memory; the A image's duplicates are then
// overwritten with zeros (zero_tail below), which makes the B image's irrelevant.  `kdim` = K.
template <bool TR, int ROWS, int NWAVES = 8, bool KT = false>
__device__ __forceinline__ void stage_dma(char* lds, const bf16_t* __restrict__ P, int ld, int rows, int row0, int k0, int kdim = 0) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  constexpr int PER_WAVE = ROWS / 8 / NWAVES, NB = ROWS / 16;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int blk = w * PER_WAVE + j;
    const bf16_t* src;
    if (!TR) {
      const int row = blk * 8 + (l >> 3), slot = l & 7;
      const int c = slot ^ ((row >> 1) & 7);
      const int gr = min(row0 + row, rows - 1);
      src = P + (size_t)gr * ld + (KT ? min(k0 + c * 8, kdim - 8) : k0 + c * 8);
    } else {
      const int p = blk * 8 + (l >> 3), slot = l & 7;
      const int q = p / NB, b = p % NB;
      const int kr = ((slot >> 1) - b) & 3;
      int col = row0 + b * 16 + (slot & 1) * 8;
      if (col >= rows) col = 0;
      src = P + (size_t)(KT ? min(k0 + q * 4 + kr, kdim - 1) : k0 + q * 4 + kr) * ld + col;
    }
    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(lds + blk * 1024), 16, 0, 0);
  }
}

// After a wave's DMAs of a partial K-tile have landed (its own vmcnt wait): zero the 16-byte pieces of the K-contiguous image whose
// source chunk lies past K -- each lane overwrites exactly the piece its own DMA instruction wrote (same address arithmetic as
// stage_dma<false>), so no other wave's in-flight DMA can undo it.  `ctail` = first invalid chunk of the tile = (K % 64) / 8.
template <int ROWS, int NWAVES = 8>
__device__ __forceinline__ void zero_tail(char* lds, int ctail) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  constexpr int PER_WAVE = ROWS / 8 / NWAVES;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int blk = w * PER_WAVE + j;
    const int row = blk * 8 + (l >> 3), slot = l & 7;
    const int c = slot ^ ((row >> 1) & 7);
    if (c >= ctail) *reinterpret_cast<uint4*>(lds + blk * 1024 + l * 16) = make_uint4(0, 0, 0, 0);
  }
}

// RAW_TR: issue the LDS reads as inline asm.  The waitcnt pass cannot see which LDS bytes ds_read_b64_tr_b16 touches, so
// after any LDS-DMA it puts `s_waitcnt vmcnt(0)` in front of the builtin form -- which in a K-loop that keeps DMAs in flight
// across phases serialises every phase behind the DMA issued just before it; and before every LDS-DMA it waits lgkmcnt(0)
// for the tracked ds_reads issued just before (possible WAR through LDS), so the DMA issue cannot overlap their latency.
// Callers of the RAW_TR form order DMA and reads themselves (explicit vmcnt / lgkmcnt waits + barriers) and must wait
// lgkmcnt(0) before using the result.
template <bool TR, int ROWS, bool RAW_TR = false>
__device__ __forceinline__ bf16x8 read_frag2(const char* lds, int rb, int ks) {
  const int l = threadIdx.x & 63;
  if (!TR) {
    const int row = rb * 32 + (l & 31);
    const int c = ks * 2 + (l >> 5);
    const char* p = lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    if (RAW_TR) {   // same reason, mirrored: a tracked ds_read makes the pass wait lgkmcnt(0) before the next LDS-DMA is issued
      bf16x8 v;
      asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)(lptr_t*)p));
      return v;
    }
    return *reinterpret_cast<const bf16x8*>(p);
  } else {
    constexpr int NB = ROWS / 16;
    const int i = l & 15, cb = (l >> 4) & 1, kh = l >> 5;
    const int b = rb * 2 + cb;
    const int inner = ((((i >> 2) + b) & 3) << 5) + ((i & 3) << 3);
    const int q0 = ks * 4 + kh * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo, hi;
    if (RAW_TR) {
      const unsigned a0 = (unsigned)(uintptr_t)(lptr_t*)(lds + (q0 * NB + b) * 128 + inner);
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a0), "n"(NB * 128));
    } else {
      lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (q0 * NB + b) * 128 + inner));
      hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + ((q0 + 1) * NB + b) * 128 + inner));
    }
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
  }
}

// BN = 256: waves 2(M) x 4(N), wave tile 128 x 64;  BN = 128: waves 4 x 2, wave tile 64 x 64.
// SLAB: split-K slice blockIdx.y writes its fp32 partial tile to slab[blockIdx.y] (C2 = slab base), no atomics.
template <bool TA, bool TB, int EPI, int BN, bool SLAB>
__global__ __launch_bounds__(NT2) void gemm256_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WN = BN / 64, WM = 8 / WN, MI = BM2 / (WM * 32);
  constexpr int A_BYTES = BM2 * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int kbeg = blockIdx.y * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = (kend - kbeg) / BK;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int wm = wave / WN, wn = wave % WN;
  // one tile per block (gridDim.x == ntiles); a smaller grid (multiple of 8) would walk the tiles persistently
  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
  int id;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, j = v >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    if (j >= q + (xcd < r ? 1 : 0)) continue;  // persistent grids are rounded up to a multiple of 8
  }
  const int tm = id / g.tiles_n, tn = id % g.tiles_n;
  const int m0 = tm * BM2, n0 = tn * BN;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nk > 0) {
    stage_dma<TA, BM2>(smem, g.A, g.lda, g.M, m0, kbeg);
    stage_dma<TB, BN>(smem + A_BYTES, g.B, g.ldb, g.N, n0, kbeg);
  }
  __syncthreads();  // hipcc drains the outstanding LDS-DMA (vmcnt(0)) in front of the barrier
  for (int kt = 0; kt < nk; ++kt) {
    const char* la = smem + (kt & 1) * STAGE;
    const char* lb = la + A_BYTES;
    if (kt + 1 < nk) {
      char* na = smem + ((kt + 1) & 1) * STAGE;
      stage_dma<TA, BM2>(na, g.A, g.lda, g.M, m0, kbeg + (kt + 1) * BK);
      stage_dma<TB, BN>(na + A_BYTES, g.B, g.ldb, g.N, n0, kbeg + (kt + 1) * BK);
    }
    // fragments double-buffered in registers: the ds_reads of k16-step ks+1 are in flight while step ks feeds the MFMAs
    bf16x8 fa[2][MI], fb[2][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[0][i] = read_frag2<TA, BM2>(la, wm * MI + i, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[0][j] = read_frag2<TB, BN>(lb, wn * 2 + j, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[(ks + 1) & 1][i] = read_frag2<TA, BM2>(la, wm * MI + i, ks + 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[(ks + 1) & 1][j] = read_frag2<TB, BN>(lb, wn * 2 + j, ks + 1);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this step's MFMAs (the scheduler would sink it to its use)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  // epilogue: 64-row halves of the wave tile through the wave's private 16 KiB of LDS
  GemmArgs ge = g;
  if (SLAB) {  // partial tile -> slab of this k-slice, plain fp32 stores
    ge.C = (float*)g.C2 + (size_t)blockIdx.y * g.M * g.N;
    ge.ldc = g.N; ge.alpha = 1.f; ge.bias = nullptr;
  }
  float* wl = reinterpret_cast<float*>(smem + wave * 16384);
#pragma unroll
  for (int h = 0; h < MI / 2; ++h) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          wl[(ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 64 + j * 32 + (l & 31)] = acc[h * 2 + ii][j][e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (SLAB) emit_subtile<EPI_F32>(ge, wl, m0 + wm * (MI * 32) + h * 64, n0 + wn * 64, l, false);
    else emit_subtile<EPI>(ge, wl, m0 + wm * (MI * 32) + h * 64, n0 + wn * 64, l, gridDim.y > 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();  // every wave is done with its epilogue scratch before the next tile's DMA overwrites it
  }
}

// ---- 4-phase ping-pong K-loop ("q" kernel) ------------------------------------------------------------------------------
// Same 256 x 256 x 64 tile, wave layout (2 x 4, wave tile 128 x 64), LDS images and epilogue as gemm256_kernel; only the K-loop
// schedule differs.  Each operand stage is four 16-KiB half-tiles [A0: A rows 0-127 | A1: A rows 128-255 | B0: B cols 0-127 |
// B1: B cols 128-255] (a wave reads one A half and one B half); a K-tile is computed in four phases, one 64 x 32 output
// quadrant of the wave tile (8 MFMAs over the whole BK = 64) per phase:
//   P1: read B-sub0 + A-sub0 | DMA A1 of tile t+1          | quadrant (0,0)
//   P2: read B-sub1          | -                           | quadrant (0,1)      last read of the B halves of this stage
//   P3: read A-sub1          | DMA B0 of tile t+2          | quadrant (1,1)      last read of the A halves of this stage
//   P4: -                    | DMA B1, A0 of tile t+2      | quadrant (1,0)      + the only vmcnt wait: vmcnt(6), never 0
// Every half-tile is re-filled as early as its stage allows (>= 1 phase after its last ds_read), so the youngest DMA of tile
// t+1 has had three phases to land when P4 of tile t waits for it (the first version issued all of tile t+1 during tile t
// and waited one phase after the last issue: ~1.5 % slower in the step).
// Phase = [load segment; lgkmcnt(0); s_barrier; MFMA segment at raised priority; s_barrier].  The second wave of every SIMD
// (wm = 1: waves 4-7) runs one barrier behind the first, so on each SIMD one wave's MFMA segment always overlaps its
// partner's LDS-read / DMA-issue segment.  Hazards: a half-tile is re-filled >= 1 phase after its last ds_read, whose
// lgkmcnt(0) precedes the reader's first barrier of that phase (WAR); every wave's vmcnt wait precedes its first barrier of
// P4 and the data is first read in the next phase (RAW, one barrier more for the staggered group).  LDS-DMA returns in issue
// order, which is what makes the counted vmcnt wait meaningful.
// KT: K % 64 != 0 (K % 8 == 0): forward / dgrad layouts only (A K-contiguous).  The last K-tile is loaded with clamped k indices and
// the A halves' tail pieces are zeroed by the waves that loaded them, right after the vmcnt wait that retires the tile and before
// the barrier that publishes it (SwiGLU widths: 2736 = 42.75 tiles, 5472).
// CS (weight gradients, TA: A = dY stored [K][M]): the kernel also leaves the column sums of dY -- the bias gradient of the same Linear
// (LT .. layers/attention.py:44, mlp.py:36: nn.Linear(bias=True) backward) -- as partial rows g.cs[(slice * tiles_n + tn) * 4 + wn][M].
// The four waves of a wave row read the same A fragments and the tiles_n workgroups of a tile row the same A tiles, so K-tile t of a
// slice is summed by exactly one of them: wave column t % 4 of column tile (t / 4) % tiles_n.  A lane's fragment holds 8 consecutive
// k of ONE output row m, i.e. the sum over k is lane-local (v_dot2c_f32_bf16 against (1, 1): 4 instructions per fragment, 32 per phase
// of the one K-tile in 4 * tiles_n that is this wave's) -- the separate column-sum pass re-read dY from HBM (3.1 ms per step).
template <bool TA, bool TB, int EPI, bool SLAB, bool KT = false, bool CS = false>
__global__ __launch_bounds__(NT2) void gemm256q_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HB = 128 * 128, BUF = 4 * HB;
  const int ntiles = g.tiles_m * g.tiles_n;
  // XCD-contiguous work order over (k-slice, tile): workgroups go round-robin to the 8 XCDs in launch order (x fastest), so
  // workgroup L runs on XCD L % 8; giving XCD c the c-th contiguous range of the work list (tile fastest, tn fastest inside)
  // makes the ~32 tiles an XCD holds at a time share their A panels (same tm) and all B panels of one k-slice in its L2.
  int id, slice;
  {
    const int W = ntiles * (int)gridDim.y, L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    const int q = W >> 3, r = W & 7, xcd = L & 7, j = L >> 3;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    slice = w / ntiles; id = w - slice * ntiles;
  }
  // tile order inside a k-slice: bands of `band` column tiles, row-major inside a band, so that the ~32 tiles an XCD holds at a time
  // form a block of (32 / band) tile rows x band tile columns sharing A AND B panels in its L2, instead of 2.7 rows x all 12 columns
  // (the [N, K] weight of the N = 2304 / 3072 GEMMs does not fit one 4 MiB L2 beside the A panels and was re-fetched per tile row)
  int tm, tn;
  if (g.band > 0 && g.band < g.tiles_n) {
    const int per_band = g.tiles_m * g.band, full_b = g.tiles_n / g.band;
    if (id < full_b * per_band) { const int bnd = id / per_band, r = id - bnd * per_band; tm = r / g.band; tn = bnd * g.band + (r - tm * g.band); }
    else { const int rem_b = g.tiles_n - full_b * g.band, r = id - full_b * per_band; tm = r / rem_b; tn = full_b * g.band + (r - tm * rem_b); }
  } else { tm = id / g.tiles_n; tn = id % g.tiles_n; }
  const int m0 = tm * 256, n0 = tn * 256;
  const int kbeg = slice * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = KT ? (kend - kbeg + BK - 1) / BK : (kend - kbeg) / BK;
  const int ctail = KT ? ((g.K & (BK - 1)) >> 3) : 8;   // first invalid 8-element chunk of the last tile (8: none)
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
#ifdef LT_GEMM_TIMING
  if (g.sc > 0) {   // diagnostic build only: start delay of the first resident workgroups, (g.sc & 0xff) x 0.54 us per group, (g.sc >> 8) groups
    const int L_ = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    if (L_ < 256) {
      const int n_ = ((L_ >> 3) % (int)(g.sc >> 8)) * (int)(g.sc & 0xff);
      for (int i_ = 0; i_ < n_; ++i_) __builtin_amdgcn_s_sleep(16);
    }
  }
#endif
  LT_TSTAMP(0);
#ifdef LT_GEMM_TIMING
  if (threadIdx.x == 0) {
    const int L_ = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    if (L_ < 16384) {
      lt_gemm_timing_buf[L_ * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
      lt_gemm_timing_buf[L_ * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
  }
#endif

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define LT_DMA_HALF(TILE, H)                                                                                     \
  do {                                                                                                           \
    char* dst_ = smem + ((TILE) & 1) * BUF + (H) * HB;                                                           \
    const int k0_ = kbeg + (TILE) * BK;                                                                          \
    if ((H) < 2) stage_dma<TA, 128, 8, KT>(dst_, g.A, g.lda, g.M, m0 + (H) * 128, k0_, g.K);                     \
    else stage_dma<TB, 128, 8, KT>(dst_, g.B, g.ldb, g.N, n0 + ((H) - 2) * 128, k0_, g.K);                       \
  } while (0)
// the partial last tile has landed (this wave's share): clear the A pieces past K
#define LT_ZERO_TAIL(TILE)                                                   \
  do {                                                                       \
    if (KT && ctail != 0 && (TILE) == nk - 1) {                              \
      zero_tail<128>(smem + ((TILE) & 1) * BUF, ctail);                      \
      zero_tail<128>(smem + ((TILE) & 1) * BUF + HB, ctail);                 \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                    \
    }                                                                        \
  } while (0)
#define LT_PHASE_SYNC_IN()                      \
  do {                                          \
    __builtin_amdgcn_s_waitcnt(0xC07F);         \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_setprio(1);              \
  } while (0)
#define LT_PHASE_SYNC_OUT()                     \
  do {                                          \
    __builtin_amdgcn_s_setprio(0);              \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)

  if (nk > 0) {
    LT_DMA_HALF(0, 0); LT_DMA_HALF(0, 1); LT_DMA_HALF(0, 2); LT_DMA_HALF(0, 3);
  }
  if (nk > 1) { LT_DMA_HALF(1, 2); LT_DMA_HALF(1, 3); LT_DMA_HALF(1, 0); __builtin_amdgcn_s_waitcnt(0xF76); }  // vmcnt(6): tile 0 landed
  else __builtin_amdgcn_s_waitcnt(0xF70);                                                                      // vmcnt(0)
  LT_ZERO_TAIL(0);
  // lgkmcnt(0) on every path into the loop: otherwise the waitcnt pass has to assume the kernel-argument loads may still be
  // pending at the loop header and waits lgkmcnt(0) before the first DMA address computation of every iteration -- which,
  // the counter being shared, also waits for the LDS reads just issued
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  LT_TSTAMP(1);
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of each SIMD by one barrier

  bf16x8 fa[2][4], fb0[4], fb1[4];
  const int bcol = (wn & 1) * 2;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};                 // CS: per 32-row block of the wave's 128 A rows, this lane's k half
  const int cs_period = 4 * g.tiles_n, cs_mine = __builtin_amdgcn_readfirstlane(tn * 4 + wn);   // wave-uniform: a scalar compare + branch per phase
  int cs_cnt = 0;
#define LT_CS_ADD(I0)                                                                                             \
  do {                                                                                                            \
    if (CS && cs_cnt == cs_mine) {                                                                                \
      const bf16x2_t one_ = {(__bf16)1.0f, (__bf16)1.0f};                                                         \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                            \
      _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                                                       \
        const bf16x8 f_ = fa[i_][ks_];                                                                            \
        float c_ = csum[(I0) + i_];                                                                               \
        c_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f_, f_, 0, 1), one_, c_, false);             \
        c_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f_, f_, 2, 3), one_, c_, false);             \
        c_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f_, f_, 4, 5), one_, c_, false);             \
        c_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f_, f_, 6, 7), one_, c_, false);             \
        csum[(I0) + i_] = c_;                                                                                     \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
  for (int t = 0; t < nk; ++t) {
    const char* buf = smem + (t & 1) * BUF;
    const char* la = buf + wm * HB;
    const char* lb = buf + (2 + (wn >> 1)) * HB;
    // ---- P1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb0[ks] = read_frag2<TB, 128, true>(lb, bcol, ks);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[i][ks] = read_frag2<TA, 128, true>(la, i, ks);
    if (t + 1 < nk) LT_DMA_HALF(t + 1, 1);
    LT_PHASE_SYNC_IN();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb0[ks], acc[i][0], 0, 0, 0);
    LT_PHASE_SYNC_OUT();
    // ---- P2
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb1[ks] = read_frag2<TB, 128, true>(lb, bcol + 1, ks);
    LT_PHASE_SYNC_IN();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb1[ks], acc[i][1], 0, 0, 0);
    LT_CS_ADD(0);   // A rows 0-63 of the wave's half (read in P1, last used here)
    LT_PHASE_SYNC_OUT();
    // ---- P3
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) fa[i][ks] = read_frag2<TA, 128, true>(la, 2 + i, ks);
    if (t + 2 < nk) LT_DMA_HALF(t + 2, 2);
    LT_PHASE_SYNC_IN();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb1[ks], acc[2 + i][1], 0, 0, 0);
    LT_PHASE_SYNC_OUT();
    // ---- P4
    // tile t+1 = [B0, B1, A0 issued one K-tile ago, A1 issued in P1] is complete once at most the three half-tiles issued
    // since (6 DMA instructions) are outstanding: its youngest DMA has had three phases to land
    if (t + 2 < nk) { LT_DMA_HALF(t + 2, 3); LT_DMA_HALF(t + 2, 0); __builtin_amdgcn_s_waitcnt(0xF76); }  // vmcnt(6)
    else __builtin_amdgcn_s_waitcnt(0xF70);
    LT_ZERO_TAIL(t + 1);
    LT_PHASE_SYNC_IN();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb0[ks], acc[2 + i][0], 0, 0, 0);
    LT_CS_ADD(2);   // A rows 64-127 (read in P3)
    if (CS) cs_cnt = (cs_cnt + 1 == cs_period) ? 0 : cs_cnt + 1;
    LT_PHASE_SYNC_OUT();
  }
#undef LT_CS_ADD
  if (wm == 0) __builtin_amdgcn_s_barrier();  // re-align the two wave groups
#undef LT_DMA_HALF
#undef LT_ZERO_TAIL
#undef LT_PHASE_SYNC_IN
#undef LT_PHASE_SYNC_OUT
  __syncthreads();
  LT_TSTAMP(2);
  if (CS) {   // lanes l and l + 32 hold the two k halves of row l & 31: combine, then one 128-byte store per 32-row block
    float* dst = g.cs + (size_t)((slice * g.tiles_n + tn) * 4 + wn) * g.M;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
      const int row = m0 + wm * 128 + i * 32 + (l & 31);
      if (l < 32 && row < g.M) dst[row] = v;
    }
  }
  GemmArgs ge = g;
  if (SLAB) {
    ge.C = (float*)g.C2 + (size_t)slice * g.M * g.N;
    ge.ldc = g.N; ge.alpha = 1.f; ge.bias = nullptr;
  }
  float* wl = reinterpret_cast<float*>(smem + wave * 16384);
#define LT_STAGE_HALF(H)                                                                                          \
  do {                                                                                                            \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
    _Pragma("unroll") for (int e = 0; e < 16; ++e)                                                                \
      wl[(ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 64 + j * 32 + (l & 31)] = acc[(H) * 2 + ii][j][e];   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                        \
    __builtin_amdgcn_wave_barrier();                                                                              \
  } while (0)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    LT_STAGE_HALF(h);
    if (SLAB) emit_subtile<EPI_F32>(ge, wl, m0 + wm * 128 + h * 64, n0 + wn * 64, l, false);
    else emit_subtile<EPI>(ge, wl, m0 + wm * 128 + h * 64, n0 + wn * 64, l, gridDim.y > 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#undef LT_STAGE_HALF
#ifdef LT_GEMM_TIMING
  LT_TSTAMP(3);                                  // wave 0 has issued its last store
  __builtin_amdgcn_s_waitcnt(0xF70);
  LT_TSTAMP(6);                                  // ... and they have been acknowledged
  __syncthreads();
  LT_TSTAMP(7);                                  // every wave is through its epilogue
#endif
}


// ---- static-address K-loop ("e" kernel, round 6) ---------------------------------------------------------------------------
// Same tile, wave layout, phases, hazards and epilogue as gemm256q_kernel; what changes is what a wave has to EXECUTE in its load
// segments.  In the q kernel every fragment read carried a v_add for its address and every LDS-DMA piece five VALU instructions
// (64-bit source address, readfirstlane of the destination) -- ~60 VALU per K-tile issued while the SIMD's other wave runs its MFMA
// segment at raised priority, which is where VALU is dearest (MI355X_MICROARCH.md, "Two waves per SIMD" item 2: a prioritised partner
// makes 16 VALU in a load segment cost 300-400 cycles).  Here the load segments contain NO vector ALU work:
//   * the K-loop is unrolled over the two LDS stages, so every LDS address is a loop-invariant per-lane base (<= 8 VGPRs, computed
//     once) plus an immediate `offset:` field; the half-tile buffers are laid out [half][stage] (16 KiB each) so that both stages of
//     the half a wave reads lie inside the 16-bit immediate range of one base;
//   * operands arrive by `buffer_load_dwordx4 ... offen lds` through a wave-uniform descriptor: the per-lane source offsets
//     (swizzle, row clamps) are loop-invariant VGPRs, the K advance and the piece index live in the scalar offset, the LDS
//     destination in M0 comes from scalar adds.
// Restrictions (dispatcher falls back to the q kernel): K % 64 == 0 (no partial K-tile variant), operand panels addressable with
// 31-bit byte offsets from the workgroup's base.
template <bool TR, int OFF>
__device__ __forceinline__ bf16x8 lds_frag(unsigned a) {
  if constexpr (!TR) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
    return v;
  } else {
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 1024));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
  }
}
// the four k16 steps of one 32-row block RB of the half-tile at byte offset BUF0 from the wave's base registers
//   K-contiguous image: base[ks] differs per step (the XOR swizzle is not additive), immediate = BUF0 + RB * 4096
//   transposed image  : base[RB & 1] (the k-row rotation depends on the block's parity), immediate = BUF0 + ks * 4096 + RB * 256
template <bool TR, int BUF0, int RB>
__device__ __forceinline__ void lds_frag4(bf16x8 (&f)[4], const unsigned (&base)[4]) {
  if constexpr (!TR) {
    f[0] = lds_frag<false, BUF0 + RB * 4096>(base[0]);
    f[1] = lds_frag<false, BUF0 + RB * 4096>(base[1]);
    f[2] = lds_frag<false, BUF0 + RB * 4096>(base[2]);
    f[3] = lds_frag<false, BUF0 + RB * 4096>(base[3]);
  } else {
    f[0] = lds_frag<true, BUF0 + 0 * 4096 + RB * 256>(base[RB & 1]);
    f[1] = lds_frag<true, BUF0 + 1 * 4096 + RB * 256>(base[RB & 1]);
    f[2] = lds_frag<true, BUF0 + 2 * 4096 + RB * 256>(base[RB & 1]);
    f[3] = lds_frag<true, BUF0 + 3 * 4096 + RB * 256>(base[RB & 1]);
  }
}

// loop-invariant source side of one operand's LDS-DMA
struct DmaSrcE {
  __amdgpu_buffer_rsrc_t srd;   // base = the workgroup's first row (column) of the operand at the slice's first k
  unsigned voff[4];             // per-lane byte offsets: K-contiguous [half * 2 + piece]; transposed [half] (pieces differ by `jstep`)
  int wave_off;                 // transposed: byte offset of this wave's first k-row group (w * 8 rows)
  int tile_step;                // bytes per K-tile
  int jstep;                    // transposed: bytes between the two pieces of a wave (4 k-rows)
  unsigned voff_tail[4];        // KT: per-lane byte offsets [half * 2 + piece] of the partial LAST K-tile (k indices clamped into the operand);
                                // used with the scalar offset tile * tile_step alone (no wave_off / jstep)
};
// KT (K % 64 != 0, forward / dgrad layouts): the last K-tile is loaded with clamped k indices -- duplicates instead of whatever follows the
// operand in memory -- and the A image's duplicates are overwritten with zeros by the lanes that loaded them (see the kernel), which makes
// the B image's irrelevant.  `krem` = K % 64 = valid k of the last tile.
template <bool TR>
__device__ __forceinline__ void make_dma_tail(DmaSrcE& d, int ld, int rows, int row0, int krem, int w, int l) {
  if (!TR) {
    const int ctail = krem >> 3;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = hh * 128 + (w * 2 + j) * 8 + (l >> 3), slot = l & 7;
        const int c = min(slot ^ ((row >> 1) & 7), ctail - 1);
        const int gr = min(row0 + row, rows - 1) - row0;
        d.voff_tail[hh * 2 + j] = (unsigned)gr * (unsigned)ld * 2u + (unsigned)c * 16u;
      }
  } else {
    const int b = l >> 3, slot = l & 7;
    const int kr = ((slot >> 1) - b) & 3;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      int col = row0 + hh * 128 + b * 16 + (slot & 1) * 8;
      if (col >= rows) col = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        d.voff_tail[hh * 2 + j] = (unsigned)min(w * 8 + j * 4 + kr, krem - 1) * (unsigned)ld * 2u + (unsigned)col * 2u;
    }
  }
}
template <bool TR>
__device__ __forceinline__ DmaSrcE make_dma_src(const bf16_t* P, int ld, int rows, int row0, int kbeg, int w, int l) {
  DmaSrcE d;
  if (!TR) {
    const bf16_t* base = P + (size_t)row0 * ld + kbeg;
    d.srd = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = hh * 128 + (w * 2 + j) * 8 + (l >> 3), slot = l & 7;
        const int c = slot ^ ((row >> 1) & 7);
        const int gr = min(row0 + row, rows - 1) - row0;
        d.voff[hh * 2 + j] = (unsigned)gr * (unsigned)ld * 2u + (unsigned)c * 16u;
      }
    d.wave_off = 0; d.tile_step = BK * 2; d.jstep = 0;
  } else {
    const bf16_t* base = P + (size_t)kbeg * ld;
    d.srd = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const int b = l >> 3, slot = l & 7;
    const int kr = ((slot >> 1) - b) & 3;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      int col = row0 + hh * 128 + b * 16 + (slot & 1) * 8;
      if (col >= rows) col = 0;
      d.voff[hh] = (unsigned)kr * (unsigned)ld * 2u + (unsigned)col * 2u;
      d.voff[2 + hh] = 0;
    }
    d.wave_off = w * 8 * ld * 2; d.tile_step = BK * ld * 2; d.jstep = 4 * ld * 2;
  }
  return d;
}

template <bool TA, bool TB, int EPI, bool SLAB, bool CS = false, int PH = 4, bool KT = false>
__global__ __launch_bounds__(NT2) void gemm256e_kernel(const GemmArgs g) {
  static_assert(!KT || (!TA && !SLAB && !CS), "partial last K-tile: forward / dgrad layouts only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HB = 128 * 128;            // one half-tile buffer; buffer index = half * 2 + stage (halves: A0 A1 B0 B1)
  const int ntiles = g.tiles_m * g.tiles_n;
  int id, slice;
  {
    const int W = ntiles * (int)gridDim.y, L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    const int q = W >> 3, r = W & 7, xcd = L & 7, j = L >> 3;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    slice = w / ntiles; id = w - slice * ntiles;
  }
  int tm, tn;
  if (g.band > 0 && g.band < g.tiles_n) {
    const int per_band = g.tiles_m * g.band, full_b = g.tiles_n / g.band;
    if (id < full_b * per_band) { const int bnd = id / per_band, r = id - bnd * per_band; tm = r / g.band; tn = bnd * g.band + (r - tm * g.band); }
    else { const int rem_b = g.tiles_n - full_b * g.band, r = id - full_b * per_band; tm = r / rem_b; tn = full_b * g.band + (r - tm * rem_b); }
  } else { tm = id / g.tiles_n; tn = id % g.tiles_n; }
  const int m0 = tm * 256, n0 = tn * 256;
  const int kbeg = slice * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nk = KT ? (kend - kbeg + BK - 1) / BK : (kend - kbeg) / BK;
  const int l = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: everything derived from it stays scalar
  const int wm = wave >> 2, wn = wave & 3;
  LT_TSTAMP(0);
#ifdef LT_GEMM_TIMING
  if (threadIdx.x == 0) {
    const int L_ = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    if (L_ < 16384) {
      lt_gemm_timing_buf[L_ * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
      lt_gemm_timing_buf[L_ * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
  }
#endif

  // ---- loop-invariant addresses
  DmaSrcE sa = make_dma_src<TA>(g.A, g.lda, g.M, m0, kbeg, wave, l);
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  DmaSrcE sb = make_dma_src<TB>(g.B, g.ldb, g.N, n0, kbeg, wave, l);
  const int krem = KT ? (g.K & (BK - 1)) : 0;        // valid k of the last tile (0: the last tile is whole)
  bool ztail[2] = {false, false};                    // KT: this lane's piece j of an A half lies past K in the last tile
  if (KT && krem) {
    make_dma_tail<TA>(sa, g.lda, g.M, m0, krem, wave, l);
    make_dma_tail<TB>(sb, g.ldb, g.N, n0, krem, wave, l);
#pragma unroll
    for (int j = 0; j < 2; ++j) ztail[j] = (((l & 7) ^ ((j * 4 + (l >> 4)) & 7))) >= (krem >> 3);
#pragma unroll
    for (int k = 0; k < 4; ++k) { asm volatile("" : "+v"(sa.voff_tail[k])); asm volatile("" : "+v"(sb.voff_tail[k])); }
  }
  unsigned ab[4], bb[4];
  {
    const unsigned a_half = (unsigned)(wm * 2) * HB, b_half = (unsigned)((2 + (wn >> 1)) * 2) * HB;
    const int bcol = (wn & 1) * 2;
    if (!TA) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ab[ks] = a_half + (l & 31) * 128 + ((((ks * 2 + (l >> 5)) ^ (((l & 31) >> 1) & 7))) << 4);
    } else {
      const int i = l & 15, cb = (l >> 4) & 1, kh = l >> 5;
#pragma unroll
      for (int p = 0; p < 2; ++p) ab[p] = a_half + kh * 2048 + cb * 128 + ((((i >> 2) + cb + 2 * p) & 3) << 5) + ((i & 3) << 3);
      ab[2] = ab[3] = 0;
    }
    if (!TB) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bb[ks] = b_half + bcol * 4096 + (l & 31) * 128 + ((((ks * 2 + (l >> 5)) ^ (((l & 31) >> 1) & 7))) << 4);
    } else {
      const int i = l & 15, cb = (l >> 4) & 1, kh = l >> 5;
#pragma unroll
      for (int p = 0; p < 2; ++p) bb[p] = b_half + bcol * 256 + kh * 2048 + cb * 128 + ((((i >> 2) + cb + 2 * p) & 3) << 5) + ((i & 3) << 3);
      bb[2] = bb[3] = 0;
    }
    // opaque to the optimiser from here on: no re-derivation of these inside the loop
#pragma unroll
    for (int k = 0; k < 4; ++k) { asm volatile("" : "+v"(ab[k])); asm volatile("" : "+v"(bb[k])); asm volatile("" : "+v"(sa.voff[k])); asm volatile("" : "+v"(sb.voff[k])); }
  }
  const int lds_w = wave * 2048;   // this wave's two 1-KiB pieces inside every half-tile buffer

  // one half-tile (H: 0 A rows 0-127, 1 A rows 128-255, 2 B cols 0-127, 3 B cols 128-255) of K-tile TILE into stage ST
#define LT_E_DMA(TILE, ST, H)                                                                                             \
  do {                                                                                                                    \
    const DmaSrcE& s_ = (H) < 2 ? sa : sb;                                                                                \
    constexpr bool TR_ = (H) < 2 ? TA : TB;                                                                               \
    char* d_ = smem + ((H) * 2 + (ST)) * HB + lds_w;                                                                      \
    if (KT && krem && (TILE) == nk - 1) {   /* the partial last tile: clamped sources (scalar branch) */                    \
      const int so_ = (TILE) * s_.tile_step;                                                                              \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s_.srd, (lptr_t*)d_, 16, s_.voff_tail[((H) & 1) * 2], so_, 0, 0);           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s_.srd, (lptr_t*)(d_ + 1024), 16, s_.voff_tail[((H) & 1) * 2 + 1], so_, 0, 0); \
    } else {                                                                                                              \
      const int so_ = (TILE) * s_.tile_step + s_.wave_off;                                                                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s_.srd, (lptr_t*)d_, 16, TR_ ? s_.voff[(H) & 1] : s_.voff[((H) & 1) * 2], so_, 0, 0);                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(s_.srd, (lptr_t*)(d_ + 1024), 16, TR_ ? s_.voff[(H) & 1] : s_.voff[((H) & 1) * 2 + 1], so_ + s_.jstep, 0, 0); \
    }                                                                                                                     \
  } while (0)
  // after this wave's DMAs of the partial last tile have landed (its own vmcnt wait) and before the barrier that publishes the tile: every
  // lane zeroes the 16-byte pieces of the two A halves it loaded itself whose source chunk lies past K
#define LT_E_ZERO_TAIL(TILE, ST)                                                                                          \
  do {                                                                                                                    \
    if (KT && krem && (TILE) == nk - 1) {                                                                                 \
      _Pragma("unroll") for (int hh_ = 0; hh_ < 2; ++hh_)                                                                 \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                    \
        if (ztail[j_]) *reinterpret_cast<uint4*>(smem + (hh_ * 2 + (ST)) * HB + lds_w + j_ * 1024 + l * 16) = make_uint4(0, 0, 0, 0); \
    }                                                                                                                     \
  } while (0)
#define LT_E_SYNC_IN()                          \
  do {                                          \
    __builtin_amdgcn_s_waitcnt(0xC07F);         \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_setprio(1);              \
  } while (0)
#define LT_E_SYNC_OUT()                         \
  do {                                          \
    __builtin_amdgcn_s_setprio(0);              \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_s_barrier();               \
    asm volatile("" ::: "memory");              \
  } while (0)

  if (nk > 0) { LT_E_DMA(0, 0, 0); LT_E_DMA(0, 0, 1); LT_E_DMA(0, 0, 2); LT_E_DMA(0, 0, 3); }
  if (PH == 4) {
    if (nk > 1) { LT_E_DMA(1, 1, 2); LT_E_DMA(1, 1, 3); LT_E_DMA(1, 1, 0); __builtin_amdgcn_s_waitcnt(0xF76); }  // vmcnt(6): tile 0 landed
    else __builtin_amdgcn_s_waitcnt(0xF70);
  } else {
    if (nk > 1) { LT_E_DMA(1, 1, 2); LT_E_DMA(1, 1, 3); __builtin_amdgcn_s_waitcnt(0xF74); }                     // vmcnt(4)
    else __builtin_amdgcn_s_waitcnt(0xF70);
  }
  LT_E_ZERO_TAIL(0, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  LT_TSTAMP(1);
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of each SIMD by one barrier

  bf16x8 fa[2][4], fb0[4], fb1[4];
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  const int cs_period = 4 * g.tiles_n, cs_mine = tn * 4 + wn;
  int cs_cnt = 0;
#define LT_E_CS_ADD(I0)                                                                                           \
  do {                                                                                                            \
    if (CS && cs_cnt == cs_mine) {                                                                                \
      const bf16x2_t one_ = {(__bf16)1.0f, (__bf16)1.0f};                                                         \
      _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                            \
      _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                                                       \
        const bf16x8 f_ = fa[i_][ks_];                                                                            \
        float c_ = csum[(I0) + i_];                                                                               \
        c_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f_, f_, 0, 1), one_, c_, false);             \
        c_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f_, f_, 2, 3), one_, c_, false);             \
        c_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f_, f_, 4, 5), one_, c_, false);             \
        c_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f_, f_, 6, 7), one_, c_, false);             \
        csum[(I0) + i_] = c_;                                                                                     \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
  // one K-tile in stage ST (a literal): the q kernel's four phases
#define LT_E_TILE(T, ST)                                                                                                   \
  do {                                                                                                                     \
    /* P1 */                                                                                                               \
    lds_frag4<TB, (ST) * HB, 0>(fb0, bb);                                                                                  \
    lds_frag4<TA, (ST) * HB, 0>(fa[0], ab);                                                                                \
    lds_frag4<TA, (ST) * HB, 1>(fa[1], ab);                                                                                \
    if ((T) + 1 < nk) LT_E_DMA((T) + 1, (ST) ^ 1, 1);                                                                      \
    LT_E_SYNC_IN();                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb0[ks], acc[i][0], 0, 0, 0); \
    LT_E_SYNC_OUT();                                                                                                       \
    /* P2 */                                                                                                               \
    lds_frag4<TB, (ST) * HB, 1>(fb1, bb);                                                                                  \
    LT_E_SYNC_IN();                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb1[ks], acc[i][1], 0, 0, 0); \
    LT_E_CS_ADD(0);                                                                                                        \
    LT_E_SYNC_OUT();                                                                                                       \
    /* P3 */                                                                                                               \
    lds_frag4<TA, (ST) * HB, 2>(fa[0], ab);                                                                                \
    lds_frag4<TA, (ST) * HB, 3>(fa[1], ab);                                                                                \
    if ((T) + 2 < nk) LT_E_DMA((T) + 2, (ST), 2);                                                                          \
    LT_E_SYNC_IN();                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb1[ks], acc[2 + i][1], 0, 0, 0); \
    LT_E_SYNC_OUT();                                                                                                       \
    /* P4 */                                                                                                               \
    if ((T) + 2 < nk) { LT_E_DMA((T) + 2, (ST), 3); LT_E_DMA((T) + 2, (ST), 0); __builtin_amdgcn_s_waitcnt(0xF76); }       \
    else __builtin_amdgcn_s_waitcnt(0xF70);                                                                                \
    LT_E_ZERO_TAIL((T) + 1, (ST) ^ 1);                                                                                     \
    LT_E_SYNC_IN();                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb0[ks], acc[2 + i][0], 0, 0, 0); \
    LT_E_CS_ADD(2);                                                                                                        \
    if (CS) cs_cnt = (cs_cnt + 1 == cs_period) ? 0 : cs_cnt + 1;                                                           \
    LT_E_SYNC_OUT();                                                                                                       \
  } while (0)

  // PH == 2: two phases of 16 MFMAs per K-tile (half as many barrier intervals):
  //   Pa: read B-sub0, B-sub1, A-sub0 | DMA A0, A1 of tile t+1 (other stage: its A halves were last read in Pb of tile t-1) | quadrants (0,0) (0,1)
  //   Pb: read A-sub1                 | DMA B0, B1 of tile t+2 (this stage: its B halves were last read in Pa)  + vmcnt(4)   | quadrants (1,1) (1,0)
  // tile t+1 = [B issued in Pb(t-1), A issued in Pa(t)] is complete once only the four DMA instructions of B(t+2) are outstanding; the wait
  // precedes every wave's first barrier of Pb(t) and the data is first read in Pa(t+1).  Per accumulator the MFMA order over k is unchanged.
#define LT_E_TILE2(T, ST)                                                                                                  \
  do {                                                                                                                     \
    /* Pa */                                                                                                               \
    lds_frag4<TB, (ST) * HB, 0>(fb0, bb);                                                                                  \
    lds_frag4<TB, (ST) * HB, 1>(fb1, bb);                                                                                  \
    lds_frag4<TA, (ST) * HB, 0>(fa[0], ab);                                                                                \
    lds_frag4<TA, (ST) * HB, 1>(fa[1], ab);                                                                                \
    if ((T) + 1 < nk) { LT_E_DMA((T) + 1, (ST) ^ 1, 0); LT_E_DMA((T) + 1, (ST) ^ 1, 1); }                                  \
    LT_E_SYNC_IN();                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb0[ks], acc[i][0], 0, 0, 0);                         \
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb1[ks], acc[i][1], 0, 0, 0);                         \
    }                                                                                                                      \
    LT_E_CS_ADD(0);                                                                                                        \
    LT_E_SYNC_OUT();                                                                                                       \
    /* Pb */                                                                                                               \
    lds_frag4<TA, (ST) * HB, 2>(fa[0], ab);                                                                                \
    lds_frag4<TA, (ST) * HB, 3>(fa[1], ab);                                                                                \
    if ((T) + 2 < nk) { LT_E_DMA((T) + 2, (ST), 2); LT_E_DMA((T) + 2, (ST), 3); __builtin_amdgcn_s_waitcnt(0xF74); }       \
    else __builtin_amdgcn_s_waitcnt(0xF70);                                                                                \
    LT_E_ZERO_TAIL((T) + 1, (ST) ^ 1);                                                                                     \
    LT_E_SYNC_IN();                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
      acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb1[ks], acc[2 + i][1], 0, 0, 0);                 \
      acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb0[ks], acc[2 + i][0], 0, 0, 0);                 \
    }                                                                                                                      \
    LT_E_CS_ADD(2);                                                                                                        \
    if (CS) cs_cnt = (cs_cnt + 1 == cs_period) ? 0 : cs_cnt + 1;                                                           \
    LT_E_SYNC_OUT();                                                                                                       \
  } while (0)
  // two K-tiles per iteration (one per LDS stage) in ONE straight-line body with one back-edge; an odd last tile is a peeled copy.  (A
  // `break` between the two copies made the register allocator carry the 128 accumulators through copies and scratch.)
  int t = 0;
  if (PH == 4) {
    for (; t + 1 < nk; t += 2) {
      LT_E_TILE(t, 0);
      LT_E_TILE(t + 1, 1);
    }
    if (t < nk) LT_E_TILE(t, 0);
  } else {
    for (; t + 1 < nk; t += 2) {
      LT_E_TILE2(t, 0);
      LT_E_TILE2(t + 1, 1);
    }
    if (t < nk) LT_E_TILE2(t, 0);
  }
#undef LT_E_TILE
#undef LT_E_TILE2
#undef LT_E_CS_ADD
#undef LT_E_DMA
#undef LT_E_ZERO_TAIL
#undef LT_E_SYNC_IN
#undef LT_E_SYNC_OUT
  if (wm == 0) __builtin_amdgcn_s_barrier();  // re-align the two wave groups
  __syncthreads();
  LT_TSTAMP(2);
  if (CS) {
    float* dst = g.cs + (size_t)((slice * g.tiles_n + tn) * 4 + wn) * g.M;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
      const int row = m0 + wm * 128 + i * 32 + (l & 31);
      if (l < 32 && row < g.M) dst[row] = v;
    }
  }
  GemmArgs ge = g;
  if (SLAB) {
    ge.C = (float*)g.C2 + (size_t)slice * g.M * g.N;
    ge.ldc = g.N; ge.alpha = 1.f; ge.bias = nullptr;
  }
  float* wl = reinterpret_cast<float*>(smem + wave * 16384);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          wl[(ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 64 + j * 32 + (l & 31)] = acc[h * 2 + ii][j][e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (SLAB) emit_subtile<EPI_F32>(ge, wl, m0 + wm * 128 + h * 64, n0 + wn * 64, l, false);
    else emit_subtile<EPI>(ge, wl, m0 + wm * 128 + h * 64, n0 + wn * 64, l, gridDim.y > 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#ifdef LT_GEMM_TIMING
  LT_TSTAMP(3);
  __builtin_amdgcn_s_waitcnt(0xF70);
  LT_TSTAMP(6);
  __syncthreads();
  LT_TSTAMP(7);
#endif
}

// ---- 128-row tiles for the LAST, partly filled round of a 256-row launch (round 6) --------------------------------------------------------
// The N = 768 GEMMs of a ViT-B pass (attention projection, fc2, three data gradients) are 591 tiles of 256 x 256 on 256 CUs: 2.31 rounds, the
// third 31 % full.  With LT_GEMM_TAIL128=1 the dispatcher gives the first two rounds to gemm256e_kernel and the remaining tile rows to this kernel as 128 x 256
// tiles (158 workgroups of half the work: one short round instead of a long, mostly empty one).  Same LDS images, fragment reads, static
// addressing and epilogues as the e kernel; per accumulator the same MFMA order over k (bit-identical results).  What differs:
//   * 8 waves as 2 (M) x 4 (N), wave tile 64 x 64 (64 accumulator registers), ONE A half-tile (128 rows) + two B half-tiles per K-tile;
//   * THREE LDS stages of 48 KiB and one phase of 16 MFMAs per K-tile: the loads of tile t + 2 are issued in the load segment of tile t (its
//     stage held tile t - 1, whose reads were retired before the previous phase's barrier) and waited for one whole tile later (vmcnt(6) in
//     the load segment of tile t + 1's predecessor), i.e. the same one-K-tile slack as the 256-row kernel's;
//   * the second wave of a SIMD is staggered by one barrier as there, so one wave's MFMA segment runs beside its partner's load segment.
// Forward / dgrad layouts (A K-contiguous), K % 64 == 0, no split-K.
template <bool TB, int EPI>
__global__ __launch_bounds__(NT2) void gemm128e_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HB = 128 * 128;            // one half-tile buffer; buffer index = half * 3 + stage (halves: A0 B0 B1)
  const int ntiles = g.tiles_m * g.tiles_n;
  int id;
  {
    const int L = (int)blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = L & 7, j = L >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tm = id / g.tiles_n, tn = id % g.tiles_n;
  const int m0 = tm * 128, n0 = tn * 256;
  const int nk = g.K / BK;
  const int l = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  DmaSrcE sa = make_dma_src<false>(g.A, g.lda, g.M, m0, 0, wave, l);
  DmaSrcE sb = make_dma_src<TB>(g.B, g.ldb, g.N, n0, 0, wave, l);
  unsigned ab[4], bb[4];
  {
    const unsigned b_half = (unsigned)((1 + (wn >> 1)) * 3) * HB;
    const int bcol = (wn & 1) * 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ab[ks] = (unsigned)(wm * 2) * 4096 + (l & 31) * 128 + ((((ks * 2 + (l >> 5)) ^ (((l & 31) >> 1) & 7))) << 4);
    if (!TB) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bb[ks] = b_half + bcol * 4096 + (l & 31) * 128 + ((((ks * 2 + (l >> 5)) ^ (((l & 31) >> 1) & 7))) << 4);
    } else {
      const int i = l & 15, cb = (l >> 4) & 1, kh = l >> 5;
#pragma unroll
      for (int p = 0; p < 2; ++p) bb[p] = b_half + bcol * 256 + kh * 2048 + cb * 128 + ((((i >> 2) + cb + 2 * p) & 3) << 5) + ((i & 3) << 3);
      bb[2] = bb[3] = 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { asm volatile("" : "+v"(ab[k])); asm volatile("" : "+v"(bb[k])); asm volatile("" : "+v"(sa.voff[k])); asm volatile("" : "+v"(sb.voff[k])); }
  }
  const int lds_w = wave * 2048;

  // K-tile TILE into stage ST: the A half-tile and both B half-tiles (6 DMA instructions per wave)
#define LT_S_DMA(TILE, ST)                                                                                                 \
  do {                                                                                                                     \
    const int sa_ = (TILE) * sa.tile_step, sb_ = (TILE) * sb.tile_step + sb.wave_off;                                      \
    char* da_ = smem + (ST) * HB + lds_w;                                                                                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(sa.srd, (lptr_t*)da_, 16, sa.voff[0], sa_, 0, 0);                             \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(sa.srd, (lptr_t*)(da_ + 1024), 16, sa.voff[1], sa_, 0, 0);                    \
    _Pragma("unroll") for (int hh_ = 0; hh_ < 2; ++hh_) {                                                                  \
      char* db_ = smem + ((1 + hh_) * 3 + (ST)) * HB + lds_w;                                                              \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sb.srd, (lptr_t*)db_, 16, TB ? sb.voff[hh_] : sb.voff[hh_ * 2], sb_, 0, 0);  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sb.srd, (lptr_t*)(db_ + 1024), 16, TB ? sb.voff[hh_] : sb.voff[hh_ * 2 + 1], sb_ + sb.jstep, 0, 0); \
    }                                                                                                                      \
  } while (0)

  if (nk > 0) LT_S_DMA(0, 0);
  if (nk > 1) { LT_S_DMA(1, 1); __builtin_amdgcn_s_waitcnt(0xF76); }   // vmcnt(6): tile 0 landed
  else __builtin_amdgcn_s_waitcnt(0xF70);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wm == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of each SIMD by one barrier

  bf16x8 fa[2][4], fb0[4], fb1[4];
  // one K-tile in stage ST (a literal), NXT = the stage tile T + 2 goes to
#define LT_S_TILE(T, ST, NXT)                                                                                              \
  do {                                                                                                                     \
    lds_frag4<TB, (ST) * HB, 0>(fb0, bb);                                                                                  \
    lds_frag4<TB, (ST) * HB, 1>(fb1, bb);                                                                                  \
    lds_frag4<false, (ST) * HB, 0>(fa[0], ab);                                                                             \
    lds_frag4<false, (ST) * HB, 1>(fa[1], ab);                                                                             \
    if ((T) + 2 < nk) { LT_S_DMA((T) + 2, (NXT)); __builtin_amdgcn_s_waitcnt(0xF76); }   /* tile T + 1 landed */             \
    else __builtin_amdgcn_s_waitcnt(0xF70);                                                                                \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                                    \
    asm volatile("" ::: "memory");                                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                          \
    asm volatile("" ::: "memory");                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb0[ks], acc[i][0], 0, 0, 0);                         \
      acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb1[ks], acc[i][1], 0, 0, 0);                         \
    }                                                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                                         \
    asm volatile("" ::: "memory");                                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                          \
    asm volatile("" ::: "memory");                                                                                         \
  } while (0)
  int t = 0;
  for (; t + 2 < nk; t += 3) {
    LT_S_TILE(t, 0, 2);
    LT_S_TILE(t + 1, 1, 0);
    LT_S_TILE(t + 2, 2, 1);
  }
  if (t < nk) LT_S_TILE(t, 0, 2);
  if (t + 1 < nk) LT_S_TILE(t + 1, 1, 0);
#undef LT_S_TILE
#undef LT_S_DMA
  if (wm == 0) __builtin_amdgcn_s_barrier();  // re-align the two wave groups
  __syncthreads();
  float* wl = reinterpret_cast<float*>(smem + wave * 16384);
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        wl[(ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5)) * 64 + j * 32 + (l & 31)] = acc[ii][j][e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  emit_subtile<EPI>(g, wl, m0 + wm * 64, n0 + wn * 64, l, false);
}
constexpr int LDS_BYTES_S = 9 * 128 * 128;   // three stages x (A0 + B0 + B1) x 16 KiB
template <bool TB, int EPI>
int launch_s_one(const GemmArgs& g, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128e_kernel<TB, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_S);
    if (e != hipSuccess) { lt_set_error("lt_gemm_bf16: cannot enable 144 KiB LDS: %s", hipGetErrorString(e)); return LT_ERR_HIP; }
    configured = true;
  }
  hipLaunchKernelGGL((gemm128e_kernel<TB, EPI>), dim3(g.tiles_m * g.tiles_n), dim3(NT2), LDS_BYTES_S, st, g);
  return LT_OK;
}
template <bool TB>
int launch_s(const GemmArgs& g, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_s_one<TB, EPI_BF16>(g, st);
    case EPI_BF16_GELU: return launch_s_one<TB, EPI_BF16_GELU>(g, st);
    case EPI_RESID: return launch_s_one<TB, EPI_RESID>(g, st);
    case EPI_F32: return launch_s_one<TB, EPI_F32>(g, st);
    case EPI_BF16_GELUGRAD: return launch_s_one<TB, EPI_BF16_GELUGRAD>(g, st);
    default: lt_set_error("lt_gemm_bf16: no 128-row tail kernel for epilogue %d", epi); return LT_ERR_INVALID;
  }
}

template <bool TA, bool TB, int EPI, bool SLAB, bool CS, int PH, bool KT = false>
int launch_e_ph(const GemmArgs& g, dim3 grid, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256e_kernel<TA, TB, EPI, SLAB, CS, PH, KT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { lt_set_error("lt_gemm_bf16: cannot enable 128 KiB LDS: %s", hipGetErrorString(e)); return LT_ERR_HIP; }
    configured = true;
  }
  hipLaunchKernelGGL((gemm256e_kernel<TA, TB, EPI, SLAB, CS, PH, KT>), grid, dim3(NT2), LDS_BYTES, st, g);
  return LT_OK;
}
// partial last K-tile (forward / dgrad layouts; two phases per K-tile only)
template <bool TB>
int launch_e_ktail(const GemmArgs& g, int epi, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_e_ph<false, TB, EPI_BF16, false, false, 2, true>(g, grid, st);
    case EPI_BF16_GELU: return launch_e_ph<false, TB, EPI_BF16_GELU, false, false, 2, true>(g, grid, st);
    case EPI_RESID: return launch_e_ph<false, TB, EPI_RESID, false, false, 2, true>(g, grid, st);
    case EPI_F32: return launch_e_ph<false, TB, EPI_F32, false, false, 2, true>(g, grid, st);
    case EPI_BF16_GELUGRAD: return launch_e_ph<false, TB, EPI_BF16_GELUGRAD, false, false, 2, true>(g, grid, st);
    default: lt_set_error("lt_gemm_bf16: no partial-K-tile kernel for epilogue %d", epi); return LT_ERR_INVALID;
  }
}
template <bool TA, bool TB, int EPI, bool SLAB, bool CS = false>
int launch_e_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
  // phases per K-tile: 2 (default; 16 MFMAs per barrier interval: +1.5 % on the loop-dominated shape, step 81.7 -> 80.9 ms same-process,
  // profiles/r06b_*) or the q kernel's 4 (LT_GEMM_E_PH=4, read per call: tools/ab_step.py)
  const char* e = getenv("LT_GEMM_E_PH");
  if (e && atoi(e) == 4) return launch_e_ph<TA, TB, EPI, SLAB, CS, 4>(g, grid, st);
  return launch_e_ph<TA, TB, EPI, SLAB, CS, 2>(g, grid, st);
}
template <bool TA, bool TB>
int launch_e(const GemmArgs& g, int epi, bool slab, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_e_one<TA, TB, EPI_BF16, false>(g, grid, st);
    case EPI_BF16_GELU: return launch_e_one<TA, TB, EPI_BF16_GELU, false>(g, grid, st);
    case EPI_RESID: return launch_e_one<TA, TB, EPI_RESID, false>(g, grid, st);
    case EPI_F32: return launch_e_one<TA, TB, EPI_F32, false>(g, grid, st);
    case EPI_BF16_GELUGRAD: return launch_e_one<TA, TB, EPI_BF16_GELUGRAD, false>(g, grid, st);
    case EPI_F32_ACCUM:
      if (TA && slab && g.cs) return launch_e_one<TA, TB, EPI_F32_ACCUM, true, TA>(g, grid, st);
      return slab ? launch_e_one<TA, TB, EPI_F32_ACCUM, true>(g, grid, st) : launch_e_one<TA, TB, EPI_F32_ACCUM, false>(g, grid, st);
    default: lt_set_error("lt_gemm_bf16: unknown epilogue %d", epi); return LT_ERR_INVALID;
  }
}

#ifdef LT_GEMM_TIMING
extern "C" int lt_debug_gemm_timing(void* host_dst, int64_t n_u64, int clear) {
  hipDeviceSynchronize();
  if (host_dst && hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lt_gemm_timing_buf), (size_t)n_u64 * 8) != hipSuccess) return LT_ERR_HIP;
  if (clear) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(lt_gemm_timing_buf)) != hipSuccess) return LT_ERR_HIP;
    hipMemset(p, 0, sizeof(unsigned long long) * 8 * 16384);
  }
  return LT_OK;
}
#endif

template <bool TA, bool TB, int EPI, bool SLAB, bool KT = false, bool CS = false>
int launch_q_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256q_kernel<TA, TB, EPI, SLAB, KT, CS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { lt_set_error("lt_gemm_bf16: cannot enable 128 KiB LDS: %s", hipGetErrorString(e)); return LT_ERR_HIP; }
    configured = true;
  }
  hipLaunchKernelGGL((gemm256q_kernel<TA, TB, EPI, SLAB, KT, CS>), grid, dim3(NT2), LDS_BYTES, st, g);
  return LT_OK;
}
// partial last K-tile (forward / dgrad layouts)
template <bool TB>
int launch_q_ktail(const GemmArgs& g, int epi, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_q_one<false, TB, EPI_BF16, false, true>(g, grid, st);
    case EPI_BF16_GELU: return launch_q_one<false, TB, EPI_BF16_GELU, false, true>(g, grid, st);
    case EPI_RESID: return launch_q_one<false, TB, EPI_RESID, false, true>(g, grid, st);
    case EPI_F32: return launch_q_one<false, TB, EPI_F32, false, true>(g, grid, st);
    case EPI_BF16_GELUGRAD: return launch_q_one<false, TB, EPI_BF16_GELUGRAD, false, true>(g, grid, st);
    default: lt_set_error("lt_gemm_bf16: no partial-K-tile kernel for epilogue %d", epi); return LT_ERR_INVALID;
  }
}
template <bool TA, bool TB>
int launch_q(const GemmArgs& g, int epi, bool slab, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_q_one<TA, TB, EPI_BF16, false>(g, grid, st);
    case EPI_BF16_GELU: return launch_q_one<TA, TB, EPI_BF16_GELU, false>(g, grid, st);
    case EPI_RESID: return launch_q_one<TA, TB, EPI_RESID, false>(g, grid, st);
    case EPI_F32: return launch_q_one<TA, TB, EPI_F32, false>(g, grid, st);
    case EPI_BF16_GELUGRAD: return launch_q_one<TA, TB, EPI_BF16_GELUGRAD, false>(g, grid, st);
    case EPI_F32_ACCUM:
      if (TA && slab && g.cs) return launch_q_one<TA, TB, EPI_F32_ACCUM, true, false, TA>(g, grid, st);   // + column sums of A (bias gradient)
      return slab ? launch_q_one<TA, TB, EPI_F32_ACCUM, true>(g, grid, st) : launch_q_one<TA, TB, EPI_F32_ACCUM, false>(g, grid, st);
    default: lt_set_error("lt_gemm_bf16: unknown epilogue %d", epi); return LT_ERR_INVALID;
  }
}

// out[i] (+)= alpha * sum_s slab[s][i]   (deterministic split-K reduction)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n, int S, float alpha) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 256 * 4;
  for (; i < n; i += stride) {
    float4 o = *reinterpret_cast<float4*>(out + i);
    float4 a = *reinterpret_cast<const float4*>(slabs + i);
    int s2 = 1;
    for (; s2 + 3 < S; s2 += 4) {   // four slabs in flight per thread (fixed summation order: deterministic)
      const float4 b0 = *reinterpret_cast<const float4*>(slabs + (long)s2 * n + i);
      const float4 b1 = *reinterpret_cast<const float4*>(slabs + (long)(s2 + 1) * n + i);
      const float4 b2 = *reinterpret_cast<const float4*>(slabs + (long)(s2 + 2) * n + i);
      const float4 b3 = *reinterpret_cast<const float4*>(slabs + (long)(s2 + 3) * n + i);
      a.x += (b0.x + b1.x) + (b2.x + b3.x); a.y += (b0.y + b1.y) + (b2.y + b3.y);
      a.z += (b0.z + b1.z) + (b2.z + b3.z); a.w += (b0.w + b1.w) + (b2.w + b3.w);
    }
    for (; s2 < S; ++s2) {
      const float4 b = *reinterpret_cast<const float4*>(slabs + (long)s2 * n + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    o.x += alpha * a.x; o.y += alpha * a.y; o.z += alpha * a.z; o.w += alpha * a.w;
    *reinterpret_cast<float4*>(out + i) = o;
  }
}

template <bool TA, bool TB, int EPI, int BN, bool SLAB>
int launch_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<TA, TB, EPI, BN, SLAB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { lt_set_error("lt_gemm_bf16: cannot enable 128 KiB LDS: %s", hipGetErrorString(e)); return LT_ERR_HIP; }
    configured = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<TA, TB, EPI, BN, SLAB>), grid, dim3(NT2), LDS_BYTES, st, g);
  return LT_OK;
}

template <bool TA, bool TB, int BN>
int launch(const GemmArgs& g, int epi, bool slab, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_one<TA, TB, EPI_BF16, BN, false>(g, grid, st);
    case EPI_BF16_GELU: return launch_one<TA, TB, EPI_BF16_GELU, BN, false>(g, grid, st);
    case EPI_RESID: return launch_one<TA, TB, EPI_RESID, BN, false>(g, grid, st);
    case EPI_F32: return launch_one<TA, TB, EPI_F32, BN, false>(g, grid, st);
    case EPI_BF16_GELUGRAD: return launch_one<TA, TB, EPI_BF16_GELUGRAD, BN, false>(g, grid, st);
    case EPI_F32_ACCUM:
      return slab ? launch_one<TA, TB, EPI_F32_ACCUM, BN, true>(g, grid, st) : launch_one<TA, TB, EPI_F32_ACCUM, BN, false>(g, grid, st);
    default: lt_set_error("lt_gemm_bf16: unknown epilogue %d", epi); return LT_ERR_INVALID;
  }
}

// 128 x 128 x 64 tile, 4 waves, LDS-DMA staged, 64 KiB LDS -> TWO workgroups per CU: one workgroup's epilogue (GELU / residual
// traffic) overlaps the other's MFMA main loop.  Used where the epilogue is heavy relative to a short K loop.
}  // namespace g256

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
int launch_epi(const GemmArgs& g, int epi, bool vec, dim3 grid, hipStream_t st) {
  const size_t smem = 4 * TILE_BYTES;
#define LT_CASE(E)                                                                                \
  case E:                                                                                          \
    if (vec) hipLaunchKernelGGL((gemm_kernel<TA, TB, E, true>), grid, dim3(NTHREADS), smem, st, g);  \
    else hipLaunchKernelGGL((gemm_kernel<TA, TB, E, false>), grid, dim3(NTHREADS), smem, st, g);     \
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

extern "C" int lt_layernorm_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32, float* mean, float* rstd, int rows, int D,
                                float eps, void* stream);
static int gemm_bf16_impl(const lt_gemm_desc* d, void* stream);
extern "C" int lt_gemm_bf16(const lt_gemm_desc* d, void* stream) {
  LT_CHECK_ARG(d != nullptr, "lt_gemm_bf16: null descriptor");
  const bool want_ln = d->ln_out != nullptr;
  LT_CHECK_ARG(!want_ln || (d->epilogue == LT_EPI_RESID && d->ln_weight && d->ln_bias && d->batch <= 1 && d->ldc == d->N),
               "lt_gemm_bf16: the fused LayerNorm needs the residual epilogue, its weight and bias, and densely stored output rows");
  const int rc = gemm_bf16_impl(d, stream);
  if (rc != LT_OK || !want_ln || d->M == 0) return rc;
  return lt_layernorm_fwd((const float*)d->C, d->ln_weight, d->ln_bias, d->ln_out, nullptr, d->ln_mean, d->ln_rstd, d->M, d->N, d->ln_eps, stream);
}
static int gemm_bf16_impl(const lt_gemm_desc* d, void* stream) {
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
  g.rowscale = d->rowscale; g.branch_scale = d->branch_scale == 0.f ? 1.f : d->branch_scale;
  g.alpha = d->alpha;
  g.sa = d->stride_a; g.sb = d->stride_b; g.sc = d->stride_c;
  g.band = 0;
  g.cs = nullptr;
  // column sums of the transposed A operand (bias gradient beside a weight gradient): fused into the four-phase slab kernel below when
  // it is the kernel that runs and the reduction ledger is open; every other path starts with the stand-alone column-sum launch
  LT_CHECK_ARG(!d->colsum || (d->trans_a && d->lda == d->M && d->epilogue == LT_EPI_F32_ACCUM && d->batch <= 1),
               "lt_gemm_bf16: colsum needs a weight-gradient GEMM (trans_a, lda == M, accumulating epilogue)");
  bool cs_pending = d->colsum != nullptr;
  auto cs_standalone = [&]() -> int {
    if (!cs_pending) return LT_OK;
    cs_pending = false;
    return lt_colsum_bf16(d->A, d->colsum, d->K, d->M, stream);
  };
#ifdef LT_GEMM_TIMING
  if (const char* e = getenv("LT_GEMM_STAGGER")) { int u = 0, gr = 2; sscanf(e, "%d,%d", &u, &gr); g.sc = (d->batch > 1) ? g.sc : (long)((gr << 8) | u); }
#endif
  const int batch = d->batch > 1 ? d->batch : 1;
  LT_CHECK_ARG(batch == 1 || (!d->C2 && !d->resid && !d->aux && !d->rowscale && d->split_k <= 1 && d->force_kernel <= 1),
               "lt_gemm_bf16: batched launches support the plain epilogues of the 128x128 kernel only");
  g.tiles_m = lt_cdiv(d->M, BM); g.tiles_n = lt_cdiv(d->N, BN);
  int split = d->split_k > 0 ? d->split_k : 1;
  if (d->epilogue != LT_EPI_F32_ACCUM) split = 1;
  const int ktiles = lt_cdiv(d->K, BK);
  if (split > ktiles) split = ktiles;
  g.k_per_split = lt_cdiv(ktiles, split) * BK;
  split = lt_cdiv(d->K, g.k_per_split);
  dim3 grid(g.tiles_m * g.tiles_n, split, batch);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  // vector epilogue needs 4-element alignment of every row it touches
  auto al = [](const void* p, int ld, int bytes) { return p == nullptr || (ld % 4 == 0 && ((uintptr_t)p % (4 * bytes)) == 0); };
  const bool f32out = d->epilogue == LT_EPI_RESID || d->epilogue == LT_EPI_F32 || d->epilogue == LT_EPI_F32_ACCUM;
  const bool vec = d->N % 4 == 0 && al(d->C, d->ldc, f32out ? 4 : 2) && al(d->C2, d->ldc2, 2) && al(d->resid, d->ldr, 4) &&
                   al(d->aux, d->ldaux, 2) && al(d->bias, 4, 4) && al(d->gamma, 4, 4);
  // ---- 256-row LDS-DMA kernel for the large GEMMs (forward / dgrad over tokens; wgrad with slab split-K)
  const bool same_t = d->trans_a == d->trans_b || !d->trans_a;  // (N,N), (N,T), (T,T)
  // K % 64 != 0: the four-phase kernel's partial-last-tile variant, forward / dgrad layouts and non-accumulating epilogues only
  const bool ktail = d->K % BK != 0 && d->K % 8 == 0 && d->K > BK && !d->trans_a && d->epilogue != LT_EPI_F32_ACCUM;
  const bool eligible = vec && same_t && (d->K % BK == 0 || ktail) && d->N % 8 == 0 && (!d->trans_a || d->M % 8 == 0);
  // weight gradients with few output rows (the 64- and 128-channel convolutions of a ResNet): a 256-row tile is mostly padding, but
  // the slab split-K kernel at a quarter of its rate still beats the 128-row kernel's atomics by 2x (LT_GEMM_WGRAD_MIN_M, per call)
  const char* env_wm = getenv("LT_GEMM_WGRAD_MIN_M");
  const int wgrad_min_m = env_wm ? atoi(env_wm) : 64;
  const char* env_wk = getenv("LT_GEMM_WGRAD_MIN_K");   // shortest contraction the slab kernel takes (layer4 of a ResNet-50 at 224^2: 6272 rows)
  const int wgrad_min_k = env_wk ? atoi(env_wk) : 4096;   // 8192 -> 4096: ResNet-50 distillation step 43.4 -> 42.5 ms
  bool big = eligible && d->force_kernel != 1 && batch == 1 && d->N >= (ktail ? 256 : 128) &&
             ((!d->trans_a && d->M >= 2048) || (d->trans_a && d->K >= wgrad_min_k && d->M >= wgrad_min_m));
  if (d->force_kernel == 2 || d->force_kernel == 8 || d->force_kernel == 11) {
    LT_CHECK_ARG(eligible && (!ktail || d->force_kernel == 8 || d->force_kernel == 11), "lt_gemm_bf16: shape/layout not eligible for the 256-row LDS-DMA kernel");
    big = true;
  }
  if (big && ktail) {
    LT_CHECK_ARG(d->N >= 256 || d->force_kernel == 8, "lt_gemm_bf16: partial K-tile needs the 256-wide kernel");
    g.tiles_m = lt_cdiv(d->M, 256); g.tiles_n = lt_cdiv(d->N, 256);
    g.k_per_split = lt_cdiv(d->K, BK) * BK;
    dim3 gridk(g.tiles_m * g.tiles_n, 1);
    const char* env_ek = getenv("LT_GEMM_E");   // the static-address kernel's partial-K-tile variant (round 6); LT_GEMM_E=0 / force_kernel = 8: the q kernel's
    const bool ek_fits = ((size_t)256 * d->lda + d->K) * 2 < 0x7fffffffull &&
                         (d->trans_b ? (size_t)g.k_per_split * d->ldb : (size_t)256 * d->ldb + d->K) * 2 < 0x7fffffffull;
    if (ek_fits && d->N >= 256 && d->force_kernel != 8 && (!env_ek || atoi(env_ek) != 0))
      rc = d->trans_b ? g256::launch_e_ktail<true>(g, d->epilogue, gridk, st) : g256::launch_e_ktail<false>(g, d->epilogue, gridk, st);
    else
    rc = d->trans_b ? g256::launch_q_ktail<true>(g, d->epilogue, gridk, st) : g256::launch_q_ktail<false>(g, d->epilogue, gridk, st);
    if (rc != LT_OK) return rc;
    LT_CHECK_LAUNCH("lt_gemm_bf16");
  }
  if (big) {
    const int cus = 256;
    // tile width: 256 unless the 256-wide grid quantises badly on 256 CUs and the 128-wide one does not
    auto eff = [&](int bn, int sp) { const long b = (long)lt_cdiv(d->M, 256) * lt_cdiv(d->N, bn) * sp; return (double)b / ((double)((b + cus - 1) / cus) * cus); };
    int sp = 1;
    const bool accum = d->epilogue == LT_EPI_F32_ACCUM;
    // measured: the 4x2-wave 128-wide variant only pays when N < 256 (again in round 3 for N = 384, where a third of the 256-wide tiling is
    // padding: ViT-S step 42.7 ms with it, 44.7 ms with 128-wide tiles for N <= 640)
    int bn = d->N >= 256 ? 256 : 128;
    if (accum && d->split_k != 1) {
      // few output tiles, long contraction: choose the slice count that fills whole waves of 256 workgroups.
      // Slices are capped so the fp32 slabs fit the caller's workspace (without one: <= 4 slices of atomics).
      const int kt = d->K / BK;
      const size_t mn = (size_t)d->M * d->N;
      const int cap = d->workspace ? (int)std::min<size_t>(64, d->workspace_bytes / (mn * sizeof(float))) : 4;
      // LT_GEMM_WGRAD_SLICES (read per call; tools/ab_step.py): cap on the slice count.  Fewer, longer slices leave CUs to the other
      // streams' kernels and halve the fp32 slab traffic (S slabs written + read per weight gradient) at the cost of a longer launch
      const char* env_ms = getenv("LT_GEMM_WGRAD_SLICES");
      const int max_sl = env_ms ? std::max(1, atoi(env_ms)) : 64;
      double best = 0.0;
      for (int cand_bn : {256, 128}) {
        if (cand_bn == 256 && d->N < 256) continue;
        for (int c = 1; c <= cap && c <= max_sl && kt / c >= 8; ++c) {
          const double e2 = eff(cand_bn, c) * (cand_bn == 256 ? 1.0 : 0.9);
          if (e2 > best + 0.04) { best = e2; sp = c; bn = cand_bn; }
        }
      }
    }
    const bool slab = accum && sp > 1 && d->workspace && d->workspace_bytes >= (size_t)sp * d->M * d->N * sizeof(float);
    g.tiles_m = lt_cdiv(d->M, 256); g.tiles_n = lt_cdiv(d->N, bn);
    const int ktiles2 = d->K / BK;
    g.k_per_split = lt_cdiv(ktiles2, sp) * BK;
    sp = lt_cdiv(d->K, g.k_per_split);
    if (slab) g.C2 = d->workspace;
    {   // LT_GEMM_BAND (read per call): band width of the four-phase kernel's tile order.  Default: 4 where 4 divides a column-tile count
      // >= 8 (N = 3072: fc1 + GELU reads 24 % less, the GELU' dgrad 13 % less through the L2's memory side; no change in time), none
      // elsewhere (N = 2304 in bands of 3 read 7 % MORE: profiles/r03c_band_traffic.log)
      const char* env_b = getenv("LT_GEMM_BAND");
      const int nb = env_b ? atoi(env_b) : 4;
      g.band = (!d->trans_a && g.tiles_n >= 8 && nb > 0 && (env_b || g.tiles_n % nb == 0)) ? nb : 0;
    }
    dim3 grid2(g.tiles_m * g.tiles_n, sp);
    static const int use_q = [] { const char* e = getenv("LT_GEMM_Q"); return e ? atoi(e) : 1; }();  // LT_GEMM_Q=0: fall back to the 2-stage K-loop
    const bool q_kernel = bn == 256 && d->force_kernel != 2 && (d->force_kernel == 8 || d->force_kernel == 11 || use_q);
    if (cs_pending && q_kernel && slab && d->trans_a && d->trans_b) {
      // LT_GEMM_CS=0 (read per call: tools/ab_step.py): keep the separate column-sum pass
      const char* env_cs = getenv("LT_GEMM_CS");
      const int nparts = sp * g.tiles_n * 4;
      float* part = (env_cs && atoi(env_cs) == 0) ? nullptr : lt_ledger::reserve((size_t)nparts * d->M);
      if (part) {
        g.cs = part;
        lt_ledger::record(d->colsum, part, nparts, d->M, d->M);
        cs_pending = false;
      }
    }
    rc = cs_standalone();
    if (rc != LT_OK) return rc;
    // static-address K-loop (round 6): the same tiles and phases with no vector ALU work in the load segments.  LT_GEMM_E=0 (read per call:
    // tools/ab_step.py) / force_kernel = 8 keep the q kernel; force_kernel = 11 insists.  Byte offsets from the workgroup's operand base must
    // fit 31 bits: a 256-row panel of a K-contiguous operand, a k-slice of a transposed one.
    const char* env_e = getenv("LT_GEMM_E");
    const bool e_fits = (d->trans_a ? (size_t)g.k_per_split * d->lda : (size_t)256 * d->lda + d->K) * 2 < 0x7fffffffull &&
                        (d->trans_b ? (size_t)g.k_per_split * d->ldb : (size_t)256 * d->ldb + d->K) * 2 < 0x7fffffffull;
    const bool e_kernel = q_kernel && d->K % BK == 0 && e_fits && d->force_kernel != 8 && (d->force_kernel == 11 || !env_e || atoi(env_e) != 0);
    // the partly filled last round as 128-row tiles (gemm128e_kernel; OPT-IN: LT_GEMM_TAIL128=1, read per call): when the 256 x 256 tiles are
    // not a multiple of the CU count and the tile rows left after the full rounds fit ONE round of 128 x 256 tiles, the full rounds go to
    // the e kernel and the remaining rows to the tail kernel -- N = 768 at 50 432 rows: 510 + 162 workgroups instead of 591 (2.31 rounds
    // -> 2 + a short one).  Alone on the chip every N = 768 shape gains 5-15 % (profiles/r06g_stagger_probe.md, last column); inside the
    // five-stream step, where other streams' kernels fill the last round anyway, the second launch and the narrower tiles COST 1.0 ms
    // (81.54 vs 82.55 ms, 30 alternating steps each, profiles/r06i_ab_tail128.log): off by default
    bool tail128 = false;
    int big_rows = g.tiles_m;
    if (e_kernel && !d->trans_a && sp == 1 && !slab && d->epilogue != LT_EPI_F32_ACCUM) {
      const char* env_t = getenv("LT_GEMM_TAIL128");
      const long tiles = (long)g.tiles_m * g.tiles_n, full = tiles / cus;
      if (env_t && atoi(env_t) != 0 && full >= 1 && tiles % cus != 0) {
        const int br = (int)(full * cus / g.tiles_n);
        const int rem = d->M - br * 256;
        if (br >= 1 && rem > 0 && (long)lt_cdiv(rem, 128) * g.tiles_n <= cus) { tail128 = true; big_rows = br; }
      }
    }
    if (tail128) {
      GemmArgs gb = g;
      gb.tiles_m = big_rows;
      if (!(gb.band > 0 && gb.tiles_n >= 8)) gb.band = 0;
      dim3 gridb(gb.tiles_m * gb.tiles_n, 1);
      rc = d->trans_b ? g256::launch_e<false, true>(gb, d->epilogue, false, gridb, st) : g256::launch_e<false, false>(gb, d->epilogue, false, gridb, st);
      if (rc != LT_OK) return rc;
      GemmArgs gs = g;
      const size_t r0 = (size_t)big_rows * 256;
      gs.A = g.A + r0 * g.lda;
      gs.C = f32out ? (void*)((float*)g.C + r0 * g.ldc) : (void*)((bf16_t*)g.C + r0 * g.ldc);
      if (g.C2) gs.C2 = (bf16_t*)g.C2 + r0 * g.ldc2;
      if (g.resid) gs.resid = g.resid + r0 * g.ldr;
      if (g.aux) gs.aux = g.aux + r0 * g.ldaux;
      if (g.rowscale) gs.rowscale = g.rowscale + r0;
      gs.M = d->M - (int)r0;
      gs.tiles_m = lt_cdiv(gs.M, 128);
      gs.band = 0;
      rc = d->trans_b ? g256::launch_s<true>(gs, d->epilogue, st) : g256::launch_s<false>(gs, d->epilogue, st);
    } else
    if (e_kernel) {
      if (!d->trans_a && !d->trans_b) rc = g256::launch_e<false, false>(g, d->epilogue, slab, grid2, st);
      else if (!d->trans_a) rc = g256::launch_e<false, true>(g, d->epilogue, slab, grid2, st);
      else rc = g256::launch_e<true, true>(g, d->epilogue, slab, grid2, st);
    } else
    if (q_kernel) {
      if (!d->trans_a && !d->trans_b) rc = g256::launch_q<false, false>(g, d->epilogue, slab, grid2, st);
      else if (!d->trans_a) rc = g256::launch_q<false, true>(g, d->epilogue, slab, grid2, st);
      else rc = g256::launch_q<true, true>(g, d->epilogue, slab, grid2, st);
    } else
    if (!d->trans_a && !d->trans_b) rc = bn == 256 ? g256::launch<false, false, 256>(g, d->epilogue, slab, grid2, st) : g256::launch<false, false, 128>(g, d->epilogue, slab, grid2, st);
    else if (!d->trans_a) rc = bn == 256 ? g256::launch<false, true, 256>(g, d->epilogue, slab, grid2, st) : g256::launch<false, true, 128>(g, d->epilogue, slab, grid2, st);
    else rc = bn == 256 ? g256::launch<true, true, 256>(g, d->epilogue, slab, grid2, st) : g256::launch<true, true, 128>(g, d->epilogue, slab, grid2, st);
    if (rc != LT_OK) return rc;
    if (slab) {
      const long n = (long)d->M * d->N;
      hipLaunchKernelGGL(g256::slab_reduce_kernel, dim3((unsigned)min((long)2048, (n / 4 + 255) / 256)), dim3(256), 0, st,
                         (const float*)d->workspace, (float*)d->C, n, sp, d->alpha);
    }
    LT_CHECK_LAUNCH("lt_gemm_bf16");
  }
  rc = cs_standalone();
  if (rc != LT_OK) return rc;
  if (!d->trans_a && !d->trans_b) rc = launch_epi<false, false>(g, d->epilogue, vec, grid, st);
  else if (!d->trans_a && d->trans_b) rc = launch_epi<false, true>(g, d->epilogue, vec, grid, st);
  else if (d->trans_a && d->trans_b) rc = launch_epi<true, true>(g, d->epilogue, vec, grid, st);
  else rc = launch_epi<true, false>(g, d->epilogue, vec, grid, st);
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
