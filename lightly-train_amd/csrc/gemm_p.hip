// Persistent bf16 MFMA GEMM for gfx950 whose epilogue runs UNDER the next output tile's K-loop ("1p" kernel, force_kernel 10).
//
// Why: in the four-phase 256 x 256 kernel (gemm.hip) a workgroup owns its CU (8 waves x 256 registers, 128 KiB LDS) and its life is
// [fill | K-loop | epilogue]; at K = 768 the epilogue + fill are a third to a half of it (profiles/r02L_gemm_timeline.txt) and the MFMA
// units idle meanwhile -- chip-wide, because equal tiles keep all CUs in lockstep.  Here ONE workgroup per CU walks a list of output
// tiles; when a tile's K-loop ends its accumulators are rounded to bf16 (exactly the rounding the reference's autocast applies to a
// Linear's output: alpha * acc + bias -> bf16) into a register stash, the MFMAs of the next tile start at once, and the stash is drained
// -- LayerScale / residual / GELU / GELU' arithmetic, 16-byte row stores, operand prefetches -- in the issue slots between those MFMAs.
// The operand DMA stream never stops at a tile boundary (K-tiles are numbered across tiles), so there is no pipeline fill either.
//
//   tile 192 x 256 x 64, 4 waves = one per SIMD (2 x 2, wave tile 96 x 128 = 3 x 4 v_mfma_f32_32x32x16_bf16 blocks: 192 accumulator
//   registers + 64 stash registers in the AGPR half, the rest of the stash / fragments / epilogue temporaries in the VGPR half --
//   possible because a lone wave on its SIMD may use all 512 registers).
//   LDS (152 KiB): 2 stages x (A 24 KiB + B 32 KiB) | per wave: 4 KiB bf16 transposition window (32 rows x 64 columns: one "chunk" = two
//   accumulator blocks), 4.5 KiB operand ring (2 passes x [8 rows x 64 fp32 residuals + row scales] or saved pre-activations), 1 KiB
//   LayerScale gamma (tile being drained | tile being computed), 0.5 KiB bias.
//   EVERY global read is an LDS-DMA (`buffer_load ... lds` through bounds-checked descriptors): operands, residual rows, saved
//   pre-activations, row scales, bias, gamma.  One kind of load = one in-order completion stream, so a counted `s_waitcnt vmcnt(N)`
//   retires a load exactly when N <= the number of loads issued after it.  (Register-destination loads do NOT retire in order with
//   LDS-DMAs: a first version that fetched the residual rows into registers read them before their second cache line had landed
//   whenever the DMA queue was busy.)  And every load a counted wait relies on must really go to memory: a piece whose 64 lanes are ALL
//   out of range is answered by the bounds check without a memory access and retires ahead of older loads (seen: the pass in front of an
//   M-tail's out-of-range rows read its residuals early).  So addresses are CLAMPED to valid memory instead -- whole pieces past M / N
//   are redirected to row 0 by a scalar select, drain rows by a per-lane min, absent operands borrow A's descriptor -- and what they
//   fetch is never used: stores past M are dropped by the bounds check (no per-row branch in the drain code), column tiles past N are
//   skipped per 64-column chunk (N % 64 == 0 required).
//   K-loop: the one-barrier-per-K-tile schedule of gemm1w_kernel (gemm.hip), fragment reads one 12-MFMA step ahead (inline asm), the
//   MFMAs themselves inline asm on hand-allocated accumulator registers, and BETWEEN them, one slice per MFMA: the 8 + 6 DMA pieces of
//   the next K-tiles (3 scalar instructions each) and the four slices of a drain pass (window read | arithmetic | stores | fetch of the
//   pass two ahead).
//   Drain schedule (static): chunk c of the previous tile is in the window when K-tile 2c starts; its four 8-row passes are emitted in
//   the steps ks0, ks1, ks3 of K-tile 2c and ks0 of K-tile 2c+1 (nothing in ks2, whose end holds the DMA wait: stores sit in vmcnt
//   like the DMAs do, so the youngest store is a step old when the wait comes), chunk c+1 is staged in ks1 / ks2 of K-tile 2c+1.
//   K < 768 leaves chunks over: they are flushed at the tile boundary.
// Serves: forward (A [M,K], B [N,K]) and dgrad (B [K,N]) layouts, K % 128 == 0, with EPI_BF16 / EPI_BF16_GELU / EPI_RESID (without the
// bf16 copy C2) / EPI_BF16_GELUGRAD.  Everything else stays on gemm.hip's kernels.
#include "gemm_args.h"

#include <cstdlib>

namespace lt_gemm {
namespace {

constexpr int PBM = 192, PBN = 256, PBK = 64;
constexpr int A_BYTES = PBM * 128, B_BYTES = PBN * 128;   // 24 KiB + 32 KiB per stage
constexpr int A_OFF = 0, B_OFF = 2 * A_BYTES, WAVE_OFF = 2 * (A_BYTES + B_BYTES), PW = 11776;
constexpr int WIN = 0, RING = 4096, RING_SLOT = 3072, GAMC = 10240, BIASC = 11264;   // offsets inside a wave's private region
constexpr int LDS_TOTAL = WAVE_OFF + 4 * PW;   // 161 792 B

typedef __attribute__((address_space(3))) void lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000u);
}
// one LDS-DMA piece: 64 lanes x BYTES from (descriptor base + soff + per-lane voff) to lds + lane * BYTES
// (the size operand of the builtin has to be a literal)
template <int BYTES>
__device__ __forceinline__ void dma(char* lds, rsrc_t r, int voff, int soff) {
  if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t*)lds, 16, voff, soff, 0, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t*)lds, 4, voff, soff, 0, 0);
}

// ---- operand images (same as gemm.hip's stage_dma: XOR-swizzled 128-byte rows / [4k][16] transposed pieces).  Per-lane byte offsets
// of the two piece parities are computed once per kernel; the rest of a piece's address is a wave-uniform running offset.
struct DmaLane { int a[2], b[2]; };
template <bool TB>
__device__ __forceinline__ DmaLane dma_lane(int lda, int ldb, int l) {
  DmaLane d;
  const int r = l >> 3, slot = l & 7;
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int c = slot ^ ((4 * par + (r >> 1)) & 7);
    d.a[par] = (r * lda + c * 8) * 2;
    d.b[par] = (r * ldb + c * 8) * 2;
  }
  if (TB) {   // [4k][16] pieces: k-row rotated by the piece's position, 8-column half by the slot parity
    const int kr = ((slot >> 1) - r) & 3;
    d.b[0] = (kr * ldb + r * 16 + (slot & 1) * 8) * 2;
    d.b[1] = d.b[0];
  }
  return d;
}

// fragment reads with the stage / block offsets as instruction immediates (one address register per k16 step and operand)
template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8& v, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void lds_read_tr(s16x4& v, unsigned a) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); }

// ---- the accumulator file is allocated BY HAND: a[0:191] = the twelve 32 x 32 accumulator blocks (block (i, j) at (4 i + j) * 16),
// a[192:255] = stash chunks 2..5.  hipcc keeps MFMA accumulators it manages as 16-register tuples that it copies wholesale between the
// AGPR and VGPR halves whenever elements are extracted, and with 240 of 256 AGPRs live it spilled stash registers to scratch inside the
// K-loop; explicit register names in asm take the whole file out of its hands.  The compiler-visible code must then never touch an
// AGPR: its VGPR demand stays far below 256 (no AGPR spilling), the kernel clobbers a255 once so that the descriptor allocates all 256,
// and tools/audit_gemm_p.py (no AGPR reference outside ASMSTART..ASMEND, no scratch) guards it in tests/test_host_logic.py.
template <int R> __device__ __forceinline__ float aread() { float x; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(R)); return x; }
template <int R> __device__ __forceinline__ void awrite(unsigned x) { asm volatile("v_accvgpr_write_b32 a[%c1], %0" :: "v"(x), "n"(R)); }
template <int R, bool Z> __device__ __forceinline__ void amfma(const bf16x8& a, const bf16x8& b) {
  if (Z) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" :: "v"(a), "v"(b), "n"(R), "n"(R + 15));
  else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "n"(R), "n"(R + 15));
}
constexpr int ST_AGPR0 = 192;
template <int R0>
__device__ __forceinline__ void aread8(unsigned (&s)[8]) {
  s[0] = __float_as_uint(aread<R0 + 0>()); s[1] = __float_as_uint(aread<R0 + 1>()); s[2] = __float_as_uint(aread<R0 + 2>());
  s[3] = __float_as_uint(aread<R0 + 3>()); s[4] = __float_as_uint(aread<R0 + 4>()); s[5] = __float_as_uint(aread<R0 + 5>());
  s[6] = __float_as_uint(aread<R0 + 6>()); s[7] = __float_as_uint(aread<R0 + 7>());
}

// ---- stash -> window.  A stash register holds the bf16 values of rows r, r+1 of one column (MFMA C layout: lane -> column, register ->
// row pair); the window is [32 rows][64 columns] bf16, so that a drain pass reads 8 consecutive columns of a row as one 16-byte word.
// Chunks 0, 1 live in (compiler-allocated) VGPRs, chunks 2..5 in a[192:255].  HALF: accumulator block 0 | 1 of the chunk.
template <int C, int HALF>
__device__ __forceinline__ void stage_half(unsigned short* win, const unsigned (&stv)[32], int l) {
  unsigned blk[8];
  if (C >= 2) aread8<ST_AGPR0 + ((C < 2 ? 2 : C) - 2) * 16 + HALF * 8>(blk);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned s;
    // (the empty asm: the stash is loop-invariant inside a tile's K-loop; LICM would otherwise hoist every unpacked value out of it)
    if (C < 2) { s = stv[((C < 2 ? C : 0) * 2 + HALF) * 8 + q]; asm volatile("" : "+v"(s)); }
    else s = blk[q];
    const int r = ((2 * q) & 3) + 8 * (q >> 1) + 4 * (l >> 5);
    unsigned short* d = win + r * 64 + HALF * 32 + (l & 31);
    d[0] = (unsigned short)s;
    d[64] = (unsigned short)(s >> 16);
  }
}
#define LT_STAGE_HALF(C, HALF)                                    \
  do {                                                            \
    switch (C) {                                                  \
      case 0: stage_half<0, HALF>(win, stv, l); break;            \
      case 1: stage_half<1, HALF>(win, stv, l); break;            \
      case 2: stage_half<2, HALF>(win, stv, l); break;            \
      case 3: stage_half<3, HALF>(win, stv, l); break;            \
      case 4: stage_half<4, HALF>(win, stv, l); break;            \
      case 5: stage_half<5, HALF>(win, stv, l); break;            \
      default: break;                                             \
    }                                                             \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
    __builtin_amdgcn_wave_barrier();                              \
  } while (0)

// tile boundary: accumulator block (I, J) -> bf16(alpha * acc + bias) pairs -> stash
template <int I, int J>
__device__ __forceinline__ void stash_block(unsigned (&stv)[32], float alpha, float bias) {
  constexpr int B0 = (I * 4 + J) * 16;
  float v[16];
  v[0] = aread<B0 + 0>(); v[1] = aread<B0 + 1>(); v[2] = aread<B0 + 2>(); v[3] = aread<B0 + 3>();
  v[4] = aread<B0 + 4>(); v[5] = aread<B0 + 5>(); v[6] = aread<B0 + 6>(); v[7] = aread<B0 + 7>();
  v[8] = aread<B0 + 8>(); v[9] = aread<B0 + 9>(); v[10] = aread<B0 + 10>(); v[11] = aread<B0 + 11>();
  v[12] = aread<B0 + 12>(); v[13] = aread<B0 + 13>(); v[14] = aread<B0 + 14>(); v[15] = aread<B0 + 15>();
  unsigned x[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) x[q] = pack_bf2(v[2 * q] * alpha + bias, v[2 * q + 1] * alpha + bias);
  if (I == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) stv[J * 8 + q] = x[q];
  } else {
    constexpr int S0 = ST_AGPR0 + ((I == 0 ? 1 : I) - 1) * 32 + J * 8;
    awrite<S0 + 0>(x[0]); awrite<S0 + 1>(x[1]); awrite<S0 + 2>(x[2]); awrite<S0 + 3>(x[3]);
    awrite<S0 + 4>(x[4]); awrite<S0 + 5>(x[5]); awrite<S0 + 6>(x[6]); awrite<S0 + 7>(x[7]);
  }
}

// ---- drain --------------------------------------------------------------------------------------------------------------------
struct EpiCtx {
  rsrc_t rC, rC2, rR, rAux, rRs;
  int ldc, ldc2, ldr, ldaux, N, M;
  float branch_scale;
  bool has_rs, has_resid, strict;
};
// coordinates of pass `p` (8 rows) of chunk `c` (32 rows x 64 columns) of the stashed tile, for this lane: 8 columns of one row
struct PassPos { int row, col; bool colok; };
__device__ __forceinline__ PassPos pass_pos(int prow0, int pcol0, int c, int p, int l, int N) {
  PassPos q;
  q.row = prow0 + (c >> 1) * 32 + p * 8 + (l >> 3);
  const int c0 = pcol0 + (c & 1) * 64;
  q.col = c0 + (l & 7) * 8;
  q.colok = c0 < N;
  return q;
}
__device__ __forceinline__ int voff(const PassPos& q, int ld, int esize) {   // STORE offset, or one the bounds check rejects (also + 16)
  return q.colok ? (q.row * ld + q.col) * esize : (int)0x80000000u;
}
__device__ __forceinline__ int loff(const PassPos& q, int ld, int esize, int M, int l) {   // LOAD offset: always inside the operand
  return (min(q.row, M - 1) * ld + (q.colok ? q.col : (l & 7) * 8)) * esize;
}
// fetch the operands of a pass into ring slot SLOT of the wave's LDS region: EPI_RESID 8 residual floats per lane (two 16-byte pieces)
// + the stochastic-depth row scale; EPI_BF16_GELUGRAD 8 saved pre-activations (bf16).  NL = pieces per pass.
template <int EPI, int SLOT>
__device__ __forceinline__ void ring_fetch(const EpiCtx& e, char* wreg, const PassPos& q, int l) {
  char* s = wreg + RING + SLOT * RING_SLOT;
  if (EPI == EPI_RESID) {
    const int o = loff(q, e.ldr, 4, e.M, l);
    dma<16>(s, e.rR, o, 0);
    dma<16>(s + 1024, e.rR, o, 16);
    dma<16>(s + 2048, e.rRs, (min(q.row, e.M - 1) & ~3) * 4, 0);   // the aligned 4 floats around this lane's row scale (16-byte pieces only: see above)
  } else if (EPI == EPI_BF16_GELUGRAD) {
    dma<16>(s, e.rAux, loff(q, e.ldaux, 2, e.M, l), 0);
  }
}
// one drain pass (8 rows x 64 columns out of the window -> epilogue arithmetic -> global) in four slices, placed by hand between the
// MFMAs of a K-loop step: A = window / gamma reads, B = arithmetic, C = stores, D = fetch of the operands two passes ahead.
template <int EPI>
struct Pass {
  PassPos q;
  u32x4 w;
  float4 g0, g1;
  u32x4 o0, o1;
};
// NW: vmcnt bound that retires this pass's ring slot (loads are all 16-byte LDS-DMAs, in order: NW <= the number of such DMAs issued
// after it).  The wait sits HERE, a window read and several MFMAs ahead of the ring reads in slice B: vmcnt reaching its bound does not
// mean the piece's bytes are visible to a ds_read issued in the very next cycles -- with the wait directly in front of the reads, a
// wait that actually had to block (the last piece landing just then) was followed by reads of the slot's OLD contents often enough to
// fail one launch in ten.  Slice B therefore starts with lgkmcnt(0): this slice's window read has made the round trip through the LDS
// queue behind the DMA's write.
template <int EPI, int NW>
__device__ __forceinline__ void pass_a(Pass<EPI>& s, const EpiCtx& e, const char* wreg, int gcur, int prow0, int pcol0, int c, int p, int l) {
  if (EPI == EPI_RESID || EPI == EPI_BF16_GELUGRAD) {
    if (e.strict) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(%c0)" :: "n"(NW) : "memory");
  }
  s.q = pass_pos(prow0, pcol0, c, p, l, e.N);
  s.w = *reinterpret_cast<const u32x4*>(wreg + WIN + (p * 8 + (l >> 3)) * 128 + (l & 7) * 16);
  if (EPI == EPI_RESID) {
    const float* gp = reinterpret_cast<const float*>(wreg + GAMC + gcur * 512) + (c & 1) * 64 + (l & 7) * 8;
    s.g0 = *reinterpret_cast<const float4*>(gp); s.g1 = *reinterpret_cast<const float4*>(gp + 4);
  }
}
template <int EPI, int SLOT>
__device__ __forceinline__ void pass_b(Pass<EPI>& s, const EpiCtx& e, const char* wreg, int l) {
  if (EPI == EPI_RESID || EPI == EPI_BF16_GELUGRAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float v[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(s.w[k] << 16); v[2 * k + 1] = __uint_as_float(s.w[k] & 0xffff0000u); }
  const char* rs = wreg + RING + SLOT * RING_SLOT;
  if (EPI == EPI_RESID) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(rs + l * 16), a1 = *reinterpret_cast<const f32x4*>(rs + 1024 + l * 16);
    if (!e.has_resid) { a0 = (f32x4){0.f, 0.f, 0.f, 0.f}; a1 = a0; }
    const float rsc = *reinterpret_cast<const float*>(rs + 2048 + l * 16 + (min(s.q.row, e.M - 1) & 3) * 4);
    const float sc = e.branch_scale * (e.has_rs ? rsc : 1.f);
    f32x4 o0, o1;
    o0[0] = a0[0] + sc * s.g0.x * v[0]; o0[1] = a0[1] + sc * s.g0.y * v[1]; o0[2] = a0[2] + sc * s.g0.z * v[2]; o0[3] = a0[3] + sc * s.g0.w * v[3];
    o1[0] = a1[0] + sc * s.g1.x * v[4]; o1[1] = a1[1] + sc * s.g1.y * v[5]; o1[2] = a1[2] + sc * s.g1.z * v[6]; o1[3] = a1[3] + sc * s.g1.w * v[7];
    s.o0 = __builtin_bit_cast(u32x4, o0); s.o1 = __builtin_bit_cast(u32x4, o1);
  } else {
    if (EPI == EPI_BF16_GELU) {
      s.o1 = s.w;   // the saved pre-activations are the stashed values as they are
#pragma unroll
      for (int k = 0; k < 8; k += 2) gelu2(v[k], v[k + 1]);
    } else if (EPI == EPI_BF16_GELUGRAD) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(rs + l * 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) mul_gelu_grad2(v[2 * k], v[2 * k + 1], bf2f((bf16_t)(a[k] & 0xffff)), bf2f((bf16_t)(a[k] >> 16)));
    }
    if (EPI == EPI_BF16) s.o0 = s.w;
    else s.o0 = (u32x4){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  }
}
template <int EPI>
__device__ __forceinline__ void pass_c(const Pass<EPI>& s, const EpiCtx& e) {
  if (EPI == EPI_RESID) {
    const int o = voff(s.q, e.ldc, 4);
    __builtin_amdgcn_raw_buffer_store_b128(s.o0, e.rC, o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(s.o1, e.rC, o, 16, 0);
  } else {
    if (EPI == EPI_BF16_GELU) __builtin_amdgcn_raw_buffer_store_b128(s.o1, e.rC2, voff(s.q, e.ldc2, 2), 0, 2);   // streaming
    __builtin_amdgcn_raw_buffer_store_b128(s.o0, e.rC, voff(s.q, e.ldc, 2), 0, 2);
  }
}
template <int EPI, int P>
__device__ __forceinline__ void pass_d(const EpiCtx& e, char* wreg, int prow0, int pcol0, int c, int l) {
  if (EPI == EPI_RESID || EPI == EPI_BF16_GELUGRAD)
    ring_fetch<EPI, P & 1>(e, wreg, pass_pos(prow0, pcol0, P < 2 ? c : c + 1, (P + 2) & 3, l, e.N), l);
}

template <bool TB, int EPI>
__global__ __launch_bounds__(256) void gemm1p_kernel(const GemmArgs g, const int wb, const int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int wsc = __builtin_amdgcn_readfirstlane(wave);
  const int nk = g.K / PBK;
  char* const wreg = smem + WAVE_OFF + wsc * PW;   // this wave's private LDS region
  unsigned short* const win = reinterpret_cast<unsigned short*>(wreg + WIN);

  // ---- this workgroup's tile list: XCD x (= blockIdx % 8, where the dispatcher places consecutive workgroups) owns the x-th eighth of
  // the band-ordered tile list; its workgroups take that range round-robin, so the ~32 tiles an XCD works on at a time are neighbours
  // in a band of `wb` column tiles: they share A panels (same tile row) and B panels (same tile column) in the XCD's L2
  const int NT = g.tiles_m * g.tiles_n;
  const int xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
  const int nwx = ((int)gridDim.x - xcd + 7) >> 3;
  const int lo = (int)((long)xcd * NT / 8), hi = (int)((long)(xcd + 1) * NT / 8);
  const int my_n = lo + jx < hi ? (hi - lo - jx + nwx - 1) / nwx : 0;
  const int full_b = g.tiles_n / wb, rem_b = g.tiles_n - full_b * wb, per_band = g.tiles_m * wb;
  auto tile_origin = [&](int i, int& m0, int& n0) {
    const int id = lo + jx + i * nwx;
    int tm, tn;
    if (id < full_b * per_band) { const int bnd = id / per_band, r = id - bnd * per_band; tm = r / wb; tn = bnd * wb + (r - tm * wb); }
    else { const int r = id - full_b * per_band; tm = r / rem_b; tn = full_b * wb + (r - tm * rem_b); }
    m0 = tm * PBM; n0 = tn * PBN;
  };
  if (my_n == 0) return;

  EpiCtx e;
  e.rC = make_rsrc(g.C, (long)g.M * g.ldc * (EPI == EPI_RESID ? 4 : 2));
  e.rC2 = make_rsrc(g.C2, (long)g.M * g.ldc2 * 2);
  // (absent drain operands borrow A's memory: their pieces must be real loads, their values are discarded)
  e.rR = g.resid ? make_rsrc(g.resid, (long)g.M * g.ldr * 4) : make_rsrc(g.A, (long)g.M * g.lda * 2);
  e.rAux = make_rsrc(g.aux, (long)g.M * g.ldaux * 2);
  e.rRs = g.rowscale ? make_rsrc(g.rowscale, (long)g.M * 4) : make_rsrc(g.A, (long)g.M * g.lda * 2);
  e.ldc = g.ldc; e.ldc2 = g.ldc2; e.ldr = g.resid ? g.ldr : 0; e.ldaux = g.ldaux; e.N = g.N; e.M = g.M;
  e.branch_scale = g.branch_scale; e.has_rs = g.rowscale != nullptr; e.has_resid = g.resid != nullptr; e.strict = (dbg & 2) != 0;
  const rsrc_t rBias = make_rsrc(EPI == EPI_BF16_GELUGRAD ? nullptr : g.bias, (long)g.N * 4);
  const rsrc_t rGam = make_rsrc(g.gamma, (long)g.N * 4);
  const rsrc_t rA = make_rsrc(g.A, (long)g.M * g.lda * 2);
  const rsrc_t rB = make_rsrc(g.B, (long)(TB ? g.K : g.N) * g.ldb * 2);

  asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");   // the descriptor must allocate the whole accumulator file (see above)
  unsigned stv[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) stv[q] = 0;
  Pass<EPI> ps;
  bool have_stash = false;
  int prow0 = 0, pcol0 = 0;   // origin of this wave's part of the stashed tile
  int gcur = 0;               // gamma cache half holding the stashed tile's LayerScale (the other half receives the running tile's)
  if (EPI == EPI_RESID && !g.gamma) {   // absent LayerScale = 1: both halves, once
    float* gc = reinterpret_cast<float*>(wreg + GAMC);
    gc[l] = 1.f; gc[64 + l] = 1.f; gc[128 + l] = 1.f; gc[192 + l] = 1.f;
  }

  // ---- fragment reads: inline asm (see gemm1w_kernel in gemm.hip for the discipline: results stay in the registers the instructions
  // name until LT_WAIT, no read under a branch).  Per-lane addresses: one per k16 step (A), one per step (B rows) or per step and block
  // parity (B transposed); stage and block offsets ride in the instruction's offset field.
  bf16x8 qa[2][3], qb[2][4];
  s16x4 rb[2][4][2];
  bf16x8 fb[4];
  unsigned aoff[4], boff[4], btr[4][2];
  {
    const int sw = ((l & 31) >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + (l >> 5);
      aoff[ks] = (unsigned)(uintptr_t)(lptr_t*)(smem + A_OFF + (wm * 96 + (l & 31)) * 128 + ((c ^ sw) << 4));
      boff[ks] = (unsigned)(uintptr_t)(lptr_t*)(smem + B_OFF + (wn * 128 + (l & 31)) * 128 + ((c ^ sw) << 4));
      const int i = l & 15, cb = (l >> 4) & 1, kh = l >> 5, q0 = ks * 4 + kh * 2;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int inner = ((((i >> 2) + 2 * jp + cb) & 3) << 5) + ((i & 3) << 3);
        btr[ks][jp] = (unsigned)(uintptr_t)(lptr_t*)(smem + B_OFF + (q0 * 16 + wn * 8 + cb) * 128 + inner);
      }
    }
  }
#define LT_READ(BUF, S, KS)                                                                                   \
  do {                                                                                                        \
    lds_read128<(S) * A_BYTES + 0 * 4096>(qa[BUF][0], aoff[KS]);                                              \
    lds_read128<(S) * A_BYTES + 1 * 4096>(qa[BUF][1], aoff[KS]);                                              \
    lds_read128<(S) * A_BYTES + 2 * 4096>(qa[BUF][2], aoff[KS]);                                              \
    if (TB) {                                                                                                 \
      lds_read_tr<(S) * B_BYTES + 0 * 256>(rb[BUF][0][0], btr[KS][0]); lds_read_tr<(S) * B_BYTES + 0 * 256 + 2048>(rb[BUF][0][1], btr[KS][0]); \
      lds_read_tr<(S) * B_BYTES + 1 * 256>(rb[BUF][1][0], btr[KS][1]); lds_read_tr<(S) * B_BYTES + 1 * 256 + 2048>(rb[BUF][1][1], btr[KS][1]); \
      lds_read_tr<(S) * B_BYTES + 2 * 256>(rb[BUF][2][0], btr[KS][0]); lds_read_tr<(S) * B_BYTES + 2 * 256 + 2048>(rb[BUF][2][1], btr[KS][0]); \
      lds_read_tr<(S) * B_BYTES + 3 * 256>(rb[BUF][3][0], btr[KS][1]); lds_read_tr<(S) * B_BYTES + 3 * 256 + 2048>(rb[BUF][3][1], btr[KS][1]); \
    } else {                                                                                                  \
      lds_read128<(S) * B_BYTES + 0 * 4096>(qb[BUF][0], boff[KS]);                                            \
      lds_read128<(S) * B_BYTES + 1 * 4096>(qb[BUF][1], boff[KS]);                                            \
      lds_read128<(S) * B_BYTES + 2 * 4096>(qb[BUF][2], boff[KS]);                                            \
      lds_read128<(S) * B_BYTES + 3 * 4096>(qb[BUF][3], boff[KS]);                                            \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)
#define LT_WAIT(BUF)                                                                                                                     \
  do {                                                                                                                                   \
    if (TB) {                                                                                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                \
                   : "+v"(qa[BUF][0]), "+v"(qa[BUF][1]), "+v"(qa[BUF][2]), "+v"(rb[BUF][0][0]), "+v"(rb[BUF][0][1]),                        \
                     "+v"(rb[BUF][1][0]), "+v"(rb[BUF][1][1]), "+v"(rb[BUF][2][0]), "+v"(rb[BUF][2][1]), "+v"(rb[BUF][3][0]), "+v"(rb[BUF][3][1])); \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                                 \
        union { struct { s16x4 a, b; } s; bf16x8 v; } ub_;                                                                               \
        ub_.s.a = rb[BUF][j_][0]; ub_.s.b = rb[BUF][j_][1]; fb[j_] = ub_.v;                                                              \
      }                                                                                                                                  \
    } else {                                                                                                                             \
      asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                \
                   : "+v"(qa[BUF][0]), "+v"(qa[BUF][1]), "+v"(qa[BUF][2]), "+v"(qb[BUF][0]), "+v"(qb[BUF][1]),                              \
                     "+v"(qb[BUF][2]), "+v"(qb[BUF][3]));                                                                                \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) fb[j_] = qb[BUF][j_];                                                             \
    }                                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                                   \
  } while (0)
#define LT_BARRIER()                  \
  do {                                \
    asm volatile("" ::: "memory");    \
    __builtin_amdgcn_s_barrier();     \
    asm volatile("" ::: "memory");    \
  } while (0)

  // ---- DMA cursors over the global K-tile sequence (tile list x K-tiles): B runs one K-tile ahead of the compute, A two.  K / 64 is
  // even, so every tile starts on stage 0 and the stage a K-tile computes on / refills is a compile-time constant of its code copy.
  // Each cursor keeps the byte offset of piece 0 of its K-tile (`*_so`); a piece adds a constant step.  Past the end of the tile list
  // the cursor wraps to tile (0, 0): the pieces are still issued (the counted waits rely on every K-tile putting the same number of
  // real loads into the queue) into a stage nobody reads.
  const DmaLane dl = dma_lane<TB>(g.lda, g.ldb, l);
  const int a_step = 16 * g.lda, b_step = TB ? 8 * g.ldb : 16 * g.ldb, b_kstep = TB ? 128 * g.ldb : 128;
  char* const a_lds = smem + A_OFF + wsc * 6 * 1024;
  char* const b_lds = smem + B_OFF + wsc * 8 * 1024;
  int a_i = 0, a_k = 0, a_m0, a_n0, b_i = 0, b_k = 0, b_m0, b_n0, a_so, b_so, a_run = 0, b_run = 0;
  tile_origin(0, a_m0, a_n0); b_m0 = a_m0; b_n0 = a_n0;
  // (row-major B: this wave's 64 rows of a column tile are all inside N or all outside -- N % 64 == 0 -- and outside ones are fetched
  // from rows 0..63 instead; A pieces past M are redirected piece by piece in LT_DMA_A)
  auto a_base = [&]() { a_so = ((a_m0 + wsc * 48) * g.lda + a_k * PBK) * 2; };
  auto b_base = [&]() {
    const int nn = b_n0 + wsc * 64 < g.N ? b_n0 + wsc * 64 : 0;
    b_so = TB ? ((b_k * PBK + 16 * wsc) * g.ldb + b_n0) * 2 : (nn * g.ldb + b_k * PBK) * 2;
  };
  a_base(); b_base();
  // piece J of the A / B image of the cursor's K-tile into stage S
#define LT_DMA_A(S, J)                                                                                         \
  do {                                                                                                         \
    if ((J) == 0) a_run = a_so; else a_run += a_step;                                                          \
    dma<16>(a_lds + (S) * A_BYTES + (J) * 1024, rA, dl.a[(J) & 1], a_m0 + wsc * 48 + (J) * 8 < g.M ? a_run : a_k * 128); \
  } while (0)
#define LT_DMA_B(S, J)                                                                                         \
  do {                                                                                                         \
    if ((J) == 0) b_run = b_so; else if (TB) b_run += ((J) & 1) ? 256 : b_step - 256; else b_run += b_step;    \
    dma<16>(b_lds + (S) * B_BYTES + (J) * 1024, rB, dl.b[(J) & 1], b_run);                                     \
  } while (0)
#define LT_ADV_A()                                                                                             \
  do {                                                                                                         \
    a_so += 128;                                                                                               \
    if (++a_k == nk) { a_k = 0; if (++a_i < my_n) tile_origin(a_i, a_m0, a_n0); else { a_m0 = 0; a_n0 = 0; } a_base(); } \
  } while (0)
#define LT_ADV_B()                                                                                             \
  do {                                                                                                         \
    b_so += b_kstep;                                                                                           \
    if (++b_k == nk) { b_k = 0; if (++b_i < my_n) tile_origin(b_i, b_m0, b_n0); else { b_m0 = 0; b_n0 = 0; } b_base(); } \
  } while (0)
#define LT_DMA_A_ALL(S) do { LT_DMA_A(S, 0); LT_DMA_A(S, 1); LT_DMA_A(S, 2); LT_DMA_A(S, 3); LT_DMA_A(S, 4); LT_DMA_A(S, 5); LT_ADV_A(); } while (0)
#define LT_DMA_B_ALL(S) do { LT_DMA_B(S, 0); LT_DMA_B(S, 1); LT_DMA_B(S, 2); LT_DMA_B(S, 3); LT_DMA_B(S, 4); LT_DMA_B(S, 5); LT_DMA_B(S, 6); LT_DMA_B(S, 7); LT_ADV_B(); } while (0)

  LT_DMA_A_ALL(0); LT_DMA_B_ALL(0); LT_DMA_A_ALL(1);
  __builtin_amdgcn_s_waitcnt(0xF76);   // vmcnt(6): K-tile 0 landed (LDS-DMAs complete in issue order)
  __builtin_amdgcn_s_waitcnt(0xC07F);
  LT_BARRIER();
  LT_READ(0, 0, 0);

  constexpr int NL = EPI == EPI_RESID ? 3 : (EPI == EPI_BF16_GELUGRAD ? 1 : 0);   // ring DMA pieces per pass
  // ---- a k16 step: the twelve MFMAs with one slice of other work behind each.  Z: first step of a tile (C operand = 0: no zeroing
  // pass).  The s_nop: fb may have been assembled by compiler VALU moves (transposed B), and nothing pads a VALU write -> asm MFMA read.
#define LT_MF(BUF, Z, I, J) amfma<((I) * 4 + (J)) * 16, Z>(qa[BUF][I], fb[J])
#define LT_STEP(BUF, Z, S0, S1, S2, S3, S4, S5, S6, S7, S8, S9, S10, S11)                                       \
  do {                                                                                                         \
    asm volatile("s_nop 1");                                                                                   \
    LT_MF(BUF, Z, 0, 0); S0; LT_MF(BUF, Z, 0, 1); S1; LT_MF(BUF, Z, 0, 2); S2; LT_MF(BUF, Z, 0, 3); S3;          \
    LT_MF(BUF, Z, 1, 0); S4; LT_MF(BUF, Z, 1, 1); S5; LT_MF(BUF, Z, 1, 2); S6; LT_MF(BUF, Z, 1, 3); S7;          \
    LT_MF(BUF, Z, 2, 0); S8; LT_MF(BUF, Z, 2, 1); S9; LT_MF(BUF, Z, 2, 2); S10; LT_MF(BUF, Z, 2, 3); S11;        \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#define LT_PA(EM, C, P, NW) do { if (EM) pass_a<EPI, NW>(ps, e, wreg, gcur, prow0, pcol0, C, P, l); } while (0)
#define LT_PB(EM, P) do { if (EM) pass_b<EPI, (P) & 1>(ps, e, wreg, l); } while (0)
#define LT_PC(EM) do { if (EM) pass_c<EPI>(ps, e); } while (0)
#define LT_PD(EM, C, P) do { if (EM) pass_d<EPI, P>(e, wreg, prow0, pcol0, C, l); } while (0)
#define LT_NOP ((void)0)
  // drain step outside the K-loop (tile boundary / last tile): the four slices back to back
#define LT_EMIT(C, P) do { LT_PA(true, C, P, NL); LT_PB(true, P); LT_PC(true); LT_PD(true, C, P); } while (0)
  // VAR 0: plain K-tile; 1: first K-tile of a drain pair (passes 0, 1, 2 of chunk C in ks0, ks1, ks3); 2: second (pass 3 in ks0, chunk
  // C + 1 staged in ks1 / ks2).  S: the stage it computes on.  Z: first K-tile of an output tile.
#define LT_KTILE(VAR, C, S, Z)                                                                                 \
  do {                                                                                                         \
    /* ---- ks 0: B pieces of the next K-tile */                                                               \
    LT_WAIT(0);                                                                                                \
    LT_READ(1, S, 1);                                                                                          \
    LT_STEP(0, Z, LT_PA((VAR) != 0, C, ((VAR) == 2 ? 3 : 0), ((VAR) == 2 ? NL + 6 : ((Z) ? NL : NL + 14))), LT_DMA_B(1 - (S), 0), LT_DMA_B(1 - (S), 1), LT_DMA_B(1 - (S), 2),  \
            LT_DMA_B(1 - (S), 3), LT_DMA_B(1 - (S), 4), LT_DMA_B(1 - (S), 5), LT_DMA_B(1 - (S), 6), LT_DMA_B(1 - (S), 7),        \
            LT_PB((VAR) != 0, ((VAR) == 2 ? 3 : 0)), LT_PC((VAR) != 0),                                 \
            do { LT_PD((VAR) != 0, C, ((VAR) == 2 ? 3 : 0)); LT_ADV_B(); } while (0));                          \
    /* ---- ks 1 */                                                                                            \
    LT_WAIT(1);                                                                                                \
    LT_READ(0, S, 2);                                                                                          \
    LT_STEP(1, false, LT_PA((VAR) == 1, C, 1, ((Z) ? NL + 8 : NL + 14)), LT_NOP, do { if ((VAR) == 2) LT_STAGE_HALF((C) + 1, 0); } while (0), LT_PB((VAR) == 1, 1), \
            LT_NOP, LT_NOP, LT_PC((VAR) == 1), LT_NOP, LT_NOP, LT_PD((VAR) == 1, C, 1), LT_NOP, LT_NOP);         \
    /* ---- ks 2 */                                                                                            \
    LT_WAIT(0);                                                                                                \
    LT_READ(1, S, 3);                                                                                          \
    LT_STEP(0, false, LT_NOP, do { if ((VAR) == 2) LT_STAGE_HALF((C) + 1, 1); } while (0), LT_NOP, LT_NOP, LT_NOP, LT_NOP, LT_NOP, LT_NOP, \
            LT_NOP, LT_NOP, LT_NOP, LT_NOP);                                                                   \
    LT_WAIT(1);                                                                                                \
    if ((VAR) == 1) __builtin_amdgcn_s_waitcnt(0xF70 + 2 * NL);                                                \
    else if ((VAR) == 2) __builtin_amdgcn_s_waitcnt(0xF70 + NL);                                               \
    else __builtin_amdgcn_s_waitcnt(0xF70);                                                                    \
    LT_BARRIER();                                                                                              \
    /* ---- ks 3: A pieces of the K-tile after next */                                                         \
    LT_READ(0, 1 - (S), 0);                                                                                    \
    LT_STEP(1, false, LT_PA((VAR) == 1, C, 2, NL), LT_DMA_A(S, 0), LT_DMA_A(S, 1), LT_DMA_A(S, 2), LT_DMA_A(S, 3), LT_DMA_A(S, 4), LT_DMA_A(S, 5), \
            LT_ADV_A(), LT_NOP, LT_PB((VAR) == 1, 2), LT_PC((VAR) == 1), LT_PD((VAR) == 1, C, 2));               \
  } while (0)

  for (int ti = 0; ti < my_n; ++ti) {
    int m0, n0;
    tile_origin(ti, m0, n0);
    // bias of this wave's 128 columns -> its LDS cache, LayerScale gamma -> the idle half of the gamma cache: landed once the first
    // K-tile's DMA wait has passed (they are older than its DMAs), read at the tile boundary
    dma<4>(wreg + BIASC, rBias, (n0 + wn * 128 + l) * 4, 0);
    dma<4>(wreg + BIASC + 256, rBias, (n0 + wn * 128 + 64 + l) * 4, 0);
    if (EPI == EPI_RESID && g.gamma) {
      dma<4>(wreg + GAMC + (1 - gcur) * 512, rGam, (n0 + wn * 128 + l) * 4, 0);
      dma<4>(wreg + GAMC + (1 - gcur) * 512 + 256, rGam, (n0 + wn * 128 + 64 + l) * 4, 0);
    }
    const int npairs = (have_stash && !(dbg & 1)) ? min(6, nk >> 1) : 0;
    int t = 2;
    if (npairs > 0) {
      LT_KTILE(1, 0, 0, true);
      LT_KTILE(2, 0, 1, false);
      for (int c = 1; c < npairs; ++c) {
        LT_KTILE(1, c, 0, false);
        LT_KTILE(2, c, 1, false);
        t += 2;
      }
    } else {
      LT_KTILE(0, 0, 0, true);
      LT_KTILE(0, 0, 1, false);
    }
    for (; t < nk; t += 2) {
      LT_KTILE(0, 0, 0, false);
      LT_KTILE(0, 0, 1, false);
    }
    // ---- tile boundary.  The fragments of the next K-tile's first step are in flight into qa[0] / qb[0]: retire them before any
    // compiler-scheduled code runs over the register file
    LT_WAIT(0);
    if (have_stash) {
      // K < 768 (or no drain): chunks the K-loop had no room for.  Chunk `npairs` is already in the window (staged by the last pair,
      // or at the previous boundary when there was none); the ring already holds the first two passes of the first of them.
      for (int c = npairs; c < 6; ++c) {
        if (c > npairs) { LT_STAGE_HALF(c, 0); LT_STAGE_HALF(c, 1); }
        LT_EMIT(c, 0); LT_EMIT(c, 1); LT_EMIT(c, 2); LT_EMIT(c, 3);
      }
    }
    // accumulators -> bf16 stash (alpha * acc + bias, rounded once: what the reference's autocast Linear returns).  An MFMA's result
    // may be read 12 wait states after its issue at the earliest: the last twelve of the K-loop were issued just now.
    asm volatile("s_nop 15\n\ts_nop 7");
    __builtin_amdgcn_s_waitcnt(0xF70);   // bias / gamma arrived as 4-byte pieces: not trusted to retire in order with the 16-byte ones
    {   // ... and one round trip through the LDS queue before they are read (see pass_a)
      unsigned dummy_;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dummy_) : "v"((unsigned)(uintptr_t)(lptr_t*)(wreg + BIASC)) : "memory");
    }
    {
      const float* bc = reinterpret_cast<const float*>(wreg + BIASC);
      const float bj0 = bc[l & 31], bj1 = bc[32 + (l & 31)], bj2 = bc[64 + (l & 31)], bj3 = bc[96 + (l & 31)];
      stash_block<0, 0>(stv, g.alpha, bj0); stash_block<0, 1>(stv, g.alpha, bj1); stash_block<0, 2>(stv, g.alpha, bj2); stash_block<0, 3>(stv, g.alpha, bj3);
      stash_block<1, 0>(stv, g.alpha, bj0); stash_block<1, 1>(stv, g.alpha, bj1); stash_block<1, 2>(stv, g.alpha, bj2); stash_block<1, 3>(stv, g.alpha, bj3);
      stash_block<2, 0>(stv, g.alpha, bj0); stash_block<2, 1>(stv, g.alpha, bj1); stash_block<2, 2>(stv, g.alpha, bj2); stash_block<2, 3>(stv, g.alpha, bj3);
    }
    prow0 = m0 + wm * 96; pcol0 = n0 + wn * 128;
    have_stash = true;
    gcur = 1 - gcur;
    LT_STAGE_HALF(0, 0); LT_STAGE_HALF(0, 1);
    if (NL) {   // prime the ring: passes 0 and 1 of chunk 0
      ring_fetch<EPI, 0>(e, wreg, pass_pos(prow0, pcol0, 0, 0, l, e.N), l);
      ring_fetch<EPI, 1>(e, wreg, pass_pos(prow0, pcol0, 0, 1, l, e.N), l);
    }
  }
  // ---- the last tile's epilogue: nothing left to hide it under
  for (int c = 0; c < 6; ++c) {
    if (c > 0) { LT_STAGE_HALF(c, 0); LT_STAGE_HALF(c, 1); }
    LT_EMIT(c, 0); LT_EMIT(c, 1); LT_EMIT(c, 2); LT_EMIT(c, 3);
  }
#undef LT_KTILE
#undef LT_EMIT
#undef LT_NOP
#undef LT_PA
#undef LT_PB
#undef LT_PC
#undef LT_PD
#undef LT_STEP
#undef LT_MF
#undef LT_DMA_A_ALL
#undef LT_DMA_B_ALL
#undef LT_ADV_A
#undef LT_ADV_B
#undef LT_DMA_A
#undef LT_DMA_B
#undef LT_BARRIER
#undef LT_WAIT
#undef LT_READ
}

template <bool TB, int EPI>
int launch_one(const GemmArgs& g, int wb, int grid, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm1p_kernel<TB, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) { lt_set_error("lt_gemm_bf16: cannot enable %d B of LDS: %s", LDS_TOTAL, hipGetErrorString(e)); return LT_ERR_HIP; }
    configured = true;
  }
  static const int dbg = [] { const char* s = getenv("LT_GEMM_1P_DBG"); return s ? atoi(s) : 0; }();   // 1: no drain under the K-loop (diagnostic)
  hipLaunchKernelGGL((gemm1p_kernel<TB, EPI>), dim3(grid), dim3(256), LDS_TOTAL, st, g, wb, dbg);
  return LT_OK;
}
template <bool TB>
int launch_epi(const GemmArgs& g, int epi, int wb, int grid, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_one<TB, EPI_BF16>(g, wb, grid, st);
    case EPI_BF16_GELU: return launch_one<TB, EPI_BF16_GELU>(g, wb, grid, st);
    case EPI_RESID: return launch_one<TB, EPI_RESID>(g, wb, grid, st);
    case EPI_BF16_GELUGRAD: return launch_one<TB, EPI_BF16_GELUGRAD>(g, wb, grid, st);
    default: lt_set_error("lt_gemm_bf16: the persistent kernel has no epilogue %d", epi); return LT_ERR_INVALID;
  }
}

}  // namespace

bool gemm1p_eligible(const GemmArgs& g, int epi, bool trans_a) {
  if (trans_a) return false;
  if (!(epi == EPI_BF16 || epi == EPI_BF16_GELU || epi == EPI_RESID || epi == EPI_BF16_GELUGRAD)) return false;
  if (epi == EPI_RESID && (g.C2 || !g.resid)) return false;
  if (g.K % (2 * PBK) != 0 || g.N % 64 != 0 || g.N < 256 || g.M < 192) return false;
  const bool f32 = epi == EPI_RESID;
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  if (!al16(g.C) || g.ldc % (f32 ? 4 : 8) != 0) return false;
  if (g.C2 && (!al16(g.C2) || g.ldc2 % 8 != 0)) return false;
  if (g.resid && (!al16(g.resid) || g.ldr % 4 != 0)) return false;
  if (g.aux && (!al16(g.aux) || g.ldaux % 8 != 0)) return false;
  // buffer instructions address 32-bit byte offsets (and the out-of-range marker is 2^31)
  const long lim = 0x7FFFFFF0L - 4L * PBM * (g.ldc > g.lda ? g.ldc : g.lda);
  if ((long)g.M * g.ldc * (f32 ? 4 : 2) > lim || (g.C2 && (long)g.M * g.ldc2 * 2 > lim) || (g.resid && (long)g.M * g.ldr * 4 > lim) ||
      (g.aux && (long)g.M * g.ldaux * 2 > lim) || (long)g.M * g.lda * 2 > lim || (long)(g.N + PBN) * g.ldb * 2 > lim || (long)g.K * g.ldb * 2 > lim)
    return false;
  return true;
}

int gemm1p_launch(const GemmArgs& g0, int epi, bool trans_b, hipStream_t st) {
  GemmArgs g = g0;
  g.tiles_m = lt_cdiv(g.M, PBM); g.tiles_n = lt_cdiv(g.N, PBN);
  static int cus = [] { int d = 0, n = 256; if (hipGetDevice(&d) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) n = 256; return n > 0 ? n : 256; }();
  const int nt = g.tiles_m * g.tiles_n;
  int grid = nt < cus ? nt : cus;
  static const int env_grid = [] { const char* s = getenv("LT_GEMM_1P_GRID"); return s ? atoi(s) : 0; }();   // diagnostic: fewer workgroups = more tiles each
  if (env_grid > 0 && env_grid < grid) grid = env_grid;
  // LT_GEMM_1P_TPW (read per call): tiles per workgroup -> more workgroups than CUs, each walking only a few tiles, so that workgroups
  // of other streams' kernels get CUs at the granularity of a few tiles instead of a whole launch
  if (const char* s = getenv("LT_GEMM_1P_TPW")) { const int tpw = atoi(s); if (tpw > 0) { const int gq = (nt + tpw - 1) / tpw; if (gq > grid) grid = gq; } }
  // column band: 4 tiles (or 3 where that divides and 4 does not), the whole width below 5
  int wb = g.tiles_n <= 4 ? g.tiles_n : (g.tiles_n % 4 == 0 ? 4 : (g.tiles_n % 3 == 0 ? 3 : 4));
  static const int env_wb = [] { const char* s = getenv("LT_GEMM_1P_BAND"); return s ? atoi(s) : 0; }();
  if (env_wb > 0) wb = env_wb < g.tiles_n ? env_wb : g.tiles_n;
  return trans_b ? launch_epi<true>(g, epi, wb, grid, st) : launch_epi<false>(g, epi, wb, grid, st);
}

}  // namespace lt_gemm
