// Hardware probe (test infrastructure): pins the gfx950 facts the kernels rely on.
//  1. ds_read_b64_tr_b16 lane/element semantics
//  2. v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 A/B/C fragment layouts
//  3. device properties
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__global__ void tr_probe(short* out, const int* lane_addr_elems) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int a = lane_addr_elems[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

__device__ inline __bf16 f2bf(float f) { return (__bf16)f; }

__global__ void mfma32_probe(const float* A, const float* B, float* D) {
  // A [32][16] row-major, B [16][32] row-major (k-major), D [32][32]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = f2bf(A[(l & 31) * 16 + (l >> 5) * 8 + j]);
    b[j] = f2bf(B[((l >> 5) * 8 + j) * 32 + (l & 31)]);
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = acc[r];
  }
}
__global__ void mfma16_probe(const float* A, const float* B, float* D) {
  // A [16][32], B [32][16], D [16][16]
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = f2bf(A[(l & 15) * 32 + (l >> 4) * 8 + j]);
    b[j] = f2bf(B[((l >> 4) * 8 + j) * 16 + (l & 15)]);
  }
  f32x4 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz mem=%.1f GB L2=%d sharedPerBlock=%zu maxShared=%zu\n", p.name, p.gcnArchName,
         p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1e9, p.l2CacheSize, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
  // ---- 1. tr16
  {
    short* dout; int* daddr; CK(hipMalloc(&dout, 64 * 4 * 2)); CK(hipMalloc(&daddr, 64 * 4));
    for (int variant = 0; variant < 3; ++variant) {
      int addr[64];
      for (int l = 0; l < 64; ++l) {
        if (variant == 0) addr[l] = l * 4;                       // fully linear
        else if (variant == 1) addr[l] = (l / 16) * 512 + (l % 16) * 4;  // each 16-group at its own 1KB
        else addr[l] = (l / 16) * 64 + ((l % 16) / 4) * 100 * 0 + ((l & 3) * 4) + ((l % 16) / 4) * 256;  // rows 256 elems apart
      }
      CK(hipMemcpy(daddr, addr, sizeof(addr), hipMemcpyHostToDevice));
      tr_probe<<<1, 64>>>(dout, daddr);
      short out[256]; CK(hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost));
      printf("tr16 variant %d (addr elems per lane: ", variant);
      for (int l = 0; l < 20; ++l) printf("%d ", addr[l]);
      printf("...)\n");
      for (int l = 0; l < 64; ++l) {
        printf("  lane %2d: %5d %5d %5d %5d", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
        if (l % 4 == 3) printf("\n");
      }
      // hypothesis: result lane i (in group g) elem j = value at addr[g*16 + j*4 + (i>>2)] + (i&3)
      int ok = 1;
      for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        int g = l / 16, i = l % 16;
        int expect = addr[g * 16 + j * 4 + (i >> 2)] + (i & 3);
        if (out[l * 4 + j] != (short)expect) ok = 0;
      }
      printf("tr16 variant %d hypothesis H1 (lane i elem j <- src lane j*4+(i>>2), sub (i&3)): %s\n", variant, ok ? "MATCH" : "MISMATCH");
    }
  }
  // ---- 2. MFMA layouts
  {
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32);
    srand(1);
    for (auto& x : A) x = (float)(rand() % 17 - 8);
    for (auto& x : B) x = (float)(rand() % 13 - 6);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
    float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    mfma32_probe<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("mfma 32x32x16 bf16 layout check: max err %g -> %s\n", err, err == 0 ? "MATCH" : "MISMATCH");
  }
  {
    std::vector<float> A(16 * 32), B(32 * 16), D(16 * 16), R(16 * 16);
    srand(2);
    for (auto& x : A) x = (float)(rand() % 17 - 8);
    for (auto& x : B) x = (float)(rand() % 13 - 6);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
    float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    mfma16_probe<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("mfma 16x16x32 bf16 layout check: max err %g -> %s\n", err, err == 0 ? "MATCH" : "MISMATCH");
  }
  return 0;
}
