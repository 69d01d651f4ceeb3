// Ceiling probes for the matrix loops of s2i_igemm.hip / s2i_bf16.hip (dev tool, GPU box):
//   reg  : v_mfma with both operands in registers (no LDS, no memory): the instruction's own rate under this clock;
//   lds  : the same MFMA stream with every fragment re-read from LDS (the access pattern of mma_chunk / conv_bf16_kernel,
//          conflict-free), no global traffic and no barriers: the ceiling of an LDS-fed loop;
// for fp32 (v_mfma_f32_32x32x2_f32, 4 x ds_read_b32 per 4 MFMAs) and bf16 (v_mfma_f32_32x32x16_bf16, 4 x ds_read_b128 per
// 4 MFMAs), at 1 / 2 / 3 resident 256-thread blocks per CU, on random data.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_microbench.hip -o tools/microbench/mfma_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool LDS>
__global__ __launch_bounds__(256) void f32_loop(const float* __restrict__ src, float* __restrict__ dst, int iters) {
  extern __shared__ float sm[];  // [32 k][2 * 128 + pad]
  constexpr int LD = 257;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 32 * LD; e += 256) sm[e] = src[e];
  __syncthreads();
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float a[2] = {src[tid], src[tid + 256]}, b[2] = {src[tid + 512], src[tid + 768]};
  const float* ap = sm + lh * LD + (wave >> 1) * 64 + l31;
  const float* bp = sm + lh * LD + 128 + (wave & 1) * 64 + l31;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      if (LDS) {
        a[0] = ap[2 * kk * LD]; a[1] = ap[2 * kk * LD + 32];
        b[0] = bp[2 * kk * LD]; b[1] = bp[2 * kk * LD + 32];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  dst[blockIdx.x * 256 + tid] = s;
}

template <bool LDS>
__global__ __launch_bounds__(256) void bf16_loop(const unsigned short* __restrict__ src, float* __restrict__ dst, int iters) {
  extern __shared__ unsigned char smb[];  // A: 128 rows x 64 B, B: 128 rows x 64 B, XOR-swizzled 16-byte segments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 2 * 128 * 64 / 16; e += 256)
    reinterpret_cast<uint4*>(smb)[e] = reinterpret_cast<const uint4*>(src)[e];
  __syncthreads();
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) {
    a[i] = *reinterpret_cast<const bf16x8*>(src + (tid + 256 * i) * 8);
    b[i] = *reinterpret_cast<const bf16x8*>(src + (tid + 256 * (i + 2)) * 8);
  }
  const unsigned char* ap = smb + ((wave >> 1) * 64 + l31) * 64;
  const unsigned char* bp = smb + 128 * 64 + ((wave & 1) * 64 + l31) * 64;
  const int sw = (l31 >> 2) & 3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (LDS) {
        const int so = ((ks * 2 + lh) ^ sw) << 4;
        a[0] = *reinterpret_cast<const bf16x8*>(ap + so); a[1] = *reinterpret_cast<const bf16x8*>(ap + 32 * 64 + so);
        b[0] = *reinterpret_cast<const bf16x8*>(bp + so); b[1] = *reinterpret_cast<const bf16x8*>(bp + 32 * 64 + so);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  dst[blockIdx.x * 256 + tid] = s;
}

template <typename K, typename S>
double run(K kern, const S* src, float* dst, int blocks, size_t shmem, int iters, double flop_per_iter_per_wave) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shmem, 0, src, dst, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shmem, 0, src, dst, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return flop_per_iter_per_wave * iters * 4.0 * blocks * 5 / (ms * 1e-3) / 1e12;
}

int main() {
  const int n = 1 << 16;
  std::vector<float> hf(n);
  std::vector<unsigned short> hb(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    hf[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    unsigned u; float f = hf[i]; memcpy(&u, &f, 4); hb[i] = (unsigned short)(u >> 16);
  }
  float *df, *dst; unsigned short* db;
  hipMalloc(&df, n * 4); hipMalloc(&db, n * 2); hipMalloc(&dst, 4096 * 256 * 4);
  hipMemcpy(df, hf.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), n * 2, hipMemcpyHostToDevice);
  const size_t shf = 32 * 257 * 4, shb = 2 * 128 * 64;
  hipFuncSetAttribute((const void*)f32_loop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipFuncSetAttribute((const void*)f32_loop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  printf("%-38s %10s %10s %10s   (TFLOP/s, random data)\n", "loop", "1 blk/CU", "2 blk/CU", "3 blk/CU");
  const double f32_flop = 16 * 4 * 2.0 * 32 * 32 * 2, bf_flop = 2 * 4 * 2.0 * 32 * 32 * 16;
  double r[4][3];
  for (int k = 1; k <= 3; ++k) {
    // pad LDS so that exactly k blocks fit a CU (160 KB): occupancy is set by the allocation
    const size_t pad = k == 1 ? 100 * 1024 : (k == 2 ? 60 * 1024 : 40 * 1024);
    hipFuncSetAttribute((const void*)f32_loop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void*)f32_loop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void*)bf16_loop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void*)bf16_loop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const int blocks = 256 * k * 4;
    r[0][k - 1] = run(f32_loop<false>, df, dst, blocks, shf > pad ? shf : pad, 400, f32_flop);
    r[1][k - 1] = run(f32_loop<true>, df, dst, blocks, shf > pad ? shf : pad, 400, f32_flop);
    r[2][k - 1] = run(bf16_loop<false>, db, dst, blocks, shb > pad ? shb : pad, 4000, bf_flop);
    r[3][k - 1] = run(bf16_loop<true>, db, dst, blocks, shb > pad ? shb : pad, 4000, bf_flop);
  }
  const char* names[4] = {"f32 32x32x2, operands in registers", "f32 32x32x2, fragments from LDS", "bf16 32x32x16, operands in registers",
                          "bf16 32x32x16, fragments from LDS"};
  for (int i = 0; i < 4; ++i) printf("%-38s %10.1f %10.1f %10.1f\n", names[i], r[i][0], r[i][1], r[i][2]);
  return 0;
}
