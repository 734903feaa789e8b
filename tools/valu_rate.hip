// Issue-rate micro-benchmark for the instructions of the attention softmax loop and the in-LDS GroupNorm + SiLU (gfx950): cycles per
// wave64 instruction measured with s_memtime inside the kernel, for one wave per SIMD and for two, and - the question the round-4
// attention experiments raised - whether a transcendental-heavy wave and an MFMA wave that SHARE a SIMD overlap or serialise.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/_pk/valu_rate && tools/_pk/valu_rate
// Every loop body is 64 independent instructions (16 registers x 4 passes) in one asm block, repeated ITER times; the reported figure is
// (t_end - t_start) / (64 * ITER) of wave 0 of block 0 (all CUs run the same code at the same time: the chip is loaded, clocks are what
// they are under load).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define R16(OP)                                                                                                         \
  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

enum { OP_EXP, OP_RCP, OP_FMA, OP_ADD, OP_MAX3, OP_CVT, OP_MFMA, OP_EXP_BESIDE_MFMA, OP_FMA_BESIDE_MFMA, OP_MFMA_BESIDE_EXP, N_OPS };

template <int OP>
__device__ __forceinline__ void body(float (&v)[16], f32x16 (&acc)[4], const bf16x8& a, const bf16x8& b) {
  if constexpr (OP == OP_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    R16(X) R16(X) R16(X) R16(X)
#undef X
  } else if constexpr (OP == OP_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    R16(X) R16(X) R16(X) R16(X)
#undef X
  } else if constexpr (OP == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
    R16(X) R16(X) R16(X) R16(X)
#undef X
  } else if constexpr (OP == OP_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
    R16(X) R16(X) R16(X) R16(X)
#undef X
  } else if constexpr (OP == OP_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[i]));
    R16(X) R16(X) R16(X) R16(X)
#undef X
  } else if constexpr (OP == OP_CVT) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
    R16(X) R16(X) R16(X) R16(X)
#undef X
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r)               // 64 MFMAs on four independent accumulators
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
}

// role 0: every wave runs OP.  role 1 (blocks of 8 waves): waves 0-3 run OPA, waves 4-7 run OPB - with the usual cyclic wave -> SIMD
// placement every SIMD then holds one wave of each kind.  out[block * 8 + wave] = cycles per instruction of that wave.
template <int OPA, int OPB>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, float seed) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * (float)(threadIdx.x + i);
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (float)(threadIdx.x & 7)); b[i] = (__bf16)0.5f; }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) body<OPA>(v, acc, a, b);
  } else {
    for (int it = 0; it < iters; ++it) body<OPB>(v, acc, a, b);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float keep = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) keep += v[i];
#pragma unroll
  for (int c = 0; c < 4; ++c) keep += acc[c][0] + acc[c][15];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (float)(t1 - t0) / (64.f * (float)iters);
  if (keep == 123.456f) out[0] = keep;
}

template <int OPA, int OPB>
static void run(const char* name, int threads, float* dout, std::vector<float>& h) {
  const int blocks = 256, iters = 400;
  hipLaunchKernelGGL((rate_kernel<OPA, OPB>), dim3(blocks), dim3(threads), 0, 0, dout, iters, 0.3f);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((rate_kernel<OPA, OPB>), dim3(blocks), dim3(threads), 0, 0, dout, iters, 0.3f);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
  double lo = 0, hi = 0;
  int nw = threads / 64;
  for (int w = 0; w < nw; ++w) (w < 4 ? lo : hi) += h[w];
  lo /= (nw < 4 ? nw : 4);
  if (nw > 4) hi /= (nw - 4);
  printf("%-44s %d wave(s) per SIMD: waves 0-3 %6.1f", name, nw / 4, lo);
  if (nw > 4) printf("   waves 4-7 %6.1f", hi);
  printf("   s_memtime ticks per instruction (kernel %.3f ms)\n", ms);
}

int main() {
  float* dout;
  CK(hipMalloc(&dout, 256 * 8 * 4));
  std::vector<float> h(256 * 8);
  run<OP_EXP, OP_EXP>("v_exp_f32", 256, dout, h);
  run<OP_EXP, OP_EXP>("v_exp_f32", 512, dout, h);
  run<OP_RCP, OP_RCP>("v_rcp_f32", 256, dout, h);
  run<OP_FMA, OP_FMA>("v_fma_f32", 256, dout, h);
  run<OP_FMA, OP_FMA>("v_fma_f32", 512, dout, h);
  run<OP_ADD, OP_ADD>("v_add_f32", 256, dout, h);
  run<OP_MAX3, OP_MAX3>("v_max3_f32", 256, dout, h);
  run<OP_CVT, OP_CVT>("v_cvt_pk_bf16_f32", 256, dout, h);
  run<OP_MFMA, OP_MFMA>("v_mfma_f32_32x32x16_bf16 (4 accumulators)", 256, dout, h);
  run<OP_MFMA, OP_MFMA>("v_mfma_f32_32x32x16_bf16 (4 accumulators)", 512, dout, h);
  run<OP_MFMA, OP_EXP>("mfma (waves 0-3) beside v_exp_f32 (waves 4-7)", 512, dout, h);
  run<OP_MFMA, OP_FMA>("mfma (waves 0-3) beside v_fma_f32 (waves 4-7)", 512, dout, h);
  run<OP_EXP, OP_FMA>("v_exp_f32 (waves 0-3) beside v_fma_f32 (4-7)", 512, dout, h);
  run<OP_EXP, OP_MFMA>("v_exp_f32 (waves 0-3) beside mfma (waves 4-7)", 512, dout, h);
  run<OP_FMA, OP_MFMA>("v_fma_f32 (waves 0-3) beside mfma (waves 4-7)", 512, dout, h);
  run<OP_RCP, OP_RCP>("v_rcp_f32", 512, dout, h);
  return 0;
}
