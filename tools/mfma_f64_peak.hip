// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (the ceiling the dense synthesis GEMM is priced
// against).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double seed) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 2e-3;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
  int ncu = 256;
  int blocks = ncu * blocks_per_cu;
  double* d;
  hipMalloc(&d, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(d, 100, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(d, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * NACC * 2048.0;
  printf("acc=%d blocks/CU=%d  %.2f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / ms * 1e-9);
  hipFree(d);
}

int main() {
  run<1>(1, 5000);
  run<2>(1, 5000);
  run<4>(1, 5000);
  run<16>(1, 2500);
  run<4>(2, 5000);
  run<16>(2, 2500);
  run<16>(2, 20000);
  return 0;
}
