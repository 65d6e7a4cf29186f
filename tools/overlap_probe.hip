// Probe: does the MFMA-bound synthesis GEMM overlap with the HBM-bound spline passes when they run on two streams?
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iscri_amd/csrc -Iinclude tools/overlap_probe.hip \
//        scri_amd/csrc/kernels_gemm.o scri_amd/csrc/kernels_spline.o -o tools/overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "kernels.h"
using namespace bms;
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const long long N = 100000, K = 285, P = 1369;
  const int chunks = argc > 1 ? atoi(argv[1]) : 8;
  const long long ldg = 2 * P + (16 - (2 * P) % 16) % 16, ldb = ((2 * P + 127) / 128) * 128;
  double *A, *B, *Y[2], *R[2], *G, *x;
  SplineTable* tab;
  CK(hipMalloc(&A, N * K * 16));
  CK(hipMalloc(&B, 288 * ldb * 8));
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&Y[i], N * ldg * 8)); CK(hipMalloc(&R[i], N * ldg * 8)); }
  CK(hipMalloc(&G, N * 2 * P * 8));
  CK(hipMalloc(&x, N * 8));
  CK(hipMalloc(&tab, N * sizeof(SplineTable)));
  CK(hipMemset(A, 0, N * K * 16));
  CK(hipMemset(B, 0, 288 * ldb * 8));
  std::vector<double> hx(N);
  for (long long i = 0; i < N; ++i) hx[i] = 0.1 * i;
  CK(hipMemcpy(x, hx.data(), N * 8, hipMemcpyHostToDevice));
  hipStream_t s1, s2;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;   // 0: plain streams; 1: CU masks, first G bits / rest; 2: per 32-bit word
  const int gcu = argc > 3 ? atoi(argv[3]) : 192;  // CUs given to the GEMM stream
  if (mode == 0) {
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
  } else {
    uint32_t m1[8] = {0}, m2[8] = {0};
    for (int b = 0; b < 256; ++b) {
      const bool to_gemm = mode == 1 ? b < gcu : (b & 31) < gcu / 8;
      (to_gemm ? m1 : m2)[b >> 5] |= 1u << (b & 31);
    }
    CK(hipExtStreamCreateWithCUMask(&s1, 8, m1));
    CK(hipExtStreamCreateWithCUMask(&s2, 8, m2));
  }
  CK(launch_spline_table(s1, x, N, tab));
  for (int i = 0; i < 2; ++i) { CK(hipMemsetAsync(Y[i], 0, N * ldg * 8, s1)); CK(hipMemsetAsync(R[i], 0, N * ldg * 8, s1)); }
  CK(hipDeviceSynchronize());
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto gemm = [&](hipStream_t s, double* y, long long r0, long long rows) {
    return launch_zgemm3m(s, A + r0 * K * 2, 2 * K, B, ldb, y + r0 * ldg, ldg, rows, (int)P, (int)K, nullptr, nullptr);
  };
  auto spl = [&](hipStream_t s, double* y, double* r, long long r0, long long rows) {
    hipError_t e = launch_spline_forward(s, y + r0 * ldg, r + r0 * ldg, ldg, (int)P, r0, rows, N, x, tab, 320, 32);
    if (e != hipSuccess) return e;
    long long i1 = r0 + rows - 40 > r0 ? r0 + rows - 40 : r0;
    return launch_spline_backward_eval(s, y + r0 * ldg, r + r0 * ldg, ldg, (int)P, r0, rows, N, x, tab, 320, 32, x, nullptr, nullptr,
                                       0.0, r0 + 40, i1, G + (r0 + 40) * 2 * P, 2 * P);
  };
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipDeviceSynchronize());
    auto t0 = now();
    CK(gemm(s1, Y[0], 0, N));
    CK(hipDeviceSynchronize());
    auto t1 = now();
    CK(spl(s1, Y[0], R[0], 0, N));
    CK(hipDeviceSynchronize());
    auto t2 = now();
    // both at once on two streams, independent buffers (upper bound of what pipelining could give)
    CK(gemm(s1, Y[0], 0, N));
    CK(spl(s2, Y[1], R[1], 0, N));
    CK(hipDeviceSynchronize());
    auto t3 = now();
    // chunk pipeline: gemm(c+1) on s1 while spline(c) on s2, double-buffered by chunk parity (same buffers, disjoint rows)
    const long long cr = (N + chunks - 1) / chunks;
    std::vector<hipEvent_t> ev(chunks);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto t4 = now();
    for (int c = 0; c < chunks; ++c) {
      const long long r0 = c * cr, rows = (r0 + cr <= N ? cr : N - r0);
      CK(gemm(s1, Y[0], r0, rows));
      CK(hipEventRecord(ev[c], s1));
      CK(hipStreamWaitEvent(s2, ev[c], 0));
      CK(spl(s2, Y[0], R[0], r0, rows));
    }
    CK(hipDeviceSynchronize());
    auto t5 = now();
    printf("gemm %.3f ms | splines %.3f ms | sum %.3f | concurrent %.3f ms | %d-chunk pipeline %.3f ms\n", ms(t0, t1), ms(t1, t2),
           ms(t0, t2), ms(t2, t3), chunks, ms(t4, t5));
    for (auto& e : ev) hipEventDestroy(e);
  }
  return 0;
}
