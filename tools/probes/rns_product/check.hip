// Stand-alone check and timing of the int8 residue product (scri_amd/csrc/kernels_gemm_rns.hip) against the fp64 3M
// product (kernels_gemm.hip) and a long-double host product on a sample of entries.
// Build: make -C scri_amd/csrc && hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iscri_amd/csrc -Iinclude tools/rns_product_check.hip \
//        scri_amd/csrc/kernels_gemm.o scri_amd/csrc/kernels_gemm_rns.o -o tools/rns_product_check
// Usage: tools/rns_product_check [M] [N] [K] [decades of dynamic range along k]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "kernels.h"

namespace bms {
size_t rns_planes_bytes(long long n, int K);
hipError_t launch_rns_residues(hipStream_t stream, const double* X, long long s_n, long long s_k, long long n, int K, int8_t* planes, double* scale,
                               double* inv_scale);
hipError_t launch_zgemm_rns(hipStream_t stream, const int8_t* Ap, const double* a_inv, const int8_t* Bp, const double* b_inv, double* C, long long ldc,
                            long long M, int N, int K, const double* col_off, const double* col_scale);
extern int rns_supertile_rows_log2, rns_knock, rns_nbuf;
}  // namespace bms

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char** argv) {
  const long long M = argc > 1 ? atoll(argv[1]) : 100405;
  const int N = argc > 2 ? atoi(argv[2]) : 1297;
  const int K = argc > 3 ? atoi(argv[3]) : 286;
  const double decades = argc > 4 ? atof(argv[4]) : 6.0;
  if (getenv("RNS_ST")) bms::rns_supertile_rows_log2 = atoi(getenv("RNS_ST"));
  if (getenv("RNS_KNOCK")) bms::rns_knock = atoi(getenv("RNS_KNOCK"));
  if (getenv("RNS_NBUF")) bms::rns_nbuf = atoi(getenv("RNS_NBUF"));
  const bool quick = getenv("RNS_QUICK") != nullptr;  // timing only
  const int Np = (N + 63) / 64 * 64, Kp = (K + 7) / 8 * 8;
  const long long lda = 2LL * K, ldb = 2LL * Np, ldc = 2LL * Np;
  std::mt19937_64 rng(7);
  std::normal_distribution<double> g(0.0, 1.0);
  std::uniform_real_distribution<double> uexp(-3.0, 3.0);
  std::vector<double> A((size_t)M * lda), B((size_t)Kp * ldb, 0.0);
  for (long long i = 0; i < M; ++i) {
    const double rs = std::pow(10.0, uexp(rng));
    for (int k = 0; k < K; ++k) {
      const double w = rs * std::pow(10.0, -decades * k / K);
      A[i * lda + 2 * k] = w * g(rng), A[i * lda + 2 * k + 1] = w * g(rng);
    }
  }
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < N; ++j) B[k * ldb + 2 * j] = g(rng), B[k * ldb + 2 * j + 1] = g(rng);
  double *dA, *dB, *dC1, *dC2, *a_sc, *a_inv, *b_sc, *b_inv;
  int8_t *Ap, *Bp;
  CK(hipMalloc(&dA, A.size() * 8));
  CK(hipMalloc(&dB, B.size() * 8));
  CK(hipMalloc(&dC1, (size_t)M * ldc * 8));
  CK(hipMalloc(&dC2, (size_t)M * ldc * 8));
  CK(hipMalloc(&Ap, bms::rns_planes_bytes(M, K)));
  CK(hipMalloc(&Bp, bms::rns_planes_bytes(N, K)));
  CK(hipMalloc(&a_sc, M * 8));
  CK(hipMalloc(&a_inv, M * 8));
  CK(hipMalloc(&b_sc, Np * 8));
  CK(hipMalloc(&b_inv, Np * 8));
  CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dC1, 0, (size_t)M * ldc * 8));
  CK(hipMemset(dC2, 0, (size_t)M * ldc * 8));
  hipEvent_t e0, e1, e2, e3;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&e2));
  CK(hipEventCreate(&e3));
  float t_f64 = 0, t_resA = 0, t_resB = 0, t_rns = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    CK(bms::launch_zgemm3m(0, dA, lda, dB, ldb, dC1, ldc, M, N, K, nullptr, nullptr));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&t_f64, e0, e1));
    CK(hipEventRecord(e0));
    CK(bms::launch_rns_residues(0, dA, lda, 2, M, K, Ap, a_sc, a_inv));
    CK(hipEventRecord(e1));
    CK(bms::launch_rns_residues(0, dB, 2, ldb, N, K, Bp, b_sc, b_inv));
    CK(hipEventRecord(e2));
    CK(bms::launch_zgemm_rns(0, Ap, a_inv, Bp, b_inv, dC2, ldc, M, N, K, nullptr, nullptr));
    CK(hipEventRecord(e3));
    CK(hipEventSynchronize(e3));
    CK(hipEventElapsedTime(&t_resA, e0, e1));
    CK(hipEventElapsedTime(&t_resB, e1, e2));
    CK(hipEventElapsedTime(&t_rns, e2, e3));
  }
  const double flops = 8.0 * M * N * K;
  printf("M=%lld N=%d K=%d: fp64 3M product %.3f ms (%.1f TFLOP/s in 8-flop units) | residues of A %.3f ms, of B %.3f ms, int8 product %.3f ms (%.1f TFLOP/s fp64-equivalent; with the residues of A %.1f)\n",
         M, N, K, t_f64, flops / t_f64 * 1e-9, t_resA, t_resB, t_rns, flops / t_rns * 1e-9, flops / (t_rns + t_resA) * 1e-9);
  if (quick) return 0;
  std::vector<double> C1((size_t)M * ldc), C2((size_t)M * ldc);
  CK(hipMemcpy(C1.data(), dC1, C1.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(C2.data(), dC2, C2.size() * 8, hipMemcpyDeviceToHost));
  // the two against each other everywhere, and both against a long-double product on a sample
  double worst12 = 0.0;
  std::vector<double> l1(M), bmax(N, 0.0);
  for (long long i = 0; i < M; ++i) {
    double s = 0;
    for (int k = 0; k < K; ++k) s += std::fabs(A[i * lda + 2 * k]) + std::fabs(A[i * lda + 2 * k + 1]);
    l1[i] = s;
  }
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < N; ++j) bmax[j] = std::fmax(bmax[j], std::fmax(std::fabs(B[k * ldb + 2 * j]), std::fabs(B[k * ldb + 2 * j + 1])));
  for (long long i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      const double sc = l1[i] * bmax[j] + 1e-300;
      const double d = std::fmax(std::fabs(C1[i * ldc + 2 * j] - C2[i * ldc + 2 * j]), std::fabs(C1[i * ldc + 2 * j + 1] - C2[i * ldc + 2 * j + 1]));
      if (!(d / sc <= worst12)) worst12 = d / sc;
    }
  double w1 = 0, w2 = 0;
  std::uniform_int_distribution<long long> ri(0, M - 1);
  std::uniform_int_distribution<int> rj(0, N - 1);
  for (int s = 0; s < 4000; ++s) {
    const long long i = s < 64 ? (s & 1 ? M - 1 - s : s) : ri(rng);
    const int j = s < 64 ? (s & 2 ? N - 1 - (s >> 2) : (s >> 2)) : rj(rng);
    long double re = 0, im = 0;
    for (int k = 0; k < K; ++k) {
      const long double ar = A[i * lda + 2 * k], ai = A[i * lda + 2 * k + 1], br = B[k * ldb + 2 * j], bi = B[k * ldb + 2 * j + 1];
      re += ar * br - ai * bi, im += ar * bi + ai * br;
    }
    const double sc = l1[i] * bmax[j] + 1e-300;
    w1 = std::fmax(w1, std::fmax(std::fabs((double)(C1[i * ldc + 2 * j] - re)), std::fabs((double)(C1[i * ldc + 2 * j + 1] - im))) / sc);
    w2 = std::fmax(w2, std::fmax(std::fabs((double)(C2[i * ldc + 2 * j] - re)), std::fabs((double)(C2[i * ldc + 2 * j + 1] - im))) / sc);
  }
  printf("errors relative to |a_row|_1 |b_col|_max: fp64 3M vs long double %.2e, int8 residues vs long double %.2e; the two against each other (all entries) %.2e\n", w1,
         w2, worst12);
  return (w2 < 1e-15 && worst12 < 1e-14) ? 0 : 2;
}
