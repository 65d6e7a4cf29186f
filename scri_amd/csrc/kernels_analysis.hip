// Separable SWSH analysis on the equiangular grid (spinsfast.map2salm, scri/waveform_grid.py:303-307,
// scri/asymptotic_bondi_data/transformations.py:419-429).
//
// The Huffenberger-Wandelt analysis collapses exactly (oracle/spinsfast_ref.py, tests/test_oracle_wigner.py) to
//     a_lm = sum_j [q_j / n_phi] sLambda_lm(theta_j) F_m(theta_j),      F_m(theta_j) = sum_k f_jk exp(-i m phi_k)
// with real theta-quadrature weights q_j.  Step 1 (phi-DFT) is a thin real GEMM [rows n_theta x 2 n_phi] . [2 n_phi x
// 2 (2L+1)] against the matrix built by dft_matrix_kernel (run through dgemm_mfma_kernel); step 2 is
// theta_quadrature_kernel below.  Compared with the dense quadrature GEMM this is ~10x less arithmetic and sums
// n_phi and n_theta terms per output instead of 2 n_pix (shorter fp64 chains => smaller rounding error).
#include "wigner.h"
#include "kernels.h"

namespace bms {

// B[2k][2mi] = cos(m phi_k), B[2k][2mi+1] = -sin(m phi_k), B[2k+1][2mi] = sin(m phi_k), B[2k+1][2mi+1] = cos(m phi_k);
// mi = m + L, phi_k = 2 pi k / n_phi; the angle is reduced in integers before the sincos.
__global__ __launch_bounds__(256) void dft_matrix_kernel(int n_phi, int L, double* __restrict__ B, long long ldb) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nm = 2 * L + 1;
  if (id >= n_phi * nm) return;
  const int k = id / nm, mi = id % nm, m = mi - L;
  long long r = ((long long)m * k) % n_phi;
  if (r < 0) r += n_phi;
  double s, c;
  sincospi(2.0 * (double)r / (double)n_phi, &s, &c);
  double* r0 = B + (2LL * k) * ldb + 2 * mi;
  double* r1 = r0 + ldb;
  r0[0] = c;
  r0[1] = -s;
  r1[0] = s;
  r1[1] = c;
}

hipError_t launch_dft_matrix(hipStream_t stream, int n_phi, int L, double* B, long long ldb) {
  const int n = n_phi * (2 * L + 1);
  hipLaunchKernelGGL(dft_matrix_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n_phi, L, B, ldb);
  return hipGetLastError();
}

// T[o][j] = w_theta[j] * Re sYlm_o(theta_j, phi = 0);  Y: c16[n_theta][n_out] from swsh_kernel<0> at the rotors R(theta_j, 0)
__global__ __launch_bounds__(256) void theta_table_kernel(const double* __restrict__ Y, const double* __restrict__ w_theta,
                                                          int n_theta, int n_out, double* __restrict__ T) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_theta * n_out) return;
  const int o = id / n_theta, j = id % n_theta;
  T[(long long)o * n_theta + j] = w_theta[j] * Y[((long long)j * n_out + o) * 2];
}

hipError_t launch_theta_table(hipStream_t stream, const double* Y, const double* w_theta, int n_theta, int n_out, double* T) {
  const int n = n_theta * n_out;
  hipLaunchKernelGGL(theta_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, Y, w_theta, n_theta, n_out, T);
  return hipGetLastError();
}

// a[t][o] = sum_j T[o][j] F[t][j][m(o)].  Thread <-> output mode o with its T row in registers; the F tile of one time
// step (n_theta x (2L+1) complex, 19.5 kB at cfg3) is staged through LDS; outputs are written 16 B per lane, coalesced.
template <int NT, int MAXT>
__global__ __launch_bounds__(MAXT) void theta_quadrature_kernel(const double* __restrict__ F, long long n_rows, int n_theta,
                                                                int nm /* 2L+1 */, int n_out,
                                                                const int* __restrict__ m_index /* [n_out] */,
                                                                const double* __restrict__ T, double* __restrict__ out,
                                                                long long ldo) {
  extern __shared__ double Fs[];  // [n_theta][2 nm]
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = o < n_out;
  double tj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) tj[j] = (live && j < n_theta) ? T[(long long)o * n_theta + j] : 0.0;
  const int mi = live ? m_index[o] : 0;
  const int tile = n_theta * 2 * nm;  // doubles
  for (long long t = blockIdx.y; t < n_rows; t += gridDim.y) {
    const double* src = F + t * tile;
    __syncthreads();  // previous tile fully consumed
    for (int e = threadIdx.x * 2; e < tile; e += blockDim.x * 2)
      *reinterpret_cast<double2*>(Fs + e) = *reinterpret_cast<const double2*>(src + e);
    __syncthreads();
    double ar = 0.0, ai = 0.0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < n_theta) {
        const double2 f = *reinterpret_cast<const double2*>(Fs + (j * nm + mi) * 2);
        ar = fma(tj[j], f.x, ar);
        ai = fma(tj[j], f.y, ai);
      }
    }
    if (live) *reinterpret_cast<double2*>(out + t * ldo + 2LL * o) = double2{ar, ai};
  }
}

hipError_t launch_theta_quadrature(hipStream_t stream, const double* F, long long n_rows, int n_theta, int nm, int n_out,
                                   const int* m_index, const double* T, double* out, long long ldo) {
  if (n_rows <= 0 || n_out <= 0) return hipSuccess;
  const size_t lds = sizeof(double) * (size_t)n_theta * 2 * nm;
  const long long by = n_rows < 2048 ? n_rows : 2048;
  // the T row lives in registers: NT doubles per thread, so larger grids get smaller workgroups
#define LAUNCH_TQ(NT, MAXT)                                                                                              \
  {                                                                                                                      \
    int threads = ((n_out + 63) / 64) * 64;                                                                              \
    if (threads > MAXT) threads = MAXT;                                                                                  \
    const int bx = (n_out + threads - 1) / threads;                                                                      \
    hipError_t e = hipFuncSetAttribute((const void*)theta_quadrature_kernel<NT, MAXT>,                                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
    if (e != hipSuccess) return e;                                                                                       \
    hipLaunchKernelGGL((theta_quadrature_kernel<NT, MAXT>), dim3(bx, (unsigned)by), dim3(threads), lds, stream, F, n_rows,  \
                       n_theta, nm, n_out, m_index, T, out, ldo);                                                        \
  }
  if (n_theta <= 24)
    LAUNCH_TQ(24, 1024)
  else if (n_theta <= 40)
    LAUNCH_TQ(40, 1024)
  else if (n_theta <= 72)
    LAUNCH_TQ(72, 512)
  else if (n_theta <= 104)
    LAUNCH_TQ(104, 256)
  else
    return hipErrorInvalidValue;
#undef LAUNCH_TQ
  return hipGetLastError();
}

}  // namespace bms
