// Spin-weighted spherical harmonics at a grid of rotors: the matrices the dense synthesis contracts with
// (sf.SWSH_grid at scri/waveform_grid.py:470-471; sf.Modes.evaluate at
// scri/asymptotic_bondi_data/transformations.py:324-334).
//
// One thread per (pixel, m): a single l-chain of the Wigner small-d recurrence gives
// sYlm(R_p) = (-1)^s sqrt((2l+1)/4pi) ea^(m-s) eb^(-s-m) d^l_{m,-s}(ra, rb) for every l at once (SwshChain: the
// recurrence runs in double-double, the matrix entries are correctly rounded fp64).
// The matrix is written directly in the *real* layout the fp64 MFMA GEMM consumes: the complex product
// (ar + i ai)(yr + i yi) over interleaved (re, im) data is the real product of the row [ar ai] with
//   | yr  yi |
//   |-yi  yr |
// so row 2k holds (yr, yi) and row 2k+1 holds (-yi, yr) at columns (2p, 2p+1).
#include "wigner.h"
#include "pixel_math.h"
#include "kernels.h"

namespace bms {

// MODE 0: complex values Y[p][k]; MODE 1: synthesis matrix (modes as rows), real 2K x 2N layout; MODE 2: quadrature
// matrix (pixels as rows), real layout; MODE 3: synthesis matrix as plain complex Y[k][p] (row pitch ldb doubles)
template <int MODE>
__global__ __launch_bounds__(256) void swsh_kernel(const double* __restrict__ rotors, const double* __restrict__ w_pix,
                                                   int n_pix, int spin, int ell_min, int ell_max,
                                                   double* __restrict__ out, long long ldb) {
  const int nm = 2 * ell_max + 1;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)n_pix * nm) return;
  // consecutive threads -> consecutive pixels (coalesced matrix rows)
  const int p = (int)(gid % n_pix);
  const int m = (int)(gid / n_pix) - ell_max;
  const double* q = rotors + 4LL * p;
  SwshChain ch;
  ch.init(m, spin, q[0], q[1], q[2], q[3]);
  const int n_modes = LM_total_size(ell_min, ell_max);
  for (int ell = ch.ell; ell <= ell_max; ++ell) {
    if (ell >= ell_min) {
      const cplx y = ch.value();
      const double yr = y.re, yi = y.im;
      const long long k = LM_index(ell, m, ell_min);
      if (MODE == 1) {
        double* r0 = out + (2 * k) * ldb + 2LL * p;
        double* r1 = r0 + ldb;
        r0[0] = yr;
        r0[1] = yi;
        r1[0] = -yi;
        r1[1] = yr;
      } else if (MODE == 3) {
        double* r0 = out + k * ldb + 2LL * p;
        r0[0] = yr;
        r0[1] = yi;
      } else if (MODE == 2) {
        const double w = w_pix[p];
        const double wr = w * yr, wi = -w * yi;  // w conj(Y)
        double* r0 = out + (2LL * p) * ldb + 2 * k;
        double* r1 = r0 + ldb;
        r0[0] = wr;
        r0[1] = wi;
        r1[0] = -wi;
        r1[1] = wr;
      } else {
        out[((long long)p * n_modes + k) * 2] = yr;
        out[((long long)p * n_modes + k) * 2 + 1] = yi;
      }
    }
    if (ell < ell_max) ch.next();
  }
}

hipError_t launch_swsh_matrix(hipStream_t stream, const double* rotors, int n_pix, int spin, int ell_min, int ell_max,
                              double* Bmat, long long ldb) {
  const long long n = (long long)n_pix * (2 * ell_max + 1);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(swsh_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rotors,
                     (const double*)nullptr, n_pix, spin, ell_min, ell_max, Bmat, ldb);
  return hipGetLastError();
}

hipError_t launch_swsh_matrix_complex(hipStream_t stream, const double* rotors, int n_pix, int spin, int ell_min,
                                      int ell_max, double* Bmat, long long ldb) {
  const long long n = (long long)n_pix * (2 * ell_max + 1);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(swsh_kernel<3>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rotors,
                     (const double*)nullptr, n_pix, spin, ell_min, ell_max, Bmat, ldb);
  return hipGetLastError();
}

hipError_t launch_quadrature_matrix(hipStream_t stream, const double* rotors, const double* w_pix, int n_pix, int spin,
                                    int ell_min, int ell_max, double* Wmat, long long ldw) {
  const long long n = (long long)n_pix * (2 * ell_max + 1);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(swsh_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rotors, w_pix, n_pix, spin,
                     ell_min, ell_max, Wmat, ldw);
  return hipGetLastError();
}

hipError_t launch_swsh_values(hipStream_t stream, const double* rotors, int n_pix, int spin, int ell_min, int ell_max,
                              double* Y) {
  const long long n = (long long)n_pix * (2 * ell_max + 1);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(swsh_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rotors,
                     (const double*)nullptr, n_pix, spin, ell_min, ell_max, Y, 0);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- per-pixel tables
// rotor grid, conformal factor k, supertranslation alpha, time-skew coefficients and the flavour-specific per-pixel
// terms (pixel_math.h: the same code the host runs for bms_shard_plan), one thread per pixel
__global__ __launch_bounds__(128) void pixel_tables_kernel(PixelSpec P, PixelOut O, int n_pix, const int* __restrict__ perm) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_pix) pixel_tables_one(P, O, p, perm ? perm[p] : p);
}

hipError_t launch_pixel_tables(hipStream_t stream, const PixelSpec& P, const PixelOut& O, int n_pix, const int* perm) {
  hipLaunchKernelGGL(pixel_tables_kernel, dim3((n_pix + 127) / 128), dim3(128), 0, stream, P, O, n_pix, perm);
  return hipGetLastError();
}

// Column order of the grids.  A boost makes the time skew of a pixel grow like -(v.r_p)(u - tt): at late times pixels
// that are neighbours on the grid sit tens to hundreds of output rows apart, and the spline evaluation, whose lanes are
// adjacent columns marching over the knots together, would write every output row in 16-byte pieces.  Storing the
// columns sorted by the skew rate keeps the lanes of a wave within a few rows of each other at any time.  Nothing else
// cares about the order (the synthesis matrix and every per-pixel table are simply built in it); the analysis reads
// grid pixel g from column inv[g].  One workgroup, bitonic sort of (key, index) pairs in LDS; n <= 2048.
constexpr int SORT_N = 2048;
// The rings theta = 0 and theta = pi are one direction each: their n_phi pixels differ only by a rotation of the rotor
// about its own z axis, i.e. by the spin phase e^{-+ i s phi_k} of the value.  Only pixel k = 0 of a pole ring becomes a
// column (n_cols = n_pix - 2 (n_phi - 1)); the analysis re-creates the others from it.  by_key = 0: columns in grid
// order (no boost); by_key = 1: sorted by key.
// The key is the time-skew rate -(v . r) of the pixel's direction: geometry only (rotor of the pixel, no supertranslation
// sums), computed here; `sort_n`: power of two >= n actually sorted.
__global__ __launch_bounds__(1024) void pixel_sort_kernel(PixelSpec P, int n, int n_theta, int n_phi, int by_key, int sort_n,
                                                          int* __restrict__ perm, int* __restrict__ inv) {
  __shared__ double k[SORT_N];
  __shared__ int id[SORT_N];
  const int tid = threadIdx.x;
  auto duplicate = [&](int g) {
    const int j = g / n_phi, kk = g - j * n_phi;
    return (j == 0 || j == n_theta - 1) && kk > 0;
  };
  for (int i = tid; i < sort_n; i += blockDim.x) {
    double key = INFINITY;
    if (i < n && !duplicate(i)) {
      key = (double)i;
      if (by_key) {
        const int j = i / n_phi, kk = i - j * n_phi;
        const Quat R = pixel_rotor(P.frq, P.bs, j, kk, n_theta, n_phi);
        double r[3];
        rotate_z(R, r);
        key = -(P.v[0] * r[0] + P.v[1] * r[1] + P.v[2] * r[2]);
      }
    }
    k[i] = key;
    id[i] = i;
  }
  for (int size = 2; size <= sort_n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < sort_n / 2; t += blockDim.x) {
        const int a = 2 * t - (t & (stride - 1)), b = a + stride;
        const bool up = (a & size) == 0;
        const double ka = k[a], kb = k[b];
        const int ia = id[a], ib = id[b];
        const bool a_gt_b = ka > kb || (ka == kb && ia > ib);
        if (a_gt_b == up) {
          k[a] = kb, k[b] = ka;
          id[a] = ib, id[b] = ia;
        }
      }
    }
  }
  __syncthreads();
  const int n_cols = n - 2 * (n_phi - 1);
  for (int i = tid; i < n_cols; i += blockDim.x) {
    perm[i] = id[i];
    inv[id[i]] = i;
  }
  __syncthreads();  // inv of the pole representatives is visible to the whole (single) workgroup
  for (int kk = 1 + tid; kk < n_phi; kk += blockDim.x) {
    inv[kk] = inv[0];
    inv[(n_theta - 1) * n_phi + kk] = inv[(n_theta - 1) * n_phi];
  }
}

int pixel_sort_max() { return SORT_N; }

hipError_t launch_pixel_sort(hipStream_t stream, const PixelSpec& P, int n_theta, int n_phi, int by_key, int* perm, int* inv) {
  const int n = n_theta * n_phi;
  if (n > SORT_N || n_theta < 3) return hipErrorInvalidValue;
  int sort_n = 2;
  while (sort_n < n) sort_n <<= 1;
  hipLaunchKernelGGL(pixel_sort_kernel, dim3(1), dim3(1024), 0, stream, P, n, n_theta, n_phi, by_key, sort_n, perm, inv);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- pointwise

__global__ __launch_bounds__(256) void psi_mix_kernel(double* __restrict__ Y, const double* __restrict__ Yaux,
                                                      long long ld, int n_pix, long long n_rows,
                                                      const double* __restrict__ x, const double* __restrict__ alpha,
                                                      const double* __restrict__ xa, const double* __restrict__ xb,
                                                      double coeff, int power) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pix) return;
  const double al = alpha[p];
  const cplx a = {xa[2 * p], xa[2 * p + 1]}, b = {xb[2 * p], xb[2 * p + 1]};
  for (long long r = blockIdx.y; r < n_rows; r += gridDim.y) {
    const double dt = x[r] - al;
    cplx X = {dt * a.re - b.re, dt * a.im - b.im};
    cplx Xn = X;
    for (int i = 1; i < power; ++i) Xn = cmul(Xn, X);
    const double2 f = *reinterpret_cast<const double2*>(Yaux + r * ld + 2LL * p);
    cplx v = cmul(cplx{f.x, f.y}, Xn);
    double2* y = reinterpret_cast<double2*>(Y + r * ld + 2LL * p);
    double2 cur = *y;
    cur.x += coeff * v.re;
    cur.y += coeff * v.im;
    *y = cur;
  }
}

hipError_t launch_psi_mix(hipStream_t stream, double* Y, const double* Yaux, long long ld, int n_pix, long long n_rows,
                          const double* x, const double* alpha, const double* xa, const double* xb, double coeff,
                          int power) {
  if (n_rows <= 0 || n_pix <= 0) return hipSuccess;
  dim3 grid((n_pix + 255) / 256, (unsigned)(n_rows < 4096 ? n_rows : 4096));
  hipLaunchKernelGGL(psi_mix_kernel, grid, dim3(256), 0, stream, Y, Yaux, ld, n_pix, n_rows, x, alpha, xa, xb, coeff,
                     power);
  return hipGetLastError();
}

// Horner mixing of the six ABD fields on the distorted grid, in place
// (scri/asymptotic_bondi_data/transformations.py:340-385; Moreschi-Boyle signs):
//   X     = (eth k / k)_p (u_t - alpha_p) - (eth alpha)_p
//   psi0' = ((((psi4 X - 4 psi3) X + 6 psi2) X - 4 psi1) X + psi0) / k^3
//   psi1' = (((-psi4 X + 3 psi3) X - 3 psi2) X + psi1) / k^3
//   psi2' = ((psi4 X - 2 psi3) X + psi2) / k^3,   psi3' = (-psi4 X + psi3) / k^3,   psi4' = psi4 / k^3
//   sigma' = (sigma - eth^2 alpha) / k
__global__ __launch_bounds__(256) void abd_mix_kernel(AbdGrids g, long long ld, int n_pix, long long n_rows,
                                                      const double* __restrict__ u, const double* __restrict__ alpha,
                                                      const double* __restrict__ ethk_over_k,
                                                      const double* __restrict__ eth_alpha,
                                                      const double* __restrict__ etheth_alpha,
                                                      const double* __restrict__ inv_k, const double* __restrict__ inv_k3) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pix) return;
  const double al = alpha[p], ik = inv_k[p], ik3 = inv_k3[p];
  const cplx A = {ethk_over_k[2 * p], ethk_over_k[2 * p + 1]}, B = {eth_alpha[2 * p], eth_alpha[2 * p + 1]};
  const cplx EE = {etheth_alpha[2 * p], etheth_alpha[2 * p + 1]};
  for (long long r = blockIdx.y; r < n_rows; r += gridDim.y) {
    const double dt = u[r] - al;
    const cplx X = {A.re * dt - B.re, A.im * dt - B.im};
    const long long o = r * ld + 2LL * p;
    cplx f[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double2 v = *reinterpret_cast<const double2*>(g.y[i] + o);
      f[i] = {v.x, v.y};
    }
    auto axpy = [](cplx t, cplx X, double c, cplx f) {  // t*X + c*f
      cplx r = cmul(t, X);
      return cplx{r.re + c * f.re, r.im + c * f.im};
    };
    cplx t0 = f[4];
    t0 = axpy(t0, X, -4.0, f[3]);
    t0 = axpy(t0, X, 6.0, f[2]);
    t0 = axpy(t0, X, -4.0, f[1]);
    t0 = axpy(t0, X, 1.0, f[0]);
    cplx t1 = {-f[4].re, -f[4].im};
    t1 = axpy(t1, X, 3.0, f[3]);
    t1 = axpy(t1, X, -3.0, f[2]);
    t1 = axpy(t1, X, 1.0, f[1]);
    cplx t2 = f[4];
    t2 = axpy(t2, X, -2.0, f[3]);
    t2 = axpy(t2, X, 1.0, f[2]);
    cplx t3 = {-f[4].re, -f[4].im};
    t3 = axpy(t3, X, 1.0, f[3]);
    const cplx out[6] = {{t0.re * ik3, t0.im * ik3}, {t1.re * ik3, t1.im * ik3}, {t2.re * ik3, t2.im * ik3},
                         {t3.re * ik3, t3.im * ik3}, {f[4].re * ik3, f[4].im * ik3},
                         {(f[5].re - EE.re) * ik, (f[5].im - EE.im) * ik}};
#pragma unroll
    for (int i = 0; i < 6; ++i) *reinterpret_cast<double2*>(g.y[i] + o) = double2{out[i].re, out[i].im};
  }
}

hipError_t launch_abd_mix(hipStream_t stream, const AbdGrids& g, long long ld, int n_pix, long long n_rows, const double* u,
                          const double* alpha, const double* ethk_over_k, const double* eth_alpha,
                          const double* etheth_alpha, const double* inv_k, const double* inv_k3) {
  if (n_rows <= 0 || n_pix <= 0) return hipSuccess;
  dim3 grid((n_pix + 255) / 256, (unsigned)(n_rows < 2048 ? n_rows : 2048));
  hipLaunchKernelGGL(abd_mix_kernel, grid, dim3(256), 0, stream, g, ld, n_pix, n_rows, u, alpha, ethk_over_k, eth_alpha,
                     etheth_alpha, inv_k, inv_k3);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void affine_cols_kernel(double* __restrict__ Y, long long ld, int n_cols,
                                                          long long n_rows, const double* __restrict__ off,
                                                          const double* __restrict__ scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cols) return;
  const double o = off[c], s = scale[c];
  for (long long r = blockIdx.y; r < n_rows; r += gridDim.y) Y[r * ld + c] = (Y[r * ld + c] - o) * s;
}

hipError_t launch_affine_cols(hipStream_t stream, double* Y, long long ld, int n_cols, long long n_rows,
                              const double* off, const double* scale) {
  if (n_rows <= 0 || n_cols <= 0) return hipSuccess;
  dim3 grid((n_cols + 255) / 256, (unsigned)(n_rows < 4096 ? n_rows : 4096));
  hipLaunchKernelGGL(affine_cols_kernel, grid, dim3(256), 0, stream, Y, ld, n_cols, n_rows, off, scale);
  return hipGetLastError();
}

}  // namespace bms
