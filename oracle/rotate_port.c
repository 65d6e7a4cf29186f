/* CPU restatement (plain C, scalar, one thread) of the reference's two numba kernels
 *   _rotate_decomposition_basis_by_constant   scri/rotations.py:346-367
 *   _rotate_decomposition_basis_by_series     scri/rotations.py:370-392
 * including the per-time-step Wigner-D generation the reference gets from the un-vendored
 * spherical_functions (sf._Wigner_D_matrices, called at scri/rotations.py:327,381).  The D generation
 * follows the published algorithm of that package ("Wigner D matrices", spherical_functions docs): polar
 * decomposition of (Ra, Rb), the |Ra| ~ 0 / |Rb| ~ 0 branches, and otherwise the explicit sum over rho
 * evaluated in Horner form in the ratio -(rb/ra)^2 or -(ra/rb)^2 (whichever is smaller in magnitude).
 *
 * TEST INFRASTRUCTURE ONLY: the timing baseline ("B1, numba-faithful") and a second
 * independent check of the HIP rotation kernel.  Parity with the reference's bit pattern is unpinned
 * (spherical_functions is not available here); this port is validated against oracle/wigner.py.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static double binom(int n, int k) {
  if (k < 0 || k > n) return 0.0;
  if (k > n - k) k = n - k;
  double c = 1.0;
  for (int i = 1; i <= k; ++i) c = c * (double)(n - k + i) / (double)i;
  return c;
}

static int64_t matrix_offset(int ell, int ell_min) {
  return ((4LL * ell * ell * ell - ell) - (4LL * ell_min * ell_min * ell_min - ell_min)) / 3;
}

/* sqrt[(l+m)!(l-m)!/((l+mp)!(l-mp)!)] C(l+mp, rho) C(l-mp, l-rho-m) at rho = rho_min, times sign */
static double coeff0(int ell, int mp, int m, int rho) {
  /* ratio of factorials via lgamma to stay in range */
  double lg = 0.5 * (lgamma(ell + m + 1.0) + lgamma(ell - m + 1.0) - lgamma(ell + mp + 1.0) - lgamma(ell - mp + 1.0));
  return exp(lg) * binom(ell + mp, rho) * binom(ell - mp, ell - rho - m);
}

void wigner_D_matrices(double complex Ra, double complex Rb, int ell_min, int ell_max, double complex* D) {
  const double eps = 1e-15;
  double ra = cabs(Ra), rb = cabs(Rb);
  double phia = carg(Ra), phib = carg(Rb);
  for (int ell = ell_min; ell <= ell_max; ++ell) {
    double complex* Dl = D + matrix_offset(ell, ell_min);
    int n = 2 * ell + 1;
    for (int mp = -ell; mp <= ell; ++mp)
      for (int m = -ell; m <= ell; ++m) {
        double complex val;
        if (ra <= eps) { /* anti-diagonal */
          val = (mp == -m) ? (((ell + mp) & 1) ? -1.0 : 1.0) * cpow(Rb, -2 * mp) : 0.0;
          if (mp == -m) val = (((ell - m) & 1) ? -1.0 : 1.0) * cexp(I * phib * (2.0 * m)) * pow(rb, 2.0 * ell);
        } else if (rb <= eps) { /* diagonal */
          val = (mp == m) ? cexp(I * phia * (2.0 * m)) * pow(ra, 2.0 * ell) : 0.0;
        } else {
          int rho_min = (mp - m) > 0 ? (mp - m) : 0;
          int rho_max = (ell + mp) < (ell - m) ? (ell + mp) : (ell - m);
          if (rho_max < rho_min) {
            val = 0.0;
          } else if (ra >= rb) {
            /* sum_rho (-1)^rho C C ra^(2l+mp-m-2rho) rb^(2rho-mp+m): Horner in x = -(rb/ra)^2 */
            double x = -(rb * rb) / (ra * ra);
            double sum = 1.0;
            for (int rho = rho_max; rho > rho_min; --rho) {
              /* term(rho)/term(rho-1) = x (l+mp-rho+1)(l-m-rho+1) / (rho (rho-mp+m)) */
              sum = 1.0 + sum * x * ((double)(ell + mp - rho + 1) * (ell - m - rho + 1)) / ((double)rho * (rho - mp + m));
            }
            double pref = coeff0(ell, mp, m, rho_min) * pow(ra, 2 * ell + mp - m - 2 * rho_min) * pow(rb, 2 * rho_min - mp + m);
            if (rho_min & 1) pref = -pref;
            val = pref * sum * cexp(I * (phia * (mp + m) + phib * (m - mp)));
          } else {
            /* same sum ordered from rho_max downwards: Horner in x = -(ra/rb)^2 */
            double x = -(ra * ra) / (rb * rb);
            double sum = 1.0;
            for (int rho = rho_min; rho < rho_max; ++rho) {
              /* term(rho)/term(rho+1) = x (rho+1)(rho+1-mp+m) / ((l+mp-rho)(l-m-rho)) */
              sum = 1.0 + sum * x * ((double)(rho + 1) * (rho + 1 - mp + m)) / ((double)(ell + mp - rho) * (ell - m - rho));
            }
            double pref = coeff0(ell, mp, m, rho_max) * pow(ra, 2 * ell + mp - m - 2 * rho_max) * pow(rb, 2 * rho_max - mp + m);
            if (rho_max & 1) pref = -pref;
            val = pref * sum * cexp(I * (phia * (mp + m) + phib * (m - mp)));
          }
        }
        Dl[(mp + ell) * n + (m + ell)] = val;
      }
  }
}

/* scri/rotations.py:346-367 */
void rotate_by_constant(double complex* data, int64_t n_times, int64_t n_modes, int ell_min, int ell_max,
                        const double complex* D, double complex* tmp) {
  for (int64_t it = 0; it < n_times; ++it)
    for (int ell = ell_min; ell <= ell_max; ++ell) {
      int64_t i_data = (int64_t)ell * ell - (int64_t)ell_min * ell_min;
      int64_t i_D = matrix_offset(ell, ell_min);
      int n = 2 * ell + 1;
      for (int im = 0; im < n; ++im) tmp[im] = 0.0;
      for (int imp = 0; imp < n; ++imp)
        for (int im = 0; im < n; ++im) tmp[im] += data[it * n_modes + i_data + imp] * D[i_D + (int64_t)n * imp + im];
      for (int im = 0; im < n; ++im) data[it * n_modes + i_data + im] = tmp[im];
    }
}

/* scri/rotations.py:370-392; R_basis[n_times][2] = (Ra, Rb); D: work space of total_size_D_matrices */
void rotate_by_series(double complex* data, const double complex* R_basis, int64_t n_times, int64_t n_modes,
                      int ell_min, int ell_max, double complex* D) {
  for (int64_t it = 0; it < n_times; ++it) {
    wigner_D_matrices(R_basis[2 * it], R_basis[2 * it + 1], ell_min, ell_max, D);
    for (int ell = ell_min; ell <= ell_max; ++ell) {
      int64_t i_data = (int64_t)ell * ell - (int64_t)ell_min * ell_min;
      int64_t i_D = matrix_offset(ell, ell_min);
      int n = 2 * ell + 1;
      for (int im = 0; im < n; ++im) {
        double complex s = 0.0;
        for (int imp = 0; imp < n; ++imp) s += data[it * n_modes + i_data + imp] * D[i_D + im + (int64_t)n * imp];
        D[i_D + im] = s;
      }
      for (int im = 0; im < n; ++im) data[it * n_modes + i_data + im] = D[i_D + im];
    }
  }
}

/* the same kernel with the time loop shared among threads (what numba's prange over time would do): every thread has its
 * own D work space; D_all holds n_threads * d_size entries.  Returns the number of threads used. */
#ifdef _OPENMP
#include <omp.h>
#endif
int rotate_by_series_omp(double complex* data, const double complex* R_basis, int64_t n_times, int64_t n_modes, int ell_min,
                         int ell_max, double complex* D_all, int64_t d_size, int n_threads) {
  int used = 1;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
    double complex* D = D_all + (int64_t)omp_get_thread_num() * d_size;
#pragma omp for schedule(static)
    for (int64_t it = 0; it < n_times; ++it) rotate_by_series(data + it * n_modes, R_basis + 2 * it, 1, n_modes, ell_min, ell_max, D);
  }
#else
  (void)d_size;
  (void)n_threads;
  rotate_by_series(data, R_basis, n_times, n_modes, ell_min, ell_max, D_all);
#endif
  return used;
}

/* pointer-argument wrapper for ctypes */
void wigner_D_matrices_p(const double* RaRb, int ell_min, int ell_max, double complex* D) {
  wigner_D_matrices(RaRb[0] + I * RaRb[1], RaRb[2] + I * RaRb[3], ell_min, ell_max, D);
}
