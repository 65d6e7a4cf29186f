// Time-series calculus on mode data and products on the sphere: the kernels behind ModesTimeSeries
// (scri/modes_time_series.py:72-126 interpolate / derivative / antiderivative, :142-202 grid_multiply) and
// WaveformBase / AsymptoticBondiData.interpolate.
//
// scipy's CubicSpline(u, y).derivative(k) / .antiderivative(k) are evaluated from knot data instead of per-interval
// coefficient tables: the spline slopes s_j at the knots (shared-matrix solve of kernels_spline.hip: forward pass,
// then spline_slopes_kernel) and, for the antiderivatives, running integrals P1_j = int_{u_0}^{u_j} f and
// P2_j = int_{u_0}^{u_j} P1 (spline_prefix_kernel).  Any sample is then a local Hermite evaluation
// (spline_hermite_eval_kernel), lanes across columns (16-byte coalesced), one output time per block row.
#include <cstdlib>
#include "kernels.h"

namespace bms {

// ------------------------------------------------------------------------------------------------ slopes at the knots
// Thread (column, tile): s_j = r'_j - C_j s_{j+1}, started `halo` knots above the tile (0.268^halo decay), written for
// the knots of the tile.  R and S must be different buffers (a tile's start-up reads r' of its neighbour).
__global__ __launch_bounds__(64) void spline_slopes_kernel(const double* __restrict__ R, double* __restrict__ S, long long ld,
                                                           int n_cols, long long n, const SplineTable* __restrict__ table,
                                                           int tile, int halo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_cols) return;
  const long long jA = (long long)blockIdx.y * tile;
  long long jB = jA + tile;
  if (jB > n) jB = n;
  long long jE = jB - 1 + halo;
  if (jE > n - 1) jE = n - 1;
  const double* rp = R + 2LL * p;
  double* sp = S + 2LL * p;
  double2 s = *reinterpret_cast<const double2*>(rp + jE * ld);  // exact at the last knot, truncated start otherwise
  if (jE < jB) *reinterpret_cast<double2*>(sp + jE * ld) = s;
  for (long long j = jE - 1; j >= jA; --j) {
    const double C = table[j].C;
    const double2 r = *reinterpret_cast<const double2*>(rp + j * ld);
    s.x = r.x - C * s.x;
    s.y = r.y - C * s.y;
    if (j < jB) *reinterpret_cast<double2*>(sp + j * ld) = s;
  }
}

hipError_t launch_spline_slopes(hipStream_t stream, const double* R, double* S, long long ld, int n_cols, long long n,
                                const SplineTable* table, int tile, int halo) {
  if (n <= 0 || n_cols <= 0) return hipSuccess;
  dim3 grid((n_cols + 63) / 64, (unsigned)((n + tile - 1) / tile));
  hipLaunchKernelGGL(spline_slopes_kernel, grid, dim3(64), 0, stream, R, S, ld, n_cols, n, table, tile, halo);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ running integrals
// Interval j of the spline in Hermite form: f(t) = y0 + s0 t + c2 t^2 + c3 t^3, t = u - u_j, h = u_{j+1} - u_j,
//   c3 = (s0 + s1 - 2 D) / h^2,  c2 = (D - s0) / h - c3 h,  D = (y1 - y0) / h.
struct Hermite {
  double c2x, c2y, c3x, c3y;
};
__device__ __forceinline__ Hermite hermite(double2 y0, double2 y1, double2 s0, double2 s1, double h) {
  const double ih = 1.0 / h;
  const double dx = (y1.x - y0.x) * ih, dy = (y1.y - y0.y) * ih;
  const double tx = (s0.x + s1.x - 2.0 * dx) * ih, ty = (s0.y + s1.y - 2.0 * dy) * ih;
  Hermite H;
  H.c3x = tx * ih, H.c3y = ty * ih;
  H.c2x = (dx - s0.x) * ih - tx, H.c2y = (dy - s0.y) * ih - ty;
  return H;
}

// P1[j], P2[j] for all knots (P[0] = 0: scipy's antiderivative vanishes at the first knot).  The sum over intervals is
// sequential per column; columns are independent.  Phase 1: per (column, tile) totals; phase 2: running sum over tiles;
// phase 3: per (column, tile) prefix from the tile's start value.  P2 needs P1 at the tile starts, so P1's three phases
// run before P2's (same kernel, `second` selects the integrand).
__global__ __launch_bounds__(64) void spline_prefix_kernel(const double* __restrict__ Y, const double* __restrict__ S,
                                                           long long ld, int n_cols, long long n, const double* __restrict__ x,
                                                           int tile, double* __restrict__ P1, double* __restrict__ P2,
                                                           double* __restrict__ carry /* [n_tiles][2 n_cols] */, int phase,
                                                           int second) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_cols) return;
  const long long n_tiles = (n + tile - 1) / tile;
  if (phase == 2) {
    // exclusive running sum of the tile totals, one thread per column
    if (blockIdx.y != 0) return;
    double2 run = {0.0, 0.0};
    for (long long t = 0; t < n_tiles; ++t) {
      double2* cp = reinterpret_cast<double2*>(carry + t * 2LL * n_cols + 2LL * p);
      const double2 v = *cp;
      *cp = run;
      run.x += v.x, run.y += v.y;
    }
    return;
  }
  const long long jA = (long long)blockIdx.y * tile;
  long long jB = jA + tile;
  if (jB > n) jB = n;
  if (jA >= n) return;
  const double* yp = Y + 2LL * p;
  const double* sp = S + 2LL * p;
  double* out = (second ? P2 : P1) + 2LL * p;
  double2* cp = reinterpret_cast<double2*>(carry + (long long)blockIdx.y * 2LL * n_cols + 2LL * p);
  double2 run = phase == 3 ? *cp : double2{0.0, 0.0};
  double2 y0 = *reinterpret_cast<const double2*>(yp + jA * ld), s0 = *reinterpret_cast<const double2*>(sp + jA * ld);
  for (long long j = jA; j < jB; ++j) {
    if (phase == 3) *reinterpret_cast<double2*>(out + j * ld) = run;
    if (j + 1 >= n) break;
    const double2 y1 = *reinterpret_cast<const double2*>(yp + (j + 1) * ld);
    const double2 s1 = *reinterpret_cast<const double2*>(sp + (j + 1) * ld);
    const double h = x[j + 1] - x[j];
    const Hermite H = hermite(y0, y1, s0, s1, h);
    if (!second) {
      // int_0^h f = h (y0 + h (s0/2 + h (c2/3 + h c3/4)))
      run.x += h * (y0.x + h * (0.5 * s0.x + h * (H.c2x * (1.0 / 3.0) + h * H.c3x * 0.25)));
      run.y += h * (y0.y + h * (0.5 * s0.y + h * (H.c2y * (1.0 / 3.0) + h * H.c3y * 0.25)));
    } else {
      // int_0^h (P1_j + int_0^t f) = h (P1_j + h (y0/2 + h (s0/6 + h (c2/12 + h c3/20))))
      const double2 q = *reinterpret_cast<const double2*>(P1 + 2LL * p + j * ld);
      run.x += h * (q.x + h * (0.5 * y0.x + h * (s0.x * (1.0 / 6.0) + h * (H.c2x * (1.0 / 12.0) + h * H.c3x * 0.05))));
      run.y += h * (q.y + h * (0.5 * y0.y + h * (s0.y * (1.0 / 6.0) + h * (H.c2y * (1.0 / 12.0) + h * H.c3y * 0.05))));
    }
    y0 = y1, s0 = s1;
  }
  if (phase == 1) *cp = run;
}

hipError_t launch_spline_prefix(hipStream_t stream, const double* Y, const double* S, long long ld, int n_cols, long long n,
                                const double* x, double* P1, double* P2, double* carry, int order) {
  if (n <= 0 || n_cols <= 0) return hipSuccess;
  const int tile = 512;
  const long long n_tiles = (n + tile - 1) / tile;
  if (n_tiles > GRID_Y_MAX) return hipErrorInvalidValue;
  dim3 grid((n_cols + 63) / 64, (unsigned)n_tiles), one((n_cols + 63) / 64, 1);
  for (int second = 0; second < order; ++second)
    for (int phase = 1; phase <= 3; ++phase) {
      hipLaunchKernelGGL(spline_prefix_kernel, phase == 2 ? one : grid, dim3(64), 0, stream, Y, S, ld, n_cols, n, x, tile, P1, P2,
                         carry, phase, second);
    }
  return hipGetLastError();
}
long long spline_prefix_carry_size(long long n, int n_cols) { return ((n + 511) / 512) * 2LL * n_cols; }

// ---- antiderivatives of any order k >= 3 (scipy's PPoly.antiderivative(k): k integrations, each vanishing at the first knot).
// Level r = 1..k keeps its knot values A_r(x_j) in Pall + (r - 1) level_stride; with the interval's cubic c_0..c_3 (c_0 = y_j,
// c_1 = s_j) and h = x_{j+1} - x_j the Taylor shift is
//     A_r(x_j + t) = sum_{q=0}^{r-1} A_{r-q}(x_j) t^q / q!  +  t^r sum_{n=0}^{3} c_n n! / (n + r)! t^n ,
// so level r needs the levels below it at the same knot: the levels run one after the other, each as the three phases of
// spline_prefix_kernel (tile totals, running sum over tiles, prefix within the tile).
__device__ __forceinline__ double2 antiderivative_shift(const double* __restrict__ lower /* level 1 at knot j, column p */, long long level_stride,
                                                        int r, bool with_own, double2 y0, double2 s0, const Hermite& H, double t) {
  // sum_{q = (with_own ? 0 : 1)}^{r-1} A_{r-q}(x_j) t^q / q!   (A_r itself is the caller's running value when !with_own)
  double2 acc{0.0, 0.0};
  double tq = with_own ? 1.0 : t;  // t^q / q!
  for (int q = with_own ? 0 : 1; q < r; ++q) {
    const double2 a = *reinterpret_cast<const double2*>(lower + (long long)(r - q - 1) * level_stride);
    acc.x = fma(a.x, tq, acc.x);
    acc.y = fma(a.y, tq, acc.y);
    tq *= t / (double)(q + 1);
  }
  // t^r / r! (c_0 + t (c_1 / (r+1) + t (2 c_2 / ((r+1)(r+2)) + t 6 c_3 / ((r+1)(r+2)(r+3)))))
  double tr = 1.0;
  for (int q = 1; q <= r; ++q) tr *= t / (double)q;
  const double f1 = 1.0 / (r + 1.0), f2 = 2.0 * f1 / (r + 2.0), f3 = 3.0 * f2 / (r + 3.0);
  acc.x = fma(tr, y0.x + t * (s0.x * f1 + t * (H.c2x * f2 + t * H.c3x * f3)), acc.x);
  acc.y = fma(tr, y0.y + t * (s0.y * f1 + t * (H.c2y * f2 + t * H.c3y * f3)), acc.y);
  return acc;
}

__global__ __launch_bounds__(64) void spline_prefix_level_kernel(const double* __restrict__ Y, const double* __restrict__ S, long long ld,
                                                                 int n_cols, long long n, const double* __restrict__ x, int tile,
                                                                 double* __restrict__ Pall, long long level_stride,
                                                                 double* __restrict__ carry, int phase, int r) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_cols) return;
  const long long n_tiles = (n + tile - 1) / tile;
  if (phase == 2) {
    if (blockIdx.y != 0) return;
    double2 run = {0.0, 0.0};
    for (long long t = 0; t < n_tiles; ++t) {
      double2* cp = reinterpret_cast<double2*>(carry + t * 2LL * n_cols + 2LL * p);
      const double2 v = *cp;
      *cp = run;
      run.x += v.x, run.y += v.y;
    }
    return;
  }
  const long long jA = (long long)blockIdx.y * tile;
  long long jB = jA + tile;
  if (jB > n) jB = n;
  if (jA >= n) return;
  const double* yp = Y + 2LL * p;
  const double* sp = S + 2LL * p;
  double* out = Pall + (long long)(r - 1) * level_stride + 2LL * p;
  double2* cp = reinterpret_cast<double2*>(carry + (long long)blockIdx.y * 2LL * n_cols + 2LL * p);
  double2 run = phase == 3 ? *cp : double2{0.0, 0.0};
  double2 y0 = *reinterpret_cast<const double2*>(yp + jA * ld), s0 = *reinterpret_cast<const double2*>(sp + jA * ld);
  for (long long j = jA; j < jB; ++j) {
    if (phase == 3) *reinterpret_cast<double2*>(out + j * ld) = run;
    if (j + 1 >= n) break;
    const double2 y1 = *reinterpret_cast<const double2*>(yp + (j + 1) * ld);
    const double2 s1 = *reinterpret_cast<const double2*>(sp + (j + 1) * ld);
    const double h = x[j + 1] - x[j];
    const Hermite H = hermite(y0, y1, s0, s1, h);
    const double2 inc = antiderivative_shift(Pall + 2LL * p + j * ld, level_stride, r, false, y0, s0, H, h);
    run.x += inc.x, run.y += inc.y;
    y0 = y1, s0 = s1;
  }
  if (phase == 1) *cp = run;
}

hipError_t launch_spline_prefix_levels(hipStream_t stream, const double* Y, const double* S, long long ld, int n_cols, long long n,
                                       const double* x, double* Pall, long long level_stride, double* carry, int levels) {
  if (n <= 0 || n_cols <= 0) return hipSuccess;
  const int tile = 512;
  const long long n_tiles = (n + tile - 1) / tile;
  if (n_tiles > GRID_Y_MAX) return hipErrorInvalidValue;
  dim3 grid((n_cols + 63) / 64, (unsigned)n_tiles), one((n_cols + 63) / 64, 1);
  for (int r = 1; r <= levels; ++r)
    for (int phase = 1; phase <= 3; ++phase)
      hipLaunchKernelGGL(spline_prefix_level_kernel, phase == 2 ? one : grid, dim3(64), 0, stream, Y, S, ld, n_cols, n, x, tile, Pall,
                         level_stride, carry, phase, r);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void spline_antiderivative_eval_kernel(const double* __restrict__ Y, const double* __restrict__ S,
                                                                         const double* __restrict__ Pall, long long level_stride,
                                                                         long long ld, int n_cols, long long n, const double* __restrict__ x,
                                                                         const double* __restrict__ x_new, long long n_new, int k,
                                                                         double* __restrict__ out, long long ldo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = blockIdx.y; i < n_new; i += gridDim.y) {
    const double u = x_new[i];
    long long lo = 0, hi = n - 1;
    while (hi - lo > 1) {
      const long long mid = (lo + hi) >> 1;
      if (x[mid] <= u)
        lo = mid;
      else
        hi = mid;
    }
    const long long j = lo;
    if (p >= n_cols) continue;
    const double xj = x[j], h = x[j + 1] - xj, t = u - xj;
    const double2 y0 = *reinterpret_cast<const double2*>(Y + 2LL * p + j * ld);
    const double2 y1 = *reinterpret_cast<const double2*>(Y + 2LL * p + (j + 1) * ld);
    const double2 s0 = *reinterpret_cast<const double2*>(S + 2LL * p + j * ld);
    const double2 s1 = *reinterpret_cast<const double2*>(S + 2LL * p + (j + 1) * ld);
    const Hermite H = hermite(y0, y1, s0, s1, h);
    *reinterpret_cast<double2*>(out + 2LL * p + i * ldo) = antiderivative_shift(Pall + 2LL * p + j * ld, level_stride, k, true, y0, s0, H, t);
  }
}

hipError_t launch_spline_antiderivative_eval(hipStream_t stream, const double* Y, const double* S, const double* Pall, long long level_stride,
                                             long long ld, int n_cols, long long n, const double* x, const double* x_new, long long n_new,
                                             int k, double* out, long long ldo) {
  if (n_new <= 0 || n_cols <= 0) return hipSuccess;
  const int threads = n_cols >= 256 ? 256 : ((n_cols + 63) / 64) * 64;
  dim3 grid((n_cols + threads - 1) / threads, (unsigned)(n_new < 32768 ? n_new : 32768));
  hipLaunchKernelGGL(spline_antiderivative_eval_kernel, grid, dim3(threads), 0, stream, Y, S, Pall, level_stride, ld, n_cols, n, x, x_new,
                     n_new, k, out, ldo);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ evaluation
// out[i][c] = d^k/du^k spline_c (x_new[i]), k = order in [-2, 3]; interval found by binary search (the same for every
// column of a block row), end intervals extrapolate (scipy's default).  Lanes across columns.
__global__ __launch_bounds__(256) void spline_hermite_eval_kernel(const double* __restrict__ Y, const double* __restrict__ S,
                                                                  const double* __restrict__ P1, const double* __restrict__ P2,
                                                                  long long ld, int n_cols, long long n,
                                                                  const double* __restrict__ x, const double* __restrict__ x_new,
                                                                  long long n_new, int order, double* __restrict__ out,
                                                                  long long ldo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = blockIdx.y; i < n_new; i += gridDim.y) {
    const double u = x_new[i];
    // last knot <= u, clamped to [0, n-2]
    long long lo = 0, hi = n - 1;
    while (hi - lo > 1) {
      const long long mid = (lo + hi) >> 1;
      if (x[mid] <= u)
        lo = mid;
      else
        hi = mid;
    }
    const long long j = lo;
    if (p >= n_cols) continue;
    const double xj = x[j], h = x[j + 1] - xj, t = u - xj;
    const double2 y0 = *reinterpret_cast<const double2*>(Y + 2LL * p + j * ld);
    const double2 y1 = *reinterpret_cast<const double2*>(Y + 2LL * p + (j + 1) * ld);
    const double2 s0 = *reinterpret_cast<const double2*>(S + 2LL * p + j * ld);
    const double2 s1 = *reinterpret_cast<const double2*>(S + 2LL * p + (j + 1) * ld);
    const Hermite H = hermite(y0, y1, s0, s1, h);
    double2 v;
    switch (order) {
      case 0:
        v.x = y0.x + t * (s0.x + t * (H.c2x + t * H.c3x));
        v.y = y0.y + t * (s0.y + t * (H.c2y + t * H.c3y));
        break;
      case 1:
        v.x = s0.x + t * (2.0 * H.c2x + 3.0 * t * H.c3x);
        v.y = s0.y + t * (2.0 * H.c2y + 3.0 * t * H.c3y);
        break;
      case 2:
        v.x = 2.0 * H.c2x + 6.0 * t * H.c3x;
        v.y = 2.0 * H.c2y + 6.0 * t * H.c3y;
        break;
      case 3:
        v.x = 6.0 * H.c3x;
        v.y = 6.0 * H.c3y;
        break;
      case -1: {
        const double2 q1 = *reinterpret_cast<const double2*>(P1 + 2LL * p + j * ld);
        v.x = q1.x + t * (y0.x + t * (0.5 * s0.x + t * (H.c2x * (1.0 / 3.0) + t * H.c3x * 0.25)));
        v.y = q1.y + t * (y0.y + t * (0.5 * s0.y + t * (H.c2y * (1.0 / 3.0) + t * H.c3y * 0.25)));
        break;
      }
      default: {
        const double2 q1 = *reinterpret_cast<const double2*>(P1 + 2LL * p + j * ld);
        const double2 q2 = *reinterpret_cast<const double2*>(P2 + 2LL * p + j * ld);
        v.x = q2.x + t * (q1.x + t * (0.5 * y0.x + t * (s0.x * (1.0 / 6.0) + t * (H.c2x * (1.0 / 12.0) + t * H.c3x * 0.05))));
        v.y = q2.y + t * (q1.y + t * (0.5 * y0.y + t * (s0.y * (1.0 / 6.0) + t * (H.c2y * (1.0 / 12.0) + t * H.c3y * 0.05))));
      }
    }
    *reinterpret_cast<double2*>(out + 2LL * p + i * ldo) = v;
  }
}

hipError_t launch_spline_hermite_eval(hipStream_t stream, const double* Y, const double* S, const double* P1, const double* P2,
                                      long long ld, int n_cols, long long n, const double* x, const double* x_new,
                                      long long n_new, int order, double* out, long long ldo) {
  if (n_new <= 0 || n_cols <= 0) return hipSuccess;
  const int threads = n_cols >= 256 ? 256 : ((n_cols + 63) / 64) * 64;
  dim3 grid((n_cols + threads - 1) / threads, (unsigned)(n_new < 32768 ? n_new : 32768));
  hipLaunchKernelGGL(spline_hermite_eval_kernel, grid, dim3(threads), 0, stream, Y, S, P1, P2, ld, n_cols, n, x, x_new, n_new,
                     order, out, ldo);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ pointwise product
// c[i] = a[i] * b[i] (complex), the grid-space product of ModesTimeSeries.grid_multiply (modes_time_series.py:181)
__global__ __launch_bounds__(256) void cmul_kernel(const double2* __restrict__ a, const double2* __restrict__ b,
                                                   double2* __restrict__ c, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double2 u = a[i], v = b[i];
    c[i] = double2{u.x * v.x - u.y * v.y, u.x * v.y + u.y * v.x};
  }
}

hipError_t launch_cmul(hipStream_t stream, const double* a, const double* b, double* c, long long n) {
  if (n <= 0) return hipSuccess;
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(cmul_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream,
                     reinterpret_cast<const double2*>(a), reinterpret_cast<const double2*>(b), reinterpret_cast<double2*>(c), n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ angular velocity
// <Ldt>^a = sum fbar^{l,m'} <l,m'|L_a|l,m> fdot^{l,m}  and  <LL>^{ab} = Re sum fbar^{l,m'} <l,m'|L_a L_b|l,m> f^{l,m}
// (scri/mode_calculations.py:14-57, 209-313), then omega = -<LL>^{-1} <Ldt> (:403-432).  One wave per time step: the
// row of modes and of their time derivatives goes to LDS (the ladder terms couple m with m +- 1, m +- 2), each lane sums
// its modes, the 9 numbers are reduced across the wave and lane 0 solves the 3 x 3 system.
__device__ __forceinline__ double ladder(int l, int m) { return sqrt((double)((l - m) * (l + m + 1))); }  // sf.ladder_operator_coefficient

__global__ __launch_bounds__(256) void angular_velocity_kernel(const double* __restrict__ F, const double* __restrict__ Fdot,
                                                               long long ld, long long n_times, int ell_min, int n_modes,
                                                               double* __restrict__ ldt_out, double* __restrict__ ll_out,
                                                               double* __restrict__ omega_out) {
  extern __shared__ double2 rows[];  // [waves][2][n_modes]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  if (t >= n_times) return;
  double2* f = rows + (size_t)wave * 2 * n_modes;
  double2* fd = f + n_modes;
  for (int i = lane; i < n_modes; i += 64) {
    f[i] = *reinterpret_cast<const double2*>(F + t * ld + 2LL * i);
    fd[i] = *reinterpret_cast<const double2*>(Fdot + t * ld + 2LL * i);
  }
  // (single wave: LDS writes above are visible to its own later reads)
  double lx = 0, ly = 0, lz = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
  auto cmulc = [](double2 a, double2 b) { return double2{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x}; };  // conj(a) b
  for (int i = lane; i < n_modes; i += 64) {
    const int k = i + ell_min * ell_min;
    int l = (int)sqrt((double)k);
    while (l * l > k) --l;
    while ((l + 1) * (l + 1) <= k) ++l;
    const int m = k - l * (l + 1);
    const double2 a = f[i], ad = fd[i];
    const double2 zero = {0.0, 0.0};
    // ---- <Ldt>
    {
      const double2 p = m + 1 <= l ? cmulc(f[i + 1], ad) : zero, q = m - 1 >= -l ? cmulc(f[i - 1], ad) : zero;
      const double cp = m + 1 <= l ? ladder(l, m) : 0.0, cm = m - 1 >= -l ? ladder(l, -m) : 0.0;
      const double2 Lp = {p.x * cp, p.y * cp}, Lm = {q.x * cm, q.y * cm};
      const double2 Lz = cmulc(a, ad);
      lx += 0.5 * (Lp.y + Lm.y);
      ly += -0.5 * (Lp.x - Lm.x);
      lz += Lz.y * m;
    }
    // ---- <LL>
    {
      auto term = [&](bool ok, int j, double c) {
        if (!ok) return zero;
        const double2 v = cmulc(f[j], a);
        return double2{v.x * c, v.y * c};
      };
      const double2 LpLp = term(m + 2 <= l, i + 2, m + 2 <= l ? ladder(l, m + 1) * ladder(l, m) : 0.0);
      const double2 LpLm = term(m - 1 >= -l, i, m - 1 >= -l ? ladder(l, m - 1) * ladder(l, -m) : 0.0);
      const double2 LmLp = term(m + 1 <= l, i, m + 1 <= l ? ladder(l, -(m + 1)) * ladder(l, m) : 0.0);
      const double2 LmLm = term(m - 2 >= -l, i - 2, m - 2 >= -l ? ladder(l, -(m - 1)) * ladder(l, -m) : 0.0);
      const double2 LpLz = term(m + 1 <= l, i + 1, m + 1 <= l ? ladder(l, m) * m : 0.0);
      const double2 LzLp = term(m + 1 <= l, i + 1, m + 1 <= l ? (m + 1) * ladder(l, m) : 0.0);
      const double2 LmLz = term(m - 1 >= -l, i - 1, m - 1 >= -l ? ladder(l, -m) * m : 0.0);
      const double2 LzLm = term(m - 1 >= -l, i - 1, m - 1 >= -l ? (m - 1) * ladder(l, -m) : 0.0);
      const double LzLz = (a.x * a.x + a.y * a.y) * (double)(m * m);
      // real parts of the symmetrised (x, y, z) components; -i z has real part Im z
      const double LxLx = 0.25 * (LpLp.x + LmLm.x + LmLp.x + LpLm.x);
      const double LyLy = -0.25 * (LpLp.x - LmLp.x - LpLm.x + LmLm.x);
      const double LxLy_r = 0.25 * (LpLp.y - LmLm.y + LmLp.y - LpLm.y);   // Re(-i/4 (LpLp - LmLm + LmLp - LpLm))
      const double LyLx_r = 0.25 * (LpLp.y - LmLp.y + LpLm.y - LmLm.y);   // Re(-i/4 (LpLp - LmLp + LpLm - LmLm))
      const double LxLz_r = 0.5 * (LpLz.x + LmLz.x), LzLx_r = 0.5 * (LzLp.x + LzLm.x);
      const double LyLz_r = 0.5 * (LpLz.y - LmLz.y), LzLy_r = 0.5 * (LzLp.y - LzLm.y);
      xx += LxLx;
      yy += LyLy;
      zz += LzLz;
      xy += 0.5 * (LxLy_r + LyLx_r);
      xz += 0.5 * (LxLz_r + LzLx_r);
      yz += 0.5 * (LyLz_r + LzLy_r);
    }
  }
  double v[9] = {lx, ly, lz, xx, xy, xz, yy, yz, zz};
#pragma unroll
  for (int c = 0; c < 9; ++c)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_xor(v[c], off, 64);
  if (lane != 0) return;
  if (ldt_out) ldt_out[3 * t] = v[0], ldt_out[3 * t + 1] = v[1], ldt_out[3 * t + 2] = v[2];
  const double A[3][3] = {{v[3], v[4], v[5]}, {v[4], v[6], v[7]}, {v[5], v[7], v[8]}};
  if (ll_out)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) ll_out[9 * t + 3 * r + c] = A[r][c];
  if (omega_out) {
    // omega = -A^{-1} l by Gaussian elimination with partial pivoting (numpy.linalg.solve = LAPACK gesv)
    double M[3][4] = {{A[0][0], A[0][1], A[0][2], -v[0]}, {A[1][0], A[1][1], A[1][2], -v[1]}, {A[2][0], A[2][1], A[2][2], -v[2]}};
    for (int c = 0; c < 3; ++c) {
      int piv = c;
      for (int r = c + 1; r < 3; ++r)
        if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
      for (int k = 0; k < 4; ++k) {
        const double tmp = M[c][k];
        M[c][k] = M[piv][k];
        M[piv][k] = tmp;
      }
      for (int r = c + 1; r < 3; ++r) {
        const double fct = M[r][c] / M[c][c];
        for (int k = c; k < 4; ++k) M[r][k] -= fct * M[c][k];
      }
    }
    double x[3];
    for (int r = 2; r >= 0; --r) {
      double sum = M[r][3];
      for (int k = r + 1; k < 3; ++k) sum -= M[r][k] * x[k];
      x[r] = sum / M[r][r];
    }
    omega_out[3 * t] = x[0], omega_out[3 * t + 1] = x[1], omega_out[3 * t + 2] = x[2];
  }
}

hipError_t launch_angular_velocity(hipStream_t stream, const double* F, const double* Fdot, long long ld, long long n_times,
                                   int ell_min, int n_modes, double* ldt_out, double* ll_out, double* omega_out) {
  if (n_times <= 0) return hipSuccess;
  const int waves = 4;
  const size_t lds = (size_t)waves * 2 * n_modes * sizeof(double2);
  if (lds > 64 * 1024) {
    hipError_t e = allow_dynamic_lds((const void*)angular_velocity_kernel);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(angular_velocity_kernel, dim3((unsigned)((n_times + waves - 1) / waves)), dim3(64 * waves), lds, stream, F,
                     Fdot, ld, n_times, ell_min, n_modes, ldt_out, ll_out, omega_out);
  return hipGetLastError();
}

}  // namespace bms
