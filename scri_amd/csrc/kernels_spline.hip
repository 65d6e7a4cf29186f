// Not-a-knot cubic-spline interpolation along the time axis, one spline per grid pixel
// (scipy InterpolatedUnivariateSpline at scri/waveform_grid.py:574-588; scipy CubicSpline at
// scri/asymptotic_bondi_data/transformations.py:403-412, scri/waveform_base.py:964).
//
// The reference fits 2 n_pix (WM) or 12 n_pix (ABD) independent splines of length N in a Python loop,
// each on its own abscissa u'_p = k_p (u - alpha_p).  Cubic-spline interpolation is invariant under affine
// maps of the abscissa, so every pixel shares ONE tridiagonal system on the original knots u_j (only the
// right-hand sides differ) and is evaluated at u_eval = u'/k_p + alpha_p.  The Thomas factors of the shared
// matrix are computed once (SplineTable); per column the fit costs 3 FMA per knot forward and 1 backward.
//
// The inverse of the (strictly diagonally dominant, ratio 1/2) spline matrix decays like (2 - sqrt 3)^n
// = 0.268^n, for any knot spacing, so the time axis is cut into tiles whose recurrences start `halo` knots
// outside the tile (0.268^32 = 5e-19: below fp64 rounding of the global solve).  That gives
// n_pix x n_tiles independent threads, lanes across adjacent pixels (16 B coalesced complex accesses),
// marching in time.  Both kernels are HBM-bound streaming passes.
#include <cstdlib>
#include "wigner.h"
#include "kernels.h"

namespace bms {

// ------------------------------------------------------------------------------------------------ table
// Row j of the shared system  a_j s_{j-1} + b_j s_j + c_j s_{j+1} = r_j  (scipy CubicSpline, bc 'not-a-knot'):
//   interior : a = h_j, b = 2 (h_{j-1} + h_j), c = h_{j-1}, r = 3 (h_j/h_{j-1}) D_{j-1} + 3 (h_{j-1}/h_j) D_j
//   j = 0    : b = h_1, c = h_0 + h_1,   r = ((3 h_0 + 2 h_1) h_1/(d h_0)) D_0 + (h_0^2/(d h_1)) D_1,  d = h_0 + h_1
//   j = n-1  : a = h_{n-3} + h_{n-2}, b = h_{n-3},
//              r = (h_{n-2}^2/(d h_{n-3})) D_{n-3} + ((2 d + h_{n-2}) h_{n-3}/(d h_{n-2})) D_{n-2}, d = h_{n-3} + h_{n-2}
// with D_j = y_{j+1} - y_j.  Thomas: m_j = 1/(b_j - a_j C_{j-1}), C_j = c_j m_j, r'_j = (r_j - a_j r'_{j-1}) m_j.
// Table entry: P, Q = the two RHS coefficients times m_j; A = a_j m_j; C = C_j.
struct RowCoef {
  double a, b, c, p, q;
};
__device__ __forceinline__ RowCoef spline_row(const double* __restrict__ x, long long j, long long n) {
  RowCoef r;
  if (j == 0) {
    const double h0 = x[1] - x[0], h1 = x[2] - x[1], d = x[2] - x[0];
    r.a = 0.0;
    r.b = h1;
    r.c = d;
    r.p = (h0 + 2.0 * d) * h1 / (d * h0);
    r.q = h0 * h0 / (d * h1);
  } else if (j == n - 1) {
    const double hm = x[n - 2] - x[n - 3], hl = x[n - 1] - x[n - 2], d = x[n - 1] - x[n - 3];
    r.a = d;
    r.b = hm;
    r.c = 0.0;
    r.p = hl * hl / (d * hm);
    r.q = (2.0 * d + hl) * hm / (d * hl);
  } else {
    const double hm = x[j] - x[j - 1], hp = x[j + 1] - x[j];
    r.a = hp;
    r.b = 2.0 * (hm + hp);
    r.c = hm;
    r.p = 3.0 * hp / hm;
    r.q = 3.0 * hm / hp;
  }
  return r;
}

constexpr int TABLE_WARMUP = 40;  // |dC_j/dC_{j-1}| <= 1/9 for any spacing: 9^-40 ~ 1e-38

// x and table are indexed by global knot number; entries [j0, j1) are produced and x is read on [j0 - 41, j1]
// (clipped to [0, n)) only, so a shard can pass pointers that are backed by memory just around its own rows.
__global__ __launch_bounds__(256) void spline_table_kernel(const double* __restrict__ x, long long n,
                                                           SplineTable* __restrict__ table, long long j0, long long j1) {
  const long long j = j0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= j1) return;
  long long js = j - TABLE_WARMUP;
  double C = 0.25;  // any start in [0, 1/2]
  if (js <= 0) {
    js = 0;
  }
  RowCoef r;
  double m = 0.0;
  for (long long i = js; i <= j; ++i) {
    r = spline_row(x, i, n);
    m = 1.0 / (i == 0 ? r.b : (r.b - r.a * C));
    C = r.c * m;
  }
  SplineTable e;
  e.P = r.p * m;
  e.Q = r.q * m;
  e.A = r.a * m;
  e.C = C;
  table[j] = e;
}

hipError_t launch_spline_table(hipStream_t stream, const double* x, long long n, SplineTable* table, long long j0,
                               long long j1) {
  if (n < 4 || j0 < 0 || j1 > n) return hipErrorInvalidValue;
  if (j1 <= j0) return hipSuccess;
  hipLaunchKernelGGL(spline_table_kernel, dim3((unsigned)((j1 - j0 + 255) / 256)), dim3(256), 0, stream, x, n, table, j0, j1);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ forward
// Thread (pixel p, tile): r'_j for knots j in [jA, jB), recurrence started `halo` knots earlier.
// Rows of Y/R: buffer row r <-> knot g0 + r.
__global__ __launch_bounds__(64) void spline_forward_kernel(const double* __restrict__ Y, double* __restrict__ R,
                                                            long long ld, int n_cols, long long g0, long long n_rows,
                                                            long long n, const SplineTable* __restrict__ table,
                                                            int tile, int halo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_cols) return;
  const long long jA = g0 + (long long)blockIdx.y * tile;
  long long jB = jA + tile;
  const long long jend = g0 + n_rows;  // one past the last available knot
  if (jB > jend) jB = jend;
  long long jS = jA - halo;
  if (jS < g0) jS = g0;
  const double* yp = Y + 2LL * p - g0 * ld;  // row pointer by absolute knot: yp + j*ld
  double* rp = R + 2LL * p - g0 * ld;
  auto ld2 = [&](long long j) { return *reinterpret_cast<const double2*>(yp + j * ld); };

  double2 rprev = {0.0, 0.0};
  long long j = jS;
  double2 ym, y0, y1;
  if (j == 0) {
    // true first row: uses D_0 and D_1
    const double2 a = ld2(0), b = ld2(1), c = ld2(2);
    const SplineTable e = table[0];
    rprev.x = e.P * (b.x - a.x) + e.Q * (c.x - b.x);
    rprev.y = e.P * (b.y - a.y) + e.Q * (c.y - b.y);
    if (jA == 0) *reinterpret_cast<double2*>(rp) = rprev;
    ym = a;
    y0 = b;
    y1 = c;
    j = 1;
  } else {
    ym = ld2(j - 1 < g0 ? g0 : j - 1);
    y0 = ld2(j);
    y1 = (j + 1 < jend) ? ld2(j + 1) : y0;
  }
  // main loop, 4 knots per trip: the 4 next rows are requested together (4 x 16 B per lane in flight) before
  // the dependent recurrence consumes them -- the loop is a latency-bound HBM stream otherwise.
  auto step = [&](long long jj, double2 ynext, bool have_next) {
    const SplineTable e = table[jj];
    double2 dlo, dhi;
    if (jj == n - 1) {  // true last row: D_{n-3} and D_{n-2}
      const double2 a = ld2(n - 3), b = ld2(n - 2), c = ld2(n - 1);
      dlo = {b.x - a.x, b.y - a.y};
      dhi = {c.x - b.x, c.y - b.y};
    } else {
      dlo = {y0.x - ym.x, y0.y - ym.y};
      dhi = {y1.x - y0.x, y1.y - y0.y};
    }
    double2 rj;
    rj.x = e.P * dlo.x + e.Q * dhi.x - e.A * rprev.x;
    rj.y = e.P * dlo.y + e.Q * dhi.y - e.A * rprev.y;
    if (jj >= jA) *reinterpret_cast<double2*>(rp + jj * ld) = rj;
    rprev = rj;
    ym = y0;
    y0 = y1;
    if (have_next) y1 = ynext;
  };
  for (; j + 4 <= jB && j + 5 < jend; j += 4) {
    const double2 n0 = ld2(j + 2), n1 = ld2(j + 3), n2 = ld2(j + 4), n3 = ld2(j + 5);
    step(j, n0, true);
    step(j + 1, n1, true);
    step(j + 2, n2, true);
    step(j + 3, n3, true);
  }
  for (; j < jB; ++j) {
    const bool have = j + 2 < jend;
    const double2 nx = have ? ld2(j + 2) : y1;
    step(j, nx, have);
  }
}

hipError_t launch_spline_forward(hipStream_t stream, const double* Y, double* R, long long ld, int n_cols,
                                 long long g0, long long n_rows, long long n_knots, const double* x,
                                 const SplineTable* table, int tile, int halo) {
  static const int tile_env = BMS_PROBE_ENV("SCRI_AMD_SPLINE_TILE_FWD") ? atoi(BMS_PROBE_ENV("SCRI_AMD_SPLINE_TILE_FWD")) : 0;
  if (tile_env > 0) tile = tile_env;
  (void)x;
  if (n_rows <= 0 || n_cols <= 0) return hipSuccess;
  const long long n_tiles = (n_rows + tile - 1) / tile;
  if (n_tiles > GRID_Y_MAX) return hipErrorInvalidValue;
  dim3 grid((n_cols + 63) / 64, (unsigned)n_tiles);
  hipLaunchKernelGGL(spline_forward_kernel, grid, dim3(64), 0, stream, Y, R, ld, n_cols, g0, n_rows, n_knots, table,
                     tile, halo);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ backward + evaluate
// Thread (pixel p, tile): s_j = r'_j - C_j s_{j+1} from `halo` knots above the tile down to jA; while passing the
// intervals [x_j, x_{j+1}) of the tile, evaluate every output sample whose abscissa
//     u_eval(i) = b_i + (skew_a[p] (b_i - tt) + skew_b[p])          (= u'_i / k_p + alpha_p for b = x)
// falls inside, with the polynomial form scipy uses (c3 t^3 + c2 t^2 + c1 t + c0, t = u_eval - x_j).
constexpr int RING_ROWS = 16;  // power of two; output rows a lane may hold back before they are written

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// One wave = 64 adjacent pixels marching backward over the knots of one time tile.  The lanes cross from one output
// row to the next at different knots (each pixel has its own time offset), so storing a sample the moment it is
// produced tears every 1 KB output row into pieces written at different times (measured: 1.75x the ideal HBM write
// bytes).  Instead each lane parks its samples in its own column of an LDS ring and the wave writes a row once every
// lane is past it: whole rows, one store instruction each.  Row indices are kept relative to i_lo.
__global__ __launch_bounds__(64) void spline_backward_eval_kernel(
    const double* __restrict__ Y, const double* __restrict__ R, long long ld, int n_cols, long long g0, long long n_rows,
    long long n, const double* __restrict__ x, const SplineTable* __restrict__ table, int tile, int halo,
    const double* __restrict__ base, const double* __restrict__ skew_a, const double* __restrict__ skew_b, double tt,
    long long i_lo, long long i_hi, double* __restrict__ out, long long ldo) {
  __shared__ double2 ring[RING_ROWS][64];
  const int lane = threadIdx.x;
  int p = blockIdx.x * blockDim.x + lane;
  bool alive = p < n_cols;
  if (!alive) p = n_cols - 1;
  const long long jend = g0 + n_rows;
  const long long jA = g0 + (long long)blockIdx.y * tile;
  long long jB = jA + tile;
  if (jB > jend) jB = jend;
  // intervals handled: j in [jA, jI) with jI = min(jB, n-1)
  const long long jI = jB < n - 1 ? jB : n - 1;
  if (jI <= jA) return;
  const bool open_top = (jI == n - 1);  // claims everything above
  const bool open_bottom = (jA == 0);   // claims everything below
  const double sa = skew_a ? skew_a[p] : 0.0, sb = skew_b ? skew_b[p] : 0.0;
  const double* bp = base + i_lo;
  const int n_i = (int)(i_hi - i_lo);
  auto ueval = [&](int i) {
    const double xi = bp[i];
    return xi + (sa * (xi - tt) + sb);
  };
  // largest i in [0, n_i) with u_eval(i) < x[jI] (all of them if open_top)
  int i;
  if (open_top) {
    i = n_i - 1;
  } else {
    const double xt = x[jI];
    int lo = 0, hi = n_i;  // first index with ueval >= xt
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ueval(mid) < xt)
        lo = mid + 1;
      else
        hi = mid;
    }
    i = lo - 1;
  }
  if (i < 0) alive = false;
  if (alive && !open_bottom && ueval(i) < x[jA]) alive = false;  // nothing lands in this tile
  if (!__any(alive)) return;
  if (!alive) i = -1;

  const double* yp = Y + 2LL * p - g0 * ld;
  const double* rp = R + 2LL * p - g0 * ld;
  double* op = out + 2LL * p;
  auto ld2 = [&](const double* base, long long j) { return *reinterpret_cast<const double2*>(base + j * ld); };

  long long jE = jI + halo;
  if (jE > jend - 1) jE = jend - 1;
  // s at jE: exact at the true last knot, otherwise the truncated start (decays as 0.268^halo)
  double2 s1 = ld2(rp, jE);
  {
    long long j = jE - 1;
    for (; j - 3 >= jI; j -= 4) {
      const double2 r0 = ld2(rp, j), r1 = ld2(rp, j - 1), r2 = ld2(rp, j - 2), r3 = ld2(rp, j - 3);
      const double C0 = table[j].C, C1 = table[j - 1].C, C2 = table[j - 2].C, C3 = table[j - 3].C;
      s1.x = r0.x - C0 * s1.x, s1.y = r0.y - C0 * s1.y;
      s1.x = r1.x - C1 * s1.x, s1.y = r1.y - C1 * s1.y;
      s1.x = r2.x - C2 * s1.x, s1.y = r2.y - C2 * s1.y;
      s1.x = r3.x - C3 * s1.x, s1.y = r3.y - C3 * s1.y;
    }
    for (; j >= jI; --j) {
      const double C = table[j].C;
      const double2 r = ld2(rp, j);
      s1.x = r.x - C * s1.x;
      s1.y = r.y - C * s1.y;
    }
  }
  // now s1 = s_{jI}
  double2 y1 = ld2(yp, jI);
  double ue = alive ? ueval(i) : 0.0;
  // this lane's samples of rows (i, fl] sit in the ring; every row above `ftop` (wave-uniform) is already in memory
  int fl = i;
  int ftop = wave_max_i32(i);
  auto park = [&](double2 v) {
    if (fl - i >= RING_ROWS) {  // ring full (output much denser than the knots): let the oldest row go
      *reinterpret_cast<double2*>(op + fl * ldo) = ring[fl & (RING_ROWS - 1)][lane];
      --fl;
    }
    ring[i & (RING_ROWS - 1)][lane] = v;
  };
  // write rows (bot, ftop]: each lane contributes the rows it has parked
  auto flush = [&](int bot) {
    for (int row = ftop; row > bot; --row)
      if (row <= fl && row > i) *reinterpret_cast<double2*>(op + row * ldo) = ring[row & (RING_ROWS - 1)][lane];
    const int keep = bot > i ? bot : i;
    if (fl > keep) fl = keep;
    if (ftop > bot) ftop = bot;
  };
  // one interval: s_j from the recurrence, then every output sample that lands in [x_j, x_{j+1})
  auto interval = [&](long long jj, double2 r, double2 y0) {
    const double C = table[jj].C;
    double2 s0;
    s0.x = r.x - C * s1.x;
    s0.y = r.y - C * s1.y;
    const double xj = x[jj];
    const bool last_interval = (jj == jA) && open_bottom;
    if (i >= 0 && (ue >= xj || last_interval)) {
      const double h = x[jj + 1] - xj;
      const double ih = 1.0 / h;
      // scipy: slope = dy/h; t = (s0 + s1 - 2 slope)/h; c3 = t/h; c2 = (slope - s0)/h - t; c1 = s0; c0 = y0
      const double slx = (y1.x - y0.x) * ih, sly = (y1.y - y0.y) * ih;
      const double tx = (s0.x + s1.x - 2.0 * slx) * ih, ty = (s0.y + s1.y - 2.0 * sly) * ih;
      const double c3x = tx * ih, c3y = ty * ih;
      const double c2x = (slx - s0.x) * ih - tx, c2y = (sly - s0.y) * ih - ty;
      while (i >= 0 && (ue >= xj || last_interval)) {
        // t = u_eval - x_j, formed as (x_i - x_j) + skew to keep the small difference exact
        const double xi = bp[i];
        const double t = (xi - xj) + (sa * (xi - tt) + sb);
        double2 v;
        v.x = ((c3x * t + c2x) * t + s0.x) * t + y0.x;
        v.y = ((c3y * t + c2y) * t + s0.y) * t + y0.y;
        park(v);
        --i;
        if (i >= 0) ue = ueval(i);
      }
    }
    s1 = s0;
    y1 = y0;
  };
  long long j = jI - 1;
  if (j - 3 >= jA) {
    // software pipeline: the 8 x 16 B of the NEXT four knots are requested before the dependent chain of the current
    // four starts (rows below jA are clamped: loaded, never used)
    double2 r0 = ld2(rp, j), r1 = ld2(rp, j - 1), r2 = ld2(rp, j - 2), r3 = ld2(rp, j - 3);
    double2 q0 = ld2(yp, j), q1 = ld2(yp, j - 1), q2 = ld2(yp, j - 2), q3 = ld2(yp, j - 3);
    for (; j - 3 >= jA; j -= 4) {
      const long long jn = j - 4;
      const long long c0 = jn >= jA ? jn : jA, c1 = jn - 1 >= jA ? jn - 1 : jA, c2 = jn - 2 >= jA ? jn - 2 : jA,
                      c3 = jn - 3 >= jA ? jn - 3 : jA;
      const double2 nr0 = ld2(rp, c0), nr1 = ld2(rp, c1), nr2 = ld2(rp, c2), nr3 = ld2(rp, c3);
      const double2 nq0 = ld2(yp, c0), nq1 = ld2(yp, c1), nq2 = ld2(yp, c2), nq3 = ld2(yp, c3);
      interval(j, r0, q0);
      interval(j - 1, r1, q1);
      interval(j - 2, r2, q2);
      interval(j - 3, r3, q3);
      // rows every lane has left behind (a finished lane, i = -1, holds nobody back)
      flush(wave_max_i32(i));
      if (!__any(i >= 0)) break;
      r0 = nr0, r1 = nr1, r2 = nr2, r3 = nr3;
      q0 = nq0, q1 = nq1, q2 = nq2, q3 = nq3;
    }
  }
  if (__any(i >= 0))
    for (; j >= jA; --j) interval(j, ld2(rp, j), ld2(yp, j));
  // lanes stop at different rows at the bottom of the tile: whatever is still parked goes out now
  const int low = -wave_max_i32(fl > i ? -i : -0x7fffffff);
  flush(low);
}

hipError_t launch_spline_backward_eval(hipStream_t stream, const double* Y, const double* R, long long ld, int n_cols,
                                       long long g0, long long n_rows, long long n_knots, const double* x,
                                       const SplineTable* table, int tile, int halo, const double* base,
                                       const double* skew_a, const double* skew_b, double tt, long long i_lo,
                                       long long i_hi, double* out, long long ldo) {
  static const int tile_env = BMS_PROBE_ENV("SCRI_AMD_SPLINE_TILE_BWD") ? atoi(BMS_PROBE_ENV("SCRI_AMD_SPLINE_TILE_BWD")) : 0;
  if (tile_env > 0) tile = tile_env;
  if (n_rows <= 0 || n_cols <= 0 || i_hi <= i_lo) return hipSuccess;
  const long long n_tiles = (n_rows + tile - 1) / tile;
  if (n_tiles > GRID_Y_MAX) return hipErrorInvalidValue;
  dim3 grid((n_cols + 63) / 64, (unsigned)n_tiles);
  hipLaunchKernelGGL(spline_backward_eval_kernel, grid, dim3(64), 0, stream, Y, R, ld, n_cols, g0, n_rows, n_knots, x,
                     table, tile, halo, base, skew_a, skew_b, tt, i_lo, i_hi, out, ldo);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 2 or 3 samples
// scipy's CubicSpline (the ABD flavour's interpolant, scri/asymptotic_bondi_data/transformations.py:398-411) degenerates
// to the straight line through 2 samples and to the parabola through 3 (not-a-knot with no interior knot to spare): the
// Lagrange form, one thread per (output sample, column).
__global__ __launch_bounds__(256) void short_series_eval_kernel(const double* __restrict__ Y, long long ld, int n_cols, int n,
                                                                const double* __restrict__ x, const double* __restrict__ base,
                                                                const double* __restrict__ skew_a, const double* __restrict__ skew_b,
                                                                double tt, long long i_lo, long long i_hi, double* __restrict__ out,
                                                                long long ldo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = i_lo + blockIdx.y;
  if (p >= n_cols || i >= i_hi) return;
  const double xi = base[i];
  const double u = xi + ((skew_a ? skew_a[p] : 0.0) * (xi - tt) + (skew_b ? skew_b[p] : 0.0));
  const double2 y0 = *reinterpret_cast<const double2*>(Y + 2LL * p), y1 = *reinterpret_cast<const double2*>(Y + ld + 2LL * p);
  double2 v;
  if (n == 2) {
    const double w = (u - x[0]) / (x[1] - x[0]);
    v.x = y0.x + w * (y1.x - y0.x);
    v.y = y0.y + w * (y1.y - y0.y);
  } else {
    const double2 y2 = *reinterpret_cast<const double2*>(Y + 2 * ld + 2LL * p);
    const double l0 = (u - x[1]) * (u - x[2]) / ((x[0] - x[1]) * (x[0] - x[2]));
    const double l1 = (u - x[0]) * (u - x[2]) / ((x[1] - x[0]) * (x[1] - x[2]));
    const double l2 = (u - x[0]) * (u - x[1]) / ((x[2] - x[0]) * (x[2] - x[1]));
    v.x = l0 * y0.x + l1 * y1.x + l2 * y2.x;
    v.y = l0 * y0.y + l1 * y1.y + l2 * y2.y;
  }
  *reinterpret_cast<double2*>(out + (i - i_lo) * ldo + 2LL * p) = v;
}

hipError_t launch_short_series_eval(hipStream_t stream, const double* Y, long long ld, int n_cols, int n, const double* x,
                                    const double* base, const double* skew_a, const double* skew_b, double tt, long long i_lo,
                                    long long i_hi, double* out, long long ldo) {
  if (n < 2 || n > 3) return hipErrorInvalidValue;
  if (n_cols <= 0 || i_hi <= i_lo) return hipSuccess;
  hipLaunchKernelGGL(short_series_eval_kernel, dim3((n_cols + 255) / 256, (unsigned)(i_hi - i_lo)), dim3(256), 0, stream, Y, ld, n_cols, n, x,
                     base, skew_a, skew_b, tt, i_lo, i_hi, out, ldo);
  return hipGetLastError();
}

}  // namespace bms
