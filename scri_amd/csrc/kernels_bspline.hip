// The per-pixel not-a-knot cubic spline of kernels_spline.hip in B-spline form (same interpolant:
// scipy InterpolatedUnivariateSpline at scri/waveform_grid.py:574-588 is this very representation, FITPACK's).
//
// Why a second form.  In the slope form an evaluation needs the samples y AND the slopes s, so the grid is written once
// (synthesis), read and written by the forward elimination, and read twice more (y and the eliminated right-hand sides) by
// the back substitution: 5 passes over 2 GB at cfg3.  In B-spline form s(u) = sum_k c_k B_k(u) the evaluation needs the
// coefficients only, and the forward elimination of the collocation system N c = y is a linear map along TIME with
// coefficients shared by all columns, while the synthesis (modes -> grid, per-column offset and scale) is linear along
// the COLUMN axis -- the two commute:
//     F[ (A.B - off) scale ] = ( F[A].B - F[1] off ) scale .
// So the elimination runs on the 285 mode columns instead of the 1297 grid columns, the constant series F[1] rides along
// as one more column of A (multiplying a row -off of B: K goes from 285 to 286, inside the same 8-wide k-chunk), the
// unchanged GEMM writes eliminated coefficients c' directly, and the only pass over the grid is the back substitution +
// evaluation below: 1 read + 1 write.
//
// Not-a-knot = cubic spline space on the knots x_0, x_2, x_3, ..., x_{n-3}, x_{n-1} (x_1 and x_{n-2} are no knots), clamped:
//     tau_0..3 = x_0,  tau_{k+2} = x_k (k = 2..n-3),  tau_n..n+3 = x_{n-1};   B_k supported on [tau_k, tau_{k+4}].
// On the data interval [x_j, x_{j+1}] the four B-splines B_{f_j} .. B_{f_j+3}, f_j = clamp(j - 1, 0, n - 4), are non-zero.
// Collocation rows (site x_i): tridiagonal (c_{i-1}, c_i, c_{i+1}) for 2 <= i <= n-3, four entries c_0..c_3 at i = 1 and
// c_{n-4}..c_{n-1} at i = n-2, c_0 = y_0, c_{n-1} = y_{n-1}.  The matrix is totally positive: elimination without pivoting
// is stable, and the factors decay like 0.268^n on average for any mesh (same tiling + halo as the slope form).
#include <cstdlib>
#include "wigner.h"
#include <algorithm>
#include <cmath>

#include "kernels.h"

namespace bms {

// ------------------------------------------------------------------------------------------------ table
// Power-basis coefficients (in t = u - x_j) of the four B-splines that live on data interval j: Cox-de Boor recursion
// carried out on polynomials.  P[4 d + q] = coefficient of t^d of B_{f_j + q}.
__device__ __forceinline__ void bspline_local(const double* __restrict__ x, long long n, long long j, double P[16]) {
  long long f = j - 1;
  if (f < 0) f = 0;
  if (f > n - 4) f = n - 4;
  const long long mu = f + 3;  // knot span [tau_mu, tau_mu+1) containing the interval
  const double xj = x[j];
  auto tau = [&](long long k) { return k <= 3 ? x[0] : (k >= n ? x[n - 1] : x[k - 2]); };
  double prev[4][4], cur[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int d = 0; d < 4; ++d) prev[i][d] = 0.0;
  prev[0][0] = 1.0;
#pragma unroll
  for (int d = 1; d <= 3; ++d) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) cur[i][g] = 0.0;
#pragma unroll
    for (int i = 0; i <= d; ++i) {
      const long long k = mu - d + i;
      if (i >= 1) {  // (u - tau_k) / (tau_{k+d} - tau_k) * B_{k, d-1}
        const double tk = tau(k);
        const double inv = 1.0 / (tau(k + d) - tk);
        const double c0 = (xj - tk) * inv;
#pragma unroll
        for (int g = 0; g < 4; ++g) cur[i][g] += c0 * prev[i - 1][g];
#pragma unroll
        for (int g = 1; g < 4; ++g) cur[i][g] += inv * prev[i - 1][g - 1];
      }
      if (i <= d - 1) {  // (tau_{k+d+1} - u) / (tau_{k+d+1} - tau_{k+1}) * B_{k+1, d-1}
        const double tk = tau(k + d + 1);
        const double inv = 1.0 / (tk - tau(k + 1));
        const double c0 = (tk - xj) * inv;
#pragma unroll
        for (int g = 0; g < 4; ++g) cur[i][g] += c0 * prev[i][g];
#pragma unroll
        for (int g = 1; g < 4; ++g) cur[i][g] -= inv * prev[i][g - 1];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) prev[i][g] = cur[i][g];
  }
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int q = 0; q < 4; ++q) P[4 * d + q] = prev[q][d];
}

constexpr int BS_TABLE_WARMUP = 40;  // 0.268^40 ~ 1e-23

// Pass 1, one thread per knot: the interval's power-basis table (its first row = the collocation row of site x_j).
__global__ __launch_bounds__(128) void bspline_basis_kernel(const double* __restrict__ x, long long n, BsplineTable* __restrict__ table,
                                                            long long j0, long long j1) {
  const long long j = j0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= j1) return;
  BsplineTable e;
  if (j <= n - 2) {
    bspline_local(x, n, j, e.m);
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) e.m[q] = 0.0;
  }
  e.G = 0.0, e.D = 0.0, e.x = x[j], e.pad = 0.0;
  table[j] = e;
}

// Pass 2: elimination factors of row j from the recursion run over rows [j - 40, j] (exactly from row 0 near the start, and
// never below `j_lo`, the first row pass 1 produced): the pivot recursion forgets its start geometrically.
__global__ __launch_bounds__(128) void bspline_factors_kernel(long long n, BsplineTable* __restrict__ table,
                                                              BsplineForward* __restrict__ fwd, long long j_lo, long long j0,
                                                              long long j1) {
  const long long j = j0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= j1) return;
  long long js = j - BS_TABLE_WARMUP;
  if (js < j_lo) js = j_lo;
  if (js <= 2) js = 0;
  double g1 = 0.25, g2 = 0.25, d1 = 0.0;  // G_{i-1}, G_{i-2}, D_1
  double p_ = 1.0, a_ = 0.0, e_ = 0.0, g_ = 0.0, dd_ = 0.0;
  for (long long i = js; i <= j; ++i) {
    const double v0 = table[i].m[0], v1 = table[i].m[1], v2 = table[i].m[2], v3 = table[i].m[3];  // B_{f_i + q}(x_i)
    e_ = 0.0, dd_ = 0.0;
    if (i == 0 || i == n - 1) {
      p_ = 1.0, a_ = 0.0, g_ = 0.0;
    } else if (i == 1) {  // a c_0 + b c_1 + c c_2 + d c_3 = y_1
      const double ib = 1.0 / v1;
      p_ = ib, a_ = v0 * ib, g_ = v2 * ib, dd_ = v3 * ib;
      d1 = dd_;
    } else if (i == n - 2) {  // e c_{n-4} + l c_{n-3} + d c_{n-2} + u c_{n-1} = y_{n-2}
      const double lp = v1 - v0 * g2;
      const double m = 1.0 / (v2 - lp * g1);
      p_ = m, a_ = lp * m, e_ = v0 * m, g_ = v3 * m;
    } else {  // l c_{i-1} + d c_i + u c_{i+1} = y_i   (B_{i+2} vanishes at its first knot x_i)
      const double m = 1.0 / (v1 - v0 * g1);
      p_ = m, a_ = v0 * m, g_ = (i == 2 ? v2 - v0 * d1 : v2) * m;
    }
    g2 = g1;
    g1 = g_;
  }
  table[j].G = g_;
  table[j].D = dd_;
  fwd[j] = BsplineForward{p_, a_, e_, table[j].x};
}

// entries [j0, j1); x and the tables are indexed by global knot number and backed from knot j_lo on
hipError_t launch_bspline_table(hipStream_t stream, const double* x, long long n, BsplineTable* table, BsplineForward* fwd,
                                long long j_lo, long long j0, long long j1) {
  if (n < 8 || j0 < 0 || j1 > n || j_lo > j0) return hipErrorInvalidValue;
  if (j1 <= j0) return hipSuccess;
  // pass 1 covers the warm-up rows too; it reads x[j - 2 .. j + 3] (clipped at the true ends)
  long long b0 = j0 - BS_TABLE_WARMUP;
  const long long floor = j_lo == 0 ? 0 : j_lo + 2;
  if (b0 < floor) b0 = floor;
  if (b0 > j0) b0 = j0;
  hipLaunchKernelGGL(bspline_basis_kernel, dim3((unsigned)((j1 - b0 + 127) / 128)), dim3(128), 0, stream, x, n, table, b0, j1);
  hipLaunchKernelGGL(bspline_factors_kernel, dim3((unsigned)((j1 - j0 + 127) / 128)), dim3(128), 0, stream, n, table, fwd, b0, j0, j1);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ forward, on the modes
// Thread (column p, tile): c'_j = P_j y_j - A_j c'_{j-1} - E_j c'_{j-2} for knots of the tile, started `halo` knots
// earlier.  With `with_ones`, column n_modes of the output is the eliminated constant series 1.  (The same kernel
// eliminates grid columns where the synthesis is followed by a time-dependent mixing stage and cannot be commuted.)
__global__ __launch_bounds__(64) void bspline_forward_modes_kernel(const double* __restrict__ A, long long lda, int n_modes,
                                                                   double* __restrict__ O, long long ldo, long long g0,
                                                                   long long n_rows, const BsplineForward* __restrict__ table,
                                                                   int tile, int halo, int with_ones) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_modes + with_ones) return;
  const bool ones = p == n_modes;
  const long long jA = g0 + (long long)blockIdx.y * tile;
  long long jB = jA + tile;
  const long long jend = g0 + n_rows;
  if (jB > jend) jB = jend;
  long long jS = jA - halo;
  if (jS < g0) jS = g0;
  const double* ap = A + 2LL * (ones ? 0 : p) - g0 * lda;
  double* op = O + 2LL * p - g0 * ldo;
  const double2 one = {1.0, 0.0};
  auto ld2 = [&](long long j) { return ones ? one : *reinterpret_cast<const double2*>(ap + j * lda); };
  // The row factors are the same for every lane; as scalar loads each would be an L2 round trip inside the dependent
  // chain.  A zero the compiler cannot see through keeps them vector loads, requested together with the data.
  int opaque_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
  const BsplineForward* tv = table + opaque_zero;
  auto ldt = [&](long long j, double2& pa, double& e) {
    pa = *reinterpret_cast<const double2*>(&tv[j].P);
    e = tv[j].E;
  };
  double2 c1 = {0.0, 0.0}, c2 = {0.0, 0.0};
  auto step = [&](long long j, double2 y, double2 pa, double E) {
    double2 c0;
    c0.x = pa.x * y.x - pa.y * c1.x - E * c2.x;
    c0.y = pa.x * y.y - pa.y * c1.y - E * c2.y;
    if (j >= jA) *reinterpret_cast<double2*>(op + j * ldo) = c0;
    c2 = c1;
    c1 = c0;
  };
  long long j = jS;
  for (; j + 4 <= jB; j += 4) {
    double2 y[4], pa[4];
    double e[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      y[g] = ld2(j + g);
      ldt(j + g, pa[g], e[g]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) step(j + g, y[g], pa[g], e[g]);
  }
  for (; j < jB; ++j) {
    double2 pa;
    double e;
    ldt(j, pa, e);
    step(j, ld2(j), pa, e);
  }
}

hipError_t launch_bspline_forward_modes(hipStream_t stream, const double* A, long long lda, int n_modes, double* Aout,
                                        long long ldo, long long g0, long long n_rows, long long n_knots,
                                        const BsplineForward* table, int tile, int halo, int with_ones) {
  (void)n_knots;
  if (n_rows <= 0 || n_modes <= 0) return hipSuccess;
  static const int tile_env = BMS_PROBE_ENV("SCRI_AMD_BSPLINE_TILE_FWD") ? atoi(BMS_PROBE_ENV("SCRI_AMD_BSPLINE_TILE_FWD")) : 0;
  tile = tile_env > 0 ? tile_env : tile;
  const long long n_tiles = (n_rows + tile - 1) / tile;
  if (n_tiles > GRID_Y_MAX) return hipErrorInvalidValue;  // (21 M knots at the 320-knot tile: the engine's chunks stay far below)
  dim3 grid((n_modes + with_ones + 63) / 64, (unsigned)n_tiles);
  hipLaunchKernelGGL(bspline_forward_modes_kernel, grid, dim3(64), 0, stream, A, lda, n_modes, Aout, ldo, g0, n_rows, table,
                     tile, halo, with_ones);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ backward, on the modes
// The back substitution c_k = c'_k - G_k c_{k+1} - D_k c_{k+2} is, like the elimination, a linear map along time whose
// coefficients all columns share, so it commutes with the synthesis as well: run on the n_modes + 1 columns of the
// eliminated modes it leaves B-spline COEFFICIENTS of the modes, the synthesis product of those is the coefficient grid
// itself, and what remains to do on the grid is a 4-tap evaluation without any recurrence (kernels_gemm_eval.hip).
// Thread (column p, tile): knots of the tile marching down, started `halo` knots above it (out of place: the tile above
// still reads the eliminated rows this one would overwrite).
__global__ __launch_bounds__(64) void bspline_backward_modes_kernel(const double* __restrict__ A, long long lda, int n_cols,
                                                                    double* __restrict__ O, long long ldo, long long g0,
                                                                    long long n_rows, const BsplineTable* __restrict__ table,
                                                                    int tile, int halo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_cols) return;
  const long long jA = g0 + (long long)blockIdx.y * tile;
  long long jB = jA + tile;
  const long long jend = g0 + n_rows;
  if (jB > jend) jB = jend;
  long long jE = jB + halo;
  if (jE > jend) jE = jend;
  const double* ap = A + 2LL * p - g0 * lda;
  double* op = O + 2LL * p - g0 * ldo;
  int opaque_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
  const BsplineTable* tv = table + opaque_zero;  // (row factors as vector loads, requested with the data: see the elimination)
  auto ld2 = [&](long long j) { return *reinterpret_cast<const double2*>(ap + j * lda); };
  auto ldt = [&](long long j) { return *reinterpret_cast<const double2*>(&tv[j].G); };  // (G_j, D_j)
  // w0 = c_{k+1}, w1 = c_{k+2}; the march starts from the eliminated value of the top row (exact at the true end, where
  // G = D = 0, and forgotten like 0.268^halo elsewhere)
  double2 w0 = {0.0, 0.0}, w1 = {0.0, 0.0};
  auto step = [&](long long j, double2 r, double2 gd) {
    double2 c;
    c.x = r.x - gd.x * w0.x - gd.y * w1.x;
    c.y = r.y - gd.x * w0.y - gd.y * w1.y;
    if (j < jB) *reinterpret_cast<double2*>(op + j * ldo) = c;
    w1 = w0;
    w0 = c;
  };
  long long j = jE - 1;
  for (; j - 3 >= jA; j -= 4) {
    double2 r[4], gd[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      r[g] = ld2(j - g);
      gd[g] = ldt(j - g);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) step(j - g, r[g], gd[g]);
  }
  for (; j >= jA; --j) step(j, ld2(j), ldt(j));
}

hipError_t launch_bspline_backward_modes(hipStream_t stream, const double* A, long long lda, int n_cols, double* Aout, long long ldo,
                                         long long g0, long long n_rows, const BsplineTable* table, int tile, int halo) {
  if (n_rows <= 0 || n_cols <= 0) return hipSuccess;
  static const int tile_env = BMS_PROBE_ENV("SCRI_AMD_BSPLINE_TILE_FWD") ? atoi(BMS_PROBE_ENV("SCRI_AMD_BSPLINE_TILE_FWD")) : 0;
  tile = tile_env > 0 ? tile_env : tile;
  const long long n_tiles = (n_rows + tile - 1) / tile;
  if (n_tiles > GRID_Y_MAX) return hipErrorInvalidValue;
  dim3 grid((n_cols + 63) / 64, (unsigned)n_tiles);
  hipLaunchKernelGGL(bspline_backward_modes_kernel, grid, dim3(64), 0, stream, A, lda, n_cols, Aout, ldo, g0, n_rows, table, tile, halo);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ both sweeps, on the modes
// Elimination and back substitution of a tile in ONE pass over memory: a thread keeps its column's TT + 2 HH rows in REGISTERS
// (the tile and HH knots of run-in either side: the elimination starts HH knots below the tile, runs to HH knots above it, and
// the back substitution comes back down from there; both recurrences forget their starts like 0.268^HH), so the modes are read
// once and their coefficients written once -- 2 x 16 (n_modes + 1) bytes per knot instead of the 4 x of the two separate sweeps
// (the run-in rows are re-read by the neighbouring tiles: L2 traffic, not HBM).  Rows outside [g0, g0 + n_rows) are virtual:
// zero data behind identity factors.  The per-knot factors are wave-uniform and travel through LDS (one coalesced load by the
// wave, broadcast reads in the sweeps); every register index is a compile-time constant.
template <int TT, int HH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void bspline_solve_modes_kernel(
    const double* __restrict__ A, long long lda, int n_modes, double* __restrict__ O, long long ldo, long long g0, long long n_rows,
    const BsplineForward* __restrict__ fwd, const BsplineTable* __restrict__ table, int with_ones) {
  constexpr int NR = TT + 2 * HH;
  __shared__ __attribute__((aligned(16))) double fw[NR][4];  // P, A, E of row r (identity for virtual rows)
  __shared__ __attribute__((aligned(16))) double bw[NR][2];  // G, D
  const int lane = threadIdx.x;
  // Blocks b, b + 8, .. share an XCD (round-robin dispatch) and with it an L2: every XCD gets a CONTIGUOUS range of time tiles, walked
  // in time order, so that the run-in rows a tile shares with its neighbours are L2 hits there instead of a second read from HBM by
  // another XCD.
  const int n_real = 2 * (n_modes + with_ones), ncb = (n_real + 63) / 64;
  const long long n_tiles = (n_rows + TT - 1) / TT, per_xcd = (n_tiles + 7) / 8;
  const long long q = blockIdx.x >> 3;
  const long long tile = (long long)(blockIdx.x & 7) * per_xcd + q / ncb;
  if (q / ncb >= per_xcd || tile >= n_tiles) return;
  // a thread = one REAL column (the factors are real: Re and Im of a mode are two independent recurrences), so that the
  // NR rows of a column are NR x 2 of the 256 registers a thread has at two waves per SIMD
  const int p = (int)(q % ncb) * 64 + lane;
  const long long jA = g0 + tile * TT, jend = g0 + n_rows;
  const long long j0 = jA - HH;  // row of register 0
  for (int r = lane; r < NR; r += 64) {
    const long long j = j0 + r;
    const bool ok = j >= g0 && j < jend;
    double2 pa = {1.0, 0.0}, ex = {0.0, 0.0}, gd = {0.0, 0.0};
    if (ok) {
      pa = *reinterpret_cast<const double2*>(&fwd[j].P);
      ex = *reinterpret_cast<const double2*>(&fwd[j].E);
      gd = *reinterpret_cast<const double2*>(&table[j].G);
    }
    *reinterpret_cast<double2*>(&fw[r][0]) = pa;
    fw[r][2] = ex.x;
    *reinterpret_cast<double2*>(&bw[r][0]) = gd;
  }
  const bool live = p < n_real;
  const bool ones = with_ones && p >= 2 * n_modes;  // the constant series 1 + 0 i
  const double one_value = (p & 1) ? 0.0 : 1.0;
  const double* ap = A + (live && !ones ? p : 0);
  double v[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const long long j = j0 + r;
    const bool ok = j >= g0 && j < jend;  // (wave-uniform)
    v[r] = ok ? (ones ? one_value : ap[(j - g0) * lda]) : 0.0;
  }
  __syncthreads();
  // elimination: c'_j = P_j y_j - A_j c'_{j-1} - E_j c'_{j-2}
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const double2 pa = *reinterpret_cast<const double2*>(&fw[r][0]);
    const double e = fw[r][2];
    const double c1 = r >= 1 ? v[r - 1] : 0.0, c2 = r >= 2 ? v[r - 2] : 0.0;
    v[r] = pa.x * v[r] - pa.y * c1 - e * c2;
  }
  // back substitution: c_k = c'_k - G_k c_{k+1} - D_k c_{k+2}
#pragma unroll
  for (int r = NR - 1; r >= HH; --r) {
    const double2 gd = *reinterpret_cast<const double2*>(&bw[r][0]);
    const double w0 = r + 1 < NR ? v[r + 1] : 0.0, w1 = r + 2 < NR ? v[r + 2] : 0.0;
    v[r] = v[r] - gd.x * w0 - gd.y * w1;
  }
  if (!live) return;
  double* op = O + p;
#pragma unroll
  for (int r = HH; r < HH + TT; ++r) {
    const long long j = j0 + r;
    if (j < jend) op[(j - g0) * ldo] = v[r];
  }
}

hipError_t launch_bspline_solve_modes(hipStream_t stream, const double* A, long long lda, int n_modes, double* Aout, long long ldo,
                                      long long g0, long long n_rows, const BsplineForward* fwd, const BsplineTable* table, int with_ones) {
  if (n_rows <= 0 || n_modes <= 0) return hipSuccess;
  constexpr int TT = 48, HH = 32;
  const long long ncb = (2 * (n_modes + with_ones) + 63) / 64, n_tiles = (n_rows + TT - 1) / TT, per_xcd = (n_tiles + 7) / 8;
  dim3 grid((unsigned)(8 * per_xcd * ncb));
  hipLaunchKernelGGL((bspline_solve_modes_kernel<TT, HH>), grid, dim3(64), 0, stream, A, lda, n_modes, Aout, ldo, g0, n_rows, fwd, table, with_ones);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ ABD: mixing + forward
// The Horner mixing of the six AsymptoticBondiData fields (kernels_swsh.hip, abd_mix_kernel; transformations.py:340-385)
// has time-dependent coefficients, so the elimination cannot move onto the modes -- but the two grid passes can be one:
// read the six synthesised fields once, mix in registers, run the six recurrences, write the six eliminated fields.
// NF = 6: psi0 .. psi4 and sigma; NF = 5: sigma stays out (it mixes with nothing and takes the evaluating product: engine_abd.hip)
template <int NF>
__global__ __launch_bounds__(64) void abd_mix_forward_kernel(AbdGrids Yg, AbdGrids Rg, long long ld, int n_cols, long long g0,
                                                             long long n_rows, const BsplineForward* __restrict__ table, int tile,
                                                             int halo, const double* __restrict__ alpha,
                                                             const double* __restrict__ ethk_over_k,
                                                             const double* __restrict__ eth_alpha,
                                                             const double* __restrict__ etheth_alpha,
                                                             const double* __restrict__ inv_k, const double* __restrict__ inv_k3) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_cols) return;
  const long long jA = g0 + (long long)blockIdx.y * tile;
  long long jB = jA + tile;
  const long long jend = g0 + n_rows;
  if (jB > jend) jB = jend;
  long long jS = jA - halo;
  if (jS < g0) jS = g0;
  const double al = alpha[p], ik = inv_k[p], ik3 = inv_k3[p];
  const cplx A = {ethk_over_k[2 * p], ethk_over_k[2 * p + 1]}, B = {eth_alpha[2 * p], eth_alpha[2 * p + 1]};
  const cplx EE = {etheth_alpha[2 * p], etheth_alpha[2 * p + 1]};
  int opaque_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
  const BsplineForward* tv = table + opaque_zero;  // row factors as vector loads, requested with the data
  const long long col = 2LL * p - g0 * ld;
  double2 c1[NF], c2[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) c1[f] = c2[f] = double2{0.0, 0.0};
  auto load = [&](long long j, double2* y, double2& pa, double2& ex) {
#pragma unroll
    for (int f = 0; f < NF; ++f) y[f] = *reinterpret_cast<const double2*>(Yg.y[f] + col + j * ld);
    pa = *reinterpret_cast<const double2*>(&tv[j].P);
    ex = *reinterpret_cast<const double2*>(&tv[j].E);  // (E, x_j)
  };
  auto step = [&](long long j, const double2* y, double2 pa, double2 ex) {
    const double dt = ex.y - al;
    const cplx X = {A.re * dt - B.re, A.im * dt - B.im};
    cplx f[6];
    f[5] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NF; ++i) f[i] = {y[i].x, y[i].y};
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
    const cplx mixed[6] = {{t0.re * ik3, t0.im * ik3}, {t1.re * ik3, t1.im * ik3}, {t2.re * ik3, t2.im * ik3},
                           {t3.re * ik3, t3.im * ik3}, {f[4].re * ik3, f[4].im * ik3},
                           {(f[5].re - EE.re) * ik, (f[5].im - EE.im) * ik}};
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      double2 c0;
      c0.x = pa.x * mixed[i].re - pa.y * c1[i].x - ex.x * c2[i].x;
      c0.y = pa.x * mixed[i].im - pa.y * c1[i].y - ex.x * c2[i].y;
      if (j >= jA) *reinterpret_cast<double2*>(Rg.y[i] + col + j * ld) = c0;
      c2[i] = c1[i];
      c1[i] = c0;
    }
  };
  long long j = jS;
  for (; j + 2 <= jB; j += 2) {
    double2 y0[NF], y1[NF], pa0, pa1, ex0, ex1;
    load(j, y0, pa0, ex0);
    load(j + 1, y1, pa1, ex1);
    step(j, y0, pa0, ex0);
    step(j + 1, y1, pa1, ex1);
  }
  for (; j < jB; ++j) {
    double2 y0[NF], pa0, ex0;
    load(j, y0, pa0, ex0);
    step(j, y0, pa0, ex0);
  }
}

hipError_t launch_abd_mix_forward(hipStream_t stream, const AbdGrids& Y, const AbdGrids& R, long long ld, int n_cols, long long g0,
                                  long long n_rows, const BsplineForward* table, int tile, int halo, const double* alpha,
                                  const double* ethk_over_k, const double* eth_alpha, const double* etheth_alpha,
                                  const double* inv_k, const double* inv_k3, int n_fields) {
  if (n_rows <= 0 || n_cols <= 0) return hipSuccess;
  const long long n_tiles = (n_rows + tile - 1) / tile;
  if (n_tiles > GRID_Y_MAX) return hipErrorInvalidValue;
  dim3 grid((n_cols + 63) / 64, (unsigned)n_tiles);
  if (n_fields == 5)
    hipLaunchKernelGGL(abd_mix_forward_kernel<5>, grid, dim3(64), 0, stream, Y, R, ld, n_cols, g0, n_rows, table, tile, halo, alpha,
                       ethk_over_k, eth_alpha, etheth_alpha, inv_k, inv_k3);
  else
    hipLaunchKernelGGL(abd_mix_forward_kernel<6>, grid, dim3(64), 0, stream, Y, R, ld, n_cols, g0, n_rows, table, tile, halo, alpha,
                       ethk_over_k, eth_alpha, etheth_alpha, inv_k, inv_k3);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void negated_row_kernel(const double* __restrict__ off, double* __restrict__ row, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) row[e] = -off[e];
}
hipError_t launch_negated_row(hipStream_t stream, const double* off, double* row, int n) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(negated_row_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, off, row, n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ backward + evaluate
// One wave = 64 adjacent grid columns marching backward over the knots of one time tile:
//     c_k = c'_k - G_k c_{k+1} - D_k c_{k+2}            (started `halo` knots above the tile)
// and, as soon as c_k is known, every output sample on the interval [x_{k+1}, x_{k+2}) is evaluated from the window
// (c_k, c_{k+1}, c_{k+2}, c_{k+3}) with the interval's power-basis table.  The two end intervals share their neighbour's
// window (no knot at x_1, x_{n-2}).  Output rows are parked in an LDS ring and written whole, as in the slope form.
//
// What bounds the march is neither HBM nor the arithmetic but the latency of the per-knot table values: as scalar loads
// each is an L2 round trip in the dependent chain of every knot (measured: ~4000 cycles per knot whatever the prefetch
// depth of the data, with or without the stores, with or without the polynomial).  So the table entries of a whole group
// of knots travel like the data: requested one group ahead with coalesced vector loads, dropped into LDS, and read back as
// broadcasts.
constexpr int BS_WORDS = sizeof(BsplineTable) / sizeof(double);  // 20
constexpr int BS_W_G = 16, BS_W_D = 17, BS_W_X = 18;

// maximum over the wave, on the cross-lane data path of the VALU (six LDS round trips as ds_bpermute, several times per group
// of knots in the dependent chain of the march)
__device__ __forceinline__ int bs_wave_max_i32(int v) {
#define BS_DPP_MAX(ctrl, rows)                                                      \
  {                                                                                 \
    const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, rows, 0xf, false);        \
    v = o > v ? o : v;                                                              \
  }
  BS_DPP_MAX(0x111, 0xf)  // row_shr:1
  BS_DPP_MAX(0x112, 0xf)  // row_shr:2
  BS_DPP_MAX(0x114, 0xf)  // row_shr:4
  BS_DPP_MAX(0x118, 0xf)  // row_shr:8   -> lane 15 of every row of 16 holds the row's maximum
  BS_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  BS_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's
#undef BS_DPP_MAX
  return __builtin_amdgcn_readlane(v, 63);
}

template <int RING, int GS, int DEFER, int SLOTS, bool NT_IO, bool USEWIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void bspline_backward_eval_kernel(
    const double* __restrict__ C, long long ld, int n_cols, long long g0, long long n_rows, long long n,
    const double* __restrict__ x, const BsplineTable* __restrict__ table, int tile, int halo, const double* __restrict__ base,
    const double* __restrict__ skew_a, const double* __restrict__ skew_b, double tt, long long i_lo, long long i_hi,
    double* __restrict__ out, long long ldo, int tile_first) {
  constexpr int TN = (GS + 2) * BS_WORDS;  // staged words per group: entries k+2 .. k-GS+1
  constexpr int TU = (TN + 63) / 64;
  __shared__ double2 ring[RING][64];
  __shared__ double tbuf[2][TU * 64];
  const int lane = threadIdx.x;
  int p = blockIdx.x * blockDim.x + lane;
  bool alive = p < n_cols;
  if (!alive) p = n_cols - 1;
  const long long jend = g0 + n_rows;
  const long long jA = g0 + (long long)(blockIdx.y + tile_first) * tile;
  long long jB = jA + tile;
  if (jB > jend) jB = jend;
  const long long jI = jB < n - 1 ? jB : n - 1;  // intervals handled: [jA, jI)
  if (jI <= jA) return;
  const bool open_top = (jI == n - 1);
  const bool open_bottom = (jA == 0);
  const double sa = skew_a ? skew_a[p] : 0.0, sb = skew_b ? skew_b[p] : 0.0;
  const double* bp = base + i_lo;
  const int n_i = (int)(i_hi - i_lo);
  auto ueval = [&](int i) {
    const double xi = bp[i];
    return xi + (sa * (xi - tt) + sb);
  };
  int i;
  if (open_top) {
    i = n_i - 1;
  } else {
    const double xt = x[jI];
    int lo = 0, hi = n_i;
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
  if (alive && !open_bottom && ueval(i) < x[jA]) alive = false;
  if (!__any(alive)) return;
  if (!alive) i = -1;

  const double* cp = C + 2LL * p - g0 * ld;
  double* op = out + 2LL * p;
  long long k_min = jA - 1 < n - 4 ? jA - 1 : n - 4;
  if (k_min < g0) k_min = g0;  // (the first interval of a shard's buffer would need the row before it: the caller's margin
                               // keeps every output sample away from there)
  long long jE = jI + halo;
  if (jE > jend - 1) jE = jend - 1;
  // rows of C are read once and rows of `out` written once (2 GB each at cfg3): non-temporal, they need not stay in the caches
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  auto ld2 = [&](long long j) {
    const v2d_t v = NT_IO ? __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(cp + (j >= k_min ? j : k_min) * ld))
                          : *reinterpret_cast<const v2d_t*>(cp + (j >= k_min ? j : k_min) * ld);
    return double2{v.x, v.y};
  };
  auto st2 = [&](double* q, double2 v) {
    if (NT_IO)
      __builtin_nontemporal_store(v2d_t{v.x, v.y}, reinterpret_cast<v2d_t*>(q));
    else
      *reinterpret_cast<double2*>(q) = v;
  };

  // this lane's share of a group's table words: word (lane + 64 u) of the block = word tw of entry k_top - te.  The words
  // of the block that the table leaves free (TN .. 64 TU) carry a window of the output abscissae, base[w_top - WIN + 1 ..
  // w_top]: the march needs base[i - 1] after every sample, and as a load from global memory that value queues up behind
  // the rows just requested for the next group (loads return in order), i.e. it costs the full latency of HBM once per
  // group, in the dependent chain.
  constexpr int WIN = 64 * TU - TN;
  static_assert(WIN >= 16, "no room for the abscissa window");
  int te[TU], tw[TU];
#pragma unroll
  for (int u = 0; u < TU; ++u) {
    const int idx = lane + 64 * u;
    te[u] = idx < TN ? idx / BS_WORDS : -1;
    tw[u] = idx < TN ? idx % BS_WORDS : idx - TN;
  }
  auto tfetch = [&](long long k_top, int w_top, double* tl) {
#pragma unroll
    for (int u = 0; u < TU; ++u)
      if (te[u] >= 0) {
        long long kk = k_top - te[u];
        kk = kk < k_min ? k_min : (kk > jE ? jE : kk);
        tl[u] = reinterpret_cast<const double*>(table + kk)[tw[u]];
      } else {
        int wi = w_top - (WIN - 1) + tw[u];
        wi = wi < 0 ? 0 : (wi > n_i - 1 ? n_i - 1 : wi);
        tl[u] = bp[wi];
      }
  };

  // the output abscissa of the sample in hand and of the one after it: the gather of base[i - 1] is in flight while sample i
  // is evaluated (it sits in the dependent chain of the march otherwise)
  double xi_cur = alive ? bp[i] : 0.0, xi_nxt = (alive && i > 0) ? bp[i - 1] : 0.0;
  double sk = sa * (xi_cur - tt) + sb;  // time skew of this column at the sample in hand
  const double* wv = nullptr;  // the abscissa window of the group in hand (LDS) and the index of its first entry
  int w_lo = 0;
  double ue = xi_cur + sk;
  int fl = i;
  int ftop = bs_wave_max_i32(i);
  auto park = [&](double2 v) {
    if (fl - i >= RING) {  // ring full (output much denser than the knots): let the oldest row go
      st2(op + fl * ldo, ring[fl & (RING - 1)][lane]);
      --fl;
    }
    ring[i & (RING - 1)][lane] = v;
  };
  auto flush = [&](int bot) {
    for (int row = ftop; row > bot; --row)
      if (row <= fl && row > i) st2(op + row * ldo, ring[row & (RING - 1)][lane]);
    const int keep = bot > i ? bot : i;
    if (fl > keep) fl = keep;
    if (ftop > bot) ftop = bot;
  };
  // every output sample on data interval jj (table entry at tb), from the window q0..q3 = c_{f_jj} .. c_{f_jj + 3}
  auto interval = [&](long long jj, const double* tb, double2 q0, double2 q1, double2 q2, double2 q3) {
    const double xj = tb[BS_W_X];
    const bool last_interval = (jj == 0);  // only the tile with jA = 0 gets here: claims everything below
    // B-spline form: the four basis values at t (cubic each, 12 multiply-adds for the wave), then 8 for the sample --
    // the power form of the interval (32 to build + 6 per sample) loses when an interval holds about one sample
    while (i >= 0 && (ue >= xj || last_interval)) {
      // t = u_eval - x_j, formed as (x_i - x_j) + skew to keep the small difference exact
      const double t = (xi_cur - xj) + sk;
      const double b0 = fma(fma(fma(tb[12], t, tb[8]), t, tb[4]), t, tb[0]);
      const double b1 = fma(fma(fma(tb[13], t, tb[9]), t, tb[5]), t, tb[1]);
      const double b2 = fma(fma(fma(tb[14], t, tb[10]), t, tb[6]), t, tb[2]);
      const double b3 = fma(fma(fma(tb[15], t, tb[11]), t, tb[7]), t, tb[3]);
      double2 v;
      v.x = fma(b3, q3.x, fma(b2, q2.x, fma(b1, q1.x, b0 * q0.x)));
      v.y = fma(b3, q3.y, fma(b2, q2.y, fma(b1, q1.y, b0 * q0.y)));
      park(v);
      --i;
      xi_cur = xi_nxt;
      if (i > 0) {
        const int rel = (i - 1) - w_lo;  // position in the staged window
        if constexpr (!USEWIN) {
          // Tiles in which the lanes of a wave stand too far apart for the window (their columns' time skew differs, and the
          // difference grows with |u|: hundreds of rows across the grid at the end of a 1e6-step series) gather base[i - 1] one
          // sample ahead, as round 1 did; the launcher decides tile by tile.
          xi_nxt = bp[i - 1];
        } else if (rel >= 0 && rel < WIN) {
          xi_nxt = wv[rel];
        } else {  // (output much denser than the knots, or columns of very different skew in one wave)
          double far = bp[i - 1];
          asm volatile("" : "+v"(far));  // the wait for it stays on this path
          xi_nxt = far;
        }
      }
      sk = sa * (xi_cur - tt) + sb;
      ue = xi_cur + sk;
    }
  };
  const bool has_top_extra = (n - 2 >= jA) && (n - 2 < jI);
  // window above the coefficient being computed: w0 = c_{k+1}, w1 = c_{k+2}, w2 = c_{k+3}
  double2 w0 = *reinterpret_cast<const double2*>(cp + jE * ld), w1 = {0.0, 0.0}, w2 = {0.0, 0.0};
  auto stepk = [&](long long k, double2 r, const double* tk /* entry k; entries k+1, k+2 sit below it in LDS */) {
    const double G = tk[BS_W_G], D = tk[BS_W_D];
    double2 c;
    c.x = r.x - G * w0.x - D * w1.x;
    c.y = r.y - G * w0.y - D * w1.y;
    if (k == n - 4 && has_top_extra) interval(n - 2, tk - 2 * BS_WORDS, c, w0, w1, w2);
    if (k + 1 >= jA && k + 1 < jI && k <= n - 4) interval(k + 1, tk - BS_WORDS, c, w0, w1, w2);
    if (k == 0 && open_bottom) interval(0, tk, c, w0, w1, w2);
    w2 = w1, w1 = w0, w0 = c;
  };

  // march: knots jE-1 .. k_min in groups of GS; above the tile only the recurrence runs (its start decays as 0.268^halo).
  // SLOTS register sets hold the groups in hand and in flight: the group computed in a trip was requested SLOTS - 1 trips
  // earlier (one trip ~ 900 cycles of dependent arithmetic, a load from HBM under load 2-3 us).
  long long k = jE - 1;
  double2 rr[SLOTS][GS];
  double tl[SLOTS][TU];
  int wtop[SLOTS];  // last index of the abscissa window that travels with the slot's table words
#pragma unroll
  for (int d = 0; d < SLOTS - 1; ++d) {
    const long long kd = k - (long long)d * GS;
    wtop[d] = ftop - 1;
    if (kd >= k_min) {
#pragma unroll
      for (int g = 0; g < GS; ++g) rr[d][g] = ld2(kd - g);
      tfetch(kd + 2, wtop[d], tl[d]);
    }
  }
  int buf = 0;
  int pending_bot = 0x7fffffff;  // rows above it are complete but not yet written (DEFER: written one group late)
  bool done = false;
  while (!done) {
#pragma unroll
    for (int ph = 0; ph < SLOTS; ++ph) {
      if (done || k < k_min) {
        done = true;
        continue;
      }
      double* tb = tbuf[buf];
#pragma unroll
      for (int u = 0; u < TU; ++u) tb[lane + 64 * u] = tl[ph][u];
      wv = tb + TN;
      w_lo = wtop[ph] - (WIN - 1);
      // (the wait for this group's loads has just drained the memory pipe: the stores of the previous group go out now, a
      // whole group ahead of the next wait, instead of right in front of it)
      if (DEFER && pending_bot != 0x7fffffff) flush(pending_bot);
      // the group SLOTS - 1 trips ahead is requested into the set that became free in the previous trip
      const int nslot = (ph + SLOTS - 1) % SLOTS;  // (a constant once the loop is unrolled)
      const long long kn = k - (long long)(SLOTS - 1) * GS;
      if (kn >= k_min) {
#pragma unroll
        for (int g = 0; g < GS; ++g) rr[nslot][g] = ld2(kn - g);
        wtop[nslot] = bs_wave_max_i32(i) - 1;  // nobody will ask for an entry above where the first lane stands now
        tfetch(kn + 2, wtop[nslot], tl[nslot]);
      }
#pragma unroll
      for (int g = 0; g < GS; ++g) {
        if (k - g >= k_min) stepk(k - g, rr[ph][g], tb + (2 + g) * BS_WORDS);
        if (!DEFER && ((g & 3) == 3 || g == GS - 1)) flush(bs_wave_max_i32(i));  // rows every lane has left behind
      }
      if (DEFER) pending_bot = bs_wave_max_i32(i);
      if (!__any(i >= 0)) done = true;
      buf ^= 1;
      k -= GS;
    }
  }
  if (DEFER && pending_bot != 0x7fffffff) flush(pending_bot);
  // lanes stop at different rows at the bottom of the tile: whatever is still parked goes out now
  const int low = -bs_wave_max_i32(fl > i ? -i : -0x7fffffff);
  flush(low);
}

hipError_t launch_bspline_backward_eval(hipStream_t stream, const double* C, long long ld, int n_cols, long long g0,
                                        long long n_rows, long long n_knots, const double* x, const BsplineTable* table,
                                        int tile, int halo, const double* base, const double* skew_a, const double* skew_b,
                                        double tt, long long i_lo, long long i_hi, double* out, long long ldo, const BsplineSpread* spread) {
  static const int tile_env = BMS_PROBE_ENV("SCRI_AMD_SPLINE_TILE_BWD") ? atoi(BMS_PROBE_ENV("SCRI_AMD_SPLINE_TILE_BWD")) : 0;
  if (n_rows <= 0 || n_cols <= 0 || i_hi <= i_lo) return hipSuccess;
  if (tile_env > 0) {
    tile = tile_env;
  } else {
    // A wave's time grows with tile + halo knots and the launch takes a whole number of rounds of resident waves
    // (16 per CU) plus a tail: measured over 96..1280-knot tiles on the three benchmark shapes (21, 154 and 7 column
    // blocks), time ~ (ceil(waves / slots) + 1/4) (tile + halo) ranks the candidates within a few per cent.
    static const long long slots = [] {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
      return 16LL * cus;
    }();
    const int cand[4] = {96, 160, 320, 640};
    double best = 1e300;
    for (int t : cand) {
      const long long waves = ((n_cols + 63) / 64) * ((n_rows + t - 1) / t);
      const double cost = ((double)((waves + slots - 1) / slots) + 0.25) * (t + halo);
      if (cost < best) best = cost, tile = t;
    }
  }
  const long long n_tiles = (n_rows + tile - 1) / tile;
  static const int xp = BMS_PROBE_ENV("SCRI_AMD_BS_XP") ? atoi(BMS_PROBE_ENV("SCRI_AMD_BS_XP")) : 0;
  // Tile by tile: may the march read the output abscissae from the window staged with the table words?  Only while the lanes
  // of a wave stay within a few samples of each other, i.e. while (range of the skew rate within 64 columns) x |x - tt| + (range
  // of the skew offset) is a few time steps; elsewhere (and when the caller gives no bound) the lanes gather them.
  auto fits = [&](long long y) {
    if (!spread || !spread->x || xp == 8) return false;
    const long long jA = g0 + y * tile, jB = std::min(g0 + n_rows, jA + tile);
    if (jB - jA < 2) return false;
    const double xm = std::max(std::fabs(spread->x[jA] - tt), std::fabs(spread->x[jB - 1] - tt));
    const double dx = (spread->x[jB - 1] - spread->x[jA]) / (double)(jB - 1 - jA);
    return dx > 0 && (spread->skew_rate_range * xm + spread->skew_offset_range) <= 10.0 * dx;
  };
#define BS_GO(R, G_, X_, S_, N_, W_)                                                                                                  \
  hipLaunchKernelGGL((bspline_backward_eval_kernel<R, G_, X_, S_, N_, W_>), grid, dim3(64), 0, stream, C, ld, n_cols, g0, n_rows, n_knots, x, table, \
                     tile, halo, base, skew_a, skew_b, tt, i_lo, i_hi, out, ldo, (int)y0)
  for (long long y0 = 0; y0 < n_tiles;) {
    const bool w = fits(y0);
    long long y1 = y0 + 1;
    while (y1 < n_tiles && y1 - y0 < GRID_Y_MAX && fits(y1) == w) ++y1;
    dim3 grid((n_cols + 63) / 64, (unsigned)(y1 - y0));
    if (xp == 1) {  // (ring rows, knots per group, deferred stores, register sets, non-temporal rows, abscissa window)
      BS_GO(8, 3, 0, 2, false, false);
    } else if (xp == 4) {
      BS_GO(8, 3, 1, 2, false, false);
    } else if (w) {
      BS_GO(8, 3, 1, 2, true, true);
    } else {  // (lanes spread over many samples write their rows in pieces: non-temporal stores of pieces cost 12.7 -> 20 ms at cfg4)
      BS_GO(8, 3, 1, 2, false, false);
    }
    y0 = y1;
  }
#undef BS_GO
  return hipGetLastError();
}

}  // namespace bms
