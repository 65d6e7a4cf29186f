// Launchers of the gfx950 kernels (C++ internal interface between the engine and the kernels).
#pragma once
#include <algorithm>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env.h"

#include <map>
#include <mutex>
#include <utility>

namespace bms {

// y extent of a launch grid (HIP: 65 535); launchers whose y counts time tiles cut longer launches into slices or refuse them
constexpr long long GRID_Y_MAX = 65535;

// Dynamic LDS beyond 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel.  It is set ONCE per (kernel, device), to
// everything the CU has beyond the kernel's static LDS: a per-launch value is a race between host threads that launch the same kernel
// with different sizes (thread A sets 124 KB, thread B sets 60 KB, A's launch is refused) -- contexts are meant to be driven from
// several threads (include/scri_amd.h).
// The per-CU LDS size is asked of the device (the opt-in limit where the runtime reports one, else the per-block limit; this library
// only runs on gfx950, whose 160 KB is the floor assumed when neither query answers with more than the 64 KB default).  Only a SUCCESS
// is remembered: a transient failure is tried again by the next launch instead of failing that kernel for the life of the process.
inline hipError_t allow_dynamic_lds(const void* fn) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, bool> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> guard(mu);
  const auto key = std::make_pair(fn, dev);
  if (done.count(key)) return hipSuccess;
  int lds = 0, v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess) lds = std::max(lds, v);
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess) lds = std::max(lds, v);
  (void)hipGetLastError();
  if (lds <= 64 * 1024) lds = 160 * 1024;
  hipFuncAttributes attr;
  hipError_t e = hipFuncGetAttributes(&attr, fn);
  if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds - (int)attr.sharedSizeBytes);
  if (e == hipSuccess)
    done[key] = true;
  else
    (void)hipGetLastError();
  return e;
}

constexpr int ROT_MB = 4;  // mu-block of the packed Delta tables (see kernels_rotate.hip)

// ---- rotation (scri/rotations.py:346-392)
int rotate_waves_per_block(int ell_max);
hipError_t launch_rotate_modes(hipStream_t stream, double* data, long long n_times, long long ld, int ell_min, int ell_max,
                               const double* RaRb, long long rotor_stride, const double* delta,
                               const long long* delta_off);

// MFMA formulation (kernels_rotate_mfma.hip), ell_max <= ~32: btab = per-l packed B images, boff = offsets (doubles)
struct RotGeom;
void rotate_mfma_table_shape(int ell, int* kpad, int* pd);
int rotate_mfma_supported(int ell_max);
hipError_t launch_rotate_modes_mfma(hipStream_t stream, double* data, long long n_times, long long ld, int ell_min,
                                    int ell_max, const double* RaRb, long long rotor_stride, const double* btab,
                                    const long long* boff);

// Wave-autonomous formulation with all tables of the l range resident in the LDS (kernels_rotate_resident.hip)
constexpr int RR_MAXL = 24;
struct RotResPlan {
  int ell_min, ell_max, n_groups, kpad_max, tab_doubles, waves;
  int grp_lo[4], grp_hi[4];
  int tab_off[RR_MAXL];  // doubles, per l - ell_min
};
bool rotate_resident_plan(int ell_min, int ell_max, RotResPlan* P, size_t* lds_bytes);
void rotate_resident_pack(const RotResPlan& P, int ell, const double* Delta, double* image);
hipError_t launch_rotate_modes_resident(hipStream_t stream, double* data, long long n_times, long long ld, const double* RaRb,
                                        long long rotor_stride, const double* tab_global, const RotResPlan& P, size_t lds_bytes,
                                        unsigned int* counter, int n_cu);

// ---- mode-space operators on resident series (kernels_modes.hip)
struct ModeMapSide {
  const double* data;  // c16[n_rows][ld]; nullptr: side absent
  long long ld;        // row stride (complex)
  const int* idx;      // source column per output column, -1: zero
  const double* coef;  // c16 per output column
  int conj;            // conjugate the source
};
// s_t = sum_j |data[t][j]|^2 (or its square root), terms added in column order without contraction (waveform_base.py:19-35)
hipError_t launch_row_norm(hipStream_t stream, const double* data, long long ld, long long n_rows, int n_cols, int take_sqrt, double* out);
// trailing data dimensions: the reference's layout (trailing index fastest) <-> one block of unit-stride columns per series
hipError_t launch_series_to_blocks(hipStream_t stream, const double* in, long long ld_in, double* out, long long n_rows, int n_modes, int F);
hipError_t launch_blocks_to_series(hipStream_t stream, const double* in, long long block_rows, double* out, long long n_rows, int n_cols, int F);
hipError_t launch_mode_map(hipStream_t stream, double* out, long long ld_out, long long n_rows, int n_cols, const ModeMapSide& A,
                           const ModeMapSide& B, const double* row_scale);

// ---- SWSH matrices (sf.SWSH_grid, waveform_grid.py:470-484)
// Bmat[2k][2p] = Re Y_k(R_p), [2k][2p+1] = Im, [2k+1][2p] = -Im, [2k+1][2p+1] = Re;  k = LM_index(l,m,ell_min)
hipError_t launch_swsh_matrix(hipStream_t stream, const double* rotors /* f8[n_pix][4] */, int n_pix, int spin,
                              int ell_min, int ell_max, double* Bmat, long long ldb);
// the same harmonics as a plain complex matrix Y[k][p] (row pitch ldb doubles), the B operand of launch_zgemm3m
hipError_t launch_swsh_matrix_complex(hipStream_t stream, const double* rotors, int n_pix, int spin, int ell_min,
                                      int ell_max, double* Bmat, long long ldb);
// analysis (quadrature) matrix: W[p][k] = w_pix[p] conj(Y_k(R_p)) in the real layout with pixels as rows:
// Wmat[2p][2k] = Re W, [2p][2k+1] = Im W, [2p+1][2k] = -Im W, [2p+1][2k+1] = Re W
hipError_t launch_quadrature_matrix(hipStream_t stream, const double* rotors, const double* w_pix, int n_pix, int spin,
                                    int ell_min, int ell_max, double* Wmat, long long ldw);
// complex values Y[p][k] (row = pixel), for tests / small per-pixel tables
hipError_t launch_swsh_values(hipStream_t stream, const double* rotors, int n_pix, int spin, int ell_min, int ell_max,
                              double* Y /* c16[n_pix][n_modes] */);

// ---- per-pixel set-up tables on the device (pixel_math.h)
struct PixelSpec;
struct PixelOut;
// perm (device, may be null): column p holds grid pixel perm[p]
hipError_t launch_pixel_tables(hipStream_t stream, const PixelSpec& P, const PixelOut& O, int n_pix, const int* perm);
// Column plan of an n_theta x n_phi grid (n_theta n_phi <= pixel_sort_max(), n_theta >= 3): the pole rings contribute one
// column each (n_cols = n_pix - 2 (n_phi - 1)); perm[n_cols] = grid pixel of a column, in grid order (by_key = 0) or sorted by
// key (by_key = 1); inv[n_pix] = column of a grid pixel (pole pixels share their ring's column)
int pixel_sort_max();
hipError_t launch_pixel_sort(hipStream_t stream, const PixelSpec& P, int n_theta, int n_phi, int by_key, int* perm, int* inv);

// ---- separable analysis (kernels_analysis.hip): phi-DFT matrix for the GEMM, theta table, theta quadrature
hipError_t launch_dft_matrix(hipStream_t stream, int n_phi, int L, double* B, long long ldb);
hipError_t launch_theta_table(hipStream_t stream, const double* Y, const double* w_theta, int n_theta, int n_out, double* T);
hipError_t launch_theta_quadrature(hipStream_t stream, const double* F, long long n_rows, int n_theta, int nm, int n_out,
                                   const int* m_index, const double* T, double* out, long long ldo);
constexpr int MAX_THETA_SEPARABLE = 104;
// large grids (40 < n_theta <= 104): folded phi-DFT to F[t][m][jp] (jp = large_analysis_jp(n_theta) rings), then the theta
// quadrature as MFMA products batched over time; T from launch_theta_table; F: n_rows * (2L+1) * jp complex of work space
int large_analysis_supported(int n_theta, int n_phi, int L);
int large_analysis_jp(int n_theta);
hipError_t launch_analysis_large(hipStream_t stream, const double* G, long long ldg, long long n_rows, int n_theta, int n_phi,
                                 int L, int ell_min_out, const double* T, double* F, double* out, long long ldo);
// fused single-kernel analysis (n_theta <= 40, n_out <= 1024): D = cos|sin DFT matrix [4 ks][pd] from launch_dft_cs_matrix
struct FusedGeom;
int fused_analysis_supported(int n_theta, int n_phi, int L, int n_out);
size_t fused_dft_table_size(int n_phi, int L);  // doubles
hipError_t launch_dft_cs_matrix(hipStream_t stream, int n_phi, int L, double* D);
// col_of_pixel (device, may be null): grid pixel g is column col_of_pixel[g] of G; then the pixels k > 0 of the two pole
// rings are read from their ring's column times the spin phase e^{-+ i spin phi_k}
hipError_t launch_analysis_fused(hipStream_t stream, const double* G, long long ldg, long long n_rows, int n_theta, int n_phi,
                                 int L, int n_out, const int* m_index, const double* T, const double* D, double* out,
                                 long long ldo, const int* col_of_pixel, int spin, bool allow_split = true);

// series of 2 or 3 samples (ABD flavour): line / parabola through the samples, rows 0..n-1 of Y
hipError_t launch_short_series_eval(hipStream_t stream, const double* Y, long long ld, int n_cols, int n, const double* x,
                                    const double* base, const double* skew_a, const double* skew_b, double tt, long long i_lo,
                                    long long i_hi, double* out, long long ldo);

// ---- separable synthesis for boost-free transformations (kernels_synthesis.hip)
struct SynGeom {
  int n_theta, n_phi, L, n_modes, nk;
  int nth, nph;      // theta and phi waves
  int n_lists, len;  // lists of mode entries the theta threads walk, entries per list (8 .. 20)
};
int synthesis_split_plan(int n_theta, int n_phi, int ell_min, int ell_max, SynGeom& g, std::vector<int>& meta, size_t& lds_bytes,
                         int& nt, int& len);
// A: [n_rows][lda] modes (complex, n_modes + 1 per row: the last one multiplies `off`), Tsyn[n_modes][n_theta] = sLambda_lm(theta_j),
// off: complex per grid pixel or null; Y[n_rows][ldy] = grid rows in grid order; scale: null or 2 (equal) doubles per pixel
// multiplying the result -- it takes synthesis_split_scale_bytes more LDS than the plan's lds_bytes
size_t synthesis_split_scale_bytes(int n_theta, int n_phi);
hipError_t launch_synthesis_split(hipStream_t stream, const double* A, long long lda, long long n_rows, const SynGeom& g, int nt,
                                  const double* Tsyn, const int* meta, const double* off, double* Y, long long ldy, size_t lds_bytes,
                                  int n_cu, const double* scale = nullptr);

// the one-kernel form with the spline evaluation in it (kernels_synthesis_eval.hip): A = B-spline coefficients of the (rotated) modes
// over the knots e.g0 .. e.g0 + n_rows - 1 (n_modes per row), the output e.out = SAMPLES at the output rows [e.i_lo, e.i_hi) in grid
// order; e.skew_b per grid pixel (e.skew_a unused: no boost), s_min / s_max = its range.
// g, nt, Tsyn, meta: synthesis_split_plan's.
struct SplineEval;
int synthesis_eval_supported(const SynGeom& g, int nt, size_t* lds_bytes, int* nph, int* ring_rows, int* staged_abscissae);
hipError_t launch_synthesis_eval(hipStream_t stream, const double* A, long long lda, long long n_rows, const SynGeom& g, int nt,
                                 const double* Tsyn, const int* meta, const SplineEval& e, double s_min, double s_max, int n_cu);
// Y[k][col[j]] -= c[j] Y[k][one_col] (complex; c: n complex numbers, col: n column indices, all device memory): the h / sigma term
// subtracted on the modes
hipError_t launch_sub_const_modes(hipStream_t stream, double* Y, long long ld, long long n_rows, int n, const int* col, const double* c, int one_col);

// two-kernel form for the grids the one-kernel form does not take (kernels_synthesis_large.hip): n_theta <= 104, n_phi <= 127,
// l_max <= 33, any l_min; F = n_rows x (2 l_max + 1) x large_analysis_jp(n_theta) complex of work space.  With `off` the row has
// one more complex number at column n_modes (the eliminated constant series), which multiplies -off[pixel]; `scale` (2 doubles per
// pixel, equal) multiplies the result.
int large_synthesis_supported(int n_theta, int n_phi, int ell_min, int ell_max);
hipError_t launch_synthesis_large(hipStream_t stream, const double* A, long long lda, long long n_rows, int n_theta, int n_phi,
                                  int ell_min, int ell_max, const double* Tsyn, const double* off, double* F, double* Y,
                                  long long ldy, const double* scale = nullptr);

// AsymptoticBondiData without a boost: theta stage per field, then the phi stage of all six fields fused with their Horner mixing
// (time-independent without a boost: the elimination has moved onto the modes); per-pixel tables in grid order
hipError_t launch_theta_synthesis(hipStream_t stream, const double* A, long long lda, long long n_rows, int n_theta, int ell_min,
                                  int ell_max, const double* Tsyn, double* F);
int abd_mix6_supported(int n_theta, int n_phi, int ell_max);
hipError_t launch_phi_synthesis_mix6(hipStream_t stream, const double* const F6[6], long long n_rows, int n_theta, int n_phi,
                                     int ell_max, const double* eth_alpha, const double* etheth_alpha, const double* cst, long long ldc,
                                     double* const out6[6], long long ldo);

// ---- dense fp64 GEMM on MFMA: C[M x N] = (A[M x K] * B[K x N] - col_off[N]) * col_scale[N]
// A row-major (lda), B row-major (ldb, zero padded to a multiple of 128 columns and 16 rows), C row-major (ldc).
hipError_t launch_dgemm(hipStream_t stream, const double* A, long long lda, const double* B, long long ldb, double* C,
                        long long ldc, long long M, int N, int K, const double* col_off, const double* col_scale);

// ---- complex128 GEMM on MFMA with 3 real products per complex one: C = (A . B - col_off) * col_scale
// A[M x K], B[K x N], C[M x N] complex interleaved, row-major, pitches lda/ldb/ldc in DOUBLES; B zero padded to a multiple
// of 64 complex columns and 8 rows; col_off/col_scale per real column (2N entries) or null.  The last K chunk reads up to 7 rows of B
// past row K - 1: the kernel zero-guards A for k >= K, so those rows need not be zero (AsymptoticBondiData keeps sigma's offset row
// right behind the rows psi0 multiplies) -- they only have to exist.
hipError_t launch_zgemm3m(hipStream_t stream, const double* A, long long lda, const double* B, long long ldb, double* C,
                          long long ldc, long long M, int N, int K, const double* col_off, const double* col_scale);

// ---- the same product with the spline evaluation in its epilogue (kernels_gemm_eval.hip): A holds B-spline coefficients of the
// modes (both sweeps of the collocation solve done on the modes), the product is the coefficient grid, and each 64-knot tile
// evaluates the output samples whose four coefficients it holds -- only the samples are written.
struct BsplineTable;
struct SplineEval {
  const BsplineTable* table;  // per knot (global index), as for launch_bspline_backward_eval
  const double* x;            // knots = output abscissae before the skew (global index)
  const double* skew_a;       // per column of the launch (may be null)
  const double* skew_b;
  double tt;
  long long g0;       // knot of row 0 of A
  long long n_knots;  // of the whole series
  long long i_lo, i_hi;  // output rows (indices into x); out row 0 = i_lo
  double* out;
  long long ldo;
  int search_halfwidth;  // bound on |row of a sample - knot of its window| (0: none known)
  double inv_dx;         // 1 / (mean step of the knots of this launch), or 0: row guesses allowed at all (the kernel uses each tile's own step)
  unsigned long long* stats;  // device, or null: [0] += tiles whose samples left the staged window, [1] += marches continued from global memory
  double* side;          // 6 rows of ldc doubles per 64-row tile (zgemm3m_eval_side_rows(M) rows), or null: overlapping tiles
  long long side_ld;
  int step = 0;          // rows a tile advances: 0 = automatic (64 with `side`, 61 without), 61 or 64 (the context's GEMM_EVAL_STEP option)
};
long long zgemm3m_eval_side_rows(long long M);
hipError_t launch_zgemm3m_eval(hipStream_t stream, const double* A, long long lda, const double* B, long long ldb, long long M, int N,
                               int K, const double* col_scale, const SplineEval& e);

// ---- shared-matrix not-a-knot cubic spline along time (waveform_grid.py:574-588)
struct SplineTable {  // per knot j
  double P, Q, A, C;  // r'_j = P (y_j - y_{j-1}) + Q (y_{j+1} - y_j) - A r'_{j-1};  s_j = r'_j - C s_{j+1}
};
// entries [j0, j1) of the table; x / table are indexed by global knot number and touched on [j0 - 41, j1] only
hipError_t launch_spline_table(hipStream_t stream, const double* x, long long n, SplineTable* table, long long j0,
                               long long j1);
// forward elimination over rows [row_lo, row_hi) of the knot axis; Y, R are [rows][ld] doubles holding
// global rows g0.. (row index r maps to knot g0 + r)
hipError_t launch_spline_forward(hipStream_t stream, const double* Y, double* R, long long ld, int n_cols /* complex */,
                                 long long g0, long long n_rows, long long n_knots, const double* x,
                                 const SplineTable* table, int tile, int halo);
// back substitution + evaluation at u_eval(i) = base_i + skew_a[p] (base_i - tt) + skew_b[p] for output rows
// i in [i_lo, i_hi) (indices into base); out row index = i - i_lo.  skew_a/skew_b may be NULL (= 0).
hipError_t launch_spline_backward_eval(hipStream_t stream, const double* Y, const double* R, long long ld, int n_cols,
                                       long long g0, long long n_rows, long long n_knots, const double* x,
                                       const SplineTable* table, int tile, int halo, const double* base,
                                       const double* skew_a, const double* skew_b, double tt, long long i_lo,
                                       long long i_hi, double* out, long long ldo);

// ---- the same spline in B-spline form (kernels_bspline.hip): coefficients c with s(u) = sum_k c_k B_k(u) on the
// not-a-knot knot vector.  Forward elimination of the collocation system commutes with the synthesis contraction, so it
// runs on the MODES and the grid pass is the back substitution + evaluation alone.
struct BsplineTable {  // per knot j: what the back substitution + evaluation reads (20 words, staged through LDS)
  double m[16];        // m[4 d + q]: coefficient of t^d of B_{f_j + q} on [x_j, x_{j+1}], t = u - x_j, f_j = clamp(j-1, 0, n-4)
  double G, D;         // c_j = c'_j - G c_{j+1} - D c_{j+2}
  double x;            // x_j
  double pad;
};
struct BsplineForward {  // per knot j: c'_j = P y_j - A c'_{j-1} - E c'_{j-2}
  double P, A, E, x;     // x = x_j
};
hipError_t launch_bspline_table(hipStream_t stream, const double* x, long long n, BsplineTable* table, BsplineForward* fwd,
                                long long j_lo /* first knot the arrays are backed for */, long long j0, long long j1);
// Aout[r][0 .. n_modes] = forward-eliminated [A | 1] over rows r = knots g0 .. g0 + n_rows (complex; with_ones: the extra
// last column is the eliminated constant series, which carries the per-column offset through the contraction)
hipError_t launch_bspline_forward_modes(hipStream_t stream, const double* A, long long lda, int n_modes, double* Aout,
                                        long long ldo, long long g0, long long n_rows, long long n_knots,
                                        const BsplineForward* table, int tile, int halo, int with_ones);
// the back substitution on the eliminated modes: Aout = B-spline coefficients of the n_cols columns (out of place)
hipError_t launch_bspline_backward_modes(hipStream_t stream, const double* A, long long lda, int n_cols, double* Aout, long long ldo,
                                         long long g0, long long n_rows, const BsplineTable* table, int tile, int halo);
// both sweeps in one pass over memory (a thread keeps its column's tile + run-in rows in registers): Aout[r][0 .. n_modes] =
// B-spline coefficients of [A | 1]
hipError_t launch_bspline_solve_modes(hipStream_t stream, const double* A, long long lda, int n_modes, double* Aout, long long ldo,
                                      long long g0, long long n_rows, const BsplineForward* fwd, const BsplineTable* table, int with_ones);
// AsymptoticBondiData: Horner mixing of the six synthesised fields (as launch_abd_mix) fused with their elimination
struct AbdGrids;
hipError_t launch_abd_mix_forward(hipStream_t stream, const AbdGrids& Y, const AbdGrids& R, long long ld, int n_cols, long long g0,
                                  long long n_rows, const BsplineForward* table, int tile, int halo, const double* alpha,
                                  const double* ethk_over_k, const double* eth_alpha, const double* etheth_alpha,
                                  const double* inv_k, const double* inv_k3, int n_fields = 6 /* 5: without sigma */);
// B[row][0 .. 2 n_cols) = -off[0 .. 2 n_cols): the synthesis-matrix row that multiplies the constant column
hipError_t launch_negated_row(hipStream_t stream, const double* off, double* row, int n);
// back substitution + evaluation (arguments as launch_spline_backward_eval, C = eliminated grid coefficients)
// how far apart the lanes of a wave can stand (host side, optional): ranges of skew_a / skew_b within any block of 64 columns of the
// launch, and the time axis indexed by global knot number
struct BsplineSpread {
  double skew_rate_range, skew_offset_range;
  const double* x;
};
hipError_t launch_bspline_backward_eval(hipStream_t stream, const double* C, long long ld, int n_cols, long long g0,
                                        long long n_rows, long long n_knots, const double* x, const BsplineTable* table,
                                        int tile, int halo, const double* base, const double* skew_a, const double* skew_b,
                                        double tt, long long i_lo, long long i_hi, double* out, long long ldo, const BsplineSpread* spread = nullptr);

// ---- time-series calculus and grid products (kernels_series.hip; scri/modes_time_series.py:72-202)
// spline slopes s_j at all knots from the forward-pass result R (S != R)
hipError_t launch_spline_slopes(hipStream_t stream, const double* R, double* S, long long ld, int n_cols, long long n,
                                const SplineTable* table, int tile, int halo);
// running integrals at the knots: P1 = int f (order >= 1), P2 = int P1 (order 2); carry: spline_prefix_carry_size doubles
long long spline_prefix_carry_size(long long n, int n_cols);
// antiderivatives of order k >= 3: knot values of the levels 1..k in Pall + (r - 1) level_stride (doubles), then the evaluation
hipError_t launch_spline_prefix_levels(hipStream_t stream, const double* Y, const double* S, long long ld, int n_cols, long long n,
                                       const double* x, double* Pall, long long level_stride, double* carry, int levels);
hipError_t launch_spline_antiderivative_eval(hipStream_t stream, const double* Y, const double* S, const double* Pall, long long level_stride,
                                             long long ld, int n_cols, long long n, const double* x, const double* x_new, long long n_new,
                                             int k, double* out, long long ldo);
hipError_t launch_spline_prefix(hipStream_t stream, const double* Y, const double* S, long long ld, int n_cols, long long n,
                                const double* x, double* P1, double* P2, double* carry, int order);
// out[i][c] = (d/du)^order spline_c(x_new[i]), order in [-2, 3] (negative: antiderivatives vanishing at x[0])
hipError_t launch_spline_hermite_eval(hipStream_t stream, const double* Y, const double* S, const double* P1, const double* P2,
                                      long long ld, int n_cols, long long n, const double* x, const double* x_new,
                                      long long n_new, int order, double* out, long long ldo);
// c = a * b, n complex numbers
hipError_t launch_cmul(hipStream_t stream, const double* a, const double* b, double* c, long long n);

// <Ldt> f8[n][3], <LL> f8[n][9], omega = -<LL>^-1 <Ldt> f8[n][3] (any may be null) from modes F and their time
// derivative Fdot, both c16[n][ld/2] (scri/mode_calculations.py:14-57, 209-313, 403-432)
hipError_t launch_angular_velocity(hipStream_t stream, const double* F, const double* Fdot, long long ld, long long n_times,
                                   int ell_min, int n_modes, double* ldt_out, double* ll_out, double* omega_out);

// ---- bit transforms of the storage formats (kernels_bits.hip; scri/utilities.py:194-406)
// rows of n_cols 64-bit words; forward: out[i] = in[i-1] ^ in[i]; reverse: running XOR (carry: xor_carry_words words)
long long xor_carry_words(long long n_rows, long long n_cols);
hipError_t launch_xor_timeseries(hipStream_t stream, const void* in, void* out, void* carry, long long n_rows, long long n_cols,
                                 int reverse);
// n elements of bit_width bits; widths[n_widths] from the most significant piece down, summing to bit_width
hipError_t launch_multishuffle(hipStream_t stream, const void* in, void* out, long long n, const int* widths, int n_widths,
                               int bit_width, int forward);
// acc[0] += sum d_j mod 65535-ish, acc[1] += sum (N-j) d_j (both to be reduced mod 65535 by the caller); acc zeroed before
hipError_t launch_fletcher32(hipStream_t stream, const void* data, long long n_words, unsigned long long* acc);

// ---- pointwise helpers
// Y[t][p] += coeff * Yaux[t][p] * X[t][p]^power,  X = (x_t - alpha_p) * xa_p - xb_p   (waveform_grid.py:516-550)
hipError_t launch_psi_mix(hipStream_t stream, double* Y, const double* Yaux, long long ld, int n_pix, long long n_rows,
                          const double* x /* times of the rows */, const double* alpha, const double* xa /* c16[n_pix] */,
                          const double* xb /* c16[n_pix] */, double coeff, int power);
// Horner mixing of the six ABD grids in place (transformations.py:340-385)
struct AbdGrids {
  double* y[6];
};
hipError_t launch_abd_mix(hipStream_t stream, const AbdGrids& g, long long ld, int n_pix, long long n_rows, const double* u,
                          const double* alpha, const double* ethk_over_k /* c16[n_pix] */,
                          const double* eth_alpha /* c16 */, const double* etheth_alpha /* c16 */, const double* inv_k,
                          const double* inv_k3);
// y[t][col] = (y[t][col] - off[col]) * scale[col]
hipError_t launch_affine_cols(hipStream_t stream, double* Y, long long ld, int n_cols, long long n_rows,
                              const double* off, const double* scale);

}  // namespace bms
