// The synthesis product with the spline evaluation in its epilogue (scri/waveform_grid.py:475-484 followed by :574-588).
//
// Both sweeps of the spline's collocation solve are linear maps along time with coefficients shared by all columns, so both
// commute with the synthesis (kernels_bspline.hip): solved completely on the n_modes + 1 columns of the modes, the product
//     C[knot][column] = c_full[knot][mode] . Y[mode][column]
// IS the grid of B-spline coefficients, and the value of column p at its own distorted time u_eval(i, p) is a 4-tap
// combination  sum_q b_q(t) C[f + q][p]  of four consecutive knots -- no recurrence left.  A 64-knot x 64-column tile of C
// that has just been accumulated holds everything the samples with f .. f + 3 inside the tile need, so the tile is parked in
// the operand LDS (half of its columns at a time: 33 KB), evaluated from there and only the SAMPLES leave for HBM: the
// coefficient grid (2.08 GB written by the product and read back by bspline_backward_eval_kernel at cfg3) never exists.
//
// Ownership.  Data interval jj = [x_jj, x_jj+1) is evaluated from the window starting at f(jj) = clamp(jj - 1, 0, n - 4)
// (not-a-knot: the two end intervals share their neighbour's window).  Row tiles step by ROW_STEP knots:
//   * ROW_STEP = 61: consecutive tiles overlap by 3 knots and tile bm owns the windows f in [61 bm, 61 bm + 61) -- every
//     window lies inside one tile, at the price of 64/61 of the matrix work;
//   * ROW_STEP = 64: no overlap; the windows with f mod 64 >= 61 straddle two tiles: the tiles drop their first and last
//     three rows into a side buffer and spline_straddle_eval_kernel evaluates those windows (3/64 of the samples) from it.
// A sample belongs to the interval jj = clamp(#{knots <= u_eval} - 1, 0, n - 2), exactly the rule of the marching kernels
// (kernels_bspline.hip: an interval claims what lies at or above its left knot and below its right one).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"

namespace bms {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int E_BM = 64, E_BN = 64, E_KC = 8;
constexpr int E_PA = 10;  // LDS pitch of an A row (complex)
constexpr int E_PB = 64;  // LDS pitch of a B row (complex)
constexpr int E_ASZ = E_BM * E_PA;
constexpr int E_BSZ = E_KC * E_PB;
constexpr int E_PC = 33;  // LDS pitch (complex) of a row of the parked half tile: 64 rows x 32 columns

// Samples of column p (skew sa, sb) whose windows start in [fa, fb): evaluated from `win(f_local, q)` = C[f + q][p].
// x, table: global knot index; bp = output abscissae of rows i_lo ..; out: row 0 = output row i_lo.
struct EvalArgs {
  const BsplineTable* table;
  const double* x;
  const double* skew_a;
  const double* skew_b;
  double tt;
  long long g0, n, i_lo, i_hi;
  double* out;
  long long ldo;
  int search_halfwidth;  // samples of a window lie within this many rows of its knots (host bound; 0: search everything)
  double inv_dx;         // 1 / (mean step of the launch's knots): turns an abscissa into a first guess of its row (0: no guess)
  double* side;          // ROW_STEP = 64: [n_row_tiles][6][side_ld] first / last three rows of every tile, or null
  long long side_ld;
  unsigned long long* stats;  // [0] += tiles (and boundary blocks) whose samples left the staged window, [1] += marches that went on from global memory (may be null)
  int dbg;               // timing experiments (results wrong): 1 no evaluation, 2 no sample stores, 4 no straddle kernel, 8 no search; 32 (results right): launch-wide sample window
  unsigned long long* trace;  // (debug) 5 words per block: hw id | xcc id << 32, tile, clock at start / K loop end / exit
};

// (the knock-out bits and the timeline exist in probe builds only: env.h)
#if BMS_PROBES
#define EV_DBG(ev) ((ev).dbg)
#define EV_TRACE(ev) ((ev).trace)
#else
#define EV_DBG(ev) 0
#define EV_TRACE(ev) (static_cast<unsigned long long*>(nullptr))
#endif

// first i in [0, n_i) with u_eval(i) >= y (n_i if none); guess = index where it would lie without any skew
__device__ __forceinline__ int eval_lower_bound(const double* __restrict__ bp, int n_i, double sa, double sb, double tt, double y,
                                                long long guess, int halfwidth) {
  auto ue = [&](int i) {
    const double xi = bp[i];
    return xi + (sa * (xi - tt) + sb);
  };
  int lo = 0, hi = n_i;
  if (halfwidth > 0) {
    long long a = guess - halfwidth, b = guess + halfwidth;
    if (a < 0) a = 0;
    if (b > n_i) b = n_i;
    if (a < b) {
      // bracket: everything below a is < y, everything from b on is >= y
      const bool ok_lo = (a == 0) || ue((int)a - 1) < y;
      const bool ok_hi = (b == n_i) || !(ue((int)b) < y);
      if (ok_lo && ok_hi) lo = (int)a, hi = (int)b;
    }
  }
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ue(mid) < y)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__device__ __forceinline__ double2 eval_ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ double2 eval_ld2(const double __attribute__((address_space(3))) * p) {
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  const v2d_t v = *(const v2d_t __attribute__((address_space(3)))*)p;
  return double2{v.x, v.y};
}

// State of the march over the windows of one column: interval in hand (relative to kT), sample in hand, where it goes.
struct EvalMarch {
  int jl, jl_end, jl_open, f_lo, f_hi, i, n_i;
  double sa, sb, tt;
  double* op;
  long long ldo;
};

// One loop whose turn moves on to the next interval if the sample in hand lies beyond this one AND evaluates the sample if it
// then lies inside: the lanes of a wave take about (intervals + 2) turns together, whatever the order in which their samples and
// knots interleave.  Everything is indexed relative to kT in 32 bits, and knots, window and table are re-read every turn rather
// than carried and shifted in registers: the evaluating waves share their SIMDs' issue slots with the matrix instructions of the
// other workgroups, so every vector instruction saved here is matrix time gained.  Returns false if the source could not supply
// a sample's abscissa (the LDS window left behind: the caller goes on from global memory); the state is then at that sample.
// INTERIOR: the caller's windows touch neither end of the series (no shared end windows, no open last interval).
template <bool INTERIOR, class SRC, class WIN>
__device__ __forceinline__ bool eval_march(EvalMarch& m, const SRC& src, WIN win, int dbg) {
  if (!src.has(m.i)) return false;
  double xi = src.samp(m.i);
  double sk = m.sa * (xi - m.tt) + m.sb;
  double ue = xi + sk;
  const double inf = __builtin_huge_val();
  double xhi = (!INTERIOR && m.jl >= m.jl_open) ? inf : src.knot(m.jl + 1);
  while (true) {
    if (!(ue < xhi)) {  // the sample lies beyond this interval
      if (++m.jl >= m.jl_end) return true;
      xhi = (!INTERIOR && m.jl >= m.jl_open) ? inf : src.knot(m.jl + 1);
    }
    if (ue < xhi) {
      const auto tb = src.tab(m.jl);
      int fl = m.jl - 1;
      if (!INTERIOR) fl = fl < m.f_lo ? m.f_lo : (fl > m.f_hi ? m.f_hi : fl);
      const double2 q0 = win(fl, 0), q1 = win(fl, 1), q2 = win(fl, 2), q3 = win(fl, 3);
      // t = u_eval - x_j, formed as (x_i - x_j) + skew to keep the small difference exact (as the marching kernels do)
      const double t = (xi - src.knot(m.jl)) + sk;
      const double2 m0 = eval_ld2(tb + 0), m1 = eval_ld2(tb + 2), m2 = eval_ld2(tb + 4), m3 = eval_ld2(tb + 6);
      const double2 m4 = eval_ld2(tb + 8), m5 = eval_ld2(tb + 10), m6 = eval_ld2(tb + 12), m7 = eval_ld2(tb + 14);
      const double b0 = fma(fma(fma(m6.x, t, m4.x), t, m2.x), t, m0.x);
      const double b1 = fma(fma(fma(m6.y, t, m4.y), t, m2.y), t, m0.y);
      const double b2 = fma(fma(fma(m7.x, t, m5.x), t, m3.x), t, m1.x);
      const double b3 = fma(fma(fma(m7.y, t, m5.y), t, m3.y), t, m1.y);
      double2 v;
      v.x = fma(b3, q3.x, fma(b2, q2.x, fma(b1, q1.x, b0 * q0.x)));
      v.y = fma(b3, q3.y, fma(b2, q2.y, fma(b1, q1.y, b0 * q0.y)));
      if (!(dbg & 2)) *reinterpret_cast<double2*>(m.op) = v;
      m.op += m.ldo;
      if (++m.i >= m.n_i) return true;
      if (!src.has(m.i)) return false;
      xi = src.samp(m.i);
      sk = m.sa * (xi - m.tt) + m.sb;
      ue = xi + sk;
    }
  }
}

// Where the march reads its tables from (jl = interval relative to the knot kT of the caller's row 0): global memory ...
struct EvalFromGlobal {
  const EvalArgs& ev;
  const double* bp;
  long long kT;
  __device__ __forceinline__ double knot(int jl) const { return ev.x[kT + jl]; }
  __device__ __forceinline__ const double* tab(int jl) const { return ev.table[kT + jl].m; }
  __device__ __forceinline__ bool has(int) const { return true; }
  __device__ __forceinline__ double samp(int i) const { return bp[i]; }
};
// ... or the copies a tile staged in LDS before its K loop: knots and power-basis tables of its 64 rows, and the E_XS output
// abscissae from row i_a on (host bound on the skew: every sample of the tile lies in that window; a thread that finds otherwise
// goes back to global memory).  Pure LDS reads: a flat or global load in the loop would put s_waitcnt vmcnt(0) into every turn,
// i.e. a wait for the previous turn's sample to reach memory.
constexpr int E_XS = 256;
typedef const double __attribute__((address_space(3))) * lds_cdp;     // (typed as LDS pointers: through generic ones the compiler
typedef const double2 __attribute__((address_space(3))) * lds_cd2p;   //  emits flat loads, which count as memory AND LDS accesses)
struct EvalFromLds {
  lds_cdp t_lds;   // [64][16]
  lds_cdp xk_lds;  // [64]
  lds_cdp xs_lds;  // [E_XS]
  int i_a;
  double inv_dx;   // 1 / (mean step of the staged abscissae): the window's OWN step, not the launch's (0: no guess)
  __device__ __forceinline__ double knot(int jl) const { return xk_lds[jl]; }
  __device__ __forceinline__ lds_cdp tab(int jl) const { return t_lds + 16 * jl; }
  __device__ __forceinline__ bool has(int i) const { return (unsigned)(i - i_a) < (unsigned)E_XS; }
  __device__ __forceinline__ double samp(int i) const { return xs_lds[i - i_a]; }
  // first i with u_eval(i) >= y, or -1 if the window cannot tell
  __device__ __forceinline__ int first(double y, double sa, double sb, double tt, int n_i) const {
    int lo = 0, hi = E_XS;
    if (inv_dx > 0.0) {
      // On a (nearly) uniform axis the row follows from the abscissa: u_eval(i) >= y <=> x_i >= (y - sb + sa tt) / (1 + sa).  The
      // guess is CHECKED against its two neighbours' exact u_eval and only narrows the search: right or one off on the benchmark
      // axes, where it replaces nine dependent LDS round trips (and ~90 vector instructions) by two.
      const double z = (y - (sb - sa * tt)) / (1.0 + sa);
      const double gf = (z - xs_lds[0]) * inv_dx;
      int g = gf < 0.0 ? 0 : (gf > (double)(E_XS - 1) ? E_XS - 1 : (int)gf);
      const double xa = xs_lds[g], xb = xs_lds[g + 1 < E_XS ? g + 1 : E_XS - 1];
      const bool below = xa + (sa * (xa - tt) + sb) < y;  // row g lies below y: the answer is above g
      const bool next_below = xb + (sa * (xb - tt) + sb) < y;
      if (below && !next_below && g + 1 < E_XS)
        lo = hi = g + 1;
      else if (!below && g > 0) {
        const double xc = xs_lds[g - 1];
        if (xc + (sa * (xc - tt) + sb) < y) lo = hi = g;
      }
    }
#pragma unroll
    for (int st = 0; st < 9; ++st) {  // E_XS = 2^8: 257 possible answers
      const int mid = lo < hi ? (lo + hi) >> 1 : (lo < E_XS ? lo : E_XS - 1);  // (converged: any valid entry, the comparison changes nothing)
      const double xi = xs_lds[mid];
      if (lo < hi) {
        if (xi + (sa * (xi - tt) + sb) < y)
          lo = mid + 1;
        else
          hi = mid;
      }
    }
    // inside the window the answer is exact; at its edges only if the window ends where the rows do
    if ((lo > 0 || i_a == 0) && (lo < E_XS || i_a + E_XS >= n_i)) return (i_a + lo) < n_i ? (i_a + lo) : n_i;
    return -1;
  }
};

// The window of output rows the samples of a tile can have, from the time skews of the tile's OWN columns at its first and last knot
// (smin, smax): a sample of global row I sits near knot I + skew / dx, so the knots [kT, kT + rows) are met by the rows
// [kT - smax / dx, kT + rows - smin / dx].  Columns are stored sorted by skew rate, so the spread within a tile stays a few dozen rows
// where the skew itself is hundreds (late times of a long series, strong boosts): a window centred per tile keeps those tiles on the
// LDS path, which a bound for the whole launch (ev.search_halfwidth) sends to global memory as soon as the largest skew anywhere exceeds
// 92 rows -- 1e6 steps: 46.9 -> 44.4 ms per transform.  The margin (3 rows + 3 % of the skew) absorbs a local step that differs from the mean one;
// beyond it the search reports the miss and the thread goes to global memory, as before.
__device__ __forceinline__ bool eval_launch_wide_window(const EvalArgs& ev) {
  return !(ev.inv_dx > 0.0) || (EV_DBG(ev) & 32) || (ev.search_halfwidth > 0 && 2 * ev.search_halfwidth + 72 <= E_XS);
}
// inv_dx: 1 / (mean step of the TILE's knots).  The launch-wide mean misplaces the window on a graded axis: where the steps are 6x
// shorter than the mean (the late part of an inspiral -> merger series) a skew of 40 mean steps is 240 rows.
__device__ __forceinline__ bool eval_tile_window(const EvalArgs& ev, long long kT, int rows, int n_i, double smin, double smax, double inv_dx, int* i_a) {
  if (eval_launch_wide_window(ev)) {  // the bound for the whole launch fits the window (or nothing better is known): no reduction needed
    const bool ok = ev.search_halfwidth > 0 && 2 * ev.search_halfwidth + 72 <= E_XS;
    long long ia = kT - ev.i_lo - ev.search_halfwidth - 2;
    ia = ia > n_i - 1 ? n_i - 1 : ia;
    *i_a = (int)(ia < 0 ? 0 : ia);
    return ok;
  }
  const double r_hi = smax * inv_dx, r_lo = smin * inv_dx;
  const double amax = fabs(r_hi) > fabs(r_lo) ? fabs(r_hi) : fabs(r_lo);
  if (!(amax < 1e8)) {
    *i_a = 0;
    return false;
  }
  const long long margin = 3 + (long long)(0.03 * amax);
  long long ia = kT - ev.i_lo - (long long)ceil(r_hi) - margin - 1;
  const long long ib = kT + rows - ev.i_lo - (long long)floor(r_lo) + margin + 1;
  const bool ok = ib - ia <= E_XS;
  ia = ia > n_i - 1 ? n_i - 1 : ia;
  *i_a = (int)(ia < 0 ? 0 : ia);
  return ok;
}
__device__ __forceinline__ void wave_min_max(double& mn, double& mx) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const double a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
}

// The windows [fa, fb) (relative to kT, the knot of the caller's row 0) of one column; win(fl, q) returns C[kT + fl + q][p];
// lds: the staged copies, if from_lds (by value: a struct whose address is taken lives in scratch memory, and a scratch load in
// the loop waits, through vmcnt, for the previous turn's sample to reach memory).
template <class WIN>
__device__ __forceinline__ void eval_windows(const EvalArgs& ev, const bool from_lds, const EvalFromLds lds, int col, double sa, double sb, long long kT,
                                             int fa, int fb, WIN win) {
  const long long n = ev.n;
  {
    const long long cap = n - 3 - kT;  // window starts are 0 .. n - 4
    if (fb > cap) fb = (int)cap;
  }
  if (fa >= fb) return;
  constexpr int FAR = 1 << 24;
  EvalMarch m;
  m.f_lo = kT > FAR ? -FAR : (int)(-kT), m.f_hi = n - 4 - kT > FAR ? FAR : (int)(n - 4 - kT);  // windows 0 .. n - 4, relative
  m.jl_open = n - 2 - kT > FAR ? FAR : (int)(n - 2 - kT);                                      // the last interval claims everything above
  m.jl = fa == m.f_lo ? m.f_lo : fa + 1;                                                       // window 0 serves intervals 0 and 1
  m.jl_end = fb - 1 == m.f_hi ? m.jl_open + 1 : fb + 1;                                        // window n - 4 serves n - 3 and n - 2
  m.sa = sa, m.sb = sb;
  m.n_i = (int)(ev.i_hi - ev.i_lo);
  m.tt = ev.tt;
  m.ldo = ev.ldo;
  const double* bp = ev.x + ev.i_lo;
  int i = -1;
  if (kT + m.jl == 0)
    i = 0;
  else if (from_lds)
    i = lds.first(lds.knot(m.jl), m.sa, m.sb, m.tt, m.n_i);
  if (i < 0) {
    if (from_lds && ev.stats) atomicAdd(ev.stats + 1, 1ull);
    i = eval_lower_bound(bp, m.n_i, m.sa, m.sb, m.tt, ev.x[kT + m.jl], kT + m.jl - ev.i_lo, ev.search_halfwidth);
  }
  if (i >= m.n_i) return;
  m.i = i;
  m.op = ev.out + 2LL * col + (long long)i * ev.ldo;
  const bool interior = fa > m.f_lo && fb - 1 < m.f_hi && fb + 1 < m.jl_open;
  if (from_lds && interior) {
    if (eval_march<true>(m, lds, win, EV_DBG(ev))) return;
  } else if (from_lds) {
    if (eval_march<false>(m, lds, win, EV_DBG(ev))) return;
  }
  if (from_lds && ev.stats) atomicAdd(ev.stats + 1, 1ull);
  eval_march<false>(m, EvalFromGlobal{ev, bp, kT}, win, EV_DBG(ev));
}

template <int ROW_STEP>
__global__ __launch_bounds__(256, 3) void zgemm3m_eval_kernel(const double* __restrict__ A, long long lda, const double* __restrict__ B,
                                                             long long ldb, long long M, int N, int K, int nbm, int nbn,
                                                             int st_rows_log2, const double* __restrict__ col_scale, EvalArgs ev) {
  __shared__ __attribute__((aligned(16))) double2 lds[2 * E_ASZ + 2 * E_BSZ];
  static_assert(sizeof(double2) * (2 * E_ASZ + 2 * E_BSZ) >= sizeof(double2) * E_BM * E_PC, "the parked half tile must fit the operand LDS");
  __shared__ __attribute__((aligned(16))) double t_lds[64 * 16];
  __shared__ __attribute__((aligned(16))) double xk_lds[64];
  __shared__ __attribute__((aligned(16))) double xs_lds[E_XS];
  __shared__ __attribute__((aligned(16))) double2 sk_lds[E_BN];  // (skew_a, skew_b) of the tile's columns
  __shared__ double win_lds[4];  // smallest / largest skew of the tile (eval_tile_window); 1 / mean step of its knots; of its staged abscissae
  double2* As = lds;
  double2* Bs = lds + 2 * E_ASZ;

  // same XCD-aware super-tile map as zgemm3m_mfma_kernel
  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int q = b >> 3;
  const int st_cols_log2 = 6 - st_rows_log2;
  const int nsn = (nbn + (1 << st_cols_log2) - 1) >> st_cols_log2;
  const int S = (q >> 6) * 8 + xcd;
  const int r = q & 63;
  const int bm = ((S / nsn) << st_rows_log2) + (r >> st_cols_log2);
  const int bn = ((S % nsn) << st_cols_log2) + (r & ((1 << st_cols_log2) - 1));
  if (bm >= nbm || bn >= nbn) return;
  const long long m0 = (long long)bm * ROW_STEP;
  const int n0 = bn * E_BN;
  unsigned long long tr_t0 = 0;
  if (EV_TRACE(ev)) tr_t0 = __builtin_readcyclecounter();  // (debug timeline: tools/probes/gemm_eval_trace.py)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // (wave-uniform: the epilogue branches on it around barriers)
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, fk = lane >> 4;

  // ---- what the evaluation will read, requested now and in LDS long before the K loop ends: the power-basis tables and knots of
  // the tile's 64 rows and the window of output abscissae its samples can lie in
  const long long rows_valid = (M - m0) < E_BM ? (M - m0) : E_BM;
  const long long kT = ev.g0 + m0;  // knot of the tile's row 0
  const int n_i = (int)(ev.i_hi - ev.i_lo);
  if (tid < E_BN) {
    // (read in the epilogue from LDS: as global loads there, their s_waitcnt vmcnt(0) also waited for every sample the first half
    // of the tile had just stored to reach memory)
    const int col = n0 + tid < N ? n0 + tid : N - 1;
    const double sa = ev.skew_a ? ev.skew_a[col] : 0.0, sb = ev.skew_b ? ev.skew_b[col] : 0.0;
    sk_lds[tid] = double2{sa, sb};
    if (!eval_launch_wide_window(ev)) {
      // smallest and largest time skew of the tile's columns over its knots (affine in the knot: the ends decide)
      const double s0 = sa * (ev.x[kT] - ev.tt) + sb, s1 = sa * (ev.x[kT + rows_valid - 1] - ev.tt) + sb;
      double mn = s0 < s1 ? s0 : s1, mx = s0 < s1 ? s1 : s0;
      wave_min_max(mn, mx);
      if (tid == 0) win_lds[0] = mn, win_lds[1] = mx;
    }
    if (tid == 0) {  // the tile's own mean step (a graded axis: the launch's mean can be several times off)
      const double span = ev.x[kT + rows_valid - 1] - ev.x[kT];
      win_lds[2] = (rows_valid > 1 && span > 0.0) ? (double)(rows_valid - 1) / span : ev.inv_dx;
    }
  }

  v4d p1[2][2], p2[2][2], p3[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) p1[i][j] = p2[i][j] = p3[i][j] = v4d{0.0, 0.0, 0.0, 0.0};

  const int a_row = tid >> 3, a_k = tid & 7;
  const int b_row = tid >> 6, b_col = tid & 63;
  const double* a_ptr = A + (m0 + a_row) * lda + 2 * a_k;
  const double* b_ptr = B + (long long)b_row * ldb + 2 * (n0 + b_col);
  const bool a_ok0 = (m0 + a_row) < M, a_ok1 = (m0 + a_row + 32) < M;
  const double2 zero2 = {0.0, 0.0};
  double2 ra0, ra1, rb0, rb1;
  const int nk = (K + E_KC - 1) / E_KC;

#define E_LOAD_GLOBAL(kt)                                                                      \
  {                                                                                            \
    const int k0 = (kt)*E_KC;                                                                  \
    const bool kok = (k0 + a_k) < K;                                                           \
    ra0 = (a_ok0 && kok) ? *reinterpret_cast<const double2*>(a_ptr + 2 * k0) : zero2;          \
    ra1 = (a_ok1 && kok) ? *reinterpret_cast<const double2*>(a_ptr + 32 * lda + 2 * k0) : zero2; \
    const double* bp_ = b_ptr + (long long)k0 * ldb;                                           \
    rb0 = *reinterpret_cast<const double2*>(bp_);                                              \
    rb1 = *reinterpret_cast<const double2*>(bp_ + 4 * ldb);                                    \
  }
#define E_STORE_LDS(buf)                                        \
  {                                                             \
    double2* as_w = As + (buf)*E_ASZ + a_row * E_PA + a_k;      \
    as_w[0] = ra0;                                              \
    as_w[32 * E_PA] = ra1;                                      \
    double2* bs_w = Bs + (buf)*E_BSZ + b_row * E_PB + b_col;    \
    bs_w[0] = rb0;                                              \
    bs_w[4 * E_PB] = rb1;                                       \
  }

  E_LOAD_GLOBAL(0);
  E_STORE_LDS(0);
  __syncthreads();
  // (behind the first barrier: the tile's skew range is known to every thread; the copies requested here are in LDS long before the
  // K loop, whose every turn ends in a barrier, is through)
  int i_a;
  const bool wide = eval_launch_wide_window(ev);
  const bool from_lds = eval_tile_window(ev, kT, (int)rows_valid, n_i, wide ? 0.0 : win_lds[0], wide ? 0.0 : win_lds[1], win_lds[2], &i_a);
  if (!from_lds && tid == 0 && ev.stats) atomicAdd(ev.stats, 1ull);
  if (from_lds) {
    if (tid == 0) {  // mean step of the window of abscissae staged below: the first guess of a sample's row (EvalFromLds::first)
      const int i_b = i_a + E_XS - 1 < n_i - 1 ? i_a + E_XS - 1 : n_i - 1;
      const double span = ev.x[ev.i_lo + i_b] - ev.x[ev.i_lo + i_a];
      win_lds[3] = (ev.inv_dx > 0.0 && i_b > i_a && span > 0.0) ? (double)(i_b - i_a) / span : 0.0;
    }
    {
      long long row = tid >> 2;
      if (row > rows_valid - 1) row = rows_valid - 1;
      const double* src = ev.table[kT + row].m + 4 * (tid & 3);
      const double2 v0 = *reinterpret_cast<const double2*>(src), v1 = *reinterpret_cast<const double2*>(src + 2);
      double2* dst = reinterpret_cast<double2*>(t_lds + 16 * (tid >> 2) + 4 * (tid & 3));
      dst[0] = v0, dst[1] = v1;
    }
    if (tid < 64) {
      long long row = tid;
      if (row > rows_valid - 1) row = rows_valid - 1;
      xk_lds[tid] = ev.x[kT + row];
    }
    {
      int i = i_a + tid;
      if (i > n_i - 1) i = n_i - 1;
      xs_lds[tid] = ev.x[ev.i_lo + i];
    }
  }

  // A panel whose columns all lie in its first half (the last one: 17 of 64 columns at cfg3, 9 at cfg5) would leave the waves of the
  // second half without products and the other two with a full load; there the four waves take 16 rows x 32 columns each instead
  // (half the products per SIMD, all four SIMDs busy).
  const bool narrow = n0 + 32 >= N;
  const int rbase = narrow ? wave * 16 : wm * 32, cbase = narrow ? 0 : wn * 32;
  const bool wave_has_columns = n0 + cbase < N;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) E_LOAD_GLOBAL(kt + 1);
    const double2* as = As + buf * E_ASZ + (rbase + fi) * E_PA + fk;
    const double2* bs = Bs + buf * E_BSZ + fk * E_PB + cbase + fi;
#pragma unroll
    for (int kk = 0; wave_has_columns && kk < E_KC / 4; ++kk) {
      const double2 a0 = as[kk * 4];
      const double2 b0 = bs[kk * 4 * E_PB], b1 = bs[kk * 4 * E_PB + 16];
      const double sa0 = a0.x + a0.y, sb0 = b0.x + b0.y, sb1 = b1.x + b1.y;
      p1[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b0.x, p1[0][0], 0, 0, 0);
      p1[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, b1.x, p1[0][1], 0, 0, 0);
      p2[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b0.y, p2[0][0], 0, 0, 0);
      p2[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, b1.y, p2[0][1], 0, 0, 0);
      p3[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa0, sb0, p3[0][0], 0, 0, 0);
      p3[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa0, sb1, p3[0][1], 0, 0, 0);
      if (!narrow) {
        const double2 a1 = as[16 * E_PA + kk * 4];
        const double sa1 = a1.x + a1.y;
        p1[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b0.x, p1[1][0], 0, 0, 0);
        p1[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b1.x, p1[1][1], 0, 0, 0);
        p2[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b0.y, p2[1][0], 0, 0, 0);
        p2[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b1.y, p2[1][1], 0, 0, 0);
        p3[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa1, sb0, p3[1][0], 0, 0, 0);
        p3[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa1, sb1, p3[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) E_STORE_LDS(buf ^ 1);
    __syncthreads();
  }
#undef E_LOAD_GLOBAL
#undef E_STORE_LDS

  unsigned long long tr_t1 = 0;
  if (EV_TRACE(ev)) tr_t1 = __builtin_readcyclecounter();
  // ---- epilogue: recombine and scale, then the tile is parked in the operand LDS half by half (columns 0..31 by the waves wn = 0,
  // then 32..63) and every thread evaluates 8 windows of one column of the parked half
  double2 cv[2][2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + cbase + j * 16 + fi;
    const bool okc = col < N;
    const double sc_r = (okc && col_scale) ? col_scale[2 * col] : 1.0, sc_i = (okc && col_scale) ? col_scale[2 * col + 1] : 1.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const double re = p1[i][j][rr] - p2[i][j][rr];
        const double im = (p3[i][j][rr] - p1[i][j][rr]) - p2[i][j][rr];
        cv[i][j][rr] = double2{re * sc_r, im * sc_i};
      }
  }
  double2* Cs = lds;  // [64][E_PC]
  const int ec = tid & 31, eg = tid >> 5;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (narrow ? h == 0 : wn == h) {  // (a narrow panel: every wave parks its 16 rows of the first half)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (i == 0 || !narrow) Cs[(rbase + i * 16 + fk + 4 * rr) * E_PC + j * 16 + fi] = cv[i][j][rr];
    }
    __syncthreads();
    const int col = n0 + h * 32 + ec;
    if (col < N && !(EV_DBG(ev) & 1)) {
      if (ROW_STEP == 64 && ev.side != nullptr && eg < 6) {
        // rows 0..2 and 61..63 of the tile, for the windows that straddle the tile boundary (side row eg of tile bm)
        const int row = eg < 3 ? eg : 58 + eg;
        if (row < rows_valid) *reinterpret_cast<double2*>(ev.side + ((long long)bm * 6 + eg) * ev.side_ld + 2LL * col) = Cs[row * E_PC + ec];
      }
      int fa = 8 * eg, fb = fa + 8;
      const int f_own = 61, f_rows = (int)rows_valid - 3;  // windows this tile owns / has the rows for
      if (fb > f_own) fb = f_own;
      if (fb > f_rows) fb = f_rows;
      const auto win = [&](int fl, int qq) { return Cs[(fl + qq) * E_PC + ec]; };
      const EvalFromLds src{(lds_cdp)t_lds, (lds_cdp)xk_lds, (lds_cdp)xs_lds, i_a, from_lds ? win_lds[3] : 0.0};
      const double2 sk = sk_lds[h * 32 + ec];
      eval_windows(ev, from_lds, src, col, sk.x, sk.y, kT, fa, fb, win);
    }
    if (h == 0) __syncthreads();
  }
  if (EV_TRACE(ev) && tid == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    unsigned long long* tp = ev.trace + 5LL * b;
    tp[0] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    tp[1] = (unsigned long long)bm | ((unsigned long long)bn << 32);
    tp[2] = tr_t0, tp[3] = tr_t1, tp[4] = __builtin_readcyclecounter();
  }
}

// The windows that straddle two 64-knot tiles (ROW_STEP = 64): f = 64 b + 61 .. 64 b + 63 for boundary b, from the six side rows
// [tile b: rows 61..63][tile b + 1: rows 0..2].  Block = one wave: (the 64 columns of a column panel of the product, boundary) -- the
// same columns as a tile of the product, so that their skews lie as close together and the window of output abscissae is placed the
// same way; tables of the few intervals involved and that window are staged in LDS as in the product's epilogue.
__global__ __launch_bounds__(64) void spline_straddle_eval_kernel(int N, int n_row_tiles, int n_col_panels, EvalArgs ev, long long M) {
  __shared__ __attribute__((aligned(16))) double2 w_lds[6][64];
  __shared__ __attribute__((aligned(16))) double t_lds[8 * 16];
  __shared__ __attribute__((aligned(16))) double xk_lds[8];
  __shared__ __attribute__((aligned(16))) double xs_lds[E_XS];
  const int tid = threadIdx.x;
  const int bnd = blockIdx.x / n_col_panels;  // (one-dimensional grid: boundary-major, the panels of a boundary next to each other)
  const int col = (blockIdx.x - bnd * n_col_panels) * 64 + tid;
  if (bnd + 1 >= n_row_tiles) return;
  const long long kT = ev.g0 + 64LL * bnd + 61;  // knot of side row 0 of this boundary
  const int n_i = (int)(ev.i_hi - ev.i_lo);
  const long long k_last = ev.g0 + M - 1;
  const int colc = col < N ? col : N - 1;
  const double sa_t = ev.skew_a ? ev.skew_a[colc] : 0.0, sb_t = ev.skew_b ? ev.skew_b[colc] : 0.0;
  double smin = 0.0, smax = 0.0;
  if (!eval_launch_wide_window(ev)) {
    // skew range of the block's 64 columns over the six knots of the boundary
    const double s0 = sa_t * (ev.x[kT] - ev.tt) + sb_t, s1 = sa_t * (ev.x[kT + 5 > k_last ? k_last : kT + 5] - ev.tt) + sb_t;
    smin = s0 < s1 ? s0 : s1, smax = s0 < s1 ? s1 : s0;
    wave_min_max(smin, smax);
  }
  // the local mean step, over the 64 knots around the boundary (six knots alone are too few on a jittered axis)
  double inv_dx_loc = ev.inv_dx;
  {
    const long long ka = kT - 29 < ev.g0 ? ev.g0 : kT - 29, kb = kT + 34 > k_last ? k_last : kT + 34;
    const double span = ev.x[kb] - ev.x[ka];
    if (kb > ka && span > 0.0) inv_dx_loc = (double)(kb - ka) / span;
  }
  int i_a;
  const bool from_lds = eval_tile_window(ev, kT, 6, n_i, smin, smax, inv_dx_loc, &i_a);
  if (!from_lds && tid == 0 && ev.stats) atomicAdd(ev.stats, 1ull);
  double inv_dx_win = 0.0;
  if (from_lds) {
    {
      const int i_b = i_a + E_XS - 1 < n_i - 1 ? i_a + E_XS - 1 : n_i - 1;
      const double span = ev.x[ev.i_lo + i_b] - ev.x[ev.i_lo + i_a];
      if (ev.inv_dx > 0.0 && i_b > i_a && span > 0.0) inv_dx_win = (double)(i_b - i_a) / span;
    }
    {  // 8 rows x 8 pairs of table words
      long long jj = kT + (tid >> 3);
      if (jj > k_last) jj = k_last;
      reinterpret_cast<double2*>(t_lds)[tid] = *reinterpret_cast<const double2*>(ev.table[jj].m + 2 * (tid & 7));
    }
    if (tid < 8) {
      long long jj = kT + tid;
      if (jj > k_last) jj = k_last;
      xk_lds[tid] = ev.x[jj];
    }
#pragma unroll
    for (int r = 0; r < E_XS / 64; ++r) {
      int i = i_a + tid + 64 * r;
      if (i > n_i - 1) i = n_i - 1;
      xs_lds[tid + 64 * r] = ev.x[ev.i_lo + i];
    }
    __syncthreads();
  }
  if (col >= N) return;
  int fb = 3;
  {
    const long long f_rows = ev.g0 + M - 3 - kT;
    if (fb > f_rows) fb = (int)f_rows;
  }
  const double* s0 = ev.side + ((long long)bnd * 6 + 3) * ev.side_ld + 2LL * col;        // tile bnd, rows 61..63
  const double* s1 = ev.side + ((long long)(bnd + 1) * 6 + 0) * ev.side_ld + 2LL * col;  // tile bnd + 1, rows 0..2
  // the six rows of this column sit in LDS (indexed by the march: a register array would live in scratch memory)
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    w_lds[e][tid] = *reinterpret_cast<const double2*>(s0 + e * ev.side_ld);
    w_lds[3 + e][tid] = *reinterpret_cast<const double2*>(s1 + e * ev.side_ld);
  }
  const auto win = [&](int fl, int qq) { return w_lds[fl + qq][tid]; };
  const EvalFromLds src{(lds_cdp)t_lds, (lds_cdp)xk_lds, (lds_cdp)xs_lds, i_a, inv_dx_win};
  eval_windows(ev, from_lds, src, col, sa_t, sb_t, kT, 0, fb, win);
}

hipError_t launch_zgemm3m_eval(hipStream_t stream, const double* A, long long lda, const double* B, long long ldb, long long M, int N,
                               int K, const double* col_scale, const SplineEval& e) {
  if (M <= 0 || N <= 0 || e.i_hi <= e.i_lo) return hipSuccess;
  if (M < 4) return hipErrorInvalidValue;
  EvalArgs ev;
  ev.table = e.table, ev.x = e.x, ev.skew_a = e.skew_a, ev.skew_b = e.skew_b, ev.tt = e.tt, ev.g0 = e.g0, ev.n = e.n_knots;
  ev.i_lo = e.i_lo, ev.i_hi = e.i_hi, ev.out = e.out, ev.ldo = e.ldo, ev.search_halfwidth = e.search_halfwidth;
  ev.inv_dx = e.inv_dx;
  ev.side = e.side, ev.side_ld = e.side_ld;
  ev.stats = e.stats;
  ev.dbg = 0;
#if BMS_PROBES
  static const int dbg_env = BMS_PROBE_ENV("SCRI_AMD_GEMM_EVAL_DBG") ? atoi(BMS_PROBE_ENV("SCRI_AMD_GEMM_EVAL_DBG")) : 0;
  ev.dbg = dbg_env;
#endif
  const int step = (e.step == 61 || e.step == 64) ? e.step : (e.side ? 64 : 61);  // (e.step: the context's GEMM_EVAL_STEP option, 0 = automatic)
  if (step == 64 && !e.side) return hipErrorInvalidValue;
  if (step == 61) ev.side = nullptr;
  const int nbm = step == 61 ? (int)((M - 3 + 60) / 61) : (int)((M + 63) / 64);
  const int nbn = (N + E_BN - 1) / E_BN;
  static const int st_env = BMS_PROBE_ENV("SCRI_AMD_ZGEMM_ST_ROWS_LOG2") ? atoi(BMS_PROBE_ENV("SCRI_AMD_ZGEMM_ST_ROWS_LOG2")) : -1;
  const int st_rows_log2 = (st_env >= 0 && st_env <= 6) ? st_env : (nbm >= 512 ? 6 : 5);
  const int sr = 1 << st_rows_log2, sc = 64 >> st_rows_log2;
  const long long n_super = (long long)((nbm + sr - 1) / sr) * ((nbn + sc - 1) / sc);
  const long long grid = ((n_super + 7) / 8) * 8 * 64;
  ev.trace = nullptr;
#if BMS_PROBES
  const char* trace_path = BMS_PROBE_ENV("SCRI_AMD_GEMM_EVAL_TRACE");
  if (trace_path) {
    if (hipMalloc(&ev.trace, 40 * (size_t)grid) != hipSuccess) ev.trace = nullptr;
    if (ev.trace) (void)hipMemsetAsync(ev.trace, 0, 40 * (size_t)grid, stream);
  }
#endif
  if (step == 61)
    hipLaunchKernelGGL(zgemm3m_eval_kernel<61>, dim3((unsigned)grid), dim3(256), 0, stream, A, lda, B, ldb, M, N, K, nbm, nbn, st_rows_log2,
                       col_scale, ev);
  else {
    hipLaunchKernelGGL(zgemm3m_eval_kernel<64>, dim3((unsigned)grid), dim3(256), 0, stream, A, lda, B, ldb, M, N, K, nbm, nbn, st_rows_log2,
                       col_scale, ev);
    // one block per (tile boundary, column panel), panels fastest, in a one-dimensional grid (a grid's y extent stops at 65 535:
    // a chunk of more than 4.19 M rows has more boundaries than that)
    const long long n_straddle = (long long)(nbm - 1) * nbn;
    if (n_straddle > 0x7fffffffLL) return hipErrorInvalidValue;
    if (nbm > 1 && !(EV_DBG(ev) & 4))
      hipLaunchKernelGGL(spline_straddle_eval_kernel, dim3((unsigned)n_straddle), dim3(64), 0, stream, N, nbm, nbn, ev, M);
  }
#if BMS_PROBES
  if (ev.trace) {  // (debug: blocks the host)
    std::vector<unsigned long long> h(5 * (size_t)grid);
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h.data(), ev.trace, 40 * (size_t)grid, hipMemcpyDeviceToHost);
    (void)hipFree(ev.trace);
    if (FILE* f = fopen(trace_path, "wb")) {
      fwrite(h.data(), 8, h.size(), f);
      fclose(f);
    }
  }
#endif
  return hipGetLastError();
}

long long zgemm3m_eval_side_rows(long long M) { return 6 * ((M + 63) / 64); }

}  // namespace bms
