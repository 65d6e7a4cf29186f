// Separable synthesis WITH the spline evaluation in it: boost-free transformations of WaveformModes (scri/waveform_grid.py:462-484 with
// beta = 0, then the per-pixel splines of :574-588).  VERDICT r4 item 1: "no marching pass over the grid".
//
// The whole collocation solve of the time spline runs on the MODES (bspline_solve_modes_kernel: it commutes with the synthesis, which
// is linear along the columns with time-independent coefficients), so the synthesis of row k of the solved modes IS row k of the grid
// of B-spline coefficients C[k][pixel], and a sample of pixel p at the output time u' is four taps
//     s_p(u) = sum_{q < 4} b_q(t) C[f + q][p],   u = x_i + skew_b[p],  u in [x_{f+1}, x_{f+2}),  t = u - x_{f+1}
// through the interval's 4 x 4 power-basis table -- no recurrence.  synthesis_split_kernel + bspline_backward_eval_kernel wrote the
// coefficient grid to HBM and read it back (2 x 20.8 KB per time step at l <= 16: 0.96 of the 2.7 ms of the transformation); here a
// workgroup walks CONSECUTIVE knots, one per trip, and the lanes of the phi waves -- each lane produces the same <= 10 pixels in
// every trip -- keep the last three coefficient rows of their own pixels in LDS (lane-private slots: no barrier, 3 x 16 B per pixel
// = 66 KB at 37 x 37), evaluate the window (C[k-3], C[k-2], C[k-1], C[k]) as soon as row k exists and store SAMPLES only.
// Without a boost a pixel's time skew is a constant (skew_a = 0): the output rows of a pixel follow its knots at a fixed distance.
//
//   theta waves: as in synthesis_split_kernel, for one row: thread (list of modes, rings j and j + n_pair) keeps sLambda_lm of its
//                list in registers, reads a_lm from LDS (staged one trip ahead), flushes F_m(theta_j) to LDS after every m.
//   phi waves:   wave w owns rings 16 w .. 16 w + 15 as the A rows of its MFMA tiles (folded DFT over m, twiddles in registers);
//                its results are coefficients; per pixel: read the three older coefficients of the pixel, put the new one in
//                the oldest one's place, evaluate every output sample of the pixel that lies in the interval the window serves
//                (a cursor per pixel; the abscissae come from a window of the time axis staged in LDS per segment), subtract
//                the pixel's offset (h / sigma: the B-spline coefficients of a constant are that constant), store.
//   A workgroup walks segments of <= 512 windows (a segment restarts with a three-row run-in: 0.6 %).
// Not-a-knot ends: the two end intervals are the same cubic as their neighbours (x_1 and x_{n-2} are no knots), so window 0 also serves
// everything below x_1 and window n - 4 everything above x_{n-2}, with the neighbour's table.
// HBM traffic = the algorithmic minimum of a synthesis: 16 n_modes bytes read, 16 n_pix bytes of SAMPLES written per step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace bms {

typedef double v4d_t __attribute__((ext_vector_type(4)));
typedef const double __attribute__((address_space(3))) * xw_p;  // (typed LDS pointer: through a generic one the compiler emits flat loads)

constexpr int SE_MAX_WAVES = 8;

// (the time axis, its tables and the output are kernel parameters of their own, `__restrict__`: inside a struct the compiler must assume
// that the stores of samples alias them and reads the wave-uniform table words with vector loads -- whose waits then drain the stores)
struct SynEvalArgs {
  const double* skew_b;       // per grid pixel
  long long g0;               // knot of row 0 of A
  long long n;                // knots of the whole series
  long long i_lo, i_hi;       // output rows (indices into x); out row 0 = i_lo
  long long ldo;
  double s_min, s_max;        // range of skew_b over the pixels
  int seg;                    // windows per segment
  int xw;                     // output abscissae staged per segment (512 or 1024)
  int ev_first[SE_MAX_WAVES], ev_count[SE_MAX_WAVES];  // evaluation: the pixels [first, first + count) of every wave (count <= 256)
  int knock;                  // probe builds (timing only, results wrong): 1 no evaluation, 4 no theta products, 8 no sample stores
  unsigned long long* trace;  // probe builds: clock stamps of workgroup 0, steps 8 .. 23: [step][wave][point] (SE_TRACE_POINTS points)
};
#if BMS_PROBES
constexpr int SE_TRACE_POINTS = 6, SE_TRACE_STEPS = 16;
#define SE_KNOCK(ev) ((ev).knock)
#define SE_STAMP(ev, st, wave, lane, pt)                                                                                         \
  if ((ev).trace && blockIdx.x == 0 && (lane) == 0 && (st) >= 8 && (st) < 8 + SE_TRACE_STEPS)                                     \
    (ev).trace[(((st)-8) * SE_MAX_WAVES + (wave)) * SE_TRACE_POINTS + (pt)] = __builtin_readcyclecounter();
#else
#define SE_KNOCK(ev) 0
#define SE_STAMP(ev, st, wave, lane, pt)
#endif

// first i in [0, n_i) with bp[i] + s >= y (n_i if none)
__device__ __forceinline__ int se_first_sample(const double* __restrict__ bp, int n_i, double s, double y) {
  int lo = 0, hi = n_i;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (bp[mid] + s < y)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

template <int NT, int LEN, int RR>
__global__ __launch_bounds__(64 * SE_MAX_WAVES, 1) void synthesis_eval_kernel(const double* __restrict__ A, long long lda, long long n_rows, SynGeom g,
                                                                             int nph, const double* __restrict__ Tsyn,
                                                                             const int* __restrict__ meta,
                                                                             const double* __restrict__ xg /* knots = output abscissae before the skew */,
                                                                             const BsplineTable* __restrict__ tabg /* per knot */,
                                                                             double* __restrict__ outg, SynEvalArgs ev) {
  // pitch (complex) of an F_m row: the next odd number (synthesis_split_kernel's conflict-free multiple of 16 changed nothing there, and
  // the four coefficient rows leave no room for it)
  const int PJ = g.n_theta | 1;
  // (16-byte aligned, and NO static LDS beside it: two ints placed in front of the dynamic region put every 16-byte access of the kernel
  // on an 8-byte boundary -- SQ_LDS_UNALIGNED_STALL 1.8e9 of 4.2e9 wave-cycles, everything 5x slower)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int fsz = (2 * g.L + 1) * PJ;
  const int na = g.n_modes;
  const int npix = g.n_theta * g.n_phi;
  double2* Fs = reinterpret_cast<double2*>(lds);  // [2 buffers][2L+1][PJ]
  double2* abuf = Fs + 2 * fsz;                   // [2 buffers][n_modes]
  // [RR][n_pix]: the newest coefficient rows of every pixel (slot = row mod RR).  RR = 4: the window of a step and nothing else, so the
  // step needs a second barrier before the next row replaces the oldest one; RR = 5: the row being produced has a slot of its own and
  // a wave goes from its role's work straight into its share of the evaluation -- one barrier per step (grids up to 37 x 37 at l <= 16)
  double2* ring = abuf + 2 * na;
  double* xw = reinterpret_cast<double*>(ring + RR * npix);  // [ev.xw]
  const int XW = ev.xw;
  int* metal = reinterpret_cast<int*>(xw + XW);             // [n_lists][LEN]
  int& seg_ia = metal[g.n_lists * LEN];        // first output row (relative to i_lo) of the staged abscissae
  int& seg_wide = metal[g.n_lists * LEN + 1];  // the segment's samples do not fit the staged window: its abscissae are read from global memory
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool is_phi = wave < nph;  // waves 0 .. nph - 1 (one per SIMD) carry the MFMA work
  const int th = wave - nph;       // rank among the theta waves
  for (int e = tid; e < g.n_lists * LEN; e += blockDim.x) {
    const int mt = meta[e];
    metal[e] = ((mt & 1023) * 16) | (((mt >> 10) & 63) << 16) | ((mt & (1 << 16)) ? 1 << 24 : 0);
  }
  const int n_i = (int)(ev.i_hi - ev.i_lo);
  const double* __restrict__ bp = xg + ev.i_lo;
  const double inf = __builtin_huge_val();
  // windows of this launch: f = g0 .. g0 + n_rows - 4 (window f = rows f .. f + 3)
  const long long fA = ev.g0, fB = ev.g0 + n_rows - 3;
  const long long n_seg = (fB - fA + ev.seg - 1) / ev.seg;

  // ---- theta role: thread (list, rings j and j + n_pair), sLambda_lm of the list in registers
  const int tt = 64 * th + lane;
  const int npair = (g.n_theta + 1) / 2;
  const bool active = !is_phi && tt < g.n_lists * npair;
  const int li = active ? tt / npair : 0, j = active ? tt - li * npair : 0;
  const bool second = active && j + npair < g.n_theta;
  double treg[LEN], tre2[LEN];
#pragma unroll
  for (int e = 0; e < LEN; ++e) treg[e] = tre2[e] = 0.0;
  if (!is_phi) {
    int mts[LEN];
#pragma unroll
    for (int e = 0; e < LEN; ++e) mts[e] = meta[li * LEN + e];
#pragma unroll
    for (int e = 0; e < LEN; ++e) {
      const bool ok = active && (mts[e] & (1 << 17));
      treg[e] = ok ? Tsyn[(long long)(mts[e] & 1023) * g.n_theta + j] : 0.0;
      tre2[e] = (ok && second) ? Tsyn[(long long)(mts[e] & 1023) * g.n_theta + j + npair] : 0.0;
    }
  }
  const int2* ml = reinterpret_cast<const int2*>(metal) + li * (LEN / 2);
  const bool h1 = !is_phi && tt < na;  // this thread carries element tt of the next row of modes

  // ---- phi role: twiddles; the lane's pixels: e = 2 v (ring 16 w + fk + 4 v, column k = fi + 1), 2 v + 1 (column n_phi - k); 8, 9: the
  // side product's (ring 16 w + 4 ((lane >> 2) & 3) + (lane >> 4), columns kx and n_phi - kx); -1: none
  const int fi = lane & 15, fk = lane >> 4;
  double cs[4], sn[4], cx[4], sx[4];
  const int jx = lane & 3;  // column of the side product: k = 0, 17, 18, 19
  const int kx = jx == 0 ? 0 : 16 + jx;
  int pix[10], ringi[5], fa = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) cs[s] = sn[s] = cx[s] = sx[s] = 0.0;
#pragma unroll
  for (int e = 0; e < 10; ++e) pix[e] = -1;
#pragma unroll
  for (int e = 0; e < 5; ++e) ringi[e] = 0;
  if (is_phi) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int m = 4 * s + fk + 1;
      double sv, cv;
      sincospi(2.0 * (double)(((long long)m * (fi + 1)) % g.n_phi) / (double)g.n_phi, &sv, &cv);
      const bool ok = fi + 1 < g.nk && m <= g.L;
      cs[s] = ok ? cv : 0.0;
      sn[s] = ok ? sv : 0.0;
      sincospi(2.0 * (double)(((long long)m * kx) % g.n_phi) / (double)g.n_phi, &sv, &cv);
      const bool okx = kx < g.nk && m <= g.L;
      cx[s] = okx ? cv : 0.0;
      sx[s] = okx ? sv : 0.0;
    }
    const int q = 16 * wave + fi;  // A row of this lane: ring q
    fa = q < g.n_theta ? q : 0;
    const int kk = fi + 1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int jv = 16 * wave + fk + 4 * v;
      const bool ok = jv < g.n_theta && kk < g.nk;
      ringi[v] = jv < g.n_theta ? jv : 0;
      pix[2 * v] = ok ? jv * g.n_phi + kk : -1;
      pix[2 * v + 1] = (ok && 2 * kk != g.n_phi) ? jv * g.n_phi + g.n_phi - kk : -1;
    }
    const int qs = 16 * wave + 4 * ((lane >> 2) & 3) + (lane >> 4);
    const bool ok = qs < g.n_theta && kx < g.nk;
    ringi[4] = qs < g.n_theta ? qs : 0;
    pix[8] = ok ? qs * g.n_phi + kx : -1;
    pix[9] = (ok && kx >= 1 && 2 * kx != g.n_phi) ? qs * g.n_phi + g.n_phi - kx : -1;
  }

  // ---- evaluation (every thread, whatever its role).  Wave w owns the pixels [ev_first, ev_first + ev_count), lane l the pixels
  // ev_first + l + 64 e: the shares follow what the waves' roles leave them -- a theta wave that shares its SIMD with a phi wave (waves
  // nph + 1 .. nph + 3 at eight waves: SIMD = wave mod 4) is the last to finish its row and takes the fewest.
  constexpr int EP = 4;
  const int nthr = blockDim.x;
  const int ev_first = ev.ev_first[wave], ev_count = ev.ev_count[wave];
  double sbv[EP];
  int icur[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    sbv[e] = lane + 64 * e < ev_count ? ev.skew_b[ev_first + lane + 64 * e] : 0.0;
    icur[e] = 0;
  }

  for (long long sg = blockIdx.x; sg < n_seg; sg += gridDim.x) {
    // ---------------------------------------------------------------------------------------------- head of a segment
    const long long F0 = fA + sg * ev.seg;
    const long long F1 = F0 + ev.seg < fB ? F0 + ev.seg : fB;
    const long long k_first = F0;              // rows F0 .. F1 + 2 (windows F0 .. F1 - 1)
    const int n_seg_rows = (int)(F1 + 3 - F0);
    // window f serves the samples in [x[f + 1], x[f + 2]); window 0 everything below too, window n - 4 everything above
    const double seg_lo = F0 == 0 ? -inf : xg[F0 + 1];
    __syncthreads();  // (the previous segment's readers of xw, seg_ia, abuf, F and the ring are through)
    if (tid == 0) {
      // the first sample of ANY pixel at or above lo is at row first(s_max, lo) or later: staged from one row below that ...
      int ia = se_first_sample(bp, n_i, ev.s_max, seg_lo) - 1;
      ia = ia < 0 ? 0 : ia;
      seg_ia = ia;
      // ... and the last one below the segment's upper end before row first(s_min, hi) (+ 1: a cursor rests on the row after its last sample)
      const double hi = F1 - 1 == ev.n - 4 ? inf : xg[F1 + 1];
      const int ib = hi == inf ? n_i : se_first_sample(bp, n_i, ev.s_min, hi);
      seg_wide = ib + 2 - ia > XW ? 1 : 0;
    }
    for (int e = tid; e < na; e += nthr) abuf[e] = *reinterpret_cast<const double2*>(A + (k_first - ev.g0) * lda + 2LL * e);
    __syncthreads();
    const int i_a = seg_ia;
    const bool wide = seg_wide != 0;
    for (int e = tid; e < XW; e += nthr) {
      int i = i_a + e;
      if (i > n_i - 1) i = n_i - 1;
      xw[e] = bp[i < 0 ? 0 : i];
    }
    __syncthreads();

    // WIN: the abscissae of the segment's samples are read from the staged window (LDS only: a global load in the loop would put a
    // wait for the previous samples' stores into every turn); otherwise -- a segment wider than the window: skews spread over more
    // rows than the host's bound promised -- from global memory, slowly.
    auto run_segment = [&](auto win_tag) {
      constexpr bool WIN = decltype(win_tag)::value;
      const xw_p xwl = (xw_p)xw;
      auto xs = [&](int i) -> double {
        if (WIN) return xwl[i - i_a];
        return bp[i < n_i ? i : n_i - 1];
      };
      // cursors: the first sample of each pixel that the segment serves; it is at or above row i_a
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        if (lane + 64 * e >= ev_count) continue;
        int lo = i_a, hi = n_i;
        if (F0 == 0) lo = hi = 0;
        if (WIN && hi > i_a + XW - 2) hi = i_a + XW - 2;  // (a window that holds every sample of the segment holds its first ones)
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (xs(mid) + sbv[e] < seg_lo)
            lo = mid + 1;
          else
            hi = mid;
        }
        icur[e] = lo;
      }
      double2 x1{0.0, 0.0};
      // step s:  theta: F(row s) -> Fs[s & 1];  phi: C(row s - 1) from Fs[(s - 1) & 1] -> ring[(s - 1) mod RR];
      //          everybody: the window that ends at row s - 1 (RR = 4: behind a barrier) or s - 2 (RR = 5: no barrier in between)
      constexpr int LAG = RR == 4 ? 1 : 2;
      for (int st = 0; st <= n_seg_rows + LAG - 1; ++st) {
        SE_STAMP(ev, st, wave, lane, 0)
        // the window this step evaluates (rows k - 3 .. k, k = k_first + st - LAG) and the interval it serves: wave-uniform, scalar
        // loads -- requested HERE, a role's work ahead of their use (a table row is a scalar-cache miss every step)
        const bool do_eval = st >= 3 + LAG;
        const long long f = k_first + st - LAG - 3, jj = do_eval ? f + 1 : ev.g0 + 1;
        const double xlo = xg[jj];
        const double xhi = f == ev.n - 4 ? inf : xg[jj + 1];
        const double* __restrict__ tb = tabg[jj].m;
        const double m00 = tb[0], m01 = tb[1], m02 = tb[2], m03 = tb[3], m10 = tb[4], m11 = tb[5], m12 = tb[6], m13 = tb[7];
        const double m20 = tb[8], m21 = tb[9], m22 = tb[10], m23 = tb[11], m30 = tb[12], m31 = tb[13], m32 = tb[14], m33 = tb[15];
        if (!is_phi) {
          // ------------------------------------------------------------------------------------------ theta waves
          if (st < n_seg_rows) {
            if (h1) {  // the next row's modes (at the last row: that row again)
              const long long kn = st + 1 < n_seg_rows ? k_first + st + 1 : k_first + st;
              x1 = *reinterpret_cast<const double2*>(A + (kn - ev.g0) * lda + 2LL * tt);
            }
            double2* Fb = Fs + (st & 1) * fsz;
            const char* a0 = reinterpret_cast<const char*>(abuf + (st & 1) * na);
            double r0 = 0.0, i0 = 0.0, s0 = 0.0, k0 = 0.0;  // ring j (r, i), ring j + npair (s, k)
            static_assert(LEN % 2 == 0, "entries are walked in pairs");
            double2 u[2][2];
            int2 mq[3];
            auto read2 = [&](const int2& mt, double2(&d)[2]) {
              d[0] = *reinterpret_cast<const double2*>(a0 + (mt.x & 0xffff));
              d[1] = *reinterpret_cast<const double2*>(a0 + (mt.y & 0xffff));
            };
            auto entry = [&](int e, int mt, const double2& xv) {
              r0 = fma(treg[e], xv.x, r0);
              i0 = fma(treg[e], xv.y, i0);
              s0 = fma(tre2[e], xv.x, s0);
              k0 = fma(tre2[e], xv.y, k0);
              if (mt & (1 << 24)) {
                const int at = ((mt >> 16) & 63) * PJ + j;
                if (active) Fb[at] = double2{r0, i0};
                if (second) Fb[at + npair] = double2{s0, k0};
                r0 = i0 = s0 = k0 = 0.0;
              }
            };
            mq[0] = ml[0];
            if (LEN > 2) mq[1] = ml[1];
            read2(mq[0], u[0]);
#pragma unroll
            for (int pi = 0; pi < LEN / 2; ++pi) {
              if (SE_KNOCK(ev) & 4) break;
              const int b = pi & 1;
              if (pi + 2 < LEN / 2) mq[(pi + 2) % 3] = ml[pi + 2];
              if (pi + 1 < LEN / 2) read2(mq[(pi + 1) % 3], u[b ^ 1]);
              entry(2 * pi, mq[pi % 3].x, u[b][0]);
              entry(2 * pi + 1, mq[pi % 3].y, u[b][1]);
              asm volatile("" : "+v"(r0), "+v"(i0), "+v"(s0), "+v"(k0)::"memory");
            }
          }
        } else if (st >= 1 && st <= n_seg_rows) {
          // -------------------------------------------------------------------------------------------- phi waves
          const double2* Fb = Fs + ((st - 1) & 1) * fsz;
          v4d_t ure{0.0, 0.0, 0.0, 0.0}, uim = ure, vre = ure, vim = ure;
          double xur = 0.0, xui = 0.0, xvr = 0.0, xvi = 0.0;
          const int ksm = (g.L + 3) / 4;  // k steps that carry an m <= L at all
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s >= ksm) break;
            const int m = 4 * s + fk + 1;
            const int mm = m <= g.L ? m : 0;  // (rows beyond L: their twiddles are zero)
            const double2 fp = Fb[fa + (g.L + mm) * PJ], fm = Fb[fa + (g.L - mm) * PJ];
            const double px = fp.x + fm.x, py = fp.y + fm.y, qx = fm.y - fp.y, qy = fp.x - fm.x;
            ure = __builtin_amdgcn_mfma_f64_16x16x4f64(px, cs[s], ure, 0, 0, 0);
            uim = __builtin_amdgcn_mfma_f64_16x16x4f64(py, cs[s], uim, 0, 0, 0);
            vre = __builtin_amdgcn_mfma_f64_16x16x4f64(qx, sn[s], vre, 0, 0, 0);
            vim = __builtin_amdgcn_mfma_f64_16x16x4f64(qy, sn[s], vim, 0, 0, 0);
            xur = __builtin_amdgcn_mfma_f64_4x4x4f64(px, cx[s], xur, 0, 0, 0);
            xui = __builtin_amdgcn_mfma_f64_4x4x4f64(py, cx[s], xui, 0, 0, 0);
            xvr = __builtin_amdgcn_mfma_f64_4x4x4f64(qx, sx[s], xvr, 0, 0, 0);
            xvi = __builtin_amdgcn_mfma_f64_4x4x4f64(qy, sx[s], xvi, 0, 0, 0);
          }
          double2* rk = ring + (int)((k_first + st - 1) % RR) * npix;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            if (pix[2 * v] < 0) continue;
            const double2 f0 = Fb[g.L * PJ + ringi[v]];
            const double ux = f0.x + ure[v], uy = f0.y + uim[v];
            rk[pix[2 * v]] = double2{ux + vre[v], uy + vim[v]};
            if (pix[2 * v + 1] >= 0) rk[pix[2 * v + 1]] = double2{ux - vre[v], uy - vim[v]};
          }
          if (pix[8] >= 0) {
            const double2 f0 = Fb[g.L * PJ + ringi[4]];
            const double ux = f0.x + xur, uy = f0.y + xui;
            rk[pix[8]] = double2{ux + xvr, uy + xvi};
            if (pix[9] >= 0) rk[pix[9]] = double2{ux - xvr, uy - xvi};
          }
        }
        SE_STAMP(ev, st, wave, lane, 1)
        if (RR == 4) __syncthreads();  // F of row st, the coefficients of row st - 1 complete
        SE_STAMP(ev, st, wave, lane, 2)
        // ---------------------------------------------------------------------------------------------- evaluation
        if (do_eval && !(SE_KNOCK(ev) & 1)) {
          const double2* r0p = ring + (int)(f % RR) * npix + ev_first + lane;
          const double2* r1p = ring + (int)((f + 1) % RR) * npix + ev_first + lane;
          const double2* r2p = ring + (int)((f + 2) % RR) * npix + ev_first + lane;
          const double2* r3p = ring + (int)((f + 3) % RR) * npix + ev_first + lane;
          // what the samples of this turn read is requested for two pixels at a time (the four coefficients and the abscissae at the
          // cursor: one exposed LDS round trip per pair instead of three per pixel)
#pragma unroll
          for (int e0 = 0; e0 < EP; e0 += 2) {
            if (64 * e0 >= ev_count) break;  // (wave-uniform)
            double2 q0[2], q1[2], q2[2], q3[2];
            double xi[2], xj[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const int o = lane + 64 * (e0 + d) < ev_count ? 64 * (e0 + d) : 0;
              q0[d] = r0p[o], q1[d] = r1p[o], q2[d] = r2p[o], q3[d] = r3p[o];
              const int ic = icur[e0 + d] < n_i - 1 ? icur[e0 + d] : n_i - 1;
              xi[d] = xs(ic);
              xj[d] = xs(ic + 1 < n_i ? ic + 1 : ic);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              if (lane + 64 * (e0 + d) >= ev_count) continue;
              const int p = ev_first + lane + 64 * (e0 + d);
              const double sb = sbv[e0 + d];
              int ic = icur[e0 + d];
              double xc = xi[d], xn = xj[d];
              while (ic < n_i) {
                if (!(xc + sb < xhi)) break;
                const double t = (xc - xlo) + sb;  // ((x_i - x_j) + skew: the small difference stays exact, as in the marching kernels)
                const double b0 = fma(fma(fma(m30, t, m20), t, m10), t, m00);
                const double b1 = fma(fma(fma(m31, t, m21), t, m11), t, m01);
                const double b2 = fma(fma(fma(m32, t, m22), t, m12), t, m02);
                const double b3 = fma(fma(fma(m33, t, m23), t, m13), t, m03);
                double2 v;
                v.x = fma(b3, q3[d].x, fma(b2, q2[d].x, fma(b1, q1[d].x, b0 * q0[d].x)));
                v.y = fma(b3, q3[d].y, fma(b2, q2[d].y, fma(b1, q1[d].y, b0 * q0[d].y)));
                if (!(SE_KNOCK(ev) & 8) || v.x == 1.2345e300) *reinterpret_cast<double2*>(outg + (long long)ic * ev.ldo + 2LL * p) = v;
                ++ic;
                xc = xn;
                xn = xs(ic + 1 < n_i ? ic + 1 : ic);
              }
              icur[e0 + d] = ic;
            }
          }
        }
        SE_STAMP(ev, st, wave, lane, 3)
        if (h1 && st < n_seg_rows) abuf[((st + 1) & 1) * na + tt] = x1;
        SE_STAMP(ev, st, wave, lane, 4)
        __syncthreads();  // RR = 4: the windows are read, the next row may replace the oldest one; RR = 5: F of row st, the coefficients
                          // of row st - 1 complete; the next row's modes are in place
        SE_STAMP(ev, st, wave, lane, 5)
      }
    };
    if (wide)
      run_segment(std::false_type{});
    else
      run_segment(std::true_type{});
  }
}

// y[k][col[j]] -= c[j] * y[k][one_col]: the inhomogeneous term of h / sigma on the MODES (its per-direction values are the synthesis
// of the modes c: sum c_lm sY_lm(R_p), pixel_math.h), so that the grid kernel need not carry a per-pixel offset
__global__ __launch_bounds__(256) void sub_const_modes_kernel(double* __restrict__ Y, long long ld, long long n_rows, int n, const int* __restrict__ col,
                                                              const double* __restrict__ c, int one_col) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_rows * n) return;
  const long long k = e / n;
  const int jc = (int)(e - k * n);
  double* row = Y + k * ld;
  const double2 one = *reinterpret_cast<const double2*>(row + 2LL * one_col);
  double2 v = *reinterpret_cast<double2*>(row + 2LL * col[jc]);
  const double cr = c[2 * jc], ci = c[2 * jc + 1];
  v.x -= cr * one.x - ci * one.y;
  v.y -= cr * one.y + ci * one.x;
  *reinterpret_cast<double2*>(row + 2LL * col[jc]) = v;
}
hipError_t launch_sub_const_modes(hipStream_t stream, double* Y, long long ld, long long n_rows, int n, const int* col, const double* c, int one_col) {
  if (n_rows <= 0 || n <= 0) return hipSuccess;
  const long long total = n_rows * n;
  hipLaunchKernelGGL(sub_const_modes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, Y, ld, n_rows, n, col, c, one_col);
  return hipGetLastError();
}

// Does the fused kernel take the shape, with how many ring rows, staged abscissae and how much LDS?  (the lists and tables are
// synthesis_split_plan's: g, nt, len)
static size_t se_lds_bytes(const SynGeom& g, int rr, int xw) {
  const size_t npix = (size_t)g.n_theta * g.n_phi;
  return sizeof(double2) * ((size_t)2 * (2 * g.L + 1) * (g.n_theta | 1) + 2 * (size_t)g.n_modes + rr * npix) + sizeof(double) * xw +
         sizeof(int) * ((size_t)g.n_lists * g.len + 4);
}
int synthesis_eval_supported(const SynGeom& g, int nt, size_t* lds_bytes, int* nph_out, int* rr_out, int* xw_out) {
  if (!nt) return 0;
  const int nph = (g.n_theta + 15) / 16;
  if (nph > 3 || g.nth + nph > SE_MAX_WAVES || g.n_modes > 64 * g.nth) return 0;
  const size_t npix = (size_t)g.n_theta * g.n_phi;
  if (npix > 3 * 64 * (size_t)(g.nth + nph)) return 0;  // (a wave's share of the evaluation stays within four pixels per lane)
  const size_t cap = 160 * 1024;
  // five ring rows (one barrier per step) where they fit, with the longer window of abscissae if that fits too
  for (int rr = 5; rr >= 4; --rr)
    for (int xw = 1024; xw >= 512; xw -= 512)
      if (se_lds_bytes(g, rr, xw) <= cap) {
        *lds_bytes = se_lds_bytes(g, rr, xw);
        *nph_out = nph, *rr_out = rr, *xw_out = xw;
        return 1;
      }
  return 0;
}

hipError_t launch_synthesis_eval(hipStream_t stream, const double* A, long long lda, long long n_rows, const SynGeom& g, int nt,
                                 const double* Tsyn, const int* meta, const SplineEval& e, double s_min, double s_max, int n_cu) {
  if (n_rows < 4 || e.i_hi <= e.i_lo) return n_rows < 4 && e.i_hi > e.i_lo ? hipErrorInvalidValue : hipSuccess;
  size_t lds_bytes = 0;
  int nph = 0, rr = 0, xw = 0;
  if (!synthesis_eval_supported(g, nt, &lds_bytes, &nph, &rr, &xw)) return hipErrorInvalidValue;
  SynEvalArgs ev;
  ev.skew_b = e.skew_b, ev.g0 = e.g0, ev.n = e.n_knots, ev.i_lo = e.i_lo, ev.i_hi = e.i_hi;
  ev.ldo = e.ldo, ev.s_min = s_min, ev.s_max = s_max;
  ev.xw = xw;
  // shares of the evaluation per lane of a phi wave / a theta wave on a phi wave's SIMD / a theta wave on a SIMD of its own
  // (timeline of the l <= 16 kernel: the three kinds finish their rows after 3.3 / 5.0 / 3.2 thousand cycles)
  int share[3] = {40, 8, 30};
  ev.knock = BMS_PROBE_ENV("SCRI_AMD_SE_KNOCK") ? atoi(BMS_PROBE_ENV("SCRI_AMD_SE_KNOCK")) : 0;
  if (const char* sh = BMS_PROBE_ENV("SCRI_AMD_SE_SHARES")) sscanf(sh, "%d,%d,%d", &share[0], &share[1], &share[2]);
  {
    // wave w: phi (w < nph), theta on a phi wave's SIMD (SIMD = wave mod 4), theta on a SIMD without one; at most four pixels per
    // lane, what a wave cannot take goes to the others in proportion
    const int n_waves = g.nth + nph, npix = g.n_theta * g.n_phi;
    int cnt[SE_MAX_WAVES] = {0}, left = npix;
    bool full[SE_MAX_WAVES] = {false};
    auto sh_of = [&](int w) { return share[w < nph ? 0 : ((w & 3) < nph ? 1 : 2)] > 0 ? share[w < nph ? 0 : ((w & 3) < nph ? 1 : 2)] : 1; };
    while (left > 0) {
      int total = 0;
      for (int w = 0; w < n_waves; ++w) total += full[w] ? 0 : sh_of(w);
      if (!total) return hipErrorInvalidValue;  // (synthesis_eval_supported keeps n_pix within 192 per wave on average)
      int given = 0;
      for (int w = 0; w < n_waves; ++w) {
        if (full[w]) continue;
        int add = (int)(((long long)left * sh_of(w) + total - 1) / total);
        add = std::min(add, std::min(left - given, 256 - cnt[w]));
        cnt[w] += add, given += add;
        if (cnt[w] == 256) full[w] = true;
      }
      left -= given;
    }
    int first = 0;
    for (int w = 0; w < SE_MAX_WAVES; ++w) {
      ev.ev_first[w] = first, ev.ev_count[w] = w < n_waves ? cnt[w] : 0;
      first += ev.ev_count[w];
    }
  }
  ev.trace = nullptr;
#if BMS_PROBES
  if (BMS_PROBE_ENV("SCRI_AMD_SE_RING4") && rr == 5) rr = 4, lds_bytes = se_lds_bytes(g, 4, xw);
  const char* trace_path = BMS_PROBE_ENV("SCRI_AMD_SE_TRACE");
  const size_t trace_words = (size_t)SE_TRACE_STEPS * SE_MAX_WAVES * SE_TRACE_POINTS;
  if (trace_path && hipMalloc(&ev.trace, 8 * trace_words) == hipSuccess) (void)hipMemsetAsync(ev.trace, 0, 8 * trace_words, stream);
  struct TraceDump {  // (debug: blocks the host)
    unsigned long long* p;
    const char* path;
    size_t words;
    hipStream_t s;
    ~TraceDump() {
      if (!p) return;
      std::vector<unsigned long long> h(words);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h.data(), p, 8 * words, hipMemcpyDeviceToHost);
      (void)hipFree(p);
      if (FILE* f = fopen(path, "wb")) {
        fwrite(h.data(), 8, words, f);
        fclose(f);
      }
    }
  } dump{ev.trace, trace_path, trace_words, stream};
#endif
  // segments: their samples (one per window and pixel, the pixels' skews spread over up to a few hundred rows) have to fit the window
  const long long n_win = n_rows - 3;
  const long long seg_max = xw / 2;
  // whole rounds of segments over the workgroups: the smallest number of rounds whose segments fit the window
  long long rounds = 1, seg = (n_win + n_cu - 1) / n_cu;
  while (seg > seg_max) {
    ++rounds;
    seg = (n_win + rounds * n_cu - 1) / (rounds * n_cu);
  }
  seg = seg < 64 ? 64 : seg;
  ev.seg = (int)seg;
  const long long n_seg = (n_win + seg - 1) / seg;
  const dim3 grid((unsigned)(n_seg < n_cu ? n_seg : n_cu)), block(64 * (g.nth + nph));
#define SE_GO(NT, LEN, RR)                                                                                                       \
  {                                                                                                                              \
    hipError_t er = allow_dynamic_lds((const void*)synthesis_eval_kernel<NT, LEN, RR>);                                          \
    if (er != hipSuccess) return er;                                                                                             \
    hipLaunchKernelGGL((synthesis_eval_kernel<NT, LEN, RR>), grid, block, lds_bytes, stream, A, lda, n_rows, g, nph, Tsyn, meta, \
                       e.x, e.table, e.out, ev);                                                                                 \
    return hipGetLastError();                                                                                                    \
  }
#define SE_LEN(RR)                  \
  {                                 \
    if (g.len == 8) SE_GO(40, 8, RR)   \
    if (g.len == 12) SE_GO(40, 12, RR) \
    if (g.len == 16) SE_GO(40, 16, RR) \
    SE_GO(40, 20, RR)                  \
  }
  if (rr == 5) SE_LEN(5)
  SE_LEN(4)
#undef SE_LEN
#undef SE_GO
}

}  // namespace bms
