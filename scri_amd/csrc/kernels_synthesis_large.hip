// Separable synthesis for the grids AsymptoticBondiData works on (40 < n_theta <= 104: 49 x 49 at l_max = 12, 99 x 99 at
// l_max = 24) and for every other boost-free shape the one-kernel form of kernels_synthesis.hip does not take (l_max > 16,
// the psi-mixing WaveformModes types).  Without a boost the output grid is the equiangular grid seen through the constant
// frame rotation (scri/asymptotic_bondi_data/transformations.py:312-334 with beta = 0, scri/waveform_grid.py:130-174), so
// after rotating the modes once the map modes -> grid factors per time step into
//     F_m(theta_j) = sum_l sLambda_lm(theta_j) a_lm ,        g(theta_j, phi_k) = sum_m F_m(theta_j) e^{i m phi_k} .
// F of one time step (2L+1 rows of n_theta complex numbers: 78 KB at l_max = 24) no longer fits a workgroup's LDS next to the
// operands of a batched product, so -- as in the large-grid analysis, whose mirror image this is (kernels_analysis.hip:
// phi_dft_folded_kernel + theta_quadrature_mfma_kernel) -- the two stages are two kernels with F[t][m][ring] in HBM between
// them, both on the matrix pipe with their constant operand in REGISTERS for the whole kernel:
//
//   theta stage   workgroup = (m, block of time steps).  sLambda_lm(theta_j) of that m (l = max(|m|, l_min) .. L by n_theta
//                 rings) are the B fragments; a wave multiplies 8 time steps (16 real rows: Re and Im are independent) per
//                 trip, gathers their a_lm (16 bytes each, l apart) one trip ahead and stores F in 256-byte runs.
//   phi stage     one wave per (time step, 8 rings).  With P_m = F_m + F_-m, Q_m = i (F_m - F_-m):
//                 g_k = F_0 + sum_{m>=1} P_m cos(m phi_k) + Q_m sin(m phi_k),  g_{n-k} = F_0 + sum P_m cos - Q_m sin  (k <= n/2):
//                 two real products [16 rows = 8 rings x (Re, Im)] x [m = 1..L] x [k = 0..n_phi/2] with cos | sin twiddles
//                 in registers; the tile of the grid row (8 rings = 8 n_phi contiguous complex numbers) is assembled in LDS
//                 and leaves as whole 1 KB wave stores.
// HBM traffic per time step and field: 16 n_modes read, 2 x 16 (2L+1) jp for F, 16 n_pix written -- 0.32 MB at l_max = 24
// against the 49 MFLOP of the dense product with the 9801 x 625 matrix of sYlm values (0.65 us on the matrix pipe).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"
#include "wigner.h"

namespace bms {

typedef double v4d_t __attribute__((ext_vector_type(4)));

// ---- theta stage: F[t][mi][j] = sum_l Tsyn[(l, m)][j] A[t][(l, m)]
template <int KL, int NTJ>
__global__ __launch_bounds__(256, NTJ <= 4 ? 3 : 2) void theta_synthesis_mfma_kernel(const double* __restrict__ A, long long lda, long long n_rows,
                                                                   int n_theta, int L, int ell_min, int jp, int rows_per_block,
                                                                   const double* __restrict__ Tsyn, double* __restrict__ F) {
  constexpr int PQ = 4 * KL + 2;  // 2 x odd: conflict-free fragment reads
  __shared__ double As[4][16 * PQ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fi = lane & 15, fk = lane >> 4;
  const int mi = blockIdx.x, m = mi - L, nm = 2 * L + 1;
  // the rings of an m may be shared out over gridDim.z workgroups (tiles of 16 rings from tile0 on): half the table fragments per
  // lane, so that three waves instead of two fit a SIMD -- the gathers of the modes are then done once per part
  const int tile0 = blockIdx.z * NTJ;
  const int am = m < 0 ? -m : m;
  const int l0 = am > ell_min ? am : ell_min;  // first l of this m
  const int nl = L - l0 + 1;                   // its number of l values (k extent of the product)
  const int ks = (nl + 3) / 4;                 // k steps that carry an l at all
  // B fragments: Tsyn[(l0 + 4 s + fk, m)][16 n + fi]
  double bq[KL][NTJ];
#pragma unroll
  for (int s = 0; s < KL; ++s) {
    const int l = l0 + 4 * s + fk;
    const long long o = (long long)l * (l + 1) - (long long)ell_min * ell_min + m;
#pragma unroll
    for (int n = 0; n < NTJ; ++n) {
      const int j = 16 * (n + tile0) + fi;
      bq[s][n] = (l <= L && j < n_theta) ? Tsyn[o * n_theta + j] : 0.0;
    }
  }
  double* Aw = As[wave];
  for (int e = lane; e < 16 * PQ; e += 64) Aw[e] = 0.0;
  const long long t_begin = (long long)blockIdx.y * rows_per_block;
  long long t_end = t_begin + rows_per_block;
  if (t_end > n_rows) t_end = n_rows;
  // This lane's pieces of a trip (8 time steps x nl modes of this m): element e = lane + 64 i is mode l0 + li of time step r.
  constexpr int NPRE = (8 * 4 * KL + 63) / 64;
  int goff[NPRE], soff[NPRE];  // offset (doubles) within the row (-1: nothing to load), r << 16 | LDS slot
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int e = lane + 64 * i;
    const int r = e / (4 * KL), li = e - r * (4 * KL);
    const int l = l0 + li;
    const bool ok = r < 8 && li < nl;
    goff[i] = ok ? 2 * (l * (l + 1) - ell_min * ell_min + m) : -1;
    soff[i] = (r << 16) | (((r & 3) + 8 * (r >> 2)) * PQ + li);
  }
  double2 pre[NPRE];
#define TS_LOAD(T0)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < NPRE; ++i) {                                                                 \
    const int r = soff[i] >> 16;                                                                                     \
    pre[i] = (goff[i] >= 0 && (T0) + r < t_end) ? *reinterpret_cast<const double2*>(A + ((T0) + r) * lda + goff[i])   \
                                                 : double2{0.0, 0.0};                                                 \
  }
  if (t_begin + 8 * wave < t_end) TS_LOAD(t_begin + 8 * wave)
  for (long long t0 = t_begin + 8 * wave; t0 < t_end; t0 += 32) {
    // ---- operand: row g + 8 h (+4 for Im) holds time step t0 + g + 4 h
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
      if (goff[i] >= 0) {
        const int slot = soff[i] & 0xffff;
        Aw[slot] = pre[i].x;
        Aw[slot + 4 * PQ] = pre[i].y;
      }
    if (t0 + 32 < t_end) TS_LOAD(t0 + 32)
    v4d_t acc[NTJ];
#pragma unroll
    for (int n = 0; n < NTJ; ++n) acc[n] = v4d_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KL; ++s) {
      if (s < ks) {  // (wave-uniform: the high |m| have few l; a `break` here keeps the loop from being unrolled at KL = 9)
        const double a = Aw[fi * PQ + 4 * s + fk];
#pragma unroll
        for (int n = 0; n < NTJ; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[s][n], acc[n], 0, 0, 0);
      }
    }
    // results r = 2h (re), 2h+1 (im) of time step t0 + fk + 4 h, ring j = 16 n + fi (the padding rings j >= n_theta get zeros)
#pragma unroll
    for (int n = 0; n < NTJ; ++n) {
      const int j = 16 * (n + tile0) + fi;
      if (j < jp) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const long long t = t0 + fk + 4 * h;
          if (t < t_end) *reinterpret_cast<double2*>(F + ((t * nm + mi) * (long long)jp + j) * 2) = double2{acc[n][2 * h], acc[n][2 * h + 1]};
        }
      }
    }
  }
#undef TS_LOAD
}

// ---- phi stage: Y[t][ring][k] = (sum_m F[t][m][ring] e^{i m phi_k} (- off[pixel] x the row's constant)) (x scale[pixel], the
// conformal factor's power behind a boost along the grid's axis).  One wave per (time step, tile of 8 rings).
template <int KM, int NTC>
__global__ __launch_bounds__(64) void phi_synthesis_folded_kernel(const double* __restrict__ F, long long n_rows, int n_theta,
                                                                  int n_phi, int L, int jp, const double* __restrict__ off,
                                                                  const double* __restrict__ cst, long long ldc,
                                                                  const double* __restrict__ scale, double* __restrict__ Y,
                                                                  long long ldy) {
  extern __shared__ double2 sm[];  // Ft[(2L+1)][8] tile of F, then Gt[8][n_phi] tile of the grid row
  const int nm = 2 * L + 1;
  double2* Ft = sm;
  double2* Gt = sm + nm * 8;
  const int lane = threadIdx.x, fi = lane & 15, fk = lane >> 4;
  const int nk = n_phi / 2 + 1, mt = (n_theta + 7) / 8;
  // twiddles of this lane's B fragments: cos / sin (m phi_k), m = 4 s + fk + 1, k = 16 n + fi
  double bc[KM][NTC], bs[KM][NTC];
#pragma unroll
  for (int s = 0; s < KM; ++s)
#pragma unroll
    for (int n = 0; n < NTC; ++n) {
      const int m = 4 * s + fk + 1, k = 16 * n + fi;
      double sn = 0.0, co = 0.0;
      if (k < nk && m <= L) sincospi(2.0 * (double)(((long long)m * k) % n_phi) / (double)n_phi, &sn, &co);
      bc[s][n] = co, bs[s][n] = sn;
    }
  const int km = (L + 3) / 4;  // k steps that carry an m <= L
  const long long n_items = n_rows * mt;
  // This lane's pieces of an item's F tile ((2L+1) x 8 rings, 128-byte pieces): element e = lane + 64 i is ring e & 7 of row e >> 3
  constexpr int NPRE = ((2 * 4 * KM + 1) * 8 + 63) / 64;
  double2 pre[NPRE];
#define PS_LOAD(ITEM)                                                                                              \
  {                                                                                                                \
    const long long t_ = (ITEM) / mt;                                                                              \
    const int rt_ = (int)((ITEM)-t_ * mt);                                                                         \
    const int rings_ = n_theta - 8 * rt_ < 8 ? n_theta - 8 * rt_ : 8;                                              \
    const double* f_ = F + ((t_ * nm) * (long long)jp + 8 * rt_) * 2;                                              \
    _Pragma("unroll") for (int i = 0; i < NPRE; ++i) {                                                             \
      const int e_ = lane + 64 * i;                                                                                \
      pre[i] = ((e_ >> 3) < nm && (e_ & 7) < rings_) ? *reinterpret_cast<const double2*>(f_ + ((long long)(e_ >> 3) * jp + (e_ & 7)) * 2) \
                                                      : double2{0.0, 0.0};                                         \
    }                                                                                                              \
  }
  if (blockIdx.x < n_items) PS_LOAD((long long)blockIdx.x)
  // operand row fi of the products: ring (fi & 3) + 4 (fi >> 3) of the tile, Re (part 0) or Im (part 1)
  const int ag = (fi & 3) + 4 * (fi >> 3), part = (fi >> 2) & 1;
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const long long t = item / mt;
    const int rt = (int)(item - t * mt);
    const int rings = n_theta - 8 * rt < 8 ? n_theta - 8 * rt : 8;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int e = lane + 64 * i;
      if ((e >> 3) < nm) Ft[e] = pre[i];
    }
    if (item + gridDim.x < n_items) PS_LOAD(item + gridDim.x)
    // (one wave: its LDS writes are visible to its own reads in program order)
    v4d_t u[NTC], v[NTC];
#pragma unroll
    for (int n = 0; n < NTC; ++n) u[n] = v[n] = v4d_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KM; ++s) {
      if (s < km) {
        const int m = 4 * s + fk + 1;
        const int mm = m <= L ? m : 0;  // (rows beyond L: their twiddles are zero)
        const double2 fp = Ft[(L + mm) * 8 + ag], fm = Ft[(L - mm) * 8 + ag];
        const double ap = part ? fp.y + fm.y : fp.x + fm.x;  // P_m = F_m + F_-m
        const double aq = part ? fp.x - fm.x : fm.y - fp.y;  // Q_m = i (F_m - F_-m)
#pragma unroll
        for (int n = 0; n < NTC; ++n) {
          u[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, bc[s][n], u[n], 0, 0, 0);
          v[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bs[s][n], v[n], 0, 0, 0);
        }
      }
    }
    // results: rows fk + 4 r = (Re, Im) of ring fk, (Re, Im) of ring fk + 4; column k = 16 n + fi
    const double cv = off ? cst[t * ldc] : 0.0;  // the row's eliminated-constant value (real: engine_abd.hip)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ring = fk + 4 * h;
      const double2 f0 = Ft[L * 8 + ring];
      const long long pix0 = (long long)(8 * rt + ring) * n_phi;
#pragma unroll
      for (int n = 0; n < NTC; ++n) {
        const int k = 16 * n + fi;
        if (k < nk && ring < rings) {
          const double ux = f0.x + u[n][2 * h], uy = f0.y + u[n][2 * h + 1];
          double2 a{ux + v[n][2 * h], uy + v[n][2 * h + 1]};
          if (off) {
            const double2 o = *reinterpret_cast<const double2*>(off + 2 * (pix0 + k));
            a.x -= o.x * cv, a.y -= o.y * cv;
          }
          Gt[ring * n_phi + k] = a;
          const int k2 = n_phi - k;
          if (k >= 1 && k2 != k) {
            double2 b{ux - v[n][2 * h], uy - v[n][2 * h + 1]};
            if (off) {
              const double2 o = *reinterpret_cast<const double2*>(off + 2 * (pix0 + k2));
              b.x -= o.x * cv, b.y -= o.y * cv;
            }
            Gt[ring * n_phi + k2] = b;
          }
        }
      }
    }
    // ---- the tile is rings x n_phi contiguous complex numbers of the grid row
    double* y = Y + t * ldy + 2LL * (8 * rt) * n_phi;
    const int n_el = rings * n_phi;
    if (scale) {
      const double* sc = scale + 2LL * (8 * rt) * n_phi;  // (stored twice per pixel: engine_tables.hip's col_scale)
      for (int e = lane; e < n_el; e += 64) {
        const double2 g = Gt[e];
        const double w = sc[2 * e];
        *reinterpret_cast<double2*>(y + 2LL * e) = double2{g.x * w, g.y * w};
      }
    } else
      for (int e = lane; e < n_el; e += 64) *reinterpret_cast<double2*>(y + 2LL * e) = Gt[e];
  }
#undef PS_LOAD
}

// ---- AsymptoticBondiData without a boost: phi stage of all SIX fields + their Horner mixing in one pass.
// Without a boost k = 1 and eth k = 0, so the mixing variable X = (eth k / k)(u - alpha) - eth alpha of
// scri/asymptotic_bondi_data/transformations.py:322-385 is -eth alpha: constant in time.  A per-pixel linear map with
// time-independent coefficients commutes with the spline's forward elimination (linear along time with per-knot coefficients), so
// the elimination runs on the MODES (bspline_forward_modes_kernel, 6 x 625 columns instead of 6 x 9801), the six fields are
// synthesised as eliminated coefficients and mixed on their way out of this kernel -- the separate mixing + elimination pass over
// the six grids (47 GB at l <= 24, 25 000 steps) is gone.  sigma' = sigma - eth^2 alpha becomes F[sigma] - eth^2 alpha F[1] with
// the eliminated constant series F[1] (the extra column of the eliminated modes), as for h and sigma of the WaveformModes flavour.
//
// One wave per (time step, 4 rings).  The 16 rows of an MFMA are 2 fields x 4 rings x (Re, Im): three passes (fields 0|1, 2|3, 4|5)
// with the cos | sin twiddles in registers throughout; the F tile of a pass sits in the LDS area its own results will overwrite (all
// of it is in registers or accumulators by then); after the third pass the area holds the six fields on the tile's 4 n_phi pixels,
// which are mixed pixel by pixel and leave as whole wave stores (4 n_phi contiguous complex numbers per field).
struct Mix6Args {
  const double* F[6];   // F[f][t][m][jp]
  double* out[6];       // eliminated coefficients of the mixed fields, [rows][ldo]
  const double* eth_alpha;     // c16[n_pix]  (eth alpha / sqrt 2 of the reference)
  const double* etheth_alpha;  // c16[n_pix]
  const double* cst;    // eliminated constant series: cst[t * ldc]
  long long ldc, ldo;
};

template <int KM, int NTC>
__global__ __launch_bounds__(64) void phi_synthesis_mix6_kernel(Mix6Args a, long long n_rows, int n_theta, int n_phi, int L, int jp) {
  extern __shared__ double2 sm[];  // [6][4 n_phi]
  const int nm = 2 * L + 1, tile = 4 * n_phi;
  const int lane = threadIdx.x, fi = lane & 15, fk = lane >> 4;
  const int nk = n_phi / 2 + 1, mt = (n_theta + 3) / 4;
  double bc[KM][NTC], bs[KM][NTC];
#pragma unroll
  for (int s = 0; s < KM; ++s)
#pragma unroll
    for (int n = 0; n < NTC; ++n) {
      const int m = 4 * s + fk + 1, k = 16 * n + fi;
      double sn = 0.0, co = 0.0;
      if (k < nk && m <= L) sincospi(2.0 * (double)(((long long)m * k) % n_phi) / (double)n_phi, &sn, &co);
      bc[s][n] = co, bs[s][n] = sn;
    }
  const int km = (L + 3) / 4;
  // This lane's pieces of a pass's two F tiles ((2L+1) x 4 rings each, 64-byte pieces): element e = lane + 64 i is ring e & 3 of
  // row (e >> 2) % nm of field e / (4 nm) of the pair
  constexpr int NPRE = ((2 * 4 * KM + 1) * 8 + 63) / 64;
  double2 pre[NPRE];
#define M6_LOAD(ITEM, PASS)                                                                                             \
  {                                                                                                                     \
    const long long item_ = (ITEM);                                                                                     \
    const long long t_ = item_ / mt;                                                                                    \
    const int rt_ = (int)(item_ - t_ * mt);                                                                             \
    const int rings_ = n_theta - 4 * rt_ < 4 ? n_theta - 4 * rt_ : 4;                                                   \
    const long long o_ = ((t_ * nm) * (long long)jp + 4 * rt_) * 2;                                                     \
    const double *fa_ = a.F[2 * (PASS)] + o_, *fb_ = a.F[2 * (PASS) + 1] + o_;                                          \
    _Pragma("unroll") for (int i = 0; i < NPRE; ++i) {                                                                  \
      const int e_ = lane + 64 * i;                                                                                     \
      const int fld_ = e_ >= 4 * nm ? 1 : 0, rem_ = e_ - fld_ * 4 * nm;                                                 \
      const double* f_ = fld_ ? fb_ : fa_;                                                                              \
      pre[i] = (e_ < 8 * nm && (rem_ & 3) < rings_) ? *reinterpret_cast<const double2*>(f_ + ((long long)(rem_ >> 2) * jp + (rem_ & 3)) * 2) \
                                                    : double2{0.0, 0.0};                                               \
    }                                                                                                                   \
  }
  // operand row fi: field fi >> 3 of the pair, ring fi & 3, Re (part 0) or Im (part 1)
  const int ab = fi >> 3, ag = fi & 3, part = (fi >> 2) & 1;
  constexpr int MAXIT = 8;  // 64-lane sweeps over the tile's 4 n_phi pixels (n_phi <= 127)
  double2 tB[MAXIT], tE[MAXIT];
  const long long n_items = n_rows * mt;
  if (blockIdx.x < n_items) M6_LOAD((long long)blockIdx.x, 0)
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const long long t = item / mt;
    const int rt = (int)(item - t * mt);
    const int rings = n_theta - 4 * rt < 4 ? n_theta - 4 * rt : 4;
    const long long pix0 = (long long)(4 * rt) * n_phi;
    const int n_el = rings * n_phi;
    // the per-pixel constants of the mixing are requested now and arrive under the three passes (asked for inside the mixing loop
    // each sweep waited out an L2 round trip)
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int e = lane + 64 * it;
      const long long p = pix0 + (e < n_el ? e : 0);
      tB[it] = *reinterpret_cast<const double2*>(a.eth_alpha + 2 * p);
      tE[it] = *reinterpret_cast<const double2*>(a.etheth_alpha + 2 * p);
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {  // (unrolled: the field pointers of a pass are then scalars, not a private array)
      // the pass's F tiles go where its results will: fields 2 pass, 2 pass + 1
      double2* Ft = sm + (size_t)(2 * pass) * tile;  // [2][nm][4]  (2 x 4 nm <= 2 x 4 n_phi: nm <= n_phi for every grid that resolves L)
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int e = lane + 64 * i;
        if (e < 8 * nm) Ft[e] = pre[i];
      }
      if (pass == 0) M6_LOAD(item, 1)
      if (pass == 1) M6_LOAD(item, 2)
      if (pass == 2 && item + gridDim.x < n_items) M6_LOAD(item + gridDim.x, 0)
      v4d_t u[NTC], v[NTC];
#pragma unroll
      for (int n = 0; n < NTC; ++n) u[n] = v[n] = v4d_t{0.0, 0.0, 0.0, 0.0};
      const double2* Fb = Ft + ab * 4 * nm;
#pragma unroll
      for (int s = 0; s < KM; ++s) {
        if (s < km) {
          const int m = 4 * s + fk + 1;
          const int mm = m <= L ? m : 0;
          const double2 fp = Fb[(L + mm) * 4 + ag], fm = Fb[(L - mm) * 4 + ag];
          const double ap = part ? fp.y + fm.y : fp.x + fm.x;
          const double aq = part ? fp.x - fm.x : fm.y - fp.y;
#pragma unroll
          for (int n = 0; n < NTC; ++n) {
            u[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, bc[s][n], u[n], 0, 0, 0);
            v[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bs[s][n], v[n], 0, 0, 0);
          }
        }
      }
      // results: rows fk + 4 r = (Re, Im) of ring fk of the pair's first field, (Re, Im) of ring fk of its second; column k = 16 n + fi
      const double2 f0a = Ft[L * 4 + fk], f0b = Ft[4 * nm + L * 4 + fk];
      // (every read of the F tiles is done: their area now takes the results)
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      double2* Ga = sm + (size_t)(2 * pass) * tile + fk * n_phi;
      double2* Gb = Ga + tile;
#pragma unroll
      for (int n = 0; n < NTC; ++n) {
        const int k = 16 * n + fi;
        if (k < nk) {
          const double uax = f0a.x + u[n][0], uay = f0a.y + u[n][1], ubx = f0b.x + u[n][2], uby = f0b.y + u[n][3];
          Ga[k] = double2{uax + v[n][0], uay + v[n][1]};
          Gb[k] = double2{ubx + v[n][2], uby + v[n][3]};
          const int k2 = n_phi - k;
          if (k >= 1 && k2 != k) {
            Ga[k2] = double2{uax - v[n][0], uay - v[n][1]};
            Gb[k2] = double2{ubx - v[n][2], uby - v[n][3]};
          }
        }
      }
    }
    // ---- mixing (transformations.py:340-385 with X = -eth alpha, 1 / k = 1 / k^3 = 1) and the way out
    const double cv = a.cst[t * a.ldc];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int e = lane + 64 * it;
      if (e < n_el) {
        const long long p = pix0 + e;
        cplx f[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double2 w = sm[(size_t)i * tile + e];
          f[i] = {w.x, w.y};
        }
        const double2 EE = tE[it];
        const cplx X = {-tB[it].x, -tB[it].y};
        auto axpy = [](cplx tt, cplx X, double c, cplx ff) {  // tt X + c ff
          cplx r = cmul(tt, X);
          return cplx{r.re + c * ff.re, r.im + c * ff.im};
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
        const cplx mixed[6] = {t0, t1, t2, t3, f[4], {f[5].re - EE.x * cv, f[5].im - EE.y * cv}};
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<double2*>(a.out[i] + t * a.ldo + 2 * p) = double2{mixed[i].re, mixed[i].im};
      }
    }
  }
#undef M6_LOAD
}

int large_synthesis_supported(int n_theta, int n_phi, int ell_min, int ell_max) {
  const int nk = n_phi / 2 + 1;
  return n_theta >= 2 && n_theta <= 104 && n_phi >= 1 && nk <= 64 && ell_max >= 1 && ell_max <= 33 && ell_min >= 0 && ell_min <= ell_max;
}

// A: [n_rows][lda] modes (complex; with `off` one more complex number per row at column n_modes: it multiplies `off`);
// Tsyn[n_modes][n_theta]; F: n_rows x (2 ell_max + 1) x large_analysis_jp(n_theta) complex of work space; Y[n_rows][ldy];
// scale: NULL or 2 doubles per pixel (the first is used) multiplying the result.
hipError_t launch_synthesis_large(hipStream_t stream, const double* A, long long lda, long long n_rows, int n_theta, int n_phi,
                                  int ell_min, int ell_max, const double* Tsyn, const double* off, double* F, double* Y,
                                  long long ldy, const double* scale) {
  if (n_rows <= 0) return hipSuccess;
  const int L = ell_max, nm = 2 * L + 1, jp = large_analysis_jp(n_theta);
  const int n_modes = (L + 1) * (L + 1) - ell_min * ell_min;
  static const int rows_per_block = BMS_PROBE_ENV("SCRI_AMD_TS_ROWS") ? atoi(BMS_PROBE_ENV("SCRI_AMD_TS_ROWS")) : 256;
  const dim3 grid1(nm, (unsigned)((n_rows + rows_per_block - 1) / rows_per_block));
  const int kl = (L + 1 - ell_min + 3) / 4, ntj_all = (n_theta + 15) / 16;
  // SCRI_AMD_TS_SPLIT=1 and more than four ring tiles (n_theta > 64): two workgroups per m, each with half of them
  // (measured, round 4: 99 x 99 / l <= 24, 25 000 steps, six fields: 14.65 ms split against 13.09 ms whole -- three waves per SIMD at
  // 153 registers do not make up for gathering every mode twice; off unless asked for)
  static const int ts_split = BMS_PROBE_ENV("SCRI_AMD_TS_SPLIT") ? atoi(BMS_PROBE_ENV("SCRI_AMD_TS_SPLIT")) : 0;
  const int parts = ts_split && ntj_all > 4 ? 2 : 1;
  const int ntj = (ntj_all + parts - 1) / parts;
  const dim3 grid1z(grid1.x, grid1.y, parts);
#define TS_GO(KL, NTJ)                                                                                                      \
  hipLaunchKernelGGL((theta_synthesis_mfma_kernel<KL, NTJ>), grid1z, dim3(256), 0, stream, A, lda, n_rows, n_theta, L, ell_min, jp, \
                     rows_per_block, Tsyn, F)
#define TS_KL(NTJ)  \
  if (kl <= 5)      \
    TS_GO(5, NTJ);  \
  else if (kl <= 7) \
    TS_GO(7, NTJ);  \
  else              \
    TS_GO(9, NTJ);
  if (ntj <= 3) {
    TS_KL(3)
  } else if (ntj == 4) {
    TS_KL(4)
  } else if (ntj <= 5) {
    TS_KL(5)
  } else {
    TS_KL(7)
  }
#undef TS_KL
#undef TS_GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int nk = n_phi / 2 + 1, mt = (n_theta + 7) / 8;
  const int km = (L + 3) / 4, ntc = (nk + 15) / 16;
  const size_t lds = sizeof(double2) * ((size_t)nm * 8 + (size_t)8 * n_phi);
  const long long items = n_rows * mt;
  const unsigned grid2 = (unsigned)(items < 256 * 8 ? items : 256 * 8);
#define PS_GO(KM, NTC)                                                                                                          \
  hipLaunchKernelGGL((phi_synthesis_folded_kernel<KM, NTC>), dim3(grid2), dim3(64), lds, stream, F, n_rows, n_theta, n_phi, L, jp, off, \
                     A + 2LL * n_modes, lda, scale, Y, ldy)
#define PS_KM(NTC)  \
  if (km <= 5)      \
    PS_GO(5, NTC);  \
  else if (km <= 7) \
    PS_GO(7, NTC);  \
  else              \
    PS_GO(9, NTC);
  if (ntc <= 2) {
    PS_KM(2)
  } else if (ntc == 3) {
    PS_KM(3)
  } else {
    PS_KM(4)
  }
#undef PS_KM
#undef PS_GO
  return hipGetLastError();
}

// theta stage alone (the six fields of the fused ABD route: one launch per field into its own F)
hipError_t launch_theta_synthesis(hipStream_t stream, const double* A, long long lda, long long n_rows, int n_theta, int ell_min,
                                  int ell_max, const double* Tsyn, double* F) {
  if (n_rows <= 0) return hipSuccess;
  const int L = ell_max, nm = 2 * L + 1, jp = large_analysis_jp(n_theta);
  static const int rows_per_block = BMS_PROBE_ENV("SCRI_AMD_TS_ROWS") ? atoi(BMS_PROBE_ENV("SCRI_AMD_TS_ROWS")) : 256;
  const dim3 grid1(nm, (unsigned)((n_rows + rows_per_block - 1) / rows_per_block));
  const int kl = (L + 1 - ell_min + 3) / 4, ntj_all = (n_theta + 15) / 16;
  // SCRI_AMD_TS_SPLIT=1 and more than four ring tiles (n_theta > 64): two workgroups per m, each with half of them
  // (measured, round 4: 99 x 99 / l <= 24, 25 000 steps, six fields: 14.65 ms split against 13.09 ms whole -- three waves per SIMD at
  // 153 registers do not make up for gathering every mode twice; off unless asked for)
  static const int ts_split = BMS_PROBE_ENV("SCRI_AMD_TS_SPLIT") ? atoi(BMS_PROBE_ENV("SCRI_AMD_TS_SPLIT")) : 0;
  const int parts = ts_split && ntj_all > 4 ? 2 : 1;
  const int ntj = (ntj_all + parts - 1) / parts;
  const dim3 grid1z(grid1.x, grid1.y, parts);
#define TS_GO(KL, NTJ)                                                                                                      \
  hipLaunchKernelGGL((theta_synthesis_mfma_kernel<KL, NTJ>), grid1z, dim3(256), 0, stream, A, lda, n_rows, n_theta, L, ell_min, jp, \
                     rows_per_block, Tsyn, F)
#define TS_KL(NTJ)  \
  if (kl <= 5)      \
    TS_GO(5, NTJ);  \
  else if (kl <= 7) \
    TS_GO(7, NTJ);  \
  else              \
    TS_GO(9, NTJ);
  if (ntj <= 3) {
    TS_KL(3)
  } else if (ntj == 4) {
    TS_KL(4)
  } else if (ntj <= 5) {
    TS_KL(5)
  } else {
    TS_KL(7)
  }
#undef TS_KL
#undef TS_GO
  return hipGetLastError();
}

int abd_mix6_supported(int n_theta, int n_phi, int ell_max) {
  // (the F tiles of a pass borrow the LDS area of its results: 2 L + 1 <= n_phi; three workgroups or more per CU)
  return large_synthesis_supported(n_theta, n_phi, 0, ell_max) && 2 * ell_max + 1 <= n_phi && (size_t)6 * 4 * n_phi * 16 <= 53 * 1024;
}

// F6: six F buffers (each n_rows x (2 L + 1) x jp complex); out6: six grids [n_rows][ldo]; per-pixel tables in grid order
hipError_t launch_phi_synthesis_mix6(hipStream_t stream, const double* const F6[6], long long n_rows, int n_theta, int n_phi,
                                     int ell_max, const double* eth_alpha, const double* etheth_alpha, const double* cst, long long ldc,
                                     double* const out6[6], long long ldo) {
  if (n_rows <= 0) return hipSuccess;
  Mix6Args a;
  for (int f = 0; f < 6; ++f) a.F[f] = F6[f], a.out[f] = out6[f];
  a.eth_alpha = eth_alpha, a.etheth_alpha = etheth_alpha, a.cst = cst, a.ldc = ldc, a.ldo = ldo;
  const int L = ell_max, jp = large_analysis_jp(n_theta);
  const int nk = n_phi / 2 + 1, mt = (n_theta + 3) / 4;
  const int km = (L + 3) / 4, ntc = (nk + 15) / 16;
  const size_t lds = sizeof(double2) * (size_t)6 * 4 * n_phi;
  const long long items = n_rows * mt;
  const unsigned grid = (unsigned)(items < 256 * 4 ? items : 256 * 4);
#define M6_GO(KM, NTC)                                                                                                    \
  {                                                                                                                       \
    hipError_t e_ = allow_dynamic_lds((const void*)phi_synthesis_mix6_kernel<KM, NTC>); \
    if (e_ != hipSuccess) return e_;                                                                                      \
    hipLaunchKernelGGL((phi_synthesis_mix6_kernel<KM, NTC>), dim3(grid), dim3(64), lds, stream, a, n_rows, n_theta, n_phi, L, jp);  \
  }
#define M6_KM(NTC)  \
  if (km <= 5)      \
    M6_GO(5, NTC)   \
  else if (km <= 7) \
    M6_GO(7, NTC)   \
  else              \
    M6_GO(9, NTC)
  if (ntc <= 2) {
    M6_KM(2)
  } else if (ntc == 3) {
    M6_KM(3)
  } else {
    M6_KM(4)
  }
#undef M6_KM
#undef M6_GO
  return hipGetLastError();
}

}  // namespace bms
