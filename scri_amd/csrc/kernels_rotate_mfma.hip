// Time-series / constant Wigner-D rotation of mode weights (scri/rotations.py:346-392) on the fp64 matrix cores.
//
//   out[t,l,m] = sum_m' data[t,l,m'] D^l_{m',m}(R_t),     D^l_{m',m} = ea^(m'+m) eb^(m-m') d^l_{m',m}(b)
//   d^l_{m',m}(b) = i^(m'-m) sum_mu Delta^l_{mu,m'} Delta^l_{mu,m} exp(-i mu b),       Delta^l = d^l(pi/2) (constants)
//
//   =>  g_m' = f_m' p1^m'        p1 = i ea conj(eb)          (phase, per time step)
//       h_mu = p2^mu sum_m' Delta_{mu,m'} g_m'       p2 = exp(-i b)       (REAL constant matrix: a GEMM over time steps)
//       o_m  = p3^m  sum_mu Delta_{mu,m} h_mu        p3 = -i ea eb        (REAL constant matrix: a GEMM over time steps)
//
// The per-step D matrix of the reference (6535 complex numbers at l <= 16) is never formed, and because Delta is real
// the Re and Im parts of a row are two independent real rows.  A workgroup owns 64 time steps = 128 real rows
// (rows 0..63 Re, 64..127 Im): wave w owns the 16-row tiles w (Re) and w+4 (Im) of the same 16 time steps, so the two
// accumulators of a lane hold Re and Im of the same (t, mu) and the phase products need no cross-lane traffic.  Per l:
// the rows are staged (phase p1 applied) into an LDS A-operand image, Delta^l and its transpose sit in LDS as B operands
// (conflict-free pitches), the two products run as v_mfma_f64_16x16x4_f64 chains entirely wave-local (in place in LDS),
// and the rotated rows are written back in place.  Every mode is read and written exactly once from HBM
// (2 * 16 * n_modes + 32 B per step); arithmetic 2 stages x 2 x (2l+1)^2 x 2 flop per (t, l) plus tile padding.
// Rotors with |Rb| ~ 0 / |Ra| ~ 0 take exact diagonal / anti-diagonal branches (identity stays bit-exact).
#include "wigner.h"
#include "kernels.h"

namespace bms {

typedef double v4dr __attribute__((ext_vector_type(4)));

constexpr int RM_TB = 64;       // time steps per workgroup
constexpr int RM_MAX_NT = 5;    // column tiles of 16: 2l+1 <= 80 (l <= 32 supported by the launcher)
constexpr int RM_LD = 9;        // elements a thread loads per batch in the staging pass ((2l+1)/4 <= 9 for l <= 16)

struct RotGeom {
  int pa;       // LDS pitch of an A row (doubles), 2 x odd
  int b_doubles;  // doubles reserved for the two B images
};

template <int MAXNT>
__global__ __launch_bounds__(256, 2) void rotate_modes_mfma_kernel(double* __restrict__ data, long long n_times, long long ld,
                                                                int ell_min, int ell_max, const double* __restrict__ RaRb,
                                                                long long rotor_stride, const double* __restrict__ btab,
                                                                const long long* __restrict__ boff, RotGeom geo) {
  extern __shared__ double lds[];
  double* As = lds;                       // [128][pa]
  double* Bs = lds + 128 * geo.pa;        // B1 | B2 of the current l
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long t0 = (long long)blockIdx.x * RM_TB;

  // ---- per-thread rotor for the load/store stage: thread <-> (time step tl, quarter q of the m range)
  const int tl = tid >> 2, q = tid & 3;
  const long long tg = t0 + tl;
  const bool live = tg < n_times;
  cplx p1 = {1, 0}, p3 = {1, 0}, ea2 = {1, 0}, eb2 = {1, 0};
  bool z_only = false, flip = false;
  {
    cplx Ra = {1.0, 0.0}, Rb = {0.0, 0.0};
    if (live) {
      const double* r = RaRb + tg * rotor_stride;
      Ra = {r[0], r[1]};
      Rb = {r[2], r[3]};
    }
    double ra, rb;
    cplx ea, eb;
    spinor_polar(Ra, Rb, ra, rb, ea, eb);
    z_only = rb <= 1e-15;
    flip = ra <= 1e-15;
    p1 = cmul(cplx{0.0, 1.0}, cmul(ea, cconj(eb)));
    p3 = cmul(cplx{0.0, -1.0}, cmul(ea, eb));
    ea2 = cmul(ea, ea);
    eb2 = cmul(eb, eb);
  }
  // ---- per-lane exp(-i b) of the 4 time steps whose accumulator rows this lane holds: t = 16 wave + (lane>>4) + 4 r
  // exp(-i mu b) for mu = 16 nt + (lane & 15) - l is  pA[r] * pN[r] * pS[r]^nt  with pA = p2^(lane&15), pS = p2^16,
  // pN = p2^(-l) (advanced by one factor conj(p2) per l): one complex product per (r, tile) instead of a power
  cplx pA[4], pN[4], pC[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long tt = t0 + 16 * wave + (lane >> 4) + 4 * r;
    cplx Ra = {1.0, 0.0}, Rb = {0.0, 0.0};
    if (tt < n_times) {
      const double* rr = RaRb + tt * rotor_stride;
      Ra = {rr[0], rr[1]};
      Rb = {rr[2], rr[3]};
    }
    double ra, rb;
    cplx ea, eb;
    spinor_polar(Ra, Rb, ra, rb, ea, eb);
    const cplx p2 = {ra * ra - rb * rb, -2.0 * ra * rb};
    pA[r] = cpow_unit(p2, lane & 15);
    pC[r] = cconj(p2);
    pN[r] = cpow_unit(p2, -ell_min);
  }

  // rows of the next l are requested while the products of the current one run (PF: a quarter row fits the registers)
  constexpr bool PF = MAXNT <= 3;
  constexpr int PFN = PF ? 4 * MAXNT : 1;
  double2 fpre[PFN];
  auto prefetch = [&](int ell) {
    const int n = 2 * ell + 1, kpad = 4 * ((n + 3) / 4), cq = (kpad + 3) / 4;
    const double* src = data + (tg * ld + ((long long)ell * ell - (long long)ell_min * ell_min)) * 2;
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
      const int c = q * cq + u;
      fpre[u] = (live && u < cq && c < n) ? *reinterpret_cast<const double2*>(src + 2 * c) : double2{0.0, 0.0};
    }
  };
  if (PF) prefetch(ell_min);

  for (int ell = ell_min; ell <= ell_max; ++ell) {
    const int n = 2 * ell + 1;
    const int kpad = 4 * ((n + 3) / 4);
    const int ntl = (n + 15) / 16;
    int pd = 16 * ntl;
    while ((pd & 31) != 16) ++pd;
    const long long col0 = (long long)ell * ell - (long long)ell_min * ell_min;
    __syncthreads();  // previous l fully stored
    // ---- B operands of this l: B1[k = m'][mu] = Delta[mu][m'], B2[k = mu][m] = Delta[mu][m]
    {
      const double* src = btab + boff[ell];
      const int nb = 2 * kpad * pd;
      for (int e = tid; e < nb; e += 256) Bs[e] = src[e];
    }
    // ---- stage rows: f_m' p1^m' -> A image (Re row tl, Im row 64 + tl); pad columns zeroed.
    // All loads of a thread are issued before the dependent phase products (one memory latency per l, not per element).
    if (PF) {
      const int cq = (kpad + 3) / 4;
      cplx w = cpow_unit(p1, q * cq - ell);
#pragma unroll
      for (int u = 0; u < PFN; ++u) {
        const int c = q * cq + u;
        if (u < cq && c < kpad) {
          As[tl * geo.pa + c] = fpre[u].x * w.re - fpre[u].y * w.im;
          As[(64 + tl) * geo.pa + c] = fpre[u].x * w.im + fpre[u].y * w.re;
        }
        w = cmul(w, p1);
      }
    } else {
      const int cq = (kpad + 3) / 4;
      const int c_lo = q * cq, c_hi = (c_lo + cq < kpad) ? c_lo + cq : kpad;
      const double* src = data + (tg * ld + col0) * 2;
      for (int cb = c_lo; cb < c_hi; cb += RM_LD) {
        double2 f[RM_LD];
#pragma unroll
        for (int u = 0; u < RM_LD; ++u) {
          const int c = cb + u;
          f[u] = (live && c < n && c < c_hi) ? *reinterpret_cast<const double2*>(src + 2 * c) : double2{0.0, 0.0};
        }
        cplx w = cpow_unit(p1, cb - ell);
#pragma unroll
        for (int u = 0; u < RM_LD; ++u) {
          const int c = cb + u;
          if (c < c_hi) {
            As[tl * geo.pa + c] = f[u].x * w.re - f[u].y * w.im;
            As[(64 + tl) * geo.pa + c] = f[u].x * w.im + f[u].y * w.re;
          }
          w = cmul(w, p1);
        }
      }
    }
    __syncthreads();
    if (PF && ell < ell_max) prefetch(ell + 1);
    // ---- two wave-local products, in place in the A image
#pragma unroll 1
    for (int stage = 0; stage < 2; ++stage) {
      const double* B = Bs + stage * kpad * pd;
      const double* a_re = As + (16 * wave + (lane & 15)) * geo.pa + (lane >> 4);
      const double* a_im = a_re + 64 * geo.pa;
      const double* bp = B + (lane >> 4) * pd + (lane & 15);
      v4dr acc_re[MAXNT], acc_im[MAXNT];
#pragma unroll
      for (int nt = 0; nt < MAXNT; ++nt) {
        acc_re[nt] = v4dr{0, 0, 0, 0};
        acc_im[nt] = v4dr{0, 0, 0, 0};
      }
      for (int s = 0; s < kpad / 4; ++s) {
        const double ar = a_re[4 * s], ai = a_im[4 * s];
#pragma unroll
        for (int nt = 0; nt < MAXNT; ++nt) {
          if (nt < ntl) {
            const double b = bp[4 * s * pd + 16 * nt];
            acc_re[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, b, acc_re[nt], 0, 0, 0);
            acc_im[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b, acc_im[nt], 0, 0, 0);
          }
        }
      }
      // first product: phase exp(-i mu b); second product: the phase p3^m is applied by the store pass.
      // accumulator element r of tile nt = (row 16 wave + (lane>>4) + 4 r, column 16 nt + (lane&15)); write back in place
      cplx wr[4], pS[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wr[r] = cmul(pA[r], pN[r]);
        cplx s2 = cmul(pC[r], pC[r]);  // conj(p2)^2 ... ^16, conjugated back: p2^16
        s2 = cmul(s2, s2);
        s2 = cmul(s2, s2);
        s2 = cmul(s2, s2);
        pS[r] = cconj(s2);
      }
#pragma unroll
      for (int nt = 0; nt < MAXNT; ++nt) {
        if (nt < ntl) {
          const int col = 16 * nt + (lane & 15);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double xr = acc_re[nt][r], xi = acc_im[nt][r];
            double* dst = As + (16 * wave + (lane >> 4) + 4 * r) * geo.pa + col;
            if (col < kpad) {
              if (stage == 0) {
                dst[0] = xr * wr[r].re - xi * wr[r].im;
                dst[64 * geo.pa] = xr * wr[r].im + xi * wr[r].re;
              } else {
                dst[0] = xr;
                dst[64 * geo.pa] = xi;
              }
            }
            wr[r] = cmul(wr[r], pS[r]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) pN[r] = cmul(pN[r], pC[r]);  // p2^-(l+1)
    __syncthreads();
    // ---- write back (thread <-> (tl, quarter)); exact branches for pure z rotations / pi flips re-read the input
    if (live) {
      const int cq = (n + 3) / 4;
      const int c_lo = q * cq, c_hi = (c_lo + cq < n) ? c_lo + cq : n;
      double* dst = data + (tg * ld + col0) * 2;
      if (!(z_only || flip)) {
        cplx w = cpow_unit(p3, c_lo - ell);
        for (int c = c_lo; c < c_hi; ++c) {
          const double xr = As[tl * geo.pa + c], xi = As[(64 + tl) * geo.pa + c];
          *reinterpret_cast<double2*>(dst + 2 * c) = double2{xr * w.re - xi * w.im, xr * w.im + xi * w.re};
          w = cmul(w, p3);
        }
      } else if (z_only) {
        for (int c = c_lo; c < c_hi; ++c) {  // D_mm = ea^(2m)
          const double2 f = *reinterpret_cast<const double2*>(dst + 2 * c);
          const cplx o = cmul(cplx{f.x, f.y}, cpow_unit(ea2, c - ell));
          *reinterpret_cast<double2*>(dst + 2 * c) = double2{o.re, o.im};
        }
      }
    }
    if (flip && !z_only) {
      // D_{-m,m} = (-1)^(l-m) eb^(2m): out_m = f_{-m} D_{-m,m}; read everything before anyone of the row writes
      cplx o[MAXNT * 16 / 4 + 1];
      const int cq = (n + 3) / 4;
      const int c_lo = q * cq, c_hi = (c_lo + cq < n) ? c_lo + cq : n;
      double* dst = data + (tg * ld + col0) * 2;
      int k = 0;
      if (live)
        for (int c = c_lo; c < c_hi; ++c, ++k) {
          const int m = c - ell;
          const double2 f = *reinterpret_cast<const double2*>(dst + 2 * (n - 1 - c));
          cplx w = cpow_unit(eb2, m);
          if ((ell - m) & 1) w = {-w.re, -w.im};
          o[k] = cmul(cplx{f.x, f.y}, w);
        }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();  // the 4 quarters of a row live in 4 adjacent lanes of one wave
      k = 0;
      if (live)
        for (int c = c_lo; c < c_hi; ++c, ++k) *reinterpret_cast<double2*>(dst + 2 * c) = double2{o[k].re, o[k].im};
    }
  }
}

// LDS geometry for ell_max; returns 0 if unsupported
size_t rotate_mfma_lds_bytes(int ell_max, RotGeom& geo) {
  const int n = 2 * ell_max + 1;
  const int ntl = (n + 15) / 16;
  if (ntl > RM_MAX_NT) return 0;
  const int kpad = 4 * ((n + 3) / 4);
  int pa = kpad;
  while ((pa & 3) != 2) ++pa;
  int pd = 16 * ntl;
  while ((pd & 31) != 16) ++pd;
  geo.pa = pa;
  geo.b_doubles = 2 * kpad * pd;
  const size_t bytes = sizeof(double) * ((size_t)128 * pa + geo.b_doubles);
  return bytes <= 160u * 1024u ? bytes : 0;
}

int rotate_mfma_supported(int ell_max) {
  RotGeom geo;
  return rotate_mfma_lds_bytes(ell_max, geo) != 0;
}

// size/offset (doubles) of the packed B images of one l
void rotate_mfma_table_shape(int ell, int* kpad, int* pd) {
  const int n = 2 * ell + 1;
  *kpad = 4 * ((n + 3) / 4);
  int p = 16 * ((n + 15) / 16);
  while ((p & 31) != 16) ++p;
  *pd = p;
}

hipError_t launch_rotate_modes_mfma(hipStream_t stream, double* data, long long n_times, long long ld, int ell_min,
                                    int ell_max, const double* RaRb, long long rotor_stride, const double* btab,
                                    const long long* boff) {
  if (n_times <= 0) return hipSuccess;
  RotGeom geo;
  const size_t lds = rotate_mfma_lds_bytes(ell_max, geo);
  if (!lds) return hipErrorInvalidValue;
  const long long blocks = (n_times + RM_TB - 1) / RM_TB;
  const int ntl = (2 * ell_max + 1 + 15) / 16;
#define RM_LAUNCH(NTMAX)                                                                                                 \
  {                                                                                                                      \
    hipError_t e = allow_dynamic_lds((const void*)rotate_modes_mfma_kernel<NTMAX>);                            \
    if (e != hipSuccess) return e;                                                                                       \
    hipLaunchKernelGGL(rotate_modes_mfma_kernel<NTMAX>, dim3((unsigned)blocks), dim3(256), lds, stream, data, n_times, ld, \
                       ell_min, ell_max, RaRb, rotor_stride, btab, boff, geo);                                           \
  }
  if (ntl <= 2)
    RM_LAUNCH(2)
  else if (ntl <= 3)
    RM_LAUNCH(3)
  else
    RM_LAUNCH(5)
#undef RM_LAUNCH
  return hipGetLastError();
}

}  // namespace bms
