// Separable SWSH analysis on the equiangular grid (spinsfast.map2salm, scri/waveform_grid.py:303-307,
// scri/asymptotic_bondi_data/transformations.py:419-429).
//
// The Huffenberger-Wandelt analysis collapses exactly (oracle/spinsfast_ref.py, tests/test_oracle_wigner.py) to
//     a_lm = sum_j [q_j / n_phi] sLambda_lm(theta_j) F_m(theta_j),      F_m(theta_j) = sum_k f_jk exp(-i m phi_k)
// with real theta-quadrature weights q_j.  Step 1 (phi-DFT) is a thin real GEMM [rows n_theta x 2 n_phi] . [2 n_phi x
// 2 (2L+1)] against the matrix built by dft_matrix_kernel (run through dgemm_mfma_kernel); step 2 is
// theta_quadrature_kernel below.  Compared with the dense quadrature GEMM this is ~10x less arithmetic and sums
// n_phi and n_theta terms per output instead of 2 n_pix (shorter fp64 chains => smaller rounding error).
#include "wigner.h"
#include <cstdlib>
#include "kernels.h"

namespace bms {

// B[2k][2mi] = cos(m phi_k), B[2k][2mi+1] = -sin(m phi_k), B[2k+1][2mi] = sin(m phi_k), B[2k+1][2mi+1] = cos(m phi_k);
// mi = m + L, phi_k = 2 pi k / n_phi; the angle is reduced in integers before the sincos.
__global__ __launch_bounds__(256) void dft_matrix_kernel(int n_phi, int L, double* __restrict__ B, long long ldb) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nm = 2 * L + 1;
  if (id >= n_phi * nm) return;
  const int k = id / nm, mi = id % nm, m = mi - L;
  long long r = ((long long)m * k) % n_phi;
  if (r < 0) r += n_phi;
  double s, c;
  sincospi(2.0 * (double)r / (double)n_phi, &s, &c);
  double* r0 = B + (2LL * k) * ldb + 2 * mi;
  double* r1 = r0 + ldb;
  r0[0] = c;
  r0[1] = -s;
  r1[0] = s;
  r1[1] = c;
}

hipError_t launch_dft_matrix(hipStream_t stream, int n_phi, int L, double* B, long long ldb) {
  const int n = n_phi * (2 * L + 1);
  hipLaunchKernelGGL(dft_matrix_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n_phi, L, B, ldb);
  return hipGetLastError();
}

// T[o][j] = w_theta[j] * Re sYlm_o(theta_j, phi = 0);  Y: c16[n_theta][n_out] from swsh_kernel<0> at the rotors R(theta_j, 0)
__global__ __launch_bounds__(256) void theta_table_kernel(const double* __restrict__ Y, const double* __restrict__ w_theta,
                                                          int n_theta, int n_out, double* __restrict__ T) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_theta * n_out) return;
  const int o = id / n_theta, j = id % n_theta;
  T[(long long)o * n_theta + j] = w_theta[j] * Y[((long long)j * n_out + o) * 2];
}

hipError_t launch_theta_table(hipStream_t stream, const double* Y, const double* w_theta, int n_theta, int n_out, double* T) {
  const int n = n_theta * n_out;
  hipLaunchKernelGGL(theta_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, Y, w_theta, n_theta, n_out, T);
  return hipGetLastError();
}

// a[t][o] = sum_j T[o][j] F[t][j][m(o)].  Thread <-> output mode o with its T row in registers; the F tile of one time
// step (n_theta x (2L+1) complex, 19.5 kB at cfg3) is staged through LDS; outputs are written 16 B per lane, coalesced.
template <int NT, int MAXT>
__global__ __launch_bounds__(MAXT) void theta_quadrature_kernel(const double* __restrict__ F, long long n_rows, int n_theta,
                                                                int nm /* 2L+1 */, int n_out,
                                                                const int* __restrict__ m_index /* [n_out] */,
                                                                const double* __restrict__ T, double* __restrict__ out,
                                                                long long ldo) {
  extern __shared__ double Fs[];  // [n_theta][2 nm]
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = o < n_out;
  double tj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) tj[j] = (live && j < n_theta) ? T[(long long)o * n_theta + j] : 0.0;
  const int mi = live ? m_index[o] : 0;
  const int tile = n_theta * 2 * nm;  // doubles
  // rows n_theta..NT-1 of the LDS tile stay zero (and carry zero weights): the sum below needs no bounds test, and a
  // branch around an LDS read would make the compiler wait for every read before issuing the next
  for (int e = tile + threadIdx.x; e < NT * 2 * nm; e += blockDim.x) Fs[e] = 0.0;
  for (long long t = blockIdx.y; t < n_rows; t += gridDim.y) {
    const double* src = F + t * tile;
    __syncthreads();  // previous tile fully consumed
    for (int e = threadIdx.x * 2; e < tile; e += blockDim.x * 2)
      *reinterpret_cast<double2*>(Fs + e) = *reinterpret_cast<const double2*>(src + e);
    __syncthreads();
    double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
#pragma unroll
    for (int j = 0; j + 1 < NT; j += 2) {
      const double2 f0 = *reinterpret_cast<const double2*>(Fs + (j * nm + mi) * 2);
      const double2 f1 = *reinterpret_cast<const double2*>(Fs + ((j + 1) * nm + mi) * 2);
      ar = fma(tj[j], f0.x, ar);
      ai = fma(tj[j], f0.y, ai);
      br = fma(tj[j + 1], f1.x, br);
      bi = fma(tj[j + 1], f1.y, bi);
    }
    if (live) *reinterpret_cast<double2*>(out + t * ldo + 2LL * o) = double2{ar + br, ai + bi};
  }
}

hipError_t launch_theta_quadrature(hipStream_t stream, const double* F, long long n_rows, int n_theta, int nm, int n_out,
                                   const int* m_index, const double* T, double* out, long long ldo) {
  if (n_rows <= 0 || n_out <= 0) return hipSuccess;
  const int nt_pad = n_theta <= 24 ? 24 : n_theta <= 40 ? 40 : n_theta <= 72 ? 72 : 104;
  const size_t lds = sizeof(double) * (size_t)nt_pad * 2 * nm;
  const long long by = n_rows < 2048 ? n_rows : 2048;
  // the T row lives in registers: NT doubles per thread, so larger grids get smaller workgroups
#define LAUNCH_TQ(NT, MAXT)                                                                                              \
  {                                                                                                                      \
    int threads = ((n_out + 63) / 64) * 64;                                                                              \
    if (threads > MAXT) threads = MAXT;                                                                                  \
    const int bx = (n_out + threads - 1) / threads;                                                                      \
    hipError_t e = allow_dynamic_lds((const void*)theta_quadrature_kernel<NT, MAXT>);                            \
    if (e != hipSuccess) return e;                                                                                       \
    hipLaunchKernelGGL((theta_quadrature_kernel<NT, MAXT>), dim3(bx, (unsigned)by), dim3(threads), lds, stream, F, n_rows,  \
                       n_theta, nm, n_out, m_index, T, out, ldo);                                                        \
  }
  if (n_theta <= 24)
    LAUNCH_TQ(24, 1024)
  else if (n_theta <= 40)
    LAUNCH_TQ(40, 1024)
  else if (n_theta <= 72)
    LAUNCH_TQ(72, 512)
  else if (n_theta <= 104)
    LAUNCH_TQ(104, 256)
  else
    return hipErrorInvalidValue;
#undef LAUNCH_TQ
  return hipGetLastError();
}



// =====================================================================================================================
// Fused analysis: one workgroup per time step, the grid row never leaves the chip between the two steps.
//
//   step 0 (fold):  the ring samples are folded about phi = 0:  e_k = x_k + x_{n-k},  o_k = x_k - x_{n-k}  (k = 1..n/2;
//                   e_0 = x_0, o_0 = 0; the Nyquist sample of an even ring is its own partner), because
//                   C_m = sum_k x_k cos(m phi_k) = sum_{k <= n/2} e_k cos(m phi_k),  S_m = sum_k x_k sin(m phi_k) =
//                   sum_{k <= n/2} o_k sin(m phi_k):  half the terms.  Real and imaginary parts are separate real rows.
//   step 1 (MFMA):  C = E . cos, S = O . sin for m = 1..L as two real products [2 n_theta x n/2+1] x [n/2+1 x L]
//                   (L = 16: exactly one 16-column tile each; 2.9x fewer MFMAs than the unfolded cos|sin form, 8x
//                   fewer than the interleaved complex form).  The operand rows are ordered so that the four results
//                   a lane holds are (Re, Im) of two rings, and both signs of m come out of registers:
//                   F_{+-m}(j) = (C_re +- S_im) + i (C_im -+ S_re).   m = 0 is a plain sum of e_k.
//   step 2 (VALU):  a_o = sum_j T[o][j] F_{m(o)}(j): one thread per output mode, its T row in registers, F read as
//                   16-byte (re, im) pairs.
// HBM traffic = the algorithmic minimum: read the grid row once (16 n_pix B), write the modes once (16 n_out B).
// =====================================================================================================================
typedef double v4d_t __attribute__((ext_vector_type(4)));

constexpr int F_PA = 26;    // LDS pitch (doubles) of the folded operand rows: 2 x odd >= 24 -> conflict-free fragment reads
constexpr int F_KSMAX = 6;  // k-steps of 4 along the folded ring: n_phi / 2 + 1 <= 24 (kernels are built for 3, 5 and 6)
constexpr int F_EPT_MAX = 4;  // (ring, sample pair) items per thread: n_theta (n_phi/2 + 1) <= 40 x 21 <= 4 x 256 threads (kernels for 2, 3, 4)

struct FusedGeom {
  int n_theta, n_phi, L, n_out;
  int mt;  // 16-row operand tiles (8 rings x {re, im} each)
  int nk;  // folded ring length n_phi / 2 + 1
  int ks;  // k-steps of 4 covering nk, rounded up to a built kernel: 3, 5 or 6 (tables are zero padded to it)
  int n_sec;  // output modes beyond one per thread (<= 64): handled by the first threads with their T rows in LDS
  int spin;   // spin weight of the field: phase of the de-duplicated pole pixels (used with col_of_pixel)
};

__host__ __device__ inline int fused_pd(int L) { return L <= 16 ? 16 : 48; }  // = 16 mod 32: B fragment halves on different banks

// Dc[k][c] = cos((c+1) phi_k), Ds[k][c] = sin((c+1) phi_k) for k < n_phi/2 + 1, c < L; both [4 ks][pd], zero padded
__global__ __launch_bounds__(256) void dft_cs_matrix_kernel(int n_phi, int L, int ks, int pd, double* __restrict__ D) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int nk = n_phi / 2 + 1;
  if (id >= nk * L) return;
  const int k = id / L, c = id % L;
  const long long r = ((long long)(c + 1) * k) % n_phi;
  double sn, co;
  sincospi(2.0 * (double)r / (double)n_phi, &sn, &co);
  D[(long long)k * pd + c] = co;
  D[(long long)(4 * ks + k) * pd + c] = sn;
}

// operand row of ring j, real part (imaginary part: + 4): rows g + 4 r of a 16-row MFMA tile sit in the lanes with
// lane >> 4 = g, so (re, im) of rings g and g + 4 of the tile are the four results one lane holds
__device__ __forceinline__ int fused_row(int j) { return (j >> 3) * 16 + (j & 3) + 8 * ((j >> 2) & 1); }

// NT: max n_theta (T row length held in registers); NTC: 16-column tiles covering m = 1..L; KS: k-steps.
// Everything inside the row loop is branch-free on purpose (zero padding instead of bounds tests): a uniform branch
// around an LDS read makes the compiler wait for each read before issuing the next.
template <int NT, int NTC, int KS, int F_EPT>
__device__ __forceinline__ void analysis_fused_body(const double* __restrict__ G, long long ldg, long long n_rows,
                                                    const FusedGeom& g, const int* __restrict__ m_index,
                                                    const double* __restrict__ T, const double* __restrict__ Dg,
                                                    double* __restrict__ out, long long ldo,
                                                    const int* __restrict__ col_of_pixel) {
  constexpr int PD = NTC == 1 ? 16 : 48;
  constexpr int PJ = NT + 1;  // odd pitch (complex) of an F_m row: consecutive m on distinct 16-byte slots
  extern __shared__ double lds[];
  double* Es = lds;                       // [16 mt][F_PA]
  double* Os = Es + 16 * g.mt * F_PA;     // [16 mt][F_PA]
  double* Dc = Os + 16 * g.mt * F_PA;     // [4 ks][PD]
  double* Dn = Dc + 4 * g.ks * PD;        // [4 ks][PD]
  double2* Fs = reinterpret_cast<double2*>(Dn + 4 * g.ks * PD);  // [2L+1][PJ]
  double* T2 = reinterpret_cast<double*>(Fs + (2 * g.L + 1) * PJ);  // [n_sec][PJ]
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nwaves = nthreads >> 6;
  const int fi = lane & 15, fk = lane >> 4;

  // one-time set-up: zero the operands (their padding stays zero), copy the DFT matrices, T row to registers
  for (int e = tid; e < 2 * 16 * g.mt * F_PA; e += nthreads) Es[e] = 0.0;
  for (int e = tid; e < 2 * 4 * g.ks * PD; e += nthreads) Dc[e] = Dg[e];
  for (int e = tid; e < (2 * g.L + 1) * PJ; e += nthreads) Fs[e] = double2{0.0, 0.0};
  const int o = tid;
  const bool live = o < g.n_out;
  double tj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) tj[j] = (live && j < g.n_theta) ? T[(long long)o * g.n_theta + j] : 0.0;
  const double2* fp = Fs + (live ? m_index[o] : 0) * PJ;
  // a second output mode for the first n_sec threads (keeps the workgroup at 4 waves = one per SIMD, so that two
  // workgroups share a CU and one's MFMA step runs under the other's loads and quadrature)
  const bool sec = tid < g.n_sec;
  const int o2 = nthreads + tid;
  for (int e = tid; e < g.n_sec * PJ; e += nthreads) {
    const int r = e / PJ, j = e - r * PJ;
    T2[e] = j < g.n_theta ? T[(long long)(nthreads + r) * g.n_theta + j] : 0.0;
  }
  const double2* fp2 = Fs + (sec ? m_index[o2] : 0) * PJ;
  const double* t2 = T2 + (sec ? tid : 0) * PJ;

  // this thread's (ring, sample pair) items: global offsets of the two samples, LDS slot of the folded pair.
  // A sample that is its own partner (k = 0, Nyquist) is loaded twice and its partner weighted by 0.
  // Registers are what limits the rows in flight (two rows ahead are being fetched while one is processed), so the
  // per-item constants are kept small: a bit per item for "own partner", and the two pole-ring phases a thread can
  // meet (items of the north ring come with q = 0 only, those of the south ring with one q per thread).
  int slot[F_EPT], off1[F_EPT], off2[F_EPT];
  unsigned self_mask = 0;  // bit q: sample pair q is one sample (k = 0 or Nyquist): partner weighted by 0
  int q_north = -1, q_south = -1;
  double2 pn_a{1.0, 0.0}, pn_b{1.0, 0.0}, ps_a{1.0, 0.0}, ps_b{1.0, 0.0};
#pragma unroll
  for (int q = 0; q < F_EPT; ++q) {
    const int e = tid + q * nthreads;
    const bool ok = e < g.n_theta * g.nk;
    const int j = ok ? e / g.nk : 0, kk = ok ? e - j * g.nk : 0;
    const int k2 = kk == 0 ? 0 : g.n_phi - kk;
    slot[q] = ok ? fused_row(j) * F_PA + kk : -1;
    const int g1 = j * g.n_phi + kk, g2 = j * g.n_phi + k2;
    off1[q] = 2 * (col_of_pixel ? col_of_pixel[g1] : g1);
    off2[q] = 2 * (col_of_pixel ? col_of_pixel[g2] : g2);
    if (k2 == kk) self_mask |= 1u << q;
    // a pole ring is stored as its pixel k = 0: pixel k is that value times e^{-i s phi_k} (north) / e^{+i s phi_k} (south)
    if (col_of_pixel && ok && (j == 0 || j == g.n_theta - 1)) {
      const double sg = j == 0 ? -1.0 : 1.0;
      const double a1 = sg * (double)((g.spin * kk) % g.n_phi) / (double)g.n_phi;
      const double a2 = sg * (double)((g.spin * k2) % g.n_phi) / (double)g.n_phi;
      double2 fa, fb;
      sincospi(2.0 * a1, &fa.y, &fa.x);
      sincospi(2.0 * a2, &fb.y, &fb.x);
      if (j == 0 && g.n_theta > 1) {
        q_north = q;
        pn_a = fa;
        pn_b = fb;
      } else {
        q_south = q;
        ps_a = fa;
        ps_b = fb;
      }
    }
  }
  // two rows in flight where the registers allow it (up to 3 items per thread), else one
  constexpr int AHEAD = F_EPT <= 3 ? 2 : 1;
  double2 pa0[F_EPT], pb0[F_EPT], pa1[F_EPT], pb1[F_EPT];
  auto fetch = [&](long long t, double2(&pa)[F_EPT], double2(&pb)[F_EPT]) {
#pragma unroll
    for (int q = 0; q < F_EPT; ++q) {
      pa[q] = *reinterpret_cast<const double2*>(G + t * ldg + off1[q]);
      pb[q] = *reinterpret_cast<const double2*>(G + t * ldg + off2[q]);
    }
  };
  const long long stride = gridDim.x;
  auto one_row = [&](const long long t, double2(&pa)[F_EPT], double2(&pb)[F_EPT]) {
    // ---- step 0: fold into the operands
#pragma unroll
    for (int q = 0; q < F_EPT; ++q) {
      const double2 fa = q == q_north ? pn_a : (q == q_south ? ps_a : double2{1.0, 0.0});
      const double2 fb = q == q_north ? pn_b : (q == q_south ? ps_b : double2{1.0, 0.0});
      const double w = (self_mask >> q) & 1u ? 0.0 : 1.0;
      const double ax = pa[q].x * fa.x - pa[q].y * fa.y, ay = pa[q].x * fa.y + pa[q].y * fa.x;
      const double bx = w * (pb[q].x * fb.x - pb[q].y * fb.y), by = w * (pb[q].x * fb.y + pb[q].y * fb.x);
      if (slot[q] >= 0) {
        Es[slot[q]] = ax + bx;
        Es[slot[q] + 4 * F_PA] = ay + by;
        Os[slot[q]] = w * ax - bx;
        Os[slot[q] + 4 * F_PA] = w * ay - by;
      }
    }
    __syncthreads();  // operands complete; every thread has finished step 2 of the previous row
    const long long tn = t + AHEAD * stride;
    if (tn < n_rows) fetch(tn, pa, pb);
    // ---- step 1: one 16-row tile (8 rings) per wave trip
    for (int tm = wave; tm < g.mt; tm += nwaves) {
      const double* ep = Es + (tm * 16 + fi) * F_PA + fk;
      const double* op = Os + (tm * 16 + fi) * F_PA + fk;
      const double* cp = Dc + fk * PD + fi;
      const double* sp = Dn + fk * PD + fi;
      v4d_t ac[NTC], as[NTC];
#pragma unroll
      for (int n = 0; n < NTC; ++n) ac[n] = as[n] = v4d_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double ae = ep[4 * s], ao = op[4 * s];
#pragma unroll
        for (int n = 0; n < NTC; ++n) {
          ac[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, cp[4 * s * PD + 16 * n], ac[n], 0, 0, 0);
          as[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao, sp[4 * s * PD + 16 * n], as[n], 0, 0, 0);
        }
      }
      // m = 0: plain sums of the folded rows (16 lanes, one operand row each)
      if (fk == 0) {
        const double* rp = Es + (tm * 16 + fi) * F_PA;
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int k = 0; k < 4 * KS; k += 2) {
          c0 += rp[k];
          c1 += rp[k + 1];
        }
        const int ring = tm * 8 + (fi & 3) + 4 * (fi >> 3), part = (fi >> 2) & 1;
        reinterpret_cast<double*>(Fs + g.L * PJ + ring)[part] = c0 + c1;
      }
      // both signs of m from registers: results r = 2h (re), 2h+1 (im) of ring 8 tm + fk + 4h, column m = 16 n + fi + 1
#pragma unroll
      for (int n = 0; n < NTC; ++n) {
        const int m = 16 * n + fi + 1;
        if (m <= g.L) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int ring = tm * 8 + fk + 4 * h;
            const double cre = ac[n][2 * h], cim = ac[n][2 * h + 1], sre = as[n][2 * h], sim = as[n][2 * h + 1];
            Fs[(g.L + m) * PJ + ring] = double2{cre + sim, cim - sre};
            Fs[(g.L - m) * PJ + ring] = double2{cre - sim, cim + sre};
          }
        }
      }
    }
    __syncthreads();  // F complete
    // ---- step 2: theta quadrature
    if (live) {
      // rings beyond n_theta carry zero weights and zero F: no bounds tests
      double pr0 = 0.0, pi0 = 0.0, pr1 = 0.0, pi1 = 0.0;
#pragma unroll
      for (int j = 0; j + 1 < NT; j += 2) {
        const double2 v0 = fp[j], v1 = fp[j + 1];
        pr0 = fma(tj[j], v0.x, pr0);
        pi0 = fma(tj[j], v0.y, pi0);
        pr1 = fma(tj[j + 1], v1.x, pr1);
        pi1 = fma(tj[j + 1], v1.y, pi1);
      }
      *reinterpret_cast<double2*>(out + t * ldo + 2LL * o) = double2{pr0 + pr1, pi0 + pi1};
    }
    if (sec) {
      double pr0 = 0.0, pi0 = 0.0, pr1 = 0.0, pi1 = 0.0;
#pragma unroll
      for (int j = 0; j + 1 < NT; j += 2) {
        const double2 v0 = fp2[j], v1 = fp2[j + 1];
        const double w0 = t2[j], w1 = t2[j + 1];
        pr0 = fma(w0, v0.x, pr0);
        pi0 = fma(w0, v0.y, pi0);
        pr1 = fma(w1, v1.x, pr1);
        pi1 = fma(w1, v1.y, pi1);
      }
      *reinterpret_cast<double2*>(out + t * ldo + 2LL * o2) = double2{pr0 + pr1, pi0 + pi1};
    }
  };
  long long t = blockIdx.x;
  if (t < n_rows) fetch(t, pa0, pb0);
  if constexpr (AHEAD == 2) {
    if (t + stride < n_rows) fetch(t + stride, pa1, pb1);
    __syncthreads();
    for (; t < n_rows; t += 2 * stride) {
      one_row(t, pa0, pb0);
      if (t + stride < n_rows) one_row(t + stride, pa1, pb1);
    }
  } else {
    __syncthreads();
    for (; t < n_rows; t += stride) one_row(t, pa0, pb0);
  }
}

template <int NT, int KS, int EPT>
__global__ __launch_bounds__(512) void analysis_fused_kernel(
    const double* __restrict__ G, long long ldg, long long n_rows, FusedGeom g, const int* __restrict__ m_index,
    const double* __restrict__ T, const double* __restrict__ Dg, double* __restrict__ out, long long ldo,
    const int* __restrict__ col_of_pixel) {
  analysis_fused_body<NT, 1, KS, EPT>(G, ldg, n_rows, g, m_index, T, Dg, out, ldo, col_of_pixel);
}
// 16 < L <= 32: two column tiles per product
template <int NT, int KS, int EPT>
__global__ __launch_bounds__(512) void analysis_fused_wide_kernel(
    const double* __restrict__ G, long long ldg, long long n_rows, FusedGeom g, const int* __restrict__ m_index,
    const double* __restrict__ T, const double* __restrict__ Dg, double* __restrict__ out, long long ldo,
    const int* __restrict__ col_of_pixel) {
  analysis_fused_body<NT, 2, KS, EPT>(G, ldg, n_rows, g, m_index, T, Dg, out, ldo, col_of_pixel);
}

static int fused_ks(int n_phi) {
  const int ks = (n_phi / 2 + 1 + 3) / 4;
  return ks <= 3 ? 3 : ks <= 5 ? 5 : 6;
}

static void fused_geometry(int n_theta, int n_phi, int L, int n_out, FusedGeom& g, size_t& lds_bytes, int& pj) {
  g.n_theta = n_theta;
  g.n_phi = n_phi;
  g.L = L;
  g.n_out = n_out;
  g.mt = (n_theta + 7) / 8;
  g.nk = n_phi / 2 + 1;
  g.ks = fused_ks(n_phi);
  g.n_sec = (n_out > 256 && n_out <= 320) ? n_out - 256 : 0;
  pj = (n_theta <= 24 ? 24 : 40) + 1;
  lds_bytes = sizeof(double) * ((size_t)2 * 16 * g.mt * F_PA + (size_t)2 * 4 * g.ks * fused_pd(L) + (size_t)2 * (2 * L + 1) * pj + (size_t)g.n_sec * pj);
}

int fused_analysis_supported(int n_theta, int n_phi, int L, int n_out) {
  if (n_theta > 40 || n_phi < 2 || n_phi / 2 + 1 > 4 * F_KSMAX || n_out > 512 || L < 1 || L > 32) return 0;
  return 1;
}

// doubles in the cos|sin table of launch_dft_cs_matrix
size_t fused_dft_table_size(int n_phi, int L) {
  return (size_t)2 * 4 * fused_ks(n_phi) * fused_pd(L);
}

hipError_t launch_dft_cs_matrix(hipStream_t stream, int n_phi, int L, double* D) {
  const int nk = n_phi / 2 + 1, ks = fused_ks(n_phi);
  const int n = nk * L;
  hipLaunchKernelGGL(dft_cs_matrix_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n_phi, L, ks, fused_pd(L), D);
  return hipGetLastError();
}

// =====================================================================================================================
// The same analysis with the two halves of the work on different waves of one workgroup (one workgroup per CU, two time
// rows per trip), because the one-role kernel above is bound by its own chain -- fold, barrier, MFMA, barrier,
// quadrature -- not by memory (with its phases switched off one at a time: loads alone 0.37 ms at cfg3, the chain without
// loads 0.80 ms, together 0.82 ms).
//
//   front waves: wave w owns 16 rings of the 2 n_rings of a row pair as the 16 A-rows of its MFMA tile.  Lane (i, c) loads
//                the sample pair (ring i, k = 4 s + c | n_phi - k) itself, 16 bytes each, so the folded operands
//                e = x_k + x_{n-k}, o = x_k - x_{n-k} exist in registers in exactly the A-fragment layout: no LDS
//                operands, no barrier before the products.  Re and Im go through separate products (A = Re e, Im e,
//                Re o, Im o against the cos | sin twiddles, which the lane keeps in registers), 4 KS MFMAs per wave and
//                trip, no padding but the last tile's.  m = 0 is the lane sum of e reduced over c.  Results are combined
//                to F_{+-m} in registers and written to LDS.  A pair's samples are requested during the fold of the pair
//                before it.
//   back waves:  one thread per output mode, its T row in registers, both rows of the pair: the theta quadrature reading
//                F as 16-byte pairs, while the front waves already work on the next pair (F is double buffered; one
//                barrier per trip); they also prefetch into L2 for the front waves (see there).
//   pole rings of the de-duplicated grids (col_of_pixel) have a closed form, F_m = n_phi x value at m = -+spin, and take no
//   part in the products.
// =====================================================================================================================
struct SplitGeom {
  int n_theta, n_phi, L, n_out, nk, spin;
  int nf, nb;     // front and back waves
  int rings;      // rings per row that go through the products (n_theta, or n_theta - 2 with closed-form poles)
  int poles;      // 1: rings 0 and n_theta - 1 are stored as one value each
  int ahead;      // 1: the back waves prefetch into L2
};
constexpr int SPLIT_MAX_WAVES = 10;
#ifndef SPLIT_AUX
#define SPLIT_AUX 0  // cache policy bits of the sample requests (2 = non-temporal)
#endif

template <int NT, int KS>
__global__ __launch_bounds__(640, 1) void analysis_split_kernel(
    const double* __restrict__ G, long long ldg, long long n_rows, SplitGeom g, const int* __restrict__ m_index,
    const double* __restrict__ T, double* __restrict__ out, long long ldo, const int* __restrict__ col_of_pixel) {
  constexpr int PJ = NT + 1;  // odd pitch (complex) of an F_m row
  extern __shared__ double lds[];
  // [1 KB per wave: landing area of the prefetch requests, never read][2 buffers][2 rows][2L+1][PJ]
  double2* Fs = reinterpret_cast<double2*>(lds) + 64 * SPLIT_MAX_WAVES;
  const int fsz = (2 * g.L + 1) * PJ;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  for (int e = tid; e < 4 * fsz; e += blockDim.x) Fs[e] = double2{0.0, 0.0};
  __syncthreads();
  const long long n_pairs = (n_rows + 1) / 2;
  const long long stride = gridDim.x;
  long long p = blockIdx.x;
  // Roles.  A workgroup's waves go to the SIMDs round robin, so waves w, w + 4, w + 8 share one.  The front waves carry
  // the MFMA work (one tile of 4 KS products each): waves 0..3 first, one per SIMD, and a fifth and sixth on the SIMDs
  // that hold only two waves (6 and 7) -- with waves 0..4 in front, waves 0 and 4 made one SIMD the bottleneck.
  const int n_waves = g.nf + g.nb;
  const auto front_of = [&](int w) {  // tile of wave w, or -1
    const int extra = n_waves >= 8 ? 6 : 4;  // where the fifth and sixth front waves sit
    const int f = w < 4 ? w : (w >= extra && w < extra + 2 ? 4 + w - extra : -1);
    return f < g.nf ? f : -1;
  };
  const int front = front_of(wave);
  const bool is_front = front >= 0;
  int back = wave;  // rank among the waves that are not front
  for (int w = 0; w < wave; ++w)
    if (front_of(w) >= 0) --back;

  if (is_front) {
    const int fi = lane & 15, fk = lane >> 4;
    // A-row of this lane: ring slot q of the pair's 2 x rings regular rings
    const int q = 16 * front + fi;
    const bool okr = q < 2 * g.rings;
    const int r = okr ? q / g.rings : 0;
    const int j = (okr ? q - r * g.rings : 0) + g.poles;
    int offa[KS], offb[KS];
    double cs[KS], sn[KS];
    unsigned va = 0, vb = 0;  // bit s: sample k = 4 s + fk exists / has a partner other than itself
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = 4 * s + fk;
      const bool ok = okr && k < g.nk;
      const int k2 = k == 0 ? 0 : g.n_phi - k;
      const int g1 = ok ? j * g.n_phi + k : 0, g2 = ok ? j * g.n_phi + k2 : 0;
      offa[s] = 8 * (2 * (col_of_pixel ? col_of_pixel[g1] : g1) + r * (int)ldg);  // bytes from the pair's first row
      offb[s] = 8 * (2 * (col_of_pixel ? col_of_pixel[g2] : g2) + r * (int)ldg);
      if (ok) va |= 1u << s;
      if (ok && k2 != k) vb |= 1u << s;
      // B fragment: column m = fi + 1 of the cos | sin matrices, row k
      const int m = fi + 1;
      const long long rr = ((long long)m * k) % g.n_phi;
      double sv, cv;
      sincospi(2.0 * (double)rr / (double)g.n_phi, &sv, &cv);
      const bool okb = k < g.nk && m <= g.L;
      cs[s] = okb ? cv : 0.0;
      sn[s] = okb ? sv : 0.0;
    }
    // where the four results of a lane go: D rows fk + 4 v = ring slots 16 front + fk + 4 v, column m = fi + 1
    int fo[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int qv = 16 * front + fk + 4 * v;
      const bool ok = qv < 2 * g.rings && fi + 1 <= g.L;
      const int rv = ok ? qv / g.rings : 0;
      fo[v] = ok ? rv * fsz + (qv - rv * g.rings) + g.poles : -1;
    }
    const int f0 = okr ? r * fsz + g.L * PJ + j : -1;  // m = 0 of this lane's own ring (written by the lanes fk = 0)
    // closed-form poles: lanes 0..3 of the first front wave take (row, pole) = (lane >> 1, lane & 1)
    const bool pole_lane = g.poles && front == 0 && lane < 4;
    const int pole_m = (lane & 1) ? g.spin : -g.spin;
    const int pole_ring = (lane & 1) ? g.n_theta - 1 : 0;
    const int pole_off = pole_lane ? 8 * (2 * col_of_pixel[pole_ring * g.n_phi] + (lane >> 1) * (int)ldg) : 0;
    const int pole_fo = (pole_lane && pole_m >= -g.L && pole_m <= g.L) ? (lane >> 1) * fsz + (g.L + pole_m) * PJ + pole_ring : -1;
    // Buffer loads: the pair's two rows (t0, t0 + 1 with t0 = min(2 pp, n_rows - 2): the last pair of an odd series
    // overlaps the one before it) behind a wave-uniform descriptor, 32-bit lane offsets.  Flat loads made the compiler
    // carry a 64-bit address per load through the loop and spill; a spill reload inside the loop waits with vmcnt(0),
    // i.e. for every sample in flight.
    auto descriptor = [&](long long pp) {
      if (pp >= n_pairs) pp = n_pairs - 1;  // (past the end: any valid pair, the samples are never used)
      long long t = 2 * pp;
      if (t > n_rows - 2) t = n_rows - 2;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(G + t * ldg), 0, (int)(16 * ldg), 0x00020000);
    };
    double2 xa[KS], xb[KS], xp;
    {
      const auto rsrc = descriptor(p);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        xa[s] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, offa[s], 0, 0));
        xb[s] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, offb[s], 0, 0));
      }
      xp = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, pole_off, 0, 0));
    }
    for (int it = 0; p < n_pairs; p += stride, it ^= 1) {
      double2* Fb = Fs + it * 2 * fsz;
      // k-step by k-step: fold, ask for the same samples of the next pair (their registers are free from here on, and
      // the request is in flight under everything up to the next trip's fold), four products
      const auto next = descriptor(p + stride);
      const double2 pole_value = xp;
      xp = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(next, pole_off, 0, 0));
      v4d_t cr{0.0, 0.0, 0.0, 0.0}, ci = cr, sr = cr, si = cr;
      double e0x = 0.0, e0y = 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double wa = (va >> s) & 1u ? 1.0 : 0.0, wb = (vb >> s) & 1u ? 1.0 : 0.0;
        const double ex = wa * xa[s].x + wb * xb[s].x, ey = wa * xa[s].y + wb * xb[s].y;
        const double ox = wb * (xa[s].x - xb[s].x), oy = wb * (xa[s].y - xb[s].y);
        xa[s] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(next, offa[s], 0, SPLIT_AUX));
        xb[s] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(next, offb[s], 0, SPLIT_AUX));
        e0x += ex;
        e0y += ey;
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(ex, cs[s], cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ey, cs[s], ci, 0, 0, 0);
        sr = __builtin_amdgcn_mfma_f64_16x16x4f64(ox, sn[s], sr, 0, 0, 0);
        si = __builtin_amdgcn_mfma_f64_16x16x4f64(oy, sn[s], si, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // m = 0: sum over the four k-lanes of a ring
      e0x += __shfl_xor(e0x, 16);
      e0y += __shfl_xor(e0y, 16);
      e0x += __shfl_xor(e0x, 32);
      e0y += __shfl_xor(e0y, 32);
      if (fk == 0 && f0 >= 0) Fb[f0] = double2{e0x, e0y};
      const int m = fi + 1;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (fo[v] >= 0) {
          Fb[fo[v] + (g.L + m) * PJ] = double2{cr[v] + si[v], ci[v] - sr[v]};
          Fb[fo[v] + (g.L - m) * PJ] = double2{cr[v] - si[v], ci[v] + sr[v]};
        }
      }
      if (pole_fo >= 0) Fb[pole_fo] = double2{(double)g.n_phi * pole_value.x, (double)g.n_phi * pole_value.y};
      __syncthreads();  // F of this pair complete; the back waves have finished with the other buffer
    }
  } else {
    const int o = 64 * back + lane;
    const bool live = o < g.n_out;
    double tj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tj[j] = (live && j < g.n_theta) ? T[(long long)o * g.n_theta + j] : 0.0;
    const int fbase = (live ? m_index[o] : 0) * PJ;
    // Prefetch (g.ahead > 0).  The front waves' requests are 16-byte pieces, 16 to 64 different lines per instruction, in
    // an order the MFMA fragment dictates; the memory system serves whole rows requested in order much better.  The back
    // waves have the time and need no registers for it: LDS-DMA requests for the rows the front waves will ask for in
    // their next trip but one, 1 KB per instruction, into a landing area nobody reads -- a prefetch into this XCD's L2
    // that no s_waitcnt of the compiler knows about.  It only pays while three pairs of rows of every CU of the XCD fit
    // the L2 (the launcher decides): beyond that it evicts what it fetched.
    const unsigned landing = (unsigned)wave * 1024u;  // LDS byte address (the dynamic segment starts at 0)
    const int pair_bytes = (int)(16 * ldg);
    auto prefetch = [&](long long pp) {
      if (pp >= n_pairs) return;
      long long t = 2 * pp;
      if (t > n_rows - 2) t = n_rows - 2;
      const char* base = reinterpret_cast<const char*>(G + t * ldg);
      for (int c = 16 * o; c < pair_bytes; c += 1024 * g.nb) {
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(base + c), "s"(landing)
            : "memory");
      }
    };
    if (g.ahead) {
      prefetch(p + stride);
      prefetch(p + 2 * stride);
    }
    for (int it = 0; p < n_pairs; p += stride, it ^= 1) {
      __syncthreads();
      const double2* fa = Fs + it * 2 * fsz + fbase;
      const double2* fb = fa + fsz;
      double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const double2 u = fa[j], w = fb[j];
        ar = fma(tj[j], u.x, ar);
        ai = fma(tj[j], u.y, ai);
        br = fma(tj[j], w.x, br);
        bi = fma(tj[j], w.y, bi);
      }
      if (live) {
        long long t = 2 * p;
        if (t > n_rows - 2) t = n_rows - 2;
        *reinterpret_cast<double2*>(out + t * ldo + 2LL * o) = double2{ar, ai};
        *reinterpret_cast<double2*>(out + (t + 1) * ldo + 2LL * o) = double2{br, bi};
      }
      // (the front waves are in their next trip and have asked for pair p + 2 stride at its start)
      if (g.ahead) prefetch(p + 3 * stride);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no request may outlive the workgroup's LDS
  }
}

int split_analysis_supported(int n_theta, int n_phi, int L, int n_out, bool dedup_poles, SplitGeom& g, size_t& lds_bytes, int& nt) {
  if (n_theta < 3 || n_theta > 40 || n_phi < 2 || L < 1 || L > 16) return 0;
  // (n_rows >= 2 and 16 ldg < 2^31 are checked by the caller)
  g.n_theta = n_theta;
  g.n_phi = n_phi;
  g.L = L;
  g.n_out = n_out;
  g.nk = n_phi / 2 + 1;
  if (g.nk > 20) return 0;  // five k-steps: with six the kernel no longer fits the 168 registers of 10 waves per CU
  g.poles = dedup_poles ? 1 : 0;
  g.rings = n_theta - 2 * g.poles;
  g.nf = (2 * g.rings + 15) / 16;
  g.nb = (n_out + 63) / 64;
  if (g.nf > 6 || g.nf + g.nb > SPLIT_MAX_WAVES) return 0;
  nt = n_theta <= 24 ? 24 : (n_theta <= 38 ? 38 : 40);
  lds_bytes = sizeof(double2) * ((size_t)4 * (2 * L + 1) * (nt + 1) + 64 * SPLIT_MAX_WAVES);
  return 1;
}

template <typename K>
static hipError_t launch_fused_t(K kernel, hipStream_t stream, dim3 grid, dim3 block, size_t lds, const double* G, long long ldg,
                                 long long n_rows, const FusedGeom& g, const int* m_index, const double* T, const double* D,
                                 double* out, long long ldo, const int* col_of_pixel) {
  hipError_t e = allow_dynamic_lds((const void*)kernel);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, grid, block, lds, stream, G, ldg, n_rows, g, m_index, T, D, out, ldo, col_of_pixel);
  return hipGetLastError();
}

hipError_t launch_analysis_fused(hipStream_t stream, const double* G, long long ldg, long long n_rows, int n_theta, int n_phi,
                                 int L, int n_out, const int* m_index, const double* T, const double* D, double* out,
                                 long long ldo, const int* col_of_pixel, int spin, bool allow_split) {
  if (n_rows <= 0) return hipSuccess;
  FusedGeom g;
  size_t lds;
  int pj;
  fused_geometry(n_theta, n_phi, L, n_out, g, lds, pj);
  {
    SplitGeom sg;
    size_t slds;
    int nt;
    if (allow_split && n_rows >= 2 && ldg < (1LL << 26) && split_analysis_supported(n_theta, n_phi, L, n_out, col_of_pixel != nullptr, sg, slds, nt)) {
      sg.spin = spin;
      // Prefetch: measured at cfg3 it takes the gathered grid of the transformation from 0.67 to 0.52 ms per 1e5 rows (its
      // 16-byte pieces land on 64 different lines per request and are expensive as L2 misses), costs the same shape in
      // natural order 0.51 -> 0.58 ms (64-byte pieces: nothing to gain) and the small rows of cfg2 0.22 -> 0.27 ms.  So:
      // gathered grids whose rows are long, while three pairs of rows of each of the XCD's 32 CUs fit its 4 MiB L2.
      const long long pair_bytes = 16 * ldg;
      sg.ahead = BMS_PROBE_ENV("SCRI_AMD_SPLIT_PREFETCH") ? atoi(BMS_PROBE_ENV("SCRI_AMD_SPLIT_PREFETCH"))
                                                   : (col_of_pixel && pair_bytes >= (32 << 10) && 96 * pair_bytes <= (4LL << 20) ? 1 : 0);
      static const long long cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return (long long)n;
      }();
      const long long n_pairs = (n_rows + 1) / 2;
      // one workgroup per CU at cfg3 (10 waves); small shapes need fewer waves and get two or three
      const long long per_cu = 10 / (sg.nf + sg.nb) < 3 ? 10 / (sg.nf + sg.nb) : 3;
      const long long max_blocks = cus * per_cu;
      const dim3 sgrid((unsigned)(n_pairs < max_blocks ? n_pairs : max_blocks)), sblock(64 * (sg.nf + sg.nb));
      const int ks = (sg.nk + 3) / 4;
#define SPLIT_GO(NT, KS)                                                                                                   \
  {                                                                                                                        \
    hipError_t e = allow_dynamic_lds((const void*)analysis_split_kernel<NT, KS>);                                                                         \
    if (e != hipSuccess) return e;                                                                                         \
    hipLaunchKernelGGL((analysis_split_kernel<NT, KS>), sgrid, sblock, slds, stream, G, ldg, n_rows, sg, m_index, T, out, ldo, \
                       col_of_pixel);                                                                                      \
    return hipGetLastError();                                                                                              \
  }
#define SPLIT_KS(NT)              \
  {                               \
    if (ks <= 3) SPLIT_GO(NT, 3)  \
    SPLIT_GO(NT, 5)               \
  }
      if (nt == 24) SPLIT_KS(24)
      if (nt == 38) SPLIT_KS(38)
      SPLIT_KS(40)
#undef SPLIT_KS
#undef SPLIT_GO
    }
  }
  g.spin = spin;
  // one thread per output mode (up to 64 modes beyond 256 ride along as second modes), never fewer than 4 waves
  int threads = g.n_sec ? 256 : ((n_out + 63) / 64) * 64;
  if (threads < 256) threads = 256;
  // persistent workgroups, two per CU (what registers and LDS allow): measured 0.76 ms at 2 x CUs vs 1.00 ms at 3 x
  // (tail) and 1.16 ms at 1 x on cfg3
  static const long long max_blocks = [] {
    if (BMS_PROBE_ENV("SCRI_AMD_FUSED_BLOCKS")) return atoll(BMS_PROBE_ENV("SCRI_AMD_FUSED_BLOCKS"));
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    return 2LL * cus;
  }();
  const long long blocks = n_rows < max_blocks ? n_rows : max_blocks;
  const dim3 grid((unsigned)blocks), block(threads);
#define FUSED_GO(K) return launch_fused_t(K, stream, grid, block, lds, G, ldg, n_rows, g, m_index, T, D, out, ldo, col_of_pixel)
#define FUSED_EPT(KERNEL, NT, KS)                        \
  {                                                      \
    if (ept <= 2) FUSED_GO((KERNEL<NT, KS, 2>));         \
    if (ept == 3) FUSED_GO((KERNEL<NT, KS, 3>));         \
    FUSED_GO((KERNEL<NT, KS, 4>));                       \
  }
#define FUSED_KS(NT)                                              \
  if (L <= 16) {                                                  \
    if (g.ks == 3) FUSED_EPT(analysis_fused_kernel, NT, 3)        \
    if (g.ks == 5) FUSED_EPT(analysis_fused_kernel, NT, 5)        \
    FUSED_EPT(analysis_fused_kernel, NT, 6)                       \
  }                                                               \
  if (g.ks == 3) FUSED_EPT(analysis_fused_wide_kernel, NT, 3)     \
  if (g.ks == 5) FUSED_EPT(analysis_fused_wide_kernel, NT, 5)     \
  FUSED_EPT(analysis_fused_wide_kernel, NT, 6)
  // items per thread of the fold: as few as the grid needs (each one costs registers for the two rows in flight)
  const int ept = (n_theta * g.nk + threads - 1) / threads;
  if (ept > F_EPT_MAX) return hipErrorInvalidValue;
  if (n_theta <= 24) {
    FUSED_KS(24)
  }
  FUSED_KS(40)
#undef FUSED_EPT
#undef FUSED_KS
#undef FUSED_GO
}

// =====================================================================================================================
// Large grids (n_theta > 40: the ABD working grids, 99 x 99 at l_max = 24): the row of one time step no longer fits a
// workgroup's LDS next to its Fourier coefficients, so the two steps are two kernels with F[t][m][ring] in HBM between
// them -- but with the same arithmetic as the fused kernel: phi-folded cos/sin products on the MFMA pipe with the
// twiddles held in REGISTERS for the whole kernel, +-m recombined in registers, and the theta quadrature as MFMA
// products batched over time with the quadrature table of one m in registers.
// =====================================================================================================================

// ---- step 1: F[t][mi][ring] = sum_k G[t][ring][k] exp(-i m phi_k).  One wave per (time step, tile of 8 rings).
template <int KS, int NTC>
__global__ __launch_bounds__(64) void phi_dft_folded_kernel(const double* __restrict__ G, long long ldg, long long n_rows,
                                                            int n_theta, int n_phi, int L, int jp, double* __restrict__ F) {
  constexpr int PA = 4 * KS + 2;  // 2 x odd: conflict-free fragment reads
  __shared__ double Es[16 * PA], Os[16 * PA];
  extern __shared__ double2 Ft[];  // [2L+1][8] tile of F
  const int lane = threadIdx.x, fi = lane & 15, fk = lane >> 4;
  const int nk = n_phi / 2 + 1, mt = (n_theta + 7) / 8, nm = 2 * L + 1;
  // twiddles of this lane's B fragments: cos / sin (m phi_k), k = 4 s + fk, m = 16 n + fi + 1
  double bc[KS][NTC], bs[KS][NTC];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int n = 0; n < NTC; ++n) {
      const int k = 4 * s + fk, m = 16 * n + fi + 1;
      double sn = 0.0, co = 0.0;
      if (k < nk && m <= L) sincospi(2.0 * (double)(((long long)m * k) % n_phi) / (double)n_phi, &sn, &co);
      bc[s][n] = co, bs[s][n] = sn;
    }
  for (int e = lane; e < 16 * PA; e += 64) Es[e] = Os[e] = 0.0;
  const long long n_items = n_rows * mt;
  // This lane's pieces of an item (8 rings x nk folded pairs): element e = lane + 64 i is pair kk of ring r.  Offsets and
  // operand slots are the same for every item; the loads of the next item are issued before the products of this one.
  constexpr int NPRE = (8 * 4 * KS + 63) / 64;
  int ga[NPRE], gb[NPRE], so[NPRE];  // offsets (doubles) of x_k and x_{n-k} in the tile (-1: none), r << 16 | LDS slot
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int e = lane + 64 * i;
    const int r = e / nk, kk = e - r * nk;
    const int k2 = kk == 0 ? 0 : n_phi - kk;
    ga[i] = e < 8 * nk ? 2 * (r * n_phi + kk) : -1;
    gb[i] = (e < 8 * nk && k2 != kk) ? 2 * (r * n_phi + k2) : -1;
    so[i] = (r << 16) | (((r & 3) + 8 * (r >> 2)) * PA + kk);
  }
  double2 pa[NPRE], pb[NPRE];
#define DFT_LOAD(ITEM)                                                                                          \
  {                                                                                                             \
    const long long t_ = (ITEM) / mt;                                                                           \
    const int rt_ = (int)((ITEM)-t_ * mt);                                                                      \
    const int rings_ = n_theta - 8 * rt_ < 8 ? n_theta - 8 * rt_ : 8;                                           \
    const double* g_ = G + t_ * ldg + 2LL * (8 * rt_) * n_phi;                                                  \
    _Pragma("unroll") for (int i = 0; i < NPRE; ++i) {                                                          \
      const bool in_ = (so[i] >> 16) < rings_;                                                                  \
      pa[i] = (in_ && ga[i] >= 0) ? *reinterpret_cast<const double2*>(g_ + ga[i]) : double2{0.0, 0.0};          \
      pb[i] = (in_ && gb[i] >= 0) ? *reinterpret_cast<const double2*>(g_ + gb[i]) : double2{0.0, 0.0};          \
    }                                                                                                           \
  }
  if (blockIdx.x < n_items) DFT_LOAD((long long)blockIdx.x)
  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const long long t = item / mt;
    const int rt = (int)(item - t * mt);
    const int rings = n_theta - 8 * rt < 8 ? n_theta - 8 * rt : 8;
    // ---- fold 8 rings into the operands (row g + 8 h (+4 for Im) holds ring g + 4 h of the tile)
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
      if (ga[i] >= 0) {
        const int slot = so[i] & 0xffff;
        const bool in = (so[i] >> 16) < rings;
        const double wb = (in && gb[i] >= 0) ? 1.0 : 0.0;
        Es[slot] = pa[i].x + pb[i].x;
        Es[slot + 4 * PA] = pa[i].y + pb[i].y;
        Os[slot] = wb * (pa[i].x - pb[i].x);
        Os[slot + 4 * PA] = wb * (pa[i].y - pb[i].y);
      }
    if (item + gridDim.x < n_items) DFT_LOAD(item + gridDim.x)
    // (one wave: its LDS writes are visible to its own reads in program order)
    v4d_t ac[NTC], as[NTC];
#pragma unroll
    for (int n = 0; n < NTC; ++n) ac[n] = as[n] = v4d_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double ae = Es[fi * PA + 4 * s + fk], ao = Os[fi * PA + 4 * s + fk];
#pragma unroll
      for (int n = 0; n < NTC; ++n) {
        ac[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, bc[s][n], ac[n], 0, 0, 0);
        as[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao, bs[s][n], as[n], 0, 0, 0);
      }
    }
    // m = 0: plain sums of the folded rows
    if (fk == 0) {
      double c0 = 0.0, c1 = 0.0;
#pragma unroll
      for (int k = 0; k < 4 * KS; k += 2) {
        c0 += Es[fi * PA + k];
        c1 += Es[fi * PA + k + 1];
      }
      const int ring = (fi & 3) + 4 * (fi >> 3), part = (fi >> 2) & 1;
      reinterpret_cast<double*>(Ft + L * 8 + ring)[part] = c0 + c1;
    }
#pragma unroll
    for (int n = 0; n < NTC; ++n) {
      const int m = 16 * n + fi + 1;
      if (m <= L) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ring = fk + 4 * h;
          const double cre = ac[n][2 * h], cim = ac[n][2 * h + 1], sre = as[n][2 * h], sim = as[n][2 * h + 1];
          Ft[(L + m) * 8 + ring] = double2{cre + sim, cim - sre};
          Ft[(L - m) * 8 + ring] = double2{cre - sim, cim + sre};
        }
      }
    }
    // ---- the tile goes out as 128-byte pieces: F[t][mi][8 rt .. 8 rt + 7]
    double* f = F + ((t * nm) * (long long)jp + 8 * rt) * 2;
    for (int e = lane; e < nm * 8; e += 64) {
      const int mi = e >> 3, r = e & 7;
      if (r < rings) *reinterpret_cast<double2*>(f + ((long long)mi * jp + r) * 2) = Ft[e];
    }
  }
}

#undef DFT_LOAD

// ---- step 2: out[t][(l, m)] = sum_j T[(l,m)][j] F[t][m][j].  Workgroup = (m, block of time steps); the quadrature
// table of that m sits in registers as MFMA B fragments; a wave multiplies 8 time steps (16 real rows) per trip.
template <int KQ, int NTQ>
__global__ __launch_bounds__(256) void theta_quadrature_mfma_kernel(const double* __restrict__ F, long long n_rows, int n_theta,
                                                                    int L, int jp, int ell_min_out, int rows_per_block,
                                                                    const double* __restrict__ T, double* __restrict__ out,
                                                                    long long ldo) {
  constexpr int PQ = 4 * KQ + 2;  // 2 x odd
  __shared__ double As[4][16 * PQ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fi = lane & 15, fk = lane >> 4;
  const int mi = blockIdx.x, m = mi - L, nm = 2 * L + 1;
  const int am = m < 0 ? -m : m;
  const int l0 = am > ell_min_out ? am : ell_min_out;  // first l of this m
  // B fragments: T[(l0 + 16 n + fi, m)][4 s + fk]
  double bq[KQ][NTQ];
#pragma unroll
  for (int n = 0; n < NTQ; ++n) {
    const int l = l0 + 16 * n + fi;
    const long long o = (long long)l * (l + 1) - (long long)ell_min_out * ell_min_out + m;
#pragma unroll
    for (int s = 0; s < KQ; ++s) {
      const int j = 4 * s + fk;
      bq[s][n] = (l <= L && j < n_theta) ? T[o * n_theta + j] : 0.0;
    }
  }
  double* A = As[wave];
  for (int e = lane; e < 16 * PQ; e += 64) A[e] = 0.0;
  const long long t_begin = (long long)blockIdx.y * rows_per_block;
  long long t_end = t_begin + rows_per_block;
  if (t_end > n_rows) t_end = n_rows;
  // This lane's pieces of a trip (8 time steps of F[.][mi][0 .. jp)): element e = lane + 64 i is ring j of time step r.
  // Their addresses relative to the trip and their operand slots do not change from trip to trip; the loads of the next
  // trip are issued before the products of the current one.
  constexpr int NPRE = (8 * (4 * KQ + 4) + 63) / 64;
  int goff[NPRE], soff[NPRE];  // global offset (doubles) from the trip's first time step (-1: nothing to load), r << 16 | LDS slot
  const long long tstride = (long long)nm * jp * 2;
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int e = lane + 64 * i;
    const int r = e / jp, j = e - r * jp;
    const bool ok = e < 8 * jp && j < n_theta && j < 4 * KQ;
    goff[i] = ok ? 2 * j : -1;
    soff[i] = (r << 16) | (((r & 3) + 8 * (r >> 2)) * PQ + j);
  }
  double2 pre[NPRE];
  const double* Fm = F + (long long)mi * jp * 2;
#define TQ_LOAD(T0)                                                                                                   \
  _Pragma("unroll") for (int i = 0; i < NPRE; ++i) {                                                                  \
    const int r = soff[i] >> 16;                                                                                      \
    pre[i] = (goff[i] >= 0 && (T0) + r < t_end) ? *reinterpret_cast<const double2*>(Fm + ((T0) + r) * tstride + goff[i]) \
                                                 : double2{0.0, 0.0};                                                  \
  }
  if (t_begin + 8 * wave < t_end) TQ_LOAD(t_begin + 8 * wave)
  for (long long t0 = t_begin + 8 * wave; t0 < t_end; t0 += 32) {
    // ---- operand: row g + 8 h (+4 for Im) holds time step t0 + g + 4 h
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
      if (goff[i] >= 0) {
        const int slot = soff[i] & 0xffff;
        A[slot] = pre[i].x;
        A[slot + 4 * PQ] = pre[i].y;
      }
    if (t0 + 32 < t_end) TQ_LOAD(t0 + 32)
    v4d_t acc[NTQ];
#pragma unroll
    for (int n = 0; n < NTQ; ++n) acc[n] = v4d_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KQ; ++s) {
      const double a = A[fi * PQ + 4 * s + fk];
#pragma unroll
      for (int n = 0; n < NTQ; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[s][n], acc[n], 0, 0, 0);
    }
    // results r = 2h (re), 2h+1 (im) of time step t0 + fk + 4 h, column l = l0 + 16 n + fi
#pragma unroll
    for (int n = 0; n < NTQ; ++n) {
      const int l = l0 + 16 * n + fi;
      if (l <= L) {
        const long long o = (long long)l * (l + 1) - (long long)ell_min_out * ell_min_out + m;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const long long t = t0 + fk + 4 * h;
          if (t < t_end) *reinterpret_cast<double2*>(out + t * ldo + 2 * o) = double2{acc[n][2 * h], acc[n][2 * h + 1]};
        }
      }
    }
  }
}

int large_analysis_supported(int n_theta, int n_phi, int L) {
  const int ks = (n_phi / 2 + 1 + 3) / 4;
  return n_theta <= 104 && ks <= 16 && L >= 1 && L <= 32;
}
int large_analysis_jp(int n_theta) { return ((n_theta + 7) / 8) * 8; }

hipError_t launch_analysis_large(hipStream_t stream, const double* G, long long ldg, long long n_rows, int n_theta, int n_phi,
                                 int L, int ell_min_out, const double* T, double* F, double* out, long long ldo) {
  if (n_rows <= 0) return hipSuccess;
  const int jp = large_analysis_jp(n_theta), nm = 2 * L + 1;
  const int ks = (n_phi / 2 + 1 + 3) / 4;
  const int mt = (n_theta + 7) / 8;
  const size_t lds1 = sizeof(double2) * (size_t)nm * 8;
  const long long items = n_rows * mt;
  const unsigned grid1 = (unsigned)(items < 256 * 16 ? items : 256 * 16);
#define DFT_GO(KS, NTC)                                                                                              \
  hipLaunchKernelGGL((phi_dft_folded_kernel<KS, NTC>), dim3(grid1), dim3(64), lds1, stream, G, ldg, n_rows, n_theta, n_phi, L, jp, F)
#define DFT_KS(NTC)      \
  if (ks <= 7)           \
    DFT_GO(7, NTC);      \
  else if (ks <= 10)     \
    DFT_GO(10, NTC);     \
  else if (ks <= 13)     \
    DFT_GO(13, NTC);     \
  else                   \
    DFT_GO(16, NTC);
  if (L <= 16) {
    DFT_KS(1)
  } else {
    DFT_KS(2)
  }
#undef DFT_KS
#undef DFT_GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int kq = (n_theta + 3) / 4;
  static const int rows_per_block = BMS_PROBE_ENV("SCRI_AMD_TQ_ROWS") ? atoi(BMS_PROBE_ENV("SCRI_AMD_TQ_ROWS")) : 256;
  const dim3 grid2(nm, (unsigned)((n_rows + rows_per_block - 1) / rows_per_block));
  // columns of one m: l = max(|m|, ell_min_out) .. L, at most L + 1 - ell_min_out
  const int ntq = (L + 1 - ell_min_out + 15) / 16;
#define TQ_GO(KQ, NTQ)                                                                                                     \
  hipLaunchKernelGGL((theta_quadrature_mfma_kernel<KQ, NTQ>), grid2, dim3(256), 0, stream, F, n_rows, n_theta, L, jp, ell_min_out, \
                     rows_per_block, T, out, ldo)
#define TQ_KQ(NTQ)   \
  if (kq <= 11)      \
    TQ_GO(11, NTQ);  \
  else if (kq <= 18) \
    TQ_GO(18, NTQ);  \
  else               \
    TQ_GO(26, NTQ);
  if (ntq <= 1) {
    TQ_KQ(1)
  } else if (ntq == 2) {
    TQ_KQ(2)
  } else {
    TQ_KQ(3)
  }
#undef TQ_KQ
#undef TQ_GO
#undef TQ_LOAD
  return hipGetLastError();
}

}  // namespace bms
