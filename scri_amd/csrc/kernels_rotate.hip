// Time-series / constant Wigner-D rotation of mode weights (scri/rotations.py:346-392) for gfx950.
//
// Reference semantics: for each time t and each l, out[t,l,m] = sum_m' data[t,l,m'] D^l_{m',m}(R_t).
//
// MI355X design.  The per-step D matrix of the reference (6535 complex numbers at l<=16, built
// with an O(l) sum per element) is never formed.  With R = (ra ea, rb eb) and
//   D^l_{m',m} = ea^(m'+m) eb^(m-m') d^l_{m',m}(b),
//   d^l_{m',m}(b) = i^(m'-m) sum_mu Delta^l_{mu,m'} Delta^l_{mu,m} exp(-i mu b),  Delta^l = d^l(pi/2),
// the rotation factors into  phase -> real constant matrix -> phase -> real constant matrix -> phase:
//   g_m'  = f_m' (i ea conj(eb))^m'
//   h_mu  = exp(-i mu b) sum_m' Delta_{mu,m'} g_m'
//   out_m = (-i ea eb)^m sum_mu Delta_{mu,m} h_mu
// Delta^l are run-time constants (computed once per context in extended precision on the host),
// fetched through the scalar cache as wave-uniform operands.  Because Delta is real, the real and
// imaginary parts of a row transform independently: lane = (time step, re|im), 32 time steps per
// wavefront, every lane busy for every l.  Rows are staged through LDS (coalesced global access
// along the mode axis, conflict-free per-lane access with an odd pitch); the kernel reads and
// writes each mode exactly once (in place), i.e. the algorithmic 2*16*n_modes + 32 B per step.
#include "wigner.h"
#include "kernels.h"

namespace bms {

constexpr int ROT_MAX_WAVES = 4;        // wavefronts per workgroup (fewer when LDS-limited at large l)
constexpr int ROT_TPW = 32;             // time steps per wavefront (lane pair = re/im)

// exchange a double with the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: no LDS traffic
__device__ __forceinline__ double swap_pair(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// Delta table layout (built by the host, see engine): for each l, first the "direct" blocks then the
// "transposed" blocks.  A block holds ROT_MB rows: for column c (0..2l), ROT_MB consecutive doubles
// Delta[mu = b*MB + j][c] (direct) or Delta[c][m = b*MB + j] (transposed); rows beyond 2l are zero.
__global__ __launch_bounds__(ROT_MAX_WAVES * 64) void rotate_modes_kernel(
    double* __restrict__ data, long long n_times, long long ld /* complex elements per row */, int ell_min,
    int ell_max, const double* __restrict__ RaRb /* c16[n][2] or c16[1][2] */, long long rotor_stride /* 0 or 4 doubles */,
    const double* __restrict__ delta /* packed table from l = 0 */, const long long* __restrict__ delta_off /* per l */) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int part = lane & 1;  // 0 = real, 1 = imaginary
  const int tl = lane >> 1;   // local time step within the wave
  const int nmax = 2 * ell_max + 1;
  const int pitch = nmax | 1;                                    // odd pitch in doubles -> conflict-free
  double* bufA = lds + (size_t)wave * (2 * 64 * pitch);          // per-lane vectors: g / out
  double* bufB = bufA + 64 * pitch;                              // per-lane vectors: h
  double* myA = bufA + lane * pitch;
  double* myB = bufB + lane * pitch;

  const int nwaves = blockDim.x >> 6;
  const long long t0 = ((long long)blockIdx.x * nwaves + wave) * ROT_TPW;
  const long long t = t0 + tl;
  const bool live = t < n_times;

  // rotor of this lane's time step
  cplx Ra = {1.0, 0.0}, Rb = {0.0, 0.0};
  if (live) {
    const double* r = RaRb + t * rotor_stride;
    Ra = {r[0], r[1]};
    Rb = {r[2], r[3]};
  }
  double ra, rb;
  cplx ea, eb;
  spinor_polar(Ra, Rb, ra, rb, ea, eb);
  const bool z_only = rb <= 1e-15;  // pure rotation about z: D diagonal, D_mm = ea^(2m)
  const bool flip = ra <= 1e-15;    // rotation by pi about an axis in the x-y plane: D anti-diagonal
  // unit phases of the three stages
  const cplx p1 = cmul(cplx{0.0, 1.0}, cmul(ea, cconj(eb)));  // i ea conj(eb)
  const cplx p2 = {ra * ra - rb * rb, -2.0 * ra * rb};         // exp(-i b)
  const cplx p3 = cmul(cplx{0.0, -1.0}, cmul(ea, eb));         // -i ea eb
  const cplx ea2 = cmul(ea, ea), eb2 = cmul(eb, eb);

  for (int ell = ell_min; ell <= ell_max; ++ell) {
    const int n = 2 * ell + 1;
    const long long col0 = (long long)ell * ell - (long long)ell_min * ell_min;  // first mode of this l
    // ---- stage rows [32 t] x [n complex] into LDS: coalesced along the mode axis.
    // bufA of lane (2*tl + part) receives the re (part 0) or im (part 1) parts: element c at myA[c].
    for (int r = 0; r < ROT_TPW; ++r) {
      const long long tr = t0 + r;
      if (tr >= n_times) break;
      const double* src = data + (tr * ld + col0) * 2;
      for (int e = lane; e < 2 * n; e += 64) {
        bufA[(2 * r + (e & 1)) * pitch + (e >> 1)] = src[e];
      }
    }
    __builtin_amdgcn_wave_barrier();  // one wave: LDS operations complete in issue order

    // ---- phase 1: g_m' = f_m' p1^m'   (pair exchange for the complex product)
    {
      cplx w = cpow_unit(p1, -ell);
      for (int c = 0; c < n; ++c) {
        double mine = myA[c];
        double other = swap_pair(mine);
        // part 0 holds re: re' = re*w.re - im*w.im ; part 1 holds im: im' = re*w.im + im*w.re
        double v = part == 0 ? mine * w.re - other * w.im : other * w.im + mine * w.re;
        myA[c] = v;
        w = cmul(w, p1);
      }
    }
    // ---- stage 1: h_mu = p2^mu sum_m' Delta[mu][m'] g_m'
    const double* dl = delta + delta_off[ell];
    const int nblk = (n + ROT_MB - 1) / ROT_MB;
    {
      cplx w = cpow_unit(p2, -ell);
      for (int b = 0; b < nblk; ++b) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const double* db = dl + (long long)b * ROT_MB * n;
#pragma unroll 4
        for (int c = 0; c < n; ++c) {
          const double g = myA[c];
          a0 = fma(db[c * ROT_MB + 0], g, a0);
          a1 = fma(db[c * ROT_MB + 1], g, a1);
          a2 = fma(db[c * ROT_MB + 2], g, a2);
          a3 = fma(db[c * ROT_MB + 3], g, a3);
        }
        double acc[ROT_MB] = {a0, a1, a2, a3};
#pragma unroll
        for (int j = 0; j < ROT_MB; ++j) {
          const int mu = b * ROT_MB + j;
          if (mu < n) {
            double mine = acc[j];
            double other = swap_pair(mine);
            myB[mu] = part == 0 ? mine * w.re - other * w.im : other * w.im + mine * w.re;
            w = cmul(w, p2);
          }
        }
      }
    }
    // ---- stage 2: out_m = p3^m sum_mu Delta[mu][m] h_mu   (Delta^T: use the symmetric-table twin)
    {
      const double* dt = dl + (long long)nblk * ROT_MB * n;  // transposed blocks follow the direct ones
      cplx w = cpow_unit(p3, -ell);
      for (int b = 0; b < nblk; ++b) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const double* db = dt + (long long)b * ROT_MB * n;
#pragma unroll 4
        for (int c = 0; c < n; ++c) {
          const double h = myB[c];
          a0 = fma(db[c * ROT_MB + 0], h, a0);
          a1 = fma(db[c * ROT_MB + 1], h, a1);
          a2 = fma(db[c * ROT_MB + 2], h, a2);
          a3 = fma(db[c * ROT_MB + 3], h, a3);
        }
        double acc[ROT_MB] = {a0, a1, a2, a3};
#pragma unroll
        for (int j = 0; j < ROT_MB; ++j) {
          const int mi = b * ROT_MB + j;
          if (mi < n) {
            double mine = acc[j];
            double other = swap_pair(mine);
            myA[mi] = part == 0 ? mine * w.re - other * w.im : other * w.im + mine * w.re;
            w = cmul(w, p3);
          }
        }
      }
    }
    // ---- exact special cases (kept bit-exact like the reference's |Rb| ~ 0 / |Ra| ~ 0 branches):
    // the inputs were overwritten in LDS, so re-read them from global (rare path).
    if (live && (z_only || flip)) {
      const double* src = data + (t * ld + col0) * 2;
      for (int c = 0; c < n; ++c) {
        const int m = c - ell;
        cplx f, w;
        if (z_only) {
          f = {src[2 * c], src[2 * c + 1]};
          w = cpow_unit(ea2, m);  // D_mm = ea^(2m)
        } else {
          f = {src[2 * (n - 1 - c)], src[2 * (n - 1 - c) + 1]};  // f_{-m}
          w = cpow_unit(eb2, m);                                  // D_{-m,m} = (-1)^(l-m) eb^(2m)
          if ((ell - m) & 1) w = {-w.re, -w.im};
        }
        cplx o = cmul(f, w);
        myA[c] = part == 0 ? o.re : o.im;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- write back, coalesced along the mode axis
    for (int r = 0; r < ROT_TPW; ++r) {
      const long long tr = t0 + r;
      if (tr >= n_times) break;
      double* dst = data + (tr * ld + col0) * 2;
      for (int e = lane; e < 2 * n; e += 64) {
        dst[e] = bufA[(2 * r + (e & 1)) * pitch + (e >> 1)];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int rotate_waves_per_block(int ell_max) {
  const int pitch = (2 * ell_max + 1) | 1;
  const size_t per_wave = (size_t)2 * 64 * pitch * sizeof(double);
  int w = (int)((160u * 1024u) / per_wave);
  if (w > ROT_MAX_WAVES) w = ROT_MAX_WAVES;
  return w;  // 0 => unsupported ell_max
}

hipError_t launch_rotate_modes(hipStream_t stream, double* data, long long n_times, long long ld, int ell_min, int ell_max,
                               const double* RaRb, long long rotor_stride, const double* delta,
                               const long long* delta_off) {
  if (n_times <= 0) return hipSuccess;
  const int waves = rotate_waves_per_block(ell_max);
  if (waves < 1) return hipErrorInvalidValue;
  const int pitch = (2 * ell_max + 1) | 1;
  const size_t lds = (size_t)waves * 2 * 64 * pitch * sizeof(double);
  hipError_t e = allow_dynamic_lds((const void*)rotate_modes_kernel);
  if (e != hipSuccess) return e;
  const long long tpb = (long long)waves * ROT_TPW;
  const long long blocks = (n_times + tpb - 1) / tpb;
  hipLaunchKernelGGL(rotate_modes_kernel, dim3((unsigned)blocks), dim3(waves * 64), lds, stream, data, n_times, ld,
                     ell_min, ell_max, RaRb, rotor_stride, delta, delta_off);
  return hipGetLastError();
}

}  // namespace bms
