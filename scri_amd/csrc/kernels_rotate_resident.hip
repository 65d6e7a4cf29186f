// Time-series / constant Wigner-D rotation of mode weights (scri/rotations.py:346-392), wave-autonomous version for
// l ranges whose Delta tables fit the LDS (l <= 16 of the headline configurations: 76 KB).
//
// Same factorisation as kernels_rotate_mfma.hip (the per-step D matrix of the reference is never formed):
//   out_m = p3^m sum_mu Delta_{mu,m} [ p2^mu sum_m' Delta_{mu,m'} ( p1^m' f_m' ) ],   Delta^l = d^l(pi/2) (real constants)
//   p1 = i ea conj(eb),  p2 = exp(-i beta),  p3 = -i ea eb   (unit phases of the time step's rotor)
// and with Delta_{mu,m'} = (-1)^(mu-m') Delta_{m',mu} ONE table T[y][x] = Delta_{y,x} serves both products:
//   stage 1:  c_x = sum_y T[y][x] (-p1)^y f_y ,   h_x = (-p2)^x c_x          (x = mu, y = m')
//   stage 2:  o_x = p3^x sum_y T[y][x] h_y                                    (x = m,  y = mu)
// Both are MFMA products with the TABLE as the A operand (rows = output index x) and the DATA as the B operand (columns =
// 16 time steps), so the accumulators of a lane belong to ONE time step (lane & 15): every phase is a running product of
// that time step's rotor, in registers, with no cross-lane traffic; Re and Im parts are two independent accumulators
// sharing the table operand.
//
// What the previous kernel lost its time on (0.21 of the HBM roofline: 37 % of the wave cycles waiting, two workgroup
// barriers and a re-staging of both table images per l, quarter-row fetches of 16 bytes per lane) is gone by construction:
//   * all tables of the l range are loaded into the LDS once per workgroup (no swizzle needed for odd tile counts; an XOR of
//     the column tile with the row parity keeps ds_read_b64 conflict free for 32-column tables without padding them to 48);
//   * a wave owns its 16 time steps and a private 256 B x kpad LDS image [y][time] (complex, 16 B slots, XOR-swizzled so
//     that the transposing write from the load layout, the operand reads and the accumulator write-back are all conflict
//     free) -- there is NO workgroup barrier after the table load;
//   * rows are fetched with 4 adjacent lanes covering 64 contiguous bytes of a row (16 rows per instruction) and the fetch
//     of the next l is in flight under the two products of the current one;
//   * work units are (16-step tile, l group), dealt round-robin to the waves with the group rotating from round to round:
//     waves never wait for each other, and the rows and the rotor of a wave's next unit are requested under the products
//     of its current one.
// Every mode is still read and written exactly once from HBM (2 x 16 n_modes + 32 B per step).
// Rotors with |Rb| ~ 0 / |Ra| ~ 0 take exact diagonal / anti-diagonal branches (identity stays bit-exact).
#include <cstdlib>
#include "wigner.h"
#include "kernels.h"

namespace bms {

typedef double v4dq __attribute__((ext_vector_type(4)));

// table geometry of one l: rows kpad = 4 ceil(n / 4), column tiles ntl = ceil(n / 16), pitch 16 ntl
static inline void rr_shape(int ell, int* kpad, int* ntl) {
  const int n = 2 * ell + 1;
  *kpad = 4 * ((n + 3) / 4);
  *ntl = (n + 15) / 16;
}

// One product: acc[x][t] (+)= sum_y T[y][x] b_y(t) over the kpad rows of the image, operands of k step s + 1 requested
// before the MFMAs of step s are issued.  PHASE: multiply the operand by w (advanced by w4 per step) as it is read.
template <int NT, bool PHASE>
__device__ __forceinline__ void rr_product(const double2* __restrict__ bp, const double* __restrict__ ap, int cq, int pd, int swz,
                                           cplx w, cplx w4, v4dq (&acc_re)[NT], v4dq (&acc_im)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    acc_re[nt] = v4dq{0, 0, 0, 0};
    acc_im[nt] = v4dq{0, 0, 0, 0};
  }
  auto fetch = [&](int s, double2& b, double (&a)[NT]) {
    b = bp[s * 64];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a[nt] = ap[4 * s * pd + 16 * (nt ^ swz)];
  };
  auto step = [&](double2 b, const double (&a)[NT]) {
    if (PHASE) {
      const double br = b.x * w.re - b.y * w.im, bi = b.x * w.im + b.y * w.re;
      b.x = br;
      b.y = bi;
      w = cmul(w, w4);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc_re[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[nt], b.x, acc_re[nt], 0, 0, 0);
      acc_im[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[nt], b.y, acc_im[nt], 0, 0, 0);
    }
  };
  // two register sets: the operands of step s + 1 are on their way while the MFMAs of step s issue
  double2 b0, b1;
  double a0[NT], a1[NT];
  fetch(0, b0, a0);
  int s = 0;
  for (; s + 2 <= cq; s += 2) {
    fetch(s + 1, b1, a1);
    step(b0, a0);
    if (s + 2 < cq) fetch(s + 2, b0, a0);
    step(b1, a1);
  }
  if (s < cq) step(b0, a0);
}

struct RRLane {       // per-lane state of a work unit (MFMA orientation: time step l15, row block g)
  cplx q1, q2, p3;    // unit phases of the rotor: q1 = -i ea conj(eb), q2 = -exp(-i beta), p3 = -i ea eb
  cplx q1_4, q2_4, p3_4;
  cplx s1, s2, s3;    // q1^(g - l), q2^(g - l), p3^(g - l) of the current l (one factor conj(q) per l)
  bool live, z_only, flip, any_special;
  int g, slot_m;
};

BMS_HD cplx rr_pow4(cplx z) {
  z = cmul(z, z);
  return cmul(z, z);
}

// rotor (Ra, Rb) of a time step -> phases; `ell` = first l of the unit
__device__ __forceinline__ void rr_setup(RRLane& L, cplx Ra, cplx Rb, int ell) {
  double ra, rb;
  cplx ea, eb;
  spinor_polar(Ra, Rb, ra, rb, ea, eb);
  L.z_only = rb <= 1e-15;
  L.flip = ra <= 1e-15;
  L.q1 = cmul(cplx{0.0, -1.0}, cmul(ea, cconj(eb)));
  L.q2 = {-(ra * ra - rb * rb), 2.0 * ra * rb};
  L.p3 = cmul(cplx{0.0, -1.0}, cmul(ea, eb));
  L.q1_4 = rr_pow4(L.q1);
  L.q2_4 = rr_pow4(L.q2);
  L.p3_4 = rr_pow4(L.p3);
  L.s1 = cpow_unit(L.q1, L.g - ell);
  L.s2 = cpow_unit(L.q2, L.g - ell);
  L.s3 = cpow_unit(L.p3, L.g - ell);
  L.any_special = __any(L.live && (L.z_only || L.flip));
}

// Both products, the phase between them and the store of one l with NT column tiles
template <int NT, int MAXNT>
__device__ __forceinline__ void rr_one_ell(double2* __restrict__ S2, const double* __restrict__ Tl, int ell, double* __restrict__ dst,
                                           const double* __restrict__ rotor, const RRLane& L, double2 (&O)[4 * MAXNT]) {
  const int n = 2 * ell + 1, cq = (n + 3) / 4, pd = 16 * NT;
  const int swz = (NT & 1) ? 0 : (L.g & 1);  // even tile counts: column tile XOR row parity (pitch = 0 mod 32 doubles)
  const int l15 = L.slot_m ^ (L.g << 1);
  const double2* bp = S2 + L.g * 16 + L.slot_m;
  const double* ap = Tl + L.g * pd + l15;
  v4dq acc_re[NT], acc_im[NT];
  // stage 1: c_x = sum_y T[y][x] (-p1)^(y - l) f_y
  rr_product<NT, true>(bp, ap, cq, pd, swz, L.s1, L.q1_4, acc_re, acc_im);
  {
    // h_x = (-p2)^(x - l) c_x back into the image (rows n <= x < kpad are zero: the table's columns beyond n are)
    cplx v = L.s2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = 16 * nt + 4 * r + L.g;
        const double xr = acc_re[nt][r], xi = acc_im[nt][r];
        if (x < 4 * cq) S2[x * 16 + L.slot_m] = double2{xr * v.re - xi * v.im, xr * v.im + xi * v.re};
        v = cmul(v, L.q2_4);
      }
    }
  }
  // stage 2: o_x = p3^(x - l) sum_y T[y][x] h_y
  rr_product<NT, false>(bp, ap, cq, pd, swz, cplx{1.0, 0.0}, cplx{1.0, 0.0}, acc_re, acc_im);
  {
    cplx v = L.s3;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double xr = acc_re[nt][r], xi = acc_im[nt][r];
        acc_re[nt][r] = xr * v.re - xi * v.im;
        acc_im[nt][r] = xr * v.im + xi * v.re;
        v = cmul(v, L.p3_4);
      }
    }
  }
  if (L.any_special) {
    // exact branches re-read the input (still in HBM: nothing of this l has been stored yet) and the rotor
    if (L.live && (L.z_only || L.flip)) {
      const cplx Ra = {rotor[0], rotor[1]}, Rb = {rotor[2], rotor[3]};
      double ra, rb;
      cplx ea, eb;
      spinor_polar(Ra, Rb, ra, rb, ea, eb);
      const cplx e2 = L.z_only ? cmul(ea, ea) : cmul(eb, eb);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int x = 16 * nt + 4 * r + L.g;
          if (x < n) {
            const int m = x - ell;
            // z_only: D_mm = ea^(2m);  flip: D_{-m,m} = (-1)^(l-m) eb^(2m), out_m = f_{-m} D_{-m,m}
            const double2 f = *reinterpret_cast<const double2*>(dst + 2 * (L.z_only ? x : n - 1 - x));
            cplx wv = cpow_unit(e2, m);
            if (!L.z_only && ((ell - m) & 1)) wv = {-wv.re, -wv.im};
            const cplx val = cmul(cplx{f.x, f.y}, wv);
            acc_re[nt][r] = val.re;
            acc_im[nt][r] = val.im;
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0);  // every re-read of the wave has returned before any of its stores
    __builtin_amdgcn_wave_barrier();
  }
  // the rotated row leaves through O: element 4 nt + r = column x = 16 nt + 4 r + g of time step l15.  The caller stores it
  // one step later, after the rows of the next l have been consumed: a wait for those rows then never waits for stores.
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) O[4 * nt + r] = double2{acc_re[nt][r], acc_im[nt][r]};
  }
}

// Row fetch of one l in load orientation (time step lane >> 2, columns 4 u + (lane & 3)): 4 adjacent lanes cover 64
// contiguous bytes of a row, 16 rows per instruction; columns beyond n and rows beyond the series read as zero.
template <int MAXNT>
__device__ __forceinline__ void rr_fetch(double2 (&F)[4 * MAXNT], const double* __restrict__ data, long long ld, long long n_times,
                                         long long t0, int ell, int ell_min, int lrow, int lk) {
  const long long tl = t0 + lrow;
  const bool live = tl < n_times;
  const int n = 2 * ell + 1, cq = (n + 3) / 4;
  const double2* src = reinterpret_cast<const double2*>(data + (tl * ld + ((long long)ell * ell - (long long)ell_min * ell_min) + lk) * 2);
#pragma unroll
  for (int u = 0; u < 4 * MAXNT; ++u) {
    double2 v = double2{0.0, 0.0};
    if (live && u < cq && 4 * u + lk < n) v = src[4 * u];
    F[u] = v;
  }
}

template <int MAXNT>
__global__ __launch_bounds__(RR_THREADS, 1) void rotate_modes_resident_kernel(double* __restrict__ data, long long n_times,
                                                                              long long ld, const double* __restrict__ RaRb,
                                                                              long long rotor_stride,
                                                                              const double* __restrict__ tab_global, RotResPlan P,
                                                                              unsigned int* __restrict__ counter) {
  extern __shared__ double lds[];
  {
    const double2* src = reinterpret_cast<const double2*>(tab_global);
    double2* dst = reinterpret_cast<double2*>(lds);
    for (int e = threadIdx.x; e < P.tab_doubles / 2; e += RR_THREADS) dst[e] = src[e];
  }
  __syncthreads();  // the only workgroup barrier of the kernel

  // (readfirstlane: the wave index is uniform, so units, l and every loop bound below live in scalar registers)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int l15 = lane & 15, g = lane >> 4;  // MFMA orientation: time step l15, k / row block g
  const int lrow = lane >> 2, lk = lane & 3; // load orientation: time step lrow, column 4 u + lk
  double2* S2 = reinterpret_cast<double2*>(lds + P.tab_doubles + (size_t)wave * P.kpad_max * 32);
  const int slot_l = lrow ^ (lk << 1);       // swizzled time slot in load orientation (row y of an element has y & 3 = lk)
  const long long n_tiles = (n_times + 15) / 16;
  const unsigned int n_units = (unsigned int)(n_tiles * P.n_groups);

  // work units (16-step tile, l group) are dealt round-robin to the waves of the launch; the group a wave gets rotates
  // from round to round so that every wave sees every group (their costs differ)
  const unsigned int n_waves = gridDim.x * RR_WAVES;
  unsigned int round = 0;
  unsigned int unit = blockIdx.x * RR_WAVES + wave;
  auto unit_tile = [&](unsigned int u) { return (long long)(u / P.n_groups) * 16; };
  auto unit_group = [&](unsigned int u, unsigned int rnd) { return (int)((u % P.n_groups + rnd) % P.n_groups); };
  double2 F[4 * MAXNT];  // rows of the (unit, l) to come, requested one step ahead
  // rotor of time step t0 + l15 (the last step's for lanes beyond the series: their results are never stored)
  auto load_rotor = [&](long long t0, double2& a, double2& b) {
    long long tm = t0 + l15;
    tm = tm < n_times ? tm : n_times - 1;
    const double2* r = reinterpret_cast<const double2*>(RaRb + tm * rotor_stride);
    a = r[0];
    b = r[1];
  };

  if (unit >= n_units) return;
  long long t0 = unit_tile(unit);
  int grp = unit_group(unit, round);
  int ell = P.grp_lo[grp], ell_hi = P.grp_hi[grp];
  double2 Ra_n, Rb_n;
  load_rotor(t0, Ra_n, Rb_n);
  rr_fetch<MAXNT>(F, data, ld, n_times, t0, ell, P.ell_min, lrow, lk);
  RRLane L;
  L.g = g;
  L.slot_m = l15 ^ (g << 1);  // swizzled time slot of this lane's accesses in MFMA orientation
  L.live = t0 + l15 < n_times;
  rr_setup(L, cplx{Ra_n.x, Ra_n.y}, cplx{Rb_n.x, Rb_n.y}, ell);

  double2 O[4 * MAXNT];   // rotated row of the previous step, stored one step late
  double* dst_prev = nullptr;
  int n_prev = 0;
  bool live_prev = false;
  auto store_prev = [&]() {
    if (live_prev) {
#pragma unroll
      for (int j = 0; j < 4 * MAXNT; ++j) {
        const int x = 4 * j + g;
        if (x < n_prev) *reinterpret_cast<double2*>(dst_prev + 2 * x) = O[j];
      }
    }
  };

  for (;;) {
    const int n = 2 * ell + 1;
    const int cq = (n + 3) / 4;  // k steps = kpad / 4
    // ---- rows of this l -> LDS image [y][time]
#pragma unroll
    for (int u = 0; u < 4 * MAXNT; ++u)
      if (u < cq) S2[(4 * u + lk) * 16 + slot_l] = F[u];
    // ---- the previous step's row goes out now
    store_prev();
    // ---- what comes next: the next l of this unit, or the first l of the next unit (its rows and rotor are requested now
    // and arrive under the products below)
    const long long t0_cur = t0;
    const int ell_cur = ell;
    bool new_unit = false, done = false;
    if (ell < ell_hi) {
      ++ell;
    } else {
      unit += n_waves;
      ++round;
      if (unit >= n_units) {
        done = true;
      } else {
        new_unit = true;
        t0 = unit_tile(unit);
        grp = unit_group(unit, round);
        ell = P.grp_lo[grp];
        ell_hi = P.grp_hi[grp];
        load_rotor(t0, Ra_n, Rb_n);
      }
    }
    if (!done) rr_fetch<MAXNT>(F, data, ld, n_times, t0, ell, P.ell_min, lrow, lk);

    const double* Tl = lds + P.tab_off[ell_cur - P.ell_min];
    const long long tm = t0_cur + l15;
    double* dst = data + (tm * ld + ((long long)ell_cur * ell_cur - (long long)P.ell_min * P.ell_min)) * 2;
    const double* rot = RaRb + tm * rotor_stride;
    if (MAXNT >= 3 && n > 32)
      rr_one_ell<3, MAXNT>(S2, Tl, ell_cur, dst, rot, L, O);
    else if (MAXNT >= 2 && n > 16)
      rr_one_ell<2, MAXNT>(S2, Tl, ell_cur, dst, rot, L, O);
    else
      rr_one_ell<1, MAXNT>(S2, Tl, ell_cur, dst, rot, L, O);
    dst_prev = dst;
    n_prev = n;
    live_prev = L.live;
    if (done) break;
    if (new_unit) {
      L.live = t0 + l15 < n_times;
      rr_setup(L, cplx{Ra_n.x, Ra_n.y}, cplx{Rb_n.x, Rb_n.y}, ell);
    } else {
      L.s1 = cmul(L.s1, cconj(L.q1));
      L.s2 = cmul(L.s2, cconj(L.q2));
      L.s3 = cmul(L.s3, cconj(L.p3));
    }
  }
  store_prev();
}

// ---------------------------------------------------------------------------------------------------- host side

// Plan for an l range: table offsets, l groups of similar cost, LDS size.  Returns false if the range does not fit.
bool rotate_resident_plan(int ell_min, int ell_max, RotResPlan* P, size_t* lds_bytes) {
  if (ell_max - ell_min + 1 > RR_MAXL) return false;
  int kpad, ntl, tab = 0, kmax = 0, nt_max = 0;
  for (int l = ell_min; l <= ell_max; ++l) {
    rr_shape(l, &kpad, &ntl);
    P->tab_off[l - ell_min] = tab;
    tab += kpad * 16 * ntl;
    kmax = kpad > kmax ? kpad : kmax;
    nt_max = ntl > nt_max ? ntl : nt_max;
  }
  if (nt_max > 3) return false;
  P->ell_min = ell_min;
  P->ell_max = ell_max;
  P->tab_doubles = tab;
  P->kpad_max = kmax;
  const size_t bytes = sizeof(double) * ((size_t)tab + (size_t)RR_WAVES * kmax * 32);
  if (bytes > 160u * 1024u) return false;
  *lds_bytes = bytes;
  // l groups: contiguous, similar MFMA cost (ntl * kpad / 4 products per stage + a constant per l)
  auto cost = [](int l) {
    int k, t;
    rr_shape(l, &k, &t);
    return t * (k / 4) + 3;
  };
  int total = 0;
  for (int l = ell_min; l <= ell_max; ++l) total += cost(l);
  const int nl = ell_max - ell_min + 1;
  // two groups from 4 l on: measured on l = 2..16 and 2..8 (tools/bench_rotation.py, 1e5 and 1e6 steps), 1 / 2 / 3 / 4 groups
  // are within 5 % of each other; 2 is best at 1e5 steps, where the number of rounds per wave is small
  int G = nl >= 4 ? 2 : 1;
  if (const char* e = getenv("SCRI_AMD_ROTATE_GROUPS")) G = atoi(e) < 1 ? 1 : (atoi(e) > 4 ? 4 : atoi(e));
  P->n_groups = 0;
  int l = ell_min, used = 0;
  for (int gidx = 0; gidx < G && l <= ell_max; ++gidx) {
    const int target = (total * (gidx + 1)) / G;
    const int lo = l;
    do {
      used += cost(l);
      ++l;
    } while (l <= ell_max && gidx < G - 1 && used + cost(l) / 2 <= target);
    if (gidx == G - 1) l = ell_max + 1;
    P->grp_lo[P->n_groups] = lo;
    P->grp_hi[P->n_groups] = l - 1;
    ++P->n_groups;
  }
  return true;
}

// LDS image of the tables: per l, T[y][16 (nt ^ swz(y)) + i] = Delta[y][x = 16 nt + i], zero padded
void rotate_resident_pack(const RotResPlan& P, int ell, const double* Delta /* (2l+1)^2 row-major */, double* image) {
  int kpad, ntl;
  rr_shape(ell, &kpad, &ntl);
  const int n = 2 * ell + 1, pd = 16 * ntl;
  double* T = image + P.tab_off[ell - P.ell_min];
  for (int e = 0; e < kpad * pd; ++e) T[e] = 0.0;
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      const int nt = x / 16, i = x % 16;
      const int pt = (ntl & 1) ? nt : (nt ^ (y & 1));
      T[(size_t)y * pd + 16 * pt + i] = Delta[(size_t)y * n + x];
    }
}

hipError_t launch_rotate_modes_resident(hipStream_t stream, double* data, long long n_times, long long ld, const double* RaRb,
                                        long long rotor_stride, const double* tab_global, const RotResPlan& P, size_t lds_bytes,
                                        unsigned int* counter, int n_cu) {
  if (n_times <= 0) return hipSuccess;
  hipError_t e;
  (void)counter;
  const long long n_units = ((n_times + 15) / 16) * P.n_groups;
  long long blocks = (n_units + RR_WAVES - 1) / RR_WAVES;
  if (blocks > n_cu) blocks = n_cu;
  int nt_max = (2 * P.ell_max + 1 + 15) / 16;
#define RR_LAUNCH(NT)                                                                                                        \
  {                                                                                                                          \
    e = hipFuncSetAttribute((const void*)rotate_modes_resident_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                            (int)lds_bytes);                                                                                 \
    if (e != hipSuccess) return e;                                                                                           \
    hipLaunchKernelGGL(rotate_modes_resident_kernel<NT>, dim3((unsigned)blocks), dim3(RR_THREADS), lds_bytes, stream, data,  \
                       n_times, ld, RaRb, rotor_stride, tab_global, P, counter);                                             \
  }
  if (nt_max <= 1)
    RR_LAUNCH(1)
  else if (nt_max <= 2)
    RR_LAUNCH(2)
  else
    RR_LAUNCH(3)
#undef RR_LAUNCH
  return hipGetLastError();
}

}  // namespace bms
