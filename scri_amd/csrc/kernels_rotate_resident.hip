// Time-series / constant Wigner-D rotation of mode weights (scri/rotations.py:346-392), wave-autonomous version for
// l ranges whose Delta tables fit the LDS (l <= 16 of the headline configurations: 47 KB).
//
// Same factorisation as kernels_rotate_mfma.hip (the per-step D matrix of the reference is never formed):
//   out_m = p3^m sum_mu Delta_{mu,m} [ p2^mu sum_m' Delta_{mu,m'} ( p1^m' f_m' ) ],   Delta^l = d^l(pi/2) (real constants)
//   p1 = i ea conj(eb),  p2 = exp(-i beta),  p3 = -i ea eb   (unit phases of the time step's rotor)
// and with Delta_{mu,m'} = (-1)^(mu-m') Delta_{m',mu} ONE table T[y][x] = Delta_{y,x} serves both products:
//   stage 1:  c_x = sum_y T[y][x] (-p1)^y f_y ,   h_x = (-p2)^x c_x          (x = mu, y = m')
//   stage 2:  o_x = p3^x sum_y T[y][x] h_y                                    (x = m,  y = mu)
// Column symmetry of d(pi/2): T[y][-x] = (-1)^(l+y) T[y][x].  With the rows split by the parity of l + y (= the parity of the
// 0-based row index iy = y + l: class A even, class B odd),
//   P_x = sum_{y in A} T[y][x] b_y ,  Q_x = sum_{y in B} T[y][x] b_y ,   c_x = P_x + Q_x ,  c_{-x} = P_x - Q_x      (x >= 0)
// so only the l + 1 columns x >= 0 are ever multiplied: one 16-column tile up to l = 15 (the full table needed two from l = 8 and
// three at l = 16), the k steps of the two classes together are those of the full product.  The operand image in the LDS holds
// the rows of class A, then those of class B (each padded to a multiple of 4 with zero rows); the phases of the outputs
// x and -x are conjugates of each other and start at power g whatever l, so they are four constants per lane and work unit.
// Both are MFMA products with the TABLE as the A operand (rows = output index x) and the DATA as the B operand (columns =
// 16 time steps), so the accumulators of a lane belong to ONE time step (lane & 15): every phase is a product of
// that time step's rotor, in registers, with no cross-lane traffic; Re and Im parts are two independent accumulators
// sharing the table operand.
//
//   * all tables of the l range are loaded into the LDS once per workgroup;
//   * a wave owns its 16 time steps and a private 256 B x kpad LDS image [row][time] (complex, 16 B slots, XOR-swizzled so
//     that the transposing write from the load layout and the operand reads are conflict free) -- there is NO workgroup
//     barrier after the table load;
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
typedef int v4iq __attribute__((ext_vector_type(4)));

constexpr int RR_OOB = 0x7fff0000;  // byte offset beyond every descriptor: loads return zero, stores are dropped

// geometry of one l: rows of class A (l + 1 of them) and of class B (l) padded to multiples of 4; columns x' = 0..15 in the
// 16-column table, x' = 16..19 (l >= 16) in a 4-column side table
BMS_HD void rr_shape(int ell, int* ka, int* kb) {
  *ka = 4 * ((ell + 4) / 4);
  *kb = ell > 4 ? 4 * ((ell + 3) / 4) : 4;
}
// side columns of a kernel whose l range ends at l_max: 4 NX, NX = 0 (l_max <= 15), 1 (<= 19), 2 (<= 23), 3 (<= 27); every
// l >= 16 of the range carries a side table of that width (zero beyond its own l)
BMS_HD int rr_side_tiles(int ell_max) { return ell_max < 16 ? 0 : (ell_max - 16) / 4 + 1; }
BMS_HD int rr_table_doubles(int ell, int nx) {
  int ka, kb;
  rr_shape(ell, &ka, &kb);
  return (ka + kb) * (ell >= 16 ? 16 + 4 * nx : 16);
}

// Both partial products of one stage: P[x][t] = sum over the class A rows, Q[x][t] = sum over the class B rows of
// T[row][x] b_row(t); the rows of B follow those of A in the image and in the table (k steps 0..cqA-1, cqA..cqT-1), the
// operands of k step j + 1 are requested before the MFMAs of step j are issued, across the class boundary too.
// PHASE: multiply the operand by w (advanced by w8 per step: rows of a class are 2 apart in y) as it is read.
// NX > 0: columns x' = 16 + 4 i + (0..3), i < NX, through the 4-block 4x4x4 MFMA (same data operand; output lane (t, g) =
// column 16 + 4 i + g).
template <int NX>
struct RRAcc {
  v4dq P_re, P_im, Q_re, Q_im;
  double p_re[NX ? NX : 1], p_im[NX ? NX : 1], q_re[NX ? NX : 1], q_im[NX ? NX : 1];
};
template <bool PHASE, int NX, int MAXQ>
__device__ __forceinline__ void rr_products(const double2* __restrict__ bpA, const double2* __restrict__ bpB,
                                            const double* __restrict__ ap, const double* __restrict__ ax, int cqA, int cqT, cplx wA,
                                            cplx wB, cplx w8, RRAcc<NX>& C) {
  // k steps unrolled up to the largest count of the l range (every bound is wave-uniform: scalar branches); two operand
  // sets, the first step of a class starts its accumulators from the zero operand
  constexpr int NXA = NX ? NX : 1;
  double2 b[2];
  double a[2], x[2][NXA];
  auto fetch = [&](int j, int s) {
    b[s] = (j < cqA ? bpA : bpB)[j * 64];
    a[s] = ap[64 * j];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[s][i] = ax[16 * NX * j + 4 * i];
  };
  fetch(0, 0);
  cplx w = wA;
  const v4dq Z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NX; ++i) C.p_re[i] = C.p_im[i] = C.q_re[i] = C.q_im[i] = 0.0;
#pragma unroll
  for (int j = 0; j < MAXQ; ++j) {
    if (j < cqT) {
      if (j + 1 < MAXQ && j + 1 < cqT) fetch(j + 1, (j + 1) & 1);
      double2 bb = b[j & 1];
      const double aa = a[j & 1];
      if (PHASE) {
        if (j == cqA) w = wB;
        const double br = bb.x * w.re - bb.y * w.im, bi = bb.x * w.im + bb.y * w.re;
        bb.x = br;
        bb.y = bi;
        w = cmul(w, w8);
      }
      if (j == 0) {
        C.P_re = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb.x, Z, 0, 0, 0);
        C.P_im = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb.y, Z, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          C.p_re[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[j & 1][i], bb.x, 0.0, 0, 0, 0);
          C.p_im[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[j & 1][i], bb.y, 0.0, 0, 0, 0);
        }
      } else if (j < cqA) {
        C.P_re = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb.x, C.P_re, 0, 0, 0);
        C.P_im = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb.y, C.P_im, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          C.p_re[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[j & 1][i], bb.x, C.p_re[i], 0, 0, 0);
          C.p_im[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[j & 1][i], bb.y, C.p_im[i], 0, 0, 0);
        }
      } else if (j == cqA) {
        C.Q_re = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb.x, Z, 0, 0, 0);
        C.Q_im = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb.y, Z, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          C.q_re[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[j & 1][i], bb.x, 0.0, 0, 0, 0);
          C.q_im[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[j & 1][i], bb.y, 0.0, 0, 0, 0);
        }
      } else {
        C.Q_re = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb.x, C.Q_re, 0, 0, 0);
        C.Q_im = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bb.y, C.Q_im, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          C.q_re[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[j & 1][i], bb.x, C.q_re[i], 0, 0, 0);
          C.q_im[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x[j & 1][i], bb.y, C.q_im[i], 0, 0, 0);
        }
      }
    }
  }
}

struct RRLane {       // per-lane state of a work unit (MFMA orientation: time step l15, row block g)
  cplx q1, q1_8;      // q1 = -i ea conj(eb) and its 8th power (one k step of a class = 4 rows = 8 in y)
  cplx s1;            // q1^(2 g - l) of the current l (one factor conj(q1) per l)
  cplx v2, q2_4;      // q2^g and q2^4, q2 = -exp(-i beta): the outputs x' = 4 r + g start at power g whatever l
  cplx v3, p3_4;      // p3^g and p3^4, p3 = -i ea eb
  bool live, z_only, flip, any_special;
  int g, l15;
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
  const cplx q2 = {-(ra * ra - rb * rb), 2.0 * ra * rb};
  const cplx p3 = cmul(cplx{0.0, -1.0}, cmul(ea, eb));
  const cplx q1_4 = rr_pow4(L.q1);
  L.q1_8 = cmul(q1_4, q1_4);
  L.q2_4 = rr_pow4(q2);
  L.p3_4 = rr_pow4(p3);
  L.s1 = cpow_unit(L.q1, 2 * L.g - ell);
  L.v2 = cpow_unit(q2, L.g);
  L.v3 = cpow_unit(p3, L.g);
  L.any_special = __any(L.live && (L.z_only || L.flip));
}

// Both stages, the phase between them and the stores of one l.  A lane's outputs are x = x' and -x', x' = 4 r + g (slots
// r = 0..3) and 16 + 4 i + g (slots 4 + i, i < NX: l >= 16).  `rsrc` covers the 16 rows of the tile; `col0` = byte offset of this l's first
// mode in the lane's row.
template <int NX, int MAXQ, int ELLC = -1, int CQA = 0, int CQB = 0>
__device__ __forceinline__ void rr_one_ell(double2* __restrict__ S2, int dump, const double* __restrict__ Tl, int ell, int ka, int kb,
                                           __amdgpu_buffer_rsrc_t rsrc, int col0, const double* __restrict__ row,
                                           const double* __restrict__ rotor, const RRLane& L) {
  constexpr int NJ = 4 + NX;
  // (ELLC >= 0: l as a compile-time constant -- the k loops become straight-line code, the idle output slots disappear)
  // (CQA, CQB: only the k-step counts of the two classes -- the l ranges with too many l for one instance each)
  if (ELLC >= 0) {
    ell = ELLC;
    rr_shape(ELLC, &ka, &kb);
  } else if (CQA) {
    ka = 4 * CQA, kb = 4 * CQB;
  }
  const int cqA = ka >> 2, cqT = (ka + kb) >> 2;
  const double2* bpA = S2 + L.g * 16 + (L.l15 ^ (L.g << 1));
  const double2* bpB = S2 + L.g * 16 + (L.l15 ^ ((L.g ^ 2) << 1));
  const double* ap = Tl + L.g * 16 + L.l15;
  const double* ax = Tl + (ka + kb) * 16 + L.g * 4 * NX + (L.l15 & 3);
  RRAcc<NX> C;
#ifndef RR_PRIO
#define RR_PRIO 3
#endif
  // The wave raises its priority for its vector-instruction phases (epilogues, phases, image writes) and drops it for its matrix
  // products, so that the products of the SIMD's other wave hold up its short instructions a little less: 1 - 1.5 % (l <= 16, 1e5
  // steps, three alternating rounds: 0.2665 / 0.2639 / 0.2598 ms without, 0.2599 / 0.2601 / 0.2574 with priority 3).  The fp64
  // matrix and vector instructions of a SIMD share its ALUs; priorities only choose who goes first.
  if (RR_PRIO) __builtin_amdgcn_s_setprio(0);
  // stage 1: c_x = sum_y T[y][x] q1^y f_y
  rr_products<true, NX, MAXQ>(bpA, bpB, ap, ax, cqA, cqT, L.s1, cmul(L.s1, L.q1), L.q1_8, C);
  if (RR_PRIO) __builtin_amdgcn_s_setprio(RR_PRIO);
  {
    // h_{+-x'} = q2^(+-x') (P +- Q) back into the image: rows iy = l +- x' are of one class c (x' = g mod 2), at position
    // iy >> 1 of it; positions 4 apart share the swizzle key, so two addresses per sign serve all slots.  Outputs beyond l
    // go to the dump row.
    const int c = (ell + L.g) & 1, base = c ? ka : 0;
    const int hp = (ell + L.g) >> 1, hm = (ell - L.g) >> 1;
    auto at = [&](int half) { return (base + half) * 16 + (L.l15 ^ ((((half & 3) ^ (c << 1))) << 1)); };
    const int ap0 = at(hp), ap1 = at(hp + 2), am0 = at(hm), am1 = at(hm - 2);
    cplx v = L.v2;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (4 * j <= ell) {  // (wave-uniform: slots whose columns all lie beyond l are skipped)
        const int r = j & 3, xp = 4 * j + L.g;
        const int i = j < 4 ? 0 : j - 4;
        const double pr = j < 4 ? C.P_re[r] : C.p_re[i], pi = j < 4 ? C.P_im[r] : C.p_im[i];
        const double qr = j < 4 ? C.Q_re[r] : C.q_re[i], qi = j < 4 ? C.Q_im[r] : C.q_im[i];
        const double sr = pr + qr, si = pi + qi, dr = pr - qr, di = pi - qi;
        const int step = 64 * (j >> 1);
        const bool ok = xp <= ell;
        S2[ok ? ((j & 1) ? ap1 : ap0) + step : dump] = double2{sr * v.re - si * v.im, sr * v.im + si * v.re};
        S2[ok && xp > 0 ? ((j & 1) ? am1 : am0) - step : dump] = double2{dr * v.re + di * v.im, di * v.re - dr * v.im};
        if (j + 1 < NJ && 4 * (j + 1) <= ell) v = cmul(v, L.q2_4);
      }
    }
  }
  // stage 2: o_x = p3^x sum_y T[y][x] h_y
  if (RR_PRIO) __builtin_amdgcn_s_setprio(0);
  rr_products<false, NX, MAXQ>(bpA, bpB, ap, ax, cqA, cqT, cplx{1.0, 0.0}, cplx{1.0, 0.0}, cplx{1.0, 0.0}, C);
  if (RR_PRIO) __builtin_amdgcn_s_setprio(RR_PRIO);
  cplx e2 = {1.0, 0.0};
  const bool special = L.live && (L.z_only || L.flip);
  if (L.any_special && special) {
    // exact branches re-read the input (still in HBM: a lane stores the two modes of a slot after it has read both) and the rotor
    const cplx Ra = {rotor[0], rotor[1]}, Rb = {rotor[2], rotor[3]};
    double ra, rb;
    cplx ea, eb;
    spinor_polar(Ra, Rb, ra, rb, ea, eb);
    e2 = L.z_only ? cmul(ea, ea) : cmul(eb, eb);
  }
  cplx v = L.v3;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    // every slot issues its two stores (the number of stores of an iteration is what the wait for the next rows counts on);
    // slots whose columns all lie beyond l (wave-uniform) store nothing: offsets out of range, no arithmetic
    double2 op = double2{0.0, 0.0}, om = double2{0.0, 0.0};
    int offp = RR_OOB, offm = RR_OOB;
    if (4 * j <= ell) {
      const int r = j & 3, xp = 4 * j + L.g;
      const int i = j < 4 ? 0 : j - 4;
      const double pr = j < 4 ? C.P_re[r] : C.p_re[i], pi = j < 4 ? C.P_im[r] : C.p_im[i];
      const double qr = j < 4 ? C.Q_re[r] : C.q_re[i], qi = j < 4 ? C.Q_im[r] : C.q_im[i];
      const double sr = pr + qr, si = pi + qi, dr = pr - qr, di = pi - qi;
      op = double2{sr * v.re - si * v.im, sr * v.im + si * v.re};
      om = double2{dr * v.re + di * v.im, di * v.re - dr * v.im};
      if (j + 1 < NJ && 4 * (j + 1) <= ell) v = cmul(v, L.p3_4);
      const bool ok = xp <= ell;
      if (L.any_special) {
        if (special && ok) {
          // z_only: D_mm = ea^(2m);  flip: D_{-m,m} = (-1)^(l-m) eb^(2m), out_m = f_{-m} D_{-m,m}
          const double2 fa = *reinterpret_cast<const double2*>(row + 2 * (ell + xp));
          const double2 fb = *reinterpret_cast<const double2*>(row + 2 * (ell - xp));
          const double2 fp = L.z_only ? fa : fb, fm = L.z_only ? fb : fa;
          cplx wv = cpow_unit(e2, xp);
          if (!L.z_only && ((ell - xp) & 1)) wv = {-wv.re, -wv.im};
          const cplx vp = cmul(cplx{fp.x, fp.y}, wv), vm = cmul(cplx{fm.x, fm.y}, cconj(wv));
          op = double2{vp.re, vp.im};
          om = double2{vm.re, vm.im};
        }
      }
      offp = ok ? col0 + 16 * (ell + xp) : RR_OOB;
      offm = ok && xp > 0 ? col0 + 16 * (ell - xp) : RR_OOB;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4iq, op), rsrc, offp, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4iq, om), rsrc, offm, 0, 0);
  }
}

// NU = row slots of a lane in load orientation = KA(l_max) / 2: 4 (l <= 7), 6 (l <= 11), 8 (l <= 15), 10 (l <= 19), 12 (l <= 23),
// 14 (l <= 27)
template <int NU, int W>
__global__ __launch_bounds__(64 * W, 1) void rotate_modes_resident_kernel(double* __restrict__ data, long long n_times,
                                                                              long long ld, const double* __restrict__ RaRb,
                                                                              long long rotor_stride,
                                                                              const double* __restrict__ tab_global, RotResPlan P) {
  constexpr bool SPEC = true;
  constexpr int NXK = NU > 8 ? (NU - 8) / 2 : 0;  // side tiles of the kernel (rr_side_tiles of its largest l)
  const long long n_modes = (long long)(P.ell_max + 1) * (P.ell_max + 1) - (long long)P.ell_min * P.ell_min;
  extern __shared__ double lds[];
  {
    const double2* src = reinterpret_cast<const double2*>(tab_global);
    double2* dst = reinterpret_cast<double2*>(lds);
    for (int e = threadIdx.x; e < P.tab_doubles / 2; e += 64 * W) dst[e] = src[e];
  }
  __syncthreads();  // the only workgroup barrier of the kernel

  // (readfirstlane: the wave index is uniform, so units, l and every loop bound below live in scalar registers)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int l15 = lane & 15, g = lane >> 4;  // MFMA orientation: time step l15, k / row block g
  const int lrow = lane >> 2, lk = lane & 3; // load orientation: time step lrow, column 4 u + lk
  // per wave: image of kpad_max rows of 16 complex slots + one dump row for the writes of lanes without an output
  double2* S2 = reinterpret_cast<double2*>(lds + P.tab_doubles + (size_t)wave * (P.kpad_max + 1) * 32);
  const int dump = P.kpad_max * 16 + l15;
  // load orientation -> image: column iy = 4 u + lk is row 2 u + (lk >> 1) of class lk & 1; swizzle key = (that row & 3),
  // XOR 2 for class B, so the four lanes of a time step write four different keys
  const int half0 = lk >> 1, cls = lk & 1;
  const int slot_e = lrow ^ ((half0 ^ (cls << 1)) << 1);  // u even; u odd: key ^ 2 = slot ^ 4
  const long long n_tiles = (n_times + 15) / 16;
  const unsigned int n_units = (unsigned int)(n_tiles * P.n_groups);
  const int ld_b = (int)(ld * 16);
  const int off_load = lrow * ld_b + lk * 16, off_store = l15 * ld_b;  // byte offsets of a lane's row within its tile

  // work units (16-step tile, l group) are dealt round-robin to the waves of the launch; the group a wave gets rotates
  // from round to round so that every wave sees every group (their costs differ)
  const unsigned int n_waves = gridDim.x * W;
  auto unit_tile = [&](unsigned int u) { return (long long)(u / P.n_groups) * 16; };
  // (the rotation by the round keeps a wave from meeting the same group every time; when the number of waves is no multiple of
  // the number of groups u % n_groups changes from round to round by itself -- and a rotation would map two units of a tile that
  // straddles a round boundary onto the same group)
  const bool rotate_groups = n_waves % P.n_groups == 0;
  auto unit_group = [&](unsigned int u, unsigned int rnd) { return (int)((u % P.n_groups + (rotate_groups ? rnd : 0u)) % P.n_groups); };
  // The deal ends with a partial round: `tail_units` < n_waves units for n_waves waves -- at 1e5 steps, l <= 16 that is 212 units
  // of 12 500 whose round costs as much as each of the six full ones.  Those units are cut into `tail_split` pieces of
  // consecutive l (equal shares of the group's cost), one piece per wave, so that the last round takes 1 / tail_split of a full one.
  const unsigned int full_rounds = n_units / n_waves, tail_first = full_rounds * n_waves, tail_units = n_units - tail_first;
  unsigned int tail_split = tail_units ? n_waves / tail_units : 1u;
  if (tail_split > (unsigned int)RR_MAXL) tail_split = RR_MAXL;
  auto ell_cost = [](int ell) {
    int a, b;
    rr_shape(ell, &a, &b);
    return (a + b) / 4 + 3;  // k steps of a stage + the fixed part of an l (as the host's grouping)
  };
  // descriptor of the rows of a tile: rows beyond the series are out of range (loads give zero, stores are dropped)
  auto tile_rsrc = [&](long long t0) {
    const long long rows = n_times - t0 < 16 ? n_times - t0 : 16;
    return __builtin_amdgcn_make_buffer_rsrc(data + t0 * ld * 2, 0, (int)(((rows - 1) * ld + n_modes) * 16), 0x00020000);
  };
  double2 F[NU];  // rows of the (unit, l) to come, requested one step ahead; columns beyond 2 l + 1 read as zero
  auto fetch = [&](__amdgpu_buffer_rsrc_t rs, int ell) {
    const int n = 2 * ell + 1, col = (ell * ell - P.ell_min * P.ell_min) * 16 + off_load;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int off = RR_OOB;  // (slots wholly beyond the row, wave-uniform, skip the arithmetic but still issue their load)
      if (4 * u < n) off = 4 * u + lk < n ? col + 64 * u : RR_OOB;
      F[u] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };
  // rotor of time step t0 + l15 (the last step's for lanes beyond the series: their results are never stored)
  auto load_rotor = [&](long long t0, double2& a, double2& b) {
    long long tm = t0 + l15;
    tm = tm < n_times ? tm : n_times - 1;
    const double2* r = reinterpret_cast<const double2*>(RaRb + tm * rotor_stride);
    a = r[0];
    b = r[1];
  };

  // position in a wave's sequence of (unit, l) steps
  struct Step {
    unsigned int unit, round;
    long long t0;
    int ell, ell_hi;
    bool done;
  };
  auto enter_unit = [&](Step& s) {
    if (s.unit < tail_first) {
      s.done = false;
      s.t0 = unit_tile(s.unit);
      const int grp = unit_group(s.unit, s.round);
      s.ell = P.grp_lo[grp];
      s.ell_hi = P.grp_hi[grp];
      return;
    }
    // the partial last round: wave w takes piece w % tail_split of unit tail_first + w / tail_split
    const unsigned int w = s.unit - tail_first;
    s.done = w >= tail_units * tail_split;
    if (s.done) return;
    const unsigned int u = tail_first + w / tail_split, piece = w % tail_split;
    s.t0 = unit_tile(u);
    const int grp = unit_group(u, full_rounds);
    const int lo = P.grp_lo[grp], hi = P.grp_hi[grp];
    int total = 0;
    for (int l = lo; l <= hi; ++l) total += ell_cost(l);
    // piece k = the l whose running cost (midpoint) falls into [k, k + 1) total / tail_split: contiguous, possibly empty
    int first = hi + 1, last = lo - 1, run = 0;
    for (int l = lo; l <= hi; ++l) {
      const int cst = ell_cost(l);
      const unsigned int k = (unsigned int)(((2 * run + cst) * (int)tail_split) / (2 * total));
      run += cst;
      if (k == piece) {
        first = l < first ? l : first;
        last = l;
      }
    }
    s.done = first > last;
    s.ell = first;
    s.ell_hi = last;
  };
  auto advance = [&](Step& s) {
    if (s.ell < s.ell_hi) {
      ++s.ell;
    } else {
      s.unit += n_waves;
      ++s.round;
      enter_unit(s);
    }
  };
  Step cur;
  cur.unit = blockIdx.x * W + wave;
  cur.round = 0;
  enter_unit(cur);
  if (cur.done) return;
  double2 Ra_n, Rb_n;
  load_rotor(cur.t0, Ra_n, Rb_n);
  __amdgpu_buffer_rsrc_t rs_cur = tile_rsrc(cur.t0);
  fetch(rs_cur, cur.ell);
  Step n1 = cur;
  advance(n1);
  // The wait for the rows at the top of the loop must let the (younger) stores of the previous l stay in flight.  The
  // compiler merges the loop entry with the back edge and takes the smaller vmcnt; with as many (out-of-range, dropped)
  // stores behind the first fetch as an iteration issues behind the others, both agree.
#pragma unroll
  for (int j = 0; j < 8; ++j)
    __builtin_amdgcn_raw_buffer_store_b128(v4iq{j, 0, 0, 0}, rs_cur, RR_OOB + 256 * j + 16 * lane, 0, 0);
  RRLane L;
  L.g = g;
  L.l15 = l15;
  L.live = cur.t0 + l15 < n_times;
  rr_setup(L, cplx{Ra_n.x, Ra_n.y}, cplx{Rb_n.x, Rb_n.y}, cur.ell);

  for (;;) {
    int ka, kb;
    rr_shape(cur.ell, &ka, &kb);
    // ---- rows of this l -> LDS image [class A rows | class B rows][time]; the padding rows of both classes are written
    // too (F holds zeros there)
    {
      const int rb = half0 + (cls ? ka : 0);
#pragma unroll
      for (int u = 0; u < NU; ++u)
        if (2 * u < ka) S2[!cls || 2 * u + half0 < kb ? (rb + 2 * u) * 16 + (slot_e ^ ((u & 1) << 2)) : dump] = F[u];
    }
    // ---- what comes next: the next l of this unit, or the first l of the next unit -- its rows (and rotor) are requested now
    // and arrive under the products below
    const bool new_unit = !n1.done && n1.unit != cur.unit;
    __amdgpu_buffer_rsrc_t rs_next = rs_cur;
    if (new_unit) {
      load_rotor(n1.t0, Ra_n, Rb_n);
      rs_next = tile_rsrc(n1.t0);
    }
    if (!n1.done) fetch(rs_next, n1.ell);

    const double* Tl = lds + P.tab_off[cur.ell - P.ell_min];
    const long long tm = cur.t0 + l15;
    const int col0 = (cur.ell * cur.ell - P.ell_min * P.ell_min) * 16 + off_store;
    const double* row = data + (tm * ld + ((long long)cur.ell * cur.ell - (long long)P.ell_min * P.ell_min)) * 2;
    const double* rot = RaRb + tm * rotor_stride;
    // (with the two instances in one kernel the compiler's vmcnt bookkeeping at their join forgets that the stores are younger
    // than the rows in flight: an iteration of these kernels starts by draining its predecessor's stores.  Measured, that
    // costs less than running the side products of l >= 16 behind a run-time flag in a single instance: 2.45 vs 2.55 ms per
    // 1e6 steps at l <= 16)
    if (SPEC && NU >= 8 && NU <= 10) {
#define RR_CASE(E) \
  case E: rr_one_ell<0, NU, E>(S2, dump, Tl, cur.ell, ka, kb, rs_cur, col0, row, rot, L); break;
#define RR_PAIR(A, B, X) \
  case 4 * A + B: rr_one_ell<X, NU, -1, A, B>(S2, dump, Tl, cur.ell, ka, kb, rs_cur, col0, row, rot, L); break;
      // one instance per l up to 7 (the idle output slots and every loop bound disappear), one per pair of k-step counts above:
      // l = 8, 9..11, 12, 13..15, 16, 17..19
      switch (cur.ell) {
        RR_CASE(0) RR_CASE(1) RR_CASE(2) RR_CASE(3) RR_CASE(4) RR_CASE(5) RR_CASE(6) RR_CASE(7)
        default:
          switch (ka + (kb >> 2)) {
            RR_PAIR(3, 2, 0) RR_PAIR(3, 3, 0) RR_PAIR(4, 3, 0) RR_PAIR(4, 4, 0)
            default:
              if (NXK > 0) switch (ka + (kb >> 2)) {
                  RR_PAIR(5, 4, NXK)
                  default: rr_one_ell<NXK, NU, -1, 5, 5>(S2, dump, Tl, cur.ell, ka, kb, rs_cur, col0, row, rot, L); break;
                }
              break;
          }
          break;
      }
#undef RR_PAIR
#undef RR_CASE
    } else if (NXK > 0 && cur.ell >= 16)
      rr_one_ell<NXK, NU>(S2, dump, Tl, cur.ell, ka, kb, rs_cur, col0, row, rot, L);
    else if (SPEC && NU <= 6) {
#define RR_CASE(E) \
  case E: rr_one_ell<0, NU, E>(S2, dump, Tl, cur.ell, ka, kb, rs_cur, col0, row, rot, L); break;
      switch (cur.ell) {
        RR_CASE(0) RR_CASE(1) RR_CASE(2) RR_CASE(3) RR_CASE(4) RR_CASE(5) RR_CASE(6) RR_CASE(7)
        default:
          if (NU > 4) switch (cur.ell) {
              RR_CASE(8) RR_CASE(9) RR_CASE(10)
              default: rr_one_ell<0, NU, 11>(S2, dump, Tl, cur.ell, ka, kb, rs_cur, col0, row, rot, L); break;
            }
          break;
      }
#undef RR_CASE
    } else
      rr_one_ell<0, NU>(S2, dump, Tl, cur.ell, ka, kb, rs_cur, col0, row, rot, L);
    if (n1.done) break;
    if (new_unit) {
      L.live = n1.t0 + l15 < n_times;
      rr_setup(L, cplx{Ra_n.x, Ra_n.y}, cplx{Rb_n.x, Rb_n.y}, n1.ell);
    } else {
      L.s1 = cmul(L.s1, cconj(L.q1));
    }
    cur = n1;
    rs_cur = rs_next;
    advance(n1);
  }
}

// ---------------------------------------------------------------------------------------------------- host side

// Plan for an l range: table offsets, l groups of similar cost, LDS size.  Returns false if the range does not fit.
bool rotate_resident_plan(int ell_min, int ell_max, RotResPlan* P, size_t* lds_bytes) {
  if (ell_max - ell_min + 1 > RR_MAXL) return false;
  if (ell_max > 27) return false;  // output slots of a lane: x' = 4 r + g and 16 + 4 i + g, i < 3
  int ka, kb, tab = 0, kmax = 0;
  for (int l = ell_min; l <= ell_max; ++l) {
    rr_shape(l, &ka, &kb);
    P->tab_off[l - ell_min] = tab;
    tab += rr_table_doubles(l, rr_side_tiles(ell_max));
    kmax = ka + kb > kmax ? ka + kb : kmax;
  }
  P->ell_min = ell_min;
  P->ell_max = ell_max;
  P->tab_doubles = tab;
  P->kpad_max = kmax;
  // 8 waves (two per SIMD, 256 registers each); 12 (168 registers: spills in the per-l instances) on request for l_max <= 11
  P->waves = 8;
  if (const char* e = BMS_PROBE_ENV("SCRI_AMD_ROTATE_WAVES")) P->waves = atoi(e) == 12 && ell_max <= 11 ? 12 : 8;
  const size_t bytes = sizeof(double) * ((size_t)tab + (size_t)P->waves * (kmax + 1) * 32);  // + the dump row of a wave
  if (bytes > 160u * 1024u) return false;
  *lds_bytes = bytes;
  // l groups: contiguous, similar MFMA cost ((ka + kb) / 4 products per stage + a constant per l)
  auto cost = [](int l) {
    int a, b;
    rr_shape(l, &a, &b);
    return (a + b) / 4 + 3;
  };
  int total = 0;
  for (int l = ell_min; l <= ell_max; ++l) total += cost(l);
  const int nl = ell_max - ell_min + 1;
  // one group up to 9 l, two beyond: measured on l = 2..16 and 2..8 (tools/rotation_sweep.sh, 1e5 and 1e6 steps) with the partial
  // last round of the deal cut into per-l pieces (kernel): l <= 16: 0.265 / 0.259 / 0.264 / 0.265 ms per 1e5 steps for 1 / 2 / 3 / 4
  // groups, l <= 8: 0.0665 / 0.0709 / 0.0761 / 0.0804 (every unit pays one rotor set-up)
  int G = nl >= 10 ? 2 : 1;
  if (const char* e = BMS_PROBE_ENV("SCRI_AMD_ROTATE_GROUPS")) G = atoi(e) < 1 ? 1 : (atoi(e) > 4 ? 4 : atoi(e));
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

// LDS image of the tables: per l, rows R = class A (iy = 0, 2, ..) then class B (iy = 1, 3, ..), each padded with zero rows,
// columns x' = 0..l (m = x' >= 0):  T[R][x'] = Delta[iy(R)][l + x'] for x' < 16, then the side table [R][x' - 16] (l >= 16)
void rotate_resident_pack(const RotResPlan& P, int ell, const double* Delta /* (2l+1)^2 row-major */, double* image) {
  int ka, kb;
  rr_shape(ell, &ka, &kb);
  const int n = 2 * ell + 1;
  double* T = image + P.tab_off[ell - P.ell_min];
  double* X = T + (ka + kb) * 16;
  const int nx = rr_side_tiles(P.ell_max);
  for (int e = 0; e < rr_table_doubles(ell, nx); ++e) T[e] = 0.0;
  for (int R = 0; R < ka + kb; ++R) {
    const int iy = R < ka ? 2 * R : 2 * (R - ka) + 1;
    if (iy >= n) continue;
    for (int xp = 0; xp <= ell; ++xp) {
      const double d = Delta[(size_t)iy * n + ell + xp];
      if (xp < 16)
        T[R * 16 + xp] = d;
      else
        X[R * 4 * nx + xp - 16] = d;
    }
  }
}

hipError_t launch_rotate_modes_resident(hipStream_t stream, double* data, long long n_times, long long ld, const double* RaRb,
                                        long long rotor_stride, const double* tab_global, const RotResPlan& P, size_t lds_bytes,
                                        unsigned int* counter, int n_cu) {
  if (n_times <= 0) return hipSuccess;
  hipError_t e;
  (void)counter;
  const long long n_units = ((n_times + 15) / 16) * P.n_groups;
  long long blocks = (n_units + P.waves - 1) / P.waves;
  if (blocks > n_cu) blocks = n_cu;
  if (ld * 256 > 0x7ffe0000LL) return hipErrorInvalidValue;  // 32-bit byte offsets within a 16-row tile
  int ka, kb;
  rr_shape(P.ell_max, &ka, &kb);
  const int nu = ka / 2;
#define RR_LAUNCH(NU, W)                                                                                                       \
  {                                                                                                                          \
    e = allow_dynamic_lds((const void*)rotate_modes_resident_kernel<NU, W>);                                                                                 \
    if (e != hipSuccess) return e;                                                                                           \
    hipLaunchKernelGGL((rotate_modes_resident_kernel<NU, W>), dim3((unsigned)blocks), dim3(64 * W), lds_bytes, stream, data, \
                       n_times, ld, RaRb, rotor_stride, tab_global, P);                                                      \
  }
#define RR_LAUNCH_W(NU)  \
  if (P.waves == 12)     \
    RR_LAUNCH(NU, 12)    \
  else                   \
    RR_LAUNCH(NU, 8)
  if (nu <= 4) {
    RR_LAUNCH_W(4)
  } else if (nu <= 6) {
    RR_LAUNCH_W(6)
  } else if (nu <= 8) {
    RR_LAUNCH(8, 8)
  } else if (nu <= 10) {
    RR_LAUNCH(10, 8)
  } else if (nu <= 12) {
    RR_LAUNCH(12, 8)
  } else {
    RR_LAUNCH(14, 8)
  }
#undef RR_LAUNCH_W
#undef RR_LAUNCH
  return hipGetLastError();
}

}  // namespace bms
