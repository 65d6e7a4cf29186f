// Per-pixel set-up math shared by the host (bms_rotor_grid, bms_shard_plan: no GPU needed) and the device
// (pixel_tables_kernel): the rotor grid R_jk of scri/waveform_grid.py:130-174 (== boosted_grid,
// scri/asymptotic_bondi_data/transformations.py:100-148), the direction R z R^-1, and sum_k c_k sYlm_k(R).
#pragma once
#include "wigner.h"

namespace bms {

struct Quat {
  double w, x, y, z;
};
BMS_HD Quat qmul(const Quat& a, const Quat& b) {
#pragma clang fp contract(off)  // same roundings on host and device (acos near 1 amplifies an ulp to 1e-8)
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
BMS_HD Quat from_spherical_coords(double theta, double phi) {
#pragma clang fp contract(off)  // same roundings on host and device (acos near 1 amplifies an ulp to 1e-8)
  const double ct = cos(theta / 2), st = sin(theta / 2), cp = cos(phi / 2), sp = sin(phi / 2);
  return {cp * ct, -sp * st, cp * st, sp * ct};
}
// (theta, phi) of a rotor = (beta, alpha) of its Euler angles (numpy-quaternion as_spherical_coords).
// numpy-quaternion takes theta = 2 acos(sqrt((w^2 + z^2)/|q|^2)), which turns a 1-ulp rounding of w^2 + z^2 into
// 3e-8 rad at the pole pixels (acos near 1); theta = 2 atan2(|(x, y)|, |(w, z)|) is the same angle, well conditioned
// everywhere, so host and device agree to rounding and the value is the intended one.
BMS_HD void as_spherical_coords(const Quat& q, double& theta, double& phi) {
#pragma clang fp contract(off)
  phi = atan2(q.z, q.w) + atan2(-q.x, q.y);
  theta = 2 * atan2(sqrt(q.x * q.x + q.y * q.y), sqrt(q.w * q.w + q.z * q.z));
}
// q z q^-1
BMS_HD void rotate_z(const Quat& q, double r[3]) {
#pragma clang fp contract(off)  // same roundings on host and device (acos near 1 amplifies an ulp to 1e-8)
  const double n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  r[0] = 2 * (q.x * q.z + q.w * q.y) / n;
  r[1] = 2 * (q.y * q.z - q.w * q.x) / n;
  r[2] = (q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z) / n;
}

struct BoostSpec {
  double vhat[3];
  double rapidity;
  int boosted;  // beta > 3e-14 (waveform_grid.py:138)
};
BMS_HD BoostSpec make_boost_spec(const double v[3]) {
  BoostSpec b;
  const double beta = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  b.rapidity = atanh(beta);
  b.boosted = beta > 3e-14;
  for (int i = 0; i < 3; ++i) b.vhat[i] = b.boosted ? v[i] / beta : 0.0;
  return b;
}

// rotor of pixel (j, k): B'(dir(q)) q,  q = frame_rotation * from_spherical_coords(theta'_j, phi'_k)
BMS_HD Quat pixel_rotor(const Quat& frq, const BoostSpec& bs, int j, int k, int n_theta, int n_phi) {
#pragma clang fp contract(off)  // same roundings on host and device (acos near 1 amplifies an ulp to 1e-8)
  const double th = M_PI * j / (n_theta - 1);  // np.linspace(0, pi, n_theta)
  const double ph = (2 * M_PI) * k / n_phi;    // np.linspace(0, 2 pi, n_phi, endpoint=False)
  const Quat rq = qmul(frq, from_spherical_coords(th, ph));
  if (!bs.boosted) return rq;
  double tp, pp;
  as_spherical_coords(rq, tp, pp);
  const double rp[3] = {cos(pp) * sin(tp), sin(pp) * sin(tp), cos(tp)};
  double dot = bs.vhat[0] * rp[0] + bs.vhat[1] * rp[1] + bs.vhat[2] * rp[2];
  if (dot > 1.0) dot = 1.0;
  if (dot < -1.0) dot = -1.0;
  const double Thetaprm = acos(dot);
  const double Theta = 2 * atan(exp(-bs.rapidity) * tan(Thetaprm / 2.0));
  const double c[3] = {rp[1] * bs.vhat[2] - rp[2] * bs.vhat[1], rp[2] * bs.vhat[0] - rp[0] * bs.vhat[2],
                       rp[0] * bs.vhat[1] - rp[1] * bs.vhat[0]};
  const double cn = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  if (!(cn > 1e-200)) return rq;
  const double ang = (Thetaprm - Theta) / 2;
  const double s = sin(ang), co = cos(ang);
  const Quat B = {co, s * c[0] / cn, s * c[1] / cn, s * c[2] / cn};
  return qmul(B, rq);
}

// sum_k coef[k] sYlm_k(R) for modes l = 0..lmax
BMS_HD cplx eval_modes(const cplx* coef, int lmax, int spin, const Quat& q) {
  // terms are summed in double-double too: alpha enters the results as a time shift (u - alpha), unattenuated
  ddc sum = {{0.0, 0.0}, {0.0, 0.0}};
  for (int m = -lmax; m <= lmax; ++m) {
    const int am = m < 0 ? -m : m, as = spin < 0 ? -spin : spin;
    bool any = false;
    for (int ell = am > as ? am : as; ell <= lmax && !any; ++ell) {
      const cplx c = coef[LM_index(ell, m, 0)];
      any = c.re != 0.0 || c.im != 0.0;
    }
    if (!any) continue;
    SwshChain ch;
    ch.init(m, spin, q.w, q.x, q.y, q.z);
    for (int ell = ch.ell; ell <= lmax; ++ell) {
      const cplx c = coef[LM_index(ell, m, 0)];
      if (c.re != 0.0 || c.im != 0.0) {
        const cplx y = ch.value();
        sum.re = dd_add(sum.re, dd_sub(two_prod(c.re, y.re), two_prod(c.im, y.im)));
        sum.im = dd_add(sum.im, dd_add(two_prod(c.re, y.im), two_prod(c.im, y.re)));
      }
      if (ell < lmax) ch.next();
    }
  }
  return cplx{dd_to_double(sum.re), dd_to_double(sum.im)};
}

// What the engine needs per pixel; evaluated identically on host and device.
struct PixelSpec {
  Quat frq;
  double v[3];
  BoostSpec bs;
  double gamma, tt;
  int n_theta, n_phi, lst;
  int mode;              // 0: WM (h / sigma / none), 1: WM psi mixing, 2: ABD
  int spin;              // spin of the WM inhomogeneous term evaluation
  int conformal_weight;  // WM
  const cplx* st;        // supertranslation modes               (device pointer on the device)
  const cplx* c0;        // WM: type-term coefficients or NULL; psi: eth alpha / sqrt2; ABD: eth alpha / sqrt2
  const cplx* c1;        // ABD: eth eth alpha / 2
  cplx cv[4];            // l <= 1 modes of v.r (psi: times 1/sqrt2)
};
struct PixelOut {
  double *rotors, *k, *alpha, *skew_a, *skew_b;  // always
  double *col_off, *col_scale;                   // WM
  double *xa, *xb;                               // WM psi
  double *ethk, *etha, *ethetha, *ik, *ik3;      // ABD
};

// `p`: column (slot) the results are stored at; `p_grid`: the grid pixel they belong to (the two differ when the columns
// are stored sorted by time skew, see pixel_sort_kernel)
BMS_HD void pixel_tables_one(const PixelSpec& P, const PixelOut& O, int p, int p_grid) {
  const int j = p_grid / P.n_phi, kk = p_grid - j * P.n_phi;
  const Quat R = pixel_rotor(P.frq, P.bs, j, kk, P.n_theta, P.n_phi);
  O.rotors[4 * p] = R.w, O.rotors[4 * p + 1] = R.x, O.rotors[4 * p + 2] = R.y, O.rotors[4 * p + 3] = R.z;
  double r[3];
  rotate_z(R, r);
  const double vr = P.v[0] * r[0] + P.v[1] * r[1] + P.v[2] * r[2];
  const double kc = 1.0 / (P.gamma * (1 - vr));
  const double al = eval_modes(P.st, P.lst, 0, R).re;
  O.k[p] = kc;
  O.alpha[p] = al;
  O.skew_a[p] = -vr;  // 1/(gamma k) - 1
  O.skew_b[p] = al - P.tt;
  if (P.mode == 0 || P.mode == 1) {
    cplx off = {0.0, 0.0};
    if (P.mode == 0 && P.c0) off = eval_modes(P.c0, P.lst, P.spin, R);
    const double kw = pow(kc, (double)P.conformal_weight);
    O.col_off[2 * p] = off.re, O.col_off[2 * p + 1] = off.im;
    O.col_scale[2 * p] = kw, O.col_scale[2 * p + 1] = kw;
  }
  if (P.mode == 1) {
    const cplx A = eval_modes(P.c0, P.lst, 1, R);
    const cplx V = eval_modes(P.cv, 1, 1, R);
    O.xa[2 * p] = P.gamma * kc * V.re, O.xa[2 * p + 1] = P.gamma * kc * V.im;
    O.xb[2 * p] = A.re, O.xb[2 * p + 1] = A.im;
  }
  if (P.mode == 2) {
    const cplx ev = eval_modes(P.cv, 1, 1, R);
    O.ethk[2 * p] = ev.re / (1 - vr), O.ethk[2 * p + 1] = ev.im / (1 - vr);
    const cplx e1 = eval_modes(P.c0, P.lst, 1, R);
    const cplx e2 = eval_modes(P.c1, P.lst, 2, R);
    O.etha[2 * p] = e1.re, O.etha[2 * p + 1] = e1.im;
    O.ethetha[2 * p] = e2.re, O.ethetha[2 * p + 1] = e2.im;
    const double one_over_k = P.gamma * (1 - vr);
    O.ik[p] = one_over_k;
    O.ik3[p] = one_over_k * one_over_k * one_over_k;
    // sigma' = (sigma - eth eth alpha) / k as a per-column scale of the evaluating product (kernels_gemm_eval.hip)
    if (O.col_scale) O.col_scale[2 * p] = one_over_k, O.col_scale[2 * p + 1] = one_over_k;
  }
}

}  // namespace bms
