// Building blocks, series operators and storage-format bit transforms
// (engine.h: the split of the engine by entry family; include/scri_amd.h: the C ABI)
#include "engine.h"

// ====================================================================================================== building blocks

extern "C" int bms_ring_colatitudes(const double fr[4], const double v[3], int n_theta, int n_phi, double* thetas_out) try {
  if (!fr || !v || !thetas_out) return fail(nullptr, BMS_ERR_INVALID, "NULL argument");
  if (n_theta < 2 || n_phi < 1) return fail(nullptr, BMS_ERR_INVALID, "bad grid size");
  bms_transformation tr{};
  for (int i = 0; i < 4; ++i) tr.frame_rotation[i] = fr[i];
  for (int i = 0; i < 3; ++i) tr.boost_velocity[i] = v[i];
  tr.n_theta = n_theta, tr.n_phi = n_phi;
  std::vector<double> thetas;
  if (!separable_rotor_grid(&tr, thetas)) return 0;
  std::memcpy(thetas_out, thetas.data(), sizeof(double) * n_theta);
  return 1;
} BMS_CATCH(nullptr)

extern "C" int bms_rotor_grid(bms_ctx* c, const double fr[4], const double v[3], int n_theta, int n_phi, double* out) try {
  // ctx == NULL: pure host evaluation; otherwise the GPU kernel the transforms use (same pixel_math.h code)
  if (!fr || !v || !out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  if (n_theta < 2 || n_phi < 1) return fail(c, BMS_ERR_INVALID, "bad grid size");
  if (!c) {
    std::vector<Quat> R;
    build_rotor_grid(fr, v, n_theta, n_phi, R);
    std::memcpy(out, R.data(), sizeof(Quat) * R.size());
    return BMS_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const cplx zero4[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  bms_transformation tr{};
  tr.supertranslation = zero4;
  tr.ell_max_supertranslation = 1;
  for (int i = 0; i < 4; ++i) tr.frame_rotation[i] = fr[i];
  for (int i = 0; i < 3; ++i) tr.boost_velocity[i] = v[i];
  tr.n_theta = n_theta;
  tr.n_phi = n_phi;
  PixelTables T;
  DevPixel DP;
  int rc = device_pixel_tables(c, &tr, T, -1, 0, 0, nullptr, nullptr, nullptr, DP, 0);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(out, DP.rotors, sizeof(double) * 4 * T.n_pix, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BMS_OK;
} BMS_CATCH(c)

extern "C" int bms_conformal_factors(bms_ctx* c, const double v[3], const double* rotors, int64_t n, double* k, void* ethk_over_k,
                                     double* one_over_k, double* one_over_k_cubed) try {
  if (!v || !rotors || !k || !ethk_over_k || !one_over_k || !one_over_k_cubed) return fail(c, BMS_ERR_INVALID, "NULL argument");
  const double b2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (!(b2 < 1.0)) return fail(c, BMS_ERR_INVALID, "boost speed must be < 1");
  const double gamma = 1 / std::sqrt(1 - b2);
  // l <= 1 modes of v.r, evaluated with spin weight 1: eth(v.r) (the same coefficients the ABD transformation uses)
  const cplx cv[4] = {{0, 0},
                      {v[0] * std::sqrt(2 * M_PI / 3), v[1] * std::sqrt(2 * M_PI / 3)},
                      {v[2] * std::sqrt(4 * M_PI / 3), 0},
                      {-v[0] * std::sqrt(2 * M_PI / 3), v[1] * std::sqrt(2 * M_PI / 3)}};
  cplx* e = (cplx*)ethk_over_k;
  for (int64_t p = 0; p < n; ++p) {
    const Quat R = {rotors[4 * p], rotors[4 * p + 1], rotors[4 * p + 2], rotors[4 * p + 3]};
    double r[3];
    rotate_z(R, r);
    const double vr = v[0] * r[0] + v[1] * r[1] + v[2] * r[2];
    const cplx ev = eval_modes(cv, 1, 1, R);
    one_over_k[p] = gamma * (1 - vr);
    k[p] = 1.0 / one_over_k[p];
    e[p] = {ev.re / (1 - vr), ev.im / (1 - vr)};
    one_over_k_cubed[p] = one_over_k[p] * one_over_k[p] * one_over_k[p];
  }
  return BMS_OK;
} BMS_CATCH(c)

extern "C" int bms_swsh_grid(bms_ctx* c, const double* rotors, int64_t n, int spin, int ell_min, int ell_max, void* Y) try {
  if (!rotors || !Y) return BMS_ERR_INVALID;
  if (n < 0 || ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad sizes");
  if (ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", ell_max, MAX_ELL);
  if (!c) {  // host evaluation of the same header the kernel compiles (wigner.h: SwshChain), as bms_rotor_grid(ctx = NULL)
    const int nm = LM_total_size(ell_min, ell_max);
    cplx* out = (cplx*)Y;
    for (int64_t p = 0; p < n; ++p) {
      for (int k = 0; k < nm; ++k) out[p * nm + k] = {0.0, 0.0};
      for (int m = -ell_max; m <= ell_max; ++m) {
        SwshChain ch;
        ch.init(m, spin, rotors[4 * p], rotors[4 * p + 1], rotors[4 * p + 2], rotors[4 * p + 3]);
        for (int ell = ch.ell; ell <= ell_max; ++ell) {
          if (ell >= ell_min) out[p * nm + LM_index(ell, m, ell_min)] = ch.value();
          if (ell < ell_max) ch.next();
        }
      }
    }
    return BMS_OK;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  if (n <= 0) return BMS_OK;
  const size_t nm = LM_total_size(ell_min, ell_max);
  void* vp;
  int rc = upload(c, "rotors", rotors, 32 * (size_t)n, &vp);
  if (rc) return rc;
  double* dY;
  if ((rc = dev_buf_t(c, "swsh_vals", (size_t)n * nm * 2, &dY))) return rc;
  HIP_TRY(c, hipMemsetAsync(dY, 0, 16 * (size_t)n * nm, c->stream));
  HIP_TRY(c, launch_swsh_values(c->stream, (const double*)vp, (int)n, spin, ell_min, ell_max, dY));
  HIP_TRY(c, hipMemcpyAsync(Y, dY, 16 * (size_t)n * nm, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BMS_OK;
} BMS_CATCH(c)

extern "C" int bms_map2salm(bms_ctx* c, const void* grid, int mem, int64_t n_maps, int n_theta, int n_phi, int spin,
                            int ell_min, int ell_max, void* modes_out) try {
  if (!c || !grid || !modes_out) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_theta < 2 || n_phi < 1 || ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad sizes");
  if (ell_max > MAX_ELL || (long long)n_theta * n_phi > (1LL << 26)) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d on %d x %d: too large", ell_max, n_theta, n_phi);
  if (n_maps <= 0) return BMS_OK;
  const int n_pix = n_theta * n_phi, n_out = LM_total_size(ell_min, ell_max);
  int rc;
  AnalysisPlan ana;
  if ((rc = build_analysis(c, "m2s", n_theta, n_phi, spin, ell_min, ell_max, ana))) return rc;
  const double* d_in;
  if ((rc = stage_in(c, "in_data", grid, mem, (size_t)n_maps * n_pix * 16, &d_in))) return rc;
  double* d_out = (double*)modes_out;
  if (mem == BMS_HOST)
    if ((rc = dev_buf_t(c, "out_data", (size_t)n_maps * n_out * 2, &d_out))) return rc;
  if ((rc = run_analysis(c, ana, d_in, n_maps, d_out, 2LL * n_out))) return rc;
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(modes_out, d_out, (size_t)n_maps * n_out * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BMS_OK;
} BMS_CATCH(c)

extern "C" int bms_cubic_spline(bms_ctx* c, const double* x, int64_t n, const void* y, int64_t ld, int64_t n_cols, int mem,
                                const double* x_new, int64_t n_new, void* out) try {
  if (!c || !x || !y || !x_new || !out) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n < 4) return fail(c, BMS_ERR_UNSUPPORTED, "cubic spline needs at least 4 knots, got %lld", (long long)n);
  if (ld < n_cols || n_cols <= 0) return fail(c, BMS_ERR_INVALID, "bad column count / stride");
  for (int64_t i = 1; i < n; ++i)
    if (!(x[i] > x[i - 1])) return fail(c, BMS_ERR_INVALID, "knots must be strictly increasing");
  for (int64_t i = 1; i < n_new; ++i)
    if (!(x_new[i] >= x_new[i - 1])) return fail(c, BMS_ERR_INVALID, "evaluation points must be non-decreasing");
  if (n_new <= 0) return BMS_OK;
  int rc;
  double* d_x;
  void* d_xn;
  SplineTable* d_tab;
  if ((rc = upload_times(c, x, n, 0, n, 0, n, &d_x, &d_tab))) return rc;
  if ((rc = upload(c, "times_new", x_new, 8 * (size_t)n_new, &d_xn))) return rc;
  const double* d_y;
  if ((rc = stage_in(c, "in_data", y, mem, (size_t)n * ld * 16, &d_y))) return rc;
  double* d_R;
  if ((rc = dev_buf_t(c, "R", (size_t)n * ld * 2, &d_R))) return rc;
  double* d_out = (double*)out;
  if (mem == BMS_HOST)
    if ((rc = dev_buf_t(c, "out_data", (size_t)n_new * n_cols * 2, &d_out))) return rc;
  const int tile = spline_tile_for(x, n);
  TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_spline_forward(c->stream, d_y, d_R, 2 * ld, (int)n_cols, 0, n, n, (const double*)d_x, d_tab, tile, SPLINE_HALO));
  TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_spline_backward_eval(c->stream, d_y, d_R, 2 * ld, (int)n_cols, 0, n, n, (const double*)d_x, d_tab, tile,
                                         SPLINE_HALO, (const double*)d_xn, nullptr, nullptr, 0.0, 0, n_new, d_out, 2 * n_cols));
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(out, d_out, (size_t)n_new * n_cols * 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BMS_OK;
} BMS_CATCH(c)

// scipy CubicSpline(x, y).derivative(k) / .antiderivative(-k) evaluated at x_new (ModesTimeSeries.interpolate with
// derivative_order, .dot / .ddot / .int / .iint: scri/modes_time_series.py:72-126)
extern "C" int bms_spline_derivative(bms_ctx* c, const double* x, int64_t n, const void* y, int64_t ld, int64_t n_cols, int mem,
                                     const double* x_new, int64_t n_new, int order, void* out) try {
  if (!c || !x || !y || !x_new || !out) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n < 4) return fail(c, BMS_ERR_UNSUPPORTED, "cubic spline needs at least 4 knots, got %lld", (long long)n);
  if (ld < n_cols || n_cols <= 0) return fail(c, BMS_ERR_INVALID, "bad column count / stride");
  if (order < -16 || order > 3) return fail(c, BMS_ERR_INVALID, "derivative order %d outside [-16, 3]", order);
  for (int64_t i = 1; i < n; ++i)
    if (!(x[i] > x[i - 1])) return fail(c, BMS_ERR_INVALID, "knots must be strictly increasing");
  if (n_new <= 0) return BMS_OK;
  int rc;
  double* d_x;
  void* d_xn;
  SplineTable* d_tab;
  if ((rc = upload_times(c, x, n, 0, n, 0, n, &d_x, &d_tab))) return rc;
  if ((rc = upload(c, "times_new", x_new, 8 * (size_t)n_new, &d_xn))) return rc;
  const double* d_y;
  if ((rc = stage_in(c, "in_data", y, mem, (size_t)n * ld * 16, &d_y))) return rc;
  double *d_R, *d_S, *d_P1 = nullptr, *d_P2 = nullptr, *d_carry = nullptr;
  if ((rc = dev_buf_t(c, "R", (size_t)n * ld * 2, &d_R))) return rc;
  if ((rc = dev_buf_t(c, "S", (size_t)n * ld * 2, &d_S))) return rc;
  hipStream_t S = c->stream;
  const int tile = spline_tile_for(x, n);
  TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_spline_forward(S, d_y, d_R, 2 * ld, (int)n_cols, 0, n, n, d_x, d_tab, tile, SPLINE_HALO));
  TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_spline_slopes(S, d_R, d_S, 2 * ld, (int)n_cols, n, d_tab, tile, SPLINE_HALO));
  double* d_out = (double*)out;
  if (mem == BMS_HOST)
    if ((rc = dev_buf_t(c, "out_data", (size_t)n_new * n_cols * 2, &d_out))) return rc;
  if (order < -2) {  // any antiderivative order (scri/modes_time_series.py:88-89): one array of knot values per level
    const int k = -order;
    double* d_Pall;
    const long long level_stride = (long long)n * ld * 2;
    if ((rc = dev_buf_t(c, "P_levels", (size_t)k * level_stride, &d_Pall))) return rc;
    if ((rc = dev_buf_t(c, "P_carry", (size_t)spline_prefix_carry_size(n, (int)n_cols), &d_carry))) return rc;
    TIMED(c, BMS_TAG_POINTWISE, launch_spline_prefix_levels(S, d_y, d_S, 2 * ld, (int)n_cols, n, d_x, d_Pall, level_stride, d_carry, k));
    TIMED(c, BMS_TAG_POINTWISE, launch_spline_antiderivative_eval(S, d_y, d_S, d_Pall, level_stride, 2 * ld, (int)n_cols, n, d_x,
                                                                  (const double*)d_xn, n_new, k, d_out, 2 * n_cols));
    if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(out, d_out, (size_t)n_new * n_cols * 16, hipMemcpyDeviceToHost, S));
    HIP_TRY(c, hipStreamSynchronize(S));
    return BMS_OK;
  }
  if (order < 0) {
    if ((rc = dev_buf_t(c, "P1", (size_t)n * ld * 2, &d_P1))) return rc;
    if (order < -1)
      if ((rc = dev_buf_t(c, "P2", (size_t)n * ld * 2, &d_P2))) return rc;
    if ((rc = dev_buf_t(c, "P_carry", (size_t)spline_prefix_carry_size(n, (int)n_cols), &d_carry))) return rc;
    TIMED(c, BMS_TAG_POINTWISE, launch_spline_prefix(S, d_y, d_S, 2 * ld, (int)n_cols, n, d_x, d_P1, d_P2, d_carry, -order));
  }
  TIMED(c, BMS_TAG_POINTWISE, launch_spline_hermite_eval(S, d_y, d_S, d_P1, d_P2, 2 * ld, (int)n_cols, n, d_x, (const double*)d_xn,
                                                         n_new, order, d_out, 2 * n_cols));
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(out, d_out, (size_t)n_new * n_cols * 16, hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));
  return BMS_OK;
} BMS_CATCH(c)

// Angular velocity of a waveform from its modes (scri/mode_calculations.py:403-432 with LdtVector :46-57 and LLMatrix
// :298-313; data_dot = CubicSpline(t, data).derivative()(t), scri/waveform_base.py:690-691 = the spline's knot slopes).
extern "C" int bms_angular_velocity(bms_ctx* c, const double* t, int64_t n, const void* data, int64_t ld, int ell_min, int ell_max,
                                    int mem, double* ldt_out, double* ll_out, double* omega_out) try {
  if (!c || !t || !data) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n < 4) return fail(c, BMS_ERR_UNSUPPORTED, "the time derivative needs at least 4 time steps, got %lld", (long long)n);
  if (ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad ell range");
  if (ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", ell_max, MAX_ELL);
  const int n_modes = LM_total_size(ell_min, ell_max);
  if (ld < n_modes) return fail(c, BMS_ERR_INVALID, "row stride smaller than the number of modes");
  for (int64_t i = 1; i < n; ++i)
    if (!(t[i] > t[i - 1])) return fail(c, BMS_ERR_INVALID, "time array must be strictly increasing (index %lld)", (long long)i);
  int rc;
  double* d_x;
  SplineTable* d_tab;
  if ((rc = upload_times(c, t, n, 0, n, 0, n, &d_x, &d_tab))) return rc;
  const double* d_y;
  if ((rc = stage_in(c, "in_data", data, mem, (size_t)n * ld * 16, &d_y))) return rc;
  double *d_R, *d_S, *d_res;
  if ((rc = dev_buf_t(c, "R", (size_t)n * ld * 2, &d_R))) return rc;
  if ((rc = dev_buf_t(c, "S", (size_t)n * ld * 2, &d_S))) return rc;
  if ((rc = dev_buf_t(c, "av_out", (size_t)n * 15, &d_res))) return rc;
  hipStream_t S = c->stream;
  const int tile = spline_tile_for(t, n);
  TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_spline_forward(S, d_y, d_R, 2 * ld, n_modes, 0, n, n, d_x, d_tab, tile, SPLINE_HALO));
  TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_spline_slopes(S, d_R, d_S, 2 * ld, n_modes, n, d_tab, tile, SPLINE_HALO));
  double *d_ldt = d_res, *d_ll = d_res + 3 * n, *d_om = d_res + 12 * n;
  TIMED(c, BMS_TAG_POINTWISE, launch_angular_velocity(S, d_y, d_S, 2 * ld, n, ell_min, n_modes, d_ldt, d_ll, d_om));
  if (ldt_out) HIP_TRY(c, hipMemcpyAsync(ldt_out, d_ldt, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, S));
  if (ll_out) HIP_TRY(c, hipMemcpyAsync(ll_out, d_ll, sizeof(double) * 9 * n, hipMemcpyDeviceToHost, S));
  if (omega_out) HIP_TRY(c, hipMemcpyAsync(omega_out, d_om, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));
  return BMS_OK;
} BMS_CATCH(c)

// Frame from angular velocity: dR/dt = (1/2) Omega R with Omega(t) the not-a-knot cubic spline through omega[n][3]
// (quaternion.integrate_angular_velocity as called by corotating_frame, scri/mode_calculations.py:470-471).  The state is
// four numbers marching in time: host code.  Each sampling interval is cut into sub-steps of bounded rotation angle; a
// sub-step is one fourth-order Magnus step (two Gauss points, one commutator), applied as an exact rotor exponential,
// so |R| = 1 is preserved to rounding and a constant angular velocity is integrated exactly.
namespace {
void host_spline_slopes(const double* x, int64_t n, const double* y, int64_t stride, std::vector<double>& s) {
  // scipy CubicSpline(bc_type='not-a-knot'): tridiagonal system for the knot slopes (same rows as kernels_spline.hip)
  std::vector<double> a(n), b(n), c(n), r(n);
  auto D = [&](int64_t j) { return y[(j + 1) * stride] - y[j * stride]; };
  {
    const double h0 = x[1] - x[0], h1 = x[2] - x[1], d = x[2] - x[0];
    a[0] = 0, b[0] = h1, c[0] = d;
    r[0] = ((h0 + 2 * d) * h1 / (d * h0)) * D(0) + (h0 * h0 / (d * h1)) * D(1);
  }
  for (int64_t j = 1; j < n - 1; ++j) {
    const double hm = x[j] - x[j - 1], hp = x[j + 1] - x[j];
    a[j] = hp, b[j] = 2 * (hm + hp), c[j] = hm;
    r[j] = 3 * (hp / hm) * D(j - 1) + 3 * (hm / hp) * D(j);
  }
  {
    const double hm = x[n - 2] - x[n - 3], hl = x[n - 1] - x[n - 2], d = x[n - 1] - x[n - 3];
    a[n - 1] = d, b[n - 1] = hm, c[n - 1] = 0;
    r[n - 1] = (hl * hl / (d * hm)) * D(n - 3) + ((2 * d + hl) * hm / (d * hl)) * D(n - 2);
  }
  for (int64_t j = 1; j < n; ++j) {
    const double m = a[j] / b[j - 1];
    b[j] -= m * c[j - 1];
    r[j] -= m * r[j - 1];
  }
  s.resize(n);
  s[n - 1] = r[n - 1] / b[n - 1];
  for (int64_t j = n - 2; j >= 0; --j) s[j] = (r[j] - c[j] * s[j + 1]) / b[j];
}
}  // namespace

extern "C" int bms_integrate_angular_velocity(bms_ctx* c, const double* t, int64_t n, const double* omega, const double R0[4],
                                              double tolerance, double* R_out) try {
  // pure host routine: ctx may be NULL
  if (!t || !omega || !R0 || !R_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  if (n < 4) return fail(c, BMS_ERR_UNSUPPORTED, "the spline of the angular velocity needs at least 4 time steps, got %lld", (long long)n);
  for (int64_t i = 1; i < n; ++i)
    if (!(t[i] > t[i - 1])) return fail(c, BMS_ERR_INVALID, "time array must be strictly increasing (index %lld)", (long long)i);
  if (!(tolerance > 0)) tolerance = 1e-12;
  std::vector<double> sl[3];
  for (int k = 0; k < 3; ++k) host_spline_slopes(t, n, omega + k, 3, sl[k]);
  // rotation angle per sub-step: the Magnus-4 defect scales like angle^5 times the relative change of Omega
  double amax = 2.0 * std::pow(tolerance, 0.2);
  amax = std::min(0.2, std::max(1e-3, amax));
  Quat R = {R0[0], R0[1], R0[2], R0[3]};
  R_out[0] = R.w, R_out[1] = R.x, R_out[2] = R.y, R_out[3] = R.z;
  const double g1 = 0.5 - std::sqrt(3.0) / 6.0, g2 = 0.5 + std::sqrt(3.0) / 6.0;
  for (int64_t j = 0; j + 1 < n; ++j) {
    const double h = t[j + 1] - t[j];
    double y0[3], y1[3], s0[3], s1[3], c2[3], c3[3];
    double wmax = 0;
    for (int k = 0; k < 3; ++k) {
      y0[k] = omega[3 * j + k], y1[k] = omega[3 * (j + 1) + k], s0[k] = sl[k][j], s1[k] = sl[k][j + 1];
      const double dd = (y1[k] - y0[k]) / h, tt = (s0[k] + s1[k] - 2 * dd) / h;
      c3[k] = tt / h, c2[k] = (dd - s0[k]) / h - tt;
    }
    wmax = std::max(std::sqrt(y0[0] * y0[0] + y0[1] * y0[1] + y0[2] * y0[2]), std::sqrt(y1[0] * y1[0] + y1[1] * y1[1] + y1[2] * y1[2]));
    const int64_t m = std::max<int64_t>(1, (int64_t)std::ceil(wmax * h / amax));
    const double hs = h / m;
    auto om = [&](double tau, double* w) {
      for (int k = 0; k < 3; ++k) w[k] = y0[k] + tau * (s0[k] + tau * (c2[k] + tau * c3[k]));
    };
    for (int64_t q = 0; q < m; ++q) {
      double wa[3], wb[3];
      om((q + g1) * hs, wa);
      om((q + g2) * hs, wb);
      // Magnus: Theta = h/2 (A1 + A2) + (sqrt3/12) h^2 [A2, A1], A = Omega/2 as a pure quaternion, [A2, A1] = 2 (a2 x a1)
      // => rotation vector (for exp(Theta), Theta = theta/2 as a vector): theta/2 = h/4 (wa + wb) + (sqrt3/24) h^2 (wb x wa)
      const double cx = wb[1] * wa[2] - wb[2] * wa[1], cy = wb[2] * wa[0] - wb[0] * wa[2], cz = wb[0] * wa[1] - wb[1] * wa[0];
      const double k1 = hs / 4, k2 = std::sqrt(3.0) / 24 * hs * hs;
      const double vx = k1 * (wa[0] + wb[0]) + k2 * cx, vy = k1 * (wa[1] + wb[1]) + k2 * cy, vz = k1 * (wa[2] + wb[2]) + k2 * cz;
      const double vn = std::sqrt(vx * vx + vy * vy + vz * vz);
      const double sc = vn > 1e-300 ? std::sin(vn) / vn : 1.0;
      const Quat E = {std::cos(vn), sc * vx, sc * vy, sc * vz};
      R = qmul(E, R);
    }
    const double nr = std::sqrt(R.w * R.w + R.x * R.x + R.y * R.y + R.z * R.z);
    R = {R.w / nr, R.x / nr, R.y / nr, R.z / nr};
    double* o = R_out + 4 * (j + 1);
    o[0] = R.w, o[1] = R.x, o[2] = R.y, o[3] = R.z;
  }
  return BMS_OK;
} BMS_CATCH(c)

// spinsfast.salm2map(modes, s, ell_max, n_theta, n_phi): values on the equiangular grid (sf.Modes.grid, used by the
// super-rest-frame iteration, scri/asymptotic_bondi_data/map_to_superrest_frame.py:171,216); modes from l = 0
extern "C" int bms_salm2map(bms_ctx* c, const void* modes, int mem, int64_t n_maps, int spin, int ell_max, int n_theta, int n_phi,
                            void* grid_out) try {
  if (!c || !modes || !grid_out) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (ell_max < 0 || n_theta < 2 || n_phi < 1 || std::abs(spin) > 4) return fail(c, BMS_ERR_INVALID, "bad sizes");
  if (ell_max > MAX_ELL || (long long)n_theta * n_phi > (1LL << 26)) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d on %d x %d: too large", ell_max, n_theta, n_phi);
  if (n_maps <= 0) return BMS_OK;
  const int n_pix = n_theta * n_phi, nm = (ell_max + 1) * (ell_max + 1);
  hipStream_t S = c->stream;
  int rc;
  const long long P2 = 2LL * n_pix, ldb = round_up(P2, 128);
  // the equiangular grid itself: separable (kernels_synthesis_large.hip) wherever that kernel takes the shape
  SynthesisPlan syn;
  if (n_theta >= 3 && ell_max >= 1 && !c->opt.on(OPT_NO_SEPARABLE_SYNTHESIS))
    if ((rc = build_synthesis(c, n_theta, n_phi, spin, 0, ell_max, syn))) return rc;
  double* d_B = nullptr;
  if (!syn.large) {
    std::vector<double> rot(4 * (size_t)n_pix);
    for (int j = 0; j < n_theta; ++j)
      for (int k = 0; k < n_phi; ++k) {
        const Quat q = from_spherical_coords(M_PI * j / (n_theta - 1), (2 * M_PI) * k / n_phi);
        double* r = &rot[4 * ((size_t)j * n_phi + k)];
        r[0] = q.w, r[1] = q.x, r[2] = q.y, r[3] = q.z;
      }
    void* vp;
    if ((rc = upload(c, "gm_rotors", rot.data(), 8 * rot.size(), &vp))) return rc;
    if ((rc = dev_buf_t(c, "gm_Ba", (size_t)round_up(nm, 8) * ldb, &d_B))) return rc;
    HIP_TRY(c, hipMemsetAsync(d_B, 0, sizeof(double) * round_up(nm, 8) * ldb, S));
    TIMED(c, BMS_TAG_SETUP, launch_swsh_matrix_complex(S, (const double*)vp, n_pix, spin, 0, ell_max, d_B, ldb));
  }
  const double* d_a;
  if ((rc = stage_in(c, "in_data", modes, mem, (size_t)n_maps * nm * 16, &d_a))) return rc;
  double* d_G = (double*)grid_out;
  if (mem == BMS_HOST)
    if ((rc = dev_buf_t(c, "out_data", (size_t)n_maps * P2, &d_G))) return rc;
  if (syn.large) {
    SynthesisPlan two = syn;
    two.nt = 0;  // (the one-kernel form wants padded rows)
    if ((rc = run_synthesis(c, two, d_a, 2LL * nm, n_maps, nullptr, d_G, P2))) return rc;
  } else
  TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m(S, d_a, 2LL * nm, d_B, ldb, d_G, P2, n_maps, n_pix, nm, nullptr, nullptr));
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(grid_out, d_G, (size_t)n_maps * n_pix * 16, hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));
  return BMS_OK;
} BMS_CATCH(c)

// sf.Modes.evaluate(R): the field at arbitrary directions / frames, out[t][p] = sum_k modes[t][k] sY_k(R_p) -- the dense contraction of
// scri/asymptotic_bondi_data/transformations.py:312-334 (`self.psi0.evaluate(distorted_grid_rotors)`), waveform_grid.py:475-484 and
// bms_transformations.py:179 as a building block of its own: the harmonics at the rotors (swsh_kernel) and one 3M product on the matrix
// cores.  modes: c16[n_rows][ld] holding l = ell_min..ell_max; rotors: host f8[n_rotors][4]; out: c16[n_rows][n_rotors], where `mem` says.
extern "C" int bms_evaluate_modes(bms_ctx* c, const void* modes, int mem, int64_t n_rows, int64_t ld, int spin, int ell_min, int ell_max,
                                  const double* rotors, int64_t n_rotors, void* out) try {
  if (!c || !modes || !rotors || !out) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  if (n_rows < 0 || n_rotors < 0 || ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad sizes");
  if (ell_max > MAX_ELL || n_rotors > (1LL << 26)) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d at %lld directions: too large", ell_max, (long long)n_rotors);
  const int nm = LM_total_size(ell_min, ell_max);
  if (ld < nm) return fail(c, BMS_ERR_INVALID, "row stride smaller than the number of modes");
  if (n_rows == 0 || n_rotors == 0) return BMS_OK;
  hipStream_t S = c->stream;
  int rc;
  void* d_rot;
  if ((rc = upload(c, "ev_rotors", rotors, 32 * (size_t)n_rotors, &d_rot))) return rc;
  const int n_pix = (int)n_rotors;
  const long long ldb = 2 * round_up(n_pix, 64), k_pad = round_up(nm, 8);
  double* d_B;
  if ((rc = dev_buf_t(c, "ev_B", (size_t)k_pad * ldb, &d_B))) return rc;
  HIP_TRY(c, hipMemsetAsync(d_B, 0, sizeof(double) * k_pad * ldb, S));  // (padding, and the rows l < |s| the harmonics do not have)
  TIMED(c, BMS_TAG_SETUP, launch_swsh_matrix_complex(S, (const double*)d_rot, n_pix, spin, ell_min, ell_max, d_B, ldb));
  const double* d_a = (const double*)modes;
  long long lda = 2 * ld;
  double* d_out = (double*)out;
  if (mem == BMS_HOST) {
    double* a;
    if ((rc = dev_buf_t(c, "in_data", (size_t)n_rows * nm * 2, &a))) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(a, (size_t)nm * 16, modes, (size_t)ld * 16, (size_t)nm * 16, (size_t)n_rows, hipMemcpyHostToDevice, S));
    d_a = a, lda = 2LL * nm;
    if ((rc = dev_buf_t(c, "out_data", (size_t)n_rows * n_pix * 2, &d_out))) return rc;
  }
  TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m(S, d_a, lda, d_B, ldb, d_out, 2LL * n_pix, n_rows, n_pix, nm, nullptr, nullptr));
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(out, d_out, (size_t)n_rows * n_pix * 16, hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));  // (the rotors were staged from the caller's array)
  return BMS_OK;
} BMS_CATCH(c)

// Mode-space operators of sf.Modes / ModesTimeSeries (eth, ethbar, bar, real, sums of different l ranges, scalar and per-row
// factors) as one map along the mode axis, see kernels_modes.hip.  Tables idx_* / coef_* are host arrays of n_cols entries;
// a, b, out and row_scale live in `mem`.  b may be NULL (one-sided map); out may alias neither input unless every idx is
// the identity.
extern "C" int bms_mode_map(bms_ctx* c, void* out, int64_t ld_out, int64_t n_rows, int n_cols, const void* a, int64_t ld_a,
                            const int32_t* idx_a, const void* coef_a, int conj_a, const void* b, int64_t ld_b,
                            const int32_t* idx_b, const void* coef_b, int conj_b, const double* row_scale, int mem) try {
  if (!c || !out || !a || !idx_a || !coef_a) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  if (b && (!idx_b || !coef_b)) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_rows < 0 || n_cols <= 0 || ld_out < n_cols || ld_a <= 0 || (b && ld_b <= 0)) return fail(c, BMS_ERR_INVALID, "bad sizes");
  if (n_rows == 0) return BMS_OK;
  int max_a = -1, max_b = -1;
  for (int j = 0; j < n_cols; ++j) {
    max_a = std::max(max_a, (int)idx_a[j]);
    if (b) max_b = std::max(max_b, (int)idx_b[j]);
  }
  if (max_a >= ld_a || (b && max_b >= ld_b)) return fail(c, BMS_ERR_INVALID, "a source column lies beyond the row stride");
  hipStream_t S = c->stream;
  int rc;
  void* vp;
  ModeMapSide A{}, B{};
  if ((rc = upload(c, "mm_idx_a", idx_a, sizeof(int32_t) * n_cols, &vp))) return rc;
  A.idx = (const int*)vp;
  if ((rc = upload(c, "mm_coef_a", coef_a, 16 * (size_t)n_cols, &vp))) return rc;
  A.coef = (const double*)vp;
  A.ld = ld_a;
  A.conj = conj_a;
  if (b) {
    if ((rc = upload(c, "mm_idx_b", idx_b, sizeof(int32_t) * n_cols, &vp))) return rc;
    B.idx = (const int*)vp;
    if ((rc = upload(c, "mm_coef_b", coef_b, 16 * (size_t)n_cols, &vp))) return rc;
    B.coef = (const double*)vp;
    B.ld = ld_b;
    B.conj = conj_b;
  }
  const double* d_rs = row_scale;
  double* d_out = (double*)out;
  if (mem == BMS_HOST) {
    const double* d;
    if ((rc = stage_in(c, "in_data", a, mem, ((size_t)(n_rows - 1) * ld_a + max_a + 1) * 16, &d))) return rc;
    A.data = d;
    if (b) {
      if ((rc = stage_in(c, "in_aux0", b, mem, ((size_t)(n_rows - 1) * ld_b + max_b + 1) * 16, &d))) return rc;
      B.data = d;
    }
    if (row_scale) {
      if ((rc = upload(c, "mm_rows", row_scale, 8 * (size_t)n_rows, &vp))) return rc;
      d_rs = (const double*)vp;
    }
    if ((rc = dev_buf_t(c, "out_data", (size_t)n_rows * n_cols * 2, &d_out))) return rc;
  } else {
    A.data = (const double*)a;
    B.data = (const double*)b;
  }
  TIMED(c, BMS_TAG_POINTWISE, launch_mode_map(S, d_out, mem == BMS_HOST ? n_cols : ld_out, n_rows, n_cols, A, B, d_rs));
  if (mem == BMS_HOST)
    HIP_TRY(c, hipMemcpy2DAsync(out, (size_t)ld_out * 16, d_out, (size_t)n_cols * 16, (size_t)n_cols * 16, (size_t)n_rows,
                                hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));  // the tables were staged from caller memory
  return BMS_OK;
} BMS_CATCH(c)

extern "C" int bms_row_norm(bms_ctx* c, const void* data, int64_t ld, int64_t n_rows, int n_cols, int mem, int take_sqrt, double* out) try {
  if (!c || !data || !out) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_rows < 0 || n_cols < 0 || ld < n_cols) return fail(c, BMS_ERR_INVALID, "bad sizes");
  if (n_rows == 0) return BMS_OK;
  hipStream_t S = c->stream;
  int rc;
  const double* d_in = (const double*)data;
  double* d_out = out;
  if (mem == BMS_HOST) {
    if (n_cols == 0) {
      std::memset(out, 0, sizeof(double) * n_rows);
      return BMS_OK;
    }
    if ((rc = stage_in(c, "in_data", data, mem, ((size_t)(n_rows - 1) * ld + n_cols) * 16, &d_in))) return rc;
    if ((rc = dev_buf_t(c, "norm_out", (size_t)n_rows, &d_out))) return rc;
  }
  TIMED(c, BMS_TAG_POINTWISE, launch_row_norm(S, d_in, ld, n_rows, n_cols, take_sqrt, d_out));
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(out, d_out, sizeof(double) * n_rows, hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));
  return BMS_OK;
} BMS_CATCH(c)

// ModesTimeSeries.grid_multiply (scri/modes_time_series.py:142-202): both mode sets (l_min = 0) are synthesised on the
// (2W+1) x (2W+1) equiangular grid, multiplied there, and the product (spin s_a + s_b) is analysed up to output_ell_max.
extern "C" int bms_grid_multiply(bms_ctx* c, const void* a, int spin_a, int ell_max_a, const void* b, int spin_b, int ell_max_b,
                                 int mem, int64_t n_times, int working_ell_max, int output_ell_max, void* out) try {
  if (!c || !a || !b || !out) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (ell_max_a < 0 || ell_max_b < 0 || working_ell_max < 1 || output_ell_max < 0 || output_ell_max > working_ell_max)
    return fail(c, BMS_ERR_INVALID, "bad l ranges (need 0 <= output_ell_max <= working_ell_max)");
  if (ell_max_a > MAX_ELL || ell_max_b > MAX_ELL || working_ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "l beyond %d", MAX_ELL);
  // (factors up to |s| = 4, as bms_salm2map / bms_map2salm take them: the boost flux multiplies ethbar h, s = -3)
  if (std::abs(spin_a) > 4 || std::abs(spin_b) > 4 || std::abs(spin_a + spin_b) > 4)
    return fail(c, BMS_ERR_UNSUPPORTED, "spin weights beyond +-4 are not supported");
  if (n_times <= 0) return BMS_OK;
  // A grid that resolves the product (band limit B = l_a + l_b <= working_ell_max) gives the modes l <= output_ell_max exactly
  // (up to rounding) as soon as 2 W + 1 > B + output_ell_max and 2 W - 1 >= B -- phi sampling and the extended theta transform of
  // map2salm -- so the smallest such W serves: the reference's default (W = B, output l_a) needs 2.3 times fewer pixels, and for
  // l_a + l_b <= 25 it is a grid the separable synthesis and the fused analysis take (n_theta <= 40).  A caller's smaller W
  // (aliasing, as in the reference) is kept as given.
  if (working_ell_max >= ell_max_a + ell_max_b && !c->opt.on(OPT_GRID_MULTIPLY_FULL_GRID)) {
    const int B = ell_max_a + ell_max_b;
    working_ell_max = std::max({(B + output_ell_max + 1) / 2, (B + 2) / 2, output_ell_max, 1});
  }
  const int n_theta = 2 * working_ell_max + 1, n_phi = n_theta, n_pix = n_theta * n_phi;
  const int nma = (ell_max_a + 1) * (ell_max_a + 1), nmb = (ell_max_b + 1) * (ell_max_b + 1);
  const int n_out = (output_ell_max + 1) * (output_ell_max + 1);
  hipStream_t S = c->stream;
  int rc;
  // grid rotors R(theta_j, phi_k) in the natural order the analysis expects
  std::vector<double> rot(4 * (size_t)n_pix);
  for (int j = 0; j < n_theta; ++j)
    for (int k = 0; k < n_phi; ++k) {
      const Quat q = from_spherical_coords(M_PI * j / (n_theta - 1), (2 * M_PI) * k / n_phi);
      double* r = &rot[4 * ((size_t)j * n_phi + k)];
      r[0] = q.w, r[1] = q.x, r[2] = q.y, r[3] = q.z;
    }
  // the equiangular grid itself: both syntheses are separable where the kernel takes the shape (kernels_synthesis.hip)
  SynthesisPlan syn_a, syn_b;
  bool sep = n_times >= 2 && ell_max_a >= 1 && ell_max_b >= 1 && !c->opt.on(OPT_NO_SEPARABLE_SYNTHESIS);
  if (sep) {
    if ((rc = build_synthesis(c, n_theta, n_phi, spin_a, 0, ell_max_a, syn_a))) return rc;
    if ((rc = build_synthesis(c, n_theta, n_phi, spin_b, 0, ell_max_b, syn_b))) return rc;
    sep = (syn_a.nt != 0 || syn_a.large) && (syn_b.nt != 0 || syn_b.large);
  }
  void* vp;
  const long long P2 = 2LL * n_pix, ldb = round_up(P2, 128);
  double *d_Ba = nullptr, *d_Bb = nullptr;
  if (!sep) {
    if ((rc = upload(c, "gm_rotors", rot.data(), 8 * rot.size(), &vp))) return rc;
    const double* d_rot = (const double*)vp;
    if ((rc = dev_buf_t(c, "gm_Ba", (size_t)round_up(nma, 8) * ldb, &d_Ba))) return rc;
    if ((rc = dev_buf_t(c, "gm_Bb", (size_t)round_up(nmb, 8) * ldb, &d_Bb))) return rc;
    HIP_TRY(c, hipMemsetAsync(d_Ba, 0, sizeof(double) * round_up(nma, 8) * ldb, S));
    HIP_TRY(c, hipMemsetAsync(d_Bb, 0, sizeof(double) * round_up(nmb, 8) * ldb, S));
    TIMED(c, BMS_TAG_SETUP, launch_swsh_matrix_complex(S, d_rot, n_pix, spin_a, 0, ell_max_a, d_Ba, ldb));
    TIMED(c, BMS_TAG_SETUP, launch_swsh_matrix_complex(S, d_rot, n_pix, spin_b, 0, ell_max_b, d_Bb, ldb));
  }
  AnalysisPlan ana;
  if ((rc = build_analysis(c, "gm", n_theta, n_phi, spin_a + spin_b, 0, output_ell_max, ana))) return rc;
  const double *d_a, *d_b;
  if ((rc = stage_in(c, "in_data", a, mem, (size_t)n_times * nma * 16, &d_a))) return rc;
  if ((rc = stage_in(c, "in_aux0", b, mem, (size_t)n_times * nmb * 16, &d_b))) return rc;
  double* d_out = (double*)out;
  if (mem == BMS_HOST)
    if ((rc = dev_buf_t(c, "out_data", (size_t)n_times * n_out * 2, &d_out))) return rc;
  // (the separable kernel reads one more complex number per row -- the eliminated constant of the transformation's series: the
  // operands are copied to rows with a zero there)
  // (the two-kernel form reads the plain rows)
  double *d_ap = nullptr, *d_bp = nullptr;
  if (sep && syn_a.nt) {
    if ((rc = dev_buf_t(c, "gm_a_pad", (size_t)n_times * (nma + 1) * 2, &d_ap))) return rc;
    HIP_TRY(c, hipMemset2DAsync(d_ap + 2 * nma, (size_t)(nma + 1) * 16, 0, 16, (size_t)n_times, S));
    HIP_TRY(c, hipMemcpy2DAsync(d_ap, (size_t)(nma + 1) * 16, d_a, (size_t)nma * 16, (size_t)nma * 16, (size_t)n_times, hipMemcpyDeviceToDevice, S));
  }
  if (sep && syn_b.nt) {
    if ((rc = dev_buf_t(c, "gm_b_pad", (size_t)n_times * (nmb + 1) * 2, &d_bp))) return rc;
    HIP_TRY(c, hipMemset2DAsync(d_bp + 2 * nmb, (size_t)(nmb + 1) * 16, 0, 16, (size_t)n_times, S));
    HIP_TRY(c, hipMemcpy2DAsync(d_bp, (size_t)(nmb + 1) * 16, d_b, (size_t)nmb * 16, (size_t)nmb * 16, (size_t)n_times, hipMemcpyDeviceToDevice, S));
  }
  // chunks of time rows: two grids of 16 n_pix bytes per row
  int64_t chunk = (int64_t)std::max(64.0, (double)c->ws_limit / (2.0 * P2 * 8.0));
  chunk = std::min<int64_t>(chunk, n_times);
  for (int64_t r0 = 0, rows; r0 < n_times; r0 += rows) {
    rows = std::min<int64_t>(chunk, n_times - r0);
    if (n_times - (r0 + rows) == 1) --rows;  // (never a last chunk of one row: the separable kernel walks rows in pairs)
    double *d_Ga, *d_Gb;
    if ((rc = dev_buf_t(c, "Y", (size_t)rows * P2, &d_Ga))) return rc;
    if ((rc = dev_buf_t(c, "R", (size_t)rows * P2, &d_Gb))) return rc;
    if (sep) {
      if ((rc = syn_a.nt ? run_synthesis(c, syn_a, d_ap + r0 * (nma + 1) * 2, 2LL * (nma + 1), rows, nullptr, d_Ga, P2)
                         : run_synthesis(c, syn_a, d_a + r0 * nma * 2, 2LL * nma, rows, nullptr, d_Ga, P2)))
        return rc;
      if ((rc = syn_b.nt ? run_synthesis(c, syn_b, d_bp + r0 * (nmb + 1) * 2, 2LL * (nmb + 1), rows, nullptr, d_Gb, P2)
                         : run_synthesis(c, syn_b, d_b + r0 * nmb * 2, 2LL * nmb, rows, nullptr, d_Gb, P2)))
        return rc;
    } else {
      TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m(S, d_a + r0 * nma * 2, 2LL * nma, d_Ba, ldb, d_Ga, P2, rows, n_pix, nma, nullptr, nullptr));
      TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m(S, d_b + r0 * nmb * 2, 2LL * nmb, d_Bb, ldb, d_Gb, P2, rows, n_pix, nmb, nullptr, nullptr));
    }
    TIMED(c, BMS_TAG_POINTWISE, launch_cmul(S, d_Ga, d_Gb, d_Ga, rows * (long long)n_pix));
    if ((rc = run_analysis(c, ana, d_Ga, rows, d_out + r0 * n_out * 2, 2LL * n_out))) return rc;
  }
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(out, d_out, (size_t)n_times * n_out * 16, hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));
  return BMS_OK;
} BMS_CATCH(c)

// ====================================================================================================== storage formats
// scri/utilities.py:194-232: XOR differencing of a time series in place (rows of 64-bit words)
extern "C" int bms_xor_timeseries(bms_ctx* c, void* data, int mem, int64_t n_rows, int64_t words_per_row, int reverse) try {
  if (!c || !data) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_rows < 0 || words_per_row < 0) return fail(c, BMS_ERR_INVALID, "negative size");
  if (n_rows == 0 || words_per_row == 0) return BMS_OK;
  const size_t bytes = (size_t)n_rows * words_per_row * 8;
  hipStream_t S = c->stream;
  int rc;
  uint64_t *d_in = (uint64_t*)data, *d_out, *d_carry = nullptr;
  if (mem == BMS_HOST) {
    if ((rc = dev_buf_t(c, "bits_in", bytes / 8, &d_in))) return rc;
    HIP_TRY(c, hipMemcpyAsync(d_in, data, bytes, hipMemcpyHostToDevice, S));
  }
  if ((rc = dev_buf_t(c, "bits_out", bytes / 8, &d_out))) return rc;
  if (reverse)
    if ((rc = dev_buf_t(c, "bits_carry", (size_t)xor_carry_words(n_rows, words_per_row), &d_carry))) return rc;
  TIMED(c, BMS_TAG_POINTWISE, launch_xor_timeseries(S, d_in, d_out, d_carry, n_rows, words_per_row, reverse));
  HIP_TRY(c, hipMemcpyAsync(data, d_out, bytes, mem == BMS_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, S));
  HIP_TRY(c, hipStreamSynchronize(S));
  return BMS_OK;
} BMS_CATCH(c)

// scri/utilities.py:271-406: the function multishuffle(shuffle_widths, forward) returns, applied to n elements
extern "C" int bms_multishuffle(bms_ctx* c, const void* in, void* out, int mem, int64_t n, const int* widths, int n_widths,
                                int forward) try {
  if (!c || !in || !out || !widths) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  int bit_width = 0;
  for (int i = 0; i < n_widths; ++i) {
    if (widths[i] < 1) return fail(c, BMS_ERR_INVALID, "shuffle widths must be positive");
    bit_width += widths[i];
  }
  if (n_widths < 1 || n_widths > 64 || (bit_width != 8 && bit_width != 16 && bit_width != 32 && bit_width != 64))
    return fail(c, BMS_ERR_INVALID, "Total bit width must be one of [8, 16, 32, 64], not %d", bit_width);
  if (n < 0) return fail(c, BMS_ERR_INVALID, "negative size");
  if (n == 0) return BMS_OK;
  const size_t bytes = (size_t)n * (bit_width / 8);
  hipStream_t S = c->stream;
  int rc;
  const void* d_in = in;
  void* d_out = out;
  if (mem == BMS_HOST) {
    uint64_t *a, *b;
    if ((rc = dev_buf_t(c, "bits_in", (bytes + 7) / 8 + 1, &a))) return rc;
    if ((rc = dev_buf_t(c, "bits_out", (bytes + 7) / 8 + 1, &b))) return rc;
    HIP_TRY(c, hipMemcpyAsync(a, in, bytes, hipMemcpyHostToDevice, S));
    d_in = a, d_out = b;
  }
  TIMED(c, BMS_TAG_POINTWISE, launch_multishuffle(S, d_in, d_out, n, widths, n_widths, bit_width, forward));
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));
  return BMS_OK;
} BMS_CATCH(c)

// scri/utilities.py:235-268: Fletcher-32 over the data viewed as 16-bit words (n_bytes must be even)
extern "C" int bms_fletcher32(bms_ctx* c, const void* data, int mem, int64_t n_bytes, uint32_t* checksum) try {
  if (!c || !checksum || (!data && n_bytes)) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_bytes < 0 || (n_bytes & 1)) return fail(c, BMS_ERR_INVALID, "the data must be viewable as 16-bit words");
  *checksum = 0;
  if (n_bytes == 0) return BMS_OK;
  hipStream_t S = c->stream;
  int rc;
  const void* d_in = data;
  if (mem == BMS_HOST) {
    uint64_t* a;
    if ((rc = dev_buf_t(c, "bits_in", (size_t)(n_bytes + 7) / 8, &a))) return rc;
    HIP_TRY(c, hipMemcpyAsync(a, data, (size_t)n_bytes, hipMemcpyHostToDevice, S));
    d_in = a;
  }
  unsigned long long* d_acc;
  if ((rc = dev_buf_t(c, "bits_acc", 2, &d_acc))) return rc;
  HIP_TRY(c, hipMemsetAsync(d_acc, 0, 16, S));
  TIMED(c, BMS_TAG_POINTWISE, launch_fletcher32(S, d_in, n_bytes / 2, d_acc));
  unsigned long long acc[2];
  HIP_TRY(c, hipMemcpyAsync(acc, d_acc, 16, hipMemcpyDeviceToHost, S));
  HIP_TRY(c, hipStreamSynchronize(S));
  *checksum = (uint32_t)((acc[1] % 65535) << 16 | (acc[0] % 65535));
  return BMS_OK;
} BMS_CATCH(c)
