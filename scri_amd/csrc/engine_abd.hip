// AsymptoticBondiData transformation: bms_transform_abd (+ _shard, _pipelined, _pipelined_part)
// (engine.h: the split of the engine by entry family; include/scri_amd.h: the C ABI)
#include "engine.h"

// ====================================================================================================== ABD flavour
static int transform_abd_impl(bms_ctx* c, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                              const bms_transformation* tr, const bms_shard* sh, double* u_out, void* raw_out, int64_t* n_times_out,
                              int64_t* first_index_out);

extern "C" int bms_transform_abd_shard(bms_ctx* c, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                                       const bms_transformation* tr, const bms_shard* sh, double* u_out, void* raw_out,
                                       int64_t* n_times_out, int64_t* first_index_out) try {
  if (!c) return BMS_ERR_INVALID;
  return with_smaller_chunks(c, [&] { return transform_abd_impl(c, u, raw, mem, n_times, ell_max, tr, sh, u_out, raw_out, n_times_out, first_index_out); });
} BMS_CATCH(c)

static int transform_abd_impl(bms_ctx* c, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                              const bms_transformation* tr, const bms_shard* sh, double* u_out, void* raw_out, int64_t* n_times_out,
                              int64_t* first_index_out) {
  if (!u || !raw || !tr || !u_out || !raw_out || !n_times_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t n = n_times;
  int64_t t_lo, t_hi;
  time_window(n, sh, t_lo, t_hi);
  bool regular_mesh = true;
  // (scipy's CubicSpline, the interpolant of this flavour, takes 2 and 3 samples too: line and parabola)
  int rc = validate_common(c, n, u, tr, t_lo, t_hi, &regular_mesh, 2);
  if (rc) return rc;
  const bool short_series = n < 4;
  if (short_series && sh && !(sh->data_row0 == 0 && sh->data_rows == n && sh->col_parts <= 1))
    return fail(c, BMS_ERR_UNSUPPORTED, "a series of %lld samples cannot be sharded", (long long)n);
  if (!regular_mesh && sh && !(sh->data_row0 == 0 && sh->data_rows == n && sh->col_parts <= 1))
    return fail(c, BMS_ERR_UNSUPPORTED,
                "the time steps vary by more than 1e3 within 48 samples: such a series is transformed with exact untiled spline "
                "recurrences, which a time shard cannot provide");
  if (ell_max < 0 || tr->ell_max_out < 0) return fail(c, BMS_ERR_INVALID, "bad ell_max");
  if (ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", ell_max, MAX_ELL);
  static const int spins[6] = {2, 1, 0, -1, -2, 2};  // psi0..psi4, sigma
  const int nm = (ell_max + 1) * (ell_max + 1);
  const int n_out = (tr->ell_max_out + 1) * (tr->ell_max_out + 1);
  const int lst = tr->ell_max_supertranslation;
  const cplx* st = (const cplx*)tr->supertranslation;

  // ---- per-pixel tables on the GPU (transformations.py:306-321), output window on the host (:391-396)
  hipStream_t S = c->stream;
  std::vector<cplx> c1((size_t)(lst + 1) * (lst + 1)), c2((size_t)(lst + 1) * (lst + 1));
  for (int l = 0; l <= lst; ++l)
    for (int m = -l; m <= l; ++m) {
      const cplx a = st[LM_index(l, m, 0)];
      const double f1 = std::sqrt((double)l * (l + 1.0)) / std::sqrt(2.0);                                          // eth alpha / sqrt2
      const double f2 = 0.5 * (std::sqrt((double)l * (l + 1.0)) * (l >= 1 ? std::sqrt((l - 1.0) * (l + 2.0)) : 0.0));  // eth eth alpha / 2
      c1[LM_index(l, m, 0)] = {f1 * a.re, f1 * a.im};
      c2[LM_index(l, m, 0)] = {f2 * a.re, f2 * a.im};
    }
  const double* v = tr->boost_velocity;
  const cplx cv[4] = {{0, 0},
                      {v[0] * std::sqrt(2 * M_PI / 3), v[1] * std::sqrt(2 * M_PI / 3)},
                      {v[2] * std::sqrt(4 * M_PI / 3), 0},
                      {-v[0] * std::sqrt(2 * M_PI / 3), v[1] * std::sqrt(2 * M_PI / 3)}};
  // Without a boost (every supertranslation and rotation step of map_to_superrest_frame, map_to_superrest_frame.py:443,610,641)
  // the six dense products give way to the separable synthesis on the rotated modes (kernels_synthesis_large.hip).
  // A boost along the polar axis of the rotated grid keeps the rings (separable_rotor_grid): the same synthesis at the aberrated
  // colatitudes; the mixing is time dependent then and stays on the grid.
  SynthesisPlan syn5[5];
  const bool no_boost = v[0] == 0 && v[1] == 0 && v[2] == 0;
  std::vector<double> ring_theta;
  bool sep = !(sh && sh->col_parts > 1) && tr->n_theta >= 3 && !c->opt.on(OPT_NO_SEPARABLE_SYNTHESIS) &&
             (no_boost || (axis_boost_pays(c, (ell_max + 1) * (ell_max + 1), tr->n_theta, tr->n_phi) &&
                           large_synthesis_route(c, tr->n_theta, tr->n_phi, 0, ell_max) && separable_rotor_grid(c, tr, ring_theta)));
  for (int si = 0; si < 5 && sep; ++si) {
    if ((rc = build_synthesis(c, tr->n_theta, tr->n_phi, si - 2, 0, ell_max, syn5[si], no_boost ? nullptr : &ring_theta))) return rc;
    sep = syn5[si].large;
  }
  PixelTables T;
  DevPixel DP;
  // (pieces of a pipelined call share the per-direction tables and the knot tables of the whole series)
  PieceTables* shared = c->async_pieces ? static_cast<PieceTables*>(c->piece_tables) : nullptr;
  if (shared && c->piece_tables_valid) {
    T = shared->T;
    DP = shared->DP;
  } else {
    if ((rc = device_pixel_tables(c, tr, T, 2, 0, 0, &c1, &c2, cv, DP, sep ? 0 : column_plan(c, tr, n_out)))) return rc;
    if (shared) {
      shared->T = T;
      shared->DP = DP;
      c->piece_tables_valid = true;
    }
  }
  const int n_cols = T.n_pix;
  const bool col_split = sh && sh->col_parts > 1;
  int cA, cB;
  if ((rc = column_range(c, sh, n_cols, cA, cB))) return rc;
  const int n_pix = cB - cA;  // columns this call synthesises and splines
  // window: timeprime = (u - tt) / gamma  (division, unlike the WaveformModes flavour)
  int64_t i_lo, i_hi;
  output_window_abd(T, u, n, i_lo, i_hi);
  // the shard's share of the window, and the rows of the global series it was given
  int64_t row0 = 0, rows_avail = n, fs_out = n;
  if (sh) {
    row0 = sh->data_row0;
    rows_avail = sh->data_rows;
    if (row0 < 0 || rows_avail < 0 || row0 + rows_avail > n || sh->out_i1 < sh->out_i0) return fail(c, BMS_ERR_INVALID, "bad shard description");
    i_lo = std::max(i_lo, sh->out_i0);
    i_hi = std::max(i_lo, std::min(i_hi, sh->out_i1));
    fs_out = sh->out_i1 - sh->out_i0;
  }
  const int64_t n_new = i_hi - i_lo;
  *n_times_out = n_new;
  if (first_index_out) *first_index_out = i_lo;
  for (int64_t i = 0; i < n_new; ++i) u_out[i] = (u[i_lo + i] - T.tt) / T.gamma;
  if (n_new == 0) return BMS_OK;
  double *d_rot = DP.rotors, *d_skewa = DP.skew_a + cA, *d_skewb = DP.skew_b + cA, *d_alpha = DP.alpha + cA, *d_ethk = DP.ethk + 2 * cA,
         *d_etha = DP.etha + 2 * cA, *d_ethetha = DP.ethetha + 2 * cA, *d_ik = DP.ik + cA, *d_ik3 = DP.ik3 + cA;
  double* d_x;
  // the Horner mixing has time-dependent coefficients, so the elimination stays on the grid; the B-spline form still saves
  // the back substitution its second input stream (kernels_bspline.hip)
  const bool bsg = n >= 8 && regular_mesh && !c->opt.on(OPT_NO_BSPLINE);
  SplineTable* d_tab = nullptr;
  BsplineTable* d_bstab = nullptr;
  BsplineForward* d_bsfwd = nullptr;
  if (short_series) {
    void* vp;
    if ((rc = upload(c, "times", u, 8 * (size_t)n, &vp))) return rc;
    d_x = (double*)vp;
  } else if (bsg && shared && shared->times_valid) {
    d_x = shared->d_x, d_bstab = shared->d_bstab, d_bsfwd = shared->d_bsfwd;
  } else if (bsg && shared) {
    rc = upload_times_bspline(c, u, n, 0, n, 0, n, &d_x, &d_bstab, &d_bsfwd);
    shared->d_x = d_x, shared->d_bstab = d_bstab, shared->d_bsfwd = d_bsfwd;
    shared->times_valid = rc == BMS_OK;
  } else if (bsg)
    rc = upload_times_bspline(c, u, n, t_lo, t_hi, row0, row0 + rows_avail, &d_x, &d_bstab, &d_bsfwd);
  else
    rc = upload_times(c, u, n, t_lo, t_hi, row0, row0 + rows_avail, &d_x, &d_tab);
  if (rc) return rc;

  const long long P2 = 2LL * n_pix, ldg = round_up(P2, 16), ldb = round_up(2LL * n_cols, 128);
  const int K = 2 * nm;
  const long long brows = round_up(K, 16);
  // five distinct spins: matrices / analysis plans indexed by spin + 2
  double *d_B[5], *d_At[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  long long ld_at = 0;
  AnalysisPlan ana[5];
  for (int si = 0; si < 5; ++si) {
    char nm1[32], tag[16];
    snprintf(nm1, sizeof nm1, "abd_B%d", si);
    snprintf(tag, sizeof tag, "abd%d", si);
    if (!sep) {
      if ((rc = dev_buf_t(c, nm1, (size_t)brows * ldb, &d_B[si]))) return rc;
      HIP_TRY(c, hipMemsetAsync(d_B[si], 0, sizeof(double) * brows * ldb, S));
      TIMED(c, BMS_TAG_SETUP, launch_swsh_matrix_complex(S, d_rot, n_cols, si - 2, 0, ell_max, d_B[si], ldb));
    }
    if ((rc = build_analysis(c, tag, T.n_theta, T.n_phi, si - 2, 0, tr->ell_max_out, ana[si]))) return rc;
    if (col_split && n_pix > 0) {
      snprintf(nm1, sizeof nm1, "abd_At%d", si);
      if ((rc = part_analysis_matrix(c, ana[si], nm1, n_cols, DP.col_of_pixel, &d_At[si], &ld_at))) return rc;
    }
  }

  const long long ldG = (!col_split && analysis_reads_contiguous_rows(ana[2])) ? P2 : ldg;  // row stride of the evaluated grids
  const double* d_raw;
  if ((rc = stage_in(c, "in_data", raw, mem, (size_t)6 * rows_avail * nm * 16, &d_raw))) return rc;
  double* d_out = (double*)raw_out;
  if (mem == BMS_HOST)
    if ((rc = dev_buf_t(c, "out_data", (size_t)6 * fs_out * n_out * 2, &d_out))) return rc;
  // Without a boost the Horner mixing has time-independent coefficients (k = 1, eth k = 0: X = -eth alpha), so it commutes with the
  // spline's forward elimination: that runs on the MODES (6 x (l_max+1)^2 columns + the constant series, instead of six grids), the
  // fields are synthesised as eliminated coefficients and mixed on their way out of the phi stage (phi_synthesis_mix6_kernel)
  const bool fused_mix = sep && no_boost && bsg && !short_series && !c->opt.on(OPT_NO_FUSED_ABD_MIX) && large_synthesis_route(c, tr->n_theta, tr->n_phi, 0, ell_max) &&
                         abd_mix6_supported(tr->n_theta, tr->n_phi, ell_max);
  const long long ld_af = round_up(2LL * (nm + 1), 16);
  double* d_Af = nullptr;
  if (fused_mix) {
    if ((rc = dev_buf_t(c, "abd_Af", (size_t)6 * rows_avail * ld_af, &d_Af))) return rc;
    for (int f = 0; f < 6; ++f)
      TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_bspline_forward_modes(S, d_raw + (size_t)f * rows_avail * nm * 2, 2LL * nm, nm, d_Af + (size_t)f * rows_avail * ld_af,
                                                                    ld_af, row0, rows_avail, n, d_bsfwd, SPLINE_TILE, SPLINE_HALO, 1));
    const double* q = tr->frame_rotation;
    if (!(q[0] == 1.0 && q[1] == 0.0 && q[2] == 0.0 && q[3] == 0.0)) {
      const double sp[4] = {q[0], q[3], q[2], q[1]};  // (w + i z, y + i x); the constant column stays
      if ((rc = rotate_impl(c, d_Af, BMS_DEVICE, 6 * rows_avail, ld_af / 2, 0, ell_max, sp, false, false))) return rc;
    }
  } else if (sep) {
    // the six fields as seen from the rotated frame, sYlm(F G) = sum_m' D_{m m'}(F) sYlm'(G): [6][rows][nm] is one series of
    // 6 x rows steps for the rotation kernel -- in place in the staging copy of a host caller, in a copy of device data
    const double* q = tr->frame_rotation;
    if (!(q[0] == 1.0 && q[1] == 0.0 && q[2] == 0.0 && q[3] == 0.0)) {
      const double sp[4] = {q[0], q[3], q[2], q[1]};  // (w + i z, y + i x)
      double* d_copy = const_cast<double*>(d_raw);
      if (mem == BMS_DEVICE) {
        if ((rc = dev_buf_t(c, "abd_rot_in", (size_t)6 * rows_avail * nm * 2, &d_copy))) return rc;
        HIP_TRY(c, hipMemcpyAsync(d_copy, d_raw, (size_t)6 * rows_avail * nm * 16, hipMemcpyDeviceToDevice, S));
        d_raw = d_copy;
      }
      if ((rc = rotate_impl(c, d_copy, BMS_DEVICE, 6 * rows_avail, nm, 0, ell_max, sp, false, false))) return rc;
    }
  }

  if (n_pix == 0) {  // more parts than column tiles: this part contributes nothing
    for (int f = 0; f < 6; ++f) {
      if (mem == BMS_HOST)
        std::memset((char*)raw_out + (size_t)f * fs_out * n_out * 16, 0, (size_t)n_new * n_out * 16);
      else
        HIP_TRY(c, hipMemsetAsync(d_out + (size_t)f * fs_out * n_out * 2, 0, (size_t)n_new * n_out * 16, S));
    }
    HIP_TRY(c, hipStreamSynchronize(S));
    return BMS_OK;
  }

  // sigma' = (sigma - eth eth alpha) / k mixes with nothing and its offset and scale do not depend on time: it takes the evaluating
  // product of the WaveformModes route (its spline solved on the modes, evaluated in the product's epilogue, kernels_gemm_eval.hip) and
  // stays out of the mixing + elimination pass and of the back substitution -- two of the four passes over its grid
  const bool sigma_eval = !sep && bsg && !short_series && rows_avail >= 8 && !c->opt.on(OPT_NO_ABD_SIGMA_EVAL);
  double* d_As = nullptr;
  if (sigma_eval) {
    if ((rc = dev_buf_t(c, "abd_As", (size_t)rows_avail * ld_af, &d_As))) return rc;
    TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_bspline_solve_modes(S, d_raw + (size_t)5 * rows_avail * nm * 2, 2LL * nm, nm, d_As, ld_af, row0, rows_avail,
                                                                d_bsfwd, d_bstab, 1));
    // row nm of the spin-2 harmonics (psi0 shares them and stops at row nm - 1) multiplies the solved constant series: -eth eth alpha.
    // Two invariants keep psi0's product, which shares the matrix, correct: the matrix has that row and the 8 rows a K chunk may read
    // past it (checked here), and launch_zgemm3m multiplies rows k >= K of B by ZERO-guarded A (kernels.h says so), because psi0's last
    // K chunk reads this now non-zero row.
    if (brows < (long long)nm + 1 + 8) return fail(c, BMS_ERR_HIP, "internal: the spin-2 matrix has %lld rows, sigma's offset row needs %d", brows, nm + 9);
    TIMED(c, BMS_TAG_SETUP, launch_negated_row(S, DP.ethetha, d_B[4] + (size_t)nm * ldb, 2 * n_cols));
  }

  // ---- chunk loop: 6 fields x (Y, R, G)
  const BsplineSpread spread = skew_spread(T, cA, cB, u);
  const int margin = SPLINE_HALO + 2;
  const double bytes_per_row = 19.0 * ldg * 8.0;  // 6 x (Y, R, G) + F
  int64_t chunk = (int64_t)((double)c->ws_limit / bytes_per_row - 4.0 * margin);
  if (chunk < 4 * margin && chunk < n_new)
    return fail(c, BMS_ERR_NOMEM, "work space limit of %llu bytes holds fewer than %d rows of the six %d-column grids (%.0f bytes each); raise it with bms_ctx_set_workspace_limit",
                (unsigned long long)c->ws_limit, 8 * margin, n_cols, bytes_per_row);
  chunk = std::min<int64_t>(chunk, n_new);
  if (!regular_mesh && chunk < n_new)
    return fail(c, BMS_ERR_UNSUPPORTED, "irregular time axis (steps vary by more than 1e3 within 48 samples): the series does not fit the work space in one piece");
  const int spline_tile = regular_mesh ? SPLINE_TILE : (int)std::min<int64_t>(n + 1, 0x7fffffff);  // one tile: exact recurrences
  for (int64_t c0 = i_lo; c0 < i_hi; c0 += chunk) {
    const int64_t c1_ = std::min<int64_t>(c0 + chunk, i_hi);
    int64_t ja, jb;
    needed_knots(T, u, n, c0, c1_, ja, jb);
    const int64_t g0 = regular_mesh ? std::max<int64_t>(0, ja - margin) : 0, g1 = regular_mesh ? std::min<int64_t>(n, jb + margin + 1) : n;
    const int64_t rows_in = g1 - g0, rows_out = c1_ - c0;
    if (g0 < row0 || g1 > row0 + rows_avail)
      return fail(c, BMS_ERR_INVALID,
                  "shard holds rows [%lld, %lld) but outputs [%lld, %lld) need rows [%lld, %lld): halo too small "
                  "(use bms_shard_plan)",
                  (long long)row0, (long long)(row0 + rows_avail), (long long)c0, (long long)c1_, (long long)g0, (long long)g1);
    double *d_Y = nullptr, *d_R, *d_G;
    if (!fused_mix)
      if ((rc = dev_buf_t(c, "Y", (size_t)6 * rows_in * ldg, &d_Y))) return rc;
    if ((rc = dev_buf_t(c, "R", (size_t)6 * rows_in * ldg, &d_R))) return rc;
    if ((rc = dev_buf_t(c, "G", (size_t)6 * rows_out * ldG, &d_G))) return rc;
    AbdGrids grids;
    if (fused_mix) {
      const size_t f_stride = (size_t)rows_in * (2 * ell_max + 1) * large_analysis_jp(T.n_theta) * 2;
      double* d_F6;
      if ((rc = dev_buf_t(c, "Fsyn6", 6 * f_stride, &d_F6))) return rc;
      const double* F6[6];
      double* out6[6];
      for (int f = 0; f < 6; ++f) {
        F6[f] = d_F6 + f * f_stride;
        out6[f] = d_R + (size_t)f * rows_in * ldg;
        TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_theta_synthesis(S, d_Af + ((size_t)f * rows_avail + (g0 - row0)) * ld_af, ld_af, rows_in, T.n_theta, 0, ell_max,
                                                                syn5[spins[f] + 2].d_T, d_F6 + f * f_stride));
      }
      TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_phi_synthesis_mix6(S, F6, rows_in, T.n_theta, T.n_phi, ell_max, d_etha, d_ethetha,
                                                                 d_Af + (size_t)(g0 - row0) * ld_af + 2LL * nm, ld_af, out6, ldg));
    }
    for (int f = 0; f < 6 && !fused_mix; ++f) {
      grids.y[f] = d_Y + (size_t)f * rows_in * ldg;
      if (f == 5 && sigma_eval) continue;  // (straight to its samples, below)
      if (sep) {
        if ((rc = run_synthesis(c, syn5[spins[f] + 2], d_raw + ((size_t)f * rows_avail + (g0 - row0)) * nm * 2, 2LL * nm, rows_in, nullptr, grids.y[f], ldg)))
          return rc;
      } else
      {
        // (a field of spin weight s has no modes below l = |s|: the s^2 leading columns of its rows and the matching -- zero -- rows of
        // the harmonics stay out of the product; at l_max = 24 that is 78 instead of 79 k-chunks for five of the six fields)
        const int skip = spins[f] * spins[f] < nm ? spins[f] * spins[f] : 0;
        TIMED(c, BMS_TAG_GEMM_SYNTHESIS,
              launch_zgemm3m(S, d_raw + ((size_t)f * rows_avail + (g0 - row0)) * nm * 2 + 2 * skip, 2LL * nm, d_B[spins[f] + 2] + 2 * cA + (size_t)skip * ldb, ldb,
                             grids.y[f], ldg, rows_in, n_pix, K / 2 - skip, nullptr, nullptr));
      }
    }
    if (fused_mix) {
      // (synthesised, mixed and eliminated above)
    } else if (bsg) {  // mixing and elimination of the six fields in one pass over the grids
      AbdGrids elim;
      for (int f = 0; f < 6; ++f) elim.y[f] = d_R + (size_t)f * rows_in * ldg;
      TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_abd_mix_forward(S, grids, elim, ldg, n_pix, g0, rows_in, d_bsfwd, SPLINE_TILE, SPLINE_HALO, d_alpha, d_ethk,
                                                              d_etha, d_ethetha, d_ik, d_ik3, sigma_eval ? 5 : 6));
    } else {
      TIMED(c, BMS_TAG_POINTWISE,
            launch_abd_mix(S, grids, ldg, n_pix, rows_in, d_x + g0, d_alpha, d_ethk, d_etha, d_ethetha, d_ik, d_ik3));
    }
    for (int f = 0; f < 6; ++f) {
      double* Rf = d_R + (size_t)f * rows_in * ldg;
      double* Gf = d_G + (size_t)f * rows_out * ldG;
      if (short_series) {
        TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_short_series_eval(S, grids.y[f], ldg, n_pix, (int)n, d_x, d_x, d_skewa, d_skewb, T.tt, c0, c1_, Gf, ldG));
      } else if (f == 5 && sigma_eval) {
        SplineEval ev;
        ev.table = d_bstab, ev.x = d_x, ev.skew_a = d_skewa, ev.skew_b = d_skewb, ev.tt = T.tt, ev.g0 = g0, ev.n_knots = n;
        ev.i_lo = c0, ev.i_hi = c1_, ev.out = Gf, ev.ldo = ldG;
        ev.search_halfwidth = eval_search_halfwidth(T, cA, cB, u, g0, g1);
        ev.inv_dx = (g1 - g0 >= 2 && u[g1 - 1] > u[g0]) ? (double)(g1 - 1 - g0) / (u[g1 - 1] - u[g0]) : 0.0;
        ev.side = nullptr, ev.side_ld = ldg;
        if (!c->d_eval_stats) {
          HIP_TRY(c, hipMalloc(&c->d_eval_stats, 16));
          HIP_TRY(c, hipMemsetAsync(c->d_eval_stats, 0, 16, S));
        }
        ev.stats = c->d_eval_stats;
        ev.step = c->opt.v[OPT_GEMM_EVAL_STEP] == 61 ? 61 : 64;
        c->eval_tiles += eval_tile_count(rows_in, n_pix, ev.step);
        if (ev.step != 61)
          if ((rc = dev_buf_t(c, "Cside", (size_t)zgemm3m_eval_side_rows(rows_in) * ldg, &ev.side))) return rc;
        const int skip = 4 < nm ? 4 : 0;  // (spin 2: no modes below l = 2)
        TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m_eval(S, d_As + (g0 - row0) * ld_af + 2 * skip, ld_af, d_B[4] + 2 * cA + (size_t)skip * ldb, ldb, rows_in,
                                                             n_pix, nm - skip + 1, DP.col_scale + 2 * cA, ev));
      } else if (bsg) {
        TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_bspline_backward_eval(S, Rf, ldg, n_pix, g0, rows_in, n, d_x, d_bstab, SPLINE_TILE, SPLINE_HALO,
                                                                       d_x, d_skewa, d_skewb, T.tt, c0, c1_, Gf, ldG, &spread));
      } else {
        TIMED(c, BMS_TAG_SPLINE_FORWARD,
              launch_spline_forward(S, grids.y[f], Rf, ldg, n_pix, g0, rows_in, n, d_x, d_tab, spline_tile, SPLINE_HALO));
        TIMED(c, BMS_TAG_SPLINE_BACKWARD,
              launch_spline_backward_eval(S, grids.y[f], Rf, ldg, n_pix, g0, rows_in, n, d_x, d_tab, spline_tile, SPLINE_HALO,
                                          d_x, d_skewa, d_skewb, T.tt, c0, c1_, Gf, ldG));
      }
      double* out_f = d_out + ((size_t)f * fs_out + (c0 - i_lo)) * n_out * 2;
      if (col_split) {
        TIMED(c, BMS_TAG_GEMM_ANALYSIS, launch_zgemm3m(S, Gf, ldG, d_At[spins[f] + 2] + (size_t)cA * ld_at, ld_at, out_f, 2LL * n_out, rows_out, n_out, n_pix,
                                                       nullptr, nullptr));
      } else if ((rc = run_analysis(c, ana[spins[f] + 2], Gf, rows_out, out_f, 2LL * n_out, DP.col_of_pixel, ldG)))
        return rc;
    }
  }
  if (mem == BMS_HOST) {
    for (int f = 0; f < 6; ++f)
      HIP_TRY(c, hipMemcpyAsync((char*)raw_out + (size_t)f * fs_out * n_out * 16, d_out + (size_t)f * fs_out * n_out * 2,
                                (size_t)n_new * n_out * 16, hipMemcpyDeviceToHost, S));
  }
  if (!c->async_pieces) HIP_TRY(c, hipStreamSynchronize(S));  // (a piece of a pipelined call returns without waiting)
  return BMS_OK;
}

// AsymptoticBondiData.transform with host arrays in and out, as the three-stage pipeline of bms_transform_modes_pipelined: the
// rows (+ halo) of the six fields of shard k + 1 travel up, the kernels of shard k run and the results of shard k - 1 travel
// down at the same time.  raw: host c16[6][n][(ell_max+1)^2]; raw_out: host c16[6][i_hi - i_lo][n_out] (best page-locked).
extern "C" int bms_transform_abd_pipelined(bms_ctx* c, const double* u, const void* raw, int64_t n, int ell_max,
                                           const bms_transformation* tr, int pieces, double* u_out, void* raw_out, int64_t* n_times_out) try {
  return bms_transform_abd_pipelined_part(c, u, raw, n, ell_max, tr, pieces, 0, pieces < 1 ? 1 : pieces, u_out, raw_out, n_times_out);
} BMS_CATCH(c)

// Pieces [piece0, piece1) of the same plan (see bms_transform_modes_pipelined_part): u_out / raw_out are the arrays of the whole window.
extern "C" int bms_transform_abd_pipelined_part(bms_ctx* c, const double* u, const void* raw, int64_t n, int ell_max, const bms_transformation* tr,
                                                int pieces, int piece0, int piece1, double* u_out, void* raw_out, int64_t* n_times_out) try {
  if (!c) return BMS_ERR_INVALID;
  if (!u || !raw || !tr || !u_out || !raw_out || !n_times_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  HIP_TRY(c, hipSetDevice(c->device));
  bool regular = true;
  int rc = validate_common(c, n, u, tr, 0, n, &regular);
  if (rc) return rc;
  if (!regular) return fail(c, BMS_ERR_UNSUPPORTED, "the time steps vary by more than 1e3 within 48 samples: not sharded");
  if (ell_max < 0 || tr->ell_max_out < 0) return fail(c, BMS_ERR_INVALID, "bad ell_max");
  if (ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", ell_max, MAX_ELL);
  PixelTables T;
  {
    DevPixel DP;
    const cplx cv0[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    if ((rc = device_pixel_tables(c, tr, T, 0, 0, 0, nullptr, nullptr, cv0, DP, 0))) return rc;
  }
  int64_t i_lo, i_hi;
  output_window_abd(T, u, n, i_lo, i_hi);
  const int64_t n_new = i_hi - i_lo;
  *n_times_out = n_new;
  if (n_new <= 0) return BMS_OK;
  if (pieces < 1) pieces = 1;
  if (pieces > n_new / 8) pieces = (int)std::max<int64_t>(1, n_new / 8);
  const int p0 = std::min(std::max(piece0, 0), pieces), p1 = std::min(std::max(piece1, p0), pieces);
  if (p1 <= p0) return BMS_OK;
  const int64_t nm = (int64_t)(ell_max + 1) * (ell_max + 1), n_out = (int64_t)(tr->ell_max_out + 1) * (tr->ell_max_out + 1);
  std::vector<int64_t> cut(pieces + 1), r0(pieces), r1(pieces);
  int64_t max_rows = 0, max_out = 0;
  for (int k = 0; k <= pieces; ++k) cut[k] = i_lo + (n_new * k) / pieces;
  for (int k = p0; k < p1; ++k) {
    int64_t ja, jb;
    needed_knots(T, u, n, cut[k], cut[k + 1], ja, jb);
    const int margin = SPLINE_HALO + 2;  // as bms_shard_plan
    r0[k] = std::max<int64_t>(0, ja - margin);
    r1[k] = std::min<int64_t>(n, jb + margin + 1);
    max_rows = std::max(max_rows, r1[k] - r0[k]);
    max_out = std::max(max_out, cut[k + 1] - cut[k]);
  }
  double *d_in[2], *d_out[2];
  if ((rc = dev_buf_t(c, "pipe_in0", (size_t)6 * max_rows * nm * 2, &d_in[0]))) return rc;
  if ((rc = dev_buf_t(c, "pipe_in1", (size_t)6 * max_rows * nm * 2, &d_in[1]))) return rc;
  if ((rc = dev_buf_t(c, "pipe_out0", (size_t)6 * max_out * n_out * 2, &d_out[0]))) return rc;
  if ((rc = dev_buf_t(c, "pipe_out1", (size_t)6 * max_out * n_out * 2, &d_out[1]))) return rc;
  if (!c->pipe_up) {
    HIP_TRY(c, hipStreamCreateWithFlags(&c->pipe_up, hipStreamNonBlocking));
    HIP_TRY(c, create_download_stream(c));
  }
  std::vector<hipEvent_t> ev_up(pieces), ev_c(pieces), ev_dn(pieces);
  for (int k = p0; k < p1; ++k) ev_up[k] = ScopedTimer::get(c), ev_c[k] = ScopedTimer::get(c), ev_dn[k] = ScopedTimer::get(c);
  auto give_back = [&]() {
    for (int k = p0; k < p1; ++k) c->event_pool.push_back(ev_up[k]), c->event_pool.push_back(ev_c[k]), c->event_pool.push_back(ev_dn[k]);
  };
  const char* host_in = (const char*)raw;
  char* host_out = (char*)raw_out;
  auto upload_piece = [&](int k) -> hipError_t {  // the six fields' rows [r0, r1) -> c16[6][rows][nm]
    const int64_t rows = r1[k] - r0[k];
    for (int f = 0; f < 6; ++f) {
      const hipError_t e = hipMemcpyAsync(d_in[(k - p0) & 1] + (size_t)f * rows * nm * 2, host_in + ((size_t)f * n + r0[k]) * nm * 16, (size_t)rows * nm * 16,
                                          hipMemcpyHostToDevice, c->pipe_up);
      if (e != hipSuccess) return e;
    }
    return hipEventRecord(ev_up[k], c->pipe_up);
  };
  PieceTables shared_tables;
  struct AsyncScope {
    bms_ctx* c;
    ~AsyncScope() {
      c->async_pieces = false;
      c->piece_tables_valid = false;
      c->piece_tables = nullptr;
    }
  } scope{c};
  c->piece_tables = &shared_tables;
  c->piece_tables_valid = false;
  c->async_pieces = true;
  hipError_t he = upload_piece(p0);
  if (he != hipSuccess) {
    give_back();
    return fail(c, BMS_ERR_HIP, "pipelined upload: %s", hipGetErrorString(he));
  }
  for (int k = p0; k < p1 && rc == BMS_OK; ++k) {
    if (k > p0 && k + 1 < p1 && (he = upload_piece(k + 1)) != hipSuccess) break;
    if ((he = hipStreamWaitEvent(c->stream, ev_up[k], 0)) != hipSuccess) break;
    if (k >= p0 + 2 && (he = hipStreamWaitEvent(c->stream, ev_dn[k - 2], 0)) != hipSuccess) break;  // its output buffer has left
    const bms_shard sh = {r0[k], r1[k] - r0[k], cut[k], cut[k + 1], 0, 0};
    int64_t got = 0, first = 0;
    rc = transform_abd_impl(c, u, d_in[(k - p0) & 1], BMS_DEVICE, n, ell_max, tr, &sh, u_out + (cut[k] - i_lo), d_out[(k - p0) & 1], &got, &first);
    if (rc) break;
    if (k == p0 && p0 + 1 < p1 && (he = upload_piece(p0 + 1)) != hipSuccess) break;  // (after piece 0's blocking table read-back)
    if (got != cut[k + 1] - cut[k] || first != cut[k]) {
      rc = fail(c, BMS_ERR_HIP, "pipelined shard [%lld, %lld) produced %lld rows from %lld", (long long)cut[k], (long long)cut[k + 1],
                (long long)got, (long long)first);
      break;
    }
    if ((he = hipEventRecord(ev_c[k], c->stream)) != hipSuccess) break;
    if ((he = hipEventSynchronize(ev_c[k])) != hipSuccess) break;
    for (int f = 0; f < 6 && he == hipSuccess; ++f)
      he = hipMemcpyAsync(host_out + ((size_t)f * n_new + (cut[k] - i_lo)) * n_out * 16, d_out[(k - p0) & 1] + (size_t)f * got * n_out * 2,
                          (size_t)got * n_out * 16, hipMemcpyDeviceToHost, c->pipe_down);
    if (he != hipSuccess) break;
    if ((he = hipEventRecord(ev_dn[k], c->pipe_down)) != hipSuccess) break;
  }
  (void)hipStreamSynchronize(c->pipe_up);
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->pipe_down);
  give_back();
  if (rc) return rc;
  if (he != hipSuccess) return fail(c, BMS_ERR_HIP, "pipelined transfer: %s", hipGetErrorString(he));
  return BMS_OK;
} BMS_CATCH(c)

// the six fields over several contexts of one process (see bms_transform_modes_multi)
extern "C" int bms_transform_abd_multi(bms_ctx* const* ctxs, int n_ctx, const double* u, const void* raw, int64_t n, int ell_max,
                                       const bms_transformation* tr, int pieces, double* u_out, void* raw_out, int64_t* n_times_out) try {
  return run_dealt_over_contexts(ctxs, n_ctx, pieces, n_times_out, [&](bms_ctx* c, int p0, int p1, int64_t* got) {
    return bms_transform_abd_pipelined_part(c, u, raw, n, ell_max, tr, pieces < 1 ? 1 : pieces, p0, p1, u_out, raw_out, got);
  });
} BMS_CATCH(ctxs && n_ctx > 0 ? ctxs[0] : nullptr)

extern "C" int bms_transform_abd(bms_ctx* c, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                                 const bms_transformation* tr, double* u_out, void* raw_out, int64_t* n_times_out) try {
  return bms_transform_abd_shard(c, u, raw, mem, n_times, ell_max, tr, nullptr, u_out, raw_out, n_times_out, nullptr);
} BMS_CATCH(c)
