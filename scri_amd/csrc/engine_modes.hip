// WaveformModes transformation: bms_transform_modes (+ _shard, _pipelined, _pipelined_part, _series), bms_modes_to_grid, bms_shard_plan, bms_output_window
// (engine.h: the split of the engine by entry family; include/scri_amd.h: the C ABI)
#include "engine.h"

constexpr int SYN_EVAL_MIN_ELL = 15;  // the evaluating separable synthesis (kernels_synthesis_eval.hip) by default from this l_max on

struct PointwiseWM {
  // WM flavour: y = (f0 + sum_i coeff_i f_i X^power_i - off) * scale, see bms_transform_modes
  const double* d_off = nullptr;
  const double* d_scale = nullptr;
  int n_aux = 0;
  double coeff[4];
  int power[4];
  const double *d_alpha = nullptr, *d_xa = nullptr, *d_xb = nullptr;
};

extern "C" int bms_transform_modes(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, double* t_out,
                                   void* data_out, int64_t* n_times_out) try {
  return bms_transform_modes_shard(c, in, tr, nullptr, t_out, data_out, n_times_out, nullptr);
} BMS_CATCH(c)

extern "C" int bms_shard_plan(bms_ctx* c, const double* t, int64_t n, const bms_transformation* tr, int64_t out_i0,
                              int64_t out_i1, int64_t need_rows[2], int64_t window[2]) try {
  // pure host planning: ctx may be NULL (errors then go to bms_last_error(NULL))
  if (!t || !tr || !need_rows || !window) return fail(c, BMS_ERR_INVALID, "NULL argument");
  int rc = validate_common(c, n, t, tr);
  if (rc) return rc;
  PixelTables T;
  build_pixel_tables(tr, T);
  int64_t i_lo, i_hi;
  output_window(T, t, n, i_lo, i_hi);
  window[0] = i_lo;
  window[1] = i_hi;
  const int64_t a = std::max(i_lo, out_i0), b = std::min(i_hi, out_i1);
  if (b <= a) {
    need_rows[0] = need_rows[1] = 0;
    return BMS_OK;
  }
  int64_t ja, jb;
  needed_knots(T, t, n, a, b, ja, jb);
  const int margin = SPLINE_HALO + 2;
  need_rows[0] = std::max<int64_t>(0, ja - margin);
  need_rows[1] = std::min<int64_t>(n, jb + margin + 1);
  return BMS_OK;
} BMS_CATCH(c)

extern "C" int bms_output_window(bms_ctx* c, const double* t, int64_t n, const bms_transformation* tr, int abd, int64_t window[2]) try {
  if (!c) return BMS_ERR_INVALID;
  if (!t || !tr || !window) return fail(c, BMS_ERR_INVALID, "NULL argument");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = validate_common(c, n, t, tr, 0, 0, nullptr, abd ? 2 : 4);
  if (rc) return rc;
  PixelTables T;
  DevPixel DP;
  const cplx cv[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  if ((rc = device_pixel_tables(c, tr, T, 0, 0, 0, nullptr, nullptr, cv, DP, 0))) return rc;
  if (abd)
    output_window_abd(T, t, n, window[0], window[1]);
  else
    output_window(T, t, n, window[0], window[1]);
  return BMS_OK;
} BMS_CATCH(c)

static int transform_modes_impl(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, const bms_shard* sh, double* t_out,
                                void* data_out, int64_t* n_times_out, int64_t* first_index_out, void* grid_out, bool walk_first = false);

extern "C" int bms_transform_modes_shard(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr,
                                         const bms_shard* sh, double* t_out, void* data_out, int64_t* n_times_out,
                                         int64_t* first_index_out) try {
  if (!c) return BMS_ERR_INVALID;
  if (!data_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  return with_smaller_chunks(c, [&] { return transform_modes_impl(c, in, tr, sh, t_out, data_out, n_times_out, first_index_out, nullptr); });
} BMS_CATCH(c)

// Host arrays in, host arrays out, as a three-stage pipeline over time shards of the OUTPUT range: the upload of shard k + 1
// (its rows + halo, bms_shard_plan), the kernels of shard k and the download of shard k - 1 run on three streams, ordered by
// events; the host thread only enqueues.  A long series in host memory waits for PCIe, not for the kernels (cfg3: 456 MB each
// way at 57 GB/s = 8 ms per direction against 6 ms of kernels): one call does upload -> kernels -> download one after the
// other (26 ms), this does them side by side.  Uploads run at full rate from page-locked memory (bms_host_register /
// bms_host_alloc); from pageable memory the runtime stages them.  data_out: host c16[i_hi - i_lo][n_out] (best page-locked).
// Results are those of the sharded path (equal to the one-call path to rounding).  No psi companions (aux) here.
extern "C" int bms_transform_modes_pipelined(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, int pieces,
                                             double* t_out, void* data_out, int64_t* n_times_out) try {
  return bms_transform_modes_pipelined_part(c, in, tr, pieces, 0, pieces < 1 ? 1 : pieces, t_out, data_out, n_times_out);
} BMS_CATCH(c)

// The same for pieces [piece0, piece1) of the `pieces` the output window is cut into: t_out / data_out are the arrays of the WHOLE
// window (every piece lands at its own place), *n_times_out is the whole window's row count.  One process that owns several GPUs
// deals the pieces of one transformation over one context per device, one host thread each (scri_amd/engine.py, `devices=`): every
// context ships its own rows + halo at upload time, so there is no GPU-to-GPU traffic at all, and the results are
// those of the one-context call with the same `pieces`, bit for bit (a piece's arithmetic depends on its cut only).
extern "C" int bms_transform_modes_pipelined_part(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, int pieces, int piece0,
                                                  int piece1, double* t_out, void* data_out, int64_t* n_times_out) try {
  if (!c) return BMS_ERR_INVALID;
  if (!in || !tr || !t_out || !data_out || !n_times_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  if (in->mem != BMS_HOST || in->n_aux != 0) return fail(c, BMS_ERR_INVALID, "the pipelined path takes host data without auxiliary fields");
  if (!in->t || !in->data) return fail(c, BMS_ERR_INVALID, "NULL argument");
  if (in->ell_min < 0 || in->ell_max < in->ell_min) return fail(c, BMS_ERR_INVALID, "bad ell range");
  if (in->ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", in->ell_max, MAX_ELL);
  if (in->ld < LM_total_size(in->ell_min, in->ell_max)) return fail(c, BMS_ERR_INVALID, "row stride smaller than the number of modes");
  if (tr->ell_max_out < std::abs(in->spin_weight)) return fail(c, BMS_ERR_INVALID, "ell_max_out < |s|");
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t n = in->n_times;
  bool regular = true;
  int rc = validate_common(c, n, in->t, tr, 0, n, &regular);
  if (rc) return rc;
  if (!regular) return fail(c, BMS_ERR_UNSUPPORTED, "the time steps vary by more than 1e3 within 48 samples: not sharded");
  // per-direction tables once (on the device, read back), for the window and for every piece's row range
  PixelTables T;
  {
    DevPixel DP;
    const cplx cv0[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    if ((rc = device_pixel_tables(c, tr, T, 0, 0, 0, nullptr, nullptr, cv0, DP, 0))) return rc;
  }
  int64_t i_lo, i_hi;
  output_window(T, in->t, n, i_lo, i_hi);
  const int64_t n_new = i_hi - i_lo;
  *n_times_out = n_new;
  if (n_new <= 0) return BMS_OK;
  if (pieces < 1) pieces = 1;
  if (pieces > n_new / 8) pieces = (int)std::max<int64_t>(1, n_new / 8);
  // (a clamped count keeps the pieces that exist: a caller that dealt a larger count over its contexts still covers every one once)
  const int p0 = std::min(std::max(piece0, 0), pieces), p1 = std::min(std::max(piece1, p0), pieces);
  if (p1 <= p0) return BMS_OK;
  const int n_modes = LM_total_size(in->ell_min, in->ell_max);
  const int s_abs = std::abs(in->spin_weight);
  const int n_out = LM_total_size(s_abs, tr->ell_max_out);
  // plan: output cuts and the input rows each piece needs
  std::vector<int64_t> cut(pieces + 1), r0(pieces), r1(pieces);
  int64_t max_rows = 0, max_out = 0;
  for (int k = 0; k <= pieces; ++k) cut[k] = i_lo + (n_new * k) / pieces;
  for (int k = p0; k < p1; ++k) {
    int64_t ja, jb;
    needed_knots(T, in->t, n, cut[k], cut[k + 1], ja, jb);
    const int margin = SPLINE_HALO + 2;  // as bms_shard_plan
    r0[k] = std::max<int64_t>(0, ja - margin);
    r1[k] = std::min<int64_t>(n, jb + margin + 1);
    max_rows = std::max(max_rows, r1[k] - r0[k]);
    max_out = std::max(max_out, cut[k + 1] - cut[k]);
  }
  double *d_in[2], *d_out[2];
  if ((rc = dev_buf_t(c, "pipe_in0", (size_t)max_rows * n_modes * 2, &d_in[0]))) return rc;
  if ((rc = dev_buf_t(c, "pipe_in1", (size_t)max_rows * n_modes * 2, &d_in[1]))) return rc;
  if ((rc = dev_buf_t(c, "pipe_out0", (size_t)max_out * n_out * 2, &d_out[0]))) return rc;
  if ((rc = dev_buf_t(c, "pipe_out1", (size_t)max_out * n_out * 2, &d_out[1]))) return rc;
  if (!c->pipe_up) {
    HIP_TRY(c, hipStreamCreateWithFlags(&c->pipe_up, hipStreamNonBlocking));
    HIP_TRY(c, create_download_stream(c));
  }
  std::vector<hipEvent_t> ev_up(pieces), ev_c(pieces), ev_dn(pieces);
  for (int k = p0; k < p1; ++k) {
    ev_up[k] = ScopedTimer::get(c);
    ev_c[k] = ScopedTimer::get(c);
    ev_dn[k] = ScopedTimer::get(c);
  }
  auto give_back = [&]() {
    for (int k = p0; k < p1; ++k) {
      c->event_pool.push_back(ev_up[k]);
      c->event_pool.push_back(ev_c[k]);
      c->event_pool.push_back(ev_dn[k]);
    }
  };
  const char* host_in = (const char*)in->data;
  char* host_out = (char*)data_out;
  // The host waits for a piece's kernels before it issues the download.  SCRI_AMD_PIPE_EVENTS=1 (experiment): the streams wait for
  // each other through events and the host runs ahead, so that the kernels of consecutive pieces follow each other without the
  // host's round trip in between -- measured, three alternating runs: 14.9 / 15.1 / 16.2 ms with the host wait, 14.9 / 15.1 / 13.7
  // with events: no difference, the transfers and not the kernels' gaps set the time.
  const bool host_wait = BMS_PROBE_ENV("SCRI_AMD_PIPE_EVENTS") == nullptr;
  auto upload_piece = [&](int k) -> hipError_t {
    // the buffer was read by the kernels of piece k - 2 (host_wait: the host has waited for them before it gets here)
    const int64_t rows = r1[k] - r0[k];
    if (!host_wait && k >= p0 + 2) {
      const hipError_t ew = hipStreamWaitEvent(c->pipe_up, ev_c[k - 2], 0);
      if (ew != hipSuccess) return ew;
    }
    hipError_t e = in->ld == n_modes
                       ? hipMemcpyAsync(d_in[(k - p0) & 1], host_in + (size_t)r0[k] * in->ld * 16, (size_t)rows * n_modes * 16, hipMemcpyHostToDevice, c->pipe_up)
                       : hipMemcpy2DAsync(d_in[(k - p0) & 1], (size_t)n_modes * 16, host_in + (size_t)r0[k] * in->ld * 16, (size_t)in->ld * 16,
                                          (size_t)n_modes * 16, (size_t)rows, hipMemcpyHostToDevice, c->pipe_up);
    if (e != hipSuccess) return e;
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
    // piece k + 1 travels while piece k is transformed; its buffer was read by the kernels of piece k - 1.  (Piece 0 reads
    // its per-direction tables back with a blocking copy, which waits for every upload under way: piece 1 is sent after it.)
    auto send_next = [&]() -> hipError_t {
      if (k + 1 >= p1) return hipSuccess;
      return upload_piece(k + 1);
    };
    if (k > p0 && (he = send_next()) != hipSuccess) break;
    if ((he = hipStreamWaitEvent(c->stream, ev_up[k], 0)) != hipSuccess) break;
    if (k >= p0 + 2 && (he = hipStreamWaitEvent(c->stream, ev_dn[k - 2], 0)) != hipSuccess) break;  // its output buffer has left
    bms_wm_input piece = *in;
    piece.data = d_in[(k - p0) & 1];
    piece.ld = n_modes;
    piece.mem = BMS_DEVICE;
    const bms_shard sh = {r0[k], r1[k] - r0[k], cut[k], cut[k + 1], 0, 0};
    int64_t got = 0, first = 0;
    rc = transform_modes_impl(c, &piece, tr, &sh, t_out + (cut[k] - i_lo), d_out[(k - p0) & 1], &got, &first, nullptr);
    if (rc) break;
    if (k == p0 && (he = send_next()) != hipSuccess) break;
    if (got != cut[k + 1] - cut[k] || first != cut[k]) {
      rc = fail(c, BMS_ERR_HIP, "pipelined shard [%lld, %lld) produced %lld rows from %lld", (long long)cut[k], (long long)cut[k + 1],
                (long long)got, (long long)first);
      break;
    }
    if ((he = hipEventRecord(ev_c[k], c->stream)) != hipSuccess) break;
    // The host waits for the piece's kernels and then issues the download (the next upload is already on its way).  In the
    // rocprofv3 trace of this loop the uploads run on a DMA engine beside the kernels; the downloads are executed by the runtime
    // as shader copies (__amd_rocclr_copyBuffer) that take turns with the compute kernels.  Storing the results straight into the
    // page-locked array from the analysis kernel (on a side stream, with a small grid) was tried: the stores leave at 42 GB/s
    // instead of 57 and every memory-bound kernel running beside them crawls -- 19.8 ms against 15.7 ms per cfg3 transform.
    if ((he = host_wait ? hipEventSynchronize(ev_c[k]) : hipStreamWaitEvent(c->pipe_down, ev_c[k], 0)) != hipSuccess) break;
    if ((he = hipMemcpyAsync(host_out + (size_t)(cut[k] - i_lo) * n_out * 16, d_out[(k - p0) & 1], (size_t)got * n_out * 16,
                             hipMemcpyDeviceToHost, c->pipe_down)) != hipSuccess)
      break;
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

// One process, several GPUs, ONE call: the `pieces` time shards of the pipelined plan dealt in contiguous runs over the n_ctx contexts
// (one per device; several on one device are allowed), one host thread each, every context running bms_transform_modes_pipelined_part
// on its run.  Every device receives its own rows + halo at upload time: no GPU-to-GPU traffic.  The result depends on `pieces` only.
// On failure the status of the first failing context is returned and its message is copied to ctxs[0] ("context k (device d): ...").
extern "C" int bms_transform_modes_multi(bms_ctx* const* ctxs, int n_ctx, const bms_wm_input* in, const bms_transformation* tr, int pieces,
                                         double* t_out, void* data_out, int64_t* n_times_out) try {
  return run_dealt_over_contexts(ctxs, n_ctx, pieces, n_times_out, [&](bms_ctx* c, int p0, int p1, int64_t* got) {
    return bms_transform_modes_pipelined_part(c, in, tr, pieces < 1 ? 1 : pieces, p0, p1, t_out, data_out, got);
  });
} BMS_CATCH(ctxs && n_ctx > 0 ? ctxs[0] : nullptr)

// WaveformGrid.from_modes on its own (scri/waveform_grid.py:331-613): the field on the distorted grid at the new time slices,
// c16[N'][n_theta * n_phi] in grid order (no column plan), without the analysis back to modes
extern "C" int bms_modes_to_grid(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, double* t_out, void* grid_out,
                                 int64_t* n_times_out) try {
  if (!c) return BMS_ERR_INVALID;
  if (!grid_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  return with_smaller_chunks(c, [&] { return transform_modes_impl(c, in, tr, nullptr, t_out, nullptr, n_times_out, nullptr, grid_out); });
} BMS_CATCH(c)

// Several series under ONE transformation (the extra trailing data dimensions of scri/waveform_grid.py:299-308, 574-594: every
// trailing index is an independent series on the same time axis), in the REFERENCE'S layout: in->data is c16[n_times][in->ld] with
// element (mode, j) of a row at column mode * n_series + j (the trailing index fastest, as numpy stores data[N, n_modes, F]); the psi
// companions likewise.  data_out: c16[n_times][n_out * n_series], grid_out: c16[n_times][n_theta n_phi n_series], same convention; the
// first *n_times_out rows are written.  The block crosses PCIe once as it is; on the device every series becomes a block of
// unit-stride columns (series_to_blocks_kernel), the time axis with its spline tables, the per-direction tables and the window are
// set up once and shared (the mechanism of the pipelined call's pieces), the kernels run per series, and the results are put back
// into the reference's layout (blocks_to_series_kernel) before they leave.  No strided copy on the host.
extern "C" int bms_transform_modes_series(bms_ctx* c, const bms_wm_input* in, int n_series, const bms_transformation* tr, double* t_out,
                                          void* data_out, void* grid_out, int64_t* n_times_out) try {
  if (!c) return BMS_ERR_INVALID;
  if (!in || !tr || !t_out || !n_times_out || (!data_out == !grid_out)) return fail(c, BMS_ERR_INVALID, "NULL argument (exactly one of data_out / grid_out)");
  if (!in->t || !in->data) return fail(c, BMS_ERR_INVALID, "NULL argument");
  if (!valid_mem(in->mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", in->mem);
  if (n_series < 1) return fail(c, BMS_ERR_INVALID, "n_series must be positive");
  if (in->ell_min < 0 || in->ell_max < in->ell_min || in->n_aux < 0 || in->n_aux > 4) return fail(c, BMS_ERR_INVALID, "bad ell range or n_aux");
  if (in->ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", in->ell_max, MAX_ELL);
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t n = in->n_times;
  if (n < 1) return fail(c, BMS_ERR_INVALID, "n_times must be positive");
  const int n_modes = LM_total_size(in->ell_min, in->ell_max);
  if (in->ld < (int64_t)n_series * n_modes) return fail(c, BMS_ERR_INVALID, "ld = %lld is less than n_series * n_modes = %lld", (long long)in->ld, (long long)n_series * n_modes);
  const int n_out = grid_out ? tr->n_theta * tr->n_phi : LM_total_size(std::abs(in->spin_weight), tr->ell_max_out);
  if (n_out <= 0) return fail(c, BMS_ERR_INVALID, "empty output l range");
  hipStream_t S = c->stream;
  int rc;
  // one field of the input (the data or a psi companion): upload as it is, then one block of columns per series
  auto to_blocks = [&](const char* raw_name, const char* blk_name, const void* src, int64_t ld, int nm, const double** blocks) -> int {
    const double* d_raw = (const double*)src;
    if (in->mem == BMS_HOST) {
      double* d = nullptr;
      int r = dev_buf_t(c, raw_name, (size_t)n * ld * 2, &d);
      if (r) return r;
      HIP_TRY(c, hipMemcpyAsync(d, src, (size_t)n * ld * 16, hipMemcpyHostToDevice, S));
      d_raw = d;
    }
    double* d_blk = nullptr;
    int r = dev_buf_t(c, blk_name, (size_t)n * n_series * nm * 2, &d_blk);
    if (r) return r;
    TIMED(c, BMS_TAG_POINTWISE, launch_series_to_blocks(S, d_raw, ld, d_blk, n, nm, n_series));
    *blocks = d_blk;
    return BMS_OK;
  };
  bms_wm_input dev = *in;
  dev.mem = BMS_DEVICE;
  int aux_modes[4] = {0, 0, 0, 0};
  const double* d_blocks = nullptr;
  if ((rc = to_blocks("series_raw", "series_in", in->data, in->ld, n_modes, &d_blocks))) return rc;
  dev.data = d_blocks;
  dev.ld = (int64_t)n_series * n_modes;
  for (int i = 0; i < in->n_aux; ++i) {
    if (in->aux_ell_min[i] < 0 || in->aux_ell_max[i] < in->aux_ell_min[i]) return fail(c, BMS_ERR_INVALID, "bad l range of auxiliary field %d", i);
    aux_modes[i] = LM_total_size(in->aux_ell_min[i], in->aux_ell_max[i]);
    if (in->aux_ld[i] < (int64_t)n_series * aux_modes[i]) return fail(c, BMS_ERR_INVALID, "auxiliary field %d: row stride too small for %d series", i, n_series);
    const char* raw_names[4] = {"series_aux_raw0", "series_aux_raw1", "series_aux_raw2", "series_aux_raw3"};
    const char* blk_names[4] = {"series_aux0", "series_aux1", "series_aux2", "series_aux3"};
    const double* d_aux = nullptr;
    if ((rc = to_blocks(raw_names[i], blk_names[i], in->aux_data[i], in->aux_ld[i], aux_modes[i], &d_aux))) return rc;
    dev.aux_data[i] = d_aux;
    dev.aux_ld[i] = (int64_t)n_series * aux_modes[i];
  }
  double* d_res = nullptr;  // series-major results, c16[n_series][n][n_out]
  if ((rc = dev_buf_t(c, "series_res", (size_t)n_series * n * n_out * 2, &d_res))) return rc;
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
  int64_t n_new = 0;
  rc = BMS_OK;
  for (int j = 0; j < n_series && rc == BMS_OK; ++j) {
    bms_wm_input one = dev;
    one.data = (const double*)dev.data + (size_t)2 * j * n_modes;
    for (int i = 0; i < in->n_aux; ++i) one.aux_data[i] = (const double*)dev.aux_data[i] + (size_t)2 * j * aux_modes[i];
    double* out_j = d_res + (size_t)j * n * n_out * 2;
    int64_t got = 0;
    rc = with_smaller_chunks(c, [&] {
      return transform_modes_impl(c, &one, tr, nullptr, t_out, grid_out ? nullptr : out_j, &got, nullptr, grid_out ? out_j : nullptr);
    });
    if (rc == BMS_OK && j > 0 && got != n_new) rc = fail(c, BMS_ERR_HIP, "series %d produced %lld rows, series 0 %lld", j, (long long)got, (long long)n_new);
    n_new = got;
  }
  hipError_t es = hipSuccess;
  if (rc == BMS_OK && n_new > 0) {
    void* user = grid_out ? grid_out : data_out;
    double* d_final = (double*)user;  // device callers: straight into their buffer
    if (in->mem == BMS_HOST && (rc = dev_buf_t(c, "series_out", (size_t)n_new * n_out * n_series * 2, &d_final)) == BMS_OK) {
    }
    if (rc == BMS_OK) {
      es = launch_blocks_to_series(S, d_res, n, d_final, n_new, n_out, n_series);
      if (es == hipSuccess && in->mem == BMS_HOST)
        es = hipMemcpyAsync(user, d_final, (size_t)n_new * n_out * n_series * 16, hipMemcpyDeviceToHost, S);
    }
  }
  const hipError_t ew = hipStreamSynchronize(S);
  if (c->aux) (void)hipStreamSynchronize(c->aux);
  if (rc) return rc;
  if (es != hipSuccess || ew != hipSuccess) return fail(c, BMS_ERR_HIP, "bms_transform_modes_series: %s", hipGetErrorString(es != hipSuccess ? es : ew));
  *n_times_out = n_new;
  return BMS_OK;
} BMS_CATCH(c)

static int transform_modes_impl(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, const bms_shard* sh, double* t_out,
                                void* data_out, int64_t* n_times_out, int64_t* first_index_out, void* grid_out, bool walk_first) {
  if (!in || !tr || !t_out || !n_times_out || !in->t || !in->data) return fail(c, BMS_ERR_INVALID, "NULL argument");
  if (!valid_mem(in->mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", in->mem);
  if (in->n_aux < 0 || in->n_aux > 4) return fail(c, BMS_ERR_INVALID, "0..4 auxiliary fields, got %d", in->n_aux);
  for (int a = 0; a < in->n_aux; ++a)
    if (!in->aux_data[a]) return fail(c, BMS_ERR_INVALID, "auxiliary field %d is NULL", a);
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t n = in->n_times;
  int64_t t_lo, t_hi;
  time_window(n, sh, t_lo, t_hi);
  bool regular_mesh = true;
  HostTrace trace0(c);
  struct DrainOnExit {  // whatever path leaves this call, nothing enqueued here still reads the caller's buffers
    hipStream_t s;
    bool skip;  // pieces of the pipelined path: its own buffers, drained by the pipeline
    ~DrainOnExit() {
      if (!skip) (void)hipStreamSynchronize(s);
    }
  } drain{c->stream, c->async_pieces};
  // The time axis goes to the device and its spline tables are built BEFORE the host walks it (validate_common: 45 - 60 us per 1e5
  // samples during which the GPU would have nothing of this call yet).  Speculative: what the walk can find -- samples out of
  // order (the call fails; the tables built from them are never used) or a graded axis (the slope form uploads its own) -- is rare.
  double* d_x = nullptr;
  SplineTable* d_tab = nullptr;
  BsplineTable* d_bstab = nullptr;
  BsplineForward* d_bsfwd = nullptr;
  int rc;
  const bool times_ahead = n >= 8 && in->t && !c->async_pieces && !c->opt.on(OPT_NO_BSPLINE) &&
                           (!sh || (sh->data_row0 >= 0 && sh->data_rows >= 0 && sh->data_row0 + sh->data_rows <= n));
  if (times_ahead) {
    const int64_t r0 = sh ? sh->data_row0 : 0, r1 = r0 + (sh ? sh->data_rows : n);
    if ((rc = upload_times_bspline(c, in->t, n, t_lo, t_hi, r0, r1, &d_x, &d_bstab, &d_bsfwd))) return rc;
  }
  // The walk itself is put off as well, to the moment the host would otherwise sit waiting for the per-direction tables: until then
  // the axis is taken to be what it nearly always is (increasing, not graded).  A walk that finds otherwise drains what was queued
  // and either fails the call as it always did or starts it again, walk first.
  const bool walk_later = times_ahead && c->aux && !walk_first && !c->opt.on(OPT_WALK_FIRST);
  int walk_rc = BMS_OK;
  bool walked = false, walk_regular = true;
  if (walk_later) {
    rc = validate_transformation(c, n, in->t, tr, 4);
  } else
    rc = validate_common(c, n, in->t, tr, t_lo, t_hi, &regular_mesh);
  if (rc) return rc;
  trace0.mark("time upload + spline factors (enqueue), checks");
  const int s = in->spin_weight;
  if (in->ell_min < 0 || in->ell_max < in->ell_min) return fail(c, BMS_ERR_INVALID, "bad ell range");
  if (in->ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", in->ell_max, MAX_ELL);
  const int n_modes = LM_total_size(in->ell_min, in->ell_max);
  if (in->ld < n_modes) return fail(c, BMS_ERR_INVALID, "row stride smaller than the number of modes");
  if (tr->ell_max_out < std::abs(s)) return fail(c, BMS_ERR_INVALID, "ell_max_out < |s|");
  if (in->type_term == BMS_TERM_PSI && (in->n_aux < 1 || in->n_aux > 4)) return fail(c, BMS_ERR_INVALID, "BMS_TERM_PSI needs 1..4 auxiliary fields");
  const int ell_min_out = std::abs(s);
  const int n_out = LM_total_size(ell_min_out, tr->ell_max_out);
  const int lst = tr->ell_max_supertranslation;
  const cplx* st = (const cplx*)tr->supertranslation;

  // ---------------------------------------------------------------- per-pixel tables (GPU) and output window (host)
  HostTrace trace(c);
  hipStream_t S = c->stream;
  const bool nontrivial = [&] {
    const double* v = tr->boost_velocity;
    if (v[0] != 0 || v[1] != 0 || v[2] != 0) return true;
    for (int i = 1; i < (lst + 1) * (lst + 1); ++i)
      if (st[i].re != 0 || st[i].im != 0) return true;
    return false;
  }();
  const bool apply_term = nontrivial;
  const bool psi = apply_term && in->type_term == BMS_TERM_PSI;
  std::vector<cplx> coef0;
  cplx cv[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  if (apply_term && (in->type_term == BMS_TERM_H || in->type_term == BMS_TERM_SIGMA)) {
    // h:     2 ethbar_GHP(ethbar_GHP(alpha, 0), -1) = +sqrt((l-1) l (l+1) (l+2)) alpha_lm, evaluated with s = -2
    // sigma: eth_GHP(eth_GHP(alpha, 0), 1) = (1/2) sqrt(l (l+1)) sqrt((l-1)(l+2)) alpha_lm, evaluated with s = +2
    coef0.resize((size_t)(lst + 1) * (lst + 1));
    for (int l = 0; l <= lst; ++l)
      for (int m = -l; m <= l; ++m) {
        const cplx a = st[LM_index(l, m, 0)];
        double f;
        if (in->type_term == BMS_TERM_H)
          f = 2 * ((-std::sqrt((double)l * (l + 1.0))) / std::sqrt(2.0)) * ((l >= 1 ? -std::sqrt((l - 1.0) * (l + 2.0)) : 0.0) / std::sqrt(2.0));
        else
          f = (std::sqrt((double)l * (l + 1.0)) / std::sqrt(2.0)) * ((l >= 1 ? std::sqrt((l - 1.0) * (l + 2.0)) : 0.0) / std::sqrt(2.0));
        if (l < 2) f = 0.0;
        coef0[LM_index(l, m, 0)] = {f * a.re, f * a.im};
      }
  } else if (psi) {
    // eth u'/k = (t - alpha) gamma k eth(v.r)/sqrt2 - eth alpha/sqrt2, exactly as waveform_grid.py:508-523
    coef0.resize((size_t)(lst + 1) * (lst + 1));
    for (int l = 0; l <= lst; ++l)
      for (int m = -l; m <= l; ++m) {
        const cplx a = st[LM_index(l, m, 0)];
        const double f = (1 / std::sqrt(2.0)) * (std::sqrt((double)l * (l + 1.0)) / std::sqrt(2.0));
        coef0[LM_index(l, m, 0)] = {f * a.re, f * a.im};
      }
    const double* v = tr->boost_velocity;
    const double is2 = 1 / std::sqrt(2.0);
    cv[1] = {is2 * v[0] * std::sqrt(2 * M_PI / 3), is2 * v[1] * std::sqrt(2 * M_PI / 3)};
    cv[2] = {is2 * v[2] * std::sqrt(4 * M_PI / 3), 0};
    cv[3] = {-is2 * v[0] * std::sqrt(2 * M_PI / 3), is2 * v[1] * std::sqrt(2 * M_PI / 3)};
  }
  // shard: rows [row0, row0 + rows) of the global data are present
  const int64_t row0 = sh ? sh->data_row0 : 0;
  const int64_t rows_avail = sh ? sh->data_rows : n;
  if (sh && (row0 < 0 || rows_avail < 0 || row0 + rows_avail > n)) return fail(c, BMS_ERR_INVALID, "shard rows outside [0, n_times)");
  if (sh && sh->out_i1 < sh->out_i0) return fail(c, BMS_ERR_INVALID, "shard output range [%lld, %lld) is reversed", (long long)sh->out_i0, (long long)sh->out_i1);
  // Without psi mixing the map modes -> grid values is linear along the columns with time-independent coefficients, so
  // the spline's forward elimination is done on the modes (B-spline form, kernels_bspline.hip) and the grid is passed over
  // once, by the back substitution + evaluation.
  const bool bsg = n >= 8 && regular_mesh && !c->opt.on(OPT_NO_BSPLINE);  // B-spline form (else: the slope form, kernels_spline.hip)
  const bool bs = bsg && !psi;                                                 // ... with the elimination commuted onto the modes
  // (a "shard" that holds every row of every column is the whole series: only its output range is restricted)
  if (!regular_mesh && sh != nullptr && !(sh->data_row0 == 0 && sh->data_rows == n && sh->col_parts <= 1))
    return fail(c, BMS_ERR_UNSUPPORTED,
                "the time steps vary by more than 1e3 within 48 samples: such a series is transformed with exact untiled spline "
                "recurrences, which a time shard cannot provide");
  // Everything that depends on the time axis and the input modes only goes to the main stream first; the per-direction
  // tables, whose window the host has to wait for, are computed beside it on the auxiliary stream.
  FieldPlan F[5];
  F[0].ell_min = in->ell_min;
  F[0].ell_max = in->ell_max;
  F[0].spin = s;
  F[0].ld = in->ld;
  if ((rc = stage_in(c, "in_data", in->data, in->mem, (size_t)rows_avail * in->ld * 16, &F[0].d_data))) return rc;
  // (pieces of a pipelined call: the knot tables depend on the time axis only and are built once, for the whole series --
  // per piece they cost a blocking upload from pageable memory and two kernels that crawl while results leave over PCIe)
  PieceTables* shared = c->async_pieces ? static_cast<PieceTables*>(c->piece_tables) : nullptr;
  if (bsg && shared && shared->times_valid) {
    d_x = shared->d_x, d_bstab = shared->d_bstab, d_bsfwd = shared->d_bsfwd;
  } else if (bsg && shared) {
    rc = upload_times_bspline(c, in->t, n, 0, n, 0, n, &d_x, &d_bstab, &d_bsfwd);
    shared->d_x = d_x, shared->d_bstab = d_bstab, shared->d_bsfwd = d_bsfwd;
    shared->times_valid = rc == BMS_OK;
  } else if (bsg && times_ahead)
    rc = BMS_OK;  // (on their way since the top of the call)
  else if (bsg)
    rc = upload_times_bspline(c, in->t, n, t_lo, t_hi, row0, row0 + rows_avail, &d_x, &d_bstab, &d_bsfwd);
  else
    rc = upload_times(c, in->t, n, t_lo, t_hi, row0, row0 + rows_avail, &d_x, &d_tab);
  if (rc) return rc;
  const long long ld_af = round_up(2LL * (n_modes + 1), 16);  // rows on 128-byte lines
  double* d_Af = nullptr;
  trace.mark("input staging, time upload, spline factors (enqueue)");
  // Without a boost the grid is the equiangular grid seen through the constant frame rotation: rotate the (eliminated) modes
  // once and synthesise ring by ring (kernels_synthesis.hip) instead of multiplying with the dense sYlm matrix.  The grid
  // keeps its natural column order for that.
  // A boost ALONG the polar axis of the rotated grid only moves its rings (separable_rotor_grid): the same route with the tables of
  // the aberrated colatitudes, and the conformal factor's power applied on the way out of the phi stage.
  SynthesisPlan syn;
  const bool no_boost = tr->boost_velocity[0] == 0 && tr->boost_velocity[1] == 0 && tr->boost_velocity[2] == 0;
  std::vector<double> ring_theta;
  bool axis_boost = false;
  // Small shapes stay on the dense route even without a boost: since the product evaluates the spline itself, it competes with separable
  // synthesis PLUS back substitution on the grid, and up to l <= 8 (77 modes x 21 x 21) it wins -- 1e5 steps, supertranslation + frame
  // rotation: l <= 4 0.62 -> 0.54 ms, l <= 6 0.89 -> 0.70, l <= 8 1.20 -> 1.12; from l <= 10 (1.65 vs 1.81) the separable route is ahead
  // (tools/probes/dense_vs_separable_small.py).
  const bool small_dense = no_boost && rows_avail >= 8 && (long long)n_modes * tr->n_theta * tr->n_phi <= 40000 && !c->opt.on(OPT_NO_SMALL_DENSE);
  if (bs && !small_dense && rows_avail >= 2 && !(sh && sh->col_parts > 1) && tr->n_theta >= 3 && !c->opt.on(OPT_NO_SEPARABLE_SYNTHESIS)) {
    if (no_boost) {
      if ((rc = build_synthesis(c, tr->n_theta, tr->n_phi, s, in->ell_min, in->ell_max, syn))) return rc;
    } else if (axis_boost_pays(c, n_modes, tr->n_theta, tr->n_phi) && large_synthesis_route(c, tr->n_theta, tr->n_phi, in->ell_min, in->ell_max) &&
               separable_rotor_grid(c, tr, ring_theta)) {
      if ((rc = build_synthesis(c, tr->n_theta, tr->n_phi, s, in->ell_min, in->ell_max, syn, &ring_theta))) return rc;
      axis_boost = syn.large || syn.nt != 0;
    }
  }
  const bool sep = no_boost ? (syn.nt != 0 || syn.large) : axis_boost;
  // Dense route: the back substitution commutes with the synthesis product as well, so it runs on the modes too and the product's
  // epilogue evaluates the spline (kernels_gemm_eval.hip): the grid of coefficients never reaches HBM.
  const bool gemm_eval = bs && !sep && rows_avail >= 8 && !c->opt.on(OPT_NO_GEMM_EVAL);
  // Separable route without a boost, grids the one-kernel synthesis takes: the same step -- the whole solve on the modes, and the
  // synthesis kernel evaluates the spline from the last four coefficient rows it has produced (kernels_synthesis_eval.hip).  NOT the
  // default (SCRI_AMD_SYNTHESIS_EVAL selects it): built for VERDICT r4 item 1, correct on every axis, and slower than the two kernels
  // it replaces -- 3.5 against 0.65 + 0.96 ms at l <= 16, 1e5 steps (docs/HISTORY.md 4.0 (xxiii), profiles/r05_b_synthesis_eval_*).  A pixel's
  // samples trail its knots by skew_b / dt rows; the kernel stages a window of the time axis per segment, so the SPREAD of the skews
  // over the pixels has to stay within a few hundred rows: bounded here, before the per-direction tables exist, by the supertranslation's
  // coefficients (|Y_lm| <= sqrt((2 l + 1) / 4 pi); the l = 0 part is the time translation and shifts every pixel alike).
  bool syn_eval = false;
  {
    size_t se_lds = 0;
    int se_nph = 0, se_rr = 0, se_xw = 0;
    // Shape rule (round 6): from l_max = SYN_EVAL_MIN_ELL on the fused kernel is ahead of the two it replaces on every time axis
    // (profiles/r06_*_boost_free_routes_by_ell.txt), below it is behind; options SYNTHESIS_EVAL / NO_SYNTHESIS_EVAL force either.
    const bool want_syn_eval = c->opt.on(OPT_SYNTHESIS_EVAL) || (!c->opt.on(OPT_NO_SYNTHESIS_EVAL) && in->ell_max >= SYN_EVAL_MIN_ELL);
    if (sep && no_boost && bs && syn.nt != 0 && rows_avail >= 8 && in->t && want_syn_eval &&
        synthesis_eval_supported(syn.g, syn.nt, &se_lds, &se_nph, &se_rr, &se_xw)) {
      double bound = 0.0;
      for (int l = 1; l <= lst; ++l)
        for (int m = -l; m <= l; ++m) {
          const cplx a = st[LM_index(l, m, 0)];
          bound += std::sqrt(a.re * a.re + a.im * a.im) * std::sqrt((2 * l + 1) / (4 * M_PI));
        }
      const int64_t r0 = row0, r1 = row0 + rows_avail;
      double dx = (in->t[r1 - 1] - in->t[r0]) / (double)(r1 - 1 - r0);
      for (int64_t k = r0; k + 64 < r1; k += 64) dx = std::min(dx, (in->t[k + 64] - in->t[k]) / 64.0);
      syn_eval = dx > 0.0 && 2.0 * bound / dx <= (se_xw >= 1024 ? 384.0 : 160.0);  // (segment + spread + slack within the staged window)
      // the inhomogeneous term of h / sigma is subtracted on the modes (its per-direction values are the synthesis of coef0): it must lie
      // in the band of the data
      for (int l = 0; l <= lst && !coef0.empty(); ++l)
        for (int m = -l; m <= l; ++m) {
          const cplx a = coef0[LM_index(l, m, 0)];
          if ((a.re != 0 || a.im != 0) && (l < in->ell_min || l > in->ell_max)) syn_eval = false;
        }
    }
  }
  std::vector<int> term_col;  // (function scope: the upload below is asynchronous; the call ends with a synchronisation)
  std::vector<double> term_val;
  double* d_Ac = nullptr;
  if (gemm_eval || syn_eval) {
    // both sweeps of the spline solve on the modes: in one pass over memory (a thread keeps its column's tile in registers), or --
    // SCRI_AMD_TWO_SWEEPS, the form the kernel was checked against -- as elimination and back substitution one after the other.
    // (Queued BEFORE the per-direction tables of the auxiliary stream: behind them -- so that their few small workgroups find free
    // SIMDs, which this kernel's 2 x 245 registers per lane do not leave -- the host's wait shrinks from 300 to 250 us, but the solve
    // starts that much later and the product waits for it: 5.56 against 5.41 ms per transform.)
    if ((rc = dev_buf_t(c, "Afull", (size_t)rows_avail * ld_af, &d_Ac))) return rc;
    if (c->opt.on(OPT_TWO_SWEEPS)) {
      if ((rc = dev_buf_t(c, "Afwd", (size_t)rows_avail * ld_af, &d_Af))) return rc;
      TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_bspline_forward_modes(S, F[0].d_data, F[0].ld * 2, n_modes, d_Af, ld_af, row0, rows_avail, n, d_bsfwd,
                                                                    SPLINE_TILE, SPLINE_HALO, 1));
      TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_bspline_backward_modes(S, d_Af, ld_af, n_modes + 1, d_Ac, ld_af, row0, rows_avail, d_bstab, SPLINE_TILE, SPLINE_HALO));
    } else
      TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_bspline_solve_modes(S, F[0].d_data, F[0].ld * 2, n_modes, d_Ac, ld_af, row0, rows_avail, d_bsfwd, d_bstab, 1));
    if (syn_eval && !coef0.empty()) {
      // C - off . 1 on the modes: column (l, m) of the solved modes loses coef0_lm times the solved constant column
      for (int l = in->ell_min; l <= std::min(lst, in->ell_max); ++l)
        for (int m = -l; m <= l; ++m) {
          const cplx a = coef0[LM_index(l, m, 0)];
          if (a.re == 0 && a.im == 0) continue;
          term_col.push_back(LM_index(l, m, in->ell_min));
          term_val.push_back(a.re);
          term_val.push_back(a.im);
        }
      if (!term_col.empty()) {
        void *vc, *vv;
        if ((rc = upload(c, "term_col", term_col.data(), sizeof(int) * term_col.size(), &vc))) return rc;
        if ((rc = upload(c, "term_val", term_val.data(), sizeof(double) * term_val.size(), &vv))) return rc;
        TIMED(c, BMS_TAG_POINTWISE, launch_sub_const_modes(S, d_Ac, ld_af, rows_avail, (int)term_col.size(), (const int*)vc, (const double*)vv, n_modes));
      }
    }
  } else if (bs && rows_avail > 0) {
    if ((rc = dev_buf_t(c, "Afwd", (size_t)rows_avail * ld_af, &d_Af))) return rc;
    TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_bspline_forward_modes(S, F[0].d_data, F[0].ld * 2, n_modes, d_Af, ld_af, row0, rows_avail, n, d_bsfwd,
                                                                  SPLINE_TILE, SPLINE_HALO, 1));
  }
  trace.mark("elimination / solve on the modes (enqueue)");
  if (sep) {
    const double* q = tr->frame_rotation;
    if (!(q[0] == 1.0 && q[1] == 0.0 && q[2] == 0.0 && q[3] == 0.0)) {
      // sYlm(F G) = sum_m' D_{m m'}(F) sYlm'(G): the modes as seen from the rotated frame (the constant column stays)
      const double sp[4] = {q[0], q[3], q[2], q[1]};  // (w + i z, y + i x)
      if ((rc = rotate_impl(c, syn_eval ? d_Ac : d_Af, BMS_DEVICE, rows_avail, ld_af / 2, in->ell_min, in->ell_max, sp, false, false))) return rc;
    }
  }
  // The psi-mixing types (whose elimination stays on the grid) and the slope-form fallback of the others: without a boost every
  // field goes through the two-kernel separable synthesis of its own spin; mixing, offset and scale follow on the grid exactly
  // as they do behind the dense product.
  SynthesisPlan syn_f[5];
  bool sep_fields = false;
  if (!bs && rows_avail >= 1 && !(sh && sh->col_parts > 1) && tr->n_theta >= 3 && !c->opt.on(OPT_NO_SEPARABLE_SYNTHESIS) &&
      (no_boost || (axis_boost_pays(c, n_modes, tr->n_theta, tr->n_phi) && separable_rotor_grid(c, tr, ring_theta)))) {
    sep_fields = true;
    for (int fi = 0; fi < 1 + (psi ? in->n_aux : 0) && sep_fields; ++fi) {
      const int f_spin = fi ? in->aux_spin[fi - 1] : s, f_lo = fi ? in->aux_ell_min[fi - 1] : in->ell_min, f_hi = fi ? in->aux_ell_max[fi - 1] : in->ell_max;
      if (f_lo < 0 || f_hi < f_lo) {
        sep_fields = false;  // (reported below, where the auxiliary fields are checked)
        break;
      }
      if ((rc = build_synthesis(c, tr->n_theta, tr->n_phi, f_spin, f_lo, f_hi, syn_f[fi], no_boost ? nullptr : &ring_theta))) return rc;
      sep_fields = syn_f[fi].large;
    }
  }
  PixelTables T;
  DevPixel DP;
  const int col_plan = (grid_out || sep || sep_fields) ? 0 : column_plan(c, tr, n_out);
  bool B_built = false;
  if (shared && c->piece_tables_valid) {
    // (the pieces of a pipelined call share the per-direction tables of the first one, built in ITS column order: a piece that
    // chose the other synthesis route -- it would have to hold fewer than two rows -- must not read them in a different order)
    if (shared->col_plan != col_plan)
      return fail(c, BMS_ERR_UNSUPPORTED, "a piece of the pipelined call chose another synthesis route than the first one (column plan %d vs %d)",
                  col_plan, shared->col_plan);
    T = shared->T;
    DP = shared->DP;
  } else {
    // dense route of a single field: its synthesis matrix (and the row that carries the offsets) is built on the auxiliary stream
    // right behind the per-direction tables, beside the spline solve on the main stream
    std::function<int(hipStream_t, const DevPixel&, int)> build_B;
    if (bs && !sep && !sep_fields && !psi && c->aux) {
      build_B = [&](hipStream_t PS, const DevPixel& D, int n_cols_) -> int {
        FieldPlan& f = F[0];
        f.K = 2 * LM_total_size(f.ell_min, f.ell_max);
        f.ldb = round_up(2LL * n_cols_, 128);
        const long long rows = round_up(f.K, 16);
        int rc2;
        if ((rc2 = dev_buf_t(c, "Bsyn0", (size_t)rows * f.ldb, &f.d_B))) return rc2;
        HIP_TRY(c, hipMemsetAsync(f.d_B, 0, sizeof(double) * rows * f.ldb, PS));
        TIMED_ON(c, PS, BMS_TAG_SETUP, launch_swsh_matrix_complex(PS, D.rotors, n_cols_, f.spin, f.ell_min, f.ell_max, f.d_B, f.ldb));
        TIMED_ON(c, PS, BMS_TAG_SETUP, launch_negated_row(PS, D.col_off, f.d_B + (size_t)(f.K / 2) * f.ldb, 2 * n_cols_));
        B_built = true;
        return BMS_OK;
      };
    }
    std::function<void()> walk;
    if (walk_later)
      walk = [&] {
        walk_rc = walk_time_axis(c, in->t, t_lo, t_hi, &walk_regular);
        walked = true;
      };
    rc = device_pixel_tables(c, tr, T, psi ? 1 : 0, s, in->conformal_weight, coef0.empty() ? nullptr : &coef0, nullptr, cv, DP, col_plan, c->aux, build_B,
                             walk);
    if (walk_later) {
      if (!walked) walk();  // (device_pixel_tables left before its wait)
      if (walk_rc) return walk_rc;
      if (!walk_regular) {  // graded axis: everything above was planned for a regular one
        HIP_TRY(c, hipStreamSynchronize(c->aux));
        HIP_TRY(c, hipStreamSynchronize(S));
        return transform_modes_impl(c, in, tr, sh, t_out, data_out, n_times_out, first_index_out, grid_out, true);
      }
    }
    if (rc) return rc;
    if (shared) {
      shared->T = T;
      shared->DP = DP;
      shared->col_plan = col_plan;
      c->piece_tables_valid = true;
    }
  }
  trace.mark("pixel tables (GPU, auxiliary stream) + copy back");
  const int n_cols = T.n_pix;
  const bool col_split = sh && sh->col_parts > 1;
  int cA, cB;
  if ((rc = column_range(c, sh, n_cols, cA, cB))) return rc;
  const int n_pix = cB - cA;  // columns this call synthesises and splines
  int64_t i_lo, i_hi;
  output_window(T, in->t, n, i_lo, i_hi);
  // produce outputs with global index in [out_i0, out_i1)
  if (sh) {
    i_lo = std::max(i_lo, sh->out_i0);
    i_hi = std::max(i_lo, std::min(i_hi, sh->out_i1));
  }
  if (first_index_out) *first_index_out = i_lo;
  const int64_t n_new = i_hi - i_lo;
  *n_times_out = n_new;
  if (n_new == 0) {
    HIP_TRY(c, hipStreamSynchronize(S));  // the work enqueued above reads the caller's buffers
    return BMS_OK;
  }
  double *d_rot = DP.rotors, *d_off = DP.col_off + 2 * cA, *d_scale = DP.col_scale + 2 * cA, *d_skewa = DP.skew_a + cA, *d_skewb = DP.skew_b + cA;
  double *d_alpha = DP.alpha + cA, *d_xa = DP.xa + 2 * cA, *d_xb = DP.xb + 2 * cA;
  trace.mark("window (host)");

  const long long P2 = 2LL * n_pix;
  const long long ldg = round_up(P2, 16);
  // synthesis matrices
  const int n_fields = 1 + (psi ? in->n_aux : 0);
  for (int a = 0; a < (psi ? in->n_aux : 0); ++a) {
    FieldPlan& f = F[1 + a];
    f.ell_min = in->aux_ell_min[a];
    f.ell_max = in->aux_ell_max[a];
    f.spin = in->aux_spin[a];
    f.ld = in->aux_ld[a];
    if (f.ell_min < 0 || f.ell_max < f.ell_min || f.ld < LM_total_size(f.ell_min, f.ell_max)) return fail(c, BMS_ERR_INVALID, "bad auxiliary field %d", a);
    char nm[32];
    snprintf(nm, sizeof nm, "in_aux%d", a);
    if ((rc = stage_in(c, nm, in->aux_data[a], in->mem, (size_t)rows_avail * f.ld * 16, &f.d_data))) return rc;
  }
  const long long ldb = round_up(2LL * n_cols, 128);
  for (int fi = 0; fi < n_fields; ++fi) {
    FieldPlan& f = F[fi];
    f.K = 2 * LM_total_size(f.ell_min, f.ell_max);
    f.ldb = ldb;
    const long long rows = round_up(f.K, 16);
    char nm[32];
    snprintf(nm, sizeof nm, "Bsyn%d", fi);
    if (sep) continue;  // (no dense sYlm matrix)
    if (sep_fields && psi && bsg && no_boost) {
      // Without a boost the psi mixing is time independent too (X = xa (t - alpha) - xb with xa = 0) and commutes with the spline's
      // forward elimination, as the offset and scale of the other types do: every field is eliminated as MODES
      // (bspline_forward_modes_kernel), rotated, synthesised as eliminated coefficients and mixed -- the elimination pass over the
      // grid is skipped below.
      const int nmf = f.K / 2;
      const long long ld_e = round_up(2LL * nmf, 16);
      double* d_e;
      snprintf(nm, sizeof nm, "psi_Af%d", fi);
      if ((rc = dev_buf_t(c, nm, (size_t)rows_avail * ld_e, &d_e))) return rc;
      TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_bspline_forward_modes(S, f.d_data, f.ld * 2, nmf, d_e, ld_e, row0, rows_avail, n, d_bsfwd, SPLINE_TILE, SPLINE_HALO, 0));
      f.d_data = d_e;
      f.ld = ld_e / 2;
      const double* q = tr->frame_rotation;
      if (!(q[0] == 1.0 && q[1] == 0.0 && q[2] == 0.0 && q[3] == 0.0)) {
        const double sp[4] = {q[0], q[3], q[2], q[1]};  // (w + i z, y + i x)
        if ((rc = rotate_impl(c, d_e, BMS_DEVICE, rows_avail, f.ld, f.ell_min, f.ell_max, sp, false, false))) return rc;
      }
      continue;
    }
    if (sep_fields) {
      // the field as seen from the rotated frame: rotated in place in the staging copy (host callers), in a copy otherwise
      const double* q = tr->frame_rotation;
      if (!(q[0] == 1.0 && q[1] == 0.0 && q[2] == 0.0 && q[3] == 0.0)) {
        const double sp[4] = {q[0], q[3], q[2], q[1]};  // (w + i z, y + i x)
        double* d_copy = const_cast<double*>(f.d_data);
        if (in->mem == BMS_DEVICE) {
          snprintf(nm, sizeof nm, "rot_in%d", fi);
          if ((rc = dev_buf_t(c, nm, (size_t)rows_avail * f.ld * 2, &d_copy))) return rc;
          HIP_TRY(c, hipMemcpyAsync(d_copy, f.d_data, (size_t)rows_avail * f.ld * 16, hipMemcpyDeviceToDevice, S));
          f.d_data = d_copy;
        }
        if ((rc = rotate_impl(c, d_copy, BMS_DEVICE, rows_avail, f.ld, f.ell_min, f.ell_max, sp, false, false))) return rc;
      }
      continue;
    }
    if (fi == 0 && B_built) continue;  // (built behind the per-direction tables on the auxiliary stream)
    if ((rc = dev_buf_t(c, nm, (size_t)rows * ldb, &f.d_B))) return rc;
    HIP_TRY(c, hipMemsetAsync(f.d_B, 0, sizeof(double) * rows * ldb, S));
    TIMED(c, BMS_TAG_SETUP, launch_swsh_matrix_complex(S, d_rot, n_cols, f.spin, f.ell_min, f.ell_max, f.d_B, ldb));
  }
  const int n_modes_in = F[0].K / 2;
  if (bs && !sep && !B_built)  // row n_modes of B multiplies the eliminated constant series: it carries the per-column offset
    TIMED(c, BMS_TAG_SETUP, launch_negated_row(S, DP.col_off, F[0].d_B + (size_t)n_modes_in * ldb, 2 * n_cols));
  trace.mark("uploads + synthesis matrices");
  AnalysisPlan ana;
  if ((rc = build_analysis(c, "wm", T.n_theta, T.n_phi, s, ell_min_out, tr->ell_max_out, ana))) return rc;
  // rows of the evaluated grid start on a 128-byte line too (7 % off the back substitution) wherever the consumer takes a stride
  const long long ldG = (grid_out || (!col_split && analysis_reads_contiguous_rows(ana))) ? P2 : ldg;
  double* d_At = nullptr;
  long long ld_at = 0;
  if (col_split && n_pix > 0)
    if ((rc = part_analysis_matrix(c, ana, "At", n_cols, DP.col_of_pixel, &d_At, &ld_at))) return rc;
  trace.mark("analysis plan");
  // spline factors

  // output staging
  double* d_out = (double*)data_out;
  if (in->mem == BMS_HOST && !grid_out)
    if ((rc = dev_buf_t(c, "out_data", (size_t)n_new * n_out * 2, &d_out))) return rc;
  if (grid_out && col_split) return fail(c, BMS_ERR_INVALID, "the grid output does not combine with a column partition");

  if (n_pix == 0) {  // more parts than column tiles: this part contributes nothing
    if (in->mem == BMS_HOST)
      std::memset(data_out, 0, (size_t)n_new * n_out * 16);
    else
      HIP_TRY(c, hipMemsetAsync(d_out, 0, (size_t)n_new * n_out * 16, S));
    for (int64_t i = 0; i < n_new; ++i) t_out[i] = (1 / T.gamma) * (in->t[i_lo + i] - T.tt);
    HIP_TRY(c, hipStreamSynchronize(S));
    return BMS_OK;
  }

  // ---------------------------------------------------------------- chunk loop over output samples
  const BsplineSpread spread = skew_spread(T, cA, cB, in->t);
  const int margin = SPLINE_HALO + 2;
  // bytes per output row ~ (Y + R + G [+ Yaux]) * ldg * 8
  const double bytes_per_row = (4.0 + (psi ? 1.0 : 0.0)) * ldg * 8.0;  // Y, R, G, F (+ Yaux)
  // rows the work space limit allows; a chunk shorter than a few spline halos would spend its time re-synthesising them,
  // so below that the limit is reported as too small rather than silently exceeded
  int64_t chunk = (int64_t)((double)c->ws_limit / bytes_per_row - 4.0 * margin);
  if (chunk < 4 * margin && chunk < n_new)
    return fail(c, BMS_ERR_NOMEM, "work space limit of %llu bytes holds fewer than %d rows of the %d-column grids (%.0f bytes each); raise it with bms_ctx_set_workspace_limit",
                (unsigned long long)c->ws_limit, 8 * margin, n_cols, bytes_per_row);
  chunk = std::min<int64_t>(chunk, n_new);
  if (!regular_mesh && chunk < n_new)
    return fail(c, BMS_ERR_UNSUPPORTED, "irregular time axis (steps vary by more than 1e3 within 48 samples): the series does not fit the work space in one piece");
  const int spline_tile = regular_mesh ? SPLINE_TILE : (int)std::min<int64_t>(n + 1, 0x7fffffff);  // one tile: exact recurrences
  for (int64_t c0 = i_lo; c0 < i_hi; c0 += chunk) {
    const int64_t c1 = std::min<int64_t>(c0 + chunk, i_hi);
    int64_t ja, jb;
    needed_knots(T, in->t, n, c0, c1, ja, jb);
    // (irregular time axis: the whole series, so that the single-tile recurrences start and end at the true ends)
    const int64_t g0 = regular_mesh ? std::max<int64_t>(0, ja - margin) : 0, g1 = regular_mesh ? std::min<int64_t>(n, jb + margin + 1) : n;
    const int64_t rows_in = g1 - g0, rows_out = c1 - c0;
    if (g0 < row0 || g1 > row0 + rows_avail)
      return fail(c, BMS_ERR_INVALID,
                  "shard holds rows [%lld, %lld) but outputs [%lld, %lld) need rows [%lld, %lld): halo too small "
                  "(use bms_shard_plan)",
                  (long long)row0, (long long)(row0 + rows_avail), (long long)c0, (long long)c1, (long long)g0, (long long)g1);
    double *d_Y = nullptr, *d_R = nullptr, *d_G, *d_Yaux = nullptr;
    if (!gemm_eval && !syn_eval)
      if ((rc = dev_buf_t(c, "Y", (size_t)rows_in * ldg, &d_Y))) return rc;
    if (!bs)
      if ((rc = dev_buf_t(c, "R", (size_t)rows_in * ldg, &d_R))) return rc;  // eliminated rows (either form)
    if (grid_out && in->mem == BMS_DEVICE)
      d_G = (double*)grid_out + (size_t)(c0 - i_lo) * P2;  // straight into the caller's grid
    else if ((rc = dev_buf_t(c, "G", (size_t)rows_out * ldG, &d_G)))
      return rc;
    if (gemm_eval) {
      SplineEval ev;
      ev.table = d_bstab, ev.x = d_x, ev.skew_a = d_skewa, ev.skew_b = d_skewb, ev.tt = T.tt, ev.g0 = g0, ev.n_knots = n;
      ev.i_lo = c0, ev.i_hi = c1, ev.out = d_G, ev.ldo = ldG;
      ev.search_halfwidth = eval_search_halfwidth(T, cA, cB, in->t, g0, g1);
      ev.inv_dx = (g1 - g0 >= 2 && in->t[g1 - 1] > in->t[g0]) ? (double)(g1 - 1 - g0) / (in->t[g1 - 1] - in->t[g0]) : 0.0;
      ev.side = nullptr, ev.side_ld = ldg;
      if (!c->d_eval_stats) {
        HIP_TRY(c, hipMalloc(&c->d_eval_stats, 16));
        HIP_TRY(c, hipMemsetAsync(c->d_eval_stats, 0, 16, S));
      }
      ev.stats = c->d_eval_stats;
      const int eval_step = c->opt.v[OPT_GEMM_EVAL_STEP] ? (int)c->opt.v[OPT_GEMM_EVAL_STEP] : 64;
      ev.step = eval_step;
      c->eval_tiles += eval_tile_count(rows_in, n_pix, eval_step);
      if (eval_step != 61)
        if ((rc = dev_buf_t(c, "Cside", (size_t)zgemm3m_eval_side_rows(rows_in) * ldg, &ev.side))) return rc;
      TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m_eval(S, d_Ac + (g0 - row0) * ld_af, ld_af, F[0].d_B + 2 * cA, ldb, rows_in, n_pix, n_modes_in + 1,
                                                           d_scale, ev));
    } else if (syn_eval) {
      SplineEval ev;
      ev.table = d_bstab, ev.x = d_x, ev.skew_a = nullptr, ev.skew_b = d_skewb, ev.tt = T.tt, ev.g0 = g0, ev.n_knots = n;
      ev.i_lo = c0, ev.i_hi = c1, ev.out = d_G, ev.ldo = ldG;
      ev.search_halfwidth = 0, ev.inv_dx = 0.0, ev.side = nullptr, ev.side_ld = 0, ev.stats = nullptr;
      double s_min = 0.0, s_max = 0.0;
      if ((int)T.skew_b.size() < cB) return fail(c, BMS_ERR_HIP, "internal: per-direction skews missing on the host");
      s_min = s_max = T.skew_b[cA];
      for (int p = cA; p < cB; ++p) s_min = std::min(s_min, T.skew_b[p]), s_max = std::max(s_max, T.skew_b[p]);
      TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_synthesis_eval(S, d_Ac + (g0 - row0) * ld_af, ld_af, rows_in, syn.g, syn.nt, syn.d_T, syn.d_meta, ev, s_min,
                                                             s_max, c->n_cu));
    } else if (bs) {
      if (sep) {  // (k = 1 without a boost: no column scale)
        if ((rc = run_synthesis(c, syn, d_Af + (g0 - row0) * ld_af, ld_af, rows_in, coef0.empty() ? nullptr : DP.col_off, d_Y, ldg,
                                axis_boost ? DP.col_scale : nullptr)))
          return rc;
      } else
      TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m(S, d_Af + (g0 - row0) * ld_af, ld_af, F[0].d_B + 2 * cA, ldb, d_Y, ldg, rows_in, n_pix,
                                                      n_modes_in + 1, nullptr, d_scale));
      TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_bspline_backward_eval(S, d_Y, ldg, n_pix, g0, rows_in, n, d_x, d_bstab, SPLINE_TILE, SPLINE_HALO, d_x, d_skewa,
                                                                     d_skewb, T.tt, c0, c1, d_G, ldG, &spread));
    } else {
    if (psi)
      if ((rc = dev_buf_t(c, "Yaux", (size_t)rows_in * ldg, &d_Yaux))) return rc;
    // synthesis (+ fused affine map when there is no psi mixing)
    if (sep_fields) {
      if ((rc = run_synthesis(c, syn_f[0], F[0].d_data + (g0 - row0) * F[0].ld * 2, F[0].ld * 2, rows_in, nullptr, d_Y, ldg))) return rc;
      if (!psi) TIMED(c, BMS_TAG_POINTWISE, launch_affine_cols(S, d_Y, ldg, (int)P2, rows_in, d_off, d_scale));
    } else
    TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m(S, F[0].d_data + (g0 - row0) * F[0].ld * 2, F[0].ld * 2, F[0].d_B + 2 * cA, ldb, d_Y, ldg, rows_in, n_pix,
                            F[0].K / 2, psi ? nullptr : d_off, psi ? nullptr : d_scale));
    if (psi) {
      for (int a = 0; a < in->n_aux; ++a) {
        const FieldPlan& f = F[1 + a];
        if (sep_fields) {
          if ((rc = run_synthesis(c, syn_f[1 + a], f.d_data + (g0 - row0) * f.ld * 2, f.ld * 2, rows_in, nullptr, d_Yaux, ldg))) return rc;
        } else
        TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_zgemm3m(S, f.d_data + (g0 - row0) * f.ld * 2, f.ld * 2, f.d_B + 2 * cA, ldb, d_Yaux, ldg, rows_in, n_pix, f.K / 2,
                                nullptr, nullptr));
        TIMED(c, BMS_TAG_POINTWISE, launch_psi_mix(S, d_Y, d_Yaux, ldg, n_pix, rows_in, d_x + g0, d_alpha, d_xa, d_xb, in->aux_coeff[a],
                                  in->aux_power[a]));
      }
      TIMED(c, BMS_TAG_POINTWISE, launch_affine_cols(S, d_Y, ldg, (int)P2, rows_in, d_off, d_scale));
    }
    // spline along time on the shared knots, evaluated on the distorted slices
    if (bsg && sep_fields && psi && no_boost) {  // (eliminated on the modes above: d_Y holds coefficients already)
      TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_bspline_backward_eval(S, d_Y, ldg, n_pix, g0, rows_in, n, d_x, d_bstab, SPLINE_TILE, SPLINE_HALO, d_x, d_skewa,
                                                                     d_skewb, T.tt, c0, c1, d_G, ldG, &spread));
    } else if (bsg) {  // mixing is time dependent: eliminate on the grid, then the coefficient-only back substitution
      TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_bspline_forward_modes(S, d_Y, ldg, n_pix, d_R, ldg, g0, rows_in, n, d_bsfwd, SPLINE_TILE, SPLINE_HALO, 0));
      TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_bspline_backward_eval(S, d_R, ldg, n_pix, g0, rows_in, n, d_x, d_bstab, SPLINE_TILE, SPLINE_HALO, d_x, d_skewa,
                                                                     d_skewb, T.tt, c0, c1, d_G, ldG, &spread));
    } else {
    TIMED(c, BMS_TAG_SPLINE_FORWARD, launch_spline_forward(S, d_Y, d_R, ldg, n_pix, g0, rows_in, n, d_x, d_tab, spline_tile, SPLINE_HALO));
    TIMED(c, BMS_TAG_SPLINE_BACKWARD, launch_spline_backward_eval(S, d_Y, d_R, ldg, n_pix, g0, rows_in, n, d_x, d_tab, spline_tile, SPLINE_HALO,
                                           d_x, d_skewa, d_skewb, T.tt, c0, c1, d_G, ldG));
    }
    }
    // analysis
    if (grid_out) {
      if (in->mem == BMS_HOST)
        HIP_TRY(c, hipMemcpyAsync((double*)grid_out + (size_t)(c0 - i_lo) * P2, d_G, sizeof(double) * (size_t)rows_out * P2, hipMemcpyDeviceToHost, S));
      if (in->mem == BMS_HOST && c1 < i_hi) HIP_TRY(c, hipStreamSynchronize(S));  // the staging buffer is reused by the next chunk
    } else if (col_split) {
      TIMED(c, BMS_TAG_GEMM_ANALYSIS, launch_zgemm3m(S, d_G, ldG, d_At + (size_t)cA * ld_at, ld_at, d_out + (c0 - i_lo) * n_out * 2, 2LL * n_out,
                                                     rows_out, n_out, n_pix, nullptr, nullptr));
    } else if ((rc = run_analysis(c, ana, d_G, rows_out, d_out + (c0 - i_lo) * n_out * 2, 2LL * n_out, DP.col_of_pixel, ldG)))
      return rc;
  }
  trace.mark("chunk loop (enqueue)");
  if (in->mem == BMS_HOST && !grid_out)
    HIP_TRY(c, hipMemcpyAsync(data_out, d_out, (size_t)n_new * n_out * 16, hipMemcpyDeviceToHost, S));
  // the new time axis is host work: done while the GPU runs
  for (int64_t i = 0; i < n_new; ++i) t_out[i] = (1 / T.gamma) * (in->t[i_lo + i] - T.tt);
  // host tables above are stack/vector memory: wait for the uploads (and results) before returning
  if (!c->async_pieces) HIP_TRY(c, hipStreamSynchronize(S));
  trace.mark("final synchronize");
  return BMS_OK;
}
