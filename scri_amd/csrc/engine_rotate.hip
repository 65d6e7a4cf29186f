// Rotations of the decomposition basis: bms_rotate_const, bms_rotate_const_D, bms_rotate_series, bms_wigner_D
// (engine.h: the split of the engine by entry family; include/scri_amd.h: the C ABI)
#include "engine.h"

// ====================================================================================================== rotation

constexpr int ROT_RING = 64;
// sync_after = false (internal callers, device data, constant rotor): the call returns with the work enqueued
int rotate_impl(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max, const void* spinors, bool series,
                bool sync_after) {
  if (!c) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_times < 0 || ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad sizes");
  if (ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", ell_max, MAX_ELL);
  const int64_t n_modes = LM_total_size(ell_min, ell_max);
  if (ld < n_modes) return fail(c, BMS_ERR_INVALID, "row stride %lld smaller than %lld modes", (long long)ld, (long long)n_modes);
  if (n_times == 0) return BMS_OK;
  if (!data) return fail(c, BMS_ERR_INVALID, "NULL data");
  // three kernels: tables resident in the LDS, tables staged per l (l <= 33), VALU beyond.  The l range is walked in
  // segments: as many leading l as fit the LDS-resident kernel (l <= 27 and 160 KB: 2..16 of the headline configurations in
  // one launch, 2..19 and 20..24 of an l <= 24 series), the rest through the staged / VALU kernel -- a segment is a column range of the
  // same rows, so each launch gets the pointer of its first mode and the common row stride.
  struct Segment {
    int lo, hi, kind;  // 0 resident, 1 staged MFMA, 2 VALU
    RotResPlan plan;
    size_t lds;
    const double* tab;
  };
  std::vector<Segment> segs;
  int rc = BMS_OK;
  {
    const bool allow_res = !c->opt.on(OPT_ROTATE_VALU) && !c->opt.on(OPT_ROTATE_STAGED) && ld * 256 <= 0x7ffe0000LL;
    int l = ell_min;
    while (l <= ell_max) {
      Segment sg{};
      bool placed = false;
      if (allow_res) {
        for (int hi = std::min(ell_max, 27); hi >= l && !placed; --hi) {
          size_t lds = 0;
          if (!rotate_resident_plan(l, hi, &sg.plan, &lds)) continue;
          bool ok = false;
          unsigned int* d_counter = nullptr;
          if ((rc = ensure_delta_resident(c, l, hi, &ok, &sg.plan, &sg.lds, &sg.tab, &d_counter))) return rc;
          if (!ok) continue;
          sg.lo = l;
          sg.hi = hi;
          sg.kind = 0;
          placed = true;
        }
      }
      if (!placed) {
        sg.lo = l;
        sg.hi = ell_max;
        sg.kind = (rotate_mfma_supported(ell_max) && !c->opt.on(OPT_ROTATE_VALU)) ? 1 : 2;
        if (sg.kind == 2 && rotate_waves_per_block(ell_max) < 1)
          return fail(c, BMS_ERR_UNSUPPORTED, "ell_max=%d too large for the rotation kernels", ell_max);
      }
      segs.push_back(sg);
      l = sg.hi + 1;
    }
  }
  const double* d_delta = nullptr;
  const long long* d_off = nullptr;
  if (segs.back().kind != 0) {
    rc = segs.back().kind == 1 ? ensure_delta_mfma(c, ell_max, &d_delta, &d_off) : ensure_delta(c, ell_max, &d_delta, &d_off);
    if (rc) return rc;
  }
  double* d_data = (double*)data;
  const double* d_rot = (const double*)spinors;
  const size_t data_bytes = ((size_t)(n_times - 1) * ld + n_modes) * 16;  // a strided view ends with its last row's modes
  const size_t rot_bytes = (series ? (size_t)n_times : 1) * 32;
  auto launch_segments = [&](double* base, int64_t rows, int64_t pitch, const double* rot) -> int {
    for (const Segment& sg : segs) {
      double* seg_data = base + 2 * ((long long)sg.lo * sg.lo - (long long)ell_min * ell_min);
      if (sg.kind == 0)
        TIMED(c, BMS_TAG_ROTATE, launch_rotate_modes_resident(c->stream, seg_data, rows, pitch, rot, series ? 4 : 0, sg.tab, sg.plan, sg.lds, nullptr, c->n_cu));
      else if (sg.kind == 1)
        TIMED(c, BMS_TAG_ROTATE, launch_rotate_modes_mfma(c->stream, seg_data, rows, pitch, sg.lo, sg.hi, rot, series ? 4 : 0, d_delta, d_off));
      else
        TIMED(c, BMS_TAG_ROTATE, launch_rotate_modes(c->stream, seg_data, rows, pitch, sg.lo, sg.hi, rot, series ? 4 : 0, d_delta, d_off));
    }
    return BMS_OK;
  };
  // A long series in HOST memory is all transfer (cfg2: 117 MB each way at 57 GB/s against 0.1 ms of kernel), and one call sends it, turns it
  // and brings it back one after the other.  The link is full duplex: blocks of rows go up on one stream, are rotated on the context's
  // stream and come back on a third, ordered by events, two staging buffers -- the rows are independent, so there is no halo; the
  // result is the one-call result to the last bit or two (a row's rounding depends on the launch geometry of the kernel: 5e-16,
  // tools/probes/rot_block_probe.py).  From page-locked memory (bms_host_register / bms_host_alloc) the copies run at the rate
  // of the link and side by side (l <= 16, 1e5 steps: 16.3 -> 9 ms); from pageable memory the runtime stages them and little is gained.
  const size_t row_bytes = (size_t)n_modes * 16;
  int blocks = c->opt.on(OPT_NO_ROTATE_PIPELINE) ? 1 : (int)std::min<size_t>(16, (size_t)n_times * row_bytes / (12u << 20));  // (blocks of >= 12 MB, <= 16 of them: host_rotation_by_blocks.py)
  if (const char* e = BMS_PROBE_ENV("SCRI_AMD_ROTATE_BLOCKS")) blocks = atoi(e);
  if (mem == BMS_HOST && blocks >= 2 && n_times >= 64 * blocks) {
    const int64_t rows_max = (n_times + blocks - 1) / blocks;
    double* d_buf[2];
    if ((rc = dev_buf_t(c, "rot_pipe0", (size_t)rows_max * n_modes * 2, &d_buf[0]))) return rc;
    if ((rc = dev_buf_t(c, "rot_pipe1", (size_t)rows_max * n_modes * 2, &d_buf[1]))) return rc;
    const double* d_r = nullptr;
    {
      double* r = nullptr;
      if ((rc = dev_buf_t(c, "rot_spinors", rot_bytes / 8, &r))) return rc;
      HIP_TRY(c, hipMemcpyAsync(r, spinors, rot_bytes, hipMemcpyHostToDevice, c->stream));
      d_r = r;
    }
    if (!c->pipe_up) {
      HIP_TRY(c, hipStreamCreateWithFlags(&c->pipe_up, hipStreamNonBlocking));
      HIP_TRY(c, create_download_stream(c));
    }
    std::vector<hipEvent_t> ev(3 * (size_t)blocks);
    for (auto& e : ev) e = ScopedTimer::get(c);
    hipError_t he = hipSuccess;
    char* host = (char*)data;
    for (int k = 0; k < blocks && he == hipSuccess && rc == BMS_OK; ++k) {
      const int64_t r0 = (int64_t)n_times * k / blocks, r1 = (int64_t)n_times * (k + 1) / blocks, rows = r1 - r0;
      double* buf = d_buf[k & 1];
      hipEvent_t up = ev[3 * k], done = ev[3 * k + 1], down = ev[3 * k + 2];
      if (k >= 2 && (he = hipStreamWaitEvent(c->pipe_up, ev[3 * (k - 2) + 2], 0)) != hipSuccess) break;  // the buffer's last rows have left
      if ((he = hipMemcpy2DAsync(buf, row_bytes, host + (size_t)r0 * ld * 16, (size_t)ld * 16, row_bytes, (size_t)rows, hipMemcpyHostToDevice,
                                 c->pipe_up)) != hipSuccess)
        break;
      if ((he = hipEventRecord(up, c->pipe_up)) != hipSuccess) break;
      if ((he = hipStreamWaitEvent(c->stream, up, 0)) != hipSuccess) break;
      rc = launch_segments(buf, rows, n_modes, series ? d_r + 4 * r0 : d_r);
      if (rc) break;
      if ((he = hipEventRecord(done, c->stream)) != hipSuccess) break;
      if ((he = hipStreamWaitEvent(c->pipe_down, done, 0)) != hipSuccess) break;
      if ((he = hipMemcpy2DAsync(host + (size_t)r0 * ld * 16, (size_t)ld * 16, buf, row_bytes, row_bytes, (size_t)rows, hipMemcpyDeviceToHost,
                                 c->pipe_down)) != hipSuccess)
        break;
      he = hipEventRecord(down, c->pipe_down);
    }
    (void)hipStreamSynchronize(c->pipe_up);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->pipe_down);
    for (auto e : ev) c->event_pool.push_back(e);
    if (rc) return rc;
    if (he != hipSuccess) return fail(c, BMS_ERR_HIP, "rotation of a host series in blocks: %s", hipGetErrorString(he));
    return BMS_OK;
  }
  if (!series && mem == BMS_DEVICE && !sync_after) {
    if (!c->rot_ring_host) {
      HIP_TRY(c, hipHostMalloc((void**)&c->rot_ring_host, 32 * ROT_RING, hipHostMallocDefault));
      HIP_TRY(c, hipMalloc((void**)&c->rot_ring_dev, 32 * ROT_RING));
    }
    if (c->rot_ring_next == ROT_RING) {  // a lap: the slots are free again once the stream has passed them
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->rot_ring_next = 0;
    }
    const int slot = c->rot_ring_next++;
    std::memcpy(c->rot_ring_host + 4 * slot, spinors, 32);
    HIP_TRY(c, hipMemcpyAsync(c->rot_ring_dev + 4 * slot, c->rot_ring_host + 4 * slot, 32, hipMemcpyHostToDevice, c->stream));
    d_rot = c->rot_ring_dev + 4 * slot;
  } else if (mem == BMS_HOST || !series) {
    double* r = nullptr;
    rc = dev_buf_t(c, "rot_spinors", rot_bytes / 8, &r);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(r, spinors, rot_bytes, hipMemcpyHostToDevice, c->stream));
    d_rot = r;
  }
  if (mem == BMS_HOST) {
    rc = dev_buf_t(c, "rot_data", data_bytes / 8, &d_data);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(d_data, data, data_bytes, hipMemcpyHostToDevice, c->stream));
  }
  if ((rc = launch_segments(d_data, n_times, ld, d_rot))) return rc;
  if (mem == BMS_HOST) {
    HIP_TRY(c, hipMemcpyAsync(data, d_data, data_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else if (!series && sync_after) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the constant rotor was staged from a host stack copy
  }
  return BMS_OK;
}

extern "C" int bms_rotate_const(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max,
                                const double q[4]) try {
  if (!c || !q) return BMS_ERR_INVALID;
  const double sp[4] = {q[0], q[3], q[2], q[1]};  // (w + i z, y + i x)
  return rotate_impl(c, data, mem, n_times, ld, ell_min, ell_max, sp, false);
} BMS_CATCH(c)

extern "C" int bms_rotate_series(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min,
                                 int ell_max, const void* spinors) try {
  if (!c || !spinors) return BMS_ERR_INVALID;
  return rotate_impl(c, data, mem, n_times, ld, ell_min, ell_max, spinors, true);
} BMS_CATCH(c)

// The reference's numba kernel takes the packed Wigner matrices it is handed, not a rotor (scri/rotations.py:346-367:
// `_rotate_decomposition_basis_by_constant(data, ell_min, ell_max, D, tmp)` with D from sf._Wigner_D_matrices, :327):
//     data[t, l, m] <- sum_m' data[t, l, m'] D^l[m', m],   D block l row-major (m', m) at _linear_matrix_offset(l, ell_min).
// One complex GEMM per l on the synthesis kernel ([N x (2l+1)] . [(2l+1) x (2l+1)], operands zero padded to its 8 x 64
// panels), out of place into a work buffer, then copied over the input.  This is the seam a binding replaces the numba
// kernel at; callers that have the rotor use bms_rotate_const, which never forms D.
extern "C" int bms_rotate_const_D(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max,
                                  const void* D_host) try {
  if (!c || !data || !D_host) return BMS_ERR_INVALID;
  if (!valid_mem(mem)) return fail(c, BMS_ERR_INVALID, "mem is BMS_HOST or BMS_DEVICE, got %d", mem);
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_times < 0 || ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad sizes");
  if (ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", ell_max, MAX_ELL);
  const int64_t n_modes = LM_total_size(ell_min, ell_max);
  if (ld < n_modes) return fail(c, BMS_ERR_INVALID, "row stride %lld smaller than %lld modes", (long long)ld, (long long)n_modes);
  if (n_times == 0) return BMS_OK;
  // padded B operands, one after the other
  auto round_up = [](long long a, long long b) { return (a + b - 1) / b * b; };
  std::vector<size_t> boff(ell_max + 2, 0);
  for (int l = ell_min; l <= ell_max; ++l)
    boff[l + 1] = boff[l] + (size_t)round_up(2 * l + 1, 8) * (size_t)round_up(2 * l + 1, 64) * 2;
  std::vector<double> B(boff[ell_max + 1], 0.0);
  const double* D = (const double*)D_host;
  for (int l = ell_min; l <= ell_max; ++l) {
    const int n = 2 * l + 1;
    const size_t pitch = (size_t)round_up(n, 64) * 2;
    const long long off = linear_matrix_offset(l, ell_min);
    for (int r = 0; r < n; ++r)
      for (int q = 0; q < n; ++q) {
        B[boff[l] + r * pitch + 2 * q] = D[2 * (off + (long long)r * n + q)];
        B[boff[l] + r * pitch + 2 * q + 1] = D[2 * (off + (long long)r * n + q) + 1];
      }
  }
  void* vp;
  int rc = upload(c, "rotD_B", B.data(), 8 * B.size(), &vp);
  if (rc) return rc;
  const double* d_B = (const double*)vp;
  const size_t data_bytes = ((size_t)(n_times - 1) * ld + n_modes) * 16;  // a strided view ends with its last row's modes
  double* d_data = (double*)data;
  // a long series in host memory: blocks of rows on three streams, as rotate_impl does (the product is out of place anyway: a block comes
  // up into one buffer, its products go into a second one, and that one goes home -- no copy on the device)
  const size_t row_bytes = (size_t)n_modes * 16;
  const int blocks = c->opt.on(OPT_NO_ROTATE_PIPELINE) ? 1 : (int)std::min<size_t>(16, (size_t)n_times * row_bytes / (12u << 20));
  if (mem == BMS_HOST && blocks >= 2 && n_times >= 64 * blocks) {
    const int64_t rows_max = (n_times + blocks - 1) / blocks;
    double *d_in[2], *d_out[2];
    if ((rc = dev_buf_t(c, "rot_pipe0", (size_t)rows_max * n_modes * 2, &d_in[0]))) return rc;
    if ((rc = dev_buf_t(c, "rot_pipe1", (size_t)rows_max * n_modes * 2, &d_in[1]))) return rc;
    if ((rc = dev_buf_t(c, "rotD_pipe0", (size_t)rows_max * n_modes * 2, &d_out[0]))) return rc;
    if ((rc = dev_buf_t(c, "rotD_pipe1", (size_t)rows_max * n_modes * 2, &d_out[1]))) return rc;
    if (!c->pipe_up) {
      HIP_TRY(c, hipStreamCreateWithFlags(&c->pipe_up, hipStreamNonBlocking));
      HIP_TRY(c, create_download_stream(c));
    }
    std::vector<hipEvent_t> ev(3 * (size_t)blocks);
    for (auto& e : ev) e = ScopedTimer::get(c);
    auto products = [&](const double* in, double* out, int64_t rows) -> int {
      for (int l = ell_min; l <= ell_max; ++l) {
        const int n = 2 * l + 1;
        const long long col = (long long)l * l - (long long)ell_min * ell_min;
        TIMED(c, BMS_TAG_ROTATE, launch_zgemm3m(c->stream, in + 2 * col, 2 * n_modes, d_B + boff[l], round_up(n, 64) * 2, out + 2 * col, 2 * n_modes,
                                                rows, n, n, nullptr, nullptr));
      }
      return BMS_OK;
    };
    hipError_t he = hipSuccess;
    char* host = (char*)data;
    for (int k = 0; k < blocks && he == hipSuccess && rc == BMS_OK; ++k) {
      const int64_t r0 = (int64_t)n_times * k / blocks, r1 = (int64_t)n_times * (k + 1) / blocks, rows = r1 - r0;
      hipEvent_t up = ev[3 * k], done = ev[3 * k + 1], down = ev[3 * k + 2];
      if (k >= 2 && (he = hipStreamWaitEvent(c->pipe_up, ev[3 * (k - 2) + 1], 0)) != hipSuccess) break;  // the input buffer has been read
      if ((he = hipMemcpy2DAsync(d_in[k & 1], row_bytes, host + (size_t)r0 * ld * 16, (size_t)ld * 16, row_bytes, (size_t)rows, hipMemcpyHostToDevice,
                                 c->pipe_up)) != hipSuccess)
        break;
      if ((he = hipEventRecord(up, c->pipe_up)) != hipSuccess) break;
      if ((he = hipStreamWaitEvent(c->stream, up, 0)) != hipSuccess) break;
      if (k >= 2 && (he = hipStreamWaitEvent(c->stream, ev[3 * (k - 2) + 2], 0)) != hipSuccess) break;  // the output buffer has left
      if ((rc = products(d_in[k & 1], d_out[k & 1], rows))) break;
      if ((he = hipEventRecord(done, c->stream)) != hipSuccess) break;
      if ((he = hipStreamWaitEvent(c->pipe_down, done, 0)) != hipSuccess) break;
      if ((he = hipMemcpy2DAsync(host + (size_t)r0 * ld * 16, (size_t)ld * 16, d_out[k & 1], row_bytes, row_bytes, (size_t)rows, hipMemcpyDeviceToHost,
                                 c->pipe_down)) != hipSuccess)
        break;
      he = hipEventRecord(down, c->pipe_down);
    }
    (void)hipStreamSynchronize(c->pipe_up);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->pipe_down);
    for (auto e : ev) c->event_pool.push_back(e);
    if (rc) return rc;
    if (he != hipSuccess) return fail(c, BMS_ERR_HIP, "rotation of a host series in blocks: %s", hipGetErrorString(he));
    return BMS_OK;
  }
  if (mem == BMS_HOST) {
    if ((rc = dev_buf_t(c, "rot_data", data_bytes / 8, &d_data))) return rc;
    HIP_TRY(c, hipMemcpyAsync(d_data, data, data_bytes, hipMemcpyHostToDevice, c->stream));
  }
  double* d_tmp;
  if ((rc = dev_buf_t(c, "rotD_out", (size_t)n_times * n_modes * 2, &d_tmp))) return rc;
  for (int l = ell_min; l <= ell_max; ++l) {
    const int n = 2 * l + 1;
    const long long col = (long long)l * l - (long long)ell_min * ell_min;
    TIMED(c, BMS_TAG_ROTATE, launch_zgemm3m(c->stream, d_data + 2 * col, 2 * ld, d_B + boff[l], round_up(n, 64) * 2,
                                            d_tmp + 2 * col, 2 * n_modes, n_times, n, n, nullptr, nullptr));
  }
  HIP_TRY(c, hipMemcpy2DAsync(d_data, (size_t)ld * 16, d_tmp, (size_t)n_modes * 16, (size_t)n_modes * 16, (size_t)n_times,
                              hipMemcpyDeviceToDevice, c->stream));
  if (mem == BMS_HOST) HIP_TRY(c, hipMemcpyAsync(data, d_data, data_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // B was staged from a host vector
  return BMS_OK;
} BMS_CATCH(c)

// D matrices through the rotation kernel itself: rotate the identity blocks (row (l, m') = delta_{m'})
extern "C" int bms_wigner_D(bms_ctx* c, const double q[4], int ell_min, int ell_max, void* D_host) try {
  if (!c || !q || !D_host) return BMS_ERR_INVALID;
  if (ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad ell range");
  if (ell_max > MAX_ELL) return fail(c, BMS_ERR_UNSUPPORTED, "ell_max = %d is beyond %d", ell_max, MAX_ELL);
  const int n_modes = LM_total_size(ell_min, ell_max);
  const int n_rows = 2 * ell_max + 1;  // row r: in every l block, unit vector at m' = r - l_max (if |m'| <= l)
  std::vector<double> buf((size_t)n_rows * n_modes * 2, 0.0);
  for (int r = 0; r < n_rows; ++r) {
    const int mp = r - ell_max;
    for (int l = std::max(ell_min, std::abs(mp)); l <= ell_max; ++l) buf[((size_t)r * n_modes + LM_index(l, mp, ell_min)) * 2] = 1.0;
  }
  int rc = bms_rotate_const(c, buf.data(), BMS_HOST, n_rows, n_modes, ell_min, ell_max, q);
  if (rc) return rc;
  double* D = (double*)D_host;
  for (int l = ell_min; l <= ell_max; ++l) {
    const long long off = linear_matrix_offset(l, ell_min);
    const int n = 2 * l + 1;
    for (int mp = -l; mp <= l; ++mp)
      for (int m = -l; m <= l; ++m) {
        const size_t src = ((size_t)(mp + ell_max) * n_modes + LM_index(l, m, ell_min)) * 2;
        const size_t dst = (size_t)(off + (long long)(mp + l) * n + (m + l)) * 2;
        D[dst] = buf[src];
        D[dst + 1] = buf[src + 1];
      }
  }
  return BMS_OK;
} BMS_CATCH(c)
