// Context of the engine: bms_ctx_*, bms_host_*, bms_last_error, bms_version
// (engine.h: the split of the engine by entry family; include/scri_amd.h: the C ABI)
#include "engine.h"

// Host side of the engine -- context: device, streams, work-space slab, route options, page-locked host memory, timing.
//
// Pipeline of one BMS transformation (WaveformModes flavour, scri/waveform_grid.py:331-613 + 274-329):
//   host   : rotor grid R_jk (n_pix quaternions), per-pixel scalars (k, alpha, inhomogeneous term), output
//            time window, theta-quadrature weights                       [O(n_pix) work, no time dependence]
//   GPU    : SWSH synthesis matrix and quadrature matrix (kernels_swsh.hip), spline factor table
//   GPU xN : per chunk of output times:  synthesis GEMM (+ fused affine epilogue)  ->  spline forward
//            -> spline backward + evaluation on the distorted time slices  ->  analysis GEMM
// Nothing in this file falls back to the CPU for the data path; the host only prepares O(n_pix) tables.

static thread_local std::string g_create_error;

int fail(bms_ctx* c, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    g_create_error = buf;
  return code;
}

int exception_status(bms_ctx* c) noexcept {
  int code = BMS_ERR_INTERNAL;
  char buf[512];
  try {
    throw;
  } catch (const std::bad_alloc&) {
    code = BMS_ERR_NOMEM;
    snprintf(buf, sizeof buf, "host allocation failed (std::bad_alloc)");
  } catch (const std::exception& e) {
    snprintf(buf, sizeof buf, "unexpected C++ exception: %s", e.what());
  } catch (...) {
    snprintf(buf, sizeof buf, "unexpected C++ exception");
  }
  try {
    (c ? c->err : g_create_error) = buf;
  } catch (...) {
  }
  return code;
}

// The stream the results of a pipelined call leave on.  The runtime executes device-to-host copies as shader copies
// (__amd_rocclr_copyBuffer) that take turns with the compute kernels on every CU.  SCRI_AMD_DOWN_CUS = n (experiment) confines the
// stream to n CUs spread over the chip (hipExtStreamCreateWithCUMask).  Measured (tools/host_mode_rate.py, cfg3 from and to host
// memory, three alternating runs on one box): 14.8 / 13.9 / 15.0 ms unconfined, 14.9 / 14.9 / 15.2 ms on 8 CUs -- no difference
// beyond the run-to-run spread (a first sweep that read 13.6 ms on 8 CUs against 14.8 was that spread), so the default stays
// unconfined.
hipError_t create_download_stream(bms_ctx* c) {
  const char* e = BMS_PROBE_ENV("SCRI_AMD_DOWN_CUS");
  const int want = e ? atoi(e) : 0;
  if (want > 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) {
      const int n_cu = prop.multiProcessorCount, words = (n_cu + 31) / 32;
      std::vector<uint32_t> mask(words, 0u);
      const int stride = std::max(1, n_cu / want);
      int set = 0;
      for (int i = 0; i < n_cu && set < want; i += stride, ++set) mask[i / 32] |= 1u << (i % 32);
      if (hipExtStreamCreateWithCUMask(&c->pipe_down, (uint32_t)words, mask.data()) == hipSuccess) return hipSuccess;
      (void)hipGetLastError();
    }
  }
  return hipStreamCreateWithFlags(&c->pipe_down, hipStreamNonBlocking);
}

// grow-only device buffer by name
int dev_buf(bms_ctx* c, const char* name, size_t bytes, void** out) {
  DevBuf& b = c->bufs[name];
  if (b.cap < bytes) {
    size_t want = bytes + bytes / 16 + 4096;
    want = (want + 255) & ~(size_t)255;
    if (b.p) {
      // the old block goes back (to its slab or to the runtime): nothing queued may still use it
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if (c->aux) HIP_TRY(c, hipStreamSynchronize(c->aux));
      // (a pipelined call's uploads and downloads run on their own streams and may still use the old block)
      if (c->pipe_up) HIP_TRY(c, hipStreamSynchronize(c->pipe_up));
      if (c->pipe_down) HIP_TRY(c, hipStreamSynchronize(c->pipe_down));
      if (b.slab >= 0)
        c->slabs[b.slab].give(b.slab_off, b.cap);
      else
        HIP_TRY(c, hipFree(b.p));
      b.p = nullptr;
      b.cap = 0;
      b.slab = -1;
    }
    // from a reserved slab if one has room (bms_ctx_reserve): no allocation
    for (int i = (int)c->slabs.size() - 1; i >= 0; --i) {
      size_t off;
      if (c->slabs[i].take(want, &off)) {
        b.p = c->slabs[i].base + off;
        b.cap = want;
        b.slab = i, b.slab_off = off;
        *out = b.p;
        if (c->opt.on(OPT_TRACE))
          fprintf(stderr, "[scri_amd] work space '%s' grows to %.3f GB: from slab %d at %.3f GB\n", name, want / 1073741824.0, i, off / 1073741824.0);
        return BMS_OK;
      }
    }
    const auto t_a = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&b.p, want);
    if (c->opt.on(OPT_TRACE))
      fprintf(stderr, "[scri_amd] work space '%s' grows to %.3f GB: hipMalloc %.1f ms\n", name, want / 1073741824.0,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_a).count());
    if (e != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      c->alloc_failed = true;
      return fail(c, BMS_ERR_NOMEM, "hipMalloc of %zu bytes for work space '%s' failed: %s", want, name,
                  hipGetErrorString(e));
    }
    b.cap = want;
  }
  *out = b.p;
  return BMS_OK;
}

extern "C" int bms_version(void) { return 1; }

// Default cap of the chunked grids: other tenants of the GPU (torch tensors of the caller, further ranks of a dry run, a smaller
// device) shrink it; tables, F arrays and the grow-only named buffers come on top, hence a third and not all of what is free.
static uint64_t default_ws_limit() {
  size_t free_b = 0, total_b = 0;
  uint64_t lim = 96ull << 30;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) lim = std::min<uint64_t>(lim, (uint64_t)free_b / 3);
  (void)hipGetLastError();
  return std::max<uint64_t>(lim, 256ull << 20);
}

extern "C" int bms_ctx_create(int device, bms_ctx** out) try {
  if (!out) return fail(nullptr, BMS_ERR_INVALID, "bms_ctx_create: ctx pointer is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(nullptr, BMS_ERR_NODEVICE,
                "no HIP device available (%s): scri_amd has no CPU fallback and needs an MI355X (gfx950)",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  }
  if (device < 0 || device >= n) return fail(nullptr, BMS_ERR_INVALID, "device %d out of range [0, %d)", device, n);
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail(nullptr, BMS_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, BMS_ERR_NODEVICE, "device %d is %s; this library contains gfx950 (MI355X) code only", device,
                prop.gcnArchName);
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, BMS_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  bms_ctx* c = new bms_ctx;
  c->device = device;
  e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(nullptr, BMS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  c->stream = c->own_stream;
  if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess) c->aux = nullptr;
  c->ws_limit = default_ws_limit();
  c->opt.read_environment();  // the ONLY place the route switches meet the environment: they are this context's defaults from here on
  *out = c;
  return BMS_OK;
} BMS_CATCH(nullptr)

// Route options of one context (env.h lists them; names with or without the SCRI_AMD_ prefix).  Setting one drops the context's cached
// plans (their shape may depend on the route); like every entry point it must not run while another thread uses the same context.
extern "C" int bms_ctx_set_option(bms_ctx* c, const char* name, int64_t value) try {
  if (!c) return BMS_ERR_INVALID;
  const int i = route_option_index(name);
  if (i < 0) return fail(c, BMS_ERR_INVALID, "bms_ctx_set_option: no route option named '%s'", name ? name : "(null)");
  if (i == OPT_GEMM_EVAL_STEP && value != 0 && value != 61 && value != 64)
    return fail(c, BMS_ERR_INVALID, "GEMM_EVAL_STEP is 0 (automatic), 61 or 64; got %lld", (long long)value);
  if (i == OPT_AXIS_BOOST_MIN_WORK)
    c->opt.v[i] = value < 0 ? -1 : value;  // (a count of multiply-adds; 0: the axis-boost route whenever it applies; < 0: the built-in threshold)
  else if (i == OPT_GEMM_EVAL_STEP)
    c->opt.v[i] = value;
  else
    c->opt.v[i] = value != 0;
  c->plans.clear();
  c->syn_plans.clear();
  c->syn_plans_axis.clear();
  c->ring_verdict = -1;
  return BMS_OK;
} BMS_CATCH(c)
extern "C" int bms_ctx_get_option(bms_ctx* c, const char* name, int64_t* value) try {
  if (!c || !value) return BMS_ERR_INVALID;
  const int i = route_option_index(name);
  if (i < 0) return fail(c, BMS_ERR_INVALID, "bms_ctx_get_option: no route option named '%s'", name ? name : "(null)");
  *value = c->opt.v[i];
  return BMS_OK;
} BMS_CATCH(c)

extern "C" void bms_ctx_destroy(bms_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& kv : c->bufs)
    if (kv.second.p && kv.second.slab < 0) (void)hipFree(kv.second.p);
  for (auto& sl : c->slabs) (void)hipFree(sl.base);
  if (c->d_eval_stats) (void)hipFree(c->d_eval_stats);
  for (auto& t : c->timed) {
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  for (auto e : c->event_pool) (void)hipEventDestroy(e);
  if (c->rot_ring_host) (void)hipHostFree(c->rot_ring_host);
  if (c->pix_back_host) (void)hipHostFree(c->pix_back_host);
  if (c->ev_tables) (void)hipEventDestroy(c->ev_tables);
  if (c->ev_aux_done) (void)hipEventDestroy(c->ev_aux_done);
  if (c->rot_ring_dev) (void)hipFree(c->rot_ring_dev);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->aux) (void)hipStreamDestroy(c->aux);
  if (c->pipe_up) (void)hipStreamDestroy(c->pipe_up);
  if (c->pipe_down) (void)hipStreamDestroy(c->pipe_down);
  delete c;
}

extern "C" const char* bms_last_error(const bms_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

extern "C" void* bms_host_alloc(uint64_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
  return p;
}
extern "C" void bms_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
// Page-lock a caller's array in place: uploads from it then run at PCIe rate without the runtime's staging copy.  Costs about
// what one upload of the array costs, so it pays for arrays that are transformed more than once (scri_amd/engine.py does it
// on the second sighting of an array and undoes it when the array is freed).
extern "C" int bms_host_register(void* p, uint64_t bytes) try {
  if (!p || !bytes) return BMS_ERR_INVALID;
  if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return BMS_ERR_HIP;
  }
  return BMS_OK;
} BMS_CATCH(nullptr)
extern "C" int bms_host_unregister(void* p) try {
  if (!p) return BMS_ERR_INVALID;
  if (hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return BMS_ERR_HIP;
  }
  return BMS_OK;
} BMS_CATCH(nullptr)

// The page-locked rotor ring is reused once the stream has passed its slots: before the context moves to another stream the
// old one is drained, so that no slot still waits for its copy on a stream nobody will synchronise any more.
static int switch_stream(bms_ctx* c, hipStream_t s) {
  if (s == c->stream) return BMS_OK;
  if (c->rot_ring_next) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->rot_ring_next = 0;
  }
  c->stream = s;
  return BMS_OK;
}

extern "C" int bms_ctx_set_stream(bms_ctx* c, void* s) try {
  if (!c) return BMS_ERR_INVALID;
  return switch_stream(c, s ? (hipStream_t)s : c->own_stream);
} BMS_CATCH(c)

// The device's default (null) stream has the handle 0, which bms_ctx_set_stream reads as "back to the context's own
// stream"; a caller whose allocations, copies and memsets are queued on the null stream (torch's default stream) names it
// here, so that the engine's kernels are ordered behind them instead of racing them on a non-blocking stream.
extern "C" int bms_ctx_use_default_stream(bms_ctx* c) try {
  if (!c) return BMS_ERR_INVALID;
  return switch_stream(c, nullptr);
} BMS_CATCH(c)

extern "C" int bms_ctx_set_workspace_limit(bms_ctx* c, uint64_t bytes) try {
  if (!c) return BMS_ERR_INVALID;
  c->ws_limit_set = bytes != 0;
  if (!bytes) HIP_TRY(c, hipSetDevice(c->device));  // (the default is sized from THIS context's device)
  c->ws_limit = bytes ? bytes : default_ws_limit();
  return BMS_OK;
} BMS_CATCH(c)

// Device allocations are slow on this platform -- 70 to 120 ms per GB for the tens of GB a full-size call needs (measured inside the first
// device-resident map_to_superrest_frame of a process: 'R' grows to 22.8 GB: 2 657 ms, to 32.1 GB: 2 342 ms), and memory a process has
// merely held before does not come back faster (a throw-away allocation of the whole cap up front changed nothing:
// profiles/r05_a_superrest_reserve_*).  bms_ctx_reserve therefore takes ONE allocation of `bytes` (0: one and a half times the work-space cap)
// that the context's named work-space buffers are carved from afterwards: the first full-size call of the process then allocates
// nothing.  A buffer that outgrows its region gives it back to the slab and takes a larger one (first fit, neighbours coalesced: the
// last buffer grows in place); without room it falls back to an allocation of its own.  Further calls add slabs.
extern "C" int bms_ctx_reserve(bms_ctx* c, uint64_t bytes) try {
  if (!c) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!bytes) {
    // the chunk grids may take the whole cap; tables, staging copies and the F arrays of the separable routes come on top, and a buffer
    // that grows needs its new region while its neighbours still hold theirs (the device-resident map_to_superrest_frame at 1e5 steps,
    // l <= 12: 115 GB in all for a 96 GB cap): half as much again, within four fifths of what is free now
    bytes = c->ws_limit + c->ws_limit / 2;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) bytes = std::min<uint64_t>(bytes, (uint64_t)free_b / 5 * 4);
    (void)hipGetLastError();
  }
  Slab sl;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(c, e == hipErrorOutOfMemory ? BMS_ERR_NOMEM : BMS_ERR_HIP, "bms_ctx_reserve: hipMalloc of %llu bytes failed: %s",
                (unsigned long long)bytes, hipGetErrorString(e));
  }
  sl.base = (char*)p, sl.cap = bytes;
  sl.free[0] = bytes;
  c->slabs.push_back(sl);
  return BMS_OK;
} BMS_CATCH(c)

// Diagnostics of the evaluating product: out[0] = tiles and tile-boundary blocks launched since the last reset, out[1] = those whose
// samples did not fit the window of output abscissae staged in LDS (they search and read the axis in global memory: same results,
// slower), out[2] = per-column marches that started in the window and had to go on from global memory.
extern "C" int bms_ctx_get_eval_stats(bms_ctx* c, int64_t* out /*[3]*/, int reset) try {
  if (!c || !out) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  unsigned long long h[2] = {0, 0};
  if (c->d_eval_stats) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyAsync(h, c->d_eval_stats, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (reset) {
      HIP_TRY(c, hipMemsetAsync(c->d_eval_stats, 0, 16, c->stream));  // (on the stream the kernels count on: a following launch is ordered behind it)
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
  }
  out[0] = (int64_t)c->eval_tiles, out[1] = (int64_t)h[0], out[2] = (int64_t)h[1];
  if (reset) c->eval_tiles = 0;
  return BMS_OK;
} BMS_CATCH(c)

extern "C" int bms_ctx_enable_timing(bms_ctx* c, int on) try {
  if (!c) return BMS_ERR_INVALID;
  c->timing = on != 0;
  return BMS_OK;
} BMS_CATCH(c)

// accumulate finished event pairs into per-tag totals; returns totals since the last reset
extern "C" int bms_ctx_get_timing(bms_ctx* c, double* ms /*[BMS_TAG_COUNT]*/, int64_t* calls /*[BMS_TAG_COUNT]*/, int reset) try {
  if (!c) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (auto& t : c->timed) {
    float f = 0.f;
    if (hipEventElapsedTime(&f, t.a, t.b) == hipSuccess) {
      c->tag_ms[t.tag] += f;
      c->tag_calls[t.tag] += 1;
    }
    c->event_pool.push_back(t.a);
    c->event_pool.push_back(t.b);
  }
  c->timed.clear();
  for (int i = 0; i < BMS_TAG_COUNT; ++i) {
    if (ms) ms[i] = c->tag_ms[i];
    if (calls) calls[i] = c->tag_calls[i];
    if (reset) {
      c->tag_ms[i] = 0;
      c->tag_calls[i] = 0;
    }
  }
  return BMS_OK;
} BMS_CATCH(c)

extern "C" int bms_ctx_synchronize(bms_ctx* c) try {
  if (!c) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BMS_OK;
} BMS_CATCH(c)
