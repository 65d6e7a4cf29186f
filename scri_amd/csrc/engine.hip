// Host side of the engine: context, set-up tables, chunked pipeline, C ABI (include/scri_amd.h).
//
// Pipeline of one BMS transformation (WaveformModes flavour, scri/waveform_grid.py:331-613 + 274-329):
//   host   : rotor grid R_jk (n_pix quaternions), per-pixel scalars (k, alpha, inhomogeneous term), output
//            time window, theta-quadrature weights                       [O(n_pix) work, no time dependence]
//   GPU    : SWSH synthesis matrix and quadrature matrix (kernels_swsh.hip), spline factor table
//   GPU xN : per chunk of output times:  synthesis GEMM (+ fused affine epilogue)  ->  spline forward
//            -> spline backward + evaluation on the distorted time slices  ->  analysis GEMM
// Nothing in this file falls back to the CPU for the data path; the host only prepares O(n_pix) tables.
#include <algorithm>
#include <array>
#include <cmath>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/scri_amd.h"
#include "kernels.h"
#include "pixel_math.h"
#include "wigner.h"

using namespace bms;

// ====================================================================================================== context

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int slab = -1;  // >= 0: carved from that reserved slab (bms_ctx_reserve) at offset `slab_off`, not an allocation of its own
  size_t slab_off = 0;
};
struct Slab {  // one device allocation that the named work-space buffers are carved from (first fit, free regions coalesced)
  char* base = nullptr;
  size_t cap = 0;
  std::map<size_t, size_t> free;  // offset -> length
  bool take(size_t want, size_t* off) {
    for (auto it = free.begin(); it != free.end(); ++it)
      if (it->second >= want) {
        *off = it->first;
        const size_t rest = it->second - want;
        free.erase(it);
        if (rest) free[*off + want] = rest;
        return true;
      }
    return false;
  }
  void give(size_t off, size_t len) {
    auto nx = free.lower_bound(off);
    if (nx != free.end() && off + len == nx->first) {
      len += nx->second;
      nx = free.erase(nx);
    }
    if (nx != free.begin()) {
      auto pv = std::prev(nx);
      if (pv->first + pv->second == off) {
        pv->second += len;
        return;
      }
    }
    free[off] = len;
  }
};

// ---------------------------------------------------------------------------------------------- analysis plan
// Separable analysis (kernels_analysis.hip) when n_theta <= MAX_THETA_SEPARABLE, dense quadrature GEMM otherwise.
struct AnalysisPlan {
  bool separable = true;
  bool fused = false;  // single-kernel analysis (kernels_analysis.hip, analysis_fused_kernel)
  bool large = false;  // folded phi-DFT + MFMA theta quadrature for grids too large for the fused kernel
  int ell_min_out = 0;
  int spin = 0;
  double* d_dcs = nullptr;
  int n_theta = 0, n_phi = 0, n_pix = 0, n_out = 0, L = 0, nm = 0;
  // separable
  double* d_dft = nullptr;
  long long ld_dft = 0;
  double* d_T = nullptr;
  int* d_mindex = nullptr;
  // dense
  double* d_W = nullptr;
  long long ldw = 0;
};

// separable synthesis of boost-free transformations (kernels_synthesis.hip)
struct SynthesisPlan {
  SynGeom g;
  int nt = 0;          // != 0: the one-kernel form takes the shape (synthesis_split_kernel)
  bool large = false;  // the two-kernel form does (kernels_synthesis_large.hip)
  int n_theta = 0, n_phi = 0, ell_min = 0, ell_max = 0;
  size_t lds = 0;
  double* d_T = nullptr;  // [n_modes][n_theta] sLambda_lm(theta_j)
  int* d_meta = nullptr;
};

struct bms_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t aux = nullptr;  // set-up kernels whose results the host waits for run here, beside the main stream's work
  std::string err;
  RouteOptions opt;  // route switches of THIS context: defaults from the environment at bms_ctx_create, then bms_ctx_set_option (env.h)
  // cap on the grids of one chunk of the time axis: min(96 GB, a third of the memory that was free when the context was created)
  // unless the caller sets one (one GPU's cfg5 rows -- 25 000 steps, six 99 x 99 grids, 76 GB -- are ONE chunk on an otherwise empty
  // MI355X); a call that still runs out of memory halves it and tries again (with_smaller_chunks)
  uint64_t ws_limit = 96ull << 30;
  bool ws_limit_set = false;  // by the caller: then it is kept as given
  uint64_t sticky_limit = 0;  // the reduced cap a call of this context last succeeded with after allocation failures (with_smaller_chunks) ...
  int sticky_left = 0;        // ... and for how many more calls it is tried first
  // evaluating product (kernels_gemm_eval.hip): how often its samples left the window of abscissae a tile stages in LDS
  unsigned long long* d_eval_stats = nullptr;  // device: [0] tiles / boundary blocks off the LDS path, [1] marches continued from global memory
  uint64_t eval_tiles = 0;                     // tiles + boundary blocks launched since the last reset
  bool alloc_failed = false;  // a device allocation of the running call failed (as opposed to a cap that is too small by plan)
  std::map<std::string, DevBuf> bufs;  // grow-only named work space
  std::vector<Slab> slabs;             // reserved by bms_ctx_reserve; the newest one with room serves the named buffers
  int delta_lmax = -1;                 // Delta tables cached up to this l
  int delta_mfma_lmax = -1;            // ... in the MFMA B-image packing
  hipStream_t pipe_up = nullptr, pipe_down = nullptr;  // bms_transform_modes_pipelined: uploads and downloads beside the kernels
  // set by bms_transform_modes_pipelined around its per-piece calls: the pieces share one transformation, so the
  // per-direction tables are computed (and read back) once, and a piece returns without waiting for its kernels
  bool async_pieces = false;
  bool piece_tables_valid = false;
  void* piece_tables = nullptr;  // PieceTables*
  // constant rotors of internal rotations travel through a page-locked ring (a truly asynchronous copy: no stream
  // synchronisation to protect a stack copy); the ring is drained once per lap
  double* rot_ring_host = nullptr;
  // per-direction scalars on their way back to the host (device_pixel_tables): page-locked, so that the copy is asynchronous and what is
  // queued behind it starts without the host; two events order the host and the main stream behind the auxiliary one
  double* pix_back_host = nullptr;
  size_t pix_back_cap = 0;
  hipEvent_t ev_tables = nullptr, ev_aux_done = nullptr;
  double* rot_ring_dev = nullptr;
  int rot_ring_next = 0;
  std::map<std::pair<int, int>, RotResPlan> rot_res_plans;  // LDS-resident table images built so far, by (ell_min, ell_max)
  int n_cu = 0;
  // analysis tables depend on the grid, the spin and the l range only: kept per tag until a call asks for other ones
  std::map<std::string, std::pair<std::array<int, 6>, AnalysisPlan>> plans;
  std::map<std::array<int, 5>, SynthesisPlan> syn_plans;  // by (n_theta, n_phi, spin, ell_min, ell_max)
  // the last answer of separable_rotor_grid (frame rotation, boost, grid): repeated transformations skip the walk over the rotors
  double ring_key[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int ring_verdict = -1;  // -1: nothing kept
  std::vector<double> ring_thetas;
  // the same behind a boost along the grid's axis: the tables belong to the ring colatitudes of the last such transformation
  std::map<std::array<int, 5>, std::pair<std::vector<double>, SynthesisPlan>> syn_plans_axis;
  // optional per-kernel timing with HIP events on the context's stream (bms_ctx_enable_timing)
  bool timing = false;
  struct Timed {
    int tag;
    hipEvent_t a, b;
  };
  std::vector<Timed> timed;
  std::vector<hipEvent_t> event_pool;
  double tag_ms[BMS_TAG_COUNT] = {0};
  long long tag_calls[BMS_TAG_COUNT] = {0};
};

static inline void note_alloc_failure(bms_ctx* c) {
  if (c) c->alloc_failed = true;
}

struct ScopedTimer {  // brackets one kernel launch with two events when timing is enabled
  bms_ctx* c;
  int tag;
  hipStream_t stream;  // the stream the bracketed kernel is launched on (events recorded elsewhere would bracket unrelated work)
  hipEvent_t a = nullptr, b = nullptr;
  static hipEvent_t get(bms_ctx* c) {
    if (!c->event_pool.empty()) {
      hipEvent_t e = c->event_pool.back();
      c->event_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  ScopedTimer(bms_ctx* c_, int tag_, hipStream_t stream_ = nullptr) : c(c_), tag(tag_), stream(stream_ ? stream_ : c_->stream) {
    if (c->timing) {
      a = get(c);
      b = get(c);
      (void)hipEventRecord(a, stream);
    }
  }
  ~ScopedTimer() {
    if (c->timing) {
      (void)hipEventRecord(b, stream);
      c->timed.push_back({tag, a, b});
      // a context that never asks for its timings must not collect events for ever: pairs that have completed are folded
      // into the totals once a few thousand are pending (no synchronisation: unfinished pairs stay)
      if (c->timed.size() > 4096) {
        size_t keep = 0;
        for (auto& t : c->timed) {
          float f = 0.f;
          if (hipEventQuery(t.b) == hipSuccess && hipEventElapsedTime(&f, t.a, t.b) == hipSuccess) {
            c->tag_ms[t.tag] += f;
            c->tag_calls[t.tag] += 1;
            c->event_pool.push_back(t.a);
            c->event_pool.push_back(t.b);
          } else {
            c->timed[keep++] = t;
          }
        }
        c->timed.resize(keep);
        (void)hipGetLastError();
      }
    }
  }
};
#define TIMED(ctx, tag, expr)     \
  do {                            \
    ScopedTimer st__(ctx, tag);   \
    HIP_TRY(ctx, expr);           \
  } while (0)
// the same for a kernel launched on another stream than the context's main one
#define TIMED_ON(ctx, strm, tag, expr)  \
  do {                                  \
    ScopedTimer st__(ctx, tag, strm);   \
    HIP_TRY(ctx, expr);                 \
  } while (0)

static thread_local std::string g_create_error;

// host-side phase timing of a call, printed when SCRI_AMD_TRACE is set (debugging aid)
struct HostTrace {
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit HostTrace(const bms_ctx* c) : on(c && c->opt.on(OPT_TRACE)), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[scri_amd] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
    t0 = t1;
  }
};

static int fail(bms_ctx* c, int code, const char* fmt, ...) {
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

static inline void note_alloc_failure(bms_ctx* c);
#define HIP_TRY(ctx, expr)                                                                                   \
  do {                                                                                                       \
    hipError_t e__ = (expr);                                                                                 \
    if (e__ != hipSuccess) {                                                                                 \
      if (e__ == hipErrorOutOfMemory) note_alloc_failure(ctx);                                               \
      return fail(ctx, e__ == hipErrorOutOfMemory ? BMS_ERR_NOMEM : BMS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                  hipGetErrorString(e__), __FILE__, __LINE__);                                               \
    }                                                                                                        \
  } while (0)

// The stream the results of a pipelined call leave on.  The runtime executes device-to-host copies as shader copies
// (__amd_rocclr_copyBuffer) that take turns with the compute kernels on every CU.  SCRI_AMD_DOWN_CUS = n (experiment) confines the
// stream to n CUs spread over the chip (hipExtStreamCreateWithCUMask).  Measured (tools/host_mode_rate.py, cfg3 from and to host
// memory, three alternating runs on one box): 14.8 / 13.9 / 15.0 ms unconfined, 14.9 / 14.9 / 15.2 ms on 8 CUs -- no difference
// beyond the run-to-run spread (a first sweep that read 13.6 ms on 8 CUs against 14.8 was that spread), so the default stays
// unconfined.
static hipError_t create_download_stream(bms_ctx* c) {
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
static int dev_buf(bms_ctx* c, const char* name, size_t bytes, void** out) {
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
template <class T>
static int dev_buf_t(bms_ctx* c, const char* name, size_t count, T** out) {
  void* p = nullptr;
  int rc = dev_buf(c, name, count * sizeof(T), &p);
  *out = static_cast<T*>(p);
  return rc;
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

// Runs `call` again with half the work space cap while it fails because a device ALLOCATION failed (the buffer that failed was
// released before the attempt, so a smaller chunk finds room).  Not retried: a cap the caller set, and the planning error "the cap
// holds fewer than N rows" -- halving only makes that one worse.  The cap itself is restored after the call; the reduced value that
// worked is only REMEMBERED for a bounded number of calls (below), so one transient shortage (a temporary tensor of the caller) does
// not leave every later call of the context with chunks up to 32x smaller.  If every attempt fails the FIRST message is reported.
template <class F>
static int with_smaller_chunks(bms_ctx* c, F call) {
  c->alloc_failed = false;
  const uint64_t tiles0 = c->eval_tiles;
  // A context that had to halve its cap keeps the reduced one for the next calls (sticky_left): under STEADY memory pressure (a
  // co-resident tensor of the caller) every call would otherwise free its grown buffers, fail the same multi-GB allocation and
  // re-allocate smaller ones -- seconds per call at 70-120 ms per GB.  The full cap is tried again after 16 calls, or as soon as
  // the device reports room for it.
  const uint64_t full = c->ws_limit;
  if (c->sticky_left > 0 && c->sticky_limit && !c->ws_limit_set) {
    size_t free_b = 0, total_b = 0;
    const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)free_b > full + full / 2;
    (void)hipGetLastError();
    if (room || --c->sticky_left == 0)
      c->sticky_limit = 0;
    else
      c->ws_limit = std::min(full, c->sticky_limit);
  }
  int rc = call();
  if (rc != BMS_ERR_NOMEM || !c->alloc_failed || c->ws_limit_set) {
    c->ws_limit = full;
    return rc;
  }
  const std::string first = c->err;
  for (int attempt = 0; rc == BMS_ERR_NOMEM && c->alloc_failed && attempt < 5 && c->ws_limit > (512ull << 20); ++attempt) {
    c->ws_limit /= 2;
    c->alloc_failed = false;
    c->eval_tiles = tiles0;  // (diagnostic counter: the failed attempt's launches, if any, are not counted twice)
    rc = call();
  }
  if (rc == BMS_OK) c->sticky_limit = c->ws_limit, c->sticky_left = 16;
  c->ws_limit = full;
  if (rc == BMS_ERR_NOMEM) c->err = first;
  return rc;
}

extern "C" int bms_ctx_create(int device, bms_ctx** out) {
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
}

// Route options of one context (env.h lists them; names with or without the SCRI_AMD_ prefix).  Setting one drops the context's cached
// plans (their shape may depend on the route); like every entry point it must not run while another thread uses the same context.
extern "C" int bms_ctx_set_option(bms_ctx* c, const char* name, int64_t value) {
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
}
extern "C" int bms_ctx_get_option(bms_ctx* c, const char* name, int64_t* value) {
  if (!c || !value) return BMS_ERR_INVALID;
  const int i = route_option_index(name);
  if (i < 0) return fail(c, BMS_ERR_INVALID, "bms_ctx_get_option: no route option named '%s'", name ? name : "(null)");
  *value = c->opt.v[i];
  return BMS_OK;
}

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
extern "C" int bms_host_register(void* p, uint64_t bytes) {
  if (!p || !bytes) return BMS_ERR_INVALID;
  if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return BMS_ERR_HIP;
  }
  return BMS_OK;
}
extern "C" int bms_host_unregister(void* p) {
  if (!p) return BMS_ERR_INVALID;
  if (hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
    return BMS_ERR_HIP;
  }
  return BMS_OK;
}

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

extern "C" int bms_ctx_set_stream(bms_ctx* c, void* s) {
  if (!c) return BMS_ERR_INVALID;
  return switch_stream(c, s ? (hipStream_t)s : c->own_stream);
}

// The device's default (null) stream has the handle 0, which bms_ctx_set_stream reads as "back to the context's own
// stream"; a caller whose allocations, copies and memsets are queued on the null stream (torch's default stream) names it
// here, so that the engine's kernels are ordered behind them instead of racing them on a non-blocking stream.
extern "C" int bms_ctx_use_default_stream(bms_ctx* c) {
  if (!c) return BMS_ERR_INVALID;
  return switch_stream(c, nullptr);
}

extern "C" int bms_ctx_set_workspace_limit(bms_ctx* c, uint64_t bytes) {
  if (!c) return BMS_ERR_INVALID;
  c->ws_limit_set = bytes != 0;
  if (!bytes) HIP_TRY(c, hipSetDevice(c->device));  // (the default is sized from THIS context's device)
  c->ws_limit = bytes ? bytes : default_ws_limit();
  return BMS_OK;
}

// Device allocations are slow on this platform -- 70 to 120 ms per GB for the tens of GB a full-size call needs (measured inside the first
// device-resident map_to_superrest_frame of a process: 'R' grows to 22.8 GB: 2 657 ms, to 32.1 GB: 2 342 ms), and memory a process has
// merely held before does not come back faster (a throw-away allocation of the whole cap up front changed nothing:
// profiles/r05_a_superrest_reserve_*).  bms_ctx_reserve therefore takes ONE allocation of `bytes` (0: one and a half times the work-space cap)
// that the context's named work-space buffers are carved from afterwards: the first full-size call of the process then allocates
// nothing.  A buffer that outgrows its region gives it back to the slab and takes a larger one (first fit, neighbours coalesced: the
// last buffer grows in place); without room it falls back to an allocation of its own.  Further calls add slabs.
extern "C" int bms_ctx_reserve(bms_ctx* c, uint64_t bytes) {
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
}

// Diagnostics of the evaluating product: out[0] = tiles and tile-boundary blocks launched since the last reset, out[1] = those whose
// samples did not fit the window of output abscissae staged in LDS (they search and read the axis in global memory: same results,
// slower), out[2] = per-column marches that started in the window and had to go on from global memory.
extern "C" int bms_ctx_get_eval_stats(bms_ctx* c, int64_t* out /*[3]*/, int reset) {
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
}

extern "C" int bms_ctx_enable_timing(bms_ctx* c, int on) {
  if (!c) return BMS_ERR_INVALID;
  c->timing = on != 0;
  return BMS_OK;
}

// accumulate finished event pairs into per-tag totals; returns totals since the last reset
extern "C" int bms_ctx_get_timing(bms_ctx* c, double* ms /*[BMS_TAG_COUNT]*/, int64_t* calls /*[BMS_TAG_COUNT]*/, int reset) {
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
}

extern "C" int bms_ctx_synchronize(bms_ctx* c) {
  if (!c) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BMS_OK;
}

// ====================================================================================================== host math

namespace {

// R_jk of scri/waveform_grid.py:130-174 == boosted_grid, transformations.py:100-148 (host loop over pixel_rotor)
void build_rotor_grid(const double fr[4], const double v[3], int n_theta, int n_phi, std::vector<Quat>& R) {
  R.resize((size_t)n_theta * n_phi);
  const Quat frq = {fr[0], fr[1], fr[2], fr[3]};
  const BoostSpec bs = make_boost_spec(v);
  for (int j = 0; j < n_theta; ++j)
    for (int k = 0; k < n_phi; ++k) R[(size_t)j * n_phi + k] = pixel_rotor(frq, bs, j, k, n_theta, n_phi);
}

// theta quadrature weights of the equiangular analysis (spinsfast.map2salm; H&W 2010): with M = 2 n_theta - 2,
//   q_j = (2 pi / M) e_j sum_{p even, -M/2 < p <= M/2} 2 cos(p theta_j) / (1 - p^2),  e_j = 1 at the poles else 2
void theta_quadrature_weights(int n_theta, std::vector<double>& q) {
  const int M = 2 * n_theta - 2;
  q.assign(n_theta, 0.0);
  for (int j = 0; j < n_theta; ++j) {
    const double th = M_PI * j / (n_theta - 1);
    double E = 0.0;
    for (int p = -M / 2 + 1; p <= M / 2; ++p)
      if ((p & 1) == 0) E += 2.0 / (1.0 - (double)p * p) * std::cos(p * th);
    q[j] = (2 * M_PI / M) * E * ((j == 0 || j == n_theta - 1) ? 1.0 : 2.0);
  }
}

// Delta^l = d^l(pi/2) in extended precision (same recurrence as DChain), packed for kernels_rotate.hip
template <class T>
void delta_matrix(int ell, std::vector<double>& D /* (2l+1)^2 row-major [mu][m] */) {
  const int n = 2 * ell + 1;
  D.assign((size_t)n * n, 0.0);
  const T r = std::sqrt((T)0.5);
  for (int mp = -ell; mp <= ell; ++mp)
    for (int m = -ell; m <= ell; ++m) {
      const int l0 = std::max(std::abs(mp), std::abs(m));
      // start value
      auto sb = [&](int k) {
        T c = 1;
        int nn = 2 * l0, kk = std::min(k, 2 * l0 - k);
        for (int i = 1; i <= kk; ++i) c = c * (T)(nn - kk + i) / (T)i;
        return std::sqrt(c);
      };
      T d0;
      const T pw = std::pow(r, (T)(2 * l0));
      if (l0 == mp)
        d0 = (((l0 - m) & 1) ? -1 : 1) * sb(l0 - m) * pw;
      else if (l0 == -mp)
        d0 = sb(l0 + m) * pw;
      else if (l0 == m)
        d0 = sb(l0 - mp) * pw;
      else
        d0 = (((l0 + mp) & 1) ? -1 : 1) * sb(l0 + mp) * pw;
      T dm1 = 0;
      for (int l = l0; l < ell; ++l) {
        T d1;
        if (l == 0) {
          d1 = 0;  // cos(pi/2) = 0
        } else {
          const T L = l, L1 = l + 1;
          const T c1 = (2 * L + 1) * (-(T)(mp * m));  // l(l+1) cos(b) = 0
          const T c2 = L1 * std::sqrt((L * L - (T)(mp * mp)) * (L * L - (T)(m * m)));
          const T den = L * std::sqrt((L1 * L1 - (T)(mp * mp)) * (L1 * L1 - (T)(m * m)));
          d1 = (c1 * d0 - c2 * dm1) / den;
        }
        dm1 = d0;
        d0 = d1;
      }
      D[(size_t)(mp + ell) * n + (m + ell)] = (double)d0;
    }
}

}  // namespace

static int ensure_delta(bms_ctx* c, int lmax, const double** d_delta, const long long** d_off) {
  double* dd = nullptr;
  long long* doff = nullptr;
  if (c->delta_lmax >= lmax) {
    *d_delta = (const double*)c->bufs["delta"].p;
    *d_off = (const long long*)c->bufs["delta_off"].p;
    return BMS_OK;
  }
  std::vector<long long> off(lmax + 1);
  long long total = 0;
  for (int l = 0; l <= lmax; ++l) {
    off[l] = total;
    const int n = 2 * l + 1, nblk = (n + ROT_MB - 1) / ROT_MB;
    total += 2LL * nblk * ROT_MB * n;
  }
  std::vector<double> packed((size_t)total, 0.0), D;
  for (int l = 0; l <= lmax; ++l) {
    delta_matrix<long double>(l, D);
    const int n = 2 * l + 1, nblk = (n + ROT_MB - 1) / ROT_MB;
    double* direct = packed.data() + off[l];
    double* transp = direct + (size_t)nblk * ROT_MB * n;
    for (int b = 0; b < nblk; ++b)
      for (int col = 0; col < n; ++col)
        for (int j = 0; j < ROT_MB; ++j) {
          const int row = b * ROT_MB + j;
          if (row < n) {
            direct[((size_t)b * n + col) * ROT_MB + j] = D[(size_t)row * n + col];  // Delta[mu=row][m'=col]
            transp[((size_t)b * n + col) * ROT_MB + j] = D[(size_t)col * n + row];  // Delta[mu=col][m=row]
          }
        }
  }
  int rc = dev_buf_t(c, "delta", (size_t)total, &dd);
  if (rc) return rc;
  rc = dev_buf_t(c, "delta_off", (size_t)lmax + 1, &doff);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(dd, packed.data(), sizeof(double) * total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(doff, off.data(), sizeof(long long) * (lmax + 1), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // host vectors go out of scope
  c->delta_lmax = lmax;
  *d_delta = dd;
  *d_off = doff;
  return BMS_OK;
}

// B images for rotate_modes_mfma_kernel: per l, B1[k = m'][mu] = Delta[mu][m'] then B2[k = mu][m] = Delta[mu][m],
// each [kpad][pd] zero padded
static int ensure_delta_mfma(bms_ctx* c, int lmax, const double** d_tab, const long long** d_off) {
  if (c->delta_mfma_lmax >= lmax) {
    *d_tab = (const double*)c->bufs["delta_mfma"].p;
    *d_off = (const long long*)c->bufs["delta_mfma_off"].p;
    return BMS_OK;
  }
  std::vector<long long> off(lmax + 1);
  long long total = 0;
  for (int l = 0; l <= lmax; ++l) {
    int kpad, pd;
    rotate_mfma_table_shape(l, &kpad, &pd);
    off[l] = total;
    total += 2LL * kpad * pd;
  }
  std::vector<double> packed((size_t)total, 0.0), D;
  for (int l = 0; l <= lmax; ++l) {
    int kpad, pd;
    rotate_mfma_table_shape(l, &kpad, &pd);
    delta_matrix<long double>(l, D);
    const int n = 2 * l + 1;
    double* B1 = packed.data() + off[l];
    double* B2 = B1 + (size_t)kpad * pd;
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) {
        B1[(size_t)a * pd + b] = D[(size_t)b * n + a];  // k = m' = a, column mu = b
        B2[(size_t)a * pd + b] = D[(size_t)a * n + b];  // k = mu = a, column m = b
      }
  }
  double* dd = nullptr;
  long long* doff = nullptr;
  int rc = dev_buf_t(c, "delta_mfma", (size_t)total, &dd);
  if (rc) return rc;
  rc = dev_buf_t(c, "delta_mfma_off", (size_t)lmax + 1, &doff);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(dd, packed.data(), sizeof(double) * total, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(doff, off.data(), sizeof(long long) * (lmax + 1), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->delta_mfma_lmax = lmax;
  *d_tab = dd;
  *d_off = doff;
  return BMS_OK;
}

// LDS image of Delta^l for l = ell_min..ell_max (kernels_rotate_resident.hip); false if the range does not fit the LDS
static int ensure_delta_resident(bms_ctx* c, int ell_min, int ell_max, bool* ok, RotResPlan* P, size_t* lds_bytes,
                                 const double** d_tab, unsigned int** d_counter) {
  *ok = rotate_resident_plan(ell_min, ell_max, P, lds_bytes);
  if (!*ok) return BMS_OK;
  char name[64];
  snprintf(name, sizeof name, "rot_res_tab_%d_%d", ell_min, ell_max);
  int rc = dev_buf_t(c, "rot_res_counter", 4, d_counter);
  if (rc) return rc;
  if (!c->n_cu) {
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount;
  }
  double* dt = nullptr;
  if ((rc = dev_buf_t(c, name, (size_t)P->tab_doubles, &dt))) return rc;
  *d_tab = dt;
  if (c->rot_res_plans.count({ell_min, ell_max})) return BMS_OK;
  std::vector<double> image((size_t)P->tab_doubles, 0.0), D;
  for (int l = ell_min; l <= ell_max; ++l) {
    delta_matrix<long double>(l, D);
    rotate_resident_pack(*P, l, D.data(), image.data());
  }
  HIP_TRY(c, hipMemcpyAsync(dt, image.data(), sizeof(double) * image.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the host vector goes out of scope
  c->rot_res_plans[{ell_min, ell_max}] = *P;
  return BMS_OK;
}

// ====================================================================================================== rotation

constexpr int ROT_RING = 64;
// sync_after = false (internal callers, device data, constant rotor): the call returns with the work enqueued
static int rotate_impl(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max,
                       const void* spinors, bool series, bool sync_after = true) {
  if (!c) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_times < 0 || ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad sizes");
  const int64_t n_modes = LM_total_size(ell_min, ell_max);
  if (ld < n_modes) return fail(c, BMS_ERR_INVALID, "row stride %lld smaller than %lld modes", (long long)ld, (long long)n_modes);
  if (n_times == 0) return BMS_OK;
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
  for (const Segment& sg : segs) {
    double* seg_data = d_data + 2 * ((long long)sg.lo * sg.lo - (long long)ell_min * ell_min);
    if (sg.kind == 0)
      TIMED(c, BMS_TAG_ROTATE, launch_rotate_modes_resident(c->stream, seg_data, n_times, ld, d_rot, series ? 4 : 0, sg.tab, sg.plan, sg.lds,
                                                            nullptr, c->n_cu));
    else if (sg.kind == 1)
      TIMED(c, BMS_TAG_ROTATE, launch_rotate_modes_mfma(c->stream, seg_data, n_times, ld, sg.lo, sg.hi, d_rot, series ? 4 : 0, d_delta, d_off));
    else
      TIMED(c, BMS_TAG_ROTATE, launch_rotate_modes(c->stream, seg_data, n_times, ld, sg.lo, sg.hi, d_rot, series ? 4 : 0, d_delta, d_off));
  }
  if (mem == BMS_HOST) {
    HIP_TRY(c, hipMemcpyAsync(data, d_data, data_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else if (!series && sync_after) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the constant rotor was staged from a host stack copy
  }
  return BMS_OK;
}

extern "C" int bms_rotate_const(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max,
                                const double q[4]) {
  if (!c || !q) return BMS_ERR_INVALID;
  const double sp[4] = {q[0], q[3], q[2], q[1]};  // (w + i z, y + i x)
  return rotate_impl(c, data, mem, n_times, ld, ell_min, ell_max, sp, false);
}

extern "C" int bms_rotate_series(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min,
                                 int ell_max, const void* spinors) {
  if (!c || !spinors) return BMS_ERR_INVALID;
  return rotate_impl(c, data, mem, n_times, ld, ell_min, ell_max, spinors, true);
}

// The reference's numba kernel takes the packed Wigner matrices it is handed, not a rotor (scri/rotations.py:346-367:
// `_rotate_decomposition_basis_by_constant(data, ell_min, ell_max, D, tmp)` with D from sf._Wigner_D_matrices, :327):
//     data[t, l, m] <- sum_m' data[t, l, m'] D^l[m', m],   D block l row-major (m', m) at _linear_matrix_offset(l, ell_min).
// One complex GEMM per l on the synthesis kernel ([N x (2l+1)] . [(2l+1) x (2l+1)], operands zero padded to its 8 x 64
// panels), out of place into a work buffer, then copied over the input.  This is the seam a binding replaces the numba
// kernel at; callers that have the rotor use bms_rotate_const, which never forms D.
static int upload(bms_ctx* c, const char* name, const void* host, size_t bytes, void** dev);
extern "C" int bms_rotate_const_D(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max,
                                  const void* D_host) {
  if (!c || !data || !D_host) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_times < 0 || ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad sizes");
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
}

// D matrices through the rotation kernel itself: rotate the identity blocks (row (l, m') = delta_{m'})
extern "C" int bms_wigner_D(bms_ctx* c, const double q[4], int ell_min, int ell_max, void* D_host) {
  if (!c || !q || !D_host) return BMS_ERR_INVALID;
  if (ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad ell range");
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
}

// ====================================================================================================== transform

namespace {

constexpr int SPLINE_TILE = 320;  // knots per (pixel group, tile) wave: measured sweep 128..640 on cfg3, best at 320 (halo re-reads 10 %)
constexpr int SPLINE_HALO = 32;

inline long long round_up(long long a, long long b) { return (a + b - 1) / b * b; }

struct PixelTables {
  int n_theta = 0, n_phi = 0, n_pix = 0;
  std::vector<Quat> R;
  std::vector<double> k, alpha, skew_a, skew_b;
  double beta = 0, gamma = 1, tt = 0;
  bool nontrivial = false;  // beta != 0 or any supertranslation mode beyond l = 0 nonzero
  double uprm_scale_min = 0, uprm_scale_max = 0;
};

// scalars of the transformation + allocation of the host-side per-pixel arrays
void init_pixel_tables(const bms_transformation* tr, PixelTables& T) {
  T.n_theta = tr->n_theta;
  T.n_phi = tr->n_phi;
  T.n_pix = tr->n_theta * tr->n_phi;
  const double* v = tr->boost_velocity;
  T.beta = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  T.gamma = 1 / std::sqrt(1 - T.beta * T.beta);
  const cplx* st = (const cplx*)tr->supertranslation;
  const int nst = (tr->ell_max_supertranslation + 1) * (tr->ell_max_supertranslation + 1);
  T.tt = st[0].re / std::sqrt(4 * M_PI);  // constant_from_ell_0_mode(supertranslation[0]).real
  T.nontrivial = T.beta != 0;
  for (int i = 1; i < nst; ++i)
    if (st[i].re != 0 || st[i].im != 0) T.nontrivial = true;
  T.k.resize(T.n_pix);
  T.alpha.resize(T.n_pix);
  T.skew_a.resize(T.n_pix);
  T.skew_b.resize(T.n_pix);
}

PixelSpec base_pixel_spec(const bms_transformation* tr, const PixelTables& T) {
  PixelSpec P{};
  P.frq = {tr->frame_rotation[0], tr->frame_rotation[1], tr->frame_rotation[2], tr->frame_rotation[3]};
  for (int i = 0; i < 3; ++i) P.v[i] = tr->boost_velocity[i];
  P.bs = make_boost_spec(tr->boost_velocity);
  P.gamma = T.gamma;
  P.tt = T.tt;
  P.n_theta = tr->n_theta;
  P.n_phi = tr->n_phi;
  P.lst = tr->ell_max_supertranslation;
  P.mode = -1;
  return P;
}

// host-only evaluation of the per-pixel scalars (bms_shard_plan: no GPU needed); same code as pixel_tables_kernel
void build_pixel_tables(const bms_transformation* tr, PixelTables& T) {
  init_pixel_tables(tr, T);
  PixelSpec P = base_pixel_spec(tr, T);
  P.st = (const cplx*)tr->supertranslation;
  T.R.resize(T.n_pix);
  PixelOut O{};
  O.rotors = (double*)T.R.data();
  O.k = T.k.data();
  O.alpha = T.alpha.data();
  O.skew_a = T.skew_a.data();
  O.skew_b = T.skew_b.data();
  for (int p = 0; p < T.n_pix; ++p) pixel_tables_one(P, O, p, p);
}

// output time window (waveform_grid.py:564-568 == transformations.py:391-396)
void output_window(const PixelTables& T, const double* t, int64_t n, int64_t& i_lo, int64_t& i_hi) {
  double umin = -INFINITY, umax = INFINITY;
  for (int p = 0; p < T.n_pix; ++p) {
    umin = std::max(umin, T.k[p] * (t[0] - T.alpha[p]));
    umax = std::min(umax, T.k[p] * (t[n - 1] - T.alpha[p]));
  }
  const double ig = 1 / T.gamma;
  // uprm_i = (1/gamma) (t_i - tt) is non-decreasing in i
  i_lo = std::partition_point(t, t + n, [&](double ti) { return ig * (ti - T.tt) < umin; }) - t;
  i_hi = std::partition_point(t, t + n, [&](double ti) { return ig * (ti - T.tt) <= umax; }) - t;
  if (i_hi < i_lo) i_hi = i_lo;
}

// the AsymptoticBondiData flavour divides: timeprime = (u - tt) / gamma (transformations.py:391-396)
void output_window_abd(const PixelTables& T, const double* u, int64_t n, int64_t& i_lo, int64_t& i_hi) {
  double umin = -INFINITY, umax = INFINITY;
  for (int p = 0; p < T.n_pix; ++p) {
    umin = std::max(umin, T.k[p] * (u[0] - T.alpha[p]));
    umax = std::min(umax, T.k[p] * (u[n - 1] - T.alpha[p]));
  }
  i_lo = std::partition_point(u, u + n, [&](double ui) { return (ui - T.tt) / T.gamma < umin; }) - u;
  i_hi = std::partition_point(u, u + n, [&](double ui) { return (ui - T.tt) / T.gamma <= umax; }) - u;
  if (i_hi < i_lo) i_hi = i_lo;
}

// knots needed to evaluate output samples [c0, c1): [ja, jb] inclusive (before halo)
void needed_knots(const PixelTables& T, const double* t, int64_t n, int64_t c0, int64_t c1, int64_t& ja, int64_t& jb) {
  double lo = INFINITY, hi = -INFINITY;
  const double x0 = t[c0], x1 = t[c1 - 1];
  for (int p = 0; p < T.n_pix; ++p) {
    lo = std::min(lo, x0 + (T.skew_a[p] * (x0 - T.tt) + T.skew_b[p]));
    hi = std::max(hi, x1 + (T.skew_a[p] * (x1 - T.tt) + T.skew_b[p]));
  }
  ja = (std::upper_bound(t, t + n, lo) - t) - 1;  // last knot <= lo
  jb = std::lower_bound(t, t + n, hi) - t;        // first knot >= hi
  ja = std::max<int64_t>(ja, 0);
  jb = std::min<int64_t>(jb, n - 1);
}

struct FieldPlan {  // one field to synthesise: input modes, its SWSH matrix
  const double* d_data = nullptr;  // device c16[n][ld]
  int64_t ld = 0;
  int ell_min = 0, ell_max = 0, spin = 0;
  double* d_B = nullptr;  // synthesis matrix
  long long ldb = 0;
  int K = 0;  // 2 * n_modes
};

}  // namespace


static int upload(bms_ctx* c, const char* name, const void* host, size_t bytes, void** dev);

static int build_analysis(bms_ctx* c, const char* tag, int n_theta, int n_phi, int spin, int ell_min_out, int ell_max_out,
                          AnalysisPlan& A) {
  hipStream_t S = c->stream;
  const std::array<int, 6> key = {n_theta, n_phi, spin, ell_min_out, ell_max_out, (c->opt.on(OPT_NO_FUSED_ANALYSIS) ? 1 : 0) + (c->opt.on(OPT_NO_LARGE_ANALYSIS) ? 2 : 0)};
  {
    auto it = c->plans.find(tag);
    if (it != c->plans.end() && it->second.first == key && !c->opt.on(OPT_NO_PLAN_CACHE)) {
      A = it->second.second;
      return BMS_OK;
    }
    c->plans.erase(tag);
  }
  A.n_theta = n_theta;
  A.n_phi = n_phi;
  A.n_pix = n_theta * n_phi;
  A.n_out = LM_total_size(ell_min_out, ell_max_out);
  A.L = ell_max_out;
  A.nm = 2 * ell_max_out + 1;
  A.separable = n_theta <= MAX_THETA_SEPARABLE;
  std::vector<double> qth;
  theta_quadrature_weights(n_theta, qth);
  int rc;
  void* vp;
  char nm_[64];
  A.fused = A.separable && fused_analysis_supported(n_theta, n_phi, A.L, A.n_out) && !c->opt.on(OPT_NO_FUSED_ANALYSIS);
  A.large = A.separable && !A.fused && large_analysis_supported(n_theta, n_phi, A.L) && !c->opt.on(OPT_NO_LARGE_ANALYSIS) &&
            !c->opt.on(OPT_NO_FUSED_ANALYSIS);
  A.ell_min_out = ell_min_out;
  A.spin = spin;
  if (A.separable) {
    if (A.large) {
      // twiddles are computed in the kernel; only the theta table below is needed
    } else if (A.fused) {
      const size_t nd = fused_dft_table_size(n_phi, A.L);
      snprintf(nm_, sizeof nm_, "dcs_%d_%d", n_phi, A.L);
      if ((rc = dev_buf_t(c, nm_, nd, &A.d_dcs))) return rc;
      HIP_TRY(c, hipMemsetAsync(A.d_dcs, 0, sizeof(double) * nd, S));
      TIMED(c, BMS_TAG_SETUP, launch_dft_cs_matrix(S, n_phi, A.L, A.d_dcs));
    } else {
      // phi-DFT matrix [2 n_phi -> 16] x [2 (2L+1) -> 128]
      A.ld_dft = round_up(2LL * A.nm, 128);
      const long long rows = round_up(2LL * n_phi, 16);
      snprintf(nm_, sizeof nm_, "dft_%d_%d", n_phi, A.L);
      if ((rc = dev_buf_t(c, nm_, (size_t)rows * A.ld_dft, &A.d_dft))) return rc;
      HIP_TRY(c, hipMemsetAsync(A.d_dft, 0, sizeof(double) * rows * A.ld_dft, S));
      TIMED(c, BMS_TAG_SETUP, launch_dft_matrix(S, n_phi, A.L, A.d_dft, A.ld_dft));
    }
    // theta table from sLambda_lm(theta_j) = sYlm(R(theta_j, 0))
    std::vector<double> rot(4 * (size_t)n_theta), wth(n_theta);
    std::vector<int> mindex(A.n_out);
    for (int j = 0; j < n_theta; ++j) {
      const Quat q = from_spherical_coords(M_PI * j / (n_theta - 1), 0.0);
      rot[4 * j] = q.w, rot[4 * j + 1] = q.x, rot[4 * j + 2] = q.y, rot[4 * j + 3] = q.z;
      wth[j] = qth[j] / n_phi;
    }
    for (int l = ell_min_out; l <= ell_max_out; ++l)
      for (int m = -l; m <= l; ++m) mindex[LM_index(l, m, ell_min_out)] = m + A.L;
    snprintf(nm_, sizeof nm_, "ana_rot_%s", tag);
    if ((rc = upload(c, nm_, rot.data(), 8 * rot.size(), &vp))) return rc;
    const double* d_rot = (const double*)vp;
    snprintf(nm_, sizeof nm_, "ana_wth_%s", tag);
    if ((rc = upload(c, nm_, wth.data(), 8 * wth.size(), &vp))) return rc;
    const double* d_wth = (const double*)vp;
    snprintf(nm_, sizeof nm_, "ana_mi_%s", tag);
    if ((rc = upload(c, nm_, mindex.data(), sizeof(int) * mindex.size(), &vp))) return rc;
    A.d_mindex = (int*)vp;
    double* d_Y;
    snprintf(nm_, sizeof nm_, "ana_Y_%s", tag);
    if ((rc = dev_buf_t(c, nm_, (size_t)n_theta * A.n_out * 2, &d_Y))) return rc;
    HIP_TRY(c, hipMemsetAsync(d_Y, 0, 16 * (size_t)n_theta * A.n_out, S));
    TIMED(c, BMS_TAG_SETUP, launch_swsh_values(S, d_rot, n_theta, spin, ell_min_out, ell_max_out, d_Y));
    snprintf(nm_, sizeof nm_, "ana_T_%s", tag);
    if ((rc = dev_buf_t(c, nm_, (size_t)n_theta * A.n_out, &A.d_T))) return rc;
    TIMED(c, BMS_TAG_SETUP, launch_theta_table(S, d_Y, d_wth, n_theta, A.n_out, A.d_T));
    HIP_TRY(c, hipStreamSynchronize(S));  // host vectors above go out of scope
  } else {
    std::vector<double> wpix((size_t)A.n_pix), grid_rot(4 * (size_t)A.n_pix);
    for (int j = 0; j < n_theta; ++j)
      for (int k = 0; k < n_phi; ++k) {
        const int p = j * n_phi + k;
        wpix[p] = qth[j] / n_phi;
        const Quat q = from_spherical_coords(M_PI * j / (n_theta - 1), (2 * M_PI) * k / n_phi);
        grid_rot[4 * p] = q.w, grid_rot[4 * p + 1] = q.x, grid_rot[4 * p + 2] = q.y, grid_rot[4 * p + 3] = q.z;
      }
    if ((rc = upload(c, "grid_rotors", grid_rot.data(), 8 * grid_rot.size(), &vp))) return rc;
    const double* d_grot = (const double*)vp;
    if ((rc = upload(c, "wpix", wpix.data(), 8 * wpix.size(), &vp))) return rc;
    const double* d_wpix = (const double*)vp;
    A.ldw = round_up(2LL * A.n_out, 128);
    const long long wrows = round_up(2LL * A.n_pix, 16);
    snprintf(nm_, sizeof nm_, "Wana_%s", tag);
    if ((rc = dev_buf_t(c, nm_, (size_t)wrows * A.ldw, &A.d_W))) return rc;
    HIP_TRY(c, hipMemsetAsync(A.d_W, 0, sizeof(double) * wrows * A.ldw, S));
    TIMED(c, BMS_TAG_SETUP, launch_quadrature_matrix(S, d_grot, d_wpix, A.n_pix, spin, ell_min_out, ell_max_out, A.d_W, A.ldw));
    HIP_TRY(c, hipStreamSynchronize(S));
  }
  c->plans[tag] = {key, A};
  return BMS_OK;
}

// G: [rows][2 n_pix] (row stride exactly 2 n_pix doubles) -> out[rows][ldo] complex modes
static bool analysis_reads_contiguous_rows(const AnalysisPlan& A) { return !A.fused && !A.large && A.separable; }

static int run_analysis(bms_ctx* c, const AnalysisPlan& A, const double* d_G, long long rows, double* d_out, long long ldo,
                        const int* col_of_pixel = nullptr, long long ld_cols = 0) {
  hipStream_t S = c->stream;
  const long long P2 = 2LL * A.n_pix, ld = ld_cols ? ld_cols : P2;  // row stride of d_G
  if (A.fused) {
    TIMED(c, BMS_TAG_ANALYSIS_FUSED, launch_analysis_fused(S, d_G, ld, rows, A.n_theta, A.n_phi, A.L, A.n_out,
                                                           A.d_mindex, A.d_T, A.d_dcs, d_out, ldo, col_of_pixel, A.spin, !c->opt.on(OPT_NO_SPLIT_ANALYSIS)));
  } else if (A.large) {
    if (col_of_pixel) return fail(c, BMS_ERR_UNSUPPORTED, "internal: sorted columns need the fused analysis");
    double* d_F;
    int rc = dev_buf_t(c, "Fphi", (size_t)rows * A.nm * large_analysis_jp(A.n_theta) * 2, &d_F);
    if (rc) return rc;
    TIMED(c, BMS_TAG_ANALYSIS_LARGE, launch_analysis_large(S, d_G, ld, rows, A.n_theta, A.n_phi, A.L, A.ell_min_out, A.d_T, d_F, d_out, ldo));
  } else if (A.separable) {
    if (col_of_pixel) return fail(c, BMS_ERR_UNSUPPORTED, "internal: sorted columns need the fused analysis");
    if (ld != P2) return fail(c, BMS_ERR_UNSUPPORTED, "internal: the separable analysis reads contiguous rows");
    double* d_F;
    int rc = dev_buf_t(c, "Fphi", (size_t)rows * A.n_theta * 2 * A.nm, &d_F);
    if (rc) return rc;
    TIMED(c, BMS_TAG_GEMM_ANALYSIS, launch_dgemm(S, d_G, 2LL * A.n_phi, A.d_dft, A.ld_dft, d_F, 2LL * A.nm, rows * A.n_theta,
                                                 2 * A.nm, 2 * A.n_phi, nullptr, nullptr));
    TIMED(c, BMS_TAG_THETA_QUADRATURE,
          launch_theta_quadrature(S, d_F, rows, A.n_theta, A.nm, A.n_out, A.d_mindex, A.d_T, d_out, ldo));
  } else {
    TIMED(c, BMS_TAG_GEMM_ANALYSIS, launch_dgemm(S, d_G, ld, A.d_W, A.ldw, d_out, ldo, rows, 2 * A.n_out, (int)P2, nullptr, nullptr));
  }
  return BMS_OK;
}

static int upload(bms_ctx* c, const char* name, const void* host, size_t bytes, void** dev) {
  int rc = dev_buf(c, name, bytes, dev);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, c->stream));
  return BMS_OK;
}

// times [lo, hi) and the spline table of knots [j0, j1) on the device; both pointers are indexed by GLOBAL knot number
static int upload_times(bms_ctx* c, const double* t, int64_t n, int64_t lo, int64_t hi, int64_t j0, int64_t j1, double** d_x,
                        SplineTable** d_tab) {
  void* vp;
  int rc = upload(c, "times", t + lo, 8 * (size_t)(hi - lo), &vp);
  if (rc) return rc;
  *d_x = (double*)vp - lo;
  SplineTable* tab;
  if ((rc = dev_buf_t(c, "spline_table", (size_t)(hi - lo), &tab))) return rc;
  *d_tab = tab - lo;
  TIMED(c, BMS_TAG_SETUP, launch_spline_table(c->stream, *d_x, n, *d_tab, std::max(j0, lo), std::min(j1, hi)));
  return BMS_OK;
}

// the same for the B-spline form of the spline (kernels_bspline.hip)
static int upload_times_bspline(bms_ctx* c, const double* t, int64_t n, int64_t lo, int64_t hi, int64_t j0, int64_t j1, double** d_x,
                                BsplineTable** d_tab, BsplineForward** d_fwd) {
  void* vp;
  int rc = upload(c, "times", t + lo, 8 * (size_t)(hi - lo), &vp);
  if (rc) return rc;
  *d_x = (double*)vp - lo;
  BsplineTable* tab;
  BsplineForward* fwd;
  if ((rc = dev_buf_t(c, "bspline_table", (size_t)(hi - lo), &tab))) return rc;
  if ((rc = dev_buf_t(c, "bspline_forward", (size_t)(hi - lo), &fwd))) return rc;
  *d_tab = tab - lo;
  *d_fwd = fwd - lo;
  TIMED(c, BMS_TAG_SETUP, launch_bspline_table(c->stream, *d_x, n, *d_tab, *d_fwd, lo, std::max(j0, lo), std::min(j1, hi)));
  return BMS_OK;
}

static int stage_in(bms_ctx* c, const char* name, const void* src, int mem, size_t bytes, const double** dev) {
  if (mem == BMS_DEVICE) {
    *dev = (const double*)src;
    return BMS_OK;
  }
  void* p;
  int rc = upload(c, name, src, bytes, &p);
  *dev = (const double*)p;
  return rc;
}

// Time samples a call touches on the device: the rows it holds plus the spline-table warm-up margin.  A shard of an
// 8 x 1e5-step series uploads and checks 1e5 + 128 samples, not 8e5 (the host still sees the global array: window
// search and chunk planning are binary searches on it).
constexpr int64_t TIME_MARGIN = 64;
static void time_window(int64_t n, const bms_shard* sh, int64_t& lo, int64_t& hi) {
  lo = 0, hi = n;
  if (sh && sh->data_row0 >= 0 && sh->data_rows >= 0 && sh->data_row0 + sh->data_rows <= n) {
    lo = std::max<int64_t>(0, sh->data_row0 - TIME_MARGIN);
    hi = std::min<int64_t>(n, sh->data_row0 + sh->data_rows + TIME_MARGIN);
  }
}

// `regular` (optional): whether the tiled spline recurrences may be trusted on this time axis.  Their truncated starts
// rely on the factors of the spline systems decaying over a 32-knot halo; that holds for any mesh whose steps do not
// grow or shrink geometrically over many knots in a row (a sudden jump of any size is harmless), and fails for sustained
// grading: a ratio of 1.3 per step over 33 knots (steps varying 4e3-fold inside the halo) costs 5e-14, 1.4 already 2e-12,
// 2.0 1e-5.  Criterion: steps within any 48 consecutive knots vary by at most 1e3 (then <= 1e-14); otherwise the caller
// runs the exact single-tile recurrences of the slope form.
// spline tile for the whole-series building blocks (slope form): one tile = exact recurrences on an irregular axis
static int spline_tile_for(const double* x, int64_t n);

// (the two halves of validate_common, for the caller that has the GPU start on the call before the host walks the time axis)
static int validate_transformation(bms_ctx* c, int64_t n, const double* t, const bms_transformation* tr, int64_t n_min) {
  if (n < n_min) return fail(c, BMS_ERR_INVALID, "need at least %lld time steps, got %lld", (long long)n_min, (long long)n);
  if (!(t[n - 1] > t[0])) return fail(c, BMS_ERR_INVALID, "time array must be strictly increasing (first/last)");
  if (tr->n_theta < 2 || tr->n_phi < 1) return fail(c, BMS_ERR_INVALID, "bad grid size %d x %d", tr->n_theta, tr->n_phi);
  if (tr->ell_max_supertranslation < 1 || !tr->supertranslation) return fail(c, BMS_ERR_INVALID, "supertranslation must hold at least l <= 1");
  const double* v = tr->boost_velocity;
  if (!(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] < 1.0)) return fail(c, BMS_ERR_INVALID, "boost speed must be < 1");
  return BMS_OK;
}
static int walk_time_axis(bms_ctx* c, const double* t, int64_t lo, int64_t hi, bool* regular) {
  double bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {0.0, 0.0, 0.0};  // step range of the last three 16-step blocks
  bool reg = true;
  // (block by block, the block's minimum and maximum by a branch-free inner loop the compiler vectorises -- this walk is host
  // time during which the GPU has nothing of the call yet: 63 us per 1e5 samples as an element-by-element loop with its early exit)
  for (int64_t b0 = std::max<int64_t>(lo, 0) + 1; b0 < hi; b0 += 16) {
    const int64_t b1 = std::min<int64_t>(b0 + 16, hi);
    double mn_b = INFINITY, mx_b = -INFINITY;
    bool nan_b = false;
    for (int64_t i = b0; i < b1; ++i) {
      const double h = t[i] - t[i - 1];
      mn_b = h < mn_b ? h : mn_b;
      mx_b = h > mx_b ? h : mx_b;
      nan_b |= h != h;
    }
    if (!(mn_b > 0) || nan_b) {
      for (int64_t i = b0; i < b1; ++i)
        if (!(t[i] - t[i - 1] > 0)) return fail(c, BMS_ERR_INVALID, "time array must be strictly increasing (index %lld)", (long long)i);
    }
    bmin[2] = mn_b, bmax[2] = mx_b;
    const double mn = std::min(bmin[0], std::min(bmin[1], bmin[2])), mx = std::max(bmax[0], std::max(bmax[1], bmax[2]));
    if (mx > 1e3 * mn) reg = false;
    bmin[0] = bmin[1], bmin[1] = bmin[2];
    bmax[0] = bmax[1], bmax[1] = bmax[2];
  }
  if (regular) *regular = reg || BMS_PROBE_ENV("SCRI_AMD_ASSUME_REGULAR_MESH") != nullptr;  // (the switch exists to show what the guard prevents)
  return BMS_OK;
}
static int validate_common(bms_ctx* c, int64_t n, const double* t, const bms_transformation* tr, int64_t lo = 0, int64_t hi = -1,
                           bool* regular = nullptr, int64_t n_min = 4) {
  // (order of the checks as it always was: size, first/last, the walk, then the transformation)
  if (n < n_min) return fail(c, BMS_ERR_INVALID, "need at least %lld time steps, got %lld", (long long)n_min, (long long)n);
  if (!(t[n - 1] > t[0])) return fail(c, BMS_ERR_INVALID, "time array must be strictly increasing (first/last)");
  int rc = walk_time_axis(c, t, lo, hi < 0 ? n : hi, regular);
  if (rc) return rc;
  return validate_transformation(c, n, t, tr, n_min);
}


static int spline_tile_for(const double* x, int64_t n) {
  double bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {0.0, 0.0, 0.0};
  int64_t in_block = 0;
  for (int64_t i = 1; i < n; ++i) {
    const double h = x[i] - x[i - 1];
    bmin[2] = std::min(bmin[2], h);
    bmax[2] = std::max(bmax[2], h);
    if (++in_block == 16 || i == n - 1) {
      const double mn = std::min(bmin[0], std::min(bmin[1], bmin[2])), mx = std::max(bmax[0], std::max(bmax[1], bmax[2]));
      if (mx > 1e3 * mn && !BMS_PROBE_ENV("SCRI_AMD_ASSUME_REGULAR_MESH")) return (int)std::min<int64_t>(n + 1, 0x7fffffff);
      bmin[0] = bmin[1], bmin[1] = bmin[2], bmin[2] = INFINITY;
      bmax[0] = bmax[1], bmax[1] = bmax[2], bmax[2] = 0.0;
      in_block = 0;
    }
  }
  return SPLINE_TILE;
}

// Per-pixel tables on the GPU.  `coef0/coef1` (host, (lst+1)^2 complex each, may be null) are uploaded next to the
// supertranslation modes; the four scalars the host needs for the output window and the chunk plan come back in T.
struct DevPixel {
  double *rotors, *k, *alpha, *skew_a, *skew_b, *col_off, *col_scale, *xa, *xb, *ethk, *etha, *ethetha, *ik, *ik3;
  const int* col_of_pixel = nullptr;  // set when the grid is stored as a column plan (kernels_swsh.hip, pixel_sort_kernel)
};

struct PieceTables {  // tables shared by the pieces of one pipelined call: per direction, and per knot of the WHOLE series
  PixelTables T;
  DevPixel DP;
  int col_plan = 0;  // the column order T / DP were built in
  bool times_valid = false;
  double* d_x = nullptr;
  BsplineTable* d_bstab = nullptr;
  BsplineForward* d_bsfwd = nullptr;
};
// How far apart the lanes of a back-substitution wave can stand: ranges of the skew rate and offset within any block of 64
// columns [cA + 64 b, ...) of the launch (host copies of the per-column tables).
static BsplineSpread skew_spread(const PixelTables& T, int cA, int cB, const double* x_host) {
  BsplineSpread sp = {0.0, 0.0, x_host};
  if ((int)T.skew_a.size() < cB || (int)T.skew_b.size() < cB) {
    sp.x = nullptr;  // (no host copy: the kernel gathers)
    return sp;
  }
  for (int c0 = cA; c0 < cB; c0 += 64) {
    double a0 = T.skew_a[c0], a1 = a0, b0 = T.skew_b[c0], b1 = b0;
    for (int p = c0; p < std::min(cB, c0 + 64); ++p) {
      a0 = std::min(a0, T.skew_a[p]), a1 = std::max(a1, T.skew_a[p]);
      b0 = std::min(b0, T.skew_b[p]), b1 = std::max(b1, T.skew_b[p]);
    }
    sp.skew_rate_range = std::max(sp.skew_rate_range, a1 - a0);
    sp.skew_offset_range = std::max(sp.skew_offset_range, b1 - b0);
  }
  return sp;
}

// Bound on how many rows a sample can lie from the knots of its spline window within knots [g0, g1): |skew| / (mean step) with a margin
// (the evaluation verifies the bracket it searches and falls back to the whole range: kernels_gemm_eval.hip).  0: no bound known.
static int eval_search_halfwidth(const PixelTables& T, int cA, int cB, const double* x_host, int64_t g0, int64_t g1) {
  if ((int)T.skew_a.size() < cB || (int)T.skew_b.size() < cB || g1 - g0 < 2) return 0;
  double am = 0.0, bm = 0.0;
  for (int p = cA; p < cB; ++p) am = std::max(am, std::fabs(T.skew_a[p])), bm = std::max(bm, std::fabs(T.skew_b[p]));
  const double xm = std::max(std::fabs(x_host[g0] - T.tt), std::fabs(x_host[g1 - 1] - T.tt));
  // the SHORTEST local step counts (mean over 64 knots, tile by tile): on a graded axis the launch's mean step understates the rows a
  // skew spans where the steps are short, and a bound that is too small sends every tile there through the global-memory search
  double dx = (x_host[g1 - 1] - x_host[g0]) / (double)(g1 - 1 - g0);
  for (int64_t k = g0; k + 64 < g1; k += 64) dx = std::min(dx, (x_host[k + 64] - x_host[k]) / 64.0);
  if (!(dx > 0.0)) return 0;
  const double rows = 1.25 * (am * xm + bm) / dx + 3.0;
  if (!(rows < 1e6)) return 0;
  return (int)std::ceil(rows);
}

// Is the rotor grid of this transformation of the form F R(Theta_j, phi'_k), rings of the rotated equiangular grid at
// colatitudes Theta_j?  Always without a boost (Theta_j = theta'_j); with one exactly when it points along the polar axis of the
// rotated grid: the aberration then moves whole rings, B'(r') F R(theta', phi') = F R(Theta(theta'), phi') with no spin phase
// (scri/waveform_grid.py:141-161: the rotation is about r' x v, which lies in the ring's tangent plane).  SURVEY section 7,
// step 4(b).  The form is CHECKED on the rotors themselves (pixel_rotor, the code the dense route uses), not assumed.
// The two-kernel synthesis moves (2 l_max + 1) x n_theta numbers per time step through HBM twice; the dense product it replaces
// costs n_modes x n_pix multiply-adds per step and overtakes it only from about l_max = 13 on the default grids (measured:
// tools/axis_boost_probe.py; l <= 8 on 17 x 17: 0.45 ms dense, 0.81 ms separable per 10^5 steps; l <= 16 on 33 x 33: 4.2 and 2.3).
static bool large_synthesis_route(const bms_ctx* c, int n_theta, int n_phi, int ell_min, int ell_max) {
  return !c->opt.on(OPT_NO_SEPARABLE_SYNTHESIS) && !c->opt.on(OPT_NO_LARGE_SYNTHESIS) && large_synthesis_supported(n_theta, n_phi, ell_min, ell_max) != 0;
}
static bool axis_boost_pays(const bms_ctx* c, int n_modes, int n_theta, int n_phi) {
  const long long o = c->opt.v[OPT_AXIS_BOOST_MIN_WORK];  // (< 0: the built-in threshold; 0: always)
  const long long min_work = o < 0 ? 160000 : o;
  return (long long)n_modes * n_theta * n_phi >= min_work;
}
static bool separable_rotor_grid(const bms_transformation* tr, std::vector<double>& thetas, bool axis_boost_off = false) {
  const int n_theta = tr->n_theta, n_phi = tr->n_phi;
  const double* fr = tr->frame_rotation;
  const Quat F = {fr[0], fr[1], fr[2], fr[3]};
  const BoostSpec bs = make_boost_spec(tr->boost_velocity);
  thetas.resize(n_theta);
  if (!bs.boosted) {
    for (int j = 0; j < n_theta; ++j) thetas[j] = M_PI * j / (n_theta - 1);
    return true;
  }
  if (axis_boost_off) return false;
  double zf[3];
  rotate_z(F, zf);
  const double cx = bs.vhat[1] * zf[2] - bs.vhat[2] * zf[1], cy = bs.vhat[2] * zf[0] - bs.vhat[0] * zf[2], cz = bs.vhat[0] * zf[1] - bs.vhat[1] * zf[0];
  if (std::sqrt(cx * cx + cy * cy + cz * cz) > 1e-15) return false;  // (parallel to rounding, nothing looser)
  const double n2 = F.w * F.w + F.x * F.x + F.y * F.y + F.z * F.z;
  const Quat Finv = {F.w / n2, -F.x / n2, -F.y / n2, -F.z / n2};
  // Every pixel of the two rings at either end (acos near 1 turns an ulp of r'.v into 1e-8 rad there, as it does in the reference:
  // a frame whose axis is parallel to v only to rounding can fail right there and then keeps the dense route), a few per ring
  // elsewhere.
  const int ks[4] = {0, 1 % n_phi, n_phi / 3, n_phi - 1};
  for (int j = 0; j < n_theta; ++j) {
    double th = 0.0, ph = 0.0;
    as_spherical_coords(qmul(Finv, pixel_rotor(F, bs, j, 0, n_theta, n_phi)), th, ph);
    thetas[j] = th;
    const bool end_ring = j < 2 || j >= n_theta - 2;
    for (int i = 0; i < (end_ring ? n_phi : 4); ++i) {
      const int k = end_ring ? i : ks[i];
      const Quat G = qmul(Finv, pixel_rotor(F, bs, j, k, n_theta, n_phi));
      const Quat E = from_spherical_coords(th, (2 * M_PI) * k / n_phi);
      const double sgn = (G.w * E.w + G.x * E.x + G.y * E.y + G.z * E.z) < 0 ? -1.0 : 1.0;
      const double d = std::fabs(G.w - sgn * E.w) + std::fabs(G.x - sgn * E.x) + std::fabs(G.y - sgn * E.y) + std::fabs(G.z - sgn * E.z);
      if (!(d <= 1e-13)) return false;
    }
  }
  return true;
}

// ... with the context remembering the last answer
static bool separable_rotor_grid(bms_ctx* c, const bms_transformation* tr, std::vector<double>& thetas) {
  const double key[9] = {tr->frame_rotation[0], tr->frame_rotation[1], tr->frame_rotation[2], tr->frame_rotation[3], tr->boost_velocity[0],
                         tr->boost_velocity[1], tr->boost_velocity[2], (double)tr->n_theta, (double)tr->n_phi};
  const bool switched_off = c->opt.on(OPT_NO_AXIS_BOOST_SEPARABLE);
  if (!switched_off && c->ring_verdict >= 0 && std::memcmp(key, c->ring_key, sizeof key) == 0) {
    if (c->ring_verdict) thetas = c->ring_thetas;
    return c->ring_verdict != 0;
  }
  const bool yes = separable_rotor_grid(tr, thetas, switched_off);
  if (!switched_off) {
    std::memcpy(c->ring_key, key, sizeof key);
    c->ring_verdict = yes ? 1 : 0;
    c->ring_thetas = yes ? thetas : std::vector<double>();
  }
  return yes;
}

// Tables of the separable synthesis, built once per (grid, spin, l range) and kept in the context.  Returns with P.nt = 0
// and P.large = false when the shape is one neither kernel takes.
static int run_synthesis(bms_ctx* c, const SynthesisPlan& P, const double* A, long long lda, long long rows, const double* off, double* Y,
                         long long ldy, const double* scale = nullptr);
// thetas != nullptr: the rings' colatitudes (a boost along the grid's polar axis): tables of their own, kept until a transformation
// with other colatitudes asks for the same shape
static int build_synthesis(bms_ctx* c, int n_theta, int n_phi, int spin, int ell_min, int ell_max, SynthesisPlan& P,
                           const std::vector<double>* thetas = nullptr) {
  const std::array<int, 5> key = {n_theta, n_phi, spin, ell_min, ell_max};
  auto it = c->syn_plans.find(key);
  if (!thetas && it != c->syn_plans.end()) {
    P = it->second;
    return BMS_OK;
  }
  if (thetas) {
    auto ia = c->syn_plans_axis.find(key);
    if (ia != c->syn_plans_axis.end() && ia->second.first == *thetas) {
      P = ia->second.second;
      return BMS_OK;
    }
  }
  P = SynthesisPlan();
  if (!c->n_cu) {
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount;
  }
  std::vector<int> meta;
  int len = 0;
  if (c->opt.on(OPT_NO_SEPARABLE_SYNTHESIS) || !synthesis_split_plan(n_theta, n_phi, ell_min, ell_max, P.g, meta, P.lds, P.nt, len)) P.nt = 0;
  // (behind an axis boost the one-kernel form also keeps the per-pixel scale in LDS)
  if (thetas && P.nt && P.lds + synthesis_split_scale_bytes(n_theta, n_phi) > 160 * 1024) P.nt = 0;
  P.large = !c->opt.on(OPT_NO_SEPARABLE_SYNTHESIS) && !c->opt.on(OPT_NO_LARGE_SYNTHESIS) && large_synthesis_supported(n_theta, n_phi, ell_min, ell_max) != 0;
  P.n_theta = n_theta, P.n_phi = n_phi, P.ell_min = ell_min, P.ell_max = ell_max;
  if (!P.nt && !P.large) return BMS_OK;
  hipStream_t S = c->stream;
  int rc;
  void* vp;
  char nm_[96];
  const int n_modes = LM_total_size(ell_min, ell_max);
  std::vector<double> rot(4 * (size_t)n_theta), one(n_theta, 1.0);
  for (int j = 0; j < n_theta; ++j) {
    const Quat q = from_spherical_coords(thetas ? (*thetas)[j] : M_PI * j / (n_theta - 1), 0.0);
    rot[4 * j] = q.w, rot[4 * j + 1] = q.x, rot[4 * j + 2] = q.y, rot[4 * j + 3] = q.z;
  }
  snprintf(nm_, sizeof nm_, thetas ? "syn_rotb_%d" : "syn_rot_%d", n_theta);
  if ((rc = upload(c, nm_, rot.data(), 8 * rot.size(), &vp))) return rc;
  const double* d_rot = (const double*)vp;
  snprintf(nm_, sizeof nm_, "syn_one_%d", n_theta);
  if ((rc = upload(c, nm_, one.data(), 8 * one.size(), &vp))) return rc;
  const double* d_one = (const double*)vp;
  if (P.nt) {
    snprintf(nm_, sizeof nm_, "syn_meta_%d_%d_%d_%d_%d", n_theta, n_phi, spin, ell_min, ell_max);
    if ((rc = upload(c, nm_, meta.data(), sizeof(int) * meta.size(), &vp))) return rc;
    P.d_meta = (int*)vp;
  }
  double* d_Y;
  if ((rc = dev_buf_t(c, "syn_Y", (size_t)n_theta * n_modes * 2, &d_Y))) return rc;
  HIP_TRY(c, hipMemsetAsync(d_Y, 0, 16 * (size_t)n_theta * n_modes, S));
  TIMED(c, BMS_TAG_SETUP, launch_swsh_values(S, d_rot, n_theta, spin, ell_min, ell_max, d_Y));
  if (thetas)  // (one buffer per cache entry: the key's n_phi is part of the name)
    snprintf(nm_, sizeof nm_, "syn_Tb_%d_%d_%d_%d_%d", n_theta, n_phi, spin, ell_min, ell_max);
  else
    snprintf(nm_, sizeof nm_, "syn_T_%d_%d_%d_%d", n_theta, spin, ell_min, ell_max);
  if ((rc = dev_buf_t(c, nm_, (size_t)n_theta * n_modes, &P.d_T))) return rc;
  TIMED(c, BMS_TAG_SETUP, launch_theta_table(S, d_Y, d_one, n_theta, n_modes, P.d_T));  // weights 1: the plain sLambda values
  HIP_TRY(c, hipStreamSynchronize(S));  // host vectors above go out of scope
  if (!thetas)
    c->syn_plans[key] = P;
  else
    c->syn_plans_axis[key] = std::make_pair(*thetas, P);
  return BMS_OK;
}

// One separable synthesis: A[rows][lda] (complex; n_modes (+ 1 with `off`, always for the one-kernel form) per row) -> Y[rows][ldy]
static int run_synthesis(bms_ctx* c, const SynthesisPlan& P, const double* A, long long lda, long long rows, const double* off, double* Y,
                         long long ldy, const double* scale) {
  hipStream_t S = c->stream;
  if (rows <= 0) return BMS_OK;
  if (P.nt && rows >= 2 && !c->opt.on(OPT_NO_SPLIT_SYNTHESIS)) {
    TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_synthesis_split(S, A, lda, rows, P.g, P.nt, P.d_T, P.d_meta, off, Y, ldy, P.lds, c->n_cu, scale));
  } else if (P.large) {
    double* d_F;
    int rc = dev_buf_t(c, "Fphi", (size_t)rows * (2 * P.ell_max + 1) * large_analysis_jp(P.n_theta) * 2, &d_F);
    if (rc) return rc;
    TIMED(c, BMS_TAG_GEMM_SYNTHESIS, launch_synthesis_large(S, A, lda, rows, P.n_theta, P.n_phi, P.ell_min, P.ell_max, P.d_T, off, d_F, Y, ldy, scale));
  } else
    return fail(c, BMS_ERR_UNSUPPORTED, "internal: no separable synthesis for a chunk of %lld row(s) of this shape", rows);
  return BMS_OK;
}

// Column plan of the grids: 0 = one column per grid pixel, in grid order.  When the analysis can read the columns in any
// order (the fused kernel) the two pole rings are stored once each (1) and, with a boost, whose time skew grows with |u|,
// the columns are also sorted by the skew rate (2).
static int column_plan(const bms_ctx* c, const bms_transformation* tr, int n_out) {
  if (c->opt.on(OPT_NO_COLUMN_SORT) || c->opt.on(OPT_NO_FUSED_ANALYSIS)) return 0;
  if (tr->n_theta < 3 || tr->n_theta * tr->n_phi > pixel_sort_max() || tr->n_theta > MAX_THETA_SEPARABLE ||
      !fused_analysis_supported(tr->n_theta, tr->n_phi, tr->ell_max_out, n_out))
    return 0;
  const double* v = tr->boost_velocity;
  return (v[0] == 0 && v[1] == 0 && v[2] == 0) ? 1 : 2;
}
// On return T.n_pix is the number of COLUMNS (everything downstream is per column); T.n_theta * T.n_phi stays the grid.
static int device_pixel_tables(bms_ctx* c, const bms_transformation* tr, PixelTables& T, int mode, int spin, int cw,
                               const std::vector<cplx>* coef0, const std::vector<cplx>* coef1, const cplx cv[4], DevPixel& D,
                               int plan, hipStream_t PS = nullptr,
                               const std::function<int(hipStream_t, const DevPixel&, int)>& behind_tables = nullptr,
                               const std::function<void()>& while_waiting = nullptr) {
  if (!PS) PS = c->stream;
  init_pixel_tables(tr, T);
  const int n_pix = T.n_pix, lst = tr->ell_max_supertranslation, nst = (lst + 1) * (lst + 1);
  const int n_cols = plan ? n_pix - 2 * (tr->n_phi - 1) : n_pix;
  PixelSpec P = base_pixel_spec(tr, T);
  P.mode = mode;
  P.spin = spin;
  P.conformal_weight = cw;
  if (cv)
    for (int i = 0; i < 4; ++i) P.cv[i] = cv[i];
  // one upload for the (up to three) coefficient sets
  std::vector<cplx> coefs((size_t)3 * nst, cplx{0.0, 0.0});
  std::memcpy(coefs.data(), tr->supertranslation, sizeof(cplx) * nst);
  if (coef0) std::memcpy(coefs.data() + nst, coef0->data(), sizeof(cplx) * nst);
  if (coef1) std::memcpy(coefs.data() + 2 * nst, coef1->data(), sizeof(cplx) * nst);
  cplx* d_coefs;
  int rc = dev_buf_t(c, "pix_coefs", (size_t)3 * nst, &d_coefs);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(d_coefs, coefs.data(), sizeof(cplx) * 3 * nst, hipMemcpyHostToDevice, PS));
  P.st = d_coefs;
  P.c0 = coef0 ? d_coefs + nst : nullptr;
  P.c1 = coef1 ? d_coefs + 2 * nst : nullptr;
  // one device block for all per-pixel outputs: 4 (rotor) + 4 scalars + 4 (off, scale) + 4 (xa, xb) + 6 + 2 doubles per pixel
  double* blk;
  if ((rc = dev_buf_t(c, "pix_block", (size_t)24 * n_pix, &blk))) return rc;
  D.rotors = blk;
  D.k = blk + 4 * (size_t)n_pix;
  D.alpha = D.k + n_pix;
  D.skew_a = D.alpha + n_pix;
  D.skew_b = D.skew_a + n_pix;
  D.col_off = D.skew_b + n_pix;
  D.col_scale = D.col_off + 2 * (size_t)n_pix;
  D.xa = D.col_scale + 2 * (size_t)n_pix;
  D.xb = D.xa + 2 * (size_t)n_pix;
  D.ethk = D.xb + 2 * (size_t)n_pix;  // ABD block aliases nothing: 16 + 2 + 2 + 2 + 1 + 1 = 24
  D.etha = D.col_off;                 // ABD never uses the WM arrays: reuse them
  D.ethetha = D.xa;
  D.ik = D.ethk + 2 * (size_t)n_pix;
  D.ik3 = D.ik + n_pix;
  PixelOut O{};
  O.rotors = D.rotors, O.k = D.k, O.alpha = D.alpha, O.skew_a = D.skew_a, O.skew_b = D.skew_b;
  O.col_off = D.col_off, O.col_scale = D.col_scale, O.xa = D.xa, O.xb = D.xb;
  O.ethk = D.ethk, O.etha = D.etha, O.ethetha = D.ethetha, O.ik = D.ik, O.ik3 = D.ik3;
  int* d_perm = nullptr;
  if (plan) {
    if ((rc = dev_buf_t(c, "pix_perm", (size_t)2 * n_pix, &d_perm))) return rc;
    TIMED_ON(c, PS, BMS_TAG_SETUP, launch_pixel_sort(PS, P, tr->n_theta, tr->n_phi, plan == 2, d_perm, d_perm + n_pix));
    D.col_of_pixel = d_perm + n_pix;
  }
  TIMED_ON(c, PS, BMS_TAG_SETUP, launch_pixel_tables(PS, P, O, n_cols, d_perm));
  // k, alpha, skew_a, skew_b are contiguous (n_pix apart): one copy back, into page-locked memory
  if (c->pix_back_cap < (size_t)4 * n_pix) {
    if (c->pix_back_host) (void)hipHostFree(c->pix_back_host);
    c->pix_back_host = nullptr, c->pix_back_cap = 0;
    HIP_TRY(c, hipHostMalloc((void**)&c->pix_back_host, sizeof(double) * 4 * n_pix, hipHostMallocDefault));
    c->pix_back_cap = (size_t)4 * n_pix;
  }
  const double* back = c->pix_back_host;
  HIP_TRY(c, hipMemcpyAsync(c->pix_back_host, D.k, sizeof(double) * 4 * n_pix, hipMemcpyDeviceToHost, PS));
  if (behind_tables && PS != c->stream) {
    // what needs the device tables only (the synthesis matrix: the rotors) is queued behind them on the same stream: it runs while the
    // main stream still works on the modes; the host waits for the copy alone, the main stream for all of it
    if (!c->ev_tables) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_tables, hipEventDisableTiming));
    if (!c->ev_aux_done) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_aux_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_tables, PS));
    rc = behind_tables(PS, D, n_cols);
    HIP_TRY(c, hipEventRecord(c->ev_aux_done, PS));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_aux_done, 0));
    if (while_waiting) while_waiting();  // host work of the caller that needs nothing from here
    HIP_TRY(c, hipEventSynchronize(c->ev_tables));
    if (rc) return rc;
  } else {
    if (behind_tables && (rc = behind_tables(PS, D, n_cols))) return rc;
    if (while_waiting) while_waiting();
    HIP_TRY(c, hipStreamSynchronize(PS));
  }
  T.n_pix = n_cols;
  T.k.assign(back, back + n_cols);
  T.alpha.assign(back + n_pix, back + n_pix + n_cols);
  T.skew_a.assign(back + 2 * (size_t)n_pix, back + 2 * (size_t)n_pix + n_cols);
  T.skew_b.assign(back + 3 * (size_t)n_pix, back + 3 * (size_t)n_pix + n_cols);
  return BMS_OK;
}

// unit maps: row r of the (zeroed) [n][2 n] matrix gets 1 + 0i in complex column r
__global__ __launch_bounds__(256) void unit_maps_kernel(double* __restrict__ I, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) I[(size_t)r * 2 * n + 2 * r] = 1.0;
}

// pixel-column partition (plan B of SURVEY 8(e)): whole 64-column GEMM tiles [cA, cB) of the column plan
static int column_range(bms_ctx* c, const bms_shard* sh, int n_cols, int& cA, int& cB) {
  cA = 0, cB = n_cols;
  if (!sh || sh->col_parts <= 1) return BMS_OK;
  if (sh->col_part < 0 || sh->col_part >= sh->col_parts) return fail(c, BMS_ERR_INVALID, "column part %d outside [0, %d)", sh->col_part, sh->col_parts);
  const long long n_tiles = (n_cols + 63) / 64;
  cA = (int)std::min<long long>(n_cols, 64 * (n_tiles * sh->col_part / sh->col_parts));
  cB = (int)std::min<long long>(n_cols, 64 * (n_tiles * (sh->col_part + 1) / sh->col_parts));
  return BMS_OK;
}
// The analysis is linear in the grid columns, so a part's contribution is G[:, cA:cB] . At[cA:cB, :] with At = the analysis
// of the n_cols unit maps (row r = modes of "1 in column r"), produced by the same analysis kernel the unsplit path runs.
static int part_analysis_matrix(bms_ctx* c, const AnalysisPlan& A, const char* name, int n_cols, const int* col_of_pixel,
                                double** d_At, long long* ld_at) {
  hipStream_t S = c->stream;
  *ld_at = round_up(2LL * A.n_out, 128);
  const size_t at_rows = (size_t)round_up(n_cols, 8) + 8;
  double* d_I;
  int rc;
  if ((rc = dev_buf_t(c, name, at_rows * *ld_at, d_At))) return rc;
  if ((rc = dev_buf_t(c, "unit_maps", (size_t)n_cols * 2 * n_cols, &d_I))) return rc;
  HIP_TRY(c, hipMemsetAsync(*d_At, 0, sizeof(double) * at_rows * *ld_at, S));
  HIP_TRY(c, hipMemsetAsync(d_I, 0, sizeof(double) * (size_t)n_cols * 2 * n_cols, S));
  hipLaunchKernelGGL(unit_maps_kernel, dim3((n_cols + 255) / 256), dim3(256), 0, S, d_I, n_cols);
  HIP_TRY(c, hipGetLastError());
  return run_analysis(c, A, d_I, n_cols, *d_At, *ld_at, col_of_pixel, 2LL * n_cols);
}

// The shared pipeline: `nf` synthesised fields -> pointwise stage -> spline -> analysis, chunked over time.
// tiles + tile-boundary blocks one launch of the evaluating product works through (the denominator of bms_ctx_get_eval_stats): 64-row
// tiles with a boundary block between neighbours, or overlapping tiles that advance 61 rows
static uint64_t eval_tile_count(long long rows, int n_cols, int step) {
  const uint64_t nbn = (uint64_t)((n_cols + 63) / 64);
  if (step == 61) return (uint64_t)std::max<long long>(0, (rows - 3 + 60) / 61) * nbn;
  const uint64_t nbm = (uint64_t)((rows + 63) / 64);
  return nbm * nbn + (nbm > 0 ? nbm - 1 : 0) * nbn;
}

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
                                   void* data_out, int64_t* n_times_out) {
  return bms_transform_modes_shard(c, in, tr, nullptr, t_out, data_out, n_times_out, nullptr);
}

extern "C" int bms_shard_plan(bms_ctx* c, const double* t, int64_t n, const bms_transformation* tr, int64_t out_i0,
                              int64_t out_i1, int64_t need_rows[2], int64_t window[2]) {
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
}

extern "C" int bms_output_window(bms_ctx* c, const double* t, int64_t n, const bms_transformation* tr, int abd, int64_t window[2]) {
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
}

static int transform_modes_impl(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, const bms_shard* sh, double* t_out,
                                void* data_out, int64_t* n_times_out, int64_t* first_index_out, void* grid_out, bool walk_first = false);

extern "C" int bms_transform_modes_shard(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr,
                                         const bms_shard* sh, double* t_out, void* data_out, int64_t* n_times_out,
                                         int64_t* first_index_out) {
  if (!c) return BMS_ERR_INVALID;
  if (!data_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  return with_smaller_chunks(c, [&] { return transform_modes_impl(c, in, tr, sh, t_out, data_out, n_times_out, first_index_out, nullptr); });
}

// Host arrays in, host arrays out, as a three-stage pipeline over time shards of the OUTPUT range: the upload of shard k + 1
// (its rows + halo, bms_shard_plan), the kernels of shard k and the download of shard k - 1 run on three streams, ordered by
// events; the host thread only enqueues.  A long series in host memory waits for PCIe, not for the kernels (cfg3: 456 MB each
// way at 57 GB/s = 8 ms per direction against 6 ms of kernels): one call does upload -> kernels -> download one after the
// other (26 ms), this does them side by side.  Uploads run at full rate from page-locked memory (bms_host_register /
// bms_host_alloc); from pageable memory the runtime stages them.  data_out: host c16[i_hi - i_lo][n_out] (best page-locked).
// Results are those of the sharded path (equal to the one-call path to rounding).  No psi companions (aux) here.
extern "C" int bms_transform_modes_pipelined(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, int pieces,
                                             double* t_out, void* data_out, int64_t* n_times_out) {
  return bms_transform_modes_pipelined_part(c, in, tr, pieces, 0, pieces < 1 ? 1 : pieces, t_out, data_out, n_times_out);
}

// The same for pieces [piece0, piece1) of the `pieces` the output window is cut into: t_out / data_out are the arrays of the WHOLE
// window (every piece lands at its own place), *n_times_out is the whole window's row count.  One process that owns several GPUs
// deals the pieces of one transformation over one context per device, one host thread each (scri_amd/engine.py, `devices=`): every
// context ships its own rows + halo at upload time, so there is no GPU-to-GPU traffic at all, and the results are
// those of the one-context call with the same `pieces`, bit for bit (a piece's arithmetic depends on its cut only).
extern "C" int bms_transform_modes_pipelined_part(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, int pieces, int piece0,
                                                  int piece1, double* t_out, void* data_out, int64_t* n_times_out) {
  if (!c) return BMS_ERR_INVALID;
  if (!in || !tr || !t_out || !data_out || !n_times_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  if (in->mem != BMS_HOST || in->n_aux != 0) return fail(c, BMS_ERR_INVALID, "the pipelined path takes host data without auxiliary fields");
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
}

// WaveformGrid.from_modes on its own (scri/waveform_grid.py:331-613): the field on the distorted grid at the new time slices,
// c16[N'][n_theta * n_phi] in grid order (no column plan), without the analysis back to modes
extern "C" int bms_modes_to_grid(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, double* t_out, void* grid_out,
                                 int64_t* n_times_out) {
  if (!c) return BMS_ERR_INVALID;
  if (!grid_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  return with_smaller_chunks(c, [&] { return transform_modes_impl(c, in, tr, nullptr, t_out, nullptr, n_times_out, nullptr, grid_out); });
}

// Several series under ONE transformation (the extra trailing data dimensions of scri/waveform_grid.py:299-308, 574-594: every
// trailing index is an independent series on the same time axis).  in->data: c16[n_times][>= n_series * n_modes], series j in columns
// [j n_modes, (j + 1) n_modes) of every row (in->ld the row stride); the psi companions likewise with their own mode counts.
// data_out: c16[n_series][n_times][n_out] (series-major; the first *n_times_out rows of each block are written), or grid_out:
// c16[n_series][n_times][n_theta n_phi] for WaveformGrid.from_modes.  The series cross PCIe once as one block, and the time axis
// with its spline tables, the per-direction tables and the window are set up once and shared (the mechanism of the pipelined call's
// pieces); per series run the kernels only.
extern "C" int bms_transform_modes_series(bms_ctx* c, const bms_wm_input* in, int n_series, const bms_transformation* tr, double* t_out,
                                          void* data_out, void* grid_out, int64_t* n_times_out) {
  if (!c) return BMS_ERR_INVALID;
  if (!in || !tr || !t_out || !n_times_out || (!data_out == !grid_out)) return fail(c, BMS_ERR_INVALID, "NULL argument (exactly one of data_out / grid_out)");
  if (n_series < 1) return fail(c, BMS_ERR_INVALID, "n_series must be positive");
  if (in->ell_min < 0 || in->ell_max < in->ell_min || in->n_aux < 0 || in->n_aux > 4) return fail(c, BMS_ERR_INVALID, "bad ell range or n_aux");
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t n = in->n_times;
  const int n_modes = LM_total_size(in->ell_min, in->ell_max);
  if (in->ld < (int64_t)n_series * n_modes) return fail(c, BMS_ERR_INVALID, "ld = %lld is less than n_series * n_modes = %lld", (long long)in->ld, (long long)n_series * n_modes);
  const int n_out = grid_out ? tr->n_theta * tr->n_phi : LM_total_size(std::abs(in->spin_weight), tr->ell_max_out);
  if (n_out <= 0) return fail(c, BMS_ERR_INVALID, "empty output l range");
  hipStream_t S = c->stream;
  bms_wm_input dev = *in;
  int aux_modes[4] = {0, 0, 0, 0};
  int rc;
  if (in->mem == BMS_HOST) {  // one upload of everything
    double* d = nullptr;
    if ((rc = dev_buf_t(c, "series_in", (size_t)n * in->ld * 2, &d))) return rc;
    HIP_TRY(c, hipMemcpyAsync(d, in->data, (size_t)n * in->ld * 16, hipMemcpyHostToDevice, S));
    dev.data = d;
    dev.mem = BMS_DEVICE;
  }
  for (int i = 0; i < in->n_aux; ++i) {
    if (in->aux_ell_min[i] < 0 || in->aux_ell_max[i] < in->aux_ell_min[i]) return fail(c, BMS_ERR_INVALID, "bad l range of auxiliary field %d", i);
    aux_modes[i] = LM_total_size(in->aux_ell_min[i], in->aux_ell_max[i]);
    if (in->aux_ld[i] < (int64_t)n_series * aux_modes[i]) return fail(c, BMS_ERR_INVALID, "auxiliary field %d: row stride too small for %d series", i, n_series);
    if (in->mem == BMS_HOST) {
      const char* names[4] = {"series_aux0", "series_aux1", "series_aux2", "series_aux3"};
      double* d = nullptr;
      if ((rc = dev_buf_t(c, names[i], (size_t)n * in->aux_ld[i] * 2, &d))) return rc;
      HIP_TRY(c, hipMemcpyAsync(d, in->aux_data[i], (size_t)n * in->aux_ld[i] * 16, hipMemcpyHostToDevice, S));
      dev.aux_data[i] = d;
    }
  }
  double* d_res = (double*)(grid_out ? grid_out : data_out);
  if (in->mem == BMS_HOST)
    if ((rc = dev_buf_t(c, "series_out", (size_t)n_series * std::max<int64_t>(n, 1) * n_out * 2, &d_res))) return rc;
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
    double* out_j = d_res + (size_t)j * std::max<int64_t>(n, 1) * n_out * 2;
    int64_t got = 0;
    rc = with_smaller_chunks(c, [&] {
      return transform_modes_impl(c, &one, tr, nullptr, t_out, grid_out ? nullptr : out_j, &got, nullptr, grid_out ? out_j : nullptr);
    });
    if (rc == BMS_OK && j > 0 && got != n_new) rc = fail(c, BMS_ERR_HIP, "series %d produced %lld rows, series 0 %lld", j, (long long)got, (long long)n_new);
    n_new = got;
  }
  if (rc == BMS_OK && in->mem == BMS_HOST && n_new > 0) {
    char* host = (char*)(grid_out ? grid_out : data_out);
    for (int j = 0; j < n_series; ++j) {
      const hipError_t e = hipMemcpyAsync(host + (size_t)j * n * n_out * 16, d_res + (size_t)j * n * n_out * 2, (size_t)n_new * n_out * 16, hipMemcpyDeviceToHost, S);
      if (e != hipSuccess) {
        rc = fail(c, BMS_ERR_HIP, "download of series %d: %s", j, hipGetErrorString(e));
        break;
      }
    }
  }
  const hipError_t es = hipStreamSynchronize(S);
  if (c->aux) (void)hipStreamSynchronize(c->aux);
  if (rc) return rc;
  if (es != hipSuccess) return fail(c, BMS_ERR_HIP, "bms_transform_modes_series: %s", hipGetErrorString(es));
  *n_times_out = n_new;
  return BMS_OK;
}

static int transform_modes_impl(bms_ctx* c, const bms_wm_input* in, const bms_transformation* tr, const bms_shard* sh, double* t_out,
                                void* data_out, int64_t* n_times_out, int64_t* first_index_out, void* grid_out, bool walk_first) {
  if (!in || !tr || !t_out || !n_times_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
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

// ====================================================================================================== building blocks

extern "C" int bms_ring_colatitudes(const double fr[4], const double v[3], int n_theta, int n_phi, double* thetas_out) {
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
}

extern "C" int bms_rotor_grid(bms_ctx* c, const double fr[4], const double v[3], int n_theta, int n_phi, double* out) {
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
}

extern "C" int bms_conformal_factors(bms_ctx* c, const double v[3], const double* rotors, int64_t n, double* k, void* ethk_over_k,
                                     double* one_over_k, double* one_over_k_cubed) {
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
}

extern "C" int bms_swsh_grid(bms_ctx* c, const double* rotors, int64_t n, int spin, int ell_min, int ell_max, void* Y) {
  if (!rotors || !Y) return BMS_ERR_INVALID;
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
}

extern "C" int bms_map2salm(bms_ctx* c, const void* grid, int mem, int64_t n_maps, int n_theta, int n_phi, int spin,
                            int ell_min, int ell_max, void* modes_out) {
  if (!c || !grid || !modes_out) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (n_theta < 2 || n_phi < 1 || ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad sizes");
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
}

extern "C" int bms_cubic_spline(bms_ctx* c, const double* x, int64_t n, const void* y, int64_t ld, int64_t n_cols, int mem,
                                const double* x_new, int64_t n_new, void* out) {
  if (!c || !x || !y || !x_new || !out) return BMS_ERR_INVALID;
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
}

// scipy CubicSpline(x, y).derivative(k) / .antiderivative(-k) evaluated at x_new (ModesTimeSeries.interpolate with
// derivative_order, .dot / .ddot / .int / .iint: scri/modes_time_series.py:72-126)
extern "C" int bms_spline_derivative(bms_ctx* c, const double* x, int64_t n, const void* y, int64_t ld, int64_t n_cols, int mem,
                                     const double* x_new, int64_t n_new, int order, void* out) {
  if (!c || !x || !y || !x_new || !out) return BMS_ERR_INVALID;
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
}

// Angular velocity of a waveform from its modes (scri/mode_calculations.py:403-432 with LdtVector :46-57 and LLMatrix
// :298-313; data_dot = CubicSpline(t, data).derivative()(t), scri/waveform_base.py:690-691 = the spline's knot slopes).
extern "C" int bms_angular_velocity(bms_ctx* c, const double* t, int64_t n, const void* data, int64_t ld, int ell_min, int ell_max,
                                    int mem, double* ldt_out, double* ll_out, double* omega_out) {
  if (!c || !t || !data) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (n < 4) return fail(c, BMS_ERR_UNSUPPORTED, "the time derivative needs at least 4 time steps, got %lld", (long long)n);
  if (ell_min < 0 || ell_max < ell_min) return fail(c, BMS_ERR_INVALID, "bad ell range");
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
}

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
                                              double tolerance, double* R_out) {
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
}

// spinsfast.salm2map(modes, s, ell_max, n_theta, n_phi): values on the equiangular grid (sf.Modes.grid, used by the
// super-rest-frame iteration, scri/asymptotic_bondi_data/map_to_superrest_frame.py:171,216); modes from l = 0
extern "C" int bms_salm2map(bms_ctx* c, const void* modes, int mem, int64_t n_maps, int spin, int ell_max, int n_theta, int n_phi,
                            void* grid_out) {
  if (!c || !modes || !grid_out) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (ell_max < 0 || n_theta < 2 || n_phi < 1 || std::abs(spin) > 4) return fail(c, BMS_ERR_INVALID, "bad sizes");
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
}

// Mode-space operators of sf.Modes / ModesTimeSeries (eth, ethbar, bar, real, sums of different l ranges, scalar and per-row
// factors) as one map along the mode axis, see kernels_modes.hip.  Tables idx_* / coef_* are host arrays of n_cols entries;
// a, b, out and row_scale live in `mem`.  b may be NULL (one-sided map); out may alias neither input unless every idx is
// the identity.
extern "C" int bms_mode_map(bms_ctx* c, void* out, int64_t ld_out, int64_t n_rows, int n_cols, const void* a, int64_t ld_a,
                            const int32_t* idx_a, const void* coef_a, int conj_a, const void* b, int64_t ld_b,
                            const int32_t* idx_b, const void* coef_b, int conj_b, const double* row_scale, int mem) {
  if (!c || !out || !a || !idx_a || !coef_a) return BMS_ERR_INVALID;
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
}

extern "C" int bms_row_norm(bms_ctx* c, const void* data, int64_t ld, int64_t n_rows, int n_cols, int mem, int take_sqrt, double* out) {
  if (!c || !data || !out) return BMS_ERR_INVALID;
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
}

// ModesTimeSeries.grid_multiply (scri/modes_time_series.py:142-202): both mode sets (l_min = 0) are synthesised on the
// (2W+1) x (2W+1) equiangular grid, multiplied there, and the product (spin s_a + s_b) is analysed up to output_ell_max.
extern "C" int bms_grid_multiply(bms_ctx* c, const void* a, int spin_a, int ell_max_a, const void* b, int spin_b, int ell_max_b,
                                 int mem, int64_t n_times, int working_ell_max, int output_ell_max, void* out) {
  if (!c || !a || !b || !out) return BMS_ERR_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  if (ell_max_a < 0 || ell_max_b < 0 || working_ell_max < 1 || output_ell_max < 0 || output_ell_max > working_ell_max)
    return fail(c, BMS_ERR_INVALID, "bad l ranges (need 0 <= output_ell_max <= working_ell_max)");
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
}

// ====================================================================================================== storage formats
// scri/utilities.py:194-232: XOR differencing of a time series in place (rows of 64-bit words)
extern "C" int bms_xor_timeseries(bms_ctx* c, void* data, int mem, int64_t n_rows, int64_t words_per_row, int reverse) {
  if (!c || !data) return BMS_ERR_INVALID;
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
}

// scri/utilities.py:271-406: the function multishuffle(shuffle_widths, forward) returns, applied to n elements
extern "C" int bms_multishuffle(bms_ctx* c, const void* in, void* out, int mem, int64_t n, const int* widths, int n_widths,
                                int forward) {
  if (!c || !in || !out || !widths) return BMS_ERR_INVALID;
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
}

// scri/utilities.py:235-268: Fletcher-32 over the data viewed as 16-bit words (n_bytes must be even)
extern "C" int bms_fletcher32(bms_ctx* c, const void* data, int mem, int64_t n_bytes, uint32_t* checksum) {
  if (!c || !checksum || (!data && n_bytes)) return BMS_ERR_INVALID;
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
}

// ====================================================================================================== ABD flavour
static int transform_abd_impl(bms_ctx* c, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                              const bms_transformation* tr, const bms_shard* sh, double* u_out, void* raw_out, int64_t* n_times_out,
                              int64_t* first_index_out);

extern "C" int bms_transform_abd_shard(bms_ctx* c, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                                       const bms_transformation* tr, const bms_shard* sh, double* u_out, void* raw_out,
                                       int64_t* n_times_out, int64_t* first_index_out) {
  if (!c) return BMS_ERR_INVALID;
  return with_smaller_chunks(c, [&] { return transform_abd_impl(c, u, raw, mem, n_times, ell_max, tr, sh, u_out, raw_out, n_times_out, first_index_out); });
}

static int transform_abd_impl(bms_ctx* c, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                              const bms_transformation* tr, const bms_shard* sh, double* u_out, void* raw_out, int64_t* n_times_out,
                              int64_t* first_index_out) {
  if (!u || !raw || !tr || !u_out || !raw_out || !n_times_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
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
    // row nm of the spin-2 harmonics (psi0 shares them and stops at row nm - 1) multiplies the solved constant series: -eth eth alpha
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
                                           const bms_transformation* tr, int pieces, double* u_out, void* raw_out, int64_t* n_times_out) {
  return bms_transform_abd_pipelined_part(c, u, raw, n, ell_max, tr, pieces, 0, pieces < 1 ? 1 : pieces, u_out, raw_out, n_times_out);
}

// Pieces [piece0, piece1) of the same plan (see bms_transform_modes_pipelined_part): u_out / raw_out are the arrays of the whole window.
extern "C" int bms_transform_abd_pipelined_part(bms_ctx* c, const double* u, const void* raw, int64_t n, int ell_max, const bms_transformation* tr,
                                                int pieces, int piece0, int piece1, double* u_out, void* raw_out, int64_t* n_times_out) {
  if (!c) return BMS_ERR_INVALID;
  if (!u || !raw || !tr || !u_out || !raw_out || !n_times_out) return fail(c, BMS_ERR_INVALID, "NULL argument");
  HIP_TRY(c, hipSetDevice(c->device));
  bool regular = true;
  int rc = validate_common(c, n, u, tr, 0, n, &regular);
  if (rc) return rc;
  if (!regular) return fail(c, BMS_ERR_UNSUPPORTED, "the time steps vary by more than 1e3 within 48 samples: not sharded");
  if (ell_max < 0 || tr->ell_max_out < 0) return fail(c, BMS_ERR_INVALID, "bad ell_max");
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
}

extern "C" int bms_transform_abd(bms_ctx* c, const double* u, const void* raw, int mem, int64_t n_times, int ell_max,
                                 const bms_transformation* tr, double* u_out, void* raw_out, int64_t* n_times_out) {
  return bms_transform_abd_shard(c, u, raw, mem, n_times, ell_max, tr, nullptr, u_out, raw_out, n_times_out, nullptr);
}
