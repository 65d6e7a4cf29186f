// Internal header of the engine (the host side of libscri_amd.so): the context, its work space and plans, the timing and error
// macros, the tables of one transformation, and the helpers the entry-point families share.  The C ABI is include/scri_amd.h; this
// file is not installed.  The engine is split by entry family:
//   engine_context.hip   context, work-space slab, route options, page-locked host memory, timing            (bms_ctx_*, bms_host_*)
//   engine_tables.hip    planners: per-direction tables, output window, shard plan, time-axis checks, analysis / synthesis plans
//   engine_rotate.hip    rotations of the decomposition basis                                               (bms_rotate_*, bms_wigner_D)
//   engine_modes.hip     WaveformModes transform: one call, shard, pipelined, series, grid                   (bms_transform_modes*, ...)
//   engine_abd.hip       AsymptoticBondiData transform                                                       (bms_transform_abd*)
//   engine_blocks.hip    building blocks, series and bit operators               (bms_rotor_grid ... bms_grid_multiply, bms_xor_timeseries ...)
#pragma once
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
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/scri_amd.h"
#include "kernels.h"
#include "pixel_math.h"
#include "wigner.h"

using namespace bms;

#define BMS_INTERNAL __attribute__((visibility("hidden")))

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

BMS_INTERNAL int fail(bms_ctx* c, int code, const char* fmt, ...);

// No C++ exception leaves the library: its callers are C, cgo, ctypes, and an exception that unwinds into them ends the process.  Every
// extern "C" entry that returns a status is a function-try-block closed by BMS_CATCH, which turns what was thrown -- std::bad_alloc
// from the host-side tables, std::system_error from a thread that could not start -- into a status and a message on the context.
BMS_INTERNAL int exception_status(bms_ctx* c) noexcept;
#define BMS_CATCH(ctx) \
  catch (...) { return exception_status(ctx); }

#define HIP_TRY(ctx, expr)                                                                                   \
  do {                                                                                                       \
    hipError_t e__ = (expr);                                                                                 \
    if (e__ != hipSuccess) {                                                                                 \
      if (e__ == hipErrorOutOfMemory) note_alloc_failure(ctx);                                               \
      return fail(ctx, e__ == hipErrorOutOfMemory ? BMS_ERR_NOMEM : BMS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                  hipGetErrorString(e__), __FILE__, __LINE__);                                               \
    }                                                                                                        \
  } while (0)

inline bool valid_mem(int mem) { return mem == BMS_HOST || mem == BMS_DEVICE; }
constexpr int MAX_ELL = 8192;  // 2 (l + 1)^2 fits an int with room; tables of such an l are far beyond any device's memory anyway

BMS_INTERNAL hipError_t create_download_stream(bms_ctx* c);
BMS_INTERNAL int dev_buf(bms_ctx* c, const char* name, size_t bytes, void** out);
template <class T>
inline int dev_buf_t(bms_ctx* c, const char* name, size_t count, T** out) {
  void* p = nullptr;
  int rc = dev_buf(c, name, count * sizeof(T), &p);
  *out = static_cast<T*>(p);
  return rc;
}

// Runs `call` again with half the work space cap while it fails because a device ALLOCATION failed (the buffer that failed was
// released before the attempt, so a smaller chunk finds room).  Not retried: a cap the caller set, and the planning error "the cap
// holds fewer than N rows" -- halving only makes that one worse.  The cap itself is restored after the call; the reduced value that
// worked is only REMEMBERED for a bounded number of calls (below), so one transient shortage (a temporary tensor of the caller) does
// not leave every later call of the context with chunks up to 32x smaller.  If every attempt fails the FIRST message is reported.
template <class F>
inline int with_smaller_chunks(bms_ctx* c, F call) {
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

// the dealing of the one-process multi-device entry points (bms_transform_modes_multi, bms_transform_abd_multi)
template <class PartCall>
inline int run_dealt_over_contexts(bms_ctx* const* ctxs, int n_ctx, int pieces, int64_t* n_times_out, PartCall part) {
  if (!ctxs || n_ctx < 1) return BMS_ERR_INVALID;
  for (int k = 0; k < n_ctx; ++k)
    if (!ctxs[k]) return BMS_ERR_INVALID;
  for (int k = 0; k < n_ctx; ++k)
    for (int j = 0; j < k; ++j)
      if (ctxs[j] == ctxs[k]) return fail(ctxs[0], BMS_ERR_INVALID, "context %d is listed twice: one context serves one host thread at a time", k);
  if (pieces < 1) pieces = 1;
  std::vector<int> rc(n_ctx, BMS_OK);
  std::vector<int64_t> got(n_ctx, -1);
  auto run = [&](int k) noexcept {
    const int p0 = (int)(((long long)pieces * k) / n_ctx), p1 = (int)(((long long)pieces * (k + 1)) / n_ctx);
    try {
      if (p1 > p0) rc[k] = part(ctxs[k], p0, p1, &got[k]);
    } catch (...) {
      rc[k] = exception_status(ctxs[k]);
    }
  };
  std::vector<std::thread> threads;
  threads.reserve(n_ctx);
  int started = 1;
  try {
    for (; started < n_ctx; ++started) threads.emplace_back(run, started);
  } catch (...) {  // a thread that could not start: the contexts without one report it, the others finish their share first
    for (int k = started; k < n_ctx; ++k) rc[k] = exception_status(ctxs[k]);
  }
  run(0);
  for (auto& th : threads) th.join();
  for (int k = 0; k < n_ctx; ++k)
    if (rc[k] != BMS_OK) {
      if (k != 0) {
        const std::string msg = ctxs[k]->err;
        return fail(ctxs[0], rc[k], "context %d (device %d): %s", k, ctxs[k]->device, msg.c_str());
      }
      return rc[k];
    }
  int64_t n_new = -1;
  for (int k = 0; k < n_ctx; ++k) {
    if (got[k] < 0) continue;  // (a context without a shard of its own)
    if (n_new >= 0 && got[k] != n_new)
      return fail(ctxs[0], BMS_ERR_HIP, "the output window has %lld rows on context %d and %lld on another", (long long)got[k], k, (long long)n_new);
    n_new = got[k];
  }
  if (n_times_out) *n_times_out = std::max<int64_t>(n_new, 0);
  return BMS_OK;
}


// ====================================================================================================== tables of one transformation

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

struct FieldPlan {  // one field to synthesise: input modes, its SWSH matrix
  const double* d_data = nullptr;  // device c16[n][ld]
  int64_t ld = 0;
  int ell_min = 0, ell_max = 0, spin = 0;
  double* d_B = nullptr;  // synthesis matrix
  long long ldb = 0;
  int K = 0;  // 2 * n_modes
};

// Time samples a call touches on the device: the rows it holds plus the spline-table warm-up margin.  A shard of an
// 8 x 1e5-step series uploads and checks 1e5 + 128 samples, not 8e5 (the host still sees the global array: window
// search and chunk planning are binary searches on it).
constexpr int64_t TIME_MARGIN = 64;

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

// ---- shared helpers (engine_tables.hip unless noted)
BMS_INTERNAL void build_rotor_grid(const double fr[4], const double v[3], int n_theta, int n_phi, std::vector<Quat>& R);
BMS_INTERNAL void theta_quadrature_weights(int n_theta, std::vector<double>& q);
BMS_INTERNAL int ensure_delta(bms_ctx* c, int lmax, const double** d_delta, const long long** d_off);
BMS_INTERNAL int ensure_delta_mfma(bms_ctx* c, int lmax, const double** d_tab, const long long** d_off);
BMS_INTERNAL int ensure_delta_resident(bms_ctx* c, int ell_min, int ell_max, bool* ok, RotResPlan* P, size_t* lds_bytes,
                                 const double** d_tab, unsigned int** d_counter);
BMS_INTERNAL void init_pixel_tables(const bms_transformation* tr, PixelTables& T);
BMS_INTERNAL PixelSpec base_pixel_spec(const bms_transformation* tr, const PixelTables& T);
BMS_INTERNAL void build_pixel_tables(const bms_transformation* tr, PixelTables& T);
BMS_INTERNAL void output_window(const PixelTables& T, const double* t, int64_t n, int64_t& i_lo, int64_t& i_hi);
BMS_INTERNAL void output_window_abd(const PixelTables& T, const double* u, int64_t n, int64_t& i_lo, int64_t& i_hi);
BMS_INTERNAL void needed_knots(const PixelTables& T, const double* t, int64_t n, int64_t c0, int64_t c1, int64_t& ja, int64_t& jb);
BMS_INTERNAL int build_analysis(bms_ctx* c, const char* tag, int n_theta, int n_phi, int spin, int ell_min_out, int ell_max_out,
                          AnalysisPlan& A);
inline bool analysis_reads_contiguous_rows(const AnalysisPlan& A) { return !A.fused && !A.large && A.separable; }
BMS_INTERNAL int run_analysis(bms_ctx* c, const AnalysisPlan& A, const double* d_G, long long rows, double* d_out, long long ldo,
                        const int* col_of_pixel = nullptr, long long ld_cols = 0);
BMS_INTERNAL int upload(bms_ctx* c, const char* name, const void* host, size_t bytes, void** dev);
BMS_INTERNAL int upload_times(bms_ctx* c, const double* t, int64_t n, int64_t lo, int64_t hi, int64_t j0, int64_t j1, double** d_x,
                        SplineTable** d_tab);
BMS_INTERNAL int upload_times_bspline(bms_ctx* c, const double* t, int64_t n, int64_t lo, int64_t hi, int64_t j0, int64_t j1, double** d_x,
                                BsplineTable** d_tab, BsplineForward** d_fwd);
BMS_INTERNAL int stage_in(bms_ctx* c, const char* name, const void* src, int mem, size_t bytes, const double** dev);
BMS_INTERNAL void time_window(int64_t n, const bms_shard* sh, int64_t& lo, int64_t& hi);
BMS_INTERNAL int validate_transformation(bms_ctx* c, int64_t n, const double* t, const bms_transformation* tr, int64_t n_min);
BMS_INTERNAL int walk_time_axis(bms_ctx* c, const double* t, int64_t lo, int64_t hi, bool* regular);
BMS_INTERNAL int validate_common(bms_ctx* c, int64_t n, const double* t, const bms_transformation* tr, int64_t lo = 0, int64_t hi = -1,
                           bool* regular = nullptr, int64_t n_min = 4);
BMS_INTERNAL int spline_tile_for(const double* x, int64_t n);
BMS_INTERNAL BsplineSpread skew_spread(const PixelTables& T, int cA, int cB, const double* x_host);
BMS_INTERNAL int eval_search_halfwidth(const PixelTables& T, int cA, int cB, const double* x_host, int64_t g0, int64_t g1);
BMS_INTERNAL bool large_synthesis_route(const bms_ctx* c, int n_theta, int n_phi, int ell_min, int ell_max);
BMS_INTERNAL bool axis_boost_pays(const bms_ctx* c, int n_modes, int n_theta, int n_phi);
BMS_INTERNAL bool separable_rotor_grid(const bms_transformation* tr, std::vector<double>& thetas, bool axis_boost_off = false);
BMS_INTERNAL bool separable_rotor_grid(bms_ctx* c, const bms_transformation* tr, std::vector<double>& thetas);
BMS_INTERNAL int build_synthesis(bms_ctx* c, int n_theta, int n_phi, int spin, int ell_min, int ell_max, SynthesisPlan& P,
                           const std::vector<double>* thetas = nullptr);
BMS_INTERNAL int run_synthesis(bms_ctx* c, const SynthesisPlan& P, const double* A, long long lda, long long rows, const double* off, double* Y,
                         long long ldy, const double* scale = nullptr);
BMS_INTERNAL int column_plan(const bms_ctx* c, const bms_transformation* tr, int n_out);
BMS_INTERNAL int device_pixel_tables(bms_ctx* c, const bms_transformation* tr, PixelTables& T, int mode, int spin, int cw,
                               const std::vector<cplx>* coef0, const std::vector<cplx>* coef1, const cplx cv[4], DevPixel& D,
                               int plan, hipStream_t PS = nullptr,
                               const std::function<int(hipStream_t, const DevPixel&, int)>& behind_tables = nullptr,
                               const std::function<void()>& while_waiting = nullptr);
BMS_INTERNAL int column_range(bms_ctx* c, const bms_shard* sh, int n_cols, int& cA, int& cB);
BMS_INTERNAL int part_analysis_matrix(bms_ctx* c, const AnalysisPlan& A, const char* name, int n_cols, const int* col_of_pixel,
                                double** d_At, long long* ld_at);
BMS_INTERNAL uint64_t eval_tile_count(long long rows, int n_cols, int step);
// engine_rotate.hip: in-place rotation of resident or host modes by one rotor (series = false) or one per row; the transformations
// rotate the modes into the frame of the separable synthesis with it (sync_after = false: the caller's stream carries on)
BMS_INTERNAL int rotate_impl(bms_ctx* c, void* data, int mem, int64_t n_times, int64_t ld, int ell_min, int ell_max, const void* spinors, bool series,
                             bool sync_after = true);
