// Switches of the library, in two classes.
//
// Route options (bms::RouteOptions, one set PER CONTEXT)
//     Choices between routes that produce the SAME results (to rounding): the A/B switches DESIGN.md section 1 describes, which the parity tests use
//     to reach every route.  A context reads the SCRI_AMD_<NAME> environment variables ONCE, in bms_ctx_create, as its defaults; after
//     that only bms_ctx_set_option / bms_ctx_get_option (include/scri_amd.h) touch them.  No call path reads the environment, so two
//     contexts of one process can run different routes side by side and a setenv in one thread cannot race a call in another.
// BMS_PROBE_ENV("SCRI_AMD_...")
//     Probe switches: knock-outs (results WRONG, timing only), traces that allocate and block the host inside a launcher, guards
//     switched off, and tuning knobs whose values are not validated.  They exist only in a build with -DSCRI_AMD_PROBES (`make
//     PROBES=1` -> libscri_amd_probes.so, which the scripts under tools/probes load through SCRI_AMD_LIB_PATH); in the default library
//     the macro is a null pointer constant and the names are not even in the binary (tests/test_abi.py checks `strings`).
#pragma once
#include <cstdlib>
#include <cstring>

namespace bms {

// One X-macro entry per option (enumerator and name); flags are 0 / 1; GEMM_EVAL_STEP carries 0 (automatic) / 61 / 64 and AXIS_BOOST_MIN_WORK a count of multiply-adds
// (-1 = the built-in threshold, 0 = the axis-boost route whenever it applies)
#define BMS_ROUTE_OPTIONS(X)                                   \
  X(TRACE, "TRACE")                                            \
  X(ROTATE_VALU, "ROTATE_VALU")                                \
  X(ROTATE_STAGED, "ROTATE_STAGED")                            \
  X(NO_FUSED_ANALYSIS, "NO_FUSED_ANALYSIS")                    \
  X(NO_LARGE_ANALYSIS, "NO_LARGE_ANALYSIS")                    \
  X(NO_SPLIT_ANALYSIS, "NO_SPLIT_ANALYSIS")                    \
  X(NO_PLAN_CACHE, "NO_PLAN_CACHE")                            \
  X(AXIS_BOOST_MIN_WORK, "AXIS_BOOST_MIN_WORK")                \
  X(NO_AXIS_BOOST_SEPARABLE, "NO_AXIS_BOOST_SEPARABLE")        \
  X(NO_SPLIT_SYNTHESIS, "NO_SPLIT_SYNTHESIS")                  \
  X(NO_COLUMN_SORT, "NO_COLUMN_SORT")                          \
  X(NO_BSPLINE, "NO_BSPLINE")                                  \
  X(WALK_FIRST, "WALK_FIRST")                                  \
  X(NO_SMALL_DENSE, "NO_SMALL_DENSE")                          \
  X(NO_SEPARABLE_SYNTHESIS, "NO_SEPARABLE_SYNTHESIS")          \
  X(NO_LARGE_SYNTHESIS, "NO_LARGE_SYNTHESIS")                  \
  X(NO_GEMM_EVAL, "NO_GEMM_EVAL")                              \
  X(SYNTHESIS_EVAL, "SYNTHESIS_EVAL")                          \
  X(NO_SYNTHESIS_EVAL, "NO_SYNTHESIS_EVAL")                    \
  X(TWO_SWEEPS, "TWO_SWEEPS")                                  \
  X(GEMM_EVAL_STEP, "GEMM_EVAL_STEP")                          \
  X(GRID_MULTIPLY_FULL_GRID, "GRID_MULTIPLY_FULL_GRID")        \
  X(NO_ABD_SIGMA_EVAL, "NO_ABD_SIGMA_EVAL")                    \
  X(NO_FUSED_ABD_MIX, "NO_FUSED_ABD_MIX")                      \
  X(NO_ROTATE_PIPELINE, "NO_ROTATE_PIPELINE")

enum RouteOpt : int {
#define X(e, n) OPT_##e,
  BMS_ROUTE_OPTIONS(X)
#undef X
      OPT_COUNT
};

inline const char* route_option_name(int i) {
  static const char* const names[OPT_COUNT] = {
#define X(e, n) n,
      BMS_ROUTE_OPTIONS(X)
#undef X
  };
  return (i >= 0 && i < OPT_COUNT) ? names[i] : nullptr;
}

// "NO_GEMM_EVAL" or "SCRI_AMD_NO_GEMM_EVAL" -> index, -1 if there is no such option
inline int route_option_index(const char* name) {
  if (!name) return -1;
  if (std::strncmp(name, "SCRI_AMD_", 9) == 0) name += 9;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (std::strcmp(name, route_option_name(i)) == 0) return i;
  return -1;
}

struct RouteOptions {
  long long v[OPT_COUNT] = {0};
  RouteOptions() { v[OPT_AXIS_BOOST_MIN_WORK] = -1; }
  bool on(RouteOpt o) const { return v[o] != 0; }
  // The context's defaults, read from the environment ONCE (bms_ctx_create is the only caller): a flag is on when its variable is set to
  // anything but "0" or the empty string; the numeric options take the number.
  void read_environment() {
    for (int i = 0; i < OPT_COUNT; ++i) {
      char full[64] = "SCRI_AMD_";
      std::strncat(full, route_option_name(i), sizeof full - 10);
      const char* e = std::getenv(full);
      if (!e) continue;
      if (i == OPT_AXIS_BOOST_MIN_WORK) {
        v[i] = std::atoll(e) > 0 ? std::atoll(e) : 0;
      } else if (i == OPT_GEMM_EVAL_STEP) {
        v[i] = std::atoll(e);
      } else {
        v[i] = (e[0] == 0 || std::strcmp(e, "0") == 0) ? 0 : 1;
      }
    }
  }
};

}  // namespace bms

#ifdef SCRI_AMD_PROBES
#define BMS_PROBE_ENV(name) std::getenv(name)
#define BMS_PROBES 1
#else
#define BMS_PROBE_ENV(name) (static_cast<const char*>(nullptr))
#define BMS_PROBES 0
#endif
