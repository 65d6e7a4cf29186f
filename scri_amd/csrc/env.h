// Environment switches of the library, in two classes.
//
// route_env("SCRI_AMD_...")      run-time switches between routes that produce the SAME results (to rounding): the A/B switches
//                                DESIGN.md section 7b lists, which the parity tests use to reach every route.  Read per call.
// BMS_PROBE_ENV("SCRI_AMD_...")  probe switches: knock-outs (results WRONG, timing only), traces that allocate and block the host
//                                inside a launcher, guards switched off, and tuning knobs whose values are not validated.  They exist
//                                only in a build with -DSCRI_AMD_PROBES (`make PROBES=1` -> libscri_amd_probes.so, which the scripts
//                                under tools/probes load through SCRI_AMD_LIB_PATH); in the default library the macro is a null
//                                pointer constant and the names are not even in the binary (tests/test_abi.py checks `strings`).
#pragma once
#include <cstdlib>

namespace bms {
inline const char* route_env(const char* name) { return std::getenv(name); }
}  // namespace bms

#ifdef SCRI_AMD_PROBES
#define BMS_PROBE_ENV(name) std::getenv(name)
#define BMS_PROBES 1
#else
#define BMS_PROBE_ENV(name) (static_cast<const char*>(nullptr))
#define BMS_PROBES 0
#endif
