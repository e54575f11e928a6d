// qocx_diag.h - what the release library leaves out.
//
// libqocx.so (make, __graft_entry__.build()) is the product: no environment variable changes
// which kernels it runs and no knob makes it return wrong numbers. libqocx_diag.so (make diag,
// -DQOCX_DIAG) is the measurement build the scripts under tools/ load: it adds the timing
// experiments whose results are garbage by design (knobs "dbg_skip", "sweep3_dbg", "k1a_dbg"), the
// kernel builds that execute in-kernel clock stamps ("sweep3_stamps", "lindblad_stamps",
// "k1a_stamps") and the QOCX_* environment switches of earlier experiments.
#ifndef QOCX_DIAG_H
#define QOCX_DIAG_H

#include <stdlib.h>

namespace qocx {

#ifdef QOCX_DIAG
constexpr bool kDiagBuild = true;
inline const char* diag_getenv(const char* name) { return getenv(name); }
#else
constexpr bool kDiagBuild = false;
inline const char* diag_getenv(const char*) { return nullptr; }
#endif

}  // namespace qocx

// the debug bit masks of the argument blocks (SweepArgs::dbg, LuArgs::dbg, FactorArgs::skip_q ...):
// constant zero in the product library, so the kernels carry none of those branches
#ifdef QOCX_DIAG
#define QOCX_DBG_BITS(x) (x)
#else
#define QOCX_DBG_BITS(x) 0
#endif

#endif
