// diag.hpp — see below.
#pragma once
#include <cstdlib>

namespace t4a {

// Environment switches.  The production library reads the few that select a documented fallback path or a debugging print
// (first table); the EXPERIMENT switches of four rounds of tuning — exchange knobs, plan overrides, measured-neutral
// alternatives — are compiled in only with -DT4A_DIAG_SWITCHES (T4A_EXTRA_FLAGS of build.py; tools/build_variant_lib.sh): without it
// diag_env() is a constant and the branches behind it fold away.
#ifdef T4A_DIAG_SWITCHES
inline const char* diag_env(const char* name) { return std::getenv(name); }
#else
inline const char* diag_env(const char*) { return nullptr; }
#endif

} // namespace t4a
