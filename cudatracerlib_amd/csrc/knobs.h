// knobs.h — measurement knobs.  The shipped library reads NO tuning variable from the environment: knob_env() returns nullptr unless the library was built
// with -DCTL_MEASUREMENT_KNOBS (python -m cudatracerlib_amd.build --out libctl_knobs.so -DCTL_MEASUREMENT_KNOBS; tools/exp.sh selects it with CTL_AMD_LIB).
// Every knob is the handle of an experiment recorded in DESIGN.md §3 and its default is the measured choice.  Product settings (CTL_CACHE_DIR, CTL_BVH_MODE,
// CTL_LOADER_LENIENT, CTL_SUN_SEED, CTL_VERBOSE) are read with getenv where they are used and each has an API call.
#pragma once
#include <cstdlib>
namespace ctl {
inline const char* knob_env(const char* name) {
#ifdef CTL_MEASUREMENT_KNOBS
    return std::getenv(name);
#else
    (void)name; return nullptr;
#endif
}
}  // namespace ctl
