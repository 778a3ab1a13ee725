// oem_knobs.cpp -- tuning / test switches.
//
// The product library (liboarfish_em.so) is built from this file WITHOUT OEM_TESTING: knob() then
// returns the built-in default whatever the environment says, so no environment variable can change
// the layout or the kernels of a production store.  The test-only library
// (liboarfish_em_testing.so, -DOEM_TESTING) reads the variable, which is how tests/ and scripts/ force
// the fallback paths and run A/B timings without a rebuild.
#include <cstdlib>

#include "oem_internal.h"

namespace oem {

long knob(const char *name, long dflt)
{
#ifdef OEM_TESTING
    const char *e = getenv(name);
    return (e && *e) ? atol(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

} // namespace oem
