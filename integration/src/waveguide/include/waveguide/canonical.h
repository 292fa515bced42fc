// Forwarding header in place of src/waveguide/include/waveguide/canonical.h (reference lines 29-176):
// `detail::canonical_impl` and both `canonical` overloads (single band: waveguide.h of the engine; multiple
// bands with constant spacing: setup.h) keep their signatures -- src/combined/src/waveguide_base.cpp:22-43
// calls them as before.
#pragma once

#include "waveguide/waveguide.h"

#include "wayverb_amd/setup.h"
