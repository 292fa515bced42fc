// Forwarding header a wayverb maintainer puts in place of src/waveguide/include/waveguide/waveguide.h
// (reference lines 36-126: the OpenCL run loop) -- see INTEGRATION.md section 2.  `waveguide::run<pre, post>`,
// the step pre-/post-processors and `canonical` then come from the MI355X engine; the ray tracer and
// src/combined are not touched.
#pragma once

#include "core/callback_accumulator.h"
#include "core/cl/common.h"
#include "core/environment.h"
#include "core/exceptions.h"

#include "utilities/range.h"

#define WAYVERB_AMD_HAVE_REFERENCE_CORE  // the five headers above define wayverb::core / util: wayverb_amd adds none
#include "wayverb_amd/cl_mirror.h"
