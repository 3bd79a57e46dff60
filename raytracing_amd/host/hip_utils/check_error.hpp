// hip_utils/check_error.hpp -- fail-fast convention of the C++ host layer.
// Replaces checkCudaErrors / rt::internal::check_cuda_error
// (reference: include/cuda_utils/check_error.cuh:3-9, src/check_error.cu:7-15):
// a non-zero status prints "<backend> error: <code> (<expr>) at file:line" to stderr and
// exits with EXIT_FAILURE.  The C-ABI itself never exits; this macro is where the host
// layer chooses to.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "mrt.h"

#define checkRtErrors(val) rt::internal::check_rt_error((val), #val, __FILE__, __LINE__)

namespace rt {
namespace internal {

inline void check_rt_error(int result, const char *const val, const char *const file, int line)
{
  if (result) {
    std::fprintf(stderr, "MRT error: %d (%s: %s) at %s:%d\n", result, val, mrt_status_string(result), file, line);
    std::exit(EXIT_FAILURE);
  }
}

} // namespace internal
} // namespace rt
