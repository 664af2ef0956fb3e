// Shared helpers for libserl_mi355.so (host side): error reporting and HIP call checking.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/serl_mi355.h"

namespace serl {

void set_error(const char* fmt, ...);

#define SERL_HIP(call)                                                                    \
  do {                                                                                    \
    hipError_t _e = (call);                                                               \
    if (_e != hipSuccess) {                                                               \
      serl::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__,    \
                      __LINE__);                                                          \
      return SERL_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

#define SERL_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      serl::set_error(__VA_ARGS__);    \
      return SERL_ERR_INVALID;         \
    }                                  \
  } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace serl
