// Optional per-launch HIP-event timing (off by default): bench.py turns it on to measure each hot
// kernel's average launch duration on the stream it runs on, live, inside the timed region.
#pragma once
#include "common.h"

namespace serl {
struct ProfScope {
  int slot = -1;
  hipStream_t stream;
  ProfScope(const char* name, hipStream_t s);
  ~ProfScope();
};
}  // namespace serl
