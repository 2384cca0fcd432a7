// timers.h -- per-stage device timers behind the C ABI (chebhip_timers_*, include/chebhip.h; SURVEY 5.1: the
// reference has no timers at all).  A StageTimer brackets one callback with a hipEvent pair on the callback's own
// stream; nothing is synchronised until the counters are read.  Disabled (the default) it costs one load and branch.
#pragma once
#include <hip/hip_runtime.h>

namespace chebhip {
struct StageTimer {
  int id; hipStream_t st; hipEvent_t e0 = nullptr; bool on;
  StageTimer(int stage, void *stream);
  ~StageTimer();
};
}  // namespace chebhip
