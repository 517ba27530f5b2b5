// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <functional>
#include <tuple>
#include <type_traits>
#include <vector>
#include "../../include/ubresnet_hip.h"

extern "C" void ubr_set_error(const char* fmt, ...);

#define UBR_CHECK(cond, ...)                      \
  do {                                            \
    if (!(cond)) {                                \
      ubr_set_error(__VA_ARGS__);                 \
      return UBR_EINVAL;                          \
    }                                             \
  } while (0)

#define UBR_LAUNCH_CHECK(name)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      ubr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return UBR_ELAUNCH;                                                        \
    }                                                                            \
  } while (0)

static inline int ubr_esize(int dtype) { return dtype == UBR_F32 ? 4 : 2; }
static inline int ubr_cpu(int dtype) { return dtype == UBR_F32 ? 4 : 8; }
static inline bool ubr_dtype_ok(int dtype) { return dtype == UBR_F32 || dtype == UBR_BF16 || dtype == UBR_F16; }
static inline int ubr_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline bool ubr_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// Launch tape (include/ubresnet_hip.h, "Launch plans"): every kernel launch of this library goes through ubr_launch.
// While a tape is recording on the calling thread the launch is executed AND appended to the tape as a closure over
// the kernel pointer, its launch geometry and its by-value arguments; ubr_tape_replay re-issues the closures from C++
// (one call per pass instead of ~560 Python -> ctypes -> validate -> launch round trips per train step).
// ---------------------------------------------------------------------------------------------
struct ubr_tape {
  enum Kind { LAUNCH = 0, FORK = 1, MARK = 2 };
  struct Node {
    int kind, slot, slot2;
    std::function<void(hipStream_t)> fn;    // LAUNCH (kernels and memsets)
    hipEvent_t ev;                          // FORK / MARK
    int label;                              // LAUNCH: the host's tag for this launch (ubr_tape_set_label), -1 = none
  };
  std::vector<Node> nodes;
  std::vector<hipEvent_t> marks;            // MARK events by id
  hipStream_t rec[UBR_TAPE_MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  int nstreams = 0;
  bool recording = false, bad = false;
  int paused = 0;
  int cur_label = -1;
  int slot_of(hipStream_t st) const {
    for (int i = 0; i < nstreams; ++i) if (rec[i] == st) return i;
    return -1;
  }
  void push(hipStream_t st, std::function<void(hipStream_t)> fn) {
    const int s = slot_of(st);
    if (s < 0) { bad = true; return; }      // a launch on a stream the tape does not know: replay would be wrong
    nodes.push_back(Node{LAUNCH, s, -1, std::move(fn), nullptr, cur_label});
  }
};
ubr_tape* ubr_tape_current();                // the tape recording on this thread (nullptr: none); ubr_tape.hip

template <typename... KA, typename... A>
static inline void ubr_launch(void (*fn)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t st, A&&... a) {
  fn<<<grid, block, lds, st>>>(a...);
  ubr_tape* t = ubr_tape_current();
  if (t != nullptr && t->recording && t->paused == 0) {
    std::tuple<std::decay_t<KA>...> args(a...);
    t->push(st, [fn, grid, block, lds, args](hipStream_t s) {
      std::apply([&](const auto&... x) { fn<<<grid, block, lds, s>>>(x...); }, args);
    });
  }
}
