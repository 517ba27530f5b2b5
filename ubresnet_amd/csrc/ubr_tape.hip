// Launch plans: record the kernel sequence of a pass once, replay it from C++ (see ubr_host.h, include/ubresnet_hip.h).
//
// The reference leaves scheduling to PyTorch's autograd engine (one Python-level nn layer call per op,
// models/ub_uresnet.py:88-147); round 1 of this package issued ~560 launches per train step from Python through
// ctypes (10.6 ms of host time per 13.7 ms step).  A tape is the prebuilt plan of one pass for one shape: descriptors are
// validated and tile configurations chosen once, at record time; a replay is a tight loop of hipLaunchKernel calls with
// the fork/join structure of the two-stream backward expressed as event nodes.
#include "ubr_host.h"

static thread_local ubr_tape* g_tape = nullptr;
ubr_tape* ubr_tape_current() { return g_tape; }

extern "C" ubr_tape* ubr_tape_create(void) { return new ubr_tape(); }

extern "C" void ubr_tape_destroy(ubr_tape* t) {
  if (t == nullptr) return;
  if (g_tape == t) g_tape = nullptr;
  for (auto& n : t->nodes)
    if (n.kind != ubr_tape::LAUNCH && n.ev != nullptr) (void)hipEventDestroy(n.ev);
  delete t;
}

extern "C" int ubr_tape_begin(ubr_tape* t, int nstreams, void* const* streams) {
  UBR_CHECK(t != nullptr && streams != nullptr && nstreams >= 1 && nstreams <= UBR_TAPE_MAX_STREAMS, "ubr_tape_begin: bad arguments");
  UBR_CHECK(g_tape == nullptr, "ubr_tape_begin: another tape is recording on this thread");
  UBR_CHECK(t->nodes.empty(), "ubr_tape_begin: tape already holds a recording");
  for (int i = 0; i < nstreams; ++i) {
    for (int j = 0; j < i; ++j) UBR_CHECK(streams[i] != streams[j], "ubr_tape_begin: stream slots %d and %d are the same stream", j, i);
    t->rec[i] = (hipStream_t)streams[i];
  }
  t->nstreams = nstreams; t->recording = true; t->bad = false; t->paused = 0;
  g_tape = t;
  return UBR_OK;
}

extern "C" int ubr_tape_end(ubr_tape* t) {
  UBR_CHECK(t != nullptr && g_tape == t, "ubr_tape_end: this tape is not recording on this thread");
  g_tape = nullptr;
  t->recording = false;
  UBR_CHECK(t->paused == 0, "ubr_tape_end: unbalanced ubr_tape_pause");
  if (t->bad) { ubr_set_error("ubr_tape_end: a launch went to a stream outside the tape's slots; the recording is unusable"); return UBR_EINVAL; }
  return UBR_OK;
}

extern "C" int ubr_tape_pause(ubr_tape* t, int on) {
  UBR_CHECK(t != nullptr && g_tape == t, "ubr_tape_pause: this tape is not recording on this thread");
  t->paused += on ? 1 : -1;
  UBR_CHECK(t->paused >= 0, "ubr_tape_pause: unbalanced resume");
  return UBR_OK;
}

static int new_event(hipEvent_t* ev) {
  hipError_t e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (e != hipSuccess) { ubr_set_error("ubr_tape: hipEventCreate: %s", hipGetErrorString(e)); return UBR_ELAUNCH; }
  return UBR_OK;
}

extern "C" int ubr_tape_fork(ubr_tape* t, int from_slot, int to_slot) {
  UBR_CHECK(t != nullptr && g_tape == t && t->recording, "ubr_tape_fork: this tape is not recording on this thread");
  UBR_CHECK(from_slot >= 0 && from_slot < t->nstreams && to_slot >= 0 && to_slot < t->nstreams && from_slot != to_slot, "ubr_tape_fork: bad slots %d -> %d", from_slot, to_slot);
  if (t->paused) return UBR_OK;
  hipEvent_t ev;
  int rc = new_event(&ev);
  if (rc != UBR_OK) return rc;
  t->nodes.push_back(ubr_tape::Node{ubr_tape::FORK, from_slot, to_slot, nullptr, ev, -1});
  return UBR_OK;
}

extern "C" int ubr_tape_mark(ubr_tape* t, int slot) {
  if (t == nullptr || g_tape != t || !t->recording || slot < 0 || slot >= t->nstreams) { ubr_set_error("ubr_tape_mark: bad arguments"); return UBR_EINVAL; }
  hipEvent_t ev;
  int rc = new_event(&ev);
  if (rc != UBR_OK) return rc;
  t->nodes.push_back(ubr_tape::Node{ubr_tape::MARK, slot, (int)t->marks.size(), nullptr, ev, -1});
  t->marks.push_back(ev);
  return (int)t->marks.size() - 1;
}

extern "C" int ubr_tape_wait_mark(const ubr_tape* t, int mark, void* stream) {
  UBR_CHECK(t != nullptr && mark >= 0 && mark < (int)t->marks.size(), "ubr_tape_wait_mark: bad mark %d", mark);
  hipError_t e = hipStreamWaitEvent((hipStream_t)stream, t->marks[mark], 0);
  if (e != hipSuccess) { ubr_set_error("ubr_tape_wait_mark: %s", hipGetErrorString(e)); return UBR_ELAUNCH; }
  return UBR_OK;
}

extern "C" int ubr_tape_size(const ubr_tape* t) { return t == nullptr ? 0 : (int)t->nodes.size(); }

extern "C" int ubr_tape_replay(const ubr_tape* t, int nstreams, void* const* streams) {
  UBR_CHECK(t != nullptr && !t->recording && !t->bad, "ubr_tape_replay: tape is empty, recording or unusable");
  UBR_CHECK(streams != nullptr && nstreams == t->nstreams, "ubr_tape_replay: expected %d stream(s)", t ? t->nstreams : 0);
  for (int i = 0; i < nstreams; ++i)
    for (int j = 0; j < i; ++j) UBR_CHECK(streams[i] != streams[j], "ubr_tape_replay: stream slots %d and %d are the same stream", j, i);
  for (const auto& n : t->nodes) {
    hipStream_t s = (hipStream_t)streams[n.slot];
    if (n.kind == ubr_tape::LAUNCH) {
      n.fn(s);
    } else if (n.kind == ubr_tape::FORK) {
      hipError_t e = hipEventRecord(n.ev, s);
      if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)streams[n.slot2], n.ev, 0);
      if (e != hipSuccess) { ubr_set_error("ubr_tape_replay: fork: %s", hipGetErrorString(e)); return UBR_ELAUNCH; }
    } else {
      hipError_t e = hipEventRecord(n.ev, s);
      if (e != hipSuccess) { ubr_set_error("ubr_tape_replay: mark: %s", hipGetErrorString(e)); return UBR_ELAUNCH; }
    }
  }
  UBR_LAUNCH_CHECK("ubr_tape_replay");
  return UBR_OK;
}

extern "C" int ubr_tape_set_label(ubr_tape* t, int label) {
  UBR_CHECK(t != nullptr && g_tape == t && t->recording, "ubr_tape_set_label: this tape is not recording on this thread");
  t->cur_label = label;
  return UBR_OK;
}

// Replay with a pair of timing events around every launch, on the launch's own stream: per-launch durations under the same
// two-stream overlap as an ordinary replay (bench.py's breakdown; the Python-scheduled path paces the streams differently).
// Synchronises the tape's streams before returning.  ms[i] / label[i] for node i; label -2 marks fork / mark nodes.
extern "C" int ubr_tape_replay_timed(const ubr_tape* t, int nstreams, void* const* streams, float* ms, int32_t* label, int cap) {
  UBR_CHECK(t != nullptr && !t->recording && !t->bad, "ubr_tape_replay_timed: tape is empty, recording or unusable");
  UBR_CHECK(streams != nullptr && nstreams == t->nstreams && ms != nullptr && label != nullptr, "ubr_tape_replay_timed: bad arguments");
  const int n = (int)t->nodes.size();
  UBR_CHECK(cap >= n, "ubr_tape_replay_timed: %d nodes, room for %d", n, cap);
  std::vector<hipEvent_t> e0(n, nullptr), e1(n, nullptr);
  int rc = UBR_OK;
  for (int i = 0; i < n && rc == UBR_OK; ++i) {
    const auto& nd = t->nodes[i];
    hipStream_t s = (hipStream_t)streams[nd.slot];
    hipError_t e = hipSuccess;
    if (nd.kind == ubr_tape::LAUNCH) {
      e = hipEventCreate(&e0[i]);
      if (e == hipSuccess) e = hipEventCreate(&e1[i]);
      if (e == hipSuccess) e = hipEventRecord(e0[i], s);
      if (e == hipSuccess) { nd.fn(s); e = hipEventRecord(e1[i], s); }
      label[i] = nd.label;
    } else if (nd.kind == ubr_tape::FORK) {
      e = hipEventRecord(nd.ev, s);
      if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)streams[nd.slot2], nd.ev, 0);
      label[i] = -2; ms[i] = 0.f;
    } else {
      e = hipEventRecord(nd.ev, s);
      label[i] = -2; ms[i] = 0.f;
    }
    if (e != hipSuccess) { ubr_set_error("ubr_tape_replay_timed: %s", hipGetErrorString(e)); rc = UBR_ELAUNCH; }
  }
  for (int i = 0; i < nstreams; ++i) (void)hipStreamSynchronize((hipStream_t)streams[i]);
  for (int i = 0; i < n; ++i) {
    if (e0[i] != nullptr && e1[i] != nullptr && rc == UBR_OK) {
      float v = 0.f;
      if (hipEventElapsedTime(&v, e0[i], e1[i]) != hipSuccess) v = 0.f;
      ms[i] = v;
    }
    if (e0[i] != nullptr) (void)hipEventDestroy(e0[i]);
    if (e1[i] != nullptr) (void)hipEventDestroy(e1[i]);
  }
  if (rc != UBR_OK) return rc;
  UBR_LAUNCH_CHECK("ubr_tape_replay_timed");
  return UBR_OK;
}
