// A second host thread that issues launches (round 5).
//
// At the reference's own shapes (B = 100 / 200, D = 16, H = 32) a training step is ~22 launches of 4 - 40 us of device work
// each, and the step is bound by what ONE host thread spends calling hipLaunchKernel / hipEventRecord / hipStreamWaitEvent
// (~185 us against ~160 us of dependent device work; tools/host_calls.py).  About a third of those calls start work that is
// NOT on the step's chain -- the next batch's index plan (six launches), the look-ahead catch-up of its rows (two), the
// optimizer's window slice (one): side-stream work the launch stream only meets again a step later.  This file hands those
// calls to a worker thread: the caller enqueues a job (the C-ABI call's arguments by value, the events to wait for on the job's
// stream first, the event to record behind it) and goes on queueing the chain; the worker makes the HIP calls.
//
// Ordering contract.  A job's launches are ordered on the DEVICE by its stream and events exactly as if the caller had made
// the calls itself at the moment the worker makes them.  What the host must add: before the caller lets anything wait for
// (hipStreamWaitEvent) or re-record an event a job records or waits for, the job must have been ISSUED --
// score_async_wait(ticket) blocks until it has (a bounded spin: the worker is at most a few tens of microseconds behind).
// A wait bound to an event before its record call, or to a later re-record, is how a stream deadlocks.
//
// One worker per process, started on first use, spinning while jobs keep coming and asleep otherwise.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include "common.h"

namespace {

enum { JOB_PLAN = 1, JOB_CATCHUP_THROUGH = 2, JOB_CATCHUP_ROWS = 3 };
struct Job {
  int kind, device;
  score_config_t cfg; score_state_t st; score_batch_t batch; int32_t n_shards, dedup;
  score_adam_table_t tab; const int32_t* ids; int64_t n_ids, lo, hi; uint32_t step; float alpha;
  hipStream_t stream; hipEvent_t waits[3]; hipEvent_t record;
};
constexpr uint64_t RING = 64;

struct Worker {
  Job ring[RING];
  std::atomic<uint64_t> head{0};      // jobs submitted (written by submitters under `sub`)
  std::atomic<uint64_t> done{0};      // jobs issued (written by the worker)
  std::atomic<int> err{0};            // first failure since the last score_async_wait
  std::atomic<bool> stop{false}, asleep{false}, started{false};
  std::mutex sub, mu;
  std::condition_variable cv;
  std::thread th;
  int cur_device = -1;

  int run(const Job& j) {
    if (j.device != cur_device) {
      if (hipSetDevice(j.device) != hipSuccess) return SCORE_E_BADARG;
      cur_device = j.device;
    }
    for (int i = 0; i < 3; ++i)
      if (j.waits[i]) { hipError_t e = hipStreamWaitEvent(j.stream, j.waits[i], 0); if (e != hipSuccess) return (int)e; }
    int rc = 0;
    if (j.kind == JOB_PLAN) rc = score_index_plan(&j.cfg, &j.st, &j.batch, j.n_shards, j.dedup, j.stream);
    else if (j.kind == JOB_CATCHUP_THROUGH) rc = score_adam_catchup_ids_through(&j.tab, j.ids, j.n_ids, j.step, j.alpha, j.stream);
    else if (j.kind == JOB_CATCHUP_ROWS) rc = score_adam_catchup_rows(&j.tab, j.lo, j.hi, j.step, j.stream);
    else rc = SCORE_E_BADARG;
    // (the record is made even after a failure: somebody is going to wait for this event, and an event that is never recorded
    //  again would let that wait bind to an OLD record -- the failure itself comes back from score_async_wait)
    if (j.record) { hipError_t e = hipEventRecord(j.record, j.stream); if (rc == 0 && e != hipSuccess) rc = (int)e; }
    return rc;
  }

  void loop() {
    uint64_t idle = 0;
    while (!stop.load(std::memory_order_acquire)) {
      const uint64_t d = done.load(std::memory_order_relaxed);
      if (head.load(std::memory_order_acquire) != d) {
        const int rc = run(ring[d % RING]);
        if (rc != 0) { int z = 0; err.compare_exchange_strong(z, rc); }
        done.store(d + 1, std::memory_order_release);
        idle = 0;
        continue;
      }
      if (++idle < 200000) {            // ~a few ms of polling after the last job: a training loop keeps the worker hot
        __builtin_ia32_pause();
        continue;
      }
      std::unique_lock<std::mutex> lk(mu);
      asleep.store(true, std::memory_order_release);
      cv.wait_for(lk, std::chrono::milliseconds(50), [&] {
        return stop.load(std::memory_order_acquire) || head.load(std::memory_order_acquire) != done.load(std::memory_order_relaxed);
      });
      asleep.store(false, std::memory_order_release);
      idle = 0;
    }
  }

  int submit(const Job& j, uint64_t* ticket) {
    std::lock_guard<std::mutex> g(sub);
    if (!started.load(std::memory_order_acquire)) {
      th = std::thread([this] { loop(); });
      started.store(true, std::memory_order_release);
    }
    const uint64_t h = head.load(std::memory_order_relaxed);
    while (h - done.load(std::memory_order_acquire) >= RING) __builtin_ia32_pause();      // (64 jobs behind: never in practice)
    ring[h % RING] = j;
    head.store(h + 1, std::memory_order_release);
    if (asleep.load(std::memory_order_acquire)) { std::lock_guard<std::mutex> lk(mu); cv.notify_one(); }
    if (ticket) *ticket = h + 1;
    return 0;
  }

  int wait(uint64_t ticket) {
    if (ticket == 0) ticket = head.load(std::memory_order_acquire);      // 0: everything submitted so far
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    while (done.load(std::memory_order_acquire) < ticket) {
      __builtin_ia32_pause();
      if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return SCORE_E_BADARG;
    }
    return err.exchange(0);
  }

  ~Worker() {
    stop.store(true, std::memory_order_release);
    { std::lock_guard<std::mutex> lk(mu); cv.notify_one(); }
    if (th.joinable()) th.join();
  }
};
Worker g_worker;

int fill_common(Job* j, void* stream, void* const* wait_events, int n_wait, void* record_event) {
  if (n_wait < 0 || n_wait > 3 || (n_wait > 0 && !wait_events)) return SCORE_E_BADARG;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SCORE_E_BADARG;
  j->device = dev;
  j->stream = (hipStream_t)stream;
  for (int i = 0; i < 3; ++i) j->waits[i] = i < n_wait ? (hipEvent_t)wait_events[i] : nullptr;
  j->record = (hipEvent_t)record_event;
  return 0;
}

}  // namespace

extern "C" int score_async_index_plan(const score_config_t* cfg, const score_state_t* st, const score_batch_t* batch, int32_t n_shards,
                                      int32_t dedup, void* stream, void* const* wait_events, int32_t n_wait, void* record_event,
                                      uint64_t* ticket) {
  if (!cfg || !st || !batch) return SCORE_E_BADARG;
  Job j;
  memset(&j, 0, sizeof(j));
  SCORE_TRY(fill_common(&j, stream, wait_events, n_wait, record_event));
  j.kind = JOB_PLAN; j.cfg = *cfg; j.st = *st; j.batch = *batch; j.n_shards = n_shards; j.dedup = dedup;
  return g_worker.submit(j, ticket);
}

extern "C" int score_async_adam_catchup_ids_through(const score_adam_table_t* t, const int32_t* ids, int64_t n_ids, uint32_t step,
                                                    float alpha, void* stream, void* const* wait_events, int32_t n_wait,
                                                    void* record_event, uint64_t* ticket) {
  if (!t || !ids) return SCORE_E_BADARG;
  Job j;
  memset(&j, 0, sizeof(j));
  SCORE_TRY(fill_common(&j, stream, wait_events, n_wait, record_event));
  j.kind = JOB_CATCHUP_THROUGH; j.tab = *t; j.ids = ids; j.n_ids = n_ids; j.step = step; j.alpha = alpha;
  return g_worker.submit(j, ticket);
}

extern "C" int score_async_adam_catchup_rows(const score_adam_table_t* t, int64_t row_begin, int64_t row_end, uint32_t upto,
                                             void* stream, void* const* wait_events, int32_t n_wait, void* record_event,
                                             uint64_t* ticket) {
  if (!t) return SCORE_E_BADARG;
  Job j;
  memset(&j, 0, sizeof(j));
  SCORE_TRY(fill_common(&j, stream, wait_events, n_wait, record_event));
  j.kind = JOB_CATCHUP_ROWS; j.tab = *t; j.lo = row_begin; j.hi = row_end; j.step = upto;
  return g_worker.submit(j, ticket);
}

extern "C" int score_async_wait(uint64_t ticket) { return g_worker.wait(ticket); }
