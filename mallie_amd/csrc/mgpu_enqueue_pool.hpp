// mgpu_enqueue_pool.hpp -- host-side worker threads of the multi-GPU frame object (mgpu_frame.hip).  Plain C++ (no HIP): the
// hand-off is timed and run under TSan on a CPU by tests/cpp/pool_driver.cc.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// Worker threads for a frame object that drives several GPUs from one process (MgpuFrame::pool): one per member but the first,
// whose launch phase the calling thread does itself.  run() hands every worker the same job (called with a member index) and
// returns when all of them have finished: the first non-zero return code in member order, and that worker's error text in the caller's thread-local
// message buffer.
// Hand-off: a render call's launch phase is ~40 us of host time per member, and a condition-variable wake alone costs 10-30 us on
// an idle core -- so nothing on the fast path sleeps.  A worker SPINS on its sequence number for kSpinUs after its last job (render
// calls of a running renderer follow each other within a frame time) and only then parks on its condition variable; run() pays the
// futex wake only for a worker that has parked, and itself spins (then yields) for the workers' completion.
// (MGPU_FRAME_ENQUEUE_SPIN_US: the spin window, default 2000; 0 = park at once, the round-5 behaviour.)
struct EnqueuePool {
  char *(*thread_error)() = nullptr; // the calling thread's 512-byte message buffer (thread-local in the user)
  struct alignas(64) Worker {
    std::thread th;
    std::atomic<unsigned long long> posted{0}, finished{0}; // jobs handed over / completed (sequence numbers)
    std::atomic<bool> parked{false}, quit{false};
    std::mutex mu;              // guards the parked wait only
    std::condition_variable cv;
    const std::function<int(size_t)> *job = nullptr; // written before `posted` moves, read after
    int rc = 0; // MGPU_OK
    char err[512] = "";
  };
  std::vector<Worker *> workers; // worker k serves member k + 1
  long spin_us = 2000;
  static void relax() { __builtin_ia32_pause(); }
  EnqueuePool(size_t members, char *(*thread_error_buffer)()) : thread_error(thread_error_buffer) {
    if (const char *e = getenv("MGPU_FRAME_ENQUEUE_SPIN_US")) spin_us = atol(e) < 0 ? 0 : atol(e);
    for (size_t i = 1; i < members; ++i) {
      Worker *w = new Worker();
      workers.push_back(w);
      try {
        start(w, i);
      } catch (...) {
        stop_all();
        throw;
      }
    }
  }
  void start(Worker *w, size_t member) {
    const long spin = spin_us;
    char *(*terr)() = thread_error;
    w->th = std::thread([w, member, spin, terr] {
      unsigned long long seen = 0;
      for (;;) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned n = 0;
        while (w->posted.load(std::memory_order_acquire) == seen && !w->quit.load(std::memory_order_acquire)) {
          if ((++n & 63u) == 0 || spin == 0) {
            if (spin == 0 || std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin) {
              // park.  `parked` is published (seq_cst) BEFORE the predicate is read, and run() moves `posted` (seq_cst) BEFORE it reads
              // `parked`: one of the two sees the other, so a job is never posted to a sleeper that nobody wakes.
              std::unique_lock<std::mutex> lk(w->mu);
              w->parked.store(true);
              w->cv.wait(lk, [&] { return w->posted.load() != seen || w->quit.load(); });
              w->parked.store(false);
              break;
            }
          }
          relax();
        }
        if (w->quit.load(std::memory_order_acquire)) return;
        seen = w->posted.load(std::memory_order_acquire);
        const int rc = (*w->job)(member);
        w->rc = rc;
        if (rc) snprintf(w->err, sizeof(w->err), "%s", terr()); // this thread's message, for the caller's thread
        w->finished.store(seen, std::memory_order_release);
      }
    });
  }
  void stop_all() {
    for (Worker *w : workers) {
      w->quit.store(true);
      {
        std::lock_guard<std::mutex> lk(w->mu);
      }
      w->cv.notify_all();
      if (w->th.joinable()) w->th.join();
      delete w;
    }
    workers.clear();
  }
  ~EnqueuePool() { stop_all(); }
  int run(const std::function<int(size_t)> &job) {
    for (Worker *w : workers) {
      w->job = &job;
      w->posted.fetch_add(1); // seq_cst: ordered against the worker's `parked` store (see start())
      if (w->parked.load()) {
        { std::lock_guard<std::mutex> lk(w->mu); } // the worker is inside cv.wait (it holds mu from the store to the wait)
        w->cv.notify_one();
      }
    }
    int rc = job(0); // the caller's thread is member 0's worker
    for (Worker *w : workers) {
      const unsigned long long want = w->posted.load(std::memory_order_relaxed);
      unsigned n = 0;
      while (w->finished.load(std::memory_order_acquire) != want) {
        if (++n > 4096) std::this_thread::yield(); // a launch that blocks (full queue) must not keep a core spinning hot
        else relax();
      }
      if (w->rc && !rc) {
        rc = w->rc;
        snprintf(thread_error(), 512, "%s", w->err);
      }
    }
    return rc;
  }
};

