// pool_driver.cc -- the multi-GPU frame object's enqueue pool (mallie_amd/csrc/mgpu_enqueue_pool.hpp) on a CPU: correctness under
// contention (every member's job runs exactly once per call, error codes and texts reach the caller, spin and park paths both
// taken) and the cost of one hand-off.  Built plain, and with -fsanitize=thread, by tests/test_host_cpu.py.
//   pool_driver stress            -> "pool stress ok ..." (exit 0) or a message (exit 1)
//   pool_driver time <members>    -> hand-off cost per call with a job of ~40 us per member (the launch phase's host time on the GPU
//                                    box): serial loop, pool parking at once (round 5), pool spinning first
#include "../../mallie_amd/csrc/mgpu_enqueue_pool.hpp"

#include <cstring>

static thread_local char g_err[512] = "";
static char *err_buf() { return g_err; }

static void burn_us(long us) {
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() < us * 1000) {
  }
}

static int stress() {
  for (const char *spin : {"0", "50", "2000"}) {
    setenv("MGPU_FRAME_ENQUEUE_SPIN_US", spin, 1);
    const size_t n = 8;
    EnqueuePool pool(n, err_buf);
    std::vector<std::atomic<int>> hits(n);
    for (auto &h : hits) h = 0;
    for (int call = 0; call < 3000; ++call) {
      const int fail_at = call % 7 == 3 ? (call % (int)n) : -1;
      const std::function<int(size_t)> job = [&](size_t m) -> int {
        hits[m].fetch_add(1);
        if ((int)m == fail_at) {
          snprintf(err_buf(), 512, "member %zu refuses call %d", m, call);
          return -4;
        }
        return 0;
      };
      const int rc = pool.run(job);
      if ((fail_at >= 0) != (rc != 0)) return printf("call %d: rc %d, expected %s\n", call, rc, fail_at >= 0 ? "an error" : "0"), 1;
      if (rc) {
        char want[64];
        snprintf(want, sizeof want, "member %d refuses call %d", fail_at, call);
        if (strcmp(err_buf(), want)) return printf("call %d: error text '%s', wanted '%s'\n", call, err_buf(), want), 1;
      }
      if (call % 500 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(atol(spin) >= 2000 ? 5 : 1)); // lets the workers park
    }
    for (size_t m = 0; m < n; ++m)
      if (hits[m].load() != 3000) return printf("spin %s: member %zu ran %d times of 3000\n", spin, m, hits[m].load()), 1;
  }
  printf("pool stress ok: 3 spin windows x 3000 calls x 8 members, every job ran once, errors delivered\n");
  return 0;
}

static double time_calls(EnqueuePool *pool, size_t members, int calls, long job_us, long gap_us) {
  const std::function<int(size_t)> job = [&](size_t) -> int { burn_us(job_us); return 0; };
  double total = 0.0;
  for (int c = 0; c < calls; ++c) {
    const auto t0 = std::chrono::steady_clock::now();
    if (pool) pool->run(job);
    else
      for (size_t m = 0; m < members; ++m) job(m);
    total += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() * 1e-3;
    burn_us(gap_us); // the caller between two render calls (waiting for a frame)
  }
  return total / calls;
}

int main(int argc, char **argv) {
  if (argc >= 2 && !strcmp(argv[1], "stress")) return stress();
  if (argc >= 3 && !strcmp(argv[1], "time")) {
    const size_t members = (size_t)atol(argv[2]);
    const long job_us = argc >= 4 ? atol(argv[3]) : 40, gap_us = argc >= 5 ? atol(argv[4]) : 600;
    printf("members %zu, %ld us of host work per member and call, %ld us between calls, %u hardware threads\n", members, job_us, gap_us,
           std::thread::hardware_concurrency());
    printf("  serial loop on the caller's thread        : %7.1f us per call\n", time_calls(nullptr, members, 2000, job_us, gap_us));
    for (const char *spin : {"0", "2000"}) {
      setenv("MGPU_FRAME_ENQUEUE_SPIN_US", spin, 1);
      EnqueuePool pool(members, err_buf);
      time_calls(&pool, members, 200, job_us, gap_us);
      printf("  pool, workers spin %4s us before parking   : %7.1f us per call\n", spin, time_calls(&pool, members, 2000, job_us, gap_us));
    }
    return 0;
  }
  fprintf(stderr, "usage: pool_driver stress | time <members> [job_us] [gap_us]\n");
  return 2;
}
