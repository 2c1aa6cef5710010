// stress_driver.cc -- TEST INFRASTRUCTURE: hammers the threaded host side of libmallie_mgpu.so (mgpu_api.hip: the resident trace
// server's mailbox, the submission queue of one-ray callers, render-ahead, frames on several streams, scene create / destroy)
// from many threads at once, for runs under AddressSanitizer / ThreadSanitizer builds of the library (tools/sanitize_gpu.sh).
// It is also a correctness test: every one-ray call must return the record of the batched call, every frame the bytes of the
// first frame rendered with the same arguments.
//
//   stress_driver [threads=16] [rays_per_thread=1500] [frames=12]        exit code 0 = consistent
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/mgpu.h"

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static double rnd() { // [0, 1)
  lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(lcg_state >> 11) * (1.0 / 9007199254740992.0);
}

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    int rc_ = (call);                                                                                \
    if (rc_ != MGPU_OK) {                                                                            \
      fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, mgpu_last_error()); \
      exit(2);                                                                                       \
    }                                                                                                \
  } while (0)

struct Mesh {
  std::vector<double> verts;
  std::vector<uint32_t> faces;
  MgpuNode *nodes = nullptr;
  uint32_t *indices = nullptr;
  size_t nn = 0;
};

static void make_mesh(Mesh &m, int nt) { // a soup of small triangles in [-4, 4]^3 plus two large ones
  for (int k = 0; k < nt; ++k) {
    const double c[3] = {8 * rnd() - 4, 8 * rnd() - 4, 8 * rnd() - 4};
    for (int v = 0; v < 3; ++v)
      for (int a = 0; a < 3; ++a) m.verts.push_back(k < 2 ? 12 * rnd() - 6 : c[a] + 0.6 * rnd() - 0.3);
    for (int v = 0; v < 3; ++v) m.faces.push_back((uint32_t)(3 * k + v));
  }
  int stats[3];
  CHECK(mgpu_bvh_build(m.verts.data(), m.verts.size() / 3, m.faces.data(), m.faces.size() / 3, 0.2, 16, 256, 64, &m.nodes, &m.nn, &m.indices, stats));
}

static MgpuScene *make_scene(const Mesh &m) {
  MgpuScene *s = nullptr;
  CHECK(mgpu_scene_create(m.verts.data(), m.verts.size() / 3, m.faces.data(), m.faces.size() / 3, nullptr, nullptr, nullptr, m.nodes, m.nn,
                          m.indices, nullptr, 0, 0, &s));
  return s;
}

int main(int argc, char **argv) {
  const int n_threads = argc > 1 ? atoi(argv[1]) : 16, per_thread = argc > 2 ? atoi(argv[2]) : 1500, n_frames = argc > 3 ? atoi(argv[3]) : 12;
  if (mgpu_device_count() < 1) {
    fprintf(stderr, "no HIP device\n");
    return 3;
  }
  Mesh mesh, small;
  make_mesh(mesh, 1500);
  make_mesh(small, 40);
  MgpuScene *sc = make_scene(mesh);

  // rays: from a shell around the soup towards it
  const size_t n_rays = (size_t)n_threads * (size_t)per_thread;
  std::vector<MgpuRay> rays(n_rays);
  memset(rays.data(), 0, rays.size() * sizeof(MgpuRay));
  for (size_t i = 0; i < n_rays; ++i) {
    double o[3], t[3], len = 0;
    for (int a = 0; a < 3; ++a) {
      o[a] = 20 * rnd() - 10;
      t[a] = 6 * rnd() - 3;
    }
    for (int a = 0; a < 3; ++a) len += (t[a] - o[a]) * (t[a] - o[a]);
    len = len > 0 ? 1.0 / __builtin_sqrt(len) : 1.0;
    for (int a = 0; a < 3; ++a) {
      rays[i].org[a] = o[a];
      rays[i].dir[a] = (t[a] - o[a]) * len;
    }
  }
  std::vector<MgpuIntersection> want(n_rays), got(n_rays);
  std::vector<uint8_t> want_hit(n_rays), got_hit(n_rays);
  CHECK(mgpu_trace(sc, rays.data(), n_rays, want.data(), want_hit.data(), nullptr));

  // reference frames (hash-seeded: a frame depends on its arguments alone)
  const int W = 200, H = 120, mpl = 5, passes = 2;
  const double eye[3] = {0, 0, 16}, lookat[3] = {0, 0, 0}, up[3] = {0, 1, 0}, quat[4] = {0, 0, 0, 0};
  double frame[12];
  CHECK(mgpu_camera_frame(eye, lookat, up, quat, 45.0, W, H, frame));
  std::vector<std::vector<float>> ref_frames(n_frames, std::vector<float>((size_t)3 * W * H));
  for (int k = 0; k < n_frames; ++k)
    CHECK(mgpu_render(sc, frame, frame + 3, frame + 6, frame + 9, W, H, 0, 0, W, H, mpl, passes, nullptr, MGPU_RNG_HASH, nullptr, 7, (uint32_t)(k * passes),
                      ref_frames[k].data(), nullptr, nullptr));

  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  // (1) one-ray callers, the reference's pattern of Scene::Trace from every OpenMP thread: resident server / submission queue
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&, t]() {
      for (int k = 0; k < per_thread; ++k) {
        const size_t i = (size_t)t + (size_t)k * (size_t)n_threads;
        if (mgpu_trace(sc, &rays[i], 1, &got[i], &got_hit[i], nullptr) != MGPU_OK) bad++;
      }
    });
  // (2) a progressive renderer on the same scene while they run: render-ahead on, the frames must be the reference frames
  th.emplace_back([&]() {
    std::vector<float> img((size_t)3 * W * H);
    if (mgpu_scene_set_render_ahead(sc, 1) != MGPU_OK) bad++;
    for (int k = 0; k < n_frames; ++k) {
      if (mgpu_render(sc, frame, frame + 3, frame + 6, frame + 9, W, H, 0, 0, W, H, mpl, passes, nullptr, MGPU_RNG_HASH, nullptr, 7, (uint32_t)(k * passes),
                      img.data(), nullptr, nullptr) != MGPU_OK)
        bad++;
      else if (memcmp(img.data(), ref_frames[k].data(), img.size() * sizeof(float)) != 0)
        bad++;
    }
    if (mgpu_scene_set_render_ahead(sc, 0) != MGPU_OK) bad++;
  });
  // (3) other scenes coming and going on the same device (allocation, layout kernels, destruction with a server possibly alive)
  th.emplace_back([&]() {
    for (int k = 0; k < 10; ++k) {
      MgpuScene *s2 = make_scene(small);
      MgpuIntersection rec;
      uint8_t hit;
      for (int j = 0; j < 20; ++j)
        if (mgpu_trace(s2, &rays[(size_t)(k * 20 + j) % n_rays], 1, &rec, &hit, nullptr) != MGPU_OK) bad++;
      if (mgpu_scene_destroy(s2) != MGPU_OK) bad++;
    }
  });
  for (auto &t : th) t.join();

  size_t diff = 0;
  for (size_t i = 0; i < n_rays; ++i)
    if (got_hit[i] != want_hit[i] || (want_hit[i] && memcmp(&got[i], &want[i], sizeof(MgpuIntersection)) != 0)) ++diff;
  uint64_t launches = 0, calls = 0, qb = 0, qc = 0;
  int alive = 0;
  double dev_us = 0;
  CHECK(mgpu_trace_server_stats(sc, &launches, &calls, &alive, &dev_us));
  CHECK(mgpu_trace_queue_stats(sc, &qb, &qc));
  CHECK(mgpu_scene_destroy(sc));
  mgpu_free(mesh.nodes);
  mgpu_free(mesh.indices);
  mgpu_free(small.nodes);
  mgpu_free(small.indices);
  printf("stress: %d threads x %d one-ray calls (%llu served by %llu server launches, %llu by %llu queue launches), %d frames beside them: "
         "%zu records differ from the batched call, %d failed calls or frames\n",
         n_threads, per_thread, (unsigned long long)calls, (unsigned long long)launches, (unsigned long long)qc, (unsigned long long)qb, n_frames, diff,
         bad.load());
  fflush(stdout);
  const int code = (diff == 0 && bad.load() == 0) ? 0 : 1;
  // ROCm's AddressSanitizer runtime trips over its own device allocator when libamdhip64's finalizers run after it has been torn
  // down (CHECK dev_runtime_unloaded_ in __cxa_finalize): the sanitized run leaves without finalizers, everything of ours is freed
  if (getenv("STRESS_FAST_EXIT")) _exit(code);
  return code;
}
