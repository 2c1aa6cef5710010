// tests/cpp/facade_driver.cc -- a Mallie-style driver written ONLY against the reference's header names
// ("scene.h", "render.h", "camera.h": here the forwarding headers in include/mallie/) and linked with
// libmallie_mgpu.so, the way main_console.cc (main_console.cc:57-79) uses the reference's own objects.
// It doubles as the harness of tests/test_facade.py.
//
//   facade_driver mesh   <obj|eson|vox> <file> <scale> <out_prefix>          (CPU only: loader + BVH build)
//   facade_driver render <obj|eson|vox> <file> <W> <H> <plane> <passes> <maxPathLength> <seed> <out.f32>   (GPU)
//   facade_driver trace  <obj|eson|vox> <file> <rays.bin> <out.bin>                                          (GPU)
//   facade_driver envrays <W> <H> <eye[3]> <lookat[3]> <uv.bin> <out.bin>     (CPU: Camera::GenerateEnvRay / StereoEnvRay)
//   facade_driver plane  <a> <b> <c> <d> <rays.bin> <out.bin>                  (CPU: Plane::intersect)
//   facade_driver step   <obj|eson|vox> <file> <W> <H> <plane> <calls> <maxPathLength> <seed> <step> <out.bin>   (GPU)
#include <string>
#include <chrono>
#include <thread>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>

#include "common.h"
#include "scene.h"
#include "render.h"
#include "camera.h"
#include "prim-plane.h"

// one diagnostic of the C ABI underneath (include/mgpu.h), declared here so that the driver needs the Mallie headers only
extern "C" int mgpu_trace_queue_stats(MgpuScene *scene, uint64_t *launches, uint64_t *calls);
extern "C" int mgpu_trace_server_stats(MgpuScene *scene, uint64_t *launches, uint64_t *calls, int *alive, double *device_us);

static bool init(mallie::Scene &scene, const char *kind, const char *file, double scale) {
  std::string obj, eson, vox, mat;
  if (!strcmp(kind, "obj")) obj = file; else if (!strcmp(kind, "vox")) vox = file; else eson = file;
  return scene.Init(obj, eson, vox, mat, scale, false);
}

static void wr(FILE *fp, const void *p, size_t n) { if (n && fwrite(p, 1, n, fp) != n) { perror("write"); exit(2); } }

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  if (!strcmp(argv[1], "mesh") && argc >= 6) {
    mallie::Scene scene;
    if (!init(scene, argv[2], argv[3], atof(argv[4]))) return 3;
    const Mesh &m = scene.GetMesh();
    std::string out = argv[5];
    FILE *fp = fopen((out + ".mesh").c_str(), "wb");
    uint64_t nv = m.numVertices, nf = m.numFaces;
    uint8_t hn = m.facevarying_normals ? 1 : 0, hu = m.facevarying_uvs ? 1 : 0;
    wr(fp, &nv, 8); wr(fp, &nf, 8); wr(fp, &hn, 1); wr(fp, &hu, 1);
    wr(fp, m.vertices, 24 * nv); wr(fp, m.faces, 12 * nf); wr(fp, m.materialIDs, 4 * nf);
    if (hn) wr(fp, m.facevarying_normals, 72 * nf);
    if (hu) wr(fp, m.facevarying_uvs, 48 * nf);
    fclose(fp);
    { // Scene::GetMaterial(0..255): the palette materials of a .vox scene, the default material otherwise
      FILE *fm = fopen((out + ".mat").c_str(), "wb");
      for (int i = 0; i < 256; i++) {
        const Material &mt = scene.GetMaterial(i);
        double d[3] = {mt.diffuse[0], mt.diffuse[1], mt.diffuse[2]};
        wr(fm, d, 24);
      }
      fclose(fm);
    }
    // BVHAccel::Dump writes the reference's binary layout (bvh_accel.cc:484-512)
    if (!scene.GetAccel().Dump((out + ".bvh").c_str())) return 4;
    // and Load must read it back
    BVHAccel again;
    if (!again.Load((out + ".bvh").c_str()) || again.GetNodes().size() != scene.GetAccel().GetNodes().size()) return 5;
    return 0;
  }
  if (!strcmp(argv[1], "render") && argc >= 11) {
    mallie::Scene scene;
    if (!init(scene, argv[2], argv[3], 1.0)) return 3;
    mallie::RenderConfig config;                    // eye (0,0,-5) default is overridden like config.json does
    config.width = atoi(argv[4]); config.height = atoi(argv[5]); config.plane = atoi(argv[6]) != 0;
    const int passes = atoi(argv[7]);
    config.eye[0] = 0; config.eye[1] = 0; config.eye[2] = 20;
    mallie::SetMaxPathLength(atoi(argv[8]));
    mallie::SetRenderSeed(strtoull(argv[9], NULL, 10));
    std::vector<float> image(3 * (size_t)config.width * config.height), accum(image.size(), 0.0f);
    std::vector<int> count((size_t)config.width * config.height, 0);
    for (int p = 0; p < passes; p++) {              // the RenderThread loop of main_sdl.cc:569-643
      mallie::Render(scene, config, image, count, config.eye, config.lookat, config.up, config.quat, 1);
      for (size_t i = 0; i < image.size(); i++) accum[i] += image[i];   // AccumImage
    }
    FILE *fp = fopen(argv[10], "wb");
    wr(fp, &accum[0], 4 * accum.size()); wr(fp, &count[0], 4 * count.size());
    fclose(fp);
    // same frame in one multi-pass launch must give the same bits
    mallie::SetRenderSeed(strtoull(argv[9], NULL, 10));
    std::vector<float> image2(image.size()); std::vector<int> count2(count.size(), 0);
    if (!mallie::RenderPasses(scene, config, image2, count2, config.eye, config.lookat, config.up, config.quat, passes)) return 6;
    if (memcmp(&image2[0], &accum[0], 4 * accum.size()) != 0) { fprintf(stderr, "RenderPasses != Render+AccumImage\n"); return 7; }
    return 0;
  }
  if (!strcmp(argv[1], "envrays") && argc >= 12) { // the two equirectangular ray generators on a list of (u, v)
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    double eye[3], lookat[3], up[3] = {0, 1, 0}, quat[4] = {0, 0, 0, 0}, fr[12];
    for (int k = 0; k < 3; k++) { eye[k] = atof(argv[4 + k]); lookat[k] = atof(argv[7 + k]); }
    mallie::Camera cam(eye, lookat, up);
    cam.BuildCameraFrame(fr, fr + 3, fr + 6, fr + 9, 45.0, quat, W, H);
    FILE *fi = fopen(argv[10], "rb");
    FILE *fo = fopen(argv[11], "wb");
    double uv[2];
    while (fread(uv, 8, 2, fi) == 2) {
      const Ray m = cam.GenerateEnvRay(uv[0], uv[1]), s = cam.GenerateStereoEnvRay(uv[0], uv[1]);
      wr(fo, &m.org, 24); wr(fo, &m.dir, 24); wr(fo, &s.org, 24); wr(fo, &s.dir, 24);
    }
    fclose(fi); fclose(fo);
    return 0;
  }
  if (!strcmp(argv[1], "plane") && argc >= 8) { // Plane::intersect on rays of (org, dir, starting t)
    mallie::Plane pl;
    pl.set((float)atof(argv[2]), (float)atof(argv[3]), (float)atof(argv[4]), (float)atof(argv[5]));
    FILE *fi = fopen(argv[6], "rb");
    FILE *fo = fopen(argv[7], "wb");
    double r[7];
    while (fread(r, 8, 7, fi) == 7) {
      Ray ray;
      memset(&ray, 0, sizeof(ray));
      ray.org = real3(r[0], r[1], r[2]);
      ray.dir = real3(r[3], r[4], r[5]);
      Intersection is;
      memset(&is, 0, sizeof(is));
      is.t = r[6];
      is.faceID = 7;
      const double hit = pl.intersect(&is, ray) ? 1.0 : 0.0, mat = (double)is.materialID, face = (double)is.faceID, uv = is.u + is.v;
      wr(fo, &hit, 8); wr(fo, &is.t, 8); wr(fo, &is.position, 24); wr(fo, &is.geometricNormal, 24); wr(fo, &is.normal, 24);
      wr(fo, &is.tangent, 24); wr(fo, &is.binormal, 24); wr(fo, is.texcoord, 16); wr(fo, &mat, 8); wr(fo, &face, 8); wr(fo, &uv, 8);
    }
    fclose(fi); fclose(fo);
    return 0;
  }
  if (!strcmp(argv[1], "step") && argc >= 12) { // Render(..., step): `calls` consecutive calls, image of the last + count
    mallie::Scene scene;
    if (!init(scene, argv[2], argv[3], 1.0)) return 3;
    mallie::RenderConfig config;
    config.width = atoi(argv[4]); config.height = atoi(argv[5]); config.plane = atoi(argv[6]) != 0;
    const int calls = atoi(argv[7]), step = atoi(argv[10]);
    config.eye[0] = 0; config.eye[1] = 0; config.eye[2] = 20;
    mallie::SetMaxPathLength(atoi(argv[8]));
    mallie::SetRenderSeed(strtoull(argv[9], NULL, 10));
    std::vector<float> image(3 * (size_t)config.width * config.height);
    std::vector<int> count((size_t)config.width * config.height, 0);
    for (int p = 0; p < calls; p++)
      mallie::Render(scene, config, image, count, config.eye, config.lookat, config.up, config.quat, step);
    FILE *fp = fopen(argv[11], "wb");
    wr(fp, &image[0], 4 * image.size()); wr(fp, &count[0], 4 * count.size());
    fclose(fp);
    return 0;
  }
  if (!strcmp(argv[1], "panoramic") && argc >= 9) {   // what main_console.cc:95-111 does for one frame
    mallie::Scene scene;
    if (!init(scene, argv[2], argv[3], 1.0)) return 3;
    mallie::RenderConfig config;
    config.width = atoi(argv[4]); config.height = atoi(argv[5]);
    const bool stereo = atoi(argv[6]) != 0;
    mallie::SetRenderSeed(strtoull(argv[7], NULL, 10));
    std::vector<float> image(3 * (size_t)config.width * config.height);
    std::vector<int> count((size_t)config.width * config.height, 0);
    double eye[3] = {0.0, 1.0, 4.0};
    mallie::RenderPanoramic(scene, config, image, count, eye, config.lookat, config.up, config.quat, stereo);
    FILE *fp = fopen(argv[8], "wb");
    wr(fp, &image[0], 4 * image.size()); wr(fp, &count[0], 4 * count.size());
    fclose(fp);
    return 0;
  }
  if (!strcmp(argv[1], "trace_mt") && argc >= 6) { // Scene::Trace from 4 host threads at once, as the reference's OpenMP loop does
    mallie::Scene scene;
    if (!init(scene, argv[2], argv[3], 1.0)) return 3;
    FILE *fi = fopen(argv[4], "rb");
    fseek(fi, 0, SEEK_END); size_t n = ftell(fi) / 48; rewind(fi);
    std::vector<double> rays(6 * n);
    if (fread(&rays[0], 48, n, fi) != n) return 4;
    fclose(fi);
    std::vector<Intersection> rec(n);
    std::vector<uint32_t> hits(n, 0);
    memset(&rec[0], 0, sizeof(Intersection) * n);
    const int nthreads = argc >= 7 ? atoi(argv[6]) : 4;
    { // first call outside the clock: device scene upload, staging
      Ray ray;
      ray.org = real3(rays[0], rays[1], rays[2]);
      ray.dir = real3(rays[3], rays[4], rays[5]);
      Intersection warm;
      memset(&warm, 0, sizeof(warm));
      scene.Trace(warm, ray);
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++)
      pool.push_back(std::thread([&, t]() {
        for (size_t i = t; i < n; i += nthreads) {
          Ray ray;
          ray.org = real3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
          ray.dir = real3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
          hits[i] = scene.Trace(rec[i], ray) ? 1 : 0;
        }
      }));
    for (size_t t = 0; t < pool.size(); t++) pool[t].join();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("\ntrace_mt: %d threads, %zu Scene::Trace calls, %.1f us per call (wall / calls), %.0f calls/s\n", nthreads, n, us / n, 1e6 * n / us);
    uint64_t launches = 0, calls = 0;
    if (mgpu_trace_queue_stats(scene.DeviceScene(), &launches, &calls) == 0 && launches)
      printf("trace_mt: submission queue: %llu calls in %llu launches (%.2f per launch)\n", (unsigned long long)calls,
             (unsigned long long)launches, (double)calls / (double)launches);
    int alive = 0;
    double dev_us = 0.0;
    if (mgpu_trace_server_stats(scene.DeviceScene(), &launches, &calls, &alive, &dev_us) == 0 && launches)
      printf("trace_mt: resident server: %llu calls in %llu launches, %.2f us on the device per call, alive at the end: %d\n",
             (unsigned long long)calls, (unsigned long long)launches, dev_us, alive);
    FILE *fo = fopen(argv[5], "wb");
    for (size_t i = 0; i < n; i++) {
      wr(fo, &hits[i], 4); wr(fo, &rec[i].faceID, 4); wr(fo, &rec[i].t, 8); wr(fo, &rec[i].u, 8); wr(fo, &rec[i].v, 8);
      wr(fo, &rec[i].normal, 24);
    }
    fclose(fo);
    return 0;
  }
  if (!strcmp(argv[1], "trace") && argc >= 6) {
    mallie::Scene scene;
    if (!init(scene, argv[2], argv[3], 1.0)) return 3;
    FILE *fi = fopen(argv[4], "rb");
    fseek(fi, 0, SEEK_END); size_t n = ftell(fi) / 48; rewind(fi);
    std::vector<double> rays(6 * n);
    if (fread(&rays[0], 48, n, fi) != n) return 4;
    fclose(fi);
    FILE *fo = fopen(argv[5], "wb");
    for (size_t i = 0; i < n; i++) {                // single-ray Scene::Trace, as PathTrace calls it (render.cc:403)
      Ray ray;
      ray.org = real3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
      ray.dir = real3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
      Intersection isect;
      memset(&isect, 0, sizeof(isect));
      const bool hit = scene.Trace(isect, ray);
      uint32_t h = hit ? 1 : 0;
      wr(fo, &h, 4); wr(fo, &isect.faceID, 4); wr(fo, &isect.t, 8); wr(fo, &isect.u, 8); wr(fo, &isect.v, 8);
      wr(fo, &isect.normal, 24);
    }
    fclose(fo);
    return 0;
  }
  return 2;
}
