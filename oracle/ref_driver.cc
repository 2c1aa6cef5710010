// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never shipped, never linked by the product).
//
// A small driver that is linked against the UNMODIFIED reference sources where they lie under
// /root/reference (see oracle/Makefile, target `ref`) and dumps golden vectors for the render hot path:
// camera frames, Mesh arrays, the BVH, (Ray -> Intersection) batches and Render() images.
// The resulting binary lives only in oracle/_ref/ (git-ignored); only its OUTPUT vectors are committed
// under tests/golden/ (see oracle/make_goldens.py).
//
// Reference entry points exercised (file:line in /root/reference):
//   mallie::Scene::Init            scene.cc:66      mallie::Scene::Trace   scene.cc:253
//   mallie::Render                 render.cc:593    Camera::BuildCameraFrame camera.cc:40
//   mallie::RenderPanoramic        render.cc:710
//   BVHAccel::GetNodes/GetIndices  bvh_accel.h:74   Camera::GenerateRay    camera.cc:222
//
// Rules learnt in SURVEY.md section 0: one process per (scene, config) because Render() keeps
// `static bool initial_pass`; OMP_NUM_THREADS=1 for a deterministic RNG stream; CWD=/root/reference so
// tinyobj resolves the .mtl; <string> must precede scene.h.
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <omp.h>

#include "common.h"
#include "scene.h"
#include "render.h"
#include "camera.h"
#include "prim-plane.h"

namespace {

// Scene keeps mesh_/accel_ protected; a derived type may read them (no reference source is modified).
struct SceneProbe : public mallie::Scene {
  const Mesh &mesh() const { return mesh_; }
  const BVHAccel &accel() const { return accel_; }
};

void die(const char *msg) {
  fprintf(stderr, "ref_driver: %s\n", msg);
  exit(2);
}

void wr(FILE *fp, const void *p, size_t n) {
  if (n && fwrite(p, 1, n, fp) != n) die("short write");
}

FILE *xopen(const std::string &path, const char *mode) {
  FILE *fp = fopen(path.c_str(), mode);
  if (!fp) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  return fp;
}

bool init_scene(SceneProbe &scene, const char *kind, const char *file, double scale) {
  std::string obj, eson, vox, mat;
  if (!strcmp(kind, "obj")) obj = file;
  else if (!strcmp(kind, "eson")) eson = file;
  else if (!strcmp(kind, "vox")) vox = file;
  else die("kind must be obj|eson|vox");
  return scene.Init(obj, eson, vox, mat, scale, false);
}

// mesh <kind> <file> <scale> <out_prefix>
//   <out>.mesh : u64 nv, u64 nf, u8 has_normals, u8 has_uvs, f64 verts[3nv], u32 faces[3nf], u32 matIDs[nf],
//                f64 normals[9nf]?, f64 uvs[6nf]?
//   <out>.bvh  : BVHAccel::Dump layout (u64 nn, 64B nodes, u64 ni, u32 indices) written from GetNodes()/GetIndices()
int cmd_mesh(int argc, char **argv) {
  if (argc < 6) die("mesh <kind> <file> <scale> <out_prefix>");
  SceneProbe scene;
  if (!init_scene(scene, argv[2], argv[3], atof(argv[4]))) die("Scene::Init failed");
  const Mesh &m = scene.mesh();
  std::string out = argv[5];
  {
    FILE *fp = xopen(out + ".mesh", "wb");
    uint64_t nv = m.numVertices, nf = m.numFaces;
    uint8_t hn = m.facevarying_normals ? 1 : 0, hu = m.facevarying_uvs ? 1 : 0;
    wr(fp, &nv, 8); wr(fp, &nf, 8); wr(fp, &hn, 1); wr(fp, &hu, 1);
    wr(fp, m.vertices, sizeof(double) * 3 * nv);
    wr(fp, m.faces, sizeof(unsigned) * 3 * nf);
    wr(fp, m.materialIDs, sizeof(unsigned) * nf);
    if (hn) wr(fp, m.facevarying_normals, sizeof(double) * 9 * nf);
    if (hu) wr(fp, m.facevarying_uvs, sizeof(double) * 6 * nf);
    fclose(fp);
  }
  {
    const std::vector<BVHNode> &nodes = scene.accel().GetNodes();
    const std::vector<unsigned int> &idx = scene.accel().GetIndices();
    FILE *fp = xopen(out + ".bvh", "wb");
    uint64_t nn = nodes.size(), ni = idx.size();
    wr(fp, &nn, 8); wr(fp, &nodes[0], sizeof(BVHNode) * nn);
    wr(fp, &ni, 8); wr(fp, &idx[0], sizeof(unsigned) * ni);
    fclose(fp);
    { // Scene::GetMaterial(0..255) (scene.h:58-65): the 256 palette materials of a .vox scene, the default otherwise
      FILE *fm = xopen(out + ".mat", "wb");
      for (int i = 0; i < 256; i++) {
        const Material &mt = scene.GetMaterial(i);
        double d[3] = {mt.diffuse[0], mt.diffuse[1], mt.diffuse[2]};
        wr(fm, d, 24);
      }
      fclose(fm);
    }
    BVHBuildStatistics st = scene.accel().GetStatistics();
    fprintf(stderr, "\nBVH nodes=%llu leaves=%d branches=%d depth=%d sizeof(BVHNode)=%zu\n",
            (unsigned long long)nn, st.numLeafNodes, st.numBranchNodes, st.maxTreeDepth, sizeof(BVHNode));
  }
  return 0;
}

// trace <kind> <file> <scale> <rays.bin> <out.bin>
//   rays.bin : f64[6] per ray (org, dir)
//   out.bin  : per ray 1 record of 152 bytes:
//     u32 hit, u32 faceID, u32 materialID, u32 f0, u32 f1, u32 f2, (pad to 24) f64 t,u,v, f64 position[3],
//     f64 geometricNormal[3], f64 normal[3], f64 texcoord[2]       (all zero when hit==0)
int cmd_trace(int argc, char **argv) {
  if (argc < 7) die("trace <kind> <file> <scale> <rays.bin> <out.bin>");
  SceneProbe scene;
  if (!init_scene(scene, argv[2], argv[3], atof(argv[4]))) die("Scene::Init failed");
  FILE *fi = xopen(argv[5], "rb");
  fseek(fi, 0, SEEK_END); long sz = ftell(fi); rewind(fi);
  size_t n = sz / 48;
  std::vector<double> rays(6 * n);
  if (fread(&rays[0], 48, n, fi) != n) die("short read");
  fclose(fi);
  FILE *fo = xopen(argv[6], "wb");
  for (size_t i = 0; i < n; i++) {
    Ray ray;
    ray.org = real3(rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]);
    ray.dir = real3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    Intersection isect;
    memset(&isect, 0, sizeof(isect));
    bool hit = scene.Trace(isect, ray);
    uint32_t ih[6] = {0, 0, 0, 0, 0, 0};
    double d[16];
    memset(d, 0, sizeof(d));
    if (hit) {
      ih[0] = 1; ih[1] = isect.faceID; ih[2] = isect.materialID;
      ih[3] = isect.f0; ih[4] = isect.f1; ih[5] = isect.f2;
      d[0] = isect.t; d[1] = isect.u; d[2] = isect.v;
      for (int k = 0; k < 3; k++) {
        d[3 + k] = isect.position[k];
        d[6 + k] = isect.geometricNormal[k];
        d[9 + k] = isect.normal[k];
      }
      d[12] = isect.texcoord[0]; d[13] = isect.texcoord[1];
    }
    wr(fo, ih, sizeof(ih));
    wr(fo, d, 14 * sizeof(double));
  }
  fclose(fo);
  fprintf(stderr, "traced %zu rays, sizeof(Ray)=%zu sizeof(Intersection)=%zu\n", n, sizeof(Ray), sizeof(Intersection));
  return 0;
}

// camera <W> <H> <fov> <ex ey ez> <lx ly lz> <ux uy uz> <q0 q1 q2 q3> <out.bin>  (appends 12 f64: origin,corner,du,dv)
// then, for probe pixel coordinates given on stdin as "u v" pairs, appends GenerateRay(u,v) org+dir (6 f64 each).
int cmd_camera(int argc, char **argv) {
  if (argc < 19) die("camera W H fov eye[3] lookat[3] up[3] quat[4] out");
  int W = atoi(argv[2]), H = atoi(argv[3]);
  double fov = atof(argv[4]);
  double eye[3], lookat[3], up[3], quat[4];
  for (int k = 0; k < 3; k++) { eye[k] = atof(argv[5 + k]); lookat[k] = atof(argv[8 + k]); up[k] = atof(argv[11 + k]); }
  for (int k = 0; k < 4; k++) quat[k] = atof(argv[14 + k]);
  mallie::Camera cam(eye, lookat, up);
  double o[3], c[3], du[3], dv[3];
  cam.BuildCameraFrame(o, c, du, dv, fov, quat, W, H);
  FILE *fo = xopen(argv[18], "wb");
  wr(fo, o, 24); wr(fo, c, 24); wr(fo, du, 24); wr(fo, dv, 24);
  double u, v;
  while (scanf("%lf %lf", &u, &v) == 2) {
    Ray r = cam.GenerateRay(u, v);
    double d[6] = {r.org[0], r.org[1], r.org[2], r.dir[0], r.dir[1], r.dir[2]};
    wr(fo, d, 48);
  }
  fclose(fo);
  return 0;
}

// render <kind> <file> <scale> <W> <H> <plane> <passes> <eye[3]> <lookat[3]> <up[3]> <quat[4]> <out_prefix> [step]
//   writes <out>.pass<k>.f32 (3*W*H float32, the image exactly as Render() left it after pass k) and
//   <out>.count.i32 (W*H int32 after the last pass).  Run with OMP_NUM_THREADS=1.
int cmd_render(int argc, char **argv) {
  if (argc < 23) die("render kind file scale W H plane passes eye[3] lookat[3] up[3] quat[4] out");
  SceneProbe scene;
  if (!init_scene(scene, argv[2], argv[3], atof(argv[4]))) die("Scene::Init failed");
  mallie::RenderConfig config;
  config.width = atoi(argv[5]);
  config.height = atoi(argv[6]);
  config.plane = atoi(argv[7]) != 0;
  int passes = atoi(argv[8]);
  for (int k = 0; k < 3; k++) {
    config.eye[k] = atof(argv[9 + k]);
    config.lookat[k] = atof(argv[12 + k]);
    config.up[k] = atof(argv[15 + k]);
  }
  for (int k = 0; k < 4; k++) config.quat[k] = atof(argv[18 + k]);
  std::string out = argv[22];
  const int step = argc > 23 ? atoi(argv[23]) : 1; // Render()'s last argument (render.cc:597)
  std::vector<float> image(3 * (size_t)config.width * config.height);
  std::vector<int> count((size_t)config.width * config.height, 0);
  for (int p = 0; p < passes; p++) {
    mallie::Render(scene, config, image, count, config.eye, config.lookat, config.up, config.quat, step);
    char suffix[64];
    snprintf(suffix, sizeof(suffix), ".pass%d.f32", p);
    FILE *fp = xopen(out + suffix, "wb");
    wr(fp, &image[0], sizeof(float) * image.size());
    fclose(fp);
  }
  FILE *fp = xopen(out + ".count.i32", "wb");
  wr(fp, &count[0], sizeof(int) * count.size());
  fclose(fp);
  fprintf(stderr, "\nrendered %d pass(es) %dx%d sizeof(RenderConfig)=%zu sizeof(Camera)=%zu\n", passes, config.width,
          config.height, sizeof(mallie::RenderConfig), sizeof(mallie::Camera));
  return 0;
}

// panoramic <kind> <file> <scale> <W> <H> <stereo> <eye[3]> <lookat[3]> <up[3]> <quat[4]> <out_prefix>
//   one mallie::RenderPanoramic() call (render.cc:710; what main_console.cc:111 runs): <out>.f32 (3*W*H float32) and
//   <out>.count.i32.  Run with OMP_NUM_THREADS=1 in a fresh process (its own `static bool initial_pass`).
int cmd_panoramic(int argc, char **argv) {
  if (argc < 22) die("panoramic kind file scale W H stereo eye[3] lookat[3] up[3] quat[4] out");
  SceneProbe scene;
  if (!init_scene(scene, argv[2], argv[3], atof(argv[4]))) die("Scene::Init failed");
  mallie::RenderConfig config;
  config.width = atoi(argv[5]);
  config.height = atoi(argv[6]);
  const bool stereo = atoi(argv[7]) != 0;
  for (int k = 0; k < 3; k++) {
    config.eye[k] = atof(argv[8 + k]);
    config.lookat[k] = atof(argv[11 + k]);
    config.up[k] = atof(argv[14 + k]);
  }
  for (int k = 0; k < 4; k++) config.quat[k] = atof(argv[17 + k]);
  std::string out = argv[21];
  std::vector<float> image(3 * (size_t)config.width * config.height);
  std::vector<int> count((size_t)config.width * config.height, 0);
  mallie::RenderPanoramic(scene, config, image, count, config.eye, config.lookat, config.up, config.quat, stereo);
  FILE *fp = xopen(out + ".f32", "wb");
  wr(fp, &image[0], sizeof(float) * image.size());
  fclose(fp);
  fp = xopen(out + ".count.i32", "wb");
  wr(fp, &count[0], sizeof(int) * count.size());
  fclose(fp);
  return 0;
}


// envrays <W> <H> <fov> <eye[3]> <lookat[3]> <up[3]> <quat[4]> <out.bin>: for "u v" pairs on stdin, appends
// Camera::GenerateEnvRay(u, v) and Camera::GenerateStereoEnvRay(u, v) (camera.cc:242-329) as org+dir (2 x 6 f64)
int cmd_envrays(int argc, char **argv) {
  if (argc < 19) die("envrays W H fov eye[3] lookat[3] up[3] quat[4] out");
  int W = atoi(argv[2]), H = atoi(argv[3]);
  double fov = atof(argv[4]);
  double eye[3], lookat[3], up[3], quat[4];
  for (int k = 0; k < 3; k++) { eye[k] = atof(argv[5 + k]); lookat[k] = atof(argv[8 + k]); up[k] = atof(argv[11 + k]); }
  for (int k = 0; k < 4; k++) quat[k] = atof(argv[14 + k]);
  mallie::Camera cam(eye, lookat, up);
  double o[3], c[3], du[3], dv[3];
  cam.BuildCameraFrame(o, c, du, dv, fov, quat, W, H);
  FILE *fo = xopen(argv[18], "wb");
  double u, v;
  while (scanf("%lf %lf", &u, &v) == 2) {
    Ray a = cam.GenerateEnvRay(u, v), b = cam.GenerateStereoEnvRay(u, v);
    double d[12] = {a.org[0], a.org[1], a.org[2], a.dir[0], a.dir[1], a.dir[2],
                    b.org[0], b.org[1], b.org[2], b.dir[0], b.dir[1], b.dir[2]};
    wr(fo, d, sizeof(d));
  }
  fclose(fo);
  return 0;
}

// plane <a> <b> <c> <d> <rays.bin> <out.bin>: Plane::intersect (prim-plane.cc:8-44) for rays of 7 f64 (org, dir, the
// Intersection::t to start from); per ray 20 f64: hit, t, position, geometricNormal, normal, tangent, binormal,
// texcoord, then materialID and faceID as doubles... (the record starts zeroed, faceID 7 to show it is left alone)
int cmd_plane(int argc, char **argv) {
  if (argc < 8) die("plane a b c d rays.bin out.bin");
  mallie::Plane pl;
  pl.set((float)atof(argv[2]), (float)atof(argv[3]), (float)atof(argv[4]), (float)atof(argv[5]));
  FILE *fi = xopen(argv[6], "rb");
  fseek(fi, 0, SEEK_END); long sz = ftell(fi); rewind(fi);
  size_t n = sz / 56;
  std::vector<double> rays(7 * n);
  if (fread(&rays[0], 56, n, fi) != n) die("short read");
  fclose(fi);
  FILE *fo = xopen(argv[7], "wb");
  for (size_t i = 0; i < n; i++) {
    Ray ray;
    memset(&ray, 0, sizeof(ray));
    ray.org = real3(rays[7 * i + 0], rays[7 * i + 1], rays[7 * i + 2]);
    ray.dir = real3(rays[7 * i + 3], rays[7 * i + 4], rays[7 * i + 5]);
    Intersection is;
    memset(&is, 0, sizeof(is));
    is.t = rays[7 * i + 6];
    is.faceID = 7;
    bool hit = pl.intersect(&is, ray);
    double d[22] = {hit ? 1.0 : 0.0, is.t, is.position[0], is.position[1], is.position[2], is.geometricNormal[0],
                    is.geometricNormal[1], is.geometricNormal[2], is.normal[0], is.normal[1], is.normal[2], is.tangent[0],
                    is.tangent[1], is.tangent[2], is.binormal[0], is.binormal[1], is.binormal[2], is.texcoord[0],
                    is.texcoord[1], (double)is.materialID, (double)is.faceID, is.u + is.v};
    wr(fo, d, sizeof(d));
  }
  fclose(fo);
  return 0;
}

// bench <kind> <file> <W> <H> <plane> <passes> <eye[3]> <lookat[3]>
// Times `passes` calls of the reference's own mallie::Render() (its OpenMP loop on every host thread the environment
// gives it, render.cc:593-708) after one untimed call, and prints one line per pass and a summary: the "kind": "reference"
// CPU baseline of bench.py.  Nothing is written; the images are the reference's business.
int cmd_bench(int argc, char **argv) {
  if (argc < 14) die("bench kind file W H plane passes eye[3] lookat[3]");
  SceneProbe scene;
  if (!init_scene(scene, argv[2], argv[3], 1.0)) die("Scene::Init failed");
  mallie::RenderConfig config;
  config.width = atoi(argv[4]);
  config.height = atoi(argv[5]);
  config.plane = atoi(argv[6]) != 0;
  const int passes = atoi(argv[7]);
  for (int k = 0; k < 3; k++) {
    config.eye[k] = atof(argv[8 + k]);
    config.lookat[k] = atof(argv[11 + k]);
  }
  std::vector<float> image(3 * (size_t)config.width * config.height);
  std::vector<int> count((size_t)config.width * config.height, 0);
  mallie::Render(scene, config, image, count, config.eye, config.lookat, config.up, config.quat, 1); // first call: static init
  double total = 0.0;
  for (int p = 0; p < passes; p++) {
    const double t0 = omp_get_wtime();
    mallie::Render(scene, config, image, count, config.eye, config.lookat, config.up, config.quat, 1);
    const double dt = omp_get_wtime() - t0;
    total += dt;
    printf("\nref_bench_pass %d %.6f\n", p, dt);
  }
  printf("\nref_bench passes=%d seconds=%.6f threads=%d width=%d height=%d\n", passes, total, omp_get_max_threads(), config.width,
         config.height);
  return 0;
}

} // namespace

int main(int argc, char **argv) {
  if (argc < 2) die("usage: ref_driver mesh|trace|camera|render ...");
  if (!strcmp(argv[1], "envrays")) return cmd_envrays(argc, argv);
  if (!strcmp(argv[1], "plane")) return cmd_plane(argc, argv);
  if (!strcmp(argv[1], "mesh")) return cmd_mesh(argc, argv);
  if (!strcmp(argv[1], "trace")) return cmd_trace(argc, argv);
  if (!strcmp(argv[1], "camera")) return cmd_camera(argc, argv);
  if (!strcmp(argv[1], "render")) return cmd_render(argc, argv);
  if (!strcmp(argv[1], "panoramic")) return cmd_panoramic(argc, argv);
  if (!strcmp(argv[1], "bench")) return cmd_bench(argc, argv);
  die("unknown command");
  return 2;
}
