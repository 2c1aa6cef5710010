// oracle/ref_aov_driver.cc -- TEST INFRASTRUCTURE ONLY (never shipped, never linked by the product).
//
// ShowNormal / ShowUV (render.cc:458-516) live in an anonymous namespace of the reference's render.cc and have no
// caller, so no object file exports them.  This translation unit #includes the UNMODIFIED render.cc where it lies under
// /root/reference (include path, see oracle/Makefile) and calls the two functions from inside the same translation unit,
// one pixel after the other in scanline order with the reference's own thread-0 RNG stream -- the loop nest of Render()
// (render.cc:657-681) with PathTrace replaced by the AOV function.  Only its OUTPUT vectors are committed
// (oracle/make_goldens.py aov -> tests/golden/aov_*.npz).
//
//   ref_aov_driver <obj|eson|vox> <file> <W> <H> <eye[3]> <lookat[3]> <normal|uv> <out.f32>      (OMP_NUM_THREADS=1)
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "render.cc" // the reference source itself, unmodified

int main(int argc, char **argv) {
  if (argc < 13) {
    fprintf(stderr, "usage: ref_aov_driver kind file W H eye[3] lookat[3] normal|uv out\n");
    return 2;
  }
  std::string obj, eson, vox, mat;
  if (!strcmp(argv[1], "obj")) obj = argv[2];
  else if (!strcmp(argv[1], "vox")) vox = argv[2];
  else eson = argv[2];
  mallie::Scene scene;
  if (!scene.Init(obj, eson, vox, mat, 1.0, false)) return 3;
  mallie::RenderConfig config;
  config.width = atoi(argv[3]);
  config.height = atoi(argv[4]);
  double eye[3], lookat[3], up[3] = {0, 1, 0}, quat[4] = {0, 0, 0, 0};
  for (int k = 0; k < 3; k++) { eye[k] = atof(argv[5 + k]); lookat[k] = atof(argv[8 + k]); }
  const bool uv = !strcmp(argv[11], "uv");
  double origin[3], corner[3], du[3], dv[3];
  mallie::Camera camera(eye, lookat, up);
  camera.BuildCameraFrame(origin, corner, du, dv, config.fov, quat, config.width, config.height);
  std::vector<float> image(3 * (size_t)config.width * config.height, 0.0f);
  std::vector<int> count((size_t)config.width * config.height, 0);
  mallie::init_randomreal();
  for (int y = 0; y < config.height; y++)
    for (int x = 0; x < config.width; x++) {
      real3 r = uv ? mallie::ShowUV(scene, camera, config, image, count, x, y, 1)
                   : mallie::ShowNormal(scene, camera, config, image, count, x, y, 1);
      for (int k = 0; k < 3; k++) image[3 * ((size_t)y * config.width + x) + k] = r[k];
    }
  FILE *fp = fopen(argv[12], "wb");
  if (!fp || fwrite(&image[0], 4, image.size(), fp) != image.size()) return 4;
  fclose(fp);
  return 0;
}
