// tests/cpp/console_driver.cc -- a console-mode caller in the shape of the reference's own (main_console.cc:57-79 for the
// perspective frame, :104-111 for the stereo panorama): it is written only against the reference's header names
// ("scene.h", "render.h"; here the forwarding headers of include/mallie/) and linked with libmallie_mgpu.so.  It
// loads a scene, renders ONE frame the way the console driver does, converts it to 8 bits per channel with that driver's
// transform (value / count, times 255.5, truncated and clamped) and writes a binary PPM instead of a JPEG (the JPEG
// encoder is outside the hot path).
//
//   console_driver <obj|eson|vox> <file> <W> <H> <plane 0|1> <frame|pano> <out.ppm>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "scene.h"
#include "render.h"

namespace {

unsigned char to_byte(float x) {
  const int i = (int)(x * 255.5);
  return (unsigned char)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

void to_ldr(std::vector<unsigned char> &ldr, const std::vector<float> &hdr, const std::vector<int> &samples) {
  ldr.resize(hdr.size());
  for (size_t k = 0; k < hdr.size(); ++k) ldr[k] = to_byte(hdr[k] / samples[k / 3]);
}

bool write_ppm(const char *path, const std::vector<unsigned char> &rgb, int w, int h) {
  FILE *fp = fopen(path, "wb");
  if (!fp) return false;
  fprintf(fp, "P6\n%d %d\n255\n", w, h);
  const bool ok = fwrite(&rgb[0], 1, rgb.size(), fp) == rgb.size();
  fclose(fp);
  return ok;
}

} // namespace

int main(int argc, char **argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: console_driver <obj|eson|vox> <file> <W> <H> <plane> <frame|pano> <out.ppm>\n");
    return 2;
  }
  std::string obj, eson, vox, material;
  if (!strcmp(argv[1], "obj")) obj = argv[2];
  else if (!strcmp(argv[1], "vox")) vox = argv[2];
  else eson = argv[2];
  mallie::Scene scene;
  if (!scene.Init(obj, eson, vox, material, 1.0, false)) return 3;

  mallie::RenderConfig config;
  config.width = atoi(argv[3]);
  config.height = atoi(argv[4]);
  config.plane = atoi(argv[5]) != 0;
  config.eye[0] = 0.0; config.eye[1] = 0.0; config.eye[2] = 20.0; // config.json's eye

  printf("[Mallie] Console mode\n");
  std::vector<float> image((size_t)config.width * config.height * 3);
  std::vector<int> count((size_t)config.width * config.height);
  if (!strcmp(argv[6], "pano")) {
    double eye[3] = {0.0, 1.0, 4.0}; // frame 0 of the turntable: radius 4, height 1
    mallie::RenderPanoramic(scene, config, image, count, eye, config.lookat, config.up, config.quat, /* stereo = */ true);
  } else {
    mallie::Render(scene, config, image, count, config.eye, config.lookat, config.up, config.quat, 1);
  }
  std::vector<unsigned char> out;
  to_ldr(out, image, count);
  if (!write_ppm(argv[7], out, config.width, config.height)) return 4;
  printf("\n[Mallie] Output %s\n", argv[7]);
  return 0;
}
