// host/mesh_io.cc -- placeholder; the readers land with SURVEY.md 8(f) N2.
#include <cstdio>
#include "mesh_io.hpp"

namespace mesh_io {
bool LoadObj(Mesh &, const char *filename) {
  fprintf(stderr, "mesh_io::LoadObj(%s): reader not built yet\n", filename);
  return false;
}
bool LoadESON(Mesh &, const char *filename) {
  fprintf(stderr, "mesh_io::LoadESON(%s): reader not built yet\n", filename);
  return false;
}
} // namespace mesh_io
