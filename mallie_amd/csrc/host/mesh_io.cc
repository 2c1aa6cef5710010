// host/mesh_io.cc -- the mesh readers behind Scene::Init: Wavefront .obj and .eson.
//
// The arrays they produce feed the BVH builder, whose result depends on face ORDER and on the exact vertex VALUES, so
// the readers reproduce what the reference's MeshLoader::LoadObj / LoadESON (importers/mesh_loader.cc:26-310) make of a
// file, including the behaviour of the tinyobj version it vendors (importers/tiny_obj_loader.cc):
//   * numbers go text -> double by tinyobj's own digit-accumulating parser (tiny_obj_loader.cc:124-233), not strtod,
//     and are then rounded to float;
//   * faces are collected per group and flushed at every usemtl / g / o / EOF with the material that was current
//     BEFORE the statement; each flush de-duplicates (v,vt,vn) triples from scratch; polygons become triangle fans;
//   * a group flushed by `usemtl` whose shape later ends on an EMPTY face list is lost (tiny_obj_loader.cc:789-812);
//   * mtllib is resolved relative to the current directory; unknown or unreadable materials give id -1;
//   * without `vn` the facevarying normal is normalize(cross(v2-v0, v1-v0)) (mesh_loader.cc:15-22) -- note the
//     operand order, the opposite of BuildIntersection's geometric normal.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "mesh_io.hpp"

namespace {

// ---- number parsing (tiny_obj_loader.cc:124-233) ---------------------------------------------------------------------
bool parse_decimal(const char *s, const char *end, double *out) {
  if (s >= end) return false;
  double mant = 0.0;
  int exp10 = 0;
  bool neg = false, exp_neg = false;
  const char *c = s;
  if (*c == '+' || *c == '-') {
    neg = (*c == '-');
    ++c;
  } else if (!isdigit((unsigned char)*c)) {
    return false;
  }
  int ndig = 0;
  bool more = false;
  while ((more = (c != end)) && isdigit((unsigned char)*c)) {
    mant *= 10;
    mant += (int)(*c - '0');
    ++c;
    ++ndig;
  }
  if (ndig == 0) return false;
  if (more) {
    bool to_exponent = false;
    if (*c == '.') {
      ++c;
      int pos = 1;
      while ((more = (c != end)) && isdigit((unsigned char)*c)) {
        mant += (int)(*c - '0') * pow(10, -pos); // one rounding per digit, as the vendored parser does
        ++pos;
        ++c;
      }
      to_exponent = more;
    } else if (*c == 'e' || *c == 'E') {
      to_exponent = true;
    }
    if (to_exponent && (*c == 'e' || *c == 'E')) {
      ++c;
      if ((more = (c != end)) && (*c == '+' || *c == '-')) {
        exp_neg = (*c == '-');
        ++c;
      } else if (!isdigit((unsigned char)*c)) {
        return false;
      }
      int nexp = 0;
      while ((more = (c != end)) && isdigit((unsigned char)*c)) {
        exp10 = exp10 * 10 + (int)(*c - '0');
        ++c;
        ++nexp;
      }
      if (exp_neg) exp10 = -exp10;
      if (nexp == 0) return false;
    }
  }
  *out = (neg ? -1 : 1) * ldexp(mant * pow(5, exp10), exp10);
  return true;
}

float next_float(const char *&tok) {
  tok += strspn(tok, " \t");
  const char *end = tok + strcspn(tok, " \t\r");
  double v = 0.0;
  parse_decimal(tok, end, &v);
  tok = end;
  return (float)v;
}

inline bool is_blank(char c) { return c == ' ' || c == '\t'; }
inline bool is_eol(char c) { return c == '\r' || c == '\n' || c == '\0'; }
inline int rebase(int idx, int n) { return idx > 0 ? idx - 1 : (idx == 0 ? 0 : n + idx); }

struct Corner {
  int v, vt, vn;
  bool operator<(const Corner &o) const {
    if (v != o.v) return v < o.v;
    if (vn != o.vn) return vn < o.vn;
    if (vt != o.vt) return vt < o.vt;
    return false;
  }
};

Corner next_corner(const char *&tok, int nv, int nvn, int nvt) {
  Corner c = {-1, -1, -1};
  c.v = rebase(atoi(tok), nv);
  tok += strcspn(tok, "/ \t\r");
  if (tok[0] != '/') return c;
  ++tok;
  if (tok[0] == '/') { // v//vn
    ++tok;
    c.vn = rebase(atoi(tok), nvn);
    tok += strcspn(tok, "/ \t\r");
    return c;
  }
  c.vt = rebase(atoi(tok), nvt);
  tok += strcspn(tok, "/ \t\r");
  if (tok[0] != '/') return c;
  ++tok;
  c.vn = rebase(atoi(tok), nvn);
  tok += strcspn(tok, "/ \t\r");
  return c;
}

struct Shape {
  std::vector<float> pos, nrm, uv;
  std::vector<unsigned int> idx;
  std::vector<int> mat;
};

struct ObjFile {
  std::vector<float> v, vn, vt;
  std::vector<std::vector<Corner> > group;
  std::vector<Shape> shapes;
  Shape cur;
  std::map<std::string, int> materials;
  int material;
  bool bad_index; // a face named a v / vn / vt entry the file does not have (the reference reads out of bounds there)
  ObjFile() : material(-1), bad_index(false) {}

  // tiny_obj_loader.cc:352-403: returns false (and adds nothing) for an empty group
  bool flush() {
    if (group.empty()) return false;
    std::map<Corner, unsigned int> seen; // fresh per flush: the reference passes its cache BY VALUE
    for (size_t g = 0; g < group.size(); g++) {
      const std::vector<Corner> &poly = group[g];
      for (size_t k = 2; k < poly.size(); k++) {
        const Corner tri[3] = {poly[0], poly[k - 1], poly[k]};
        for (int c = 0; c < 3; c++) {
          std::map<Corner, unsigned int>::iterator it = seen.find(tri[c]);
          unsigned int id;
          if (it != seen.end()) {
            id = it->second;
          } else {
            const Corner &q = tri[c];
            if (q.v < 0 || (size_t)q.v >= v.size() / 3 || (q.vn >= 0 && (size_t)q.vn >= vn.size() / 3) ||
                (q.vt >= 0 && (size_t)q.vt >= vt.size() / 2)) {
              bad_index = true; // the whole file is refused (LoadObj), nothing of this corner is read
              cur.idx.push_back(0u);
              continue;
            }
            for (int a = 0; a < 3; a++) cur.pos.push_back(v[3 * (size_t)q.v + a]);
            if (q.vn >= 0) for (int a = 0; a < 3; a++) cur.nrm.push_back(vn[3 * (size_t)q.vn + a]);
            if (q.vt >= 0) for (int a = 0; a < 2; a++) cur.uv.push_back(vt[2 * (size_t)q.vt + a]);
            id = (unsigned int)(cur.pos.size() / 3 - 1);
            seen[q] = id;
          }
          cur.idx.push_back(id);
        }
        cur.mat.push_back(material);
      }
    }
    return true;
  }

  void end_shape() { // `g`, `o`, EOF: the shape survives only if THIS flush had faces
    if (flush()) shapes.push_back(cur);
    cur = Shape();
    group.clear();
  }
};

void read_mtl_names(const char *path, std::map<std::string, int> &names) {
  // tiny_obj_loader.cc:406-598: only the name -> index map matters to the mesh. The last (possibly unnamed) material is
  // always flushed, so an unreadable file yields {"" -> 0}.
  names.clear();
  std::ifstream in(path);
  std::string name;
  int count = 0;
  std::string line;
  std::vector<char> buf(8192);
  while (in.peek() != -1) {
    in.getline(&buf[0], (std::streamsize)buf.size());
    line = &buf[0];
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    const char *tok = line.c_str();
    tok += strspn(tok, " \t");
    if (strncmp(tok, "newmtl", 6) == 0 && is_blank(tok[6])) {
      if (!name.empty()) names.insert(std::make_pair(name, count++));
      char nb[4096];
      nb[0] = 0;
      sscanf(tok + 7, "%s", nb);
      name = nb;
    }
  }
  names.insert(std::make_pair(name, count));
}

bool read_obj(const char *filename, ObjFile &o) {
  std::ifstream in(filename);
  if (!in) {
    fprintf(stderr, "Cannot open file [%s]\n", filename);
    return false;
  }
  std::vector<char> buf(8192);
  std::string line, name;
  while (in.peek() != -1) {
    in.getline(&buf[0], (std::streamsize)buf.size());
    line = &buf[0];
    if (!line.empty() && line[line.size() - 1] == '\n') line.erase(line.size() - 1);
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    if (line.empty()) continue;
    const char *tok = line.c_str();
    tok += strspn(tok, " \t");
    if (tok[0] == '\0' || tok[0] == '#') continue;
    if (tok[0] == 'v' && is_blank(tok[1])) {
      tok += 2;
      for (int a = 0; a < 3; a++) o.v.push_back(next_float(tok));
    } else if (tok[0] == 'v' && tok[1] == 'n' && is_blank(tok[2])) {
      tok += 3;
      for (int a = 0; a < 3; a++) o.vn.push_back(next_float(tok));
    } else if (tok[0] == 'v' && tok[1] == 't' && is_blank(tok[2])) {
      tok += 3;
      for (int a = 0; a < 2; a++) o.vt.push_back(next_float(tok));
    } else if (tok[0] == 'f' && is_blank(tok[1])) {
      tok += 2;
      tok += strspn(tok, " \t");
      std::vector<Corner> poly;
      while (!is_eol(tok[0])) {
        poly.push_back(next_corner(tok, (int)(o.v.size() / 3), (int)(o.vn.size() / 3), (int)(o.vt.size() / 2)));
        tok += strspn(tok, " \t\r");
      }
      o.group.push_back(poly);
    } else if (strncmp(tok, "usemtl", 6) == 0 && is_blank(tok[6])) {
      char nb[4096];
      nb[0] = 0;
      sscanf(tok + 7, "%s", nb);
      if (o.flush()) o.group.clear(); // flushed under the PREVIOUS material; the shape stays open
      std::map<std::string, int>::iterator it = o.materials.find(nb);
      o.material = (it != o.materials.end()) ? it->second : -1;
    } else if (strncmp(tok, "mtllib", 6) == 0 && is_blank(tok[6])) {
      char nb[4096];
      nb[0] = 0;
      sscanf(tok + 7, "%s", nb);
      read_mtl_names(nb, o.materials); // relative to the current directory
    } else if ((tok[0] == 'g' || tok[0] == 'o') && is_blank(tok[1])) {
      o.end_shape();
    }
  }
  o.end_shape();
  if (o.bad_index) {
    fprintf(stderr, "Mallie:err\tface index out of range in [%s]\n", filename);
    return false;
  }
  return true;
}

void face_normal(const float *p0, const float *p1, const float *p2, real out[3]) { // mesh_loader.cc:15-22
  const real a[3] = {(real)p1[0] - (real)p0[0], (real)p1[1] - (real)p0[1], (real)p1[2] - (real)p0[2]};
  const real b[3] = {(real)p2[0] - (real)p0[0], (real)p2[1] - (real)p0[1], (real)p2[2] - (real)p0[2]};
  real3 n = vcross(real3(b[0], b[1], b[2]), real3(a[0], a[1], a[2]));
  n.normalize();
  out[0] = n.x;
  out[1] = n.y;
  out[2] = n.z;
}

} // namespace

namespace mesh_io {

bool LoadObj(Mesh &mesh, const char *filename) {
  ObjFile o;
  if (!read_obj(filename, o)) return false;
  size_t nv = 0, nf = 0;
  for (size_t i = 0; i < o.shapes.size(); i++) {
    nv += o.shapes[i].pos.size() / 3;
    nf += o.shapes[i].idx.size() / 3;
  }
  printf("[LoadOBJ] # of shapes in .obj : %zu\n[LoadOBJ] # of faces: %zu\n[LoadOBJ] # of vertices: %zu\n", o.shapes.size(),
         nf, nv);
  memset(&mesh, 0, sizeof(mesh));
  mesh.numVertices = nv;
  mesh.numFaces = nf;
  mesh.vertices = new real[3 * nv];
  mesh.faces = new unsigned int[3 * nf];
  mesh.materialIDs = new unsigned int[nf];
  mesh.facevarying_normals = new real[9 * nf];
  mesh.facevarying_uvs = new real[6 * nf];
  memset(mesh.facevarying_uvs, 0, sizeof(real) * 6 * nf);
  size_t v0 = 0, f0 = 0;
  for (size_t s = 0; s < o.shapes.size(); s++) {
    const Shape &sh = o.shapes[s];
    const size_t sf = sh.idx.size() / 3, sv = sh.pos.size() / 3;
    for (size_t i = 0; i < 3 * sv; i++) mesh.vertices[3 * v0 + i] = sh.pos[i];
    for (size_t f = 0; f < sf; f++) {
      const unsigned int id[3] = {sh.idx[3 * f], sh.idx[3 * f + 1], sh.idx[3 * f + 2]};
      for (int c = 0; c < 3; c++) mesh.faces[3 * (f0 + f) + c] = id[c] + (unsigned int)v0;
      mesh.materialIDs[f0 + f] = (unsigned int)sh.mat[f];
      real *fn = &mesh.facevarying_normals[9 * (f0 + f)];
      if (!sh.nrm.empty()) { // indexed by the VERTEX id, as the reference does (mesh_loader.cc:99-132)
        for (int c = 0; c < 3; c++)
          for (int a = 0; a < 3; a++) {
            const size_t at = 3 * (size_t)id[c] + a;
            fn[3 * c + a] = at < sh.nrm.size() ? (real)sh.nrm[at] : 0.0;
          }
      } else {
        real n[3];
        face_normal(&sh.pos[3 * id[0]], &sh.pos[3 * id[1]], &sh.pos[3 * id[2]], n);
        for (int c = 0; c < 3; c++)
          for (int a = 0; a < 3; a++) fn[3 * c + a] = n[a];
      }
      if (!sh.uv.empty()) {
        real *fu = &mesh.facevarying_uvs[6 * (f0 + f)];
        for (int c = 0; c < 3; c++)
          for (int a = 0; a < 2; a++) {
            const size_t at = 2 * (size_t)id[c] + a;
            fu[2 * c + a] = at < sh.uv.size() ? (real)sh.uv[at] : 0.0;
          }
      }
    }
    v0 += sv;
    f0 += sf;
  }
  return true;
}

// ESON container (importers/eson.cc:133-313): i64 total size, then elements {u8 type, cstring key, payload}; payloads:
// 1 = f64, 2 = i64, 4 = string (i64 n + bytes), 6 = binary (i64 n + bytes), 7 = object (i64 n + one element).
bool LoadESON(Mesh &mesh, const char *filename) {
  FILE *fp = fopen(filename, "rb");
  if (!fp) {
    fprintf(stderr, "Failed to load file: %s\n", filename);
    return false;
  }
  fseek(fp, 0, SEEK_END);
  const long len = ftell(fp);
  rewind(fp);
  std::vector<unsigned char> buf((size_t)(len > 0 ? len : 0));
  const bool read_ok = len > 8 && fread(&buf[0], 1, (size_t)len, fp) == (size_t)len;
  fclose(fp);
  if (!read_ok) return false;
  struct Blob { const unsigned char *p; long long n; };
  std::map<std::string, Blob> bin;
  std::map<std::string, long long> ints;
  long long total;
  memcpy(&total, &buf[0], 8);
  size_t at = 8;
  const size_t stop = (size_t)std::min<long long>(total, len);
  while (at < stop) {
    const unsigned char type = buf[at++];
    const char *key = (const char *)&buf[at];
    const size_t klen = strnlen(key, stop - at);
    if (at + klen >= stop) return false;
    at += klen + 1;
    long long n = 0;
    if (type == 1 || type == 2) {
      if (at + 8 > stop) return false;
      if (type == 2) {
        memcpy(&n, &buf[at], 8);
        ints[key] = n;
      }
      at += 8;
    } else if (type == 4 || type == 6 || type == 7) {
      if (at + 8 > stop) return false;
      memcpy(&n, &buf[at], 8);
      at += 8;
      if (n < 0 || at + (size_t)n > stop) return false;
      if (type == 6) {
        Blob b = {&buf[at], n};
        bin[key] = b;
      }
      at += (size_t)n;
    } else {
      return false;
    }
  }
  if (!ints.count("num_vertices") || !ints.count("num_faces") || !bin.count("vertices") || !bin.count("faces")) return false;
  const size_t nv = (size_t)ints["num_vertices"], nf = (size_t)ints["num_faces"];
  if ((size_t)bin["vertices"].n < 12 * nv || (size_t)bin["faces"].n < 12 * nf) return false;
  printf("# of vertices: %zu\n# of faces   : %zu\n", nv, nf);
  memset(&mesh, 0, sizeof(mesh));
  mesh.numVertices = nv;
  mesh.numFaces = nf;
  mesh.vertices = new real[3 * nv];
  mesh.faces = new unsigned int[3 * nf];
  mesh.materialIDs = new unsigned int[nf];
  const float *fv = (const float *)bin["vertices"].p; // float32 positions, int32 faces, uint16 material ids
  const int *ff = (const int *)bin["faces"].p;
  for (size_t i = 0; i < 3 * nv; i++) mesh.vertices[i] = fv[i];
  for (size_t i = 0; i < 3 * nf; i++) mesh.faces[i] = (unsigned int)ff[i];
  if (bin.count("material_ids") && (size_t)bin["material_ids"].n >= 2 * nf) {
    const unsigned short *m = (const unsigned short *)bin["material_ids"].p;
    for (size_t i = 0; i < nf; i++) mesh.materialIDs[i] = m[i];
  } else {
    for (size_t i = 0; i < nf; i++) mesh.materialIDs[i] = 0;
  }
  // the reference leaves normals / uvs unset for ESON input (mesh_loader.cc:300-306)
  return true;
}

// ---- MagicaVoxel .vox (MeshLoader::LoadMagicaVoxel, mesh_loader.cc:312-400; MagicaVoxelLoader::Load,
// importers/magicavoxel_loader.cc:60-157) ---------------------------------------------------------------------------
// The format's default palette (used when the file carries no RGBA chunk), generated instead of tabulated: index 0 is
// transparent black; 1..215 walk a 6x6x6 colour cube with levels ff, cc, 99, 66, 33, 00 -- bits 16-23 fastest, then
// bits 8-15, then bits 0-7 -- stopping before its all-zero corner; then four ramps of ten (bits 0-7, 8-15, 16-23, grey)
// with levels ee dd bb aa 88 77 55 44 22 11.  Alpha (bits 24-31) is ff.  Checked entry by entry against the reference's
// reader through tests/golden/objload_vox_default.npz.
static void default_vox_palette(unsigned int pal[256]) {
  static const unsigned int ramp[10] = {0xee, 0xdd, 0xbb, 0xaa, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
  pal[0] = 0x00000000u;
  for (unsigned int k = 0; k < 215; k++) {
    const unsigned int c0 = 0xffu - 0x33u * (k % 6), c1 = 0xffu - 0x33u * ((k / 6) % 6), c2 = 0xffu - 0x33u * (k / 36);
    pal[1 + k] = 0xff000000u | (c0 << 16) | (c1 << 8) | c2;
  }
  for (unsigned int i = 0; i < 10; i++) {
    pal[216 + i] = 0xff000000u | ramp[i];
    pal[226 + i] = 0xff000000u | (ramp[i] << 8);
    pal[236 + i] = 0xff000000u | (ramp[i] << 16);
    pal[246 + i] = 0xff000000u | (ramp[i] << 16) | (ramp[i] << 8) | ramp[i];
  }
}

bool LoadMagicaVoxel(Mesh &mesh, std::vector<Material> &materials, const char *filename) {
  FILE *fp = fopen(filename, "rb");
  if (!fp) {
    fprintf(stderr, "Failed to load file.\n");
    fprintf(stderr, "Failed to load .vox file.\n");
    return false;
  }
  unsigned char magic[4] = {0, 0, 0, 0};
  if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "VOX ", 4) != 0) {
    fprintf(stderr, "Bad magic number.\n");
    fprintf(stderr, "Failed to load .vox file.\n");
    fclose(fp);
    return false;
  }
  int ver = 0;
  if (fread(&ver, sizeof(int), 1, fp) != 1) ver = 0;
  int size[3] = {0, 0, 0};
  std::vector<unsigned char> voxel; // x, y, z, colour index per voxel
  std::vector<unsigned int> palette;
  for (;;) { // chunks: id[4], int chunkSize, int childChunkSize, content; a parent's children follow it in the stream
    unsigned char id[4];
    int chunkSize = 0, childChunkSize = 0;
    if (fread(id, 1, 4, fp) != 4) break;
    if (fread(&chunkSize, sizeof(int), 1, fp) != 1 || fread(&childChunkSize, sizeof(int), 1, fp) != 1) break;
    if (!memcmp(id, "SIZE", 4)) {
      if (fread(size, sizeof(int), 3, fp) != 3) break;
      fseek(fp, chunkSize - 4 * 3, SEEK_CUR);
    } else if (!memcmp(id, "XYZI", 4)) {
      int numVoxels = 0;
      if (fread(&numVoxels, sizeof(int), 1, fp) != 1 || numVoxels < 0) break;
      voxel.resize((size_t)numVoxels * 4);
      if (numVoxels && fread(&voxel[0], 1, voxel.size(), fp) != voxel.size()) break;
    } else if (!memcmp(id, "RGBA", 4)) {
      palette.resize(256);
      if (fread(&palette[0], sizeof(unsigned int), 256, fp) != 256) break;
    } else {
      fseek(fp, chunkSize, SEEK_CUR); // MAIN and anything unknown: skip the content, read on into the children
    }
  }
  fclose(fp);

  // materials: 256 palette entries, diffuse = byte / 255.0f (float division, then widened)
  unsigned int pal[256];
  if (palette.empty()) default_vox_palette(pal);
  else memcpy(pal, &palette[0], sizeof(pal));
  materials.clear();
  for (int i = 0; i < 256; i++) {
    Material m;
    m.diffuse[2] = (float)((pal[i] >> 0) & 0xff) / 255.0f;
    m.diffuse[1] = (float)((pal[i] >> 8) & 0xff) / 255.0f;
    m.diffuse[0] = (float)((pal[i] >> 16) & 0xff) / 255.0f;
    materials.push_back(m);
  }

  // one cube (8 vertices, 12 triangles) per voxel, centred on the grid, MagicaVoxel's Z-up turned into Y-up
  static const float P[8][3] = {{-1, -1, 1}, {-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, -1}, {-1, 1, -1}, {1, 1, -1}, {1, -1, -1}};
  static const int F[6][4] = {{0, 3, 2, 1}, {2, 3, 7, 6}, {0, 4, 7, 3}, {1, 2, 6, 5}, {4, 5, 6, 7}, {0, 1, 5, 4}};
  const size_t numVoxels = voxel.size() / 4;
  memset(&mesh, 0, sizeof(mesh));
  mesh.numFaces = numVoxels * 12;
  mesh.numVertices = numVoxels * 8;
  mesh.vertices = new real[mesh.numVertices * 3 + 1];
  mesh.faces = new unsigned int[mesh.numFaces * 3 + 1];
  mesh.materialIDs = new unsigned int[mesh.numFaces + 1];
  const float posOffset[3] = {-0.5f * (float)size[0], -0.5f * (float)size[1], -0.5f * (float)size[2]};
  size_t voffset = 0, foffset = 0;
  for (size_t i = 0; i < numVoxels; i++) {
    const int x = voxel[4 * i + 0], y = voxel[4 * i + 1], z = voxel[4 * i + 2];
    const int col = (int)voxel[4 * i + 3] - 1; // palette index 0 is "no voxel": it becomes material id -1
    for (int j = 0; j < 8; j++) { // float arithmetic, stored as double
      mesh.vertices[3 * (voffset + j) + 0] = posOffset[0] + (float)x + 0.5f * P[j][0];
      mesh.vertices[3 * (voffset + j) + 2] = posOffset[1] + -((float)y + 0.5f * P[j][1]);
      mesh.vertices[3 * (voffset + j) + 1] = posOffset[2] + (float)z + 0.5f * P[j][2];
    }
    for (int f = 0; f < 6; f++) { // each quad as the triangles (0,1,2) and (0,2,3)
      mesh.faces[foffset + 6 * f + 0] = (unsigned int)(voffset + F[f][0]);
      mesh.faces[foffset + 6 * f + 1] = (unsigned int)(voffset + F[f][1]);
      mesh.faces[foffset + 6 * f + 2] = (unsigned int)(voffset + F[f][2]);
      mesh.faces[foffset + 6 * f + 3] = (unsigned int)(voffset + F[f][0]);
      mesh.faces[foffset + 6 * f + 4] = (unsigned int)(voffset + F[f][2]);
      mesh.faces[foffset + 6 * f + 5] = (unsigned int)(voffset + F[f][3]);
      mesh.materialIDs[foffset / 3 + 2 * f + 0] = (unsigned int)col;
      mesh.materialIDs[foffset / 3 + 2 * f + 1] = (unsigned int)col;
    }
    voffset += 8;
    foffset += 3 * 12;
  }
  return true; // as the reference: an empty grid loads and then fails in the BVH build
}

} // namespace mesh_io
