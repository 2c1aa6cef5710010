// host/mesh_io.hpp -- mesh file readers of the facade (Scene::Init's inputs; MeshLoader in the reference, mesh_loader.h).
#pragma once
#include "../../../include/mallie/mallie_api.hpp"

namespace mesh_io {
// Both fill `mesh` with new[]-allocated arrays owned by the caller (Scene frees them) and return false on failure.
bool LoadObj(Mesh &mesh, const char *filename);  // MeshLoader::LoadObj,  mesh_loader.cc:26-210
bool LoadESON(Mesh &mesh, const char *filename); // MeshLoader::LoadESON, mesh_loader.cc:212-310
// MeshLoader::LoadMagicaVoxel, mesh_loader.cc:312-400: one cube per voxel, 256 palette materials
bool LoadMagicaVoxel(Mesh &mesh, std::vector<Material> &materials, const char *filename);
} // namespace mesh_io
