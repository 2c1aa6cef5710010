// Forwarding header: lets sources written against lighttransport/mallie's "mesh.h" build against the MI355X path.
#ifndef MALLIE_MI355X_FWD_MESH_H_
#define MALLIE_MI355X_FWD_MESH_H_
#include "mallie_api.hpp"
#endif
