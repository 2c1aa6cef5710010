// Forwarding header: lets sources written against lighttransport/mallie's "prim-plane.h" build against the MI355X path.
#ifndef MALLIE_MI355X_FWD_PRIM_PLANE_H_
#define MALLIE_MI355X_FWD_PRIM_PLANE_H_
#include "mallie_api.hpp"
#endif
