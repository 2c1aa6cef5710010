// Forwarding header: lets sources written against lighttransport/mallie's "bvh_accel.h" build against the MI355X path.
#ifndef MALLIE_MI355X_FWD_BVH_ACCEL_H_
#define MALLIE_MI355X_FWD_BVH_ACCEL_H_
#include "mallie_api.hpp"
#endif
