// Forwarding header: lets sources written against lighttransport/mallie's "material.h" build against the MI355X path.
#ifndef MALLIE_MI355X_FWD_MATERIAL_H_
#define MALLIE_MI355X_FWD_MATERIAL_H_
#include "mallie_api.hpp"
#endif
