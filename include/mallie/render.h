// Forwarding header: lets sources written against lighttransport/mallie's "render.h" build against the MI355X path.
#ifndef MALLIE_MI355X_FWD_RENDER_H_
#define MALLIE_MI355X_FWD_RENDER_H_
#include "mallie_api.hpp"
#endif
