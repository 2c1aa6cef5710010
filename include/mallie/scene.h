// Forwarding header: lets sources written against lighttransport/mallie's "scene.h" build against the MI355X path.
#ifndef MALLIE_MI355X_FWD_SCENE_H_
#define MALLIE_MI355X_FWD_SCENE_H_
#include "mallie_api.hpp"
#endif
