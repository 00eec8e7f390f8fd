// mitsuba_loader.cpp — see mitsuba_loader.h (implementation lands with SURVEY §8f n1).
#include "mitsuba_loader.h"
namespace ctl {
void parse_mitsuba_scene(scene_builder&, const char*, int32_t*, int32_t*) { throw unsupported_error("ParseMitsubaScene: loader not built in this revision"); }
}
