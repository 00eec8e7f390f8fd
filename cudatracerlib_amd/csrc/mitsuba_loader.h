// mitsuba_loader.h — Mitsuba-0.5 XML scene loader feeding the scene builder
// (ParseMitsubaScene, Engine/SceneLoader/Mitsuba/MitsubaLoader.h:13).
#pragma once
#include "scene_builder.h"
#include <stdexcept>
#include <cstdint>

namespace ctl {
struct io_error : std::runtime_error { using std::runtime_error::runtime_error; };
struct unsupported_error : std::runtime_error { using std::runtime_error::runtime_error; };
// width/height: in = override (<= 0: take the film size of the file), out = the size used
void parse_mitsuba_scene(scene_builder& b, const char* xml_path, int32_t* width_inout, int32_t* height_inout);
}
