// rt_build.h — the scene's trees built on the GPU (rt_build.hip); same results as build_bvh / build_wide_bvh of host_scene.h.
#pragma once
#include <string>

#include "host_scene.h"

namespace rt {

// BVH::build (src/scene/mod.rs:99) on the current device.  Returns false with a reason when the input is outside what the device builder
// reproduces exactly (the caller then builds on the host).  *ms: device time, uploads and downloads included.
bool device_build_bvh(const HostScene& sc, FlatBVH* out, double* ms, std::string* err);
// The 8-wide occlusion tree from the binary tree, on the current device.
bool device_build_wide_bvh(const HostScene& sc, const FlatBVH& bvh, WideBVH* out, double* ms, std::string* err);

}  // namespace rt
