"""mallie_amd -- MI355X-native render hot path of lighttransport/mallie (Render -> PathTrace -> BVHAccel::Traverse).

The compute lives in libmallie_mgpu.so (hand-written HIP for gfx950 behind the C ABI of include/mgpu.h); this package
is the thin Python host layer used by the tests, the benchmark and multi-GPU plumbing.  There is no CPU fallback:
importing works anywhere (the library only needs the HIP runtime), but every compute call needs a GPU.
"""
from .mgpu import (MgpuError, Scene, Stats, RNG_HASH, RNG_STREAM, RNG_TABLE, NODE_DT, RAY_DT, ISECT_DT, abi_version,
                   bvh_build, camera_frame, device_count, hash_state, lib_path, plane_from_bbox, load_library, tonemap_device,
                   TONEMAP_LINEAR_RGB8, TONEMAP_GAMMA22_BGRA8, Frame, frame_rows, frame_unique_id, frame_plan, frame_block_plan)

__all__ = ["MgpuError", "Scene", "Stats", "RNG_HASH", "RNG_STREAM", "RNG_TABLE", "NODE_DT", "RAY_DT", "ISECT_DT",
           "abi_version", "bvh_build", "camera_frame", "device_count", "hash_state", "lib_path", "plane_from_bbox",
           "load_library", "tonemap_device", "TONEMAP_LINEAR_RGB8", "TONEMAP_GAMMA22_BGRA8", "Frame", "frame_rows", "frame_unique_id", "frame_plan", "frame_block_plan"]
