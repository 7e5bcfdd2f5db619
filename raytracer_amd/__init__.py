"""raytracer_amd -- MI355X-native hot path of the kd-tree path tracer.

The product is ``libraytracer_amd.so``: C host code exporting the reference
renderer's C API (``include/raytracer.h``, ``spath_raytracer.h``,
``ss_raytracer.h``, ``path_raytracer.h``, ``scene.h``, ``kdtree.h``,
``parallel.h``) on top of hand-written HIP kernels reached through the C-ABI
layer ``include/rt_hip.h``.  This module is a thin ctypes binding used by the
tests and the benchmark; it mirrors the public structs, checks that mirror
against the library at import time, and fails loudly when the library is
missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import scenes  # noqa: F401  (re-exported)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIB_PATH: another build of the library (A/B timing of two builds in one session)
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(_HERE, "libraytracer_amd.so")

SS_RAYTRACER, PATH_RAYTRACER, SPLIT_PATH_RAYTRACER = 0, 1, 2
RT_RNG_LIBC, RT_RNG_SPLITMIX = 0, 1

KERNEL_IDS = {"generate_rays": 0, "kd_intersect": 1, "spath_init": 2, "spath_trace": 3, "xorshift": 4,
              "resolve_avg": 5, "cast_rays": 6, "path_trace": 7, "f_buffer_average": 8, "f_to_byte": 9,
              "spath_run": 10, "pixel_order": 11, "debug_draw": 12}

HIT_DT = np.dtype([("tri", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])

vec4 = C.c_float * 4
mat4 = C.c_float * 16


class rcl_ctx(C.Structure):
    _fields_ = [("device_ordinal", C.c_int), ("device", C.c_void_p), ("simt_size", C.c_uint),
                ("num_simt_per_multiprocessor", C.c_uint), ("num_multiprocessors", C.c_uint),
                ("num_cores", C.c_uint)]


class rcl_program(C.Structure):
    _fields_ = [("program", C.c_void_p), ("raw_kernels", C.c_void_p), ("raw_data", C.c_void_p)]


class rcl_img_buf(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("image", C.c_void_p), ("size", C.c_size_t)]


class AABB(C.Structure):
    _fields_ = [("max", vec4), ("min", vec4)]


class kd_tree_flat(C.Structure):
    _fields_ = [("nodes", C.POINTER(C.c_uint32)), ("leaf_tris", C.POINTER(C.c_float)), ("num_nodes", C.c_uint32),
                ("num_leaf_lines", C.c_uint32), ("root_ref", C.c_uint32), ("max_depth", C.c_uint32),
                ("plain_splits", C.c_uint32)]


class rt_kd_view(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("leaf_tris", C.c_void_p), ("tri_shade", C.c_void_p),
                ("num_nodes", C.c_uint32), ("num_leaf_lines", C.c_uint32), ("num_tris", C.c_uint32),
                ("root_ref", C.c_uint32), ("scene_min", C.c_float * 3), ("scene_max", C.c_float * 3),
                ("max_depth", C.c_uint32), ("plain_splits", C.c_uint32)]


class rt_spath_buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ray_o", "ray_d", "origin_o", "origin_d", "hits", "source_hits",
                                          "prog_a", "prog_b", "random", "out_tex")]


class scene_t(C.Structure):
    pass


class kd_tree(C.Structure):
    _fields_ = [("root", C.c_void_p), ("k", C.c_uint), ("num_nodes_total", C.c_uint),
                ("num_tris_padded", C.c_uint), ("num_traversal_nodes", C.c_uint), ("num_leaves", C.c_uint),
                ("num_indices_total", C.c_uint), ("max_recurse", C.c_uint), ("tri_for_leaf_threshold", C.c_uint),
                ("s", C.POINTER(scene_t)), ("bounds", AABB), ("buffer", C.POINTER(C.c_uint8)),
                ("buffer_size", C.c_uint), ("cl_kd_tree_buffer", C.c_void_p), ("flat", kd_tree_flat),
                ("view", rt_kd_view), ("arena", C.c_void_p), ("build_threads", C.c_uint)]


scene_t._fields_ = [
    ("camera_world_matrix", mat4),
    ("materials", C.c_void_p), ("cl_material_buffer", C.c_void_p), ("num_materials", C.c_uint), ("materials_changed", C.c_bool),
    ("spheres", C.c_void_p), ("cl_sphere_buffer", C.c_void_p), ("num_spheres", C.c_uint), ("spheres_changed", C.c_bool),
    ("planes", C.c_void_p), ("cl_plane_buffer", C.c_void_p), ("num_planes", C.c_uint), ("planes_changed", C.c_bool),
    ("meshes", C.c_void_p), ("cl_mesh_buffer", C.c_void_p), ("num_meshes", C.c_uint), ("meshes_changed", C.c_bool),
    ("kdt", C.POINTER(kd_tree)),
    ("mesh_verts", C.c_void_p), ("cl_mesh_vert_buffer", rcl_img_buf), ("num_mesh_verts", C.c_uint),
    ("mesh_nrmls", C.c_void_p), ("cl_mesh_nrml_buffer", rcl_img_buf), ("num_mesh_nrmls", C.c_uint),
    ("mesh_texcoords", C.c_void_p), ("cl_mesh_texcoord_buffer", rcl_img_buf), ("num_mesh_texcoords", C.c_uint),
    ("mesh_indices", C.c_void_p), ("cl_mesh_index_buffer", rcl_img_buf), ("num_mesh_indices", C.c_uint),
]


class raytracer_context(C.Structure):
    _fields_ = [("width", C.c_uint), ("height", C.c_uint), ("ray_buffer", C.c_void_p),
                ("output_buffer", C.POINTER(C.c_uint32)), ("stat_scene", C.POINTER(scene_t)),
                ("ic_ctx", C.c_void_p), ("block_size_y", C.c_uint), ("block_size_x", C.c_uint),
                ("event_stack", C.c_uint * 32), ("event_position", C.c_uint), ("render_complete", C.c_bool),
                ("rcl", C.POINTER(rcl_ctx)), ("program", C.POINTER(rcl_program)),
                ("cl_ray_buffer", C.c_void_p), ("cl_output_buffer", C.c_void_p),
                ("path_output_buffer", C.c_void_p), ("num_samples", C.c_uint), ("current_sample", C.c_uint),
                ("cl_path_output_buffer", C.c_void_p), ("cl_path_fresh_frame_buffer", C.c_void_p),
                ("cl_ray_dir_buffer", C.c_void_p), ("tile_y0", C.c_uint), ("full_height", C.c_uint),
                ("tile_band", C.c_uint), ("tile_stride", C.c_uint)]


class rt_rows(C.Structure):
    _fields_ = [("first", C.c_uint32), ("band", C.c_uint32), ("stride", C.c_uint32), ("full_height", C.c_uint32)]


class ss_raytracer_context(C.Structure):
    _fields_ = [("rctx", C.POINTER(raytracer_context)), ("up_to_date", C.c_bool)]


class path_raytracer_context(C.Structure):
    _fields_ = [("rctx", C.POINTER(raytracer_context)), ("up_to_date", C.c_bool), ("num_samples", C.c_uint),
                ("current_sample", C.c_uint), ("render_complete", C.c_bool), ("start_time", C.c_int),
                ("cl_path_output_buffer", C.c_void_p), ("cl_path_fresh_frame_buffer", C.c_void_p)]


class spath_raytracer_context(C.Structure):
    _fields_ = [("rctx", C.POINTER(raytracer_context)), ("up_to_date", C.c_bool), ("num_iterations", C.c_uint),
                ("current_iteration", C.c_uint), ("render_complete", C.c_bool), ("start_time", C.c_uint),
                ("random_buffer", C.POINTER(C.c_uint)),
                ("cl_path_output_buffer", C.c_void_p), ("cl_path_ray_origin_buffer", C.c_void_p),
                ("cl_path_collision_result_buffer", C.c_void_p), ("cl_spath_progress_buffer", C.c_void_p),
                ("cl_path_origin_collision_result_buffer", C.c_void_p), ("cl_random_buffer", C.c_void_p),
                ("cl_bad_api_design_buffer", C.c_void_p), ("cl_path_ray_origin_dir_buffer", C.c_void_p),
                ("cl_spath_progress_b_buffer", C.c_void_p), ("rng_mode", C.c_int), ("rng_seed", C.c_uint),
                ("rng_launch", C.c_uint), ("resolve_interval", C.c_uint), ("separate_kernels", C.c_int),
                ("quiet", C.c_bool), ("max_batch", C.c_uint), ("pending", C.c_uint),
                ("pending_values", C.POINTER(C.c_uint)), ("cl_random_value_buffer", C.c_void_p),
                ("cl_pixel_cost_buffer", C.c_void_p), ("cl_pixel_order_buffer", C.c_void_p),
                ("cl_cost_hist_buffer", C.c_void_p), ("cl_tile_scratch_buffer", C.c_void_p),
                ("order_valid", C.c_bool), ("no_reorder", C.c_int),
                ("schedule", C.c_int), ("schedule_fixed", C.c_int), ("trial_state", C.c_int),
                ("trial_ms", C.c_float * 5), ("calibration_iterations", C.c_uint),
                ("run_ahead", C.c_int), ("launched_upto", C.c_uint), ("cl_snapshot_buffer", C.c_void_p),
                ("snapshot_slots", C.c_uint), ("snap_first_iteration", C.c_uint), ("snap_count", C.c_uint),
                ("launches", C.c_uint), ("values_uploaded", C.c_void_p * 2),
                ("span_marks", C.c_void_p * 128), ("span_count", C.c_uint), ("span_launches", C.c_uint),
                ("span_ms", C.c_double)]


class rt_device_props(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("compute_units", C.c_int), ("wave_size", C.c_int),
                ("max_waves_per_cu", C.c_int), ("global_mem_bytes", C.c_size_t), ("lds_bytes_per_cu", C.c_size_t)]


_ABI_STRUCTS = {"rcl_ctx": rcl_ctx, "rcl_program": rcl_program, "rcl_img_buf": rcl_img_buf, "scene": scene_t,
                "kd_tree": kd_tree, "kd_tree_flat": kd_tree_flat, "rt_kd_view": rt_kd_view,
                "rt_spath_buffers": rt_spath_buffers, "raytracer_context": raytracer_context,
                "spath_raytracer_context": spath_raytracer_context,
                "path_raytracer_context": path_raytracer_context, "ss_raytracer_context": ss_raytracer_context,
                "AABB": AABB, "rt_device_props": rt_device_props}
_ABI_OFFSETS = [("raytracer_context", "stat_scene"), ("raytracer_context", "rcl"),
                ("raytracer_context", "cl_ray_buffer"), ("raytracer_context", "cl_ray_dir_buffer"),
                ("raytracer_context", "tile_y0"), ("raytracer_context", "tile_stride"), ("raytracer_context", "num_samples"),
                ("scene", "kdt"), ("scene", "mesh_indices"), ("scene", "num_mesh_indices"),
                ("kd_tree", "bounds"), ("kd_tree", "buffer"), ("kd_tree", "flat"), ("kd_tree", "view"),
                ("kd_tree", "build_threads"), ("spath_raytracer_context", "cl_random_buffer"),
                ("spath_raytracer_context", "rng_mode"), ("spath_raytracer_context", "quiet"), ("spath_raytracer_context", "max_batch"),
                ("spath_raytracer_context", "cl_random_value_buffer"), ("spath_raytracer_context", "no_reorder"),
                ("spath_raytracer_context", "trial_ms"), ("spath_raytracer_context", "run_ahead"),
                ("spath_raytracer_context", "cl_snapshot_buffer"), ("spath_raytracer_context", "launches"),
                ("spath_raytracer_context", "span_ms"),
                ("path_raytracer_context", "cl_path_output_buffer")]

# every symbol the headers under include/ declare
EXPORTED_SYMBOLS = [
    # rt_hip.h
    "rt_device_count", "rt_device_open", "rt_device_close", "rt_device_get_props", "rt_device_sync",
    "rt_device_stream", "rt_device_ordinal", "rt_device_select_stream", "rt_mem_alloc", "rt_mem_free", "rt_mem_upload", "rt_mem_download",
    "rt_mem_zero", "rt_mem_copy", "rt_host_alloc", "rt_host_free", "rt_mem_upload_queued", "rt_event_wait", "rt_event_record", "rt_event_elapsed_ms", "rt_event_release", "rt_timing_enable", "rt_timing_reset", "rt_timing_get", "rt_kernel_name",
    "rt_k_generate_rays", "rt_k_kd_intersect", "rt_k_kd_intersect_noalloc", "rt_k_spath_init", "rt_k_spath_trace", "rt_k_xorshift",
    "rt_k_spath_run", "rt_k_spath_run_snap", "rt_k_pixel_order", "rt_k_resolve_avg", "rt_k_kd_ray_draw", "rt_k_kd_test_draw", "rt_k_cast_rays", "rt_k_path_trace", "rt_k_f_buffer_average",
    "rt_k_f_to_byte",
    # loader.h
    "load_scene_json", "load_scene_json_url", "load_file", "rt_set_resource_root", "rt_obj_parse", "rt_obj_free",
    # parallel.h
    "cl_info", "create_context", "load_program_raw", "load_program_url", "gen_rgb_image", "gen_grayscale_buffer",
    "gen_1d_image", "gen_1d_image_buffer", "retrieve_buf", "zero_buffer", "zero_buffer_img", "get_workgroup_size",
    "rt_time_ms",
    # geom.h
    "AABB_divide", "AABB_divide_world", "AABB_surface_area", "AABB_clip", "AABB_ilerp", "AABB_is_planar",
    "AABB_construct_from_vertices", "AABB_construct_from_triangle", "does_collide_sphere", "does_collide_plane",
    "generate_ray", "matvec_mul",
    # scene.h
    "scene_init_resources", "scene_resource_push", "scene_generate_resources", "scene_from_arrays", "scene_free",
    # kdtree.h
    "kd_tree_init", "kd_tree_node_init", "kd_tree_node_is_leaf", "kd_tree_construct", "kd_tree_generate_serialized",
    "kd_tree_flatten", "kd_tree_free",
    # raytracer.h
    "raytracer_init", "raytracer_build", "raytracer_prepass", "raytracer_render", "raytracer_refined_render",
    "_raytracer_gen_ray_buffer", "_raytracer_path_trace", "_raytracer_average_buffers", "_raytracer_push_path",
    "_raytracer_cast_rays", "raytracer_set_tile", "raytracer_set_tile_bands", "raytracer_get_rows", "raytracer_destroy",
    # ss_raytracer.h / path_raytracer.h / spath_raytracer.h
    "init_ss_raytracer_context", "ss_raytracer_render", "ss_raytracer_prepass",
    "init_path_raytracer_context", "path_raytracer_render", "path_raytracer_prepass",
    "init_spath_raytracer_context", "spath_raytracer_render", "spath_raytracer_prepass",
    "spath_raytracer_ray_test", "spath_raytracer_kd_test", "spath_raytracer_save", "spath_raytracer_restore",
    "spath_raytracer_resolve", "spath_raytracer_flush", "spath_raytracer_fill_buffers", "spath_raytracer_calibrate",
    "ss_raytracer_destroy", "path_raytracer_destroy", "spath_raytracer_destroy",
]

_lib = None


def build(force: bool = False) -> None:
    """Compile libraytracer_amd.so in-tree (gcc + hipcc --offload-arch=gfx950)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)


def lib():
    """The loaded C library.  Raises if it has not been built: the product has
    no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or make -C raytracer_amd/csrc); there is no CPU fallback")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.rt_abi_query.restype = C.c_size_t
    L.rt_abi_query.argtypes = [C.c_char_p]
    for name, st in _ABI_STRUCTS.items():
        want = L.rt_abi_query(f"sizeof:{name}".encode())
        if want != C.sizeof(st):
            raise RuntimeError(f"ABI mismatch: sizeof({name}) is {want} in the library, {C.sizeof(st)} in the binding")
    for sname, member in _ABI_OFFSETS:
        want = L.rt_abi_query(f"offsetof:{sname}.{member}".encode())
        have = getattr(_ABI_STRUCTS[sname], member).offset
        if want != have:
            raise RuntimeError(f"ABI mismatch: offsetof({sname}.{member}) is {want} in the library, {have} in the binding")

    vp, u32 = C.c_void_p, C.c_uint32
    L.rt_device_count.restype = C.c_int
    L.rt_device_open.restype = vp; L.rt_device_open.argtypes = [C.c_int]
    L.rt_device_close.argtypes = [vp]
    L.rt_device_get_props.argtypes = [vp, C.POINTER(rt_device_props)]
    L.rt_device_sync.argtypes = [vp]
    L.rt_device_stream.restype = vp; L.rt_device_stream.argtypes = [vp]
    L.rt_device_select_stream.argtypes = [vp, C.c_int]
    L.rt_mem_alloc.restype = vp; L.rt_mem_alloc.argtypes = [vp, C.c_size_t]
    L.rt_mem_free.argtypes = [vp, vp]
    L.rt_mem_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.rt_mem_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.rt_mem_zero.argtypes = [vp, vp, C.c_size_t]
    L.rt_mem_copy.argtypes = [vp, vp, vp, C.c_size_t]
    L.rt_timing_enable.argtypes = [vp, C.c_int]
    L.rt_timing_reset.argtypes = [vp]
    L.rt_timing_get.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.rt_kernel_name.restype = C.c_char_p; L.rt_kernel_name.argtypes = [C.c_int]
    L.load_scene_json.restype = C.POINTER(scene_t); L.load_scene_json.argtypes = [C.c_char_p]
    L.load_scene_json_url.restype = C.POINTER(scene_t); L.load_scene_json_url.argtypes = [C.c_char_p]
    L.rt_set_resource_root.argtypes = [C.c_char_p]
    L.rt_obj_parse.argtypes = [C.POINTER(rt_obj), C.c_char_p, C.c_size_t]
    L.rt_obj_free.argtypes = [C.POINTER(rt_obj)]
    L.rt_k_generate_rays.argtypes = [vp, vp, vp, u32, u32, C.POINTER(rt_rows), C.c_float]
    L.rt_k_kd_intersect.argtypes = [vp, C.POINTER(rt_kd_view), vp, vp, vp, u32]
    L.rt_k_kd_intersect_noalloc.restype = C.c_int
    L.rt_k_kd_intersect_noalloc.argtypes = [vp, C.POINTER(rt_kd_view), vp, vp, vp, u32]
    L.rt_k_spath_init.argtypes = [vp, C.POINTER(rt_kd_view), C.POINTER(rt_spath_buffers), u32, u32, C.POINTER(rt_rows), u32]
    L.rt_k_spath_trace.argtypes = [vp, C.POINTER(rt_kd_view), C.POINTER(rt_spath_buffers), u32, u32, C.c_int]
    L.rt_k_xorshift.argtypes = [vp, vp, u32]
    L.rt_k_spath_run.argtypes = [vp, C.POINTER(rt_kd_view), C.POINTER(rt_spath_buffers), u32, u32, vp, vp, vp, C.c_int]
    L.rt_k_spath_run_snap.argtypes = [vp, C.POINTER(rt_kd_view), C.POINTER(rt_spath_buffers), u32, u32, vp, vp, vp, C.c_int,
                                      vp, u32, u32]
    L.rt_k_pixel_order.argtypes = [vp, vp, vp, vp, u32, u32, u32, vp]
    L.rt_k_resolve_avg.argtypes = [vp, vp, vp, u32]
    L.rt_k_kd_ray_draw.argtypes = [vp, vp, vp, u32]
    L.rt_k_kd_test_draw.argtypes = [vp, C.POINTER(rt_kd_view), vp, vp, u32]
    L.rt_k_f_buffer_average.argtypes = [vp, vp, vp, u32, u32]
    L.rt_k_f_to_byte.argtypes = [vp, vp, vp, u32]

    L.create_context.argtypes = [C.POINTER(rcl_ctx)]
    L.scene_from_arrays.restype = C.POINTER(scene_t)
    L.scene_from_arrays.argtypes = [vp, C.c_uint, vp, C.c_uint, vp, C.c_uint, vp, C.c_uint, vp, C.c_uint,
                                    vp, C.c_uint, vp, C.c_uint]
    L.scene_free.argtypes = [C.POINTER(scene_t)]
    L.kd_tree_init.restype = C.POINTER(kd_tree)
    for f in ("kd_tree_construct", "kd_tree_generate_serialized", "kd_tree_flatten", "kd_tree_free"):
        getattr(L, f).argtypes = [C.POINTER(kd_tree)]
    L.raytracer_init.restype = C.POINTER(raytracer_context)
    L.raytracer_init.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_uint32), C.POINTER(rcl_ctx)]
    for f in ("raytracer_build", "raytracer_prepass", "raytracer_render", "raytracer_refined_render",
              "_raytracer_gen_ray_buffer", "_raytracer_push_path", "_raytracer_cast_rays", "raytracer_destroy"):
        getattr(L, f).argtypes = [C.POINTER(raytracer_context)]
    L._raytracer_path_trace.argtypes = [C.POINTER(raytracer_context), C.c_uint]
    L._raytracer_average_buffers.argtypes = [C.POINTER(raytracer_context), C.c_uint]
    L.raytracer_set_tile.argtypes = [C.POINTER(raytracer_context), C.c_uint, C.c_uint]
    L.raytracer_set_tile_bands.argtypes = [C.POINTER(raytracer_context), C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    L.raytracer_get_rows.argtypes = [C.POINTER(raytracer_context), C.POINTER(rt_rows)]
    L.init_ss_raytracer_context.restype = C.POINTER(ss_raytracer_context)
    L.init_ss_raytracer_context.argtypes = [C.POINTER(raytracer_context)]
    L.ss_raytracer_render.argtypes = [C.POINTER(ss_raytracer_context)]
    L.ss_raytracer_prepass.argtypes = [C.POINTER(ss_raytracer_context)]
    L.init_path_raytracer_context.restype = C.POINTER(path_raytracer_context)
    L.init_path_raytracer_context.argtypes = [C.POINTER(raytracer_context)]
    L.path_raytracer_render.argtypes = [C.POINTER(path_raytracer_context)]
    L.path_raytracer_prepass.argtypes = [C.POINTER(path_raytracer_context)]
    L.init_spath_raytracer_context.restype = C.POINTER(spath_raytracer_context)
    L.init_spath_raytracer_context.argtypes = [C.POINTER(raytracer_context)]
    for f in ("spath_raytracer_render", "spath_raytracer_prepass", "spath_raytracer_resolve", "spath_raytracer_flush",
              "spath_raytracer_calibrate", "spath_raytracer_ray_test", "spath_raytracer_kd_test"):
        getattr(L, f).argtypes = [C.POINTER(spath_raytracer_context)]
    for f in ("spath_raytracer_save", "spath_raytracer_restore"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [C.POINTER(spath_raytracer_context), C.c_char_p]
    L.spath_raytracer_fill_buffers.argtypes = [C.POINTER(spath_raytracer_context), C.POINTER(rt_spath_buffers)]
    L.spath_raytracer_destroy.argtypes = [C.POINTER(spath_raytracer_context)]
    L.path_raytracer_destroy.argtypes = [C.POINTER(path_raytracer_context)]
    L.ss_raytracer_destroy.argtypes = [C.POINTER(ss_raytracer_context)]
    _lib = L
    return L


def _vp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class rt_obj(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("num_vertices", C.c_uint),
                ("normals", C.POINTER(C.c_float)), ("num_normals", C.c_uint),
                ("texcoords", C.POINTER(C.c_float)), ("num_texcoords", C.c_uint),
                ("faces", C.POINTER(C.c_int)), ("num_faces", C.c_uint),
                ("shape_face_offset", C.POINTER(C.c_uint)), ("shape_length", C.POINTER(C.c_uint)),
                ("num_shapes", C.c_uint), ("num_face_lines", C.c_uint), ("num_nontriangle_lines", C.c_uint)]


def obj_parse(data: bytes) -> dict:
    """One OBJ file through the library's reader (rt_obj_parse); `data` is the file's bytes."""
    L = lib()
    buf = data + b"\0"
    o = rt_obj()
    rc = L.rt_obj_parse(C.byref(o), buf, len(buf))
    if rc != 0:
        raise RuntimeError(f"rt_obj_parse returned {rc}")

    def arr(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
    out = {
        "vertices": arr(o.vertices, 3 * o.num_vertices, np.float32).reshape(-1, 3),
        "normals": arr(o.normals, 3 * o.num_normals, np.float32).reshape(-1, 3),
        "texcoords": arr(o.texcoords, 2 * o.num_texcoords, np.float32).reshape(-1, 2),
        "faces": arr(o.faces, 3 * o.num_faces, np.int32).reshape(-1, 3),
        "shape_face_offset": arr(o.shape_face_offset, o.num_shapes, np.uint32),
        "shape_length": arr(o.shape_length, o.num_shapes, np.uint32),
        "num_face_lines": int(o.num_face_lines), "num_nontriangle_lines": int(o.num_nontriangle_lines),
    }
    L.rt_obj_free(C.byref(o))
    return out


def load_scene(url: str | None, resource_root: str, text: str | None = None) -> "scenes.SceneArrays":
    """A .rsc scene file (and the OBJ meshes it names, both relative to
    `resource_root`) through load_scene_json_url -- or, with `text`, a scene
    description held in memory through load_scene_json -- copied into SceneArrays."""
    L = lib()
    L.rt_set_resource_root(resource_root.encode())
    sp = L.load_scene_json(text.encode()) if text is not None else L.load_scene_json_url(url.encode())
    s = sp.contents

    def rec(ptr, n, dt):
        if not n:
            return np.zeros(0, dt)
        raw = C.string_at(ptr, n * np.dtype(dt).itemsize)
        return np.frombuffer(raw, dt).copy()
    out = scenes.SceneArrays(
        verts4=rec(s.mesh_verts, s.num_mesh_verts * 4, np.float32).reshape(-1, 4),
        normals4=rec(s.mesh_nrmls, s.num_mesh_nrmls * 4, np.float32).reshape(-1, 4),
        idx4=rec(s.mesh_indices, s.num_mesh_indices * 4, np.int32).reshape(-1, 4),
        meshes=rec(s.meshes, s.num_meshes, scenes.MESH_DT),
        materials=rec(s.materials, s.num_materials, scenes.MATERIAL_DT),
        spheres=rec(s.spheres, s.num_spheres, scenes.SPHERE_DT),
        planes=rec(s.planes, s.num_planes, scenes.PLANE_DT))
    out.texcoords = rec(s.mesh_texcoords, s.num_mesh_texcoords * 2, np.float32).reshape(-1, 2)
    L.scene_free(sp)
    L.rt_set_resource_root(None)
    return out


def make_scene(sc: "scenes.SceneArrays"):
    """scene* (host side) from SceneArrays; owned by the caller (scene_free)."""
    L = lib()
    return L.scene_from_arrays(_vp(sc.materials), sc.materials.shape[0], _vp(sc.spheres), sc.spheres.shape[0],
                               _vp(sc.planes), sc.planes.shape[0], _vp(sc.meshes), sc.meshes.shape[0],
                               _vp(sc.verts4), sc.verts4.shape[0], _vp(sc.normals4), sc.normals4.shape[0],
                               _vp(sc.idx4), sc.idx4.shape[0])


class HostKdTree:
    """kd-tree built by the library on the host only (no GPU needed): the
    reference's wire format and the flattened device layout."""

    def __init__(self, sc: "scenes.SceneArrays", threads: int = 0):
        L = lib()
        self._scene = make_scene(sc)
        self._t = L.kd_tree_init()
        self._t.contents.s = self._scene
        self._t.contents.build_threads = threads
        L.kd_tree_construct(self._t)
        L.kd_tree_generate_serialized(self._t)
        L.kd_tree_flatten(self._t)
        t = self._t.contents
        self.counts = {"nodes_total": t.num_nodes_total, "traversal": t.num_traversal_nodes,
                       "leaves": t.num_leaves, "indices_total": t.num_indices_total,
                       "tris_padded": t.num_tris_padded}
        self.blob = np.ctypeslib.as_array(t.buffer, shape=(max(int(t.buffer_size), 1),)).copy()[:t.buffer_size]
        self.bounds_min = np.array(t.bounds.min[:3], np.float32)
        self.bounds_max = np.array(t.bounds.max[:3], np.float32)
        f = t.flat
        self.nodes = np.ctypeslib.as_array(f.nodes, shape=(max(f.num_nodes, 1), 8)).copy()[:f.num_nodes]
        self.leaf_tris = np.ctypeslib.as_array(f.leaf_tris, shape=(max(f.num_leaf_lines, 1), 16)).copy()[:f.num_leaf_lines]   # 64-byte lines
        self.root_ref = int(f.root_ref)

    def __del__(self):
        try:
            L = lib()
            L.kd_tree_free(self._t)
            L.scene_free(self._scene)
        except Exception:
            pass


class Renderer:
    """One raytracer_context on one GPU, driven exactly as the reference's
    render thread drives it (src/startup.c:79-218): create_context ->
    raytracer_init -> stat_scene -> raytracer_build -> init_X / X_prepass /
    X_render."""

    def __init__(self, sc: "scenes.SceneArrays", width: int, height: int, device: int | None = None,
                 tile_y0: int = 0, full_height: int | None = None, build_threads: int = 0,
                 bands: tuple[int, int] | None = None):
        """`height` local rows.  full_height given: they are rows tile_y0.. of a frame that high (a contiguous
        strip), or, with bands=(band, stride), the interleaved rows tile_y0 + (j // band) * stride + j % band."""
        L = lib()
        self.L = L
        self.sc = sc
        self.width, self.height = width, height
        if device is not None:
            os.environ["RT_DEVICE_ORDINAL"] = str(device)
        self.rcl = rcl_ctx()
        L.create_context(C.byref(self.rcl))
        self.dev = self.rcl.device
        self.output = np.zeros(width * height, np.uint32)
        self.rctx = L.raytracer_init(width, height, self.output.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(self.rcl))
        self.scene = make_scene(sc)
        self.rctx.contents.stat_scene = self.scene
        if full_height is not None and bands is not None:
            L.raytracer_set_tile_bands(self.rctx, tile_y0, bands[0], bands[1], full_height)
        elif full_height is not None:
            L.raytracer_set_tile(self.rctx, tile_y0, full_height)
        self._build_threads = build_threads
        self.spath = None
        self.ss = None
        self.path = None
        self.built = False

    # -- protocol ---------------------------------------------------------
    def build(self):
        self.L.raytracer_build(self.rctx)
        self.built = True
        return self

    @property
    def kdt(self):
        return self.scene.contents.kdt.contents

    def init_spath(self, seed: int = 1, separate_kernels: bool = False, resolve_interval: int = 50, quiet: bool = True,
                   max_batch: int | None = None, reorder: bool = True, schedule: int | None = None,
                   run_ahead: bool = True):
        if not self.built:
            self.build()
        self.spath = self.L.init_spath_raytracer_context(self.rctx)
        c = self.spath.contents
        c.rng_mode = RT_RNG_SPLITMIX
        c.rng_seed = seed
        c.separate_kernels = 1 if separate_kernels else 0
        c.resolve_interval = resolve_interval
        c.quiet = quiet
        if max_batch is not None:
            c.max_batch = max_batch
        c.no_reorder = 0 if reorder else 1
        c.run_ahead = 1 if run_ahead else 0
        if schedule is not None:         # None: the prepass ends with a calibration of the schedules
            c.schedule = schedule
            c.schedule_fixed = 1
        return self.spath

    def spath_prepass(self, num_iterations: int | None = None):
        self.L.spath_raytracer_prepass(self.spath)
        if num_iterations is not None:
            self.spath.contents.num_iterations = num_iterations

    def spath_render(self, calls: int = 1, flush: bool = True):
        """`calls` render calls; by default the iterations still pending in the
        library's batch are launched before returning (not waited for)."""
        f = self.L.spath_raytracer_render
        c = self.spath.contents
        # The library may launch iterations ahead of the calls that ask for them, never past num_iterations:
        # while these calls run, the render ends (at the latest) where they end, so that the device holds
        # exactly the state after `calls` more iterations when they return.
        total = int(c.num_iterations)
        c.num_iterations = min(total, int(c.current_iteration) + calls)
        try:
            for _ in range(calls):
                f(self.spath)
            if flush:
                self.L.spath_raytracer_flush(self.spath)
        finally:
            c.num_iterations = total

    def spath_resolve(self) -> np.ndarray:
        self.L.spath_raytracer_resolve(self.spath)
        return self.output.copy()

    def spath_buffers(self) -> rt_spath_buffers:
        b = rt_spath_buffers()
        self.L.spath_raytracer_fill_buffers(self.spath, C.byref(b))
        return b

    def ss_render(self) -> np.ndarray:
        if not self.built:
            self.build()
        if self.ss is None:
            self.ss = self.L.init_ss_raytracer_context(self.rctx)
            self.L.ss_raytracer_prepass(self.ss)
        self.L.ss_raytracer_render(self.ss)
        return self.output.copy()

    def init_path(self, num_samples: int = 128):
        if not self.built:
            self.build()
        self.path = self.L.init_path_raytracer_context(self.rctx)
        self.path.contents.num_samples = num_samples
        self.L.path_raytracer_prepass(self.path)
        return self.path

    def path_render(self) -> np.ndarray:
        self.L.path_raytracer_render(self.path)
        return self.output.copy()

    # -- device access ----------------------------------------------------
    def sync(self):
        self.L.rt_device_sync(self.dev)

    def download(self, mem, dtype, count: int) -> np.ndarray:
        out = np.zeros(count, dtype)
        self.L.rt_mem_download(self.dev, _vp(out), mem, out.nbytes)
        return out

    def upload(self, mem, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.L.rt_mem_upload(self.dev, mem, _vp(arr), arr.nbytes)

    def timing(self, enable: bool):
        self.L.rt_timing_enable(self.dev, 1 if enable else 0)

    def timing_reset(self):
        self.L.rt_timing_reset(self.dev)

    def timing_get(self, kernel: str):
        n, ms = C.c_uint64(), C.c_double()
        self.L.rt_timing_get(self.dev, KERNEL_IDS[kernel], C.byref(n), C.byref(ms))
        return int(n.value), float(ms.value)

    def device_props(self) -> dict:
        p = rt_device_props()
        self.L.rt_device_get_props(self.dev, C.byref(p))
        return {"name": p.name.decode(), "compute_units": p.compute_units, "wave_size": p.wave_size,
                "max_waves_per_cu": p.max_waves_per_cu, "global_mem_bytes": p.global_mem_bytes,
                "lds_bytes_per_cu": p.lds_bytes_per_cu}

    # state snapshots in the oracle's layouts ------------------------------
    def spath_state(self) -> dict:
        """Per-pixel buffers of the split path tracer, converted to the
        reference's record layouts (rays 8 floats, progress 12 words)."""
        n = self.width * self.height
        b = self.spath_buffers()
        f4 = lambda m: self.download(m, np.float32, 4 * n).reshape(n, 4)
        ro, rd = f4(b.ray_o), f4(b.ray_d)
        oo, od = f4(b.origin_o), f4(b.origin_d)
        pa, pb = f4(b.prog_a), f4(b.prog_b)
        prog = np.zeros((n, 12), np.uint32)
        prog[:, 0] = pa[:, 3].view(np.uint32)
        prog[:, 1] = pb[:, 3].view(np.uint32)
        prog[:, 4:7] = pa[:, :3].view(np.uint32)
        prog[:, 8:11] = pb[:, :3].view(np.uint32)
        return {"rays": np.concatenate([ro, rd], axis=1), "origin_rays": np.concatenate([oo, od], axis=1),
                "results": self.download(b.hits, HIT_DT, n), "source_results": self.download(b.source_hits, HIT_DT, n),
                "progress": prog, "random": self.download(b.random, np.uint32, n), "out_tex": f4(b.out_tex)}

    def close(self):
        if self.rctx is not None:
            if self.spath is not None:
                self.L.spath_raytracer_destroy(self.spath)
            if self.path is not None:
                self.L.path_raytracer_destroy(self.path)
            if self.ss is not None:
                self.L.ss_raytracer_destroy(self.ss)
            self.spath = self.path = self.ss = None
            self.L.raytracer_destroy(self.rctx)
            self.L.scene_free(self.scene)
            self.L.rt_device_close(self.dev)
            self.rctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
