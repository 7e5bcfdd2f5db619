/*
 * abi_check.c -- lets a foreign-language binding verify that its mirror of the
 * public structs matches this build (sizes and the offsets of the members it
 * touches).  Returns (size_t)-1 for an unknown name.
 */
#include <stddef.h>
#include <string.h>

#include "kdtree.h"
#include "path_raytracer.h"
#include "raytracer.h"
#include "spath_raytracer.h"
#include "ss_raytracer.h"

#define SIZE_OF(T)  if (strcmp(name, "sizeof:" #T) == 0) return sizeof(T)
#define OFFSET_OF(T, m) if (strcmp(name, "offsetof:" #T "." #m) == 0) return offsetof(T, m)

size_t rt_abi_query(const char *name)
{
    SIZE_OF(rcl_ctx); SIZE_OF(rcl_program); SIZE_OF(rcl_img_buf); SIZE_OF(scene); SIZE_OF(kd_tree);
    SIZE_OF(kd_tree_flat); SIZE_OF(rt_kd_view); SIZE_OF(rt_scene_view); SIZE_OF(rt_spath_buffers);
    SIZE_OF(raytracer_context); SIZE_OF(spath_raytracer_context); SIZE_OF(path_raytracer_context);
    SIZE_OF(ss_raytracer_context); SIZE_OF(material); SIZE_OF(mesh); SIZE_OF(sphere); SIZE_OF(plane);
    SIZE_OF(ray); SIZE_OF(AABB); SIZE_OF(kd_tree_collision_result); SIZE_OF(rt_device_props);
    SIZE_OF(_skd_tree_traversal_node); SIZE_OF(_skd_tree_leaf_node);
    OFFSET_OF(raytracer_context, stat_scene); OFFSET_OF(raytracer_context, rcl);
    OFFSET_OF(raytracer_context, cl_ray_buffer); OFFSET_OF(raytracer_context, cl_ray_dir_buffer);
    OFFSET_OF(raytracer_context, tile_y0); OFFSET_OF(raytracer_context, tile_stride); OFFSET_OF(raytracer_context, num_samples);
    OFFSET_OF(scene, kdt); OFFSET_OF(scene, mesh_indices); OFFSET_OF(scene, num_mesh_indices);
    OFFSET_OF(kd_tree, bounds); OFFSET_OF(kd_tree, buffer); OFFSET_OF(kd_tree, flat); OFFSET_OF(kd_tree, view);
    OFFSET_OF(kd_tree, build_threads);
    OFFSET_OF(spath_raytracer_context, cl_random_buffer); OFFSET_OF(spath_raytracer_context, rng_mode);
    OFFSET_OF(spath_raytracer_context, quiet); OFFSET_OF(spath_raytracer_context, max_batch);
    OFFSET_OF(spath_raytracer_context, cl_random_value_buffer); OFFSET_OF(spath_raytracer_context, no_reorder);
    OFFSET_OF(spath_raytracer_context, trial_ms); OFFSET_OF(spath_raytracer_context, run_ahead);
    OFFSET_OF(spath_raytracer_context, cl_snapshot_buffer); OFFSET_OF(spath_raytracer_context, launches);
    OFFSET_OF(spath_raytracer_context, span_ms);
    OFFSET_OF(path_raytracer_context, cl_path_output_buffer);
    OFFSET_OF(mesh, material_index); OFFSET_OF(material, reflectivity);
    OFFSET_OF(sphere, material_index); OFFSET_OF(plane, material_index);
    return (size_t)-1;
}
