/*
 * scene.h -- scene container shared between host and device.
 *
 * Drop-in for the reference's include/scene.h:1-87: same records, member names
 * and the two entry points the integrators call.
 */
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <stdbool.h>
#include "geom.h"
#include "parallel.h"

typedef struct _rt_ctx raytracer_context;
struct kd_tree;

/* reference: include/scene.h:7-14, 32 bytes */
typedef struct {
    vec4  colour;
    float reflectivity;
} RT_ALIGNED(16) material;

/* reference: include/scene.h:18-29, 128 bytes */
typedef struct {
    mat4 model;
    vec4 max;
    vec4 min;
    int index_offset;
    int num_indices;
    int material_index;
} RT_ALIGNED(32) mesh;

/* reference: include/scene.h:31-82 */
typedef struct scene {
    mat4 camera_world_matrix;

    material    *materials;
    rt_mem       cl_material_buffer;
    unsigned int num_materials;
    bool         materials_changed;

    sphere      *spheres;
    rt_mem       cl_sphere_buffer;
    unsigned int num_spheres;
    bool         spheres_changed;

    plane       *planes;
    rt_mem       cl_plane_buffer;
    unsigned int num_planes;
    bool         planes_changed;

    mesh        *meshes;
    rt_mem       cl_mesh_buffer;
    unsigned int num_meshes;
    bool         meshes_changed;

    struct kd_tree *kdt;

    vec3        *mesh_verts;
    rcl_img_buf  cl_mesh_vert_buffer;
    unsigned int num_mesh_verts;

    vec3        *mesh_nrmls;
    rcl_img_buf  cl_mesh_nrml_buffer;
    unsigned int num_mesh_nrmls;

    vec2        *mesh_texcoords;
    rcl_img_buf  cl_mesh_texcoord_buffer;
    unsigned int num_mesh_texcoords;

    ivec3       *mesh_indices;       /* {v, vn, vt, mesh} per index */
    rcl_img_buf  cl_mesh_index_buffer;
    unsigned int num_mesh_indices;
} scene;

/* reference: src/scene.c:7-65 and 68-145 */
void scene_init_resources(raytracer_context *rctx);
void scene_resource_push(raytracer_context *rctx);
void scene_generate_resources(raytracer_context *rctx);   /* declared by the reference, never defined there */

/* Builds a scene from caller-owned arrays in the layouts above (the arrays
   are copied).  The reference fills the same members from .rsc/.obj files
   (src/loader.c:228-450); this entry point is for callers that already hold
   the data. */
scene *scene_from_arrays(const material *materials, unsigned int num_materials,
                         const sphere *spheres, unsigned int num_spheres,
                         const plane *planes, unsigned int num_planes,
                         const mesh *meshes, unsigned int num_meshes,
                         const float *verts4, unsigned int num_verts,
                         const float *normals4, unsigned int num_normals,
                         const int *indices4, unsigned int num_indices);
void scene_free(scene *s);

#endif
