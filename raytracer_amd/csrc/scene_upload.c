/*
 * scene_upload.c -- moves the scene to the GPU.
 *
 * scene_init_resources uploads the scene arrays as they are (the brute-force
 * integrators read them); scene_resource_push serialises the kd-tree in the
 * reference's wire format for whoever reads kdt->buffer, flattens it into the
 * device layout of rt_hip.h and uploads that together with the per-triangle
 * shading table.  reference: src/scene.c:7-65 and 68-145.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kdtree.h"
#include "raytracer.h"
#include "scene.h"

static rt_mem upload_array(raytracer_context *rctx, const void *src, size_t bytes)
{
    rt_mem m = rt_mem_alloc(rctx->rcl->device, bytes);
    if (src && bytes) rt_mem_upload(rctx->rcl->device, m, src, bytes);
    return m;
}

static rcl_img_buf upload_img(raytracer_context *rctx, const void *src, size_t bytes)
{
    rcl_img_buf ib;
    ib.buffer = upload_array(rctx, src, bytes);
    ib.image = ib.buffer;
    ib.size = bytes;
    return ib;
}

void scene_init_resources(raytracer_context *rctx)
{
    scene *s = rctx->stat_scene;
    s->kdt = kd_tree_init();

    s->cl_sphere_buffer = upload_array(rctx, s->spheres, sizeof(sphere) * (size_t)s->num_spheres);
    s->cl_plane_buffer = upload_array(rctx, s->planes, sizeof(plane) * (size_t)s->num_planes);
    s->cl_material_buffer = upload_array(rctx, s->materials, sizeof(material) * (size_t)s->num_materials);
    s->cl_mesh_buffer = upload_array(rctx, s->meshes, sizeof(mesh) * (size_t)s->num_meshes);

    s->cl_mesh_vert_buffer = upload_img(rctx, s->mesh_verts, sizeof(vec3) * (size_t)s->num_mesh_verts);
    s->cl_mesh_nrml_buffer = upload_img(rctx, s->mesh_nrmls, sizeof(vec3) * (size_t)s->num_mesh_nrmls);
    s->cl_mesh_index_buffer = upload_img(rctx, s->mesh_indices, sizeof(ivec3) * (size_t)s->num_mesh_indices);
}

/* declared by the reference (include/scene.h:87) without a definition; here it
   is the build step scene_init_resources + kd_tree_construct */
void scene_generate_resources(raytracer_context *rctx)
{
    scene_init_resources(rctx);
    rctx->stat_scene->kdt->s = rctx->stat_scene;
    kd_tree_construct(rctx->stat_scene->kdt);
}

/* SCENE_MIN / SCENE_MAX reach the reference's kernels as text printed with %f
   and parsed back as a double literal (reference: src/raytracer.c:53-64), so
   the box the traversal clips against is that rounded copy of the bounds. */
static float through_percent_f(float v)
{
    char text[64];
    snprintf(text, sizeof text, "%f", v);
    return (float)strtod(text, NULL);
}

/* 64 bytes per triangle: {n0.xyz, colour.r}, {n1.xyz, colour.g}, {n2.xyz, colour.b}, {reflectivity,0,0,0} */
static float *build_tri_shade(const scene *s, unsigned int num_tris)
{
    float *t = (float *)calloc((size_t)(num_tris ? num_tris : 1) * 16, sizeof(float));
    for (unsigned int i = 0; i < num_tris; i++) {
        const int *c0 = s->mesh_indices[3 * (size_t)i];
        const int mesh_id = c0[3];
        if (mesh_id < 0 || (unsigned)mesh_id >= s->num_meshes) {
            fprintf(stderr, "ERROR: triangle %u names mesh %d of %u\n", i, mesh_id, s->num_meshes);
            exit(1);
        }
        const int mat_id = s->meshes[mesh_id].material_index;
        if (mat_id < 0 || (unsigned)mat_id >= s->num_materials) {
            fprintf(stderr, "ERROR: mesh %d names material %d of %u\n", mesh_id, mat_id, s->num_materials);
            exit(1);
        }
        const material *m = &s->materials[mat_id];
        float *o = t + 16 * (size_t)i;
        for (int c = 0; c < 3; c++) {
            const int ni = s->mesh_indices[3 * (size_t)i + c][1];
            if (ni < 0 || (unsigned)ni >= s->num_mesh_nrmls) {
                fprintf(stderr, "ERROR: triangle %u names normal %d of %u\n", i, ni, s->num_mesh_nrmls);
                exit(1);
            }
            o[4 * c + 0] = s->mesh_nrmls[ni][0];
            o[4 * c + 1] = s->mesh_nrmls[ni][1];
            o[4 * c + 2] = s->mesh_nrmls[ni][2];
            o[4 * c + 3] = m->colour[c];
        }
        o[12] = m->reflectivity;
    }
    return t;
}

void scene_resource_push(raytracer_context *rctx)
{
    scene *s = rctx->stat_scene;
    kd_tree *kdt = s->kdt;
    rt_device *dev = rctx->rcl->device;

    kd_tree_generate_serialized(kdt);

    const unsigned int num_tris = s->num_mesh_indices / 3;
    int first_push = (kdt->cl_kd_tree_buffer == NULL);
    if (first_push && kdt->root) {
        kd_tree_flatten(kdt);
        kdt->cl_kd_tree_buffer = upload_array(rctx, kdt->flat.nodes, sizeof(uint32_t) * 8 * (size_t)kdt->flat.num_nodes);
        kdt->view.nodes = kdt->cl_kd_tree_buffer;
        /* the kernels may read slot 0 / node 0 as a harmless dummy: keep both allocated */
        kdt->view.leaf_tris = rt_mem_alloc(dev, 64u * ((size_t)kdt->flat.num_leaf_lines + 1));
        rt_mem_zero(dev, kdt->view.leaf_tris, 64u * ((size_t)kdt->flat.num_leaf_lines + 1));
        if (kdt->flat.num_leaf_lines)
            rt_mem_upload(dev, kdt->view.leaf_tris, kdt->flat.leaf_tris, 64u * (size_t)kdt->flat.num_leaf_lines);
        kdt->view.num_nodes = kdt->flat.num_nodes;
        kdt->view.num_leaf_lines = kdt->flat.num_leaf_lines;
        kdt->view.num_tris = num_tris;
        kdt->view.root_ref = kdt->flat.root_ref;
        kdt->view.max_depth = kdt->flat.max_depth;
        kdt->view.plain_splits = kdt->flat.plain_splits;
        for (int k = 0; k < 3; k++) {
            kdt->view.scene_min[k] = through_percent_f(kdt->bounds.min[k]);
            kdt->view.scene_max[k] = through_percent_f(kdt->bounds.max[k]);
        }
    }
    if (!first_push && kdt->root && num_tris != kdt->view.num_tris) {
        /* the flattened tree, its triangle slots and the shading table describe the triangles of the build */
        fprintf(stderr, "ERROR: the scene has %u triangles now, the kd-tree was built for %u: geometry cannot change "
                        "after raytracer_build\n", num_tris, kdt->view.num_tris);
        exit(1);
    }
    if (kdt->root && (first_push || s->materials_changed || s->meshes_changed)) {
        float *table = build_tri_shade(s, num_tris);
        size_t bytes = sizeof(float) * 16 * (size_t)(num_tris ? num_tris : 1);
        if (!kdt->view.tri_shade) kdt->view.tri_shade = rt_mem_alloc(dev, bytes);
        rt_mem_upload(dev, kdt->view.tri_shade, table, bytes);
        free(table);
    }

    if (s->meshes_changed && s->num_meshes)
        rt_mem_upload(dev, s->cl_mesh_buffer, s->meshes, sizeof(mesh) * (size_t)s->num_meshes);
    if (s->spheres_changed && s->num_spheres)
        rt_mem_upload(dev, s->cl_sphere_buffer, s->spheres, sizeof(sphere) * (size_t)s->num_spheres);
    if (s->planes_changed && s->num_planes)
        rt_mem_upload(dev, s->cl_plane_buffer, s->planes, sizeof(plane) * (size_t)s->num_planes);
    if (s->materials_changed && s->num_materials)
        rt_mem_upload(dev, s->cl_material_buffer, s->materials, sizeof(material) * (size_t)s->num_materials);

    /* the reference leaves the flags set and re-uploads on every prepass; here
       they are consumed, so an unchanged scene is not pushed again */
    s->meshes_changed = s->spheres_changed = s->planes_changed = s->materials_changed = false;
}

static void *dup_bytes(const void *src, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (src && bytes) memcpy(p, src, bytes);
    return p;
}

scene *scene_from_arrays(const material *materials, unsigned int num_materials,
                         const sphere *spheres, unsigned int num_spheres,
                         const plane *planes, unsigned int num_planes,
                         const mesh *meshes, unsigned int num_meshes,
                         const float *verts4, unsigned int num_verts,
                         const float *normals4, unsigned int num_normals,
                         const int *indices4, unsigned int num_indices)
{
    scene *s = (scene *)calloc(1, sizeof(scene));
    for (int i = 0; i < 4; i++) s->camera_world_matrix[5 * i] = 1.0f;
    s->materials = (material *)dup_bytes(materials, sizeof(material) * (size_t)num_materials);
    s->num_materials = num_materials;
    s->spheres = (sphere *)dup_bytes(spheres, sizeof(sphere) * (size_t)num_spheres);
    s->num_spheres = num_spheres;
    s->planes = (plane *)dup_bytes(planes, sizeof(plane) * (size_t)num_planes);
    s->num_planes = num_planes;
    s->meshes = (mesh *)dup_bytes(meshes, sizeof(mesh) * (size_t)num_meshes);
    s->num_meshes = num_meshes;
    s->mesh_verts = (vec3 *)dup_bytes(verts4, sizeof(vec3) * (size_t)num_verts);
    s->num_mesh_verts = num_verts;
    s->mesh_nrmls = (vec3 *)dup_bytes(normals4, sizeof(vec3) * (size_t)num_normals);
    s->num_mesh_nrmls = num_normals;
    s->mesh_texcoords = (vec2 *)malloc(1);
    s->num_mesh_texcoords = 0;
    s->mesh_indices = (ivec3 *)dup_bytes(indices4, sizeof(ivec3) * (size_t)num_indices);
    s->num_mesh_indices = num_indices;
    s->materials_changed = s->spheres_changed = s->planes_changed = s->meshes_changed = true;
    return s;
}

void scene_free(scene *s)
{
    if (!s) return;
    free(s->materials); free(s->spheres); free(s->planes); free(s->meshes);
    free(s->mesh_verts); free(s->mesh_nrmls); free(s->mesh_texcoords); free(s->mesh_indices);
    free(s);
}
