/*
 * host_core.c -- the core renderer context: creation, scene build, primary-ray
 * generation and the legacy single-context render entry points.
 * reference: src/raytracer.c:9-340.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kdtree.h"
#include "raytracer.h"

#define RT_FOV_DEG 80.0f                 /* reference: src/kernels/util.cl:2 */
#define RT_PI_F 3.14159274101257f

/* reference: src/raytracer.c:9-29 */
raytracer_context *raytracer_init(unsigned int width, unsigned int height, uint32_t *output_buffer, rcl_ctx *rcl)
{
    raytracer_context *rctx = (raytracer_context *)calloc(1, sizeof(raytracer_context));
    rctx->width = width;
    rctx->height = height;
    rctx->ray_buffer = (float *)malloc((size_t)width * height * sizeof(ray));
    rctx->output_buffer = output_buffer;
    rctx->rcl = rcl;
    rctx->program = (rcl_program *)calloc(1, sizeof(rcl_program));
    rctx->ic_ctx = NULL;
    rctx->render_complete = false;
    rctx->num_samples = 64;
    rctx->current_sample = 0;
    rctx->event_position = 0;
    return rctx;
}

void raytracer_set_tile(raytracer_context *rctx, unsigned int y0, unsigned int full_height)
{
    rctx->tile_y0 = y0;
    rctx->full_height = full_height;
    rctx->tile_band = rctx->tile_stride = 0;
}

void raytracer_set_tile_bands(raytracer_context *rctx, unsigned int first, unsigned int band, unsigned int stride,
                              unsigned int full_height)
{
    rctx->tile_y0 = first;
    rctx->full_height = full_height;
    rctx->tile_band = band;
    rctx->tile_stride = stride;
}

void raytracer_get_rows(const raytracer_context *rctx, rt_rows *rows)
{
    rows->first = rctx->tile_y0;
    rows->band = rctx->tile_band;
    rows->stride = rctx->tile_stride;
    rows->full_height = rctx->full_height ? rctx->full_height : rctx->height;
}

/* reference: src/raytracer.c:81-126 (no kernels are compiled here: they ship prebuilt) */
void raytracer_build(raytracer_context *rctx)
{
    if (!rctx->stat_scene) { fprintf(stderr, "ERROR: raytracer_build: stat_scene is not set\n"); exit(1); }
    if (!rctx->rcl || !rctx->rcl->device) { fprintf(stderr, "ERROR: raytracer_build: no device context\n"); exit(1); }
    printf("Building Scene...\n");
    scene_init_resources(rctx);
    rctx->stat_scene->kdt->s = rctx->stat_scene;

    printf("Building/Rebuilding k-d tree.\n");
    unsigned int t0 = rt_time_ms();
    kd_tree_construct(rctx->stat_scene->kdt);
    printf("Done Building/Rebuilding k-d tree (%u ms, %u nodes).\n", rt_time_ms() - t0,
           rctx->stat_scene->kdt->num_nodes_total);

    rt_device *dev = rctx->rcl->device;
    const size_t n = (size_t)rctx->width * rctx->height;
    rctx->cl_ray_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    rctx->cl_ray_dir_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    rctx->cl_path_output_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    rctx->cl_output_buffer = rt_mem_alloc(dev, n * 4);
    rctx->cl_path_fresh_frame_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    fflush(stdout);
}

/* reference: src/raytracer.c:128-135 */
void raytracer_prepass(raytracer_context *rctx)
{
    scene_resource_push(rctx);
}

/* reference: src/raytracer.c:182-206 */
void _raytracer_gen_ray_buffer(raytracer_context *rctx)
{
    const float tan_half = tanf(RT_FOV_DEG / 2 * RT_PI_F / 180);
    rt_rows rows;
    raytracer_get_rows(rctx, &rows);
    rt_k_generate_rays(rctx->rcl->device, rctx->cl_ray_buffer, rctx->cl_ray_dir_buffer, rctx->width, rctx->height,
                       &rows, tan_half);
    rt_device_sync(rctx->rcl->device);
}

static void fill_scene_view(const raytracer_context *rctx, rt_scene_view *v)
{
    const scene *s = rctx->stat_scene;
    v->idx4 = s->cl_mesh_index_buffer.buffer;
    v->verts4 = s->cl_mesh_vert_buffer.buffer;
    v->normals4 = s->cl_mesh_nrml_buffer.buffer;
    v->meshes = s->cl_mesh_buffer;
    v->materials = s->cl_material_buffer;
    v->spheres = s->cl_sphere_buffer;
    v->planes = s->cl_plane_buffer;
    v->num_meshes = s->num_meshes;
    v->num_spheres = s->num_spheres;
    v->num_planes = s->num_planes;
    v->num_materials = s->num_materials;
}

/* shared by the legacy path and the SS integrator */
void rt_host_cast_rays(raytracer_context *rctx)
{
    rt_scene_view v;
    fill_scene_view(rctx, &v);
    const unsigned int n = rctx->width * rctx->height;
    rt_k_cast_rays(rctx->rcl->device, &v, rctx->cl_output_buffer, rctx->cl_ray_buffer, rctx->cl_ray_dir_buffer, n);
    rt_mem_download(rctx->rcl->device, rctx->output_buffer, rctx->cl_output_buffer, (size_t)n * 4);
}

/* shared by the legacy path and the PATH integrator; the reference splits the
   frame into 2x2 launches and refuses odd sizes (src/raytracer.c:243-298) */
void rt_host_path_trace(raytracer_context *rctx, rt_mem fresh_frame, unsigned int sample_num)
{
    if ((rctx->width / 2) * 2 * (rctx->height / 2) * 2 != rctx->width * rctx->height) {
        printf("Watchdog divisions are incorrect!\n");
        exit(1);
    }
    float origin[4] = {0.f, 0.f, 0.f, 1.f};
    float *cam = matvec_mul(rctx->stat_scene->camera_world_matrix, origin);
    rt_scene_view v;
    fill_scene_view(rctx, &v);
    rt_k_path_trace(rctx->rcl->device, &v, fresh_frame, rctx->cl_ray_dir_buffer, rctx->width, rctx->height, cam,
                    sample_num);
    free(cam);
    rt_device_sync(rctx->rcl->device);
}

/* reference: src/raytracer.c:301-340 */
void _raytracer_cast_rays(raytracer_context *rctx)
{
    scene_resource_push(rctx);
    rt_host_cast_rays(rctx);
}

/* reference: src/raytracer.c:137-142 */
void raytracer_render(raytracer_context *rctx)
{
    _raytracer_gen_ray_buffer(rctx);
    _raytracer_cast_rays(rctx);
}

/* reference: src/raytracer.c:245-298 */
void _raytracer_path_trace(raytracer_context *rctx, unsigned int sample_num)
{
    rt_host_path_trace(rctx, rctx->cl_path_fresh_frame_buffer, sample_num);
}

/* declared by the reference (include/raytracer.h:97, "DEPRECATED") and never
   defined there; here: the running mean of the PATH integrator */
void _raytracer_average_buffers(raytracer_context *rctx, unsigned int sample_num)
{
    rt_k_f_buffer_average(rctx->rcl->device, rctx->cl_path_output_buffer, rctx->cl_path_fresh_frame_buffer,
                          rctx->width * rctx->height, sample_num);
    rt_device_sync(rctx->rcl->device);
}

/* reference: src/raytracer.c:209-240 */
void _raytracer_push_path(raytracer_context *rctx)
{
    const unsigned int n = rctx->width * rctx->height;
    rt_k_f_to_byte(rctx->rcl->device, rctx->cl_output_buffer, rctx->cl_path_output_buffer, n);
    rt_mem_download(rctx->rcl->device, rctx->output_buffer, rctx->cl_output_buffer, (size_t)n * 4);
}

/* reference: src/raytracer.c:145-180 (the averaging call is commented out there) */
void raytracer_refined_render(raytracer_context *rctx)
{
    rctx->current_sample++;
    if (rctx->current_sample > rctx->num_samples) {
        rctx->render_complete = true;
        return;
    }
    _raytracer_gen_ray_buffer(rctx);
    _raytracer_path_trace(rctx, rctx->current_sample);
    if (rctx->current_sample == 1) {
        rt_mem_copy(rctx->rcl->device, rctx->cl_path_output_buffer, rctx->cl_path_fresh_frame_buffer,
                    (size_t)rctx->width * rctx->height * sizeof(vec4));
        rt_device_sync(rctx->rcl->device);
    }
    _raytracer_push_path(rctx);
}

void raytracer_destroy(raytracer_context *rctx)
{
    if (!rctx) return;
    rt_device *dev = rctx->rcl ? rctx->rcl->device : NULL;
    if (dev) {
        rt_mem_free(dev, rctx->cl_ray_buffer);
        rt_mem_free(dev, rctx->cl_ray_dir_buffer);
        rt_mem_free(dev, rctx->cl_path_output_buffer);
        rt_mem_free(dev, rctx->cl_output_buffer);
        rt_mem_free(dev, rctx->cl_path_fresh_frame_buffer);
        scene *s = rctx->stat_scene;
        if (s) {
            rt_mem_free(dev, s->cl_sphere_buffer); rt_mem_free(dev, s->cl_plane_buffer);
            rt_mem_free(dev, s->cl_material_buffer); rt_mem_free(dev, s->cl_mesh_buffer);
            rt_mem_free(dev, s->cl_mesh_vert_buffer.buffer); rt_mem_free(dev, s->cl_mesh_nrml_buffer.buffer);
            rt_mem_free(dev, s->cl_mesh_index_buffer.buffer);
            s->cl_sphere_buffer = s->cl_plane_buffer = s->cl_material_buffer = s->cl_mesh_buffer = NULL;
            s->cl_mesh_vert_buffer.buffer = s->cl_mesh_nrml_buffer.buffer = s->cl_mesh_index_buffer.buffer = NULL;
            if (s->kdt) {
                rt_mem_free(dev, s->kdt->view.nodes);
                rt_mem_free(dev, s->kdt->view.leaf_tris);
                rt_mem_free(dev, s->kdt->view.tri_shade);
                kd_tree_free(s->kdt);
                s->kdt = NULL;
            }
        }
    }
    free(rctx->ray_buffer);
    free(rctx->program);
    free(rctx);
}
