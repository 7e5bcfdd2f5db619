/*
 * raytracer.h -- core context of the renderer.
 *
 * Drop-in for the reference's include/raytracer.h:1-99: same context members
 * and the same ten entry points; device handles are rt_mem instead of cl_mem.
 *
 * Call protocol (reference: src/startup.c:79-218): raytracer_init -> assign
 * stat_scene -> raytracer_build (once) -> init_X_context -> X_prepass ->
 * X_render until render_complete.  The caller owns output_buffer (width*height
 * uint32, 0xAARRGGBB as packed by the resolve kernels) and may read it as soon
 * as a render call returns.
 */
#ifndef RT_RAYTRACER_H
#define RT_RAYTRACER_H

#include <stdbool.h>
#include <stdint.h>
#include "parallel.h"
#include "scene.h"

#define SS_RAYTRACER 0
#define PATH_RAYTRACER 1
#define SPLIT_PATH_RAYTRACER 2

typedef struct _rt_ctx raytracer_context;
struct ic_context;

/* reference: include/raytracer.h:53-86 */
struct _rt_ctx {
    unsigned int width, height;

    float    *ray_buffer;
    uint32_t *output_buffer;

    scene *stat_scene;
    struct ic_context *ic_ctx;       /* irradiance cache: not part of this library, always NULL */

    unsigned int block_size_y;
    unsigned int block_size_x;

    unsigned int event_stack[32];
    unsigned int event_position;

    bool render_complete;

    rcl_ctx     *rcl;
    rcl_program *program;

    rt_mem cl_ray_buffer;            /* ray origins, float4 per pixel */
    rt_mem cl_output_buffer;         /* packed ARGB, uint32 per pixel */

    vec4        *path_output_buffer;
    unsigned int num_samples;
    unsigned int current_sample;
    rt_mem cl_path_output_buffer;
    rt_mem cl_path_fresh_frame_buffer;

    /* not in the reference */
    rt_mem cl_ray_dir_buffer;        /* ray directions, float4 per pixel */
    unsigned int tile_y0;            /* this context renders rows [tile_y0, tile_y0+height) ... */
    unsigned int full_height;        /* ... of a frame full_height rows high (0 = height) */
    unsigned int tile_band;          /* != 0: the rows come in bands of tile_band rows, tile_stride frame rows apart */
    unsigned int tile_stride;
};

raytracer_context *raytracer_init(unsigned int width, unsigned int height,
                                  uint32_t *output_buffer, rcl_ctx *ctx);
void raytracer_build(raytracer_context *rctx);
void raytracer_prepass(raytracer_context *rctx);
void raytracer_render(raytracer_context *rctx);
void raytracer_refined_render(raytracer_context *rctx);
void _raytracer_gen_ray_buffer(raytracer_context *rctx);
void _raytracer_path_trace(raytracer_context *rctx, unsigned int sample_num);
void _raytracer_average_buffers(raytracer_context *rctx, unsigned int sample_num);   /* deprecated in the reference */
void _raytracer_push_path(raytracer_context *rctx);
void _raytracer_cast_rays(raytracer_context *rctx);

/* not in the reference: multi-GPU row strips.  Call before raytracer_build. */
void raytracer_set_tile(raytracer_context *rctx, unsigned int y0, unsigned int full_height);
/* Interleaved partition: local row j is frame row first + (j / band) * stride + (j % band).  With `world` ranks,
   band rows each, rank r uses first = r * band, stride = world * band: every rank gets rows from all over the frame. */
void raytracer_set_tile_bands(raytracer_context *rctx, unsigned int first, unsigned int band, unsigned int stride,
                              unsigned int full_height);
/* The row mapping of this context as the kernels take it. */
void raytracer_get_rows(const raytracer_context *rctx, rt_rows *rows);
void raytracer_destroy(raytracer_context *rctx);

#endif
