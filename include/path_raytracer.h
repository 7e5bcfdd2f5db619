/*
 * path_raytracer.h -- brute-force progressive path tracer (4 samples x 4
 * bounces per call, running mean).  Drop-in for the reference's
 * include/path_raytracer.h:1-24.
 */
#ifndef RT_PATH_RAYTRACER_H
#define RT_PATH_RAYTRACER_H

#include <stdbool.h>
#include "rt_hip.h"

struct _rt_ctx;

typedef struct path_raytracer_context {
    struct _rt_ctx *rctx;
    bool up_to_date;

    unsigned int num_samples;
    unsigned int current_sample;
    bool render_complete;
    int  start_time;

    rt_mem cl_path_output_buffer;
    rt_mem cl_path_fresh_frame_buffer;
} path_raytracer_context;

path_raytracer_context *init_path_raytracer_context(struct _rt_ctx *rctx);   /* src/path_raytracer.c:3-23 */
void path_raytracer_render(path_raytracer_context *ctx);                      /* src/path_raytracer.c:147-181 */
void path_raytracer_prepass(path_raytracer_context *ctx);                     /* src/path_raytracer.c:183-188 */
void path_raytracer_destroy(path_raytracer_context *ctx);                     /* not in the reference */

#endif
