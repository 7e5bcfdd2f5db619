/*
 * host_path.c -- brute-force progressive path tracer driver.
 * reference: src/path_raytracer.c:3-188.
 */
#include <stdio.h>
#include <stdlib.h>

#include "path_raytracer.h"
#include "raytracer.h"

void rt_host_path_trace(raytracer_context *rctx, rt_mem fresh_frame, unsigned int sample_num);

path_raytracer_context *init_path_raytracer_context(struct _rt_ctx *rctx)
{
    path_raytracer_context *c = (path_raytracer_context *)calloc(1, sizeof(path_raytracer_context));
    c->rctx = rctx;
    c->up_to_date = false;
    c->num_samples = 128;            /* reference: src/path_raytracer.c:8 */
    const size_t bytes = (size_t)rctx->width * rctx->height * sizeof(vec4);
    c->cl_path_fresh_frame_buffer = rt_mem_alloc(rctx->rcl->device, bytes);
    c->cl_path_output_buffer = rt_mem_alloc(rctx->rcl->device, bytes);
    return c;
}

void path_raytracer_prepass(path_raytracer_context *c)
{
    raytracer_prepass(c->rctx);
    c->current_sample = 0;
    c->render_complete = false;
    c->start_time = (int)rt_time_ms();
}

/* One call = one frame of 4 samples x 4 bounces folded into the running mean,
   converted and read back (reference: src/path_raytracer.c:147-181). */
void path_raytracer_render(path_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    rt_device *dev = rctx->rcl->device;
    const unsigned int n = rctx->width * rctx->height;

    c->current_sample++;
    if (c->current_sample > c->num_samples) {
        if (!c->render_complete) printf("Render took %d ms\n", (int)rt_time_ms() - c->start_time);
        c->render_complete = true;
        return;
    }
    _raytracer_gen_ray_buffer(rctx);
    rt_host_path_trace(rctx, c->cl_path_fresh_frame_buffer, c->current_sample);
    if (c->current_sample == 1)
        rt_mem_copy(dev, c->cl_path_output_buffer, c->cl_path_fresh_frame_buffer, (size_t)n * sizeof(vec4));
    rt_k_f_buffer_average(dev, c->cl_path_output_buffer, c->cl_path_fresh_frame_buffer, n, c->current_sample);
    rt_k_f_to_byte(dev, rctx->cl_output_buffer, c->cl_path_output_buffer, n);
    rt_mem_download(dev, rctx->output_buffer, rctx->cl_output_buffer, (size_t)n * 4);
}

void path_raytracer_destroy(path_raytracer_context *c)
{
    if (!c) return;
    rt_device *dev = c->rctx->rcl->device;
    rt_mem_free(dev, c->cl_path_fresh_frame_buffer);
    rt_mem_free(dev, c->cl_path_output_buffer);
    free(c);
}
