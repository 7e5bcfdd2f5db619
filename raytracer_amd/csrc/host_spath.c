/*
 * host_spath.c -- split kd-tree path tracer driver: one bounce of every
 * pixel's path per render call.  reference: src/spath_raytracer.c:15-436.
 *
 * The reference synchronises the host after each of its three launches per
 * iteration.  Nothing outside this library can see the per-pixel buffers
 * between two resolves, so render calls are only recorded (iteration counter,
 * the call's random value) and launched in batches: one rt_k_spath_run launch
 * covers every pending iteration, ending at the latest at the next resolve
 * point, where the frame is resolved and read back exactly when the reference
 * would have done it.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kdtree.h"
#include "raytracer.h"
#include "spath_raytracer.h"

static uint32_t splitmix32(uint32_t seed, uint32_t index)
{
    uint32_t z = index + seed * 0x9E3779B9u + 0x9E3779B9u;
    z = (z ^ (z >> 16)) * 0x85EBCA6Bu;
    z = (z ^ (z >> 13)) * 0xC2B2AE35u;
    return z ^ (z >> 16);
}

/* the per-launch random_value (rand() in the reference, src/spath_raytracer.c:283,321) */
static uint32_t next_random_value(spath_raytracer_context *c)
{
    if (c->rng_mode == RT_RNG_SPLITMIX) return splitmix32(c->rng_seed + 1u, c->rng_launch++);
    c->rng_launch++;
    return (uint32_t)rand();
}

void spath_raytracer_fill_buffers(spath_raytracer_context *c, rt_spath_buffers *b)
{
    b->ray_o = c->rctx->cl_ray_buffer;
    b->ray_d = c->rctx->cl_ray_dir_buffer;
    b->origin_o = c->cl_path_ray_origin_buffer;
    b->origin_d = c->cl_path_ray_origin_dir_buffer;
    b->hits = c->cl_path_collision_result_buffer;
    b->source_hits = c->cl_path_origin_collision_result_buffer;
    b->prog_a = c->cl_spath_progress_buffer;
    b->prog_b = c->cl_spath_progress_b_buffer;
    b->random = c->cl_random_buffer;
    b->out_tex = c->cl_path_output_buffer;
}

/* reference: src/spath_raytracer.c:39-116 */
spath_raytracer_context *init_spath_raytracer_context(struct _rt_ctx *rctx)
{
    spath_raytracer_context *c = (spath_raytracer_context *)calloc(1, sizeof(spath_raytracer_context));
    rt_device *dev = rctx->rcl->device;
    const size_t n = (size_t)rctx->width * rctx->height;
    c->rctx = rctx;
    c->up_to_date = false;
    c->cl_path_output_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    c->cl_path_ray_origin_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    c->cl_path_ray_origin_dir_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    c->cl_path_collision_result_buffer = rt_mem_alloc(dev, n * sizeof(kd_tree_collision_result));
    c->cl_path_origin_collision_result_buffer = rt_mem_alloc(dev, n * sizeof(kd_tree_collision_result));
    c->cl_random_buffer = rt_mem_alloc(dev, n * sizeof(unsigned int));
    c->random_buffer = (unsigned int *)malloc(n * sizeof(unsigned int));
    c->cl_spath_progress_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    c->cl_spath_progress_b_buffer = rt_mem_alloc(dev, n * sizeof(vec4));
    rt_mem_zero(dev, c->cl_spath_progress_buffer, n * sizeof(vec4));
    rt_mem_zero(dev, c->cl_spath_progress_b_buffer, n * sizeof(vec4));
    c->cl_bad_api_design_buffer = NULL;
    c->rng_mode = RT_RNG_LIBC;
    c->rng_seed = 1;
    c->resolve_interval = 50;        /* reference: src/spath_raytracer.c:389-390 */
    c->separate_kernels = 0;
    c->quiet = false;
    c->max_batch = 64;
    c->pending = 0;
    c->pending_values = (unsigned int *)malloc(sizeof(unsigned int) * 1024);
    c->cl_random_value_buffer = rt_mem_alloc(dev, sizeof(unsigned int) * 1024);
    c->cl_pixel_cost_buffer = rt_mem_alloc(dev, ((size_t)n + 65536) * sizeof(unsigned int));   /* + room for a diagnostic build's per-wave time stamps */
    c->cl_pixel_order_buffer = rt_mem_alloc(dev, n * sizeof(unsigned int));
    c->cl_cost_hist_buffer = rt_mem_alloc(dev, 256 * sizeof(unsigned int));
    /* whole 8x8 screen tiles are ordered when the frame divides into them */
    c->cl_tile_scratch_buffer = (rctx->width % 8 == 0 && rctx->height % 8 == 0)
                                    ? rt_mem_alloc(dev, 2 * (size_t)(n / 64) * sizeof(unsigned int)) : NULL;
    c->order_valid = false;
    c->no_reorder = 0;
    return c;
}

/* reference: src/spath_raytracer.c:118-132 */
static void upload_random_buffer(spath_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    const size_t n = (size_t)rctx->width * rctx->height;
    if (c->rng_mode == RT_RNG_SPLITMIX) {
        /* the word of a pixel depends on its position in the full frame only */
        rt_rows rows;
        raytracer_get_rows(rctx, &rows);
        for (unsigned int j = 0; j < rctx->height; j++) {
            const size_t row = rows.band ? rows.first + (size_t)(j / rows.band) * rows.stride + j % rows.band
                                         : (size_t)rows.first + j;
            for (unsigned int x = 0; x < rctx->width; x++)
                c->random_buffer[(size_t)j * rctx->width + x] = splitmix32(c->rng_seed, (uint32_t)(row * rctx->width + x));
        }
    } else {
        for (size_t i = 0; i < n; i++) c->random_buffer[i] = (unsigned int)rand();
    }
    rt_mem_upload(rctx->rcl->device, c->cl_random_buffer, c->random_buffer, n * sizeof(unsigned int));
}

/* The parts of a schedule trial: the default schedule first and last, the alternatives between. */
static const int trial_schedule[RT_TRIAL_PARTS] = { RT_SCHEDULE_ONE_LEAF_SLOT, RT_SCHEDULE_TWO_LEAF_SLOTS,
                                                    RT_SCHEDULE_SHORT_CHAINS, RT_SCHEDULE_ONE_LEAF_SLOT };
static const char *schedule_name[RT_SCHEDULE_COUNT] = { "one leaf slot", "two leaf slots", "short chains" };

/* Reads the timings of a queued schedule trial (waits for the launches) and keeps the fastest schedule.  An
   alternative has to beat the better of the two default runs by 5 % to displace the default, so a slow first
   launch (cold caches, clocks still ramping) cannot decide it. */
static void finish_schedule_trial(spath_raytracer_context *c)
{
    if (c->trial_state != 1) return;
    rt_device *dev = c->rctx->rcl->device;
    float t[RT_TRIAL_PARTS];
    for (int k = 0; k < RT_TRIAL_PARTS; k++)
        t[k] = rt_event_elapsed_ms(dev, c->trial_marks[k], c->trial_marks[k + 1]) / (float)c->trial_iters[k];
    float best = t[0] < t[RT_TRIAL_PARTS - 1] ? t[0] : t[RT_TRIAL_PARTS - 1];
    best *= 0.95f;
    c->schedule = RT_SCHEDULE_ONE_LEAF_SLOT;
    for (int k = 1; k < RT_TRIAL_PARTS - 1; k++)
        if (t[k] < best) { best = t[k]; c->schedule = trial_schedule[k]; }
    for (int k = 0; k <= RT_TRIAL_PARTS; k++) { rt_event_release(dev, c->trial_marks[k]); c->trial_marks[k] = NULL; }
    c->trial_state = 2;
    if (!c->quiet)
        printf("schedule trial: %.3f / %.3f / %.3f / %.3f ms per iteration -> %s\n", t[0], t[1], t[2], t[3],
               schedule_name[c->schedule]);
}

/* Launches the iterations recorded since the last flush as one batch. */
void spath_raytracer_flush(spath_raytracer_context *c)
{
    finish_schedule_trial(c);        /* also with nothing pending: a caller that wants the decision flushes */
    if (!c->pending) return;
    raytracer_context *rctx = c->rctx;
    rt_device *dev = rctx->rcl->device;
    const unsigned int n = rctx->width * rctx->height;
    rt_spath_buffers b;
    spath_raytracer_fill_buffers(c, &b);
    /* blocking upload: also orders this batch behind the previous one, so the
       table is not overwritten while a launch still reads it */
    rt_mem_upload(dev, c->cl_random_value_buffer, c->pending_values, sizeof(unsigned int) * c->pending);
    if (!c->order_valid && !c->no_reorder && n >= 65536) {
        /* nothing is known about costs yet: all pixels in one bucket, i.e. plain screen-tile order */
        rt_mem_zero(dev, c->cl_pixel_cost_buffer, sizeof(unsigned int) * (size_t)n);
        rt_k_pixel_order(dev, c->cl_pixel_cost_buffer, c->cl_pixel_order_buffer, c->cl_cost_hist_buffer, n, 0,
                         rctx->width, c->cl_tile_scratch_buffer);
        c->order_valid = true;
    }
    finish_schedule_trial(c);
    const rt_mem order = (c->order_valid && !c->no_reorder) ? c->cl_pixel_order_buffer : NULL;
    unsigned int cost_iters = c->pending;      /* iterations the cost buffer will describe: those of the last launch */
    if (!c->schedule_fixed && c->trial_state == 0 && c->pending >= 4 * RT_TRIAL_PARTS && n >= 65536) {
        /* the parts of this batch run back to back in the same pixel order, one schedule each */
        unsigned int done = 0;
        c->trial_marks[0] = rt_event_record(dev);
        for (int k = 0; k < RT_TRIAL_PARTS; k++) {
            const unsigned int part = (c->pending - done) / (unsigned int)(RT_TRIAL_PARTS - k);
            c->trial_iters[k] = part;
            rt_k_spath_run(dev, &rctx->stat_scene->kdt->view, &b, n, part,
                           (char *)c->cl_random_value_buffer + sizeof(unsigned int) * done,
                           c->cl_pixel_cost_buffer, order, trial_schedule[k]);
            c->trial_marks[k + 1] = rt_event_record(dev);
            done += part;
            cost_iters = part;
        }
        c->trial_state = 1;
    } else {
        rt_k_spath_run(dev, &rctx->stat_scene->kdt->view, &b, n, c->pending, c->cl_random_value_buffer,
                       c->cl_pixel_cost_buffer, order, c->schedule);
    }
    if (!c->no_reorder && n >= 65536) {
        /* the next batch serves the pixels that were heaviest in this one first; bucket width ~ 1/16 of a
           typical pixel's cost (about 32 steps per iteration): finer buckets order better but scatter the
           pixels of a screen tile over more chunks, and the rays of a chunk then diverge more */
        unsigned int shift = 0;
        while ((32u * cost_iters) >> shift > 16u) shift++;
        rt_k_pixel_order(dev, c->cl_pixel_cost_buffer, c->cl_pixel_order_buffer, c->cl_cost_hist_buffer, n, shift,
                         rctx->width, c->cl_tile_scratch_buffer);
        c->order_valid = true;
    }
    c->pending = 0;
}

void spath_raytracer_resolve(spath_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    const unsigned int n = rctx->width * rctx->height;
    spath_raytracer_flush(c);
    rt_k_resolve_avg(rctx->rcl->device, rctx->cl_output_buffer, c->cl_path_output_buffer, n);
    rt_mem_download(rctx->rcl->device, rctx->output_buffer, rctx->cl_output_buffer, (size_t)n * 4);
}

/* reference: src/spath_raytracer.c:404-436 */
void spath_raytracer_prepass(spath_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    rt_device *dev = rctx->rcl->device;
    const unsigned int n = rctx->width * rctx->height;
    const size_t n16 = (size_t)n * sizeof(vec4);

    finish_schedule_trial(c);        /* a decision taken for this scene is kept across renders */
    c->render_complete = false;
    c->num_iterations = 2048 * 4;
    c->rng_launch = 0;
    c->pending = 0;
    c->order_valid = false;
    if (c->rng_mode == RT_RNG_LIBC) srand((unsigned int)rt_time_ms());
    c->start_time = rt_time_ms();

    rt_mem_zero(dev, c->cl_path_output_buffer, n16);
    raytracer_prepass(rctx);
    kd_tree *kdt = rctx->stat_scene->kdt;
    if (!kdt->root) {
        fprintf(stderr, "ERROR: the split path tracer needs mesh triangles (the scene has none)\n");
        exit(1);
    }

    c->current_iteration = 0;
    rt_mem_zero(dev, c->cl_spath_progress_buffer, n16);
    rt_mem_zero(dev, c->cl_spath_progress_b_buffer, n16);
    _raytracer_gen_ray_buffer(rctx);

    rt_spath_buffers b;
    spath_raytracer_fill_buffers(c, &b);
    rt_k_kd_intersect(dev, &kdt->view, b.ray_o, b.ray_d, b.hits, n);
    rt_rows rows;
    raytracer_get_rows(rctx, &rows);
    rt_k_spath_init(dev, &kdt->view, &b, rctx->width, rctx->height, &rows, next_random_value(c));
    upload_random_buffer(c);

    /* the reference clears the ray buffer here (src/spath_raytracer.c:432-433),
       discarding the first-bounce rays the init kernel has just written: the
       first iteration traces null rays and finishes one sky-only sample */
    rt_mem_zero(dev, rctx->cl_ray_buffer, n16);
    rt_mem_zero(dev, rctx->cl_ray_dir_buffer, n16);
    rt_device_sync(dev);
}

/* reference: src/spath_raytracer.c:355-402 */
void spath_raytracer_render(spath_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    rt_device *dev = rctx->rcl->device;
    kd_tree *kdt = rctx->stat_scene->kdt;
    const unsigned int n = rctx->width * rctx->height;
    rt_spath_buffers b;
    spath_raytracer_fill_buffers(c, &b);

    c->current_iteration++;
    if (c->current_iteration > c->num_iterations) {
        /* the reference still steps the RNG buffer on this call (src/spath_raytracer.c:364-373) */
        spath_raytracer_flush(c);
        rt_k_xorshift(dev, b.random, n);
        if (!c->render_complete) {
            spath_raytracer_resolve(c);
            if (!c->quiet) printf("Render took: %u ms\n", rt_time_ms() - c->start_time);
        }
        c->render_complete = true;
        return;
    }

    const uint32_t rv = next_random_value(c);
    if (c->separate_kernels) {
        rt_k_xorshift(dev, b.random, n);
        rt_k_kd_intersect(dev, &kdt->view, b.ray_o, b.ray_d, b.hits, n);
        rt_k_spath_trace(dev, &kdt->view, &b, n, rv, 0);
    } else {
        unsigned int cap = c->max_batch ? c->max_batch : 64;
        if (cap > 1024) cap = 1024;
        c->pending_values[c->pending++] = rv;
        if (c->pending >= cap || c->current_iteration == c->num_iterations) spath_raytracer_flush(c);
    }

    if (c->resolve_interval && c->current_iteration % c->resolve_interval == 0) {
        spath_raytracer_resolve(c);
        if (!c->quiet)
            printf("iteration %u  sample %u.%u/%u  elapsed %u ms\n", c->current_iteration, c->current_iteration / 4,
                   c->current_iteration % 4, c->num_iterations / 4, rt_time_ms() - c->start_time);
    }
}

void spath_raytracer_destroy(spath_raytracer_context *c)
{
    if (!c) return;
    rt_device *dev = c->rctx->rcl->device;
    finish_schedule_trial(c);
    rt_mem_free(dev, c->cl_path_output_buffer);
    rt_mem_free(dev, c->cl_path_ray_origin_buffer);
    rt_mem_free(dev, c->cl_path_ray_origin_dir_buffer);
    rt_mem_free(dev, c->cl_path_collision_result_buffer);
    rt_mem_free(dev, c->cl_path_origin_collision_result_buffer);
    rt_mem_free(dev, c->cl_random_buffer);
    rt_mem_free(dev, c->cl_spath_progress_buffer);
    rt_mem_free(dev, c->cl_spath_progress_b_buffer);
    rt_mem_free(dev, c->cl_random_value_buffer);
    rt_mem_free(dev, c->cl_pixel_cost_buffer);
    rt_mem_free(dev, c->cl_pixel_order_buffer);
    rt_mem_free(dev, c->cl_cost_hist_buffer);
    if (c->cl_tile_scratch_buffer) rt_mem_free(dev, c->cl_tile_scratch_buffer);
    free(c->pending_values);
    free(c->random_buffer);
    free(c);
}
