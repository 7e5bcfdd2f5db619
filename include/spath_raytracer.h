/*
 * spath_raytracer.h -- split (segmented) kd-tree path tracer: one bounce of
 * every pixel's path per render call.  Drop-in for the reference's
 * include/spath_raytracer.h:1-37.
 */
#ifndef RT_SPATH_RAYTRACER_H
#define RT_SPATH_RAYTRACER_H

#include <stdbool.h>
#include "rt_hip.h"

struct _rt_ctx;

/* where the per-pixel random buffer and the per-launch random value come from */
enum {
    RT_RNG_LIBC = 0,      /* srand(clock) + rand(), as the reference (src/spath_raytracer.c:118-132,283,321,409) */
    RT_RNG_SPLITMIX = 1   /* counter-based: random_buffer[i] = splitmix32(rng_seed, global pixel index),
                             value of launch k = splitmix32(rng_seed + 1, k); reproducible */
};

#define RT_TRIAL_PARTS 4   /* launches of a schedule trial: default, two alternatives, default */

typedef struct spath_raytracer_context {
    struct _rt_ctx *rctx;
    bool up_to_date;

    unsigned int num_iterations;      /* samples per pixel * 4 bounces */
    unsigned int current_iteration;
    bool render_complete;

    unsigned int start_time;

    unsigned int *random_buffer;

    rt_mem cl_path_output_buffer;                    /* float4 accumulation */
    rt_mem cl_path_ray_origin_buffer;                /* saved primary ray origins */
    rt_mem cl_path_collision_result_buffer;          /* hits of the current rays */
    rt_mem cl_spath_progress_buffer;                 /* {mask, sample_num} */
    rt_mem cl_path_origin_collision_result_buffer;   /* saved primary hits */
    rt_mem cl_random_buffer;
    rt_mem cl_bad_api_design_buffer;                 /* unused: axis masks / work counter of the reference */

    /* not in the reference */
    rt_mem cl_path_ray_origin_dir_buffer;            /* saved primary ray directions */
    rt_mem cl_spath_progress_b_buffer;               /* {accum, bounce_num} */
    int          rng_mode;            /* RT_RNG_* */
    unsigned int rng_seed;
    unsigned int rng_launch;          /* launches drawn so far */
    unsigned int resolve_interval;    /* resolve + readback every this many iterations (reference: 50) */
    int          separate_kernels;    /* 1: three launches per iteration, as the reference; 0: iterations are batched
                                         into multi-iteration launches (rt_k_spath_run) */
    bool         quiet;
    unsigned int max_batch;           /* most iterations folded into one launch (0 = 64); a batch always ends at a resolve point */
    unsigned int pending;             /* iterations accepted by spath_raytracer_render but not launched yet */
    unsigned int *pending_values;     /* their random values */
    rt_mem       cl_random_value_buffer;
    rt_mem       cl_pixel_cost_buffer;   /* walk steps per pixel in the last batch */
    rt_mem       cl_pixel_order_buffer;  /* pixels by decreasing cost: service order of the next batch */
    rt_mem       cl_cost_hist_buffer;
    rt_mem       cl_tile_scratch_buffer; /* per-tile cost and order (2 * pixels / 64 words) */
    bool         order_valid;
    int          no_reorder;             /* 1: always serve pixels in index order */
    int          schedule;               /* RT_SCHEDULE_* the batches run with */
    int          schedule_fixed;         /* 1: keep `schedule`; 0: the first batch of a render that is long enough is
                                            split in RT_TRIAL_PARTS launches (default schedule, the alternatives,
                                            default again), and an alternative is kept if it beats both default runs
                                            by 5 % */
    int          trial_state;            /* 0: not tried yet, 1: trial queued, 2: decided */
    rt_event     trial_marks[RT_TRIAL_PARTS + 1];
    unsigned int trial_iters[RT_TRIAL_PARTS];
} spath_raytracer_context;

spath_raytracer_context *init_spath_raytracer_context(struct _rt_ctx *rctx);   /* src/spath_raytracer.c:39-116 */
void spath_raytracer_render(spath_raytracer_context *ctx);                      /* src/spath_raytracer.c:355-402 */
void spath_raytracer_prepass(spath_raytracer_context *ctx);                     /* src/spath_raytracer.c:404-436 */

/* not in the reference */
void spath_raytracer_flush(spath_raytracer_context *ctx);     /* launch whatever render calls are still pending */
void spath_raytracer_resolve(spath_raytracer_context *ctx);   /* flush, resolve + blocking readback into output_buffer */
void spath_raytracer_fill_buffers(spath_raytracer_context *ctx, rt_spath_buffers *out);
void spath_raytracer_destroy(spath_raytracer_context *ctx);   /* the reference never releases its buffers */

#endif
