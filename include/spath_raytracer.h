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

#define RT_SPAN_MAX 64      /* render launches whose device time is pending between two resolve-point lines */
#define RT_TRIAL_PARTS 5   /* timed launches of a calibration: default schedule (two and two slots), the three others (one
                              and one, short chains, three and three), default again */

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
    unsigned int max_batch;           /* most iterations folded into one launch; 0 = eight resolve intervals (64 when
                                         resolve_interval is 0), at most RT_RUN_MAX_ITERS */
    unsigned int pending;             /* iterations accepted by spath_raytracer_render but not launched yet */
    unsigned int *pending_values;     /* random values of the launches being put together: two tables of RT_RUN_MAX_ITERS
                                         words in page-locked memory, used in turn, so that a launch can be queued while
                                         the one before it still runs */
    rt_mem       cl_random_value_buffer;   /* the same two tables on the device */
    rt_mem       cl_pixel_cost_buffer;   /* walk steps per pixel in the last batch */
    rt_mem       cl_pixel_order_buffer;  /* pixels by decreasing cost: service order of the next batch */
    rt_mem       cl_cost_hist_buffer;
    rt_mem       cl_tile_scratch_buffer; /* per-tile cost and order (2 * pixels / 64 words) */
    bool         order_valid;
    int          no_reorder;             /* 1: always serve pixels in index order */
    int          schedule;               /* RT_SCHEDULE_* the batches run with */
    int          schedule_fixed;         /* 1: keep `schedule`; 0: spath_raytracer_prepass ends with
                                            spath_raytracer_calibrate, which times the schedules on this scene and frame */
    int          trial_state;            /* 0: not calibrated yet, 2: decided (kept for later renders of this context) */
    float        trial_ms[RT_TRIAL_PARTS];   /* ms per iteration of the calibration's launches: default schedule, the
                                                three alternatives, default again */
    unsigned int calibration_iterations; /* iterations per calibration launch (0 = 12) */

    /* Launches that run ahead of the caller.  A pixel's iterations depend on nothing but the pixel, the iteration
       number and the random streams, and the caller sees the device's state only through output_buffer at the
       resolve points, so the render call that finds its iteration not launched yet launches it together with the
       following ones (up to max_batch, never past num_iterations) in ONE launch; the calls for those iterations then
       find their work done.  Where such a launch crosses a resolve point the kernel keeps a copy of each pixel's
       accumulator as it stood after that iteration, and the frame handed to the caller there is resolved from the
       copy: the same bits, at the same call, as with one launch per iteration.  Without it every launch ends in a
       stretch where the GPU only finishes its slowest pixels; longer launches have fewer of those. */
    int          run_ahead;              /* 1 (default): as above; 0: a launch holds only iterations the caller has asked for */
    unsigned int launched_upto;          /* iterations launched so far (>= current_iteration when running ahead) */
    rt_mem       cl_snapshot_buffer;     /* [slot][pixel] float4: out_tex at the resolve points of the last launch */
    unsigned int snapshot_slots;         /* slots allocated */
    unsigned int snap_first_iteration;   /* iteration whose accumulator slot 0 holds (0: the last launch took none) */
    unsigned int snap_count;
    unsigned int launches;               /* rt_k_spath_run launches since the prepass (diagnostic) */
    rt_event     values_uploaded[2];     /* marker behind the last upload from each of the two value tables */
    /* Device time of this context's own render launches since the last line printed at a resolve point (quiet == false
       only): marker pairs around each launch on the device's stream, summed when the line is printed.  The device-wide
       rt_timing_* table is left to whoever else uses the rt_device. */
    rt_event     span_marks[2 * RT_SPAN_MAX];
    unsigned int span_count;             /* pairs not summed yet */
    unsigned int span_launches;          /* launches summed into span_ms */
    double       span_ms;
} spath_raytracer_context;

spath_raytracer_context *init_spath_raytracer_context(struct _rt_ctx *rctx);   /* src/spath_raytracer.c:39-116 */
void spath_raytracer_render(spath_raytracer_context *ctx);                      /* src/spath_raytracer.c:355-402 */
void spath_raytracer_prepass(spath_raytracer_context *ctx);                     /* src/spath_raytracer.c:404-436 */
void spath_raytracer_ray_test(spath_raytracer_context *ctx);                    /* src/spath_raytracer.c:173-199 (debug view) */
void spath_raytracer_kd_test(spath_raytracer_context *ctx);                     /* src/spath_raytracer.c:201-235 (debug view) */

/* not in the reference */
void spath_raytracer_flush(spath_raytracer_context *ctx);     /* launch whatever render calls are still pending */
void spath_raytracer_resolve(spath_raytracer_context *ctx);   /* flush, then resolve the accumulator as the device holds it
                                                                  + blocking readback into output_buffer.  Fatal when
                                                                  launches have run ahead of current_iteration (the
                                                                  frame would hold iterations the caller has not asked
                                                                  for): between resolve points use run_ahead = 0, or end
                                                                  the render at the iteration wanted (num_iterations)
                                                                  before it is launched */
/* Also fatal, for the same reason: a spath_raytracer_render call that finds num_iterations lowered below launched_upto,
   or separate_kernels switched on while launched_upto is ahead of current_iteration.  Set both before the first render
   call after a prepass (or render with run_ahead = 0). */
/* Times the kernel's slot schedules (RT_SCHEDULE_*) on this scene and frame and keeps the fastest in ctx->schedule.
   Called by spath_raytracer_prepass when schedule_fixed is 0 and no decision exists yet; needs the state a prepass
   leaves.  Runs a few throw-away iterations on the per-pixel buffers and puts them back exactly (copies taken
   before), draws nothing from the random streams, and leaves the pixel order of the first real launch behind:
   rendered results do not depend on it.  Frames under 65,536 pixels and separate_kernels mode keep the default. */
void spath_raytracer_calibrate(spath_raytracer_context *ctx);

/* Checkpoint / resume (the reference keeps its progressive state on the device only and has neither).
   spath_raytracer_save launches what is pending, waits, and writes the render's state -- every per-pixel buffer,
   the iteration the device has reached (launches may have run ahead of current_iteration: the checkpoint is taken
   there) and the position in the random streams -- to `path`.  spath_raytracer_restore reads it into a context of
   the same frame size and row mapping on the same scene that has been through spath_raytracer_prepass, and sets
   current_iteration to the checkpoint's; rendering on from there gives, in RT_RNG_SPLITMIX mode, bit for bit what the
   uninterrupted render gives (RT_RNG_LIBC draws from rand(), whose state is not the library's to save).  Both
   return 0, or non-zero with a message on stderr and the context untouched when the file cannot be written / read or
   does not fit the context. */
int spath_raytracer_save(spath_raytracer_context *ctx, const char *path);
int spath_raytracer_restore(spath_raytracer_context *ctx, const char *path);
void spath_raytracer_fill_buffers(spath_raytracer_context *ctx, rt_spath_buffers *out);
void spath_raytracer_destroy(spath_raytracer_context *ctx);   /* the reference never releases its buffers */

#endif
