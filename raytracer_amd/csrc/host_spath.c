/*
 * host_spath.c -- split kd-tree path tracer driver: one bounce of every
 * pixel's path per render call.  reference: src/spath_raytracer.c:15-436.
 *
 * The reference synchronises the host after each of its three launches per
 * iteration.  Nothing outside this library can see the per-pixel buffers
 * between two resolves, so iterations are launched many at a time: one
 * rt_k_spath_run launch covers up to max_batch iterations, and (run_ahead) it
 * is issued by the first render call that needs one of them, for the calls
 * after it as well.  The frame is resolved and read back exactly when the
 * reference would have done it (every resolve_interval-th call), from the
 * copy of the accumulator the kernel kept at that iteration.
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

#define RT_SCHEDULE_DEFAULT RT_SCHEDULE_TWO_AND_TWO     /* see trial_schedule below */

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
    c->max_batch = 0;
    c->run_ahead = 1;
    c->pending = 0;
    c->pending_values = (unsigned int *)rt_host_alloc(dev, sizeof(unsigned int) * 2 * RT_RUN_MAX_ITERS);
    c->cl_random_value_buffer = rt_mem_alloc(dev, sizeof(unsigned int) * 2 * RT_RUN_MAX_ITERS);
    c->cl_pixel_cost_buffer = rt_mem_alloc(dev, ((size_t)n + 131072) * sizeof(unsigned int));   /* + room for a diagnostic build's per-wave time stamps */
    c->cl_pixel_order_buffer = rt_mem_alloc(dev, n * sizeof(unsigned int));
    c->cl_cost_hist_buffer = rt_mem_alloc(dev, 256 * sizeof(unsigned int));
    /* whole 8x8 screen tiles are ordered when the frame divides into them */
    c->cl_tile_scratch_buffer = (rctx->width % 8 == 0 && rctx->height % 8 == 0)
                                    ? rt_mem_alloc(dev, 2 * (size_t)(n / 64) * sizeof(unsigned int)) : NULL;
    c->order_valid = false;
    c->no_reorder = 0;
    c->schedule = RT_SCHEDULE_DEFAULT;       /* until a calibration decides otherwise (frames too small for one keep it) */
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

/* The timed launches of a calibration: the default schedule first and last, the alternatives between.  The default
   is two and two slots: over the workloads measured it is never behind one and one slot (2 to 10 % ahead on the mesh
   scenes, 3 % on a 20-iteration launch of the million-triangle scene), so an alternative has to earn its place
   against it. */
static const int trial_schedule[RT_TRIAL_PARTS] = { RT_SCHEDULE_DEFAULT, RT_SCHEDULE_ONE_LEAF_SLOT,
                                                    RT_SCHEDULE_SHORT_CHAINS, RT_SCHEDULE_THREE_AND_THREE,
                                                    RT_SCHEDULE_DEFAULT };
static const char *schedule_name[RT_SCHEDULE_COUNT] = { "one and one slot", "two and two slots", "short chains",
                                                        "three and three slots" };

#define RT_ORDER_MIN_PIXELS 65536u     /* smaller frames are served in index order with the default schedule */

static unsigned int launch_cap(const spath_raytracer_context *c)
{
    /* eight resolve intervals by default: every launch ends in a drain about one pixel's share of it long, so the
       longer the launch the smaller the drain's part (the million-triangle frame at 2048 x 2048, 1024 iterations
       sustained: 7.3 Grays/s in launches of 200, 7.55 in launches of 400, 7.64 in launches of 800); the frames of the
       resolve points inside a launch reach the caller together when it ends (0.2 s apart on that frame) */
    unsigned int cap = c->max_batch ? c->max_batch : (c->resolve_interval ? 8u * c->resolve_interval : 64u);
    if (c->resolve_interval && cap > RT_RUN_MAX_SNAPSHOTS * c->resolve_interval) cap = RT_RUN_MAX_SNAPSHOTS * c->resolve_interval;
    if (cap > RT_RUN_MAX_ITERS) cap = RT_RUN_MAX_ITERS;
    return cap ? cap : 1u;
}

/* The order in which the next launch serves the pixels, from the walk costs the last one (of cost_iters iterations)
   recorded: heaviest first.  Bucket width ~ 1/16 of a typical pixel's cost (about 32 steps per iteration): finer
   buckets order better but scatter the pixels of a screen tile over more chunks, and the rays of a chunk then
   diverge more.  cost_iters == 0: nothing is known yet, all pixels in one bucket, i.e. plain screen-tile order. */
static void update_pixel_order(spath_raytracer_context *c, unsigned int cost_iters)
{
    raytracer_context *rctx = c->rctx;
    rt_device *dev = rctx->rcl->device;
    const unsigned int n = rctx->width * rctx->height;
    unsigned int shift = 0;
    if (cost_iters == 0) rt_mem_zero(dev, c->cl_pixel_cost_buffer, sizeof(unsigned int) * (size_t)n);
    else while ((32u * cost_iters) >> shift > 16u) shift++;
    rt_k_pixel_order(dev, c->cl_pixel_cost_buffer, c->cl_pixel_order_buffer, c->cl_cost_hist_buffer, n, shift,
                     rctx->width, c->cl_tile_scratch_buffer);
    c->order_valid = true;
}

/* Device time of this context's render launches (for the line printed at the resolve points of a context that is
   not quiet): a marker pair around each launch, summed here.  fold waits for the last pair's launch to end. */
static void span_fold(spath_raytracer_context *c)
{
    rt_device *dev = c->rctx->rcl->device;
    for (unsigned int k = 0; k < c->span_count; k++) {
        c->span_ms += rt_event_elapsed_ms(dev, c->span_marks[2 * k], c->span_marks[2 * k + 1]);
        c->span_launches++;
        rt_event_release(dev, c->span_marks[2 * k]);
        rt_event_release(dev, c->span_marks[2 * k + 1]);
    }
    c->span_count = 0;
}

static void span_begin(spath_raytracer_context *c)
{
    if (c->quiet) return;
    if (c->span_count == RT_SPAN_MAX) span_fold(c);
    c->span_marks[2 * c->span_count] = rt_event_record(c->rctx->rcl->device);
}

static void span_end(spath_raytracer_context *c)
{
    if (c->quiet) return;
    c->span_marks[2 * c->span_count + 1] = rt_event_record(c->rctx->rcl->device);
    c->span_count++;
}

static void span_reset(spath_raytracer_context *c)
{
    span_fold(c);
    c->span_ms = 0.0;
    c->span_launches = 0;
}

/* One rt_k_spath_run launch over iterations launched_upto+1 .. launched_upto+count; their random values are drawn
   here, in iteration order. */
static void launch_iterations(spath_raytracer_context *c, unsigned int count)
{
    raytracer_context *rctx = c->rctx;
    rt_device *dev = rctx->rcl->device;
    const unsigned int n = rctx->width * rctx->height;
    const bool reorder = !c->no_reorder && n >= RT_ORDER_MIN_PIXELS;
    rt_spath_buffers b;
    spath_raytracer_fill_buffers(c, &b);
    /* the value tables are used in turn: this launch may be queued while the previous one still reads its table;
       the one before that has long finished when its table comes round again, and the marker makes sure */
    const unsigned int slot = c->launches & 1u;
    unsigned int *values = c->pending_values + (size_t)slot * RT_RUN_MAX_ITERS;
    const rt_mem d_values = (char *)c->cl_random_value_buffer + sizeof(unsigned int) * (size_t)slot * RT_RUN_MAX_ITERS;
    if (c->values_uploaded[slot]) {
        rt_event_wait(dev, c->values_uploaded[slot]);
        rt_event_release(dev, c->values_uploaded[slot]);
        c->values_uploaded[slot] = NULL;
    }
    for (unsigned int k = 0; k < count; k++) values[k] = next_random_value(c);
    rt_mem_upload_queued(dev, d_values, values, sizeof(unsigned int) * count);
    if (reorder && !c->order_valid) update_pixel_order(c, 0);

    /* resolve points inside this launch: iterations first, first + r, ... <= last */
    const unsigned int r = c->resolve_interval, a = c->launched_upto + 1, last = c->launched_upto + count;
    const unsigned int first = r ? ((a + r - 1) / r) * r : 0;
    c->snap_count = (r && first <= last) ? (last - first) / r + 1 : 0;
    c->snap_first_iteration = c->snap_count ? first : 0;
    if (c->snap_count > c->snapshot_slots) {
        if (c->cl_snapshot_buffer) rt_mem_free(dev, c->cl_snapshot_buffer);
        c->cl_snapshot_buffer = rt_mem_alloc(dev, (size_t)c->snap_count * n * sizeof(vec4));
        c->snapshot_slots = c->snap_count;
    }
    span_begin(c);
    rt_k_spath_run_snap(dev, &rctx->stat_scene->kdt->view, &b, n, count, d_values,
                        c->cl_pixel_cost_buffer, (reorder && c->order_valid) ? c->cl_pixel_order_buffer : NULL, c->schedule,
                        c->snap_count ? c->cl_snapshot_buffer : NULL, c->snap_count ? first - a + 1 : 0, r);
    span_end(c);
    c->values_uploaded[slot] = rt_event_record(dev);      /* the launch has read the table once this marker is passed */
    if (reorder) update_pixel_order(c, count);
    c->launched_upto = last;
    c->launches++;
}

/* Launches the iterations the caller has asked for and that are not launched yet (with run_ahead: up to a full
   launch from there on).  Never past num_iterations. */
void spath_raytracer_flush(spath_raytracer_context *c)
{
    const unsigned int asked = c->current_iteration < c->num_iterations ? c->current_iteration : c->num_iterations;
    while (!c->separate_kernels && c->launched_upto < asked) {
        const unsigned int count = (c->run_ahead ? c->num_iterations : asked) - c->launched_upto;
        const unsigned int cap = launch_cap(c);
        launch_iterations(c, count < cap ? count : cap);
    }
    c->pending = 0;
}

static void resolve_from(spath_raytracer_context *c, rt_mem accum)
{
    raytracer_context *rctx = c->rctx;
    const unsigned int n = rctx->width * rctx->height;
    rt_k_resolve_avg(rctx->rcl->device, rctx->cl_output_buffer, accum, n);
    rt_mem_download(rctx->rcl->device, rctx->output_buffer, rctx->cl_output_buffer, (size_t)n * 4);
}

/* Launches run ahead of the calls that ask for their iterations; what the device holds is the state after iteration
   launched_upto.  A caller that changes the plan while launches are ahead of it -- ends the render earlier than the
   iterations already launched, or switches to one launch per kernel per iteration in mid-flight -- would be handed
   frames with more iterations in them than it asked for: that is refused, loudly, like every other misuse. */
static void refuse_if_ahead(const spath_raytracer_context *c, const char *who, const char *what)
{
    fprintf(stderr, "ERROR: %s: %s while iterations up to %u are already launched (the caller is at %u of %u); change it "
                    "before the render starts, right after a prepass, or render with run_ahead = 0\n", who, what,
            c->launched_upto, c->current_iteration, c->num_iterations);
    exit(1);
}

/* The frame as the device holds it.  Between the library's resolve points that is only the caller's frame when no
   launch has run ahead of its calls (run_ahead = 0, one launch per kernel per iteration, or a render that has reached
   its end); otherwise the call is refused. */
void spath_raytracer_resolve(spath_raytracer_context *c)
{
    spath_raytracer_flush(c);
    const unsigned int at = c->current_iteration < c->num_iterations ? c->current_iteration : c->num_iterations;
    if (c->launched_upto > at) refuse_if_ahead(c, "spath_raytracer_resolve", "the frame was asked for between resolve points");
    resolve_from(c, c->cl_path_output_buffer);
}

/* The frame as it stands after iteration `iteration` (a resolve point the caller has just reached). */
static void resolve_at(spath_raytracer_context *c, unsigned int iteration)
{
    const size_t n = (size_t)c->rctx->width * c->rctx->height;
    spath_raytracer_flush(c);
    if (c->snap_count && iteration >= c->snap_first_iteration &&
        (iteration - c->snap_first_iteration) % c->resolve_interval == 0 &&
        (iteration - c->snap_first_iteration) / c->resolve_interval < c->snap_count) {
        const size_t slot = (iteration - c->snap_first_iteration) / c->resolve_interval;
        resolve_from(c, (char *)c->cl_snapshot_buffer + slot * n * sizeof(vec4));
    } else if (c->launched_upto <= iteration) {
        resolve_from(c, c->cl_path_output_buffer);       /* one launch per kernel per iteration: nothing has run ahead */
    } else {
        fprintf(stderr, "ERROR: no copy of the accumulator after iteration %u was kept (resolve_interval changed during a render?)\n", iteration);
        exit(1);
    }
}

void spath_raytracer_calibrate(spath_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    rt_device *dev = rctx->rcl->device;
    const unsigned int n = rctx->width * rctx->height;
    if (c->separate_kernels || n < RT_ORDER_MIN_PIXELS) return;
    unsigned int iters = c->calibration_iterations ? c->calibration_iterations : 12u;
    if (iters > RT_RUN_MAX_ITERS) iters = RT_RUN_MAX_ITERS;

    /* copies of everything a launch writes */
    rt_spath_buffers b;
    spath_raytracer_fill_buffers(c, &b);
    rt_mem live[7] = { b.ray_o, b.ray_d, b.hits, b.prog_a, b.prog_b, b.random, b.out_tex };
    const size_t bytes[7] = { 16, 16, 16, 16, 16, 4, 16 };
    rt_mem saved[7];
    for (int k = 0; k < 7; k++) {
        saved[k] = rt_mem_alloc(dev, bytes[k] * n);
        rt_mem_copy(dev, saved[k], live[k], bytes[k] * n);
    }
    /* random values of its own: the render's streams stay untouched */
    for (unsigned int k = 0; k < iters; k++) c->pending_values[k] = splitmix32(0xca11b8u, k);
    rt_mem_upload(dev, c->cl_random_value_buffer, c->pending_values, sizeof(unsigned int) * iters);

    const bool reorder = !c->no_reorder;
    const rt_mem order = reorder ? c->cl_pixel_order_buffer : NULL;
    rt_event marks[RT_TRIAL_PARTS][2];
    for (int k = -1; k < RT_TRIAL_PARTS; k++) {
        /* launch -1 (default schedule, screen-tile order) only measures the pixels' costs: the timed launches and
           the render's first launch serve the heaviest pixels first, as every later launch of the render will */
        if (k == -1 && reorder) update_pixel_order(c, 0);
        if (k >= 0) marks[k][0] = rt_event_record(dev);
        rt_k_spath_run(dev, &rctx->stat_scene->kdt->view, &b, n, iters, c->cl_random_value_buffer, c->cl_pixel_cost_buffer,
                       order, k < 0 ? RT_SCHEDULE_DEFAULT : trial_schedule[k]);
        if (k >= 0) marks[k][1] = rt_event_record(dev);
        if (k == -1 && reorder) update_pixel_order(c, iters);
        for (int j = 0; j < 7; j++) rt_mem_copy(dev, live[j], saved[j], bytes[j] * n);
    }
    for (int k = 0; k < RT_TRIAL_PARTS; k++) {
        c->trial_ms[k] = rt_event_elapsed_ms(dev, marks[k][0], marks[k][1]) / (float)iters;
        rt_event_release(dev, marks[k][0]);
        rt_event_release(dev, marks[k][1]);
    }
    rt_device_sync(dev);
    for (int k = 0; k < 7; k++) rt_mem_free(dev, saved[k]);

    /* An alternative has to beat the better of the two default runs by 5 % to displace the default, so a slow
       first launch (clocks still ramping) cannot decide it. */
    const float *t = c->trial_ms;
    float best = (t[0] < t[RT_TRIAL_PARTS - 1] ? t[0] : t[RT_TRIAL_PARTS - 1]) * 0.95f;
    c->schedule = RT_SCHEDULE_DEFAULT;
    for (int k = 1; k < RT_TRIAL_PARTS - 1; k++)
        if (t[k] < best) { best = t[k]; c->schedule = trial_schedule[k]; }
    c->trial_state = 2;
    if (!c->quiet)
        printf("schedule calibration: %.3f / %.3f / %.3f / %.3f / %.3f ms per iteration -> %s\n", t[0], t[1], t[2], t[3],
               t[4], schedule_name[c->schedule]);
}

/* The reference's two debug views (commented out of its render loop, src/spath_raytracer.c:400): the current rays'
   origins, and the normals at the current rays' hits, drawn into output_buffer.
   reference: src/spath_raytracer.c:173-199 and :201-235 */
void spath_raytracer_ray_test(spath_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    const unsigned int n = rctx->width * rctx->height;
    spath_raytracer_flush(c);
    rt_k_kd_ray_draw(rctx->rcl->device, rctx->cl_output_buffer, rctx->cl_ray_buffer, n);
    rt_mem_download(rctx->rcl->device, rctx->output_buffer, rctx->cl_output_buffer, (size_t)n * 4);
}

void spath_raytracer_kd_test(spath_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    const unsigned int n = rctx->width * rctx->height;
    spath_raytracer_flush(c);
    rt_k_kd_test_draw(rctx->rcl->device, &rctx->stat_scene->kdt->view, rctx->cl_output_buffer,
                      c->cl_path_collision_result_buffer, n);
    rt_mem_download(rctx->rcl->device, rctx->output_buffer, rctx->cl_output_buffer, (size_t)n * 4);
}

/* reference: src/spath_raytracer.c:404-436 */
void spath_raytracer_prepass(spath_raytracer_context *c)
{
    raytracer_context *rctx = c->rctx;
    rt_device *dev = rctx->rcl->device;
    const unsigned int n = rctx->width * rctx->height;
    const size_t n16 = (size_t)n * sizeof(vec4);

    c->render_complete = false;
    c->num_iterations = 2048 * 4;
    c->rng_launch = 0;
    c->pending = 0;
    c->launched_upto = 0;
    c->launches = 0;
    c->snap_count = 0;
    c->snap_first_iteration = 0;
    c->order_valid = false;
    if (c->rng_mode == RT_RNG_LIBC) srand((unsigned int)rt_time_ms());
    c->start_time = rt_time_ms();
    span_reset(c);

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
    /* a decision taken for this scene and frame is kept across renders */
    if (!c->schedule_fixed && c->trial_state == 0) spath_raytracer_calibrate(c);
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
    if (c->launched_upto > c->num_iterations) refuse_if_ahead(c, "spath_raytracer_render", "num_iterations was lowered");
    if (c->separate_kernels && c->launched_upto >= c->current_iteration)
        refuse_if_ahead(c, "spath_raytracer_render", "separate_kernels was switched on");
    if (c->current_iteration > c->num_iterations) {
        /* the reference still steps the RNG buffer on this call (src/spath_raytracer.c:364-373) */
        spath_raytracer_flush(c);
        rt_k_xorshift(dev, b.random, n);
        if (!c->render_complete) {
            resolve_from(c, c->cl_path_output_buffer);
            if (!c->quiet) printf("Render took: %u ms\n", rt_time_ms() - c->start_time);
        }
        c->render_complete = true;
        return;
    }

    if (c->separate_kernels) {
        const uint32_t rv = next_random_value(c);
        span_begin(c);
        rt_k_xorshift(dev, b.random, n);
        rt_k_kd_intersect(dev, &kdt->view, b.ray_o, b.ray_d, b.hits, n);
        rt_k_spath_trace(dev, &kdt->view, &b, n, rv, 0);
        span_end(c);
        c->launched_upto = c->current_iteration;
    } else if (c->current_iteration > c->launched_upto) {
        /* running ahead: this call launches its iteration and the following ones; otherwise iterations are
           collected until the launch is full, the render ends or the frame is due */
        c->pending = c->current_iteration - c->launched_upto;
        if (c->run_ahead || c->pending >= launch_cap(c) || c->current_iteration == c->num_iterations) spath_raytracer_flush(c);
    }

    if (c->resolve_interval && c->current_iteration % c->resolve_interval == 0) {
        const unsigned int t0 = rt_time_ms();
        unsigned int launches = 0;
        double run_ms = 0.0;
        if (!c->quiet) {
            /* device time of this context's render launches since the last line; the reference prints host times of
               its three launches here (src/spath_raytracer.c:396-399): traversal and shading are one kernel in this
               library (one marker pair per launch, or per iteration's three launches with separate_kernels) */
            span_fold(c);
            launches = c->span_launches;
            run_ms = c->span_ms;
            c->span_ms = 0.0;
            c->span_launches = 0;
        }
        const unsigned int t1 = rt_time_ms();
        resolve_at(c, c->current_iteration);
        if (!c->quiet) {
            const unsigned int t2 = rt_time_ms(), el = t2 - c->start_time;
            printf("collision+trace: %.1f (device, %u launches), wait: %u, draw: %u    %u.%u/%u    %u:%u:%u\n", run_ms,
                   launches, t1 - t0, t2 - t1, c->current_iteration / 4, c->current_iteration % 4,
                   c->num_iterations / 4, (el / 1000) / 60, (el / 1000) % 60, el % 1000);
        }
    }
}

/* ---- checkpoint / resume ------------------------------------------------------------------------------------ */
#define RT_CHECKPOINT_VERSION 2u
typedef struct {
    char     magic[8];                 /* "RTSPATH1" */
    uint32_t version, header_bytes;    /* RT_CHECKPOINT_VERSION, sizeof(spath_checkpoint_header): a file written by a build
                                          with another layout (or byte order: both fields would not match) is refused */
    uint32_t width, height;            /* local frame of the context */
    rt_rows  rows;                     /* which rows of the full frame they are */
    uint32_t num_tris, tree_bytes;     /* the scene the state belongs to */
    uint32_t iteration;                /* iterations the buffers have been through */
    uint32_t num_iterations;
    int32_t  rng_mode;
    uint32_t rng_seed, rng_launch;
    uint32_t reserved[4];
} spath_checkpoint_header;

#define RT_CHECKPOINT_BUFFERS 10
static void checkpoint_buffers(spath_raytracer_context *c, rt_mem mem[RT_CHECKPOINT_BUFFERS], size_t bytes[RT_CHECKPOINT_BUFFERS])
{
    rt_spath_buffers b;
    spath_raytracer_fill_buffers(c, &b);
    const rt_mem m[RT_CHECKPOINT_BUFFERS] = { b.ray_o, b.ray_d, b.origin_o, b.origin_d, b.hits, b.source_hits, b.prog_a,
                                               b.prog_b, b.random, b.out_tex };
    for (int k = 0; k < RT_CHECKPOINT_BUFFERS; k++) { mem[k] = m[k]; bytes[k] = 16; }
    bytes[8] = 4;
}

static void checkpoint_header(spath_raytracer_context *c, spath_checkpoint_header *hd)
{
    raytracer_context *rctx = c->rctx;
    memset(hd, 0, sizeof *hd);
    memcpy(hd->magic, "RTSPATH1", 8);
    hd->version = RT_CHECKPOINT_VERSION;
    hd->header_bytes = (uint32_t)sizeof *hd;
    hd->width = rctx->width;
    hd->height = rctx->height;
    raytracer_get_rows(rctx, &hd->rows);
    hd->num_tris = rctx->stat_scene->kdt->view.num_tris;
    hd->tree_bytes = rctx->stat_scene->kdt->buffer_size;
    hd->iteration = c->launched_upto;
    hd->num_iterations = c->num_iterations;
    hd->rng_mode = c->rng_mode;
    hd->rng_seed = c->rng_seed;
    hd->rng_launch = c->rng_launch;
}

int spath_raytracer_save(spath_raytracer_context *c, const char *path)
{
    rt_device *dev = c->rctx->rcl->device;
    const size_t n = (size_t)c->rctx->width * c->rctx->height;
    spath_raytracer_flush(c);
    rt_device_sync(dev);
    FILE *f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "spath_raytracer_save: cannot write %s\n", path); return 1; }
    spath_checkpoint_header hd;
    checkpoint_header(c, &hd);
    rt_mem mem[RT_CHECKPOINT_BUFFERS];
    size_t bytes[RT_CHECKPOINT_BUFFERS];
    checkpoint_buffers(c, mem, bytes);
    void *host = malloc(16 * (n ? n : 1));
    if (!host) { fprintf(stderr, "spath_raytracer_save: out of host memory\n"); fclose(f); remove(path); return 3; }
    int bad = fwrite(&hd, sizeof hd, 1, f) != 1;
    for (int k = 0; k < RT_CHECKPOINT_BUFFERS && !bad; k++) {
        rt_mem_download(dev, host, mem[k], bytes[k] * n);
        bad = n && fwrite(host, bytes[k], n, f) != n;
    }
    free(host);
    if (fclose(f) != 0) bad = 1;
    if (bad) { fprintf(stderr, "spath_raytracer_save: writing %s failed\n", path); remove(path); return 2; }
    return 0;
}

int spath_raytracer_restore(spath_raytracer_context *c, const char *path)
{
    rt_device *dev = c->rctx->rcl->device;
    const size_t n = (size_t)c->rctx->width * c->rctx->height;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "spath_raytracer_restore: cannot read %s\n", path); return 1; }
    spath_checkpoint_header hd, mine;
    checkpoint_header(c, &mine);
    rt_mem mem[RT_CHECKPOINT_BUFFERS];
    size_t bytes[RT_CHECKPOINT_BUFFERS], total = sizeof hd;
    checkpoint_buffers(c, mem, bytes);
    for (int k = 0; k < RT_CHECKPOINT_BUFFERS; k++) total += bytes[k] * n;
    int bad = fread(&hd, sizeof hd, 1, f) != 1 || memcmp(hd.magic, mine.magic, 8) != 0 ||
              hd.version != RT_CHECKPOINT_VERSION || hd.header_bytes != (uint32_t)sizeof hd;
    if (!bad)       /* a damaged header: an iteration past the render's end, or an RNG mode that does not exist */
        bad = hd.iteration > hd.num_iterations || (hd.rng_mode != RT_RNG_LIBC && hd.rng_mode != RT_RNG_SPLITMIX);
    if (!bad)
        bad = hd.width != mine.width || hd.height != mine.height || memcmp(&hd.rows, &mine.rows, sizeof hd.rows) != 0 ||
              hd.num_tris != mine.num_tris || hd.tree_bytes != mine.tree_bytes;
    if (!bad) { fseek(f, 0, SEEK_END); bad = (size_t)ftell(f) != total; fseek(f, (long)sizeof hd, SEEK_SET); }
    if (bad) {
        fprintf(stderr, "spath_raytracer_restore: %s is not a checkpoint of this frame and scene\n", path);
        fclose(f);
        return 2;
    }
    /* everything is read before anything is changed */
    void *host[RT_CHECKPOINT_BUFFERS] = { 0 };
    for (int k = 0; k < RT_CHECKPOINT_BUFFERS && !bad; k++) {
        host[k] = malloc(bytes[k] * (n ? n : 1));
        bad = !host[k] || (n && fread(host[k], bytes[k], n, f) != n);
    }
    fclose(f);
    if (!bad) {
        spath_raytracer_flush(c);
        rt_device_sync(dev);
        for (int k = 0; k < RT_CHECKPOINT_BUFFERS; k++) rt_mem_upload(dev, mem[k], host[k], bytes[k] * n);
        c->current_iteration = c->launched_upto = hd.iteration;
        c->num_iterations = hd.num_iterations;
        c->rng_mode = hd.rng_mode;
        c->rng_seed = hd.rng_seed;
        c->rng_launch = hd.rng_launch;
        c->render_complete = false;
        c->pending = 0;
        c->snap_count = 0;
        c->snap_first_iteration = 0;
        c->order_valid = false;             /* costs of the restored pixels are not known to this context */
    } else {
        fprintf(stderr, "spath_raytracer_restore: reading %s failed\n", path);
    }
    for (int k = 0; k < RT_CHECKPOINT_BUFFERS; k++) free(host[k]);
    return bad ? 3 : 0;
}

void spath_raytracer_destroy(spath_raytracer_context *c)
{
    if (!c) return;
    rt_device *dev = c->rctx->rcl->device;
    if (c->cl_snapshot_buffer) rt_mem_free(dev, c->cl_snapshot_buffer);
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
    for (int k = 0; k < 2; k++)
        if (c->values_uploaded[k]) rt_event_release(dev, c->values_uploaded[k]);
    span_fold(c);
    rt_host_free(dev, c->pending_values);
    free(c->random_buffer);
    free(c);
}
