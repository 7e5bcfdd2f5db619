/*
 * plumbing.c -- the context and buffer helpers of parallel.h on top of the HIP
 * C-ABI layer (rt_hip.h).  Takes the place of the reference's OpenCL plumbing
 * (reference: src/parallel.c); there is no OpenCL and no fallback device.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "parallel.h"
#include "raytracer.h"

/* kernel names the reference requests from load_program_raw
   (reference: include/raytracer.h:15-25) that this library implements */
static const char *const k_known_kernels[] = {
    "cast_ray_test", "generate_rays", "path_trace", "f_buffer_average", "f_buffer_to_byte_buffer",
    "kdtree_intersection", "segmented_path_trace", "f_buffer_to_byte_buffer_avg",
    "segmented_path_trace_init", "xorshift_batch"};

unsigned int rt_time_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned int)((unsigned long long)ts.tv_sec * 1000ull + (unsigned long long)ts.tv_nsec / 1000000ull);
}

/* reference: src/parallel.c:4-83 */
void cl_info(void)
{
    int n = rt_device_count();
    printf("HIP devices: %d\n", n);
    for (int i = 0; i < n; i++) {
        rt_device *d = rt_device_open(i);
        rt_device_props p;
        rt_device_get_props(d, &p);
        printf("%d. Device: %s\n", i, p.name);
        printf(" %d.1 Parallel compute units: %d\n", i, p.compute_units);
        printf(" %d.2 Wavefront size: %d\n", i, p.wave_size);
        printf(" %d.3 Max waves per compute unit: %d\n", i, p.max_waves_per_cu);
        printf(" %d.4 Global memory: %zu MiB\n", i, p.global_mem_bytes >> 20);
        printf(" %d.5 LDS per compute unit: %zu KiB\n", i, p.lds_bytes_per_cu >> 10);
        rt_device_close(d);
    }
    printf("\n");
}

static int env_int(const char *name, int fallback)
{
    const char *v = getenv(name);
    if (!v || !*v) return fallback;
    return atoi(v);
}

/* reference: src/parallel.c:93-279.  The reference hands this function an
   uninitialised struct (src/startup.c:89-90), so the device is chosen from the
   environment: RT_DEVICE_ORDINAL, else LOCAL_RANK (one process per GPU), else 0. */
void create_context(rcl_ctx *ctx)
{
    int n = rt_device_count();
    if (n <= 0) {
        fprintf(stderr, "Error: no HIP GPU device is visible (there is no CPU fallback).\n");
        exit(1);
    }
    int ordinal = env_int("RT_DEVICE_ORDINAL", env_int("LOCAL_RANK", 0));
    if (ordinal < 0 || ordinal >= n) {
        fprintf(stderr, "Error: device %d asked for (RT_DEVICE_ORDINAL / LOCAL_RANK) but %d HIP device(s) are visible.\n", ordinal, n);
        exit(1);
    }
    memset(ctx, 0, sizeof *ctx);
    ctx->device_ordinal = ordinal;
    ctx->device = rt_device_open(ordinal);
    rt_device_props p;
    rt_device_get_props(ctx->device, &p);
    ctx->simt_size = (unsigned)p.wave_size;
    ctx->num_simt_per_multiprocessor = 4;
    ctx->num_multiprocessors = (unsigned)p.compute_units;
    ctx->num_cores = ctx->num_multiprocessors * ctx->num_simt_per_multiprocessor * ctx->simt_size;
}

/* reference: src/parallel.c:444-528.  Nothing is compiled at run time. */
void load_program_raw(rcl_ctx *ctx, char *data, char **kernels, unsigned int num_kernels,
                      rcl_program *program, char **macros, unsigned int num_macros)
{
    (void)ctx; (void)data; (void)macros; (void)num_macros;
    for (unsigned int i = 0; i < num_kernels; i++) {
        int found = 0;
        for (size_t j = 0; j < sizeof k_known_kernels / sizeof k_known_kernels[0]; j++)
            if (strcmp(kernels[i], k_known_kernels[j]) == 0) found = 1;
        if (!found) printf("Note: kernel '%s' is not part of this library.\n", kernels[i]);
    }
    if (program) { program->program = NULL; program->raw_kernels = NULL; program->raw_data = NULL; }
}

void load_program_url(rcl_ctx *ctx, char *url, char **kernels, unsigned int num_kernels,
                      rcl_program *program, char **macros, unsigned int num_macros)
{
    (void)url;
    load_program_raw(ctx, NULL, kernels, num_kernels, program, macros, num_macros);
}

/* reference: src/parallel.c:326-402 */
rt_mem gen_rgb_image(raytracer_context *rctx, const unsigned int width, const unsigned int height)
{
    size_t bytes = (size_t)width * height * sizeof(vec4);
    rt_mem m = rt_mem_alloc(rctx->rcl->device, bytes);
    rt_mem_zero(rctx->rcl->device, m, bytes);
    return m;
}

rt_mem gen_grayscale_buffer(raytracer_context *rctx, const unsigned int width, const unsigned int height)
{
    size_t bytes = (size_t)width * height * sizeof(float);
    rt_mem m = rt_mem_alloc(rctx->rcl->device, bytes);
    rt_mem_zero(rctx->rcl->device, m, bytes);
    return m;
}

rt_mem gen_1d_image(raytracer_context *rctx, size_t t, void *ptr)
{
    rt_mem m = rt_mem_alloc(rctx->rcl->device, t);
    if (ptr && t) rt_mem_upload(rctx->rcl->device, m, ptr, t);
    return m;
}

/* reference: src/parallel.c:282-324 */
rcl_img_buf gen_1d_image_buffer(raytracer_context *rctx, size_t t, void *ptr)
{
    rcl_img_buf ib;
    ib.size = t;
    ib.buffer = gen_1d_image(rctx, t, ptr);
    ib.image = ib.buffer;
    return ib;
}

/* reference: src/parallel.c:404-442 */
void retrieve_buf(raytracer_context *rctx, rt_mem g_buf, void *c_buf, size_t size)
{
    rt_mem_download(rctx->rcl->device, c_buf, g_buf, size);
}

void zero_buffer(raytracer_context *rctx, rt_mem buf, size_t size)
{
    rt_mem_zero(rctx->rcl->device, buf, size);
}

void zero_buffer_img(raytracer_context *rctx, rt_mem buf, size_t element, const unsigned int width,
                     const unsigned int height)
{
    rt_mem_zero(rctx->rcl->device, buf, element * (size_t)width * height);
}

size_t get_workgroup_size(raytracer_context *rctx, void *kernel)
{
    (void)rctx; (void)kernel;
    return 256;
}
