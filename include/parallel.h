/*
 * parallel.h -- device context and buffer helpers the host code calls.
 *
 * Drop-in for the reference's include/parallel.h:15-68 with the OpenCL handle
 * types replaced by the HIP C-ABI layer's (rt_hip.h).  Field and function
 * names are kept; there is no OpenCL behind them and no run-time kernel
 * compilation (kernels are built ahead of time for gfx950).
 */
#ifndef RT_PARALLEL_H
#define RT_PARALLEL_H

#include <stddef.h>
#include "geom.h"
#include "rt_hip.h"

typedef struct _rt_ctx raytracer_context;

/* reference: include/parallel.h:15-27 */
typedef struct {
    int        device_ordinal;   /* which GPU create_context opens; -1 = LOCAL_RANK env or 0 */
    rt_device *device;           /* replaces platform_id/device_id/context/commands */

    unsigned int simt_size;                     /* 64 on gfx950 */
    unsigned int num_simt_per_multiprocessor;   /* SIMDs per CU */
    unsigned int num_multiprocessors;           /* CUs */
    unsigned int num_cores;
} rcl_ctx;

/* reference: include/parallel.h:29-35.  Nothing is compiled at run time; the
   struct survives so raytracer_context keeps its `program` member. */
typedef struct {
    void  *program;
    void **raw_kernels;
    char  *raw_data;
} rcl_program;

/* reference: include/parallel.h:37-42 -- a buffer that the reference also
   wrapped in a 1-D image view; here both members name the same allocation */
typedef struct rcl_img_buf {
    rt_mem buffer;
    rt_mem image;
    size_t size;
} rcl_img_buf;

void cl_info(void);                       /* prints the visible GPUs (src/parallel.c:4-83) */
void create_context(rcl_ctx *context);    /* src/parallel.c:93-279 */

/* src/parallel.c:444-558: kept for source compatibility; they only check that
   every requested kernel name is one this library implements */
void load_program_raw(rcl_ctx *ctx, char *data, char **kernels, unsigned int num_kernels,
                      rcl_program *program, char **macros, unsigned int num_macros);
void load_program_url(rcl_ctx *ctx, char *url, char **kernels, unsigned int num_kernels,
                      rcl_program *program, char **macros, unsigned int num_macros);

/* src/parallel.c:282-402 */
rt_mem      gen_rgb_image(raytracer_context *rctx, const unsigned int width, const unsigned int height);
rt_mem      gen_grayscale_buffer(raytracer_context *rctx, const unsigned int width, const unsigned int height);
rt_mem      gen_1d_image(raytracer_context *rctx, size_t t, void *ptr);
rcl_img_buf gen_1d_image_buffer(raytracer_context *rctx, size_t t, void *ptr);

/* src/parallel.c:404-442 */
void retrieve_buf(raytracer_context *rctx, rt_mem g_buf, void *c_buf, size_t size);
void zero_buffer(raytracer_context *rctx, rt_mem buf, size_t size);
void zero_buffer_img(raytracer_context *rctx, rt_mem buf, size_t element,
                     const unsigned int width, const unsigned int height);
size_t get_workgroup_size(raytracer_context *rctx, void *kernel);

/* monotonic milliseconds; stands in for os_get_time_mili(abst) */
unsigned int rt_time_ms(void);

#endif
