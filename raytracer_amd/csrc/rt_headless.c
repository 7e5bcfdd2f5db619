/*
 * rt_headless -- the reference's render thread without a window.
 *
 * Drives the library with the call sequence of the reference's run()
 * (src/startup.c:79-218): cl_info, create_context, raytracer_init,
 * load_scene_json_url, raytracer_build, then the chosen integrator's
 * init / prepass / render until it reports completion, and writes the
 * caller-owned 0xAARRGGBB frame as a binary PPM where the reference would
 * have blitted it to the window.
 *
 *   rt_headless <resource root> <scene.rsc> <width> <height> <samples> <out.ppm> [ss|path|spath]
 *
 * RT_RNG_SEED=<n> in the environment makes the split path tracer's random
 * streams reproducible.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "loader.h"
#include "path_raytracer.h"
#include "raytracer.h"
#include "spath_raytracer.h"
#include "ss_raytracer.h"

static int write_ppm(const char *path, const uint32_t *frame, unsigned int w, unsigned int h)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const unsigned char rgb[3] = {(unsigned char)(frame[i] >> 16), (unsigned char)(frame[i] >> 8), (unsigned char)frame[i]};
        fwrite(rgb, 1, 3, f);
    }
    return fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s <resource root> <scene.rsc> <width> <height> <samples> <out.ppm> [ss|path|spath]\n", argv[0]);
        return 2;
    }
    const unsigned int width = (unsigned int)atoi(argv[3]), height = (unsigned int)atoi(argv[4]);
    const unsigned int samples = (unsigned int)atoi(argv[5]);
    const char *mode = argc > 7 ? argv[7] : "spath";
    if (!width || !height || !samples) { fprintf(stderr, "width, height and samples must be positive\n"); return 2; }

    uint32_t *bitmap = (uint32_t *)calloc((size_t)width * height, sizeof(uint32_t));
    rt_set_resource_root(argv[1]);

    cl_info();
    rcl_ctx *rcl = (rcl_ctx *)calloc(1, sizeof(rcl_ctx));
    rcl->device_ordinal = -1;
    create_context(rcl);
    raytracer_context *rctx = raytracer_init(width, height, bitmap, rcl);
    rctx->stat_scene = load_scene_json_url(argv[2]);
    rctx->num_samples = samples;
    raytracer_build(rctx);

    if (!strcmp(mode, "ss")) {
        ss_raytracer_context *c = init_ss_raytracer_context(rctx);
        ss_raytracer_prepass(c);
        ss_raytracer_render(c);
    } else if (!strcmp(mode, "path")) {
        path_raytracer_context *c = init_path_raytracer_context(rctx);
        c->num_samples = samples;
        path_raytracer_prepass(c);
        while (!c->render_complete) path_raytracer_render(c);
    } else {
        spath_raytracer_context *c = init_spath_raytracer_context(rctx);
        const char *seed = getenv("RT_RNG_SEED");    /* reproducible RNG streams instead of srand(clock) / rand() */
        if (seed) { c->rng_mode = RT_RNG_SPLITMIX; c->rng_seed = (unsigned int)strtoul(seed, NULL, 10); }
        spath_raytracer_prepass(c);                  /* sets the reference's default of 2048 samples (src/spath_raytracer.c:408) */
        c->num_iterations = samples * 4;             /* four bounces (= iterations) per sample */
        while (!c->render_complete) spath_raytracer_render(c);
    }
    if (write_ppm(argv[6], bitmap, width, height)) { fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
    printf("wrote %s (%ux%u)\n", argv[6], width, height);
    return 0;
}
