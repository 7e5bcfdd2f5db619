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
 *   rt_headless <resource root> <scene.rsc> <width> <height> <samples> <out.ppm> script <events file>
 *
 * The second form replays the reference's event protocol (src/startup.c:118-216, src/ui.c:24-98): the events
 * file holds, one per line, the JSON messages the web UI would send over its socket -- {"type":1,"action":
 * {"type":0|1|2|10}} queues a switch to the single-sweep, path or split path integrator or a clear on
 * rctx->event_stack (32 deep, last in first out, as the reference's), {"type":2} asks for the serialised
 * kd-tree, which is written to <out.ppm>.kdtree -- and the harness's own lines "turn N" (N turns of the render
 * thread's loop: pop at most one event, then one render call of the current integrator) and "frame FILE"
 * (write the caller-owned bitmap as it stands).
 *
 * RT_RNG_SEED=<n> in the environment makes the split path tracer's random
 * streams reproducible.
 */
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kdtree.h"
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

/* ---- the reference's render-thread loop and UI event handler, driven by a script ------------------------- */
typedef struct {
    raytracer_context *rctx;
    ss_raytracer_context *ss;
    path_raytracer_context *path;
    spath_raytracer_context *spath;
    int current;                       /* current_renderer of src/startup.c:116 */
    bool built;
    unsigned int samples;
    uint32_t *bitmap;
} loop_state;

/* the number that follows "key": in a message, -1 if the key is missing */
static int json_number(const char *text, const char *key)
{
    char pat[64];
    snprintf(pat, sizeof pat, "\"%s\"", key);
    const char *p = strstr(text, pat);
    if (!p) return -1;
    p += strlen(pat);
    while (*p == ' ' || *p == '\t' || *p == ':') p++;
    if (*p < '0' || *p > '9') return -1;
    return atoi(p);
}

/* src/ui.c:24-98 without the socket */
static void ui_message(loop_state *st, const char *msg, const char *out_path)
{
    raytracer_context *rctx = st->rctx;
    switch (json_number(msg, "type")) {
    case 0:
        printf("{ \"type\":0, \"message\":\"Nothing Right Now.\"}\n");
        return;
    case 1: {
        const char *action = strstr(msg, "\"action\"");
        const int what = action ? json_number(action, "type") : -1;
        if (what == SS_RAYTRACER || what == PATH_RAYTRACER || what == SPLIT_PATH_RAYTRACER || what == 3) {
            if (rctx->event_position == 32) return;                    /* the stack is full: dropped, as the reference does */
            static const char *name[] = {"Single Bounce", "Path Raytracer", "Split Path Raytracer", "(scene change: not supported)"};
            printf("UI Event Queued: Switch To %s\n", name[what]);
            rctx->event_stack[rctx->event_position++] = (unsigned int)what;
        } else if (what == 10) {
            if (rctx->event_position == 32) return;                    /* (the reference forgets this check) */
            printf("UI Event Queued: Clear\n");
            rctx->event_stack[rctx->event_position++] = 10;
        }
        return;
    }
    case 2: {
        kd_tree *kdt = rctx->stat_scene ? rctx->stat_scene->kdt : NULL;
        printf("GE2 requested k-d tree.\n");
        if (kdt && kdt->buffer) {
            char path[1024];
            snprintf(path, sizeof path, "%s.kdtree", out_path);
            FILE *f = fopen(path, "wb");
            if (f) { fwrite(kdt->buffer, 1, kdt->buffer_size, f); fclose(f); }
        } else {
            printf("ERROR: no k-d tree.\n");
        }
        return;
    }
    default:
        return;
    }
}

/* one turn of the loop of src/startup.c:118-216 */
static void loop_turn(loop_state *st)
{
    raytracer_context *rctx = st->rctx;
    if (rctx->event_position) {
        if (!st->built) {
            raytracer_build(rctx);
            st->built = true;
        }
        const unsigned int ev = rctx->event_stack[--rctx->event_position];
        if ((int)ev != st->current && (ev == SS_RAYTRACER || ev == PATH_RAYTRACER || ev == SPLIT_PATH_RAYTRACER || ev == 10))
            memset(st->bitmap, 0x40, (size_t)rctx->width * rctx->height * 4);      /* os_draw_weird: any placeholder */
        switch (ev) {
        case SS_RAYTRACER:
            printf("Switching To SS Raytracer\n");
            if (st->current == SS_RAYTRACER) break;
            st->current = SS_RAYTRACER;
            if (!st->ss) st->ss = init_ss_raytracer_context(rctx);
            ss_raytracer_prepass(st->ss);
            break;
        case PATH_RAYTRACER:
            printf("Switching To Path Tracer\n");
            if (st->current == PATH_RAYTRACER) break;
            st->current = PATH_RAYTRACER;
            if (!st->path) st->path = init_path_raytracer_context(rctx);
            st->path->num_samples = st->samples;
            path_raytracer_prepass(st->path);
            break;
        case SPLIT_PATH_RAYTRACER:
            printf("Switching To Split Path Tracer\n");
            if (st->current == SPLIT_PATH_RAYTRACER) break;
            st->current = SPLIT_PATH_RAYTRACER;
            if (!st->spath) {
                st->spath = init_spath_raytracer_context(rctx);
                const char *seed = getenv("RT_RNG_SEED");
                if (seed) { st->spath->rng_mode = RT_RNG_SPLITMIX; st->spath->rng_seed = (unsigned int)strtoul(seed, NULL, 10); }
            }
            spath_raytracer_prepass(st->spath);
            st->spath->num_iterations = st->samples * 4;
            break;
        case 10:
            printf("Clearing\n");
            st->current = 10;
            break;
        default:
            break;
        }
    }
    switch (st->current) {
    case SS_RAYTRACER: ss_raytracer_render(st->ss); break;
    case PATH_RAYTRACER: path_raytracer_render(st->path); break;
    case SPLIT_PATH_RAYTRACER: spath_raytracer_render(st->spath); break;
    default: break;
    }
}

static int run_script(loop_state *st, const char *script, const char *out_path)
{
    FILE *f = fopen(script, "r");
    if (!f) { fprintf(stderr, "cannot read %s\n", script); return 1; }
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (*p == '{') {
            ui_message(st, p, out_path);
        } else if (!strncmp(p, "turn", 4)) {
            for (int n = atoi(p + 4); n > 0; n--) loop_turn(st);
        } else if (!strncmp(p, "frame", 5)) {
            char name[1024];
            if (sscanf(p + 5, "%1023s", name) == 1 && write_ppm(name, st->bitmap, st->rctx->width, st->rctx->height))
                fprintf(stderr, "cannot write %s\n", name);
        }
    }
    fclose(f);
    return 0;
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

    if (!strcmp(mode, "script")) {
        if (argc < 9) { fprintf(stderr, "script mode needs an events file\n"); return 2; }
        loop_state st = {0};
        st.rctx = rctx;
        st.current = -1;
        st.samples = samples;
        st.bitmap = bitmap;
        /* the reference builds at the first event (src/startup.c:122-127); the build above is that build, done early */
        st.built = true;
        if (run_script(&st, argv[8], argv[6])) return 1;
    } else if (!strcmp(mode, "ss")) {
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
