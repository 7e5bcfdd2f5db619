/*
 * ss_raytracer.h -- single-sweep ray caster (primary + shadow + two mirror
 * bounces, brute-force intersection).  Drop-in for the reference's
 * include/ss_raytracer.h:1-15.
 */
#ifndef RT_SS_RAYTRACER_H
#define RT_SS_RAYTRACER_H

#include <stdbool.h>

struct _rt_ctx;

typedef struct ss_raytracer_context {
    struct _rt_ctx *rctx;
    bool up_to_date;
} ss_raytracer_context;

ss_raytracer_context *init_ss_raytracer_context(struct _rt_ctx *rctx);
void ss_raytracer_render(ss_raytracer_context *ctx);    /* reference: src/ss_raytracer.c:7-56 */
void ss_raytracer_prepass(ss_raytracer_context *ctx);   /* reference: src/ss_raytracer.c:72-75 */
void ss_raytracer_destroy(ss_raytracer_context *ctx);   /* not in the reference */

#endif
