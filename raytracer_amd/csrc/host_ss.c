/*
 * host_ss.c -- single-sweep ray caster driver.  reference: src/ss_raytracer.c:7-75.
 * The reference slices the frame into 16 launches to stay under a display
 * watchdog; one launch covers the frame here.
 */
#include <stdio.h>
#include <stdlib.h>

#include "raytracer.h"
#include "ss_raytracer.h"

void rt_host_cast_rays(raytracer_context *rctx);

ss_raytracer_context *init_ss_raytracer_context(struct _rt_ctx *rctx)
{
    ss_raytracer_context *c = (ss_raytracer_context *)malloc(sizeof(ss_raytracer_context));
    c->rctx = rctx;
    c->up_to_date = false;
    return c;
}

void ss_raytracer_prepass(ss_raytracer_context *c) { raytracer_prepass(c->rctx); }

void ss_raytracer_render(ss_raytracer_context *c)
{
    _raytracer_gen_ray_buffer(c->rctx);
    rt_host_cast_rays(c->rctx);
}

void ss_raytracer_destroy(ss_raytracer_context *c) { free(c); }
