/*
 * aabb.c -- axis-aligned box helpers and the few other host-side geometry
 * routines of geom.h (reference: src/geom.c).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "geom.h"

#define GEOM_EPS 0.0000001f

/* reference: src/geom.c:24-34 (column-major) */
float *matvec_mul(mat4 m, vec4 v)
{
    float *out = (float *)malloc(sizeof(vec4));
    for (int r = 0; r < 4; r++)
        out[r] = m[r + 0 * 4] * v[0] + m[r + 1 * 4] * v[1] + m[r + 2 * 4] * v[2] + m[r + 3 * 4] * v[3];
    return out;
}

static void split_box(const AABB *src, uint8_t k, float plane, AABB *left, AABB *right)
{
    *left = *src;
    *right = *src;
    left->max[k] = plane;
    right->min[k] = plane;
}

/* reference: src/geom.c:44-57 -- b is a fraction of the box extent along k */
void AABB_divide(AABB source, uint8_t k, float b, AABB *left, AABB *right)
{
    split_box(&source, k, source.min[k] + (source.max[k] - source.min[k]) * b, left, right);
}

/* reference: src/geom.c:60-72 -- world_b is a world-space coordinate */
void AABB_divide_world(AABB source, uint8_t k, float world_b, AABB *left, AABB *right)
{
    split_box(&source, k, world_b, left, right);
}

/* reference: src/geom.c:75-84 */
float AABB_surface_area(AABB source)
{
    float d0 = source.max[0] - source.min[0];
    float d1 = source.max[1] - source.min[1];
    float d2 = source.max[2] - source.min[2];
    return (d0 * d1 * 2 + d1 * d2 * 2 + d0 * d2 * 2);
}

/* reference: src/geom.c:86-97 */
void AABB_clip(AABB *result, AABB *target, AABB *container)
{
    *result = *target;
    for (int i = 0; i < 3; i++) {
        if (result->min[i] < container->min[i]) result->min[i] = container->min[i];
        if (result->max[i] > container->max[i]) result->max[i] = container->max[i];
    }
}

static void box_reset(AABB *b)
{
    for (int k = 0; k < 3; k++) { b->min[k] = 1000000; b->max[k] = -1000000; }
}

static void box_grow(AABB *b, const float *p)
{
    for (int k = 0; k < 3; k++) {
        if (p[k] < b->min[k]) b->min[k] = p[k];
        if (p[k] > b->max[k]) b->max[k] = p[k];
    }
}

/* reference: src/geom.c:99-120 */
void AABB_construct_from_triangle(AABB *result, ivec3 *indices, vec3 *vertices)
{
    box_reset(result);
    for (int i = 0; i < 3; i++) box_grow(result, vertices[indices[i][0]]);
}

/* reference: src/geom.c:122-141 */
void AABB_construct_from_vertices(AABB *result, vec3 *vertices, unsigned int num_vertices)
{
    box_reset(result);
    for (unsigned int i = 0; i < num_vertices; i++) box_grow(result, vertices[i]);
}

/* reference: src/geom.c:143-148 */
bool AABB_is_planar(AABB *source, uint8_t k) { return source->max[k] - source->min[k] <= GEOM_EPS; }

/* reference: src/geom.c:150-153 */
float AABB_ilerp(AABB source, uint8_t k, float world_b)
{
    return (world_b - source.min[k]) / (source.max[k] - source.min[k]);
}

/* reference: src/geom.c:155-177 (quadratic solved as in :5-22); -1 = no hit */
float does_collide_sphere(sphere s, ray r)
{
    float L[3] = {r.orig[0] - s.pos[0], r.orig[1] - s.pos[1], r.orig[2] - s.pos[2]};
    float a = 1.0f;
    float b = (r.dir[0] * L[0] + r.dir[1] * L[1] + r.dir[2] * L[2]) * 2.0f;
    float c = (L[0] * L[0] + L[1] * L[1] + L[2] * L[2]) - (s.radius * s.radius);
    float discr = b * b - 4 * a * c;
    float t0, t1;
    if (discr < 0) return -1.0f;
    if (discr == 0) t0 = t1 = -0.5 * b / a;
    else {
        float q = (b > 0) ? -0.5 * (b + sqrt(discr)) : -0.5 * (b - sqrt(discr));
        t0 = q / a;
        t1 = c / q;
    }
    if (t0 > t1) { float t = t0; t0 = t1; t1 = t; }
    if (t0 < 0) {
        t0 = t1;
        if (t0 < 0) return -1.0f;
    }
    return t0;
}

/* reference: src/geom.c:179-192 (returns t only for intersections behind the
   origin, as the reference does) */
float does_collide_plane(plane p, ray r)
{
    float denom = r.dir[0] * p.norm[0] + r.dir[1] * p.norm[1] + r.dir[2] * p.norm[2];
    if (denom > 1e-6) {
        float l[3] = {p.pos[0] - r.orig[0], p.pos[1] - r.orig[1], p.pos[2] - r.orig[2]};
        float t = (l[0] * p.norm[0] + l[1] * p.norm[1] + l[2] * p.norm[2]) / denom;
        if (t >= 0) return -1.0;
        return t;
    }
    return -1.0;
}

/* reference: src/geom.c:194-218 */
ray generate_ray(int x, int y, int width, int height, float fov)
{
    ray r;
    memset(&r, 0, sizeof r);
    float aspect_ratio = width / (float)height;
    float cam_x = (2 * (((float)x + 0.5) / width) - 1) * tan(fov / 2 * M_PI / 180) * aspect_ratio;
    float cam_y = (1 - 2 * (((float)y + 0.5) / height)) * tan(fov / 2 * M_PI / 180);
    float len = sqrtf(cam_x * cam_x + cam_y * cam_y + 1.0f);
    r.dir[0] = cam_x / len;
    r.dir[1] = cam_y / len;
    r.dir[2] = -1.0f / len;
    return r;
}
