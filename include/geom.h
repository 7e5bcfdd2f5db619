/*
 * geom.h -- host-side geometry types and AABB helpers.
 *
 * Drop-in for the reference's include/geom.h:1-79: same type names, field
 * names and function signatures, so code written against the reference
 * compiles unchanged.  Vectors are 4 floats wide (one float of padding), as the
 * device expects.
 */
#ifndef RT_GEOM_H
#define RT_GEOM_H

#include <stdbool.h>
#include <stdint.h>

#define RT_ALIGNED(n) __attribute__((aligned(n)))

typedef int   ivec3[4];
typedef float vec2[2];
typedef float vec3[4];
typedef float vec4[4];
typedef float mat4[16];

/* reference: include/geom.h:15-20 */
typedef struct ray {
    vec3 orig;
    vec3 dir;
} ray;

/* reference: include/geom.h:27-31 (max before min) */
typedef struct AABB {
    vec3 max;
    vec3 min;
} AABB;

/* reference: include/geom.h:49-57, 32 bytes */
typedef struct {
    vec4  pos;
    float radius;
    int   material_index;
} RT_ALIGNED(16) sphere;

/* reference: include/geom.h:67-75, 48 bytes */
typedef struct plane {
    vec4 pos;
    vec4 norm;
    int  material_index;
} RT_ALIGNED(16) plane;

/* reference: src/geom.c:44-153 */
void  AABB_divide(AABB source, uint8_t k, float b, AABB *left, AABB *right);
void  AABB_divide_world(AABB source, uint8_t k, float world_b, AABB *left, AABB *right);
float AABB_surface_area(AABB source);
void  AABB_clip(AABB *result, AABB *target, AABB *container);
float AABB_ilerp(AABB source, uint8_t k, float world_b);
bool  AABB_is_planar(AABB *source, uint8_t k);
void  AABB_construct_from_vertices(AABB *result, vec3 *vertices, unsigned int num_vertices);
void  AABB_construct_from_triangle(AABB *result, ivec3 *indices, vec3 *vertices);

/* reference: src/geom.c:155-218 (CPU-side helpers, not used by the renderers) */
float does_collide_sphere(sphere s, ray r);
float does_collide_plane(plane p, ray r);
ray   generate_ray(int x, int y, int width, int height, float fov);

/* reference: src/geom.c:24-34 -- column-major mat4 * vec4; the result is
   malloc'd and owned by the caller */
float *matvec_mul(mat4 m, vec4 v);

#endif
