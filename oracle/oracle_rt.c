/*
 * oracle_rt.c -- CPU restatement of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under raytracer_amd/ may include, link or
 * call this file; it exists so that tests/, the smoke check and the benchmark's
 * cpu_baseline leg can compare the HIP path with an independent statement of
 * the reference algorithm.
 *
 * What it restates (citations are path:line in the reference checkout):
 *   - SAH kd-tree builder + serialiser ........ src/kdtree.c:82-102,213-459,469-553
 *     AABB helpers ............................. src/geom.c:44-153
 *   - generate_rays ............................ src/kernels/ray.cl:109-133
 *   - kdtree_intersection (on the serialised blob, exactly the reference's
 *     byte layout) ............................. src/kernels/kdtree.cl:49-75,90-321
 *   - Moeller-Trumbore, slab tests, sphere/plane src/kernels/collision.cl:71-230
 *   - brute-force collide_all .................. src/kernels/collision.cl:240-285,382-468
 *   - segmented_path_trace_init / _trace ....... src/kernels/path.cl:36-258
 *   - xorshift_batch, resolve, averages ........ src/kernels/path.cl:394-458
 *   - get_random, get_colour, reflect .......... src/kernels/util.cl:57-108
 *   - cast_ray_test + shade .................... src/kernels/ray.cl:2-105
 *   - path_trace ............................... src/kernels/path.cl:260-366
 *
 * Pinning status.
 *   Builder + serialiser: PINNED.  oracle/_ref builds the reference's own
 *   kdtree.c/geom.c from /root/reference and tests compare blobs byte for byte
 *   (padding bytes, which the reference leaves uninitialised, are masked).
 *   Device kernels: PINNED against the reference's own kernels.  The kernel
 *   files compile unmodified for gfx950 with ROCm's clang and OpenCL library
 *   (oracle/build_ref_kernels.py -> oracle/_ref/); tests/test_gpu_refkernels.py
 *   runs them on the GPU (oracle/refcl.py) and holds them against this file:
 *   kdtree_intersection bit for bit once dot/cross/min/max are defined as
 *   ROCm's library defines them (orc_set_builtins(1)), the shading kernels one
 *   step at a time to 1e-5, whole renders as radiance.  Outputs of those
 *   kernels are frozen in tests/golden/refkernel_ico4.npz and checked by the
 *   CPU suite, so the pin holds where there is no GPU.  A host build of the
 *   kernels would need stand-ins for an OpenCL built-in library and is not
 *   attempted.
 *
 * Arithmetic conventions (the OpenCL spec leaves these to the implementation;
 * this oracle fixes ONE conformant choice so a GPU result can be compared bit
 * for bit):
 *   - no contraction of the source's own a*b+c (build with -ffp-contract=off);
 *   - dot / cross / min / max as ROCm's OpenCL library defines them on gfx950
 *     (the default, mode 1 below): dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)),
 *     cross with one fma per component, min/max = minNum/maxNum.  With these the
 *     reference's compiled kdtree_intersection equals this file bit for bit.
 *     Mode 0 is the plain reading: dot summed left to right, textbook cross,
 *     min(a,b) = b<a ? b : a, max(a,b) = a<b ? b : a  (OpenCL 1.2, 6.12.4);
 *   - normalize(v) = v / sqrt(dot(v,v)), IEEE sqrt and divide;
 *   - cos/sin of the hemisphere angle use orc_sincos() below, a <=2 ulp
 *     Cody-Waite + minimax evaluation written in plain float operations
 *     (OpenCL allows 4 ulp); ORC_LIBM_TRIG=1 in the environment switches
 *     to the C library's cosf/sinf to measure what that choice is worth;
 *   - uninitialised sample_num/bounce_num in segmented_path_trace_init
 *     (path.cl:65-67) are taken to be 0;
 *   - a traversal stack deeper than the reference's 16 entries
 *     (kdtree.cl:78,120 -- undefined behaviour there) is simply allowed to
 *     grow; rays that needed it are counted in orc_kd_stats.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))
#define ORC_API_DECL __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* small vector helpers                                                       */
/* ------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_ld(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
/* Which definition of the OpenCL built-ins dot / cross / min / max is in force.
 * 1 (default): the one ROCm's OpenCL library has on gfx950 (read off the code
 * it links into the reference's kernels): dot = fma(a.z,b.z, fma(a.y,b.y,
 * a.x*b.x)), cross.x = fma(a.y,b.z, -(b.y*a.z)) and so on, min/max = IEEE
 * minNum/maxNum; the reference's kernels, compiled by ROCm's clang and run on
 * the GPU (oracle/refcl.py), equal this restatement bit for bit with it, and
 * the library is built to it.  0: the plain reading of the header comment, kept
 * to measure what the choice is worth.  Nothing else differs between the modes. */
static int g_builtins = 1;
ORC_API_DECL void orc_set_builtins(int mode) { g_builtins = mode; }

static inline float v3_dot(v3 a, v3 b)
{
    if (g_builtins == 1) return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
    return (a.x * b.x + a.y * b.y) + a.z * b.z;
}
static inline v3 v3_cross(v3 a, v3 b)
{
    if (g_builtins == 1)
        return V3(fmaf(a.y, b.z, -(b.y * a.z)), fmaf(a.z, b.x, -(b.z * a.x)), fmaf(a.x, b.y, -(b.x * a.y)));
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float cl_min(float a, float b) { return g_builtins == 1 ? fminf(a, b) : (b < a ? b : a); }
static inline float cl_max(float a, float b) { return g_builtins == 1 ? fmaxf(a, b) : (a < b ? b : a); }
static inline v3 v3_normalize(v3 a)
{
    float len = sqrtf(v3_dot(a, a));
    return V3(a.x / len, a.y / len, a.z / len);
}

/* ------------------------------------------------------------------------- */
/* trig for theta in [0, 2*pi]                                                */
/* ------------------------------------------------------------------------- */
static int g_libm_trig = -1;

static void orc_sincos(float theta, float *s_out, float *c_out)
{
    if (g_libm_trig < 0) {
        const char *e = getenv("ORC_LIBM_TRIG");
        g_libm_trig = (e && e[0] == '1') ? 1 : 0;
    }
    if (g_libm_trig) { *s_out = sinf(theta); *c_out = cosf(theta); return; }

    /* quadrant: q = round(theta * 2/pi), theta >= 0 */
    int q = (int)(theta * 0.636619772367581343f + 0.5f);
    float fq = (float)q;
    /* pi/2 split in three parts; the first two have short mantissas so the
       products with q <= 4 are exact */
    float r = theta - fq * 1.5703125f;
    r = r - fq * 4.837512969970703125e-4f;
    r = r - fq * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    switch (q & 3) {
    case 0: *s_out = sp;  *c_out = cp;  break;
    case 1: *s_out = cp;  *c_out = -sp; break;
    case 2: *s_out = -sp; *c_out = -cp; break;
    default: *s_out = -cp; *c_out = sp; break;
    }
}

ORC_API void orc_sincos_array(const float *theta, float *s, float *c, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) orc_sincos(theta[i], &s[i], &c[i]);
}

/* ========================================================================= */
/* 1. kd-tree builder (src/kdtree.c, src/geom.c)                              */
/* ========================================================================= */

/* Field order follows the reference AABB {max,min} (include/geom.h:27-31). */
typedef struct { float max[4]; float min[4]; } orc_aabb;

typedef struct orc_node {
    uint8_t k;
    float b;
    struct orc_node *left, *right;
    uint32_t *tris;     /* index-buffer offsets (3 * triangle number) */
    uint32_t ntris;
} orc_node;

typedef struct {
    orc_node *root;
    uint32_t num_nodes_total, num_tris_padded, num_traversal_nodes, num_leaves, num_indices_total;
    uint32_t max_recurse, leaf_threshold;
    uint32_t max_depth;          /* instrumentation, not in the reference */
    uint32_t count_mismatch;     /* nodes where sweep counts != classification (reference UB) */
    orc_aabb bounds;
    const float *verts;          /* 4 floats per vertex */
    const int32_t *idx;          /* 4 ints per index: v, vn, vt, mesh */
    uint32_t nverts, nidx;
} orc_tree;

enum { EV_END = 0, EV_PLANAR = 1, EV_START = 2 };   /* kdtree.c:10-12 */
enum { SIDE_BOTH = 0, SIDE_LEFT = 1, SIDE_RIGHT = 2 };

typedef struct { float b; uint8_t type; } orc_event;

/* geom.c:75-84 */
static float aabb_area(const orc_aabb *a)
{
    float d0 = a->max[0] - a->min[0], d1 = a->max[1] - a->min[1], d2 = a->max[2] - a->min[2];
    return (d0 * d1 * 2 + d1 * d2 * 2 + d0 * d2 * 2);
}

/* geom.c:143-148 */
static int aabb_planar(const orc_aabb *a, int k) { return (a->max[k] - a->min[k] <= 0.0000001f); }

/* geom.c:99-120 */
static void aabb_from_tri(orc_aabb *r, const orc_tree *t, uint32_t off)
{
    for (int k = 0; k < 3; k++) { r->min[k] = 1000000; r->max[k] = -1000000; }
    for (int i = 0; i < 3; i++) {
        const float *v = t->verts + 4 * (size_t)t->idx[4 * (size_t)(off + i)];
        for (int k = 0; k < 3; k++) {
            if (v[k] < r->min[k]) r->min[k] = v[k];
            if (v[k] > r->max[k]) r->max[k] = v[k];
        }
    }
}

/* kdtree.c:66-73 */
static float sah_lambda(int NL, int NR, float PL, float PR)
{
    const float E = 0.001f;
    if ((NL <= E || NR <= E) && !(PL >= 1.0f - E || PR >= 1.0f - E)) return 0.8f;
    return 1.0f;
}

/* kdtree.c:77-80 */
static float sah_cost(float PL, float PR, uint32_t NL, uint32_t NR)
{
    return sah_lambda((int)NL, (int)NR, PL, PR) * (2.0f + 1.0f * (PL * NL + PR * NR));
}

/* kdtree.c:82-102 with geom.c:44-57 (AABB_divide) folded in */
static float sah_eval(int k, float b_rel, const orc_aabb *V, int NL, int NR, int NP, uint8_t *side)
{
    orc_aabb VL = *V, VR = *V;
    float split = V->min[k] + (V->max[k] - V->min[k]) * b_rel;
    VL.max[k] = split;
    VR.min[k] = split;
    float PL = aabb_area(&VL) / aabb_area(V);
    float PR = aabb_area(&VR) / aabb_area(V);
    if (PL >= 1 - 0.001f || PR >= 1 - 0.001f) { *side = SIDE_BOTH; return 1000000000.0f; }
    float CPL = sah_cost(PL, PR, (uint32_t)(NL + NP), (uint32_t)NR);
    float CPR = sah_cost(PL, PR, (uint32_t)NL, (uint32_t)(NR + NP));
    if (CPL < CPR) { *side = SIDE_LEFT; return CPL; }
    *side = SIDE_RIGHT;
    return CPR;
}

/* kdtree.c:50-56 restricted to one axis (k is constant inside one buffer) */
static int event_cmp(const void *pa, const void *pb)
{
    const orc_event *a = (const orc_event *)pa, *b = (const orc_event *)pb;
    if (a->b < b->b) return -1;
    if (b->b < a->b) return 1;
    return (int)a->type - (int)b->type;
}

typedef struct { float b; uint8_t k, side; uint32_t NL, NR, NP; float cost; } orc_plane;

/* kdtree.c:213-303.  The sweep only looks at (b, type) of each event and at
   run lengths, so any sort that orders by b then type gives the same plane. */
static orc_plane find_plane(const orc_tree *t, const orc_aabb *V, const uint32_t *tris, uint32_t n)
{
    orc_plane best;
    memset(&best, 0, sizeof best);
    float best_cost = INFINITY;
    orc_event *ev = (orc_event *)malloc(sizeof(orc_event) * 2 * (size_t)n);

    for (int k = 0; k < 3; k++) {
        uint32_t ne = 0;
        for (uint32_t i = 0; i < n; i++) {
            orc_aabb tv, B;
            aabb_from_tri(&tv, t, tris[i]);
            B = tv;                                  /* geom.c:86-97 */
            for (int a = 0; a < 3; a++) {
                if (B.min[a] < V->min[a]) B.min[a] = V->min[a];
                if (B.max[a] > V->max[a]) B.max[a] = V->max[a];
            }
            if (aabb_planar(&B, k)) {
                ev[ne].b = B.min[k]; ev[ne++].type = EV_PLANAR;
            } else {
                ev[ne].b = B.min[k]; ev[ne++].type = EV_START;
                ev[ne].b = B.max[k]; ev[ne++].type = EV_END;
            }
        }
        qsort(ev, ne, sizeof(orc_event), event_cmp);

        int NL = 0, NP = 0, NR = (int)n;
        for (uint32_t i = 0; i < ne;) {
            float pb = ev[i].b;
            int Ps = 0, Pe = 0, Pp = 0;
            while (i < ne && ev[i].b == pb && ev[i].type == EV_END) { Pe++; i++; }
            while (i < ne && ev[i].b == pb && ev[i].type == EV_PLANAR) { Pp++; i++; }
            while (i < ne && ev[i].b == pb && ev[i].type == EV_START) { Ps++; i++; }
            NP = Pp; NR -= Pp; NR -= Pe;
            uint8_t side;
            float rel = (pb - V->min[k]) / (V->max[k] - V->min[k]);   /* geom.c:150-153 */
            float cost = sah_eval(k, rel, V, NL, NR, NP, &side);
            if (cost < best_cost) {
                best_cost = cost;
                best.b = pb; best.k = (uint8_t)k; best.side = side;
                best.NL = (uint32_t)NL; best.NR = (uint32_t)NR; best.NP = (uint32_t)NP;
                best.cost = cost;
            }
            NL += Ps; NL += NP; NP = 0;
        }
    }
    free(ev);
    return best;
}

static orc_node *build_rec(orc_tree *t, orc_aabb V, uint32_t *tris, uint32_t n, uint32_t depth)
{
    orc_node *node = (orc_node *)calloc(1, sizeof(orc_node));
    node->b = 0.5f;
    t->num_nodes_total++;
    if (depth > t->max_depth) t->max_depth = depth;

    /* kdtree.c:362-373 */
    int stop = 0;
    for (int k = 0; k < 3; k++) if (aabb_planar(&V, k)) stop = 1;
    if (depth == t->max_recurse) stop = 1;
    if (n <= t->leaf_threshold) stop = 1;

    orc_plane p;
    if (!stop) {
        p = find_plane(t, &V, tris, n);
        if (p.cost > 1.0f * (float)n) stop = 1;          /* kdtree.c:392 */
    }
    if (stop) {
        node->tris = tris; node->ntris = n;
        t->num_leaves++;
        t->num_indices_total += n;
        t->num_tris_padded += n % 8;
        return node;
    }
    t->num_traversal_nodes++;
    node->k = p.k; node->b = p.b;

    orc_aabb VL = V, VR = V;                              /* geom.c:60-72 */
    VL.max[p.k] = p.b;
    VR.min[p.k] = p.b;

    /* kdtree.c:305-360.  The reference sizes the child lists from the sweep
       counts; here they are sized from the classification itself and a
       disagreement (heap overflow / garbage entries in the reference) is
       recorded instead. */
    uint32_t *TL = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *TR = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t nl = 0, nr = 0;
    for (uint32_t i = 0; i < n; i++) {
        int isL = 0, isR = 0;
        for (int j = 0; j < 3; j++) {
            float c = t->verts[4 * (size_t)t->idx[4 * (size_t)(tris[i] + j)] + p.k];
            if (c < p.b) isL = 1;
            if (c > p.b) isR = 1;
        }
        if (isL && isR) { TR[nr++] = tris[i]; TL[nl++] = tris[i]; }
        else if (!isL && !isR) { if (p.side == SIDE_LEFT) TL[nl++] = tris[i]; else TR[nr++] = tris[i]; }
        else if (isL) TL[nl++] = tris[i];
        else TR[nr++] = tris[i];
    }
    uint32_t expL = p.NL + (p.side == SIDE_LEFT ? p.NP : 0);
    uint32_t expR = p.NR + (p.side == SIDE_RIGHT ? p.NP : 0);
    if (expL != nl || expR != nr) t->count_mismatch++;

    node->left = build_rec(t, VL, TL, nl, depth + 1);
    node->right = build_rec(t, VR, TR, nr, depth + 1);
    return node;
}

ORC_API orc_tree *orc_kd_build(const float *verts4, uint32_t nverts, const int32_t *idx4, uint32_t nidx)
{
    orc_tree *t = (orc_tree *)calloc(1, sizeof(orc_tree));
    t->max_recurse = 50; t->leaf_threshold = 2;           /* kdtree.c:172-174 */
    t->verts = verts4; t->idx = idx4; t->nverts = nverts; t->nidx = nidx;
    if (nidx == 0) return t;

    orc_aabb V;                                           /* geom.c:122-141 */
    for (int k = 0; k < 3; k++) { V.min[k] = 1000000; V.max[k] = -1000000; }
    V.min[3] = V.max[3] = 0;
    for (uint32_t i = 0; i < nverts; i++)
        for (int k = 0; k < 3; k++) {
            float c = verts4[4 * (size_t)i + k];
            if (c < V.min[k]) V.min[k] = c;
            if (c > V.max[k]) V.max[k] = c;
        }
    t->bounds = V;
    uint32_t n = nidx / 3;
    uint32_t *tris = (uint32_t *)malloc(sizeof(uint32_t) * n);
    for (uint32_t i = 0; i < n; i++) tris[i] = i * 3;
    t->root = build_rec(t, V, tris, n, 0);
    return t;
}

static void free_rec(orc_node *n)
{
    if (!n) return;
    free_rec(n->left); free_rec(n->right);
    free(n->tris); free(n);
}
ORC_API void orc_kd_free(orc_tree *t) { if (t) { free_rec(t->root); free(t); } }

/* counts: [nodes_total, traversal, leaves, indices_total, tris_padded, max_depth, count_mismatch] */
ORC_API void orc_kd_counts(const orc_tree *t, uint32_t *out7)
{
    out7[0] = t->num_nodes_total; out7[1] = t->num_traversal_nodes; out7[2] = t->num_leaves;
    out7[3] = t->num_indices_total; out7[4] = t->num_tris_padded; out7[5] = t->max_depth;
    out7[6] = t->count_mismatch;
}
ORC_API void orc_kd_bounds(const orc_tree *t, float *min3, float *max3)
{
    for (int k = 0; k < 3; k++) { min3[k] = t->bounds.min[k]; max3[k] = t->bounds.max[k]; }
}

/* kdtree.c:525-531 -- the reference's (over-)estimate of the blob size */
ORC_API uint32_t orc_kd_blob_size(const orc_tree *t)
{
    return t->num_traversal_nodes * 32u + t->num_leaves * 16u +
           (t->num_indices_total + t->num_tris_padded) * 4u;
}

/* kdtree.c:469-513.  Interior node: u8 type=2 @0, u8 k @1, f32 b @4, u64 left @8,
   u64 right @16 (8-byte word indices), 32 bytes.  Leaf: u8 type=1 @0, u32 n @4,
   16 bytes, then n u32 offsets padded to 8 bytes.  Padding is written as zero
   here (the reference leaves it uninitialised). */
static uint32_t serialize_rec(const orc_node *n, uint8_t *buf, uint32_t off)
{
    if (!n->left) {
        buf[off] = 1;
        memcpy(buf + off + 4, &n->ntris, 4);
        off += 16;
        for (uint32_t i = 0; i < n->ntris; i++) { memcpy(buf + off, &n->tris[i], 4); off += 4; }
        if (n->ntris % 2) off += 4;
        return off;
    }
    uint32_t start = off;
    off += 32;
    uint32_t left_end = serialize_rec(n->left, buf, off);
    uint32_t right_end = serialize_rec(n->right, buf, left_end);
    uint64_t li = off / 8, ri = left_end / 8;
    buf[start] = 2; buf[start + 1] = n->k;
    memcpy(buf + start + 4, &n->b, 4);
    memcpy(buf + start + 8, &li, 8);
    memcpy(buf + start + 16, &ri, 8);
    return right_end;
}

/* returns bytes actually used (<= orc_kd_blob_size) */
ORC_API uint32_t orc_kd_serialize(const orc_tree *t, uint8_t *buf)
{
    if (!t->root) return 0;
    memset(buf, 0, orc_kd_blob_size(t));
    return serialize_rec(t->root, buf, 0);
}

/* SCENE_MIN / SCENE_MAX as the kernels see them: printed with %f and re-read
   as a double literal converted to float (src/raytracer.c:53-64). */
ORC_API void orc_scene_bounds_rounded(const float *min3, const float *max3, float *omin3, float *omax3)
{
    char s[64];
    for (int k = 0; k < 3; k++) {
        snprintf(s, sizeof s, "%f", min3[k]); omin3[k] = (float)strtod(s, NULL);
        snprintf(s, sizeof s, "%f", max3[k]); omax3[k] = (float)strtod(s, NULL);
    }
}

/* ========================================================================= */
/* 2. primary rays (ray.cl:109-133)                                           */
/* ========================================================================= */
/* rays: 8 floats per ray {orig.xyz, pad, dir.xyz, pad}.  The image may be a
   strip of rows [y0, y0+h) of a full_h-high frame (full_h = h, y0 = 0 for the
   reference's single-device case). */
ORC_API void orc_generate_rays(float *rays, uint32_t w, uint32_t h, uint32_t y0, uint32_t full_h)
{
    const float tan_half = tanf(80.0f / 2 * 3.14159274101257f / 180);
    const float aspect = w / (float)full_h;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            float cam_x = (float)((2 * (((float)x + 0.5) / w) - 1) * tan_half * aspect);
            float cam_y = (float)((1 - 2 * (((float)(y + y0) + 0.5) / full_h)) * tan_half);
            v3 d = v3_normalize(V3(cam_x, cam_y, -1.0f));
            float *r = rays + 8 * ((size_t)x + (size_t)y * w);
            r[0] = r[1] = r[2] = r[3] = 0; r[4] = d.x; r[5] = d.y; r[6] = d.z; r[7] = 0;
        }
}

/* ========================================================================= */
/* 3. intersection primitives (collision.cl)                                  */
/* ========================================================================= */
#define ORC_EPS 0.0000001f
#define ORC_FAR 100000000.0f

/* collision.cl:139-172.  Returns 1 on hit; tuv = {t,u,v}. */
static inline __attribute__((always_inline)) int tri_hit(v3 v0, v3 v1, v3 v2, v3 o, v3 d, float *t, float *u, float *v)
{
    v3 ab = v3_sub(v1, v0), ac = v3_sub(v2, v0);
    v3 pvec = v3_cross(d, ac);
    float det = v3_dot(ab, pvec);
    if (det < ORC_EPS) return 0;
    float inv = 1.f / det;
    v3 tvec = v3_sub(o, v0);
    *u = v3_dot(tvec, pvec) * inv;
    if (*u < 0 || *u > 1) return 0;
    v3 qvec = v3_cross(tvec, ab);
    *v = v3_dot(d, qvec) * inv;
    if (*v < 0 || *u + *v > 1) return 0;
    *t = v3_dot(ac, qvec) * inv;
    return 1;
}

/* collision.cl:71-86 */
static inline __attribute__((always_inline)) int slab_t(v3 vmin, v3 vmax, v3 o, v3 d, float *tmin, float *tmax)
{
    v3 inv = V3(1 / d.x, 1 / d.y, 1 / d.z);
    v3 t0 = v3_mul(v3_sub(vmin, o), inv), t1 = v3_mul(v3_sub(vmax, o), inv);
    v3 sm = V3(cl_min(t0.x, t1.x), cl_min(t0.y, t1.y), cl_min(t0.z, t1.z));
    v3 bg = V3(cl_max(t0.x, t1.x), cl_max(t0.y, t1.y), cl_max(t0.z, t1.z));
    *tmin = cl_max(*tmin, cl_max(sm.x, cl_max(sm.y, sm.z)));
    *tmax = cl_min(*tmax, cl_min(bg.x, cl_min(bg.y, bg.z)));
    return *tmin < *tmax;
}

/* collision.cl:106-121 */
static int slab_hit(v3 vmin, v3 vmax, v3 o, v3 d)
{
    v3 a = V3((vmin.x - o.x) / d.x, (vmin.y - o.y) / d.y, (vmin.z - o.z) / d.z);
    v3 b = V3((vmax.x - o.x) / d.x, (vmax.y - o.y) / d.y, (vmax.z - o.z) / d.z);
    v3 rmin = V3(cl_min(a.x, b.x), cl_min(a.y, b.y), cl_min(a.z, b.z));
    v3 rmax = V3(cl_max(a.x, b.x), cl_max(a.y, b.y), cl_max(a.z, b.z));
    float mm = cl_min(cl_min(rmax.x, rmax.y), rmax.z);
    float xm = cl_max(cl_max(rmin.x, rmin.y), rmin.z);
    v3 minmax = V3(mm, mm, mm), maxmin = V3(xm, xm, xm);
    if (v3_dot(minmax, minmax) >= v3_dot(maxmin, maxmin)) return v3_dot(maxmin, maxmin) > 0.001f;
    return 0;
}

/* ========================================================================= */
/* 4. kd-tree traversal on the serialised blob (kdtree.cl:90-321)             */
/* ========================================================================= */
typedef struct { uint32_t tri; float t, u, v; } orc_hit;

typedef struct {
    uint64_t rays, nodes_visited, leaves_visited, tri_refs, tri_tests, pushes;
    uint64_t rays_over_16;      /* rays whose stack exceeded the reference's STACK_SIZE */
    uint32_t max_stack;
    uint32_t pad;
    uint64_t empty_leaves;      /* visited leaves that hold no triangle */
} orc_kd_stats;

#define ORC_STACK 128

/* Two builds of the traversal, chosen at load time: one for CPUs with an FMA unit (fmaf becomes one
 * instruction) and a portable one (fmaf is the C library's, same result). */
__attribute__((target_clones("fma", "default")))
static orc_hit kd_trace_one(const uint8_t *blob, const int32_t *idx4, const float *verts4,
                            const float *ray, v3 smin, v3 smax, orc_kd_stats *st)
{
    const uint64_t *words = (const uint64_t *)blob;
    v3 o = v3_ld(ray), d = v3_ld(ray + 4);
    /* get_elem (kdtree.cl:67-75) sums the masked components, which turns -0 into +0 */
    float oc[3] = {o.x + 0.0f + 0.0f, 0.0f + o.y + 0.0f, 0.0f + 0.0f + o.z};
    float dc[3] = {d.x + 0.0f + 0.0f, 0.0f + d.y + 0.0f, 0.0f + 0.0f + d.z};

    float t_hit = INFINITY, hu = 0, hv = 0;
    uint32_t tri_indx = 0;
    float scene_t_min = 0, scene_t_max = INFINITY;
    if (!slab_t(smin, smax, o, d, &scene_t_min, &scene_t_max)) scene_t_max = -INFINITY;
    float t_min, t_max;
    t_max = t_min = scene_t_min;

    struct { uint64_t node; float mn, mx; } stack[ORC_STACK];
    uint32_t sp = 0, max_sp = 0;
    int early = 0;
    int root_is_leaf = (blob[0] == 1);

    while (t_max < scene_t_max && !early) {
        uint64_t cur;
        if (sp == 0) { cur = 0; t_min = t_max; t_max = scene_t_max; }
        else { sp--; cur = stack[sp].node; t_min = stack[sp].mn; t_max = stack[sp].mx; }

        /* descend (kdtree.cl:229-260) */
        while (blob[cur * 8] != 1) {
            const uint8_t *n = blob + cur * 8;
            uint8_t k = n[1]; if (k > 2) k = 2;
            float b; memcpy(&b, n + 4, 4);
            uint64_t left = words[cur + 1], right = words[cur + 2];
            if (st) st->nodes_visited++;
            float t_split = (b - oc[k]) / dc[k];
            int left_close = (oc[k] < b) || (oc[k] == b && dc[k] <= 0);
            uint64_t first = left_close ? left : right, second = left_close ? right : left;
            int c1 = (t_split > t_max || t_split <= 0);
            int c2 = (t_split < t_min);
            cur = (c1 || !c2) ? first : second;
            if (!c1 && !c2) {
                if (sp >= ORC_STACK) { fprintf(stderr, "oracle: traversal stack overflow\n"); abort(); }
                stack[sp].node = second; stack[sp].mn = t_split; stack[sp].mx = t_max; sp++;
                if (sp > max_sp) max_sp = sp;
                if (st) st->pushes++;
                t_max = t_split;
            }
        }
        /* leaf (kdtree.cl:263-302) */
        uint32_t ntri; memcpy(&ntri, blob + cur * 8 + 4, 4);
        const uint32_t *offs = (const uint32_t *)(blob + (cur + 2) * 8);
        if (st) { st->leaves_visited++; st->tri_refs += ntri; st->empty_leaves += (ntri == 0); }
        for (uint32_t i = 0; i < ntri; i++) {
            uint32_t off = offs[i];
            v3 a = v3_ld(verts4 + 4 * (size_t)idx4[4 * (size_t)(off + 0)]);
            v3 b = v3_ld(verts4 + 4 * (size_t)idx4[4 * (size_t)(off + 1)]);
            v3 c = v3_ld(verts4 + 4 * (size_t)idx4[4 * (size_t)(off + 2)]);
            float t, u, v;
            if (st) st->tri_tests++;
            if (tri_hit(a, b, c, o, d, &t, &u, &v)) {
                if (t <= 0) continue;
                if (t < t_hit) {
                    t_hit = t; hu = u; hv = v; tri_indx = off;
                    if (t_hit < t_max) { early = 1; break; }
                }
            }
        }
        /* A tree that is a single leaf has no interior root to restart from;
           the reference reads the leaf header as a node (undefined).  Here the
           leaf is tested once and the walk ends. */
        if (root_is_leaf) break;
    }
    if (st) {
        st->rays++;
        if (max_sp > st->max_stack) st->max_stack = max_sp;
        if (max_sp > 16) st->rays_over_16++;
    }
    orc_hit h = {0, 0, 0, 0};
    if (!isinf(t_hit)) { h.tri = tri_indx; h.t = t_hit; h.u = hu; h.v = hv; }
    return h;
}

ORC_API void orc_kd_intersect(const uint8_t *blob, const int32_t *idx4, const float *verts4,
                              const float *rays, uint32_t n, const float *smin3, const float *smax3,
                              orc_hit *out, orc_kd_stats *stats, int nthreads)
{
    v3 smin = v3_ld(smin3), smax = v3_ld(smax3);
    orc_kd_stats total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        orc_kd_stats loc;
        memset(&loc, 0, sizeof loc);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1024)
#endif
        for (int64_t i = 0; i < (int64_t)n; i++)
            out[i] = kd_trace_one(blob, idx4, verts4, rays + 8 * i, smin, smax, stats ? &loc : NULL);
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.rays += loc.rays; total.nodes_visited += loc.nodes_visited;
            total.leaves_visited += loc.leaves_visited; total.tri_refs += loc.tri_refs;
            total.tri_tests += loc.tri_tests; total.pushes += loc.pushes;
            total.rays_over_16 += loc.rays_over_16; total.empty_leaves += loc.empty_leaves;
            if (loc.max_stack > total.max_stack) total.max_stack = loc.max_stack;
        }
    }
    if (stats) *stats = total;
}

/* deepest traversal stack each ray needs (the reference's private stack holds 16 entries, kdtree.cl:78,120) */
ORC_API void orc_kd_ray_stack(const uint8_t *blob, const int32_t *idx4, const float *verts4,
                              const float *rays, uint32_t n, const float *smin3, const float *smax3,
                              uint32_t *depth, int nthreads)
{
    v3 smin = v3_ld(smin3), smax = v3_ld(smax3);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < (int64_t)n; i++) {
        orc_kd_stats st;
        memset(&st, 0, sizeof st);
        kd_trace_one(blob, idx4, verts4, rays + 8 * i, smin, smax, &st);
        depth[i] = st.max_stack;
    }
}

/* per-ray work profile: steps[2*i] = interior nodes visited, steps[2*i+1] = triangles tested */
ORC_API void orc_kd_ray_steps(const uint8_t *blob, const int32_t *idx4, const float *verts4,
                              const float *rays, uint32_t n, const float *smin3, const float *smax3,
                              uint32_t *steps, int nthreads)
{
    v3 smin = v3_ld(smin3), smax = v3_ld(smax3);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < (int64_t)n; i++) {
        orc_kd_stats st;
        memset(&st, 0, sizeof st);
        (void)kd_trace_one(blob, idx4, verts4, rays + 8 * i, smin, smax, &st);
        steps[2 * i] = (uint32_t)st.nodes_visited;
        steps[2 * i + 1] = (uint32_t)st.tri_tests;
    }
}

/* exact nearest-hit brute force over every triangle (same Moeller-Trumbore);
   used by property tests, not a restatement of any reference function */
ORC_API void orc_brute_nearest(const int32_t *idx4, const float *verts4, uint32_t nidx,
                               const float *rays, uint32_t n, orc_hit *out, int nthreads)
{
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
#endif
    for (int64_t r = 0; r < (int64_t)n; r++) {
        v3 o = v3_ld(rays + 8 * r), d = v3_ld(rays + 8 * r + 4);
        orc_hit h = {0, 0, 0, 0};
        float best = INFINITY;
        for (uint32_t off = 0; off + 2 < nidx; off += 3) {
            v3 a = v3_ld(verts4 + 4 * (size_t)idx4[4 * (size_t)off]);
            v3 b = v3_ld(verts4 + 4 * (size_t)idx4[4 * (size_t)(off + 1)]);
            v3 c = v3_ld(verts4 + 4 * (size_t)idx4[4 * (size_t)(off + 2)]);
            float t, u, v;
            if (tri_hit(a, b, c, o, d, &t, &u, &v) && t > 0 && t < best) {
                best = t; h.tri = off; h.t = t; h.u = u; h.v = v;
            }
        }
        out[r] = h;
    }
}

/* ========================================================================= */
/* 5. RNG, colour packing (util.cl)                                           */
/* ========================================================================= */
/* util.cl:76-91 */
static float get_random(uint32_t *s0, uint32_t *s1)
{
    *s0 = 36969 * ((*s0) & 65535) + ((*s0) >> 16);
    *s1 = 18000 * ((*s1) & 65535) + ((*s1) >> 16);
    uint32_t ires = ((*s0) << 16) + (*s1);
    union { float f; uint32_t ui; } res;
    res.ui = (ires & 0x007fffff) | 0x40000000;
    return (res.f - 2.0f) / 2.0f;
}

/* util.cl:57-74; a NaN channel is defined as 0 here (undefined in OpenCL) */
static uint32_t get_colour(float x, float y, float z, float w)
{
    float c[4] = {x, y, z, w};
    uint32_t q[4];
    for (int i = 0; i < 4; i++) {
        float f = cl_min(cl_max(c[i], 0.0f), 1.0f);
        if (!(f == f)) f = 0.0f;
        q[i] = (uint32_t)(f * 255);
    }
    uint32_t out = 0;
    out |= 0xff000000 & q[3] << 24;
    out |= 0x00ff0000 & q[0] << 16;
    out |= 0x0000ff00 & q[1] << 8;
    out |= 0x000000ff & q[2];
    return out;
}

/* path.cl:11-23 */
static v3 cosine_hemisphere(float u1, float u2, v3 normal)
{
    const float r = sqrtf(u1);
    const float theta = 2.f * 3.14159274101257f * u2;
    v3 w = normal;
    v3 axis = fabsf(w.x) > 0.1f ? V3(0.0f, 1.0f, 0.0f) : V3(1.0f, 0.0f, 0.0f);
    v3 u = v3_normalize(v3_cross(axis, w));
    v3 v = v3_cross(w, u);
    float s, c;
    orc_sincos(theta, &s, &c);
    v3 a = v3_scale(v3_scale(u, c), r);
    v3 b = v3_scale(v3_scale(v, s), r);
    v3 cc = v3_scale(w, sqrtf(1.0f - u1));
    return v3_normalize(v3_add(v3_add(a, b), cc));
}

/* path.cl:421-426 */
ORC_API void orc_xorshift(uint32_t *data, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) { uint32_t d = data[i]; data[i] = ((d << 1) | (d >> 31)) + 1; }
}

/* ========================================================================= */
/* 6. split path tracer shading (path.cl:36-258)                              */
/* ========================================================================= */
/* Scene arrays in the reference's device layouts:
 *   idx4      int[4] per index   {v, vn, vt, mesh}            (loader.c:184-192)
 *   normals4  float[4] per normal
 *   meshes    128 bytes each, material_index at byte 104      (scene.h:18-29)
 *   materials 32 bytes each, colour float[4] @0, reflectivity @16 (scene.h:7-14)
 * progress: 12 floats/uints per pixel {sample_num, bounce_num, pad, pad, mask.xyz, pad, accum.xyz, pad}
 *           (path.cl:28-34, 48 bytes) */
typedef struct {
    const int32_t *idx4;
    const float *normals4;
    const uint8_t *meshes;
    const uint8_t *materials;
} orc_scene_view;

static void surface_at(const orc_scene_view *s, const orc_hit *h, v3 *normal, v3 *colour, float *refl)
{
    const int32_t *i1 = s->idx4 + 4 * (size_t)h->tri;
    const int32_t *i2 = i1 + 4, *i3 = i1 + 8;
    int32_t mat_index;
    memcpy(&mat_index, s->meshes + 128 * (size_t)i1[3] + 104, 4);
    const float *mat = (const float *)(s->materials + 32 * (size_t)mat_index);
    *colour = v3_ld(mat);
    *refl = mat[4];
    v3 n1 = v3_ld(s->normals4 + 4 * (size_t)i1[1]);
    v3 n2 = v3_ld(s->normals4 + 4 * (size_t)i2[1]);
    v3 n3 = v3_ld(s->normals4 + 4 * (size_t)i3[1]);
    *normal = v3_add(v3_add(v3_scale(n1, 1 - h->u - h->v), v3_scale(n2, h->u)), v3_scale(n3, h->v));
}

/* path.cl:36-113.  x,y of a pixel are its coordinates in the full frame:
   rows of a strip start at y0. */
ORC_API void orc_spath_init(float *out_tex, float *rays, float *origin_rays, const orc_hit *results,
                            orc_hit *source_results, uint32_t *progress, const int32_t *idx4,
                            const float *normals4, const uint8_t *meshes, const uint8_t *materials,
                            uint32_t w, uint32_t h, uint32_t y0, uint32_t random_value)
{
    orc_scene_view sv = {idx4, normals4, meshes, materials};
    const float sky[3] = {(float)0.16 * 2, (float)0.2 * 2, (float)0.2 * 2};
    for (uint32_t py = 0; py < h; py++)
        for (uint32_t px = 0; px < w; px++) {
            size_t off = (size_t)px + (size_t)py * w;
            orc_hit res = results[off];
            float *r = rays + 8 * off;
            memcpy(origin_rays + 8 * off, r, 32);
            source_results[off] = res;
            v3 mask = V3(1.0f, 1.0f, 1.0f);
            if (res.t == 0) {
                out_tex[4 * off + 0] += sky[0]; out_tex[4 * off + 1] += sky[1];
                out_tex[4 * off + 2] += sky[2]; out_tex[4 * off + 3] += 0.0f;
            }
            uint32_t seed1 = random_value * px, seed2 = random_value * (py + y0);
            for (int i = 0; i < 7; i++) get_random(&seed1, &seed2);
            float rand1 = get_random(&seed1, &seed2);
            float rand2 = get_random(&seed1, &seed2);

            v3 normal, colour; float refl;
            surface_at(&sv, &res, &normal, &colour, &refl);
            v3 o = v3_ld(r), d = v3_ld(r + 4);
            v3 pos = v3_add(o, v3_scale(d, res.t));
            mask = v3_mul(mask, colour);
            v3 dir = cosine_hemisphere(rand1, rand2, normal);
            v3 so = v3_add(pos, v3_scale(normal, 0.0001f));
            r[0] = so.x; r[1] = so.y; r[2] = so.z; r[3] = 0;
            r[4] = dir.x; r[5] = dir.y; r[6] = dir.z; r[7] = 0;
            uint32_t *p = progress + 12 * off;
            float *pf = (float *)p;
            p[0] = 0; p[1] = 0; p[2] = 0; p[3] = 0;
            pf[4] = mask.x; pf[5] = mask.y; pf[6] = mask.z; pf[7] = 0;
            pf[8] = 0; pf[9] = 0; pf[10] = 0; pf[11] = 0;
        }
}

/* path.cl:115-258, non-Windows branch (accum is added to out_tex as .zyx) */
ORC_API void orc_spath_trace(float *out_tex, float *rays, const float *origin_rays, const orc_hit *results,
                             const orc_hit *source_results, uint32_t *progress,
                             const uint32_t *random_buffer, const int32_t *idx4, const float *normals4,
                             const uint8_t *meshes, const uint8_t *materials, uint32_t n,
                             uint32_t random_value, int nthreads)
{
    orc_scene_view sv = {idx4, normals4, meshes, materials};
    const v3 sky = V3((float)0.16, (float)0.2, (float)0.2);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int64_t off = 0; off < (int64_t)n; off++) {
        uint32_t *p = progress + 12 * off;
        float *pf = (float *)p;
        float *r = rays + 8 * off;
        uint32_t sample_num = p[0], bounce_num = p[1];
        v3 mask = v3_ld(pf + 4), accum = v3_ld(pf + 8);
        if (sample_num == 2048) { memset(r, 0, 32); continue; }

        orc_hit res = results[off];
        v3 o = v3_ld(r), d = v3_ld(r + 4);
        v3 normal, colour; float refl;
        surface_at(&sv, &res, &normal, &colour, &refl);
        v3 pos = v3_add(o, v3_scale(d, res.t));

        uint32_t seed1 = random_buffer[off] * random_value, seed2 = random_buffer[off];
        float rand1 = get_random(&seed1, &seed2);
        float rand2 = get_random(&seed2, &seed1);

        v3 sdir = cosine_hemisphere(rand1, rand2, normal);
        v3 sorig = v3_add(pos, v3_scale(normal, 0.0001f));

        if (res.t == 0) {
            bounce_num = 4;
            accum = v3_add(accum, v3_mul(mask, sky));
        } else {
            float e = (float)(refl == 1.f);
            accum = v3_add(accum, v3_scale(v3_scale(mask, e), 2.f));
            mask = v3_mul(mask, colour);
            mask = v3_scale(mask, v3_dot(sdir, normal));
        }
        bounce_num++;
        if (bounce_num >= 4) {
            bounce_num = 0;
            sample_num++;
            float *ot = out_tex + 4 * off;
            ot[0] += accum.z; ot[1] += accum.y; ot[2] += accum.x; ot[3] += 1.0f;

            res = source_results[off];
            o = v3_ld(origin_rays + 8 * off); d = v3_ld(origin_rays + 8 * off + 4);
            mask = V3(1.0f, 1.0f, 1.0f);
            accum = V3(0, 0, 0);
            if (res.t == 0) { ot[0] += sky.x; ot[1] += sky.y; ot[2] += sky.z; ot[3] += 0.0f; }
            surface_at(&sv, &res, &normal, &colour, &refl);
            pos = v3_add(o, v3_scale(d, res.t));
            mask = v3_mul(mask, colour);
            if ((float)(refl == 1.)) accum = v3_add(accum, v3_scale(mask, 2));
            sdir = cosine_hemisphere(rand1, rand2, normal);
            sorig = v3_add(pos, v3_scale(normal, 0.0001f));
        }
        r[0] = sorig.x; r[1] = sorig.y; r[2] = sorig.z; r[3] = 0;
        r[4] = sdir.x; r[5] = sdir.y; r[6] = sdir.z; r[7] = 0;
        p[0] = sample_num; p[1] = bounce_num;
        pf[4] = mask.x; pf[5] = mask.y; pf[6] = mask.z;
        pf[8] = accum.x; pf[9] = accum.y; pf[10] = accum.z;
    }
}

/* path.cl:428-444 */
ORC_API void orc_resolve_avg(uint32_t *out, const float *out_tex, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        const float *d = out_tex + 4 * (size_t)i;
        if (d[3] == 0) out[i] = get_colour(0, 0, 0, 0);
        else out[i] = get_colour(d[0] / d[3], d[1] / d[3], d[2] / d[3], d[3] / d[3]);
    }
}

/* path.cl:447-458 */
ORC_API void orc_f_to_byte(uint32_t *out, const float *tex, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
        out[i] = get_colour(tex[4 * (size_t)i], tex[4 * (size_t)i + 1], tex[4 * (size_t)i + 2], tex[4 * (size_t)i + 3]);
}

/* path.cl:394-419 */
ORC_API void orc_f_buffer_average(float *out_tex, const float *fresh, uint32_t n, uint32_t sample)
{
    for (size_t i = 0; i < 4 * (size_t)n; i++) {
        float old = (out_tex[i] - 0.5f) * ((float)sample - 1.f);
        float nw = fresh[i] - 0.5f;
        out_tex[i] = (old + nw) / (float)sample;
    }
}

/* ========================================================================= */
/* 7. brute-force scene intersection (collision.cl:179-285,382-468)           */
/* ========================================================================= */
/* spheres: 32 bytes {pos float4, radius @16, material_index @20} (geom.h:49-57)
   planes : 48 bytes {pos float4, normal float4 @16, material_index @32} (geom.h:67-75) */
typedef struct {
    orc_scene_view sv;
    const float *verts4;
    const uint8_t *spheres; uint32_t nspheres;
    const uint8_t *planes; uint32_t nplanes;
    uint32_t nmeshes;
} orc_full_scene;

typedef struct { int did_hit; v3 normal, point; float dist; v3 colour; float refl; } orc_coll;

static void load_mat(const orc_full_scene *s, int32_t mi, v3 *colour, float *refl)
{
    const float *m = (const float *)(s->sv.materials + 32 * (size_t)mi);
    *colour = v3_ld(m); *refl = m[4];
}

/* collision.cl:411-454 */
static int collide_primitives(const orc_full_scene *s, v3 o, v3 d, orc_coll *res)
{
    float dist = ORC_FAR;
    res->did_hit = 0; res->dist = ORC_FAR;
    for (uint32_t i = 0; i < s->nspheres; i++) {
        const float *sp = (const float *)(s->spheres + 32 * (size_t)i);
        int32_t mi; memcpy(&mi, sp + 5, 4);
        v3 c = v3_ld(sp); float radius = sp[4];
        /* collision.cl:179-206 */
        v3 L = v3_sub(c, o);
        float b = v3_dot(d, L);
        float cc = v3_dot(L, L) - (radius * radius);
        float disc = b * b - cc;
        float t;
        if (disc < 0.0f) continue;
        t = b - sqrtf(disc);
        if (t < 0.0f) { t = b + sqrtf(disc); if (t < 0.0f) continue; }
        if (t < dist) {
            dist = t; res->did_hit = 1; res->dist = dist;
            res->point = v3_add(v3_scale(d, dist), o);
            res->normal = v3_normalize(v3_sub(res->point, c));
            load_mat(s, mi, &res->colour, &res->refl);
        }
    }
    for (uint32_t i = 0; i < s->nplanes; i++) {
        const float *pl = (const float *)(s->planes + 48 * (size_t)i);
        int32_t mi; memcpy(&mi, pl + 8, 4);
        v3 pp = v3_ld(pl), pn = v3_ld(pl + 4);
        /* collision.cl:214-230 */
        float denom = v3_dot(d, pn);
        if (denom < ORC_EPS) {
            v3 l = v3_sub(pp, o);
            float t = v3_dot(l, pn) / denom;
            if (t >= 0 && t < dist) {
                dist = t; res->did_hit = 1; res->dist = dist;
                res->point = v3_add(v3_scale(d, dist), o);
                res->normal = pn;
                load_mat(s, mi, &res->colour, &res->refl);
            }
        }
    }
    return dist != ORC_FAR;
}

/* collision.cl:240-285 and 382-409 */
static int collide_meshes(const orc_full_scene *s, v3 o, v3 d, orc_coll *res)
{
    float dist = ORC_FAR;
    res->did_hit = 0; res->dist = ORC_FAR;
    for (uint32_t m = 0; m < s->nmeshes; m++) {
        const uint8_t *mp = s->sv.meshes + 128 * (size_t)m;
        const float *mf = (const float *)mp;
        v3 bmax = v3_ld(mf + 16), bmin = v3_ld(mf + 20);
        int32_t index_offset, num_indices, mi;
        memcpy(&index_offset, mp + 96, 4); memcpy(&num_indices, mp + 100, 4); memcpy(&mi, mp + 104, 4);
        if (!slab_hit(bmin, bmax, o, d)) continue;
        float min_t = ORC_FAR;
        v3 normal = V3(0, 0, 0);
        for (int i = 0; i < num_indices / 3; i++) {
            const int32_t *i0 = s->sv.idx4 + 4 * (size_t)(i * 3 + index_offset);
            const int32_t *i1 = i0 + 4, *i2 = i0 + 8;
            v3 a = v3_ld(s->verts4 + 4 * (size_t)i0[0]);
            v3 b = v3_ld(s->verts4 + 4 * (size_t)i1[0]);
            v3 c = v3_ld(s->verts4 + 4 * (size_t)i2[0]);
            float t = 0, u = 0, v = 0;
            if (tri_hit(a, b, c, o, d, &t, &u, &v) && t < min_t && t > 0) {
                min_t = t;
                v3 n0 = v3_ld(s->sv.normals4 + 4 * (size_t)i0[1]);
                v3 n1 = v3_ld(s->sv.normals4 + 4 * (size_t)i1[1]);
                v3 n2 = v3_ld(s->sv.normals4 + 4 * (size_t)i2[1]);
                normal = v3_add(v3_add(v3_scale(n0, 1 - u - v), v3_scale(n1, u)), v3_scale(n2, v));
            }
        }
        if (min_t != ORC_FAR && min_t < dist) {
            dist = min_t;
            res->dist = dist; res->normal = normal;
            res->point = v3_add(v3_scale(d, dist), o);
            load_mat(s, mi, &res->colour, &res->refl);
            res->did_hit = 1;
        }
    }
    return res->did_hit;
}

/* collision.cl:456-468 */
static int collide_all(const orc_full_scene *s, v3 o, v3 d, orc_coll *res)
{
    float dist = ORC_FAR;
    if (collide_primitives(s, o, d, res)) dist = res->dist;
    orc_coll m;
    if (collide_meshes(s, o, d, &m))
        if (m.dist < dist) *res = m;
    return res->did_hit;
}

/* ray.cl:2-14; returns rgba */
static void ss_shade(const orc_full_scene *s, const orc_coll *c, float out[4])
{
    v3 l = v3_normalize(v3_sub(V3(1, 2, 0), c->point));
    v3 ro = v3_add(c->point, v3_scale(l, 0.00001f));
    orc_coll tmp;
    float vis = collide_all(s, ro, l, &tmp) ? 0.0f : 1.0f;
    out[0] = vis * c->colour.x; out[1] = vis * c->colour.y; out[2] = vis * c->colour.z; out[3] = vis * 1.0f;
}

static v3 reflect3(v3 i, v3 n) { return v3_sub(i, v3_scale(n, 2.f * v3_dot(i, n))); }  /* util.cl:105-108 */
static void mix4(float *a, const float *b, float t)      /* a = a + (b-a)*t */
{
    for (int i = 0; i < 4; i++) a[i] = a[i] + (b[i] - a[i]) * t;
}

/* ray.cl:17-105, non-Windows branch (colour packed as .zyxw) */
ORC_API void orc_cast_ray_test(uint32_t *out, const float *rays, uint32_t n, const int32_t *idx4,
                               const float *verts4, const float *normals4, const uint8_t *meshes,
                               uint32_t nmeshes, const uint8_t *materials, const uint8_t *spheres,
                               uint32_t nspheres, const uint8_t *planes, uint32_t nplanes, int nthreads)
{
    orc_full_scene s = {{idx4, normals4, meshes, materials}, verts4, spheres, nspheres, planes, nplanes, nmeshes};
    const float sky[4] = {(float)0.84, (float)0.87, (float)0.93, 0};
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < (int64_t)n; i++) {
        v3 o = v3_ld(rays + 8 * i), d = v3_ld(rays + 8 * i + 4);
        orc_coll result;
        if (!collide_all(&s, o, d, &result)) { out[i] = get_colour(sky[0], sky[1], sky[2], sky[3]); continue; }
        float colour[4];
        ss_shade(&s, &result, colour);
        v3 rdir[2]; orc_coll results[2]; float colours[2][4];
        int early = 2;
        for (int j = 0; j < 2; j++) {
            v3 ro;
            if (j == 0) { ro = v3_add(result.point, v3_scale(result.normal, 0.0001f)); rdir[0] = reflect3(d, result.normal); }
            else { ro = v3_add(results[0].point, v3_scale(results[0].normal, 0.0001f)); rdir[1] = reflect3(rdir[0], results[0].normal); }
            if (collide_all(&s, ro, rdir[j], &results[j])) ss_shade(&s, &results[j], colours[j]);
            else { memcpy(colours[j], sky, 16); early = j; break; }
        }
        for (int j = early - 1; j > -1; j--) {
            if (j == 1) mix4(colours[j], sky, results[j].refl);
            else mix4(colours[j], colours[j + 1], results[j].refl);
        }
        mix4(colour, colours[0], result.refl);
        out[i] = get_colour(colour[2], colour[1], colour[0], colour[3]);
    }
}

/* path.cl:260-366, non-Windows branch.  sin() of the seed hash uses the C
   library; a GPU result may differ in the few pixels where that rounding
   moves the integer seed. */
ORC_API void orc_path_trace(float *out_tex, const float *rays, uint32_t w, uint32_t h, const float *cam_pos3,
                            uint32_t magic, const int32_t *idx4, const float *verts4, const float *normals4,
                            const uint8_t *meshes, uint32_t nmeshes, const uint8_t *materials,
                            const uint8_t *spheres, uint32_t nspheres, const uint8_t *planes,
                            uint32_t nplanes, int nthreads)
{
    orc_full_scene s = {{idx4, normals4, meshes, materials}, verts4, spheres, nspheres, planes, nplanes, nmeshes};
    const v3 sky = V3((float)0.16, (float)0.2, (float)0.2);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
    for (int64_t y = 0; y < (int64_t)h; y++)
        for (uint32_t x = 0; x < w; x++) {
            size_t off = (size_t)x + (size_t)y * w;
            v3 d = v3_ld(rays + 8 * off + 4);
            v3 o = v3_ld(cam_pos3);
            union { float f; uint32_t ui; } res;
            res.f = (float)magic * 3.14159274101257f + x;
            uint32_t seed1 = res.ui + (int)(sinf((float)x) * 7.1f);
            res.f = (float)magic * 3.14159274101257f + y;
            uint32_t seed2 = (uint32_t)y + (int)(sinf((float)res.ui) * 7 * 3.f);
            float *ot = out_tex + 4 * off;
            orc_coll first;
            if (!collide_all(&s, o, d, &first)) { ot[0] = sky.x; ot[1] = sky.y; ot[2] = sky.z; ot[3] = 0; continue; }
            v3 fin = V3(0, 0, 0);
            for (int i = 0; i < 4; i++) {
                v3 accum = V3(0, 0, 0), mask = V3(1, 1, 1);
                float rand1 = get_random(&seed1, &seed2);
                float rand2 = get_random(&seed1, &seed2);
                v3 sdir = cosine_hemisphere(rand1, rand2, first.normal);
                v3 so = v3_add(first.point, v3_scale(first.normal, 0.0001f));
                mask = v3_mul(mask, first.colour);
                for (int b = 0; b < 4; b++) {
                    orc_coll r;
                    if (!collide_all(&s, so, sdir, &r)) { accum = v3_add(accum, v3_mul(mask, sky)); break; }
                    rand1 = get_random(&seed1, &seed2);
                    rand2 = get_random(&seed1, &seed2);
                    sdir = cosine_hemisphere(rand1, rand2, r.normal);
                    so = v3_add(r.point, v3_scale(r.normal, 0.0001f));
                    accum = v3_add(accum, v3_scale(v3_scale(mask, (float)(r.refl == 1.)), 2));
                    mask = v3_mul(mask, r.colour);
                    mask = v3_scale(mask, v3_dot(sdir, r.normal));
                }
                accum = V3(cl_min(cl_max(accum.x, 0.f), 1.f), cl_min(cl_max(accum.y, 0.f), 1.f),
                           cl_min(cl_max(accum.z, 0.f), 1.f));
                fin = v3_add(fin, v3_scale(accum, 1.f / 4));
            }
            ot[0] = fin.z; ot[1] = fin.y; ot[2] = fin.x; ot[3] = 1;
        }
}

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
