/*
 * kd_build.c -- SAH kd-tree construction, wire-format serialisation and
 * flattening for the GPU.
 *
 * The tree this builder produces is, node for node and triangle for triangle,
 * the one the reference's O(n log^2 n) sweep builder produces
 * (reference: src/kdtree.c:82-102 SAH, :213-303 plane search, :305-360
 * classification, :362-400 termination, src/geom.c:44-153 box helpers): the
 * candidate planes, the cost arithmetic (single precision, same operation
 * order), the strict "<" tie-breaking over axes and positions and the order of
 * triangles inside the child lists are all kept, because the order of
 * triangles in a leaf decides which triangle the traversal reports.
 *
 * How it differs in mechanics: per-triangle extents are computed once; the
 * sweep only needs each event's position and kind, so events are 64-bit keys
 * (order-preserving float bits << 2 | kind) sorted with an LSD radix sort or,
 * for small nodes, a merge sort; the three axes of a large node and the two
 * subtrees of every large node are independent OpenMP tasks; everything that
 * is allocated is released by kd_tree_free.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "kdtree.h"

#define EV_END    0u
#define EV_PLANAR 1u
#define EV_START  2u

#define SIDE_LEFT  1
#define SIDE_RIGHT 2

#define PLANAR_EPS 0.0000001f      /* reference: src/geom.c:3,143-148 */
#define SAH_EPS    0.001f          /* reference: src/kdtree.c:4 */

#define TASK_MIN_TRIS   2048       /* subtrees smaller than this are built inline */
#define AXIS_TASK_TRIS  65536      /* nodes larger than this search the three axes in parallel */

typedef struct build_ctx {
    const float (*verts)[4];
    const int (*idx)[4];
    float *ext;                    /* per triangle: min.xyz, max.xyz over its three corners */
    unsigned int max_recurse, leaf_threshold;
} build_ctx;

typedef struct plane_choice {
    float b, cost;
    uint8_t k, side;
    unsigned int NL, NR, NP;
} plane_choice;

/* ---- order-preserving float <-> key ------------------------------------- */
static inline uint32_t float_key(float f)
{
    uint32_t u;
    f = f + 0.0f;                  /* -0 and +0 compare equal in the sweep: give them one key */
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float key_float(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* ---- sorting of 34-bit keys --------------------------------------------- */
static void sort_small(uint64_t *a, uint64_t *tmp, size_t n)
{
    /* insertion-sorted runs of 8, then bottom-up merging */
    for (size_t s = 0; s < n; s += 8) {
        size_t e = s + 8 < n ? s + 8 : n;
        for (size_t i = s + 1; i < e; i++) {
            uint64_t v = a[i];
            size_t j = i;
            while (j > s && a[j - 1] > v) { a[j] = a[j - 1]; j--; }
            a[j] = v;
        }
    }
    uint64_t *src = a, *dst = tmp;
    for (size_t width = 8; width < n; width *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * width) {
            size_t mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
            size_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) dst[o++] = (src[j] < src[i]) ? src[j++] : src[i++];
            while (i < mid) dst[o++] = src[i++];
            while (j < hi) dst[o++] = src[j++];
        }
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(uint64_t));
}

static void sort_radix(uint64_t *a, uint64_t *tmp, size_t n)
{
    /* three passes: bits 0-11, 12-23, 24-33 */
    static const int shift[3] = {0, 12, 24};
    static const int bits[3] = {12, 12, 10};
    uint64_t *src = a, *dst = tmp;
    for (int p = 0; p < 3; p++) {
        size_t count[4096];
        const size_t buckets = (size_t)1 << bits[p];
        const uint64_t mask = buckets - 1;
        memset(count, 0, buckets * sizeof(size_t));
        for (size_t i = 0; i < n; i++) count[(src[i] >> shift[p]) & mask]++;
        size_t sum = 0;
        for (size_t b = 0; b < buckets; b++) { size_t c = count[b]; count[b] = sum; sum += c; }
        for (size_t i = 0; i < n; i++) dst[count[(src[i] >> shift[p]) & mask]++] = src[i];
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(uint64_t));
}

/* ---- SAH (reference: src/kdtree.c:66-102, src/geom.c:44-57,75-84) -------- */
static inline float box_area(const float mn[3], const float mx[3])
{
    float d0 = mx[0] - mn[0], d1 = mx[1] - mn[1], d2 = mx[2] - mn[2];
    return (d0 * d1 * 2 + d1 * d2 * 2 + d0 * d2 * 2);
}

static inline float sah_lambda(int NL, int NR, float PL, float PR)
{
    if ((NL <= SAH_EPS || NR <= SAH_EPS) && !(PL >= 1.0f - SAH_EPS || PR >= 1.0f - SAH_EPS)) return 0.8f;
    return 1.0f;
}

static inline float sah_c(float PL, float PR, uint32_t NL, uint32_t NR)
{
    return sah_lambda((int)NL, (int)NR, PL, PR) * (KDTREE_KT + KDTREE_KI * (PL * NL + PR * NR));
}

static inline float sah(int k, float rel, const AABB *V, float areaV, int NL, int NR, int NP, uint8_t *side)
{
    float lmax[3] = {V->max[0], V->max[1], V->max[2]};
    float rmin[3] = {V->min[0], V->min[1], V->min[2]};
    float split = V->min[k] + (V->max[k] - V->min[k]) * rel;
    lmax[k] = split;
    rmin[k] = split;
    float PL = box_area(V->min, lmax) / areaV;
    float PR = box_area(rmin, V->max) / areaV;
    if (PL >= 1 - SAH_EPS || PR >= 1 - SAH_EPS) { *side = 0; return 1000000000.0f; }
    float CPL = sah_c(PL, PR, (uint32_t)(NL + NP), (uint32_t)NR);
    float CPR = sah_c(PL, PR, (uint32_t)NL, (uint32_t)(NR + NP));
    if (CPL < CPR) { *side = SIDE_LEFT; return CPL; }
    *side = SIDE_RIGHT;
    return CPR;
}

/* Best plane on one axis (reference: src/kdtree.c:220-300 for one k).
   cost = INFINITY when the axis has no candidate. */
static plane_choice search_axis(const build_ctx *c, const AABB *V, const unsigned int *tris, unsigned int n, int k)
{
    plane_choice best;
    memset(&best, 0, sizeof best);
    best.cost = INFINITY;
    best.k = (uint8_t)k;

    uint64_t *ev = (uint64_t *)malloc(sizeof(uint64_t) * 4 * (size_t)n);
    uint64_t *tmp = ev + 2 * (size_t)n;
    size_t ne = 0;
    const float vmin = V->min[k], vmax = V->max[k];
    for (unsigned int i = 0; i < n; i++) {
        const float *e = c->ext + 6 * (size_t)(tris[i] / 3);
        /* triangle box starts at +-1e6 (src/geom.c:99-105), then is clipped to the voxel (:86-97) */
        float lo = e[k] < 1000000 ? e[k] : 1000000;
        float hi = e[3 + k] > -1000000 ? e[3 + k] : -1000000;
        if (lo < vmin) lo = vmin;
        if (hi > vmax) hi = vmax;
        if (hi - lo <= PLANAR_EPS) {
            ev[ne++] = ((uint64_t)float_key(lo) << 2) | EV_PLANAR;
        } else {
            ev[ne++] = ((uint64_t)float_key(lo) << 2) | EV_START;
            ev[ne++] = ((uint64_t)float_key(hi) << 2) | EV_END;
        }
    }
    if (ne < 4096) sort_small(ev, tmp, ne);
    else sort_radix(ev, tmp, ne);

    const float areaV = box_area(V->min, V->max);
    const float extent = vmax - vmin;
    int NL = 0, NP = 0, NR = (int)n;
    for (size_t i = 0; i < ne;) {
        const uint64_t pos = ev[i] >> 2;
        int Ps = 0, Pe = 0, Pp = 0;
        while (i < ne && (ev[i] >> 2) == pos && (ev[i] & 3u) == EV_END) { Pe++; i++; }
        while (i < ne && (ev[i] >> 2) == pos && (ev[i] & 3u) == EV_PLANAR) { Pp++; i++; }
        while (i < ne && (ev[i] >> 2) == pos && (ev[i] & 3u) == EV_START) { Ps++; i++; }
        NP = Pp; NR -= Pp; NR -= Pe;
        const float b = key_float((uint32_t)pos);
        uint8_t side;
        float cost = sah(k, (b - vmin) / extent, V, areaV, NL, NR, NP, &side);
        if (cost < best.cost) {
            best.cost = cost; best.b = b; best.side = side;
            best.NL = (unsigned)NL; best.NR = (unsigned)NR; best.NP = (unsigned)NP;
        }
        NL += Ps; NL += NP; NP = 0;
    }
    free(ev);
    return best;
}

static plane_choice search_plane(const build_ctx *c, const AABB *V, const unsigned int *tris, unsigned int n)
{
    plane_choice r[3];
    if (n >= AXIS_TASK_TRIS) {
#pragma omp taskgroup
        {
            for (int k = 0; k < 3; k++) {
#pragma omp task shared(r) firstprivate(k)
                r[k] = search_axis(c, V, tris, n, k);
            }
        }
    } else {
        for (int k = 0; k < 3; k++) r[k] = search_axis(c, V, tris, n, k);
    }
    /* axes are visited in order and a later one wins only when strictly cheaper */
    plane_choice best = r[0];
    for (int k = 1; k < 3; k++)
        if (r[k].cost < best.cost) best = r[k];
    return best;
}

static kd_tree_node *make_node(void)
{
    kd_tree_node *n = (kd_tree_node *)malloc(sizeof(kd_tree_node));
    n->k = 0;
    n->b = 0.5f;
    n->left = n->right = NULL;
    n->triangles.triangle_buffer = NULL;
    n->triangles.num_triangles = 0;
    return n;
}

static kd_tree_node *build_node(const build_ctx *c, AABB V, unsigned int *tris, unsigned int n, unsigned int depth)
{
    kd_tree_node *node = make_node();

    /* reference: src/kdtree.c:362-373 */
    int leaf = 0;
    for (int k = 0; k < 3; k++)
        if (V.max[k] - V.min[k] <= PLANAR_EPS) leaf = 1;
    if (depth == c->max_recurse) leaf = 1;
    if (n <= c->leaf_threshold) leaf = 1;

    plane_choice p;
    memset(&p, 0, sizeof p);
    if (!leaf) {
        p = search_plane(c, &V, tris, n);
        if (p.cost > KDTREE_KI * (float)n) leaf = 1;     /* reference: src/kdtree.c:392 */
    }
    if (leaf) {
        node->triangles.triangle_buffer = tris;
        node->triangles.num_triangles = n;
        return node;
    }
    node->k = p.k;
    node->b = p.b;

    /* reference: src/kdtree.c:305-360.  A corner strictly below the plane puts
       the triangle on the left, strictly above on the right, both -> both
       lists, neither -> the side the SAH preferred.  The per-triangle extents
       answer "any corner below/above" directly. */
    unsigned int *TL = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)n);
    unsigned int *TR = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)n);
    unsigned int nl = 0, nr = 0;
    const int k = p.k;
    for (unsigned int i = 0; i < n; i++) {
        const float *e = c->ext + 6 * (size_t)(tris[i] / 3);
        const int isL = e[k] < p.b, isR = e[3 + k] > p.b;
        if (isL && isR) { TR[nr++] = tris[i]; TL[nl++] = tris[i]; }
        else if (!isL && !isR) { if (p.side == SIDE_LEFT) TL[nl++] = tris[i]; else TR[nr++] = tris[i]; }
        else if (isL) TL[nl++] = tris[i];
        else TR[nr++] = tris[i];
    }
    free(tris);
    if (nl < n) { unsigned int *s = (unsigned int *)realloc(TL, sizeof(unsigned int) * (nl ? nl : 1)); if (s) TL = s; }
    if (nr < n) { unsigned int *s = (unsigned int *)realloc(TR, sizeof(unsigned int) * (nr ? nr : 1)); if (s) TR = s; }

    AABB VL = V, VR = V;             /* reference: src/geom.c:60-72 */
    VL.max[k] = p.b;
    VR.min[k] = p.b;

    if (n >= TASK_MIN_TRIS) {
#pragma omp task shared(node) firstprivate(VL, TL, nl, depth)
        node->left = build_node(c, VL, TL, nl, depth + 1);
#pragma omp task shared(node) firstprivate(VR, TR, nr, depth)
        node->right = build_node(c, VR, TR, nr, depth + 1);
#pragma omp taskwait
    } else {
        node->left = build_node(c, VL, TL, nl, depth + 1);
        node->right = build_node(c, VR, TR, nr, depth + 1);
    }
    return node;
}

/* counters the reference keeps while recursing (src/kdtree.c:380-403) */
static void count_nodes(kd_tree *t, const kd_tree_node *n)
{
    t->num_nodes_total++;
    if (!n->left) {
        t->num_leaves++;
        t->num_indices_total += n->triangles.num_triangles;
        t->num_tris_padded += n->triangles.num_triangles % 8;
        return;
    }
    t->num_traversal_nodes++;
    count_nodes(t, n->left);
    count_nodes(t, n->right);
}

/* ---- public API ---------------------------------------------------------- */
kd_tree *kd_tree_init(void)
{
    kd_tree *t = (kd_tree *)calloc(1, sizeof(kd_tree));
    t->k = 3;
    t->max_recurse = 50;             /* reference: src/kdtree.c:172-174 */
    t->tri_for_leaf_threshold = 2;
    return t;
}

kd_tree_node *kd_tree_node_init(void) { return make_node(); }

bool kd_tree_node_is_leaf(kd_tree_node *node) { return node->left == NULL || node->right == NULL; }

void kd_tree_construct(kd_tree *tree)
{
    scene *s = tree->s;
    if (!s) { fprintf(stderr, "ERROR: kd_tree_construct: tree has no scene\n"); exit(1); }
    if (s->num_mesh_indices == 0) {
        printf("WARNING: Skipping k-d tree Construction, num_mesh_indices is 0.\n");
        return;
    }
    if (s->num_mesh_indices % 3 != 0) { fprintf(stderr, "ERROR: index count is not a multiple of 3\n"); exit(1); }

    AABB V;
    memset(&V, 0, sizeof V);
    AABB_construct_from_vertices(&V, s->mesh_verts, s->num_mesh_verts);
    tree->bounds = V;

    const unsigned int nt = s->num_mesh_indices / 3;
    build_ctx c;
    c.verts = (const float (*)[4])s->mesh_verts;
    c.idx = (const int (*)[4])s->mesh_indices;
    c.max_recurse = tree->max_recurse;
    c.leaf_threshold = tree->tri_for_leaf_threshold;
    c.ext = (float *)malloc(sizeof(float) * 6 * (size_t)nt);

    unsigned int *tris = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)nt);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)nt; i++) {
        float *e = c.ext + 6 * (size_t)i;
        for (int k = 0; k < 3; k++) { e[k] = INFINITY; e[3 + k] = -INFINITY; }
        for (int j = 0; j < 3; j++) {
            const int vi = c.idx[3 * (size_t)i + j][0];
            if (vi < 0 || (unsigned)vi >= s->num_mesh_verts) {
                fprintf(stderr, "ERROR: kd_tree_construct: vertex index %d out of range\n", vi);
                exit(1);
            }
            const float *v = c.verts[vi];
            for (int k = 0; k < 3; k++) {
                if (v[k] < e[k]) e[k] = v[k];
                if (v[k] > e[3 + k]) e[3 + k] = v[k];
            }
        }
        tris[i] = (unsigned int)i * 3;
    }

    kd_tree_node *root = NULL;
#ifdef _OPENMP
    int threads = tree->build_threads ? (int)tree->build_threads : omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#pragma omp single
#endif
    root = build_node(&c, V, tris, nt, 0);
    free(c.ext);

    tree->root = root;
    tree->num_nodes_total = tree->num_tris_padded = tree->num_traversal_nodes = 0;
    tree->num_leaves = tree->num_indices_total = 0;
    count_nodes(tree, root);
}

/* Wire format of the reference (src/kdtree.c:469-513): pre-order, child links
   in 8-byte words, leaves carry their triangle offsets inline. */
static unsigned int write_wire(const kd_tree_node *n, char *buf, unsigned int off)
{
    if (!n->left) {
        _skd_tree_leaf_node l;
        memset(&l, 0, sizeof l);
        l.type = KDTREE_LEAF;
        l.num_triangles = n->triangles.num_triangles;
        memcpy(buf + off, &l, sizeof l);
        off += sizeof l;
        memcpy(buf + off, n->triangles.triangle_buffer, sizeof(unsigned int) * (size_t)n->triangles.num_triangles);
        off += sizeof(unsigned int) * n->triangles.num_triangles;
        if (n->triangles.num_triangles % 2) off += 4;
        return off;
    }
    _skd_tree_traversal_node t;
    memset(&t, 0, sizeof t);
    t.type = KDTREE_NODE;
    t.k = n->k;
    t.b = n->b;
    const unsigned int self = off;
    off += sizeof t;
    const unsigned int left_end = write_wire(n->left, buf, off);
    const unsigned int right_end = write_wire(n->right, buf, left_end);
    t.left_ind = off / 8;
    t.right_ind = left_end / 8;
    memcpy(buf + self, &t, sizeof t);
    return right_end;
}

void kd_tree_generate_serialized(kd_tree *tree)
{
    free(tree->buffer);
    tree->buffer = NULL;
    if (!tree->s || tree->s->num_mesh_indices == 0 || !tree->root) {
        tree->buffer_size = 0;
        tree->buffer = (char *)malloc(1);
        return;
    }
    /* size formula of the reference, including its over-estimate (src/kdtree.c:525-531) */
    unsigned int need = 0;
    need += tree->num_traversal_nodes * (unsigned int)sizeof(_skd_tree_traversal_node);
    need += tree->num_leaves * (unsigned int)sizeof(_skd_tree_leaf_node);
    need += (tree->num_indices_total + tree->num_tris_padded) * (unsigned int)sizeof(unsigned int);
    tree->buffer_size = need;
    tree->buffer = (char *)calloc(need ? need : 1, 1);
    write_wire(tree->root, tree->buffer, 0);
}

/* ---- flattening ---------------------------------------------------------- */
static void *must(void *p)           /* the arrays of a 10 M-triangle tree are gigabytes: say so when they cannot be had */
{
    if (!p) { fprintf(stderr, "ERROR: out of memory while flattening the kd-tree\n"); exit(1); }
    return p;
}

#define REF_LEAF        0x80000000u
#define REF_SLOT_BITS   26
#define REF_COUNT_MAX   31u

#define LEAF_SLOT_BYTES 40u
#define LEAF_LINE_BYTES 64u

typedef struct flat_ctx {
    kd_tree_flat *f;
    const float (*verts)[4];
    const int (*idx)[4];
    uint32_t next_node, next_line, cap_lines, depth_max;
    uint32_t hole, has_hole;        /* a single line left free by an alignment step, for the next one-line leaf */
    int plain_splits;
} flat_ctx;

/* A leaf's triangles, 40 bytes each, from the start of a 64-byte line of
 * leaf_tris (layout: rt_hip.h): a ray that tests the leaf touches the fewest
 * lines the triangles fit in.  A leaf of REF_COUNT_MAX or more triangles
 * begins with a header slot: a triangle of zero extent, which no ray hits,
 * carrying the leaf's slot count where the others carry their index offset. */
static uint32_t flatten_leaf(flat_ctx *fc, const kd_tree_node *n)
{
    const uint32_t cnt = n->triangles.num_triangles;
    const uint32_t is_long = cnt >= REF_COUNT_MAX;
    const uint32_t slots = cnt + is_long;
    const uint32_t lines = (slots * LEAF_SLOT_BYTES + LEAF_LINE_BYTES - 1u) / LEAF_LINE_BYTES;
    if (cnt == 0) return REF_LEAF;
    if ((uint64_t)fc->next_line + lines + 1u >= (1u << REF_SLOT_BITS)) {    /* keeps the array below 4 GiB: 32-bit byte offsets */
        fprintf(stderr, "ERROR: kd-tree too large for the flattened layout (more than %u lines of leaf triangles)\n",
                1u << REF_SLOT_BITS);
        exit(1);
    }
    if (fc->next_line + lines + 1u > fc->cap_lines) {
        while (fc->next_line + lines + 1u > fc->cap_lines) fc->cap_lines = fc->cap_lines ? fc->cap_lines * 2u : 1024u;
        fc->f->leaf_tris = (float *)must(realloc(fc->f->leaf_tris, (size_t)fc->cap_lines * LEAF_LINE_BYTES));
    }
    /* The memory system fetches 128 bytes for the price of 64: a leaf of an even number of lines starts on an
       even line, so that it lies in lines / 2 such pairs and not one more. */
    uint32_t start;
    if (lines == 1u && fc->has_hole) {
        start = fc->hole;
        fc->has_hole = 0;
    } else {
        if (!(lines & 1u) && (fc->next_line & 1u)) {
            memset(fc->f->leaf_tris + (size_t)fc->next_line * (LEAF_LINE_BYTES / 4u), 0, LEAF_LINE_BYTES);
            if (!fc->has_hole) { fc->hole = fc->next_line; fc->has_hole = 1; }
            fc->next_line++;
        }
        start = fc->next_line;
        fc->next_line += lines;
    }
    float *o = fc->f->leaf_tris + (size_t)start * (LEAF_LINE_BYTES / 4u);
    memset(o, 0, (size_t)lines * LEAF_LINE_BYTES);
    if (is_long) { memcpy(&o[9], &slots, 4); o += LEAF_SLOT_BYTES / 4u; }
    for (uint32_t i = 0; i < cnt; i++, o += LEAF_SLOT_BYTES / 4u) {
        const unsigned int off = n->triangles.triangle_buffer[i];
        const float *a = fc->verts[fc->idx[off][0]];
        const float *b = fc->verts[fc->idx[off + 1][0]];
        const float *c = fc->verts[fc->idx[off + 2][0]];
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
        o[3] = b[0] - a[0]; o[4] = b[1] - a[1]; o[5] = b[2] - a[2];
        o[6] = c[0] - a[0]; o[7] = c[1] - a[1]; o[8] = c[2] - a[2];
        memcpy(&o[9], &off, 4);
    }
    return REF_LEAF | ((is_long ? REF_COUNT_MAX : cnt) << REF_SLOT_BITS) | start;
}

/* Returns the reference of n (layout: rt_hip.h).  An interior node becomes the
   top of a two-level record that also holds its interior children: one fetch
   serves two descend steps.  Leaves are emitted in depth-first order. */
static uint32_t flatten_rec(flat_ctx *fc, const kd_tree_node *n, uint32_t depth)
{
    if (depth > fc->depth_max) fc->depth_max = depth;
    if (!n->left) return flatten_leaf(fc, n);
    {   /* rt_kd_view.plain_splits: 0, or 2^-16 <= |b| <= 2^39 (exponent fields 111..166) */
        uint32_t bits;
        memcpy(&bits, &n->b, 4);
        bits &= 0x7fffffffu;
        if (bits != 0 && (bits < 0x37800000u || bits > 0x53000000u)) fc->plain_splits = 0;
    }
    const uint32_t self = fc->next_node++;
    uint32_t rec[8] = {0};
    uint32_t packed = (uint32_t)(n->k > 2 ? 2 : n->k);
    memcpy(&rec[0], &n->b, 4);
    const kd_tree_node *child[2] = {n->left, n->right};
    for (int side = 0; side < 2; side++) {
        const kd_tree_node *c = child[side];
        if (depth + 1 > fc->depth_max) fc->depth_max = depth + 1;
        if (!c->left) {
            rec[4 + 2 * side] = flatten_leaf(fc, c);
            packed |= 64u << side;
        } else {
            memcpy(&rec[1 + side], &c->b, 4);
            packed |= (uint32_t)(c->k > 2 ? 2 : c->k) << (2 + 2 * side);
            rec[4 + 2 * side] = flatten_rec(fc, c->left, depth + 2);
            rec[5 + 2 * side] = flatten_rec(fc, c->right, depth + 2);
        }
    }
    rec[3] = packed;
    memcpy(fc->f->nodes + 8 * (size_t)self, rec, sizeof rec);
    return self << 2;
}

/* The record references among the eight words of a record: for each child of
 * the top that is interior, the two words of that child's children, where they
 * do not hold leaf descriptors. */
static int record_children(const uint32_t *rec, int word[4])
{
    int n = 0;
    for (int side = 0; side < 2; side++) {
        if (rec[3] & (64u << side)) continue;
        for (int j = 0; j < 2; j++)
            if (!(rec[4 + 2 * side + j] & REF_LEAF)) word[n++] = 4 + 2 * side + j;
    }
    return n;
}

/* Re-orders the records (depth-first order as flatten_rec leaves them) so that
 * a ray's way down the tree touches few 128-byte lines of the array, which is
 * what a scattered fetch costs whether it uses 32 of the bytes or all: every
 * line of four records holds connected pieces of the tree.  A piece is a
 * head, then records picked one at a time as the largest subtree among the
 * children of those already in it, until the line is full; the children left
 * over head pieces of their own.  A whole subtree of at most three records
 * goes into the slots an earlier line left free, if one has as many.  Lines
 * are laid out breadth-first, the root's first.  Slots that stay free hold
 * zeros and nothing refers to them.  Updates the references in
 * the records and *root_ref; returns the new array and its record count. */
static uint32_t *cluster_records(const uint32_t *nodes, uint32_t n, uint32_t *root_ref, uint32_t *num_out)
{
    uint32_t *size = (uint32_t *)must(malloc(sizeof(uint32_t) * n));
    uint32_t *place = (uint32_t *)must(malloc(sizeof(uint32_t) * n));
    uint32_t *queue = (uint32_t *)must(malloc(sizeof(uint32_t) * n));
    int word[4];
    for (uint32_t i = n; i-- > 0;) {            /* children follow their parent in depth-first order */
        const uint32_t *rec = nodes + 8 * (size_t)i;
        const int k = record_children(rec, word);
        size[i] = 1;
        for (int j = 0; j < k; j++) size[i] += size[rec[word[j]] >> 2];
    }
    uint32_t head = 0, tail = 0, lines = 0;
    uint32_t *open[4] = {NULL, NULL, NULL, NULL}, nopen[4] = {0, 0, 0, 0};   /* first free slot of lines with 1..3 free */
    for (int f = 1; f < 4; f++) open[f] = (uint32_t *)must(malloc(sizeof(uint32_t) * n));
    queue[tail++] = *root_ref >> 2;
    while (head < tail) {
        const uint32_t first = queue[head++];
        uint32_t frontier[16];
        int nf = 0, filled = 0, room = 4;
        uint32_t slot = 0;
        /* a piece of at most three records fits whole into what another line left free */
        for (int f = (int)size[first]; f < 4; f++)
            if (nopen[f]) { slot = open[f][--nopen[f]]; room = f; break; }
        if (room == 4) slot = 4u * lines++;
        uint32_t member = first;
        for (;;) {
            place[member] = slot++;
            const uint32_t *rec = nodes + 8 * (size_t)member;
            const int k = record_children(rec, word);
            for (int j = 0; j < k; j++) frontier[nf++] = rec[word[j]] >> 2;
            if (++filled == room || nf == 0) break;
            int best = 0;
            for (int j = 1; j < nf; j++)
                if (size[frontier[j]] > size[frontier[best]]) best = j;
            member = frontier[best];
            frontier[best] = frontier[--nf];
        }
        if (filled < room) open[room - filled][nopen[room - filled]++] = slot;
        for (int j = 0; j < nf; j++) queue[tail++] = frontier[j];
    }
    for (int f = 1; f < 4; f++) free(open[f]);
    if (4u * (uint64_t)lines >= (1u << 27)) {
        fprintf(stderr, "ERROR: kd-tree too large for the flattened layout (%u lines of node records)\n", lines);
        exit(1);
    }
    uint32_t *out = (uint32_t *)must(calloc((size_t)lines * 4u, sizeof(uint32_t) * 8));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t *rec = out + 8 * (size_t)place[i];
        memcpy(rec, nodes + 8 * (size_t)i, sizeof(uint32_t) * 8);
        const int k = record_children(rec, word);
        for (int j = 0; j < k; j++) rec[word[j]] = place[rec[word[j]] >> 2] << 2;
    }
    *root_ref = place[*root_ref >> 2] << 2;
    *num_out = lines * 4u;
    free(size); free(place); free(queue);
    return out;
}

void kd_tree_flatten(kd_tree *tree)
{
    kd_tree_flat *f = &tree->flat;
    free(f->nodes); free(f->leaf_tris);
    memset(f, 0, sizeof *f);
    if (!tree->root) return;
    /* also keeps both device arrays below 4 GiB: the kernels use 32-bit byte offsets */
    if (tree->num_traversal_nodes >= (1u << 27)) {
        fprintf(stderr, "ERROR: kd-tree too large for the flattened layout (%u interior nodes)\n",
                tree->num_traversal_nodes);
        exit(1);
    }
    /* every record holds at least its top node, so the interior-node count bounds the record count */
    f->nodes = (uint32_t *)must(malloc(sizeof(uint32_t) * 8 * (size_t)(tree->num_traversal_nodes ? tree->num_traversal_nodes : 1)));
    flat_ctx fc;
    fc.f = f;
    fc.verts = (const float (*)[4])tree->s->mesh_verts;
    fc.idx = (const int (*)[4])tree->s->mesh_indices;
    fc.next_node = fc.next_line = fc.cap_lines = fc.depth_max = fc.hole = fc.has_hole = 0;
    fc.plain_splits = 1;
    f->root_ref = flatten_rec(&fc, tree->root, 0);
    f->num_nodes = fc.next_node;
    f->num_leaf_lines = fc.next_line;
    f->leaf_tris = (float *)must(realloc(f->leaf_tris, (size_t)(fc.next_line ? fc.next_line : 1) * LEAF_LINE_BYTES));
    if (f->num_nodes) {
        uint32_t *clustered = cluster_records(f->nodes, f->num_nodes, &f->root_ref, &f->num_nodes);
        free(f->nodes);
        f->nodes = clustered;
    }
    f->max_depth = fc.depth_max;
    f->plain_splits = (uint32_t)fc.plain_splits;
    if (fc.depth_max >= RT_MAX_STACK) {        /* a ray pushes at most one entry per level */
        fprintf(stderr, "ERROR: kd-tree depth %u exceeds the traversal stack\n", fc.depth_max);
        exit(1);
    }
}

static void free_nodes(kd_tree_node *n)
{
    if (!n) return;
    free_nodes(n->left);
    free_nodes(n->right);
    free(n->triangles.triangle_buffer);
    free(n);
}

void kd_tree_free(kd_tree *tree)
{
    if (!tree) return;
    free_nodes(tree->root);
    free(tree->buffer);
    free(tree->flat.nodes);
    free(tree->flat.leaf_tris);
    free(tree);
}
