// HIP kernels of the path tracer for gfx950 (wave64) and their C-ABI launchers.
//
// Arithmetic contract: results must be bit-identical to a plain IEEE-754
// single-precision evaluation in source order, so this file is built with
// -ffp-contract=off and without any fast-math option; divisions and square
// roots are the correctly rounded ones.  The OpenCL built-ins whose rounding
// the specification leaves open are defined as ROCm's OpenCL library defines
// them on this GPU, so that hit records equal, bit for bit, what the
// reference's own kernels give when ROCm's compiler builds them (without
// contraction, with correctly rounded division) for the same device:
// dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)), cross with one fma per
// component, min/max = IEEE minNum/maxNum.
#include "rt_internal.hpp"

#define RT_INF __builtin_huge_valf()
#define RT_EPS 0.0000001f          /* EPSILON,   reference: src/kernels/util.cl:8 */
#define RT_FAR 100000000.0f        /* FAR_PLANE, reference: src/kernels/util.cl:9 */
#define RT_PI_F 3.14159274101257f  /* M_PI_F */

namespace {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(f3 a, f3 b)
{
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3(__builtin_fmaf(a.y, b.z, -(b.y * a.z)), __builtin_fmaf(a.z, b.x, -(b.z * a.x)),
               __builtin_fmaf(a.x, b.y, -(b.x * a.y)));
}
__device__ __forceinline__ float cl_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float cl_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ f3 normalize3(f3 a)
{
    float len = sqrtf(dot3(a, a));
    return mk3(a.x / len, a.y / len, a.z / len);
}
__device__ __forceinline__ float4 mk4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// sin and cos of an angle in [0, 2*pi]: three-constant Cody-Waite reduction to
// [-pi/4, pi/4] and degree-7/8 minimax polynomials, plain float operations
// only, so the host can reproduce the bits.
__device__ __forceinline__ void sincos_2pi(float theta, float &s, float &c)
{
    int q = (int)(theta * 0.636619772367581343f + 0.5f);
    float fq = (float)q;
    float r = theta - fq * 1.5703125f;
    r = r - fq * 4.837512969970703125e-4f;
    r = r - fq * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    switch (q & 3) {
    case 0: s = sp; c = cp; break;
    case 1: s = cp; c = -sp; break;
    case 2: s = -sp; c = -cp; break;
    default: s = -cp; c = sp; break;
    }
}

// get_random, reference: src/kernels/util.cl:76-91
__device__ __forceinline__ float get_random(uint32_t &s0, uint32_t &s1)
{
    s0 = 36969u * (s0 & 65535u) + (s0 >> 16);
    s1 = 18000u * (s1 & 65535u) + (s1 >> 16);
    uint32_t ires = (s0 << 16) + s1;
    float f = __uint_as_float((ires & 0x007fffffu) | 0x40000000u);
    return (f - 2.0f) / 2.0f;
}

// cosineSampleHemisphere, reference: src/kernels/path.cl:11-23
__device__ __forceinline__ f3 cosine_hemisphere(float u1, float u2, f3 w)
{
    const float r = sqrtf(u1);
    const float theta = 2.f * RT_PI_F * u2;
    f3 axis = fabsf(w.x) > 0.1f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    f3 u = normalize3(cross3(axis, w));
    f3 v = cross3(w, u);
    float s, c;
    sincos_2pi(theta, s, c);
    f3 a = (u * c) * r;
    f3 b = (v * s) * r;
    f3 cc = w * sqrtf(1.0f - u1);
    return normalize3((a + b) + cc);
}

// get_colour, reference: src/kernels/util.cl:57-74 (a NaN channel packs as 0)
__device__ __forceinline__ uint32_t quant8(float f)
{
    f = cl_min(cl_max(f, 0.0f), 1.0f);
    if (!(f == f)) f = 0.0f;
    return (uint32_t)(f * 255);
}
__device__ __forceinline__ uint32_t get_colour(float x, float y, float z, float w)
{
    uint32_t out = 0;
    out |= 0xff000000u & (quant8(w) << 24);
    out |= 0x00ff0000u & (quant8(x) << 16);
    out |= 0x0000ff00u & (quant8(y) << 8);
    out |= 0x000000ffu & quant8(z);
    return out;
}

// Moeller-Trumbore with back-face culling on precomputed edges,
// reference: src/kernels/collision.cl:139-172.  The reference returns at the
// first failed test; here every lane computes all of it and the tests are
// combined at the end (the values a failed test leaves behind are never
// used), which keeps the wave on one straight instruction stream.
__device__ __forceinline__ bool tri_hit_edges(f3 v0, f3 ab, f3 ac, f3 o, f3 d, float &t, float &u, float &v)
{
    const f3 pvec = cross3(d, ac);
    const float det = dot3(ab, pvec);
    const float inv = 1.f / det;
    const f3 tvec = o - v0;
    u = dot3(tvec, pvec) * inv;
    const f3 qvec = cross3(tvec, ab);
    v = dot3(d, qvec) * inv;
    t = dot3(ac, qvec) * inv;
    const bool miss = (det < RT_EPS) | (u < 0) | (u > 1) | (v < 0) | (u + v > 1);
    return !miss;
}

struct Hit { uint32_t tri; float t, u, v; };

// ---------------------------------------------------------------------------
// kd-tree traversal, one ray per lane, persistent waves.
//
// Decisions follow the reference walk (src/kernels/kdtree.cl:188-318) one for
// one: slab clip against the scene box; near/far choice by
// "o<b || (o==b && d<=0)"; near only when t_split > t_max or <= 0, far only
// when t_split < t_min, otherwise push the far child; in a leaf the triangles
// are tested in stored order and the walk stops at the first accepted hit that
// lies before the leaf's exit distance.  The reference's restart-from-root
// only ever fires on entry.
//
// What is different is the control structure.  The reference nests three
// data-dependent loops (pop / descend / leaf), which on a 64-wide wave leaves
// most lanes waiting for the longest one at every level.  Here every lane is a
// small state machine and the wave runs ONE flat loop: per trip, lanes that sit
// on an interior node fetch its record and descend (up to two levels, the
// record holds the node's interior children as well), then lanes that sit in a
// leaf fetch and test one triangle (and pop when the leaf is used up), and
// lanes whose ray has finished are handed new rays from a global work counter
// once enough of them are idle (ballot + prefix rank).  Loads are exec-masked:
// only the lanes that step pay for a cache lookup.
//
// Stack entries are {far child, t_split}: an entry's t_max always equals the
// t_split of the entry below it (scene_t_max for the bottom one), so it is not
// stored.  The first LDS_DEPTH entries of each lane live in LDS, laid out
// [level][lane] so that a wave's accesses never collide on a bank whatever the
// lanes' depths are; deeper entries (rare) spill to a per-lane private array.
// ---------------------------------------------------------------------------
#define RT_REF_LEAF 0x80000000u
#define RT_REF_SLOT_MASK 0x03ffffffu
#define RT_REF_COUNT(ref) (((ref) >> 26) & 31u)

struct TravScene {
    const uint4 *nodes;
    const float4 *leaf_tris;
    f3 smin, smax;
    uint32_t root_ref;
    uint32_t plain_splits;       // rt_kd_view.plain_splits
};

// Registers of one lane's walk (the ray itself, o and d, stays in plain locals
// of the kernel: selecting a component of a struct member by axis makes the
// compiler index the struct in memory, which drags all of it into scratch).
struct Walk {
    float t_min, t_max, scene_t_max, t_hit, hu, hv;
    uint32_t tri_indx, sp;
    uint32_t ref;            // where the lane stands: interior reference or leaf descriptor
    uint32_t li, lc;         // leaf progress: triangle li of lc (lc is settled at the leaf's first step)
};

__device__ __forceinline__ void walk_enter(Walk &w, uint32_t ref)
{
    w.ref = ref;
    w.li = 0;
}

// Start the walk of ray (o, d): slab clip against the scene box
// (getTBoundingBox, reference: src/kernels/collision.cl:71-86) and the first
// pass of the reference's outer loop.  false = the walk is over before it
// starts (the ray misses the box): the result is a miss.
// Also sets what the descend steps use to divide by the ray's direction: rd = the correctly rounded reciprocals of
// d's components, and `plain` = every component of the ray is in the range in which (b - o) * rd, corrected by two
// fused multiply-adds, is the correctly rounded (b - o) / d (see walk_descend) and the tree's splits are too.
__device__ __forceinline__ bool in_exponent_range(float x, uint32_t lo_bits, uint32_t hi_bits)
{
    return ((__float_as_uint(x) & 0x7fffffffu) - lo_bits) <= (hi_bits - lo_bits);
}

__device__ __forceinline__ bool walk_begin(const TravScene &P, Walk &w, f3 o, f3 d, f3 &rd, bool &plain)
{
    float scene_t_min = 0.f;
    w.scene_t_max = RT_INF;
    f3 inv = mk3(1 / d.x, 1 / d.y, 1 / d.z);
    rd = inv;
    // origin components: 0 or 2^-16 <= |o| <= 2^39; direction components: 2^-30 <= |d| <= 2^30
    const bool ox = (o.x == 0.f) | in_exponent_range(o.x, 0x37800000u, 0x53000000u);
    const bool oy = (o.y == 0.f) | in_exponent_range(o.y, 0x37800000u, 0x53000000u);
    const bool oz = (o.z == 0.f) | in_exponent_range(o.z, 0x37800000u, 0x53000000u);
    const bool dx = in_exponent_range(d.x, 0x30800000u, 0x4e800000u), dy = in_exponent_range(d.y, 0x30800000u, 0x4e800000u),
               dz = in_exponent_range(d.z, 0x30800000u, 0x4e800000u);
    plain = (int)ox & (int)oy & (int)oz & (int)dx & (int)dy & (int)dz & (int)(P.plain_splits != 0u);
    f3 t0 = (P.smin - o) * inv, t1 = (P.smax - o) * inv;
    f3 sm = mk3(cl_min(t0.x, t1.x), cl_min(t0.y, t1.y), cl_min(t0.z, t1.z));
    f3 bg = mk3(cl_max(t0.x, t1.x), cl_max(t0.y, t1.y), cl_max(t0.z, t1.z));
    scene_t_min = cl_max(scene_t_min, cl_max(sm.x, cl_max(sm.y, sm.z)));
    w.scene_t_max = cl_min(w.scene_t_max, cl_min(bg.x, cl_min(bg.y, bg.z)));
    if (!(scene_t_min < w.scene_t_max)) w.scene_t_max = -RT_INF;
    w.t_hit = RT_INF; w.hu = 0.f; w.hv = 0.f; w.tri_indx = 0; w.sp = 0;
    w.t_min = w.t_max = scene_t_min;
    if (!(w.t_max < w.scene_t_max)) return false;
    w.t_min = w.t_max; w.t_max = w.scene_t_max;     // restart at the root (kdtree.cl:208-216)
    walk_enter(w, P.root_ref);
    return true;
}

// A register with no defined contents and no instruction spent on it: the
// value of a load that only some lanes perform, for the lanes that do not.
__device__ __forceinline__ void undefined_words(uint4 &x)
{
    asm volatile("" : "=v"(x.x), "=v"(x.y), "=v"(x.z), "=v"(x.w));
}
__device__ __forceinline__ void undefined_words(uint2 &x)
{
    asm volatile("" : "=v"(x.x), "=v"(x.y));
}
template <class W4>
__device__ __forceinline__ void undefined_words(W4 &x)
{
    asm volatile("" : "=v"(x.x), "=v"(x.y), "=v"(x.z), "=v"(x.w));
}

// The walk steps below are written for a wave whose lanes are in different
// states: every lane runs the same straight instruction stream and `en` (or
// `in_leaf`) says whether the step is the lane's to take; state changes are
// selects, only loads and stack stores sit in branches.  Wrapping a step in
// "if (en)" instead costs a register copy per state word and nesting level on
// every trip of the kernels' loops.

// One descend step at an interior node with split plane (k, b) and children
// (left, right), reference: src/kernels/kdtree.cl:229-260.  get_elem (:67-75)
// adds the masked-out zeros, which turns -0 into +0.
template <int LDS_DEPTH>
__device__ __forceinline__ void walk_descend(Walk &w, f3 o, f3 d, f3 rd, bool plain, uint32_t k, float b, uint32_t left,
                                             uint32_t right, uint2 *__restrict__ col, uint2 *__restrict__ spill, bool en)
{
    // get_elem's "+ 0" (-0 becomes +0) changes nothing that is computed from the origin's component (b - (-0) and
    // b - (+0) are the same number, and so are the comparisons), and from the direction's only the sign of a
    // quotient by zero: it is applied where the division is, below
    const float ok = (k == 0) ? o.x : ((k == 1) ? o.y : o.z);
    const float dk = (k == 0) ? d.x : ((k == 1) ? d.y : d.z);
    // t_split = (b - ok) / dk, correctly rounded.  With y = RN(1/dk) kept per ray: q = RN(num * y) is within an ulp
    // of the quotient, rem = num - q * dk is exact in one fused multiply-add, and RN(q + rem * y) is the correctly
    // rounded quotient (Markstein's division by a correctly rounded reciprocal) as long as nothing underflows or
    // overflows on the way -- which the ranges behind `plain` rule out: dk in 2^-30..2^30; num = b - ok is 0 or in
    // 2^-40..2^40 because b and ok are each 0 or in 2^-16..2^39.  (A zero quotient may come out with the other
    // sign; t_split is only ever compared.)  Five instructions for the division's eleven.  Lanes that are not
    // `plain` (axis-parallel and null rays, origins within 1.5e-5 of a coordinate plane) divide.
    const float num = b - ok;
#ifdef RT_ALWAYS_DIVIDE      /* A/B builds: the division for every lane */
    const float t_split = num / (dk + 0.0f);
#else
    const float rk = (k == 0) ? rd.x : ((k == 1) ? rd.y : rd.z);
    const float q = num * rk;
    const float rem = __builtin_fmaf(-q, dk, num);
    float t_split = __builtin_fmaf(rem, rk, q);
    if (__any(en & !plain)) {
        const float exact = num / (dk + 0.0f);
        t_split = plain ? t_split : exact;
    }
#endif
    const bool left_close = (ok < b) | ((ok == b) & (dk <= 0));
    const uint32_t first = left_close ? left : right;
    const uint32_t second = left_close ? right : left;
    const bool c1 = (t_split > w.t_max) | (t_split <= 0);
    const bool c2 = (t_split < w.t_min);
    const bool push = en & !c1 & !c2;
    if (push) {
        const uint2 e = make_uint2(second, __float_as_uint(t_split));
        if (w.sp < (uint32_t)LDS_DEPTH) col[w.sp * 64] = e;
        else if (w.sp < RT_MAX_STACK) spill[w.sp - LDS_DEPTH] = e;
    }
    w.sp += push ? 1u : 0u;
    w.t_max = push ? t_split : w.t_max;
    const uint32_t to = (c1 | !c2) ? first : second;
    w.ref = en ? to : w.ref;
    // w.li needs no reset here: it is 0 whenever a lane stands on an interior node (walk_enter and every pop leave
    // it 0, only leaf steps count it up), so it is 0 on entering a leaf
}

// The descend steps a two-level record (words a, r; layout in rt_hip.h) allows the lanes with `en`, which stand
// on one of its nodes: from the record's top through one of its children when that child is interior too, so up
// to two levels; a lane that stands on a child position (popped from the stack) takes one.
template <int LDS_DEPTH>
__device__ __forceinline__ void walk_record_steps(Walk &w, f3 o, f3 d, f3 rd, bool plain, const uint4 &a, const uint4 &r,
                                                  uint2 *__restrict__ col, uint2 *__restrict__ spill, bool en)
{
    const uint32_t top = w.ref & ~3u;                    // reference of the record's top = record index * 4
    const uint32_t packed = a.w;
    const bool en1 = en & ((w.ref & 3u) == 0u);
    const uint32_t left = (packed & 64u) ? r.x : (top | 1u);
    const uint32_t right = (packed & 128u) ? r.z : (top | 2u);
    walk_descend<LDS_DEPTH>(w, o, d, rd, plain, packed & 3u, __uint_as_float(a.x), left, right, col, spill, en1);
    // now at a leaf or at a child position of this record (never at another record's top)
    const bool en2 = en & !(w.ref & RT_REF_LEAF) & ((w.ref & 3u) != 0u);
    const bool at_left = (w.ref & 3u) == 1u;
    walk_descend<LDS_DEPTH>(w, o, d, rd, plain, (packed >> (at_left ? 2 : 4)) & 3u, __uint_as_float(at_left ? a.y : a.z),
                            at_left ? r.x : r.z, at_left ? r.y : r.w, col, spill, en2);
}

// Lanes with `en` stand on an interior node: fetch its two-level record (32
// bytes) and descend.
template <int LDS_DEPTH>
__device__ __forceinline__ void walk_interior(const TravScene &P, Walk &w, f3 o, f3 d, f3 rd, bool plain,
                                              uint2 *__restrict__ col, uint2 *__restrict__ spill, bool en)
{
    uint4 a, r;
    undefined_words(a); undefined_words(r);
    if (en) {
        const uint4 *rec = (const uint4 *)((const char *)P.nodes + ((w.ref & ~3u) << 3));
        a = rec[0]; r = rec[1];
    }
    // both halves are fetched here, together: left alone, the compiler sinks the
    // words only the first step needs into that step, as a second round trip
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
    walk_record_steps<LDS_DEPTH>(w, o, d, rd, plain, a, r, col, spill, en);
}

// What leaving a leaf delivers (src/kernels/kdtree.cl:205-226): entry sp-1 of the stack -- the far child kept
// back and the distance from which it counts -- together with the t_max that goes with it, which is the t_split
// of entry sp-2 (scene_t_max at the bottom); with an empty stack, a restart at the root from t_max onwards.
struct PopData { uint2 e; uint32_t below; };

template <int LDS_DEPTH>
__device__ __forceinline__ PopData walk_peek(const TravScene &P, const Walk &w, const uint2 *__restrict__ col,
                                             const uint2 *__restrict__ spill, bool read)
{
    PopData p;
    p.e = make_uint2(P.root_ref, __float_as_uint(w.t_max));
    p.below = __float_as_uint(w.scene_t_max);
    if (read) {
        const uint32_t s = w.sp - 1u;
        if (s < (uint32_t)LDS_DEPTH) {
            p.e = col[s * 64];
            const uint32_t under = col[(s > 0u ? s - 1u : 0u) * 64].y;
            p.below = s > 0u ? under : p.below;
        } else {
            p.e = spill[s - LDS_DEPTH];
            p.below = (s - 1u < (uint32_t)LDS_DEPTH) ? col[(LDS_DEPTH - 1) * 64].y : spill[s - 1u - LDS_DEPTH].y;
        }
    }
    return p;
}

// One step of the lanes standing in a leaf (`in_leaf`), src/kernels/kdtree.cl:263-302:
// test the next triangle, and when the leaf is used up pop / restart / end.
// `has` says which of them have a triangle left, T0..T2 hold it; `pd` is what
// walk_peek read for the lanes that may leave the leaf (at least those with
// sp != 0 whose leaf this step uses up).  true when the lane's walk has ended
// (w.t_hit etc. hold the result, +inf for a miss); `moved` = the lane has left
// the leaf for the node pd names.
template <int LDS_DEPTH>
__device__ __forceinline__ bool walk_leaf_step(const TravScene &P, Walk &w, f3 o, f3 d, const float4 &T0, const float4 &T1,
                                               const float4 &T2, const PopData &pd, bool in_leaf, bool has, bool &moved)
{
    float t, u, v;
    const bool hit = tri_hit_edges(xyz(T0), xyz(T1), xyz(T2), o, d, t, u, v);
    const bool take = has & hit & !(t <= 0) & (t < w.t_hit);
    w.t_hit = take ? t : w.t_hit;
    w.hu = take ? u : w.hu;
    w.hv = take ? v : w.hv;
    w.tri_indx = take ? __float_as_uint(T0.w) : w.tri_indx;
    const bool finished = take & (t < w.t_max);
    w.li += has ? 1u : 0u;

    // leaf used up: the reference's outer loop test, then restart or pop
    const bool used_up = in_leaf & !finished & (w.li >= w.lc);
    const bool ends = used_up & (((P.root_ref & RT_REF_LEAF) != 0u) | !(w.t_max < w.scene_t_max));
    const bool next = used_up & !ends;
    const bool pop = next & (w.sp != 0u);
    w.sp -= pop ? 1u : 0u;
    w.t_min = next ? __uint_as_float(pd.e.y) : w.t_min;
    w.t_max = next ? __uint_as_float(pd.below) : w.t_max;
    w.ref = next ? pd.e.x : w.ref;
    w.li = next ? 0u : w.li;
    moved = next;
    return finished | ends;
}

template <int LDS_DEPTH>
__device__ __forceinline__ bool walk_leaf(const TravScene &P, Walk &w, f3 o, f3 d, const float4 &T0, const float4 &T1,
                                          const float4 &T2, uint2 *__restrict__ col, uint2 *__restrict__ spill,
                                          bool in_leaf, bool has)
{
    // the stack is read only by the lanes that pop: what they are is known once the triangle has been tested,
    // so the test's decisions are taken twice here (the compiler folds them)
    float t, u, v;
    const bool hit = tri_hit_edges(xyz(T0), xyz(T1), xyz(T2), o, d, t, u, v);
    const bool take = has & hit & !(t <= 0) & (t < w.t_hit);
    const bool finished = take & (t < w.t_max);
    const bool used_up = in_leaf & !finished & (w.li + (has ? 1u : 0u) >= w.lc);
    const bool ends = used_up & (((P.root_ref & RT_REF_LEAF) != 0u) | !(w.t_max < w.scene_t_max));
    const PopData pd = walk_peek<LDS_DEPTH>(P, w, col, spill, used_up & !ends & (w.sp != 0u));
    bool moved;
    return walk_leaf_step<LDS_DEPTH>(P, w, o, d, T0, T1, T2, pd, in_leaf, has, moved);
}

// A slot of leaf_tris (40 bytes, 8-byte aligned: rt_hip.h) as it is loaded, and its parts as the test takes them.
struct __attribute__((aligned(8))) SlotWords4 { uint32_t x, y, z, w; };
struct LeafSlot { SlotWords4 a, b; uint2 c; };

__device__ __forceinline__ LeafSlot leaf_slot_load(const TravScene &P, const Walk &w)
{
    const char *p = (const char *)P.leaf_tris + (((w.ref & RT_REF_SLOT_MASK) << 6) + w.li * 40u);
    LeafSlot t;
    t.a = *(const SlotWords4 *)p;
    t.b = *(const SlotWords4 *)(p + 16);
    t.c = *(const uint2 *)(p + 32);
    return t;
}
__device__ __forceinline__ float4 leaf_slot_v0(const LeafSlot &t)      // .w = the triangle's index offset
{
    return make_float4(__uint_as_float(t.a.x), __uint_as_float(t.a.y), __uint_as_float(t.a.z), __uint_as_float(t.c.y));
}
__device__ __forceinline__ float4 leaf_slot_e1(const LeafSlot &t)
{
    return make_float4(__uint_as_float(t.a.w), __uint_as_float(t.b.x), __uint_as_float(t.b.y), 0.f);
}
__device__ __forceinline__ float4 leaf_slot_e2(const LeafSlot &t)
{
    return make_float4(__uint_as_float(t.b.z), __uint_as_float(t.b.w), __uint_as_float(t.c.x), 0.f);
}

// walk_leaf with the triangle fetch: lanes that have a triangle left load it
// (40 bytes), the others pay for no cache lookup.  The slot count of the leaf
// is settled at its first step: the descriptor's triangle count, or for a
// long leaf the one its header slot carries.
template <int LDS_DEPTH>
__device__ __forceinline__ bool walk_leaf_load(const TravScene &P, Walk &w, f3 o, f3 d, uint2 *__restrict__ col,
                                               uint2 *__restrict__ spill, bool in_leaf)
{
    const bool first = w.li == 0u;
    const uint32_t count = RT_REF_COUNT(w.ref);
    w.lc = first ? count : w.lc;
    const bool has = in_leaf & (w.li < w.lc);
    LeafSlot t;
    undefined_words(t.a); undefined_words(t.b); undefined_words(t.c);
    if (has) t = leaf_slot_load(P, w);
    w.lc = (first & (count == 31u)) ? t.c.y : w.lc;
    const float4 T0 = leaf_slot_v0(t), T1 = leaf_slot_e1(t), T2 = leaf_slot_e2(t);
    return walk_leaf<LDS_DEPTH>(P, w, o, d, T0, T1, T2, col, spill, in_leaf, has);
}

__device__ __forceinline__ Hit walk_result(const Walk &w)
{
    Hit h;
    if (!isinf(w.t_hit)) { h.tri = w.tri_indx; h.t = w.t_hit; h.u = w.hu; h.v = w.hv; }
    else { h.tri = 0; h.t = 0.f; h.u = 0.f; h.v = 0.f; }
    return h;
}

// Wave-uniform hand-out of consecutive work items: one atomicAdd claims CHUNK
// items for the wave; `take` gives the calling lanes (those with want == true)
// consecutive indices by prefix rank.  Returns the lane's index or ~0u.
struct WorkQueue {
    uint32_t next = 0, end = 0;
    bool exhausted = false;
};

template <int CHUNK>
__device__ __forceinline__ uint32_t queue_take(WorkQueue &q, bool want, uint32_t n, uint32_t *__restrict__ counter)
{
    const unsigned long long mask = __ballot(want);
    const uint32_t nwant = (uint32_t)__popcll(mask);
    if (q.exhausted || nwant == 0) return ~0u;
    if (q.next >= q.end) {
        uint32_t base = 0;
        if (threadIdx.x == 0) base = atomicAdd(counter, (uint32_t)CHUNK);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n) { q.exhausted = true; return ~0u; }
        q.next = base;
        q.end = (n - base < (uint32_t)CHUNK) ? n : base + (uint32_t)CHUNK;
    }
    const uint32_t avail = q.end - q.next;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const uint32_t mine = (want && rank < avail) ? q.next + rank : ~0u;
    q.next += (nwant < avail) ? nwant : avail;
    return mine;
}

// kdtree_intersection: every ray of ray_o/ray_d[0..n) -> hits[0..n).
template <int LDS_DEPTH, int REFILL_MIN, int CHUNK>
__global__ __launch_bounds__(64) void k_kd_intersect(TravScene P, const float4 *__restrict__ ray_o,
                                                     const float4 *__restrict__ ray_d, Hit *__restrict__ hits,
                                                     uint32_t n, uint32_t *__restrict__ counter)
{
    __shared__ uint2 stack[LDS_DEPTH * 64];
    uint2 spill[RT_MAX_STACK - LDS_DEPTH];
    uint2 *const col = stack + threadIdx.x;

    WorkQueue q;
    Walk w = {};
    f3 o = mk3(0.f, 0.f, 0.f), d = o, rd = o;
    bool walking = false, plain = false;
    uint32_t ridx = 0;

    for (;;) {
        // ---- hand new rays to idle lanes -----------------------------------
        const uint32_t nidle = (uint32_t)__popcll(__ballot(!walking));
        if (!q.exhausted && nidle >= (uint32_t)REFILL_MIN) {
            const uint32_t mine = queue_take<CHUNK>(q, !walking, n, counter);
            if (mine != ~0u) {
                ridx = mine;
                o = xyz(ray_o[ridx]);
                d = xyz(ray_d[ridx]);
                walking = walk_begin(P, w, o, d, rd, plain);
                if (!walking) hits[ridx] = walk_result(w);
            }
        }
        if (!__any(walking)) {
            if (q.exhausted) break;
            continue;
        }
        // one record (up to two descend steps) for the lanes on interior nodes,
        // then one triangle for the lanes in leaves
        const bool in_node = walking & !(w.ref & RT_REF_LEAF);
        if (__any(in_node)) walk_interior<LDS_DEPTH>(P, w, o, d, rd, plain, col, spill, in_node);
        const bool in_leaf = walking & ((w.ref & RT_REF_LEAF) != 0u);
        if (__any(in_leaf)) {
            if (walk_leaf_load<LDS_DEPTH>(P, w, o, d, col, spill, in_leaf)) {
                hits[ridx] = walk_result(w);
                walking = false;
            }
        }
    }
}

// Surface data of a hit: interpolated (unnormalised) normal, material colour
// and reflectivity.  reference: src/kernels/path.cl:160-171
struct Surface { f3 normal, colour; float refl; };

__device__ __forceinline__ Surface surface_at(const float4 *__restrict__ tri_shade, const Hit &h)
{
    const float4 *rec = tri_shade + 4 * (size_t)(h.tri / 3u);
    const float4 a = rec[0], b = rec[1], c = rec[2], m = rec[3];
    Surface s;
    s.normal = (xyz(a) * (1 - h.u - h.v) + xyz(b) * h.u) + xyz(c) * h.v;
    s.colour = mk3(a.w, b.w, c.w);
    s.refl = m.x;
    return s;
}

// One step of segmented_path_trace for one pixel (src/kernels/path.cl:142-256,
// non-Windows branch: the finished sample is added to out_tex as .zyx).
__device__ __forceinline__ void spath_step(const float4 *__restrict__ tri_shade, Hit res, f3 o, f3 d,
                                           uint32_t rnd, uint32_t random_value, uint32_t i,
                                           float4 *__restrict__ out_tex, const float4 *__restrict__ origin_o,
                                           const float4 *__restrict__ origin_d, const Hit *__restrict__ source_hits,
                                           float4 &pa, float4 &pb, f3 &new_o, f3 &new_d)
{
    const f3 sky = mk3((float)0.16, (float)0.2, (float)0.2);
    uint32_t sample_num = __float_as_uint(pa.w), bounce_num = __float_as_uint(pb.w);
    f3 mask = xyz(pa), accum = xyz(pb);

    Surface sf = surface_at(tri_shade, res);
    f3 pos = o + d * res.t;

    uint32_t seed1 = rnd * random_value, seed2 = rnd;
    float rand1 = get_random(seed1, seed2);
    float rand2 = get_random(seed2, seed1);

    // The reference samples a bounce direction at the hit, updates mask with it,
    // and, when this bounce ends the sample (a miss or the fourth bounce),
    // resets mask and samples again at the primary hit with the same two random
    // numbers (path.cl:176-250).  In that case nothing of the first direction
    // survives, so the hemisphere sample is evaluated once, at whichever surface
    // the next ray leaves from.
    const bool finish = (res.t == 0) || (bounce_num + 1 >= 4);
    if (res.t == 0) {
        accum = accum + mask * sky;
    } else {
        float e = (sf.refl == 1.f) ? 1.0f : 0.0f;
        accum = accum + (mask * e) * 2.f;
    }

    Surface from = sf;
    if (finish) {
        bounce_num = 0;
        sample_num++;
        float4 ot = out_tex[i];
        ot.x += accum.z; ot.y += accum.y; ot.z += accum.x; ot.w += 1.0f;

        res = source_hits[i];
        o = xyz(origin_o[i]); d = xyz(origin_d[i]);
        mask = mk3(1.0f, 1.0f, 1.0f);
        accum = mk3(0.f, 0.f, 0.f);
        if (res.t == 0) { ot.x += sky.x; ot.y += sky.y; ot.z += sky.z; ot.w += 0.0f; }
        out_tex[i] = ot;

        from = surface_at(tri_shade, res);
        pos = o + d * res.t;
        mask = mask * from.colour;
        if (from.refl == 1.f) accum = accum + mask * 2.f;
    }
    const f3 sdir = cosine_hemisphere(rand1, rand2, from.normal);
    const f3 sorig = pos + from.normal * 0.0001f;
    if (!finish) {
        mask = mask * sf.colour;
        mask = mask * dot3(sdir, sf.normal);
        bounce_num++;
    }
    new_o = sorig; new_d = sdir;
    pa = mk4(mask, __uint_as_float(sample_num));
    pb = mk4(accum, __uint_as_float(bounce_num));
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// generate_rays, reference: src/kernels/ray.cl:109-133.  The reference's 0.5
// literals are doubles, so the camera-plane coordinates go through double.
__device__ __forceinline__ uint32_t frame_row(const rt_rows &rows, uint32_t local_row)
{
    return rows.band ? rows.first + (local_row / rows.band) * rows.stride + (local_row % rows.band)
                     : rows.first + local_row;
}

__global__ void k_generate_rays(float4 *__restrict__ ray_o, float4 *__restrict__ ray_d, uint32_t w, uint32_t h,
                                rt_rows rows, float tan_half)
{
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= w * h) return;
    uint32_t x = id % w, y = frame_row(rows, id / w);
    const uint32_t full_h = rows.full_height;
    float aspect = w / (float)full_h;
    float cam_x = (float)((2 * (((float)x + 0.5) / w) - 1) * tan_half * aspect);
    float cam_y = (float)((1 - 2 * (((float)y + 0.5) / full_h)) * tan_half);
    f3 dir = normalize3(mk3(cam_x, cam_y, -1.0f));
    ray_o[id] = make_float4(0.f, 0.f, 0.f, 0.f);
    ray_d[id] = mk4(dir, 0.f);
}

// segmented_path_trace_init, reference: src/kernels/path.cl:36-113.  The
// reference leaves sample_num / bounce_num of the record it stores
// uninitialised; they are written as 0 here.
__global__ void k_spath_init(const float4 *__restrict__ tri_shade, float4 *__restrict__ out_tex,
                             float4 *__restrict__ ray_o, float4 *__restrict__ ray_d,
                             float4 *__restrict__ origin_o, float4 *__restrict__ origin_d,
                             const Hit *__restrict__ hits, Hit *__restrict__ source_hits,
                             float4 *__restrict__ prog_a, float4 *__restrict__ prog_b, uint32_t w, uint32_t h,
                             rt_rows rows, uint32_t random_value)
{
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= w * h) return;
    uint32_t x = id % w, y = frame_row(rows, id / w);

    Hit res = hits[id];
    float4 ro = ray_o[id], rd = ray_d[id];
    origin_o[id] = ro; origin_d[id] = rd;
    source_hits[id] = res;

    if (res.t == 0) {
        float4 ot = out_tex[id];
        ot.x += (float)0.16 * 2; ot.y += (float)0.2 * 2; ot.z += (float)0.2 * 2; ot.w += 0.0f;
        out_tex[id] = ot;
    }
    uint32_t seed1 = random_value * x, seed2 = random_value * y;
    for (int i = 0; i < 7; i++) get_random(seed1, seed2);
    float rand1 = get_random(seed1, seed2);
    float rand2 = get_random(seed1, seed2);

    Surface sf = surface_at(tri_shade, res);
    f3 pos = xyz(ro) + xyz(rd) * res.t;
    f3 mask = mk3(1.0f, 1.0f, 1.0f) * sf.colour;
    f3 dir = cosine_hemisphere(rand1, rand2, sf.normal);
    f3 so = pos + sf.normal * 0.0001f;
    ray_o[id] = mk4(so, 0.f);
    ray_d[id] = mk4(dir, 0.f);
    prog_a[id] = mk4(mask, __uint_as_float(0u));
    prog_b[id] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
}

// segmented_path_trace, reference: src/kernels/path.cl:115-258
// STEP_RNG folds xorshift_batch (src/kernels/path.cl:421-426) into this launch:
// the pixel's RNG word is advanced and stored before it is used, exactly as if
// that kernel had run just before.
template <bool STEP_RNG>
__global__ void k_spath_trace(const float4 *__restrict__ tri_shade, float4 *__restrict__ out_tex,
                              float4 *__restrict__ ray_o, float4 *__restrict__ ray_d,
                              const float4 *__restrict__ origin_o, const float4 *__restrict__ origin_d,
                              const Hit *__restrict__ hits, const Hit *__restrict__ source_hits,
                              float4 *__restrict__ prog_a, float4 *__restrict__ prog_b,
                              uint32_t *__restrict__ random, uint32_t n, uint32_t random_value)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t rnd = random[i];
    if (STEP_RNG) {
        rnd = ((rnd << 1) | (rnd >> 31)) + 1;
        random[i] = rnd;
    }
    float4 pa = prog_a[i], pb = prog_b[i];
    if (__float_as_uint(pa.w) == 2048u) {       // reference: path.cl:145-152
        ray_o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ray_d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    f3 no, nd;
    spath_step(tri_shade, hits[i], xyz(ray_o[i]), xyz(ray_d[i]), rnd, random_value, i, out_tex, origin_o,
               origin_d, source_hits, pa, pb, no, nd);
    ray_o[i] = mk4(no, 0.f);
    ray_d[i] = mk4(nd, 0.f);
    prog_a[i] = pa;
    prog_b[i] = pb;
}

// ---------------------------------------------------------------------------
// Several render iterations in one launch.
//
// With one launch per bounce the whole GPU waits, every iteration, for the
// single longest ray of the frame (bounce rays: 24 steps on average, more than
// a thousand for the worst), and each wave is busy for only about half of the
// launch.  Pixels never interact, so here a lane keeps ONE pixel for `iters`
// consecutive iterations -- walk its current ray, advance its RNG word, shade,
// walk the new ray, ... exactly the per-pixel sequence of
// xorshift_batch -> kdtree_intersection -> segmented_path_trace -- with the
// path state in registers, and only then takes the next pixel from the work
// queue.  A long ray now delays nothing but its own pixel.  rv[k] is the
// random_value of the k-th iteration of the batch (the reference draws one per
// launch, src/spath_raytracer.c:321).
//
// Shading is much heavier per call than a walk step, so lanes whose walk has
// ended wait until SHADE_MIN of them can be served together (or nothing else is
// left to do); the same service pass hands pixels to free lanes.
//
// Walk slots of a trip: NI descend slots (one two-level record each), then NT
// leaf slots (one triangle each).  In each slot only the lanes standing on that
// kind of node load and step (exec-masked, so idle lanes cost the cache
// nothing); a lane that reaches its leaf in a descend slot tests its first
// triangle in the same trip.
//
// WAVES is the occupancy the register allocation is held to (waves per SIMD):
// waves spend more than half of their time waiting for gathers, so a seventh
// wave is worth more than registers; at eight the path state spills.
// ---------------------------------------------------------------------------

struct SpathDev {
    const float4 *tri_shade;
    float4 *out_tex, *ray_o, *ray_d;
    const float4 *origin_o, *origin_d;
    Hit *hits;
    const Hit *source_hits;
    float4 *prog_a, *prog_b;
    uint32_t *random;
    uint32_t *cost;             // out: trips of the loop each pixel was walked in, in this launch
    const uint32_t *order;      // in: pixel to serve as j-th work item (nullptr = j)
    float4 *snap;               // out: copies of out_tex at the launch's resolve points, [slot][pixel]
};

// A lane's progress through its pixel's iterations, in one register: iterations finished in this launch (bits
// 10..0), snapshot slots filled (15..11), iterations left until the next snapshot (31..16).  Finishing an
// iteration is one add.
#define RT_IT_MASK 0x7ffu
#define RT_IT_SLOT_SHIFT 11
#define RT_IT_LEFT_SHIFT 16
#define RT_IT_FINISHED_ONE 0xffff0001u   /* +1 finished, -1 left */
#define RT_HANDOVER_GROUPS_UPTO 40u      /* launches of at most this many iterations hand pixels over in groups */
#define RT_HANDOVER_WAIT 16u             /* trips a finished pixel is held at most while its group fills */

template <int LDS_DEPTH, int SHADE_MIN, int CHUNK, int NI, int NT, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_spath_run(TravScene P, SpathDev B, uint32_t n, uint32_t iters,
                                                  const uint32_t *__restrict__ rv, uint32_t *__restrict__ counter,
                                                  uint32_t snap_first, uint32_t snap_every, uint32_t handover_min)
{
    __shared__ uint2 stack[LDS_DEPTH * 64];
    uint2 spill[RT_MAX_STACK - LDS_DEPTH];
    uint2 *const col = stack + threadIdx.x;
    // With several slots of a kind per trip some find no lane to serve: those
    // are jumped over.  With one slot of each kind that hardly happens, and the
    // jump would cost a register copy per state word on every trip.
    constexpr bool SKIP_EMPTY = (NI + NT) > 2;

    enum { FREE = 0, WALK = 1, DONE = 2, LAST = 3 };     // LAST: the pixel's last iteration is done, its state not stored yet
    // A lane that has finished its pixel keeps it until handover_min lanes of the wave have, but for handover_wait
    // trips at most (or until nothing else is left to do): a hand-over makes the whole wave wait three times for
    // memory (the stores' acknowledgements, which the next loads wait for as well; the pixel-order lookup; the new
    // pixel's state), and that is paid once per group.  The launcher asks for groups in short launches only, in which a
    // wave hands a pixel over every few trips; in long ones the lanes finish too far apart and would only idle.
    constexpr uint32_t handover_wait = RT_HANDOVER_WAIT;
    uint32_t held = 0;           // trips since a lane of the wave began to hold a finished pixel (wave-uniform)
    WorkQueue q;
    Walk w = {};
    f3 o = mk3(0.f, 0.f, 0.f), d = o, rd = o;
    bool plain = false;
    uint32_t mode = FREE, pix = 0, it = 0, rnd = 0, steps = 0;
#ifdef RT_WAVE_TIMES
    // diagnostic build: per wave, the constant 100 MHz clock (s_memrealtime) and the shader clock counter (s_memtime)
    // at its start, when it finds the work counter dry, and at its exit; their ratio is the shader clock it ran at
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        B.cost[n + 8 * blockIdx.x] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        B.cost[n + 8 * blockIdx.x + 4] = (uint32_t)__builtin_amdgcn_s_memtime();
    }
    uint32_t exhausted_at = 0, exhausted_clk = 0, trips = 0;
#endif
    float4 pa = make_float4(0, 0, 0, 0), pb = pa;

    for (;;) {
        const uint32_t ndone = (uint32_t)__popcll(__ballot(mode == DONE));
        const uint32_t nfree = (uint32_t)__popcll(__ballot(mode == FREE));
        const uint32_t nlast = (uint32_t)__popcll(__ballot(mode == LAST));
        const uint32_t nwalk = 64u - ndone - nfree - nlast;
        held = nlast ? held + 1u : 0u;
        const bool hand_over = nlast >= handover_min || held > handover_wait;
        const uint32_t waiting = ndone + (hand_over ? nlast : 0u) + (q.exhausted ? 0u : nfree);
        if (waiting >= (uint32_t)SHADE_MIN || (nwalk == 0 && (waiting > 0 || nlast > 0))) {
            bool fresh = false;          // lane holds a new ray in w.o / w.d
            // (a) finish the iteration of lanes whose walk has ended
            if (mode == DONE) {
                const Hit h = walk_result(w);
                rnd = ((rnd << 1) | (rnd >> 31)) + 1;                 // xorshift_batch
                if (__float_as_uint(pa.w) == 2048u) {                 // reference: path.cl:145-152
                    o = mk3(0.f, 0.f, 0.f); d = mk3(0.f, 0.f, 0.f);
                } else {
                    f3 no, nd;
                    spath_step(B.tri_shade, h, o, d, rnd, rv[it & RT_IT_MASK], pix, B.out_tex, B.origin_o, B.origin_d,
                               B.source_hits, pa, pb, no, nd);
                    o = no; d = nd;
                }
                it += RT_IT_FINISHED_ONE;
                if ((it >> RT_IT_LEFT_SHIFT) == 0u) {
                    // the frame is resolved after this iteration: keep the pixel's accumulator as it stands now
                    // (snap_first-th iteration of the launch, then every snap_every-th)
                    B.snap[(size_t)((it >> RT_IT_SLOT_SHIFT) & 31u) * n + pix] = B.out_tex[pix];
                    it += (snap_every << RT_IT_LEFT_SHIFT) + (1u << RT_IT_SLOT_SHIFT);
                }
#ifdef RT_EXP_FAKE_HANDOVER
                // diagnostic build: every RT_EXP_FAKE_HANDOVER-th iteration of a pixel the lane goes through the memory
                // operations of handing the pixel back and taking one (its own again), to price a hand-over by itself;
                // RT_EXP_FAKE_PARTS: 1 = the stores, 2 = the pixel-order lookup ahead of the loads, 4 = the state loads
#ifndef RT_EXP_FAKE_PARTS
#define RT_EXP_FAKE_PARTS 7
#endif
                if ((it & RT_IT_MASK) != iters && (it & RT_IT_MASK) % RT_EXP_FAKE_HANDOVER == 0u) {
                    if (RT_EXP_FAKE_PARTS & 1) {
                        B.ray_o[pix] = mk4(o, 0.f);
                        B.ray_d[pix] = mk4(d, 0.f);
                        B.prog_a[pix] = pa;
                        B.prog_b[pix] = pb;
                        B.random[pix] = rnd;
                        B.hits[pix] = h;
                        B.cost[pix] = steps;
                    }
                    uint32_t again = pix;
                    if (RT_EXP_FAKE_PARTS & 2) {
                        again = B.order ? B.order[pix] : pix;
                        asm volatile("" : "+v"(again));
                        again = (again == 0xfffffff0u) ? 0u : pix;       // always pix, but only known once the load is back
                    }
                    if (RT_EXP_FAKE_PARTS & 4) {
                        float4 l0 = B.ray_o[again], l1 = B.ray_d[again], l2 = B.prog_a[again], l3 = B.prog_b[again];
                        uint32_t l4 = B.random[again];
                        asm volatile("" :: "v"(l0.x), "v"(l1.x), "v"(l2.x), "v"(l3.x), "v"(l4));
                    }
                }
#endif
                if ((it & RT_IT_MASK) == iters) mode = LAST;
                else fresh = true;
            }
            // (a') lanes that are through with their pixel hand it back, in groups: the buffers are left as the last
            // per-iteration launches would leave them (the walk's registers still hold the last hit)
            {
                const uint32_t through = (uint32_t)__popcll(__ballot(mode == LAST));
                const bool others_busy = __any((mode == WALK) | fresh);
                if (through >= handover_min || held > handover_wait || (through > 0 && !others_busy)) {
                    if (mode == LAST) {
                        B.ray_o[pix] = mk4(o, 0.f);
                        B.ray_d[pix] = mk4(d, 0.f);
                        B.prog_a[pix] = pa;
                        B.prog_b[pix] = pb;
                        B.random[pix] = rnd;
                        B.hits[pix] = walk_result(w);
                        B.cost[pix] = steps;
                        mode = FREE;
                    }
                }
            }
            // (b) free lanes take the next pixels
            const uint32_t mine = queue_take<CHUNK>(q, mode == FREE, n, counter);
            if (mine != ~0u) {
                pix = B.order ? B.order[mine] : mine;
                steps = 0;
                o = xyz(B.ray_o[pix]);
                d = xyz(B.ray_d[pix]);
                pa = B.prog_a[pix];
                pb = B.prog_b[pix];
                rnd = B.random[pix];
                it = snap_first << RT_IT_LEFT_SHIFT;
                fresh = true;
            }
            // (c) start the walk of every new ray
            if (fresh) mode = walk_begin(P, w, o, d, rd, plain) ? WALK : DONE;
        }
#ifdef RT_WAVE_TIMES
        trips++;
        // every 2^RT_WAVE_MILESTONE_SHIFT-th trip of the first 1024 waves: both clocks (32 milestones per wave, behind the per-wave records)
#ifndef RT_WAVE_MILESTONE_SHIFT
#define RT_WAVE_MILESTONE_SHIFT 10
#endif
        if ((trips & ((1u << RT_WAVE_MILESTONE_SHIFT) - 1u)) == 0u && (trips >> RT_WAVE_MILESTONE_SHIFT) <= 32u && blockIdx.x < 1024 && threadIdx.x == 0) {
            uint32_t *m = B.cost + n + 65536 + 64 * blockIdx.x + 2 * ((trips >> RT_WAVE_MILESTONE_SHIFT) - 1u);
            m[0] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            m[1] = (uint32_t)__builtin_amdgcn_s_memtime();
        }
        if (q.exhausted && !exhausted_at) {
            exhausted_at = (uint32_t)__builtin_amdgcn_s_memrealtime();
            exhausted_clk = (uint32_t)__builtin_amdgcn_s_memtime();
        }
#endif
        if (!__any(mode == WALK)) {
            if (q.exhausted && !__any((mode == DONE) | (mode == LAST))) break;
            continue;
        }
        // up to NI records (two descend steps each), then up to NT triangles: a
        // lane that reaches its leaf within the trip tests its triangles in the
        // same trip.  A pixel's cost is the number of trips it was walked in.
        steps += (mode == WALK) ? 1u : 0u;
#ifdef RT_EXP_DUMMY_VMEM
        // diagnostic build: RT_EXP_DUMMY_VMEM extra loads per trip from lines that stay in the vector L1 -- what does
        // one more vector-memory instruction per trip cost?  RT_EXP_DUMMY_LANES: 64 = every lane, 32 = the lower half of
        // the wave, 33 = every other lane, 1 = one lane; RT_EXP_DUMMY_WIDTH: 1 = a dword, 4 = 16 bytes per lane.
#ifndef RT_EXP_DUMMY_LANES
#define RT_EXP_DUMMY_LANES 64
#endif
#ifndef RT_EXP_DUMMY_WIDTH
#define RT_EXP_DUMMY_WIDTH 1
#endif
        if (RT_EXP_DUMMY_LANES == 64 || (RT_EXP_DUMMY_LANES == 32 && threadIdx.x < 32) ||
            (RT_EXP_DUMMY_LANES == 33 && (threadIdx.x & 1)) || (RT_EXP_DUMMY_LANES == 1 && threadIdx.x == 0)) {
#pragma unroll
            for (int k = 0; k < RT_EXP_DUMMY_VMEM; k++) {
                if (RT_EXP_DUMMY_WIDTH == 4) {
                    typedef float native4 __attribute__((ext_vector_type(4)));
                    typedef const __attribute__((address_space(1))) native4 *global_float4_ptr;
                    global_float4_ptr p = (global_float4_ptr)((const native4 *)P.nodes + threadIdx.x + 64 * k);
                    asm volatile("" : "+v"(p));
                    const native4 x = *p;
                    asm volatile("" :: "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
                } else {
                    typedef const __attribute__((address_space(1))) float *global_float_ptr;
                    global_float_ptr p = (global_float_ptr)((const float *)P.nodes + threadIdx.x + 64 * k);
                    asm volatile("" : "+v"(p));      // a new address as far as the compiler knows: a load per trip
                    const float x = *p;
                    asm volatile("" :: "v"(x));
                }
            }
        }
#endif
#pragma unroll
        for (int r = 0; r < NI; r++) {
            const bool in_node = (mode == WALK) & !(w.ref & RT_REF_LEAF);
            if (!SKIP_EMPTY || __any(in_node)) {
                walk_interior<LDS_DEPTH>(P, w, o, d, rd, plain, col, spill, in_node);
            }
        }
#pragma unroll
        for (int r = 0; r < NT; r++) {
            const bool in_leaf = (mode == WALK) & ((w.ref & RT_REF_LEAF) != 0u);
            if (!SKIP_EMPTY || __any(in_leaf)) {
                const bool fin = walk_leaf_load<LDS_DEPTH>(P, w, o, d, col, spill, in_leaf);
                mode = fin ? (uint32_t)DONE : mode;
            }
        }
    }
#ifdef RT_WAVE_TIMES
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        B.cost[n + 8 * blockIdx.x + 1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        B.cost[n + 8 * blockIdx.x + 2] = exhausted_at;
        B.cost[n + 8 * blockIdx.x + 3] = trips;
        B.cost[n + 8 * blockIdx.x + 5] = (uint32_t)__builtin_amdgcn_s_memtime();
        B.cost[n + 8 * blockIdx.x + 6] = exhausted_clk;
    }
#endif
}

// ---------------------------------------------------------------------------
// Heaviest pixels first.  A pixel's walk cost is stable from batch to batch
// (its paths all start at the same primary hit), and a batch ends when the
// last lane has finished its last pixel, so the next batch serves pixels in
// order of decreasing cost of the previous one: 256-bucket counting sort,
// buckets by cost >> shift, in three small launches (histogram, scan of the
// buckets from heaviest to lightest, scatter).  Contention on the global
// bucket counters is kept low by counting inside each workgroup in LDS first.
// ---------------------------------------------------------------------------
#define RT_COST_BUCKETS 256

__device__ __forceinline__ uint32_t cost_bucket(uint32_t c, uint32_t shift)
{
    const uint32_t b = c >> shift;
    return (RT_COST_BUCKETS - 1) - (b < RT_COST_BUCKETS - 1 ? b : RT_COST_BUCKETS - 1);   // heaviest -> bucket 0
}

// The sort visits the pixels tile by tile (8x8 pixels, tiles in rows) when the
// frame allows it, so pixels of one bucket that end up next to each other in
// the order are also neighbours on screen: a wave's rays then share nodes and
// triangles.  tiles_per_row == 0: plain index order.
__device__ __forceinline__ uint32_t tile_pixel(uint32_t i, uint32_t tiles_per_row)
{
    if (tiles_per_row == 0) return i;
    const uint32_t tile = i >> 6, in = i & 63u;
    const uint32_t tx = tile % tiles_per_row, ty = tile / tiles_per_row;
    return (ty * 8u + (in >> 3)) * (tiles_per_row * 8u) + tx * 8u + (in & 7u);
}

__global__ __launch_bounds__(256) void k_cost_hist(const uint32_t *__restrict__ cost, uint32_t n, uint32_t shift,
                                                    uint32_t tiles_per_row, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t local[RT_COST_BUCKETS];
    local[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x, e = min(n, (blockIdx.x + 1u) * 1024u); i < e; i += 256u)
        atomicAdd(&local[cost_bucket(cost[tile_pixel(i, tiles_per_row)], shift)], 1u);
    __syncthreads();
    if (local[threadIdx.x]) atomicAdd(&hist[threadIdx.x], local[threadIdx.x]);
}

// hist[b] -> first output position of bucket b (exclusive prefix sum); one workgroup.
__global__ __launch_bounds__(256) void k_cost_scan(uint32_t *__restrict__ hist)
{
    __shared__ uint32_t v[RT_COST_BUCKETS];
    v[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int b = 0; b < RT_COST_BUCKETS; b++) { const uint32_t c = v[b]; v[b] = sum; sum += c; }
    }
    __syncthreads();
    hist[threadIdx.x] = v[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_cost_scatter(const uint32_t *__restrict__ cost, uint32_t n, uint32_t shift,
                                                       uint32_t tiles_per_row, uint32_t *__restrict__ next,
                                                       uint32_t *__restrict__ order)
{
    __shared__ uint32_t local[RT_COST_BUCKETS], base[RT_COST_BUCKETS];
    local[threadIdx.x] = 0;
    __syncthreads();
    uint32_t rank[4], bucket[4];
    const uint32_t first = blockIdx.x * 1024u + threadIdx.x;
    for (int k = 0; k < 4; k++) {
        const uint32_t i = first + 256u * k;
        bucket[k] = (i < n) ? cost_bucket(cost[tile_pixel(i, tiles_per_row)], shift) : 0u;
        rank[k] = (i < n) ? atomicAdd(&local[bucket[k]], 1u) : 0u;
    }
    __syncthreads();
    base[threadIdx.x] = local[threadIdx.x] ? atomicAdd(&next[threadIdx.x], local[threadIdx.x]) : 0u;
    __syncthreads();
    for (int k = 0; k < 4; k++) {
        const uint32_t i = first + 256u * k;
        if (i < n) order[base[bucket[k]] + rank[k]] = tile_pixel(i, tiles_per_row);
    }
}

// Tile-granular order: whole 8x8 screen tiles sorted by their mean cost, each
// tile's 64 pixels kept together.  A wave's 64-item chunk is then exactly one
// tile, whose rays start next to each other and diverge least; ordering single
// pixels balances the launch slightly better but costs 10 % in divergence.
__global__ __launch_bounds__(256) void k_tile_cost(const uint32_t *__restrict__ cost, uint32_t n_tiles, uint32_t tiles_per_row,
                                                    uint32_t *__restrict__ tcost)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < 64u; k++) sum += cost[tile_pixel(t * 64u + k, tiles_per_row)];
    tcost[t] = sum >> 6;
}

__global__ __launch_bounds__(256) void k_tile_expand(const uint32_t *__restrict__ torder, uint32_t n, uint32_t tiles_per_row,
                                                      uint32_t *__restrict__ order)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    order[i] = tile_pixel(torder[i >> 6] * 64u + (i & 63u), tiles_per_row);
}

// xorshift_batch, reference: src/kernels/path.cl:421-426
__global__ void k_xorshift(uint32_t *__restrict__ data, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d = data[i];
    data[i] = ((d << 1) | (d >> 31)) + 1;
}

// f_buffer_to_byte_buffer_avg, reference: src/kernels/path.cl:428-444
__global__ void k_resolve_avg(uint32_t *__restrict__ out, const float4 *__restrict__ out_tex, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 d = out_tex[i];
    out[i] = (d.w == 0) ? get_colour(0.f, 0.f, 0.f, 0.f) : get_colour(d.x / d.w, d.y / d.w, d.z / d.w, d.w / d.w);
}

// kdtree_ray_draw, reference: src/kernels/kdtree.cl:323-341 -- the rays' origins as colours (debug probe)
__global__ void k_kd_ray_draw(uint32_t *__restrict__ out, const float4 *__restrict__ ray_o, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 o = ray_o[i];
    out[i] = get_colour((o.x + 1) / 2, (o.y + 1) / 2, (o.z + 1) / 2, 1.0f);
}

// kdtree_test_draw, reference: src/kernels/kdtree.cl:344-384 -- the interpolated normal at each hit as a colour,
// black where the ray missed (debug probe of a traversal result)
__global__ void k_kd_test_draw(uint32_t *__restrict__ out, const float4 *__restrict__ tri_shade,
                               const Hit *__restrict__ hits, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Hit h = hits[i];
    if (h.t == 0) { out[i] = get_colour(0.f, 0.f, 0.f, 0.f); return; }
    const f3 nrm = surface_at(tri_shade, h).normal;
    out[i] = get_colour((nrm.x + 1) / 2, (nrm.y + 1) / 2, (nrm.z + 1) / 2, 1.0f);
}

// f_buffer_to_byte_buffer, reference: src/kernels/path.cl:447-458
__global__ void k_f_to_byte(uint32_t *__restrict__ out, const float4 *__restrict__ tex, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 d = tex[i];
    out[i] = get_colour(d.x, d.y, d.z, d.w);
}

// f_buffer_average, reference: src/kernels/path.cl:394-419
__global__ void k_f_buffer_average(float4 *__restrict__ out_tex, const float4 *__restrict__ fresh, uint32_t n,
                                   uint32_t sample)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 o = out_tex[i], f = fresh[i];
    float s = (float)sample, sm1 = (float)sample - 1.f;
    o.x = ((o.x - 0.5f) * sm1 + (f.x - 0.5f)) / s;
    o.y = ((o.y - 0.5f) * sm1 + (f.y - 0.5f)) / s;
    o.z = ((o.z - 0.5f) * sm1 + (f.z - 0.5f)) / s;
    o.w = ((o.w - 0.5f) * sm1 + (f.w - 0.5f)) / s;
    out_tex[i] = o;
}

// ---------------------------------------------------------------------------
// brute-force scene intersection for the SS and PATH integrators
// (reference: src/kernels/collision.cl:106-121,179-285,382-468)
// ---------------------------------------------------------------------------
struct SceneDev {
    const int4 *idx4;
    const float4 *verts4, *normals4;
    const float4 *meshes;      // 8 float4 per mesh record
    const float4 *materials;   // 2 float4 per material
    const float4 *spheres;     // 2 float4 per sphere
    const float4 *planes;      // 3 float4 per plane
    uint32_t num_meshes, num_spheres, num_planes;
};

struct Coll { bool did_hit; f3 normal, point; float dist; f3 colour; float refl; };

__device__ __forceinline__ void load_mat(const SceneDev &s, int mi, f3 &colour, float &refl)
{
    colour = xyz(s.materials[2 * (size_t)mi]);
    refl = s.materials[2 * (size_t)mi + 1].x;
}

__device__ __forceinline__ bool slab_hit(f3 vmin, f3 vmax, f3 o, f3 d)
{
    f3 a = mk3((vmin.x - o.x) / d.x, (vmin.y - o.y) / d.y, (vmin.z - o.z) / d.z);
    f3 b = mk3((vmax.x - o.x) / d.x, (vmax.y - o.y) / d.y, (vmax.z - o.z) / d.z);
    f3 rmin = mk3(cl_min(a.x, b.x), cl_min(a.y, b.y), cl_min(a.z, b.z));
    f3 rmax = mk3(cl_max(a.x, b.x), cl_max(a.y, b.y), cl_max(a.z, b.z));
    float mm = cl_min(cl_min(rmax.x, rmax.y), rmax.z);
    float xm = cl_max(cl_max(rmin.x, rmin.y), rmin.z);
    f3 minmax = mk3(mm, mm, mm), maxmin = mk3(xm, xm, xm);
    if (dot3(minmax, minmax) >= dot3(maxmin, maxmin)) return dot3(maxmin, maxmin) > 0.001f;
    return false;
}

__device__ bool collide_primitives(const SceneDev &s, f3 o, f3 d, Coll &res)
{
    float dist = RT_FAR;
    res.did_hit = false; res.dist = RT_FAR;
    for (uint32_t i = 0; i < s.num_spheres; i++) {
        float4 p = s.spheres[2 * i], q = s.spheres[2 * i + 1];
        f3 c = xyz(p);
        float radius = q.x;
        int mi = __float_as_int(q.y);
        f3 L = c - o;
        float b = dot3(d, L);
        float cc = dot3(L, L) - (radius * radius);
        float disc = b * b - cc;
        if (disc < 0.0f) continue;
        float t = b - sqrtf(disc);
        if (t < 0.0f) { t = b + sqrtf(disc); if (t < 0.0f) continue; }
        if (t < dist) {
            dist = t; res.did_hit = true; res.dist = dist;
            res.point = d * dist + o;
            res.normal = normalize3(res.point - c);
            load_mat(s, mi, res.colour, res.refl);
        }
    }
    for (uint32_t i = 0; i < s.num_planes; i++) {
        f3 pp = xyz(s.planes[3 * i]), pn = xyz(s.planes[3 * i + 1]);
        int mi = __float_as_int(s.planes[3 * i + 2].x);
        float denom = dot3(d, pn);
        if (denom < RT_EPS) {
            f3 l = pp - o;
            float t = dot3(l, pn) / denom;
            if (t >= 0 && t < dist) {
                dist = t; res.did_hit = true; res.dist = dist;
                res.point = d * dist + o;
                res.normal = pn;
                load_mat(s, mi, res.colour, res.refl);
            }
        }
    }
    return dist != RT_FAR;
}

__device__ bool collide_meshes(const SceneDev &s, f3 o, f3 d, Coll &res)
{
    float dist = RT_FAR;
    res.did_hit = false; res.dist = RT_FAR;
    for (uint32_t m = 0; m < s.num_meshes; m++) {
        const float4 *mp = s.meshes + 8 * (size_t)m;
        f3 bmax = xyz(mp[4]), bmin = xyz(mp[5]);
        float4 tail = mp[6];
        int index_offset = __float_as_int(tail.x), num_indices = __float_as_int(tail.y), mi = __float_as_int(tail.z);
        if (!slab_hit(bmin, bmax, o, d)) continue;
        float min_t = RT_FAR;
        f3 normal = mk3(0.f, 0.f, 0.f);
        for (int i = 0; i < num_indices / 3; i++) {
            int4 i0 = s.idx4[i * 3 + index_offset], i1 = s.idx4[i * 3 + index_offset + 1],
                 i2 = s.idx4[i * 3 + index_offset + 2];
            f3 a = xyz(s.verts4[i0.x]), b = xyz(s.verts4[i1.x]), c = xyz(s.verts4[i2.x]);
            float t = 0.f, u = 0.f, v = 0.f;
            if (tri_hit_edges(a, b - a, c - a, o, d, t, u, v) && t < min_t && t > 0) {
                min_t = t;
                normal = (xyz(s.normals4[i0.y]) * (1 - u - v) + xyz(s.normals4[i1.y]) * u) + xyz(s.normals4[i2.y]) * v;
            }
        }
        if (min_t != RT_FAR && min_t < dist) {
            dist = min_t;
            res.dist = dist; res.normal = normal;
            res.point = d * dist + o;
            load_mat(s, mi, res.colour, res.refl);
            res.did_hit = true;
        }
    }
    return res.did_hit;
}

__device__ bool collide_all(const SceneDev &s, f3 o, f3 d, Coll &res)
{
    float dist = RT_FAR;
    if (collide_primitives(s, o, d, res)) dist = res.dist;
    Coll m;
    if (collide_meshes(s, o, d, m))
        if (m.dist < dist) res = m;
    return res.did_hit;
}

// shade, reference: src/kernels/ray.cl:2-14
__device__ float4 ss_shade(const SceneDev &s, const Coll &c)
{
    f3 l = normalize3(mk3(1.f, 2.f, 0.f) - c.point);
    f3 ro = c.point + l * 0.00001f;
    Coll tmp;
    float vis = collide_all(s, ro, l, tmp) ? 0.0f : 1.0f;
    return make_float4(vis * c.colour.x, vis * c.colour.y, vis * c.colour.z, vis * 1.0f);
}

__device__ __forceinline__ f3 reflect3(f3 i, f3 n) { return i - n * (2.f * dot3(i, n)); }
__device__ __forceinline__ float4 mix4(float4 a, float4 b, float t)
{
    return make_float4(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t);
}

// cast_ray_test, reference: src/kernels/ray.cl:17-105 (non-Windows: packed as .zyxw)
__global__ void k_cast_rays(SceneDev s, uint32_t *__restrict__ out, const float4 *__restrict__ ray_o,
                            const float4 *__restrict__ ray_d, uint32_t n)
{
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const float4 sky = make_float4((float)0.84, (float)0.87, (float)0.93, 0.f);
    f3 o = xyz(ray_o[id]), d = xyz(ray_d[id]);
    Coll result;
    if (!collide_all(s, o, d, result)) { out[id] = get_colour(sky.x, sky.y, sky.z, sky.w); return; }
    float4 colour = ss_shade(s, result);

    f3 rdir0, rdir1;
    Coll r0, r1;
    float4 c0 = sky, c1 = sky;
    int early = 2;
    {
        f3 ro = result.point + result.normal * 0.0001f;
        rdir0 = reflect3(d, result.normal);
        if (collide_all(s, ro, rdir0, r0)) {
            c0 = ss_shade(s, r0);
            f3 ro1 = r0.point + r0.normal * 0.0001f;
            rdir1 = reflect3(rdir0, r0.normal);
            if (collide_all(s, ro1, rdir1, r1)) c1 = ss_shade(s, r1);
            else { c1 = sky; early = 1; }
        } else { c0 = sky; early = 0; }
    }
    if (early == 2) c1 = mix4(c1, sky, r1.refl);
    if (early >= 1) c0 = mix4(c0, c1, r0.refl);
    colour = mix4(colour, c0, result.refl);
    out[id] = get_colour(colour.z, colour.y, colour.x, colour.w);
}

// path_trace, reference: src/kernels/path.cl:260-366 (non-Windows: stored as .zyx)
__global__ void k_path_trace(SceneDev s, float4 *__restrict__ out_tex, const float4 *__restrict__ ray_d, uint32_t w,
                             uint32_t h, f3 cam, uint32_t magic)
{
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    uint32_t off = x + y * w;
    const f3 sky = mk3((float)0.16, (float)0.2, (float)0.2);
    f3 d = xyz(ray_d[off]);
    f3 o = cam;

    float f = (float)magic * RT_PI_F + (float)(int)x;
    uint32_t seed1 = __float_as_uint(f) + (uint32_t)(int)(sinf((float)(int)x) * 7.1f);
    f = (float)magic * RT_PI_F + (float)(int)y;
    uint32_t seed2 = y + (uint32_t)(int)(sinf((float)__float_as_uint(f)) * 7 * 3.f);

    Coll first;
    if (!collide_all(s, o, d, first)) { out_tex[off] = make_float4(sky.x, sky.y, sky.z, 0.f); return; }
    f3 fin = mk3(0.f, 0.f, 0.f);
    for (int i = 0; i < 4; i++) {
        f3 accum = mk3(0.f, 0.f, 0.f), mask = mk3(1.f, 1.f, 1.f);
        float rand1 = get_random(seed1, seed2);
        float rand2 = get_random(seed1, seed2);
        f3 sdir = cosine_hemisphere(rand1, rand2, first.normal);
        f3 so = first.point + first.normal * 0.0001f;
        mask = mask * first.colour;
        for (int b = 0; b < 4; b++) {
            Coll r;
            if (!collide_all(s, so, sdir, r)) { accum = accum + mask * sky; break; }
            rand1 = get_random(seed1, seed2);
            rand2 = get_random(seed1, seed2);
            sdir = cosine_hemisphere(rand1, rand2, r.normal);
            so = r.point + r.normal * 0.0001f;
            float e = (r.refl == 1.f) ? 1.0f : 0.0f;
            accum = accum + (mask * e) * 2.f;
            mask = mask * r.colour;
            mask = mask * dot3(sdir, r.normal);
        }
        accum = mk3(cl_min(cl_max(accum.x, 0.f), 1.f), cl_min(cl_max(accum.y, 0.f), 1.f),
                    cl_min(cl_max(accum.z, 0.f), 1.f));
        fin = fin + accum * (1.f / 4);
    }
    out_tex[off] = make_float4(fin.z, fin.y, fin.x, 1.f);
}

inline unsigned blocks_for(uint32_t n, unsigned bs) { return (n + bs - 1) / bs; }
inline f3 host3(const float *p) { f3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }

SceneDev make_scene_dev(const rt_scene_view *sc)
{
    SceneDev s;
    s.idx4 = (const int4 *)sc->idx4;
    s.verts4 = (const float4 *)sc->verts4;
    s.normals4 = (const float4 *)sc->normals4;
    s.meshes = (const float4 *)sc->meshes;
    s.materials = (const float4 *)sc->materials;
    s.spheres = (const float4 *)sc->spheres;
    s.planes = (const float4 *)sc->planes;
    s.num_meshes = sc->num_meshes; s.num_spheres = sc->num_spheres; s.num_planes = sc->num_planes;
    return s;
}

void check_kd(const rt_kd_view *kd, const char *who)
{
    if (!kd || !kd->nodes || !kd->leaf_tris || !kd->tri_shade || kd->num_tris == 0) {
        fprintf(stderr, "ERROR: %s: the flattened kd-tree has not been uploaded (no mesh triangles in the scene?)\n", who);
        exit(1);
    }
    if (kd->max_depth >= RT_MAX_STACK) {
        fprintf(stderr, "ERROR: %s: kd-tree depth %u does not fit the %d-entry traversal stack\n", who, kd->max_depth, RT_MAX_STACK);
        exit(1);
    }
}

// Resident one-wave blocks per CU of a persistent kernel on this device (asked of the runtime once per device
// and kernel: two devices of one process may differ).
template <typename K>
int resident_blocks_per_cu(rt_device *dev, K kern)
{
    const void *key = (const void *)kern;
    auto it = dev->blocks_per_cu.find(key);
    if (it != dev->blocks_per_cu.end()) return it->second;
    int b = 0;
    RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kern, 64, 0));
    if (b < 1) b = 1;
    if (getenv("RT_VERBOSE")) fprintf(stderr, "[rt] device %d: persistent kernel %p keeps %d waves per CU\n", dev->ordinal, key, b);
    dev->blocks_per_cu[key] = b;
    return b;
}

// Launch geometry of the persistent traversal kernel: as many one-wave blocks
// as the chip keeps resident for this kernel (LDS-limited), never more than
// there are 64-ray groups.  RT_TRAVERSE_VARIANT (environment, read once)
// selects one of a few compiled parameter sets for tuning runs.
template <int LDS_DEPTH, int REFILL_MIN, int CHUNK>
void launch_kd_variant(rt_device *dev, const TravScene &P, const float4 *ro, const float4 *rd, Hit *hits, uint32_t n)
{
    auto kern = k_kd_intersect<LDS_DEPTH, REFILL_MIN, CHUNK>;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    const int blocks_per_cu = resident_blocks_per_cu(dev, kern);
    static int waves_per_cu = -1;
    if (waves_per_cu < 0) {
        const char *e = getenv("RT_TRAVERSE_WAVES_PER_CU");
        waves_per_cu = e ? atoi(e) : 0;
    }
    unsigned per_cu = (waves_per_cu > 0 && waves_per_cu < blocks_per_cu) ? (unsigned)waves_per_cu : (unsigned)blocks_per_cu;
    unsigned grid = per_cu * (unsigned)dev->props.multiProcessorCount;
    const unsigned need = blocks_for(n, 64);
    if (grid > need) grid = need;
    uint32_t *counter = dev->counters + 16 * dev->current;     // this stream's own 64-byte line
    RT_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(uint32_t), dev->stream));
    rt_launch_timer tm(dev, RT_K_KD_INTERSECT);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, dev->stream, P, ro, rd, hits, n, counter);
    RT_HIP_CHECK(hipGetLastError());
}

void launch_kd_intersect(rt_device *dev, const TravScene &P, const float4 *ro, const float4 *rd, Hit *hits, uint32_t n)
{
    static int variant = -1;
    if (variant < 0) {
        const char *e = getenv("RT_TRAVERSE_VARIANT");
        variant = e ? atoi(e) : 0;
    }
    switch (variant) {
    case 1: launch_kd_variant<16, 16, 256>(dev, P, ro, rd, hits, n); break;
    default: launch_kd_variant<12, 16, 256>(dev, P, ro, rd, hits, n); break;
    }
}


template <int LDS_DEPTH, int SHADE_MIN, int CHUNK, int NI, int NT, int WAVES>
void launch_run_variant(rt_device *dev, const TravScene &P, const SpathDev &B, uint32_t n, uint32_t iters,
                        const uint32_t *rv, uint32_t snap_first, uint32_t snap_every)
{
    auto kern = k_spath_run<LDS_DEPTH, SHADE_MIN, CHUNK, NI, NT, WAVES>;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    int blocks_per_cu = resident_blocks_per_cu(dev, kern);
    const char *cap = getenv("RT_WAVES_PER_CU");      // tuning: fewer resident waves than would fit
    if (cap && atoi(cap) > 0 && atoi(cap) < blocks_per_cu) blocks_per_cu = atoi(cap);
    unsigned grid = (unsigned)blocks_per_cu * (unsigned)dev->props.multiProcessorCount;
    const unsigned need = blocks_for(n, 64);
    if (grid > need) grid = need;
    uint32_t *counter = dev->counters + 16 * dev->current;
    RT_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(uint32_t), dev->stream));
    rt_launch_timer tm(dev, RT_K_SPATH_RUN);
    // pixels change hands in groups of up to 8 lanes in launches of few iterations, one by one in long ones (see
    // k_spath_run); RT_HANDOVER (environment, tuning): the group size
    uint32_t handover = iters <= RT_HANDOVER_GROUPS_UPTO ? 8u : 1u;
    static int handover_env = -2;
    if (handover_env == -2) { const char *e = getenv("RT_HANDOVER"); handover_env = e ? atoi(e) : -1; }
    if (handover_env >= 0) handover = (uint32_t)handover_env;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, dev->stream, P, B, n, iters, rv, counter, snap_first, snap_every, handover);
    RT_HIP_CHECK(hipGetLastError());
}

void launch_spath_run(rt_device *dev, const TravScene &P, const SpathDev &B, uint32_t n, uint32_t iters, const uint32_t *rv,
                      int schedule, uint32_t snap_first, uint32_t snap_every)
{
    // RT_RUN_VARIANT (environment, read once) overrides the schedule with one of the
    // compiled parameter sets, for tuning runs
    static int variant = -1;
    if (variant < 0) {
        const char *e = getenv("RT_RUN_VARIANT");
        variant = e ? atoi(e) : 0;
    }
    if (variant == 0 && schedule == RT_SCHEDULE_TWO_AND_TWO) {
        launch_run_variant<10, 16, 64, 2, 2, 7>(dev, P, B, n, iters, rv, snap_first, snap_every);
        return;
    }
    if (variant == 0 && schedule == RT_SCHEDULE_SHORT_CHAINS) {
        launch_run_variant<16, 16, 64, 2, 2, 5>(dev, P, B, n, iters, rv, snap_first, snap_every);
        return;
    }
    if (variant == 0 && schedule == RT_SCHEDULE_THREE_AND_THREE) {
        launch_run_variant<16, 16, 64, 3, 3, 4>(dev, P, B, n, iters, rv, snap_first, snap_every);
        return;
    }
    // the sets a sweep over the benchmark's workloads has favoured somewhere, beside the four schedules
    switch (variant) {
    case 7: launch_run_variant<10, 20, 64, 1, 1, 7>(dev, P, B, n, iters, rv, snap_first, snap_every); break;    // shading at 20 waiting lanes
    case 11: launch_run_variant<10, 16, 64, 1, 1, 8>(dev, P, B, n, iters, rv, snap_first, snap_every); break;   // 8 waves per SIMD
    case 14: launch_run_variant<10, 16, 64, 1, 2, 7>(dev, P, B, n, iters, rv, snap_first, snap_every); break;   // one descend, two leaf slots
    case 15: launch_run_variant<10, 16, 64, 2, 2, 7>(dev, P, B, n, iters, rv, snap_first, snap_every); break;   // = RT_SCHEDULE_TWO_AND_TWO
    case 17: launch_run_variant<16, 16, 64, 2, 2, 5>(dev, P, B, n, iters, rv, snap_first, snap_every); break;   // = RT_SCHEDULE_SHORT_CHAINS
    case 25: launch_run_variant<16, 16, 64, 2, 2, 4>(dev, P, B, n, iters, rv, snap_first, snap_every); break;   // short chains at 4 waves per SIMD
    case 28: launch_run_variant<16, 16, 64, 3, 3, 4>(dev, P, B, n, iters, rv, snap_first, snap_every); break;   // = RT_SCHEDULE_THREE_AND_THREE
#ifdef RT_EXP_MORE_SLOTS      /* tuning builds: more slots per trip at the default occupancy */
    case 40: launch_run_variant<10, 16, 64, 3, 3, 7>(dev, P, B, n, iters, rv, snap_first, snap_every); break;
    case 41: launch_run_variant<10, 16, 64, 2, 3, 7>(dev, P, B, n, iters, rv, snap_first, snap_every); break;
    case 42: launch_run_variant<10, 16, 64, 3, 2, 7>(dev, P, B, n, iters, rv, snap_first, snap_every); break;
    case 43: launch_run_variant<10, 16, 64, 4, 4, 7>(dev, P, B, n, iters, rv, snap_first, snap_every); break;
    case 44: launch_run_variant<10, 16, 64, 3, 3, 6>(dev, P, B, n, iters, rv, snap_first, snap_every); break;
#endif
    default: launch_run_variant<10, 16, 64, 1, 1, 7>(dev, P, B, n, iters, rv, snap_first, snap_every); break;
    }
}

}  // namespace

#define RT_AFTER_LAUNCH() RT_HIP_CHECK(hipGetLastError())

// ---------------------------------------------------------------------------
// Scratch memory of rt_k_kd_intersect_noalloc: the work counters its persistent
// waves claim rays from, one 64-byte line per stream of each rt_device that is
// open on a GPU (up to RT_NOALLOC_DEVICES_PER_GPU of them at a time).  The array
// lives in the code object, so every GPU the library runs on holds its own copy
// (8 KB) at its own address until the process ends.  In host code the name
// `scratch` is only a handle for hipGetSymbolAddress -- it is not a device
// address and is never passed to a kernel, a memset or a copy: set-up resolves
// it once per device into rt_device::static_counters and everything uses that.
// An rt_device takes a block of RT_MAX_STREAMS lines of its GPU's array when it
// is opened and gives it back when it is closed, so calls through different
// rt_devices (which have different streams and may overlap on the GPU) never
// share a counter.
// ---------------------------------------------------------------------------
#define RT_NOALLOC_DEVICES_PER_GPU 16
#define RT_NOALLOC_MAX_GPUS 64
#define RT_NOALLOC_SCRATCH_WORDS (16 * RT_MAX_STREAMS * RT_NOALLOC_DEVICES_PER_GPU)
static __device__ __align__(256) uint32_t scratch[RT_NOALLOC_SCRATCH_WORDS];

static std::mutex noalloc_slots_lock;
static uint32_t noalloc_slots_taken[RT_NOALLOC_MAX_GPUS];     // per GPU ordinal: bit s = block s is held by an open rt_device

void rt_kernels_setup_device(rt_device *dev)
{
    // the caller (rt_device_open) has made dev->ordinal the current device
    void *p = nullptr;
    RT_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    dev->static_counters = (uint32_t *)p;
    dev->noalloc_slot = -1;
    if (dev->ordinal < RT_NOALLOC_MAX_GPUS) {
        std::lock_guard<std::mutex> hold(noalloc_slots_lock);
        for (int s = 0; s < RT_NOALLOC_DEVICES_PER_GPU; s++)
            if (!(noalloc_slots_taken[dev->ordinal] & (1u << s))) {
                noalloc_slots_taken[dev->ordinal] |= 1u << s;
                dev->noalloc_slot = s;
                break;
            }
    }
    if (dev->noalloc_slot >= 0)       // this device's lines only: other rt_devices on this GPU may be using theirs
        RT_HIP_CHECK(hipMemset(dev->static_counters + 16 * RT_MAX_STREAMS * dev->noalloc_slot, 0,
                               sizeof(uint32_t) * 16 * RT_MAX_STREAMS));
    // the launch geometry is asked for here so that the entry point itself calls nothing that may load or allocate
    RT_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&dev->noalloc_blocks_per_cu, k_kd_intersect<12, 16, 256>, 64, 0));
    if (dev->noalloc_blocks_per_cu < 1) dev->noalloc_blocks_per_cu = 1;
}

void rt_kernels_release_device(rt_device *dev)
{
    if (dev->noalloc_slot < 0 || dev->ordinal >= RT_NOALLOC_MAX_GPUS) return;
    std::lock_guard<std::mutex> hold(noalloc_slots_lock);
    noalloc_slots_taken[dev->ordinal] &= ~(1u << dev->noalloc_slot);
    dev->noalloc_slot = -1;
}

extern "C" {

// The kernels index frame rows without bounds of their own: refuse a mapping
// that leaves the frame.
static void check_rows(const rt_rows *rows, uint32_t h, const char *who)
{
    bool ok = rows && rows->full_height > 0;
    if (ok && h) {
        const uint32_t last = h - 1;
        const uint64_t row = rows->band ? (uint64_t)rows->first + (uint64_t)(last / rows->band) * rows->stride + last % rows->band
                                        : (uint64_t)rows->first + last;
        ok = row < rows->full_height && (!rows->band || rows->stride >= rows->band);
    }
    if (!ok) {
        fprintf(stderr, "ERROR: %s: the row mapping leaves the frame\n", who);
        exit(1);
    }
}

void rt_k_generate_rays(rt_device *dev, rt_mem ray_o, rt_mem ray_d, uint32_t w, uint32_t h, const rt_rows *rows,
                        float tan_half_fov)
{
    uint32_t n = w * h;
    if (!n) return;
    check_rows(rows, h, "rt_k_generate_rays");
    rt_launch_timer tm(dev, RT_K_GENERATE_RAYS);
    hipLaunchKernelGGL(k_generate_rays, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream, (float4 *)ray_o,
                       (float4 *)ray_d, w, h, *rows, tan_half_fov);
    RT_AFTER_LAUNCH();
}

void rt_k_kd_intersect(rt_device *dev, const rt_kd_view *kd, rt_mem ray_o, rt_mem ray_d, rt_mem hits, uint32_t n)
{
    if (!n) return;
    check_kd(kd, "rt_k_kd_intersect");
    TravScene P;
    P.nodes = (const uint4 *)kd->nodes;
    P.leaf_tris = (const float4 *)kd->leaf_tris;
    P.smin = host3(kd->scene_min);
    P.smax = host3(kd->scene_max);
    P.root_ref = kd->root_ref;
    P.plain_splits = kd->plain_splits;
    launch_kd_intersect(dev, P, (const float4 *)ray_o, (const float4 *)ray_d, (Hit *)hits, n);
}

int rt_k_kd_intersect_noalloc(rt_device *dev, const rt_kd_view *kd, rt_mem ray_o, rt_mem ray_d, rt_mem hits, uint32_t n)
{
    if (!n) return 0;
    if (n > RT_NOALLOC_MAX_RAYS) {
        fprintf(stderr, "ERROR: rt_k_kd_intersect_noalloc: %u rays in one call, the limit is %u "
                        "(the 32-bit work counter is claimed in chunks past the end)\n", n, (unsigned)RT_NOALLOC_MAX_RAYS);
        return 1;
    }
    if (!kd || !kd->nodes || !kd->leaf_tris || kd->num_leaf_lines == 0 || kd->max_depth >= RT_MAX_STACK || !ray_o ||
        !ray_d || !hits || !dev->static_counters) {
        fprintf(stderr, "ERROR: rt_k_kd_intersect_noalloc: no flattened kd-tree, or a null ray / hit buffer\n");
        return 2;
    }
    if (dev->noalloc_slot < 0) {
        fprintf(stderr, "ERROR: rt_k_kd_intersect_noalloc: more than %d rt_devices are open on GPU %d; this one has no "
                        "counter lines in the library's static array\n", RT_NOALLOC_DEVICES_PER_GPU, dev->ordinal);
        return 3;
    }
    TravScene P;
    P.nodes = (const uint4 *)kd->nodes;
    P.leaf_tris = (const float4 *)kd->leaf_tris;
    P.smin = host3(kd->scene_min);
    P.smax = host3(kd->scene_max);
    P.root_ref = kd->root_ref;
    P.plain_splits = kd->plain_splits;
    RT_HIP_CHECK(hipSetDevice(dev->ordinal));
    unsigned grid = (unsigned)dev->noalloc_blocks_per_cu * (unsigned)dev->props.multiProcessorCount;
    const unsigned need = blocks_for(n, 64);
    if (grid > need) grid = need;
    // the array keeps the last call's count: clear this device's line for this stream, in stream order, before the launch
    uint32_t *counter = dev->static_counters + 16 * (RT_MAX_STREAMS * dev->noalloc_slot + dev->current);
    RT_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(uint32_t), dev->stream));
    hipLaunchKernelGGL((k_kd_intersect<12, 16, 256>), dim3(grid), dim3(64), 0, dev->stream, P, (const float4 *)ray_o,
                       (const float4 *)ray_d, (Hit *)hits, n, counter);
    RT_HIP_CHECK(hipGetLastError());
    return 0;
}

void rt_k_spath_init(rt_device *dev, const rt_kd_view *kd, const rt_spath_buffers *b, uint32_t w, uint32_t h,
                     const rt_rows *rows, uint32_t random_value)
{
    if (!(w * h)) return;
    check_kd(kd, "rt_k_spath_init");
    check_rows(rows, h, "rt_k_spath_init");
    rt_launch_timer tm(dev, RT_K_SPATH_INIT);
    hipLaunchKernelGGL(k_spath_init, dim3(blocks_for(w * h, 256)), dim3(256), 0, dev->stream,
                       (const float4 *)kd->tri_shade, (float4 *)b->out_tex, (float4 *)b->ray_o, (float4 *)b->ray_d,
                       (float4 *)b->origin_o, (float4 *)b->origin_d, (const Hit *)b->hits, (Hit *)b->source_hits,
                       (float4 *)b->prog_a, (float4 *)b->prog_b, w, h, *rows, random_value);
    RT_AFTER_LAUNCH();
}

void rt_k_spath_trace(rt_device *dev, const rt_kd_view *kd, const rt_spath_buffers *b, uint32_t n,
                      uint32_t random_value, int step_rng)
{
    if (!n) return;
    check_kd(kd, "rt_k_spath_trace");
    rt_launch_timer tm(dev, RT_K_SPATH_TRACE);
    auto kern = step_rng ? k_spath_trace<true> : k_spath_trace<false>;
    hipLaunchKernelGGL(kern, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream,
                       (const float4 *)kd->tri_shade, (float4 *)b->out_tex, (float4 *)b->ray_o, (float4 *)b->ray_d,
                       (const float4 *)b->origin_o, (const float4 *)b->origin_d, (const Hit *)b->hits,
                       (const Hit *)b->source_hits, (float4 *)b->prog_a, (float4 *)b->prog_b,
                       (uint32_t *)b->random, n, random_value);
    RT_AFTER_LAUNCH();
}

void rt_k_xorshift(rt_device *dev, rt_mem random, uint32_t n)
{
    if (!n) return;
    rt_launch_timer tm(dev, RT_K_XORSHIFT);
    hipLaunchKernelGGL(k_xorshift, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream, (uint32_t *)random, n);
    RT_AFTER_LAUNCH();
}

void rt_k_pixel_order(rt_device *dev, rt_mem cost, rt_mem order, rt_mem hist256, uint32_t n, uint32_t shift,
                      uint32_t width, rt_mem tile_scratch)
{
    if (!n) return;
    uint32_t tiles_per_row = 0;
    if (width && width % 8u == 0 && n % width == 0 && (n / width) % 8u == 0) tiles_per_row = width / 8u;
    rt_launch_timer tm(dev, RT_K_PIXEL_ORDER);
    RT_HIP_CHECK(hipMemsetAsync(hist256, 0, RT_COST_BUCKETS * sizeof(uint32_t), dev->stream));
    if (tile_scratch && tiles_per_row) {
        const uint32_t n_tiles = n / 64u;
        uint32_t *tcost = (uint32_t *)tile_scratch, *torder = tcost + n_tiles;
        hipLaunchKernelGGL(k_tile_cost, dim3(blocks_for(n_tiles, 256)), dim3(256), 0, dev->stream, (const uint32_t *)cost,
                           n_tiles, tiles_per_row, tcost);
        hipLaunchKernelGGL(k_cost_hist, dim3(blocks_for(n_tiles, 1024)), dim3(256), 0, dev->stream, (const uint32_t *)tcost,
                           n_tiles, shift, 0u, (uint32_t *)hist256);
        hipLaunchKernelGGL(k_cost_scan, dim3(1), dim3(256), 0, dev->stream, (uint32_t *)hist256);
        hipLaunchKernelGGL(k_cost_scatter, dim3(blocks_for(n_tiles, 1024)), dim3(256), 0, dev->stream,
                           (const uint32_t *)tcost, n_tiles, shift, 0u, (uint32_t *)hist256, torder);
        hipLaunchKernelGGL(k_tile_expand, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream, (const uint32_t *)torder, n,
                           tiles_per_row, (uint32_t *)order);
        RT_AFTER_LAUNCH();
        return;
    }
    hipLaunchKernelGGL(k_cost_hist, dim3(blocks_for(n, 1024)), dim3(256), 0, dev->stream, (const uint32_t *)cost, n,
                       shift, tiles_per_row, (uint32_t *)hist256);
    hipLaunchKernelGGL(k_cost_scan, dim3(1), dim3(256), 0, dev->stream, (uint32_t *)hist256);
    hipLaunchKernelGGL(k_cost_scatter, dim3(blocks_for(n, 1024)), dim3(256), 0, dev->stream, (const uint32_t *)cost, n,
                       shift, tiles_per_row, (uint32_t *)hist256, (uint32_t *)order);
    RT_AFTER_LAUNCH();
}

void rt_k_spath_run(rt_device *dev, const rt_kd_view *kd, const rt_spath_buffers *b, uint32_t n, uint32_t iters,
                    rt_mem random_values, rt_mem cost, rt_mem order, int schedule)
{
    rt_k_spath_run_snap(dev, kd, b, n, iters, random_values, cost, order, schedule, NULL, 0, 0);
}

void rt_k_spath_run_snap(rt_device *dev, const rt_kd_view *kd, const rt_spath_buffers *b, uint32_t n, uint32_t iters,
                         rt_mem random_values, rt_mem cost, rt_mem order, int schedule, rt_mem snapshots,
                         uint32_t snap_first, uint32_t snap_every)
{
    if (!n || !iters) return;
    check_kd(kd, "rt_k_spath_run");
    if (iters > RT_RUN_MAX_ITERS) {
        fprintf(stderr, "ERROR: rt_k_spath_run: %u iterations in one launch, the limit is %u\n", iters, RT_RUN_MAX_ITERS);
        exit(1);
    }
    if (!snapshots || snap_first == 0 || snap_first > iters) {
        snapshots = NULL; snap_first = 0xffffu; snap_every = 0;        // never reached
    } else if (snap_every == 0 || (iters - snap_first) / snap_every + 1 > RT_RUN_MAX_SNAPSHOTS) {
        fprintf(stderr, "ERROR: rt_k_spath_run: more than %u snapshots in one launch\n", RT_RUN_MAX_SNAPSHOTS);
        exit(1);
    }
    TravScene P;
    P.nodes = (const uint4 *)kd->nodes;
    P.leaf_tris = (const float4 *)kd->leaf_tris;
    P.smin = host3(kd->scene_min);
    P.smax = host3(kd->scene_max);
    P.root_ref = kd->root_ref;
    P.plain_splits = kd->plain_splits;
    SpathDev B;
    B.tri_shade = (const float4 *)kd->tri_shade;
    B.out_tex = (float4 *)b->out_tex; B.ray_o = (float4 *)b->ray_o; B.ray_d = (float4 *)b->ray_d;
    B.origin_o = (const float4 *)b->origin_o; B.origin_d = (const float4 *)b->origin_d;
    B.hits = (Hit *)b->hits; B.source_hits = (const Hit *)b->source_hits;
    B.prog_a = (float4 *)b->prog_a; B.prog_b = (float4 *)b->prog_b; B.random = (uint32_t *)b->random;
    B.cost = (uint32_t *)cost;
    B.order = (const uint32_t *)order;
    B.snap = (float4 *)snapshots;
    // the lanes count down to the next snapshot in 16 bits: an interval longer than a launch can be never comes round again
    if (snap_every > 0xffffu) snap_every = 0xffffu;
    if (!cost) { fprintf(stderr, "ERROR: rt_k_spath_run needs a per-pixel cost buffer\n"); exit(1); }
    launch_spath_run(dev, P, B, n, iters, (const uint32_t *)random_values, schedule, snap_first, snap_every);
}

void rt_k_resolve_avg(rt_device *dev, rt_mem out_u32, rt_mem out_tex, uint32_t n)
{
    if (!n) return;
    rt_launch_timer tm(dev, RT_K_RESOLVE_AVG);
    hipLaunchKernelGGL(k_resolve_avg, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream, (uint32_t *)out_u32,
                       (const float4 *)out_tex, n);
    RT_AFTER_LAUNCH();
}

void rt_k_kd_ray_draw(rt_device *dev, rt_mem out_u32, rt_mem ray_o, uint32_t n)
{
    if (!n) return;
    rt_launch_timer tm(dev, RT_K_DEBUG_DRAW);
    hipLaunchKernelGGL(k_kd_ray_draw, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream, (uint32_t *)out_u32,
                       (const float4 *)ray_o, n);
    RT_AFTER_LAUNCH();
}

void rt_k_kd_test_draw(rt_device *dev, const rt_kd_view *kd, rt_mem out_u32, rt_mem hits, uint32_t n)
{
    if (!n) return;
    check_kd(kd, "rt_k_kd_test_draw");
    rt_launch_timer tm(dev, RT_K_DEBUG_DRAW);
    hipLaunchKernelGGL(k_kd_test_draw, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream, (uint32_t *)out_u32,
                       (const float4 *)kd->tri_shade, (const Hit *)hits, n);
    RT_AFTER_LAUNCH();
}

void rt_k_cast_rays(rt_device *dev, const rt_scene_view *sc, rt_mem out_u32, rt_mem ray_o, rt_mem ray_d, uint32_t n)
{
    if (!n) return;
    rt_launch_timer tm(dev, RT_K_CAST_RAYS);
    hipLaunchKernelGGL(k_cast_rays, dim3(blocks_for(n, 64)), dim3(64), 0, dev->stream, make_scene_dev(sc),
                       (uint32_t *)out_u32, (const float4 *)ray_o, (const float4 *)ray_d, n);
    RT_AFTER_LAUNCH();
}

void rt_k_path_trace(rt_device *dev, const rt_scene_view *sc, rt_mem out_tex, rt_mem ray_d, uint32_t w, uint32_t h,
                     const float cam_pos[3], uint32_t magic)
{
    if (!(w * h)) return;
    rt_launch_timer tm(dev, RT_K_PATH_TRACE);
    hipLaunchKernelGGL(k_path_trace, dim3(blocks_for(w, 8), blocks_for(h, 8)), dim3(8, 8), 0, dev->stream,
                       make_scene_dev(sc), (float4 *)out_tex, (const float4 *)ray_d, w, h, host3(cam_pos), magic);
    RT_AFTER_LAUNCH();
}

void rt_k_f_buffer_average(rt_device *dev, rt_mem out_tex, rt_mem fresh, uint32_t n, uint32_t sample)
{
    if (!n) return;
    rt_launch_timer tm(dev, RT_K_F_AVERAGE);
    hipLaunchKernelGGL(k_f_buffer_average, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream, (float4 *)out_tex,
                       (const float4 *)fresh, n, sample);
    RT_AFTER_LAUNCH();
}

void rt_k_f_to_byte(rt_device *dev, rt_mem out_u32, rt_mem tex, uint32_t n)
{
    if (!n) return;
    rt_launch_timer tm(dev, RT_K_F_TO_BYTE);
    hipLaunchKernelGGL(k_f_to_byte, dim3(blocks_for(n, 256)), dim3(256), 0, dev->stream, (uint32_t *)out_u32,
                       (const float4 *)tex, n);
    RT_AFTER_LAUNCH();
}

}  // extern "C"
