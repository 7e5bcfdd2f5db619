/*
 * rt_hip.h -- C-ABI device layer of the MI355X path tracer.
 *
 * This is the boundary between the C host code (raytracer.h and friends) and
 * the hand-written HIP kernels.  It takes the place of the reference's OpenCL
 * plumbing (reference: include/parallel.h:15-68, src/parallel.c) and of its
 * clSetKernelArg / clEnqueueNDRangeKernel call sites: plain pointers and sizes
 * only, no C++ or torch types.  Every function is fatal on a device error
 * (message on stderr, exit(1)) -- the reference's ASRT_CL convention
 * (src/debug.c:46-53) without the wait on stdin.
 *
 * All work of one rt_device is issued to that device's single in-order stream
 * (the reference uses one in-order command queue, src/parallel.c:159-174).
 * Launch functions return as soon as the kernel is queued; rt_device_sync or a
 * blocking download orders the host behind them.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *rt_mem;                 /* device pointer (replaces cl_mem) */
typedef struct rt_device rt_device;   /* device ordinal + stream (replaces cl_context/cl_command_queue) */

typedef struct rt_device_props {
    char name[128];
    int compute_units;       /* replaces num_multiprocessors, src/parallel.c:176-247 */
    int wave_size;           /* replaces simt_size */
    int max_waves_per_cu;
    size_t global_mem_bytes;
    size_t lds_bytes_per_cu;
} rt_device_props;

/* ---- device, memory (src/parallel.c:93-174, 282-442) -------------------- */
int        rt_device_count(void);
rt_device *rt_device_open(int ordinal);                 /* create_context */
void       rt_device_close(rt_device *dev);
void       rt_device_get_props(rt_device *dev, rt_device_props *out);   /* cl_info */
void       rt_device_sync(rt_device *dev);              /* clFinish */
void      *rt_device_stream(rt_device *dev);            /* current hipStream_t, for callers that record their own events */
/* A device owns RT_MAX_STREAMS in-order streams.  Launches, zero-fills and
   copies go to the selected one (0 after rt_device_open); work on different
   streams may overlap on the GPU and is unordered until rt_device_sync, which
   waits for all of them.  Independent pixel ranges of one frame are rendered
   on different streams so that the long tail of one range's traversal overlaps
   the bulk of the next. */
#define RT_MAX_STREAMS 8
void       rt_device_select_stream(rt_device *dev, int index);
int        rt_device_ordinal(rt_device *dev);

rt_mem rt_mem_alloc(rt_device *dev, size_t bytes);      /* clCreateBuffer */
void   rt_mem_free(rt_device *dev, rt_mem m);
void   rt_mem_upload(rt_device *dev, rt_mem dst, const void *src, size_t bytes);   /* blocking write */
void   rt_mem_download(rt_device *dev, void *dst, rt_mem src, size_t bytes);       /* blocking read, retrieve_buf */
void   rt_mem_zero(rt_device *dev, rt_mem m, size_t bytes);                        /* zero_buffer (queued) */
/* Page-locked host memory, and a host-to-device copy from it that is only queued on the current stream: the caller
   must leave the source alone until the stream has passed the copy (rt_event_record + rt_event_elapsed_ms /
   rt_device_sync).  For small tables a launch needs, so that launches queue back to back without a host round trip. */
void  *rt_host_alloc(rt_device *dev, size_t bytes);
void   rt_host_free(rt_device *dev, void *p);
void   rt_mem_upload_queued(rt_device *dev, rt_mem dst, const void *pinned_src, size_t bytes);

void   rt_mem_copy(rt_device *dev, rt_mem dst, rt_mem src, size_t bytes);          /* clEnqueueCopyBuffer (queued) */

/* ---- timing with HIP events on the device's stream ---------------------- */
enum {
    RT_K_GENERATE_RAYS = 0, RT_K_KD_INTERSECT, RT_K_SPATH_INIT, RT_K_SPATH_TRACE, RT_K_XORSHIFT,
    RT_K_RESOLVE_AVG, RT_K_CAST_RAYS, RT_K_PATH_TRACE, RT_K_F_AVERAGE, RT_K_F_TO_BYTE, RT_K_SPATH_RUN, RT_K_PIXEL_ORDER, RT_K_DEBUG_DRAW, RT_K_COUNT
};
/* When enabled, every launch below is bracketed by two events; the totals are
   read back (after a sync) with rt_timing_get and cleared with rt_timing_reset. */
void rt_timing_enable(rt_device *dev, int on);
void rt_timing_reset(rt_device *dev);
void rt_timing_get(rt_device *dev, int kernel_id, uint64_t *launches, double *total_ms);
const char *rt_kernel_name(int kernel_id);

/* Stream markers for callers that time a stretch of queued work themselves
   (clEnqueueMarker / clGetEventProfilingInfo in OpenCL terms).  rt_event_record
   queues a marker on the current stream; rt_event_elapsed_ms waits for `stop`
   and returns the device time between the two; both are released with
   rt_event_release. */
typedef void *rt_event;
rt_event rt_event_record(rt_device *dev);
float    rt_event_elapsed_ms(rt_device *dev, rt_event start, rt_event stop);
void     rt_event_release(rt_device *dev, rt_event e);
void     rt_event_wait(rt_device *dev, rt_event e);     /* the host waits until the stream has passed the marker */

/* ---- scene data as the kernels see it ------------------------------------ */
/* Flattened kd-tree (built by kd_tree_flatten, kdtree.h).
 *   nodes      32-byte records, each an interior node ("top") together with
 *              its two children where those are interior too:
 *                word 0..2  f32 split of the top, of its left child, of its right child
 *                word 3     axis of the top (bits 1..0), of the left child (3..2),
 *                           of the right child (5..4); bit 6 / bit 7: the left /
 *                           right child is a leaf
 *                word 4, 5  references of the left child's children (word 4 =
 *                           the leaf descriptor when the left child is a leaf)
 *                word 6, 7  the same for the right child
 *              A reference is (record index << 2) | position (bit 31 clear;
 *              position 0 = the record's top, 1 / 2 = its left / right child)
 *              or a leaf descriptor (bit 31 set): bits 25..0 = the leaf's first
 *              64-byte line in leaf_tris, bits 30..26 = triangle count, 31 meaning
 *              "31 or more, the leaf begins with a header slot".  One fetch serves
 *              two descend steps and leaves cost no node fetch of their own.
 *   leaf_tris  64-byte lines.  Every leaf starts on a line and holds its triangles
 *              in consecutive 40-byte slots {v0.xyz, v1-v0, v2-v0, index offset
 *              as u32}, leaf order (a triangle referenced by several leaves is
 *              stored once per leaf), padded with zeros to the end of its last
 *              line: a ray testing the leaf touches the fewest lines its triangles
 *              fit in.  The header slot of a long leaf is a triangle of zero extent
 *              (no ray hits it; it is tested like the others) whose last word is
 *              the leaf's slot count, header included.
 *   tri_shade  64 bytes per triangle, indexed by index offset / 3:
 *              {n0.xyz, colour.r}, {n1.xyz, colour.g}, {n2.xyz, colour.b},
 *              {reflectivity, 0, 0, 0}.
 * Replaces the serialised blob + index/vertex/normal image buffers that
 * kdtree_intersection and the shading kernels gather from
 * (src/kernels/kdtree.cl:49-65,263-277; src/kernels/path.cl:160-171). */
#define RT_MAX_STACK 64   /* entries of a ray's traversal stack; a tree must be shallower than this */
typedef struct rt_kd_view {
    rt_mem nodes;
    rt_mem leaf_tris;
    rt_mem tri_shade;
    uint32_t num_nodes;          /* records */
    uint32_t num_leaf_lines;     /* 64-byte lines of leaf_tris */
    uint32_t num_tris;
    uint32_t root_ref;           /* reference of the root (a leaf descriptor for a one-leaf tree) */
    float scene_min[3];      /* SCENE_MIN / SCENE_MAX of src/raytracer.c:53-64 */
    float scene_max[3];
    uint32_t max_depth;      /* interior nodes on the longest root-to-leaf path: bounds the traversal stack; the
                                launches refuse a view with max_depth >= RT_MAX_STACK */
    uint32_t plain_splits;   /* 1: every split position is 0 or has 2^-16 <= |b| <= 2^39 (kd_tree_flatten checks): the
                                walk may then compute its plane distances (b - o) / d for rays whose origin and
                                direction components are in like ranges as (b - o) * RN(1/d) corrected by two fused
                                multiply-adds, which is the correctly rounded quotient in those ranges; 0: always the
                                division.  Results are the same either way */
} rt_kd_view;

/* Original (un-flattened) scene arrays for the brute-force integrators
 * (src/kernels/collision.cl:240-285,382-468). */
typedef struct rt_scene_view {
    rt_mem idx4;          /* int[4] per index {v, vn, vt, mesh} */
    rt_mem verts4;        /* float[4] per vertex */
    rt_mem normals4;      /* float[4] per normal */
    rt_mem meshes;        /* 128-byte mesh records */
    rt_mem materials;     /* 32-byte material records */
    rt_mem spheres;       /* 32-byte sphere records */
    rt_mem planes;        /* 48-byte plane records */
    uint32_t num_meshes, num_spheres, num_planes, num_materials;
} rt_scene_view;

/* Per-pixel state of the split path tracer (src/spath_raytracer.c:49-113),
 * structure-of-arrays, 16 bytes per element unless noted:
 *   ray_o/ray_d        current ray origin / direction (xyz, w unused)
 *   origin_o/origin_d  saved primary ray
 *   hits               {index offset u32, t, u, v} of the current ray
 *   source_hits        saved primary hit
 *   prog_a             {mask.xyz, sample_num as u32 bits}
 *   prog_b             {accum.xyz, bounce_num as u32 bits}
 *   random             u32 per pixel
 *   out_tex            float4 accumulation {sum.xyz, sample count} */
typedef struct rt_spath_buffers {
    rt_mem ray_o, ray_d, origin_o, origin_d, hits, source_hits, prog_a, prog_b, random, out_tex;
} rt_spath_buffers;

/* ---- kernel launches ------------------------------------------------------ */
/* Which rows of the full frame a context renders (multi-GPU partitions; no
   counterpart in the reference, which renders whole frames).  Local row j is
   frame row  first + (j / band) * stride + (j % band);  band == 0 means one
   contiguous strip, first + j.  Pixel coordinates that enter the arithmetic
   (camera rays, RNG seeds) are always those of the full frame. */
typedef struct rt_rows {
    uint32_t first, band, stride;
    uint32_t full_height;
} rt_rows;

/* generate_rays (src/kernels/ray.cl:109-133; host src/raytracer.c:182-206).
   h local rows mapped into the full frame by `rows`; tan_half_fov is tanf(40 deg). */
void rt_k_generate_rays(rt_device *dev, rt_mem ray_o, rt_mem ray_d, uint32_t w, uint32_t h,
                        const rt_rows *rows, float tan_half_fov);

/* kdtree_intersection (src/kernels/kdtree.cl:90-321; host src/spath_raytracer.c:136-171).
   Persistent waves pull rays from a device-side work counter (the reference
   does the same through dumb_data[16], src/kernels/kdtree.cl:125,160-166); the
   counter lives in a small buffer owned by the rt_device and is cleared on the
   stream before every launch. */
void rt_k_kd_intersect(rt_device *dev, const rt_kd_view *kd, rt_mem ray_o, rt_mem ray_d, rt_mem hits, uint32_t n);

/* rt_k_kd_intersect for callers that cannot allocate device memory (code under
   HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No
   counterpart in the reference.  Same kernel, same results, but the work counter
   lives in an 8 KB array inside the library's code object (one copy per GPU,
   resolved when the rt_device is opened) instead of memory the library
   allocates, and the call itself allocates nothing: no hipMalloc / hipFree, no
   event creation (rt_timing_* does not see it), fixed launch parameters.
   At most RT_NOALLOC_MAX_RAYS rays per call.  Returns 0 when the launch is
   queued; non-zero, with a message on stderr and nothing launched, when n is
   over the limit (1), the tree or a buffer is missing (2), or this rt_device got
   no counter lines because 16 other rt_devices were open on the same GPU when
   it was opened (3).  Every rt_device holds its own counter lines of its GPU's
   array (one per stream index) from rt_device_open to rt_device_close, so calls
   through different rt_devices, or on different streams of one, may overlap;
   two calls on the same stream are ordered by the stream. */
#define RT_NOALLOC_MAX_RAYS 0x80000000u
int rt_k_kd_intersect_noalloc(rt_device *dev, const rt_kd_view *kd, rt_mem ray_o, rt_mem ray_d, rt_mem hits, uint32_t n);

/* segmented_path_trace_init (src/kernels/path.cl:36-113; host src/spath_raytracer.c:280-316). */
void rt_k_spath_init(rt_device *dev, const rt_kd_view *kd, const rt_spath_buffers *b, uint32_t w, uint32_t h,
                     const rt_rows *rows, uint32_t random_value);

/* segmented_path_trace (src/kernels/path.cl:115-258; host src/spath_raytracer.c:318-353).
   With step_rng != 0 the per-pixel RNG word is advanced first (xorshift_batch
   folded into this launch) and written back. */
void rt_k_spath_trace(rt_device *dev, const rt_kd_view *kd, const rt_spath_buffers *b, uint32_t n,
                      uint32_t random_value, int step_rng);

/* xorshift_batch (src/kernels/path.cl:421-426; host src/spath_raytracer.c:237-250). */
void rt_k_xorshift(rt_device *dev, rt_mem random, uint32_t n);

/* `iters` consecutive render iterations -- xorshift_batch, kdtree_intersection and
   segmented_path_trace for every pixel, iters times -- in ONE launch
   (the loop body of src/spath_raytracer.c:364-385 repeated).  random_values is a
   device array of iters words: the random_value of each iteration, in order.
   Buffers end up exactly as iters rounds of the three separate launches leave them.
   cost (n words, required) receives the number of trips of the kernel's loop each pixel was walked in;
   order (n words, or NULL for 0,1,2,...) is the sequence in which pixels are
   handed to lanes -- scheduling only, results do not depend on it.  schedule
   picks how many triangle slots follow the descend slot in each trip of the
   kernel's loop (RT_SCHEDULE_*): also scheduling only. */
#define RT_SCHEDULE_ONE_LEAF_SLOT  0   /* one descend slot and one leaf slot per trip, 10 stack levels in LDS, 7 waves per SIMD */
#define RT_SCHEDULE_TWO_AND_TWO    1   /* two descend and two leaf slots per trip (a slot no lane needs is jumped over), same
                                          stack and occupancy: fewer trips per ray where rays run through long chains of
                                          small steps; the best form on most mesh scenes, by a few per cent */
#define RT_SCHEDULE_SHORT_CHAINS   2   /* two and two slots, 16 stack levels in LDS, 5 waves per SIMD: deep trees whose rays
                                          would otherwise spill their stacks to memory (unstructured soups), and frames
                                          (or a rank's share of one) with too few pixels to keep every lane busy, where a
                                          batch lasts as long as its heaviest pixel's chain of steps */
#define RT_SCHEDULE_THREE_AND_THREE 3  /* three and three slots, 16 stack levels in LDS, 4 waves per SIMD: shallow trees
                                          (tens of thousands of triangles) on small frames, where a batch is a few pixels'
                                          chains and fewer, longer trips per chain count for more than waves in flight */
#define RT_SCHEDULE_COUNT          4
#define RT_RUN_MAX_ITERS     1024u  /* iterations per launch */
#define RT_RUN_MAX_SNAPSHOTS 32u
void rt_k_spath_run(rt_device *dev, const rt_kd_view *kd, const rt_spath_buffers *b, uint32_t n, uint32_t iters,
                    rt_mem random_values, rt_mem cost, rt_mem order, int schedule);

/* rt_k_spath_run for a launch that spans points at which the reference resolves its frame
   (src/spath_raytracer.c:389-390): after the snap_first-th iteration of the launch and then after every
   snap_every-th, each pixel's out_tex entry is copied to snapshots[slot * n + pixel], slot = 0, 1, ... --
   what out_tex would hold had the launch ended there; rt_k_resolve_avg on a slot gives the frame of that
   iteration.  snapshots == NULL or snap_first == 0: none.  At most RT_RUN_MAX_SNAPSHOTS slots. */
void rt_k_spath_run_snap(rt_device *dev, const rt_kd_view *kd, const rt_spath_buffers *b, uint32_t n, uint32_t iters,
                         rt_mem random_values, rt_mem cost, rt_mem order, int schedule, rt_mem snapshots,
                         uint32_t snap_first, uint32_t snap_every);

/* The order in which the next rt_k_spath_run hands pixels to lanes: heaviest
   first, by cost >> shift (counting sort, 256 buckets), into order[0..n);
   hist256 is 256 words of scratch.  When the frame (`width` pixels wide, n /
   width rows) divides into 8x8 screen tiles and tile_scratch (2 * n / 64 words)
   is given, whole tiles are sorted by their mean cost and each tile's 64 pixels
   stay together; otherwise single pixels are sorted (ties in screen-tile order
   when the frame divides into tiles, index order if not or if width is 0).
   No counterpart in the reference: scheduling only. */
void rt_k_pixel_order(rt_device *dev, rt_mem cost, rt_mem order, rt_mem hist256, uint32_t n, uint32_t shift,
                      uint32_t width, rt_mem tile_scratch);

/* f_buffer_to_byte_buffer_avg (src/kernels/path.cl:428-444; host src/spath_raytracer.c:252-277). */
void rt_k_resolve_avg(rt_device *dev, rt_mem out_u32, rt_mem out_tex, uint32_t n);

/* kdtree_ray_draw (src/kernels/kdtree.cl:323-341; host src/spath_raytracer.c:173-199) and kdtree_test_draw
   (src/kernels/kdtree.cl:344-384; host src/spath_raytracer.c:201-235): the reference's debug probes -- ray origins,
   and the interpolated normals of a traversal's hits, as ARGB colours ((value + 1) / 2 per channel). */
void rt_k_kd_ray_draw(rt_device *dev, rt_mem out_u32, rt_mem ray_o, uint32_t n);
void rt_k_kd_test_draw(rt_device *dev, const rt_kd_view *kd, rt_mem out_u32, rt_mem hits, uint32_t n);

/* cast_ray_test (src/kernels/ray.cl:17-105; host src/ss_raytracer.c:7-56). */
void rt_k_cast_rays(rt_device *dev, const rt_scene_view *sc, rt_mem out_u32, rt_mem ray_o, rt_mem ray_d, uint32_t n);

/* path_trace (src/kernels/path.cl:260-366; host src/path_raytracer.c:28-82). */
void rt_k_path_trace(rt_device *dev, const rt_scene_view *sc, rt_mem out_tex, rt_mem ray_d, uint32_t w, uint32_t h,
                     const float cam_pos[3], uint32_t magic);

/* f_buffer_average (src/kernels/path.cl:394-419; host src/path_raytracer.c:85-109). */
void rt_k_f_buffer_average(rt_device *dev, rt_mem out_tex, rt_mem fresh, uint32_t n, uint32_t sample);

/* f_buffer_to_byte_buffer (src/kernels/path.cl:447-458; host src/path_raytracer.c:111-144). */
void rt_k_f_to_byte(rt_device *dev, rt_mem out_u32, rt_mem tex, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
