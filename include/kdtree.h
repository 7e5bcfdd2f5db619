/*
 * kdtree.h -- SAH kd-tree over the scene's triangles.
 *
 * Drop-in for the reference's include/kdtree.h:1-99.  The builder makes the
 * same decisions as the reference's (same SAH constants, event order,
 * termination and classification rules), so the tree -- and with it the order
 * of triangles inside every leaf, which decides which triangle a ray reports --
 * is identical.  Two outputs are produced from it: the reference's serialised
 * wire format (kd_tree_generate_serialized) and the flattened device layout the
 * HIP kernels read (kd_tree_flatten).
 */
#ifndef RT_KDTREE_H
#define RT_KDTREE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include "geom.h"
#include "rt_hip.h"
#include "scene.h"

#define KDTREE_KT 2.0f   /* traversal cost,    reference: include/kdtree.h:8 */
#define KDTREE_KI 1.0f   /* intersection cost, reference: include/kdtree.h:9 */

#define KDTREE_LEAF 1
#define KDTREE_NODE 2

/* wire format, reference: include/kdtree.h:16-35 */
typedef struct _skd_tree_traversal_node {
    uint8_t type;
    uint8_t k;
    float   b;
    size_t  left_ind;    /* in 8-byte words from the start of the blob */
    size_t  right_ind;
} RT_ALIGNED(16) _skd_tree_traversal_node;

typedef struct _skd_tree_leaf_node {
    uint8_t      type;
    unsigned int num_triangles;
    /* followed by num_triangles index offsets, padded to 8 bytes */
} RT_ALIGNED(16) _skd_tree_leaf_node;

typedef struct kd_tree_triangle_buffer {
    unsigned int *triangle_buffer;   /* index-buffer offset of each triangle's first corner */
    unsigned int  num_triangles;
} kd_tree_triangle_buffer;

/* reference: include/kdtree.h:44-50 */
typedef struct kd_tree_collision_result {
    unsigned int triangle_index;
    float t;
    float u;
    float v;
} kd_tree_collision_result;

/* reference: include/kdtree.h:53-63 */
typedef struct kd_tree_node {
    uint8_t k;
    float   b;
    struct kd_tree_node *left;
    struct kd_tree_node *right;
    kd_tree_triangle_buffer triangles;
} kd_tree_node;

/* flattened tree on the host, uploaded by scene_resource_push (layouts: rt_hip.h) */
typedef struct kd_tree_flat {
    uint32_t *nodes;        /* 8 words per two-level record (layout: rt_hip.h) */
    float    *leaf_tris;    /* 64-byte lines of 40-byte triangle slots, every leaf from the start of a line */
    uint32_t  num_nodes;    /* records */
    uint32_t  num_leaf_lines;
    uint32_t  root_ref;
    uint32_t  max_depth;    /* interior nodes on the longest root-to-leaf path */
    uint32_t  plain_splits; /* 1: every split position is 0 or 2^-16 <= |b| <= 2^39 (rt_kd_view.plain_splits) */
} kd_tree_flat;

/* reference: include/kdtree.h:65-91 */
typedef struct kd_tree {
    kd_tree_node *root;

    unsigned int k;

    unsigned int num_nodes_total;
    unsigned int num_tris_padded;
    unsigned int num_traversal_nodes;
    unsigned int num_leaves;
    unsigned int num_indices_total;

    unsigned int max_recurse;
    unsigned int tri_for_leaf_threshold;

    scene *s;
    AABB   bounds;

    char        *buffer;            /* serialised wire format */
    unsigned int buffer_size;
    rt_mem       cl_kd_tree_buffer; /* device copy of flat.nodes */

    /* not in the reference */
    kd_tree_flat flat;
    rt_kd_view   view;              /* device pointers + SCENE_MIN/MAX handed to the kernels */
    void        *arena;             /* node/list storage, released by kd_tree_free */
    unsigned int build_threads;     /* 0 = all available */
} kd_tree;

kd_tree      *kd_tree_init(void);
kd_tree_node *kd_tree_node_init(void);
bool          kd_tree_node_is_leaf(kd_tree_node *node);
void          kd_tree_construct(kd_tree *tree);
void          kd_tree_generate_serialized(kd_tree *tree);

/* not in the reference */
void kd_tree_flatten(kd_tree *tree);     /* fills tree->flat from tree->root */
void kd_tree_free(kd_tree *tree);

#endif
