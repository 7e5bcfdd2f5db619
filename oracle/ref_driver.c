/*
 * ref_driver.c -- thin driver around the REFERENCE's own kd-tree builder.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it pulls
 * the reference's src/geom.c and src/kdtree.c (and the headers they need) in
 * from the read-only checkout named by -I flags at build time (see Makefile,
 * REF=/root/reference), in the same order as the reference's unity build
 * (src/_compiler_sources.c:1-51), and exposes one C entry point so tests can
 * ask the real builder + serialiser for a tree blob.  The output library goes
 * to oracle/_ref/ and is never committed.
 *
 * This file covers the host-side builder only.  The reference's device kernels
 * (OpenCL C) are built separately, unmodified, for gfx950 with ROCm's clang and
 * ROCm's own OpenCL built-in library by oracle/build_ref_kernels.py, and are
 * launched by oracle/hsa_runner.c + oracle/refcl.py.
 */
#define CL_TARGET_OPENCL_VERSION 120
#include <assert.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MMX_IMPLEMENTATION
#include <vec.h>
#undef MMX_IMPLEMENTATION
#include <CL/opencl.h>
#include <debug.c>
#include <parallel.h>
#include <scene.h>
#include <kdtree.h>
#include <geom.c>
#include <kdtree.c>

/* Builds the reference tree for (verts4, idx4) and serialises it.
 *   blob_out   malloc'd by the reference (kd_tree_generate_serialized); the
 *              caller copies what it needs and calls ref_kd_release
 *   counts5    {num_nodes_total, num_traversal_nodes, num_leaves,
 *               num_indices_total, num_tris_padded}
 *   bounds6    {min.xyz, max.xyz}
 * Returns the blob size the reference computed (buffer_size). */
__attribute__((visibility("default")))
unsigned int ref_kd_build(const float *verts4, unsigned int nverts, const int *idx4, unsigned int nidx,
                          char **blob_out, unsigned int *counts5, float *bounds6)
{
    scene *s = (scene *)calloc(1, sizeof(scene));
    s->mesh_verts = (vec3 *)verts4;
    s->num_mesh_verts = nverts;
    s->mesh_indices = (ivec3 *)idx4;
    s->num_mesh_indices = nidx;

    kd_tree *tree = kd_tree_init();
    tree->s = s;
    kd_tree_construct(tree);
    kd_tree_generate_serialized(tree);
    fflush(stdout);

    *blob_out = tree->buffer;
    counts5[0] = tree->num_nodes_total;
    counts5[1] = tree->num_traversal_nodes;
    counts5[2] = tree->num_leaves;
    counts5[3] = tree->num_indices_total;
    counts5[4] = tree->num_tris_padded;
    for (int k = 0; k < 3; k++) { bounds6[k] = tree->bounds.min[k]; bounds6[3 + k] = tree->bounds.max[k]; }
    return tree->buffer_size;
}

__attribute__((visibility("default")))
void ref_kd_release(char *blob) { free(blob); }
