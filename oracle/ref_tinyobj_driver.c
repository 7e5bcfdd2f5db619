/*
 * ref_tinyobj_driver.c -- thin driver around the OBJ parser the REFERENCE vendors.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes
 * libs/tinyobj_loader_c.h from the read-only reference checkout named by the
 * -I flag at build time (see Makefile, REF=/root/reference) and exposes its
 * parse call, with the flag the reference's loader passes (src/loader.c:121),
 * so tests can compare the product's OBJ reader with the real one.  The output
 * library goes to oracle/_ref/ and is never committed.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define TINYOBJ_LOADER_C_IMPLEMENTATION
#include <tinyobj_loader_c.h>

typedef struct {
    float *vertices, *normals, *texcoords;
    unsigned int num_vertices, num_normals, num_texcoords;
    int *faces;                 /* {v, vt, vn} per corner, as tinyobj_vertex_index_t */
    unsigned int num_faces;     /* corners */
    int *face_num_verts;
    unsigned int num_face_num_verts;
    unsigned int *shape_face_offset, *shape_length;
    unsigned int num_shapes;
} ref_obj;

__attribute__((visibility("default")))
int ref_obj_parse(ref_obj *out, const char *buf, size_t len)
{
    tinyobj_attrib_t attrib;
    tinyobj_shape_t *shapes = NULL;
    size_t num_shapes = 0, num_materials = 0;
    tinyobj_material_t *materials = NULL;
    memset(out, 0, sizeof *out);
    int ret = tinyobj_parse_obj(&attrib, &shapes, &num_shapes, &materials, &num_materials, buf, len,
                                TINYOBJ_FLAG_TRIANGULATE);
    if (ret != TINYOBJ_SUCCESS) return ret;
    out->vertices = attrib.vertices;   out->num_vertices = attrib.num_vertices;
    out->normals = attrib.normals;     out->num_normals = attrib.num_normals;
    out->texcoords = attrib.texcoords; out->num_texcoords = attrib.num_texcoords;
    out->faces = (int *)attrib.faces;  out->num_faces = attrib.num_faces;
    out->face_num_verts = attrib.face_num_verts;
    out->num_face_num_verts = attrib.num_face_num_verts;
    out->num_shapes = (unsigned int)num_shapes;
    out->shape_face_offset = (unsigned int *)malloc(sizeof(unsigned int) * (num_shapes + 1));
    out->shape_length = (unsigned int *)malloc(sizeof(unsigned int) * (num_shapes + 1));
    for (size_t i = 0; i < num_shapes; i++) {
        out->shape_face_offset[i] = shapes[i].face_offset;
        out->shape_length[i] = shapes[i].length;
    }
    return 0;
}

__attribute__((visibility("default")))
unsigned int ref_obj_index_size(void) { return (unsigned int)sizeof(tinyobj_vertex_index_t); }
