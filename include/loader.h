/*
 * loader.h -- scene files (.rsc JSON + Wavefront OBJ meshes) -> scene.
 *
 * Drop-in for the reference's include/loader.h:1-6 (src/loader.c:228-458).
 * Same two entry points and the same resulting scene members: per OBJ shape one
 * mesh {model, min, max, index_offset, num_indices, material_index}, per face
 * corner one index tuple {v, vn, vt, mesh} with the offsets of the OBJ file in
 * the scene-wide arrays, vertices and normals padded to four floats.
 *
 * The reference resolves file names against "<directory of the executable>/res"
 * with macOS / Win32 calls (src/loader.c:10-97); here the root is, in order:
 * rt_set_resource_root(), the RT_RES_DIR environment variable, "<directory of
 * /proc/self/exe>/res".
 */
#ifndef RT_LOADER_H
#define RT_LOADER_H

#include "scene.h"

scene *load_scene_json(char *data);        /* src/loader.c:228-450 */
scene *load_scene_json_url(char *url);     /* src/loader.c:453-458 */

/* src/loader.c:64-97: whole file + terminating NUL; *ret_length counts the NUL.  Fatal when the file is missing. */
char *load_file(const char *url, long *ret_length);

/* Directory that takes the place of "<bin>/res"; NULL returns to the default lookup. */
void rt_set_resource_root(const char *dir);

/* One OBJ file parsed with the reference's rules (vendored parser semantics:
   triangulating fan, index fix-up against the running counts, one shape per
   'o'/'g' run counted in face lines).  Arrays are malloc'ed; free with
   rt_obj_free.  Exposed for tests and tools. */
typedef struct {
    float *vertices;  unsigned int num_vertices;     /* 3 floats each */
    float *normals;   unsigned int num_normals;      /* 3 floats each */
    float *texcoords; unsigned int num_texcoords;    /* 2 floats each */
    int   *faces;     unsigned int num_faces;        /* {v, vt, vn} per corner; num_faces counts corners */
    unsigned int *shape_face_offset, *shape_length;  /* in face lines */
    unsigned int num_shapes;
    unsigned int num_face_lines;
    unsigned int num_nontriangle_lines;              /* face lines that did not yield exactly one triangle */
} rt_obj;

int  rt_obj_parse(rt_obj *out, const char *buf, size_t len);   /* len counts the terminating NUL; 0 = ok */
void rt_obj_free(rt_obj *o);

#endif
