/*
 * host_check -- loader + kd-tree builder + serialiser + flattening without a GPU,
 * meant to be built with -fsanitize=address,undefined (make -C raytracer_amd/csrc hostcheck):
 *   build/host_check <resource root> <scene.rsc | {json text}> ...
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "loader.h"
#include "kdtree.h"
/* stubs for the device layer: this driver never touches the GPU */
void *rt_mem_alloc(rt_device *d, size_t n) { (void)d; (void)n; return NULL; }
void rt_mem_free(rt_device *d, rt_mem m) { (void)d; (void)m; }
void rt_mem_upload(rt_device *d, rt_mem m, const void *s, size_t n) { (void)d; (void)m; (void)s; (void)n; }
void rt_mem_zero(rt_device *d, rt_mem m, size_t n) { (void)d; (void)m; (void)n; }
int main(int argc, char **argv)
{
    rt_set_resource_root(argv[1]);
    for (int i = 2; i < argc; i++) {
        scene *s = argv[i][0] == '{' ? load_scene_json(argv[i]) : load_scene_json_url(argv[i]);
        kd_tree *t = kd_tree_init();
        t->s = s;
        kd_tree_construct(t);
        kd_tree_generate_serialized(t);
        kd_tree_flatten(t);
        printf("%s: %u nodes, blob %u bytes, %u records, %u slots\n", argv[i], t->num_nodes_total, t->buffer_size,
               t->flat.num_nodes, t->flat.num_leaf_lines);
        kd_tree_free(t);
        scene_free(s);
    }
    return 0;
}
