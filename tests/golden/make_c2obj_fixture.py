"""Fixture for configuration 2 on the reference's own geometry: res/scenes/path_obj3.rsc with its first mesh,
mesh/happy.obj (not in the reference checkout), replaced by mesh/test3.obj, the largest mesh the reference ships
(46,966 triangles), beside mesh/light.obj and mesh/indigo_happy_walls.obj exactly as the scene file arranges them.

Runs only where /root/reference exists.  Writes tests/golden/c2obj_scene.npz: the scene arrays as the library's
loader produces them from the reference's files (inputs only; expected outputs come from the oracle at test time)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import raytracer_amd as rt  # noqa: E402

REF = "/root/reference"
text = open(REF + "/res/scenes/path_obj3.rsc").read()
assert text.count('"mesh/happy.obj"') == 1
sc = rt.load_scene(None, REF + "/res", text=text.replace('"mesh/happy.obj"', '"mesh/test3.obj"'))
assert len(sc.meshes) >= 3 and len(sc.materials) == 5 and len(sc.spheres) == 1 and len(sc.planes) == 1
np.savez_compressed(os.path.join(HERE, "c2obj_scene.npz"),
                    verts4=sc.verts4, normals4=sc.normals4, idx4=sc.idx4, meshes=sc.meshes, materials=sc.materials,
                    spheres=sc.spheres, planes=sc.planes)
print("triangles", sc.num_tris, "vertices", len(sc.verts4), "normals", len(sc.normals4))
for m in sc.meshes:
    print("mesh: first index", int(m["index_offset"]), "indices", int(m["num_indices"]), "material", int(m["material_index"]),
          "min", m["min"][:3], "max", m["max"][:3])
