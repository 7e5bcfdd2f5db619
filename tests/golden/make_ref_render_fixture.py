"""Fixture from the one stored render of the reference that matches a scene file it still ships:
renders/3_custom_2046.PNG is res/scenes/path_obj2.rsc at 1920x1080 (written by the reference's Windows
build, whose frames are not channel-swapped).  Its radiance comes from another revision of the light
transport (walls 1.5-1.9x darker than today's kernels give, with less of the walls' blue tint, i.e. fewer
bounces), so only what does not depend on that is taken from it: which pixels show an emitter (all three
channels saturated) and which show the orange objects (red - blue > 25; misses their parts in deep shadow).

Runs only where /root/reference exists.  Writes tests/golden/ref_render_path_obj2.npz:
  scene arrays as the library's loader produces them from the reference's files (inputs),
  emitter and object masks of the stored render, bit-packed (expected outputs).
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import raytracer_amd as rt  # noqa: E402

REF = "/root/reference"
sc = rt.load_scene("scenes/path_obj2.rsc", REF + "/res")
img = np.asarray(Image.open(REF + "/renders/3_custom_2046.PNG").convert("RGB")).astype(np.int32)
assert img.shape == (1080, 1920, 3)
emitter = img.min(-1) >= 250
objects = (img[..., 0] - img[..., 2] > 25) & ~emitter
np.savez_compressed(os.path.join(HERE, "ref_render_path_obj2.npz"),
                    verts4=sc.verts4, normals4=sc.normals4, idx4=sc.idx4, meshes=sc.meshes, materials=sc.materials,
                    width=1920, height=1080, emitter_bits=np.packbits(emitter), object_bits=np.packbits(objects))
print("emitter pixels", int(emitter.sum()), "object pixels", int(objects.sum()))
