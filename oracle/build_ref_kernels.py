#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: compile the reference's OpenCL C kernels for gfx950.

The reference builds its device program at run time from a macro prelude
(src/raytracer.c:38-70: scene counts, SIMT geometry and the `%f`-printed scene
bounds) followed by its six kernel files in the order of build.command:10, and
hands the text to clBuildProgram without options (src/parallel.c:478).  This
script does the same offline with ROCm's clang and ROCm's own OpenCL built-in
library (opencl.bc / ocml.bc / ockl.bc): the prelude becomes -D options, the
kernel files are #included from where they lie under /root/reference (nothing
of them is copied), and the result is one gfx950 code object per scene and
per flavour in oracle/_ref/ plus a JSON file with each kernel's argument
layout (read from the code object's metadata) for oracle/refcl.py.

Flavours:
  default -- OpenCL C defaults, as clBuildProgram(NULL options) would give:
             a*b+c may be contracted into fma, `/` and sqrt are 2.5/3-ulp.
  strict  -- -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt: the
             plain IEEE reading of the source that oracle/oracle_rt.c restates
             (what remains implementation-defined is inside the OpenCL
             built-ins: dot, cross, normalize, sin, cos, tan).

Both inline every function (-mllvm -amdgpu-function-calls=false) so that the
kernels have a fixed private-segment size, and link the built-in library after
optimisation (the printf lowering of the reference's debug helpers needs it).

BLOCKSIZE_X is the device's SIMT width (64 on gfx950), BLOCKSIZE_Y the number
of warps per workgroup the reference's launch uses (one: local size = SIMT
width, src/spath_raytracer.c:162-163).

Run from the repository root where /root/reference is mounted:
    python oracle/build_ref_kernels.py
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REF = os.environ.get("REF", "/root/reference")
CLANG = "/opt/rocm/lib/llvm/bin/clang"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
KERNEL_FILES = ["util.cl", "collision.cl", "irradiance_cache.cl", "ray.cl", "path.cl", "kdtree.cl"]   # build.command:10
FLAVOURS = {
    "default": [],
    "strict": ["-ffp-contract=off", "-cl-fp32-correctly-rounded-divide-sqrt"],
}
SIMT_WIDTH = 64


def scene_specs():
    """Scenes the GPU tests run the reference kernels on: tag -> SceneArrays."""
    from raytracer_amd import scenes
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_render_path_obj2.npz"))      # the reference's own scene file
    path_obj2 = scenes.SceneArrays(verts4=z["verts4"], normals4=z["normals4"], idx4=z["idx4"],
                                   meshes=z["meshes"].view(scenes.MESH_DT).reshape(-1),
                                   materials=z["materials"].view(scenes.MATERIAL_DT).reshape(-1),
                                   spheres=np.zeros(0, scenes.SPHERE_DT), planes=np.zeros(0, scenes.PLANE_DT))
    return {
        "ico4": scenes.icosphere_scene(4),
        "ico6": scenes.icosphere_scene(6),
        "mixed2": scenes.mixed_scene(2),
        "spheres": scenes.sphere_plane_scene(),
        "path_obj2": path_obj2,
        "ico8": scenes.icosphere_scene(8),          # the million-triangle benchmark scene (timing runs only)
    }


def scene_macros(sc) -> dict:
    """The prelude of src/raytracer.c:38-70 for one scene."""
    import oracle
    if sc.idx4.shape[0]:
        # kdt->bounds is the box of all triangle corners (src/kdtree.c:375-400); for the large scene it is taken
        # from the arrays directly instead of through a single-threaded tree build
        used = sc.verts4[np.unique(sc.idx4[:, 0]), :3]
        lo, hi = used.min(axis=0), used.max(axis=0)
        if sc.idx4.shape[0] < 600_000:
            tree = oracle.KdTree(sc.verts4, sc.idx4)
            assert np.array_equal(lo, tree.bounds_min) and np.array_equal(hi, tree.bounds_max)
    else:
        lo = hi = np.zeros(3, np.float32)
    return {
        "SCENE_NUM_SPHERES": str(int(sc.spheres.shape[0])),
        "SCENE_NUM_PLANES": str(int(sc.planes.shape[0])),
        "SCENE_NUM_INDICES": str(int(sc.idx4.shape[0])),
        "SCENE_NUM_MESHES": str(int(sc.meshes.shape[0])),
        "SCENE_NUM_MATERIALS": str(int(sc.materials.shape[0])),
        "_OSX": "1",
        "BLOCKSIZE_X": str(SIMT_WIDTH),
        "BLOCKSIZE_Y": "1",
        "SCENE_MIN": "(%f, %f, %f)" % tuple(float(x) for x in lo),
        "SCENE_MAX": "(%f, %f, %f)" % tuple(float(x) for x in hi),
    }


def kernel_metadata(hsaco: str) -> dict:
    text = subprocess.check_output([READELF, "--notes", hsaco], text=True)
    start = text.index("amdhsa.kernels:")
    end = text.index("\n...", start)
    doc = yaml.safe_load(text[start:end])
    out = {}
    for k in doc["amdhsa.kernels"]:
        out[k[".name"]] = {
            "symbol": k[".symbol"],
            "kernarg_size": k[".kernarg_segment_size"],
            "group_size": k[".group_segment_fixed_size"],
            "private_size": k[".private_segment_fixed_size"],
            "vgprs": k[".vgpr_count"],
            "args": [{"offset": a[".offset"], "size": a[".size"], "kind": a[".value_kind"],
                      "type": a.get(".type_name", "")} for a in k.get(".args", [])],
        }
    return out


def build_one(tag: str, flavour: str, macros: dict, outdir: str) -> str:
    unit = os.path.join(outdir, f"refcl_{tag}.cl")          # only #include lines: the sources stay where they are
    with open(unit, "w") as f:
        for name in KERNEL_FILES:
            f.write(f'#include "{REF}/src/kernels/{name}"\n')
    hsaco = os.path.join(outdir, f"refcl_{tag}_{flavour}.hsaco")
    cmd = [CLANG, "-x", "cl", "-cl-std=CL1.2", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", "-w",
           "-Xclang", "-finclude-default-header", "--rocm-path=/opt/rocm",
           "-Xclang", "-mlink-builtin-bitcode-postopt", "-mllvm", "-amdgpu-function-calls=false"]
    cmd += FLAVOURS[flavour]
    cmd += [f"-D{k}={v}" for k, v in macros.items()]
    cmd += ["-o", hsaco, unit]
    subprocess.check_call(cmd)
    os.remove(unit)
    meta = {"tag": tag, "flavour": flavour, "macros": macros, "flags": FLAVOURS[flavour],
            "simt_width": SIMT_WIDTH, "kernels": kernel_metadata(hsaco)}
    with open(hsaco[:-6] + ".json", "w") as f:
        json.dump(meta, f, indent=1)
    return hsaco


def main():
    if not os.path.isdir(os.path.join(REF, "src", "kernels")):
        print("reference checkout not present: skipping the reference kernels")
        return
    outdir = os.path.join(HERE, "_ref")
    os.makedirs(outdir, exist_ok=True)
    for tag, sc in scene_specs().items():
        macros = scene_macros(sc)
        for flavour in FLAVOURS:
            print("built", os.path.relpath(build_one(tag, flavour, macros, outdir), ROOT))


if __name__ == "__main__":
    main()
