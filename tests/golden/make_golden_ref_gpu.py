"""Generate golden vectors from the REFERENCE BUILD (oracle/_ref/libgsr_ref_strict.so = the reference's own
.cu kernels compiled for gfx950, see oracle/build_ref.sh).  Must run on the MI355X box:

    gpurun -- 'python tests/golden/make_golden_ref_gpu.py gpurun_out/golden'

then copy gpurun_out/golden/ref_*.npz into tests/golden/ and commit.  Inputs are NOT stored: every scene is
rebuilt from its seed by tests/util.build_scene (a sha1 of the input bytes is stored and checked).  Stored per
scene: every integer output, the per-Gaussian forward floats of visible Gaussians, the image, and all nine
gradient tensors for the seeded dL/dpixel of tests/util.seeded_dL.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import util  # noqa: E402
from oracle.oracle import Reference  # noqa: E402

GOLDEN_SCENES = ["random_aniso", "sh_deg3", "colors_precomp", "cov3d_precomp", "culled_mix", "voxel_ties",
                 "opaque_early_stop", "capsule_axis_view", "one_gaussian", "all_culled"]


def scene_digest(s):
    h = hashlib.sha1()
    for f in s.FIELDS:
        a = getattr(s, f)
        h.update(b"-" if a is None else np.ascontiguousarray(a).tobytes())
    h.update(np.array([s.W, s.H, s.sh_degree], np.int64).tobytes())
    h.update(np.array([s.tanfovx, s.tanfovy, s.scale_modifier], np.float64).tobytes())
    return h.hexdigest()


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    ref = Reference("strict")
    for name in GOLDEN_SCENES:
        s = util.build_scene(name)
        dL = util.seeded_dL(s)
        f, g = ref.forward_backward(s, dL)
        d = dict(digest=np.array(scene_digest(s)), R=np.int64(f["R"]), radii=f["radii"], out_color=f["out_color"])
        if s.P:
            vis = f["radii"] > 0
            d.update(tiles_touched=f["tiles_touched"], keys=f["keys"], vals=f["vals"], ranges=f["ranges"],
                     n_contrib=f["n_contrib"].astype(np.uint16) if f["n_contrib"].max(initial=0) < 65536 else f["n_contrib"],
                     final_T=f["final_T"], means2D=f["means2D"][vis], depths=f["depths"][vis],
                     conic_opacity=f["conic_opacity"][vis])
            if s.shs is not None:
                d.update(rgb=f["rgb"][vis], clamped=np.packbits(f["clamped"][vis].astype(bool)))
            for k, v in g.items():
                d[k] = v
        np.savez_compressed(os.path.join(out_dir, "ref_%s.npz" % name), **d)
        print(name, "R=%d" % f["R"], "bytes=%d" % os.path.getsize(os.path.join(out_dir, "ref_%s.npz" % name)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden"))
