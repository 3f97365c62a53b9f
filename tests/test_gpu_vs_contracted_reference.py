"""The product against the reference built WITH the compiler's default FMA contraction (oracle/_ref `fast`, standing in for
what an nvcc build with -fmad=true computes): the library follows the reference's SOURCE semantics, one
rounding per written operation, so against a contracted build a few decisions (a radius, a tile count, an alpha >= 1/255 test)
land on the other side.  This is the only available proxy for "what a user of the CUDA binary sees"; the counts are printed and
held to the same bars the strict-vs-contracted reference comparison uses (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import util
from util import run_product

pytestmark = pytest.mark.gpu


def _fast():
    return util.reference_build("fast")


def _compare(name, s, gpu_device, light):
    fast = _fast()
    b = fast.forward(s)
    a, _ = run_product(s, gpu_device, light=light)
    moved_radii = int((a["radii"] != b["radii"]).sum())
    dR = abs(int(a["R"]) - int(b["R"]))
    err = np.abs(a["out_color"] - b["out_color"]).max(axis=0)
    over = int((err > 1e-4).sum())
    print("%s: product vs contracted reference build: radii moved %d / %d, |dR| = %d of %d, RGB pixels > 1e-4: %d of %d (max %.3g)"
          % (name, moved_radii, s.P, dR, int(b["R"]), over, err.size, err.max()))
    assert moved_radii <= max(2, s.P // 5000)
    assert dR <= max(8, int(b["R"]) // 50000)
    assert over <= 5e-3 * err.size
    return a, b


def test_small_scene(gpu_device):
    _compare("capsule_circle", util.build_scene("capsule_circle"), gpu_device, light=False)


def test_headline_frame(gpu_device):
    import test_gpu_fullsize as F
    s = F._scene(F.CONFIGS["thuman800k_1080p"])
    _compare("synth-THuman-800K 1080p", s, gpu_device, light=True)
