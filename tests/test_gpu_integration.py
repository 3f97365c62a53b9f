"""GPU test of INTEGRATION.md section B: the ctypes stub a maintainer of the reference would drop in as
`diff_gaussian_rasterization/_C.py` is EXTRACTED from the document, pointed at the built library and run -- forward,
backward and mark_visible with the reference's argument lists -- against the CPU oracle."""
import os
import re
import types

import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_stub():
    from diff_gaussian_rasterization import _native
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    src = [b for b in blocks if b.startswith("# diff_gaussian_rasterization/_C.py")]
    assert len(src) == 1, "INTEGRATION.md no longer carries the _C.py stub"
    code = src[0].replace('"/path/to/libgsr_hip.so"', repr(_native.LIB_PATH))
    mod = types.ModuleType("stub_C")
    exec(compile(code, "INTEGRATION.md:_C.py", "exec"), mod.__dict__)
    return mod


@pytest.mark.parametrize("name", ["capsule_circle", "cov3d_precomp", "colors_precomp"])
def test_documented_ctypes_stub_matches_oracle(name, oracle, gpu_device):
    stub = _load_stub()
    dev = gpu_device
    s = util.build_scene(name)
    dL = util.seeded_dL(s)
    o, go = oracle.forward_backward(s, dL)

    def t(a):
        return torch.empty(0) if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    args = (t(s.bg), t(s.means3D), t(s.colors_precomp), t(s.opacities), t(s.scales), t(s.rotations), s.scale_modifier,
            t(s.cov3D_precomp), t(s.viewmatrix.reshape(4, 4)), t(s.projmatrix.reshape(4, 4)), s.tanfovx, s.tanfovy, s.H, s.W,
            t(s.shs), s.sh_degree, t(s.campos), False, False)
    R, color, radii, geom, binning, img = stub.rasterize_gaussians(*args)
    assert R == o["R"]
    np.testing.assert_array_equal(radii.cpu().numpy(), o["radii"])
    err = np.abs(color.cpu().numpy() - o["out_color"]).max(axis=0)
    assert (err > 1e-4).mean() <= 2e-3
    g = stub.rasterize_gaussians_backward(args[0], args[1], radii, args[2], args[4], args[5], s.scale_modifier, args[7], args[8],
                                          args[9], s.tanfovx, s.tanfovy, t(dL), args[14], s.sh_degree, args[16], geom, R, binning,
                                          img, False)
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
    gp = {n: x.cpu().numpy() for n, x in zip(names, g)}
    if not (err > 1e-4).any():
        util.check_grads(gp, go, "INTEGRATION.md stub / " + name)
    vis = stub.mark_visible(args[1], args[8], args[9]).cpu().numpy()
    np.testing.assert_array_equal(vis, oracle.mark_visible(s.means3D, s.viewmatrix, s.projmatrix))
