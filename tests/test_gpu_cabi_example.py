"""examples/render_cabi.cpp - a C++ host program with no Python or torch in the process - driven on seeded scenes and
checked against the CPU oracle: the C ABI of include/gsr.h is usable from compiled code as it stands."""
import os
import subprocess

import numpy as np
import pytest

import util
from oracle.oracle import Oracle

EXE = os.path.join(util.ROOT, "examples", "render_cabi")


def _write_scene(path, s, dL):
    M = 0 if s.shs is None else s.shs.shape[1]
    with open(path, "wb") as f:
        np.array([s.P, s.sh_degree, M, s.W, s.H, int(dL is not None)], np.int32).tofile(f)
        np.array([s.tanfovx, s.tanfovy, s.scale_modifier], np.float32).tofile(f)
        for a in (s.bg, s.means3D, s.shs, s.opacities, s.scales, s.rotations, s.viewmatrix, s.projmatrix, s.campos):
            np.ascontiguousarray(a, np.float32).tofile(f)
        if dL is not None:
            dL.tofile(f)
    return M


def _read_out(path, s, M, has_dL):
    with open(path, "rb") as f:
        R = int(np.fromfile(f, np.int64, 1)[0])
        color = np.fromfile(f, np.float32, 3 * s.H * s.W).reshape(3, s.H, s.W)
        radii = np.fromfile(f, np.int32, s.P)
        g = None
        if has_dL:
            g = dict(dL_dmean3D=np.fromfile(f, np.float32, s.P * 3).reshape(s.P, 3),
                     dL_dopacity=np.fromfile(f, np.float32, s.P).reshape(s.P, 1),
                     dL_dsh=np.fromfile(f, np.float32, s.P * M * 3).reshape(s.P, M, 3),
                     dL_dscale=np.fromfile(f, np.float32, s.P * 3).reshape(s.P, 3),
                     dL_drot=np.fromfile(f, np.float32, s.P * 4).reshape(s.P, 4),
                     dL_dmean2D=np.fromfile(f, np.float32, s.P * 3).reshape(s.P, 3))
    return R, color, radii, g


@pytest.mark.gpu
@pytest.mark.parametrize("name,backward", [("capsule_circle", True), ("random_aniso", True), ("culled_mix", True), ("all_culled", True), ("capsule_circle", False)])
def test_cpp_host_matches_oracle(tmp_path, name, backward):
    assert os.path.exists(EXE), "examples/render_cabi is built by __graft_entry__.build()"
    s = util.build_scene(name)
    dL = util.seeded_dL(s) if backward else None
    M = _write_scene(tmp_path / "scene.bin", s, dL)
    r = subprocess.run([EXE, str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    R, color, radii, g = _read_out(tmp_path / "out.bin", s, M, backward)
    if backward:
        o, go = Oracle().forward_backward(s, dL)
    else:
        o, go = Oracle().forward(s), None
    assert R == o["R"]
    np.testing.assert_array_equal(radii, o["radii"])
    assert np.abs(color - o["out_color"]).max() <= 1e-4          # the RGB bar of BASELINE.json
    if backward:
        util.check_grads(g, go, name, names=tuple(g))
