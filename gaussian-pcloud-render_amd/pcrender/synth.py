"""Deterministic synthetic stand-ins for the reference's example point clouds (numpy, CPU).

The real THuman scans and the model checkpoint are absent from the reference mount
(/root/reference/.MISSING_LARGE_BLOBS), so every workload is synthesised with fixed seeds following
SURVEY.md section 8(d):

  "capsule-man"  a 2.2-unit-tall body of six capsules sampled area-weighted
  synth-THuman-256   200 000 voxelised points, scale_factor 256 (mirrors pcgc_rescale,
                     /root/reference/simple_raw_render.py:73-77, README `--voxelized --scale_factor 256`)
  synth-THuman-800K  800 000 float points, scale_factor 448 (README, /root/reference/README.md:45-46)
  synth-mesh-2M      2 000 000 float points, scale_factor 448
                     (/root/reference/sample_point_cloud_from_mesh.py:13, structures.py:3878)

Gaussian parameters imitate the primitive predictor's outputs (/root/reference/models/model_v2.py:292-324,
358-365 with options.yaml:113-152): per-axis scales ~ clamp(1+0.15 N,0)*(sqrt(3)/scale_factor*6)
(simple_raw_render.py:248-249), un-normalised quaternions (1,0,0,0)+0.05 N, SH with M=13 rows for degree 1
(row 0 = RGB2SH(colour), models/sh_utils.py:114-115).
Profiles:  "inference" opacity 1, SH AC rows 0;  "training" opacity U(0.2,1), SH rows 1-3 = 0.1 N.
"""
import numpy as np

SH_C0 = 0.28209479177387814

# (a, b, r) capsules: torso, head, arms, legs  -- y is up
_CAPSULES = [
    ((0.0, 0.05, 0.0), (0.0, 0.55, 0.0), 0.20),
    ((0.0, 0.85, 0.0), (0.0, 0.90, 0.0), 0.13),
    ((-0.30, 0.55, 0.0), (-0.55, -0.05, 0.0), 0.06),
    ((0.30, 0.55, 0.0), (0.55, -0.05, 0.0), 0.06),
    ((-0.11, -0.05, 0.0), (-0.14, -1.05, 0.0), 0.09),
    ((0.11, -0.05, 0.0), (0.14, -1.05, 0.0), 0.09),
]

CONFIGS = {
    "synth-THuman-256": dict(P=200_000, scale_factor=256.0, voxelized=True),
    "synth-THuman-800K": dict(P=800_000, scale_factor=448.0, voxelized=False),
    "synth-mesh-2M": dict(P=2_000_000, scale_factor=448.0, voxelized=False),
}


def capsule_man(n, rng):
    """n surface points (float64 [n,3]) on the union of capsules, area-weighted."""
    A = np.array([c[0] for c in _CAPSULES], dtype=np.float64)
    B = np.array([c[1] for c in _CAPSULES], dtype=np.float64)
    r = np.array([c[2] for c in _CAPSULES], dtype=np.float64)
    L = np.linalg.norm(B - A, axis=1)
    area = 2 * np.pi * r * L + 4 * np.pi * r * r
    which = rng.choice(len(r), size=n, p=area / area.sum())
    t = rng.random(n)
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return A[which] + (B[which] - A[which]) * t[:, None] + d * r[which, None]


def make_cloud(name, seed=0, P=None):
    """Returns dict(means3D[P,3] f32, rgb[P,3] f32 in [0,1], scale_factor) for one of CONFIGS."""
    cfg = dict(CONFIGS[name])
    if P is not None:
        cfg["P"] = int(P)
    rng = np.random.default_rng(seed)
    n, sf = cfg["P"], cfg["scale_factor"]
    if cfg["voxelized"]:
        pts = np.zeros((0, 3), dtype=np.int64)
        while pts.shape[0] < n:  # quantise + dedupe until n distinct voxels
            q = np.round(capsule_man(int(n * 1.5) + 1024, rng) * 256.0 + 512.0).astype(np.int64)
            allq = np.concatenate([pts, q], axis=0)
            _, first = np.unique(allq, axis=0, return_index=True)
            pts = allq[np.sort(first)]
        pts = pts[:n]
        means = (pts.astype(np.float64) - 512.0) / 256.0
    else:
        means = capsule_man(n, rng)
    rgb = 0.5 + 0.5 * np.sin(7.0 * means + np.array([0.0, 2.0, 4.0]))
    return dict(means3D=means.astype(np.float32), rgb=rgb.astype(np.float32), scale_factor=sf, name=name, seed=seed)


def make_gaussians(cloud, profile="inference", seed=1, sh_rows=13, sh_degree=1):
    """Per-Gaussian rasterizer inputs (all float32 numpy): means3D, scales, rotations, opacities[P,1], shs[P,M,3]."""
    rng = np.random.default_rng(seed)
    means, rgb, sf = cloud["means3D"].astype(np.float64), cloud["rgb"].astype(np.float64), cloud["scale_factor"]
    P = means.shape[0]
    radius = np.sqrt(3) / sf * 6
    scales = np.clip(1.0 + 0.15 * rng.standard_normal((P, 3)), 0.0, None) * radius
    rot = np.array([1.0, 0.0, 0.0, 0.0]) + 0.05 * rng.standard_normal((P, 4))
    means = means + 0.3 / sf * rng.standard_normal((P, 3))  # learned offsets
    shs = np.zeros((P, sh_rows, 3), dtype=np.float64)
    shs[:, 0, :] = (rgb - 0.5) / SH_C0
    if profile == "training":
        opac = rng.uniform(0.2, 1.0, size=(P, 1))
        k = min((sh_degree + 1) ** 2, sh_rows)
        shs[:, 1:k, :] = 0.1 * rng.standard_normal((P, k - 1, 3))
    elif profile == "inference":
        opac = np.ones((P, 1))
    else:
        raise ValueError(profile)
    f = np.float32
    return dict(means3D=means.astype(f), scales=scales.astype(f), rotations=rot.astype(f), opacities=opac.astype(f),
                shs=shs.astype(f), sh_degree=sh_degree, profile=profile)


def random_scene(P, W, H, seed=0, sh_degree=1, sh_rows=None, spread=1.0, scale=0.05, anisotropy=1.0):
    """Small generic test scene in front of a camera at the origin looking down +z (numpy float32 dict)."""
    rng = np.random.default_rng(seed)
    M = sh_rows if sh_rows is not None else (sh_degree + 1) ** 2
    f = np.float32
    means = np.stack([rng.uniform(-spread, spread, P), rng.uniform(-spread, spread, P), rng.uniform(0.5, 6.0, P)], 1)
    scales = np.exp(rng.normal(np.log(scale), 0.4 * anisotropy, (P, 3)))
    rot = rng.standard_normal((P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    return dict(means3D=means.astype(f), scales=scales.astype(f), rotations=rot.astype(f),
                opacities=rng.uniform(0.05, 1.0, (P, 1)).astype(f), shs=(0.6 * rng.standard_normal((P, M, 3))).astype(f),
                colors_precomp=rng.uniform(0, 1, (P, 3)).astype(f), sh_degree=sh_degree)
