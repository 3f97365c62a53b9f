"""Mesh -> point-cloud sampling, numpy only.

Mirrors what /root/reference/structures.py:3795-3899 (Mesh.sample_point_cloud, methods 'uniform' and
'uniform_quantized', driven by /root/reference/sample_point_cloud_from_mesh.py:19-33) produces, without open3d:

  * 'uniform'            area-weighted uniform sampling of the triangle soup (open3d's sample_points_uniformly);
                         colours / normals are interpolated from the vertices with the sample's barycentric weights
                         (the reference looks them up through a ray cast against the same surface point);
  * 'uniform_quantized'  the same samples pushed onto the integer grid the codec uses:
                         q = round(p * 448) + 512  (structures.py:3876-3879), then one point per occupied voxel with
                         np.unique(axis=0, return_index=True) -- i.e. voxels in lexicographic order, attributes of
                         the FIRST sample that fell into each voxel (structures.py:3881-3888).

open3d's random stream cannot be reproduced, so sample positions are only distribution-equal to the reference's;
everything after the sampling (quantisation, dedupe order, attribute pick) is the same arithmetic.
"""
import numpy as np

QUANT_SCALE = 448          # structures.py:3876
QUANT_HALF_CUBE = 512      # structures.py:3877


def read_obj(path):
    """Minimal Wavefront OBJ reader: 'v x y z [r g b]', 'vn', 'f' with v, v/vt, v//vn, v/vt/vn; polygons are fanned.
    Returns dict(vertices [V,3] f64, faces [F,3] i64, colors [V,3] f64 or None, normals [V,3] f64 or None)."""
    vs, cols, vns, faces, face_n = [], [], [], [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                vs.append([float(x) for x in p[1:4]])
                if len(p) >= 7:
                    cols.append([float(x) for x in p[4:7]])
            elif p[0] == "vn":
                vns.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                idx, nidx = [], []
                for tok in p[1:]:
                    parts = tok.split("/")
                    i = int(parts[0])
                    idx.append(i - 1 if i > 0 else len(vs) + i)
                    if len(parts) == 3 and parts[2]:
                        j = int(parts[2])
                        nidx.append(j - 1 if j > 0 else len(vns) + j)
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
                    if len(nidx) == len(idx):
                        face_n.append([nidx[0], nidx[k], nidx[k + 1]])
    v = np.asarray(vs, np.float64).reshape(-1, 3)
    fa = np.asarray(faces, np.int64).reshape(-1, 3)
    normals = None
    if vns and len(face_n) == len(faces):          # per-corner normals -> per-vertex (last writer wins, like open3d's loader)
        normals = np.zeros_like(v)
        normals[fa.reshape(-1)] = np.asarray(vns, np.float64)[np.asarray(face_n, np.int64).reshape(-1)]
    return dict(vertices=v, faces=fa, colors=np.asarray(cols, np.float64) if len(cols) == len(vs) and cols else None,
                normals=normals)


def triangle_areas(vertices, faces):
    a, b, c = vertices[faces[:, 0]], vertices[faces[:, 1]], vertices[faces[:, 2]]
    return 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)


def vertex_normals(vertices, faces):
    """Area-weighted vertex normals (what open3d's compute_vertex_normals gives before normalisation quirks)."""
    a, b, c = vertices[faces[:, 0]], vertices[faces[:, 1]], vertices[faces[:, 2]]
    fn = np.cross(b - a, c - a)
    n = np.zeros_like(vertices)
    for k in range(3):
        np.add.at(n, faces[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return n / np.where(ln > 0, ln, 1.0)


def sample_uniform(vertices, faces, num_points, seed=0, colors=None, normals=None):
    """Area-weighted uniform surface samples.  Returns dict(xyz [n,3] f64, rgb [n,3] or None, normal [n,3], face [n])."""
    vertices = np.asarray(vertices, np.float64)
    faces = np.asarray(faces, np.int64)
    if faces.shape[0] == 0 or num_points <= 0:
        z = np.zeros((0, 3))
        return dict(xyz=z, rgb=None if colors is None else z.copy(), normal=z.copy(), face=np.zeros((0,), np.int64))
    rng = np.random.default_rng(seed)
    area = triangle_areas(vertices, faces)
    tot = area.sum()
    if not tot > 0:
        raise ValueError("mesh has zero surface area")
    face = rng.choice(faces.shape[0], size=int(num_points), p=area / tot)
    r1 = np.sqrt(rng.random(int(num_points)))
    r2 = rng.random(int(num_points))
    w = np.stack([1.0 - r1, r1 * (1.0 - r2), r1 * r2], 1)          # barycentric, uniform over the triangle
    tri = faces[face]
    xyz = (vertices[tri] * w[:, :, None]).sum(1)
    if normals is None:
        normals = vertex_normals(vertices, faces)
    nrm = (np.asarray(normals, np.float64)[tri] * w[:, :, None]).sum(1)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = nrm / np.where(ln > 0, ln, 1.0)
    rgb = None if colors is None else (np.asarray(colors, np.float64)[tri] * w[:, :, None]).sum(1)
    return dict(xyz=xyz, rgb=rgb, normal=nrm, face=face)


def quantize_dedupe(xyz, *attrs):
    """structures.py:3876-3888: q = round(xyz * 448) + 512; one point per occupied voxel, voxels in np.unique's
    lexicographic order, attributes taken from the first sample of each voxel.  Returns (q [m,3] f64, attrs...)."""
    q = np.round(np.asarray(xyz, np.float64) * QUANT_SCALE)
    q += QUANT_HALF_CUBE
    _, index = np.unique(q, axis=0, return_index=True)
    return (q[index],) + tuple(None if a is None else np.asarray(a)[index] for a in attrs)


def sample_point_cloud(mesh, num_points, method="uniform_quantized", seed=0):
    """mesh: dict from read_obj (vertices, faces, optional colors / normals).  Returns dict(xyz_w, rgb, normal_w) as
    float32 arrays, the fields of the reference's PointCloud (structures.py:3890-3894)."""
    if method not in ("uniform", "uniform_quantized"):
        raise NotImplementedError(method)
    s = sample_uniform(mesh["vertices"], mesh["faces"], num_points, seed=seed, colors=mesh.get("colors"),
                       normals=mesh.get("normals"))
    xyz, rgb, nrm = s["xyz"], s["rgb"], s["normal"]
    if method == "uniform_quantized":
        xyz, rgb, nrm = quantize_dedupe(xyz, rgb, nrm)
    f = np.float32
    return dict(xyz_w=xyz.astype(f), rgb=None if rgb is None else rgb.astype(f), normal_w=nrm.astype(f))


def to_gaussian_means(q):
    """Voxel grid -> world units of the renderer, the inverse of the quantisation ((q - 512) / 448); the reference's
    renderers do the same through pcgc_rescale (simple_raw_render.py:73-77) with their own offset / factor."""
    return ((np.asarray(q, np.float64) - QUANT_HALF_CUBE) / QUANT_SCALE).astype(np.float32)
