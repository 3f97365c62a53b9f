"""Point-cloud ingest without open3d: the PLY files the reference reads and writes.

The reference loads `pcd_0.ply` with open3d (`simple_benchmark.py:171-184`) and writes point clouds with
`o3d.io.write_point_cloud(..., write_ascii=True)` (`structures.py:826-848`, `util_rescale_ply.py:8-36`):
vertex elements with double/float x y z, optional float nx ny nz, optional uchar red green blue, in ASCII or
binary_little_endian.  open3d exposes colours as float64 in [0,1] (uchar / 255) and points as float64.
`pcgc_rescale` is `(xyz - offset) / factor` (`simple_raw_render.py:73-77`, `util_rescale_ply.py:8-11`).
"""
import numpy as np

_PLY_DTYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
               "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
               "double": "f8", "float64": "f8"}


def read_ply(path):
    """Returns dict(points[N,3] float64, colors[N,3] float64 in [0,1] or None, normals[N,3] float64 or None)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("%s: list property in vertex element" % path)
                props.append((tok[2], _PLY_DTYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n, ndmin=2, dtype=np.float64) if n else np.zeros((0, len(props)))
            if data.shape != (n, len(props)):
                raise ValueError("%s: expected %d x %d vertex values, got %s" % (path, n, len(props), data.shape))
            cols = {nm: data[:, i] for i, nm in enumerate(names)}
        elif fmt in ("binary_little_endian", "binary_big_endian"):
            end = "<" if fmt == "binary_little_endian" else ">"
            rec = np.dtype([(nm, end + dt) for nm, dt in props])
            data = np.frombuffer(f.read(rec.itemsize * n), dtype=rec, count=n)
            cols = {nm: data[nm].astype(np.float64) for nm in names}
        else:
            raise ValueError("%s: unsupported PLY format %r" % (path, fmt))
    for k in ("x", "y", "z"):
        if k not in cols:
            raise ValueError("%s: vertex element has no %s" % (path, k))
    out = {"points": np.stack([cols["x"], cols["y"], cols["z"]], 1), "colors": None, "normals": None}
    if all(k in cols for k in ("red", "green", "blue")):
        scale = 255.0 if dict(props)["red"] == "u1" else 1.0
        out["colors"] = np.stack([cols["red"], cols["green"], cols["blue"]], 1) / scale
    if all(k in cols for k in ("nx", "ny", "nz")):
        out["normals"] = np.stack([cols["nx"], cols["ny"], cols["nz"]], 1)
    return out


def write_ply_ascii(path, points, colors=None, normals=None):
    """ASCII PLY in open3d's layout (double x y z, [double nx ny nz], [uchar red green blue])."""
    points = np.asarray(points, dtype=np.float64)
    n = points.shape[0]
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment Created by pcrender\nelement vertex %d\n" % n)
        f.write("property double x\nproperty double y\nproperty double z\n")
        if normals is not None:
            f.write("property double nx\nproperty double ny\nproperty double nz\n")
        if colors is not None:
            f.write("property uchar red\nproperty uchar green\nproperty uchar blue\n")
        f.write("end_header\n")
        cols = [points]
        fmts = ["%.10g"] * 3
        if normals is not None:
            cols.append(np.asarray(normals, dtype=np.float64))
            fmts += ["%.10g"] * 3
        if colors is not None:
            cols.append(np.clip(np.round(np.asarray(colors, dtype=np.float64) * 255.0), 0, 255))
            fmts += ["%d"] * 3
        np.savetxt(f, np.concatenate(cols, 1), fmt=" ".join(fmts))


def pcgc_rescale(xyz, offset=512, factor=256):
    """Voxel coordinates -> world units, `(xyz - offset) / factor`."""
    return (np.asarray(xyz) - offset) / factor


def simple_render_primitives(points, colors, sigma, scale_factor=1.0, voxelized=False, sh_rows=13):
    """Per-Gaussian inputs of the reference's model-free `Simple_Render` (simple_raw_render.py:688-726):
    isotropic scales sigma (divided by scale_factor for voxelised input), identity quaternions, opacity 1,
    SH rows = [RGB2SH(colour), zeros...] (M = 13 rows for degree 1).  float32 numpy arrays."""
    P = points.shape[0]
    s = float(sigma) / float(scale_factor) if voxelized else float(sigma)
    shs = np.zeros((P, sh_rows, 3), dtype=np.float32)
    shs[:, 0, :] = (np.asarray(colors, dtype=np.float64) - 0.5) / 0.28209479177387814
    rot = np.zeros((P, 4), dtype=np.float32)
    rot[:, 0] = 1.0
    return dict(means3D=np.asarray(points, dtype=np.float32), scales=np.full((P, 3), s, dtype=np.float32), rotations=rot,
                opacities=np.ones((P, 1), dtype=np.float32), shs=shs, sh_degree=1)
