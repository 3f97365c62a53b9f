"""Mesh -> point-cloud sampling, numpy only.

Mirrors what /root/reference/structures.py:3795-3899 (Mesh.sample_point_cloud, methods 'uniform' and
'uniform_quantized', driven by /root/reference/sample_point_cloud_from_mesh.py:19-33) produces, without open3d:

  * 'uniform'            area-weighted uniform sampling of the triangle soup (open3d's sample_points_uniformly); every sample's
                         colour comes from the TEXTURE -- the reference casts a ray through the sample (origin 1e-5 before it
                         along (1,1,1)), takes the hit triangle's barycentric weights, forms uv from the triangle's three vt
                         and reads the material's map_Kd image bilinearly (structures.py:3746-3755,
                         plib/render.py:96-180, plib/uv_mapping.py:9-61); here the sample already knows its triangle and
                         weights, the lookup is the same arithmetic.  Meshes without a texture: per-vertex OBJ colours
                         interpolated with the same weights, else white (ones, structures.py:3754).  Normals: vertex normals
                         interpolated with the weights, normalised, then oriented AGAINST the ray direction (1,1,1)
                         (structures.py:3757-3780);
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


def read_mtl(path):
    """Wavefront MTL: material name -> map_Kd file (absolute path), in file order."""
    import os
    out = {}
    cur = None
    with open(path, "r", errors="replace") as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "newmtl":
                cur = " ".join(p[1:])
                out.setdefault(cur, None)
            elif p[0] == "map_Kd" and cur is not None:
                out[cur] = os.path.join(os.path.dirname(os.path.abspath(path)), p[-1])
    return out


def load_texture(path):
    """An image file as the reference sees it: open3d's OBJ reader stores every map_Kd image flipped vertically
    (ReadTriangleMeshFromOBJ: textures_.push_back(*image.FlipVertical())), and get_ray_intersection converts it with
    skimage.img_as_float(...).astype(float32) (structures.py:3750): uint8 / 255.  Returns [h, w, c] float32."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"))
    return (img[::-1].astype(np.float32) / np.float32(255.0)).astype(np.float32)


def uv_lookup(texture, uv):
    """plib/uv_mapping.py:9-61 (UVMap, mode 'wrap'): uv taken mod 1, y = v * h - 0.5, x = u * w - 0.5, bilinear interpolation of
    the texture extended by one wrapped texel on every side.  texture [h, w, c], uv [..., 2] -> [..., c] (float64 like scipy)."""
    tex = np.asarray(texture, np.float64)
    h, w = tex.shape[0], tex.shape[1]
    uv = np.mod(np.asarray(uv, np.float64), 1.0)
    y = uv[..., 1] * h - 0.5
    x = uv[..., 0] * w - 0.5
    y0, x0 = np.floor(y), np.floor(x)
    fy, fx = (y - y0)[..., None], (x - x0)[..., None]
    y0i, x0i = y0.astype(np.int64), x0.astype(np.int64)
    ya, yb, xa, xb = np.mod(y0i, h), np.mod(y0i + 1, h), np.mod(x0i, w), np.mod(x0i + 1, w)
    return ((tex[ya, xa] * (1 - fx) + tex[ya, xb] * fx) * (1 - fy) + (tex[yb, xa] * (1 - fx) + tex[yb, xb] * fx) * fy)


def read_obj(path, load_textures=True, material_numbering="open3d"):
    """Wavefront OBJ reader: 'v x y z [r g b]', 'vt', 'vn', 'f' with v, v/vt, v//vn, v/vt/vn (polygons are fanned), 'mtllib',
    'usemtl'.  Returns dict(vertices [V,3] f64, faces [F,3] i64, colors [V,3] f64 or None, normals [V,3] f64 or None,
    triangle_uvs [F,3,2] f64 or None, material_ids [F] i64, textures: list of [h,w,3] float32 images in material order (open3d's
    mesh.textures; only materials with a map_Kd), or [] ).

    material_numbering="open3d" (default, what the reference sees): material_ids index ALL materials of the MTL file(s) in file
    order (open3d stores tinyobjloader's material index in triangle_material_ids; -1 for a `usemtl` the MTL does not define),
    while `textures` holds only the materials with a map_Kd.  The reference pairs texture t with the triangles whose
    material id == t (plib/render.py:158-173), so a textured material that FOLLOWS an untextured one in the MTL comes out black
    and its texture lands on the wrong triangles -- a quirk of the reference that a drop-in reproduces; single-material scans
    (THuman) and MTLs whose textured materials come first are unaffected.  material_numbering="textured" numbers only the
    materials that carry a map_Kd, i.e. material id == index into `textures` (every textured triangle gets its own texture)."""
    if material_numbering not in ("open3d", "textured"):
        raise ValueError("material_numbering: 'open3d' or 'textured'")
    import os
    vs, cols, vns, vts, faces, face_n, face_t, face_m = [], [], [], [], [], [], [], []
    mtl, mat_order, cur_mat = {}, [], None    # mat_order: the materials the MTL file(s) define, in file order
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
            elif p[0] == "vt":
                vts.append([float(x) for x in p[1:3]])
            elif p[0] == "mtllib":
                mp = os.path.join(os.path.dirname(os.path.abspath(path)), " ".join(p[1:]))
                if os.path.exists(mp):
                    for k, v in read_mtl(mp).items():
                        mtl[k] = v
                        if k not in mat_order:
                            mat_order.append(k)
            elif p[0] == "usemtl":
                cur_mat = " ".join(p[1:])
            elif p[0] == "f":
                idx, nidx, tidx = [], [], []
                for tok in p[1:]:
                    parts = tok.split("/")
                    i = int(parts[0])
                    idx.append(i - 1 if i > 0 else len(vs) + i)
                    if len(parts) >= 2 and parts[1]:
                        j = int(parts[1])
                        tidx.append(j - 1 if j > 0 else len(vts) + j)
                    if len(parts) == 3 and parts[2]:
                        j = int(parts[2])
                        nidx.append(j - 1 if j > 0 else len(vns) + j)
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
                    face_m.append(cur_mat)
                    if len(nidx) == len(idx):
                        face_n.append([nidx[0], nidx[k], nidx[k + 1]])
                    if len(tidx) == len(idx):
                        face_t.append([tidx[0], tidx[k], tidx[k + 1]])
    v = np.asarray(vs, np.float64).reshape(-1, 3)
    fa = np.asarray(faces, np.int64).reshape(-1, 3)
    normals = None
    if vns and len(face_n) == len(faces):          # per-corner normals -> per-vertex (last writer wins, like open3d's loader)
        normals = np.zeros_like(v)
        normals[fa.reshape(-1)] = np.asarray(vns, np.float64)[np.asarray(face_n, np.int64).reshape(-1)]
    tri_uv = None
    if vts and len(face_t) == len(faces):
        tri_uv = np.asarray(vts, np.float64)[np.asarray(face_t, np.int64)]       # [F, 3, 2]
    tex_mats = [m for m in mat_order if mtl.get(m)]          # open3d's mesh.textures: the materials with a diffuse map, file order
    numbered = mat_order if material_numbering == "open3d" else tex_mats
    mat_ids = np.asarray([numbered.index(m) if m in numbered else -1 for m in face_m], np.int64)
    textures = [load_texture(mtl[m]) for m in tex_mats] if (load_textures and tri_uv is not None) else []
    return dict(vertices=v, faces=fa, colors=np.asarray(cols, np.float64) if len(cols) == len(vs) and cols else None,
                normals=normals, triangle_uvs=tri_uv, material_ids=mat_ids, textures=textures)


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


RAY_DIRECTION = np.array([1.0, 1.0, 1.0])     # structures.py:3864 (ray_directions = ones): what the normals are oriented against


def texture_colors(face, weights, triangle_uvs, material_ids, textures):
    """plib/render.py:129-180 for samples that know their triangle: uv = sum_k w_k * vt_k of the hit triangle, every texture
    read at uv and kept where the triangle's material is that texture's (merge_textures=True: the sum over textures)."""
    uv = (np.asarray(triangle_uvs, np.float64)[face] * weights[:, :, None]).sum(1)
    out = np.zeros((face.shape[0], 3), np.float64)
    mids = np.asarray(material_ids)[face]
    for t, tex in enumerate(textures):
        out += uv_lookup(tex, uv)[..., :3] * (mids == t)[:, None]
    return out


def sample_uniform(vertices, faces, num_points, seed=0, colors=None, normals=None, triangle_uvs=None, material_ids=None,
                   textures=None, orient_normals=True):
    """Area-weighted uniform surface samples.  Returns dict(xyz [n,3] f64, rgb [n,3] or None, normal [n,3], face [n], weights
    [n,3]).  rgb: texture lookup when the mesh has triangle_uvs + textures, else interpolated vertex colours, else None."""
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
    if orient_normals:      # structures.py:3776-3780: the normal points back along the ray (sign 0 zeroes it, like np.sign there)
        nrm = nrm * (-1.0 * np.sign((nrm * RAY_DIRECTION).sum(-1, keepdims=True)))
    if triangle_uvs is not None and textures:
        rgb = texture_colors(face, w, triangle_uvs, material_ids if material_ids is not None else np.zeros(faces.shape[0], np.int64),
                             textures)
    else:
        rgb = None if colors is None else (np.asarray(colors, np.float64)[tri] * w[:, :, None]).sum(1)
    return dict(xyz=xyz, rgb=rgb, normal=nrm, face=face, weights=w)


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
                       normals=mesh.get("normals"), triangle_uvs=mesh.get("triangle_uvs"), material_ids=mesh.get("material_ids"),
                       textures=mesh.get("textures"))
    xyz, rgb, nrm = s["xyz"], s["rgb"], s["normal"]
    if rgb is None:           # no texture, no vertex colours: the reference's ray_rgbs are ones (structures.py:3754)
        rgb = np.ones_like(xyz)
    if method == "uniform_quantized":
        xyz, rgb, nrm = quantize_dedupe(xyz, rgb, nrm)
    f = np.float32
    return dict(xyz_w=xyz.astype(f), rgb=None if rgb is None else rgb.astype(f), normal_w=nrm.astype(f))


def to_gaussian_means(q):
    """Voxel grid -> world units of the renderer, the inverse of the quantisation ((q - 512) / 448); the reference's
    renderers do the same through pcgc_rescale (simple_raw_render.py:73-77) with their own offset / factor."""
    return ((np.asarray(q, np.float64) - QUANT_HALF_CUBE) / QUANT_SCALE).astype(np.float32)
