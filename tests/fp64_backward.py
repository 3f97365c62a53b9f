"""Float64 evaluation of the per-Gaussian backward chain (test infrastructure).

The plain-C oracle restates the reference's per-Gaussian backward (CR/backward.cu:144-396) in float32 with the reference's
own expression order, so on an ill-conditioned splat (a nearly singular conic makes dL/d(a,b,c) the difference of
products 10^3 times larger than the result) the oracle and the reference build share their rounding errors and agree with
each other far better than either agrees with the exact value.  This module evaluates the same mathematics in float64 from
the RENDER-level sums (dL/dmean2D, dL/dconic, dL/dcolour, which the oracle accumulates in double), written from the
formulas in SURVEY.md Appendix A.9 / A.10, and serves as the arbiter when two float32 implementations disagree.

Inputs are the float32 tensors the kernels saw (the forward state -- radii, clamp mask -- is taken as given).
"""
import numpy as np

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]
F = np.float32


def _f(x):
    """a float32 constant / input as the kernels hold it, promoted to float64"""
    return np.float64(F(x))


def gaussian_backward_fp64(s, radii, clamped, dL_dmean2D, dL_dconic, dL_dcolor, rows=None, dtype=np.float64):
    """s: oracle.Scene; radii [P] int, clamped [P,3] bool (forward state); dL_dmean2D [P,>=2], dL_dconic [P,4] (x, y, -, w),
    dL_dcolor [P,3].  rows: optional index array (evaluate only those Gaussians).  Returns dict of arrays for the
    selected rows: dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot.  dtype=np.float32 runs the same expressions in single
    precision: the distance between the two results is the rounding scale any float32 implementation has on that row."""
    dt = dtype
    idx = np.arange(s.P) if rows is None else np.asarray(rows)
    n = idx.size
    p = s.means3D[idx].astype(dt)
    vis = (np.asarray(radii)[idx] > 0)
    V = s.viewmatrix.astype(dt)
    Pm = s.projmatrix.astype(dt)
    g2 = np.asarray(dL_dmean2D, dt)[idx]
    gcon = np.asarray(dL_dconic, dt)[idx]
    gcol = np.asarray(dL_dcolor, dt)[idx]
    W3 = np.array([[V[0], V[4], V[8]], [V[1], V[5], V[9]], [V[2], V[6], V[10]]], dtype=dt)       # rows r0, r1, r2
    tr = np.array([V[12], V[13], V[14]], dtype=dt)
    fx = dt(F(s.W) / (F(2.0) * F(s.tanfovx)))                                  # focal as the kernel computes it (float32)
    fy = dt(F(s.H) / (F(2.0) * F(s.tanfovy)))
    limx, limy = dt(F(1.3) * F(s.tanfovx)), dt(F(1.3) * F(s.tanfovy))
    one, two, half = dt(1.0), dt(2.0), dt(0.5)

    # ---- 3D covariance
    mod = dt(F(s.scale_modifier))
    if s.cov3D_precomp is not None:
        c6 = s.cov3D_precomp[idx].astype(dt)
        S = np.zeros((n, 3, 3), dt)
        S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2] = c6.T
        S[:, 1, 0], S[:, 2, 0], S[:, 2, 1] = S[:, 0, 1], S[:, 0, 2], S[:, 1, 2]
        R = sc = None
    else:
        q = s.rotations[idx].astype(dt)
        r, x, y, z = q.T
        R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                      np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                      np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)   # [n,3,3] rows
        sc = s.scales[idx].astype(dt) * mod
        S = np.einsum("nij,nj,nkj->nik", R, sc * sc, R)

    # ---- EWA projection (A.3) and its backward (A.9)
    t = p @ W3.T + tr
    txtz, tytz = t[:, 0] / t[:, 2], t[:, 1] / t[:, 2]
    cx, cy = (txtz < -limx) | (txtz > limx), (tytz < -limy) | (tytz > limy)
    tx, ty, tz = np.clip(txtz, -limx, limx) * t[:, 2], np.clip(tytz, -limy, limy) * t[:, 2], t[:, 2]
    J00, J02, J11, J12 = fx / tz, -fx * tx / (tz * tz), fy / tz, -fy * ty / (tz * tz)
    m0 = J00[:, None] * W3[0] + J02[:, None] * W3[2]
    m1 = J11[:, None] * W3[1] + J12[:, None] * W3[2]
    u, w = np.einsum("nij,nj->ni", S, m0), np.einsum("nij,nj->ni", S, m1)
    a = np.einsum("ni,ni->n", m0, u) + dt(F(0.3))
    b = np.einsum("ni,ni->n", m0, w)
    c = np.einsum("ni,ni->n", m1, w) + dt(F(0.3))
    gA, gB, gC = gcon[:, 0], gcon[:, 1], gcon[:, 3]
    det = a * c - b * b
    Dn = one / (det * det + dt(F(0.0000001)))
    da = Dn * (-c * c * gA + 2 * b * c * gB + (det - a * c) * gC)
    dc = Dn * (-a * a * gC + 2 * a * b * gB + (det - a * c) * gA)
    db = Dn * 2 * (b * c * gA - (det + 2 * b * b) * gB + a * b * gC)
    dS = np.zeros((n, 6), dt)
    pairs = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    for k, (i, j) in enumerate(pairs):
        if i == j:
            dS[:, k] = m0[:, i] ** 2 * da + m0[:, i] * m1[:, i] * db + m1[:, i] ** 2 * dc
        else:
            dS[:, k] = 2 * m0[:, i] * m0[:, j] * da + (m0[:, i] * m1[:, j] + m0[:, j] * m1[:, i]) * db + 2 * m1[:, i] * m1[:, j] * dc
    dm0 = 2 * da[:, None] * u + db[:, None] * w
    dm1 = 2 * dc[:, None] * w + db[:, None] * u
    dJ00, dJ02, dJ11, dJ12 = dm0 @ W3[0], dm0 @ W3[2], dm1 @ W3[1], dm1 @ W3[2]
    iz = one / tz
    dtx = np.where(cx, dt(0.0), -fx * iz * iz * dJ02)
    dty = np.where(cy, dt(0.0), -fy * iz * iz * dJ12)
    dtz = -(fx * dJ00 + fy * dJ11) * iz * iz + 2 * (fx * tx * dJ02 + fy * ty * dJ12) * iz ** 3
    dmean = np.stack([dtx, dty, dtz], -1) @ W3                                         # R^T dt

    # ---- mean2D -> mean through the perspective divide (A.10)
    hom = lambda row: Pm[row] * p[:, 0] + Pm[4 + row] * p[:, 1] + Pm[8 + row] * p[:, 2] + Pm[12 + row]   # noqa: E731
    hx, hy, hw = hom(0), hom(1), hom(3)
    iw = one / (hw + dt(F(0.0000001)))
    kx, ky = hx * iw * iw, hy * iw * iw
    for j in range(3):
        dmean[:, j] += (Pm[4 * j] * iw - Pm[4 * j + 3] * kx) * g2[:, 0] + (Pm[4 * j + 1] * iw - Pm[4 * j + 3] * ky) * g2[:, 1]

    out = dict(dL_dcov3D=np.where(vis[:, None], dS, dt(0.0)))

    # ---- colour -> SH, and the view direction's share of the mean gradient
    if s.shs is not None:
        sh = s.shs[idx].astype(dt)
        D = s.sh_degree
        cam = s.campos.astype(dt)
        do = p - cam
        ln = np.linalg.norm(do, axis=1, keepdims=True)
        d = do / ln
        x, y, z = d.T
        gr = gcol * (~np.asarray(clamped, bool)[idx])
        dsh = np.zeros_like(sh)
        basis = [np.full(n, C0, dt)]
        ddx = [np.zeros(n, dt)]
        ddy = [np.zeros(n, dt)]
        ddz = [np.zeros(n, dt)]
        if D > 0:
            basis += [-C1 * y, C1 * z, -C1 * x]
            ddx += [np.zeros(n, dt), np.zeros(n, dt), np.full(n, -C1, dt)]
            ddy += [np.full(n, -C1, dt), np.zeros(n, dt), np.zeros(n, dt)]
            ddz += [np.zeros(n, dt), np.full(n, C1, dt), np.zeros(n, dt)]
        if D > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            basis += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
            ddx += [C2[0] * y, 0 * x, C2[2] * (-2 * x), C2[3] * z, C2[4] * 2 * x]
            ddy += [C2[0] * x, C2[1] * z, C2[2] * (-2 * y), 0 * x, C2[4] * (-2 * y)]
            ddz += [0 * x, C2[1] * y, C2[2] * 4 * z, C2[3] * x, 0 * x]
        if D > 2:
            basis += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
                      C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
            ddx += [C3[0] * 6 * xy, C3[1] * yz, C3[2] * (-2) * xy, C3[3] * (-6) * xz, C3[4] * (-3 * xx + 4 * zz - yy), C3[5] * 2 * xz,
                    C3[6] * 3 * (xx - yy)]
            ddy += [C3[0] * 3 * (xx - yy), C3[1] * xz, C3[2] * (-3 * yy + 4 * zz - xx), C3[3] * (-6) * yz, C3[4] * (-2) * xy,
                    C3[5] * (-2) * yz, C3[6] * (-6) * xy]
            ddz += [0 * x, C3[1] * xy, C3[2] * 8 * yz, C3[3] * 3 * (2 * zz - xx - yy), C3[4] * 8 * xz, C3[5] * (xx - yy), 0 * x]
        K = len(basis)
        for k in range(K):
            dsh[:, k, :] = basis[k][:, None] * gr
        dRx = sum(ddx[k][:, None] * sh[:, k, :] for k in range(K))
        dRy = sum(ddy[k][:, None] * sh[:, k, :] for k in range(K))
        dRz = sum(ddz[k][:, None] * sh[:, k, :] for k in range(K))
        ddir = np.stack([(dRx * gr).sum(1), (dRy * gr).sum(1), (dRz * gr).sum(1)], -1)
        # d normalize(v)/dv applied to ddir
        dmean += (ddir - d * (d * ddir).sum(1, keepdims=True)) / ln
        out["dL_dsh"] = np.where(vis[:, None, None], dsh, dt(0.0))
    out["dL_dmean3D"] = np.where(vis[:, None], dmean, dt(0.0))

    # ---- 3D covariance -> scale, rotation (A.10)
    if R is not None:
        G = np.zeros((n, 3, 3), dt)
        G[:, 0, 0], G[:, 1, 1], G[:, 2, 2] = dS[:, 0], dS[:, 3], dS[:, 5]
        G[:, 0, 1] = G[:, 1, 0] = half * dS[:, 1]
        G[:, 0, 2] = G[:, 2, 0] = half * dS[:, 2]
        G[:, 1, 2] = G[:, 2, 1] = half * dS[:, 4]
        cols = np.transpose(R, (0, 2, 1))                       # cols[:, k] = column k of R
        e = two * np.einsum("nij,nkj->nki", G, cols * sc[:, :, None])        # e[:, k] = 2 G a_k
        out["dL_dscale"] = np.where(vis[:, None], np.einsum("nki,nki->nk", cols, e), dt(0.0))
        Q = np.transpose(e * sc[:, :, None], (0, 2, 1))         # Q[:, i, k] = dL/dR_ik (column k = s_k e_k)
        r, x, y, z = s.rotations[idx].astype(dt).T
        q00, q01, q02, q10, q11, q12, q20, q21, q22 = [Q[:, i, j] for i in range(3) for j in range(3)]
        dq = np.stack([2 * z * (q10 - q01) + 2 * y * (q02 - q20) + 2 * x * (q21 - q12),
                       2 * y * (q01 + q10) + 2 * z * (q02 + q20) + 2 * r * (q21 - q12) - 4 * x * (q11 + q22),
                       2 * x * (q01 + q10) + 2 * r * (q02 - q20) + 2 * z * (q12 + q21) - 4 * y * (q00 + q22),
                       2 * r * (q10 - q01) + 2 * x * (q02 + q20) + 2 * y * (q12 + q21) - 4 * z * (q00 + q11)], -1)
        out["dL_drot"] = np.where(vis[:, None], dq, dt(0.0))
    return out
