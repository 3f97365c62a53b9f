"""ctypes front-end for the CPU oracle (oracle/_build/libgsr_oracle.so) and, with the same
struct layout, for the reference build (oracle/_ref/libgsr_ref_{strict,fast}.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libgsr_oracle.so")
REF_SO = {v: os.path.join(_HERE, "_ref", "libgsr_ref_%s.so" % v) for v in ("strict", "fast")}

_fp = C.POINTER(C.c_float)


class OrcInputs(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int),
        ("bg", _fp), ("means3D", _fp), ("shs", _fp), ("colors_precomp", _fp), ("opacities", _fp),
        ("scales", _fp), ("rotations", _fp), ("cov3D_precomp", _fp),
        ("viewmatrix", _fp), ("projmatrix", _fp), ("campos", _fp),
    ]


class OrcState(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("W", C.c_int), ("H", C.c_int), ("gridx", C.c_int), ("gridy", C.c_int),
        ("R", C.c_int64),
        ("depths", _fp), ("clamped", C.POINTER(C.c_uint8)), ("radii", C.POINTER(C.c_int32)),
        ("means2D", _fp), ("cov3D", _fp), ("conic_opacity", _fp), ("rgb", _fp),
        ("tiles_touched", C.POINTER(C.c_uint32)), ("point_offsets", C.POINTER(C.c_uint32)),
        ("keys_unsorted", C.POINTER(C.c_uint64)), ("vals_unsorted", C.POINTER(C.c_uint32)),
        ("keys", C.POINTER(C.c_uint64)), ("vals", C.POINTER(C.c_uint32)),
        ("ranges", C.POINTER(C.c_uint32)), ("final_T", _fp), ("n_contrib", C.POINTER(C.c_uint32)),
        ("out_color", _fp),
        ("visible", C.c_int64), ("consumed_fwd", C.c_int64), ("consumed_bwd", C.c_int64),
    ]


def build_oracle():
    """Compile the C restatement (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return ORACLE_SO


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class Scene(object):
    """Host-side inputs of one rasterizer call, kept alive for the ctypes struct."""

    FIELDS = ("bg", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp",
              "viewmatrix", "projmatrix", "campos")

    def __init__(self, W, H, tanfovx, tanfovy, bg, means3D, opacities, viewmatrix, projmatrix, campos,
                 shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 sh_degree=0, scale_modifier=1.0, prefiltered=False):
        self.W, self.H = int(W), int(H)
        self.tanfovx, self.tanfovy = float(tanfovx), float(tanfovy)
        self.sh_degree, self.scale_modifier, self.prefiltered = int(sh_degree), float(scale_modifier), bool(prefiltered)
        self.bg = _f32(bg).reshape(3)
        self.means3D = _f32(means3D).reshape(-1, 3)
        self.P = self.means3D.shape[0]
        self.opacities = _f32(opacities).reshape(-1)
        self.viewmatrix = _f32(viewmatrix).reshape(16)   # row-major torch memory of H_w2c^T == column-major H_w2c
        self.projmatrix = _f32(projmatrix).reshape(16)
        self.campos = _f32(campos).reshape(3)
        self.shs = None if shs is None else _f32(shs).reshape(self.P, -1, 3)
        self.M = 0 if self.shs is None else self.shs.shape[1]
        self.colors_precomp = None if colors_precomp is None else _f32(colors_precomp).reshape(-1, 3)
        self.scales = None if scales is None else _f32(scales).reshape(-1, 3)
        self.rotations = None if rotations is None else _f32(rotations).reshape(-1, 4)
        self.cov3D_precomp = None if cov3D_precomp is None else _f32(cov3D_precomp).reshape(-1, 6)

    def as_struct(self):
        s = OrcInputs()
        s.P, s.D, s.M, s.W, s.H = self.P, self.sh_degree, self.M, self.W, self.H
        s.tanfovx, s.tanfovy, s.scale_modifier = self.tanfovx, self.tanfovy, self.scale_modifier
        s.prefiltered = int(self.prefiltered)
        for f in self.FIELDS:
            a = getattr(self, f)
            setattr(s, f, None if a is None or a.size == 0 else a.ctypes.data_as(_fp))
        return s


GRAD_SHAPES = (("dL_dmean2D", 3), ("dL_dconic", 4), ("dL_dopacity", 1), ("dL_dcolor", 3), ("dL_dmean3D", 3),
               ("dL_dcov3D", 6), ("dL_dsh", None), ("dL_dscale", 3), ("dL_drot", 4))


class _Lib(object):
    prefix = "orc"

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        p = self.prefix
        fwd = getattr(self.lib, p + "_forward")
        fwd.restype = C.POINTER(OrcState)
        getattr(self.lib, p + "_free").argtypes = [C.POINTER(OrcState)]
        getattr(self.lib, p + "_free").restype = None
        getattr(self.lib, p + "_backward").restype = None
        getattr(self.lib, p + "_mark_visible").restype = None

    # -- helpers
    @staticmethod
    def _copy_state(st, scene):
        P, N, T, R = st.P, st.W * st.H, st.gridx * st.gridy, int(st.R)

        def arr(ptr, n, shape=None):
            if n == 0 or not ptr:
                a = np.zeros((0,), dtype=np.ctypeslib.as_array(ptr, shape=(1,)).dtype if ptr else np.float32)
            else:
                a = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
            return a.reshape(shape) if shape is not None else a

        out = dict(
            P=P, W=st.W, H=st.H, gridx=st.gridx, gridy=st.gridy, R=R,
            visible=int(st.visible), consumed_fwd=int(st.consumed_fwd), consumed_bwd=int(st.consumed_bwd),
            out_color=arr(st.out_color, 3 * N, (3, st.H, st.W)),
            radii=arr(st.radii, P),
        )
        if P:
            out.update(
                depths=arr(st.depths, P), clamped=arr(st.clamped, 3 * P, (P, 3)), means2D=arr(st.means2D, 2 * P, (P, 2)),
                cov3D=arr(st.cov3D, 6 * P, (P, 6)), conic_opacity=arr(st.conic_opacity, 4 * P, (P, 4)),
                rgb=arr(st.rgb, 3 * P, (P, 3)), tiles_touched=arr(st.tiles_touched, P),
                point_offsets=arr(st.point_offsets, P),
                keys_unsorted=arr(st.keys_unsorted, R), vals_unsorted=arr(st.vals_unsorted, R),
                keys=arr(st.keys, R), vals=arr(st.vals, R),
                ranges=arr(st.ranges, 2 * T, (T, 2)), final_T=arr(st.final_T, N, (st.H, st.W)),
                n_contrib=arr(st.n_contrib, N, (st.H, st.W)),
            )
        return out

    def forward(self, scene, keep=False, **kw):
        s = scene.as_struct()
        stp = self._call_forward(s, **kw)
        out = self._copy_state(stp.contents, scene)
        if keep:
            out["_handle"] = stp
            out["_struct"] = s
        else:
            getattr(self.lib, self.prefix + "_free")(stp)
        return out

    def free(self, out):
        h = out.pop("_handle", None)
        if h is not None:
            getattr(self.lib, self.prefix + "_free")(h)

    def forward_backward(self, scene, dL_dpix, exact=False, **kw):
        """Returns (forward dict, grads dict).  exact=True (plain-C oracle only): grads["exact"] holds the render-level sums
        dL_dmean2D [P,3], dL_dconic [P,4], dL_dopacity [P,1], dL_dcolor [P,3] as float64 arrays whose per-(pixel, entry) terms
        were evaluated in double (orc_render_backward_fp64: the arbiter between float32 implementations)."""
        s = scene.as_struct()
        stp = self._call_forward(s, **kw)
        out = self._copy_state(stp.contents, scene)
        P, M = scene.P, scene.M
        g = {}
        for name, c in GRAD_SHAPES:
            shape = (P, M, 3) if c is None else (P, c)
            g[name] = np.zeros(shape, dtype=np.float32)
        dpix = _f32(dL_dpix).reshape(3, scene.H, scene.W)
        self._call_backward(s, stp, dpix, g, **kw)
        if exact:
            x9 = np.zeros((P, 9), np.float64)
            self.lib.orc_render_backward_fp64(C.byref(s), stp, dpix.ctypes.data_as(_fp), x9.ctypes.data_as(C.POINTER(C.c_double)),
                                              C.c_int(kw.get("nthreads", 1)))
            conic = np.zeros((P, 4), np.float64)
            conic[:, [0, 1, 3]] = x9[:, 2:5]
            g["exact"] = dict(dL_dmean2D=np.concatenate([x9[:, 0:2], np.zeros((P, 1))], 1), dL_dconic=conic,
                              dL_dopacity=x9[:, 5:6].copy(), dL_dcolor=x9[:, 6:9].copy())
        getattr(self.lib, self.prefix + "_free")(stp)
        return out, g

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        m = _f32(means3D).reshape(-1, 3)
        v, p = _f32(viewmatrix).reshape(16), _f32(projmatrix).reshape(16)
        pres = np.zeros(m.shape[0], dtype=np.uint8)
        getattr(self.lib, self.prefix + "_mark_visible")(
            C.c_int(m.shape[0]), m.ctypes.data_as(_fp), v.ctypes.data_as(_fp), p.ctypes.data_as(_fp),
            pres.ctypes.data_as(C.POINTER(C.c_uint8)))
        return pres.astype(bool)


class Oracle(_Lib):
    """The plain-C CPU restatement."""
    prefix = "orc"

    def __init__(self, path=None):
        if path is None:
            path = build_oracle()
        super().__init__(path)
        self.lib.orc_get_higher_msb.restype = C.c_uint32
        self.lib.orc_get_higher_msb.argtypes = [C.c_uint32]

    def _call_forward(self, s, nthreads=1):
        return self.lib.orc_forward(C.byref(s), C.c_int(nthreads))

    def _call_backward(self, s, stp, dpix, g, nthreads=1):
        self.lib.orc_backward(C.byref(s), stp, dpix.ctypes.data_as(_fp),
                              *[g[n].ctypes.data_as(_fp) for n, _ in GRAD_SHAPES], C.c_int(nthreads))

    def get_higher_msb(self, n):
        return int(self.lib.orc_get_higher_msb(n))

    def sort_pairs(self, keys, vals, end_bit):
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        v = np.ascontiguousarray(vals, dtype=np.uint32)
        ko, vo = np.empty_like(k), np.empty_like(v)
        self.lib.orc_sort_pairs(C.c_int64(k.size), k.ctypes.data_as(C.POINTER(C.c_uint64)),
                                v.ctypes.data_as(C.POINTER(C.c_uint32)), ko.ctypes.data_as(C.POINTER(C.c_uint64)),
                                vo.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int(end_bit))
        return ko, vo

    def sh_to_rgb(self, deg, mean, campos, sh):
        mean, campos, sh = _f32(mean), _f32(campos), _f32(sh)
        rgb = np.zeros(3, np.float32)
        cl = np.zeros(3, np.uint8)
        self.lib.orc_sh_to_rgb(C.c_int(deg), C.c_int(sh.shape[0]), mean.ctypes.data_as(_fp), campos.ctypes.data_as(_fp),
                               sh.ctypes.data_as(_fp), rgb.ctypes.data_as(_fp), cl.ctypes.data_as(C.POINTER(C.c_uint8)))
        return rgb, cl.astype(bool)


class Reference(_Lib):
    """The reference's own kernels built for gfx950 (oracle/_ref).  Needs a GPU."""
    prefix = "ref"

    def __init__(self, variant="strict"):
        super().__init__(REF_SO[variant])
        self.variant = variant
        self.lib.ref_bench.restype = None

    def _call_forward(self, s):
        return self.lib.ref_forward(C.byref(s))

    def _call_backward(self, s, stp, dpix, g):
        self.lib.ref_backward(C.byref(s), stp, dpix.ctypes.data_as(_fp), *[g[n].ctypes.data_as(_fp) for n, _ in GRAD_SHAPES])

    def bench(self, scene, dL_dpix=None, warmup=2, iters=5):
        s = scene.as_struct()
        dp = None if dL_dpix is None else _f32(dL_dpix)
        f, b = C.c_float(0), C.c_float(0)
        self.lib.ref_bench(C.byref(s), None if dp is None else dp.ctypes.data_as(_fp), C.c_int(warmup), C.c_int(iters),
                           C.byref(f), C.byref(b))
        return f.value, b.value

    @staticmethod
    def available(variant="strict"):
        return os.path.exists(REF_SO[variant])
