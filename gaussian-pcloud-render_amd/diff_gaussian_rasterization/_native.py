"""ctypes binding of libgsr_hip.so (include/gsr.h).

Plays the role of the reference's pybind module ``diff_gaussian_rasterization._C`` (ext.cpp:15-19):
three entry points taking torch tensors.  Tensors are only used for device memory and the
stream; what crosses the boundary are raw pointers and sizes.

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device the
call raises.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB") or os.path.join(_HERE, "libgsr_hip.so")   # (GSR_LIB: another build of the same ABI, for A/B runs)

_fp = C.c_void_p


class GsrParams(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int), ("debug", C.c_int), ("need_backward", C.c_int), ("reference_lists", C.c_int),
        ("bg", _fp), ("means3D", _fp), ("shs", _fp), ("colors_precomp", _fp), ("opacities", _fp),
        ("scales", _fp), ("rotations", _fp), ("cov3D_precomp", _fp),
        ("viewmatrix", _fp), ("projmatrix", _fp), ("campos", _fp),
    ]


# every symbol include/gsr.h declares (tests check the library exports all of them)
SYMBOLS = ("gsr_geom_bytes", "gsr_geom_bytes_inference", "gsr_image_bytes", "gsr_binning_bytes", "gsr_forward_batch", "gsr_forward_stage1",
           "gsr_forward_stage2", "gsr_backward_batch", "gsr_backward", "gsr_mark_visible", "gsr_query", "gsr_set_profiling",
           "gsr_get_profile", "gsr_last_error", "gsr_version", "gsr_selftest", "gsr_forward_recolor", "gsr_forward_batch_channels", "gsr_d2h_count",
           "gsr_clock_probe_launch", "gsr_wall_clock_khz", "gsr_last_list_pairs", "gsr_set_forward_half_views",
           "gsr_set_backward_moments", "gsr_set_sort_mode")

GSR_RETRY = 1

Q = dict(DEPTHS=1, MEANS2D=2, CONIC_OPACITY=3, RGB=4, TILES_TOUCHED=5, POINT_LIST=6, POINT_LIST_KEYS=7, RANGES=8,
         FINAL_T=9, N_CONTRIB=10, CLAMPED=11, TILE_NEED=12, DEPTH_SORT=13, LIST_PAIRS=14)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "diff_gaussian_rasterization: %s not found - build it with "
            "`python gaussian-pcloud-render_amd/build.py` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.gsr_geom_bytes.restype = C.c_size_t
    lib.gsr_geom_bytes.argtypes = [C.c_int]
    lib.gsr_geom_bytes_inference.restype = C.c_size_t
    lib.gsr_geom_bytes_inference.argtypes = [C.c_int]
    lib.gsr_image_bytes.restype = C.c_size_t
    lib.gsr_image_bytes.argtypes = [C.c_int, C.c_int]
    lib.gsr_binning_bytes.restype = C.c_size_t
    lib.gsr_binning_bytes.argtypes = [C.c_int64]
    lib.gsr_forward_batch.restype = C.c_int
    lib.gsr_forward_batch.argtypes = [C.POINTER(GsrParams), C.c_int, _fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t, _fp, _fp,
                                      C.POINTER(C.c_int64), C.c_int, _fp]
    lib.gsr_forward_batch_channels.restype = C.c_int
    lib.gsr_forward_batch_channels.argtypes = [C.POINTER(GsrParams), C.c_int, _fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t,
                                               _fp, _fp, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp]
    lib.gsr_forward_stage1.restype = C.c_int
    lib.gsr_forward_stage1.argtypes = [C.POINTER(GsrParams), _fp, C.c_size_t, _fp, C.c_size_t, _fp,
                                       C.POINTER(C.c_int64), _fp]
    lib.gsr_forward_stage2.restype = C.c_int
    lib.gsr_forward_stage2.argtypes = [C.POINTER(GsrParams), _fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t,
                                       C.c_int64, _fp, _fp]
    lib.gsr_forward_recolor.restype = C.c_int
    lib.gsr_forward_recolor.argtypes = [C.POINTER(GsrParams), C.c_int, C.c_int, _fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t, _fp,
                                        _fp]
    lib.gsr_backward_batch.restype = C.c_int
    lib.gsr_backward_batch.argtypes = [C.POINTER(GsrParams), C.c_int, _fp, _fp, C.c_size_t, _fp, C.c_size_t, _fp,
                                       C.c_size_t] + [_fp] * 9 + [_fp]
    lib.gsr_backward.restype = C.c_int
    lib.gsr_backward.argtypes = [C.POINTER(GsrParams), _fp, C.c_int64, _fp, C.c_size_t, _fp, C.c_size_t, _fp,
                                 C.c_size_t] + [_fp] * 9 + [_fp]
    lib.gsr_mark_visible.restype = C.c_int
    lib.gsr_mark_visible.argtypes = [C.c_int, _fp, _fp, _fp, _fp, _fp]
    lib.gsr_query.restype = C.c_int
    lib.gsr_query.argtypes = [C.POINTER(GsrParams), C.c_int, _fp, _fp, C.c_size_t, _fp, C.c_int64, _fp, C.c_size_t, _fp]
    lib.gsr_set_profiling.restype = None
    lib.gsr_set_profiling.argtypes = [C.c_int]
    lib.gsr_set_forward_half_views.restype = C.c_int
    lib.gsr_set_forward_half_views.argtypes = [C.c_int]
    lib.gsr_set_backward_moments.restype = C.c_int
    lib.gsr_set_backward_moments.argtypes = [C.c_int]
    lib.gsr_set_sort_mode.restype = C.c_int
    lib.gsr_set_sort_mode.argtypes = [C.c_int, C.c_int]
    lib.gsr_get_profile.restype = C.c_int
    lib.gsr_get_profile.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    lib.gsr_selftest.restype = C.c_int
    lib.gsr_selftest.argtypes = [_fp]
    lib.gsr_d2h_count.restype = C.c_longlong
    lib.gsr_d2h_count.argtypes = []
    lib.gsr_clock_probe_launch.restype = C.c_int
    lib.gsr_clock_probe_launch.argtypes = [_fp, C.c_int, _fp]
    lib.gsr_wall_clock_khz.restype = C.c_int
    lib.gsr_wall_clock_khz.argtypes = []
    lib.gsr_last_list_pairs.restype = C.c_int
    lib.gsr_last_list_pairs.argtypes = [C.POINTER(C.c_int64), C.c_int]
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_version.restype = C.c_char_p
    return lib


lib = _load()


def _ptr(t):
    """NULL for absent optionals: the reference passes empty tensors whose data_ptr is nullptr
    (rasterize_points.cu:94-111 -> rasterizer_impl.cu:321,389,411)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib.gsr_last_error().decode("utf-8", "replace"))


def _f32c(t, device, name):
    if t.numel() == 0:
        return t
    if t.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float but found %s for %s" % (t.dtype, name))
    if t.device == device and t.is_contiguous():     # the usual case: nothing to dispatch
        return t
    return t.to(device).contiguous()


# the current stream's handle and "is `device` the current device" without building torch.cuda.Stream / device objects: this
# module's code runs between the caller's last host->device copy and the first kernel launch, i.e. while the GPU idles whenever
# the caller's loop drains the stream once per frame (the reference's does: simple_raw_render.py:260-263 builds the settings
# tensors from host arrays for every call)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_handle(device):
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class _NoContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    @staticmethod
    def give(*tensors):
        return tensors[0] if len(tensors) == 1 else tensors


_NO_CONTEXT = _NoContext()


def _on_device(device):
    """context that makes `device` the current HIP device (nothing to do when it already is)"""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_CONTEXT
    return torch.cuda.device(device)


def _require_hip(device):
    if device.type != "cuda":
        raise RuntimeError(
            "diff_gaussian_rasterization (MI355X build) needs tensors on a HIP device (torch device 'cuda'); got %s. "
            "There is no CPU path." % device)


def _params(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
            tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, debug, need_backward):
    device = means3D.device
    # (the converted tensors are returned so that they outlive the enqueued kernels' use of their pointers)
    keep = (_f32c(bg, device, "bg"), _f32c(means3D, device, "means3D"), _f32c(sh, device, "sh"),
            _f32c(colors, device, "colors_precomp"), _f32c(opacity, device, "opacities"), _f32c(scales, device, "scales"),
            _f32c(rotations, device, "rotations"), _f32c(cov3D_precomp, device, "cov3D_precomp"),
            _f32c(viewmatrix, device, "viewmatrix"), _f32c(projmatrix, device, "projmatrix"), _f32c(campos, device, "campos"))
    M = int(sh.shape[1]) if sh.numel() != 0 and sh.shape[0] != 0 else 0  # rasterize_points.cu:83-87
    # field order of GsrParams / gsr_params (include/gsr.h)
    p = GsrParams(means3D.shape[0], int(degree), M, int(W), int(H), float(tan_fovx), float(tan_fovy), float(scale_modifier),
                  int(bool(prefiltered)), int(bool(debug)), int(bool(need_backward)), int(reference_lists()),
                  *[_ptr(t) for t in keep])
    return p, keep


# ---- footprint clipping (include/gsr.h, gsr_params.reference_lists) ---------------------------------------------------
# Default: a Gaussian emits pairs only for the tiles of the reference's rectangle in which it can reach alpha >= 1/255; every
# API-visible result is unchanged.  set_reference_lists(True) / GSR_REFERENCE_LISTS=1 keeps the reference's full rectangles, so
# that the private lists, ranges and n_contrib can be compared with the reference's element by element (parity tests).
_REFERENCE_LISTS = [os.environ.get("GSR_REFERENCE_LISTS", "0") not in ("", "0")]   # process default
_TLS = threading.local()                                                              # per-thread override (tests render from several threads)


def reference_lists():
    return getattr(_TLS, "reference_lists", _REFERENCE_LISTS[0])


def set_reference_lists(on):
    """Setting of the CALLING THREAD (None: back to the process default).  Returns the thread's previous override (or None)."""
    old = getattr(_TLS, "reference_lists", None)
    if on is None:
        if hasattr(_TLS, "reference_lists"):
            del _TLS.reference_lists
    else:
        _TLS.reference_lists = bool(on)
    return old


# ---- binning-arena capacity -----------------------------------------------------------------------------------------
# The pair count of a frame (num_rendered) is only known on the device while the frame is being enqueued, so the binning
# arena is allocated by CAPACITY: the largest count this (device, P, W, H) configuration produced recently, plus slack.
# A frame that needs more gets GSR_RETRY and repeats its binning half with an exact-size arena; the first frame of a
# configuration counts first (stage 1) and then binds (stage 2), like the reference's synchronous flow.
_CAP_HINT = {}
CAP_SLACK = 1.25


def _cap_key(device, P, W, H):
    return (device.index if device.index is not None else torch.cuda.current_device(), int(P), int(W), int(H))


def _note_counts(key, counts):
    old = _CAP_HINT.get(key, 0)
    _CAP_HINT[key] = max(max(counts), int(old * 0.98))   # shrinks slowly when the frames get lighter


def reset_capacity_hints():
    _CAP_HINT.clear()


# arena sizes are functions of (P, need_backward) resp. (W, H) alone: asked of the library once per configuration
_ARENA_BYTES, _IMAGE_BYTES = {}, {}


def _arena_bytes(P, need_backward):
    k = (P, bool(need_backward))
    b = _ARENA_BYTES.get(k)
    if b is None:
        if len(_ARENA_BYTES) > 256:
            _ARENA_BYTES.clear()
        b = _ARENA_BYTES[k] = int(lib.gsr_geom_bytes(P) if need_backward else lib.gsr_geom_bytes_inference(P))
    return b


def _image_bytes(W, H):
    k = (W, H)
    b = _IMAGE_BYTES.get(k)
    if b is None:
        if len(_IMAGE_BYTES) > 256:
            _IMAGE_BYTES.clear()
        b = _IMAGE_BYTES[k] = int(lib.gsr_image_bytes(W, H))
    return b



# ---- consecutive calls overlap on the device (no caller threads) ---------------------------------------------------------
# A per-view caller issues forward(j), backward(j), forward(j+1), ... on ONE stream, so the device runs them back to back: the
# tail of a single view's render kernels (a handful of deep list walks on an otherwise idle chip) and the bandwidth-bound front
# end of the next view never share the GPU.  Four caller threads on four streams get 1.5x out of that overlap (bench.py,
# drop_in_api.four_streams); the same overlap is available to ONE caller when the library can prove that the next forward does
# not depend on the work still in flight:
#   * every forward runs on one of a few internal side streams (rotating), the caller's stream waits (on the device) for it at
#     the end of the call, so everything the caller enqueues afterwards is ordered behind the results as before;
#   * the side stream has to wait for the INPUTS only.  An input tensor seen before -- the same Python tensor object with the
#     same torch version counter, i.e. not written through torch since -- was complete when the call that first saw it
#     began; if that holds for every input, the side stream waits for the event recorded then, not for the caller stream's
#     tail: forward(j+1)'s front end runs beside backward(j).  A new tensor, a bumped version counter (optimizer step, any
#     in-place op), a tensor without a counter (inference mode): the side stream waits for the caller's stream as it is
#     now -- exactly the unoverlapped order.
# Tensors are remembered by weak reference (no lifetime is extended; a recycled address can not impersonate an old tensor).
# What the version counter does not see -- `t.data.copy_(...)`, a foreign kernel writing through `data_ptr()` -- is not seen here
# either: autograd's own in-place checks have the same blind spot (a swapped storage, `t.data = other`, IS seen: the record
# holds data_ptr / offset / shape / strides).  Off by default (below); when on, set_overlap(False) restores plain in-stream execution.  All outputs and arenas are allocated on the side stream and handed to the caller's stream with
# record_stream, so torch's caching allocator never recycles them under a kernel that still runs.
import weakref as _weakref

# OPT-IN since round 5 (GSR_OVERLAP=1 or set_overlap(True)): what the overlap is worth depends on which hardware queues the runtime
# binds the side streams to at their first launch -- 1 250-1 360 frames/s in most processes, 1 010-1 160 in some, in-order 985 --
# which a process can neither see nor choose, and the reference's literal caller (fresh settings tensors per call,
# simple_raw_render.py:260-263) never qualifies for it anyway: the default is plain stream order, the same in every process.
# Each side stream wants a hardware queue of its own (the ROCm runtime maps HIP streams onto GPU_MAX_HW_QUEUES queues, default 4,
# and streams that share a queue serialise); the variable is read when HIP starts and changes the stream-to-queue mapping of the
# whole process, so the library only touches it when asked to (GSR_SET_HW_QUEUES=1, before the process first uses the GPU).
if os.environ.get("GSR_SET_HW_QUEUES", "0") == "1" and not torch.cuda.is_initialized():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_OVERLAP_ON = os.environ.get("GSR_OVERLAP", "0") == "1"
_OVERLAP_STREAMS = int(os.environ.get("GSR_OVERLAP_STREAMS", "3"))
_OVERLAP_PRIORITY = int(os.environ.get("GSR_OVERLAP_PRIORITY", "-1"))    # -1: side streams above the caller's stream
OVERLAP_MAX_VIEWS = 3
_OVERLAP = {}            # (device index, caller stream) -> _OverlapState
OVERLAP_STATS = dict(calls=0, overlapped=0)


def set_overlap(on):
    """Switch the device-side overlap of consecutive forward calls on or off (off: every kernel on the caller's stream)."""
    global _OVERLAP_ON
    _OVERLAP_ON = bool(on)      # (the side streams are kept: the runtime maps streams to hardware queues in creation order, and a
                                # later set of streams may land on the caller's queue -- 1 150 instead of 1 290 frames/s,
                                # scripts/debug/overlap_warm.py)


class _OverlapState:
    def __init__(self, device, cur):
        # side streams above the caller's priority: a view's front end is short and on the critical path of the next frame, the
        # previous view's backward is not.  What the overlap is worth depends on which hardware queues / dispatch pipes the
        # runtime binds these streams to at their first launch, which a process can neither see nor choose: 1 250-1 360 frames/s
        # in most processes, 1 010-1 060 in some (in-order: 985), stable for the life of a process (gpurun_out/r4q*).  Choosing
        # streams with a concurrency test of small spin kernels made it WORSE (1 000-1 140 in every process: the binding seems
        # to follow what is in flight at the first launch), so the streams are taken as they come.
        self.streams = [torch.cuda.Stream(device=device, priority=_OVERLAP_PRIORITY) for _ in range(_OVERLAP_STREAMS)]
        self.turn = 0
        self.seen = {}       # id(tensor) -> (weakref, version, event, sequence number, storage key)
        self.seq = 0

    @staticmethod
    def _where(t):
        """What the tensor object points at: `param.data = other` (weight clamping, checkpoint loading, densification code) swaps
        the storage under the same Python object WITHOUT bumping the version counter, so object identity + version is not enough."""
        return (t.data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype)

    def input_event(self, tensors, cur):
        """The event the side stream has to wait for: the newest `first seen` event when every input is a known, unmodified
        tensor, else an event recorded on the caller's stream now (under which the unknown inputs are registered)."""
        self.seq += 1
        newest, missing = None, []
        for t in tensors:
            try:
                ver = t._version
            except RuntimeError:      # inference tensor: no version counter, never assumed unchanged
                return self._now(cur, []), False
            rec = self.seen.get(id(t))
            if rec is not None and rec[0]() is t and rec[1] == ver and rec[4] == self._where(t):
                if newest is None or rec[3] > newest[3]:
                    newest = rec
            else:
                missing.append((t, ver))
        if missing or newest is None:
            return self._now(cur, missing), False
        return newest[2], True

    def _now(self, cur, missing):
        ev = torch.cuda.Event()
        ev.record(cur)
        if len(self.seen) > 512:
            self.seen = {k: r for k, r in self.seen.items() if r[0]() is not None}
            if len(self.seen) > 512:
                self.seen.clear()
        for t, ver in missing:
            self.seen[id(t)] = (_weakref.ref(t), ver, ev, self.seq, self._where(t))
        return ev

    def next_stream(self):
        self.turn = (self.turn + 1) % len(self.streams)
        return self.streams[self.turn]


class _OnSideStream:
    """Context of one forward call: inside, torch's current stream is a side stream that waits only for the call's inputs;
    leaving it, the caller's stream is ordered behind the side stream and takes over the outputs handed to `give`."""

    def __init__(self, device, tensors):
        self.device, self.active, self.out = device, False, []
        if device.type != "cuda":
            return
        self.cur = torch.cuda.current_stream(device)
        key = (device.index if device.index is not None else torch.cuda.current_device(), self.cur.cuda_stream)
        st = _OVERLAP.get(key)
        if st is None:
            st = _OVERLAP[key] = _OverlapState(device, self.cur)
        live = [t for t in tensors if isinstance(t, torch.Tensor) and t.numel() != 0 and t.device.type == "cuda"]
        ev, early = st.input_event(live, self.cur)
        OVERLAP_STATS["calls"] += 1
        OVERLAP_STATS["overlapped"] += int(early)
        self.side = st.next_stream()
        self.side.wait_event(ev)
        self.active = True

    def give(self, *tensors):
        self.out += [t for t in tensors if isinstance(t, torch.Tensor) and t.numel() != 0]
        return tensors[0] if len(tensors) == 1 else tensors

    def __enter__(self):
        if self.active:
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.ctx.__exit__(*exc)
            done = torch.cuda.Event()
            done.record(self.side)
            self.cur.wait_event(done)
            for t in self.out:
                t.record_stream(self.cur)
        return False


def rasterize_gaussians_batch(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                              viewmatrices, projmatrices, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                              camposs, prefiltered, debug, need_backward=True, capacity=None, extra=None):
    """V views of one cloud in one submission (C ABI gsr_forward_batch): viewmatrices / projmatrices [V,4,4] (transposed like
    the reference's settings), camposs [V,3].  Returns (num_rendered list[V], out_color [V,3,H,W], radii [V,P],
    geomBuffer, binningBuffer, imgBuffer).  `capacity` (pairs per view) overrides the remembered arena capacity.
    extra = (values [P,nx] shared by the views or [V,P,nx] per view -- or the pair ([P,4] shared, [V,P,4] per view) for eight channels
    of which only the last four depend on the view --, view_scale [V,nx] or None, bg [nx]) with nx in (4, 8): the
    render also composites those channels with the colour's alphas (gsr_forward_batch_channels) and the result gains a 7th
    element, out_extra [V,nx,H,W]."""
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    device = means3D.device
    _require_hip(device)
    V = viewmatrices.numel() // 16
    if viewmatrices.numel() != 16 * V or projmatrices.numel() != 16 * V or camposs.numel() != 3 * V or V < 1:
        raise RuntimeError("viewmatrices, projmatrices and camposs must describe the same number of views (>= 1)")
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    byte = dict(dtype=torch.uint8, device=device)
    nx = 0
    xv = xs = xb = None
    if extra is not None:
        xv, xs, xb = extra
        if isinstance(xv, (tuple, list)):
            # split layout (gsr.h extra_per_view = 2): channels 0..3 shared by the views [P,4], channels 4..7 per view [V,P,4]
            lo, hi = xv
            if tuple(lo.shape) != (P, 4) or tuple(hi.shape) != (V, P, 4):
                raise RuntimeError("split extra channels must have shapes (num_points, 4) and (num_views, num_points, 4)")
            xv = torch.cat([_f32c(lo, device, "extra").reshape(-1), _f32c(hi, device, "extra").reshape(-1)])
            x_per_view, nx = 2, 8
        else:
            x_per_view = int(xv.dim() == 3)
            nx = int(xv.shape[-1]) if xv.dim() in (2, 3) else -1
            if nx not in (4, 8) or xv.shape[-2] != P or (x_per_view and xv.shape[0] != V):
                raise RuntimeError("extra channels must have shape (num_points, 4 or 8) or (num_views, num_points, 4 or 8)")
        if xb.numel() != nx or (xs is not None and tuple(xs.shape) != (V, nx)):
            raise RuntimeError("bg_extra must have nx entries and view_scale shape (V, nx)")
    if P == 0:  # rasterize_points.cu:81: the zero image (not the background) is returned
        e = torch.empty((0,), **byte)
        r0 = ([0] * V, torch.zeros((V, 3, H, W), dtype=torch.float32, device=device),
              torch.zeros((V, 0), dtype=torch.int32, device=device), e, e.clone(), e.clone())
        return r0 if extra is None else r0 + (torch.zeros((V, nx, H, W), dtype=torch.float32, device=device),)
    key = _cap_key(device, P, W, H)
    # everything below -- layout conversions of the inputs, the allocation of outputs and arenas, every kernel -- happens on a side
    # stream that waits for the inputs only (see _OnSideStream); the caller's stream takes the results over when the block ends
    # (only calls of up to OVERLAP_MAX_VIEWS views: a 12-view batch has no tail for the next call's front end to hide behind, and
    # running the two side by side costs 2.5 % on short bursts -- gpurun_out/r4k: 1 754 vs 1 711 frames/s at 20 frames per block)
    if _OVERLAP_ON and V <= OVERLAP_MAX_VIEWS:
        ov = _OnSideStream(device, [background, means3D, colors, opacity, scales, rotations, cov3D_precomp, viewmatrices, projmatrices,
                                    sh, camposs, xv, xs, xb])
    else:
        ov = _NO_CONTEXT
    with _on_device(device), ov:
        if nx:
            xv = _f32c(xv, device, "extra")
            xb = _f32c(xb.reshape(-1), device, "bg_extra")
            xs = None if xs is None else _f32c(xs, device, "extra_view_scale")
        f32 = dict(dtype=torch.float32, device=device)
        out_extra = torch.empty((V, nx, H, W), **f32) if nx else None
        # every pixel and every radius is written by the kernels (the reference fills both with zeros first)
        out_color = torch.empty((V, 3, H, W), **f32)
        radii = torch.empty((V, P), dtype=torch.int32, device=device)
        stream = _stream_handle(device)
        # (the matrices go over as pointers: [V,4,4] / [V,16] / [4,4] are the same 16 V floats once contiguous)
        p, keep = _params(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                          viewmatrices, projmatrices, tan_fovx, tan_fovy, H, W, sh, degree, camposs, prefiltered, debug,
                          need_backward)
        # the 64-B per-Gaussian gradient records are only carved for calls a backward may follow
        geom = torch.empty((V * _arena_bytes(P, need_backward),), **byte)
        img = torch.empty((V * _image_bytes(W, H),), **byte)
        counts = (C.c_int64 * V)()
        if capacity is None:
            hint = _CAP_HINT.get(key)
            capacity = None if hint is None else int(hint * CAP_SLACK) + 4096
        if capacity is None and V == 1 and not nx:
            # first frame of this configuration: count, then bind (one host round trip, like the reference)
            _check(lib.gsr_forward_stage1(C.byref(p), geom.data_ptr(), geom.numel(), img.data_ptr(), img.numel(),
                                          radii.data_ptr(), counts, stream))
            binning = torch.empty((lib.gsr_binning_bytes(int(counts[0] * CAP_SLACK) + 4096),), **byte)
            _check(lib.gsr_forward_stage2(C.byref(p), geom.data_ptr(), geom.numel(), binning.data_ptr(), binning.numel(),
                                          img.data_ptr(), img.numel(), counts[0], out_color.data_ptr(), stream))
        else:
            if capacity is None:
                capacity = 16 * P      # first batch of this configuration: a guess, corrected by the retry below
            def submit(binning, resume):
                if nx:
                    return lib.gsr_forward_batch_channels(
                        C.byref(p), V, geom.data_ptr(), geom.numel(), img.data_ptr(), img.numel(), binning.data_ptr(),
                        binning.numel(), radii.data_ptr(), out_color.data_ptr(), counts, resume, nx, x_per_view, xv.data_ptr(),
                        None if xs is None else xs.data_ptr(), xb.data_ptr(), out_extra.data_ptr(), stream)
                return lib.gsr_forward_batch(C.byref(p), V, geom.data_ptr(), geom.numel(), img.data_ptr(), img.numel(),
                                             binning.data_ptr(), binning.numel(), radii.data_ptr(), out_color.data_ptr(), counts,
                                             resume, stream)

            binning = torch.empty((V * lib.gsr_binning_bytes(int(capacity)),), **byte)
            rc = submit(binning, 0)
            if rc == GSR_RETRY:
                pairs = (C.c_int64 * V)()
                _check(lib.gsr_last_list_pairs(pairs, V))
                need = int(max(pairs) * CAP_SLACK) + 4096
                binning = torch.empty((V * lib.gsr_binning_bytes(need),), **byte)
                rc = submit(binning, 1)
            _check(rc)
        ov.give(out_color, radii, geom, binning, img, out_extra)
        pairs = (C.c_int64 * V)()
        _check(lib.gsr_last_list_pairs(pairs, V))   # (same host thread as the forward call: the count is kept per thread)
    del keep
    counts = [int(c) for c in counts]
    _note_counts(key, [int(c) for c in pairs])      # the arena holds the lists: sized by their pairs, not by num_rendered
    if nx:
        return counts, out_color, radii, geom, binning, img, out_extra
    return counts, out_color, radii, geom, binning, img


# ---- the per-view call's short path ----------------------------------------------------------------------------------------
# The reference's caller builds its settings with blocking host->device copies (simple_raw_render.py:79-112), which drain the
# stream once per frame: everything the host does between the last copy and this call's first kernel is time the GPU idles in.
# forward_view / backward_view are rasterize_gaussians_batch / _backward_batch cut down to what ONE view in steady state needs
# before the launch: inline argument checks, the three scratch arenas as ONE allocation (they live and die together: saved for
# the backward as one tensor), outputs in their final shape (no views), sizes from dictionaries.  Anything unusual -- first frame
# of a configuration, debug, P == 0, the overlap mode -- returns None and the caller takes the general path.
_F32 = torch.float32
_BIN_BYTES = {}
_CAP_QUANTUM = 1 << 16
_TLS_FAST = threading.local()


def _slab_layout(P, W, H, need_backward, capacity):
    g = (_arena_bytes(P, need_backward) + 255) & ~255
    i = (_image_bytes(W, H) + 255) & ~255
    b = _BIN_BYTES.get(capacity)
    if b is None:
        if len(_BIN_BYTES) > 1024:
            _BIN_BYTES.clear()
        b = _BIN_BYTES[capacity] = int(lib.gsr_binning_bytes(capacity))
    return g, i, b


def forward_view(rs, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, need_backward):
    """One view through gsr_forward_batch.  Returns (num_rendered, color [3,H,W], radii [P], arenas, layout) -- arenas: ONE uint8
    tensor holding the geometry, image and binning arenas at the byte offsets 0, layout[0], layout[0] + layout[1] -- or None when
    the call has to take the general path."""
    device = means3D.device
    P = means3D.shape[0]
    if _OVERLAP_ON or device.type != "cuda" or means3D.dim() != 2 or P == 0:
        return None
    H, W = int(rs.image_height), int(rs.image_width)
    hint = _CAP_HINT.get((device.index, P, W, H))
    if hint is None or device.index is None or device.index != torch.cuda.current_device():
        return None
    keep = []
    ptrs = []
    for t in (rs.bg, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, rs.viewmatrix, rs.projmatrix, rs.campos):
        if t.numel() == 0:
            ptrs.append(None)
            continue
        if t.dtype is not _F32 or t.device != device or not t.is_contiguous():
            t = _f32c(t, device, "an input")
            keep.append(t)
        ptrs.append(t.data_ptr())
    M = int(sh.shape[1]) if ptrs[2] is not None else 0
    p = GsrParams(P, int(rs.sh_degree), M, W, H, float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
                  int(bool(rs.prefiltered)), 0, int(need_backward), int(reference_lists()), *ptrs)
    capacity = (int(hint * CAP_SLACK) + 4096 + _CAP_QUANTUM - 1) // _CAP_QUANTUM * _CAP_QUANTUM
    g, i, b = _slab_layout(P, W, H, need_backward, capacity)
    arenas = torch.empty((g + i + b,), dtype=torch.uint8, device=device)
    color = torch.empty((3, H, W), dtype=_F32, device=device)
    radii = torch.empty((P,), dtype=torch.int32, device=device)
    counts = getattr(_TLS_FAST, "counts", None)
    if counts is None:
        counts = _TLS_FAST.counts = (C.c_int64 * 1)()
        _TLS_FAST.pairs = (C.c_int64 * 1)()
    base = arenas.data_ptr()
    stream = _raw_stream(device.index) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream
    rc = lib.gsr_forward_batch(C.byref(p), 1, base, g, base + g, i, base + g + i, b, radii.data_ptr(), color.data_ptr(), counts, 0, stream)
    pairs = _TLS_FAST.pairs
    if rc == GSR_RETRY:
        _check(lib.gsr_last_list_pairs(pairs, 1))
        capacity = (int(pairs[0] * CAP_SLACK) + 4096 + _CAP_QUANTUM - 1) // _CAP_QUANTUM * _CAP_QUANTUM
        g2, i2, b2 = _slab_layout(P, W, H, need_backward, capacity)
        grown = torch.empty((g + i + b2,), dtype=torch.uint8, device=device)
        # the geometry / image halves of the frame are kept: device copy in stream order, then only the binning half is repeated
        grown[:g + i].copy_(arenas[:g + i])
        arenas, b, base = grown, b2, grown.data_ptr()
        rc = lib.gsr_forward_batch(C.byref(p), 1, base, g, base + g, i, base + g + i, b, radii.data_ptr(), color.data_ptr(), counts, 1, stream)
    if rc != 0:
        raise RuntimeError(lib.gsr_last_error().decode("utf-8", "replace"))
    _check(lib.gsr_last_list_pairs(pairs, 1))
    _note_counts((device.index, P, W, H), (int(pairs[0]),))
    return int(counts[0]), color, radii, arenas, (g, i, b)


def backward_view(rs, means3D, radii, colors, scales, rotations, cov3D_precomp, grad_color, sh, arenas, layout):
    """Backward of forward_view (gsr_backward_batch, V = 1): the reference's 8-tuple."""
    device = means3D.device
    P = means3D.shape[0]
    H, W = int(grad_color.shape[-2]), int(grad_color.shape[-1])
    keep = []
    ptrs = []
    for t in (rs.bg, means3D, sh, colors, means3D, scales, rotations, cov3D_precomp, rs.viewmatrix, rs.projmatrix, rs.campos):
        if t.numel() == 0:       # (opacity lives in the geometry arena; the pointer only has to be non-NULL: means3D stands in)
            ptrs.append(None)
            continue
        if t.dtype is not _F32 or t.device != device or not t.is_contiguous():
            t = _f32c(t, device, "an input")
            keep.append(t)
        ptrs.append(t.data_ptr())
    M = int(sh.shape[1]) if ptrs[2] is not None else 0
    p = GsrParams(P, int(rs.sh_degree), M, W, H, float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier), 0, 0, 1,
                  int(reference_lists()), *ptrs)
    if grad_color.dtype is not _F32 or grad_color.device != device or not grad_color.is_contiguous():
        grad_color = _f32c(grad_color, device, "dL_dout_color")
    z = dict(dtype=_F32, device=device)
    has_sr = ptrs[5] is not None
    g_means2D, g_colors, g_opacity = torch.empty((P, 3), **z), torch.empty((P, 3), **z), torch.empty((P, 1), **z)
    g_means3D, g_cov3D, g_sh = torch.empty((P, 3), **z), torch.empty((P, 6), **z), torch.empty((P, M, 3), **z)
    g_scales = torch.empty((P, 3), **z) if has_sr else torch.zeros((P, 3), **z)
    g_rot = torch.empty((P, 4), **z) if has_sr else torch.zeros((P, 4), **z)
    g, i, b = layout
    base = arenas.data_ptr()
    with _on_device(device):
        stream = _raw_stream(device.index) if _raw_stream is not None and device.index is not None else torch.cuda.current_stream(device).cuda_stream
        rc = lib.gsr_backward_batch(C.byref(p), 1, radii.data_ptr(), base, g, base + g + i, b, base + g, i, grad_color.data_ptr(),
                                    g_means2D.data_ptr(), g_opacity.data_ptr(), g_colors.data_ptr(), g_means3D.data_ptr(),
                                    g_cov3D.data_ptr(), g_sh.data_ptr() if M else None, g_scales.data_ptr(), g_rot.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError(lib.gsr_last_error().decode("utf-8", "replace"))
    return g_means2D, g_colors, g_opacity, g_means3D, g_cov3D, g_sh, g_scales, g_rot


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, need_backward=True):
    """Counterpart of RasterizeGaussiansCUDA (rasterize_points.cu:35-115); same argument order, same
    6-tuple result (num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer)."""
    counts, out_color, radii, geom, binning, img = rasterize_gaussians_batch(
        background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
        tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug, need_backward=need_backward)
    return counts[0], out_color[0], radii[0], geom, binning, img


def recolor(background, means3D, colors, sh, degree, campos, image_height, image_width, num_rendered, geomBuffer,
            binningBuffer, imgBuffer, debug=False):
    """Re-render a finished forward's view(s) with other per-Gaussian colours (exactly one of `colors` / `sh` [P,M,3]
    non-empty), reusing its geometry, sorted lists and ranges (gsr_forward_recolor).  campos [3] -> color [3,H,W];
    campos [V,3] (a batch forward's arenas) -> [V,3,H,W], with `colors` [P,3] shared by the views or [V,P,3] per view."""
    device = means3D.device
    _require_hip(device)
    P, H, W = means3D.shape[0], int(image_height), int(image_width)
    single = campos.numel() == 3
    V = 1 if single else campos.reshape(-1, 3).shape[0]
    out_color = (torch.empty if P != 0 else torch.zeros)((V, 3, H, W), dtype=torch.float32, device=device)   # every pixel is written
    if P != 0:
        with torch.cuda.device(device):
            e = torch.empty(0)
            p, keep = _params(background, means3D, colors, torch.empty((1,), device=device), e, e, 1.0, e,
                              torch.empty((1,), device=device), torch.empty((1,), device=device), 1.0, 1.0, H, W, sh, degree,
                              campos, False, debug, False)
            per_view = int(colors.numel() != 0 and colors.dim() == 3)
            if per_view and tuple(colors.shape) != (V, P, 3):
                raise RuntimeError("recolor: per-view colours must have shape (V, P, 3)")
            _check(lib.gsr_forward_recolor(C.byref(p), V, per_view, geomBuffer.data_ptr(), geomBuffer.numel(), binningBuffer.data_ptr(),
                                           binningBuffer.numel(), imgBuffer.data_ptr(), imgBuffer.numel(),
                                           out_color.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
            del keep
    return out_color[0] if single else out_color


def rasterize_gaussians_backward_batch(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                       viewmatrices, projmatrices, tan_fovx, tan_fovy, dL_dout_color, sh, degree, camposs,
                                       geomBuffer, binningBuffer, imageBuffer, debug, _alloc=None):
    """Backward of rasterize_gaussians_batch: dL_dout_color [V,3,H,W], radii [V,P]; gradients summed over the views.
    Returns the reference's 8-tuple (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
    dL_drotations).  _alloc(shape, dtype=, device=): allocator of the outputs (tests hand over float-aligned views to
    exercise the C ABI's alignment fallback; default torch.empty)."""
    device = means3D.device
    _require_hip(device)
    P = means3D.shape[0]
    V, H, W = int(dL_dout_color.shape[0]), int(dL_dout_color.shape[2]), int(dL_dout_color.shape[3])
    M = int(sh.shape[1]) if sh.numel() != 0 and sh.shape[0] != 0 else 0
    z = dict(dtype=torch.float32, device=device)
    # Nothing is cleared here: the per-Gaussian backward kernel writes every output for every Gaussian (the reference
    # zero-fills nine tensors, rasterize_points.cu:151-159), and the accumulation records live in the geometry arena.
    has_sr = scales.numel() != 0 and P != 0
    e_or_z = (_alloc or torch.empty) if P != 0 else torch.zeros
    dL_dmeans2D = e_or_z((P, 3), **z)
    dL_dcolors = e_or_z((P, 3), **z)
    dL_dopacity = e_or_z((P, 1), **z)
    dL_dmeans3D = e_or_z((P, 3), **z)
    dL_dcov3D = e_or_z((P, 6), **z)
    dL_dsh = e_or_z((P, M, 3), **z)
    dL_dscales = e_or_z((P, 3), **z) if has_sr else torch.zeros((P, 3), **z)
    dL_drotations = e_or_z((P, 4), **z) if has_sr else torch.zeros((P, 4), **z)
    if P != 0:
        with _on_device(device):
            stream = _stream_handle(device)
            # (opacity lives in the geom arena; the pointer only has to be non-NULL: means3D stands in)
            p, keep = _params(background, means3D, colors, means3D, scales, rotations, scale_modifier,
                              cov3D_precomp, viewmatrices, projmatrices, tan_fovx, tan_fovy, H, W, sh, degree, camposs, False,
                              debug, True)
            dpix = _f32c(dL_dout_color, device, "dL_dout_color")
            radii_c = radii.contiguous()
            _check(lib.gsr_backward_batch(C.byref(p), V, radii_c.data_ptr(), geomBuffer.data_ptr(), geomBuffer.numel(),
                                          binningBuffer.data_ptr(), binningBuffer.numel(), imageBuffer.data_ptr(),
                                          imageBuffer.numel(), dpix.data_ptr(), dL_dmeans2D.data_ptr(),
                                          dL_dopacity.data_ptr(), dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(),
                                          dL_dcov3D.data_ptr(), _ptr(dL_dsh), dL_dscales.data_ptr(), dL_drotations.data_ptr(),
                                          stream))
            del keep
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug):
    """Counterpart of RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-196); same argument order,
    same 8-tuple (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    return rasterize_gaussians_backward_batch(
        background, means3D, radii.reshape(1, -1), colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
        projmatrix, tan_fovx, tan_fovy, dL_dout_color.reshape((1,) + tuple(dL_dout_color.shape[-3:])), sh, degree, campos,
        geomBuffer, binningBuffer, imageBuffer, debug)


def mark_visible(means3D, viewmatrix, projmatrix):
    """Counterpart of markVisible (rasterize_points.cu:198-217)."""
    device = means3D.device
    _require_hip(device)
    P = means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.bool, device=device)
    if P != 0:
        with torch.cuda.device(device):
            m = _f32c(means3D, device, "means3D")
            v = _f32c(viewmatrix, device, "viewmatrix")
            pr = _f32c(projmatrix, device, "projmatrix")
            _check(lib.gsr_mark_visible(P, m.data_ptr(), v.data_ptr(), pr.data_ptr(), present.data_ptr(),
                                        torch.cuda.current_stream(device).cuda_stream))
    return present


# ---- inspection helpers for tests / bench (not part of the reference API) -------------------------------
_QSPEC = {
    "DEPTHS": (torch.float32, lambda P, R, T, N: (P,)), "MEANS2D": (torch.float32, lambda P, R, T, N: (P, 2)),
    "CONIC_OPACITY": (torch.float32, lambda P, R, T, N: (P, 4)), "RGB": (torch.float32, lambda P, R, T, N: (P, 3)),
    "TILES_TOUCHED": (torch.int32, lambda P, R, T, N: (P,)), "POINT_LIST": (torch.int32, lambda P, R, T, N: (R,)),
    "POINT_LIST_KEYS": (torch.int64, lambda P, R, T, N: (R,)), "RANGES": (torch.int32, lambda P, R, T, N: (T, 2)),
    "FINAL_T": (torch.float32, lambda P, R, T, N: (N,)), "N_CONTRIB": (torch.int32, lambda P, R, T, N: (N,)),
    "CLAMPED": (torch.uint8, lambda P, R, T, N: (P, 3)), "TILE_NEED": (torch.int32, lambda P, R, T, N: (T,)),
    "DEPTH_SORT": (torch.int32, lambda P, R, T, N: (4,)), "LIST_PAIRS": (torch.int64, lambda P, R, T, N: (2,)),
}


def query(name, P, W, H, R, geom, binning, img, view=0, n_views=1):
    """Copy one private arena array of `view` out of the arenas of an `n_views` batch (device tensor).  Unsigned data come
    back in same-width signed dtypes."""
    if name in ("POINT_LIST", "POINT_LIST_KEYS"):
        # the lists hold LIST_PAIRS[0] pairs: R (the reference's num_rendered) with reference_lists, fewer with footprint clipping
        R = int(query("LIST_PAIRS", P, W, H, R, geom, binning, img, view=view, n_views=n_views)[0])
    dtype, shp = _QSPEC[name]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out = torch.zeros(shp(P, R, T, W * H), dtype=dtype, device=geom.device)
    if out.numel() == 0:
        return out
    p = GsrParams()
    p.P, p.W, p.H = P, W, H
    # view v's arenas start v strides into the allocations (include/gsr.h: V identically laid out single-view arenas)
    g_stride, i_stride = lib.gsr_geom_bytes_inference(P) - 256, lib.gsr_image_bytes(W, H) - 256
    b_stride = ((binning.numel() - 256) // n_views) // 256 * 256 if binning.numel() else 0
    if geom.data_ptr() % 256 or img.data_ptr() % 256 or (binning.numel() and binning.data_ptr() % 256):
        raise RuntimeError("query: arena base pointers are expected to be 256-byte aligned")
    with torch.cuda.device(geom.device):
        _check(lib.gsr_query(C.byref(p), Q[name], geom.data_ptr() + view * g_stride,
                             (binning.data_ptr() + view * b_stride) if binning.numel() else None, b_stride + 256,
                             img.data_ptr() + view * i_stride, int(R), out.data_ptr(), out.numel() * out.element_size(),
                             torch.cuda.current_stream(geom.device).cuda_stream))
    return out


def grad_records(geom, P, view=0, n_views=1):
    """The [P, 16] render-level gradient records of `view` (mean2D.xy, conic.xyw, colour rgb, opacity, 7 unused words) as the
    last backward on these arenas left them: the counterpart of the reference's internal dL_dconic / dL_dmean2D / dL_dcolors /
    dL_dopacity accumulators (rasterize_points.cu:151-159), for tests that pin the render backward on its own.  The records
    follow the n_views per-view geometry arenas inside a need_backward geometry allocation (include/gsr.h)."""
    g_stride = lib.gsr_geom_bytes_inference(P) - 256
    gr_stride = lib.gsr_geom_bytes(P) - lib.gsr_geom_bytes_inference(P)
    if geom.data_ptr() % 256:
        raise RuntimeError("grad_records: arena base pointers are expected to be 256-byte aligned")
    if geom.numel() < n_views * (g_stride + gr_stride):
        raise RuntimeError("grad_records: not a need_backward geometry arena of %d views" % n_views)
    off = n_views * g_stride + view * gr_stride
    return geom[off:off + P * 64].view(torch.float32).view(P, 16).clone()


class ClockProbe:
    """Shader-clock readings under load (bench / profiles): launch() enqueues the library's one-wave probe kernel on a side
    stream of its own, next to whatever the device is running; mhz() waits for the probes and returns one effective shader
    clock per launch, in MHz (include/gsr.h gsr_clock_probe_launch)."""

    def __init__(self, device, capacity=256, iters=256):
        import threading
        self.device, self.iters, self.n, self._mu = device, int(iters), 0, threading.Lock()
        with torch.cuda.device(device):
            self.buf = torch.zeros((capacity, 2), dtype=torch.int64, device=device)
            self.stream = torch.cuda.Stream(device=device)
            self.khz = int(lib.gsr_wall_clock_khz())
        torch.cuda.synchronize(device)

    def launch(self):
        with self._mu:
            if self.n >= self.buf.shape[0]:
                return
            slot = self.n
            self.n += 1
        with torch.cuda.device(self.device):
            _check(lib.gsr_clock_probe_launch(self.buf.data_ptr() + 16 * slot, self.iters, self.stream.cuda_stream))

    def mhz(self, first=0):
        """effective shader clock of probes first .. n-1 (MHz); [] when the wall clock rate is unknown"""
        self.stream.synchronize()
        rows = self.buf[first:self.n].cpu().numpy()
        if self.khz <= 0:
            return []
        return [float(s) / float(w) * self.khz / 1e3 for s, w in rows if w > 0]


def selftest(device):
    with torch.cuda.device(device):
        _check(lib.gsr_selftest(torch.cuda.current_stream(device).cuda_stream))


def set_backward_moments(mode):
    """How the render backward's pixel contraction takes its second moments: 0 about the quadrant centre, 1 about the four
    sub-quadrant centres (mean2D / conic sums 1.25x / 2.0x the reference build's rounding error instead of 1.85x / 4.3x, +8 % of
    that kernel's time), 2 (default) per batch of eight entries, the sub-quadrant way only where a splat lies far from the quadrant
    centre in its own sigmas (1.33x / 2.1x, +0.8 %); include/gsr.h gsr_set_backward_moments, GSR_BWD_SUBQ.  Returns the mode in force."""
    return int(lib.gsr_set_backward_moments(int(mode)))


def set_profiling(on):
    """False / 0: off; True / 1: hipEvents around every stage; 2: around the two render kernels only"""
    lib.gsr_set_profiling(int(on))


def get_profile():
    names = (C.c_char_p * 8192)()
    ms = (C.c_float * 8192)()
    n = lib.gsr_get_profile(names, ms, 8192)
    return [(names[i].decode(), float(ms[i])) for i in range(n)]
