/*
 * gsr.h -- C ABI of the MI355X-native Gaussian tile rasterizer (libgsr_hip.so).
 *
 * This is the drop-in boundary for the one hot path of huzi96/gaussian-pcloud-render:
 * the differentiable rasterizer behind diff_gaussian_rasterization.GaussianRasterizer.
 * It replaces the reference's torch/pybind binding + CUDA core:
 *
 *   reference (under /root/reference/diff-gaussian-rasterization/)      this header
 *   ------------------------------------------------------------------  -------------------------
 *   ext.cpp:16  rasterize_gaussians        -> rasterize_points.cu:35   gsr_forward_stage1 + _stage2
 *               CudaRasterizer::Rasterizer::forward  rasterizer.h:35-59
 *   ext.cpp:17  rasterize_gaussians_backward -> rasterize_points.cu:117 gsr_backward
 *               CudaRasterizer::Rasterizer::backward rasterizer.h:61-90
 *   ext.cpp:18  mark_visible               -> rasterize_points.cu:198  gsr_mark_visible
 *               CudaRasterizer::Rasterizer::markVisible rasterizer.h:28-33
 *   rasterizer_impl.h:65-72 required<GeometryState/ImageState/BinningState>()
 *                                                                       gsr_geom_bytes / gsr_image_bytes / gsr_binning_bytes
 *   auxiliary.h:166-173 CHECK_CUDA(..., debug) + thrown runtime_error   int status + gsr_last_error()
 *
 * Conventions
 *   - plain C: raw device pointers, ints, floats; no torch / ATen / pybind types.
 *   - the CALLER owns every buffer (inputs, outputs, the three opaque scratch arenas); the library
 *     never allocates device memory on the hot path.  Arena contents are private to the library
 *     (struct-of-arrays) and only need to survive from forward to backward, exactly
 *     like geomBuffer/binningBuffer/imgBuffer in diff_gaussian_rasterization/__init__.py:97.
 *   - optional inputs are NULL when absent (the reference passes data_ptr() of empty tensors, i.e.
 *     nullptr: rasterizer_impl.cu:321,389,411).
 *   - every launch goes to the explicit `stream` (the reference uses the legacy default stream).
 *   - the binning arena size depends on num_rendered, which the reference reads back in the middle of
 *     the frame to grow its arena through a std::function callback (rasterizer_impl.cu:279-285).  Here
 *     the count stays on the device: the caller hands over an arena of some CAPACITY (gsr_binning_bytes
 *     of the pair count it expects, e.g. the previous frame's plus slack), the whole frame is enqueued
 *     without a host round trip, and gsr_forward_batch reports afterwards whether the capacity was enough
 *     (GSR_RETRY: allocate gsr_binning_bytes(max num_rendered) and repeat with resume = 1).  The
 *     two-stage pair gsr_forward_stage1/_stage2 keeps the reference's synchronous shape on top of that.
 *   - a call covers a batch of V camera views of ONE cloud (V = 1: the reference's per-view call).
 *     Per-view inputs (viewmatrix, projmatrix, campos) and outputs (radii, out_color, dL_dpix) are V
 *     consecutive arrays; each scratch arena is V times the single-view size.
 *   - return value: 0 = ok, negative = error (text via gsr_last_error()), GSR_RETRY = see above.  With
 *     debug != 0 the library synchronises and checks after every kernel like CHECK_CUDA(…, true).
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_ERR_INVALID (-1)  /* bad argument (message says which)                              */
#define GSR_ERR_HIP (-2)      /* a HIP call or kernel failed                                    */
#define GSR_ERR_CAPACITY (-3) /* an arena is smaller than gsr_*_bytes() requires                */
#define GSR_ERR_TRAP (-4)     /* prefiltered=1 but a point was culled (auxiliary.h:156-160)     */
#define GSR_RETRY 1           /* forward: a view produced more pairs than the binning arena holds */

typedef void* gsr_stream_t; /* hipStream_t */

/* Arguments shared by forward and backward: the 19/21-argument lists of
 * diff_gaussian_rasterization/__init__.py:60-80,109-129 minus the tensors' metadata. */
typedef struct gsr_params {
    int P;                 /* number of Gaussians (means3D.size(0))                                  */
    int D;                 /* active SH degree (raster_settings.sh_degree)                           */
    int M;                 /* SH coefficients per Gaussian = sh.size(1), 0 when shs is absent        */
    int W, H;              /* image_width, image_height                                              */
    float tanfovx, tanfovy;
    float scale_modifier;
    int prefiltered;
    int debug;
    int need_backward;     /* 0: inference call, skip the saves only gsr_backward reads (SH clamp mask,
                              accumulated colour, per-pixel state at the list-slice boundaries the backward's
                              work items start from: every 512 entries of a tile's list in single-view
                              submissions, every 1024 from two views per submission on);
                              gsr_backward is only valid after a forward with need_backward = 1      */
    int reference_lists;   /* 0 (default): a Gaussian emits pairs for the tiles of the reference's rectangle
                              (auxiliary.h:46-56) in which alpha can reach 1/255 -- the rectangle clipped to the
                              bounding box of the ellipse  d^T conic d <= 2 log(255 opacity), conservatively against
                              fp32 rounding (csrc/tile_cull.hpp).  The reference evaluates the other tiles' entries
                              and skips them at every pixel (forward.cu:336-347), so out_color, radii, num_rendered
                              and every gradient are the same bit for bit / sum for sum; only the PRIVATE lists
                              (and the list positions in n_contrib) differ.  1: the reference's full rectangles --
                              lists, ranges and n_contrib identical to the reference's (parity tests).
                              (occupies what used to be padding: zero-initialised callers are unaffected)  */
    const float* bg;             /* [3]        device */
    const float* means3D;        /* [P,3]      device */
    const float* shs;            /* [P,M,3]    device or NULL */
    const float* colors_precomp; /* [P,3]      device or NULL */
    const float* opacities;      /* [P]        device */
    const float* scales;         /* [P,3]      device or NULL */
    const float* rotations;      /* [P,4]      device or NULL */
    const float* cov3D_precomp;  /* [P,6]      device or NULL */
    const float* viewmatrix;     /* [V][16] column-major when flattened (auxiliary.h:58-76) device */
    const float* projmatrix;     /* [V][16]    device */
    const float* campos;         /* [V][3]     device */
} gsr_params;

/* Scratch arena sizes in bytes for ONE view (256-B aligned sub-arrays inside); a batch of V views needs V times as much.
 * gsr_binning_bytes(n) is an arena that holds n pairs per view: the library derives the capacity from the arena's size. */
size_t gsr_geom_bytes(int P);
/* The geometry arena of a call with need_backward = 0 (inference): without the 64-B per-Gaussian gradient records that only
 * the backward accumulates into (they follow the V per-view arenas inside the allocation, so the layout of everything else
 * does not depend on the flag).  A forward with need_backward = 1 and every backward need V * gsr_geom_bytes(P). */
size_t gsr_geom_bytes_inference(int P);
size_t gsr_image_bytes(int W, int H);
size_t gsr_binning_bytes(int64_t num_rendered);

/* Forward of V views in one submission (rasterize_points.cu:35-115 / Rasterizer::forward rasterizer.h:35-59, looped over
 * views by the reference's caller, simple_raw_render.py:259-278): per-Gaussian preprocess (cull, EWA projection, SH colour),
 * depth ordering, (tile, Gaussian) pair emission in depth order with its prefix sum, stable radix sort by tile, tile
 * ranges, per-tile front-to-back alpha compositing.  Writes radii[V,P], out_color[V,3,H,W] (planar CHW per view) and
 * num_rendered[V] (HOST array): the number of (tile, Gaussian) pairs of the reference's tile rectangles, i.e. exactly the
 * reference's num_rendered (rasterizer_impl.cu:277-281).  The lists the library keeps hold at most that many pairs (fewer
 * with footprint clipping, see gsr_params.reference_lists; gsr_last_list_pairs reports how many).
 * Nothing on the host waits for the device until every kernel of the batch is enqueued.
 * Returns GSR_OK, or GSR_RETRY when some view's lists exceed the per-view capacity of `binning`: out_color is then
 * invalid; call again with resume = 1, the same geom / image arenas and a binning arena of at least
 * V * gsr_binning_bytes(n) bytes, n = max_v num_rendered[v] (always enough) or max_v of gsr_last_list_pairs (exact)
 * (only the binning half of the frame is repeated). */
int gsr_forward_batch(const gsr_params* p, int V, void* geom, size_t geom_bytes, void* image, size_t image_bytes,
                      void* binning, size_t binning_bytes, int* radii, float* out_color, int64_t* num_rendered,
                      int resume, gsr_stream_t stream);

/* gsr_forward_batch that also composites `nx` (4 or 8; pad with zero channels) extra per-Gaussian channels with the SAME alphas,
 * transmittances and stopping decisions as the colour: out_extra[v][k] = sum_i extra[v][i][k] * view_scale[v][k] * alpha_i * T_i +
 * T_final * bg_extra[k], term for term what a separate call with colors_precomp = those channels would produce -- the
 * reference's callers render world xyz, a hit map and normals that way, one full rasterizer call each
 * (simple_raw_render.py:410-524), recomputing identical alphas four times.  extra_per_view = 0: extra [P][nx] is shared by the
 * views; 1: extra [V][P][nx], one array per view (the reference turns every normal towards the camera of the view it renders,
 * simple_raw_render.py:264-268); 2 (nx = 8 only): split -- extra holds channels 0..3 as [P][4] shared by the views, followed by
 * channels 4..7 as [V][P][4] (world xyz + hit value once, the turned normals per view: half the memory of 1 for that use).  extra_view_scale [V][nx] (or NULL) multiplies the values per view and channel.
 * out_extra is [V][nx][H][W].  P == 0: nothing is written (like out_color).  Same GSR_RETRY / resume contract. */
int gsr_forward_batch_channels(const gsr_params* p, int V, void* geom, size_t geom_bytes, void* image, size_t image_bytes,
                               void* binning, size_t binning_bytes, int* radii, float* out_color, int64_t* num_rendered, int resume,
                               int nx, int extra_per_view, const float* extra, const float* extra_view_scale, const float* bg_extra,
                               float* out_extra, gsr_stream_t stream);

/* The reference's synchronous shape for one view.  Stage 1: preprocess, depth ordering, pair counting; returns
 * num_rendered through *num_rendered_out after a device->host read-back on `stream` (cf. rasterizer_impl.cu:281).
 * out_color is not touched. */
int gsr_forward_stage1(const gsr_params* p, void* geom, size_t geom_bytes, void* image, size_t image_bytes,
                       int* radii, int64_t* num_rendered_out, gsr_stream_t stream);

/* Stage 2: pair emission, sort, ranges, compositing into out_color[3,H,W], with a binning arena of at least
 * gsr_binning_bytes(num_rendered) bytes. */
int gsr_forward_stage2(const gsr_params* p, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
                       void* image, size_t image_bytes, int64_t num_rendered, float* out_color, gsr_stream_t stream);

/* Colour-only re-render.  The reference's caller renders four passes per view that differ only in the
 * per-Gaussian colour (world xyz / SH colour / ones / normals, simple_raw_render.py:410-524), each through the whole
 * pipeline.  After a forward this entry re-renders the same V views with other colours on the SAME geometry, lists and
 * ranges: p->colors_precomp (verbatim; [P,3] shared by the views, or [V,P,3] with colors_per_view != 0 -- the reference's
 * normal pass flips the normals' sign per view) or p->shs (evaluated like the forward does); only P, D, M, W, H, bg, means3D,
 * shs / colors_precomp, campos of *p are read.  The result is bit-identical to a full forward with those colours.  The
 * arenas stay valid for further recolor calls.  With p->need_backward = 1 (and arenas of a need_backward forward) the
 * saves the backward reads are rewritten too (per-pixel state at the list-slice boundaries, accumulated colour, SH clamp
 * mask), so a backward afterwards -- called with the SAME colour inputs -- differentiates the LAST colours rendered; with
 * need_backward = 0 a backward after the recolor is not supported. */
int gsr_forward_recolor(const gsr_params* p, int V, int colors_per_view, void* geom, size_t geom_bytes, const void* binning,
                        size_t binning_bytes, void* image, size_t image_bytes, float* out_color, gsr_stream_t stream);

/* Backward of a batch (rasterize_points.cu:117-196 / Rasterizer::backward rasterizer.h:61-90), on the arenas of ONE forward call and with
 * that call's V (the arenas' internal layout -- per-view strides, the length of the list slices whose boundary states the forward
 * saved -- follows from V; a forward batch cannot be split into several backward calls).  dL_dpix is [V,3,H,W]; the
 * per-Gaussian gradients are SUMMED over the V views (what autograd does with the reference's per-view calls on a shared
 * cloud).  Every output is written for every Gaussian (zeros where it is invisible; all M rows of dL_dsh), so nothing has
 * to be cleared by the caller -- the reference zero-fills nine tensors per call (rasterize_points.cu:151-159) -- except
 * dL_dscale / dL_drot, which are not touched when cov3D_precomp is given.  The reference's internal dL_dconic accumulator
 * lives in the geometry arena.  Valid after a forward with need_backward = 1 on the same arenas, any number of times: every
 * call returns the gradients of ITS dL_dpix (a repeated call first clears what the previous one accumulated).
 * shapes: radii[V,P] dL_dmean2D[P,3] dL_dopacity[P,1] dL_dcolor[P,3] dL_dmean3D[P,3] dL_dcov3D[P,6] dL_dsh[P,M,3]
 * dL_dscale[P,3] dL_drot[P,4].
 * Alignment: dL_drot must be 16-byte aligned (GSR_ERR_INVALID otherwise); the other gradient outputs should be -- every
 * hipMalloc / torch allocation is -- because their rows then leave the kernel as whole 16-byte chunks; an output that is only
 * float-aligned is still written correctly, through slower per-row stores. */
int gsr_backward_batch(const gsr_params* p, int V, const int* radii, const void* geom, size_t geom_bytes, const void* binning,
                       size_t binning_bytes, const void* image, size_t image_bytes, const float* dL_dpix, float* dL_dmean2D,
                       float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                       float* dL_dscale, float* dL_drot, gsr_stream_t stream);

/* One view, the reference's argument shape (num_rendered is not needed: the lists' extent lives in the arenas). */
int gsr_backward(const gsr_params* p, const int* radii, int64_t num_rendered, const void* geom, size_t geom_bytes,
                 const void* binning, size_t binning_bytes, const void* image, size_t image_bytes,
                 const float* dL_dpix /* [3,H,W] */, float* dL_dmean2D, float* dL_dopacity,
                 float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                 float* dL_drot, gsr_stream_t stream);

/* Pairs in the lists of the calling thread's last forward call, per view (out[V], HOST array): what a binning arena has to
 * hold.  Equal to num_rendered with reference_lists = 1. */
int gsr_last_list_pairs(int64_t* out, int V);

/* present[i] = (view-space z of means3D[i] > 0.2)   (rasterizer_impl.cu:54-66) */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, gsr_stream_t stream);

/* Inspection of the private arenas (view 0 of a batch; pass a view's sub-arena pointers for the others), for parity tests
 * and the roofline report only (device->device copies into caller buffers; not part of the hot path).  `what` is one of
 * GSR_Q_*. */
#define GSR_Q_DEPTHS 1         /* float  [P]    view-space z of visible Gaussians (0 otherwise)            */
#define GSR_Q_MEANS2D 2        /* float  [P,2]                                                              */
#define GSR_Q_CONIC_OPACITY 3  /* float  [P,4]                                                              */
#define GSR_Q_RGB 4            /* float  [P,3]                                                              */
#define GSR_Q_TILES_TOUCHED 5  /* uint32 [P]                                                                */
#define GSR_Q_POINT_LIST 6     /* uint32 [L]   Gaussian ids sorted by (tile, depth bits, id); L = GSR_Q_LIST_PAIRS[0] (pass it as R) */
#define GSR_Q_POINT_LIST_KEYS 7/* uint64 [L]   (tile<<32)|depth bits, rebuilt for comparison with CUB keys  */
#define GSR_Q_RANGES 8         /* uint32 [T,2]                                                              */
#define GSR_Q_FINAL_T 9        /* float  [H*W]                                                              */
#define GSR_Q_N_CONTRIB 10     /* uint32 [H*W]                                                              */
#define GSR_Q_TILE_NEED 12     /* uint32 [T]   list entries the tile's render actually walked (roofline model)    */
#define GSR_Q_CLAMPED 11       /* uint8  [P,3]                                                              */
#define GSR_Q_LIST_PAIRS 14    /* uint64 [2]   pairs in the view's lists, pairs of the reference's rectangles      */
#define GSR_Q_DEPTH_SORT 13    /* uint32 [4]   depth sort of the frame: key base, key bits compared, passes run, -  */
int gsr_query(const gsr_params* p, int what, const void* geom, const void* binning, size_t binning_bytes, const void* image,
              int64_t num_rendered, void* dst, size_t dst_bytes, gsr_stream_t stream);

/* Per-stage timing of the calls made since gsr_set_profiling(1) (ms, hipEvents on each call's launch stream; the switch is
 * process wide, the records are kept per stream and returned stream by stream).  names/ms hold up to `cap` entries;
 * returns the count and resets the records.  gsr_set_profiling(2) times only the two render kernels (one event pair per
 * forward and per backward: cheap enough to leave on inside a timed region); 0 switches it off. */
void gsr_set_profiling(int on);
int gsr_get_profile(const char** names, float* ms, int cap);

/* Shader-clock probe (bench / profiles only): enqueues a one-wave kernel on `stream` that spins through `iters` x 64 dependent FMAs
 * (about iters x 0.15 us) and stores two 64-bit counts at dst16 (device memory, 16 bytes): elapsed shader-clock ticks (s_memtime) and
 * elapsed ticks of the constant-rate wall clock (s_memrealtime, gsr_wall_clock_khz() ticks per millisecond).  Their ratio x the wall
 * rate is the shader clock the device sustained while the probe was resident -- launched on a side stream it reads the clock UNDER
 * the load of the kernels running beside it.  Nothing in the library waits for it. */
int gsr_clock_probe_launch(void* dst16, int iters, gsr_stream_t stream);
int gsr_wall_clock_khz(void);

/* Device self-test of internal primitives (the matrix-core pixel contraction of the render backward, stable radix sort vs std::stable_sort).
 * Allocates its own small buffers; not part of the hot path.  0 = pass. */
int gsr_selftest(gsr_stream_t stream);

/* Number of device->host read-backs the library has made in this process: exactly one per forward call -- the 32-byte
 * per-view counter block (num_rendered, trap and stall flags), which the pair-emission kernel stores into mapped host
 * memory and the host reads after ONE event wait, once the whole frame is enqueued -- and none per backward.  There is no
 * copy command in the stream.  For tests that guard the call path against host stalls. */
long long gsr_d2h_count(void);

/* Tuning / test hook: submissions of up to `views` views run the forward render in half-quadrant mode (csrc/render_fwd.hip: a wave
 * owns 8 x 4 pixels twice and blends four list entries per step -- a shorter critical path for the deep walks a single view's launch
 * ends on; images, final_T and n_contrib are bit-identical in both modes).  Default 1 (GSR_FWD_HALF_V in the environment), 0 = the
 * 8 x 8 kernel always; views < 0 only queries.  Returns the value in force. */
int gsr_set_forward_half_views(int views);

/* Tuning / test hook: how the two sorts of a submission run (csrc/sort.hip; the sorted lists are bit-identical either way).
 * mode 0: three launches per radix pass (digit histogram per block, row scan, stable scatter) -- least traffic; 1: one kernel reads
 * the keys once and counts the digits of every pass, then ONE scatter launch per pass whose workgroups look back at the counts of
 * the workgroups before them ("onesweep"), and the tile ranges are prefix sums of the per-tile counts -- 12 launches per single-view
 * forward instead of 23; 2 (default; GSR_SORT_MODE): look-back passes for submissions of up to `lookback_views` views (default 2;
 * GSR_SORT_LB_VIEWS), three-launch passes for larger batches.  mode < 0 / lookback_views < 0 leave the setting.  Returns the mode. */
int gsr_set_sort_mode(int mode, int lookback_views);

/* Accuracy / speed switch of the render backward's pixel contraction (GSR_BWD_SUBQ in the environment): 0 second moments about the
 * 8 x 8 quadrant centre, shifted to the splat centre per entry: the mean2D / conic sums carry 1.85x / 4.3x the reference build's own
 * rounding error (median over 500 fuzz cases against a float64 evaluation of the same sums; two orders of magnitude inside every
 * test bar); 1 about the centres of the four 4 x 4 sub-quadrants, each shifted on its own: 1.25x / 2.0x, +8 % kernel time;
 * 2 (DEFAULT) adaptive: the sub-quadrant path only for batches of eight entries that hold a splat more than sqrt(20) of its own sigmas
 * from the quadrant centre -- where that rounding comes from (sub-pixel splats seen from a quadrant away): 1.33x / 2.1x at +0.8 %
 * kernel time on the benchmark views, none of whose batches takes it (profiles/r06_bwd_accuracy.txt).  mode < 0 only queries.
 * Returns the value in force. */
int gsr_set_backward_moments(int mode);

const char* gsr_last_error(void);
const char* gsr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
