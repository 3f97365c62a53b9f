// api.hip -- the C ABI of include/gsr.h: argument checks, arena carving, kernel sequencing.
//
// Host orchestration corresponding to reference CR/rasterizer_impl.cu:198-336 (Rasterizer::forward) and
// :340-434 (Rasterizer::backward), re-planned for this library's data flow.  One submission covers a BATCH of V camera
// views of one cloud (V = 1 for the reference's per-view API); every launch below has the view as a grid dimension.
//
//   forward   preprocess (+ clears the frame's bookkeeping)      1 launch
//             depth sort of the P Gaussians                       4 passes x 3 launches, u32 key / u32 id
//             pair emission in depth order with its own prefix sum  1 launch     -> num_rendered, ON THE DEVICE
//             tile sort                                           ceil(bit/8) passes x 3 launches, u16|u32 tile / u32 id
//             tile ranges (search), tile order by work estimate   2 launches
//             render                                              1 launch
//   backward  work items, render backward, per-Gaussian backward  3 launches
//
// No host round trip inside a frame: where the reference blocks on a device->host copy of num_rendered to size its
// binning buffer (CR/rasterizer_impl.cu:279-285), the binning arena here is carved by CAPACITY (what the caller
// allocated; the Python layer sizes it from the previous frames of the same configuration), every kernel behind the pair
// emission takes its element count from device memory, and the 16-byte counter copy is enqueued right after the emission
// but only waited for once the WHOLE frame has been enqueued.  If a view's count exceeds the capacity the call returns
// GSR_RETRY with the true counts and the caller repeats the binning half with a larger arena (resume = 1).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include <atomic>
#include <map>
#include <mutex>

namespace gsr {

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const Launch& L, const char* what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && L.debug) e = hipStreamSynchronize(L.stream);  // CHECK_CUDA(…, debug) of the reference
    if (e != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] %s: %s", what, hipGetErrorString(e));
    return GSR_OK;
}

// ---- optional per-step timing (hipEvents on the launch stream) -------------------------------------
// Records are kept PER STREAM (several host threads may render at once, each on its own stream, and one frame's forward
// and backward usually come from different host threads -- autograd runs the backward on its own thread -- but on the
// same stream).  The switch is process wide.
struct ProfEntry {
    const char* name;
    int a, b;  // indices into the stream's event pool
};
struct ProfTables {
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    std::vector<ProfEntry> rec;
    int event()
    {
        if (used == pool.size()) {
            // timing events only: no system-scope fence when they are recorded (hipEventRecord's default release writes the L2s back
            // and invalidates them between two stages, so the kernel behind an event pair started cold: the per-stage pass read
            // k_render_backward 7-10 % slower than the kernel trace of the same run, profiles/r04_bench_kernel_stats.csv)
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) (void)hipEventCreate(&e);
            pool.push_back(e);
        }
        return (int)used++;
    }
};
static std::atomic<int> g_prof_on{0};    // 0 off, 1 every stage, 2 the two render kernels only (cheap enough for a timed region)
static std::mutex g_prof_mu;
static std::map<hipStream_t, ProfTables> g_prof;   // guarded by g_prof_mu

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfEntry e;
    ProfScope(const char* name, hipStream_t stream) : s(stream), on(false)
    {
        const int mode = g_prof_on;
        on = mode == 1 || (mode == 2 && strncmp(name, "render_", 7) == 0);
        if (!on) return;
        e.name = name;
        hipEvent_t ea;
        {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            ProfTables& t = g_prof[s];
            e.a = t.event();
            e.b = t.event();
            ea = t.pool[e.a];
        }
        (void)hipEventRecord(ea, s);
    }
    ~ProfScope()
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        ProfTables& t = g_prof[s];
        (void)hipEventRecord(t.pool[e.b], s);
        t.rec.push_back(e);
    }
};

// ---- per-thread landing zone for the counter read-back ----------------------------------------------------
// Host memory mapped into the device: the pair-emission kernel stores the per-view counters there itself, and the host
// reads them after an event recorded behind that kernel.  No copy command (on this runtime a small device->host copy is a
// blit kernel plus two ~10 us bubbles in the stream).
constexpr int MAX_VIEWS = 256;
// One per (host thread, device): an event belongs to the device that was current when it was created, so a thread that
// renders on several GPUs needs one landing zone for each.
struct HostLanding {
    uint64_t* pinned = nullptr;   // [MAX_VIEWS][4]: num_rendered, trap flag, stall flag, -   (host address)
    uint64_t* mapped = nullptr;   // the same memory as the device sees it
    hipEvent_t ev = nullptr;
};
static thread_local std::map<int, HostLanding> t_lands;
static int landing(HostLanding** out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] hipGetDevice: %s", hipGetErrorString(hipGetLastError()));
    HostLanding& h = t_lands[dev];
    if (!h.pinned) {
        if (hipHostMalloc((void**)&h.pinned, MAX_VIEWS * 4 * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void**)&h.mapped, h.pinned, 0) != hipSuccess ||
            // (hipEventDisableSystemFence on this event was tried -- the counters live in coherent host memory and the kernel fences
            // them itself, so the record would not need the system-scope release of the L2s -- and cost the per-view path a fifth of
            // its rate, 1 363 -> 1 050-1 130 frames/s, gpurun_out/r4m: waiting for such an event does not return when the emission
            // kernel ends but when the stream drains, and the host stops running ahead of the device)
            hipEventCreateWithFlags(&h.ev, hipEventDisableTiming) != hipSuccess) {
            h.pinned = nullptr;
            return fail(GSR_ERR_HIP, "[gsr] pinned host buffer: %s", hipGetErrorString(hipGetLastError()));
        }
    }
    *out = &h;
    return GSR_OK;
}

// pairs in the lists of the calling thread's last forward, per view (<= num_rendered: footprint clipping); gsr_last_list_pairs
static thread_local std::vector<int64_t> t_list_pairs;

// device->host read-backs the library has issued since it was loaded (one per forward call: the per-view counters)
static std::atomic<long long> g_d2h_count{0};

// ---- which way the sorts run (common.hpp sort_mode) ------------------------------------------------------------------------------
static std::atomic<int> g_sort_mode{[] { const char* e = getenv("GSR_SORT_MODE"); const int m = e ? atoi(e) : 0; return m < 0 || m > 2 ? 0 : m; }()};
static std::atomic<int> g_sort_lb_views{[] { const char* e = getenv("GSR_SORT_LB_VIEWS"); const int v = e ? atoi(e) : 2; return v < 0 ? 0 : v; }()};
static std::atomic<int> g_tickets{[] { const char* e = getenv("GSR_TICKETS"); return e && atoi(e) == 0 ? 0 : 1; }()};
int block_tickets(int set)
{
    if (set >= 0) g_tickets = set != 0;
    return g_tickets;
}
int sort_mode(int set)
{
    if (set >= 0 && set <= 2) g_sort_mode = set;
    return g_sort_mode;
}
int lookback_max_views(int set)
{
    if (set >= 0) g_sort_lb_views = set;
    return g_sort_lb_views;
}
static bool use_lookback(int V) { const int m = g_sort_mode; return m == 1 || (m == 2 && V <= g_sort_lb_views); }

static int tile_count(const gsr_params* p) { return ((p->W + TILE_X - 1) / TILE_X) * ((p->H + TILE_Y - 1) / TILE_Y); }

static int check_params(const gsr_params* p, int V = 1)
{
    if (!p) return fail(GSR_ERR_INVALID, "[gsr] params is NULL");
    if (p->P < 0 || p->W <= 0 || p->H <= 0) return fail(GSR_ERR_INVALID, "[gsr] bad sizes P=%d W=%d H=%d", p->P, p->W, p->H);
    if (V < 1 || V > MAX_VIEWS) return fail(GSR_ERR_INVALID, "[gsr] view count %d outside 1..%d", V, MAX_VIEWS);
    if (p->P == 0) return GSR_OK;
    if (!p->means3D || !p->opacities || !p->bg || !p->viewmatrix || !p->projmatrix || !p->campos)
        return fail(GSR_ERR_INVALID, "[gsr] a required input pointer is NULL");
    if ((p->shs == nullptr) == (p->colors_precomp == nullptr))
        return fail(GSR_ERR_INVALID, "Please provide excatly one of either SHs or precomputed colors!");
    if (((p->scales == nullptr || p->rotations == nullptr) && p->cov3D_precomp == nullptr) ||
        ((p->scales != nullptr || p->rotations != nullptr) && p->cov3D_precomp != nullptr))
        return fail(GSR_ERR_INVALID, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (p->shs) {
        if (p->D < 0 || p->D > 3) return fail(GSR_ERR_INVALID, "[gsr] SH degree %d outside 0..3", p->D);
        if ((p->D + 1) * (p->D + 1) > p->M) return fail(GSR_ERR_INVALID, "[gsr] SH degree %d needs %d coefficients, M=%d", p->D, (p->D + 1) * (p->D + 1), p->M);
    }
    const int64_t gx = (p->W + TILE_X - 1) / TILE_X, gy = (p->H + TILE_Y - 1) / TILE_Y;
    if (gx > 65535 || gy > 65535) return fail(GSR_ERR_INVALID, "[gsr] image too large for 16-bit tile coordinates");
    // backward work items carry the tile in BWD_TILE_BITS bits, and the render grids hold 4 workgroups per tile and view
    if (gx * gy > (1ll << BWD_TILE_BITS) || (gx * gy / 8 + 1) * 64 * V > 0x7FFFFFFFll)   // (64: the half-quadrant forward's grid)
        return fail(GSR_ERR_INVALID, "[gsr] %lld tiles x %d views: too many for one launch", (long long)(gx * gy), V);
    return GSR_OK;
}

// how many u32 tile-key bits the tile sort covers: the reference's 32+bit minus the 32 depth bits
static int tile_bits(int T) { return (int)higher_msb((uint32_t)T); }
// which of the two ping-pong buffers holds the sorted lists (depends on the pass count only)
static int sorted_buffer(int T) { return ((tile_bits(T) + RADIX_BITS - 1) / RADIX_BITS) & 1; }

static inline void* align256(void* p) { return (void*)(((uintptr_t)p + 255) & ~(uintptr_t)255); }

// Carve the caller's three allocations into V per-view arenas.  binning may be NULL (count-only forward).  with_grad: the
// V gradient-record blocks that follow the V geometry arenas are part of the geometry allocation (need_backward calls).
static int make_batch(const gsr_params* p, int V, void* geom, size_t geom_bytes, void* image, size_t image_bytes, void* binning,
                      size_t binning_bytes, bool with_grad, Batch& B)
{
    B.V = V;
    const size_t G = geom_view(nullptr, p->P).bytes, I = image_view(nullptr, p->W, p->H).bytes;
    const size_t GR = with_grad ? grad_rec_bytes(p->P) : 0;
    if (!geom || geom_bytes < (size_t)V * (G + GR) + 256)
        return fail(GSR_ERR_CAPACITY, "[gsr] geom arena too small (%zu < %zu%s)", geom_bytes, (size_t)V * (G + GR) + 256,
                    with_grad ? ", need_backward" : "");
    if (!image || image_bytes < (size_t)V * I + 256)
        return fail(GSR_ERR_CAPACITY, "[gsr] image arena too small (%zu < %zu)", image_bytes, (size_t)V * I + 256);
    B.g = geom_view(align256(geom), p->P);
    B.g_stride = G;
    B.grad_rec = with_grad ? reinterpret_cast<float*>(reinterpret_cast<char*>(align256(geom)) + (size_t)V * G) : nullptr;
    B.gr_stride = GR;
    B.iv = image_view(align256(image), p->W, p->H);
    B.iv_stride = I;
    if (binning) {
        if (binning_bytes < 256 + 256 * (size_t)V) return fail(GSR_ERR_CAPACITY, "[gsr] binning arena too small");
        const size_t per_view = ((binning_bytes - 256) / (size_t)V) / 256 * 256;
        const int64_t cap = bin_capacity_from_bytes(per_view);
        if (cap < 1) return fail(GSR_ERR_CAPACITY, "[gsr] binning arena too small (%zu bytes for %d views)", binning_bytes, V);
        B.b = bin_view(align256(binning), cap > 0xFFFFFFFFll ? 0xFFFFFFFFll : cap);
        B.b_stride = per_view;
    } else {
        B.b = bin_view(nullptr, 1);
        B.b.cap = 0;
        B.b_stride = 0;
    }
    return GSR_OK;
}

static int memset_views(hipStream_t s, void* base, size_t stride, size_t bytes, int V)
{
    if (bytes == 0) return GSR_OK;
    hipError_t e = V == 1 ? hipMemsetAsync(base, 0, bytes, s) : hipMemset2DAsync(base, stride, 0, bytes, (size_t)V, s);
    return e == hipSuccess ? GSR_OK : fail(GSR_ERR_HIP, "[gsr] memset failed: %s", hipGetErrorString(e));
}

// The whole forward for a batch.  mode 0: everything; 1 (resume): from the pair emission on, after a GSR_RETRY or a
// count-only call; 2 (count only): geometry + pair counting, no binning arena.
static int forward_impl(const gsr_params* p, int V, void* geom, size_t geom_bytes, void* image, size_t image_bytes, void* binning,
                        size_t binning_bytes, int* radii, float* out_color, int64_t* num_rendered, int mode, hipStream_t stream,
                        const ExtraChannels* X = nullptr)
{
    if (int e = check_params(p, V)) return e;
    if (!num_rendered) return fail(GSR_ERR_INVALID, "[gsr] num_rendered is NULL");
    for (int v = 0; v < V; v++) num_rendered[v] = 0;
    if (p->P == 0) return GSR_OK;  // reference rasterize_points.cu:81: nothing runs, image stays zero
    if (mode != 1 && !radii) return fail(GSR_ERR_INVALID, "[gsr] radii is NULL");
    if (mode != 2 && !out_color) return fail(GSR_ERR_INVALID, "[gsr] out_color is NULL");
    if (mode != 2 && !binning) return fail(GSR_ERR_INVALID, "[gsr] binning arena is NULL");
    Batch B;
    if (int e = make_batch(p, V, geom, geom_bytes, image, image_bytes, mode == 2 ? nullptr : binning, binning_bytes,
                           p->need_backward != 0, B))
        return e;
    HostLanding* landp = nullptr;
    if (int e = landing(&landp)) return e;
    HostLanding& t_land = *landp;
    const Launch L{stream, p->debug};
    const int T = tile_count(p);
    const int gridx = (p->W + TILE_X - 1) / TILE_X;
    const bool key16 = tile_keys16(T);
    // the emission kernel leaves the counters of every view in mapped host memory (and a sort pass that gives up waiting raises its
    // flag there); cleared before the first kernel that may store into it is enqueued
    memset(t_land.pinned, 0, (size_t)V * 4 * sizeof(uint64_t));

    if (mode == 1) {
        // the first attempt consumed the frame's bookkeeping: clear it again (k_preprocess did it the first time)
        if (int e = memset_views(L.stream, B.g.zero_begin, B.g_stride, B.g.zero_bytes, V)) return e;
        if (int e = memset_views(L.stream, B.iv.zero_begin, B.iv_stride, B.iv.zero_bytes, V)) return e;
    } else {
        if (p->prefiltered)   // the trap word is only looked at for prefiltered calls
            if (int e = memset_views(L.stream, B.g.counters, B.g_stride, 8 * sizeof(uint64_t), V)) return e;
        {
            ProfScope ps("preprocess", L.stream);
            if (int e = launch_preprocess(L, *p, B, radii)) return e;
        }
        {
            ProfScope ps("depth_sort", L.stream);
            SortJob job{{B.g.dkey[0], B.g.dkey[1]}, {B.g.dval[0], B.g.dval[1]}, B.g.hist, B.g.totals, B.g_stride, nullptr, 0, p->P, V};
            job.blk_minmax = B.g.blk_minmax;
            job.sortctl = B.g.sortctl;
            if (use_lookback(V) && lb_fits(p->P)) {
                LbJob lj{B.g.lb, lb_pass_words(p->P), B.g.ghist, B.g_stride, B.g.counters, B.g_stride, t_land.mapped};
                lj.pre_minmax = B.g.pre_minmax;
                if (int e = launch_depth_sort_lookback(L, job, lj)) return e;
            } else {
                // half-size sort blocks while whole-size ones would leave the chip short of workgroups
                job.small_blocks = div_up(p->P, RS_TILE) * V < 4096;
                int res = 0;
                if (int e = launch_radix_sort_pairs(L, job, /*iota_vals=*/true, 32, &res)) return e;
            }
            // up to 4 passes (B.g.sortctl says how many did something): the ids in depth order are in dval[passes & 1]
        }
    }
    // the event behind the emission kernel is waited for only after the rest of the frame has been enqueued.
    // Once the emission is in the stream it WILL store into the landing zone, whatever happens to the launches behind it: an
    // early return must not leave it pending (the next call of this thread clears the zone from the host and would race
    // with those late stores), so every error path below drains the stream first.
    const auto rest = [&]() -> int {
        {
            ProfScope ps("duplicate", L.stream);
            if (int e = launch_duplicate(L, p->P, B, gridx, key16, t_land.mapped)) return e;
        }
        {
            g_d2h_count++;
            const hipError_t e = hipEventRecord(t_land.ev, L.stream);
            if (e != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] num_rendered read-back: %s", hipGetErrorString(e));
        }
        if (mode != 2) {
            const int res = sorted_buffer(T);
            // look-back passes: images of up to LB_MAX_TILES tiles (the histogram over whole keys lives in LDS), arenas of up to
            // LB_MAX_BLOCKS sort blocks per view; the tile ranges are then prefix sums of the histogram
            const bool lookback = use_lookback(V) && tile_bits(T) <= 15 && lb_fits(B.b.cap);
            {
                ProfScope ps("tile_sort", L.stream);
                SortJob job{{B.b.key[0], B.b.key[1]}, {B.b.val[0], B.b.val[1]}, B.b.hist, B.b.totals, B.b_stride,
                            B.g.counters + CNT_NUM_RENDERED, B.g_stride, B.b.cap, V};
                int r2 = 0;
                if (lookback) {
                    LbJob lj{B.b.lb, lb_pass_words(B.b.cap), B.iv.tile_ghist, B.iv_stride, B.g.counters, B.g_stride, t_land.mapped};
                    lj.tile_count = B.iv.tile_count;
                    if (int e = launch_tile_sort_lookback(L, job, lj, T, tile_bits(T), &r2, key16)) return e;
                } else {
                    if (int e = launch_radix_sort_pairs(L, job, false, tile_bits(T), &r2, key16)) return e;
                }
            }
            {
                ProfScope ps("tile_ranges", L.stream);
                if (lookback) {
                    if (int e = launch_ranges_order(L, B, T)) return e;
                } else {
                    // (one launch of one workgroup per view doing both was measured: 47 us instead of 7 + 11 for a single view --
                    // 8 160 binary searches are ~100 000 scattered line requests, and ONE CU's address unit passes about one line
                    // per cycle; the search wants its 32 workgroups.  gpurun_out/r6e, profiles/r06_lookback_sort.txt)
                    if (int e = launch_tile_ranges(L, B, B.b.key[res], T, key16)) return e;
                    if (int e = launch_tile_order(L, B, T)) return e;
                }
            }
            {
                ProfScope ps("render_forward", L.stream);
                if (int e = launch_render_forward(L, *p, B, B.b.val[res], out_color, p->need_backward != 0, X)) return e;
            }
        }
        return GSR_OK;
    };
    if (int e = rest()) {
        (void)hipStreamSynchronize(L.stream);   // (keeps g_err: the failure being reported)
        return e;
    }
    if (hipEventSynchronize(t_land.ev) != hipSuccess)
        return fail(GSR_ERR_HIP, "[gsr] num_rendered read-back failed: %s", hipGetErrorString(hipGetLastError()));
    bool retry = false;
    t_list_pairs.assign((size_t)V, 0);
    for (int v = 0; v < V; v++) {
        // R: pairs of the reference's tile rectangles (what the reference calls num_rendered); L: pairs in this library's
        // lists (fewer when the rectangles are clipped to the splats' footprints) -- the arena has to hold L
        const uint64_t R = t_land.pinned[4 * v + LAND_NUM_REFERENCE], Lp = t_land.pinned[4 * v + CNT_NUM_RENDERED];
        if (t_land.pinned[4 * v + CNT_STALL])
            return fail(GSR_ERR_HIP, "[gsr] pair emission: a workgroup's pair count never arrived (view %d)", v);
        if (p->prefiltered && t_land.pinned[4 * v + CNT_TRAP])
            return fail(GSR_ERR_TRAP, "Point is filtered although prefiltered is set. This shouldn't happen!");
        // the reference keeps num_rendered in an int (CR/rasterizer_impl.cu:280); beyond that its arena sizes wrap
        if (R > 0x7FFFFFFFull)
            return fail(GSR_ERR_CAPACITY, "[gsr] num_rendered = %llu tile pairs does not fit the reference's int (scales too large?)",
                        (unsigned long long)R);
        num_rendered[v] = (int64_t)R;
        t_list_pairs[(size_t)v] = (int64_t)Lp;
        if (mode != 2 && (int64_t)Lp > B.b.cap) retry = true;
    }
    if (retry) {
        fail(GSR_RETRY, "[gsr] binning arena holds %lld pairs per view, the frame needs more (num_rendered is enough; gsr_last_list_pairs is exact): repeat with resume = 1",
             (long long)B.b.cap);
        return GSR_RETRY;
    }
    return GSR_OK;
}

// The look-back sorts on their own buffers: (1) tile-sort shaped -- 12-bit keys with many ties, explicit values, the element
// count on the device, pairs per tile checked against a host histogram; (2) depth-sort shaped -- 32-bit keys that differ in 17
// bits above a base (three passes run, the fourth leaves), values = indices, key extremes handed over like k_preprocess does.
struct DevBuffers {   // selftest allocations, freed on every way out
    std::vector<void*> all;
    ~DevBuffers() { for (void* p : all) (void)hipFree(p); }
    template <typename T_> T_* get(size_t count)
    {
        void* p = nullptr;
        if (hipMalloc(&p, count * sizeof(T_) + 16) != hipSuccess) return nullptr;
        all.push_back(p);
        return (T_*)p;
    }
};
static int selftest_lookback(hipStream_t s, const std::vector<uint32_t>& hk, const std::vector<uint32_t>& order)
{
    const int64_t n = (int64_t)hk.size();
    const int T = 3001;
    const size_t pw = lb_pass_words(n);
    DevBuffers dev;
    uint32_t* k0 = dev.get<uint32_t>(n); uint32_t* k1 = dev.get<uint32_t>(n);
    uint32_t* v0 = dev.get<uint32_t>(n); uint32_t* v1 = dev.get<uint32_t>(n);
    uint32_t* lbw = dev.get<uint32_t>(4 * pw); uint32_t* ghist = dev.get<uint32_t>(4 * RADIX);
    uint32_t* tcount = dev.get<uint32_t>(T); uint64_t* cnt = dev.get<uint64_t>(8);
    uint32_t* mm = dev.get<uint32_t>(2 * (size_t)div_up(n, PRE_THREADS)); uint32_t* ctl = dev.get<uint32_t>(4);
    if (!k0 || !k1 || !v0 || !v1 || !lbw || !ghist || !tcount || !cnt || !mm || !ctl) return fail(GSR_ERR_HIP, "[gsr] selftest: hipMalloc failed");
    const Launch L{s, 1};
    std::vector<uint32_t> iota(n), gk(n), gv(n);
    for (int64_t i = 0; i < n; i++) iota[i] = (uint32_t)i;
    // (1)
    {
        uint64_t hc[8] = {(uint64_t)n, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyAsync(k0, hk.data(), n * 4, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(v0, iota.data(), n * 4, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(cnt, hc, sizeof(hc), hipMemcpyHostToDevice, s);
        (void)hipMemsetAsync(ghist, 0, 4 * RADIX * 4, s);
        (void)hipMemsetAsync(tcount, 0, T * 4, s);
        (void)hipMemsetAsync(lbw, 0xFF, 4 * pw * 4, s);   // the histogram kernel has to clear it
        const SortJob job{{k0, k1}, {v0, v1}, nullptr, nullptr, 0, cnt + CNT_NUM_RENDERED, 0, n, 1};
        LbJob lj{lbw, pw, ghist, 0, cnt, 0, nullptr};
        lj.tile_count = tcount;
        int res = 0;
        if (int rc = launch_tile_sort_lookback(L, job, lj, T, 12, &res, false)) return rc;
        uint32_t* key[2] = {k0, k1};
        uint32_t* val[2] = {v0, v1};
        std::vector<uint32_t> tc(T), want(T, 0u);
        (void)hipMemcpyAsync(gk.data(), key[res], n * 4, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(gv.data(), val[res], n * 4, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(tc.data(), tcount, T * 4, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] selftest: look-back tile sort failed: %s", hipGetErrorString(hipGetLastError()));
        if (hc[CNT_STALL]) return fail(GSR_ERR_HIP, "[gsr] selftest: a look-back pass gave up waiting");
        for (int64_t i = 0; i < n; i++) {
            want[hk[i]]++;
            if (gv[i] != order[i] || gk[i] != hk[order[i]]) return fail(GSR_ERR_HIP, "[gsr] selftest: look-back tile sort differs from std::stable_sort at %lld", (long long)i);
        }
        for (int t = 0; t < T; t++)
            if (tc[t] != want[t]) return fail(GSR_ERR_HIP, "[gsr] selftest: pairs per tile wrong at tile %d", t);
    }
    // (2)
    {
        std::vector<uint32_t> dk(n), dord(n);
        uint32_t x = 99u, lo = 0xFFFFFFFFu, hi = 0u;
        for (int64_t i = 0; i < n; i++) {
            x = x * 1664525u + 1013904223u;
            dk[i] = (i % 11 == 3) ? CULLED_KEY : 0x3FFF8000u + (x >> 9) % 70000u;   // straddles the binade boundary at 2.0
            if (dk[i] != CULLED_KEY) { lo = dk[i] < lo ? dk[i] : lo; hi = dk[i] > hi ? dk[i] : hi; }
            dord[i] = (uint32_t)i;
        }
        const uint32_t base = lo & ~255u;
        // three passes compare the low 24 bits of (key - base): a culled key lands wherever those bits put it (its place is immaterial)
        std::stable_sort(dord.begin(), dord.end(), [&](uint32_t a, uint32_t b) { return ((dk[a] - base) & 0xFFFFFFu) < ((dk[b] - base) & 0xFFFFFFu); });
        const int n_pre = (int)div_up(n, PRE_THREADS);
        std::vector<uint32_t> hmm(2 * (size_t)n_pre);
        for (int b = 0; b < n_pre; b++) { hmm[b] = b == 7 ? lo : 0xFFFFFFFFu; hmm[n_pre + b] = b == n_pre - 1 ? hi : 0u; }
        uint64_t hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyAsync(k0, dk.data(), n * 4, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(mm, hmm.data(), hmm.size() * 4, hipMemcpyHostToDevice, s);
        (void)hipMemcpyAsync(cnt, hc, sizeof(hc), hipMemcpyHostToDevice, s);
        (void)hipMemsetAsync(ghist, 0, 4 * RADIX * 4, s);
        (void)hipMemsetAsync(lbw, 0xFF, 4 * pw * 4, s);
        SortJob job{{k0, k1}, {v0, v1}, nullptr, nullptr, 0, nullptr, 0, n, 1};
        job.sortctl = ctl;
        LbJob lj{lbw, pw, ghist, 0, cnt, 0, nullptr};
        lj.pre_minmax = mm;
        if (int rc = launch_depth_sort_lookback(L, job, lj)) return rc;
        uint32_t hctl[4];
        (void)hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] selftest: look-back depth sort failed: %s", hipGetErrorString(hipGetLastError()));
        const uint32_t passes = depth_sort_passes(hctl[SORTCTL_BITS]);
        if (hctl[SORTCTL_BASE] != base || passes != 3u) return fail(GSR_ERR_HIP, "[gsr] selftest: depth-sort control words wrong (base %08x, %u passes)", hctl[SORTCTL_BASE], passes);
        uint32_t* val[2] = {v0, v1};
        (void)hipMemcpyAsync(gv.data(), val[passes & 1], n * 4, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] selftest: copy failed");
        if (hc[CNT_STALL]) return fail(GSR_ERR_HIP, "[gsr] selftest: a look-back pass gave up waiting");
        for (int64_t i = 0; i < n; i++)
            if (gv[i] != dord[i]) return fail(GSR_ERR_HIP, "[gsr] selftest: look-back depth sort differs from std::stable_sort at %lld", (long long)i);
    }
    return GSR_OK;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

size_t gsr_geom_bytes(int P) { return geom_view(nullptr, P).bytes + grad_rec_bytes(P) + 256; }
size_t gsr_geom_bytes_inference(int P) { return geom_view(nullptr, P).bytes + 256; }
size_t gsr_image_bytes(int W, int H) { return image_view(nullptr, W, H).bytes + 256; }
size_t gsr_binning_bytes(int64_t R) { return bin_view(nullptr, R).bytes + 256; }

int gsr_forward_batch(const gsr_params* p, int V, void* geom, size_t geom_bytes, void* image, size_t image_bytes, void* binning,
                      size_t binning_bytes, int* radii, float* out_color, int64_t* num_rendered, int resume, gsr_stream_t stream)
{
    return forward_impl(p, V, geom, geom_bytes, image, image_bytes, binning, binning_bytes, radii, out_color, num_rendered,
                        resume ? 1 : 0, (hipStream_t)stream);
}

int gsr_forward_batch_channels(const gsr_params* p, int V, void* geom, size_t geom_bytes, void* image, size_t image_bytes,
                               void* binning, size_t binning_bytes, int* radii, float* out_color, int64_t* num_rendered, int resume,
                               int nx, int extra_per_view, const float* extra, const float* extra_view_scale, const float* bg_extra,
                               float* out_extra, gsr_stream_t stream)
{
    if (nx != 4 && nx != 8) return fail(GSR_ERR_INVALID, "[gsr] extra channels come in 4 or 8 (got %d): pad with zeros", nx);
    if (!extra || !bg_extra || !out_extra) return fail(GSR_ERR_INVALID, "[gsr] an extra-channel pointer is NULL");
    if (extra_per_view < 0 || extra_per_view > 2 || (extra_per_view == 2 && nx != 8))
        return fail(GSR_ERR_INVALID, "[gsr] extra_per_view is 0, 1 or (with 8 channels) 2");
    ExtraChannels X{nx, extra, extra_view_scale, bg_extra, out_extra, extra_per_view == 1 ? (size_t)p->P * (size_t)nx : (size_t)0};
    if (extra_per_view == 2) {   // [P][4] shared by the views, then [V][P][4]
        X.values_hi = extra + (size_t)p->P * 4;
        X.hi_view_stride = (size_t)p->P * 4;
    }
    return forward_impl(p, V, geom, geom_bytes, image, image_bytes, binning, binning_bytes, radii, out_color, num_rendered,
                        resume ? 1 : 0, (hipStream_t)stream, &X);
}

int gsr_forward_stage1(const gsr_params* p, void* geom, size_t geom_bytes, void* image, size_t image_bytes, int* radii,
                       int64_t* num_rendered_out, gsr_stream_t stream)
{
    if (!num_rendered_out) return fail(GSR_ERR_INVALID, "[gsr] num_rendered_out is NULL");
    return forward_impl(p, 1, geom, geom_bytes, image, image_bytes, nullptr, 0, radii, nullptr, num_rendered_out, 2, (hipStream_t)stream);
}

int gsr_forward_stage2(const gsr_params* p, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, void* image,
                       size_t image_bytes, int64_t R, float* out_color, gsr_stream_t stream)
{
    if (R < 0 || R > 0x7FFFFFFFll) return fail(GSR_ERR_INVALID, "[gsr] num_rendered out of range");
    if (p && p->P > 0 && (!binning || binning_bytes < gsr_binning_bytes(R)))
        return fail(GSR_ERR_CAPACITY, "[gsr] binning arena too small (%zu < %zu)", binning_bytes, gsr_binning_bytes(R));
    int64_t got = 0;
    const int rc = forward_impl(p, 1, geom, geom_bytes, image, image_bytes, binning, binning_bytes, nullptr, out_color, &got, 1,
                                (hipStream_t)stream);
    if (rc == GSR_OK && p && p->P > 0 && got != R) return fail(GSR_ERR_INVALID, "[gsr] stage 2 found %lld pairs, stage 1 reported %lld", (long long)got, (long long)R);
    return rc;
}

// Colour-only re-render on the geometry / lists of a finished forward (same P, views, image size).
int gsr_forward_recolor(const gsr_params* p, int V, int colors_per_view, void* geom, size_t geom_bytes, const void* binning,
                        size_t binning_bytes, void* image, size_t image_bytes, float* out_color, gsr_stream_t stream)
{
    if (!p) return fail(GSR_ERR_INVALID, "[gsr] params is NULL");
    if (V < 1 || V > MAX_VIEWS) return fail(GSR_ERR_INVALID, "[gsr] view count %d outside 1..%d", V, MAX_VIEWS);
    if (p->P <= 0) return GSR_OK;
    if ((p->shs == nullptr) == (p->colors_precomp == nullptr))
        return fail(GSR_ERR_INVALID, "Please provide excatly one of either SHs or precomputed colors!");
    if (p->shs && (!p->means3D || !p->campos || p->D < 0 || p->D > 3 || (p->D + 1) * (p->D + 1) > p->M))
        return fail(GSR_ERR_INVALID, "[gsr] recolor: bad SH arguments");
    if (!out_color || !p->bg || !binning) return fail(GSR_ERR_INVALID, "[gsr] recolor: NULL pointer");
    Batch B;
    if (int e = make_batch(p, V, geom, geom_bytes, image, image_bytes, const_cast<void*>(binning), binning_bytes, false, B)) return e;
    const Launch L{(hipStream_t)stream, p->debug};
    const int res = sorted_buffer(tile_count(p));
    {
        ProfScope ps("recolor", L.stream);
        if (int e = launch_recolor(L, *p, B, (colors_per_view && p->colors_precomp) ? (size_t)p->P * 3 : 0)) return e;
        // the render accumulates the consumed-entry counts from zero
        if (int e = memset_views(L.stream, B.iv.tile_need, B.iv_stride, (size_t)tile_count(p) * sizeof(uint32_t), V)) return e;
    }
    {
        ProfScope ps("render_forward", L.stream);
        // need_backward: the slice-boundary state and the accumulated colour are rewritten for the new colours (and k_recolor
        // refreshed the SH clamp mask), so a backward afterwards differentiates this render
        if (int e = launch_render_forward(L, *p, B, B.b.val[res], out_color, p->need_backward != 0)) return e;
    }
    return GSR_OK;
}

int gsr_backward_batch(const gsr_params* p, int V, const int* radii, const void* geom, size_t geom_bytes, const void* binning,
                       size_t binning_bytes, const void* image, size_t image_bytes, const float* dL_dpix, float* dL_dmean2D,
                       float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                       float* dL_drot, gsr_stream_t stream)
{
    if (int e = check_params(p, V)) return e;
    if (p->P == 0) return GSR_OK;
    if (!radii || !dL_dpix || !dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D)
        return fail(GSR_ERR_INVALID, "[gsr] a required backward pointer is NULL");
    if (p->shs && !dL_dsh) return fail(GSR_ERR_INVALID, "[gsr] dL_dsh is NULL");
    if (p->scales && (!dL_dscale || !dL_drot)) return fail(GSR_ERR_INVALID, "[gsr] dL_dscale/dL_drot is NULL");
    if (p->scales && ((uintptr_t)dL_drot & 15u)) return fail(GSR_ERR_INVALID, "[gsr] dL_drot must be 16-byte aligned (it is written one float4 per Gaussian)");
    if (!binning) return fail(GSR_ERR_INVALID, "[gsr] binning arena is NULL");
    Batch B;
    if (int e = make_batch(p, V, const_cast<void*>(geom), geom_bytes, const_cast<void*>(image), image_bytes,
                           const_cast<void*>(binning), binning_bytes, true, B))
        return e;
    const Launch L{(hipStream_t)stream, p->debug};
    const int res = sorted_buffer(tile_count(p));
    {
        ProfScope ps("bwd_items", L.stream);
        if (int e = launch_bwd_items(L, B, tile_count(p), p->P)) return e;
    }
    {
        ProfScope ps("render_backward", L.stream);
        if (int e = launch_render_backward(L, *p, B, B.b.val[res], dL_dpix)) return e;
    }
    {
        ProfScope ps("preprocess_backward", L.stream);
        if (int e = launch_preprocess_backward(L, *p, B, radii, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                               dL_dscale, dL_drot))
            return e;
    }
    return GSR_OK;
}

int gsr_backward(const gsr_params* p, const int* radii, int64_t R, const void* geom, size_t geom_bytes, const void* binning,
                 size_t binning_bytes, const void* image, size_t image_bytes, const float* dL_dpix, float* dL_dmean2D,
                 float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot, gsr_stream_t stream)
{
    (void)R;   // the lists' extent lives in the arenas; kept for symmetry with the reference's backward(P, D, M, R, ...)
    return gsr_backward_batch(p, 1, radii, geom, geom_bytes, binning, binning_bytes, image, image_bytes, dL_dpix, dL_dmean2D,
                              dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, stream);
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     gsr_stream_t stream)
{
    (void)projmatrix;
    if (P < 0) return fail(GSR_ERR_INVALID, "[gsr] P < 0");
    if (P == 0) return GSR_OK;
    if (!means3D || !viewmatrix || !present) return fail(GSR_ERR_INVALID, "[gsr] NULL pointer");
    const Launch L{(hipStream_t)stream, 0};
    return launch_mark_visible(L, P, means3D, viewmatrix, present);
}

// ---- inspection (tests / roofline report only) -------------------------------------------------------
namespace {
// One wave reads the shader clock (s_memtime: one tick per shader cycle) and the constant-rate wall clock (s_memrealtime) around a
// fixed chain of dependent FMAs: delta(shader) / delta(wall) x the wall clock rate is the shader clock the chip is running at
// while the probe is resident -- next to whatever else is on the device (DVFS follows the power budget, so a kernel time means
// little without the clock it was measured at).
__global__ void k_clock_probe(unsigned long long* out, int iters)
{
    const unsigned long long s0 = clock64(), w0 = wall_clock64();
    float x = (float)threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) x = __builtin_fmaf(x, 0.999f, 1e-3f);
    }
    const unsigned long long s1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = s1 - s0;
        out[1] = w1 - w0;
    }
    if (x == 12345.678f) out[1] = 0;   // keeps the chain
}
__global__ void k_query(int what, int64_t n, const Splat* __restrict__ sp, const uint8_t* __restrict__ clamped,
                        const uint32_t* __restrict__ k, int key16, const uint32_t* __restrict__ v, void* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* d = (float*)dst;
    switch (what) {
    case GSR_Q_DEPTHS: d[i] = sp[i].q2.y; break;
    case GSR_Q_MEANS2D: d[2 * i] = sp[i].q0.x; d[2 * i + 1] = sp[i].q0.y; break;
    case GSR_Q_CONIC_OPACITY:
        d[4 * i] = sp[i].q0.z; d[4 * i + 1] = sp[i].q0.w; d[4 * i + 2] = sp[i].q1.x; d[4 * i + 3] = sp[i].q1.y;
        break;
    case GSR_Q_RGB: d[3 * i] = sp[i].q1.z; d[3 * i + 1] = sp[i].q1.w; d[3 * i + 2] = sp[i].q2.x; break;
    case GSR_Q_POINT_LIST_KEYS:
        ((uint64_t*)dst)[i] = ((uint64_t)(key16 ? (uint32_t)((const uint16_t*)k)[i] : k[i]) << 32) | (uint64_t)__float_as_uint(sp[v[i]].q2.y);
        break;
    case GSR_Q_LIST_PAIRS: {
        const uint64_t* c = (const uint64_t*)k;   // k = the view's counters
        if (i == 0) { ((uint64_t*)dst)[0] = c[CNT_NUM_RENDERED]; ((uint64_t*)dst)[1] = c[CNT_NUM_REFERENCE]; }
        break;
    }
    case GSR_Q_DEPTH_SORT: {
        uint32_t* o = (uint32_t*)dst;   // k = the view's sortctl words
        if (i == 0) { o[0] = k[SORTCTL_BASE]; o[1] = k[SORTCTL_BITS]; o[2] = depth_sort_passes(k[SORTCTL_BITS]); o[3] = 0u; }
        break;
    }
    case GSR_Q_CLAMPED: {
        uint8_t* c = (uint8_t*)dst;
        c[3 * i] = clamped[i] & 1; c[3 * i + 1] = (clamped[i] >> 1) & 1; c[3 * i + 2] = (clamped[i] >> 2) & 1;
        break;
    }
    default: break;
    }
}
}  // namespace

int gsr_query(const gsr_params* p, int what, const void* geom, const void* binning, size_t binning_bytes, const void* image,
              int64_t R, void* dst, size_t dst_bytes, gsr_stream_t stream)
{
    if (!p || !dst) return fail(GSR_ERR_INVALID, "[gsr] query: NULL");
    hipStream_t s = (hipStream_t)stream;
    const int P = p->P, T = tile_count(p);
    const int64_t N = (int64_t)p->W * p->H;
    const GeomView g = geom_view(align256(const_cast<void*>(geom)), P);
    if (binning && binning_bytes < 512) return fail(GSR_ERR_CAPACITY, "[gsr] query: binning arena too small");
    const int64_t cap = binning ? bin_capacity_from_bytes(((binning_bytes - 256)) / 256 * 256) : 0;
    if (binning && R > cap && (what == GSR_Q_POINT_LIST || what == GSR_Q_POINT_LIST_KEYS))
        return fail(GSR_ERR_CAPACITY, "[gsr] query: arena holds %lld pairs, asked for %lld (the lists hold GSR_Q_LIST_PAIRS pairs)", (long long)cap, (long long)R);
    const BinView b = bin_view(align256(const_cast<void*>(binning)), cap > 0 ? cap : 1);
    const ImageView iv = image_view(align256(const_cast<void*>(image)), p->W, p->H);
    const int res = sorted_buffer(T);
    int64_t n = 0;
    size_t bytes = 0;
    const void* src = nullptr;
    switch (what) {
    case GSR_Q_DEPTHS: n = P; bytes = (size_t)P * 4; break;
    case GSR_Q_MEANS2D: n = P; bytes = (size_t)P * 8; break;
    case GSR_Q_CONIC_OPACITY: n = P; bytes = (size_t)P * 16; break;
    case GSR_Q_RGB: n = P; bytes = (size_t)P * 12; break;
    case GSR_Q_CLAMPED: n = P; bytes = (size_t)P * 3; break;
    case GSR_Q_POINT_LIST_KEYS: n = R; bytes = (size_t)R * 8; break;
    case GSR_Q_TILES_TOUCHED: src = g.tiles_touched; bytes = (size_t)P * 4; break;
    case GSR_Q_POINT_LIST: src = b.val[res]; bytes = (size_t)R * 4; break;
    case GSR_Q_RANGES: src = iv.ranges; bytes = (size_t)T * 8; break;
    case GSR_Q_FINAL_T: src = iv.final_T; bytes = (size_t)N * 4; break;
    case GSR_Q_N_CONTRIB: src = iv.n_contrib; bytes = (size_t)N * 4; break;
    case GSR_Q_TILE_NEED: src = iv.tile_need; bytes = (size_t)T * 4; break;
    case GSR_Q_DEPTH_SORT: n = 1; bytes = 16; break;
    case GSR_Q_LIST_PAIRS: n = 1; bytes = 16; break;
    default: return fail(GSR_ERR_INVALID, "[gsr] query: unknown item %d", what);
    }
    if (dst_bytes < bytes) return fail(GSR_ERR_CAPACITY, "[gsr] query %d: destination too small", what);
    if (bytes == 0) return GSR_OK;
    if (src) {
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] query copy failed");
    } else {
        hipLaunchKernelGGL(k_query, dim3((unsigned)div_up(n, 256)), dim3(256), 0, s, what, n, g.splat, g.clamped,
                           what == GSR_Q_DEPTH_SORT ? (const uint32_t*)g.sortctl
                           : what == GSR_Q_LIST_PAIRS ? (const uint32_t*)g.counters : (const uint32_t*)b.key[res],
                           (int)tile_keys16(T), b.val[res], dst);
        if (hipGetLastError() != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] query kernel failed");
    }
    return GSR_OK;
}

// Device self-test of internal primitives that have no observable output of their own: the matrix-core pixel
// contraction of the render backward, and the stable radix sort against std::stable_sort on random keys with many ties.
int gsr_selftest(gsr_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    float* d = nullptr;
    if (hipMalloc(&d, 256 * sizeof(float)) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] selftest: hipMalloc failed");
    const int rm = selftest_mm(s, d);
    (void)hipFree(d);
    if (rm != 0) return fail(GSR_ERR_HIP, "[gsr] selftest: matrix-core pixel contraction wrong at check %d", rm);

    const int64_t n = 100003;
    std::vector<uint32_t> hk(n), hv(n), order(n);
    uint32_t x = 12345u;
    for (int64_t i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        hk[i] = (x >> 8) % 3001u;  // many ties
        hv[i] = (uint32_t)i;
        order[i] = (uint32_t)i;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hk[a] < hk[b]; });
    uint32_t *k0, *k1, *v0, *v1, *hist, *tot;
    const size_t nb = (size_t)sort_hist_stride(n);
    if (hipMalloc(&k0, n * 4) != hipSuccess || hipMalloc(&k1, n * 4) != hipSuccess || hipMalloc(&v0, n * 4) != hipSuccess ||
        hipMalloc(&v1, n * 4) != hipSuccess || hipMalloc(&hist, RADIX * nb * 4) != hipSuccess || hipMalloc(&tot, RADIX * 4) != hipSuccess)
        return fail(GSR_ERR_HIP, "[gsr] selftest: hipMalloc failed");
    (void)hipMemcpyAsync(k0, hk.data(), n * 4, hipMemcpyHostToDevice, s);
    uint32_t* key[2] = {k0, k1};
    uint32_t* val[2] = {v0, v1};
    int res = 0;
    const Launch L{s, 1};
    const SortJob job{{k0, k1}, {v0, v1}, hist, tot, 0, nullptr, 0, n, 1};
    int rc = launch_radix_sort_pairs(L, job, /*iota_vals=*/true, 12, &res);
    std::vector<uint32_t> gk(n), gv(n);
    if (rc == 0) {
        (void)hipMemcpyAsync(gk.data(), key[res], n * 4, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(gv.data(), val[res], n * 4, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) rc = GSR_ERR_HIP;
    }
    (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1); (void)hipFree(hist); (void)hipFree(tot);
    if (rc != 0) return rc;
    for (int64_t i = 0; i < n; i++)
        if (gv[i] != order[i] || gk[i] != hk[order[i]]) return fail(GSR_ERR_HIP, "[gsr] selftest: radix sort differs from std::stable_sort at %lld", (long long)i);
    return selftest_lookback(s, hk, order);
}

int gsr_clock_probe_launch(void* dst16, int iters, gsr_stream_t stream)
{
    if (!dst16 || iters < 1) return fail(GSR_ERR_INVALID, "[gsr] clock probe: bad argument");
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)dst16, iters);
    if (hipGetLastError() != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] clock probe launch failed");
    return GSR_OK;
}

int gsr_wall_clock_khz(void)
{
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}

void gsr_set_profiling(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on < 0 ? 0 : on > 2 ? 1 : on;
    for (auto& kv : g_prof) { kv.second.rec.clear(); kv.second.used = 0; }
}

int gsr_get_profile(const char** names, float* ms, int cap)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& kv : g_prof) {
        ProfTables& t = kv.second;
        for (auto& e : t.rec) {
            if (n >= cap) break;
            (void)hipEventSynchronize(t.pool[e.b]);
            float d = 0;
            (void)hipEventElapsedTime(&d, t.pool[e.a], t.pool[e.b]);
            names[n] = e.name;
            ms[n] = d;
            n++;
        }
        t.rec.clear();
        t.used = 0;
    }
    return n;
}

#ifdef GSR_STATS
__attribute__((visibility("default"))) int gsr_debug_bwd_stats(unsigned long long* out8, int reset) { return gsr::debug_bwd_stats(out8, reset); }
__attribute__((visibility("default"))) int gsr_debug_scatter_times(unsigned long long* out8, int reset) { return gsr::debug_scatter_times(out8, reset); }
__attribute__((visibility("default"))) int gsr_debug_fwd_times(unsigned long long* out8, int reset) { return gsr::debug_fwd_times(out8, reset); }
__attribute__((visibility("default"))) int gsr_debug_fwd_records(unsigned* out, int n) { return gsr::debug_fwd_records(out, n); }
__attribute__((visibility("default"))) int gsr_debug_bwd_times(unsigned long long* out8, int reset) { return gsr::debug_bwd_times(out8, reset); }
__attribute__((visibility("default"))) int gsr_debug_bwd_records(unsigned* out, int n) { return gsr::debug_bwd_records(out, n); }
__attribute__((visibility("default"))) int gsr_debug_dup_times(unsigned long long* out8, int reset) { return gsr::debug_dup_times(out8, reset); }
#endif

long long gsr_d2h_count(void) { return gsr::g_d2h_count.load(); }
int gsr_set_forward_half_views(int views) { return gsr::forward_half_views(views); }
int gsr_set_sort_mode(int mode, int lookback_views)
{
    if (lookback_views >= 0) (void)gsr::lookback_max_views(lookback_views);
    return gsr::sort_mode(mode);
}
int gsr_set_backward_moments(int mode) { return gsr::backward_subquadrant_moments(mode); }

int gsr_last_list_pairs(int64_t* out, int V)
{
    if (!out || V < 0) return fail(GSR_ERR_INVALID, "[gsr] gsr_last_list_pairs: bad argument");
    for (int v = 0; v < V; v++) out[v] = (size_t)v < gsr::t_list_pairs.size() ? gsr::t_list_pairs[(size_t)v] : 0;
    return GSR_OK;
}

const char* gsr_last_error(void) { return gsr::g_err; }
const char* gsr_version(void) { return "gsr-hip 0.3 (gfx950)"; }

}  // extern "C"
