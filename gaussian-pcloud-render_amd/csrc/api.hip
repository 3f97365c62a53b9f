// api.hip -- the C ABI of include/gsr.h: argument checks, arena carving, kernel sequencing.
//
// Host orchestration corresponding to reference CR/rasterizer_impl.cu:198-336 (Rasterizer::forward) and
// :340-434 (Rasterizer::backward), re-planned for this library's data flow:
//
//   stage 1   preprocess                      (1 launch)
//             depth sort of the P Gaussians   (4 passes x 3 launches, u32 key / u32 id)
//             offsets scan in depth order     (3 launches)  -> num_rendered
//             8-byte D2H read-back + stream sync            (the reference's cudaMemcpy at :281)
//   stage 2   pair emission in depth order    (1 launch)
//             tile sort                       (ceil(bit/8) passes x 3 launches, u32 tile / u32 id)
//             tile ranges (+memset)           (1 launch)
//             tile order by list length       (3 tiny launches)
//             render                          (1 launch)
//   backward  render backward, per-Gaussian backward (2 launches)
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include <atomic>
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

// ---- optional per-step timing (hipEvents on the launch stream, pooled) ---------------------------
struct ProfEntry {
    const char* name;
    int a, b;  // indices into the event pool
};
// (several host threads may render at once: the tables are guarded, the flag is atomic)
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_used = 0;
static std::vector<ProfEntry> g_prof;

static int pool_event()
{
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_pool.push_back(e);
    }
    return (int)g_pool_used++;
}

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfEntry e;
    ProfScope(const char* name, hipStream_t stream) : s(stream), on(g_prof_on)
    {
        if (!on) return;
        e.name = name;
        hipEvent_t ea;
        {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            e.a = pool_event();
            e.b = pool_event();
            ea = g_pool[e.a];
        }
        (void)hipEventRecord(ea, s);
    }
    ~ProfScope()
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        (void)hipEventRecord(g_pool[e.b], s);
        g_prof.push_back(e);
    }
};

static int check_params(const gsr_params* p)
{
    if (!p) return fail(GSR_ERR_INVALID, "[gsr] params is NULL");
    if (p->P < 0 || p->W <= 0 || p->H <= 0) return fail(GSR_ERR_INVALID, "[gsr] bad sizes P=%d W=%d H=%d", p->P, p->W, p->H);
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
    const int gx = (p->W + TILE_X - 1) / TILE_X, gy = (p->H + TILE_Y - 1) / TILE_Y;
    if (gx > 65535 || gy > 65535) return fail(GSR_ERR_INVALID, "[gsr] image too large for 16-bit tile coordinates");
    return GSR_OK;
}

static int tile_count(const gsr_params* p) { return ((p->W + TILE_X - 1) / TILE_X) * ((p->H + TILE_Y - 1) / TILE_Y); }

// how many u32 tile-key bits the tile sort covers: the reference's 32+bit minus the 32 depth bits
static int tile_bits(int T) { return (int)higher_msb((uint32_t)T); }

}  // namespace gsr

using namespace gsr;

extern "C" {

size_t gsr_geom_bytes(int P) { return geom_view(nullptr, P).bytes + 256; }
size_t gsr_image_bytes(int W, int H) { return image_view(nullptr, W, H).bytes + 256; }
size_t gsr_binning_bytes(int64_t R) { return bin_view(nullptr, R).bytes + 256; }

static inline void* align256(void* p) { return (void*)(((uintptr_t)p + 255) & ~(uintptr_t)255); }

int gsr_forward_stage1(const gsr_params* p, void* geom, size_t geom_bytes, void* image, size_t image_bytes, int* radii,
                       int64_t* num_rendered_out, gsr_stream_t stream)
{
    if (int e = check_params(p)) return e;
    if (!num_rendered_out) return fail(GSR_ERR_INVALID, "[gsr] num_rendered_out is NULL");
    *num_rendered_out = 0;
    if (p->P == 0) return GSR_OK;  // reference rasterize_points.cu:81: nothing runs, image stays zero
    if (!geom || geom_bytes < gsr_geom_bytes(p->P)) return fail(GSR_ERR_CAPACITY, "[gsr] geom arena too small (%zu < %zu)", geom_bytes, gsr_geom_bytes(p->P));
    if (!image || image_bytes < gsr_image_bytes(p->W, p->H)) return fail(GSR_ERR_CAPACITY, "[gsr] image arena too small");
    if (!radii) return fail(GSR_ERR_INVALID, "[gsr] radii is NULL");
    const Launch L{(hipStream_t)stream, p->debug};
    const GeomView g = geom_view(align256(geom), p->P);

    if (hipMemsetAsync(g.counters, 0, 8 * sizeof(uint64_t), L.stream) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] memset failed");
    {
        ProfScope ps("preprocess", L.stream);
        if (int e = launch_preprocess(L, *p, g, radii)) return e;
    }
    int res = 0;
    {
        ProfScope ps("depth_sort", L.stream);
        uint32_t* key[2] = {g.dkey[0], g.dkey[1]};
        uint32_t* val[2] = {g.dval[0], g.dval[1]};
        if (int e = launch_radix_sort_pairs(L, p->P, key, val, /*iota_vals=*/true, 32, g.hist, g.totals, &res)) return e;
    }
    // 4 passes: the ids in depth order are back in buffer 0
    {
        ProfScope ps("offsets_scan", L.stream);
        if (int e = launch_offsets_scan(L, p->P, g.dval[res], g.tiles_touched, g.dup_offset, g.scan_tmp, g.counters)) return e;
    }
    uint64_t host[2] = {0, 0};
    if (hipMemcpyAsync(host, g.counters, sizeof(host), hipMemcpyDeviceToHost, L.stream) != hipSuccess ||
        hipStreamSynchronize(L.stream) != hipSuccess)
        return fail(GSR_ERR_HIP, "[gsr] num_rendered read-back failed: %s", hipGetErrorString(hipGetLastError()));
    if (host[1]) return fail(GSR_ERR_TRAP, "Point is filtered although prefiltered is set. This shouldn't happen!");
    // the reference keeps num_rendered in an int (CR/rasterizer_impl.cu:280); beyond that its arena sizes wrap
    if (host[0] > 0x7FFFFFFFull)
        return fail(GSR_ERR_CAPACITY, "[gsr] num_rendered = %llu tile pairs does not fit the reference's int (scales too large?)",
                    (unsigned long long)host[0]);
    *num_rendered_out = (int64_t)host[0];
    return GSR_OK;
}

int gsr_forward_stage2(const gsr_params* p, void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, void* image,
                       size_t image_bytes, int64_t R, float* out_color, gsr_stream_t stream)
{
    if (int e = check_params(p)) return e;
    if (p->P == 0) return GSR_OK;
    if (!out_color) return fail(GSR_ERR_INVALID, "[gsr] out_color is NULL");
    if (R < 0 || R > 0xFFFFFFFFll) return fail(GSR_ERR_INVALID, "[gsr] num_rendered out of range");
    if (!geom || geom_bytes < gsr_geom_bytes(p->P)) return fail(GSR_ERR_CAPACITY, "[gsr] geom arena too small");
    if (!image || image_bytes < gsr_image_bytes(p->W, p->H)) return fail(GSR_ERR_CAPACITY, "[gsr] image arena too small");
    if (!binning || binning_bytes < gsr_binning_bytes(R)) return fail(GSR_ERR_CAPACITY, "[gsr] binning arena too small (%zu < %zu)", binning_bytes, gsr_binning_bytes(R));
    const Launch L{(hipStream_t)stream, p->debug};
    const GeomView g = geom_view(align256(geom), p->P);
    const BinView b = bin_view(align256(binning), R);
    const ImageView iv = image_view(align256(image), p->W, p->H);
    const int T = tile_count(p);
    const int gridx = (p->W + TILE_X - 1) / TILE_X;

    int res = 0;
    if (R > 0) {
        {
            ProfScope ps("duplicate", L.stream);
            if (int e = launch_duplicate(L, p->P, g, g.dval[0], gridx, b.key[0], b.val[0], tile_keys16(T))) return e;
        }
        {
            ProfScope ps("tile_sort", L.stream);
            uint32_t* key[2] = {b.key[0], b.key[1]};
            uint32_t* val[2] = {b.val[0], b.val[1]};
            if (int e = launch_radix_sort_pairs(L, R, key, val, false, tile_bits(T), b.hist, b.totals, &res, tile_keys16(T))) return e;
        }
    }
    {
        ProfScope ps("tile_ranges", L.stream);
        if (int e = launch_tile_ranges(L, R, b.key[res], iv.ranges, T, tile_keys16(T))) return e;
        if (int e = launch_tile_order(L, iv, T)) return e;
    }
    {
        ProfScope ps("render_forward", L.stream);
        if (int e = launch_render_forward(L, *p, g, b.val[res], iv, out_color, p->need_backward ? b.ckpt : nullptr)) return e;
    }
    return GSR_OK;
}

// Colour-only re-render on the geometry / lists of a finished forward (same P, view, image size).
int gsr_forward_recolor(const gsr_params* p, void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes, void* image,
                        size_t image_bytes, int64_t R, float* out_color, gsr_stream_t stream)
{
    if (!p) return fail(GSR_ERR_INVALID, "[gsr] params is NULL");
    if (p->P <= 0) return GSR_OK;
    if ((p->shs == nullptr) == (p->colors_precomp == nullptr))
        return fail(GSR_ERR_INVALID, "Please provide excatly one of either SHs or precomputed colors!");
    if (p->shs && (!p->means3D || !p->campos || p->D < 0 || p->D > 3 || (p->D + 1) * (p->D + 1) > p->M))
        return fail(GSR_ERR_INVALID, "[gsr] recolor: bad SH arguments");
    if (!out_color || !p->bg) return fail(GSR_ERR_INVALID, "[gsr] recolor: NULL pointer");
    if (R < 0 || R > 0x7FFFFFFFll) return fail(GSR_ERR_INVALID, "[gsr] num_rendered out of range");
    if (!geom || geom_bytes < gsr_geom_bytes(p->P) || !image || image_bytes < gsr_image_bytes(p->W, p->H) || !binning ||
        binning_bytes < gsr_binning_bytes(R))
        return fail(GSR_ERR_CAPACITY, "[gsr] an arena is too small for recolor");
    const Launch L{(hipStream_t)stream, p->debug};
    const GeomView g = geom_view(align256(geom), p->P);
    const BinView b = bin_view(align256(const_cast<void*>(binning)), R);
    const ImageView iv = image_view(align256(image), p->W, p->H);
    const int passes = (tile_bits(tile_count(p)) + RADIX_BITS - 1) / RADIX_BITS;
    const int res = R > 0 ? (passes & 1) : 0;
    {
        ProfScope ps("recolor", L.stream);
        if (int e = launch_recolor(L, *p, g)) return e;
    }
    {
        ProfScope ps("render_forward", L.stream);
        if (int e = launch_render_forward(L, *p, g, b.val[res], iv, out_color, nullptr)) return e;
    }
    return GSR_OK;
}

int gsr_backward(const gsr_params* p, const int* radii, int64_t R, const void* geom, size_t geom_bytes, const void* binning,
                 size_t binning_bytes, const void* image, size_t image_bytes, const float* dL_dpix, float* dL_dmean2D,
                 float* grad_rec, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot, gsr_stream_t stream)
{
    if (int e = check_params(p)) return e;
    if (p->P == 0) return GSR_OK;
    if (!radii || !dL_dpix || !dL_dmean2D || !grad_rec || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D)
        return fail(GSR_ERR_INVALID, "[gsr] a required backward pointer is NULL");
    if (p->shs && !dL_dsh) return fail(GSR_ERR_INVALID, "[gsr] dL_dsh is NULL");
    if (p->scales && (!dL_dscale || !dL_drot)) return fail(GSR_ERR_INVALID, "[gsr] dL_dscale/dL_drot is NULL");
    if (!geom || geom_bytes < gsr_geom_bytes(p->P) || !image || image_bytes < gsr_image_bytes(p->W, p->H) || !binning ||
        binning_bytes < gsr_binning_bytes(R))
        return fail(GSR_ERR_CAPACITY, "[gsr] an arena is too small for backward");
    const Launch L{(hipStream_t)stream, p->debug};
    const GeomView g = geom_view(align256(const_cast<void*>(geom)), p->P);
    const BinView b = bin_view(align256(const_cast<void*>(binning)), R);
    const ImageView iv = image_view(align256(const_cast<void*>(image)), p->W, p->H);
    const int passes = (tile_bits(tile_count(p)) + RADIX_BITS - 1) / RADIX_BITS;
    const int res = R > 0 ? (passes & 1) : 0;
    {
        ProfScope ps("bwd_items", L.stream);
        if (int e = launch_bwd_items(L, iv, tile_count(p))) return e;
    }
    {
        ProfScope ps("render_backward", L.stream);
        if (int e = launch_render_backward(L, *p, g, b.val[res], iv, b.ckpt, dL_dpix, grad_rec)) return e;
    }
    {
        ProfScope ps("preprocess_backward", L.stream);
        if (int e = launch_preprocess_backward(L, *p, g, radii, grad_rec, dL_dmean2D, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                               dL_dscale, dL_drot))
            return e;
    }
    return GSR_OK;
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
    case GSR_Q_CLAMPED: {
        uint8_t* c = (uint8_t*)dst;
        c[3 * i] = clamped[i] & 1; c[3 * i + 1] = (clamped[i] >> 1) & 1; c[3 * i + 2] = (clamped[i] >> 2) & 1;
        break;
    }
    default: break;
    }
}
}  // namespace

int gsr_query(const gsr_params* p, int what, const void* geom, const void* binning, const void* image, int64_t R, void* dst,
              size_t dst_bytes, gsr_stream_t stream)
{
    if (!p || !dst) return fail(GSR_ERR_INVALID, "[gsr] query: NULL");
    hipStream_t s = (hipStream_t)stream;
    const int P = p->P, T = tile_count(p);
    const int64_t N = (int64_t)p->W * p->H;
    const GeomView g = geom_view(align256(const_cast<void*>(geom)), P);
    const BinView b = bin_view(align256(const_cast<void*>(binning)), R);
    const ImageView iv = image_view(align256(const_cast<void*>(image)), p->W, p->H);
    const int passes = (tile_bits(T) + RADIX_BITS - 1) / RADIX_BITS;
    const int res = R > 0 ? (passes & 1) : 0;
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
    default: return fail(GSR_ERR_INVALID, "[gsr] query: unknown item %d", what);
    }
    if (dst_bytes < bytes) return fail(GSR_ERR_CAPACITY, "[gsr] query %d: destination too small", what);
    if (bytes == 0) return GSR_OK;
    if (src) {
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] query copy failed");
    } else {
        hipLaunchKernelGGL(k_query, dim3((unsigned)div_up(n, 256)), dim3(256), 0, s, what, n, g.splat, g.clamped, b.key[res],
                           (int)tile_keys16(T), b.val[res], dst);
        if (hipGetLastError() != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] query kernel failed");
    }
    return GSR_OK;
}

// Device self-test of internal primitives that have no observable output of their own: the transposed wave
// reduction of the backward pass, and the stable radix sort against std::stable_sort on random keys with many ties.
int gsr_selftest(gsr_stream_t stream)
{
    hipStream_t s = (hipStream_t)stream;
    float* d = nullptr;
    if (hipMalloc(&d, 128 * sizeof(float)) != hipSuccess) return fail(GSR_ERR_HIP, "[gsr] selftest: hipMalloc failed");
    const int rr = selftest_reduce(s, d);
    (void)hipFree(d);
    if (rr != 0) return fail(GSR_ERR_HIP, "[gsr] selftest: wave reduction wrong at check %d", rr);

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
    const size_t nb = (size_t)div_up(n, RS_TILE);
    if (hipMalloc(&k0, n * 4) != hipSuccess || hipMalloc(&k1, n * 4) != hipSuccess || hipMalloc(&v0, n * 4) != hipSuccess ||
        hipMalloc(&v1, n * 4) != hipSuccess || hipMalloc(&hist, RADIX * nb * 4) != hipSuccess || hipMalloc(&tot, RADIX * 4) != hipSuccess)
        return fail(GSR_ERR_HIP, "[gsr] selftest: hipMalloc failed");
    (void)hipMemcpyAsync(k0, hk.data(), n * 4, hipMemcpyHostToDevice, s);
    uint32_t* key[2] = {k0, k1};
    uint32_t* val[2] = {v0, v1};
    int res = 0;
    const Launch L{s, 1};
    int rc = launch_radix_sort_pairs(L, n, key, val, /*iota_vals=*/true, 12, hist, tot, &res);
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
    return GSR_OK;
}

void gsr_set_profiling(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    g_prof.clear();
    g_pool_used = 0;
}

int gsr_get_profile(const char** names, float* ms, int cap)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& e : g_prof) {
        if (n >= cap) break;
        (void)hipEventSynchronize(g_pool[e.b]);
        float t = 0;
        (void)hipEventElapsedTime(&t, g_pool[e.a], g_pool[e.b]);
        names[n] = e.name;
        ms[n] = t;
        n++;
    }
    g_prof.clear();
    g_pool_used = 0;
    return n;
}

const char* gsr_last_error(void) { return gsr::g_err; }
const char* gsr_version(void) { return "gsr-hip 0.1 (gfx950)"; }

}  // extern "C"
