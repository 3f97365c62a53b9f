// preprocess.hip -- per-Gaussian forward stage: near-plane cull, clip-space projection, 3D->2D (EWA)
// covariance projection with 0.3 px^2 dilation, conic, 3-sigma radius, tile rectangle, SH->RGB.
//
// Behaviour follows reference CR/forward.cu:158-259 (preprocessCUDA) + CR/auxiliary.h:41-56,139-164
// (ndc2Pix, getRect, in_frustum).  Layout is ours: one thread per Gaussian, inputs are the caller's
// AoS tensors read as contiguous 12/16/24-B runs per lane, the outputs the render kernels need are
// packed into one 48-B Splat record per Gaussian (three coalesced 16-B stores per lane) and the
// sort key (depth bits) is emitted directly, so nothing is re-read before the depth sort.
#include "common.hpp"
#include "splat_math.hpp"
#include "tile_cull.hpp"

namespace gsr {

// SH -> RGB, reference CR/forward.cu:20-71.  `sh` points at this Gaussian's first coefficient
// (stride M rows of 3 floats; only (deg+1)^2 rows are read).
template <int deg>
__device__ __forceinline__ V3 sh_to_rgb_t(V3 pos, V3 campos, const float* sh, uint32_t* clamp_mask)
{
    V3 dir = pos - campos;
    const float len = sqrtf(dot3(dir, dir));
    dir = v3(dir.x / len, dir.y / len, dir.z / len);
#define SHV(k) v3(sh[3 * (k)], sh[3 * (k) + 1], sh[3 * (k) + 2])
    V3 result = kSH_C0 * SHV(0);
    if constexpr (deg > 0) {
        const float x = dir.x, y = dir.y, z = dir.z;
        result = ((result - (kSH_C1 * y) * SHV(1)) + (kSH_C1 * z) * SHV(2)) - (kSH_C1 * x) * SHV(3);
        if constexpr (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            result = result + (kSH_C2[0] * xy) * SHV(4);
            result = result + (kSH_C2[1] * yz) * SHV(5);
            result = result + (kSH_C2[2] * (2.0f * zz - xx - yy)) * SHV(6);
            result = result + (kSH_C2[3] * xz) * SHV(7);
            result = result + (kSH_C2[4] * (xx - yy)) * SHV(8);
            if constexpr (deg > 2) {
                result = result + (kSH_C3[0] * y * (3.0f * xx - yy)) * SHV(9);
                result = result + (kSH_C3[1] * xy * z) * SHV(10);
                result = result + (kSH_C3[2] * y * (4.0f * zz - xx - yy)) * SHV(11);
                result = result + (kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * SHV(12);
                result = result + (kSH_C3[4] * x * (4.0f * zz - xx - yy)) * SHV(13);
                result = result + (kSH_C3[5] * z * (xx - yy)) * SHV(14);
                result = result + (kSH_C3[6] * x * (xx - 3.0f * yy)) * SHV(15);
            }
        }
    }
#undef SHV
    result.x += 0.5f;
    result.y += 0.5f;
    result.z += 0.5f;
    *clamp_mask = (result.x < 0 ? 1u : 0u) | (result.y < 0 ? 2u : 0u) | (result.z < 0 ? 4u : 0u);
    return v3(fmax_(result.x, 0.0f), fmax_(result.y, 0.0f), fmax_(result.z, 0.0f));
}

// runtime-degree front end (k_recolor)
__device__ __forceinline__ V3 sh_to_rgb(int deg, V3 pos, V3 campos, const float* __restrict__ sh, uint32_t* clamp_mask)
{
    switch (deg) {
    case 0: return sh_to_rgb_t<0>(pos, campos, sh, clamp_mask);
    case 1: return sh_to_rgb_t<1>(pos, campos, sh, clamp_mask);
    case 2: return sh_to_rgb_t<2>(pos, campos, sh, clamp_mask);
    default: return sh_to_rgb_t<3>(pos, campos, sh, clamp_mask);
    }
}

// (v+1)*S-1)/2 evaluated in double, rounded once (reference CR/auxiliary.h:41-44)
__device__ __forceinline__ float ndc_to_pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

__device__ __forceinline__ uint32_t clamp_tile(float f, uint32_t grid)
{
    int v = (int)f;  // truncation toward zero, as the C cast in getRect
    v = v > 0 ? v : 0;
    return grid < (uint32_t)v ? grid : (uint32_t)v;
}

struct PreArgs {
    int P, D, M, W, H, V, vpt;            // vpt: views per thread (grid.y = ceil(V / vpt))
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    int prefiltered, need_backward, reference_lists;
    uint32_t gridx, gridy;
    const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *view, *proj, *campos;   // [V][16], [V][16], [V][3]
    Splat* splat;
    uint32_t* tiles_touched;
    uint8_t* clamped;
    uint32_t* dkey;
    uint32_t* pre_minmax;                 // [2][n_pre] per view: smallest / largest depth key of a visible Gaussian per workgroup
    int n_pre;
    float* grad_rec;
    uint64_t* counters;
    char* g_zero;                         // per-frame cleared regions of the geometry / image arenas
    size_t g_zero_bytes, g_stride, gr_stride;
    char* iv_zero;
    size_t iv_zero_bytes, iv_stride;
    int* radii;                           // [V][P]
};

// DEG = active SH degree (template parameter: the 3 (DEG+1)^2 coefficients a Gaussian needs are read ONCE into registers and
// reused for every view of the batch -- re-reading them per view pulled the 156-B-stride SH rows from HBM five times over)
// (135 registers at degree 1 = three waves per SIMD; forcing four -- amdgpu_waves_per_eu(4, 4), 128 registers -- spills: 0.029 ->
// 0.031 ms per view at 12 views per call, 0.067 -> 0.187 at one)
constexpr int MAX_VIEWS_PER_THREAD = 256;   // (= api.hip MAX_VIEWS)
template <int DEG>
__global__ __launch_bounds__(256) void k_preprocess(PreArgs a)
{
    // A thread owns one Gaussian for a.vpt consecutive views of the batch: the inputs (mean, scale / rotation -> 3D
    // covariance, opacity, precomputed colour) are read and prepared ONCE, only the per-view part is repeated, so a batch of V
    // views reads the cloud once instead of V times.
    const int v_first = blockIdx.y * a.vpt, v_last = v_first + a.vpt < a.V ? v_first + a.vpt : a.V;
    const int idx_raw = blockIdx.x * 256 + threadIdx.x;
    // this frame's bookkeeping that later kernels accumulate into (prefix-sum status words; consumed-entry counts, backward
    // item count) is cleared here, so a frame needs no memset launches
    {
        const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, nthr = (size_t)gridDim.x * 256;
        for (int vw = v_first; vw < v_last; vw++) {
            if (gtid == 0) {
                at_view(a.counters, a.g_stride, vw)[CNT_STALL] = 0;
                at_view(a.counters, a.g_stride, vw)[CNT_BWD_DIRTY] = 0;   // the gradient records are cleared below
            }
            zero_region(a.g_zero + a.g_stride * vw, a.g_zero_bytes, gtid, nthr);
            zero_region(a.iv_zero + a.iv_stride * vw, a.iv_zero_bytes, gtid, nthr);
        }
    }
    // Threads past the end of the cloud stay (the workgroup meets at a barrier below): they recompute the last Gaussian and
    // store nothing.
    const bool valid = idx_raw < a.P;
    const int idx = valid ? idx_raw : a.P - 1;
    __shared__ float4 xpose[4][256];   // per wave: 64 Splat lines on their way to coalesced stores
    __shared__ uint32_t s_mm[MAX_VIEWS_PER_THREAD][4][2];   // per view of this thread and wave: depth-key extremes

    const V3 p_orig = v3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
    float cov6[6];
    if (a.cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) cov6[i] = a.cov3D_precomp[6 * (size_t)idx + i];
    } else {
        const V3 sc = v3(a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]);
        const float4 q = *reinterpret_cast<const float4*>(a.rotations + 4 * (size_t)idx);
        cov3d_from_scale_rot(sc, a.scale_modifier, q, cov6, nullptr);
    }
    const float opacity = a.opacities[idx];
    V3 rgb_pre = v3(0, 0, 0);
    constexpr int NSH = 3 * (DEG + 1) * (DEG + 1);
    float shv[NSH];
    if (a.colors_precomp) {
        rgb_pre = v3(a.colors_precomp[3 * idx], a.colors_precomp[3 * idx + 1], a.colors_precomp[3 * idx + 2]);
    } else {
        const float* shp = a.shs + (size_t)idx * a.M * 3;
#pragma unroll
        for (int i = 0; i < NSH; i++) shv[i] = shp[i];
    }

    for (int vw = v_first; vw < v_last; vw++) {
        // uniform data: 35 scalar loads, served by the scalar cache
        float view[16], proj[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            view[i] = a.view[16 * vw + i];
            proj[i] = a.proj[16 * vw + i];
        }

        int radius_out = 0;
        uint32_t tiles = 0, etiles = 0, key = 0xFFFFFFFFu, cmask = 0;
        uint2 rect = make_uint2(0, 0), spans = make_uint2(0, 0);
        bool spans_valid = false;
        Splat s;
        s.q0 = make_float4(0, 0, 0, 0);
        s.q1 = make_float4(0, 0, 0, 0);
        s.q2 = make_float4(0, 0, 0, 0);

        const V3 p_view = xform_point_4x3(p_orig, view);
        if (p_view.z <= 0.2f) {
            if (a.prefiltered && valid) at_view(a.counters, a.g_stride, vw)[CNT_TRAP] = 1;  // reference: printf + __trap() (CR/auxiliary.h:156-160)
        } else {
            const float hx = ((proj[0] * p_orig.x + proj[4] * p_orig.y) + proj[8] * p_orig.z) + proj[12];
            const float hy = ((proj[1] * p_orig.x + proj[5] * p_orig.y) + proj[9] * p_orig.z) + proj[13];
            const float hw = ((proj[3] * p_orig.x + proj[7] * p_orig.y) + proj[11] * p_orig.z) + proj[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float proj_x = hx * p_w, proj_y = hy * p_w;

            const Cov2D c = cov2d_project(p_orig, a.focal_x, a.focal_y, a.tanfovx, a.tanfovy, cov6, view);
            const float cov_x = c.cov.m[0][0] + 0.3f;  // dilation is ON in this fork (CR/forward.cu:112-113)
            const float cov_y = c.cov.m[0][1];
            const float cov_z = c.cov.m[1][1] + 0.3f;

            const float det = (cov_x * cov_z - cov_y * cov_y);
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float conic_x = cov_z * det_inv, conic_y = -cov_y * det_inv, conic_z = cov_x * det_inv;

                const float mid = 0.5f * (cov_x + cov_z);
                const float lambda1 = mid + sqrtf(fmax_(0.1f, mid * mid - det));
                const float lambda2 = mid - sqrtf(fmax_(0.1f, mid * mid - det));
                const float my_radius = ceilf(3.f * sqrtf(fmax_(lambda1, lambda2)));
                const float px = ndc_to_pix(proj_x, a.W), py = ndc_to_pix(proj_y, a.H);

                // getRect (CR/auxiliary.h:46-56); max_radius is an int there
                const float r = (float)(int)my_radius;
                const uint32_t minx = clamp_tile((px - r) / (float)TILE_X, a.gridx);
                const uint32_t miny = clamp_tile((py - r) / (float)TILE_Y, a.gridy);
                const uint32_t maxx = clamp_tile((((px + r) + (float)TILE_X) - 1.0f) / (float)TILE_X, a.gridx);
                const uint32_t maxy = clamp_tile((((py + r) + (float)TILE_Y) - 1.0f) / (float)TILE_Y, a.gridy);
                const uint32_t ntiles = (maxx - minx) * (maxy - miny);
                if (ntiles != 0) {
                    V3 rgb = rgb_pre;
                    if (!a.colors_precomp) {
                        const V3 cam = v3(a.campos[3 * vw], a.campos[3 * vw + 1], a.campos[3 * vw + 2]);
                        rgb = sh_to_rgb_t<DEG>(p_orig, cam, shv, &cmask);
                    }
                    radius_out = (int)my_radius;
                    tiles = ntiles;
                    key = __float_as_uint(p_view.z);
                    rect = make_uint2(minx | (miny << 16), maxx | (maxy << 16));
                    // the pairs this Gaussian EMITS: the reference's rectangle, clipped to where alpha can reach 1/255
                    // (tile_cull.hpp); `tiles`, the reference's count, still goes into tiles_touched and num_rendered
                    etiles = a.reference_lists ? ntiles
                                               : clip_rect_to_footprint(px, py, cov_x, cov_y, cov_z, det, conic_x, conic_y, conic_z,
#ifdef GSR_NO_SPANS   // (A/B builds: bounding-box clipping only)
                                                                        opacity, r, rect, a.gridx, a.gridy);
#else
                                                                        opacity, r, rect, a.gridx, a.gridy, &spans, &spans_valid);
#endif
                    s.q0 = make_float4(px, py, conic_x, conic_y);
                    s.q1 = make_float4(conic_z, opacity, rgb.x, rgb.y);
                    s.q2 = make_float4(rgb.z, p_view.z, 0.f, 0.f);
                }
            }
        }

        if (valid) {
            a.radii[(size_t)vw * a.P + idx] = radius_out;
            at_view(a.tiles_touched, a.g_stride, vw)[idx] = tiles;
            at_view(a.dkey, a.g_stride, vw)[idx] = key;
        }
        {
            // smallest / largest key of a Gaussian that emits pairs, per wave (the depth sort's histogram kernel reduces the
            // workgroups' records to the key bits the frame's sort has to look at, sort.hip)
            uint32_t kmin = (valid && key != CULLED_KEY) ? key : 0xFFFFFFFFu, kmax = (valid && key != CULLED_KEY) ? key : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t x = __shfl_xor(kmin, d, 64), y = __shfl_xor(kmax, d, 64);
                kmin = x < kmin ? x : kmin;
                kmax = y > kmax ? y : kmax;
            }
            if ((threadIdx.x & 63) == 0) { s_mm[vw - v_first][threadIdx.x >> 6][0] = kmin; s_mm[vw - v_first][threadIdx.x >> 6][1] = kmax; }
        }
        // q3: what the pair emission needs per Gaussian (tile rectangle, tile count), so that it gathers ONE line per
        // Gaussian; the whole 64-B line is written here
        // what the emission needs (binning.hip emit_info): (first tile, last tile + 1, pairs emitted, pairs of the reference's
        // rectangle), or with row spans (first tile, spans of rows 0-3, spans of rows 4-7, reference pairs | SPANS_FLAG)
        const float4 q3 = spans_valid ? make_float4(__uint_as_float(rect.x), __uint_as_float(spans.x), __uint_as_float(spans.y),
                                                    __uint_as_float(tiles | SPANS_FLAG))
                                      : make_float4(__uint_as_float(rect.x), __uint_as_float(rect.y), __uint_as_float(etiles),
                                                    __uint_as_float(tiles));
        const int wave_first = idx_raw - (int)(threadIdx.x & 63);
        const bool full_wave = wave_first + 64 <= a.P;   // wave-uniform
        if (full_wave) {
            // The wave's 64 lines are 4 KB in a row.  A lane storing its own line issues four stores whose lanes are 64 B apart
            // (64 requests of 16 B each); through the wave's LDS slice the same bytes leave as four stores of consecutive
            // 16-B chunks (16 requests of 64 B each).  DS operations of a wave execute in order: no barrier.
            float4* mine = xpose[threadIdx.x >> 6] + 4 * (threadIdx.x & 63);
            mine[0] = s.q0; mine[1] = s.q1; mine[2] = s.q2; mine[3] = q3;
            __builtin_amdgcn_wave_barrier();
            const float4* rd = xpose[threadIdx.x >> 6] + (threadIdx.x & 63);
            float4* out = reinterpret_cast<float4*>(at_view(a.splat, a.g_stride, vw) + wave_first) + (threadIdx.x & 63);
            const float4 t0 = rd[0], t1 = rd[64], t2 = rd[128], t3 = rd[192];
            out[0] = t0; out[64] = t1; out[128] = t2; out[192] = t3;
            __builtin_amdgcn_wave_barrier();
        } else if (valid) {
            Splat* sp = at_view(a.splat, a.g_stride, vw) + idx;
            sp->q0 = s.q0;
            sp->q1 = s.q1;
            sp->q2 = s.q2;
            sp->q3 = q3;
        }
        if (a.need_backward && valid) {
            at_view(a.clamped, a.g_stride, vw)[idx] = (uint8_t)cmask;
            // the render backward accumulates into this Gaussian's 64-B record: cleared here, alongside the Splat line
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (full_wave) {
                // the wave's 64 records are 4 KB in a row: consecutive lanes on consecutive 16-B chunks (16 requests of 64 B
                // per store instruction instead of 64 requests of 16 B)
                float4* recw = reinterpret_cast<float4*>(at_view(a.grad_rec, a.gr_stride, vw) + (size_t)wave_first * GRAD_REC_WORDS) + (threadIdx.x & 63);
                recw[0] = z; recw[64] = z; recw[128] = z; recw[192] = z;
            } else {
                float4* rec = reinterpret_cast<float4*>(at_view(a.grad_rec, a.gr_stride, vw) + (size_t)idx * GRAD_REC_WORDS);
                rec[0] = z; rec[1] = z; rec[2] = z; rec[3] = z;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < v_last - v_first) {
        const uint32_t* m = &s_mm[threadIdx.x][0][0];
        const uint32_t lo01 = m[0] < m[2] ? m[0] : m[2], lo23 = m[4] < m[6] ? m[4] : m[6];
        const uint32_t hi01 = m[1] > m[3] ? m[1] : m[3], hi23 = m[5] > m[7] ? m[5] : m[7];
        uint32_t* mm = at_view(a.pre_minmax, a.g_stride, (uint32_t)(v_first + (int)threadIdx.x));
        mm[blockIdx.x] = lo01 < lo23 ? lo01 : lo23;
        mm[a.n_pre + blockIdx.x] = hi01 > hi23 ? hi01 : hi23;
    }
}

int launch_preprocess(const Launch& L, const gsr_params& p, const Batch& B, int* radii)
{
    const GeomView& g = B.g;
    PreArgs a;
    a.P = p.P; a.D = p.D; a.M = p.M; a.W = p.W; a.H = p.H;
    a.tanfovx = p.tanfovx; a.tanfovy = p.tanfovy;
    a.focal_y = p.H / (2.0f * p.tanfovy);  // reference CR/rasterizer_impl.cu:222-223
    a.focal_x = p.W / (2.0f * p.tanfovx);
    a.scale_modifier = p.scale_modifier;
    a.prefiltered = p.prefiltered;
    a.need_backward = p.need_backward;
    a.reference_lists = p.reference_lists;
    a.gridx = (uint32_t)((p.W + TILE_X - 1) / TILE_X);
    a.gridy = (uint32_t)((p.H + TILE_Y - 1) / TILE_Y);
    a.means3D = p.means3D; a.shs = p.shs; a.colors_precomp = p.colors_precomp; a.opacities = p.opacities;
    a.scales = p.scales; a.rotations = p.rotations; a.cov3D_precomp = p.cov3D_precomp;
    a.view = p.viewmatrix; a.proj = p.projmatrix; a.campos = p.campos;
    a.splat = g.splat; a.tiles_touched = g.tiles_touched; a.clamped = g.clamped;
    a.pre_minmax = g.pre_minmax; a.n_pre = (int)div_up(p.P, PRE_THREADS);
    a.dkey = g.dkey[0]; a.radii = radii; a.counters = g.counters; a.grad_rec = B.grad_rec; a.gr_stride = B.gr_stride;
    a.g_zero = g.zero_begin; a.g_zero_bytes = g.zero_bytes; a.g_stride = B.g_stride;
    a.iv_zero = B.iv.zero_begin; a.iv_zero_bytes = B.iv.zero_bytes; a.iv_stride = B.iv_stride;
    const int blocks = (p.P + 255) / 256;
    // all views of a Gaussian in one thread when the cloud alone fills the chip (256 CUs x 8 workgroups), otherwise the views
    // are spread over grid.y so that small clouds keep their parallelism
    a.V = B.V;
    a.vpt = 1;
    while (a.vpt < B.V && (int64_t)blocks * div_up(B.V, a.vpt * 2) >= 2048) a.vpt *= 2;
    if (a.vpt > B.V) a.vpt = B.V;
    const dim3 grid(blocks, (unsigned)div_up(B.V, a.vpt));
    switch (p.shs ? p.D : 0) {
    case 0: hipLaunchKernelGGL(k_preprocess<0>, grid, dim3(256), 0, L.stream, a); break;
    case 1: hipLaunchKernelGGL(k_preprocess<1>, grid, dim3(256), 0, L.stream, a); break;
    case 2: hipLaunchKernelGGL(k_preprocess<2>, grid, dim3(256), 0, L.stream, a); break;
    default: hipLaunchKernelGGL(k_preprocess<3>, grid, dim3(256), 0, L.stream, a); break;
    }
    return check_launch(L, "preprocess");
}

// ---- colour-only update of the Splat records (gsr_forward_recolor) -----------------------------------------
// The reference's caller renders four passes per view that differ only in the per-Gaussian colour (world xyz, SH
// colour, ones = hit map, normals; /root/reference/simple_raw_render.py:410-524), each through the full pipeline.
// Geometry, lists and ranges are identical across the passes, so a pass can reuse them: this kernel rewrites the three
// colour floats of every visible Gaussian's record (precomputed colours verbatim, or the SH colour exactly as
// k_preprocess computes it) and the caller re-launches the compositing kernel only.
__global__ __launch_bounds__(256) void k_recolor(int P, int D, int M, const float* __restrict__ means3D,
                                                 const float* __restrict__ shs, const float* __restrict__ colors_precomp,
                                                 const float* __restrict__ campos, const uint32_t* __restrict__ tiles_touched,
                                                 Splat* __restrict__ splat, uint8_t* __restrict__ clamped, size_t g_stride,
                                                 size_t colors_view_stride)
{
    const uint32_t vw = blockIdx.y;
    if (colors_precomp) colors_precomp += colors_view_stride * vw;
    tiles_touched = at_view(tiles_touched, g_stride, vw);
    splat = at_view(splat, g_stride, vw);
    if (campos) campos += 3 * vw;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P || tiles_touched[idx] == 0) return;
    V3 rgb;
    uint32_t cmask = 0;
    if (colors_precomp) {
        rgb = v3(colors_precomp[3 * idx], colors_precomp[3 * idx + 1], colors_precomp[3 * idx + 2]);
    } else {
        const V3 pos = v3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        rgb = sh_to_rgb(D, pos, v3(campos[0], campos[1], campos[2]), shs + (size_t)idx * M * 3, &cmask);
    }
    float* rec = reinterpret_cast<float*>(splat + idx);
    rec[6] = rgb.x;  // q1.z
    rec[7] = rgb.y;  // q1.w
    rec[8] = rgb.z;  // q2.x
    if (clamped) at_view(clamped, g_stride, vw)[idx] = (uint8_t)cmask;   // need_backward: the SH backward reads the new mask
}

int launch_recolor(const Launch& L, const gsr_params& p, const Batch& B, size_t colors_view_stride)
{
    hipLaunchKernelGGL(k_recolor, dim3((p.P + 255) / 256, B.V), dim3(256), 0, L.stream, p.P, p.D, p.M, p.means3D, p.shs,
                       p.colors_precomp, p.campos, B.g.tiles_touched, B.g.splat, p.need_backward ? B.g.clamped : nullptr, B.g_stride,
                       colors_view_stride);
    return check_launch(L, "recolor");
}

// reference CR/rasterizer_impl.cu:54-66 (checkFrustum): only the near-plane test is live
__global__ __launch_bounds__(256) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                      const float* __restrict__ view, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const V3 p = v3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    const float z = ((view[2] * p.x + view[6] * p.y) + view[10] * p.z) + view[14];
    present[idx] = !(z <= 0.2f);
}

int launch_mark_visible(const Launch& L, int P, const float* means3D, const float* view, uint8_t* present)
{
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, L.stream, P, means3D, view, present);
    return check_launch(L, "mark_visible");
}

}  // namespace gsr
