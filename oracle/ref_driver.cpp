// ref_driver.cpp -- thin host driver around the REFERENCE rasterizer core
// (CudaRasterizer::Rasterizer, CR/rasterizer.h:24-80) built for gfx950 by
// oracle/build_ref.sh.  TEST INFRASTRUCTURE ONLY.
//
// This file is ours; it contains no reference code.  It links against the
// reference's own translation units (hipify-perl output of
// CR/{rasterizer_impl,forward,backward}.cu placed under oracle/_ref/src/, never
// committed) and exposes host-pointer entry points with the same struct layout as
// the CPU oracle (gsr_oracle.h) so tests can diff reference / oracle / product.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <vector>

#include "gsr_oracle.h"        // orc_inputs / orc_state layouts only
#include "rasterizer.h"        // reference API (from oracle/_ref/src)
#include "rasterizer_impl.h"   // reference arena layout: GeometryState/BinningState/ImageState::fromChunk

#define HCHK(x)                                                                              \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "[ref_driver] %s failed: %s\n", #x, hipGetErrorString(e_));      \
            abort();                                                                         \
        }                                                                                    \
    } while (0)

namespace {

struct Arena {
    char* ptr = nullptr;
    size_t cap = 0;
    char* get(size_t n)
    {
        if (n > cap) {
            if (ptr) HCHK(hipFree(ptr));
            HCHK(hipMalloc(&ptr, n ? n : 1));
            cap = n;
        }
        return ptr;
    }
    void release()
    {
        if (ptr) HCHK(hipFree(ptr));
        ptr = nullptr;
        cap = 0;
    }
};

template <typename T>
T* to_dev(const T* h, size_t n)
{
    if (!h || n == 0) return nullptr;
    T* d = nullptr;
    HCHK(hipMalloc(&d, n * sizeof(T)));
    HCHK(hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
template <typename T>
T* to_host(const T* d, size_t n)
{
    T* h = (T*)calloc(n + 1, sizeof(T));
    if (d && n) HCHK(hipMemcpy(h, d, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
}

struct DevInputs {
    float *bg, *means3D, *shs, *colors, *opac, *scales, *rots, *cov3D, *view, *proj, *campos;
    void upload(const orc_inputs* in)
    {
        const size_t P = in->P;
        bg = to_dev(in->bg, 3);
        means3D = to_dev(in->means3D, 3 * P);
        shs = to_dev(in->shs, (size_t)3 * in->M * P);
        colors = to_dev(in->colors_precomp, 3 * P);
        opac = to_dev(in->opacities, P);
        scales = to_dev(in->scales, 3 * P);
        rots = to_dev(in->rotations, 4 * P);
        cov3D = to_dev(in->cov3D_precomp, 6 * P);
        view = to_dev(in->viewmatrix, 16);
        proj = to_dev(in->projmatrix, 16);
        campos = to_dev(in->campos, 3);
    }
    void release()
    {
        float* all[] = {bg, means3D, shs, colors, opac, scales, rots, cov3D, view, proj, campos};
        for (float* p : all)
            if (p) HCHK(hipFree(p));
    }
};

struct Session {
    Arena geom, binning, image;
    DevInputs din;
    float* d_out = nullptr;
    int* d_radii = nullptr;
    int R = 0;
};

std::map<const orc_state*, Session*> g_sessions;

int run_forward(const orc_inputs* in, Session* s)
{
    using namespace CudaRasterizer;
    std::function<char*(size_t)> g = [s](size_t n) { return s->geom.get(n); };
    std::function<char*(size_t)> b = [s](size_t n) { return s->binning.get(n); };
    std::function<char*(size_t)> i = [s](size_t n) { return s->image.get(n); };
    return Rasterizer::forward(g, b, i, in->P, in->D, in->M, s->din.bg, in->W, in->H, s->din.means3D, s->din.shs,
                               s->din.colors, s->din.opac, s->din.scales, in->scale_modifier, s->din.rots,
                               s->din.cov3D, s->din.view, s->din.proj, s->din.campos, in->tanfovx, in->tanfovy,
                               in->prefiltered != 0, s->d_out, s->d_radii, false);
}

struct DevGrads {
    float *mean2D, *conic, *opacity, *color, *mean3D, *cov3D, *sh, *scale, *rot;
    size_t P, M;
    void alloc(size_t P_, size_t M_)
    {
        P = P_; M = M_;
        auto z = [](size_t n) { float* d = nullptr; HCHK(hipMalloc(&d, (n ? n : 1) * 4)); HCHK(hipMemset(d, 0, (n ? n : 1) * 4)); return d; };
        mean2D = z(3 * P); conic = z(4 * P); opacity = z(P); color = z(3 * P); mean3D = z(3 * P);
        cov3D = z(6 * P); sh = z(3 * M * P); scale = z(3 * P); rot = z(4 * P);
    }
    void zero()
    {
        HCHK(hipMemsetAsync(mean2D, 0, 3 * P * 4)); HCHK(hipMemsetAsync(conic, 0, 4 * P * 4));
        HCHK(hipMemsetAsync(opacity, 0, P * 4)); HCHK(hipMemsetAsync(color, 0, 3 * P * 4));
        HCHK(hipMemsetAsync(mean3D, 0, 3 * P * 4)); HCHK(hipMemsetAsync(cov3D, 0, 6 * P * 4));
        if (M) HCHK(hipMemsetAsync(sh, 0, 3 * M * P * 4));
        HCHK(hipMemsetAsync(scale, 0, 3 * P * 4)); HCHK(hipMemsetAsync(rot, 0, 4 * P * 4));
    }
    void release()
    {
        float* all[] = {mean2D, conic, opacity, color, mean3D, cov3D, sh, scale, rot};
        for (float* p : all) HCHK(hipFree(p));
    }
};

void run_backward(const orc_inputs* in, Session* s, const float* d_dL_dpix, DevGrads& g)
{
    using namespace CudaRasterizer;
    Rasterizer::backward(in->P, in->D, in->M, s->R, s->din.bg, in->W, in->H, s->din.means3D, s->din.shs,
                         s->din.colors, s->din.scales, in->scale_modifier, s->din.rots, s->din.cov3D, s->din.view,
                         s->din.proj, s->din.campos, in->tanfovx, in->tanfovy, s->d_radii, s->geom.ptr,
                         s->binning.ptr, s->image.ptr, d_dL_dpix, g.mean2D, g.conic, g.opacity, g.color, g.mean3D,
                         g.cov3D, g.sh, g.scale, g.rot, false);
}

}  // namespace

extern "C" {

// Reference forward on host inputs; returns host copies of the output and of every arena array.
orc_state* ref_forward(const orc_inputs* in)
{
    using namespace CudaRasterizer;
    orc_state* st = (orc_state*)calloc(1, sizeof(orc_state));
    Session* s = new Session();
    const size_t P = in->P, N = (size_t)in->W * in->H;
    st->P = in->P; st->W = in->W; st->H = in->H;
    st->gridx = (in->W + 15) / 16;
    st->gridy = (in->H + 15) / 16;
    const size_t T = (size_t)st->gridx * st->gridy;
    st->consumed_fwd = st->consumed_bwd = -1;

    s->din.upload(in);
    HCHK(hipMalloc(&s->d_out, (3 * N ? 3 * N : 1) * 4));
    HCHK(hipMemset(s->d_out, 0, (3 * N ? 3 * N : 1) * 4));   // torch::full(0.0), rasterize_points.cu:68
    HCHK(hipMalloc(&s->d_radii, (P ? P : 1) * 4));
    HCHK(hipMemset(s->d_radii, 0, (P ? P : 1) * 4));          // torch::full(0), rasterize_points.cu:69
    if (P != 0) s->R = run_forward(in, s);                     // rasterize_points.cu:81
    HCHK(hipDeviceSynchronize());
    st->R = s->R;

    st->out_color = to_host(s->d_out, 3 * N);
    st->radii = (int32_t*)to_host(s->d_radii, P);
    if (P != 0) {
        char* c = s->geom.ptr;
        GeometryState gs = GeometryState::fromChunk(c, P);
        st->depths = to_host(gs.depths, P);
        st->clamped = (uint8_t*)to_host((uint8_t*)gs.clamped, 3 * P);
        st->means2D = to_host((float*)gs.means2D, 2 * P);
        st->cov3D = to_host(gs.cov3D, 6 * P);
        st->conic_opacity = to_host((float*)gs.conic_opacity, 4 * P);
        st->rgb = to_host(gs.rgb, 3 * P);
        st->tiles_touched = to_host(gs.tiles_touched, P);
        st->point_offsets = to_host(gs.point_offsets, P);
        c = s->binning.ptr;
        BinningState bs = BinningState::fromChunk(c, s->R);
        st->keys_unsorted = (uint64_t*)to_host(bs.point_list_keys_unsorted, (size_t)s->R);
        st->vals_unsorted = to_host(bs.point_list_unsorted, (size_t)s->R);
        st->keys = (uint64_t*)to_host(bs.point_list_keys, (size_t)s->R);
        st->vals = to_host(bs.point_list, (size_t)s->R);
        c = s->image.ptr;
        ImageState is = ImageState::fromChunk(c, N);
        st->ranges = (uint32_t*)to_host((uint32_t*)is.ranges, 2 * T);
        st->final_T = to_host(is.accum_alpha, N);
        st->n_contrib = to_host(is.n_contrib, N);
        int64_t vis = 0;
        for (size_t i = 0; i < P; i++) vis += st->radii[i] > 0;
        st->visible = vis;
    }
    g_sessions[st] = s;
    return st;
}

// Reference backward for a state returned by ref_forward.  Host gradient buffers are overwritten.
void ref_backward(const orc_inputs* in, const orc_state* st, const float* dL_dpix, float* dL_dmean2D,
                  float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                  float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    Session* s = g_sessions.at(st);
    const size_t P = in->P, M = in->M, N = (size_t)in->W * in->H;
    if (P == 0) return;
    float* d_pix = to_dev(dL_dpix, 3 * N);
    DevGrads g;
    g.alloc(P, M);
    run_backward(in, s, d_pix, g);
    HCHK(hipDeviceSynchronize());
    HCHK(hipMemcpy(dL_dmean2D, g.mean2D, 3 * P * 4, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(dL_dconic, g.conic, 4 * P * 4, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(dL_dopacity, g.opacity, P * 4, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(dL_dcolor, g.color, 3 * P * 4, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(dL_dmean3D, g.mean3D, 3 * P * 4, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(dL_dcov3D, g.cov3D, 6 * P * 4, hipMemcpyDeviceToHost));
    if (M) HCHK(hipMemcpy(dL_dsh, g.sh, 3 * M * P * 4, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(dL_dscale, g.scale, 3 * P * 4, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(dL_drot, g.rot, 4 * P * 4, hipMemcpyDeviceToHost));
    g.release();
    HCHK(hipFree(d_pix));
}

void ref_free(orc_state* st)
{
    if (!st) return;
    auto it = g_sessions.find(st);
    if (it != g_sessions.end()) {
        Session* s = it->second;
        s->geom.release(); s->binning.release(); s->image.release();
        s->din.release();
        if (s->d_out) HCHK(hipFree(s->d_out));
        if (s->d_radii) HCHK(hipFree(s->d_radii));
        delete s;
        g_sessions.erase(it);
    }
    free(st->depths); free(st->clamped); free(st->radii); free(st->means2D); free(st->cov3D);
    free(st->conic_opacity); free(st->rgb); free(st->tiles_touched); free(st->point_offsets);
    free(st->keys_unsorted); free(st->vals_unsorted); free(st->keys); free(st->vals);
    free(st->ranges); free(st->final_T); free(st->n_contrib); free(st->out_color);
    free(st);
}

void ref_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present)
{
    if (P == 0) return;
    float* dm = to_dev(means3D, (size_t)3 * P);
    float* dv = to_dev(view, 16);
    float* dp = to_dev(proj, 16);
    bool* dpres = nullptr;
    HCHK(hipMalloc(&dpres, P));
    HCHK(hipMemset(dpres, 0, P));
    CudaRasterizer::Rasterizer::markVisible(P, dm, dv, dp, dpres);
    HCHK(hipDeviceSynchronize());
    HCHK(hipMemcpy(present, dpres, P, hipMemcpyDeviceToHost));
    HCHK(hipFree(dm)); HCHK(hipFree(dv)); HCHK(hipFree(dp)); HCHK(hipFree(dpres));
}

// Wall time (hipEvents, default stream) of the reference's own kernels on this GPU, inputs resident in HBM:
// `iters` forwards (and backwards when dL_dpix != NULL) after `warmup` untimed ones.  Gradient buffers are
// re-zeroed inside the timed region like rasterize_points.cu:151-159 does each call.
void ref_bench(const orc_inputs* in, const float* dL_dpix, int warmup, int iters, float* ms_fwd, float* ms_bwd)
{
    Session s;
    const size_t P = in->P, N = (size_t)in->W * in->H;
    s.din.upload(in);
    HCHK(hipMalloc(&s.d_out, 3 * N * 4));
    HCHK(hipMalloc(&s.d_radii, P * 4));
    float* d_pix = dL_dpix ? to_dev(dL_dpix, 3 * N) : nullptr;
    DevGrads g;
    if (d_pix) g.alloc(P, in->M);
    hipEvent_t e0, e1, e2;
    HCHK(hipEventCreate(&e0)); HCHK(hipEventCreate(&e1)); HCHK(hipEventCreate(&e2));
    double tf = 0, tb = 0;
    for (int it = 0; it < warmup + iters; it++) {
        HCHK(hipEventRecord(e0, 0));
        HCHK(hipMemsetAsync(s.d_out, 0, 3 * N * 4));
        HCHK(hipMemsetAsync(s.d_radii, 0, P * 4));
        s.R = run_forward(in, &s);
        HCHK(hipEventRecord(e1, 0));
        if (d_pix) { g.zero(); run_backward(in, &s, d_pix, g); }
        HCHK(hipEventRecord(e2, 0));
        HCHK(hipEventSynchronize(e2));
        float a = 0, b = 0;
        HCHK(hipEventElapsedTime(&a, e0, e1));
        HCHK(hipEventElapsedTime(&b, e1, e2));
        if (it >= warmup) { tf += a; tb += b; }
    }
    *ms_fwd = (float)(tf / iters);
    *ms_bwd = (float)(tb / iters);
    if (d_pix) { g.release(); HCHK(hipFree(d_pix)); }
    s.geom.release(); s.binning.release(); s.image.release(); s.din.release();
    HCHK(hipFree(s.d_out)); HCHK(hipFree(s.d_radii));
    HCHK(hipEventDestroy(e0)); HCHK(hipEventDestroy(e1)); HCHK(hipEventDestroy(e2));
}

}  // extern "C"
