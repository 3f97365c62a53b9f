// render_cabi.cpp -- a host program WITHOUT Python or torch on top of the C ABI of include/gsr.h.
//
// What a C++ maintainer of the reference would write in place of rasterize_points.cu (which allocates torch tensors and
// calls CudaRasterizer::Rasterizer::forward / ::backward, /root/reference/diff-gaussian-rasterization/rasterize_points.cu:35-196):
// plain hipMalloc'd buffers, the synchronous two-stage forward (count, size the arena, bind), the backward.
//
//   render_cabi <scene.bin> <out.bin>
//
// scene.bin (little endian): int32 P, D, M, W, H, has_dL; float32 tanfovx, tanfovy, scale_modifier; then float32 arrays
//   bg[3] means3D[P*3] shs[P*M*3] opacities[P] scales[P*3] rotations[P*4] viewmatrix[16] projmatrix[16] campos[3]
//   and, if has_dL, dL_dpix[3*H*W].
// out.bin: int64 num_rendered; float32 out_color[3*H*W]; int32 radii[P]; if has_dL float32 dL_dmean3D[P*3] dL_dopacity[P]
//   dL_dsh[P*M*3] dL_dscale[P*3] dL_drot[P*4] dL_dmean2D[P*3].
// tests/test_gpu_cabi_example.py builds scenes, runs this program and checks the output against the CPU oracle.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/gsr.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define GSR_OK_(x) do { int rc_ = (x); if (rc_ != GSR_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, gsr_last_error()); return 3; } } while (0)

template <typename T>
static bool read_vec(FILE* f, std::vector<T>& v, size_t n)
{
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

template <typename T>
static T* to_device(const std::vector<T>& v)
{
    T* d = nullptr;
    if (v.empty()) return nullptr;
    if (hipMalloc((void**)&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[6];
    float fl[3];
    if (fread(hdr, 4, 6, f) != 6 || fread(fl, 4, 3, f) != 3) { fprintf(stderr, "short header\n"); return 1; }
    const int P = hdr[0], D = hdr[1], M = hdr[2], W = hdr[3], H = hdr[4], has_dL = hdr[5];
    std::vector<float> bg, means, shs, opac, scales, rots, view, proj, campos, dL;
    if (!read_vec(f, bg, 3) || !read_vec(f, means, (size_t)P * 3) || !read_vec(f, shs, (size_t)P * M * 3) || !read_vec(f, opac, (size_t)P) ||
        !read_vec(f, scales, (size_t)P * 3) || !read_vec(f, rots, (size_t)P * 4) || !read_vec(f, view, 16) || !read_vec(f, proj, 16) ||
        !read_vec(f, campos, 3) || !read_vec(f, dL, has_dL ? (size_t)3 * W * H : 0)) {
        fprintf(stderr, "short scene file\n");
        return 1;
    }
    fclose(f);

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    gsr_params p{};
    p.P = P; p.D = D; p.M = M; p.W = W; p.H = H;
    p.tanfovx = fl[0]; p.tanfovy = fl[1]; p.scale_modifier = fl[2];
    p.prefiltered = 0; p.debug = 0; p.need_backward = has_dL;
    p.bg = to_device(bg); p.means3D = to_device(means); p.shs = to_device(shs); p.colors_precomp = nullptr;
    p.opacities = to_device(opac); p.scales = to_device(scales); p.rotations = to_device(rots); p.cov3D_precomp = nullptr;
    p.viewmatrix = to_device(view); p.projmatrix = to_device(proj); p.campos = to_device(campos);

    // the three scratch arenas are the caller's, sized by the query functions (geomBuffer / imgBuffer / binningBuffer of the
    // reference's binding); the binning arena after stage 1 has counted the (tile, Gaussian) pairs
    const size_t gb = gsr_geom_bytes(P), ib = gsr_image_bytes(W, H);
    void *geom = nullptr, *image = nullptr, *binning = nullptr;
    float* out_color = nullptr;
    int* radii = nullptr;
    HIP_OK(hipMalloc(&geom, gb));
    HIP_OK(hipMalloc(&image, ib));
    HIP_OK(hipMalloc((void**)&out_color, (size_t)3 * W * H * sizeof(float)));
    HIP_OK(hipMalloc((void**)&radii, (size_t)(P > 0 ? P : 1) * sizeof(int)));
    HIP_OK(hipMemsetAsync(out_color, 0, (size_t)3 * W * H * sizeof(float), stream));   // P == 0: the zero image, like the reference
    int64_t R = 0;
    GSR_OK_(gsr_forward_stage1(&p, geom, gb, image, ib, radii, &R, stream));
    const size_t bb = gsr_binning_bytes(R);
    HIP_OK(hipMalloc(&binning, bb));
    GSR_OK_(gsr_forward_stage2(&p, geom, gb, binning, bb, image, ib, R, out_color, stream));

    std::vector<float> g_mean3D, g_opac, g_sh, g_scale, g_rot, g_mean2D;
    if (has_dL) {
        float *d_dL = to_device(dL), *d_m2, *d_op, *d_col, *d_m3, *d_cov, *d_sh, *d_sc, *d_rot;
        const size_t n = (size_t)(P > 0 ? P : 1);
        HIP_OK(hipMalloc((void**)&d_m2, n * 3 * 4)); HIP_OK(hipMalloc((void**)&d_op, n * 4)); HIP_OK(hipMalloc((void**)&d_col, n * 3 * 4));
        HIP_OK(hipMalloc((void**)&d_m3, n * 3 * 4)); HIP_OK(hipMalloc((void**)&d_cov, n * 6 * 4)); HIP_OK(hipMalloc((void**)&d_sh, n * (size_t)(M > 0 ? M : 1) * 3 * 4));
        HIP_OK(hipMalloc((void**)&d_sc, n * 3 * 4)); HIP_OK(hipMalloc((void**)&d_rot, n * 4 * 4));
        // nothing to clear: every gradient is written for every Gaussian
        GSR_OK_(gsr_backward(&p, radii, R, geom, gb, binning, bb, image, ib, d_dL, d_m2, d_op, d_col, d_m3, d_cov, d_sh, d_sc, d_rot, stream));
        g_mean3D.resize((size_t)P * 3); g_opac.resize(P); g_sh.resize((size_t)P * M * 3); g_scale.resize((size_t)P * 3);
        g_rot.resize((size_t)P * 4); g_mean2D.resize((size_t)P * 3);
        HIP_OK(hipStreamSynchronize(stream));
        if (P > 0) {
            HIP_OK(hipMemcpy(g_mean3D.data(), d_m3, g_mean3D.size() * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(g_opac.data(), d_op, g_opac.size() * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(g_sh.data(), d_sh, g_sh.size() * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(g_scale.data(), d_sc, g_scale.size() * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(g_rot.data(), d_rot, g_rot.size() * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(g_mean2D.data(), d_m2, g_mean2D.size() * 4, hipMemcpyDeviceToHost));
        }
    }
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<float> h_color((size_t)3 * W * H);
    std::vector<int32_t> h_radii(P);
    HIP_OK(hipMemcpy(h_color.data(), out_color, h_color.size() * 4, hipMemcpyDeviceToHost));
    if (P > 0) HIP_OK(hipMemcpy(h_radii.data(), radii, h_radii.size() * 4, hipMemcpyDeviceToHost));

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    fwrite(&R, 8, 1, o);
    fwrite(h_color.data(), 4, h_color.size(), o);
    fwrite(h_radii.data(), 4, h_radii.size(), o);
    if (has_dL) {
        fwrite(g_mean3D.data(), 4, g_mean3D.size(), o); fwrite(g_opac.data(), 4, g_opac.size(), o); fwrite(g_sh.data(), 4, g_sh.size(), o);
        fwrite(g_scale.data(), 4, g_scale.size(), o); fwrite(g_rot.data(), 4, g_rot.size(), o); fwrite(g_mean2D.data(), 4, g_mean2D.size(), o);
    }
    fclose(o);
    printf("%s: P=%d %dx%d num_rendered=%lld\n", gsr_version(), P, W, H, (long long)R);
    return 0;
}
