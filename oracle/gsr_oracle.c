/*
 * gsr_oracle.c -- CPU restatement of the reference Gaussian tile rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY (see gsr_oracle.h).  Plain C11, IEEE fp32 in the
 * reference's source operation order, no FMA contraction
 * (-ffp-contract=off; x86-64 baseline has no FMA instructions either).
 *
 * CR/ = /root/reference/diff-gaussian-rasterization/cuda_rasterizer/
 * GLM = /root/reference/diff-gaussian-rasterization/third_party/glm/glm/
 *
 * GLM conventions restated here (they fix the rounding order):
 *   mat3 is column-major, m[c][r]; glm::mat3(a,b,c,d,e,f,g,h,i) has columns
 *   (a,b,c),(d,e,f),(g,h,i)                      (GLM/detail/type_mat3x3.inl)
 *   (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2],
 *   summed left to right                         (GLM/detail/type_mat3x3.inl:486-520)
 *   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z     (GLM/detail/func_geometric.inl:48-55)
 *   length(v) = sqrt(dot(v,v))                   (GLM/detail/func_geometric.inl:8-14)
 */
#include "gsr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* CR/config.h:16 */
#define BLOCK_Y 16 /* CR/config.h:17 */

/* CR/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,  -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

typedef struct { float x, y, z; } v3;
typedef struct { float m[3][3]; } m3; /* m[c][r], column-major like glm::mat3 */

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); } /* scalar * vec */
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

static inline m3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    m3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}
static inline m3 m3_mul(m3 A, m3 B)
{
    m3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = (A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1]) + A.m[2][r] * B.m[c][2];
    return R;
}
static inline m3 m3_transpose(m3 A)
{
    m3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[r][c];
    return R;
}

static inline float fminf_cuda(float a, float b) { return a < b ? a : b; } /* CUDA min(float,float) on non-NaN */
static inline float fmaxf_cuda(float a, float b) { return a > b ? a : b; }

/* CR/auxiliary.h:41-44.  v + 1.0 promotes to double; result rounded to float on return. */
static inline float ndc2Pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

/* CR/auxiliary.h:46-56.  max_radius is an int parameter (the float radius is truncated at the call). */
static inline void getRect(float px, float py, int max_radius, uint32_t *rmin, uint32_t *rmax, int gridx, int gridy)
{
    int v;
    v = (int)((px - (float)max_radius) / (float)BLOCK_X);
    v = v > 0 ? v : 0; rmin[0] = (uint32_t)gridx < (uint32_t)v ? (uint32_t)gridx : (uint32_t)v;
    v = (int)((py - (float)max_radius) / (float)BLOCK_Y);
    v = v > 0 ? v : 0; rmin[1] = (uint32_t)gridy < (uint32_t)v ? (uint32_t)gridy : (uint32_t)v;
    v = (int)((((px + (float)max_radius) + (float)BLOCK_X) - 1.0f) / (float)BLOCK_X); /* p.x + r + BLOCK_X - 1, left to right */
    v = v > 0 ? v : 0; rmax[0] = (uint32_t)gridx < (uint32_t)v ? (uint32_t)gridx : (uint32_t)v;
    v = (int)((((py + (float)max_radius) + (float)BLOCK_Y) - 1.0f) / (float)BLOCK_Y);
    v = v > 0 ? v : 0; rmax[1] = (uint32_t)gridy < (uint32_t)v ? (uint32_t)gridy : (uint32_t)v;
}

/* CR/auxiliary.h:58-76 */
static inline v3 transformPoint4x3(v3 p, const float *m)
{
    return v3_make(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12],
                   ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
                   ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}
static inline float transformPoint4x4_w(v3 p, const float *m)
{
    return ((m[3] * p.x + m[7] * p.y) + m[11] * p.z) + m[15];
}
/* CR/auxiliary.h:89-97 */
static inline v3 transformVec4x3Transpose(v3 p, const float *m)
{
    return v3_make((m[0] * p.x + m[1] * p.y) + m[2] * p.z,
                   (m[4] * p.x + m[5] * p.y) + m[6] * p.z,
                   (m[8] * p.x + m[9] * p.y) + m[10] * p.z);
}
/* CR/auxiliary.h:107-117 */
static inline v3 dnormvdv(v3 v, v3 dv)
{
    float sum2 = (v.x * v.x + v.y * v.y) + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    v3 r;
    r.x = ((((+sum2 - v.x * v.x) * dv.x) - v.y * v.x * dv.y) - v.z * v.x * dv.z) * invsum32;
    r.y = ((-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y) - v.z * v.y * dv.z) * invsum32;
    r.z = ((-v.x * v.z * dv.x - v.y * v.z * dv.y) + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* quaternion -> R exactly as written at CR/forward.cu:136-140 / CR/backward.cu:286-290 */
static inline m3 quat_to_R(const float *rot)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    return m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

/* CR/forward.cu:121-155 */
static void computeCov3D(const float *scale, float mod, const float *rot, float *cov3D)
{
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    m3 R = quat_to_R(rot);
    m3 M = m3_mul(S, R);
    m3 Sigma = m3_mul(m3_transpose(M), M);
    cov3D[0] = Sigma.m[0][0];
    cov3D[1] = Sigma.m[0][1];
    cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1];
    cov3D[4] = Sigma.m[1][2];
    cov3D[5] = Sigma.m[2][2];
}

/* Shared by CR/forward.cu:74-116 and CR/backward.cu:160-198: t (clamped), J, W, T, cov2D before dilation. */
typedef struct { v3 t; float txtz, tytz, limx, limy; m3 J, W, T, Vrk, cov; } cov2d_ctx;
static cov2d_ctx cov2d_common(v3 mean, float fx, float fy, float tan_fovx, float tan_fovy, const float *cov3D,
                              const float *view)
{
    cov2d_ctx c;
    v3 t = transformPoint4x3(mean, view);
    c.limx = 1.3f * tan_fovx;
    c.limy = 1.3f * tan_fovy;
    c.txtz = t.x / t.z;
    c.tytz = t.y / t.z;
    t.x = fminf_cuda(c.limx, fmaxf_cuda(-c.limx, c.txtz)) * t.z;
    t.y = fminf_cuda(c.limy, fmaxf_cuda(-c.limy, c.tytz)) * t.z;
    c.t = t;
    c.J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z),
                  0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z),
                  0, 0, 0);
    c.W = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    c.T = m3_mul(c.W, c.J);
    c.Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    c.cov = m3_mul(m3_mul(m3_transpose(c.T), m3_transpose(c.Vrk)), c.T);
    return c;
}

/* CR/forward.cu:20-71 */
void orc_sh_to_rgb(int deg, int max_coeffs, const float *mean, const float *campos, const float *shp, float *rgb_out,
                   uint8_t *clamped_out)
{
    (void)max_coeffs;
    v3 pos = v3_make(mean[0], mean[1], mean[2]);
    v3 dir = v3_sub(pos, v3_make(campos[0], campos[1], campos[2]));
    float len = sqrtf(v3_dot(dir, dir));
    dir = v3_make(dir.x / len, dir.y / len, dir.z / len);
#define SH(k) v3_make(shp[3 * (k)], shp[3 * (k) + 1], shp[3 * (k) + 2])
    v3 result = v3_scale(SH_C0, SH(0));
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        result = v3_sub(v3_add(v3_sub(result, v3_scale(SH_C1 * y, SH(1))), v3_scale(SH_C1 * z, SH(2))),
                        v3_scale(SH_C1 * x, SH(3)));
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            result = v3_add(result, v3_scale(SH_C2[0] * xy, SH(4)));
            result = v3_add(result, v3_scale(SH_C2[1] * yz, SH(5)));
            result = v3_add(result, v3_scale(SH_C2[2] * (2.0f * zz - xx - yy), SH(6)));
            result = v3_add(result, v3_scale(SH_C2[3] * xz, SH(7)));
            result = v3_add(result, v3_scale(SH_C2[4] * (xx - yy), SH(8)));
            if (deg > 2) {
                result = v3_add(result, v3_scale(SH_C3[0] * y * (3.0f * xx - yy), SH(9)));
                result = v3_add(result, v3_scale(SH_C3[1] * xy * z, SH(10)));
                result = v3_add(result, v3_scale(SH_C3[2] * y * (4.0f * zz - xx - yy), SH(11)));
                result = v3_add(result, v3_scale(SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), SH(12)));
                result = v3_add(result, v3_scale(SH_C3[4] * x * (4.0f * zz - xx - yy), SH(13)));
                result = v3_add(result, v3_scale(SH_C3[5] * z * (xx - yy), SH(14)));
                result = v3_add(result, v3_scale(SH_C3[6] * x * (xx - 3.0f * yy), SH(15)));
            }
        }
    }
#undef SH
    result.x += 0.5f; result.y += 0.5f; result.z += 0.5f;
    clamped_out[0] = (result.x < 0);
    clamped_out[1] = (result.y < 0);
    clamped_out[2] = (result.z < 0);
    rgb_out[0] = fmaxf_cuda(result.x, 0.0f);
    rgb_out[1] = fmaxf_cuda(result.y, 0.0f);
    rgb_out[2] = fmaxf_cuda(result.z, 0.0f);
}

/* CR/forward.cu:158-259 (preprocessCUDA) incl. in_frustum CR/auxiliary.h:139-164 */
static void preprocess_one(const orc_inputs *in, orc_state *st, float fx, float fy, int idx)
{
    st->radii[idx] = 0;
    st->tiles_touched[idx] = 0;

    v3 p_orig = v3_make(in->means3D[3 * idx], in->means3D[3 * idx + 1], in->means3D[3 * idx + 2]);
    v3 p_view = transformPoint4x3(p_orig, in->viewmatrix);
    if (p_view.z <= 0.2f) {
        if (in->prefiltered) abort(); /* __trap() in the reference */
        return;
    }
    const float *pm = in->projmatrix;
    float hx = ((pm[0] * p_orig.x + pm[4] * p_orig.y) + pm[8] * p_orig.z) + pm[12];
    float hy = ((pm[1] * p_orig.x + pm[5] * p_orig.y) + pm[9] * p_orig.z) + pm[13];
    float hw = transformPoint4x4_w(p_orig, pm);
    float p_w = 1.0f / (hw + 0.0000001f);
    float p_proj_x = hx * p_w, p_proj_y = hy * p_w;

    const float *cov3D;
    if (in->cov3D_precomp) {
        cov3D = in->cov3D_precomp + 6 * idx;
    } else {
        computeCov3D(in->scales + 3 * idx, in->scale_modifier, in->rotations + 4 * idx, st->cov3D + 6 * idx);
        cov3D = st->cov3D + 6 * idx;
    }

    cov2d_ctx c = cov2d_common(p_orig, fx, fy, in->tanfovx, in->tanfovy, cov3D, in->viewmatrix);
    float cov_x = c.cov.m[0][0] + 0.3f; /* CR/forward.cu:112-113: dilation is ON */
    float cov_y = c.cov.m[0][1];
    float cov_z = c.cov.m[1][1] + 0.3f;

    float det = (cov_x * cov_z - cov_y * cov_y);
    if (det == 0.0f) return;
    float det_inv = 1.f / det;
    float conic_x = cov_z * det_inv, conic_y = -cov_y * det_inv, conic_z = cov_x * det_inv;

    float mid = 0.5f * (cov_x + cov_z);
    float lambda1 = mid + sqrtf(fmaxf_cuda(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf_cuda(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf_cuda(lambda1, lambda2)));
    float pix_x = ndc2Pix(p_proj_x, in->W), pix_y = ndc2Pix(p_proj_y, in->H);
    uint32_t rmin[2], rmax[2];
    getRect(pix_x, pix_y, (int)my_radius, rmin, rmax, st->gridx, st->gridy);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) return;

    if (!in->colors_precomp)
        orc_sh_to_rgb(in->D, in->M, in->means3D + 3 * idx, in->campos, in->shs + (size_t)3 * in->M * idx,
                      st->rgb + 3 * idx, st->clamped + 3 * idx);

    st->depths[idx] = p_view.z;
    st->radii[idx] = (int)my_radius;
    st->means2D[2 * idx] = pix_x;
    st->means2D[2 * idx + 1] = pix_y;
    st->conic_opacity[4 * idx + 0] = conic_x;
    st->conic_opacity[4 * idx + 1] = conic_y;
    st->conic_opacity[4 * idx + 2] = conic_z;
    st->conic_opacity[4 * idx + 3] = in->opacities[idx];
    st->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
}

/* CR/rasterizer_impl.cu:35-50 */
uint32_t orc_get_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* cub::DeviceScan::InclusiveSum call, CR/rasterizer_impl.cu:277 (u32, wraps like the device scan) */
void orc_inclusive_scan_u32(int64_t n, const uint32_t *in, uint32_t *out)
{
    uint32_t acc = 0;
    for (int64_t i = 0; i < n; i++) { acc += in[i]; out[i] = acc; }
}

/* the same scan on `nthreads` threads: per-chunk sums, then per-chunk scans from the chunk's base (u32 wrap-around included) */
static void inclusive_scan_u32_mt(int64_t n, const uint32_t *in, uint32_t *out, int nthreads)
{
    if (nthreads <= 1 || n < 65536) { orc_inclusive_scan_u32(n, in, out); return; }
    uint32_t *part = (uint32_t *)calloc((size_t)nthreads + 1, sizeof(uint32_t));
    const int64_t chunk = (n + nthreads - 1) / nthreads;
#pragma omp parallel num_threads(nthreads)
    {
        /* the runtime may grant fewer threads than asked for (OMP_THREAD_LIMIT, dynamic teams): chunks are dealt to the team
         * that actually runs, never indexed by the thread number alone */
        const int me = omp_get_thread_num(), team = omp_get_num_threads();
        for (int t = me; t < nthreads; t += team) {
            const int64_t a = t * chunk < n ? t * chunk : n, b = a + chunk < n ? a + chunk : n;
            uint32_t acc = 0;
            for (int64_t i = a; i < b; i++) acc += in[i];
            part[t + 1] = acc;
        }
#pragma omp barrier
#pragma omp single
        for (int k = 0; k < nthreads; k++) part[k + 1] += part[k];
        for (int t = me; t < nthreads; t += team) {
            const int64_t a = t * chunk < n ? t * chunk : n, b = a + chunk < n ? a + chunk : n;
            uint32_t acc = part[t];
            for (int64_t i = a; i < b; i++) { acc += in[i]; out[i] = acc; }
        }
    }
    free(part);
}

/* CR/rasterizer_impl.cu:70-111: one (parallel) loop iteration per Gaussian, like the kernel's one thread per Gaussian */
static void duplicate_with_keys(const orc_state *st, int nthreads)
{
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int idx = 0; idx < st->P; idx++) {
        if (st->radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : st->point_offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            getRect(st->means2D[2 * idx], st->means2D[2 * idx + 1], st->radii[idx], rmin, rmax, st->gridx, st->gridy);
            uint32_t dbits;
            memcpy(&dbits, &st->depths[idx], 4);
            for (uint32_t y = rmin[1]; y < rmax[1]; y++)
                for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * (uint32_t)st->gridx + x);
                    key <<= 32;
                    key |= dbits;
                    st->keys_unsorted[off] = key;
                    st->vals_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
}

/* cub::DeviceRadixSort::SortPairs(…, begin_bit=0, end_bit) call, CR/rasterizer_impl.cu:303-308:
 * stable ascending sort on key bits [0,end_bit).  LSD, 8-bit digits. */
void orc_sort_pairs(int64_t n, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, int end_bit)
{
    if (n <= 0) return;
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * n), *kb = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * n), *vb = (uint32_t *)malloc(sizeof(uint32_t) * n);
    memcpy(ka, kin, sizeof(uint64_t) * n);
    memcpy(va, vin, sizeof(uint32_t) * n);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        int64_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < n; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < n; i++) {
            int64_t dst = cnt[(ka[i] >> shift) & mask]++;
            kb[dst] = ka[i];
            vb[dst] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(kout, ka, sizeof(uint64_t) * n);
    memcpy(vout, va, sizeof(uint32_t) * n);
    free(ka); free(kb); free(va); free(vb);
}

/* The same stable LSD sort on `nthreads` threads: every thread owns a contiguous chunk, counts its digits, and scatters its chunk
 * in order to offsets[digit][thread] = (all smaller digits) + (the same digit in earlier chunks) -- stability is kept because
 * chunks are taken in order and each chunk is walked in order. */
static void sort_pairs_mt(int64_t n, const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, int end_bit, int nthreads)
{
    if (nthreads <= 1 || n < 65536) { orc_sort_pairs(n, kin, vin, kout, vout, end_bit); return; }
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * n), *kb = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * n), *vb = (uint32_t *)malloc(sizeof(uint32_t) * n);
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * 256 * (size_t)nthreads);
    const int64_t chunk = (n + nthreads - 1) / nthreads;
#pragma omp parallel num_threads(nthreads)
    {
        /* `nthreads` chunks, dealt to whatever team the runtime grants (see inclusive_scan_u32_mt) */
        const int me = omp_get_thread_num(), team = omp_get_num_threads();
#define CHUNK_BOUNDS(t) const int64_t a = (t) * chunk < n ? (t) * chunk : n, b = a + chunk < n ? a + chunk : n
        for (int t = me; t < nthreads; t += team) {
            CHUNK_BOUNDS(t);
            memcpy(ka + a, kin + a, sizeof(uint64_t) * (size_t)(b - a));
            memcpy(va + a, vin + a, sizeof(uint32_t) * (size_t)(b - a));
        }
#pragma omp barrier
        for (int shift = 0; shift < end_bit; shift += 8) {
            const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
            const uint32_t mask = (1u << bits) - 1u;
            const uint64_t *ks = ((shift / 8) & 1) ? kb : ka;
            const uint32_t *vs = ((shift / 8) & 1) ? vb : va;
            uint64_t *kd = ((shift / 8) & 1) ? ka : kb;
            uint32_t *vd = ((shift / 8) & 1) ? va : vb;
            for (int t = me; t < nthreads; t += team) {
                CHUNK_BOUNDS(t);
                int64_t *mine = cnt + 256 * (size_t)t;
                memset(mine, 0, sizeof(int64_t) * 256);
                for (int64_t i = a; i < b; i++) mine[(ks[i] >> shift) & mask]++;
            }
#pragma omp barrier
#pragma omp single
            {
                int64_t run = 0;
                for (int d = 0; d < 256; d++)
                    for (int k = 0; k < nthreads; k++) { const int64_t c = cnt[256 * (size_t)k + d]; cnt[256 * (size_t)k + d] = run; run += c; }
            }
            for (int t = me; t < nthreads; t += team) {
                CHUNK_BOUNDS(t);
                int64_t *mine = cnt + 256 * (size_t)t;
                for (int64_t i = a; i < b; i++) {
                    const int64_t dst = mine[(ks[i] >> shift) & mask]++;
                    kd[dst] = ks[i];
                    vd[dst] = vs[i];
                }
            }
#pragma omp barrier
        }
        const int passes = (end_bit + 7) / 8;
        const uint64_t *kf = (passes & 1) ? kb : ka;
        const uint32_t *vf = (passes & 1) ? vb : va;
        for (int t = me; t < nthreads; t += team) {
            CHUNK_BOUNDS(t);
            memcpy(kout + a, kf + a, sizeof(uint64_t) * (size_t)(b - a));
            memcpy(vout + a, vf + a, sizeof(uint32_t) * (size_t)(b - a));
        }
#undef CHUNK_BOUNDS
    }
    free(ka); free(kb); free(va); free(vb); free(cnt);
}

/* CR/rasterizer_impl.cu:116-138; ranges must be zeroed first (cudaMemset, :310) */
static void identify_tile_ranges_mt(int64_t L, const uint64_t *keys, uint32_t *ranges, int nthreads)
{
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t idx = 0; idx < L; idx++) {
        uint32_t currtile = (uint32_t)(keys[idx] >> 32);
        if (idx == 0)
            ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)idx;
                ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == L - 1) ranges[2 * currtile + 1] = (uint32_t)L;
    }
}

void orc_identify_tile_ranges(int64_t L, const uint64_t *keys, uint32_t *ranges) { identify_tile_ranges_mt(L, keys, ranges, 1); }

/* CR/forward.cu:264-377 (renderCUDA), one 16x16 tile.  The 256-entry staging rounds and the block-wide
 * early exit of the kernel do not change any pixel's arithmetic; each pixel walks the tile list in order
 * until it is done or the list ends. */
static void render_tile_forward(const orc_inputs *in, orc_state *st, const float *features, int tx, int ty,
                                int64_t *consumed_fwd, int64_t *consumed_bwd)
{
    const int W = in->W, H = in->H;
    const uint32_t r0 = st->ranges[2 * (ty * st->gridx + tx)], r1 = st->ranges[2 * (ty * st->gridx + tx) + 1];
    const int toDo = (int)(r1 - r0);
    uint32_t tile_need = 0, tile_last = 0;
    for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
            uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
            uint32_t pix_id = W * py + px;
            float pixf_x = (float)px, pixf_y = (float)py;
            float T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0;
            float C[3] = {0, 0, 0};
            int done = 0;
            for (int j = 0; !done && j < toDo; j++) {
                contributor++;
                uint32_t id = st->vals[r0 + j];
                float dx = st->means2D[2 * id] - pixf_x, dy = st->means2D[2 * id + 1] - pixf_y;
                const float *co = st->conic_opacity + 4 * id;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float alpha = fminf_cuda(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) { done = 1; continue; }
                for (int ch = 0; ch < 3; ch++) C[ch] += features[id * 3 + ch] * alpha * T;
                T = test_T;
                last_contributor = contributor;
            }
            st->final_T[pix_id] = T;
            st->n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < 3; ch++) st->out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * in->bg[ch];
            /* entries this pixel had to look at: up to and including the one that stopped it */
            if (contributor > tile_need) tile_need = contributor;
            if (last_contributor > tile_last) tile_last = last_contributor;
        }
    *consumed_fwd += tile_need;
    *consumed_bwd += tile_last;
}

static orc_state *state_alloc(const orc_inputs *in)
{
    orc_state *st = (orc_state *)calloc(1, sizeof(orc_state));
    const size_t P = (size_t)in->P, N = (size_t)in->W * in->H;
    st->P = in->P; st->W = in->W; st->H = in->H;
    st->gridx = (in->W + BLOCK_X - 1) / BLOCK_X;
    st->gridy = (in->H + BLOCK_Y - 1) / BLOCK_Y;
    const size_t T = (size_t)st->gridx * st->gridy;
    st->depths = (float *)calloc(P + 1, 4);
    st->clamped = (uint8_t *)calloc(3 * P + 1, 1);
    st->radii = (int32_t *)calloc(P + 1, 4);
    st->means2D = (float *)calloc(2 * P + 1, 4);
    st->cov3D = (float *)calloc(6 * P + 1, 4);
    st->conic_opacity = (float *)calloc(4 * P + 1, 4);
    st->rgb = (float *)calloc(3 * P + 1, 4);
    st->tiles_touched = (uint32_t *)calloc(P + 1, 4);
    st->point_offsets = (uint32_t *)calloc(P + 1, 4);
    st->ranges = (uint32_t *)calloc(2 * T + 1, 4);
    st->final_T = (float *)calloc(N + 1, 4);
    st->n_contrib = (uint32_t *)calloc(N + 1, 4);
    st->out_color = (float *)calloc(3 * N + 1, 4);
    return st;
}

void orc_free(orc_state *st)
{
    if (!st) return;
    free(st->depths); free(st->clamped); free(st->radii); free(st->means2D); free(st->cov3D);
    free(st->conic_opacity); free(st->rgb); free(st->tiles_touched); free(st->point_offsets);
    free(st->keys_unsorted); free(st->vals_unsorted); free(st->keys); free(st->vals);
    free(st->ranges); free(st->final_T); free(st->n_contrib); free(st->out_color);
    free(st);
}

/* CR/rasterizer_impl.cu:198-336 */
orc_state *orc_forward(const orc_inputs *in, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    orc_state *st = state_alloc(in);
    if (in->P == 0) return st; /* rasterize_points.cu:81: P==0 leaves the zero image (no background) */
    const float focal_y = in->H / (2.0f * in->tanfovy);
    const float focal_x = in->W / (2.0f * in->tanfovx);

#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < in->P; i++) preprocess_one(in, st, focal_x, focal_y, i);

    inclusive_scan_u32_mt(in->P, st->tiles_touched, st->point_offsets, nthreads);
    st->R = (int64_t)(int32_t)st->point_offsets[in->P - 1]; /* int num_rendered, :280-281 */
    int64_t vis = 0;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : vis)
    for (int i = 0; i < in->P; i++) vis += st->radii[i] > 0;
    st->visible = vis;

    const size_t R = (size_t)st->R;
    st->keys_unsorted = (uint64_t *)calloc(R + 1, 8);
    st->vals_unsorted = (uint32_t *)calloc(R + 1, 4);
    st->keys = (uint64_t *)calloc(R + 1, 8);
    st->vals = (uint32_t *)calloc(R + 1, 4);
    duplicate_with_keys(st, nthreads);

    int bit = (int)orc_get_higher_msb((uint32_t)(st->gridx * st->gridy));
    sort_pairs_mt(st->R, st->keys_unsorted, st->vals_unsorted, st->keys, st->vals, 32 + bit, nthreads);

    if (st->R > 0) identify_tile_ranges_mt(st->R, st->keys, st->ranges, nthreads);   /* (no omp_set_num_threads: process-wide) */

    const float *feature_ptr = in->colors_precomp ? in->colors_precomp : st->rgb;
    int64_t cf = 0, cb = 0;
    const int ntiles = st->gridx * st->gridy;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4) reduction(+ : cf, cb)
    for (int t = 0; t < ntiles; t++)
        render_tile_forward(in, st, feature_ptr, t % st->gridx, t / st->gridx, &cf, &cb);
    st->consumed_fwd = cf;
    st->consumed_bwd = cb;
    return st;
}

/* CR/rasterizer_impl.cu:54-66 */
void orc_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present)
{
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        v3 p = v3_make(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
        v3 pv = transformPoint4x3(p, viewmatrix);
        present[i] = !(pv.z <= 0.2f);
    }
}

/* ------------------------------------------------------------------ backward */

/* CR/backward.cu:399-557 (renderCUDA bwd), one tile.  acc_* are double accumulators standing in for the
 * reference's float atomicAdd targets.  The 256 pixels of the tile first add into a buffer private to the calling thread,
 * one row of nine doubles per list entry; the rows are then added to the shared accumulators, one atomic per (tile, entry,
 * component) instead of one per (pixel, entry, component). */
static void render_tile_backward(const orc_inputs *in, const orc_state *st, const float *colors, const float *dL_dpix,
                                 int tx, int ty, double *acc_mean2D, double *acc_conic, double *acc_opacity,
                                 double *acc_color, double *loc /* the calling thread's rows: >= 9 x (longest list) doubles */)
{
    const int W = in->W, H = in->H;
    const uint32_t r0 = st->ranges[2 * (ty * st->gridx + tx)], r1 = st->ranges[2 * (ty * st->gridx + tx) + 1];
    const int toDo = (int)(r1 - r0);
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
    if (toDo <= 0) return;
    memset(loc, 0, (size_t)toDo * 9 * sizeof(double));                  /* row = list position: x y | conic 0 1 3 | opacity | r g b */
    for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
            uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
            uint32_t pix_id = W * py + px;
            float pixf_x = (float)px, pixf_y = (float)py;
            const float T_final = st->final_T[pix_id];
            float T = T_final;
            uint32_t contributor = (uint32_t)toDo;
            const int last_contributor = (int)st->n_contrib[pix_id];
            float accum_rec[3] = {0, 0, 0}, dL_dpixel[3], last_color[3] = {0, 0, 0};
            for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpix[(size_t)i * H * W + pix_id];
            float last_alpha = 0;
            for (int j = 0; j < toDo; j++) {
                contributor--;
                if (contributor >= (uint32_t)last_contributor) continue; /* unsigned compare as in :483 */
                uint32_t id = st->vals[r1 - 1 - j];
                double *row = loc + (size_t)(toDo - 1 - j) * 9;
                float dx = st->means2D[2 * id] - pixf_x, dy = st->means2D[2 * id + 1] - pixf_y;
                const float *co = st->conic_opacity + 4 * id;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                float G = expf(power);
                float alpha = fminf_cuda(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;

                T = T / (1.f - alpha);
                float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                for (int ch = 0; ch < 3; ch++) {
                    float c = colors[id * 3 + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    float dL_dchannel = dL_dpixel[ch];
                    dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                    float add = dchannel_dcolor * dL_dchannel;
                    row[6 + ch] += (double)add;
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot_dpixel = 0;
                for (int i = 0; i < 3; i++) bg_dot_dpixel += in->bg[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                float dL_dG = co[3] * dL_dalpha;
                float gdx = G * dx, gdy = G * dy;
                float dG_ddelx = -gdx * co[0] - gdy * co[1];
                float dG_ddely = -gdy * co[2] - gdx * co[1];
                float a0 = dL_dG * dG_ddelx * ddelx_dx, a1 = dL_dG * dG_ddely * ddely_dy;
                float c0 = -0.5f * gdx * dx * dL_dG, c1 = -0.5f * gdx * dy * dL_dG, c3 = -0.5f * gdy * dy * dL_dG;
                float o0 = G * dL_dalpha;
                row[0] += (double)a0;
                row[1] += (double)a1;
                row[2] += (double)c0;
                row[3] += (double)c1;
                row[4] += (double)c3;
                row[5] += (double)o0;
            }
        }
    for (int k = 0; k < toDo; k++) {
        const double *row = loc + (size_t)k * 9;
        int any = 0;
        for (int c = 0; c < 9; c++) any |= row[c] != 0.0;
        if (!any) continue;
        const uint32_t id = st->vals[r0 + k];
#pragma omp atomic
        acc_mean2D[(size_t)id * 3 + 0] += row[0];
#pragma omp atomic
        acc_mean2D[(size_t)id * 3 + 1] += row[1];
#pragma omp atomic
        acc_conic[(size_t)id * 4 + 0] += row[2];
#pragma omp atomic
        acc_conic[(size_t)id * 4 + 1] += row[3];
#pragma omp atomic
        acc_conic[(size_t)id * 4 + 3] += row[4];
#pragma omp atomic
        acc_opacity[id] += row[5];
#pragma omp atomic
        acc_color[(size_t)id * 3 + 0] += row[6];
#pragma omp atomic
        acc_color[(size_t)id * 3 + 1] += row[7];
#pragma omp atomic
        acc_color[(size_t)id * 3 + 2] += row[8];
    }
}

/* The ARBITER of the render backward (not a restatement of any reference arithmetic): the same sums as render_tile_backward with
 * every per-(pixel, entry) term evaluated in DOUBLE -- dx, power, exp, alpha, the transmittance (rebuilt as the forward product
 * over the contributing entries), the accum_rec recurrence, the nine partial derivatives -- from the float render inputs
 * (means2D, conic_opacity, colours, dL_dpix, bg) and with the float forward's DECISIONS (which entries a pixel skips, where it
 * stops: those define the piecewise-smooth function being differentiated and are taken from the float arithmetic above, bit for
 * bit).  render_tile_backward accumulates in double too, but its terms are the reference's float32 terms, so against it the
 * reference build shows only its summation error while any other float32 formulation also shows its (equally legitimate)
 * per-term rounding; against THIS function both show their whole error.  out: rows of 9 doubles per Gaussian:
 * mean2D.x, .y | conic.x, .y, .w | opacity | colour r g b. */
static void render_tile_backward_fp64(const orc_inputs *in, const orc_state *st, const float *colors, const float *dL_dpix,
                                      int tx, int ty, double *out9, double *loc, double *alpha_d /* >= longest list */)
{
    const int W = in->W, H = in->H;
    const uint32_t r0 = st->ranges[2 * (ty * st->gridx + tx)], r1 = st->ranges[2 * (ty * st->gridx + tx) + 1];
    const int toDo = (int)(r1 - r0);
    if (toDo <= 0) return;
    memset(loc, 0, (size_t)toDo * 9 * sizeof(double));
    for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
            const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
            const uint32_t pix_id = W * py + px;
            const float pixf_x = (float)px, pixf_y = (float)py;
            const int last_contributor = (int)st->n_contrib[pix_id];
            /* forward walk: float decisions, double alpha and transmittance; alpha_d[k] < 0 marks a skipped entry */
            double T = 1.0;
            for (int k = 0; k < last_contributor && k < toDo; k++) {
                const uint32_t id = st->vals[r0 + k];
                const float *co = st->conic_opacity + 4 * id;
                const float dx = st->means2D[2 * id] - pixf_x, dy = st->means2D[2 * id + 1] - pixf_y;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                alpha_d[k] = -1.0;
                if (power > 0.0f) continue;
                const float alpha = fminf_cuda(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                const double ddx = (double)st->means2D[2 * id] - (double)pixf_x, ddy = (double)st->means2D[2 * id + 1] - (double)pixf_y;
                const double pw = -0.5 * ((double)co[0] * ddx * ddx + (double)co[2] * ddy * ddy) - (double)co[1] * ddx * ddy;
                double a = (double)co[3] * exp(pw);
                if (a > (double)0.99f) a = (double)0.99f;
                alpha_d[k] = a;
                T *= 1.0 - a;
            }
            const double T_final = T;
            double accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, dL_dpixel[3], last_alpha = 0, bg_dot = 0;
            for (int i = 0; i < 3; i++) dL_dpixel[i] = (double)dL_dpix[(size_t)i * H * W + pix_id];
            for (int i = 0; i < 3; i++) bg_dot += (double)in->bg[i] * dL_dpixel[i];
            for (int k = (last_contributor < toDo ? last_contributor : toDo) - 1; k >= 0; k--) {
                const double alpha = alpha_d[k];
                if (alpha < 0) continue;
                const uint32_t id = st->vals[r0 + k];
                const float *co = st->conic_opacity + 4 * id;
                double *row = loc + (size_t)k * 9;
                const double dx = (double)st->means2D[2 * id] - (double)pixf_x, dy = (double)st->means2D[2 * id + 1] - (double)pixf_y;
                const double G = exp(-0.5 * ((double)co[0] * dx * dx + (double)co[2] * dy * dy) - (double)co[1] * dx * dy);
                T = T / (1.0 - alpha);
                double dL_dalpha = 0;
                for (int ch = 0; ch < 3; ch++) {
                    const double c = (double)colors[id * 3 + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.0 - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dalpha += (c - accum_rec[ch]) * dL_dpixel[ch];
                    row[6 + ch] += alpha * T * dL_dpixel[ch];
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.0 - alpha)) * bg_dot;
                const double dL_dG = (double)co[3] * dL_dalpha, gdx = G * dx, gdy = G * dy;
                row[0] += dL_dG * (-gdx * (double)co[0] - gdy * (double)co[1]) * (0.5 * W);
                row[1] += dL_dG * (-gdy * (double)co[2] - gdx * (double)co[1]) * (0.5 * H);
                row[2] += -0.5 * gdx * dx * dL_dG;
                row[3] += -0.5 * gdx * dy * dL_dG;
                row[4] += -0.5 * gdy * dy * dL_dG;
                row[5] += G * dL_dalpha;
            }
        }
    for (int k = 0; k < toDo; k++) {
        const double *row = loc + (size_t)k * 9;
        double *dst = out9 + (size_t)st->vals[r0 + k] * 9;
        for (int c = 0; c < 9; c++)
            if (row[c] != 0.0) {
#pragma omp atomic
                dst[c] += row[c];
            }
    }
}

/* render-level gradient sums in double (see render_tile_backward_fp64); out9 [P][9] must be zero-filled by the caller */
void orc_render_backward_fp64(const orc_inputs *in, const orc_state *st, const float *dL_dpix, double *out9, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (in->P == 0) return;
    const float *color_ptr = in->colors_precomp ? in->colors_precomp : st->rgb;
    const int ntiles = st->gridx * st->gridy;
    size_t longest = 1;
    for (int t = 0; t < ntiles; t++) {
        const size_t len = (size_t)(st->ranges[2 * t + 1] - st->ranges[2 * t]);
        if (len > longest) longest = len;
    }
    double *rows = (double *)malloc(sizeof(double) * 10 * longest * (size_t)nthreads);
#pragma omp parallel num_threads(nthreads)
    {
        double *loc = rows + (size_t)omp_get_thread_num() * 10 * longest;
#pragma omp for schedule(dynamic, 4)
        for (int t = 0; t < ntiles; t++)
            render_tile_backward_fp64(in, st, color_ptr, dL_dpix, t % st->gridx, t / st->gridx, out9, loc, loc + 9 * longest);
    }
    free(rows);
}

/* MODEL of an alternative float32 formulation of the per-(pixel, entry) weights (not a restatement of reference arithmetic; used by
 * scripts/bwd_formulations.py to price a formulation's rounding against orc_render_backward_fp64 before it is built on the GPU).
 * The nine partial derivatives are formed from (T, dL_dalpha) exactly like render_tile_backward (float terms, double sums);
 * what differs is how T and dL_dalpha are obtained:
 *   mode 0: the reference's back-to-front walk (T = T / (1 - alpha), accum_rec recurrence) -- equals render_tile_backward
 *   mode 1: FRONT-TO-BACK: T by the forward's own multiply chain (exact), S = sum_{j<=i} w_j d_j (d = c . dL_dpixel, w = alpha T)
 *           accumulated with fmaf, dL_dalpha = T d - (S_tot - S) * (1 / (1 - alpha)), S_tot = C_final . dL_dpixel + T_final bg . dL_dpixel
 *           with C_final the float forward's accumulated colour.   */
static void render_tile_backward_model(const orc_inputs *in, const orc_state *st, const float *colors, const float *dL_dpix,
                                       int tx, int ty, double *out9, double *loc, int mode)
{
    const int W = in->W, H = in->H;
    const uint32_t r0 = st->ranges[2 * (ty * st->gridx + tx)], r1 = st->ranges[2 * (ty * st->gridx + tx) + 1];
    const int toDo = (int)(r1 - r0);
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
    if (toDo <= 0) return;
    memset(loc, 0, (size_t)toDo * 9 * sizeof(double));
    for (int ly = 0; ly < BLOCK_Y; ly++)
        for (int lx = 0; lx < BLOCK_X; lx++) {
            const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
            const uint32_t pix_id = W * py + px;
            const float pixf_x = (float)px, pixf_y = (float)py;
            const float T_final = st->final_T[pix_id];
            const int last = (int)st->n_contrib[pix_id] < toDo ? (int)st->n_contrib[pix_id] : toDo;
            float dpx[3], bg_dot = 0;
            for (int i = 0; i < 3; i++) dpx[i] = dL_dpix[(size_t)i * H * W + pix_id];
            for (int i = 0; i < 3; i++) bg_dot += in->bg[i] * dpx[i];
            /* forward colour of the pixel as the float forward accumulates it */
            float Cf[3] = {0, 0, 0}, Tf = 1.f;
            for (int k = 0; k < last; k++) {
                const uint32_t id = st->vals[r0 + k];
                const float *co = st->conic_opacity + 4 * id;
                const float dx = st->means2D[2 * id] - pixf_x, dy = st->means2D[2 * id + 1] - pixf_y;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf_cuda(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                for (int ch = 0; ch < 3; ch++) Cf[ch] += colors[id * 3 + ch] * alpha * Tf;
                Tf = Tf * (1 - alpha);
            }
            const float S_tot = fmaf(T_final, bg_dot, fmaf(Cf[2], dpx[2], fmaf(Cf[1], dpx[1], Cf[0] * dpx[0])));
            float T = mode == 0 ? T_final : 1.f, S = 0.f;
            float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
            for (int j = 0; j < last; j++) {
                const int k = mode == 0 ? last - 1 - j : j;
                const uint32_t id = st->vals[r0 + k];
                double *row = loc + (size_t)k * 9;
                const float *co = st->conic_opacity + 4 * id;
                const float dx = st->means2D[2 * id] - pixf_x, dy = st->means2D[2 * id + 1] - pixf_y;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf_cuda(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                float dL_dalpha, Ti;
                if (mode == 0) {
                    T = T / (1.f - alpha);
                    Ti = T;
                    dL_dalpha = 0.f;
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = colors[id * 3 + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * dpx[ch];
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                } else {
                    const float om = 1.f - alpha, rcp = 1.f / om;
                    const float d = fmaf(colors[id * 3 + 2], dpx[2], fmaf(colors[id * 3 + 1], dpx[1], colors[id * 3] * dpx[0]));
                    Ti = T;
                    S = fmaf(alpha * T, d, S);
                    dL_dalpha = fmaf(-(S_tot - S), rcp, T * d);
                    T = T * om;
                }
                const float dchannel_dcolor = alpha * Ti;
                for (int ch = 0; ch < 3; ch++) row[6 + ch] += (double)(dchannel_dcolor * dpx[ch]);
                const float dL_dG = co[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co[0] - gdy * co[1], dG_ddely = -gdy * co[2] - gdx * co[1];
                row[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                row[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                row[2] += (double)(-0.5f * gdx * dx * dL_dG);
                row[3] += (double)(-0.5f * gdx * dy * dL_dG);
                row[4] += (double)(-0.5f * gdy * dy * dL_dG);
                row[5] += (double)(G * dL_dalpha);
            }
        }
    for (int k = 0; k < toDo; k++) {
        const double *row = loc + (size_t)k * 9;
        double *dst = out9 + (size_t)st->vals[r0 + k] * 9;
        for (int c = 0; c < 9; c++)
            if (row[c] != 0.0) {
#pragma omp atomic
                dst[c] += row[c];
            }
    }
}

/* see render_tile_backward_model; out9 [P][9] zero-filled by the caller, rows as orc_render_backward_fp64 */
void orc_render_backward_model(const orc_inputs *in, const orc_state *st, const float *dL_dpix, double *out9, int mode, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (in->P == 0) return;
    const float *color_ptr = in->colors_precomp ? in->colors_precomp : st->rgb;
    const int ntiles = st->gridx * st->gridy;
    size_t longest = 1;
    for (int t = 0; t < ntiles; t++) {
        const size_t len = (size_t)(st->ranges[2 * t + 1] - st->ranges[2 * t]);
        if (len > longest) longest = len;
    }
    double *rows = (double *)malloc(sizeof(double) * 9 * longest * (size_t)nthreads);
#pragma omp parallel num_threads(nthreads)
    {
        double *loc = rows + (size_t)omp_get_thread_num() * 9 * longest;
#pragma omp for schedule(dynamic, 4)
        for (int t = 0; t < ntiles; t++)
            render_tile_backward_model(in, st, color_ptr, dL_dpix, t % st->gridx, t / st->gridx, out9, loc, mode);
    }
    free(rows);
}

/* CR/backward.cu:144-274 (computeCov2DCUDA) */
static void cov2d_backward_one(const orc_inputs *in, const orc_state *st, const float *cov3Ds, float h_x, float h_y,
                               const float *dL_dconics, float *dL_dmeans, float *dL_dcov, int idx)
{
    if (!(st->radii[idx] > 0)) return;
    const float *cov3D = cov3Ds + 6 * idx;
    v3 mean = v3_make(in->means3D[3 * idx], in->means3D[3 * idx + 1], in->means3D[3 * idx + 2]);
    v3 dL_dconic = v3_make(dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]);
    cov2d_ctx c = cov2d_common(mean, h_x, h_y, in->tanfovx, in->tanfovy, cov3D, in->viewmatrix);
    const v3 t = c.t;
    const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
    const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
    const m3 T = c.T, W = c.W, Vrk = c.Vrk;

    float a = c.cov.m[0][0] + 0.3f;
    float b = c.cov.m[0][1];
    float cc = c.cov.m[1][1] + 0.3f;

    float denom = a * cc - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);

    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dL_dconic.x + 2 * b * cc * dL_dconic.y + (denom - a * cc) * dL_dconic.z);
        dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * cc) * dL_dconic.x);
        dL_db = denom2inv * 2 * (b * cc * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);

        dL_dcov[6 * idx + 0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
        dL_dcov[6 * idx + 3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
        dL_dcov[6 * idx + 5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);

        dL_dcov[6 * idx + 1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
        dL_dcov[6 * idx + 2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
        dL_dcov[6 * idx + 4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
    }

    float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
    float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
    float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                    (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
    float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
    float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
    float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                    (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;

    float dL_dJ00 = W.m[0][0] * dL_dT00 + W.m[0][1] * dL_dT01 + W.m[0][2] * dL_dT02;
    float dL_dJ02 = W.m[2][0] * dL_dT00 + W.m[2][1] * dL_dT01 + W.m[2][2] * dL_dT02;
    float dL_dJ11 = W.m[1][0] * dL_dT10 + W.m[1][1] * dL_dT11 + W.m[1][2] * dL_dT12;
    float dL_dJ12 = W.m[2][0] * dL_dT10 + W.m[2][1] * dL_dT11 + W.m[2][2] * dL_dT12;

    float tz = 1.f / t.z;
    float tz2 = tz * tz;
    float tz3 = tz2 * tz;

    float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;

    v3 dL_dmean = transformVec4x3Transpose(v3_make(dL_dtx, dL_dty, dL_dtz), in->viewmatrix);
    dL_dmeans[3 * idx + 0] = dL_dmean.x; /* assignment, CR/backward.cu:273 */
    dL_dmeans[3 * idx + 1] = dL_dmean.y;
    dL_dmeans[3 * idx + 2] = dL_dmean.z;
}

/* CR/backward.cu:20-139 */
static void sh_backward_one(const orc_inputs *in, const orc_state *st, int idx, const float *dL_dcolor, float *dL_dmeans,
                            float *dL_dshs)
{
    const int deg = in->D, max_coeffs = in->M;
    v3 pos = v3_make(in->means3D[3 * idx], in->means3D[3 * idx + 1], in->means3D[3 * idx + 2]);
    v3 dir_orig = v3_sub(pos, v3_make(in->campos[0], in->campos[1], in->campos[2]));
    float len = sqrtf(v3_dot(dir_orig, dir_orig));
    v3 dir = v3_make(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
    const float *shp = in->shs + (size_t)3 * max_coeffs * idx;
#define SH(k) v3_make(shp[3 * (k)], shp[3 * (k) + 1], shp[3 * (k) + 2])
    v3 dL_dRGB = v3_make(dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]);
    dL_dRGB.x *= st->clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB.y *= st->clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB.z *= st->clamped[3 * idx + 2] ? 0 : 1;

    v3 dRGBdx = v3_make(0, 0, 0), dRGBdy = v3_make(0, 0, 0), dRGBdz = v3_make(0, 0, 0);
    float x = dir.x, y = dir.y, z = dir.z;
    float *dL_dsh = dL_dshs + (size_t)3 * max_coeffs * idx;
#define PUT(k, s) do { v3 _v = v3_scale((s), dL_dRGB); dL_dsh[3 * (k)] = _v.x; dL_dsh[3 * (k) + 1] = _v.y; dL_dsh[3 * (k) + 2] = _v.z; } while (0)
    /* vec * scalar chains below follow glm: (scalar*scalar...)*vec, left to right */
#define ACC(dst, s, v) dst = v3_add(dst, v3_scale((s), (v)))
    PUT(0, SH_C0);
    if (deg > 0) {
        float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
        PUT(1, dRGBdsh1); PUT(2, dRGBdsh2); PUT(3, dRGBdsh3);
        dRGBdx = v3_scale(-SH_C1, SH(3));
        dRGBdy = v3_scale(-SH_C1, SH(1));
        dRGBdz = v3_scale(SH_C1, SH(2));
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            PUT(4, SH_C2[0] * xy);
            PUT(5, SH_C2[1] * yz);
            PUT(6, SH_C2[2] * (2.f * zz - xx - yy));
            PUT(7, SH_C2[3] * xz);
            PUT(8, SH_C2[4] * (xx - yy));
            /* CR/backward.cu:78-80: dRGBdx += A + B + C + D  => dRGBdx = dRGBdx + (((A+B)+C)+D) */
            {
                v3 s = v3_scale(SH_C2[0] * y, SH(4));
                s = v3_add(s, v3_scale(SH_C2[2] * 2.f * -x, SH(6)));
                s = v3_add(s, v3_scale(SH_C2[3] * z, SH(7)));
                s = v3_add(s, v3_scale(SH_C2[4] * 2.f * x, SH(8)));
                dRGBdx = v3_add(dRGBdx, s);
            }
            {
                v3 s = v3_scale(SH_C2[0] * x, SH(4));
                s = v3_add(s, v3_scale(SH_C2[1] * z, SH(5)));
                s = v3_add(s, v3_scale(SH_C2[2] * 2.f * -y, SH(6)));
                s = v3_add(s, v3_scale(SH_C2[4] * 2.f * -y, SH(8)));
                dRGBdy = v3_add(dRGBdy, s);
            }
            {
                v3 s = v3_scale(SH_C2[1] * y, SH(5));
                s = v3_add(s, v3_scale(SH_C2[2] * 2.f * 2.f * z, SH(6)));
                s = v3_add(s, v3_scale(SH_C2[3] * x, SH(7)));
                dRGBdz = v3_add(dRGBdz, s);
            }
            if (deg > 2) {
                PUT(9, SH_C3[0] * y * (3.f * xx - yy));
                PUT(10, SH_C3[1] * xy * z);
                PUT(11, SH_C3[2] * y * (4.f * zz - xx - yy));
                PUT(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                PUT(13, SH_C3[4] * x * (4.f * zz - xx - yy));
                PUT(14, SH_C3[5] * z * (xx - yy));
                PUT(15, SH_C3[6] * x * (xx - 3.f * yy));
                /* CR/backward.cu:99-122: terms are SH_C3[k] * sh[j] * scalar...: (scalar*vec)*scalar*... */
#define VS(v, s) v3_make((v).x * (s), (v).y * (s), (v).z * (s))
                {
                    v3 s = VS(VS(VS(v3_scale(SH_C3[0], SH(9)), 3.f), 2.f), xy);
                    s = v3_add(s, VS(v3_scale(SH_C3[1], SH(10)), yz));
                    s = v3_add(s, VS(VS(v3_scale(SH_C3[2], SH(11)), -2.f), xy));
                    s = v3_add(s, VS(VS(VS(v3_scale(SH_C3[3], SH(12)), -3.f), 2.f), xz));
                    s = v3_add(s, VS(v3_scale(SH_C3[4], SH(13)), (-3.f * xx + 4.f * zz - yy)));
                    s = v3_add(s, VS(VS(v3_scale(SH_C3[5], SH(14)), 2.f), xz));
                    s = v3_add(s, VS(VS(v3_scale(SH_C3[6], SH(15)), 3.f), (xx - yy)));
                    dRGBdx = v3_add(dRGBdx, s);
                }
                {
                    v3 s = VS(VS(v3_scale(SH_C3[0], SH(9)), 3.f), (xx - yy));
                    s = v3_add(s, VS(v3_scale(SH_C3[1], SH(10)), xz));
                    s = v3_add(s, VS(v3_scale(SH_C3[2], SH(11)), (-3.f * yy + 4.f * zz - xx)));
                    s = v3_add(s, VS(VS(VS(v3_scale(SH_C3[3], SH(12)), -3.f), 2.f), yz));
                    s = v3_add(s, VS(VS(v3_scale(SH_C3[4], SH(13)), -2.f), xy));
                    s = v3_add(s, VS(VS(v3_scale(SH_C3[5], SH(14)), -2.f), yz));
                    s = v3_add(s, VS(VS(VS(v3_scale(SH_C3[6], SH(15)), -3.f), 2.f), xy));
                    dRGBdy = v3_add(dRGBdy, s);
                }
                {
                    v3 s = VS(v3_scale(SH_C3[1], SH(10)), xy);
                    s = v3_add(s, VS(VS(VS(v3_scale(SH_C3[2], SH(11)), 4.f), 2.f), yz));
                    s = v3_add(s, VS(VS(v3_scale(SH_C3[3], SH(12)), 3.f), (2.f * zz - xx - yy)));
                    s = v3_add(s, VS(VS(VS(v3_scale(SH_C3[4], SH(13)), 4.f), 2.f), xz));
                    s = v3_add(s, VS(v3_scale(SH_C3[5], SH(14)), (xx - yy)));
                    dRGBdz = v3_add(dRGBdz, s);
                }
#undef VS
            }
        }
    }
#undef ACC
#undef PUT
#undef SH
    v3 dL_ddir = v3_make(v3_dot(dRGBdx, dL_dRGB), v3_dot(dRGBdy, dL_dRGB), v3_dot(dRGBdz, dL_dRGB));
    v3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
    dL_dmeans[3 * idx + 0] += dL_dmean.x;
    dL_dmeans[3 * idx + 1] += dL_dmean.y;
    dL_dmeans[3 * idx + 2] += dL_dmean.z;
}

/* CR/backward.cu:278-341 */
static void cov3d_backward_one(int idx, const float *scale, float mod, const float *rot, const float *dL_dcov3Ds,
                               float *dL_dscales, float *dL_drots)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = quat_to_R(rot);
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    v3 s = v3_make(mod * scale[0], mod * scale[1], mod * scale[2]);
    S.m[0][0] = s.x; S.m[1][1] = s.y; S.m[2][2] = s.z;
    m3 M = m3_mul(S, R);
    const float *d = dL_dcov3Ds + 6 * idx;
    m3 dL_dSigma = m3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
    m3 M2;
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = M.m[c][rr] * 2.0f;
    m3 dL_dM = m3_mul(M2, dL_dSigma);
    m3 Rt = m3_transpose(R);
    m3 dL_dMt = m3_transpose(dL_dM);
#define COL(A, c) v3_make((A).m[c][0], (A).m[c][1], (A).m[c][2])
    dL_dscales[3 * idx + 0] = v3_dot(COL(Rt, 0), COL(dL_dMt, 0));
    dL_dscales[3 * idx + 1] = v3_dot(COL(Rt, 1), COL(dL_dMt, 1));
    dL_dscales[3 * idx + 2] = v3_dot(COL(Rt, 2), COL(dL_dMt, 2));
#undef COL
    for (int k = 0; k < 3; k++) { dL_dMt.m[0][k] *= s.x; dL_dMt.m[1][k] *= s.y; dL_dMt.m[2][k] *= s.z; }
#define D(c, rr) dL_dMt.m[c][rr]
    float qx = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
    float qy = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
    float qz = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
    float qw = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
#undef D
    dL_drots[4 * idx + 0] = qx; dL_drots[4 * idx + 1] = qy; dL_drots[4 * idx + 2] = qz; dL_drots[4 * idx + 3] = qw;
}

/* CR/backward.cu:346-396 (preprocessCUDA bwd) */
static void preprocess_backward_one(const orc_inputs *in, const orc_state *st, int idx, const float *dL_dmean2D,
                                    float *dL_dmeans, const float *dL_dcolor, const float *dL_dcov3D, float *dL_dsh,
                                    float *dL_dscale, float *dL_drot)
{
    if (!(st->radii[idx] > 0)) return;
    const float *proj = in->projmatrix;
    v3 m = v3_make(in->means3D[3 * idx], in->means3D[3 * idx + 1], in->means3D[3 * idx + 2]);
    float m_hom_w = transformPoint4x4_w(m, proj);
    float m_w = 1.0f / (m_hom_w + 0.0000001f);
    float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
    float gx = dL_dmean2D[3 * idx], gy = dL_dmean2D[3 * idx + 1];
    float dx = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
    float dy = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
    float dz = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
    dL_dmeans[3 * idx + 0] += dx;
    dL_dmeans[3 * idx + 1] += dy;
    dL_dmeans[3 * idx + 2] += dz;
    if (in->shs) sh_backward_one(in, st, idx, dL_dcolor, dL_dmeans, dL_dsh);
    if (in->scales)
        cov3d_backward_one(idx, in->scales + 3 * idx, in->scale_modifier, in->rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
}

/* CR/rasterizer_impl.cu:340-434 */
void orc_backward(const orc_inputs *in, const orc_state *st, const float *dL_dpix, float *dL_dmean2D, float *dL_dconic,
                  float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                  float *dL_dscale, float *dL_drot, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    const int P = in->P;
    if (P == 0) return;
    const float focal_y = in->H / (2.0f * in->tanfovy);
    const float focal_x = in->W / (2.0f * in->tanfovx);
    const float *color_ptr = in->colors_precomp ? in->colors_precomp : st->rgb;

    double *acc_mean2D = (double *)calloc((size_t)3 * P, 8), *acc_conic = (double *)calloc((size_t)4 * P, 8);
    double *acc_opacity = (double *)calloc((size_t)P, 8), *acc_color = (double *)calloc((size_t)3 * P, 8);
    const int ntiles = st->gridx * st->gridy;
    size_t longest = 1;
    for (int t = 0; t < ntiles; t++) {
        const size_t len = (size_t)(st->ranges[2 * t + 1] - st->ranges[2 * t]);
        if (len > longest) longest = len;
    }
    /* one buffer of per-entry rows per thread, allocated once (a calloc per tile from 256 threads is all page faults) */
    double *rows = (double *)malloc(sizeof(double) * 9 * longest * (size_t)nthreads);
#pragma omp parallel num_threads(nthreads)
    {
        double *loc = rows + (size_t)omp_get_thread_num() * 9 * longest;
#pragma omp for schedule(dynamic, 4)
        for (int t = 0; t < ntiles; t++)
            render_tile_backward(in, st, color_ptr, dL_dpix, t % st->gridx, t / st->gridx, acc_mean2D, acc_conic,
                                 acc_opacity, acc_color, loc);
    }
    free(rows);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t i = 0; i < (int64_t)P; i++) {
        for (int c = 0; c < 3; c++) dL_dmean2D[3 * i + c] += (float)acc_mean2D[3 * i + c];
        for (int c = 0; c < 4; c++) dL_dconic[4 * i + c] += (float)acc_conic[4 * i + c];
        dL_dopacity[i] += (float)acc_opacity[i];
        for (int c = 0; c < 3; c++) dL_dcolor[3 * i + c] += (float)acc_color[3 * i + c];
    }
    free(acc_mean2D); free(acc_conic); free(acc_opacity); free(acc_color);

    const float *cov3D_ptr = in->cov3D_precomp ? in->cov3D_precomp : st->cov3D;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < P; i++)
        cov2d_backward_one(in, st, cov3D_ptr, focal_x, focal_y, dL_dconic, dL_dmean3D, dL_dcov3D, i);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < P; i++)
        preprocess_backward_one(in, st, i, dL_dmean2D, dL_dmean3D, dL_dcolor, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
}
