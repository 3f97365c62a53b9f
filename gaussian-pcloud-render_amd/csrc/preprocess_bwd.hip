// preprocess_bwd.hip -- per-Gaussian backward: dL/d{conic, mean2D, colour} -> dL/d{mean3D, cov3D, SH,
// scale, rotation}.
//
// One kernel, one thread per visible Gaussian, fusing the reference's two launches
//   computeCov2DCUDA   CR/backward.cu:144-274   (conic -> cov2D -> cov3D and the mean through the Jacobian)
//   preprocessCUDA     CR/backward.cu:346-396   (mean2D -> mean3D, SH backward CR/backward.cu:20-139,
//                                                scale/rotation backward CR/backward.cu:278-341)
// so dL_dmean3D is accumulated in registers in the reference's order (cov2D part, + projection part, + SH
// view-direction part) and written once, and dL_dcov3D feeds the scale/rotation step without a round trip
// through HBM (it is still stored: it is the gradient returned for cov3D_precomp).  The 3D covariance is
// recomputed from scale/rotation with the forward's code instead of being saved by the forward pass.
#include "common.hpp"
#include "splat_math.hpp"

namespace gsr {

// reference CR/auxiliary.h:107-117
__device__ __forceinline__ V3 dnormvdv(V3 v, V3 dv)
{
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    V3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

struct PreBwdArgs {
    int P, D, M;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    const float *means3D, *shs, *scales, *rotations, *cov3D_precomp, *view, *proj, *campos;
    const int* radii;
    const uint8_t* clamped;
    const float* grad_rec;                         // [P][GRAD_REC_WORDS], see common.hpp
    float *dL_dmean2D, *dL_dopacity, *dL_dcolor;   // user-facing copies of record fields
    float *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
};

// SH backward (reference CR/backward.cu:20-139): writes dL_dsh rows, returns the mean gradient through the
// normalised view direction.
__device__ __forceinline__ V3 sh_backward(int deg, V3 pos, V3 campos, const float* __restrict__ sh, uint32_t cmask,
                                          V3 dL_dRGB, float* __restrict__ dL_dsh)
{
    const V3 dir_orig = pos - campos;
    const float len = sqrtf(dot3(dir_orig, dir_orig));
    const V3 dir = v3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
    dL_dRGB.x *= (cmask & 1u) ? 0 : 1;
    dL_dRGB.y *= (cmask & 2u) ? 0 : 1;
    dL_dRGB.z *= (cmask & 4u) ? 0 : 1;
    V3 dRGBdx = v3(0, 0, 0), dRGBdy = v3(0, 0, 0), dRGBdz = v3(0, 0, 0);
    const float x = dir.x, y = dir.y, z = dir.z;
#define SHV(k) v3(sh[3 * (k)], sh[3 * (k) + 1], sh[3 * (k) + 2])
#define PUT(k, s) do { const V3 t_ = (s) * dL_dRGB; dL_dsh[3 * (k)] = t_.x; dL_dsh[3 * (k) + 1] = t_.y; dL_dsh[3 * (k) + 2] = t_.z; } while (0)
    PUT(0, kSH_C0);
    if (deg > 0) {
        PUT(1, -kSH_C1 * y);
        PUT(2, kSH_C1 * z);
        PUT(3, -kSH_C1 * x);
        dRGBdx = -kSH_C1 * SHV(3);
        dRGBdy = -kSH_C1 * SHV(1);
        dRGBdz = kSH_C1 * SHV(2);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            PUT(4, kSH_C2[0] * xy);
            PUT(5, kSH_C2[1] * yz);
            PUT(6, kSH_C2[2] * (2.f * zz - xx - yy));
            PUT(7, kSH_C2[3] * xz);
            PUT(8, kSH_C2[4] * (xx - yy));
            dRGBdx = dRGBdx + ((((kSH_C2[0] * y) * SHV(4) + (kSH_C2[2] * 2.f * -x) * SHV(6)) + (kSH_C2[3] * z) * SHV(7)) +
                               (kSH_C2[4] * 2.f * x) * SHV(8));
            dRGBdy = dRGBdy + ((((kSH_C2[0] * x) * SHV(4) + (kSH_C2[1] * z) * SHV(5)) + (kSH_C2[2] * 2.f * -y) * SHV(6)) +
                               (kSH_C2[4] * 2.f * -y) * SHV(8));
            dRGBdz = dRGBdz + (((kSH_C2[1] * y) * SHV(5) + (kSH_C2[2] * 2.f * 2.f * z) * SHV(6)) + (kSH_C2[3] * x) * SHV(7));
            if (deg > 2) {
                PUT(9, kSH_C3[0] * y * (3.f * xx - yy));
                PUT(10, kSH_C3[1] * xy * z);
                PUT(11, kSH_C3[2] * y * (4.f * zz - xx - yy));
                PUT(12, kSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                PUT(13, kSH_C3[4] * x * (4.f * zz - xx - yy));
                PUT(14, kSH_C3[5] * z * (xx - yy));
                PUT(15, kSH_C3[6] * x * (xx - 3.f * yy));
                // scalar*vec first, then the remaining scalars left to right (reference CR/backward.cu:99-122)
                V3 s;
                s = (((kSH_C3[0] * SHV(9)) * 3.f) * 2.f) * xy;
                s = s + (kSH_C3[1] * SHV(10)) * yz;
                s = s + ((kSH_C3[2] * SHV(11)) * -2.f) * xy;
                s = s + (((kSH_C3[3] * SHV(12)) * -3.f) * 2.f) * xz;
                s = s + (kSH_C3[4] * SHV(13)) * (-3.f * xx + 4.f * zz - yy);
                s = s + ((kSH_C3[5] * SHV(14)) * 2.f) * xz;
                s = s + ((kSH_C3[6] * SHV(15)) * 3.f) * (xx - yy);
                dRGBdx = dRGBdx + s;
                s = ((kSH_C3[0] * SHV(9)) * 3.f) * (xx - yy);
                s = s + (kSH_C3[1] * SHV(10)) * xz;
                s = s + (kSH_C3[2] * SHV(11)) * (-3.f * yy + 4.f * zz - xx);
                s = s + (((kSH_C3[3] * SHV(12)) * -3.f) * 2.f) * yz;
                s = s + ((kSH_C3[4] * SHV(13)) * -2.f) * xy;
                s = s + ((kSH_C3[5] * SHV(14)) * -2.f) * yz;
                s = s + (((kSH_C3[6] * SHV(15)) * -3.f) * 2.f) * xy;
                dRGBdy = dRGBdy + s;
                s = (kSH_C3[1] * SHV(10)) * xy;
                s = s + (((kSH_C3[2] * SHV(11)) * 4.f) * 2.f) * yz;
                s = s + ((kSH_C3[3] * SHV(12)) * 3.f) * (2.f * zz - xx - yy);
                s = s + (((kSH_C3[4] * SHV(13)) * 4.f) * 2.f) * xz;
                s = s + (kSH_C3[5] * SHV(14)) * (xx - yy);
                dRGBdz = dRGBdz + s;
            }
        }
    }
#undef PUT
#undef SHV
    const V3 dL_ddir = v3(dot3(dRGBdx, dL_dRGB), dot3(dRGBdy, dL_dRGB), dot3(dRGBdz, dL_dRGB));
    return dnormvdv(dir_orig, dL_ddir);
}

__global__ __launch_bounds__(256) void k_preprocess_backward(PreBwdArgs a)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.P) return;
    // the render-level sums of this Gaussian: one 64-B record (zeros when nothing was accumulated, e.g. invisible)
    const float4 rec0 = *reinterpret_cast<const float4*>(a.grad_rec + (size_t)idx * GRAD_REC_WORDS);
    const float4 rec1 = *reinterpret_cast<const float4*>(a.grad_rec + (size_t)idx * GRAD_REC_WORDS + 4);
    const float rec8 = a.grad_rec[(size_t)idx * GRAD_REC_WORDS + 8];
    a.dL_dmean2D[3 * (size_t)idx + 0] = rec0.x;
    a.dL_dmean2D[3 * (size_t)idx + 1] = rec0.y;
    a.dL_dmean2D[3 * (size_t)idx + 2] = 0.f;
    a.dL_dcolor[3 * (size_t)idx + 0] = rec1.y;
    a.dL_dcolor[3 * (size_t)idx + 1] = rec1.z;
    a.dL_dcolor[3 * (size_t)idx + 2] = rec1.w;
    a.dL_dopacity[idx] = rec8;
    if (!(a.radii[idx] > 0)) {
        // invisible Gaussian: no gradient.  The per-Gaussian outputs this kernel owns are written for every index, so the
        // caller does not have to clear them first (the atomically accumulated ones and dL_dsh's unused rows it does).
#pragma unroll
        for (int i = 0; i < 3; i++) a.dL_dmean3D[3 * (size_t)idx + i] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = 0.f;
        if (a.scales) {
#pragma unroll
            for (int i = 0; i < 3; i++) a.dL_dscale[3 * (size_t)idx + i] = 0.f;
            *reinterpret_cast<float4*>(a.dL_drot + 4 * (size_t)idx) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        view[i] = a.view[i];
        proj[i] = a.proj[i];
    }
    const V3 mean = v3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);

    // ---- 3D covariance as the forward saw it
    float cov6[6];
    V3 sc = v3(0, 0, 0);
    float4 q = make_float4(0, 0, 0, 0);
    if (a.cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) cov6[i] = a.cov3D_precomp[6 * (size_t)idx + i];
    } else {
        sc = v3(a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]);
        q = *reinterpret_cast<const float4*>(a.rotations + 4 * (size_t)idx);
        cov3d_from_scale_rot(sc, a.scale_modifier, q, cov6, nullptr);
    }

    // ---- conic -> cov2D -> cov3D, and the mean through J (reference CR/backward.cu:144-274)
    const float gcx = rec0.z, gcy = rec0.w, gcz = rec1.x;
    const float h_x = a.focal_x, h_y = a.focal_y;
    const Cov2D c = cov2d_project(mean, h_x, h_y, a.tanfovx, a.tanfovy, cov6, view);
    const V3 t = c.t;
    const float x_grad_mul = (c.txtz < -c.limx || c.txtz > c.limx) ? 0 : 1;
    const float y_grad_mul = (c.tytz < -c.limy || c.tytz > c.limy) ? 0 : 1;
    const M3& T = c.T;
    const M3& W = c.W;
    const M3& Vrk = c.Vrk;

    const float ca = c.cov.m[0][0] + 0.3f;
    const float cb = c.cov.m[0][1];
    const float cc = c.cov.m[1][1] + 0.3f;
    const float denom = ca * cc - cb * cb;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcov[6];
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * gcx + 2 * cb * cc * gcy + (denom - ca * cc) * gcz);
        dL_dc = denom2inv * (-ca * ca * gcz + 2 * ca * cb * gcy + (denom - ca * cc) * gcx);
        dL_db = denom2inv * 2 * (cb * cc * gcx - (denom + 2 * cb * cb) * gcy + ca * cb * gcz);
        dcov[0] = (T.m[0][0] * T.m[0][0] * dL_da + T.m[0][0] * T.m[1][0] * dL_db + T.m[1][0] * T.m[1][0] * dL_dc);
        dcov[3] = (T.m[0][1] * T.m[0][1] * dL_da + T.m[0][1] * T.m[1][1] * dL_db + T.m[1][1] * T.m[1][1] * dL_dc);
        dcov[5] = (T.m[0][2] * T.m[0][2] * dL_da + T.m[0][2] * T.m[1][2] * dL_db + T.m[1][2] * T.m[1][2] * dL_dc);
        dcov[1] = 2 * T.m[0][0] * T.m[0][1] * dL_da + (T.m[0][0] * T.m[1][1] + T.m[0][1] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][1] * dL_dc;
        dcov[2] = 2 * T.m[0][0] * T.m[0][2] * dL_da + (T.m[0][0] * T.m[1][2] + T.m[0][2] * T.m[1][0]) * dL_db + 2 * T.m[1][0] * T.m[1][2] * dL_dc;
        dcov[4] = 2 * T.m[0][2] * T.m[0][1] * dL_da + (T.m[0][1] * T.m[1][2] + T.m[0][2] * T.m[1][1]) * dL_db + 2 * T.m[1][1] * T.m[1][2] * dL_dc;
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];

    const float dL_dT00 = 2 * (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_da +
                          (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_db;
    const float dL_dT01 = 2 * (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_da +
                          (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_db;
    const float dL_dT02 = 2 * (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_da +
                          (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_db;
    const float dL_dT10 = 2 * (T.m[1][0] * Vrk.m[0][0] + T.m[1][1] * Vrk.m[0][1] + T.m[1][2] * Vrk.m[0][2]) * dL_dc +
                          (T.m[0][0] * Vrk.m[0][0] + T.m[0][1] * Vrk.m[0][1] + T.m[0][2] * Vrk.m[0][2]) * dL_db;
    const float dL_dT11 = 2 * (T.m[1][0] * Vrk.m[1][0] + T.m[1][1] * Vrk.m[1][1] + T.m[1][2] * Vrk.m[1][2]) * dL_dc +
                          (T.m[0][0] * Vrk.m[1][0] + T.m[0][1] * Vrk.m[1][1] + T.m[0][2] * Vrk.m[1][2]) * dL_db;
    const float dL_dT12 = 2 * (T.m[1][0] * Vrk.m[2][0] + T.m[1][1] * Vrk.m[2][1] + T.m[1][2] * Vrk.m[2][2]) * dL_dc +
                          (T.m[0][0] * Vrk.m[2][0] + T.m[0][1] * Vrk.m[2][1] + T.m[0][2] * Vrk.m[2][2]) * dL_db;

    const float dL_dJ00 = W.m[0][0] * dL_dT00 + W.m[0][1] * dL_dT01 + W.m[0][2] * dL_dT02;
    const float dL_dJ02 = W.m[2][0] * dL_dT00 + W.m[2][1] * dL_dT01 + W.m[2][2] * dL_dT02;
    const float dL_dJ11 = W.m[1][0] * dL_dT10 + W.m[1][1] * dL_dT11 + W.m[1][2] * dL_dT12;
    const float dL_dJ12 = W.m[2][0] * dL_dT10 + W.m[2][1] * dL_dT11 + W.m[2][2] * dL_dT12;

    const float tz = 1.f / t.z;
    const float tz2 = tz * tz;
    const float tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;

    // transformVec4x3Transpose (reference CR/auxiliary.h:89-97); plain assignment in the reference (:273)
    V3 dmean = v3(view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz,
                  view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz,
                  view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz);

    // ---- mean2D -> mean3D through the perspective divide (reference CR/backward.cu:370-391)
    {
        const float m_hom_w = ((proj[3] * mean.x + proj[7] * mean.y) + proj[11] * mean.z) + proj[15];
        const float m_w = 1.0f / (m_hom_w + 0.0000001f);
        const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        const float gx = rec0.x, gy = rec0.y;
        V3 d;
        d.x = (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
        d.y = (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
        d.z = (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
        dmean = dmean + d;
    }

    // ---- colour -> SH (+ view-direction term on the mean)
    if (a.shs) {
        const V3 cam = v3(a.campos[0], a.campos[1], a.campos[2]);
        const V3 dL_dRGB = v3(rec1.y, rec1.z, rec1.w);
        const V3 d = sh_backward(a.D, mean, cam, a.shs + (size_t)idx * a.M * 3, a.clamped[idx], dL_dRGB,
                                 a.dL_dsh + (size_t)idx * a.M * 3);
        dmean = dmean + d;
    }
    a.dL_dmean3D[3 * (size_t)idx + 0] = dmean.x;
    a.dL_dmean3D[3 * (size_t)idx + 1] = dmean.y;
    a.dL_dmean3D[3 * (size_t)idx + 2] = dmean.z;

    // ---- cov3D -> scale, rotation (reference CR/backward.cu:278-341)
    if (a.scales) {
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        const M3 R = quat_to_cols(r, x, y, z);
        M3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
        const V3 s = v3(a.scale_modifier * sc.x, a.scale_modifier * sc.y, a.scale_modifier * sc.z);
        S.m[0][0] = s.x; S.m[1][1] = s.y; S.m[2][2] = s.z;
        const M3 M = m3_mul(S, R);
        const M3 dL_dSigma = m3_cols(dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                                     0.5f * dcov[2], 0.5f * dcov[4], dcov[5]);
        M3 M2;
#pragma unroll
        for (int cI = 0; cI < 3; cI++)
#pragma unroll
            for (int rI = 0; rI < 3; rI++) M2.m[cI][rI] = M.m[cI][rI] * 2.0f;
        const M3 dL_dM = m3_mul(M2, dL_dSigma);
        const M3 Rt = m3_transpose(R);
        M3 dL_dMt = m3_transpose(dL_dM);
#define COLV(A, k) v3((A).m[k][0], (A).m[k][1], (A).m[k][2])
        a.dL_dscale[3 * (size_t)idx + 0] = dot3(COLV(Rt, 0), COLV(dL_dMt, 0));
        a.dL_dscale[3 * (size_t)idx + 1] = dot3(COLV(Rt, 1), COLV(dL_dMt, 1));
        a.dL_dscale[3 * (size_t)idx + 2] = dot3(COLV(Rt, 2), COLV(dL_dMt, 2));
#undef COLV
#pragma unroll
        for (int k = 0; k < 3; k++) {
            dL_dMt.m[0][k] *= s.x;
            dL_dMt.m[1][k] *= s.y;
            dL_dMt.m[2][k] *= s.z;
        }
#define DM(cI, rI) dL_dMt.m[cI][rI]
        float4 dq;
        dq.x = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
        dq.y = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
        dq.z = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
        dq.w = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
        *reinterpret_cast<float4*>(a.dL_drot + 4 * (size_t)idx) = dq;
    }
}

int launch_preprocess_backward(const Launch& L, const gsr_params& p, const GeomView& g, const int* radii,
                               const float* grad_rec, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                               float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    PreBwdArgs a;
    a.P = p.P; a.D = p.D; a.M = p.M;
    a.tanfovx = p.tanfovx; a.tanfovy = p.tanfovy;
    a.focal_y = p.H / (2.0f * p.tanfovy);
    a.focal_x = p.W / (2.0f * p.tanfovx);
    a.scale_modifier = p.scale_modifier;
    a.means3D = p.means3D; a.shs = p.shs; a.scales = p.scales; a.rotations = p.rotations;
    a.cov3D_precomp = p.cov3D_precomp; a.view = p.viewmatrix; a.proj = p.projmatrix; a.campos = p.campos;
    a.radii = radii; a.clamped = g.clamped;
    a.grad_rec = grad_rec; a.dL_dmean2D = dL_dmean2D; a.dL_dopacity = dL_dopacity; a.dL_dcolor = dL_dcolor;
    a.dL_dmean3D = dL_dmean3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscale = dL_dscale; a.dL_drot = dL_drot;
    hipLaunchKernelGGL(k_preprocess_backward, dim3((p.P + 255) / 256), dim3(256), 0, L.stream, a);
    return check_launch(L, "preprocess_backward");
}

}  // namespace gsr
