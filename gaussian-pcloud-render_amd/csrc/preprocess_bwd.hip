// preprocess_bwd.hip -- per-Gaussian backward: dL/d{conic, mean2D, colour} -> dL/d{mean3D, cov3D, SH, scale, rotation}.
//
// One kernel, one thread per Gaussian, covering what the reference does in two launches
//   computeCov2DCUDA   CR/backward.cu:144-274   (conic -> 2D covariance -> 3D covariance, and the mean through the projection Jacobian)
//   preprocessCUDA     CR/backward.cu:346-396   (mean2D -> mean3D, SH backward CR/backward.cu:20-139, scale / rotation backward :278-341)
// and summing over the V views of a batch in registers (V = 1 for the per-view API): every output is written exactly once,
// for every Gaussian (zeros when it is invisible in all views; all M rows of dL_dsh), so the caller clears nothing.
//
// The gradient chain is written from its own derivation (below), not from the reference's expression list; the forward
// quantities it needs (view-space mean with the frustum clamp, the 2x3 projection M = J W, a, b, c of the dilated 2D
// covariance) come from the same code the forward pass runs (splat_math.hpp), so they are bit-identical to what the
// forward saw.  Results are compared to tolerance (float accumulation order already differs in the render backward).
//
//   2D covariance   [a b; b c] = M S M^T + 0.3 I,   M = J W (2x3, rows m0, m1),   S = 3D covariance (symmetric)
//   conic           (A, B, C) = (c, -b, a) / det,   det = a c - b^2
//   with g = dL/d(A, B, C) and D = 1 / (det^2 + 1e-7):
//       dL/da = D (-c^2 gA + 2 b c gB + (det - a c) gC)
//       dL/dc = D (-a^2 gC + 2 a b gB + (det - a c) gA)
//       dL/db = 2 D (b c gA - (det + 2 b^2) gB + a b gC)
//   a = m0.S m0, b = m0.S m1, c = m1.S m1, hence with u = S m0, w = S m1:
//       dL/dS_ii = m0_i^2 da + m0_i m1_i db + m1_i^2 dc
//       dL/dS_ij = 2 m0_i m0_j da + (m0_i m1_j + m0_j m1_i) db + 2 m1_i m1_j dc        (i < j; the 6-vector holds S_ij once)
//       dL/dm0 = 2 da u + db w,      dL/dm1 = 2 dc w + db u
//   m0 = J00 r0 + J02 r2, m1 = J11 r1 + J12 r2 with r_k the rows of the view rotation and
//   J00 = fx / tz, J02 = -fx tx / tz^2, J11 = fy / tz, J12 = -fy ty / tz^2:
//       dL/dJ00 = r0.dm0, dL/dJ02 = r2.dm0, dL/dJ11 = r1.dm1, dL/dJ12 = r2.dm1
//       dL/dtx = -(fx / tz^2) dJ02   (0 where the frustum clamp was active, likewise ty)
//       dL/dty = -(fy / tz^2) dJ12
//       dL/dtz = -(fx dJ00 + fy dJ11) / tz^2 + 2 (fx tx dJ02 + fy ty dJ12) / tz^3
//       dL/dmean = R^T dL/dt
//   S = sum_k a_k a_k^T with a_k = s_k c_k (c_k = column k of the rotation built from the raw quaternion, s = modifier * scale);
//   with G the full symmetric dL/dS (off-diagonals halved):  e_k = 2 G a_k,  dL/dscale_k = c_k . e_k  (as upstream: no factor
//   for the modifier),  dL/dc_k = s_k e_k =: column k of Q, and the quaternion derivative of the rotation entries gives
//       dL/dr = 2 z (Q10 - Q01) + 2 y (Q02 - Q20) + 2 x (Q21 - Q12)
//       dL/dx = 2 y (Q01 + Q10) + 2 z (Q02 + Q20) + 2 r (Q21 - Q12) - 4 x (Q11 + Q22)
//       dL/dy = 2 x (Q01 + Q10) + 2 r (Q02 - Q20) + 2 z (Q12 + Q21) - 4 y (Q00 + Q22)
//       dL/dz = 2 r (Q10 - Q01) + 2 x (Q02 + Q20) + 2 y (Q12 + Q21) - 4 z (Q00 + Q11)      (Q_ij = dL/dR_ij)
#include "common.hpp"
#include "splat_math.hpp"

namespace gsr {

// d normalize(v) / dv applied to dv (reference CR/auxiliary.h:107-117)
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

template <int DEG> constexpr int sh_words() { return 3 * (DEG + 1) * (DEG + 1); }

struct PreBwdArgs {
    int P, D, M, V;
    int stage_sh;                        // dL_dsh rows go through LDS (256 x 3 M floats fit)
    int stage_out;                       // the 3- and 6-float gradient rows go through LDS too
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    const float *means3D, *shs, *scales, *rotations, *cov3D_precomp;
    const float *view, *proj, *campos;   // [V][16], [V][16], [V][3]
    const int* radii;                    // [V][P]
    const uint8_t* clamped;
    const float* grad_rec;               // [P][GRAD_REC_WORDS] per view, see common.hpp
    size_t g_stride, gr_stride;
    uint64_t* counters;                  // per view (CNT_*)
    float *dL_dmean2D, *dL_dopacity, *dL_dcolor;
    float *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
};

// SH backward (reference CR/backward.cu:20-139): adds this view's dL_dsh rows into acc, returns the mean gradient through the
// normalised view direction.
template <int deg>
__device__ __forceinline__ V3 sh_backward(V3 pos, V3 campos, const float* sh, uint32_t cmask, V3 dL_dRGB,
                                          float (&acc)[sh_words<deg>()])
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
#define PUT(k, s) do { const V3 t_ = (s) * dL_dRGB; acc[3 * (k)] += t_.x; acc[3 * (k) + 1] += t_.y; acc[3 * (k) + 2] += t_.z; } while (0)
    PUT(0, kSH_C0);
    if constexpr (deg > 0) {
        PUT(1, -kSH_C1 * y);
        PUT(2, kSH_C1 * z);
        PUT(3, -kSH_C1 * x);
        dRGBdx = -kSH_C1 * SHV(3);
        dRGBdy = -kSH_C1 * SHV(1);
        dRGBdz = kSH_C1 * SHV(2);
        if constexpr (deg > 1) {
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
            if constexpr (deg > 2) {
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

// ---- 3D covariance -> scale, rotation (the sum over views entered gS linearly, so this runs once per Gaussian)
__device__ __forceinline__ void scale_rot_backward(V3 sc, float4 q, float scale_modifier, const float (&gS)[6], float (&dsc)[3], float4& dq)
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    // columns of the rotation (raw quaternion, as the forward builds it)
    const V3 c0 = v3(1.f - 2.f * (y * y + z * z), 2.f * (x * y + r * z), 2.f * (x * z - r * y));
    const V3 c1 = v3(2.f * (x * y - r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + r * x));
    const V3 c2 = v3(2.f * (x * z + r * y), 2.f * (y * z - r * x), 1.f - 2.f * (x * x + y * y));
    const V3 s = v3(scale_modifier * sc.x, scale_modifier * sc.y, scale_modifier * sc.z);
    const V3 a0 = s.x * c0, a1 = s.y * c1, a2 = s.z * c2;
    // e_k = 2 G a_k with G the full symmetric matrix (off-diagonals of the 6-vector halved)
    const float Gxx = gS[0], Gxy = 0.5f * gS[1], Gxz = 0.5f * gS[2], Gyy = gS[3], Gyz = 0.5f * gS[4], Gzz = gS[5];
#define GMUL(v) v3(2.f * (Gxx * (v).x + Gxy * (v).y + Gxz * (v).z), 2.f * (Gxy * (v).x + Gyy * (v).y + Gyz * (v).z), \
                   2.f * (Gxz * (v).x + Gyz * (v).y + Gzz * (v).z))
    const V3 e0 = GMUL(a0), e1 = GMUL(a1), e2 = GMUL(a2);
#undef GMUL
    dsc[0] = dot3(c0, e0);
    dsc[1] = dot3(c1, e1);
    dsc[2] = dot3(c2, e2);
    // Q_ij = dL/dR_ij: column k of Q is s_k e_k
    const V3 Q0 = s.x * e0, Q1 = s.y * e1, Q2 = s.z * e2;
    const float Q00 = Q0.x, Q10 = Q0.y, Q20 = Q0.z, Q01 = Q1.x, Q11 = Q1.y, Q21 = Q1.z, Q02 = Q2.x, Q12 = Q2.y, Q22 = Q2.z;
    dq.x = 2 * z * (Q10 - Q01) + 2 * y * (Q02 - Q20) + 2 * x * (Q21 - Q12);
    dq.y = 2 * y * (Q01 + Q10) + 2 * z * (Q02 + Q20) + 2 * r * (Q21 - Q12) - 4 * x * (Q11 + Q22);
    dq.z = 2 * x * (Q01 + Q10) + 2 * r * (Q02 - Q20) + 2 * z * (Q12 + Q21) - 4 * y * (Q00 + Q22);
    dq.w = 2 * r * (Q10 - Q01) + 2 * x * (Q02 + Q20) + 2 * y * (Q12 + Q21) - 4 * z * (Q00 + Q11);
}

// DEG = active SH degree (a template parameter so that the per-view SH sums take 3 (DEG+1)^2 registers, not 48)
template <int DEG>
__global__ __launch_bounds__(256) void k_preprocess_backward(PreBwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_rows[];   // [256][3 M] staging of the dL_dsh rows (a.stage_sh)
    const bool active = blockIdx.x * 256 + threadIdx.x < (unsigned)a.P;
    const int idx = active ? (int)(blockIdx.x * 256 + threadIdx.x) : a.P - 1;   // idle lanes of the last block shadow a real one

    // (The records hold this backward's sums; k_render_backward raised CNT_BWD_DIRTY so that a further backward over the same
    // forward -- retain_graph, a second gsr_backward call -- clears them first (k_bwd_items), where the reference zero-fills
    // its accumulators on every call, rasterize_points.cu:151-159.)

    // sums over the views of the batch
    float g2x = 0.f, g2y = 0.f, gop = 0.f;
    V3 gcol = v3(0, 0, 0), gmean = v3(0, 0, 0);
    float gS[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int NSH = sh_words<DEG>();
    float gsh[NSH];
#pragma unroll
    for (int i = 0; i < NSH; i++) gsh[i] = 0.f;

    const V3 mean = v3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
    // 3D covariance as the forward saw it (recomputed with the forward's code, not stored)
    float S6[6];
    V3 sc = v3(0, 0, 0);
    float4 q = make_float4(0, 0, 0, 0);
    if (a.cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) S6[i] = a.cov3D_precomp[6 * (size_t)idx + i];
    } else {
        sc = v3(a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]);
        q = *reinterpret_cast<const float4*>(a.rotations + 4 * (size_t)idx);
        cov3d_from_scale_rot(sc, a.scale_modifier, q, S6, nullptr);
    }

    // the SH coefficients the view-direction term needs: read once, reused for every view
    float shv[NSH];
    if (a.shs) {
        const float* shp = a.shs + (size_t)idx * a.M * 3;
#pragma unroll
        for (int i = 0; i < NSH; i++) shv[i] = shp[i];
    }

    for (int vw = 0; vw < a.V; vw++) {
        // the render-level sums of this Gaussian in this view: one 64-B record (zeros when nothing was accumulated)
        const float* recp = at_view(a.grad_rec, a.gr_stride, (uint32_t)vw) + (size_t)idx * GRAD_REC_WORDS;
        const float4 rec0 = *reinterpret_cast<const float4*>(recp);
        const float4 rec1 = *reinterpret_cast<const float4*>(recp + 4);
        const float rec8 = recp[8];
        g2x += rec0.x; g2y += rec0.y;
        gcol = gcol + v3(rec1.y, rec1.z, rec1.w);
        gop += rec8;
        if (!(a.radii[(size_t)vw * a.P + idx] > 0)) continue;   // invisible in this view: no geometric gradient

        const float* view = a.view + 16 * vw;
        const float* proj = a.proj + 16 * vw;
        const float fx = a.focal_x, fy = a.focal_y;
        const Cov2D c = cov2d_project(mean, fx, fy, a.tanfovx, a.tanfovy, S6, view);
        const bool clamp_x = c.txtz < -c.limx || c.txtz > c.limx, clamp_y = c.tytz < -c.limy || c.tytz > c.limy;
        const V3 m0 = v3(c.T.m[0][0], c.T.m[0][1], c.T.m[0][2]), m1 = v3(c.T.m[1][0], c.T.m[1][1], c.T.m[1][2]);
        const float ca = c.cov.m[0][0] + 0.3f, cb = c.cov.m[0][1], cc = c.cov.m[1][1] + 0.3f;

        // ---- conic -> (a, b, c)
        const float gA = rec0.z, gB = rec0.w, gC = rec1.x;
        const float det = ca * cc - cb * cb;
        const float Dn = 1.0f / ((det * det) + 0.0000001f);
        float da = 0.f, db = 0.f, dc = 0.f;
        if (Dn != 0) {
            da = Dn * (-cc * cc * gA + 2 * cb * cc * gB + (det - ca * cc) * gC);
            dc = Dn * (-ca * ca * gC + 2 * ca * cb * gB + (det - ca * cc) * gA);
            db = Dn * 2 * (cb * cc * gA - (det + 2 * cb * cb) * gB + ca * cb * gC);
            // ---- (a, b, c) -> 3D covariance (xx, xy, xz, yy, yz, zz)
            gS[0] += m0.x * m0.x * da + m0.x * m1.x * db + m1.x * m1.x * dc;
            gS[3] += m0.y * m0.y * da + m0.y * m1.y * db + m1.y * m1.y * dc;
            gS[5] += m0.z * m0.z * da + m0.z * m1.z * db + m1.z * m1.z * dc;
            gS[1] += 2 * m0.x * m0.y * da + (m0.x * m1.y + m0.y * m1.x) * db + 2 * m1.x * m1.y * dc;
            gS[2] += 2 * m0.x * m0.z * da + (m0.x * m1.z + m0.z * m1.x) * db + 2 * m1.x * m1.z * dc;
            gS[4] += 2 * m0.z * m0.y * da + (m0.y * m1.z + m0.z * m1.y) * db + 2 * m1.y * m1.z * dc;
        }
        // ---- (a, b, c) -> rows of M -> Jacobian entries -> view-space mean -> mean
        const V3 u = v3(S6[0] * m0.x + S6[1] * m0.y + S6[2] * m0.z, S6[1] * m0.x + S6[3] * m0.y + S6[4] * m0.z,
                        S6[2] * m0.x + S6[4] * m0.y + S6[5] * m0.z);
        const V3 w = v3(S6[0] * m1.x + S6[1] * m1.y + S6[2] * m1.z, S6[1] * m1.x + S6[3] * m1.y + S6[4] * m1.z,
                        S6[2] * m1.x + S6[4] * m1.y + S6[5] * m1.z);
        const V3 dm0 = (2 * da) * u + db * w, dm1 = (2 * dc) * w + db * u;
        const V3 r0 = v3(view[0], view[4], view[8]), r1 = v3(view[1], view[5], view[9]), r2 = v3(view[2], view[6], view[10]);
        const float dJ00 = dot3(r0, dm0), dJ02 = dot3(r2, dm0), dJ11 = dot3(r1, dm1), dJ12 = dot3(r2, dm1);
        const V3 t = c.t;
        const float iz = 1.f / t.z, iz2 = iz * iz, iz3 = iz2 * iz;
        const float dtx = clamp_x ? 0.f : -fx * iz2 * dJ02;
        const float dty = clamp_y ? 0.f : -fy * iz2 * dJ12;
        const float dtz = -(fx * dJ00 + fy * dJ11) * iz2 + 2 * (fx * t.x * dJ02 + fy * t.y * dJ12) * iz3;
        V3 dmean = v3(view[0] * dtx + view[1] * dty + view[2] * dtz, view[4] * dtx + view[5] * dty + view[6] * dtz,
                      view[8] * dtx + view[9] * dty + view[10] * dtz);

        // ---- mean2D -> mean through the perspective divide: ndc = (P row0 . m, P row1 . m) / (P row3 . m + 1e-7)
        {
            const float hx = proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12];
            const float hy = proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13];
            const float hw = ((proj[3] * mean.x + proj[7] * mean.y) + proj[11] * mean.z) + proj[15];
            const float iw = 1.0f / (hw + 0.0000001f);
            const float kx = hx * iw * iw, ky = hy * iw * iw;
            const float gx = rec0.x, gy = rec0.y;
            dmean.x += (proj[0] * iw - proj[3] * kx) * gx + (proj[1] * iw - proj[3] * ky) * gy;
            dmean.y += (proj[4] * iw - proj[7] * kx) * gx + (proj[5] * iw - proj[7] * ky) * gy;
            dmean.z += (proj[8] * iw - proj[11] * kx) * gx + (proj[9] * iw - proj[11] * ky) * gy;
        }

        // ---- colour -> SH (+ view-direction term on the mean)
        if (a.shs) {
            const V3 cam = v3(a.campos[3 * vw], a.campos[3 * vw + 1], a.campos[3 * vw + 2]);
            const uint8_t cm = at_view(a.clamped, a.g_stride, (uint32_t)vw)[idx];
            dmean = dmean + sh_backward<DEG>(mean, cam, shv, cm, v3(rec1.y, rec1.z, rec1.w), gsh);
        }
        gmean = gmean + dmean;
    }

    if (a.shs) {
        // dL_dsh rows (3 M floats per Gaussian, only 3 (DEG+1)^2 of them non-zero): through LDS, so that the workgroup writes
        // its 256 x 3 M floats as one contiguous, 16-B-per-lane stream instead of 3 M stride-12M scalar stores per lane
        const int W3 = 3 * a.M;
        if (a.stage_sh) {
            float* row = s_rows + (size_t)threadIdx.x * W3;
#pragma unroll
            for (int i = 0; i < NSH; i++) row[i] = gsh[i];
            for (int i = NSH; i < W3; i++) row[i] = 0.f;
            __syncthreads();
            const int rows = a.P - (int)blockIdx.x * 256 < 256 ? a.P - (int)blockIdx.x * 256 : 256;
            const int total = rows * W3;
            float* out = a.dL_dsh + (size_t)blockIdx.x * 256 * W3;   // 256 * 3 M floats per block: 16-B aligned
            for (int j = 4 * (int)threadIdx.x; j + 3 < total; j += 4 * 256)
                *reinterpret_cast<float4*>(out + j) = *reinterpret_cast<const float4*>(s_rows + j);
            if ((int)threadIdx.x < (total & 3)) out[(total & ~3) + threadIdx.x] = s_rows[(total & ~3) + threadIdx.x];
        } else if (active) {
            float* out = a.dL_dsh + (size_t)idx * W3;
#pragma unroll
            for (int i = 0; i < NSH; i++) out[i] = gsh[i];
            for (int i = NSH; i < W3; i++) out[i] = 0.f;
        }
    }
    // The 3- and 6-float rows (mean2D, colour, mean3D, 3D covariance, scale) leave through the staging area too: a lane storing
    // its own row issues scalar stores 12 or 24 B apart (12-24 partial-line requests per instruction); staged, the workgroup
    // writes each array's 256 rows as one run of 16-B chunks.
    float dsc[3] = {0.f, 0.f, 0.f};
    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scales) scale_rot_backward(sc, q, a.scale_modifier, gS, dsc, dq);
    if (a.stage_out) {
        __syncthreads();   // the dL_dsh rows have left the staging area
        float* o = s_rows;
        const uint32_t t = threadIdx.x;
        o[3 * t + 0] = g2x; o[3 * t + 1] = g2y; o[3 * t + 2] = 0.f;
        o += 768;
        o[3 * t + 0] = gcol.x; o[3 * t + 1] = gcol.y; o[3 * t + 2] = gcol.z;
        o += 768;
        o[3 * t + 0] = gmean.x; o[3 * t + 1] = gmean.y; o[3 * t + 2] = gmean.z;
        o += 768;
        o[3 * t + 0] = dsc[0]; o[3 * t + 1] = dsc[1]; o[3 * t + 2] = dsc[2];
        o += 768;
#pragma unroll
        for (int i = 0; i < 6; i++) o[6 * t + i] = gS[i];
        __syncthreads();
        const int rows = a.P - (int)blockIdx.x * 256 < 256 ? a.P - (int)blockIdx.x * 256 : 256;
        float* const dst[5] = {a.dL_dmean2D, a.dL_dcolor, a.dL_dmean3D, a.scales ? a.dL_dscale : nullptr, a.dL_dcov3D};
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int wdt = k == 4 ? 6 : 3, total = rows * wdt;
            const float* src = s_rows + 768 * k;
            if (dst[k] == nullptr) continue;
            float* out = dst[k] + (size_t)blockIdx.x * 256 * wdt;   // 256 rows of 12 or 24 B per block: 16-B aligned
            for (int j = 4 * (int)t; j + 3 < total; j += 4 * 256)
                *reinterpret_cast<float4*>(out + j) = *reinterpret_cast<const float4*>(src + j);
            if ((int)t < (total & 3)) out[(total & ~3) + t] = src[(total & ~3) + t];
        }
        if (!active) return;
        a.dL_dopacity[idx] = gop;
        if (a.scales) *reinterpret_cast<float4*>(a.dL_drot + 4 * (size_t)idx) = dq;
        return;
    }
    if (!active) return;
    a.dL_dmean2D[3 * (size_t)idx + 0] = g2x;
    a.dL_dmean2D[3 * (size_t)idx + 1] = g2y;
    a.dL_dmean2D[3 * (size_t)idx + 2] = 0.f;
    a.dL_dcolor[3 * (size_t)idx + 0] = gcol.x;
    a.dL_dcolor[3 * (size_t)idx + 1] = gcol.y;
    a.dL_dcolor[3 * (size_t)idx + 2] = gcol.z;
    a.dL_dopacity[idx] = gop;
    a.dL_dmean3D[3 * (size_t)idx + 0] = gmean.x;
    a.dL_dmean3D[3 * (size_t)idx + 1] = gmean.y;
    a.dL_dmean3D[3 * (size_t)idx + 2] = gmean.z;
#pragma unroll
    for (int i = 0; i < 6; i++) a.dL_dcov3D[6 * (size_t)idx + i] = gS[i];

    if (a.scales) {
        a.dL_dscale[3 * (size_t)idx + 0] = dsc[0];
        a.dL_dscale[3 * (size_t)idx + 1] = dsc[1];
        a.dL_dscale[3 * (size_t)idx + 2] = dsc[2];
        *reinterpret_cast<float4*>(a.dL_drot + 4 * (size_t)idx) = dq;
    }
}

int launch_preprocess_backward(const Launch& L, const gsr_params& p, const Batch& B, const int* radii, float* dL_dmean2D,
                               float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                               float* dL_dscale, float* dL_drot)
{
    PreBwdArgs a;
    a.P = p.P; a.D = p.D; a.M = p.M; a.V = B.V;
    a.tanfovx = p.tanfovx; a.tanfovy = p.tanfovy;
    a.focal_y = p.H / (2.0f * p.tanfovy);
    a.focal_x = p.W / (2.0f * p.tanfovx);
    a.scale_modifier = p.scale_modifier;
    a.means3D = p.means3D; a.shs = p.shs; a.scales = p.scales; a.rotations = p.rotations;
    a.cov3D_precomp = p.cov3D_precomp; a.view = p.viewmatrix; a.proj = p.projmatrix; a.campos = p.campos;
    a.radii = radii; a.clamped = B.g.clamped;
    a.grad_rec = B.grad_rec; a.g_stride = B.g_stride; a.gr_stride = B.gr_stride; a.counters = B.g.counters;
    a.dL_dmean2D = dL_dmean2D; a.dL_dopacity = dL_dopacity; a.dL_dcolor = dL_dcolor;
    a.dL_dmean3D = dL_dmean3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscale = dL_dscale; a.dL_drot = dL_drot;
    const dim3 grid((p.P + 255) / 256), block(256);
    size_t lds = p.shs ? (size_t)256 * 3 * p.M * sizeof(float) : 0;
    // the staged paths write 16-B chunks from each array's base (+ a multiple of 16 B per workgroup): they need 16-B aligned
    // output pointers (include/gsr.h asks for that; every torch / hipMalloc allocation is).  A caller that hands over a
    // float-aligned sub-buffer gets the per-row scalar stores instead.
    const auto al16 = [](const void* q) { return ((uintptr_t)q & 15u) == 0; };
    a.stage_sh = lds != 0 && lds <= 64 * 1024 && al16(dL_dsh);
    if (!a.stage_sh) lds = 0;
    const size_t lds_out = (size_t)(4 * 768 + 6 * 256) * sizeof(float);   // the staged 3- and 6-float rows of a workgroup
    a.stage_out = al16(dL_dmean2D) && al16(dL_dcolor) && al16(dL_dmean3D) && al16(dL_dcov3D) && (!p.scales || al16(dL_dscale));
    if (a.stage_out && lds < lds_out) lds = lds_out;
    switch (p.shs ? p.D : 0) {
    case 0: hipLaunchKernelGGL(k_preprocess_backward<0>, grid, block, lds, L.stream, a); break;
    case 1: hipLaunchKernelGGL(k_preprocess_backward<1>, grid, block, lds, L.stream, a); break;
    case 2: hipLaunchKernelGGL(k_preprocess_backward<2>, grid, block, lds, L.stream, a); break;
    default: hipLaunchKernelGGL(k_preprocess_backward<3>, grid, block, lds, L.stream, a); break;
    }
    return check_launch(L, "preprocess_backward");
}

}  // namespace gsr
