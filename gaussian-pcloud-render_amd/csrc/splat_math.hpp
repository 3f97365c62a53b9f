// splat_math.hpp -- per-Gaussian fp32 math shared by preprocess.hip and preprocess_bwd.hip.
//
// The rounding ORDER of every expression follows the reference source so integer outputs are
// reproducible (see common.hpp).  3x3 matrices are stored column-major, m[c][r], and multiplied as
//   (A*B)[c][r] = (A[0][r]*B[c][0] + A[1][r]*B[c][1]) + A[2][r]*B[c][2]
// which is the evaluation order of the GLM operator the reference relies on
// (third_party/glm/glm/detail/type_mat3x3.inl:486-520).
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

struct V3 { float x, y, z; };
struct M3 { float m[3][3]; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

__device__ __forceinline__ float fmin_(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float fmax_(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ M3 m3_cols(float a, float b, float c, float d, float e, float f, float g, float h, float i)
{
    M3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}
__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B)
{
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            R.m[c][r] = (A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1]) + A.m[2][r] * B.m[c][2];
    return R;
}
__device__ __forceinline__ M3 m3_transpose(const M3& A)
{
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++) R.m[c][r] = A.m[r][c];
    return R;
}

// view/proj are column-major flattened 4x4 (reference CR/auxiliary.h:58-76)
__device__ __forceinline__ V3 xform_point_4x3(V3 p, const float* __restrict__ m)
{
    return v3(((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12],
              ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13],
              ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14]);
}

// Rotation from the raw (un-normalised) quaternion (r,x,y,z), written as the reference writes it
// (CR/forward.cu:136-140); columns of the returned M3 are the argument triples.
__device__ __forceinline__ M3 quat_to_cols(float r, float x, float y, float z)
{
    return m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

// 3D covariance from scale + quaternion (reference CR/forward.cu:121-155): Sigma = (S R)^T (S R).
__device__ __forceinline__ void cov3d_from_scale_rot(V3 scale, float mod, float4 q, float* cov6, M3* M_out)
{
    M3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale.x;
    S.m[1][1] = mod * scale.y;
    S.m[2][2] = mod * scale.z;
    M3 R = quat_to_cols(q.x, q.y, q.z, q.w);
    M3 M = m3_mul(S, R);
    M3 Sigma = m3_mul(m3_transpose(M), M);
    cov6[0] = Sigma.m[0][0];
    cov6[1] = Sigma.m[0][1];
    cov6[2] = Sigma.m[0][2];
    cov6[3] = Sigma.m[1][1];
    cov6[4] = Sigma.m[1][2];
    cov6[5] = Sigma.m[2][2];
    if (M_out) *M_out = M;
}

// EWA projection context shared by forward (CR/forward.cu:74-116) and backward (CR/backward.cu:160-198).
struct Cov2D {
    V3 t;              // view-space mean with x,y clamped to the 1.3*tanfov frustum
    float txtz, tytz;  // unclamped ratios (for the backward clamp masks)
    float limx, limy;
    M3 W, T, Vrk, cov; // cov = T^T Vrk^T T, before the +0.3 dilation
};

__device__ __forceinline__ Cov2D cov2d_project(V3 mean, float fx, float fy, float tanx, float tany, const float* cov6,
                                               const float* __restrict__ view)
{
    Cov2D c;
    V3 t = xform_point_4x3(mean, view);
    c.limx = 1.3f * tanx;
    c.limy = 1.3f * tany;
    c.txtz = t.x / t.z;
    c.tytz = t.y / t.z;
    t.x = fmin_(c.limx, fmax_(-c.limx, c.txtz)) * t.z;
    t.y = fmin_(c.limy, fmax_(-c.limy, c.tytz)) * t.z;
    c.t = t;
    M3 J = m3_cols(fx / t.z, 0.0f, -(fx * t.x) / (t.z * t.z),
                   0.0f, fy / t.z, -(fy * t.y) / (t.z * t.z),
                   0.0f, 0.0f, 0.0f);
    c.W = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    c.T = m3_mul(c.W, J);
    c.Vrk = m3_cols(cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]);
    c.cov = m3_mul(m3_mul(m3_transpose(c.T), m3_transpose(c.Vrk)), c.T);
    return c;
}

// SH constants (reference CR/auxiliary.h:22-39)
__device__ constexpr float kSH_C0 = 0.28209479177387814f;
__device__ constexpr float kSH_C1 = 0.4886025119029199f;
__device__ constexpr float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                        -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                        0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                        -0.5900435899266435f};

}  // namespace gsr
