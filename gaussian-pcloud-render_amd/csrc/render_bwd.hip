// render_bwd.hip -- backward of the per-tile compositing: dL/dpixel -> per-Gaussian dL/d{mean2D, conic,
// opacity, colour}.
//
// Per-(pixel, Gaussian) arithmetic follows reference CR/backward.cu:399-557 (renderCUDA): the tile list is
// walked back to front, entries at or beyond the pixel's n_contrib are skipped, T is rebuilt by division,
// the same power/alpha skips apply, and nine partial derivatives come out of every contributing pair.
//
// Where the reference issues 9 float atomicAdds per contributing PAIR (256 pixels hammering the same
// Gaussian), this kernel reduces first:
//   lanes -> wave   : 6-step DPP butterfly per value (quad_perm, row mirrors, row_bcast15/31)
//   waves -> tile   : LDS float adds into a per-round accumulator row per list entry
//   tile  -> global : one hardware float atomic per (entry, component) per tile, zero sums skipped
// so global atomics drop from 9*256 to at most 9 per (tile, entry).  The walk also starts at the tile's
// largest n_contrib instead of the list end: entries no pixel of the tile consumed are never staged.
// Summation order differs from the reference's (undefined) atomic order; results agree to fp32 rounding.
#include "common.hpp"

namespace gsr {

constexpr int BRB = 256;   // entries staged per round
constexpr int NACC = 9;    // mean2D.x,y  conic.x,y,w  opacity  colour r,g,b

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

// sum over the 64 lanes; the total is valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v += dpp_get<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_get<0x140, 0xf>(v);  // row_mirror
    v += dpp_get<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_get<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

struct RenderBwdArgs {
    const uint2* ranges;
    const uint32_t* tile_order;
    const uint32_t* point_list;
    const Splat* splat;
    int W, H, gridx;
    const float* bg;
    const float* final_T;
    const uint32_t* n_contrib;
    const float* dL_dpix;
    float* dL_dmean2D;   // [P,3]
    float* dL_dconic;    // [P,4]
    float* dL_dopacity;  // [P]
    float* dL_dcolor;    // [P,3]
};

__global__ __launch_bounds__(256) void k_render_backward(RenderBwdArgs a)
{
    __shared__ float4 s_q0[BRB];   // x, y, conic.x, conic.y
    __shared__ float2 s_q1[BRB];   // conic.z, opacity
    __shared__ float4 s_col[BRB];  // r, g, b, -
    __shared__ uint32_t s_id[BRB];
    __shared__ float s_acc[BRB][NACC];
    __shared__ uint32_t s_max[4];

    const uint32_t tile = a.tile_order[blockIdx.x];
    const uint32_t tx = tile % (uint32_t)a.gridx, ty = tile / (uint32_t)a.gridx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t px = tx * TILE_X + (w & 1) * 8 + (lane & 7);
    const uint32_t py = ty * TILE_Y + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const size_t pix = (size_t)py * a.W + px, N = (size_t)a.W * a.H;

    const uint2 range = a.ranges[tile];

    const float T_final = inside ? a.final_T[pix] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? a.n_contrib[pix] : 0u;
    float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
    if (inside) {
        dpx0 = a.dL_dpix[pix];
        dpx1 = a.dL_dpix[N + pix];
        dpx2 = a.dL_dpix[2 * N + pix];
    }
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
    float bg_dot_dpixel = 0;
    bg_dot_dpixel += bg0 * dpx0;
    bg_dot_dpixel += bg1 * dpx1;
    bg_dot_dpixel += bg2 * dpx2;

    // tile-wide max of n_contrib: nothing beyond it was consumed by any pixel
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(m, d, 64);
            m = m > o ? m : o;
        }
        if (lane == 0) s_max[w] = m;
    }
    __syncthreads();
    uint32_t max_nc = s_max[0];
    max_nc = max_nc > s_max[1] ? max_nc : s_max[1];
    max_nc = max_nc > s_max[2] ? max_nc : s_max[2];
    max_nc = max_nc > s_max[3] ? max_nc : s_max[3];
    const int total = (int)max_nc;  // list entries [0,total) are walked, last first

    float acc_r0 = 0.f, acc_r1 = 0.f, acc_r2 = 0.f;      // accum_rec
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;  // last_color
    const float ddelx_dx = (float)(0.5 * a.W);
    const float ddely_dy = (float)(0.5 * a.H);

    for (int base = 0; base < total; base += BRB) {
        const int n = total - base < BRB ? total - base : BRB;
        __syncthreads();  // previous round's flush is finished before LDS is reused
        if ((int)tid < n) {
            const uint32_t id = a.point_list[range.x + (uint32_t)(total - 1 - base - (int)tid)];
            const Splat* sp = a.splat + id;
            const float4 q0 = sp->q0, q1 = sp->q1, q2 = sp->q2;
            s_id[tid] = id;
            s_q0[tid] = q0;
            s_q1[tid] = make_float2(q1.x, q1.y);
            s_col[tid] = make_float4(q1.z, q1.w, q2.x, 0.f);
        }
        for (int i = tid; i < BRB * NACC; i += 256) (&s_acc[0][0])[i] = 0.f;
        __syncthreads();

        for (int j = 0; j < n; j++) {
            const uint32_t f = (uint32_t)(total - 1 - base - j);  // 0-based index from the list front
            const float4 q0 = s_q0[j];
            const float2 q1 = s_q1[j];
            const float dx = q0.x - pixf_x, dy = q0.y - pixf_y;
            const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
            const float G = expf(power);
            const float alpha = fminf(0.99f, q1.y * G);
            const bool hit = (f < last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
            if (!__any(hit)) continue;

            float g[NACC];
#pragma unroll
            for (int k = 0; k < NACC; k++) g[k] = 0.f;
            if (hit) {
                const float4 col = s_col[j];
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                acc_r0 = last_alpha * lc0 + (1.f - last_alpha) * acc_r0;
                lc0 = col.x;
                dL_dalpha += (col.x - acc_r0) * dpx0;
                acc_r1 = last_alpha * lc1 + (1.f - last_alpha) * acc_r1;
                lc1 = col.y;
                dL_dalpha += (col.y - acc_r1) * dpx1;
                acc_r2 = last_alpha * lc2 + (1.f - last_alpha) * acc_r2;
                lc2 = col.z;
                dL_dalpha += (col.z - acc_r2) * dpx2;
                g[6] = dchannel_dcolor * dpx0;
                g[7] = dchannel_dcolor * dpx1;
                g[8] = dchannel_dcolor * dpx2;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                const float dL_dG = q1.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * q0.z - gdy * q0.w;
                const float dG_ddely = -gdy * q1.x - gdx * q0.w;
                g[0] = dL_dG * dG_ddelx * ddelx_dx;
                g[1] = dL_dG * dG_ddely * ddely_dy;
                g[2] = -0.5f * gdx * dx * dL_dG;
                g[3] = -0.5f * gdx * dy * dL_dG;
                g[4] = -0.5f * gdy * dy * dL_dG;
                g[5] = G * dL_dalpha;
            }
#pragma unroll
            for (int k = 0; k < NACC; k++) g[k] = wave_sum_to_lane63(g[k]);
            if (lane == 63) {
#pragma unroll
                for (int k = 0; k < NACC; k++) atomicAdd(&s_acc[j][k], g[k]);
            }
        }
        __syncthreads();

        if ((int)tid < n) {
            const uint32_t id = s_id[tid];
            const float* r = s_acc[tid];
            if (r[0] != 0.f) atomicAdd(a.dL_dmean2D + 3 * (size_t)id + 0, r[0]);
            if (r[1] != 0.f) atomicAdd(a.dL_dmean2D + 3 * (size_t)id + 1, r[1]);
            if (r[2] != 0.f) atomicAdd(a.dL_dconic + 4 * (size_t)id + 0, r[2]);
            if (r[3] != 0.f) atomicAdd(a.dL_dconic + 4 * (size_t)id + 1, r[3]);
            if (r[4] != 0.f) atomicAdd(a.dL_dconic + 4 * (size_t)id + 3, r[4]);
            if (r[5] != 0.f) atomicAdd(a.dL_dopacity + (size_t)id, r[5]);
            if (r[6] != 0.f) atomicAdd(a.dL_dcolor + 3 * (size_t)id + 0, r[6]);
            if (r[7] != 0.f) atomicAdd(a.dL_dcolor + 3 * (size_t)id + 1, r[7]);
            if (r[8] != 0.f) atomicAdd(a.dL_dcolor + 3 * (size_t)id + 2, r[8]);
        }
    }
}

int launch_render_backward(const Launch& L, const gsr_params& p, const GeomView& g, const uint32_t* point_list,
                           const ImageView& iv, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                           float* dL_dopacity, float* dL_dcolor)
{
    RenderBwdArgs a;
    a.ranges = iv.ranges;
    a.tile_order = iv.tile_order;
    a.point_list = point_list;
    a.splat = g.splat;
    a.W = p.W; a.H = p.H;
    a.gridx = (p.W + TILE_X - 1) / TILE_X;
    const int gridy = (p.H + TILE_Y - 1) / TILE_Y;
    a.bg = p.bg;
    a.final_T = iv.final_T;
    a.n_contrib = iv.n_contrib;
    a.dL_dpix = dL_dpix;
    a.dL_dmean2D = dL_dmean2D;
    a.dL_dconic = dL_dconic;
    a.dL_dopacity = dL_dopacity;
    a.dL_dcolor = dL_dcolor;
    hipLaunchKernelGGL(k_render_backward, dim3(a.gridx * gridy), dim3(256), 0, L.stream, a);
    return check_launch(L, "render_backward");
}

}  // namespace gsr
