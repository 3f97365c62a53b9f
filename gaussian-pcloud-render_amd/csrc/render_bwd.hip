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
#include <cstdlib>

#include "common.hpp"
#include "tile_cull.hpp"

namespace gsr {

constexpr int BRB = 256;   // entries staged per round
constexpr int BGRP = 4;    // entries evaluated per inner-loop trip (see render_fwd.hip for the latency rationale)
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

struct BEntry {
    float4 q0;   // x, y, conic.x, conic.y
    float2 q1;   // conic.z, opacity
    float4 col;  // r, g, b, -
};

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
    uint64_t* tile_clock;
};

__global__ __launch_bounds__(256) void k_render_backward(RenderBwdArgs a)
{
    __shared__ float4 s_q0[BRB + 1];   // x, y, conic.x, conic.y   (slot BRB = null entry used for padding)
    __shared__ float2 s_q1[BRB + 1];   // conic.z, opacity
    __shared__ float4 s_col[BRB + 1];  // r, g, b, -
    __shared__ uint32_t s_id[BRB];
    __shared__ float s_acc[BRB + 1][NACC];
    __shared__ uint16_t s_list[4][BRB + 2 * BGRP];  // per quadrant: staged entries that may touch it (tile_cull.hpp)
    __shared__ uint32_t s_cnt[4][4];
    __shared__ uint32_t s_max[4];

    const uint32_t tile = a.tile_order[blockIdx.x];
    const uint64_t clk0 = wall_clock64();
    const uint32_t tx = tile % (uint32_t)a.gridx, ty = tile / (uint32_t)a.gridx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t px = tx * TILE_X + (w & 1) * 8 + (lane & 7);
    const uint32_t py = ty * TILE_Y + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const float tile_px = (float)(tx * TILE_X), tile_py = (float)(ty * TILE_Y);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
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

    if (tid == 0) {
        s_q0[BRB] = make_float4(0.f, 0.f, 0.f, 0.f);  // null entry: alpha = 0, never hits
        s_q1[BRB] = make_float2(0.f, 0.f);
        s_col[BRB] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    for (int base = 0; base < total; base += BRB) {
        const int n = total - base < BRB ? total - base : BRB;
        __syncthreads();  // previous round's flush is finished before LDS is reused
        uint32_t qmask = 0;
        if ((int)tid < n) {
            const uint32_t id = a.point_list[range.x + (uint32_t)(total - 1 - base - (int)tid)];
            const Splat* sp = a.splat + id;
            const float4 q0 = sp->q0, q1 = sp->q1, q2 = sp->q2;
            s_id[tid] = id;
            s_q0[tid] = q0;
            s_q1[tid] = make_float2(q1.x, q1.y);
            s_col[tid] = make_float4(q1.z, q1.w, q2.x, 0.f);
            qmask = quadrant_mask(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tile_px, tile_py);
        }
        for (int i = tid; i < (BRB + 1) * NACC; i += 256) (&s_acc[0][0])[i] = 0.f;
        uint64_t bal[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bal[q] = __ballot((qmask >> q) & 1u);
            if (lane == 0) s_cnt[q][w] = (uint32_t)__popcll(bal[q]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t off = 0;
#pragma unroll
            for (int ww = 0; ww < 4; ww++)
                if ((uint32_t)ww < w) off += s_cnt[q][ww];
            if ((qmask >> q) & 1u) s_list[q][off + (uint32_t)__popcll(bal[q] & lt_mask)] = (uint16_t)tid;
        }
        const int nq = (int)(s_cnt[w][0] + s_cnt[w][1] + s_cnt[w][2] + s_cnt[w][3]);
        const int nq_pad = (nq + BGRP - 1) / BGRP * BGRP;
        if ((int)lane < nq_pad + BGRP - nq) s_list[w][nq + lane] = (uint16_t)BRB;
        __syncthreads();

        const uint16_t* lst = s_list[w];
        BEntry cur[BGRP], nxt[BGRP];
        uint32_t ci[BGRP], ni[BGRP];
#pragma unroll
        for (int k = 0; k < BGRP; k++) {
            ci[k] = lst[k];
            cur[k].q0 = s_q0[ci[k]];
            cur[k].q1 = s_q1[ci[k]];
            cur[k].col = s_col[ci[k]];
        }
        for (int j0 = 0; j0 < nq_pad; j0 += BGRP) {
#pragma unroll
            for (int k = 0; k < BGRP; k++) {
                ni[k] = lst[j0 + BGRP + k];
                nxt[k].q0 = s_q0[ni[k]];
                nxt[k].q1 = s_q1[ni[k]];
                nxt[k].col = s_col[ni[k]];
            }
            float dxs[BGRP], dys[BGRP], Gs[BGRP], alphas[BGRP];
            bool hits[BGRP];
            bool any_lane_hit = false;
#pragma unroll
            for (int k = 0; k < BGRP; k++) {
                // 0-based index of this entry from the list front; the null entry (slot BRB) wraps to a huge value
                const uint32_t f = (uint32_t)(total - 1 - base) - ci[k];
                const float dx = cur[k].q0.x - pixf_x, dy = cur[k].q0.y - pixf_y;
                const float power = -0.5f * (cur[k].q0.z * dx * dx + cur[k].q1.x * dy * dy) - cur[k].q0.w * dx * dy;
                const float G = expf(power);
                const float alpha = fminf(0.99f, cur[k].q1.y * G);
                dxs[k] = dx; dys[k] = dy; Gs[k] = G; alphas[k] = alpha;
                hits[k] = (f < last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                any_lane_hit = any_lane_hit || hits[k];
            }
            if (__any(any_lane_hit)) {
                // Phase 1 (branch-free): advance the per-pixel recurrences through the four entries.  Only T,
                // accum_rec, last_color and last_alpha are serial; a lane that does not hit multiplies T by 1 and
                // keeps its state (selects), so the four steps are a short dependent chain the scheduler can
                // overlap with phase 2 of the same group.
                float dLa[BGRP], Gh[BGRP], dch[BGRP];
#pragma unroll
                for (int k = 0; k < BGRP; k++) {
                    const bool hit = hits[k];
                    const float alpha = alphas[k];
                    const float4 col = cur[k].col;
                    // The reference's two divisions by (1 - alpha) share one reciprocal here (gradients are compared
                    // to tolerance, not bit-for-bit: the accumulation order differs anyway).
                    const float rcp = 1.0f / (1.f - alpha);
                    const float Tn = T * (hit ? rcp : 1.0f);
                    const float om = 1.f - last_alpha;
                    const float r0 = last_alpha * lc0 + om * acc_r0;
                    const float r1 = last_alpha * lc1 + om * acc_r1;
                    const float r2 = last_alpha * lc2 + om * acc_r2;
                    float dL_dalpha = (col.x - r0) * dpx0;
                    dL_dalpha += (col.y - r1) * dpx1;
                    dL_dalpha += (col.z - r2) * dpx2;
                    dL_dalpha *= Tn;
                    dL_dalpha += (-T_final * rcp) * bg_dot_dpixel;
                    // lanes that do not hit contribute exact zeros: three selects zero every product of phase 2
                    dLa[k] = hit ? dL_dalpha : 0.f;
                    Gh[k] = hit ? Gs[k] : 0.f;
                    dch[k] = hit ? alpha * Tn : 0.f;
                    T = Tn;
                    acc_r0 = hit ? r0 : acc_r0; acc_r1 = hit ? r1 : acc_r1; acc_r2 = hit ? r2 : acc_r2;
                    lc0 = hit ? col.x : lc0; lc1 = hit ? col.y : lc1; lc2 = hit ? col.z : lc2;
                    last_alpha = hit ? alpha : last_alpha;
                }
                // Phase 2: per entry, the nine partial derivatives, wave reduction, LDS accumulation.
#pragma unroll
                for (int k = 0; k < BGRP; k++) {
                    if (!__any(hits[k])) continue;  // wave-uniform
                    const float dx = dxs[k], dy = dys[k];
                    const float4 q0 = cur[k].q0;
                    const float2 q1 = cur[k].q1;
                    const float dL_dG = q1.y * dLa[k];
                    const float gdx = Gh[k] * dx, gdy = Gh[k] * dy;
                    const float dG_ddelx = -gdx * q0.z - gdy * q0.w;
                    const float dG_ddely = -gdy * q1.x - gdx * q0.w;
                    float g[NACC];
                    g[0] = dL_dG * dG_ddelx * ddelx_dx;
                    g[1] = dL_dG * dG_ddely * ddely_dy;
                    g[2] = -0.5f * gdx * dx * dL_dG;
                    g[3] = -0.5f * gdx * dy * dL_dG;
                    g[4] = -0.5f * gdy * dy * dL_dG;
                    g[5] = Gh[k] * dLa[k];
                    g[6] = dch[k] * dpx0;
                    g[7] = dch[k] * dpx1;
                    g[8] = dch[k] * dpx2;
#pragma unroll
                    for (int c = 0; c < NACC; c++) g[c] = wave_sum_to_lane63(g[c]);
                    if (lane == 63) {
#pragma unroll
                        for (int c = 0; c < NACC; c++) atomicAdd(&s_acc[ci[k]][c], g[c]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < BGRP; k++) {
                cur[k] = nxt[k];
                ci[k] = ni[k];
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
    __syncthreads();
    if (tid == 0) {
        a.tile_clock[4 * tile + 2] = clk0;
        a.tile_clock[4 * tile + 3] = wall_clock64();
    }
}

int launch_render_backward(const Launch& L, const gsr_params& p, const GeomView& g, const uint32_t* point_list,
                           const ImageView& iv, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                           float* dL_dopacity, float* dL_dcolor)
{
    RenderBwdArgs a;
    a.ranges = iv.ranges;
    a.tile_order = iv.tile_order_bwd;
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
    a.tile_clock = iv.tile_clock;
    // experiment knob: extra (unused) dynamic LDS caps how many tiles are resident per CU, which turns the
    // hardware dispatcher into a longest-first dynamic scheduler
    static const int lds_pad = getenv("GSR_RENDER_LDS_PAD") ? atoi(getenv("GSR_RENDER_LDS_PAD")) : 0;
    hipLaunchKernelGGL(k_render_backward, dim3(a.gridx * gridy), dim3(256), lds_pad, L.stream, a);
    return check_launch(L, "render_backward");
}

}  // namespace gsr
