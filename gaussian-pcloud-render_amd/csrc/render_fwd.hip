// render_fwd.hip -- per-tile front-to-back alpha compositing.
//
// Per-pixel arithmetic, thresholds and bookkeeping are those of reference CR/forward.cu:264-377
// (renderCUDA): power > 0 skip, alpha = min(0.99, o*exp(power)), alpha < 1/255 skip, stop when
// T*(1-alpha) < 1e-4 (that entry is not blended), C += colour*alpha*T, out = C + T*bg, plus final_T and
// n_contrib for the backward pass.
//
// Mapping (ours): one 256-thread workgroup per 16x16 tile, four wave64s each owning an 8x8 pixel
// quadrant.  The tile's Gaussian list is staged through LDS 256 entries at a time from the packed
// 48-B Splat records (one gather per entry instead of three), as SoA so the inner loop's reads are
// wave-uniform broadcasts.  The frame time of this kernel is set by the few longest tile lists (a wave
// walks its list serially), so the inner loop is built for latency, not just throughput:
//   - entries are taken GROUP at a time: all LDS reads of a group are issued together, the next group is
//     prefetched into registers while the current one is evaluated, the G exp/alpha evaluations are
//     independent (ILP), and only the short T / C recurrence is serial;
//   - per-pixel skips are selects, not branches; the only branch is the wave-uniform "all 64 pixels
//     done" ballot once per group, and the workgroup stops staging once all four waves are done;
//   - tiles are dispatched in descending list-length order (tile_order) so the longest lists start first.
#include "common.hpp"
#include "tile_cull.hpp"

namespace gsr {

constexpr int RB = 256;  // entries staged per round
constexpr int GRP = 4;   // entries evaluated per inner-loop trip

struct RenderArgs {
    const uint2* ranges;
    const uint32_t* tile_order;
    const uint32_t* point_list;
    const Splat* splat;
    int W, H, gridx;
    const float* bg;
    float* out_color;
    float* final_T;
    uint32_t* n_contrib;
    uint32_t* tile_need;
};

struct EntryRegs {
    float4 q0;   // x, y, conic.x, conic.y
    float2 q1;   // conic.z, opacity
    float4 col;  // r, g, b, -
};

__global__ __launch_bounds__(256) void k_render_forward(RenderArgs a)
{
    __shared__ float4 s_q0[RB + 1];   // slot RB is the null entry (alpha = 0) used for padding
    __shared__ float2 s_q1[RB + 1];
    __shared__ float4 s_col[RB + 1];
    __shared__ uint16_t s_list[4][RB + 2 * GRP];  // per quadrant: indices of the staged entries that may touch it
    __shared__ uint32_t s_cnt[4][4];               // [quadrant][staging wave]
    __shared__ uint32_t s_livew[2][4];
    __shared__ uint32_t s_need;

    const uint32_t tile = a.tile_order[blockIdx.x];
    const uint32_t tx = tile % (uint32_t)a.gridx, ty = tile / (uint32_t)a.gridx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // wave w owns quadrant (w&1, w>>1); lane -> (lane&7, lane>>3) inside it
    const uint32_t px = tx * TILE_X + (w & 1) * 8 + (lane & 7);
    const uint32_t py = ty * TILE_Y + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const float tile_px = (float)(tx * TILE_X), tile_py = (float)(ty * TILE_Y);
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);

    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_contributor = 0;
    uint32_t stop_at = 0;  // 1-based index of the entry that terminated this pixel
    bool done = !inside;
    if (tid == 0) {
        s_need = 0;
        s_q0[RB] = make_float4(0.f, 0.f, 0.f, 0.f);  // null entry: power = -0, alpha = 0 -> never passes 1/255
        s_q1[RB] = make_float2(0.f, 0.f);
        s_col[RB] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        const bool wave_live = !__all(done);
        if (lane == 0) s_livew[0][w] = wave_live ? 1u : 0u;
    }

    int round = 0;
    for (int base = 0; base < total; base += RB, round++) {
        // S0: everyone has finished reading the previous round's LDS and published its live flag.
        __syncthreads();
        // workgroup-wide early exit (reference: __syncthreads_count(done) == BLOCK_SIZE)
        const uint32_t* lv = s_livew[round & 1];
        if ((lv[0] | lv[1] | lv[2] | lv[3]) == 0u) break;

        const int n = total - base < RB ? total - base : RB;
        uint32_t qmask = 0;
        if ((int)tid < n) {
            const uint32_t id = a.point_list[range.x + base + tid];
            const Splat* sp = a.splat + id;
            const float4 q0 = sp->q0, q1 = sp->q1, q2 = sp->q2;
            s_q0[tid] = q0;
            s_q1[tid] = make_float2(q1.x, q1.y);
            s_col[tid] = make_float4(q1.z, q1.w, q2.x, 0.f);
            qmask = quadrant_mask(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tile_px, tile_py);
        }
        uint64_t bal[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bal[q] = __ballot((qmask >> q) & 1u);
            if (lane == 0) s_cnt[q][w] = (uint32_t)__popcll(bal[q]);
        }
        __syncthreads();  // SA: counts visible
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t off = 0;
#pragma unroll
            for (int ww = 0; ww < 4; ww++)
                if ((uint32_t)ww < w) off += s_cnt[q][ww];
            if ((qmask >> q) & 1u) s_list[q][off + (uint32_t)__popcll(bal[q] & lt_mask)] = (uint16_t)tid;
        }
        // this wave's own quadrant list: length, padding with the null entry up to a whole group + one prefetch group
        const int nq = (int)(s_cnt[w][0] + s_cnt[w][1] + s_cnt[w][2] + s_cnt[w][3]);
        const int nq_pad = (nq + GRP - 1) / GRP * GRP;
        if ((int)lane < nq_pad + GRP - nq) s_list[w][nq + lane] = (uint16_t)RB;
        __syncthreads();  // SB: lists complete

        if (!__all(done)) {
            const uint16_t* lst = s_list[w];
            EntryRegs cur[GRP], nxt[GRP];
            uint32_t ci[GRP], ni[GRP];
#pragma unroll
            for (int k = 0; k < GRP; k++) {
                ci[k] = lst[k];
                cur[k].q0 = s_q0[ci[k]];
                cur[k].q1 = s_q1[ci[k]];
                cur[k].col = s_col[ci[k]];
            }
            for (int j0 = 0; j0 < nq_pad; j0 += GRP) {
                // prefetch the next group (the list is padded by one extra group of null entries)
#pragma unroll
                for (int k = 0; k < GRP; k++) {
                    ni[k] = lst[j0 + GRP + k];
                    nxt[k].q0 = s_q0[ni[k]];
                    nxt[k].q1 = s_q1[ni[k]];
                    nxt[k].col = s_col[ni[k]];
                }
                float alpha[GRP];
                bool vis[GRP];
#pragma unroll
                for (int k = 0; k < GRP; k++) {
                    const float dx = cur[k].q0.x - pixf_x, dy = cur[k].q0.y - pixf_y;
                    const float power = -0.5f * (cur[k].q0.z * dx * dx + cur[k].q1.x * dy * dy) - cur[k].q0.w * dx * dy;
                    alpha[k] = fminf(0.99f, cur[k].q1.y * expf(power));
                    vis[k] = !(power > 0.0f) && !(alpha[k] < 1.0f / 255.0f);
                }
#pragma unroll
                for (int k = 0; k < GRP; k++) {
                    const float test_T = T * (1 - alpha[k]);
                    const bool hit = !done && vis[k];
                    const bool stop = hit && (test_T < 0.0001f);
                    const bool blend = hit && !stop;
                    // adding +0 leaves C unchanged bit-for-bit (C is never -0)
                    C0 += blend ? cur[k].col.x * alpha[k] * T : 0.f;
                    C1 += blend ? cur[k].col.y * alpha[k] * T : 0.f;
                    C2 += blend ? cur[k].col.z * alpha[k] * T : 0.f;
                    T = blend ? test_T : T;
                    const uint32_t idx1 = (uint32_t)base + ci[k] + 1u;  // 1-based position in the tile list
                    last_contributor = blend ? idx1 : last_contributor;
                    stop_at = stop ? idx1 : stop_at;
                    done = done || stop;
                }
                if (__all(done)) break;
#pragma unroll
                for (int k = 0; k < GRP; k++) {
                    cur[k] = nxt[k];
                    ci[k] = ni[k];
                }
            }
        }
        {
            const bool wave_live = !__all(done);
            if (lane == 0) s_livew[(round + 1) & 1][w] = wave_live ? 1u : 0u;
        }
    }

    // instrumentation: how many list entries this tile really needed (max over its pixels)
    __syncthreads();
    if (inside) atomicMax(&s_need, done ? stop_at : (uint32_t)total);
    __syncthreads();
    if (tid == 0) a.tile_need[tile] = s_need;

    if (inside) {
        const size_t pix = (size_t)py * a.W + px, N = (size_t)a.W * a.H;
        a.final_T[pix] = T;
        a.n_contrib[pix] = last_contributor;
        a.out_color[pix] = C0 + T * a.bg[0];
        a.out_color[N + pix] = C1 + T * a.bg[1];
        a.out_color[2 * N + pix] = C2 + T * a.bg[2];
    }
}

int launch_render_forward(const Launch& L, const gsr_params& p, const GeomView& g, const uint32_t* point_list,
                          const ImageView& iv, float* out_color)
{
    RenderArgs a;
    a.ranges = iv.ranges;
    a.tile_order = iv.tile_order;
    a.point_list = point_list;
    a.splat = g.splat;
    a.W = p.W; a.H = p.H;
    a.gridx = (p.W + TILE_X - 1) / TILE_X;
    const int gridy = (p.H + TILE_Y - 1) / TILE_Y;
    a.bg = p.bg;
    a.out_color = out_color;
    a.final_T = iv.final_T;
    a.n_contrib = iv.n_contrib;
    a.tile_need = iv.tile_need;
    hipLaunchKernelGGL(k_render_forward, dim3(a.gridx * gridy), dim3(256), 0, L.stream, a);
    return check_launch(L, "render_forward");
}

}  // namespace gsr
