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
// wave-uniform broadcasts: (x,y,A,B) b128 + (C,o) b64, and the colour b128 only on iterations where
// at least one lane blends.  A wave leaves the inner loop as soon as all its 64 pixels are done
// (ballot), the workgroup stops staging once all four waves are.  Tiles are dispatched in descending
// list-length order (tile_order) so the longest lists start first.
#include "common.hpp"

namespace gsr {

constexpr int RB = 256;  // entries staged per round

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

__global__ __launch_bounds__(256) void k_render_forward(RenderArgs a)
{
    __shared__ float4 s_q0[RB];   // x, y, conic.x, conic.y
    __shared__ float2 s_q1[RB];   // conic.z, opacity
    __shared__ float4 s_col[RB];  // r, g, b, -
    __shared__ int s_live;
    __shared__ uint32_t s_need;

    const uint32_t tile = a.tile_order[blockIdx.x];
    const uint32_t tx = tile % (uint32_t)a.gridx, ty = tile / (uint32_t)a.gridx;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // wave w owns quadrant (w&1, w>>1); lane -> (lane&7, lane>>3) inside it
    const uint32_t px = tx * TILE_X + (w & 1) * 8 + (lane & 7);
    const uint32_t py = ty * TILE_Y + (w >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixf_x = (float)px, pixf_y = (float)py;

    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);

    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_contributor = 0;
    bool done = !inside;
    uint32_t walked = 0;  // list entries this wave looked at
    if (tid == 0) s_need = 0;

    for (int base = 0; base < total; base += RB) {
        // workgroup-wide early exit (reference: __syncthreads_count(done) == BLOCK_SIZE)
        const bool wave_live = !__all(done);  // ballot over the whole wave, taken in uniform control flow
        if (tid == 0) s_live = 0;
        __syncthreads();
        if (lane == 0 && wave_live) s_live = 1;  // benign race: every writer stores 1
        __syncthreads();
        if (!s_live) break;

        const int n = total - base < RB ? total - base : RB;
        if ((int)tid < n) {
            const uint32_t id = a.point_list[range.x + base + tid];
            const Splat* sp = a.splat + id;
            const float4 q0 = sp->q0, q1 = sp->q1, q2 = sp->q2;
            s_q0[tid] = q0;
            s_q1[tid] = make_float2(q1.x, q1.y);
            s_col[tid] = make_float4(q1.z, q1.w, q2.x, 0.f);
        }
        __syncthreads();

        if (wave_live) {
            for (int j = 0; j < n; j++) {
                const float4 q0 = s_q0[j];
                const float2 q1 = s_q1[j];
                const float dx = q0.x - pixf_x, dy = q0.y - pixf_y;
                const float power = -0.5f * (q0.z * dx * dx + q1.x * dy * dy) - q0.w * dx * dy;
                const float alpha = fminf(0.99f, q1.y * expf(power));
                const float test_T = T * (1 - alpha);
                const bool hit = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                const bool stop = hit && (test_T < 0.0001f);
                const bool blend = hit && !stop;
                if (__any(blend)) {
                    const float4 col = s_col[j];
                    if (blend) {
                        C0 += col.x * alpha * T;
                        C1 += col.y * alpha * T;
                        C2 += col.z * alpha * T;
                        T = test_T;
                        last_contributor = (uint32_t)(base + j + 1);
                    }
                }
                done = done || stop;
                walked = (uint32_t)(base + j + 1);
                if (__all(done)) break;
            }
        }
    }

    __syncthreads();
    if (lane == 0) atomicMax(&s_need, walked);
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
