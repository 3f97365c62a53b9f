// render_fwd.hip -- per-tile front-to-back alpha compositing.
//
// Per-pixel arithmetic, thresholds and bookkeeping are those of reference CR/forward.cu:264-377
// (renderCUDA): power > 0 skip, alpha = min(0.99, o*exp(power)), alpha < 1/255 skip, stop when
// T*(1-alpha) < 1e-4 (that entry is not blended), C += colour*alpha*T, out = C + T*bg, plus final_T and
// n_contrib for the backward pass.  Tiles are the reference's 16x16 (they define keys and ranges).
//
// Mapping (ours): WAVE-AUTONOMOUS QUADRANTS.  The unit of work is one wave64 = one 8x8 quadrant of a tile
// (four single-wave workgroups per tile).  A wave walks its tile's list on its own:
//   - 64 list entries per round, one per lane: the lane gathers the entry's packed 48-B Splat record and
//     decides with the exact-safe footprint test (tile_cull.hpp) whether the entry can matter to THIS quadrant;
//     a ballot turns that into a 64-bit mask;
//   - the mask is consumed four set bits at a time by scalar code (s_ff1 / s_and), the chosen lanes' records are
//     broadcast with v_readlane into SGPRs, and the 64 pixels evaluate the four entries with scalar operands:
//     no LDS, no barrier, no waiting for sibling quadrants (a quadrant of a silhouette tile that sees half the
//     entries finishes in half the time and frees its SIMD slot);
//   - the next round's records (and the ids of the round after) are already in flight while a round is
//     evaluated, so the dependent id -> record gather latency is off the critical path;
//   - per-pixel skips are selects; the only branches are wave-uniform (mask empty, all 64 pixels done).
// The frame time of this kernel is set by the few longest lists (a wave walks its list serially); tiles are
// dispatched in descending list-length order (tile_order) so those start first.
#include "common.hpp"
#include "tile_cull.hpp"

namespace gsr {

constexpr int GRP = 4;  // entries evaluated per inner-loop trip

struct RenderArgs {
    const uint2* ranges;
    const uint32_t* tile_order;
    const uint32_t* point_list;
    const Splat* splat;
    int W, H, gridx, num_tiles;
    const float* bg;
    float* out_color;
    float* final_T;
    uint32_t* n_contrib;
    uint32_t* tile_need;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Prefetch loads are issued as inline asm so that hipcc's waitcnt pass does not see them: left to itself it puts
// an s_waitcnt for the NEXT round's records inside the CURRENT round's evaluation loop and re-exposes the gather
// latency every 64 entries.  The loads are retired by hand with one s_waitcnt vmcnt(0) at the rotation point; that
// asm takes the destination registers as in/out operands, so nothing can read them earlier.
__device__ __forceinline__ void prefetch16(f32x4& dst, const void* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void prefetch4(uint32_t& dst, const void* p)
{
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void retire_prefetch(f32x4& a, f32x4& b, f32x4& c, uint32_t& d)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// exp(x) for the compositing loop.  Instruction-for-instruction the core of the ocml expf that `exp(power)` of the
// reference resolves to under hipcc (extended-precision x*log2(e), v_rndne, v_exp_f32, v_ldexp_f32), minus its two
// range clamps: x > 88.7 -> inf and x < -103.3 -> 0.  Neither can change a decision or a blended value: entries
// with power > 0 are skipped before alpha is used, and for x < -103 both forms give a value < 1e-44, far below
// the 1/255 cut for any finite opacity.  For every x in [-103, 0] the result is bit-identical to expf(x).
__device__ __forceinline__ float exp_nonpos(float x)
{
    const float ph = x * 0x1.715476p+0f;
    float pl = __builtin_fmaf(x, 0x1.715476p+0f, -ph);
    pl = __builtin_fmaf(x, 0x1.4ae0bep-26f, pl);
    const float e = __builtin_rintf(ph);
    const float r = __builtin_amdgcn_exp2f((ph - e) + pl);
    return __builtin_ldexpf(r, (int)e);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// two-entry version: the multiplies / fused multiply-adds / adds become packed fp32 instructions (v_pk_*_f32, two
// IEEE operations per lane per issue slot); rounding per component is that of exp_nonpos
__device__ __forceinline__ f32x2 exp_nonpos2(f32x2 x)
{
    const f32x2 c = {0x1.715476p+0f, 0x1.715476p+0f}, cc = {0x1.4ae0bep-26f, 0x1.4ae0bep-26f};
    const f32x2 ph = x * c;
    f32x2 pl = __builtin_elementwise_fma(x, c, -ph);
    pl = __builtin_elementwise_fma(x, cc, pl);
    const f32x2 e = {__builtin_rintf(ph.x), __builtin_rintf(ph.y)};
    const f32x2 a = (ph - e) + pl;
    f32x2 r;
    r.x = __builtin_ldexpf(__builtin_amdgcn_exp2f(a.x), (int)e.x);
    r.y = __builtin_ldexpf(__builtin_amdgcn_exp2f(a.y), (int)e.y);
    return r;
}

__device__ __forceinline__ float lane_bcast(float v, int src_lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

__global__ __launch_bounds__(64) void k_render_forward(RenderArgs a)
{
    // XCD-aware work mapping: workgroup b runs on XCD b % 8 (each XCD has its own L2), so the four quadrant waves of
    // one tile are workgroups b, b+8, b+16, b+24: same XCD, dispatched together, and the tile's list and Splat records
    // are fetched into that L2 once instead of four times.
    const uint32_t order_slot = (blockIdx.x >> 5) * 8u + (blockIdx.x & 7u);
    if (order_slot >= (uint32_t)a.num_tiles) return;
    const uint32_t tile = a.tile_order[order_slot];
    const uint32_t q = (blockIdx.x >> 3) & 3u;
    const uint32_t lane = threadIdx.x;
    const uint32_t tx = tile % (uint32_t)a.gridx, ty = tile / (uint32_t)a.gridx;
    const uint32_t x0 = tx * TILE_X + (q & 1u) * 8u, y0 = ty * TILE_Y + (q >> 1) * 8u;
    const uint32_t px = x0 + (lane & 7u), py = y0 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const float x0f = (float)x0, y0f = (float)y0;

    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);

    float T = 1.0f;
    f32x2 C01 = {0.f, 0.f};
    float C2 = 0.f;
    uint32_t last_contributor = 0;
    uint32_t stop_at = 0;  // 1-based index of the entry that terminated this pixel
    bool done = !inside;
    bool all_done = __all(done);

    if (!all_done && total > 0) {
        const uint32_t* plist = a.point_list + range.x;
        // software pipeline: records of round r+1 and ids of round r+2 are in flight while round r is evaluated.
        // Lanes past the end of the list read entry total-1 again (always a valid address) and are masked by `valid`.
        const int last = total - 1;
        f32x4 c0, c1, c2, n0, n1, n2;
        uint32_t id_nxt, id_nn;
        {
            // prologue: round 0's records and round 1's ids, through the same asm path so that no compiler-tracked
            // load is pending when the loop is entered
            uint32_t id0;
            prefetch4(id0, plist + ((int)lane < total ? (int)lane : last));
            prefetch4(id_nxt, plist + (64 + (int)lane < total ? 64 + (int)lane : last));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(id0), "+v"(id_nxt)::"memory");
            const Splat* sp = a.splat + id0;
            prefetch16(c0, &sp->q0);
            prefetch16(c1, &sp->q1);
            prefetch16(c2, &sp->q2);
            retire_prefetch(c0, c1, c2, id_nxt);
        }
        for (int base = 0; base < total; base += 64) {
            {
                const Splat* sp = a.splat + id_nxt;
                prefetch16(n0, &sp->q0);
                prefetch16(n1, &sp->q1);
                prefetch16(n2, &sp->q2);
                const int i2 = base + 128 + (int)lane;
                prefetch4(id_nn, plist + (i2 < total ? i2 : last));
            }

            // which of this round's 64 entries can reach alpha >= 1/255 somewhere in this quadrant?
            const bool valid = base + (int)lane < total;
            const bool touch = valid && may_touch_8x8(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, x0f, y0f);
            uint64_t mask = __ballot(touch);

            while (mask != 0 && !all_done) {
                float ex[GRP], ey[GRP], eA[GRP], eB[GRP], eC[GRP], eo[GRP], er[GRP], eg[GRP], eb[GRP];
                uint32_t eidx[GRP];
#pragma unroll
                for (int k = 0; k < GRP; k++) {
                    const bool have = mask != 0;
                    const int j = have ? (int)__builtin_ctzll(mask) : 0;
                    mask = have ? (mask & (mask - 1)) : 0;
                    ex[k] = lane_bcast(c0.x, j); ey[k] = lane_bcast(c0.y, j);
                    eA[k] = lane_bcast(c0.z, j); eB[k] = lane_bcast(c0.w, j);
                    eC[k] = lane_bcast(c1.x, j);
                    const float o = lane_bcast(c1.y, j);
                    eo[k] = have ? o : 0.f;  // opacity 0 -> alpha 0 -> the 1/255 test drops the slot
                    er[k] = lane_bcast(c1.z, j); eg[k] = lane_bcast(c1.w, j); eb[k] = lane_bcast(c2.x, j);
                    eidx[k] = (uint32_t)(base + j + 1);  // 1-based position in the tile list
                }
                // alpha of each entry; ae = alpha where the entry counts for this pixel, else 0.  Entries are taken in
                // pairs so the arithmetic maps onto packed fp32 instructions (this kernel is VALU-issue bound).
                float alpha[GRP], ae[GRP];
                bool cnt[GRP];
#pragma unroll
                for (int k = 0; k < GRP; k += 2) {
                    const f32x2 X = {ex[k], ex[k + 1]}, Y = {ey[k], ey[k + 1]}, A2 = {eA[k], eA[k + 1]};
                    const f32x2 B2 = {eB[k], eB[k + 1]}, C2 = {eC[k], eC[k + 1]}, O2 = {eo[k], eo[k + 1]};
                    const f32x2 dx = X - pixf_x, dy = Y - pixf_y;
                    const f32x2 power = -0.5f * (A2 * dx * dx + C2 * dy * dy) - B2 * dx * dy;
                    const f32x2 al = O2 * exp_nonpos2(power);
                    alpha[k] = fminf(0.99f, al.x);
                    alpha[k + 1] = fminf(0.99f, al.y);
                    cnt[k] = !done && !(power.x > 0.0f) && !(alpha[k] < 1.0f / 255.0f);
                    cnt[k + 1] = !done && !(power.y > 0.0f) && !(alpha[k + 1] < 1.0f / 255.0f);
                    ae[k] = cnt[k] ? alpha[k] : 0.f;
                    ae[k + 1] = cnt[k + 1] ? alpha[k + 1] : 0.f;
                }
                // Optimistic pass: assume no pixel of this wave terminates inside the group.  T then only needs the
                // products T*(1-ae) (ae = 0 multiplies by exactly 1), and because T never increases and every live
                // pixel has T >= 1e-4, "some entry of the group would have stopped a pixel" is just T_after < 1e-4.
                // A pixel stops once, so the exact serial fallback runs for at most 64 groups per wave per tile.
                float Tk[GRP + 1];
                Tk[0] = T;
#pragma unroll
                for (int k = 0; k < GRP; k++) Tk[k + 1] = Tk[k] * (1 - ae[k]);
                if (!__any(Tk[GRP] < 0.0001f)) {
#pragma unroll
                    for (int k = 0; k < GRP; k++) {
                        // ae = 0 adds a zero, which leaves C unchanged bit-for-bit (C is never -0)
                        const f32x2 rg = {er[k], eg[k]};
                        C01 += rg * ae[k] * Tk[k];
                        C2 += eb[k] * ae[k] * Tk[k];
                        last_contributor = cnt[k] ? eidx[k] : last_contributor;
                    }
                    T = Tk[GRP];
                } else {
#pragma unroll
                    for (int k = 0; k < GRP; k++) {
                        const float test_T = T * (1 - alpha[k]);
                        const bool hit = !done && cnt[k];
                        const bool stop = hit && (test_T < 0.0001f);
                        const bool blend = hit && !stop;
                        C01.x += blend ? er[k] * alpha[k] * T : 0.f;
                        C01.y += blend ? eg[k] * alpha[k] * T : 0.f;
                        C2 += blend ? eb[k] * alpha[k] * T : 0.f;
                        T = blend ? test_T : T;
                        last_contributor = blend ? eidx[k] : last_contributor;
                        stop_at = stop ? eidx[k] : stop_at;
                        done = done || stop;
                    }
                    all_done = __all(done);
                }
            }
            retire_prefetch(n0, n1, n2, id_nn);
            if (all_done) break;
            c0 = n0; c1 = n1; c2 = n2;
            id_nxt = id_nn;
        }
    }

    // instrumentation: how many list entries this tile really needed (max over its pixels); tile_need is zeroed
    // before the launch
    {
        uint32_t need = inside ? (done ? stop_at : (uint32_t)total) : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(need, d, 64);
            need = need > o ? need : o;
        }
        if (lane == 0 && need != 0) atomicMax(&a.tile_need[tile], need);
    }

    if (inside) {
        const size_t pix = (size_t)py * a.W + px, N = (size_t)a.W * a.H;
        a.final_T[pix] = T;
        a.n_contrib[pix] = last_contributor;
        a.out_color[pix] = C01.x + T * a.bg[0];
        a.out_color[N + pix] = C01.y + T * a.bg[1];
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
    const int T = a.gridx * gridy;
    if (hipMemsetAsync(iv.tile_need, 0, (size_t)T * sizeof(uint32_t), L.stream) != hipSuccess) return GSR_ERR_HIP;
    a.num_tiles = T;
    hipLaunchKernelGGL(k_render_forward, dim3((unsigned)div_up(T, 8) * 32u), dim3(64), 0, L.stream, a);
    return check_launch(L, "render_forward");
}

}  // namespace gsr
