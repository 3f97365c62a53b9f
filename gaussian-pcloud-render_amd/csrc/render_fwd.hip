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
//   - the surviving lanes write their records (and the entries' list positions), compacted and interleaved in
//     pairs, into 2.5 KB of LDS that belongs to the wave; the wave reads the pairs back with same-address ds_read, so
//     the 64 pixels evaluate two entries at a time on packed fp32 instructions: no barrier (DS operations of one wave
//     execute in order), no waiting for sibling quadrants (a quadrant of a silhouette tile that sees half the entries
//     finishes in half the time and frees its SIMD slot);
//   - the next round's records (and the ids of the round after) are already in flight while a round is
//     evaluated, so the dependent id -> record gather latency is off the critical path;
//   - per-pixel skips are selects; the only branches are wave-uniform (mask empty, all 64 pixels done).
// The duration of this kernel is set by the quadrants that walk deepest (a wave walks its list serially, one
// instruction per ~5.5 cycles when it has its SIMD to itself); tiles are dispatched by descending work estimate
// (binning.hip k_tile_order).  With need_backward the per-pixel (T, C) state is left at every BWD_CHUNK-entry boundary a quadrant
// crosses, so that the backward pass can work on slices of lists (render_bwd.hip).
//
// k_render_forward<NX>, NX = 4 or 8: the same walk also composites NX extra per-Gaussian channels with the colour's alphas
// (gsr_forward_batch_channels: the reference's callers render world xyz, a hit map and normals as three more full passes,
// simple_raw_render.py:410-524).  The extra values ride in the pair records next to the colour; NX = 0 is the plain kernel.
#include <atomic>
#include <cstdlib>

#include "common.hpp"
#include "tile_cull.hpp"

namespace gsr {


struct RenderArgs {
    const uint2* ranges;
    const uint32_t* tile_order;
    const uint32_t* point_list;
    const Splat* splat;
    int W, H, gridx, num_tiles, chunk_shift;
    const float* bg;
    float* out_color;
    float* final_T;
    uint32_t* n_contrib;
    uint32_t* tile_need;
    float4* ckpt;   // chunk-boundary state for the backward pass, or NULL
    float* accum;   // [3N] accumulated colour without background (written with ckpt, for quadrants that crossed a chunk boundary)
    uint32_t V;                              // views in the batch
    size_t g_stride, b_stride, iv_stride;    // bytes between consecutive views' arenas
    // extra channels (k_render_forward<NX>, NX > 0): composited with the same alphas as the colour
    const float* extra;        // [P][NX] per-Gaussian values, shared by the views (extra_vstride = 0) or one array per view
    size_t extra_vstride;      // floats between consecutive views' arrays
    size_t extra_hi_vstride;
    const float* extra_hi;     // NX = 8, split layout: channels 4..7 as [V][P][4] (channels 0..3 then come from `extra` as [P][4],
                               // shared by the views); NULL: all NX channels interleaved in `extra`
    const float* extra_scale;  // [V][NX] per-view factors applied to them (NULL: 1), e.g. the +-1 of view-dependent normals
    const float* bg_extra;     // [NX]
    float* out_extra;          // [V][NX][H][W]
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
__device__ __forceinline__ void prefetch4f(float& dst, const void* p)
{
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void retire_prefetch(f32x4& a, f32x4& b, float& c, uint32_t& d)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

__device__ __forceinline__ void retire_prefetch_x(f32x4& a, f32x4& b, float& c, uint32_t& d, f32x4& e, f32x4& f)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)::"memory");
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

// Broadcast of the surviving entries to the 64 pixels goes through LDS: the lanes whose entry survived the footprint
// test write their records, compacted and interleaved in PAIRS (x0 x1 y0 y1 | A0 A1 B0 B1 | C0 C1 o0 o1 |
// r0 g0 r1 g1 | b0 b1), and the wave reads one pair back with five same-address ds_read (every lane gets every
// value; operands arrive as the register pairs the packed fp32 instructions want).  The obvious alternative, nine
// v_readlane per entry, costs about as many VALU issue cycles as the whole alpha evaluation (measured: one
// v_readlane ~ 2-3 plain VALU ops, scripts/probe/readlane_probe.hip), and this kernel is VALU-issue bound; scalar
// loads (s_load through the scalar cache) have the right cost but ~1 us latency, which the serial walk of the
// longest lists cannot hide.  A wave owns its 2.5 KB of LDS: no barrier anywhere (DS operations of one wave execute
// in order).
// With NX extra channels a pair record grows by 2 NX words: for every two channels (k, k+1) the quad
// (k of entry 0, k+1 of entry 0, k of entry 1, k+1 of entry 1), the layout of `rg`.
constexpr int PAIR_WORDS = 20;
template <int NX>
struct PairRec {
    f32x4 xy;   // x0 x1 y0 y1
    f32x4 ab;   // A0 A1 B0 B1
    f32x4 co;   // C0 C1 o0 o1
    f32x4 rg;   // r0 g0 r1 g1
    f32x2 b;    // b0 b1
    uint2 pos;  // 1-based list positions of the two entries
    f32x4 ex[NX > 0 ? NX / 2 : 1];
};
template <int NX>
__device__ __forceinline__ PairRec<NX> read_pair(const float* lds, int pair)
{
    const float* p = lds + pair * (PAIR_WORDS + 2 * NX);
    PairRec<NX> r;
    r.xy = *(const f32x4*)(p + 0);
    r.ab = *(const f32x4*)(p + 4);
    r.co = *(const f32x4*)(p + 8);
    r.rg = *(const f32x4*)(p + 12);
    r.b = *(const f32x2*)(p + 16);
    r.pos = *(const uint2*)(p + 18);
#pragma unroll
    for (int j = 0; j < NX / 2; j++) r.ex[j] = *(const f32x4*)(p + PAIR_WORDS + 4 * j);
    return r;
}

// One surviving entry into its compacted slot of the staging area (both forward kernels): the nine values the pixels need and the
// entry's 1-based list position, in the pair layout above.  The last survivor of an ODD count also fills its pair's second half with a
// copy of itself at opacity 0 (alpha 0: the 1/255 test drops it).  Returns true for that lane (extra channels are copied likewise).
__device__ __forceinline__ bool stage_entry(float* stage, int pw, uint32_t slot, uint32_t nsurv, const f32x4& c0, const f32x4& c1, float c2b,
                                            uint32_t my_pos)
{
    float* p = stage + (slot >> 1) * pw + (slot & 1u);
    p[0] = c0.x; p[2] = c0.y; p[4] = c0.z; p[6] = c0.w; p[8] = c1.x; p[10] = c1.y;
    float* pc = stage + (slot >> 1) * pw + 12 + 2 * (slot & 1u);
    pc[0] = c1.z; pc[1] = c1.w;
    p[16] = c2b;
    uint32_t* pp = (uint32_t*)p;
    pp[18] = my_pos;
    const bool lone = slot + 1 == nsurv && (slot & 1u) == 0;
    if (lone) {
        p[1] = c0.x; p[3] = c0.y; p[5] = c0.z; p[7] = c0.w; p[9] = c1.x; p[11] = 0.f;
        pc[2] = c1.z; pc[3] = c1.w;
        p[17] = c2b;
        pp[19] = my_pos;
    }
    return lone;
}

#ifdef GSR_STATS
// instrumentation build only: where the forward waves' time goes, summed over the waves of the launches since the last reset
// (10-ns ticks): 0 whole life, 1 waiting for the next round's records at the rotation point, 2 footprint test + staging,
// 3 pair evaluation, 4 waves, 5 rounds, 6 pairs evaluated
// (one record per wave of the LAST launch -- same-address atomics from 390 K waves would be what gets measured)
constexpr int FW_REC = 1 << 19;
__device__ unsigned g_fwd_rec[FW_REC][8];
__device__ unsigned g_fwd_hw[FW_REC][2];   // where and when each wave ran: HW_ID | XCC_ID << 28, start tick (low 32 bits)
// raw per-wave records of the last launch, ten words per wave (scripts/debug/fwd_placement.py)
int debug_fwd_records(unsigned* out, int n)
{
    static unsigned host[FW_REC][8], hw[FW_REC][2];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd_rec), sizeof(host)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_fwd_hw), sizeof(hw)) != hipSuccess) return -1;
    if (n > FW_REC) n = FW_REC;
    for (int r = 0; r < n; r++) {
        for (int i = 0; i < 8; i++) out[r * 10 + i] = host[r][i];
        out[r * 10 + 8] = hw[r][0]; out[r * 10 + 9] = hw[r][1];
    }
    return n;
}
#define FW_T(var) const unsigned long long var = wall_clock64()
int debug_fwd_times(unsigned long long* out8, int reset)
{
    static unsigned host[FW_REC][8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fwd_rec), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    for (int r = 0; r < FW_REC; r++) {
        for (int i = 0; i < 7; i++) out8[i] += host[r][i];
        if (host[r][0] > out8[7]) out8[7] = host[r][0];   // the longest-lived wave
    }
    if (reset) {
        for (int r = 0; r < FW_REC; r++) for (int i = 0; i < 8; i++) host[r][i] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_rec), host, sizeof(host)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

template <int NX>
__global__ __launch_bounds__(64) void k_render_forward(RenderArgs a)
{
#ifdef GSR_STATS
    FW_T(tw0);
    unsigned long long tw_wait = 0, tw_stage = 0, tw_eval = 0, n_rounds = 0, n_pairs = 0;
    unsigned live_cnt = 64, n_pairs_le8 = 0, n_pairs_le16 = 0, n_pairs_le32 = 0;   // pairs evaluated while <= 8 / 16 / 32 pixels were live
#endif
    static_assert(NX == 0 || NX == 4 || NX == 8, "extra channels come in quads");
    constexpr int PW = PAIR_WORDS + 2 * NX;   // words per staged pair
    // XCD-aware work mapping: workgroup b runs on XCD b % 8 (each XCD has its own L2), so the four quadrant waves of
    // one tile are workgroups b, b+8, b+16, b+24: same XCD, dispatched together, and the tile's list and Splat records
    // are fetched into that L2 once instead of four times.
    // Batches: groups of 32 workgroups (8 tiles x 4 quadrants) are dealt to the views round-robin, so the heaviest tiles of
    // EVERY view are dispatched first and the batch has one tail instead of one per view.
    const uint32_t group = blockIdx.x >> 5;
    const uint32_t view = group % a.V;
    const uint32_t order_slot = (group / a.V) * 8u + (blockIdx.x & 7u);
    if (order_slot >= (uint32_t)a.num_tiles) return;
    a.ranges = at_view(a.ranges, a.iv_stride, view);
    a.tile_order = at_view(a.tile_order, a.iv_stride, view);
    a.final_T = at_view(a.final_T, a.iv_stride, view);
    a.n_contrib = at_view(a.n_contrib, a.iv_stride, view);
    a.tile_need = at_view(a.tile_need, a.iv_stride, view);
    a.accum = at_view(a.accum, a.iv_stride, view);
    a.point_list = at_view(a.point_list, a.b_stride, view);
    if (a.ckpt) a.ckpt = at_view(a.ckpt, a.b_stride, view);
    a.splat = at_view(a.splat, a.g_stride, view);
    a.out_color += (size_t)view * 3u * (size_t)a.W * (size_t)a.H;
    if (NX > 0) a.extra += (size_t)view * a.extra_vstride;
    // split layout: a Gaussian's first quad is extra[4 id], its second extra_hi[4 (view P + id)] (both 16-B records)
    const bool x_split = NX > 4 && a.extra_hi != nullptr;
    const float* const x_hi = x_split ? a.extra_hi + (size_t)view * a.extra_hi_vstride : a.extra + 4;
    const uint32_t x_nx = x_split ? 4u : (uint32_t)NX;
    const uint32_t tile = a.tile_order[order_slot];
    const uint32_t q = (blockIdx.x >> 3) & 3u;
    const uint32_t lane = threadIdx.x;
    const uint32_t tx = tile % (uint32_t)a.gridx, ty = tile / (uint32_t)a.gridx;
    const uint32_t x0 = tx * TILE_X + (q & 1u) * 8u, y0 = ty * TILE_Y + (q >> 1) * 8u;
    const uint32_t px = x0 + (lane & 7u), py = y0 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const float x0f = (float)x0, y0f = (float)y0;
    // rectangle of the pixels that are still live (not outside the image, not terminated): the footprint test only has to
    // keep entries that can reach one of those
    float bx0 = x0f, by0 = y0f, bx1 = x0f + 7.f, by1 = y0f + 7.f;

    __shared__ __attribute__((aligned(16))) float stage[33 * PW];   // 32 pairs + one that may be read, never used

    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);

    float T = 1.0f;
    f32x2 C01 = {0.f, 0.f};
    float C2 = 0.f;
    f32x2 CX[NX > 0 ? NX / 2 : 1];   // extra channels, two per register pair
#pragma unroll
    for (int j = 0; j < (NX > 0 ? NX / 2 : 1); j++) CX[j] = f32x2{0.f, 0.f};
    f32x4 xs0 = {1.f, 1.f, 1.f, 1.f}, xs1 = {1.f, 1.f, 1.f, 1.f};   // this view's factors for the extra channels
    if (NX > 0 && a.extra_scale != nullptr) {
        const float* sc = a.extra_scale + (size_t)view * NX;
        xs0 = f32x4{sc[0], sc[1], sc[2], sc[3]};
        if (NX > 4) xs1 = f32x4{sc[4], sc[5], sc[6], sc[7]};
    }
    uint32_t last_contributor = 0;
    uint32_t stop_at = 0;  // 1-based index of the entry that terminated this pixel
    bool crossed = false;  // this quadrant walked past a BWD_CHUNK boundary (wave-uniform)
    bool done = !inside;
    bool all_done = __all(done);
#ifdef GSR_STATS
    live_cnt = (unsigned)__popcll(__ballot(!done));
#endif
    if (!all_done) {
        int ax, ay, bx, by;
        live_box(__ballot(!done), ax, ay, bx, by);
        bx0 = x0f + (float)ax; by0 = y0f + (float)ay; bx1 = x0f + (float)bx; by1 = y0f + (float)by;
    }

    if (!all_done && total > 0) {
        const uint32_t* plist = a.point_list + range.x;
        // software pipeline: records of round r+1 and ids of round r+2 are in flight while round r is evaluated.
        // Lanes past the end of the list read entry total-1 again (always a valid address) and are masked by `valid`.
        const int last = total - 1;
        f32x4 c0, c1, n0, n1;
        f32x4 cx0 = {0.f, 0.f, 0.f, 0.f}, cx1 = cx0, nx0 = cx0, nx1 = cx0;   // extra channels of the entry
        float c2b, n2b;  // blue
        uint32_t id_cur, id_nxt, id_nn;
        {
            // prologue: round 0's records and round 1's ids, through the same asm path so that no compiler-tracked
            // load is pending when the loop is entered
            prefetch4(id_cur, plist + ((int)lane < total ? (int)lane : last));
            prefetch4(id_nxt, plist + (64 + (int)lane < total ? 64 + (int)lane : last));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(id_cur), "+v"(id_nxt)::"memory");
            const Splat* sp = a.splat + id_cur;
            prefetch16(c0, &sp->q0);
            prefetch16(c1, &sp->q1);
            prefetch4f(c2b, &sp->q2);
            if (NX > 0) {
                prefetch16(cx0, a.extra + (size_t)id_cur * x_nx);
                if (NX > 4) prefetch16(cx1, x_hi + (size_t)id_cur * x_nx);
                retire_prefetch_x(c0, c1, c2b, id_nxt, cx0, cx1);
            } else {
                retire_prefetch(c0, c1, c2b, id_nxt);
            }
        }
        for (int base = 0; base < total; base += 64) {
            {
                const Splat* sp = a.splat + id_nxt;
                prefetch16(n0, &sp->q0);
                prefetch16(n1, &sp->q1);
                prefetch4f(n2b, &sp->q2);
                if (NX > 0) {
                    prefetch16(nx0, a.extra + (size_t)id_nxt * x_nx);
                    if (NX > 4) prefetch16(nx1, x_hi + (size_t)id_nxt * x_nx);
                }
                const int i2 = base + 128 + (int)lane;
                prefetch4(id_nn, plist + (i2 < total ? i2 : last));
            }

            // backward work items are chunks of BWD_CHUNK list entries: leave this quadrant's state at the boundaries it crosses
            if (a.ckpt != nullptr && base != 0 && (base & ((1 << a.chunk_shift) - 1)) == 0 && (base >> a.chunk_shift) < BWD_MAX_CHUNKS) {
                const size_t slot = (size_t)(range.x >> a.chunk_shift) + (size_t)(base >> a.chunk_shift);
                a.ckpt[slot * 256 + q * 64 + lane] = make_float4(T, C01.x, C01.y, C2);
                crossed = true;
            }

#ifdef GSR_STATS
            FW_T(ts0);
            n_rounds++;
#endif
            // which of this round's 64 entries can reach alpha >= 1/255 somewhere in this quadrant?
            const bool valid = base + (int)lane < total;
            const bool touch = valid && may_touch_rect(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, bx0, by0, bx1, by1);
            uint64_t mask = __ballot(touch);

            if (mask != 0) {
                // stage the survivors: lane -> compacted slot -> (pair, half)
                const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                const uint32_t nsurv = (uint32_t)__popcll(mask);
                if (touch) {
                    const bool lone = stage_entry(stage, PW, slot, nsurv, c0, c1, c2b, (uint32_t)(base + (int)lane + 1));
                    if (NX > 0) {
                        f32x4 e0 = cx0, e1 = cx1;
                        e0 *= xs0;            // times +-1 (or any per-view factor): one multiply per staged value, not per pixel
                        if (NX > 4) e1 *= xs1;
                        // (a lone last survivor's extra values go to both halves of its pair, like its colour)
                        for (uint32_t h = slot & 1u; h < (lone ? 2u : (slot & 1u) + 1u); h++) {
                            float* px_ = stage + (slot >> 1) * PW + PAIR_WORDS + 2 * h;
                            *(f32x2*)(px_ + 0) = f32x2{e0.x, e0.y};
                            *(f32x2*)(px_ + 4) = f32x2{e0.z, e0.w};
                            if (NX > 4) {
                                *(f32x2*)(px_ + 8) = f32x2{e1.x, e1.y};
                                *(f32x2*)(px_ + 12) = f32x2{e1.z, e1.w};
                            }
                        }
                    }
                }
                // Pair p+1 is read from LDS while pair p is evaluated.  The loop body is written out twice with the two
                // register sets swapped, so no register moves are needed to rotate them.
                auto eval_pair = [&](const PairRec<NX>& r) {
#ifdef GSR_STATS
                    n_pairs_le8 += live_cnt <= 8; n_pairs_le16 += live_cnt <= 16; n_pairs_le32 += live_cnt <= 32;
#endif
                    // this pair has landed (DS returns in order); the 5 + NX / 2 reads of the next pair stay in flight
                    __builtin_amdgcn_s_waitcnt(NX == 0 ? 0xC57F : NX == 4 ? 0xC77F : 0xC97F);   // lgkmcnt(5 / 7 / 9)
                    // alpha of each entry; ae = alpha where the entry counts for this pixel, else 0.  The two entries
                    // share packed fp32 instructions (v_pk_*_f32).
                    const uint32_t eidx0 = r.pos.x, eidx1 = r.pos.y;
                    const f32x2 X = {r.xy.x, r.xy.y}, Y = {r.xy.z, r.xy.w}, A2 = {r.ab.x, r.ab.y}, B2 = {r.ab.z, r.ab.w};
                    const f32x2 C2p = {r.co.x, r.co.y}, O2 = {r.co.z, r.co.w};
                    const f32x2 dx = X - pixf_x, dy = Y - pixf_y;
                    const f32x2 power = -0.5f * (A2 * dx * dx + C2p * dy * dy) - B2 * dx * dy;
                    const f32x2 al = O2 * exp_nonpos2(power);
                    const float alpha0 = fminf(0.99f, al.x), alpha1 = fminf(0.99f, al.y);
                    const bool cnt0 = !done && !(power.x > 0.0f) && !(alpha0 < 1.0f / 255.0f);
                    const bool cnt1 = !done && !(power.y > 0.0f) && !(alpha1 < 1.0f / 255.0f);
                    const float ae0 = cnt0 ? alpha0 : 0.f, ae1 = cnt1 ? alpha1 : 0.f;
                    // Optimistic pass: assume no pixel of this wave terminates inside the pair.  T then only needs the
                    // products T*(1-ae) (ae = 0 multiplies by exactly 1), and because T never increases and every live
                    // pixel has T >= 1e-4, "an entry of the pair would have stopped a pixel" is just T_after < 1e-4.
                    // A pixel stops once, so the exact serial fallback runs for at most 64 pairs per wave per tile.
                    const float T1 = T * (1 - ae0), T2 = T1 * (1 - ae1);
                    if (!__any(T2 < 0.0001f)) {
                        // ae = 0 adds a zero, which leaves C unchanged bit-for-bit (C is never -0)
                        const f32x2 rg0 = {r.rg.x, r.rg.y}, rg1 = {r.rg.z, r.rg.w};
                        C01 += rg0 * ae0 * T;
                        C2 += r.b.x * ae0 * T;
                        C01 += rg1 * ae1 * T1;
                        C2 += r.b.y * ae1 * T1;
#pragma unroll
                        for (int j = 0; j < NX / 2; j++) {
                            CX[j] += f32x2{r.ex[j].x, r.ex[j].y} * ae0 * T;
                            CX[j] += f32x2{r.ex[j].z, r.ex[j].w} * ae1 * T1;
                        }
                        last_contributor = cnt0 ? eidx0 : last_contributor;
                        last_contributor = cnt1 ? eidx1 : last_contributor;
                        T = T2;
                    } else {
                        // Some pixel stops inside this pair.  Only the stopping lanes differ from the optimistic update:
                        // s0 = entry 0 stops the pixel (T*(1-a0) < 1e-4): neither entry is blended; s1 = entry 0 is blended and
                        // entry 1 stops it.  With those lanes' alphas forced to 0 the same accumulation applies to everybody
                        // (the blended values are the reference's c*alpha*T and T*(1-alpha), term for term).
                        const bool s0 = cnt0 && (T1 < 0.0001f);
                        const bool s1 = !s0 && cnt1 && (T2 < 0.0001f);
                        const bool b0 = cnt0 && !s0, b1 = cnt1 && !s0 && !s1;
                        const float be0 = b0 ? alpha0 : 0.f, be1 = b1 ? alpha1 : 0.f;
                        const float U1 = T * (1 - be0), U2 = U1 * (1 - be1);
                        const f32x2 rg0 = {r.rg.x, r.rg.y}, rg1 = {r.rg.z, r.rg.w};
                        C01 += rg0 * be0 * T;
                        C2 += r.b.x * be0 * T;
                        C01 += rg1 * be1 * U1;
                        C2 += r.b.y * be1 * U1;
#pragma unroll
                        for (int j = 0; j < NX / 2; j++) {
                            CX[j] += f32x2{r.ex[j].x, r.ex[j].y} * be0 * T;
                            CX[j] += f32x2{r.ex[j].z, r.ex[j].w} * be1 * U1;
                        }
                        last_contributor = b0 ? eidx0 : last_contributor;
                        last_contributor = b1 ? eidx1 : last_contributor;
                        T = U2;
                        stop_at = s0 ? eidx0 : (s1 ? eidx1 : stop_at);
                        done = done || s0 || s1;
                        const uint64_t live = __ballot(!done);
                        all_done = live == 0;
#ifdef GSR_STATS
                        live_cnt = (unsigned)__popcll(live);
#endif
                        if (!all_done) {   // a pixel stopped: the rounds still to come only need entries that reach the rest
                            int ax, ay, bx, by;
                            live_box(live, ax, ay, bx, by);
                            bx0 = x0f + (float)ax; by0 = y0f + (float)ay; bx1 = x0f + (float)bx; by1 = y0f + (float)by;
                        }
                    }
                };
                // The next pair is always read (the staging area has a spare pair, so reading one past the last is harmless)
                // and the loop ends on the pair count alone: when every pixel is done the count is set to 0.
                int npairs = (int)((nsurv + 1u) >> 1);
                int pair = 0;
#ifdef GSR_STATS
                FW_T(ts1);
                tw_stage += ts1 - ts0;
#endif
                PairRec<NX> ra = read_pair<NX>(stage, 0), rb;
                for (;;) {
                    rb = read_pair<NX>(stage, pair + 1);
                    eval_pair(ra);
                    if (all_done) npairs = 0;
                    if (++pair >= npairs) break;
                    ra = read_pair<NX>(stage, pair + 1);
                    eval_pair(rb);
                    if (all_done) npairs = 0;
                    if (++pair >= npairs) break;
                }
#ifdef GSR_STATS
                { FW_T(ts2); tw_eval += ts2 - ts1; n_pairs += (unsigned long long)pair; }
#endif
            }
#ifdef GSR_STATS
            FW_T(ts3);
#endif
            if (NX > 0) retire_prefetch_x(n0, n1, n2b, id_nn, nx0, nx1);
            else retire_prefetch(n0, n1, n2b, id_nn);
#ifdef GSR_STATS
            { FW_T(ts4); tw_wait += ts4 - ts3; }
#endif
            if (all_done) break;
            c0 = n0; c1 = n1; c2b = n2b;
            if (NX > 0) { cx0 = nx0; cx1 = nx1; }
            id_cur = id_nxt;
            id_nxt = id_nn;
        }
    }

#ifdef GSR_STATS
    if (lane == 0 && blockIdx.x < (unsigned)FW_REC) {
        FW_T(tw1);
        unsigned* r_ = g_fwd_rec[blockIdx.x];
        r_[0] = (unsigned)(tw1 - tw0); r_[1] = (unsigned)tw_wait; r_[2] = (unsigned)tw_stage; r_[3] = (unsigned)tw_eval;
        r_[4] = 1u; r_[5] = (unsigned)n_rounds; r_[6] = (unsigned)n_pairs;
        r_[7] = n_pairs_le8 | (n_pairs_le16 << 10) | (n_pairs_le32 << 20);   // (10 bits each: a wave evaluates < 1024 pairs)
        g_fwd_hw[blockIdx.x][0] = __builtin_amdgcn_s_getreg(63492) | (__builtin_amdgcn_s_getreg(63508) << 28);   // HW_ID, XCC_ID
        g_fwd_hw[blockIdx.x][1] = (unsigned)tw0;
    }
#endif
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
        if (NX > 0) {
            float* ox = a.out_extra + (size_t)view * NX * N + pix;
#pragma unroll
            for (int j = 0; j < NX / 2; j++) {
                ox[(size_t)(2 * j) * N] = CX[j].x + T * a.bg_extra[2 * j];
                ox[(size_t)(2 * j + 1) * N] = CX[j].y + T * a.bg_extra[2 * j + 1];
            }
        }
        if (crossed) {   // only a backward slice that starts at a recorded boundary reads the final accumulated colour
            a.accum[pix] = C01.x;
            a.accum[N + pix] = C01.y;
            a.accum[2 * N + pix] = C2;
        }
    }
}

// ---- half-quadrant mode: the forward of SINGLE-VIEW submissions -------------------------------------------------------------
// A single view's launch ends on a dozen waves that evaluate 500-650 pairs each, alone on their SIMDs, one instruction per ~10
// cycles, while the chip is a third occupied: what matters there is the number of instructions on a deep
// walk's critical path, not throughput.  In this mode a wave owns HALF a quadrant (8 x 4 pixels; eight single-wave workgroups
// per tile) and keeps its pixels TWICE: lanes 0-31 and lanes 32-63 hold the same 32 pixels.  A step covers FOUR consecutive
// surviving entries: the lower half of the wave evaluates alpha for entries 0 and 1 (one staged pair record), the upper half for
// entries 2 and 3 (the next record) -- the expensive part, 60 of a pair's 90 instructions, is done for four entries in the time
// of two -- then two v_permlane32_swap hand every lane all four alphas and both halves run the same blend of the four entries in
// order (redundantly: identical registers in both halves, no further exchange).  Per (pixel, entry) the arithmetic is the
// 8 x 8 kernel's, term for term: power, exp, alpha, T (1 - alpha), (c alpha) T, the 1/255 and 1e-4 tests, in list order; the
// images, final_T and n_contrib are bit-identical (tests compare both modes with the reference build).  Twice the waves gather and
// cull the same 64 entries per round, which is why batches (throughput-bound: the 12-view launch keeps 87 % of the wave slots
// busy) stay on the 8 x 8 kernel.
__device__ __forceinline__ void swap_halves2(float& a_lo, float& a_hi, float& b_lo, float& b_hi)
{
    // v_permlane32_swap x, y: lanes 32-63 of x <-> lanes 0-31 of y.  Called with x == y == v (two registers holding the same
    // per-lane value): afterwards x holds the LOWER half's v in all 64 lanes (lane i and lane i + 32 both have lane i's), y the upper
    // half's.  Through the compiler's builtin (gfx950 only, like this whole library: common.hpp refuses other targets), so that the
    // hazard recogniser pads the wait states around the swap (VALU write -> swap read and swap write -> VALU read) whatever the
    // surrounding code becomes; earlier rounds carried hand-counted s_nop in inline asm.
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_lo), __float_as_uint(a_hi), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(b_lo), __float_as_uint(b_hi), false, false);
    a_lo = __uint_as_float(a[0]);
    a_hi = __uint_as_float(a[1]);
    b_lo = __uint_as_float(b[0]);
    b_hi = __uint_as_float(b[1]);
}

__global__ __launch_bounds__(64) void k_render_forward_half(RenderArgs a)
{
#ifdef GSR_STATS
    FW_T(tw0);
    unsigned long long tw_wait = 0, tw_stage = 0, tw_eval = 0, n_rounds = 0, n_pairs = 0;   // (n_pairs: steps of four entries here)
#endif
    constexpr int PW = PAIR_WORDS;
    // workgroup b runs on XCD b % 8: the eight half-quadrants of one tile are b, b + 8, ..., b + 56 (one L2 fetch of list and records);
    // groups of 64 workgroups (8 tiles x 8 halves) are dealt to the views round-robin like the 8 x 8 kernel's groups of 32
    const uint32_t group = blockIdx.x >> 6;
    const uint32_t view = group % a.V;
    const uint32_t order_slot = (group / a.V) * 8u + (blockIdx.x & 7u);
    if (order_slot >= (uint32_t)a.num_tiles) return;
    a.ranges = at_view(a.ranges, a.iv_stride, view);
    a.tile_order = at_view(a.tile_order, a.iv_stride, view);
    a.final_T = at_view(a.final_T, a.iv_stride, view);
    a.n_contrib = at_view(a.n_contrib, a.iv_stride, view);
    a.tile_need = at_view(a.tile_need, a.iv_stride, view);
    a.accum = at_view(a.accum, a.iv_stride, view);
    a.point_list = at_view(a.point_list, a.b_stride, view);
    if (a.ckpt) a.ckpt = at_view(a.ckpt, a.b_stride, view);
    a.splat = at_view(a.splat, a.g_stride, view);
    a.out_color += (size_t)view * 3u * (size_t)a.W * (size_t)a.H;
    const uint32_t tile = a.tile_order[order_slot];
    const uint32_t sb = (blockIdx.x >> 3) & 7u, q = sb >> 1, hf = sb & 1u;
    const uint32_t lane = threadIdx.x, pl = lane & 31u, eg = lane >> 5;   // pixel of the half, entry group
    const uint32_t tx = tile % (uint32_t)a.gridx, ty = tile / (uint32_t)a.gridx;
    const uint32_t x0 = tx * TILE_X + (q & 1u) * 8u, y0 = ty * TILE_Y + (q >> 1) * 8u + hf * 4u;
    const uint32_t px = x0 + (pl & 7u), py = y0 + (pl >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const float x0f = (float)x0, y0f = (float)y0;
    float bx0 = x0f, by0 = y0f, bx1 = x0f + 7.f, by1 = y0f + 3.f;

    __shared__ __attribute__((aligned(16))) float stage[35 * PW];   // 32 pairs + a zero pair behind an odd count + read-ahead

    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);

    float T = 1.0f;
    f32x2 C01 = {0.f, 0.f};
    float C2 = 0.f;
    uint32_t last_contributor = 0;
    uint32_t stop_at = 0;
    bool crossed = false;
    bool done = !inside;
    bool all_done = __all(done);
    if (!all_done) {
        int ax, ay, bx, by;
        live_box(__ballot(!done) & 0xFFFFFFFFull, ax, ay, bx, by);   // (both halves of the wave hold the same 32 pixels)
        bx0 = x0f + (float)ax; by0 = y0f + (float)ay; bx1 = x0f + (float)bx; by1 = y0f + (float)by;
    }

    if (!all_done && total > 0) {
        const uint32_t* plist = a.point_list + range.x;
        const int last = total - 1;
        f32x4 c0, c1, n0, n1;
        float c2b, n2b;
        uint32_t id_cur, id_nxt, id_nn;
        {
            prefetch4(id_cur, plist + ((int)lane < total ? (int)lane : last));
            prefetch4(id_nxt, plist + (64 + (int)lane < total ? 64 + (int)lane : last));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(id_cur), "+v"(id_nxt)::"memory");
            const Splat* sp = a.splat + id_cur;
            prefetch16(c0, &sp->q0);
            prefetch16(c1, &sp->q1);
            prefetch4f(c2b, &sp->q2);
            retire_prefetch(c0, c1, c2b, id_nxt);
        }
        for (int base = 0; base < total; base += 64) {
            {
                const Splat* sp = a.splat + id_nxt;
                prefetch16(n0, &sp->q0);
                prefetch16(n1, &sp->q1);
                prefetch4f(n2b, &sp->q2);
                const int i2 = base + 128 + (int)lane;
                prefetch4(id_nn, plist + (i2 < total ? i2 : last));
            }
            // the backward's slice-boundary state: pixel (x, y) of the quadrant sits at index 8 y + x = 32 hf + pl, as in the 8 x 8 kernel
            if (a.ckpt != nullptr && base != 0 && (base & ((1 << a.chunk_shift) - 1)) == 0 && (base >> a.chunk_shift) < BWD_MAX_CHUNKS) {
                const size_t slot = (size_t)(range.x >> a.chunk_shift) + (size_t)(base >> a.chunk_shift);
                if (eg == 0) a.ckpt[slot * 256 + q * 64 + hf * 32u + pl] = make_float4(T, C01.x, C01.y, C2);
                crossed = true;
            }
#ifdef GSR_STATS
            FW_T(ts0);
            n_rounds++;
#endif
            const bool valid = base + (int)lane < total;
            const bool touch = valid && may_touch_rect(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, bx0, by0, bx1, by1);
            const uint64_t mask = __ballot(touch);
            if (mask != 0) {
                const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                const uint32_t nsurv = (uint32_t)__popcll(mask);
                const int npairs = (int)((nsurv + 1u) >> 1);
                if (touch) stage_entry(stage, PW, slot, nsurv, c0, c1, c2b, (uint32_t)(base + (int)lane + 1));
                // an odd number of PAIRS: the upper half's record of the last step is a pair of opacity 0 (alpha 0: never counted)
                if ((npairs & 1) && lane < (uint32_t)PW) stage[npairs * PW + lane] = 0.f;
                const int nsteps = (npairs + 1) >> 1;
                // a step's seven LDS reads: the half's own pair record (alpha of two entries) and, from both records of the step,
                // colours and list positions.  The next step's are issued before the current one is evaluated (two register sets
                // taking turns, like the 8 x 8 kernel's pairs); the read past the last step lands in the spare records.
                struct StepRec {
                    f32x4 xy, ab, co, rgA, rgB;
                    f32x2 bA, bB;
                    uint2 posA, posB;
                };
                auto load_step = [&](int st) {
                    StepRec r;
                    const float* po = stage + (2 * st + (int)eg) * PW;
                    const float* pa = stage + (2 * st) * PW;
                    r.xy = *(const f32x4*)(po + 0);
                    r.ab = *(const f32x4*)(po + 4);
                    r.co = *(const f32x4*)(po + 8);
                    r.rgA = *(const f32x4*)(pa + 12);
                    r.rgB = *(const f32x4*)(pa + PW + 12);
                    r.bA = *(const f32x2*)(pa + 16);
                    r.bB = *(const f32x2*)(pa + PW + 16);
                    r.posA = *(const uint2*)(pa + 18);
                    r.posB = *(const uint2*)(pa + PW + 18);
                    return r;
                };
                auto eval_step = [&](const StepRec& r) {
                    const f32x4 rgA = r.rgA, rgB = r.rgB;
                    const f32x2 bA = r.bA, bB = r.bB;
                    const uint2 posA = r.posA, posB = r.posB;
                    const f32x2 X = {r.xy.x, r.xy.y}, Y = {r.xy.z, r.xy.w}, A2 = {r.ab.x, r.ab.y}, B2 = {r.ab.z, r.ab.w};
                    const f32x2 C2p = {r.co.x, r.co.y}, O2 = {r.co.z, r.co.w};
                    const f32x2 dx = X - pixf_x, dy = Y - pixf_y;
                    const f32x2 power = -0.5f * (A2 * dx * dx + C2p * dy * dy) - B2 * dx * dy;
                    const f32x2 al = O2 * exp_nonpos2(power);
                    const float alpha0 = fminf(0.99f, al.x), alpha1 = fminf(0.99f, al.y);
                    // an entry counts for a pixel that is still live when the STEP begins; pixels that stop inside the step are
                    // handled by the exact path below.  alpha >= 1/255 > 0 for every counted entry, so "counted" <=> e != 0.
                    float ea = (!done && !(power.x > 0.0f) && !(alpha0 < 1.0f / 255.0f)) ? alpha0 : 0.f;
                    float eb = (!done && !(power.y > 0.0f) && !(alpha1 < 1.0f / 255.0f)) ? alpha1 : 0.f;
                    float e0 = ea, e2 = ea, e1 = eb, e3 = eb;
                    swap_halves2(e0, e2, e1, e3);   // e0 / e1: the lower half's two entries, e2 / e3: the upper half's, in every lane
                    const float T1 = T * (1 - e0), T2 = T1 * (1 - e1), T3 = T2 * (1 - e2), T4 = T3 * (1 - e3);
                    if (!__any(T4 < 0.0001f)) {
                        // no pixel of the wave stops inside the step (T never increases, every live pixel has T >= 1e-4)
                        C01 += f32x2{rgA.x, rgA.y} * e0 * T;
                        C2 += bA.x * e0 * T;
                        C01 += f32x2{rgA.z, rgA.w} * e1 * T1;
                        C2 += bA.y * e1 * T1;
                        C01 += f32x2{rgB.x, rgB.y} * e2 * T2;
                        C2 += bB.x * e2 * T2;
                        C01 += f32x2{rgB.z, rgB.w} * e3 * T3;
                        C2 += bB.y * e3 * T3;
                        last_contributor = e0 != 0.f ? posA.x : last_contributor;
                        last_contributor = e1 != 0.f ? posA.y : last_contributor;
                        last_contributor = e2 != 0.f ? posB.x : last_contributor;
                        last_contributor = e3 != 0.f ? posB.y : last_contributor;
                        T = T4;
                    } else {
                        // some pixel stops inside these four entries: the reference's sequence, entry by entry.  s: the entry
                        // would take T below 1e-4 -> the pixel stops and the entry is NOT blended; b: the entry is blended.
                        bool stopped = false;
                        const float es[4] = {e0, e1, e2, e3};
                        const f32x2 rgs[4] = {f32x2{rgA.x, rgA.y}, f32x2{rgA.z, rgA.w}, f32x2{rgB.x, rgB.y}, f32x2{rgB.z, rgB.w}};
                        const float bs[4] = {bA.x, bA.y, bB.x, bB.y};
                        const uint32_t ps[4] = {posA.x, posA.y, posB.x, posB.y};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool c = es[k] != 0.f && !stopped;
                            const float Tn = T * (1 - es[k]);
                            const bool sk = c && (Tn < 0.0001f);
                            const bool bk = c && !sk;
                            const float be = bk ? es[k] : 0.f;
                            C01 += rgs[k] * be * T;
                            C2 += bs[k] * be * T;
                            T = T * (1 - be);
                            last_contributor = bk ? ps[k] : last_contributor;
                            stop_at = sk ? ps[k] : stop_at;
                            stopped = stopped || sk;
                        }
                        done = done || stopped;
                        const uint64_t live = __ballot(!done) & 0xFFFFFFFFull;
                        all_done = live == 0;
                        if (!all_done) {
                            int ax, ay, bx, by;
                            live_box(live, ax, ay, bx, by);
                            bx0 = x0f + (float)ax; by0 = y0f + (float)ay; bx1 = x0f + (float)bx; by1 = y0f + (float)by;
                        }
                    }
                };
                int step = 0;
#ifdef GSR_STATS
                FW_T(ts1);
                tw_stage += ts1 - ts0;
#endif
                StepRec ra = load_step(0), rb;
                for (;;) {
                    rb = load_step(step + 1);
                    eval_step(ra);
                    if (all_done || ++step >= nsteps) break;
                    ra = load_step(step + 1);
                    eval_step(rb);
                    if (all_done || ++step >= nsteps) break;
                }
#ifdef GSR_STATS
                { FW_T(ts2); tw_eval += ts2 - ts1; n_pairs += (unsigned long long)step + 1ull; }
#endif
            }
#ifdef GSR_STATS
            FW_T(ts3);
#endif
            retire_prefetch(n0, n1, n2b, id_nn);
#ifdef GSR_STATS
            { FW_T(ts4); tw_wait += ts4 - ts3; }
#endif
            if (all_done) break;
            c0 = n0; c1 = n1; c2b = n2b;
            id_cur = id_nxt;
            id_nxt = id_nn;
        }
    }
#ifdef GSR_STATS
    // (record layout of the 8 x 8 kernel; word 6 counts steps, word 7 is the tile's list length: scripts/debug/fwd_half_tail.py)
    if (lane == 0 && blockIdx.x < (unsigned)FW_REC) {
        FW_T(tw1);
        unsigned* r_ = g_fwd_rec[blockIdx.x];
        r_[0] = (unsigned)(tw1 - tw0); r_[1] = (unsigned)tw_wait; r_[2] = (unsigned)tw_stage; r_[3] = (unsigned)tw_eval;
        r_[4] = 1u; r_[5] = (unsigned)n_rounds; r_[6] = (unsigned)n_pairs; r_[7] = (unsigned)total;
        g_fwd_hw[blockIdx.x][0] = __builtin_amdgcn_s_getreg(63492) | (__builtin_amdgcn_s_getreg(63508) << 28);   // HW_ID, XCC_ID
        g_fwd_hw[blockIdx.x][1] = (unsigned)tw0;
    }
#endif
    {
        uint32_t need = inside ? (done ? stop_at : (uint32_t)total) : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(need, d, 64);
            need = need > o ? need : o;
        }
        if (lane == 0 && need != 0) atomicMax(&a.tile_need[tile], need);
    }
    if (inside && eg == 0) {
        const size_t pix = (size_t)py * a.W + px, N = (size_t)a.W * a.H;
        a.final_T[pix] = T;
        a.n_contrib[pix] = last_contributor;
        a.out_color[pix] = C01.x + T * a.bg[0];
        a.out_color[N + pix] = C01.y + T * a.bg[1];
        a.out_color[2 * N + pix] = C2 + T * a.bg[2];
        if (crossed) {
            a.accum[pix] = C01.x;
            a.accum[N + pix] = C01.y;
            a.accum[2 * N + pix] = C2;
        }
    }
}

// views per submission up to which the forward runs in half-quadrant mode: GSR_FWD_HALF_V in the environment when the library is
// first used (default 1; 0: never), or gsr_set_forward_half_views() (tests compare the two kernels in one process)
static std::atomic<int> g_fwd_half_v{-1};
int forward_half_views(int set)
{
    if (set >= 0) g_fwd_half_v.store(set);
    int v = g_fwd_half_v.load();
    if (v < 0) {
        const char* e = getenv("GSR_FWD_HALF_V");
        v = e ? atoi(e) : 1;
        if (v < 0) v = 0;
        g_fwd_half_v.store(v);
    }
    return v;
}

int launch_render_forward(const Launch& L, const gsr_params& p, const Batch& B, const uint32_t* point_list, float* out_color,
                          bool with_ckpt, const ExtraChannels* X)
{
    RenderArgs a;
    a.ranges = B.iv.ranges;
    a.tile_order = B.iv.tile_order;
    a.point_list = point_list;
    a.splat = B.g.splat;
    a.W = p.W; a.H = p.H;
    a.gridx = (p.W + TILE_X - 1) / TILE_X;
    const int gridy = (p.H + TILE_Y - 1) / TILE_Y;
    a.bg = p.bg;
    a.out_color = out_color;
    a.final_T = B.iv.final_T;
    a.n_contrib = B.iv.n_contrib;
    a.tile_need = B.iv.tile_need;
    a.ckpt = with_ckpt ? B.b.ckpt : nullptr;
    a.accum = B.iv.accum;
    a.V = (uint32_t)B.V;
    a.g_stride = B.g_stride; a.b_stride = B.b_stride; a.iv_stride = B.iv_stride;
    const int T = a.gridx * gridy;
    a.num_tiles = T;
    a.chunk_shift = B.chunk_shift();
    // tile_need was cleared at the start of the frame (k_preprocess; the host on a retry / re-render)
    a.extra = nullptr; a.extra_scale = nullptr; a.bg_extra = nullptr; a.out_extra = nullptr; a.extra_vstride = 0;
    a.extra_hi = nullptr; a.extra_hi_vstride = 0;
    const dim3 grid((unsigned)div_up(T, 8) * 32u * (unsigned)B.V);
    if (X != nullptr && X->nx > 0) {
        a.extra = X->values; a.extra_scale = X->view_scale; a.bg_extra = X->bg; a.out_extra = X->out; a.extra_vstride = X->view_stride;
        a.extra_hi = X->values_hi; a.extra_hi_vstride = X->hi_view_stride;
        if (X->nx == 4) hipLaunchKernelGGL(k_render_forward<4>, grid, dim3(64), 0, L.stream, a);
        else hipLaunchKernelGGL(k_render_forward<8>, grid, dim3(64), 0, L.stream, a);
    } else if (B.V <= forward_half_views(-1)) {
        hipLaunchKernelGGL(k_render_forward_half, dim3((unsigned)div_up(T, 8) * 64u * (unsigned)B.V), dim3(64), 0, L.stream, a);
    } else {
        hipLaunchKernelGGL(k_render_forward<0>, grid, dim3(64), 0, L.stream, a);
    }
    return check_launch(L, "render_forward");
}

}  // namespace gsr
