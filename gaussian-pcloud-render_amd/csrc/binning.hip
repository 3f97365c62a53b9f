// binning.hip -- (tile, Gaussian) pair emission with its own prefix sum, per-tile list ranges, render launch order.
//
// Pair emission follows reference CR/rasterizer_impl.cu:70-111 (duplicateWithKeys): every visible
// Gaussian emits one pair per tile of its rectangle, rows outer / columns inner.  Differences in
// structure (results identical after the sort, see sort.hip):
//   - Gaussians are visited in depth order, the key is just the tile id (u32 / u16);
//   - the prefix sum of the touched-tile counts (the reference's cub::DeviceScan::InclusiveSum, :277) is done by the
//     emission kernel itself: a workgroup publishes the pair count of its 256 Gaussians and adds up the counts its
//     predecessors published (they are dispatched earlier, so they are running or done), which removes the scan
//     launches and the host read-back between them and the emission: num_rendered stays on the device;
//   - emission is wave-cooperative: a wave owns 64 consecutive Gaussians, and its lanes write the
//     wave's whole contiguous output range 64 slots at a time (coalesced 4-B stores) instead of one
//     thread writing a 10..40-slot run on its own;
//   - pairs beyond the binning arena's capacity are counted but not written (the host retries with a larger arena).
// Tile ranges (reference CR/rasterizer_impl.cu:116-138 identifyTileRanges + the memset at :310): one thread per tile finds
// its list in the sorted keys by binary search -- T searches instead of a pass over all R keys.
#include "common.hpp"
#include "tile_cull.hpp"

namespace gsr {

__device__ __forceinline__ uint64_t agent_load(const uint64_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_store(uint64_t* p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef GSR_STATS
// instrumentation build only: per-phase time of the emission workgroups (wall_clock64 ticks of 10 ns, summed over workgroups):
// 0 ticket, 1 order + rectangle gather + scan, 2 publish -> look-back done, 3 emission; out8[4] = workgroups seen
// one record per workgroup (no atomics: 190 K same-address atomics quadruple the kernel's time and end up being what is measured)
constexpr int DUP_REC = 65536;
__device__ unsigned g_dup_rec[DUP_REC][4];
#define DUP_T(var) const unsigned long long var = wall_clock64()
#define DUP_ADD(i, v) do { if (threadIdx.x == 0) { const unsigned r_ = blockIdx.y * gridDim.x + s_ticket; if (r_ < DUP_REC) g_dup_rec[r_][i] = (unsigned)(v); } } while (0)
int debug_dup_times(unsigned long long* out8, int reset)
{
    static unsigned host[DUP_REC][4];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dup_rec), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    for (int r = 0; r < DUP_REC; r++) {
        if (host[r][0] == 0 && host[r][1] == 0) continue;
        for (int i = 0; i < 4; i++) out8[i] += host[r][i];
        out8[4]++;
    }
    if (reset) {
        for (int r = 0; r < DUP_REC; r++) for (int i = 0; i < 4; i++) host[r][i] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_dup_rec), host, sizeof(host)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define DUP_T(var) do { } while (0)
#define DUP_ADD(i, v) do { } while (0)
#endif

struct DupArgs {
    int P;
    uint32_t gridx;
    const uint32_t* order[2];        // ids in depth order: the depth sort's ping-pong buffers ...
    const uint32_t* sortctl;         // ... and how many of its passes ran this frame (an odd count leaves the result in buffer 1)
    const Splat* splat;              // q3: what the Gaussian emits + the reference's pair count, as written by k_preprocess (common.hpp)
    uint64_t* dup_status;            // [nblk] zeroed before the launch; pair count + 1 once a workgroup has published
    uint64_t* counters;
    size_t g_stride;
    void* keys;                      // NULL: count only
    uint32_t* vals;
    size_t b_stride;
    int64_t cap;
    uint64_t* host_land;             // [V][4] host memory mapped into the device: num_rendered, trap flag, stall flag, - (api.hip)
    int tickets;                     // common.hpp block_tickets
};

template <typename KeyT>
__global__ __launch_bounds__(DUP_THREADS) void k_duplicate(DupArgs a)
{
    __shared__ uint32_t s_off[4][64];
    __shared__ uint32_t s_id[4][64];
    __shared__ uint2 s_rect[4][64];
    __shared__ uint32_t s_sp2[4][64];
    __shared__ uint2 s_pre[4][64];
    __shared__ uint32_t s_flag[4][64];
    __shared__ uint64_t s_wave[4];
    __shared__ uint64_t s_ref[4];
    __shared__ uint64_t s_base;
    __shared__ uint32_t s_ticket;
    const uint32_t view = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t* status = at_view(a.dup_status, a.g_stride, view);
    // Which DUP_BLOCK Gaussians this workgroup takes is decided by a ticket drawn when it STARTS (a.tickets, the default): a workgroup
    // with a lower number has started earlier, so the look-back below only ever waits for workgroups that are already running or
    // done, whatever order the hardware dispatches blockIdx in.  (a.tickets == 0: by blockIdx, common.hpp block_tickets.)
    // The ticket word follows the status words.
    const uint32_t nblk = gridDim.x;
    DUP_T(t0);
    if (threadIdx.x == 0) s_ticket = a.tickets ? (uint32_t)atomicAdd((unsigned long long*)&status[DUP_COPIES * nblk], 1ull) : blockIdx.x;
    __syncthreads();
    DUP_T(t1);
    DUP_ADD(0, t1 - t0);
    const uint32_t blk = s_ticket;
    const uint32_t npass = depth_sort_passes(at_view(a.sortctl, a.g_stride, view)[SORTCTL_BITS]);
    const uint32_t* order = at_view((npass & 1u) ? a.order[1] : a.order[0], a.g_stride, view);
    const Splat* splat = at_view(a.splat, a.g_stride, view);

    // A wave owns DUP_G groups of 64 consecutive Gaussians (positions in depth order); the G gathers are independent and in
    // flight together, and the ticket and the look-back -- round trips to the fabric that bound a workgroup's life -- are paid
    // once per DUP_G * 256 Gaussians.
    uint32_t cnt[DUP_G], id[DUP_G], sp2[DUP_G];
    uint2 rc[DUP_G];
    uint64_t ref_cnt = 0;   // pairs of the reference's (unclipped) rectangles: only their total is needed (num_rendered)
    const int slot0 = (int)(blk * DUP_BLOCK + w * (64 * DUP_G) + lane);
#pragma unroll
    for (int g = 0; g < DUP_G; g++) {
        const int slot = slot0 + g * 64;
        id[g] = slot < a.P ? order[slot] : 0u;
    }
#pragma unroll
    for (int g = 0; g < DUP_G; g++) {
        cnt[g] = 0;
        rc[g] = make_uint2(0, 0);
        sp2[g] = 0;
        if (slot0 + g * 64 < a.P) {
            const float4 q3 = splat[id[g]].q3;   // one 16-B gather: tile rectangle + tile count
            const uint32_t w3 = __float_as_uint(q3.w);
            rc[g] = make_uint2(__float_as_uint(q3.x), __float_as_uint(q3.y));
            sp2[g] = __float_as_uint(q3.z);
            if (w3 & SPANS_FLAG) {
                // row spans: one byte per tile row (first column | columns << 4); the pair count is the sum of the high nibbles
                const uint32_t a4 = (rc[g].y >> 4) & 0x0F0F0F0Fu, b4 = (sp2[g] >> 4) & 0x0F0F0F0Fu;
                const uint32_t pa = a4 * 0x01010101u;              // byte i = columns of rows 0..i (<= 60)
                const uint32_t pb = b4 * 0x01010101u + (pa >> 24) * 0x01010101u;   // rows 0..4+i (<= 120)
                cnt[g] = pb >> 24;
                id[g] |= SPANS_FLAG;                               // travels with the id (ids are < 2^31)
            } else {
                cnt[g] = sp2[g];
            }
            ref_cnt += w3 & ~SPANS_FLAG;
        }
    }
    // ---- prefix sum: inside the wave (group after group), over the workgroup's waves, over the preceding workgroups ----
    // a Gaussian touches < 2^28 tiles and 64 of them < 2^34: sums are carried in 64 bits
    uint64_t excl[DUP_G];          // exclusive prefix of this Gaussian inside the wave
    uint64_t run = 0;              // pairs of the wave's groups so far (wave uniform)
#pragma unroll
    for (int g = 0; g < DUP_G; g++) {
        uint64_t inc64 = cnt[g];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t n = (uint64_t)__shfl_up((unsigned long long)inc64, d, 64);
            if (lane >= (uint32_t)d) inc64 += n;
        }
        excl[g] = run + inc64 - cnt[g];
        run += (uint64_t)__shfl((unsigned long long)inc64, 63, 64);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ref_cnt += (uint64_t)__shfl_xor((unsigned long long)ref_cnt, d, 64);
    if (lane == 63) { s_wave[w] = run; s_ref[w] = ref_cnt; }
    __syncthreads();
    uint64_t wave_base = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if ((uint32_t)i < w) wave_base += s_wave[i];
    const uint64_t block_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    const uint64_t block_ref = s_ref[0] + s_ref[1] + s_ref[2] + s_ref[3];
    // One 64-bit status word carries both counts of the ticket's 1 024 Gaussians: pairs emitted + 1 in the low half, pairs of
    // the reference's rectangles in the high half.  A count that does not fit its half saturates: the reference total then
    // exceeds the int the reference keeps num_rendered in, which the host reports as an error before anything uses the lists.
    const uint64_t sat = 0xFFFFFFFEull;
    const uint64_t word = ((block_total < sat ? block_total : sat) + 1ull) | ((block_ref < 0xFFFFFFFFull ? block_ref : 0xFFFFFFFFull) << 32);
    // Published DUP_COPIES times: what workgroups of one launch tell each other goes around the L2s (they are not coherent across
    // XCDs), and accesses to ONE line of that kind complete at about one per 50 ns -- the first status lines are read by every
    // workgroup behind them (782 at 800 K Gaussians: 40 us, the whole kernel).  A reader takes copy (its number mod DUP_COPIES).
    if (threadIdx.x < DUP_COPIES) agent_store(&status[threadIdx.x * nblk + blk], word);
    DUP_T(t2);
    DUP_ADD(1, t2 - t1);
    // look-back: every preceding workgroup's count (published as count + 1; 0 = not yet).  They were dispatched before
    // this one, so waiting for them cannot deadlock.
    uint64_t part = 0, part_ref = 0;
    // LB words per thread are requested back to back (one round trip to the fabric instead of LB); a word that has not been
    // published yet is polled afterwards.
    constexpr uint32_t LB = 4;
    const uint64_t* mine = status + (size_t)(blk % DUP_COPIES) * nblk;
    for (uint32_t b0 = threadIdx.x; b0 < blk; b0 += DUP_THREADS * LB) {
        uint64_t v[LB];
#pragma unroll
        for (uint32_t k = 0; k < LB; k++) {
            const uint32_t b = b0 + k * DUP_THREADS;
            v[k] = b < blk ? agent_load(&mine[b]) : 1ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < LB; k++) {
            uint32_t spins = 0;
            while ((uint32_t)v[k] == 0u) {
                v[k] = agent_load(&mine[b0 + k * DUP_THREADS]);
                if ((uint32_t)v[k] == 0u && ++spins > (1u << 24)) {   // seconds: something is badly wrong; report instead of hanging the GPU
                    at_view(a.counters, a.g_stride, view)[CNT_STALL] = 1;
                    a.host_land[4 * view + CNT_STALL] = 1;
                    v[k] = 1;
                }
            }
            part += (v[k] & 0xFFFFFFFFull) - 1;
            part_ref += v[k] >> 32;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        part += (uint64_t)__shfl_xor((unsigned long long)part, d, 64);
        part_ref += (uint64_t)__shfl_xor((unsigned long long)part_ref, d, 64);
    }
    __syncthreads();   // s_wave / s_ref are reused
    if (lane == 0) { s_wave[w] = part; s_ref[w] = part_ref; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t base = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = base;
        if (blk == nblk - 1) {
            // the frame's pair counts: for the kernels that follow (device memory) and for the host, which reads them from mapped
            // memory once this kernel has completed -- no copy command in the stream.  CNT_NUM_RENDERED = pairs in the lists (what
            // the sort, the range search and the arena capacity are about); LAND_NUM_REFERENCE = pairs of the reference's
            // rectangles, the num_rendered the API reports (equal unless footprint clipping is on)
            uint64_t* cnt = at_view(a.counters, a.g_stride, view);
            const uint64_t ref_total = s_ref[0] + s_ref[1] + s_ref[2] + s_ref[3] + block_ref;
            cnt[CNT_NUM_RENDERED] = base + block_total;
            cnt[CNT_NUM_REFERENCE] = ref_total;
            a.host_land[4 * view + CNT_NUM_RENDERED] = base + block_total;
            a.host_land[4 * view + CNT_TRAP] = cnt[CNT_TRAP];
            a.host_land[4 * view + LAND_NUM_REFERENCE] = ref_total;
            __threadfence_system();
        }
    }
    __syncthreads();
    DUP_T(t3);
    DUP_ADD(2, t3 - t2);
    if (a.keys == nullptr) return;   // count-only call (no binning arena yet)
    KeyT* keys = at_view((KeyT*)a.keys, a.b_stride, view);
    uint32_t* vals = at_view(a.vals, a.b_stride, view);
    const uint64_t wave_off = s_base + wave_base;

#pragma unroll 1
    for (int g = 0; g < DUP_G; g++) {
        // select group g's registers without dynamic indexing
        uint32_t cnt_g = cnt[0], id_g = id[0], sp2_g = sp2[0];
        uint2 rc_g = rc[0];
        uint64_t ex_g = excl[0];
#pragma unroll
        for (int k = 1; k < DUP_G; k++)
            if (g == k) { cnt_g = cnt[k]; id_g = id[k]; rc_g = rc[k]; ex_g = excl[k]; sp2_g = sp2[k]; }
        const uint64_t off64 = wave_off + ex_g;   // exclusive prefix of this Gaussian
        // the group's output range [grp_begin, grp_end); positions at or beyond the capacity are dropped
        const uint64_t grp_begin64 = (uint64_t)__shfl((unsigned long long)off64, 0, 64);
        const uint64_t grp_end64 = (uint64_t)__shfl((unsigned long long)(off64 + cnt_g), 63, 64);
        if (grp_begin64 >= (uint64_t)a.cap) break;
        const uint32_t grp_begin = (uint32_t)grp_begin64;
        const uint32_t grp_end = grp_end64 < (uint64_t)a.cap ? (uint32_t)grp_end64 : (uint32_t)a.cap;   // cap < 2^32
        // offsets relative to the group's begin (fit 32 bits: 64 Gaussians x < 2^28 tiles is rejected by the host long before)
        // The Gaussians that emit at least one pair are compacted (their offsets are then strictly increasing), and every
        // chunk of 64 output slots finds its owners without a search: the owners whose run STARTS inside the chunk raise a
        // flag at their first slot (distinct slots: plain LDS stores), one ballot of the flags gives the chunk's start mask,
        // and slot l belongs to Gaussian number (starts before the chunk) + (starts at slots <= l) - 1.  Two LDS round trips
        // per 64 pairs instead of the eight of a binary search over the offsets.
        const uint64_t nzb = __ballot(cnt_g != 0);
        const uint32_t n_nz = (uint32_t)__popcll(nzb);
        const uint32_t jslot = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzb, 0u));
        __builtin_amdgcn_wave_barrier();   // the previous group's reads of the staging arrays are done
        if (cnt_g != 0) {
            s_off[w][jslot] = (uint32_t)(off64 - grp_begin64);
            s_id[w][jslot] = id_g;
            s_rect[w][jslot] = rc_g;
            s_sp2[w][jslot] = sp2_g;
            if (id_g & SPANS_FLAG) {   // running column counts of the rows, one byte each: slot -> row without walking the rows
                const uint32_t a4 = (rc_g.y >> 4) & 0x0F0F0F0Fu, b4 = (sp2_g >> 4) & 0x0F0F0F0Fu;
                const uint32_t pa = a4 * 0x01010101u;
                s_pre[w][jslot] = make_uint2(pa, b4 * 0x01010101u + (pa >> 24) * 0x01010101u);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t my_start = lane < n_nz ? s_off[w][lane] : 0xFFFFFFFFu;   // first slot of the lane-th emitting Gaussian
        const uint32_t range = grp_end - grp_begin;
        uint32_t before = 0;                                                    // runs that start before the chunk
        for (uint32_t c = 0; c < range; c += 64) {
            s_flag[w][lane] = 0u;
            const uint32_t d = my_start - c;
            if (d < 64u) s_flag[w][d] = 1u;
            __builtin_amdgcn_wave_barrier();
            const uint32_t mine = s_flag[w][lane];
            const uint64_t starts = __ballot(mine != 0u);
            const uint32_t owner = before + mine - 1u +
                                   __builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u));
            before += (uint32_t)__popcll(starts);
            __builtin_amdgcn_wave_barrier();
            const uint32_t rel = c + lane;
            if (rel < range) {
                const uint2 r = s_rect[w][owner];
                const uint32_t oid = s_id[w][owner];
                const uint32_t minx = r.x & 0xFFFFu, miny = r.x >> 16;
                const uint32_t t = rel - s_off[w][owner];
                uint32_t row, col;
                if (oid & SPANS_FLAG) {
                    // t-th tile of the row spans: the row is the number of running counts <= t (a byte-wise compare: count byte
                    // + 127 - t has its top bit set exactly when the count exceeds t; no carries, the counts are <= 120)
                    const uint2 pre = s_pre[w][owner];
                    const uint32_t tt = (127u - t) * 0x01010101u;
                    const uint32_t xa = (pre.x + tt) & 0x80808080u, xb = (pre.y + tt) & 0x80808080u;
                    row = 8u - (uint32_t)__popc(xa) - (uint32_t)__popc(xb);
                    const uint32_t rp = row - 1u;                                     // (row 0: nothing before it)
                    const uint32_t before = row == 0u ? 0u : (((rp < 4u ? pre.x : pre.y) >> (8u * (rp & 3u))) & 0xFFu);
                    const uint32_t byte = ((row < 4u ? r.y : s_sp2[w][owner]) >> (8u * (row & 3u))) & 0xFFu;
                    col = (byte & 15u) + (t - before);
                } else {
                    const uint32_t width = (r.y & 0xFFFFu) - minx;
                    row = t / width;
                    col = t - row * width;
                }
                const uint32_t p = grp_begin + rel;
                keys[p] = (KeyT)((miny + row) * a.gridx + (minx + col));
                vals[p] = oid & ~SPANS_FLAG;
            }
        }
    }
#ifdef GSR_STATS
    if (w == 0) { DUP_T(t4); DUP_ADD(3, t4 - t3); }
#endif
}

int launch_duplicate(const Launch& L, int P, const Batch& B, int gridx, bool key16, uint64_t* host_land)
{
    DupArgs a;
    a.P = P;
    a.gridx = (uint32_t)gridx;
    a.order[0] = B.g.dval[0];
    a.order[1] = B.g.dval[1];
    a.sortctl = B.g.sortctl;
    a.splat = B.g.splat;
    a.dup_status = B.g.dup_status;
    a.counters = B.g.counters;
    a.g_stride = B.g_stride;
    a.keys = B.b.key[0];
    a.vals = B.b.val[0];
    a.b_stride = B.b_stride;
    a.cap = B.b.key[0] ? B.b.cap : 0;
    a.host_land = host_land;
    a.tickets = block_tickets(-1);
    const dim3 grid((unsigned)div_up(P, DUP_BLOCK), B.V);
    if (key16)
        hipLaunchKernelGGL(k_duplicate<uint16_t>, grid, dim3(DUP_THREADS), 0, L.stream, a);
    else
        hipLaunchKernelGGL(k_duplicate<uint32_t>, grid, dim3(DUP_THREADS), 0, L.stream, a);
    return check_launch(L, "duplicate");
}

// ---- tile ranges by search -------------------------------------------------------------------------------
// ranges[t] = [lower_bound(keys, t), lower_bound(keys, t + 1)), or (0, 0) for a tile without pairs (what the reference's
// memset leaves).  Two levels: every RANGE_SAMPLE-th key is staged in LDS and searched there, then one block of
// RANGE_SAMPLE keys is searched in global memory (12 dependent loads instead of 24).  Every tile is written, so the array
// needs no clearing.
constexpr int RANGE_SAMPLE = 4096;
constexpr int RANGE_MAX_SAMPLES = 8192;

template <typename KeyT>
__device__ __forceinline__ uint32_t lower_bound_keys(const KeyT* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ samples,
                                                     uint32_t ns, uint32_t t)
{
    uint32_t lo = 0, hi = n;   // answer in [lo, hi]
    if (ns != 0) {
        // largest sample j with keys[j * RANGE_SAMPLE] < t  ->  the answer lies in (j * S, (j + 1) * S]
        uint32_t a = 0, b = ns;   // first sample >= t is in [a, b]
        while (a < b) {
            const uint32_t m = (a + b) >> 1;
            if (samples[m] < t) a = m + 1; else b = m;
        }
        if (a == 0) return 0;     // even keys[0] >= t
        lo = (a - 1) * RANGE_SAMPLE + 1;
        hi = a < ns ? a * RANGE_SAMPLE : n;
    }
    while (lo < hi) {
        const uint32_t m = (lo + hi) >> 1;
        if ((uint32_t)keys[m] < t) lo = m + 1; else hi = m;
    }
    return lo;
}

// A workgroup of 256 threads serves 255 tiles: thread i finds the lower bound of tile t0 + i, thread 255 the one that closes
// the last tile's list; a tile's range is (its bound, its neighbour's) -- one search per thread instead of two.
constexpr int RANGE_TILES_PER_WG = 255;
template <typename KeyT>
__global__ __launch_bounds__(256) void k_tile_ranges(const uint64_t* __restrict__ counters, size_t g_stride, int64_t cap,
                                                     const KeyT* __restrict__ keys, size_t b_stride, uint2* __restrict__ ranges,
                                                     size_t iv_stride, int T)
{
    __shared__ uint32_t samples[RANGE_MAX_SAMPLES];
    __shared__ uint32_t bound[256];
    const uint32_t view = blockIdx.y;
    const uint64_t n64 = at_view(counters, g_stride, view)[CNT_NUM_RENDERED];
    const uint32_t n = (uint32_t)(n64 < (uint64_t)cap ? n64 : (uint64_t)cap);
    keys = at_view(keys, b_stride, view);
    ranges = at_view(ranges, iv_stride, view);
    uint32_t ns = (n + RANGE_SAMPLE - 1) / RANGE_SAMPLE;
    if (ns > RANGE_MAX_SAMPLES) ns = 0;   // enormous lists: plain binary search
    for (uint32_t j = threadIdx.x; j < ns; j += 256) samples[j] = (uint32_t)keys[(size_t)j * RANGE_SAMPLE];
    __syncthreads();
    const int t = blockIdx.x * RANGE_TILES_PER_WG + threadIdx.x;
    bound[threadIdx.x] = (t < T && n != 0) ? lower_bound_keys(keys, n, samples, ns, (uint32_t)t) : n;   // tile ids are < T: bound(T) = n
    __syncthreads();
    if (threadIdx.x >= RANGE_TILES_PER_WG || t >= T) return;
    const uint32_t first = bound[threadIdx.x], last = bound[threadIdx.x + 1];
    ranges[t] = first < last ? make_uint2(first, last) : make_uint2(0u, 0u);
}

int launch_tile_ranges(const Launch& L, const Batch& B, const uint32_t* sorted_keys, int T, bool key16)
{
    const dim3 grid((unsigned)div_up(T, RANGE_TILES_PER_WG), B.V);
    if (key16)
        hipLaunchKernelGGL(k_tile_ranges<uint16_t>, grid, dim3(256), 0, L.stream, B.g.counters, B.g_stride, B.b.cap,
                           (const uint16_t*)sorted_keys, B.b_stride, B.iv.ranges, B.iv_stride, T);
    else
        hipLaunchKernelGGL(k_tile_ranges<uint32_t>, grid, dim3(256), 0, L.stream, B.g.counters, B.g_stride, B.b.cap, sorted_keys,
                           B.b_stride, B.iv.ranges, B.iv_stride, T);
    return check_launch(L, "tile_ranges");
}

// ---- render launch order: tiles by descending work estimate (128 quarter-octave buckets) -------------
// Dispatching the heavy tiles first keeps the tail of the forward render kernel short.  Any permutation is correct.
// What a tile costs is the number of entries its pixels evaluate before they saturate, which is NOT its list length: a list
// consumed to its end costs its length, but the denser a tile, the sooner its pixels saturate -- on the benchmark views the waves
// that evaluate most (150-300 steps of four entries) sit on lists of 1 500-4 000 entries, lists of 12 000+ cost half of that
// (scripts/debug/fwd_half_tail.py).  Ordered by length, those medium lists queue behind every longer one and start when the launch is
// half over: a single view's forward ended on them at 0.22 ms although no wave lives longer than 0.15.  The estimate is the length up
// to a knee and falls slowly beyond it: w(L) = L for L <= K, K (K / L)^e above (K = 2 048, e = 0.3: the means per length bucket of
// two benchmark views fall like e = 0.6, but the lists that are walked to their end -- silhouette tiles, some pixel never saturates --
// are the heavy ones of their length and should not queue late; GSR_ORDER_KNEE / GSR_ORDER_EXP in the environment, knee 0 = plain
// length).  Measured (leases r5t / r5u): single-view forward 0.221-0.223 -> 0.209-0.217 ms, in a 12-view batch 0.123 -> 0.119-0.122 per
// view; the instrumented launch 211 -> 192 us against 131 us of summed wave time per slot: what is left is the spread INSIDE a length
// class (a wave lives up to 0.16 ms of a 0.19 ms launch), which only the render itself could tell.
constexpr int ORD_BUCKETS = 128;

__device__ __forceinline__ uint32_t work_bucket(uint32_t len)
{
    if (len == 0) return ORD_BUCKETS - 1;
    const uint32_t msb = 31u - (uint32_t)__builtin_clz(len);          // 0..31
    const uint32_t frac = msb >= 2 ? (len >> (msb - 2)) & 3u : 0u;    // next two bits
    const uint32_t rank = msb * 4u + frac;                            // larger = more work, 0..127
    return (ORD_BUCKETS - 2) - (rank < ORD_BUCKETS - 2 ? rank : ORD_BUCKETS - 2);
}

// One 1024-thread workgroup per view: LDS histogram, scan, LDS cursors.  (T is 8 160 at 1080p, 32 400 at 4K.)  Lanes of a
// wave that fall in the same bucket are aggregated with a ballot so the thousands of empty tiles, which all share
// one bucket, cost one LDS atomic per wave instead of 64 serialized ones.
__device__ __forceinline__ uint32_t work_estimate(uint32_t len, float knee, float expo)
{
    if (knee <= 0.f || (float)len <= knee) return len;
    // (v_log_f32 / v_exp_f32: a scheduling hint needs no accurate pow -- the library call made this kernel 10 us longer)
    return (uint32_t)(knee * __builtin_amdgcn_exp2f(expo * __builtin_amdgcn_logf(knee / (float)len)));
}

__global__ __launch_bounds__(1024) void k_tile_order(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ tile_order,
                                                     size_t iv_stride, float knee, float expo)
{
    ranges = at_view(ranges, iv_stride, blockIdx.x);
    tile_order = at_view(tile_order, iv_stride, blockIdx.x);
    __shared__ uint32_t cnt[ORD_BUCKETS];
    __shared__ uint32_t cur[ORD_BUCKETS];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    if (threadIdx.x < ORD_BUCKETS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int T_pad = (T + 1023) / 1024 * 1024;
    for (int pass = 0; pass < 2; pass++) {
        for (int t = threadIdx.x; t < T_pad; t += 1024) {
            const bool ok = t < T;
            uint32_t b = 0;
            if (ok) b = work_bucket(work_estimate(ranges[t].y - ranges[t].x, knee, expo));
            const bool empty = ok && b == ORD_BUCKETS - 1;
            const uint64_t em = __ballot(empty);
            uint32_t slot = 0;
            if (pass == 0) {
                if (empty) { if ((em & lt_mask) == 0) atomicAdd(&cnt[b], (uint32_t)__popcll(em)); }
                else if (ok) atomicAdd(&cnt[b], 1u);
            } else {
                uint32_t basev = 0;
                const int leader = em ? (int)__builtin_ctzll(em) : 0;
                if (empty && (int)lane == leader) basev = atomicAdd(&cur[b], (uint32_t)__popcll(em));
                basev = __shfl(basev, leader, 64);
                if (empty) slot = basev + (uint32_t)__popcll(em & lt_mask);
                else if (ok) slot = atomicAdd(&cur[b], 1u);
                if (ok) tile_order[slot] = (uint32_t)t;
            }
        }
        __syncthreads();
        if (pass == 0) {
            if (threadIdx.x < 64) {   // exclusive scan of 128 counts by one wave, two per lane
                const uint32_t a0 = cnt[2 * lane], a1 = cnt[2 * lane + 1];
                uint32_t inc = a0 + a1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t n = __shfl_up(inc, d, 64);
                    if (lane >= (uint32_t)d) inc += n;
                }
                cur[2 * lane] = inc - a0 - a1;
                cur[2 * lane + 1] = inc - a1;
            }
            __syncthreads();
        }
    }
}

// ---- ranges + launch order in one workgroup per view, from the pairs-per-tile counts (look-back tile sort) ------------------
// The single-read histogram kernel of the tile sort (sort.hip k_tile_hist) leaves the number of pairs of every tile: the ranges
// are their exclusive prefix sums (what the reference finds by comparing neighbouring sorted keys, CR/rasterizer_impl.cu:116-138,
// and the search kernel above by 12 dependent loads per tile), and the launch order needs nothing else either: one launch of one
// workgroup per view instead of two, off the sorted keys altogether.  A thread owns ceil(T / 1024) consecutive tiles.
__global__ __launch_bounds__(1024) void k_ranges_order(int T, const uint32_t* tile_count, uint2* __restrict__ ranges,
                                                       uint32_t* __restrict__ tile_order, size_t iv_stride, float knee, float expo)
{
    tile_count = at_view(tile_count, iv_stride, blockIdx.x);
    ranges = at_view(ranges, iv_stride, blockIdx.x);
    tile_order = at_view(tile_order, iv_stride, blockIdx.x);
    // the counts are staged in LDS with ONE round of coalesced loads (they were accumulated by memory-side atomics: every global
    // read of them is a round trip to the fabric, and the three walks below made 3 x ceil(T / 1024) dependent ones: 19 us)
    extern __shared__ uint32_t tc_lds[];
    for (int t = threadIdx.x; t < T; t += 1024) tc_lds[t] = tile_count[t];
    __syncthreads();
    tile_count = tc_lds;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t cnt[ORD_BUCKETS];
    __shared__ uint32_t cur[ORD_BUCKETS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    if (tid < ORD_BUCKETS) cnt[tid] = 0;
    const int per = (T + 1023) / 1024;
    const int t0 = (int)tid * per;
    uint32_t sum = 0;
    for (int i = 0; i < per; i++)
        if (t0 + i < T) sum += tile_count[t0 + i];
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += x;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = inc - sum;
    for (uint32_t i = 0; i < w; i++) base += wsum[i];
    // ranges + bucket counts; the thousands of empty tiles share one bucket: one LDS atomic per wave for them
    for (int i = 0; i < per; i++) {
        const int t = t0 + i;
        const bool ok = t < T;
        const uint32_t c = ok ? tile_count[t] : 0u;
        if (ok) ranges[t] = c ? make_uint2(base, base + c) : make_uint2(0u, 0u);
        base += c;
        const uint32_t b = ok ? work_bucket(work_estimate(c, knee, expo)) : 0u;
        const bool empty = ok && b == ORD_BUCKETS - 1;
        const uint64_t em = __ballot(empty);
        if (empty) { if ((em & lt_mask) == 0) atomicAdd(&cnt[b], (uint32_t)__popcll(em)); }
        else if (ok) atomicAdd(&cnt[b], 1u);
    }
    __syncthreads();
    if (tid < 64) {   // exclusive scan of 128 counts by one wave, two per lane
        const uint32_t a0 = cnt[2 * lane], a1 = cnt[2 * lane + 1];
        uint32_t in2 = a0 + a1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = __shfl_up(in2, d, 64);
            if (lane >= (uint32_t)d) in2 += x;
        }
        cur[2 * lane] = in2 - a0 - a1;
        cur[2 * lane + 1] = in2 - a1;
    }
    __syncthreads();
    for (int i = 0; i < per; i++) {
        const int t = t0 + i;
        const bool ok = t < T;
        const uint32_t c = ok ? tile_count[t] : 0u;
        const uint32_t b = ok ? work_bucket(work_estimate(c, knee, expo)) : 0u;
        const bool empty = ok && b == ORD_BUCKETS - 1;
        const uint64_t em = __ballot(empty);
        uint32_t slot = 0, basev = 0;
        const int leader = em ? (int)__builtin_ctzll(em) : 0;
        if (empty && (int)lane == leader) basev = atomicAdd(&cur[b], (uint32_t)__popcll(em));
        basev = __shfl(basev, leader, 64);
        if (empty) slot = basev + (uint32_t)__popcll(em & lt_mask);
        else if (ok) slot = atomicAdd(&cur[b], 1u);
        if (ok) tile_order[slot] = (uint32_t)t;
    }
}

static void order_params(float& knee, float& expo)
{
    static const float k = [] { const char* e = getenv("GSR_ORDER_KNEE"); return e ? (float)atof(e) : 2048.f; }();
    static const float x = [] { const char* e = getenv("GSR_ORDER_EXP"); return e ? (float)atof(e) : 0.3f; }();
    knee = k;
    expo = x;
}

int launch_ranges_order(const Launch& L, const Batch& B, int T)
{
    float knee, expo;
    order_params(knee, expo);
    static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ranges_order), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   4 * LB_MAX_TILES) == hipSuccess;
    (void)lds_ok;
    hipLaunchKernelGGL(k_ranges_order, dim3(B.V), dim3(1024), (size_t)T * 4, L.stream, T, B.iv.tile_count, B.iv.ranges, B.iv.tile_order, B.iv_stride,
                       knee, expo);
    return check_launch(L, "ranges_order");
}

int launch_tile_order(const Launch& L, const Batch& B, int T)
{
    float knee, expo;
    order_params(knee, expo);
    hipLaunchKernelGGL(k_tile_order, dim3(B.V), dim3(1024), 0, L.stream, T, B.iv.ranges, B.iv.tile_order, B.iv_stride, knee, expo);
    return check_launch(L, "tile_order");
}

// ---- backward work items ----------------------------------------------------------------------------------
// The forward render recorded how many entries each tile consumed (tile_need).  That consumed prefix is cut into
// chunks of BWD_CHUNK entries; each (tile, chunk) is an independent backward work item (the forward left the per-pixel
// state at every chunk boundary), so the longest serial walk in the backward kernel is BWD_CHUNK entries instead of a
// whole list.  Items are emitted heaviest first with the same bucket scheme as tile_order; a tile's chunk number
// BWD_MAX_CHUNKS-1 takes everything that is left.  item = tile | chunk << BWD_TILE_BITS.
// Workgroups with blockIdx.y > 0 are the guard against a REPEATED backward over one forward: when the view's gradient
// records were already consumed by a backward (CNT_BWD_DIRTY, raised by k_render_backward as it starts) they clear them before the
// render backward accumulates again; normally they read one word and leave.
constexpr int BWD_CLEAR_BLOCKS = 64;
__global__ __launch_bounds__(1024) void k_bwd_items(int T, int chunk_shift, const uint32_t* __restrict__ need, uint32_t* __restrict__ items,
                                                    uint32_t* __restrict__ count, size_t iv_stride, const uint64_t* __restrict__ counters,
                                                    size_t g_stride, float* __restrict__ grad_rec, size_t gr_stride, size_t gr_bytes)
{
    if (blockIdx.y != 0) {
        if (at_view(counters, g_stride, blockIdx.x)[CNT_BWD_DIRTY] == 0) return;
        zero_region(reinterpret_cast<char*>(at_view(grad_rec, gr_stride, blockIdx.x)), gr_bytes,
                    (size_t)(blockIdx.y - 1) * 1024 + threadIdx.x, (size_t)BWD_CLEAR_BLOCKS * 1024);
        return;
    }
    need = at_view(need, iv_stride, blockIdx.x);
    items = at_view(items, iv_stride, blockIdx.x);
    count = at_view(count, iv_stride, blockIdx.x);
    if (threadIdx.x < 8) count[BWD_QUEUE_WORD + threadIdx.x] = 0;   // the render backward's work-unit counters of this view, one per XCD
    __shared__ uint32_t cnt[ORD_BUCKETS];
    __shared__ uint32_t cur[ORD_BUCKETS];
    const uint32_t lane = threadIdx.x & 63;
    if (threadIdx.x < ORD_BUCKETS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t full_bucket = work_bucket(1u << chunk_shift);
    for (int pass = 0; pass < 2; pass++) {
        for (int t = threadIdx.x; t < T; t += 1024) {
            const uint32_t n = need[t];
            if (n == 0) continue;
            uint32_t n_full = n >> chunk_shift;
            if (n_full > BWD_MAX_CHUNKS - 1) n_full = BWD_MAX_CHUNKS - 1;
            const uint32_t rest = n - (n_full << chunk_shift);   // size of the tile's last item (may exceed the chunk length)
            if (pass == 0) {
                if (n_full) atomicAdd(&cnt[full_bucket], n_full);
                if (rest) atomicAdd(&cnt[work_bucket(rest)], 1u);
            } else {
                if (n_full) {
                    const uint32_t slot = atomicAdd(&cur[full_bucket], n_full);
                    for (uint32_t c = 0; c < n_full; c++) items[slot + c] = (uint32_t)t | (c << BWD_TILE_BITS);
                }
                if (rest) items[atomicAdd(&cur[work_bucket(rest)], 1u)] = (uint32_t)t | (n_full << BWD_TILE_BITS);
            }
        }
        __syncthreads();
        if (pass == 0) {
            if (threadIdx.x < 64) {
                const uint32_t a0 = cnt[2 * lane], a1 = cnt[2 * lane + 1];
                uint32_t inc = a0 + a1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t n = __shfl_up(inc, d, 64);
                    if (lane >= (uint32_t)d) inc += n;
                }
                cur[2 * lane] = inc - a0 - a1;
                cur[2 * lane + 1] = inc - a1;
                if (lane == 63) count[0] = inc;
            }
            __syncthreads();
        }
    }
}

int launch_bwd_items(const Launch& L, const Batch& B, int T, int P)
{
    hipLaunchKernelGGL(k_bwd_items, dim3(B.V, 1 + BWD_CLEAR_BLOCKS), dim3(1024), 0, L.stream, T, B.chunk_shift(), B.iv.tile_need, B.iv.bwd_items,
                       B.iv.bwd_count, B.iv_stride, B.g.counters, B.g_stride, B.grad_rec, B.gr_stride, grad_rec_bytes(P));
    return check_launch(L, "bwd_items");
}

}  // namespace gsr
