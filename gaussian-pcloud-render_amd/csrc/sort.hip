// sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs and the u32 prefix sum.
//
// Role in the pipeline (replaces the reference's CUB calls, CR/rasterizer_impl.cu:277,303-308):
//   the reference sorts R 64-bit keys (tile<<32 | depth bits) with 4-byte payloads in one
//   DeviceRadixSort (6 passes over 12-B pairs at 1080p).  Here the same total order is produced in two
//   steps with 4x less traffic: (1) the P Gaussians are sorted by depth bits (4 passes over P pairs),
//   (2) pairs are emitted in that order and stably sorted by tile id only (ceil(bit/8) passes over R
//   8-B pairs).  Stability of both steps gives exactly: tile, then depth bits, then Gaussian id.
//
// Every launch covers a batch of V independent sort problems (one per camera view, blockIdx.y), laid out at a fixed byte
// stride; a problem's element count may live on the device (the tile sort's num_rendered is never read back mid-frame):
// the grid is sized by the arena capacity and workgroups past the count leave at once.
//
// Two ways to run a pass (same kernels' ranking code, bit-identical results; launch_* at the end of the file choose):
//   * three launches -- per-workgroup digit histogram, per-digit row scan, stable scatter: least traffic, the choice for
//     batches (thousands of workgroups in flight hide the launch boundaries);
//   * single-read histogram + look-back ("onesweep"): ONE kernel reads the keys once and counts the digits of every pass
//     (k_depth_hist / k_tile_hist), then a pass is ONE scatter launch whose workgroups look back at the counts published by
//     the workgroups before them (common.hpp LB_*): 1 + passes launches instead of 3 x passes -- the choice for single-view
//     submissions, whose sort stages are bound by launch latency, not by bandwidth (replaces the reference's
//     cub::DeviceRadixSort::SortPairs, CR/rasterizer_impl.cu:303-308, and its scan, :277).
// The scatter ranks keys with wave64 ballots (one ballot per digit bit), reorders the workgroup's 4096
// pairs in LDS, then writes digit runs with consecutive lanes on consecutive addresses.  The key bits
// are split evenly over the passes (13 bits -> 7 + 6, not 8 + 5): narrower digits mean fewer ballots
// and longer, better coalesced runs per workgroup.
#include "common.hpp"

namespace gsr {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// inclusive wave64 scan via shuffles (not hot)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += n;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread workgroup; tmp = 4 words of LDS
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* tmp, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) tmp[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if ((uint32_t)i < w) base += tmp[i];
    if (total) *total = tmp[0] + tmp[1] + tmp[2] + tmp[3];
    __syncthreads();
    return base + inc - v;
}

struct SortView {
    size_t stride;
    const uint64_t* n_dev;
    size_t n_stride;
    int64_t cap;
    const uint32_t* ctl;   // sortctl of view 0 (depth sort, passes 1..3) or NULL
};
// The depth sort's later passes compare (key - base) and leave at once when their bits are above every difference between two
// keys of visible Gaussians (wave-uniform: one scalar load).  Returns false when the pass has nothing to do.
__device__ __forceinline__ bool pass_control(const SortView& sv, uint32_t view, int shift, uint32_t& base)
{
    base = 0u;
    if (sv.ctl == nullptr) return true;
    const uint32_t* c = at_view(sv.ctl, sv.stride, view);
    base = c[SORTCTL_BASE];
    return (uint32_t)shift < c[SORTCTL_BITS];
}
__device__ __forceinline__ int64_t view_count(const SortView& sv, uint32_t view)
{
    if (sv.n_dev == nullptr) return sv.cap;
    const uint64_t n = *at_view(sv.n_dev, sv.n_stride, view);
    return n < (uint64_t)sv.cap ? (int64_t)n : sv.cap;
}

// ---- pass kernel 1: digit histogram per sort block ------------------------------------------------
// The count matrix is [digit][block], rows padded to whole 64-B lines of HIST_GROUP blocks (k_radix_scatter's workgroup ->
// block map keeps the blocks of one line on one XCD).  Only the digits in use are written.
// (Tried: one histogram workgroup per HIST_GROUP blocks writing whole lines -- the 16 sequential sub-histograms cost more
// than the partial-line stores they save: 61 vs 55 us per launch at 12 views x 11.8 M pairs, 36 vs 11 us for the depth keys.)
constexpr int HIST_GROUP = 16;

template <typename KeyT, int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const KeyT* __restrict__ keys, SortView sv, int shift,
                                                           uint32_t mask, uint32_t* __restrict__ hist, int nblk_pad,
                                                           uint32_t* __restrict__ minmax)   // depth sort, pass 0: [2][nblk_pad], else NULL
{
    constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
    const uint32_t view = blockIdx.y;
    uint32_t kbase;
    if (!pass_control(sv, view, shift, kbase)) return;
    const int64_t n = view_count(sv, view);
    keys = at_view(keys, sv.stride, view);
    hist = at_view(hist, sv.stride, view);
    if (minmax) minmax = at_view(minmax, sv.stride, view);
    const uint32_t d = threadIdx.x;
    if ((int64_t)blockIdx.x * RS_TILE >= n) {   // past the end: this block's column of the count matrix is zero
        if (d <= mask) hist[(size_t)d * nblk_pad + blockIdx.x] = 0;
        if (minmax && d == 0) { minmax[blockIdx.x] = 0xFFFFFFFFu; minmax[nblk_pad + blockIdx.x] = 0u; }
        return;
    }
    // HIST_COPIES private histograms (four per wave, by lane mod 4): same-digit keys of one wave instruction serialise in
    // the LDS atomic unit, the copies quarter those collisions
    constexpr int HIST_COPIES = 4 * RS_WAVES;
    __shared__ __attribute__((aligned(16))) uint32_t h[HIST_COPIES][RADIX];
    __shared__ uint32_t s_min, s_max;
    for (int i = threadIdx.x; i < HIST_COPIES * RADIX / 4; i += RS_THREADS) reinterpret_cast<uint4*>(&h[0][0])[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) { s_min = 0xFFFFFFFFu; s_max = 0u; }
    __syncthreads();
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    const uint32_t w = (threadIdx.x >> 6) * 4u + (threadIdx.x & 3u);
    if (sizeof(KeyT) == 2 && RS_ITEMS % 8 == 0 && base + RS_TILE <= n) {
        // full block of 16-bit keys: 16-B loads of eight keys instead of 2-B ones (counting is order-free)
        const uint4* k4 = reinterpret_cast<const uint4*>(keys + base) + threadIdx.x * (RS_ITEMS / 8);
#pragma unroll
        for (int v = 0; v < RS_ITEMS / 8; v++) {
            const uint4 q = k4[v];
            const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                atomicAdd(&h[w][((wds[j] & 0xFFFFu) >> shift) & mask], 1u);
                atomicAdd(&h[w][((wds[j] >> 16) >> shift) & mask], 1u);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) {
            const int64_t idx = base + i * RS_THREADS + threadIdx.x;
            if (idx < n) {
                const uint32_t key = (uint32_t)keys[idx];
                atomicAdd(&h[w][((key - kbase) >> shift) & mask], 1u);
                if (key != CULLED_KEY) { kmin = key < kmin ? key : kmin; kmax = key > kmax ? key : kmax; }
            }
        }
    }
    if (minmax) {   // (uniform) smallest / largest key of a visible Gaussian in this block
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            const uint32_t a = __shfl_xor(kmin, dd, 64), b = __shfl_xor(kmax, dd, 64);
            kmin = a < kmin ? a : kmin;
            kmax = b > kmax ? b : kmax;
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_min, kmin); atomicMax(&s_max, kmax); }
    }
    __syncthreads();
    if (minmax && d == 0) { minmax[blockIdx.x] = s_min; minmax[nblk_pad + blockIdx.x] = s_max; }
    if (d <= mask) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < HIST_COPIES; i++) c += h[i][d];
        hist[(size_t)d * nblk_pad + blockIdx.x] = c;
    }
}

// ---- pass kernel 2: exclusive scan of each digit's row of workgroup counts -----------------------
// Depth sort, pass 0: one more workgroup (blockIdx.x == nrows) reduces the per-block key extremes of the histogram kernel into
// the view's sortctl words -- base = smallest key with its low 8 bits cleared, bits = position of the highest bit in which
// (largest key - base) is set -- so that the passes above `bits` leave at once: depths within a factor of two of each other
// differ in at most 24 bits (one binade = 2^23 codes), and the fourth pass of the 32-bit sort moves nothing.
__global__ __launch_bounds__(256) void k_radix_rowscan(uint32_t* __restrict__ hist, uint32_t* __restrict__ totals, int nblk_pad,
                                                       size_t stride, const uint32_t* __restrict__ ctl, int shift,
                                                       const uint32_t* __restrict__ minmax, uint32_t* __restrict__ ctl_out,
                                                       uint32_t nrows)
{
    __shared__ uint32_t tmp[4];
    if (blockIdx.x == nrows) {
        minmax = at_view(minmax, stride, blockIdx.y);
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        for (int b = threadIdx.x; b < nblk_pad; b += 256) {
            const uint32_t a = minmax[b], c = minmax[nblk_pad + b];
            kmin = a < kmin ? a : kmin;
            kmax = c > kmax ? c : kmax;
        }
        __shared__ uint32_t s_min, s_max;
        if (threadIdx.x == 0) { s_min = 0xFFFFFFFFu; s_max = 0u; }
        __syncthreads();
        atomicMin(&s_min, kmin);
        atomicMax(&s_max, kmax);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t* c = at_view(ctl_out, stride, blockIdx.y);
            const uint32_t base = s_min <= s_max ? (s_min & ~(uint32_t)(RADIX - 1)) : 0u;
            const uint32_t span = s_min <= s_max ? s_max - base : 0u;      // no visible Gaussian: one pass, any order
            const uint32_t bits = span ? 32u - (uint32_t)__builtin_clz(span) : 0u;
            c[SORTCTL_BASE] = base;
            c[SORTCTL_BITS] = bits < (uint32_t)RADIX_BITS ? (uint32_t)RADIX_BITS : bits;
        }
        return;
    }
    if (ctl != nullptr && (uint32_t)shift >= at_view(ctl, stride, blockIdx.y)[SORTCTL_BITS]) return;
    hist = at_view(hist, stride, blockIdx.y);
    totals = at_view(totals, stride, blockIdx.y);
    uint32_t* row = hist + (size_t)blockIdx.x * nblk_pad;
    uint32_t carry = 0;
    for (int b0 = 0; b0 < nblk_pad; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const uint32_t v = b < nblk_pad ? row[b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan_256(v, tmp, &tot);
        if (b < nblk_pad) row[b] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

#ifdef GSR_STATS
// instrumentation build only: per-phase time of the scatter workgroups of the LAST launch (one record per workgroup, 10-ns ticks):
// 0 keys/values loaded, 1 ranking, 2 prefix sums, 3 reorder in LDS, 4 stores issued
constexpr int SC_REC = 65536;
__device__ unsigned g_sc_rec[SC_REC][5];
#define SC_T(var) const unsigned long long var = wall_clock64()
#define SC_PUT(i, v) do { if (threadIdx.x == 0) { const unsigned r_ = blockIdx.y * gridDim.x + blockIdx.x; if (r_ < SC_REC) g_sc_rec[r_][i] = (unsigned)(v); } } while (0)
int debug_scatter_times(unsigned long long* out8, int reset)
{
    static unsigned host[SC_REC][5];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sc_rec), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    for (int r = 0; r < SC_REC; r++) {
        if (host[r][1] == 0) continue;
        for (int i = 0; i < 5; i++) out8[i] += host[r][i];
        out8[5]++;
    }
    if (reset) {
        for (int r = 0; r < SC_REC; r++) for (int i = 0; i < 5; i++) host[r][i] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sc_rec), host, sizeof(host)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define SC_T(var) do { } while (0)
#define SC_PUT(i, v) do { } while (0)
#endif

// ---- look-back bookkeeping of one pass (common.hpp LB_*) ------------------------------------------
__device__ __forceinline__ uint32_t agent_load32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct LbArgs {
    uint32_t* status;        // view 0's bookkeeping of THIS pass (the key arrays' arena: stride sv.stride)
    const uint32_t* ghist;   // view 0's digit totals of this pass [RADIX]
    size_t ghist_stride;
    uint64_t* counters;      // view 0's CNT_* words (geometry arena) and the host's landing zone: a spin that gives up says so
    size_t cnt_stride;
    uint64_t* host_land;
    uint32_t nblk_cap;       // rows the bookkeeping was laid out for
    int tickets;             // number the blocks by a ticket drawn at their start instead of by blockIdx (common.hpp block_tickets)
};

// a published word is count + 1; polls until it is there (bounded: seconds -- then the frame is reported as failed, not hung)
__device__ __forceinline__ uint32_t lb_wait(const uint32_t* p, uint32_t v, const LbArgs& lb, uint32_t view)
{
    uint32_t spins = 0;
    while (v == 0u) {
        __builtin_amdgcn_s_sleep(1);
        v = agent_load32(p);
        if (v == 0u && ++spins > (1u << 22)) {
            at_view(lb.counters, lb.cnt_stride, view)[CNT_STALL] = 1;
            if (lb.host_land) lb.host_land[4 * view + CNT_STALL] = 1;
            v = 1u;
        }
    }
    return v - 1u;
}

// ---- pass kernel 3: stable scatter ----------------------------------------------------------------
// LB = false: the block's per-digit offsets come from the count matrix the row scan left (hist / totals).
// LB = true : no count matrix -- blocks publish their digit counts and sum what the blocks before them published (two-level,
//             flat: see common.hpp).  A block only ever waits for lower-numbered blocks, numbered by a ticket drawn at their start
//             (or by blockIdx: common.hpp block_tickets).
template <int BITS, typename KeyT, int RS_ITEMS, bool LB>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const KeyT* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,  // NULL: value = index
                                                              KeyT* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, SortView sv, int shift,
                                                              uint32_t mask, const uint32_t* __restrict__ hist,
                                                              const uint32_t* __restrict__ totals, int nblk_pad, LbArgs lb)
{
    constexpr int RS_TILE = RS_THREADS * RS_ITEMS;
    const uint32_t view = blockIdx.y;
    uint32_t kbase;
    if (!pass_control(sv, view, shift, kbase)) return;
    const int64_t n = view_count(sv, view);
    uint32_t blk;
    uint32_t* lb_status = nullptr;
    if constexpr (LB) {
        __shared__ uint32_t s_ticket;
        lb_status = at_view(lb.status, sv.stride, view);
        blk = blockIdx.x;
        if (lb.tickets) {   // (wave-uniform)
            if (threadIdx.x == 0) s_ticket = atomicAdd(&lb_status[0], 1u);
            __syncthreads();
            blk = s_ticket;
        }
        totals = at_view(lb.ghist, lb.ghist_stride, view);
    } else {
        // Workgroup -> sort block: the HIST_GROUP blocks whose counters share a 64-B line of the count matrix run on ONE XCD
        // (workgroup w runs on XCD w % 8), so the line is fetched into that L2 once instead of by sixteen different L2s.
        const uint32_t wg = blockIdx.x, xcd = wg & 7u, r = wg >> 3;
        blk = ((r / HIST_GROUP) * 8u + xcd) * HIST_GROUP + (r % HIST_GROUP);
        hist = at_view(hist, sv.stride, view);
        totals = at_view(totals, sv.stride, view);
    }
    if ((int64_t)blk * RS_TILE >= n) return;
    keys_in = at_view(keys_in, sv.stride, view);
    if (vals_in) vals_in = at_view(vals_in, sv.stride, view);
    keys_out = at_view(keys_out, sv.stride, view);
    vals_out = at_view(vals_out, sv.stride, view);
    __shared__ uint32_t s_key[RS_TILE];
    __shared__ uint32_t s_val[RS_TILE];
    __shared__ uint32_t wave_cnt[RS_WAVES][RADIX];  // running per-wave digit counts, then exclusive over waves
    __shared__ uint32_t local_base[RADIX];          // first slot of digit d inside this workgroup's sorted tile
    __shared__ uint32_t global_base[RADIX];         // first global slot of this workgroup's digit-d run
    __shared__ uint32_t tmp[4];

    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t blk_base = (int64_t)blk * RS_TILE;
    const int64_t seg_base = blk_base + (int64_t)w * (RS_TILE / RS_WAVES);

    for (int i = lane; i < RADIX; i += 64) wave_cnt[w][i] = 0;
    SC_T(t0);

    // requested before the keys (loads return in order): see gbase_d below
    const uint32_t tot_d = tid <= mask ? totals[tid] : 0u;
    uint32_t hist_d = 0u;
    if constexpr (!LB) hist_d = tid <= mask ? hist[(size_t)tid * nblk_pad + blk] : 0u;

    uint32_t k[RS_ITEMS], v[RS_ITEMS], rank[RS_ITEMS];
    const bool full = blk_base + RS_TILE <= n;   // all but the last block of a view: no bounds checks
    {
        const KeyT* kp = keys_in + seg_base + lane;
        const uint32_t* vp = vals_in ? vals_in + seg_base + lane : nullptr;
        // keys are ranked (and staged) as key - kbase; kbase goes back on when they are written (kbase = 0 but in the depth sort)
        if (full) {
#pragma unroll
            for (int j = 0; j < RS_ITEMS; j++) k[j] = (uint32_t)kp[j * 64] - kbase;
            if (vp) {
#pragma unroll
                for (int j = 0; j < RS_ITEMS; j++) v[j] = vp[j * 64];
            } else {
#pragma unroll
                for (int j = 0; j < RS_ITEMS; j++) v[j] = (uint32_t)(seg_base + j * 64 + lane);
            }
        } else {
#pragma unroll
            for (int j = 0; j < RS_ITEMS; j++) {
                const int64_t idx = seg_base + j * 64 + lane;
                const bool ok = idx < n;
                k[j] = ok ? (uint32_t)kp[j * 64] - kbase : 0xFFFFFFFFu;
                v[j] = ok ? (vp ? vp[j * 64] : (uint32_t)idx) : 0u;
            }
        }
    }
    // digit d = tid: first global slot of this workgroup's digit-d run = digits below d in the whole array + digit d in the
    // blocks before this one.  Needs nothing from the keys: the two loads and the scan overlap the key loads' latency instead
    // of sitting between ranking and reorder (a fifth of the workgroup's life there, scripts/dup_times.py).
    uint32_t gbase_d = block_exclusive_scan_256(tot_d, tmp, nullptr) + hist_d;
    __builtin_amdgcn_wave_barrier();
#ifdef GSR_STATS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SC_T(t1);
    SC_PUT(0, t1 - t0);

    // stable rank inside the wave's segment: order is (j, lane).  peers = the lanes holding my digit: one ballot per digit
    // bit, folded in with one three-input bit operation per half (peers & ~(ballot ^ my bit)); the count of peers below me
    // comes from mbcnt.  This loop is most of the kernel's instructions -- the scatter is VALU-bound, not HBM-bound.
#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        const bool ok = full || seg_base + j * 64 + lane < n;
        const uint32_t d = (k[j] >> shift) & mask;
        const uint64_t okb = full ? ~0ull : __ballot(ok);
        uint32_t plo = (uint32_t)okb, phi = (uint32_t)(okb >> 32);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int32_t)k[j], (uint32_t)(shift + b), 1u);   // all ones where my bit b is set
            const uint64_t bal = __ballot(m != 0u);
            plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)bal, m, 0x90);          // a & ~(b ^ c)
            phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(bal >> 32), m, 0x90);
        }
        const uint32_t before = wave_cnt[w][d];
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        rank[j] = before + below;
        __builtin_amdgcn_wave_barrier();
        if (ok && below == 0) wave_cnt[w][d] = before + (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    SC_T(t2);
    SC_PUT(1, t2 - t1);

    // digit d = tid: exclusive prefix over waves, workgroup-local and global run starts
    uint32_t lb_run = 0u;
    {
        const uint32_t d = tid;
        uint32_t c[RS_WAVES], run = 0;
#pragma unroll
        for (int i = 0; i < RS_WAVES; i++) {
            c[i] = wave_cnt[i][d];
            wave_cnt[i][d] = run;
            run += c[i];
        }
        if constexpr (LB) {
            // publish this block's count of digit d right away: the blocks behind this one wait for it
            if (d <= mask) agent_store32(&lb_status[LB_HDR + blk * (mask + 1u) + d], run + 1u);
            lb_run = run;
        }
        const uint32_t lbase = block_exclusive_scan_256(run, tmp, nullptr);
        local_base[d] = lbase;
        global_base[d] = gbase_d;
    }
    __syncthreads();
    SC_T(t3);
    SC_PUT(2, t3 - t2);

#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        if (full || seg_base + j * 64 + lane < n) {
            const uint32_t d = (k[j] >> shift) & mask;
            const uint32_t pos = local_base[d] + wave_cnt[w][d] + rank[j];
            s_key[pos] = k[j];
            s_val[pos] = v[j];
        }
    }
    __syncthreads();
    if constexpr (LB) {
        // Look back, now that the pairs wait in LDS and their registers are free: add up what came before this block -- the sums
        // of the groups before its group and the counts of the earlier blocks of its own group (final when first seen non-zero;
        // they were published while this block reordered).  nd divides 256: 256 / nd threads share a digit's rows.
        const uint32_t d = tid, nd = mask + 1u, ndl = (uint32_t)__builtin_ctz(nd);
        const uint32_t G = blk / LB_GROUP, r = blk % LB_GROUP;
        const uint32_t dd = d & mask, part = d >> ndl, parts = (uint32_t)RS_THREADS >> ndl, rows = G + r;
        uint32_t sum_prev = 0u, sum_own = 0u;
        constexpr uint32_t LBU = 8;   // rows requested back to back: one round trip for LBU of them
        // row -> word offset from the pass' bookkeeping (32-bit: the whole of it is < 2^21 words)
        const uint32_t grp_w = LB_HDR + lb.nblk_cap * nd + dd, own_w = LB_HDR + (G * LB_GROUP - G) * nd + dd;
        for (uint32_t row0 = part; row0 < rows; row0 += parts * LBU) {
            uint32_t v[LBU];
#pragma unroll
            for (uint32_t k = 0; k < LBU; k++) {
                const uint32_t row = row0 + k * parts;
                v[k] = row < rows ? agent_load32(lb_status + ((row < G ? grp_w : own_w) + row * nd)) : 1u;
            }
#pragma unroll
            for (uint32_t k = 0; k < LBU; k++) {
                const uint32_t row = row0 + k * parts;
                const uint32_t c = v[k] != 0u ? v[k] - 1u : lb_wait(lb_status + ((row < G ? grp_w : own_w) + row * nd), 0u, lb, view);
                if (row < G) sum_prev += c; else sum_own += c;
            }
        }
        wave_cnt[0][d] = sum_prev;   // (the per-wave counts have been consumed by the reorder above)
        wave_cnt[1][d] = sum_own;
        __syncthreads();
        if (d < nd) {
            uint32_t prev = 0u, own = 0u;
            for (uint32_t q = 0; q < parts; q++) { prev += wave_cnt[0][q * nd + d]; own += wave_cnt[1][q * nd + d]; }
            global_base[d] += prev + own;
            if (r == LB_GROUP - 1) agent_store32(&lb_status[LB_HDR + (lb.nblk_cap + G) * nd + d], own + lb_run + 1u);
        }
        __syncthreads();
    }
    SC_T(t4);
    SC_PUT(3, t4 - t3);

    const int64_t rem = n - blk_base;
    const uint32_t count = rem < RS_TILE ? (uint32_t)rem : (uint32_t)RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t p = i * RS_THREADS + tid;
        if (p < count) {
            const uint32_t key = s_key[p];
            const uint32_t d = (key >> shift) & mask;
            const size_t g = (size_t)global_base[d] + (p - local_base[d]);
            keys_out[g] = (KeyT)(key + kbase);
            vals_out[g] = s_val[p];
        }
    }
#ifdef GSR_STATS
    { SC_T(t5); SC_PUT(4, t5 - t4); }
#endif
}


// ================================================================================================================================
// Single-read histograms + look-back passes
// ================================================================================================================================
constexpr int HIST_THREADS = 1024;

// ---- depth keys: key range -> control words, digit totals of every pass, bookkeeping cleared -------------------------------
// grid (G, V).  Every workgroup reduces the per-workgroup key extremes k_preprocess left (n_pre records: a few KB from the L2)
// to the frame's control words -- base = smallest key of a Gaussian that emits pairs with its low 8 bits cleared, bits =
// position of the highest bit in which (largest key - base) is set: depths within a factor of two of each other differ in at
// most 24 bits, and the passes above `bits` leave at once -- then counts the digits of (key - base) for the passes that run in
// its slice of the keys (LDS, four copies) and adds them to the view's totals.  The look-back words of the passes are cleared
// here (the scatters run behind this kernel).
struct DepthHistArgs {
    const uint32_t* keys;
    const uint32_t* pre_minmax;
    uint32_t* sortctl;
    uint32_t* ghist;
    uint32_t* lb;
    size_t stride;
    size_t lb_words;    // of all four passes
    int P, n_pre;
};

__global__ __launch_bounds__(HIST_THREADS) void k_depth_hist(DepthHistArgs a)
{
    const uint32_t view = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t* mm = at_view(a.pre_minmax, a.stride, view);
    __shared__ uint32_t s_min, s_max;
    __shared__ uint32_t h[4][4][RADIX];   // [pass][copy][digit]
    if (tid == 0) { s_min = 0xFFFFFFFFu; s_max = 0u; }
    for (uint32_t i = tid; i < 4u * 4u * RADIX; i += HIST_THREADS) (&h[0][0][0])[i] = 0u;
    __syncthreads();
    {
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        for (int b = (int)tid; b < a.n_pre; b += HIST_THREADS) {
            const uint32_t x = mm[b], y = mm[a.n_pre + b];
            kmin = x < kmin ? x : kmin;
            kmax = y > kmax ? y : kmax;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t x = __shfl_xor(kmin, d, 64), y = __shfl_xor(kmax, d, 64);
            kmin = x < kmin ? x : kmin;
            kmax = y > kmax ? y : kmax;
        }
        if (lane == 0) { atomicMin(&s_min, kmin); atomicMax(&s_max, kmax); }
    }
    __syncthreads();
    const uint32_t base = s_min <= s_max ? (s_min & ~(uint32_t)(RADIX - 1)) : 0u;
    const uint32_t span = s_min <= s_max ? s_max - base : 0u;       // no visible Gaussian: one pass, any order
    uint32_t bits = span ? 32u - (uint32_t)__builtin_clz(span) : 0u;
    bits = bits < (uint32_t)RADIX_BITS ? (uint32_t)RADIX_BITS : bits;
    const uint32_t npass = depth_sort_passes(bits);
    if (blockIdx.x == 0 && tid == 0) {
        uint32_t* c = at_view(a.sortctl, a.stride, view);
        c[SORTCTL_BASE] = base;
        c[SORTCTL_BITS] = bits;
    }
    zero_region(reinterpret_cast<char*>(at_view(a.lb, a.stride, view)), a.lb_words * sizeof(uint32_t),
                (size_t)blockIdx.x * HIST_THREADS + tid, (size_t)gridDim.x * HIST_THREADS);
    // this workgroup's slice of the keys, in 16-B loads
    const uint32_t* keys = at_view(a.keys, a.stride, view);
    const int64_t n4 = (a.P + 3) / 4;
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t q0 = per * blockIdx.x, q1 = q0 + per < n4 ? q0 + per : n4;
    const uint32_t copy = (tid >> 6) & 3u;
    for (int64_t q = q0 + tid; q < q1; q += HIST_THREADS) {
        uint32_t k[4];
        if (4 * q + 3 < a.P) {
            const uint4 v = reinterpret_cast<const uint4*>(keys)[q];
            k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) k[j] = 4 * q + j < a.P ? keys[4 * q + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (4 * q + j >= a.P) break;
            const uint32_t rel = k[j] - base;
            atomicAdd(&h[0][copy][rel & 0xFFu], 1u);
            if (npass > 1) atomicAdd(&h[1][copy][(rel >> 8) & 0xFFu], 1u);
            if (npass > 2) atomicAdd(&h[2][copy][(rel >> 16) & 0xFFu], 1u);
            if (npass > 3) atomicAdd(&h[3][copy][rel >> 24], 1u);
        }
    }
    __syncthreads();
    uint32_t* gh = at_view(a.ghist, a.stride, view);
    for (uint32_t i = tid; i < npass * RADIX; i += HIST_THREADS) {
        const uint32_t p = i >> RADIX_BITS, d = i & (RADIX - 1);
        const uint32_t c = h[p][0][d] + h[p][1][d] + h[p][2][d] + h[p][3][d];
        if (c) atomicAdd(&gh[i], c);
    }
}

// ---- tile keys: pairs per tile (-> the tile ranges, binning.hip), digit totals of both passes, bookkeeping cleared -----------
// grid (G, V), dynamic LDS = 4 << bits bytes.  ONE LDS atomic per key into a histogram over the whole key (the tile id, < 2^bits
// <= LB_MAX_TILES); the digit totals of the two passes are row / column sums of that histogram.  Keys arrive in emission order
// -- row-major runs of consecutive tile ids per Gaussian -- so the lanes of a wave mostly hit different bins.
struct TileHistArgs {
    const void* keys;          // u16 or u32
    const uint64_t* n_dev;     // the view's pair count (geometry arena)
    size_t n_stride;
    int64_t cap;
    size_t b_stride;
    uint32_t* tile_count;      // [T]        } image arena, cleared by k_preprocess / the host on a resume
    uint32_t* ghist;           // [2][RADIX] }
    size_t iv_stride;
    uint32_t* lb;
    size_t lb_words;
    int T, bits, bits0;        // key bits in all, bits of pass 0's digit
};

template <typename KeyT>
__global__ __launch_bounds__(HIST_THREADS) void k_tile_hist(TileHistArgs a)
{
    extern __shared__ uint32_t th[];
    const uint32_t view = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t nbins = 1u << a.bits;
    for (uint32_t i = tid; i < nbins; i += HIST_THREADS) th[i] = 0u;
    zero_region(reinterpret_cast<char*>(at_view(a.lb, a.b_stride, view)), a.lb_words * sizeof(uint32_t),
                (size_t)blockIdx.x * HIST_THREADS + tid, (size_t)gridDim.x * HIST_THREADS);
    const uint64_t n64 = *at_view(a.n_dev, a.n_stride, view);
    const int64_t n = n64 < (uint64_t)a.cap ? (int64_t)n64 : a.cap;
    __syncthreads();
    constexpr int KPL = 16 / (int)sizeof(KeyT);   // keys per 16-B load
    const KeyT* keys = at_view(reinterpret_cast<const KeyT*>(a.keys), a.b_stride, view);
    const int64_t nq = (n + KPL - 1) / KPL;
    const int64_t per = (nq + gridDim.x - 1) / gridDim.x;
    const int64_t q0 = per * blockIdx.x, q1 = q0 + per < nq ? q0 + per : nq;
    for (int64_t q = q0 + tid; q < q1; q += HIST_THREADS) {
        if ((q + 1) * KPL <= n) {
            const uint4 v = reinterpret_cast<const uint4*>(keys)[q];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (sizeof(KeyT) == 2) {
                    atomicAdd(&th[w[j] & 0xFFFFu], 1u);
                    atomicAdd(&th[w[j] >> 16], 1u);
                } else {
                    atomicAdd(&th[w[j]], 1u);
                }
            }
        } else {
            for (int j = 0; j < KPL; j++)
                if (q * KPL + j < n) atomicAdd(&th[(uint32_t)keys[q * KPL + j]], 1u);
        }
    }
    __syncthreads();
    uint32_t* tc = at_view(a.tile_count, a.iv_stride, view);
    for (uint32_t t = tid; t < (uint32_t)a.T; t += HIST_THREADS) {
        const uint32_t c = th[t];
        if (c) atomicAdd(&tc[t], c);
    }
    // digit totals: pass 0 = the low bits0 bits (sum over the high part: lanes on consecutive bins), pass 1 = the rest (a wave per
    // digit: lanes on consecutive bins of the digit's row, then a wave reduction)
    uint32_t* gh = at_view(a.ghist, a.iv_stride, view);
    const uint32_t nd0 = 1u << a.bits0, nd1 = nbins >> a.bits0;
    for (uint32_t d = tid; d < nd0; d += HIST_THREADS) {
        uint32_t c = 0;
        for (uint32_t hi = 0; hi < nd1; hi++) c += th[(hi << a.bits0) | d];
        if (c) atomicAdd(&gh[d], c);
    }
    if (nd1 > 1u) {
        for (uint32_t d = tid >> 6; d < nd1; d += HIST_THREADS / 64) {
            uint32_t c = 0;
            for (uint32_t lo = lane; lo < nd0; lo += 64u) c += th[(d << a.bits0) | lo];
#pragma unroll
            for (int x = 32; x >= 1; x >>= 1) c += __shfl_xor(c, x, 64);
            if (lane == 0 && c) atomicAdd(&gh[RADIX + d], c);
        }
    }
}

// Sorts on key bits [0, end_bit).  job.key[0]/val[0] hold the input (val[0] ignored when iota_vals); the
// result lands in key[*result_buffer] / val[*result_buffer] (the parity only depends on end_bit).
int launch_radix_sort_pairs(const Launch& L, const SortJob& job, bool iota_vals, int end_bit, int* result_buffer, bool key16)
{
    int cur = 0;
    if (job.cap > 0 && job.V > 0) {
        const bool small = job.small_blocks && !key16 && end_bit % RADIX_BITS == 0;   // (every digit 8 bits wide)
        const int tile = small ? RS_TILE_SMALL : RS_TILE;
        const int nblk = (int)div_up(job.cap, tile);
        const int nblk_pad = sort_hist_stride(job.cap, tile);
        SortView sv{job.stride, job.n_dev, job.n_stride, job.cap, nullptr};
        const dim3 grid_hist(nblk_pad, job.V);
        const dim3 grid((unsigned)div_up(nblk, 8 * HIST_GROUP) * 8 * HIST_GROUP, job.V);   // see the workgroup -> block map
        bool first = true;
        const int npass = (end_bit + RADIX_BITS - 1) / RADIX_BITS;
        int shift = 0;
        for (int pass = 0; pass < npass; pass++) {
            const int bits = (end_bit - shift + (npass - pass) - 1) / (npass - pass);   // even split, wider digits first
            const uint32_t mask = (1u << bits) - 1u;
            // depth sort: pass 0 produces the control words, the later passes obey them
            const bool make_ctl = job.sortctl != nullptr && pass == 0;
            sv.ctl = (job.sortctl != nullptr && pass > 0) ? job.sortctl : nullptr;
            uint32_t* minmax = make_ctl ? job.blk_minmax : nullptr;
            if (key16)
                hipLaunchKernelGGL((k_radix_hist<uint16_t, RS_ITEMS>), grid_hist, dim3(RS_THREADS), 0, L.stream, (const uint16_t*)job.key[cur], sv,
                                   shift, mask, job.hist, nblk_pad, minmax);
            else if (small)
                hipLaunchKernelGGL((k_radix_hist<uint32_t, RS_ITEMS_SMALL>), grid_hist, dim3(RS_THREADS), 0, L.stream, (const uint32_t*)job.key[cur], sv,
                                   shift, mask, job.hist, nblk_pad, minmax);
            else
                hipLaunchKernelGGL((k_radix_hist<uint32_t, RS_ITEMS>), grid_hist, dim3(RS_THREADS), 0, L.stream, (const uint32_t*)job.key[cur], sv,
                                   shift, mask, job.hist, nblk_pad, minmax);
            if (int e = check_launch(L, "radix_hist")) return e;
            hipLaunchKernelGGL(k_radix_rowscan, dim3(mask + 1u + (make_ctl ? 1u : 0u), job.V), dim3(256), 0, L.stream, job.hist,
                               job.totals, nblk_pad, job.stride, sv.ctl, shift, (const uint32_t*)minmax, job.sortctl, mask + 1u);
            if (int e = check_launch(L, "radix_rowscan")) return e;
            const uint32_t* vin = (first && iota_vals) ? (const uint32_t*)nullptr : (const uint32_t*)job.val[cur];
#define GSR_SCATTER(B)                                                                                                        \
    case B:                                                                                                                   \
        if (key16)                                                                                                            \
            hipLaunchKernelGGL((k_radix_scatter<B, uint16_t, RS_ITEMS, false>), grid, dim3(RS_THREADS), 0, L.stream,          \
                               (const uint16_t*)job.key[cur], vin, (uint16_t*)job.key[cur ^ 1], job.val[cur ^ 1], sv, shift,  \
                               mask, job.hist, job.totals, nblk_pad, LbArgs{});                                               \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_radix_scatter<B, uint32_t, RS_ITEMS, false>), grid, dim3(RS_THREADS), 0, L.stream,          \
                               (const uint32_t*)job.key[cur], vin, job.key[cur ^ 1], job.val[cur ^ 1], sv, shift, mask,       \
                               job.hist, job.totals, nblk_pad, LbArgs{});                                                     \
        break;
            if (small)
                hipLaunchKernelGGL((k_radix_scatter<RADIX_BITS, uint32_t, RS_ITEMS_SMALL, false>), grid, dim3(RS_THREADS), 0, L.stream,
                                   (const uint32_t*)job.key[cur], vin, job.key[cur ^ 1], job.val[cur ^ 1], sv, shift, mask, job.hist,
                                   job.totals, nblk_pad, LbArgs{});
            else switch (bits) {
                GSR_SCATTER(1) GSR_SCATTER(2) GSR_SCATTER(3) GSR_SCATTER(4) GSR_SCATTER(5) GSR_SCATTER(6) GSR_SCATTER(7)
                GSR_SCATTER(8)
            }
#undef GSR_SCATTER
            if (int e = check_launch(L, "radix_scatter")) return e;
            cur ^= 1;
            first = false;
            shift += bits;
        }
    } else {
        cur = ((end_bit + RADIX_BITS - 1) / RADIX_BITS) & 1;
    }
    *result_buffer = cur;
    return GSR_OK;
}

// ---- look-back launch sequences ----------------------------------------------------------------------------------------------
// Depth sort: 1 + 4 launches (the passes above the frame's key bits leave at once, like the three-launch passes).  Same result
// buffer convention: the ids in depth order end up in val[passes & 1].
int launch_depth_sort_lookback(const Launch& L, const SortJob& job, const LbJob& lj)
{
    if (job.cap <= 0 || job.V <= 0) return GSR_OK;
    const int P = (int)job.cap;
    DepthHistArgs h;
    h.keys = job.key[0];
    h.pre_minmax = lj.pre_minmax;
    h.sortctl = job.sortctl;
    h.ghist = lj.ghist;
    h.lb = lj.lb;
    h.stride = job.stride;
    h.lb_words = 4 * lj.pass_words;
    h.P = P;
    h.n_pre = (int)div_up(P, PRE_THREADS);
    int G = (int)div_up(P, HIST_THREADS * 8);
    G = G < 1 ? 1 : G > 64 ? 64 : G;
    hipLaunchKernelGGL(k_depth_hist, dim3(G, job.V), dim3(HIST_THREADS), 0, L.stream, h);
    if (int e = check_launch(L, "depth_hist")) return e;
    SortView sv{job.stride, nullptr, 0, job.cap, job.sortctl};
    const uint32_t nblk = (uint32_t)lb_blocks(job.cap);
    int cur = 0;
    for (int pass = 0; pass < 4; pass++) {
        LbArgs lb{lj.lb + (size_t)pass * lj.pass_words, lj.ghist + pass * RADIX, lj.ghist_stride, lj.counters, lj.cnt_stride, lj.host_land, nblk, block_tickets(-1)};
        hipLaunchKernelGGL((k_radix_scatter<RADIX_BITS, uint32_t, RS_ITEMS, true>), dim3(nblk, job.V), dim3(RS_THREADS), 0, L.stream,
                           (const uint32_t*)job.key[cur], pass == 0 ? (const uint32_t*)nullptr : (const uint32_t*)job.val[cur],
                           job.key[cur ^ 1], job.val[cur ^ 1], sv, pass * RADIX_BITS, (uint32_t)(RADIX - 1), (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, 0, lb);
        if (int e = check_launch(L, "radix_scatter_lookback")) return e;
        cur ^= 1;
    }
    return GSR_OK;
}

// Tile sort on key bits [0, end_bit), end_bit <= 15: 1 + (1 or 2) launches; also leaves the pairs per tile in lj.tile_count.
// The result lands in key / val[*result_buffer] like launch_radix_sort_pairs.
int launch_tile_sort_lookback(const Launch& L, const SortJob& job, const LbJob& lj, int T, int end_bit, int* result_buffer, bool key16)
{
    const int npass = (end_bit + RADIX_BITS - 1) / RADIX_BITS;
    *result_buffer = npass & 1;
    if (job.cap <= 0 || job.V <= 0) return GSR_OK;
    const int bits0 = (end_bit + npass - 1) / npass;   // even split, wider digit first (= launch_radix_sort_pairs)
    TileHistArgs h;
    h.keys = job.key[0];
    h.n_dev = job.n_dev;
    h.n_stride = job.n_stride;
    h.cap = job.cap;
    h.b_stride = job.stride;
    h.tile_count = lj.tile_count;
    h.ghist = const_cast<uint32_t*>(lj.ghist);
    h.iv_stride = lj.ghist_stride;
    h.lb = lj.lb;
    h.lb_words = (size_t)npass * lj.pass_words;
    h.T = T;
    h.bits = end_bit;
    h.bits0 = bits0;
    int G = (int)div_up(job.cap, (int64_t)HIST_THREADS * 64);
    G = G < 1 ? 1 : G > 128 ? 128 : G;
    const size_t lds = (size_t)4 << end_bit;
    static const bool lds_ok = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_hist<uint16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << 15) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_hist<uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 << 15) == hipSuccess;
    }();
    (void)lds_ok;
    if (key16)
        hipLaunchKernelGGL(k_tile_hist<uint16_t>, dim3(G, job.V), dim3(HIST_THREADS), lds, L.stream, h);
    else
        hipLaunchKernelGGL(k_tile_hist<uint32_t>, dim3(G, job.V), dim3(HIST_THREADS), lds, L.stream, h);
    if (int e = check_launch(L, "tile_hist")) return e;
    SortView sv{job.stride, job.n_dev, job.n_stride, job.cap, nullptr};
    const uint32_t nblk = (uint32_t)lb_blocks(job.cap);
    int cur = 0, shift = 0;
    for (int pass = 0; pass < npass; pass++) {
        const int bits = pass == 0 ? bits0 : end_bit - bits0;
        const uint32_t mask = (1u << bits) - 1u;
        LbArgs lb{lj.lb + (size_t)pass * lj.pass_words, lj.ghist + pass * RADIX, lj.ghist_stride, lj.counters, lj.cnt_stride, lj.host_land, nblk, block_tickets(-1)};
        const dim3 grid(nblk, job.V);
#define GSR_SCATTER_LB(B)                                                                                                     \
    case B:                                                                                                                   \
        if (key16)                                                                                                            \
            hipLaunchKernelGGL((k_radix_scatter<B, uint16_t, RS_ITEMS, true>), grid, dim3(RS_THREADS), 0, L.stream,           \
                               (const uint16_t*)job.key[cur], (const uint32_t*)job.val[cur], (uint16_t*)job.key[cur ^ 1],     \
                               job.val[cur ^ 1], sv, shift, mask, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0, lb); \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_radix_scatter<B, uint32_t, RS_ITEMS, true>), grid, dim3(RS_THREADS), 0, L.stream,           \
                               (const uint32_t*)job.key[cur], (const uint32_t*)job.val[cur], job.key[cur ^ 1],                \
                               job.val[cur ^ 1], sv, shift, mask, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0, lb); \
        break;
        switch (bits) {
            GSR_SCATTER_LB(1) GSR_SCATTER_LB(2) GSR_SCATTER_LB(3) GSR_SCATTER_LB(4) GSR_SCATTER_LB(5) GSR_SCATTER_LB(6) GSR_SCATTER_LB(7)
            GSR_SCATTER_LB(8)
        }
#undef GSR_SCATTER_LB
        if (int e = check_launch(L, "radix_scatter_lookback")) return e;
        cur ^= 1;
        shift += bits;
    }
    return GSR_OK;
}

}  // namespace gsr
