// binning.hip -- (tile, Gaussian) pair emission, per-tile list ranges, render launch order.
//
// Pair emission follows reference CR/rasterizer_impl.cu:70-111 (duplicateWithKeys): every visible
// Gaussian emits one pair per tile of its rectangle, rows outer / columns inner.  Differences in
// structure (results identical after the sort, see sort.hip):
//   - Gaussians are visited in depth order, the key is just the tile id (u32);
//   - emission is wave-cooperative: a wave owns 64 consecutive Gaussians, and its lanes write the
//     wave's whole contiguous output range 64 slots at a time (coalesced 4-B stores) instead of one
//     thread writing a 10..40-slot run on its own.
// Tile ranges follow reference CR/rasterizer_impl.cu:116-138 (identifyTileRanges) + the memset at :310.
#include "common.hpp"

namespace gsr {

template <typename KeyT>
__global__ __launch_bounds__(256) void k_duplicate(int P, const uint32_t* __restrict__ order,
                                                   const uint32_t* __restrict__ dup_offset,
                                                   const uint32_t* __restrict__ tiles_touched,
                                                   const uint2* __restrict__ rect, uint32_t gridx,
                                                   KeyT* __restrict__ keys, uint32_t* __restrict__ vals)
{
    __shared__ uint32_t s_off[4][64];
    __shared__ uint32_t s_id[4][64];
    __shared__ uint2 s_rect[4][64];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int slot = blockIdx.x * 256 + threadIdx.x;  // position in depth order

    uint32_t off = 0, cnt = 0, id = 0;
    uint2 rc = make_uint2(0, 0);
    if (slot < P) {
        id = order[slot];
        off = dup_offset[slot];
        cnt = tiles_touched[id];
        rc = rect[id];
    }
    // lanes past P replicate the end offset so the search below never selects them
    const uint32_t wave_begin = __shfl(off, 0, 64);
    uint32_t last_valid_end = off + cnt;
    {   // end of the wave's output range = max over lanes of off+cnt among valid lanes
        uint32_t e = slot < P ? off + cnt : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(e, d, 64);
            e = e > o ? e : o;
        }
        last_valid_end = e;
    }
    if (!(slot < P)) off = last_valid_end;
    s_off[w][lane] = off;
    s_id[w][lane] = id;
    s_rect[w][lane] = rc;
    __builtin_amdgcn_wave_barrier();

    for (uint32_t p = wave_begin + lane; p < last_valid_end; p += 64) {
        // largest s with s_off[s] <= p  (offsets are non-decreasing; zero-count entries are skipped)
        uint32_t lo = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const uint32_t cand = lo + step;
            if (cand < 64 && s_off[w][cand] <= p) lo = cand;
        }
        const uint2 r = s_rect[w][lo];
        const uint32_t minx = r.x & 0xFFFFu, miny = r.x >> 16, maxx = r.y & 0xFFFFu;
        const uint32_t width = maxx - minx;
        const uint32_t t = p - s_off[w][lo];
        const uint32_t row = t / width, col = t - row * width;
        keys[p] = (KeyT)((miny + row) * gridx + (minx + col));
        vals[p] = s_id[w][lo];
    }
}

int launch_duplicate(const Launch& L, int P, const GeomView& g, const uint32_t* order, int gridx, uint32_t* keys,
                     uint32_t* vals, bool key16)
{
    if (key16)
        hipLaunchKernelGGL(k_duplicate<uint16_t>, dim3((P + 255) / 256), dim3(256), 0, L.stream, P, order, g.dup_offset,
                           g.tiles_touched, g.rect, (uint32_t)gridx, (uint16_t*)keys, vals);
    else
        hipLaunchKernelGGL(k_duplicate<uint32_t>, dim3((P + 255) / 256), dim3(256), 0, L.stream, P, order, g.dup_offset,
                           g.tiles_touched, g.rect, (uint32_t)gridx, keys, vals);
    return check_launch(L, "duplicate");
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_tile_ranges(int64_t R, const KeyT* __restrict__ keys, uint2* __restrict__ ranges)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= R) return;
    const uint32_t cur = keys[idx];
    if (idx == 0)
        ranges[cur].x = 0;
    else {
        const uint32_t prev = keys[idx - 1];
        if (cur != prev) {
            ranges[prev].y = (uint32_t)idx;
            ranges[cur].x = (uint32_t)idx;
        }
    }
    if (idx == R - 1) ranges[cur].y = (uint32_t)R;
}

int launch_tile_ranges(const Launch& L, int64_t R, const uint32_t* sorted_keys, uint2* ranges, int T, bool key16)
{
    if (hipMemsetAsync(ranges, 0, (size_t)T * sizeof(uint2), L.stream) != hipSuccess) return GSR_ERR_HIP;
    if (R > 0) {
        if (key16)
            hipLaunchKernelGGL(k_tile_ranges<uint16_t>, dim3((unsigned)div_up(R, 256)), dim3(256), 0, L.stream, R,
                               (const uint16_t*)sorted_keys, ranges);
        else
            hipLaunchKernelGGL(k_tile_ranges<uint32_t>, dim3((unsigned)div_up(R, 256)), dim3(256), 0, L.stream, R, sorted_keys, ranges);
        return check_launch(L, "tile_ranges");
    }
    return GSR_OK;
}

// ---- render launch order: tiles by descending work estimate (128 quarter-octave buckets) -------------
// Per-tile work is (entries walked) x 256 pixels and the spread is 5-10x; dispatching the
// heavy tiles first keeps the tail of the forward render kernel short (it only knows the list length).
// Any permutation is correct.
constexpr int ORD_BUCKETS = 128;

__device__ __forceinline__ uint32_t work_bucket(uint32_t len)
{
    if (len == 0) return ORD_BUCKETS - 1;
    const uint32_t msb = 31u - (uint32_t)__builtin_clz(len);          // 0..31
    const uint32_t frac = msb >= 2 ? (len >> (msb - 2)) & 3u : 0u;    // next two bits
    const uint32_t rank = msb * 4u + frac;                            // larger = more work, 0..127
    return (ORD_BUCKETS - 2) - (rank < ORD_BUCKETS - 2 ? rank : ORD_BUCKETS - 2);
}

// One 1024-thread workgroup: LDS histogram, scan, LDS cursors.  (T is 8 160 at 1080p, 32 400 at 4K.)  Lanes of a
// wave that fall in the same bucket are aggregated with a ballot so the thousands of empty tiles, which all share
// one bucket, cost one LDS atomic per wave instead of 64 serialized ones.
__global__ __launch_bounds__(1024) void k_tile_order(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ tile_order)
{
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
            if (ok) b = work_bucket(ranges[t].y - ranges[t].x);
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

int launch_tile_order(const Launch& L, const ImageView& iv, int T)
{
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, L.stream, T, iv.ranges, iv.tile_order);
    return check_launch(L, "tile_order");
}

// ---- backward work items ----------------------------------------------------------------------------------
// The forward render recorded how many entries each tile consumed (tile_need).  That consumed prefix is cut into
// chunks of BWD_CHUNK entries; each (tile, chunk) is an independent backward work item (the forward left the per-pixel
// state at every chunk boundary), so the longest serial walk in the backward kernel is BWD_CHUNK entries instead of a
// whole list.  Items are emitted heaviest first with the same bucket scheme as tile_order; a tile's chunk number
// BWD_MAX_CHUNKS-1 takes everything that is left.  item = tile | chunk << 20.
__global__ __launch_bounds__(1024) void k_bwd_items(int T, const uint32_t* __restrict__ need, uint32_t* __restrict__ items,
                                                    uint32_t* __restrict__ count)
{
    __shared__ uint32_t cnt[ORD_BUCKETS];
    __shared__ uint32_t cur[ORD_BUCKETS];
    const uint32_t lane = threadIdx.x & 63;
    if (threadIdx.x < ORD_BUCKETS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t full_bucket = work_bucket(BWD_CHUNK);
    for (int pass = 0; pass < 2; pass++) {
        for (int t = threadIdx.x; t < T; t += 1024) {
            const uint32_t n = need[t];
            if (n == 0) continue;
            uint32_t n_full = n >> BWD_CHUNK_SHIFT;
            if (n_full > BWD_MAX_CHUNKS - 1) n_full = BWD_MAX_CHUNKS - 1;
            const uint32_t rest = n - (n_full << BWD_CHUNK_SHIFT);   // size of the tile's last item (may exceed BWD_CHUNK)
            if (pass == 0) {
                if (n_full) atomicAdd(&cnt[full_bucket], n_full);
                if (rest) atomicAdd(&cnt[work_bucket(rest)], 1u);
            } else {
                if (n_full) {
                    const uint32_t slot = atomicAdd(&cur[full_bucket], n_full);
                    for (uint32_t c = 0; c < n_full; c++) items[slot + c] = (uint32_t)t | (c << 20);
                }
                if (rest) items[atomicAdd(&cur[work_bucket(rest)], 1u)] = (uint32_t)t | (n_full << 20);
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

int launch_bwd_items(const Launch& L, const ImageView& iv, int T)
{
    hipLaunchKernelGGL(k_bwd_items, dim3(1), dim3(1024), 0, L.stream, T, iv.tile_need, iv.bwd_items, iv.bwd_count);
    return check_launch(L, "bwd_items");
}

}  // namespace gsr
