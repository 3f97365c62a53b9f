// sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs and the u32 prefix sum.
//
// Role in the pipeline (replaces the reference's CUB calls, CR/rasterizer_impl.cu:277,303-308):
//   the reference sorts R 64-bit keys (tile<<32 | depth bits) with 4-byte payloads in one
//   DeviceRadixSort (6 passes over 12-B pairs at 1080p).  Here the same total order is produced in two
//   steps with 4x less traffic: (1) the P Gaussians are sorted by depth bits (4 passes over P pairs),
//   (2) pairs are emitted in that order and stably sorted by tile id only (ceil(bit/8) passes over R
//   8-B pairs).  Stability of both steps gives exactly: tile, then depth bits, then Gaussian id.
//
// One pass = three launches: per-workgroup digit histogram, per-digit row scan, stable scatter.
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

// ---- pass kernel 1: digit histogram per workgroup ------------------------------------------------
template <typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const KeyT* __restrict__ keys, int64_t n, int shift,
                                                           uint32_t mask, uint32_t* __restrict__ hist, int nblk)
{
    __shared__ uint32_t h[RS_WAVES][RADIX];
    for (int i = threadIdx.x; i < RS_WAVES * RADIX; i += RS_THREADS) (&h[0][0])[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    const uint32_t w = threadIdx.x >> 6;
    if (sizeof(KeyT) == 2 && base + RS_TILE <= n) {
        // full workgroup of 16-bit keys: two 16-B loads per thread instead of sixteen 2-B ones (counting is order-free)
        const uint4* k4 = reinterpret_cast<const uint4*>(keys + base) + threadIdx.x * 2;
#pragma unroll
        for (int v = 0; v < 2; v++) {
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
            if (idx < n) atomicAdd(&h[w][((uint32_t)keys[idx] >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    const uint32_t d = threadIdx.x;
    hist[(size_t)d * nblk + blockIdx.x] = h[0][d] + h[1][d] + h[2][d] + h[3][d];
}

// ---- pass kernel 2: exclusive scan of each digit's row of workgroup counts -----------------------
__global__ __launch_bounds__(256) void k_radix_rowscan(uint32_t* __restrict__ hist, uint32_t* __restrict__ totals, int nblk)
{
    __shared__ uint32_t tmp[4];
    uint32_t* row = hist + (size_t)blockIdx.x * nblk;
    uint32_t carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const uint32_t v = b < nblk ? row[b] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan_256(v, tmp, &tot);
        if (b < nblk) row[b] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// ---- pass kernel 3: stable scatter ----------------------------------------------------------------
template <int BITS, typename KeyT>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const KeyT* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,  // NULL: value = index
                                                              KeyT* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, int64_t n, int shift,
                                                              uint32_t mask, const uint32_t* __restrict__ hist,
                                                              const uint32_t* __restrict__ totals, int nblk)
{
    __shared__ uint32_t s_key[RS_TILE];
    __shared__ uint32_t s_val[RS_TILE];
    __shared__ uint32_t wave_cnt[RS_WAVES][RADIX];  // running per-wave digit counts, then exclusive over waves
    __shared__ uint32_t local_base[RADIX];          // first slot of digit d inside this workgroup's sorted tile
    __shared__ uint32_t global_base[RADIX];         // first global slot of this workgroup's digit-d run
    __shared__ uint32_t tmp[4];

    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t blk_base = (int64_t)blockIdx.x * RS_TILE;
    const int64_t seg_base = blk_base + (int64_t)w * (RS_TILE / RS_WAVES);
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    for (int i = lane; i < RADIX; i += 64) wave_cnt[w][i] = 0;

    uint32_t k[RS_ITEMS], v[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        const int64_t idx = seg_base + j * 64 + lane;
        const bool ok = idx < n;
        k[j] = ok ? (uint32_t)keys_in[idx] : 0xFFFFFFFFu;
        v[j] = ok ? (vals_in ? vals_in[idx] : (uint32_t)idx) : 0u;
    }
    __builtin_amdgcn_wave_barrier();

    // stable rank inside the wave's segment: order is (j, lane)
#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        const int64_t idx = seg_base + j * 64 + lane;
        const bool ok = idx < n;
        const uint32_t d = (k[j] >> shift) & mask;
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t before = wave_cnt[w][d];
        rank[j] = before + (uint32_t)__popcll(peers & lt_mask);
        __builtin_amdgcn_wave_barrier();
        if (ok && (peers & lt_mask) == 0) wave_cnt[w][d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // digit d = tid: exclusive prefix over waves, workgroup-local and global run starts
    {
        const uint32_t d = tid;
        uint32_t c[RS_WAVES], run = 0;
#pragma unroll
        for (int i = 0; i < RS_WAVES; i++) {
            c[i] = wave_cnt[i][d];
            wave_cnt[i][d] = run;
            run += c[i];
        }
        const uint32_t lb = block_exclusive_scan_256(run, tmp, nullptr);
        local_base[d] = lb;
        const uint32_t gt = block_exclusive_scan_256(totals[d], tmp, nullptr);
        global_base[d] = gt + hist[(size_t)d * nblk + blockIdx.x];
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < RS_ITEMS; j++) {
        const int64_t idx = seg_base + j * 64 + lane;
        if (idx < n) {
            const uint32_t d = (k[j] >> shift) & mask;
            const uint32_t pos = local_base[d] + wave_cnt[w][d] + rank[j];
            s_key[pos] = k[j];
            s_val[pos] = v[j];
        }
    }
    __syncthreads();

    const int64_t rem = n - blk_base;
    const uint32_t count = rem < RS_TILE ? (uint32_t)rem : (uint32_t)RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t p = i * RS_THREADS + tid;
        if (p < count) {
            const uint32_t key = s_key[p];
            const uint32_t d = (key >> shift) & mask;
            const size_t g = (size_t)global_base[d] + (p - local_base[d]);
            keys_out[g] = (KeyT)key;
            vals_out[g] = s_val[p];
        }
    }
}

// Sorts on key bits [0, end_bit).  key[0]/val[0] hold the input (val[0] ignored when iota_vals); the
// result lands in key[*result_buffer] / val[*result_buffer].
int launch_radix_sort_pairs(const Launch& L, int64_t n, uint32_t* key[2], uint32_t* val[2], bool iota_vals, int end_bit,
                            uint32_t* hist, uint32_t* totals, int* result_buffer, bool key16)
{
    int cur = 0;
    if (n > 0) {
        const int nblk = (int)div_up(n, RS_TILE);
        bool first = true;
        const int npass = (end_bit + RADIX_BITS - 1) / RADIX_BITS;
        int shift = 0;
        for (int pass = 0; pass < npass; pass++) {
            const int bits = (end_bit - shift + (npass - pass) - 1) / (npass - pass);   // even split, wider digits first
            const uint32_t mask = (1u << bits) - 1u;
            if (key16)
                hipLaunchKernelGGL(k_radix_hist<uint16_t>, dim3(nblk), dim3(RS_THREADS), 0, L.stream, (const uint16_t*)key[cur], n,
                                   shift, mask, hist, nblk);
            else
                hipLaunchKernelGGL(k_radix_hist<uint32_t>, dim3(nblk), dim3(RS_THREADS), 0, L.stream, (const uint32_t*)key[cur], n,
                                   shift, mask, hist, nblk);
            if (int e = check_launch(L, "radix_hist")) return e;
            hipLaunchKernelGGL(k_radix_rowscan, dim3(RADIX), dim3(256), 0, L.stream, hist, totals, nblk);
            if (int e = check_launch(L, "radix_rowscan")) return e;
            const uint32_t* vin = (first && iota_vals) ? (const uint32_t*)nullptr : (const uint32_t*)val[cur];
#define GSR_SCATTER(B)                                                                                                     \
    case B:                                                                                                                \
        if (key16)                                                                                                         \
            hipLaunchKernelGGL((k_radix_scatter<B, uint16_t>), dim3(nblk), dim3(RS_THREADS), 0, L.stream,                  \
                               (const uint16_t*)key[cur], vin, (uint16_t*)key[cur ^ 1], val[cur ^ 1], n, shift, mask, hist, \
                               totals, nblk);                                                                              \
        else                                                                                                               \
            hipLaunchKernelGGL((k_radix_scatter<B, uint32_t>), dim3(nblk), dim3(RS_THREADS), 0, L.stream,                  \
                               (const uint32_t*)key[cur], vin, key[cur ^ 1], val[cur ^ 1], n, shift, mask, hist, totals,   \
                               nblk);                                                                                      \
        break;
            switch (bits) {
                GSR_SCATTER(1) GSR_SCATTER(2) GSR_SCATTER(3) GSR_SCATTER(4) GSR_SCATTER(5) GSR_SCATTER(6) GSR_SCATTER(7)
                GSR_SCATTER(8)
            }
#undef GSR_SCATTER
            if (int e = check_launch(L, "radix_scatter")) return e;
            cur ^= 1;
            first = false;
            shift += bits;
        }
    }
    *result_buffer = cur;
    return GSR_OK;
}

// ---- exclusive prefix sum of tiles_touched taken in depth order ------------------------------------
// (the reference's cub::DeviceScan::InclusiveSum over index order, CR/rasterizer_impl.cu:277; the total
// -- num_rendered -- is order independent.)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(int P, const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ tiles_touched,
                                                              uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t tmp[4];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < P) s += tiles_touched[order[base + i]];
    uint32_t tot;
    block_exclusive_scan_256(s, tmp, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_blocksums(uint32_t* __restrict__ block_sums, int nb,
                                                        uint64_t* __restrict__ total_out)
{
    // 64-bit throughout: a hostile cloud (huge scales) can touch P x T > 2^32 tiles; the host refuses such a frame,
    // but it has to see the true total to do so.
    __shared__ uint64_t tmp64[4];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const uint64_t v = b < nb ? (uint64_t)block_sums[b] : 0ull;
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t n = (uint64_t)__shfl_up((unsigned long long)inc, d, 64);
            if (lane >= (uint32_t)d) inc += n;
        }
        if (lane == 63) tmp64[w] = inc;
        __syncthreads();
        uint64_t base = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((uint32_t)i < w) base += tmp64[i];
        const uint64_t tot = tmp64[0] + tmp64[1] + tmp64[2] + tmp64[3];
        __syncthreads();
        if (b < nb) block_sums[b] = (uint32_t)(carry + base + inc - v);  // offsets are only used when the total fits
        carry += tot;
    }
    if (threadIdx.x == 0) total_out[0] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(int P, const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ tiles_touched,
                                                             const uint32_t* __restrict__ block_sums,
                                                             uint32_t* __restrict__ dup_offset)
{
    __shared__ uint32_t tmp[4];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        c[i] = (base + i < P) ? tiles_touched[order[base + i]] : 0u;
        s += c[i];
    }
    uint32_t run = block_sums[blockIdx.x] + block_exclusive_scan_256(s, tmp, nullptr);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < P) dup_offset[base + i] = run;
        run += c[i];
    }
}

int launch_offsets_scan(const Launch& L, int P, const uint32_t* order, const uint32_t* tiles_touched,
                        uint32_t* dup_offset, uint32_t* scan_tmp, uint64_t* total_out)
{
    const int nb = (int)div_up(P, SCAN_TILE);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_THREADS), 0, L.stream, P, order, tiles_touched, scan_tmp);
    if (int e = check_launch(L, "scan_reduce")) return e;
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(256), 0, L.stream, scan_tmp, nb, total_out);
    if (int e = check_launch(L, "scan_blocksums")) return e;
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_THREADS), 0, L.stream, P, order, tiles_touched, scan_tmp, dup_offset);
    return check_launch(L, "scan_apply");
}

}  // namespace gsr
