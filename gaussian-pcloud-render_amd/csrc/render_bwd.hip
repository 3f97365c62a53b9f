// render_bwd.hip -- backward of the per-tile compositing: dL/dpixel -> per-Gaussian dL/d{mean2D, conic,
// opacity, colour}.
//
// Per-(pixel, Gaussian) arithmetic follows reference CR/backward.cu:399-557 (renderCUDA): the tile list is
// walked back to front, entries at or beyond the pixel's n_contrib are skipped, T is rebuilt by dividing out
// (1 - alpha), the same power/alpha skips apply, and nine partial derivatives come out of every contributing pair.
//
// Mapping (ours), same wave-autonomous scheme as render_fwd.hip: one wave64 = one 8x8 quadrant walks list entries on
// its own (64 entries per round gathered one per lane, exact-safe footprint test + ballot, surviving entries staged
// in the wave's own LDS and read back four at a time), never beyond the largest n_contrib of ITS 64 pixels.  No barrier.
// The unit of work is not a whole list but a slice of BWD_CHUNK consumed entries of a tile (binning.hip k_bwd_items):
// the forward pass recorded every pixel's (T, accumulated colour) at the slice boundaries, which is all the reference's
// back-to-front recurrences need to start in the middle, so the slices of a long list are walked concurrently and the
// longest serial walk in this kernel is BWD_CHUNK entries.
//
// Where the reference issues 9 float atomicAdds per contributing PAIR (up to 256 pixels hammering one Gaussian), the
// sums over the 64 pixels of the quadrant are taken FIRST, and not by cross-lane instructions: every one of the nine
// gradients is a pixel-weighted sum of one of two per-(pixel, entry) weights, i.e. a small dense contraction over the
// pixel index, which is what the matrix cores do ("wave reduction on the matrix cores" below: two LDS rows per entry,
// sixteen v_mfma_f32_16x16x4_f32 per batch of eight entries, exact fp32).  Two global float atomic instructions per
// batch (36 lanes each, the nine addresses of an entry inside one 64-B record, zero sums skipped) then update the
// per-Gaussian gradient records: 9 atomics per (quadrant, entry) instead of 9 x 64.
// Summation order differs from the reference's (undefined) atomic order and the second moments are shifted from the
// quadrant centre (or, for batches that hold a splat far from it in its own sigmas, from the four sub-quadrant centres: MODE 2,
// the default) to the splat centre after the sum; T / (1 - alpha) is the reference's division to the last bit but rare boundary
// cases (v_rcp_f32 + one residual step).  Against the float64 value of the same sums (oracle: orc_render_backward_fp64; 503 fuzz
// cases, median of the max-element error) the colour and opacity gradients are as accurate as the reference build's (0.95x /
// 1.06x), the mean2D / conic gradients carry 1.33x / 2.1x its error (3e-7 / 5e-7 of max|g|; 1.85x / 4.3x with the quadrant-centre
// moments alone, 1.25x / 2.0x with the sub-quadrant ones everywhere): profiles/r06_bwd_accuracy.txt.
#include <atomic>
#include <cstdlib>

#include "common.hpp"
#include "tile_cull.hpp"

namespace gsr {

constexpr int BGRP = 4;  // entries evaluated per inner-loop trip
#ifndef GSR_BWD_SUBQ_DEFAULT
#define GSR_BWD_SUBQ_DEFAULT 2
#endif

#ifndef GSR_BWD_DIV
#define GSR_BWD_DIV 1   // how T / (1 - alpha) is formed (see phase 1 of the group loop)
#endif
#ifndef GSR_BWD_NOFMA
#define GSR_BWD_NOFMA 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// see render_fwd.hip: prefetch loads hidden from hipcc's waitcnt pass, retired by hand
__device__ __forceinline__ void bw_prefetch16(f32x4& dst, const void* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void bw_prefetch4(uint32_t& dst, const void* p)
{
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void bw_prefetch4f(float& dst, const void* p)
{
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}

// LDS staging of the surviving entries (see render_fwd.hip for why not v_readlane): compacted, four entries per
// group, component-major inside the group: x0..x3 | y0..y3 | A | B | C | opacity | r | g | b | id  (10 x 16 B), so one
// same-address ds_read_b128 per component hands every lane the four entries' values, pairs adjacent for the
// packed fp32 instructions.
// Groups are 44 words apart, not 40: a round's survivors are staged by slot, lane s writing word (s & 3) of group s >> 2, and with a
// stride of 40 words the sixteen groups start on only four different banks (8 g mod 32): every one of the ten staging stores of a
// round was a 4-way bank conflict.  44 puts the first eight groups on eight different bank quads (12 g mod 32): 2 lanes per bank, the
// minimum for 64 lanes.  Measured: SQ_LDS_BANK_CONFLICT 5.35e7 -> 4.67e7 per 12-view launch (-13 %), kernel time unchanged -- the
// conflicts are not what the kernel waits for (profiles/r06_bench_rocprofv3_summary.txt).
constexpr int QUAD_WORDS = 44;

// core of ocml expf without its range clamps; bit-identical to expf on [-103, 0] (see render_fwd.hip)
__device__ __forceinline__ float bw_exp_nonpos(float x)
{
    const float ph = x * 0x1.715476p+0f;
    float pl = __builtin_fmaf(x, 0x1.715476p+0f, -ph);
    pl = __builtin_fmaf(x, 0x1.4ae0bep-26f, pl);
    const float e = __builtin_rintf(ph);
    const float r = __builtin_amdgcn_exp2f((ph - e) + pl);
    return __builtin_ldexpf(r, (int)e);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 bw_exp_nonpos2(f32x2 x)
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

// ---- cross-lane helpers -------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, BANK_MASK, false));
}
// ---- wave reduction on the matrix cores ------------------------------------------------------------------
// The nine sums of an entry over the 64 pixels of a quadrant are a contraction over the pixel index:
//   dL/d{mean2D, conic, opacity} are moments  sum_p q[e][p] * {1, cx, cy, cx^2, cx cy, cy^2}(p)  of ONE per-(entry, pixel)
//   weight q = G * dL_dalpha (cx, cy = pixel offsets from the quadrant centre: per-LANE constants), shifted to the
//   splat's own centre afterwards (dx = bx - cx with bx = mean.x - quadrant centre, a per-ENTRY constant);
//   dL/dcolour[ch] = sum_p u[e][p] * dL_dpixel[ch][p] with u = alpha * T.
// So a batch of 8 entries is a 16-row matrix (rows 0..7 q, rows 8..15 u) x 64 pixels, written to LDS by the lanes
// that own the pixels and multiplied by a constant 64 x 16 basis with sixteen v_mfma_f32_16x16x4_f32 (exact fp32):
// D[i][j] = sum_p F[i][p] * data[j][p].  (An fp32 MFMA does not run beside other waves' VALU work on this part --
// scripts/probe/mfma_overlap_probe.hip: the two add up -- so the contraction costs its ~40 cycles of SIMD time per
// instruction; that is still less than half of the partial products + transposed butterfly it replaces.)
// Basis rows i = 4 g + r are laid out so that the four
// accumulator registers of a lane (g = lane >> 4, column j = lane & 15) hold everything ONE output needs:
//   g = 0: 1, cx, cy, cx^2 -> conic.x, mean.x     g = 1: 1, cx, cy, cx cy -> conic.y, mean.y
//   g = 2: 1, cx, cy, cy^2 -> conic.w, opacity    g = 3: dL_dpixel r, g, b -> colour (u columns only)
// Apart from four DPP moves that spread an entry's nine values over nine lanes for the atomics, no cross-lane
// instruction is left.  A row stores the pixels in 2x2-block order (mm_pos): K step s = 4 m + r covers row positions
// 16 m + 4 k + r (k = lane >> 4) = the four pixels of block (column pair r, row pair m), so a lane's four ds_read_b128 of
// its row are the B operands of the sixteen steps, and a step whose block no entry of the batch hits multiplies zeros
// and is skipped (12 of 16 blocks are hit per batch on the benchmark views; worth 1.7 % of the kernel).
constexpr int MM_STRIDE = 68;   // floats per LDS row: 272 B, consecutive rows 4 banks apart (conflict-free b128 reads)

// position of pixel (lane = 8 y + x) in an LDS row: 16 (y >> 1) + 4 ((x & 1) + 2 (y & 1)) + (x >> 1)
__device__ __forceinline__ uint32_t mm_pos(uint32_t lane)
{
    const uint32_t x = lane & 7u, y = lane >> 3;
    return 16u * (y >> 1) + 4u * ((x & 1u) + 2u * (y & 1u)) + (x >> 1);
}
// SUBQ: the basis polynomials of a K step are taken about the centre of the 4 x 4 SUB-QUADRANT the step's 2 x 2 pixel block lies in
// (|c| <= 1.5 instead of <= 3.5) and the steps of each sub-quadrant go to an accumulator of their own (mm_contract4): the moments a
// splat is shifted by are then at most ~2 pixels from it in the sub-quadrant that holds most of its weight, and the float32
// rounding of the moments -- (b / sigma)^2 relative to the second moment about the splat itself -- drops by the square of that ratio.
template <bool SUBQ>
__device__ __forceinline__ void mm_basis(float (&am)[16], uint32_t lane)
{
    const uint32_t i = lane & 15u, k = lane >> 4;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        // step s = 4 m + r, operand lane group k: pixel x = 2 r + (k & 1), y = 2 m + (k >> 1)
        const uint32_t x = 2u * (uint32_t)(s & 3) + (k & 1u), y = 2u * (uint32_t)(s >> 2) + (k >> 1);
        const float cx = SUBQ ? (float)(x & 3u) - 1.5f : (float)x - 3.5f, cy = SUBQ ? (float)(y & 3u) - 1.5f : (float)y - 3.5f;
        const uint32_t c = i & 3u;
        const float f = c == 0 ? 1.f : c == 1 ? cx : c == 2 ? cy : (i == 3 ? cx * cx : i == 7 ? cx * cy : cy * cy);
        am[s] = i < 12u ? f : 0.f;
    }
}
// rows 12..14 of the basis: this item's dL_dpixel, staged as rows 0..2 of the LDS matrix
__device__ __forceinline__ void mm_basis_dpx(float (&am)[16], const float* mrow, uint32_t lane)
{
    const uint32_t i = lane & 15u, k = lane >> 4;
    if (i >= 12u && i < 15u) {
        const f32x4* src = (const f32x4*)(mrow + (i - 12u) * MM_STRIDE + 4u * k);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const f32x4 t = src[4 * m];
            am[4 * m + 0] = t.x; am[4 * m + 1] = t.y; am[4 * m + 2] = t.z; am[4 * m + 3] = t.w;
        }
    }
}
// `blocks`: bit 16 m + 2 r set = some row is non-zero in 2x2 block (r, m) (a superset is fine)
__device__ __forceinline__ f32x4 mm_contract(const float* mrow, const float (&am)[16], uint32_t lane, unsigned long long blocks)
{
    const f32x4* rp = (const f32x4*)(mrow + (lane & 15u) * MM_STRIDE + 4u * (lane >> 4));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const f32x4 b = rp[4 * m];
        if ((blocks >> (16 * m + 0)) & 1ull) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(am[4 * m + 0], b.x, acc, 0, 0, 0);
        if ((blocks >> (16 * m + 2)) & 1ull) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(am[4 * m + 1], b.y, acc, 0, 0, 0);
        if ((blocks >> (16 * m + 4)) & 1ull) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(am[4 * m + 2], b.z, acc, 0, 0, 0);
        if ((blocks >> (16 * m + 6)) & 1ull) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(am[4 * m + 3], b.w, acc, 0, 0, 0);
    }
    return acc;
}

// four accumulators, one per 4 x 4 sub-quadrant sq = 2 (m >> 1) + (r >> 1) (see mm_basis<true>); same sixteen steps, same skipping
__device__ __forceinline__ void mm_contract4(const float* mrow, const float (&am)[16], uint32_t lane, unsigned long long blocks, f32x4 (&acc)[4])
{
    const f32x4* rp = (const f32x4*)(mrow + (lane & 15u) * MM_STRIDE + 4u * (lane >> 4));
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const f32x4 b = rp[4 * m];
        f32x4& aL = acc[2 * (m >> 1)];
        f32x4& aR = acc[2 * (m >> 1) + 1];
        if ((blocks >> (16 * m + 0)) & 1ull) aL = __builtin_amdgcn_mfma_f32_16x16x4f32(am[4 * m + 0], b.x, aL, 0, 0, 0);
        if ((blocks >> (16 * m + 2)) & 1ull) aL = __builtin_amdgcn_mfma_f32_16x16x4f32(am[4 * m + 1], b.y, aL, 0, 0, 0);
        if ((blocks >> (16 * m + 4)) & 1ull) aR = __builtin_amdgcn_mfma_f32_16x16x4f32(am[4 * m + 2], b.z, aR, 0, 0, 0);
        if ((blocks >> (16 * m + 6)) & 1ull) aR = __builtin_amdgcn_mfma_f32_16x16x4f32(am[4 * m + 3], b.w, aR, 0, 0, 0);
    }
}

// The adaptive mode keeps the quadrant-centre basis in `am` and takes the sub-quadrant one only for the batches that need it: about
// its sub-quadrant's centre a basis polynomial depends on the step only through (r & 1, m & 1) -- x & 3 = 2 (r & 1) + (k & 1) --
// so a lane needs FOUR values (bq), not sixteen; rows 12..14 (dL_dpixel, lanes with i >= 12) are the same in both bases.
__device__ __forceinline__ void mm_basis_subq4(float (&bq)[4], uint32_t lane)
{
    const uint32_t i = lane & 15u, k = lane >> 4, c = i & 3u;
#pragma unroll
    for (int h = 0; h < 4; h++) {   // h = 2 (m & 1) + (r & 1)
        const float cx = (float)(2u * (uint32_t)(h & 1) + (k & 1u)) - 1.5f, cy = (float)(2u * (uint32_t)(h >> 1) + (k >> 1)) - 1.5f;
        bq[h] = c == 0 ? 1.f : c == 1 ? cx : c == 2 ? cy : (i == 3 ? cx * cx : i == 7 ? cx * cy : cy * cy);
    }
}
// one sub-quadrant (sq = 2 (m >> 1) + (r >> 1)) at a time: its four K steps into one accumulator (the adaptive mode's rare path
// is written for few live registers, not for overlap)
template <int SQ>
__device__ __forceinline__ f32x4 mm_contract_sq(const float* mrow, const float (&am)[16], const float (&bq)[4], uint32_t lane,
                                                unsigned long long blocks)
{
    const f32x4* rp = (const f32x4*)(mrow + (lane & 15u) * MM_STRIDE + 4u * (lane >> 4));
    const bool poly = (lane & 15u) < 12u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mm = 0; mm < 2; mm++) {
        constexpr int r0 = 2 * (SQ & 1);
        const int m = 2 * (SQ >> 1) + mm;
        const f32x4 b = rp[4 * m];
        const float e0 = poly ? bq[2 * (m & 1)] : am[4 * m + r0], e1 = poly ? bq[2 * (m & 1) + 1] : am[4 * m + r0 + 1];
        if ((blocks >> (16 * m + 2 * r0)) & 1ull) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(e0, b[r0], acc, 0, 0, 0);
        if ((blocks >> (16 * m + 2 * r0 + 2)) & 1ull) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(e1, b[r0 + 1], acc, 0, 0, 0);
    }
    return acc;
}

#ifdef GSR_BWD_EMUL
// DIAGNOSTIC build only (scripts/leases/gpu_session_r4b.sh): the same contraction as mm_contract by plain arithmetic, pixels in
// raster order -- GSR_BWD_EMUL = 1: float fmaf chain, 2: double -- to tell the matrix cores' summation apart from the
// moment shift when the accuracy of the sums is in question.  dpx: [3][64] dL_dpixel of the item, raster order.
__device__ __forceinline__ f32x4 mm_contract_emul(const float* mrow, const float* dpx, uint32_t lane)
{
    const uint32_t j = lane & 15u;
    f32x4 acc;
    for (int r = 0; r < 4; r++) {
        const uint32_t i = 4u * (lane >> 4) + (uint32_t)r;
#if GSR_BWD_EMUL == 2
        double sum = 0.0;
#else
        float sum = 0.f;
#endif
        for (uint32_t p = 0; p < 64u; p++) {
            const float cx = (float)(p & 7u) - 3.5f, cy = (float)(p >> 3) - 3.5f;
            const uint32_t c = i & 3u;
            float f = c == 0 ? 1.f : c == 1 ? cx : c == 2 ? cy : (i == 3 ? cx * cx : i == 7 ? cx * cy : cy * cy);
            if (i >= 12u) f = i < 15u ? dpx[(i - 12u) * 64u + p] : 0.f;
            const float d = mrow[j * MM_STRIDE + mm_pos(p)];
#if GSR_BWD_EMUL == 2
            sum += (double)f * (double)d;
#else
            sum = __builtin_fmaf(f, d, sum);
#endif
        }
        acc[r] = (float)sum;
    }
    return acc;
}
#endif

__global__ void k_selftest_mm(float* out256)
{
    __shared__ __attribute__((aligned(16))) float mrow[16 * MM_STRIDE];
    const uint32_t lane = threadIdx.x;
    float am[16];
    mm_basis<false>(am, lane);
    // "dL_dpixel" of pixel p, channel c: ((p + 2 c) % 5) - 2
    const uint32_t pos = mm_pos(lane);
    for (int c = 0; c < 3; c++) mrow[c * MM_STRIDE + pos] = (float)((int)((lane + 2u * c) % 5u) - 2);
    mm_basis_dpx(am, mrow, lane);
    // data[j][p] = ((7 j + 3 p) % 11) - 5: small integers, every sum exact
    // (pixels of the 2x2 blocks (3, 0) and (1, 2) are zero in every row: those two steps are skipped)
    const bool hole = ((lane & 7u) >> 1 == 3u && (lane >> 4) == 0u) || ((lane & 7u) >> 1 == 1u && (lane >> 4) == 2u);
    for (int j = 0; j < 16; j++) mrow[j * MM_STRIDE + pos] = hole ? 0.f : (float)((int)((7u * j + 3u * lane) % 11u) - 5);
    const f32x4 acc = mm_contract(mrow, am, lane, ~((1ull << 6) | (1ull << 34)));
    out256[lane * 4 + 0] = acc.x; out256[lane * 4 + 1] = acc.y; out256[lane * 4 + 2] = acc.z; out256[lane * 4 + 3] = acc.w;
}

int selftest_mm(hipStream_t stream, float* d_scratch256)
{
    hipLaunchKernelGGL(k_selftest_mm, dim3(1), dim3(64), 0, stream, d_scratch256);
    float h[256];
    if (hipMemcpyAsync(h, d_scratch256, sizeof(h), hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int i = 4 * (l >> 4) + r, j = l & 15;
            double want = 0;
            for (int p = 0; p < 64; p++) {
                const double cx = (p & 7) - 3.5, cy = (p >> 3) - 3.5;
                const int c = i & 3;
                double f = c == 0 ? 1. : c == 1 ? cx : c == 2 ? cy : (i == 3 ? cx * cx : i == 7 ? cx * cy : cy * cy);
                if (i >= 12) f = i < 15 ? (double)((p + 2 * (i - 12)) % 5 - 2) : 0.;
                const bool hole = (((p & 7) >> 1) == 3 && (p >> 4) == 0) || (((p & 7) >> 1) == 1 && (p >> 4) == 2);
                want += hole ? 0. : f * (double)((7 * j + 3 * p) % 11 - 5);
            }
            if ((double)h[l * 4 + r] != want) return 1 + l * 4 + r;
        }
    return 0;
}

#if defined(GSR_STATS) && defined(GSR_STATS_HITS)
// instrumentation build only (GSR_EXTRA_FLAGS="-DGSR_STATS -DGSR_STATS_HITS"; the atomics slow the kernel 100x, so they are
// kept out of the timing build): 0 rounds, 1 staged entries, 2 groups, 3 groups with a hit,
// 4 entries with a hit, 5 (pixel, entry) hits, 6 batches flushed, 7 2x2 pixel blocks hit per batch (summed)
__device__ unsigned long long g_bwd_stats[8];
#define BWD_STAT(i, v) do { if (lane == 0) atomicAdd(&g_bwd_stats[i], (unsigned long long)(v)); } while (0)
#else
#define BWD_STAT(i, v) do { } while (0)
#ifdef GSR_STATS
__device__ unsigned long long g_bwd_stats[8];
#endif
#endif

#ifdef GSR_STATS
// per-workgroup (= wave) time split of the LAST backward launch, 10-ns ticks: 0 whole life, 1 waiting at the rotation point
// (next round's records AND this round's atomics: stores count in vmcnt on gfx9), 2 item set-up (per-pixel state, first
// records), 3 footprint test + staging, 4 group evaluation + reduction + atomics issue, 5 rounds, 6 groups, 7 items
constexpr int BW_REC = 1 << 17;
__device__ unsigned g_bwd_rec[BW_REC][8];
#define BW_T(var) const unsigned long long var = wall_clock64()
// raw per-workgroup records of the last launch, eight words each (scripts/debug/bwd_tail.py)
int debug_bwd_records(unsigned* out, int n)
{
    static unsigned host[BW_REC][8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_rec), sizeof(host)) != hipSuccess) return -1;
    if (n > BW_REC) n = BW_REC;
    for (int r = 0; r < n; r++) for (int i = 0; i < 8; i++) out[r * 8 + i] = host[r][i];
    return n;
}
int debug_bwd_times(unsigned long long* out8, int reset)
{
    static unsigned host[BW_REC][8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_rec), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 8; i++) out8[i] = 0;
    unsigned long long longest = 0, waves = 0;
    for (int r = 0; r < BW_REC; r++) {
        for (int i = 0; i < 8; i++) if (i != 5) out8[i] += host[r][i];
        if (host[r][0] > longest) longest = host[r][0];
        if (host[r][0] != 0) waves++;
    }
    out8[5] = longest;   // (rounds are not reported any more) the longest-lived wave
    out8[7] = (out8[7] << 20) | waves;   // items in the upper bits, waves that ran in the lower 20
    if (reset) {
        for (int r = 0; r < BW_REC; r++) for (int i = 0; i < 8; i++) host[r][i] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_rec), host, sizeof(host)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

struct RenderBwdArgs {
    const uint2* ranges;
    const uint32_t* items;       // work items: tile | chunk << BWD_TILE_BITS, heaviest first (binning.hip k_bwd_items)
    const uint32_t* item_count;  // [1]
    const float4* ckpt;          // forward state at chunk boundaries (render_fwd.hip)
    const float* accum;          // [3N] forward accumulated colour without background
    const uint32_t* point_list;
    const Splat* splat;
    int W, H, gridx, num_tiles, chunk_shift;
    const float* bg;
    const float* final_T;
    const uint32_t* n_contrib;
    const float* dL_dpix;
    float* grad_rec;     // [P][GRAD_REC_WORDS] accumulation records (common.hpp)
    uint64_t* counters;  // per view (CNT_*)
    uint32_t V;
    uint32_t dynamic;    // work units are pulled from per-XCD counters (see the item loop)
    size_t g_stride, b_stride, iv_stride, gr_stride;
};

// five waves per SIMD: 96 registers, the accumulators in VGPRs.  MODE 0 fits without scratch; in MODE 2 the cold sub-quadrant path
// spills inside itself (and eight lane constants once per wave at kernel start, reloaded only there): the hot loops and the item
// set-up are free of scratch accesses (checked in the assembly, profiles/r06_isa_mix.txt).  A fifth wave per SIMD is worth 2.6 %.
// MODE: 0 moments about the quadrant centre; 1 about the four sub-quadrant centres; 2 ADAPTIVE: per batch of eight entries, the
// sub-quadrant path only when one of them is flagged -- (b / sigma)^2 > SUBQ_M, b = splat centre - quadrant centre measured with the
// splat's own conic (one compare when the entry is staged; the flag rides in bit 31 of the staged id).  The float32 rounding of
// the quadrant-centred moments grows with that ratio; on the benchmark views no evaluated entry exceeds 20 (the alpha >= 1/255
// footprint ends at ~17: scripts/analysis/subq_flag_hist.py), so their batches never pay, while sub-pixel splats seen from a
// quadrant away -- where the 4x-the-reference errors of the conic gradient come from -- do.
#ifndef GSR_SUBQ_M
#define GSR_SUBQ_M 20.f
#endif
constexpr float SUBQ_M = GSR_SUBQ_M;
template <int MODE>
__attribute__((amdgpu_waves_per_eu(5, 5)))
__global__ __launch_bounds__(64) void k_render_backward(RenderBwdArgs a)
{
    constexpr bool SUBQ = MODE == 1;
    // Work items are (tile, chunk of BWD_CHUNK consumed list entries), one quadrant per wave; which ones a workgroup takes: see the
    // item loop.  XCD-aware either way: workgroup b runs on XCD b % 8 (each XCD has its own L2) and the four quadrant waves of an item
    // run on one XCD at about the same time, so the item's list slice and Splat records are fetched into that L2 once instead of
    // four times.  Batches: groups of 32 workgroups are dealt to the views round-robin (see render_fwd.hip).
    const uint32_t group = blockIdx.x >> 5, view = group % a.V;
    a.ranges = at_view(a.ranges, a.iv_stride, view);
    a.items = at_view(a.items, a.iv_stride, view);
    a.accum = at_view(a.accum, a.iv_stride, view);
    a.final_T = at_view(a.final_T, a.iv_stride, view);
    a.n_contrib = at_view(a.n_contrib, a.iv_stride, view);
    a.ckpt = at_view(a.ckpt, a.b_stride, view);
    a.point_list = at_view(a.point_list, a.b_stride, view);
    a.splat = at_view(a.splat, a.g_stride, view);
    a.grad_rec = at_view(a.grad_rec, a.gr_stride, view);
    a.dL_dpix += (size_t)view * 3u * (size_t)a.W * (size_t)a.H;
    const uint32_t n_items = at_view(a.item_count, a.iv_stride, view)[0];
    // From here on this view's gradient records hold sums of THIS backward: a later backward on the same arenas has to clear
    // them first (k_bwd_items reads the flag; it has finished: stream order).  Raised before anything is accumulated, so a
    // backward that fails half-way still leaves the records marked.
    if (group < a.V && (blockIdx.x & 31u) == 0 && threadIdx.x == 0) at_view(a.counters, a.g_stride, view)[CNT_BWD_DIRTY] = 1;
    // group 16 keeps the entries of a batch that is still open when its round ends (the next round restages groups 0..15)
    __shared__ __attribute__((aligned(16))) float stage[17 * QUAD_WORDS];
    __shared__ __attribute__((aligned(16))) float mrow[16 * MM_STRIDE];   // 8 entries x {q, u} rows x 64 pixels
#ifdef GSR_BWD_EMUL
    __shared__ float dpx_raster[3 * 64];
#endif
    float am[16];                                                          // A operands of the 16 K steps (basis)
    mm_basis<SUBQ>(am, threadIdx.x);
    const uint32_t mm_p = mm_pos(threadIdx.x);                              // this lane's pixel in a row
    // what this lane does with its four accumulator values after a batch's contraction (see mm_basis): g < 3 on q columns,
    // g = 3 on u columns
    const uint32_t mm_g = threadIdx.x >> 4, mm_j = threadIdx.x & 15u;
    const bool mm_u = mm_g == 3u && mm_j >= 8u;
    const uint32_t mm_gb = (mm_j >> 2) & 1u, mm_k4 = mm_j & 3u;     // own column: group of the batch, slot in the group
    // after the hand-over (flush below): which component of grad_rec this lane adds to for entries 0..3 (A) and 4..7 (B)
    const uint32_t mm_c2 = mm_g == 2u ? 8u : mm_g;
    const bool mm_onA = mm_g < 3u ? mm_j < 8u : (mm_j < 4u || mm_j >= 8u), mm_onB = mm_g < 3u ? mm_j < 8u : mm_j >= 4u;
    const uint32_t mm_cA = mm_g < 3u ? (mm_j < 4u ? 2u + mm_g : mm_c2) : (mm_j >= 12u ? 6u : mm_j >= 8u ? 5u : 7u);
    const uint32_t mm_cB = mm_g < 3u ? (mm_j < 4u ? mm_c2 : 2u + mm_g) : (mm_j >= 12u ? 5u : mm_j >= 8u ? 6u : 7u);
    const uint32_t mm_offP = mm_g == 0 ? 8u : 12u, mm_offQ = mm_g == 0 ? 12u : 16u;   // conic A|B resp. B|C
    const float mm_dd = -(mm_g == 0 ? (float)(0.5 * a.W) : (float)(0.5 * a.H));
#ifdef GSR_STATS
    BW_T(tw0);
    unsigned long long tw_wait = 0, tw_setup = 0, tw_stage = 0, tw_eval = 0, n_rounds = 0, n_groups = 0, n_items_done = 0;
#endif
  // Which items a workgroup takes.  STATIC (a.dynamic == 0; the launch has a quartet of workgroups per eight items or more): quartet
  // (b, b + 8, b + 16, b + 24) takes item (group / V) * 8 + b % 8 and every (groups per view * 8)-th after it, one quadrant each -- with
  // at least as many quartets as items that is one item per quartet in dispatch order, heaviest first.  DYNAMIC (batches, whose grid
  // would otherwise be capped far below the item count): every workgroup PULLS (item, quadrant) units of its view from the counter
  // of its XCD until none is left (the counter hands out the items with index % 8 == XCD, quadrant by quadrant), so whatever slot
  // falls free takes the heaviest unit left instead of a fixed every-n-th one, and the launch holds two workgroups per wave slot.
  // (Not for single views: 1 280 waves pulling from one address wait for the counter -- 0.39 instead of 0.25 ms, lease r5x.)
  uint32_t* const queue = const_cast<uint32_t*>(at_view(a.item_count, a.iv_stride, view)) + BWD_QUEUE_WORD + (blockIdx.x & 7u);
  const uint32_t static_stride = ((gridDim.x >> 5) / a.V) * 8u;
  uint32_t item_idx = (group / a.V) * 8u + (blockIdx.x & 7u) - static_stride, q = (blockIdx.x >> 3) & 3u;
  for (;;) {
    if (a.dynamic) {
        uint32_t unit = 0;
        if (threadIdx.x == 0) unit = atomicAdd(queue, 1u);
        unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
        item_idx = (unit >> 2) * 8u + (blockIdx.x & 7u);
        q = unit & 3u;
    } else {
        item_idx += static_stride;
    }
    if (item_idx >= n_items) break;
#ifdef GSR_STATS
    BW_T(ti0);
    n_items_done++;
#endif
    const uint32_t item = a.items[item_idx];
    const uint32_t tile = item & ((1u << BWD_TILE_BITS) - 1u), chunk = item >> BWD_TILE_BITS;
    const uint32_t lane = threadIdx.x;
    const uint32_t tx = tile % (uint32_t)a.gridx, ty = tile / (uint32_t)a.gridx;
    const uint32_t x0 = tx * TILE_X + (q & 1u) * 8u, y0 = ty * TILE_Y + (q >> 1) * 8u;
    const uint32_t px = x0 + (lane & 7u), py = y0 + (lane >> 3);
    const bool inside = px < (uint32_t)a.W && py < (uint32_t)a.H;
    const float pixf_x = (float)px, pixf_y = (float)py;
    const float x0f = (float)x0, y0f = (float)y0;
    const size_t pix = (size_t)py * a.W + px, N = (size_t)a.W * a.H;

    const uint32_t last_contributor = inside ? a.n_contrib[pix] : 0u;
    // entries [0, total) are walked, last first: nothing beyond the quadrant's largest n_contrib was consumed here
    int total;
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(m, d, 64);
            m = m > o ? m : o;
        }
        total = (int)__builtin_amdgcn_readfirstlane(m);   // every lane holds the maximum: tell the compiler it is uniform
    }
    // this item's slice of the list: entries [lo, hi0), walked last first
    const int lo = (int)(chunk << a.chunk_shift);
    if (lo >= total) continue;
    const bool last_chunk = chunk == BWD_MAX_CHUNKS - 1 || lo + (1 << a.chunk_shift) >= total;
    const int hi0 = last_chunk ? total : lo + (1 << a.chunk_shift);

    const uint2 range = a.ranges[tile];
    const float T_final = inside ? a.final_T[pix] : 0.f;
    float dpx0 = 0.f, dpx1 = 0.f, dpx2 = 0.f;
    if (inside) {
        dpx0 = a.dL_dpix[pix];
        dpx1 = a.dL_dpix[N + pix];
        dpx2 = a.dL_dpix[2 * N + pix];
    }
    float bg_dot_dpixel = 0;
    bg_dot_dpixel += a.bg[0] * dpx0;
    bg_dot_dpixel += a.bg[1] * dpx1;
    bg_dot_dpixel += a.bg[2] * dpx2;
    const float neg_tfb = -T_final * bg_dot_dpixel;   // the background term of dL_dalpha, times 1 / (1 - alpha) per entry
    mrow[mm_p] = dpx0; mrow[MM_STRIDE + mm_p] = dpx1; mrow[2 * MM_STRIDE + mm_p] = dpx2;
    mm_basis_dpx(am, mrow, lane);
#ifdef GSR_BWD_EMUL
    dpx_raster[lane] = dpx0; dpx_raster[64 + lane] = dpx1; dpx_raster[128 + lane] = dpx2;
#endif
    const float mm_sx = x0f + 3.5f, mm_sy = y0f + 3.5f;   // quadrant centre

    // Where this lane's reduced value goes.  Even lane 2i owns value i of the 32-batch: entry k = i >> 3 of the group,
    // component c = i & 7 in {mean2D.x, mean2D.y, conic.x, conic.y, conic.w, colour r, g, b}; odd lanes 1, 17, 33, 49
    // own the opacity gradient (component 8) of entries 0..3.  target = grad_rec[id][c]: the nine atomics of an entry
    // fall into one 64-B line.

    // the staging area starts as zeros, so slots of a partly filled last group always hold finite values (their
    // opacity is set to 0 every round, which is what keeps them from ever hitting)
#pragma unroll
    for (int i = 0; i < 16 * QUAD_WORDS / 64; i++) stage[i * 64 + lane] = 0.f;

    // Per-pixel state at the back end of the slice.  A pixel whose last contributor lies inside (or before) the slice
    // starts from its final state exactly like the reference; a pixel that also consumed entries behind the slice
    // starts from what the forward pass recorded at the boundary hi0: T as it was there, and the colour accumulated
    // behind it, (C_final - C_boundary) / T_boundary, which is what the reference's accum_rec recurrence would have
    // built up by the time it reaches this entry (last_alpha = 0 makes the first update take it over unchanged).
    float T = T_final;
    float s_rec = 0.f;                        // accum_rec . dL_dpixel
    float last_alpha = 0.f, last_d = 0.f;     // last alpha, last_color . dL_dpixel
    if (!last_chunk && last_contributor > (uint32_t)hi0) {
        const size_t slot = (size_t)(range.x >> a.chunk_shift) + (size_t)(hi0 >> a.chunk_shift);
        const float4 ck = a.ckpt[slot * 256 + q * 64 + lane];
        T = ck.x;
        const float behind = (a.accum[pix] - ck.y) * dpx0 + (a.accum[N + pix] - ck.z) * dpx1 + (a.accum[2 * N + pix] - ck.w) * dpx2;
        s_rec = behind / ck.x;
    }

    const uint32_t* plist = a.point_list + range.x;
    // Round r covers front indices hi-64 .. hi-1 (hi = hi0 - 64 r), lane i <-> f = hi-1-i, so the lowest set bit of
    // the ballot is the entry nearest the back.  Lanes whose f would fall before the slice re-read entry lo and are masked.
    f32x4 c0, c1, n0, n1;
    float c2b, n2b;  // blue
    uint32_t id_cur, id_nxt, id_nn;
    {
        const int f0 = hi0 - 1 - (int)lane, f1 = hi0 - 65 - (int)lane;
        bw_prefetch4(id_cur, plist + (f0 >= lo ? f0 : lo));
        bw_prefetch4(id_nxt, plist + (f1 >= lo ? f1 : lo));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(id_cur), "+v"(id_nxt)::"memory");
        const Splat* sp = a.splat + id_cur;
        bw_prefetch16(c0, &sp->q0);
        bw_prefetch16(c1, &sp->q1);
        bw_prefetch4f(c2b, &sp->q2);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2b)::"memory");
    }
#ifdef GSR_STATS
    { BW_T(ti1); tw_setup += ti1 - ti0; }
#endif
    unsigned long long hit_px = 0;   // pixels hit by some entry of the open batch
    uint32_t nb = 0, gq0 = 0;   // groups in the open batch (it may span rounds), the staging group of its first one
    for (int hi = hi0; hi > lo; hi -= 64) {
#ifdef GSR_STATS
        BW_T(tr0);
        n_rounds++;
#endif
        {
            const Splat* sp = a.splat + id_nxt;
            bw_prefetch16(n0, &sp->q0);
            bw_prefetch16(n1, &sp->q1);
            bw_prefetch4f(n2b, &sp->q2);
            const int f2 = hi - 129 - (int)lane;
            bw_prefetch4(id_nn, plist + (f2 >= lo ? f2 : lo));
        }
        const int f_lane = hi - 1 - (int)lane;
        // only pixels whose last contributor lies beyond this round's first entry can be hit by the round: test
        // the entries against their bounding rectangle (the deeper the slice, the fewer pixels are left)
        const int round_lo = hi - 64 > lo ? hi - 64 : lo;
        const uint64_t live = __ballot(last_contributor > (uint32_t)round_lo);
        uint64_t mask = 0;
        bool touch = false;
        if (live != 0) {
            int ax, ay, bx, by;
            live_box(live, ax, ay, bx, by);
            touch = f_lane >= lo && may_touch_rect(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, x0f + (float)ax, y0f + (float)ay,
                                                    x0f + (float)bx, y0f + (float)by);
            mask = __ballot(touch);
        }
        if (mask != 0) {
            const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const uint32_t nsurv = (uint32_t)__popcll(mask);
            if (lane < 4) stage[((nsurv - 1) >> 2) * QUAD_WORDS + 20 + lane] = 0.f;  // opacity row of the last group
            if (touch) {
                float* p = stage + (slot >> 2) * QUAD_WORDS + (slot & 3u);
                p[0] = c0.x; p[4] = c0.y; p[8] = c0.z; p[12] = c0.w; p[16] = c1.x; p[20] = c1.y;
                uint32_t id_st = id_cur;
                if constexpr (MODE == 2) {
                    const float bx = c0.x - mm_sx, by = c0.y - mm_sy;
                    const float mq = (c0.z * bx) * bx + 2.f * (c0.w * bx) * by + (c1.x * by) * by;
                    id_st |= mq > SUBQ_M ? 0x80000000u : 0u;
                }
                p[24] = c1.z; p[28] = c1.w; p[32] = c2b; p[36] = __builtin_bit_cast(float, id_st);
            }
        }
        BWD_STAT(0, 1);
        BWD_STAT(1, __popcll(mask));
        int quad = 0;
#ifdef GSR_STATS
        BW_T(tr1);
        tw_stage += tr1 - tr0;
#endif
        const bool last_round = hi - 64 <= lo;
        for (;;) {
          const bool more = mask != 0;
          if (more) {
            BWD_STAT(2, 1);
#ifdef GSR_STATS
            n_groups++;
#endif
            float ex[BGRP], ey[BGRP], eA[BGRP], eB[BGRP], eC[BGRP], eo[BGRP], er[BGRP], eg[BGRP], eb[BGRP];
            uint32_t ef[BGRP];
            {
                const f32x4* p = (const f32x4*)(stage + quad * QUAD_WORDS);
                const f32x4 X = p[0], Y = p[1], A4 = p[2], B4 = p[3], C4 = p[4], O4 = p[5], R4 = p[6], G4 = p[7], Bl = p[8];
#pragma unroll
                for (int k = 0; k < BGRP; k++) {
                    ex[k] = X[k]; ey[k] = Y[k]; eA[k] = A4[k]; eB[k] = B4[k]; eC[k] = C4[k]; eo[k] = O4[k];
                    er[k] = R4[k]; eg[k] = G4[k]; eb[k] = Bl[k];
                }
                quad++;
            }
#pragma unroll
            for (int k = 0; k < BGRP; k++) {
                const bool have = mask != 0;
                const int j = have ? (int)__builtin_ctzll(mask) : 0;
                mask = have ? (mask & (mask - 1)) : 0;
                ef[k] = (uint32_t)(hi - 1 - j);  // 0-based position of the entry in the tile list
            }
            // (the list position carried in an eleventh row of the staged record instead of these ~12 scalar instructions per
            // entry: measured, +-0 -- 0.2353 vs 0.2333 / 0.2394 ms per view, gpurun_out/r4i: the loop does not wait for its scalar unit)
            float Gs[BGRP], alphas[BGRP];
            bool hits[BGRP];
            bool any_lane_hit = false;
            // pairs of entries on float2: the arithmetic maps onto packed fp32 instructions (two IEEE operations per
            // lane per issue slot); per-component rounding is unchanged, so power / alpha / hit equal the forward's
#pragma unroll
            for (int k = 0; k < BGRP; k += 2) {
                const f32x2 X = {ex[k], ex[k + 1]}, Y = {ey[k], ey[k + 1]}, A2 = {eA[k], eA[k + 1]};
                const f32x2 B2 = {eB[k], eB[k + 1]}, C2 = {eC[k], eC[k + 1]}, O2 = {eo[k], eo[k + 1]};
                const f32x2 dx = X - pixf_x, dy = Y - pixf_y;
                const f32x2 power = -0.5f * (A2 * dx * dx + C2 * dy * dy) - B2 * dx * dy;
                const f32x2 G = bw_exp_nonpos2(power);
                const f32x2 al = O2 * G;
                Gs[k] = G.x; Gs[k + 1] = G.y;
                alphas[k] = fminf(0.99f, al.x);
                alphas[k + 1] = fminf(0.99f, al.y);
                hits[k] = (ef[k] < last_contributor) && !(power.x > 0.0f) && !(alphas[k] < 1.0f / 255.0f);
                hits[k + 1] = (ef[k + 1] < last_contributor) && !(power.y > 0.0f) && !(alphas[k + 1] < 1.0f / 255.0f);
                any_lane_hit = any_lane_hit || hits[k] || hits[k + 1];
            }
            if (__builtin_amdgcn_ballot_w64(any_lane_hit) != 0) {
#if defined(GSR_STATS) && defined(GSR_STATS_HITS)
            BWD_STAT(3, 1);
            for (int k = 0; k < BGRP; k++) {
                const uint64_t hm = __ballot(hits[k]);
                BWD_STAT(4, hm != 0 ? 1 : 0);
                BWD_STAT(5, __popcll(hm));
            }
#endif

            // Phase 1 (branch-free): advance the per-pixel recurrences through the four entries.
            // The reference tracks accum_rec[ch], the colour accumulated behind the current entry, only to form
            // sum_ch (c[ch] - accum_rec[ch]) * dL_dpixel[ch].  That sum is linear, so the same recurrence is run on the
            // scalar s = accum_rec . dL_dpixel and d = c . dL_dpixel (7 operations per entry instead of 18); fused
            // multiply-adds are allowed from here on (gradients are compared to tolerance; power / alpha above are not
            // touched, so hit decisions stay those of the forward pass).
            // A lane that does not hit runs the same update with alpha := 0, which is exact: T is multiplied by
            // rcp(1) = 1, its colour weight alpha*T is 0, and the pending pair (last_alpha, last_d) it leaves behind,
            // (0, d), makes the next fold s + 0*(d - s) return s unchanged, while the pair it replaces has just been folded
            // into s exactly as the next hit would have done.  One select instead of five.
            float dLa[BGRP], Gh[BGRP], dch[BGRP];
#pragma unroll
            for (int k = 0; k < BGRP; k++) {
                const bool hit = hits[k];
                const float alpha = hit ? alphas[k] : 0.f;
                const float om = 1.f - alpha;
#if GSR_BWD_DIV == 0
                const float rcp = __builtin_amdgcn_rcpf(om);
                const float Tn = T * rcp;
#elif GSR_BWD_DIV == 1
                // T / (1 - alpha) as the reference writes it (CR/backward.cu:503), without the ten instructions of the generic
                // division: v_rcp_f32 (1 ulp) gives the quotient to ~1.5 ulp, one residual step T - om * q (exact in the FMA)
                // brings it to the correctly rounded value except when the exact quotient lies within ~2^-22 ulp of a rounding
                // boundary.  No scaling is needed: om is in [0.01, 1] and T in [1e-4 * 0.01, 1].
                const float rcp = __builtin_amdgcn_rcpf(om);
                const float q0 = T * rcp;
                const float Tn = __builtin_fmaf(__builtin_fmaf(-om, q0, T), rcp, q0);
#elif GSR_BWD_DIV == 2
                const float r0 = __builtin_amdgcn_rcpf(om);
                const float rcp = __builtin_fmaf(__builtin_fmaf(-om, r0, 1.f), r0, r0);
                const float q0 = T * rcp;
                const float Tn = __builtin_fmaf(__builtin_fmaf(-om, q0, T), rcp, q0);
#else
                const float rcp = 1.f / om;
                const float Tn = T / om;
#endif
#if GSR_BWD_NOFMA
                const float d = (er[k] * dpx0 + eg[k] * dpx1) + eb[k] * dpx2;
                const float sn = last_alpha * last_d + (1.f - last_alpha) * s_rec;
                float dL_dalpha = (d - sn) * Tn;
                dL_dalpha = dL_dalpha + (-T_final * rcp) * bg_dot_dpixel;
#else
                const float d = __builtin_fmaf(eb[k], dpx2, __builtin_fmaf(eg[k], dpx1, er[k] * dpx0));
                const float sn = __builtin_fmaf(last_alpha, last_d - s_rec, s_rec);  // la*last_d + (1-la)*s
                float dL_dalpha = (d - sn) * Tn;
                dL_dalpha = __builtin_fmaf(neg_tfb, rcp, dL_dalpha);   // (-T_final / (1 - alpha)) * bg . dL_dpixel
#endif
                // lanes that do not hit contribute exact zeros: every product of phase 2 carries a factor Gh or alpha*T
                // (dL_dalpha itself stays finite, so 0 * dL_dalpha is 0)
                dLa[k] = dL_dalpha;
                Gh[k] = hit ? Gs[k] : 0.f;
                dch[k] = alpha * Tn;
                T = Tn;
                s_rec = sn;
                last_d = d;
                last_alpha = alpha;
            }
            // Phase 2 is two weights per (pixel, entry): q = G dL_dalpha and u = alpha T, one LDS row each; the sums over the
            // pixels are taken by the matrix cores once two groups have been written (mm_basis)
            {
                float* rq = mrow + nb * 4u * MM_STRIDE + mm_p;
#pragma unroll
                for (int k = 0; k < BGRP; k += 2) {
                    const f32x2 dLa2 = {dLa[k], dLa[k + 1]}, Gh2 = {Gh[k], Gh[k + 1]};
                    const f32x2 op = Gh2 * dLa2;
                    rq[k * MM_STRIDE] = op.x; rq[(k + 1) * MM_STRIDE] = op.y;
                    rq[(8 + k) * MM_STRIDE] = dch[k]; rq[(9 + k) * MM_STRIDE] = dch[k + 1];
                }
                hit_px = (nb == 0 ? 0ull : hit_px) | __builtin_amdgcn_ballot_w64(any_lane_hit);
                if (nb == 0) gq0 = (uint32_t)(quad - 1);   // (a second group is flushed at once: it is group quad - 1 then)
                nb++;
            }
            }   // any lane hit
          }     // more
          if (nb == 2u || (!more && nb != 0u && last_round)) {
            // 2x2 pixel blocks that some entry of the batch hits (bit 16 (y >> 1) + 2 (x >> 1)): the others are zero in all rows
            unsigned long long hit_blocks = hit_px | (hit_px >> 1);
            hit_blocks |= hit_blocks >> 8;
            BWD_STAT(6, 1);                                                   // batches flushed
            BWD_STAT(7, __popcll(hit_blocks & 0x0055005500550055ull));        // blocks hit (of 16 per batch)
            // per-entry constants and targets first: their LDS round trips pass while the matrix pipe works
            const uint32_t gq1 = (uint32_t)(quad > 0 ? quad - 1 : 0);   // the batch's second group, if it has one, was evaluated just now
            const float* e = stage + (mm_gb ? gq1 : gq0) * QUAD_WORDS + mm_k4;
            const float eX = e[0], eY = e[4], cP = e[mm_offP], cQ = e[mm_offQ], cO = e[20];
            uint32_t idA = __builtin_bit_cast(uint32_t, stage[gq0 * QUAD_WORDS + 36u + mm_k4]);
            uint32_t idB = __builtin_bit_cast(uint32_t, stage[gq1 * QUAD_WORDS + 36u + mm_k4]);
            bool subq_batch = SUBQ;
            if constexpr (MODE == 2) {   // (wave-uniform: every lane sees the same eight staged ids between them)
                subq_batch = __builtin_amdgcn_ballot_w64(((idA | (nb == 2u ? idB : 0u)) >> 31) != 0u) != 0ull;
                idA &= 0x7FFFFFFFu;
                idB &= 0x7FFFFFFFu;
            }
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink these loads behind the matrix instructions)
            // contraction over the 64 pixels, then every lane turns its four moments into (up to) three gradient values
            float o1, o2, o3;
            if (MODE == 1 || (MODE == 2 && __builtin_expect(subq_batch, 0))) {
                // moments about the four sub-quadrant centres (quadrant centre -+2 in x and y), each shifted to the splat centre on
                // its own and then added: sum q dx^2 etc. lose (b / sigma)^2 of their bits with b the distance to the SUB-quadrant
                // centre, and the sub-quadrants far from the splat carry little weight
                const float bx = eX - mm_sx, by = eY - mm_sy;
                float S1 = 0.f, Dx = 0.f, Dy = 0.f, t2 = 0.f, Ay = 0.f, Az = 0.f;
                const auto shift_add = [&](const f32x4 acc, const float bxq, const float byq) {
                    const float s1 = acc.x, sx = acc.y, sy = acc.z, v3 = acc.w;
                    const float dxq = __builtin_fmaf(bxq, s1, -sx), dyq = __builtin_fmaf(byq, s1, -sy);
                    const float t2q = __builtin_fmaf(mm_g == 0 ? bxq : byq, mm_g == 2u ? dyq : dxq,
                                                     __builtin_fmaf(-(mm_g == 2u ? byq : bxq), mm_g == 0 ? sx : sy, v3));
                    S1 += s1; Dx += dxq; Dy += dyq; t2 += t2q; Ay += sx; Az += sy;
                };
                if constexpr (MODE == 2) {
                    float bq[4];   // (built here, a dozen operations per flagged batch, instead of held in four registers all along)
                    mm_basis_subq4(bq, lane);
                    shift_add(mm_contract_sq<0>(mrow, am, bq, lane, hit_blocks), bx + 2.f, by + 2.f);
                    shift_add(mm_contract_sq<1>(mrow, am, bq, lane, hit_blocks), bx - 2.f, by + 2.f);
                    shift_add(mm_contract_sq<2>(mrow, am, bq, lane, hit_blocks), bx + 2.f, by - 2.f);
                    shift_add(mm_contract_sq<3>(mrow, am, bq, lane, hit_blocks), bx - 2.f, by - 2.f);
                } else {
                    f32x4 acc4[4];
                    mm_contract4(mrow, am, lane, hit_blocks, acc4);
                    shift_add(acc4[0], bx + 2.f, by + 2.f);
                    shift_add(acc4[1], bx - 2.f, by + 2.f);
                    shift_add(acc4[2], bx + 2.f, by - 2.f);
                    shift_add(acc4[3], bx - 2.f, by - 2.f);
                }
                o1 = -0.5f * cO * t2;                                                        // conic x | y | w
                o2 = (cO * mm_dd) * __builtin_fmaf(cP, Dx, cQ * Dy);                         // mean2D x | y
                if (mm_g == 2u) o2 = S1;                                                     // opacity
                if (mm_u) { o1 = S1; o2 = Ay; }                                              // colour r, g (row 3, u columns: sums of rows 12, 13)
                o3 = Az;                                                                     // colour b
            } else {
#ifdef GSR_BWD_EMUL
            const f32x4 acc = mm_contract_emul(mrow, dpx_raster, lane);
#else
            const f32x4 acc = mm_contract(mrow, am, lane, hit_blocks);
#endif
            const float S1 = acc.x, Sx = acc.y, Sy = acc.z, V3 = acc.w;
#ifdef GSR_BWD_SHIFT64
            // DIAGNOSTIC: the moment shift in double
            const double bxd = (double)eX - (double)mm_sx, byd = (double)eY - (double)mm_sy;
            const double Dxd = bxd * S1 - Sx, Dyd = byd * S1 - Sy;
            const double t2d = (mm_g == 0 ? bxd : byd) * (mm_g == 2u ? Dyd : Dxd) - (mm_g == 2u ? byd : bxd) * (mm_g == 0 ? Sx : Sy) + V3;
            o1 = (float)(-0.5 * cO * t2d);
            o2 = (float)(((double)cO * mm_dd) * ((double)cP * Dxd + (double)cQ * Dyd));
#else
            const float bx = eX - mm_sx, by = eY - mm_sy;             // splat centre - quadrant centre
            const float Dx = __builtin_fmaf(bx, S1, -Sx), Dy = __builtin_fmaf(by, S1, -Sy);   // sum q dx, sum q dy
            // second moments about the splat centre: sum q dx^2 = bx Dx - bx Sx + Sxx, sum q dx dy = by Dx - bx Sy + Sxy,
            // sum q dy^2 = by Dy - by Sy + Syy
            const float t2 = __builtin_fmaf(mm_g == 0 ? bx : by, mm_g == 2u ? Dy : Dx,
                                            __builtin_fmaf(-(mm_g == 2u ? by : bx), mm_g == 0 ? Sx : Sy, V3));
            o1 = -0.5f * cO * t2;                                                  // conic x | y | w
            o2 = (cO * mm_dd) * __builtin_fmaf(cP, Dx, cQ * Dy);                   // mean2D x | y
#endif
            if (mm_g == 2u) o2 = S1;                                                     // opacity
            if (mm_u) { o1 = acc.x; o2 = acc.y; }                                        // colour r, g (row 3, u columns)
            o3 = acc.z;                                                                  // colour b
            }
            // Hand the second and third values to idle lanes of the same 16-lane row, so that the nine values of an entry
            // sit in nine lanes and ONE atomic instruction serves four entries (the nine addresses of an entry fall into
            // one 64-B record: atomics are priced per line touched).  vA: entries 0..3 of the batch, vB: entries 4..7.
            float vA = dpp_mov<0x114, 0xf, 0xa>(o1, o2);   // row_shr:4 -> banks 1, 3: o2 of the lane four below
            vA = dpp_mov<0x128, 0x8, 0x1>(vA, o3);         // row_ror:8, row 3 bank 0: colour b of lane + 8
            float vB = dpp_mov<0x104, 0xf, 0x5>(o1, o2);   // row_shl:4 -> banks 0, 2: o2 of the lane four above
            vB = dpp_mov<0x128, 0x8, 0x2>(vB, o3);         // row 3 bank 1: colour b of lane + 8
            if (mm_onA && vA != 0.f) atomicAdd(a.grad_rec + (size_t)idA * GRAD_REC_WORDS + mm_cA, vA);
            if (mm_onB && nb == 2u && vB != 0.f) atomicAdd(a.grad_rec + (size_t)idB * GRAD_REC_WORDS + mm_cB, vB);
            nb = 0;
          }
          if (!more) break;
        }
        // a half-filled batch waits for the next round's first group: keep its entries' constants out of the restaging's way
        if (nb == 1u && gq0 != 16u) {
            if (lane < (uint32_t)QUAD_WORDS) stage[16 * QUAD_WORDS + lane] = stage[gq0 * QUAD_WORDS + lane];
            gq0 = 16u;
        }
#ifdef GSR_STATS
        BW_T(tr2);
        tw_eval += tr2 - tr1;
#endif
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(n0), "+v"(n1), "+v"(n2b), "+v"(id_nn)::"memory");
#ifdef GSR_STATS
        { BW_T(tr3); tw_wait += tr3 - tr2; }
#endif
        c0 = n0; c1 = n1; c2b = n2b;
        id_cur = id_nxt;
        id_nxt = id_nn;
    }
  }
#ifdef GSR_STATS
    if (threadIdx.x == 0 && blockIdx.x < (unsigned)BW_REC) {
        BW_T(tw1);
        unsigned* r_ = g_bwd_rec[blockIdx.x];
        r_[0] = (unsigned)(tw1 - tw0); r_[1] = (unsigned)tw_wait; r_[2] = (unsigned)tw_setup; r_[3] = (unsigned)tw_stage;
        r_[4] = (unsigned)tw_eval; r_[5] = (unsigned)tw0; r_[6] = (unsigned)n_groups; r_[7] = (unsigned)n_items_done;   // (5: start tick)
    }
#endif
}

#ifdef GSR_STATS
int debug_bwd_stats(unsigned long long* out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bwd_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// Which moments the contraction takes: 0 (default) about the quadrant centre, 1 about the four sub-quadrant centres (mean2D / conic
// sums at the reference build's accuracy, for more flush arithmetic).  GSR_BWD_SUBQ in the environment when the library is first
// used, or gsr_set_backward_moments() (tests, scripts/bwd_accuracy.py).
static std::atomic<int> g_bwd_subq{-1};
int backward_subquadrant_moments(int set)
{
    if (set >= 0) g_bwd_subq.store(set > 2 ? 2 : set);
    int v = g_bwd_subq.load();
    if (v < 0) {
        const char* e = getenv("GSR_BWD_SUBQ");
        v = e ? atoi(e) : GSR_BWD_SUBQ_DEFAULT;
        v = v < 0 || v > 2 ? GSR_BWD_SUBQ_DEFAULT : v;
        g_bwd_subq.store(v);
    }
    return v;
}

int launch_render_backward(const Launch& L, const gsr_params& p, const Batch& B, const uint32_t* point_list, const float* dL_dpix)
{
    RenderBwdArgs a;
    a.ranges = B.iv.ranges;
    a.items = B.iv.bwd_items;
    a.item_count = B.iv.bwd_count;
    a.ckpt = B.b.ckpt;
    a.accum = B.iv.accum;
    a.point_list = point_list;
    a.splat = B.g.splat;
    a.W = p.W; a.H = p.H;
    a.gridx = (p.W + TILE_X - 1) / TILE_X;
    const int gridy = (p.H + TILE_Y - 1) / TILE_Y;
    a.bg = p.bg;
    a.final_T = B.iv.final_T;
    a.n_contrib = B.iv.n_contrib;
    a.dL_dpix = dL_dpix;
    a.grad_rec = B.grad_rec; a.gr_stride = B.gr_stride;
    a.counters = B.g.counters;
    a.num_tiles = a.gridx * gridy;
    a.chunk_shift = B.chunk_shift();
    a.V = (uint32_t)B.V;
    a.g_stride = B.g_stride; a.b_stride = B.b_stride; a.iv_stride = B.iv_stride;
    // the number of items is only known on the device.  A quartet of workgroups per eight tiles covers the items of a single view one
    // to one (rarely more items than tiles); a batch would need V times that, so its workgroups -- about two per wave slot of the chip
    // (five waves per SIMD, four SIMDs per CU), dealt to the views in groups of 32 -- pull work units until their view's queues are empty
    static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    int64_t groups = div_up(a.num_tiles, 8);
#ifndef GSR_BWD_FILL
#define GSR_BWD_FILL 4096
#endif
    // (batches only: a single view's 1 280 waves per XCD pulling from ONE counter wait for it -- 0.39 instead of 0.25 ms -- however
    // many tiles the image has)
    a.dynamic = (B.V > 1 && groups * B.V > GSR_BWD_FILL) ? 1u : 0u;
    if (a.dynamic) groups = div_up((int64_t)cus * 4 * 5 * 2, 32 * (int64_t)B.V);
    if (groups < 8) groups = 8;
    const dim3 grid((unsigned)(groups * B.V) * 32u);
    switch (backward_subquadrant_moments(-1)) {
    case 1: hipLaunchKernelGGL(k_render_backward<1>, grid, dim3(64), 0, L.stream, a); break;
    case 2: hipLaunchKernelGGL(k_render_backward<2>, grid, dim3(64), 0, L.stream, a); break;
    default: hipLaunchKernelGGL(k_render_backward<0>, grid, dim3(64), 0, L.stream, a); break;
    }
    return check_launch(L, "render_backward");
}

}  // namespace gsr
