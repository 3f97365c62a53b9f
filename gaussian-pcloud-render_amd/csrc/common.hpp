// common.hpp -- shared by every translation unit of libgsr_hip.so (gfx950 only).
//
// Floating-point policy: the whole library is compiled with -ffp-contract=off and performs the
// per-Gaussian / per-pixel arithmetic in the order the reference SOURCE writes it (one IEEE
// rounding per written operation; +,-,*,/ and sqrt are correctly rounded under hipcc defaults), so
// that integer outputs (radii, tile counts, sorted lists, ranges, n_contrib) are reproducible
// bit-for-bit against the CPU oracle and the reference build ("Numerics").
//
// Views: every kernel works on a BATCH of V camera views of one cloud (V = 1 for the reference's per-view API).  The
// scratch arenas of a batch are V identically laid out single-view arenas at a fixed byte stride, so a kernel finds
// view v's arrays by adding v * stride to the view-0 pointers (at_view); the view index is blockIdx.y, or folded into
// blockIdx.x where dispatch order matters (render kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/gsr.h"

// One target: the kernels use gfx950 instructions without a second path (v_permlane32_swap, bitop3, fp32 MFMA shapes, the
// 160-KB LDS); another --offload-arch stops here instead of failing somewhere inside an assembler line.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libgsr_hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace gsr {

constexpr int TILE_X = 16;  // reference CR/config.h:16-17 -- defines keys / ranges, must not change
constexpr int TILE_Y = 16;
constexpr int TILE_PIX = TILE_X * TILE_Y;

// ---- radix sort geometry (sort.hip) -----------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
#ifndef GSR_RS_ITEMS
#define GSR_RS_ITEMS 16
#endif
constexpr int RS_ITEMS = GSR_RS_ITEMS;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // keys per workgroup per pass
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

// the count matrix of a sort is [digit][block] with rows padded to whole 64-B lines (sort.hip HIST_GROUP = 16 blocks)
// The depth sort (P keys per view) runs on half-size blocks when whole-size ones would not fill the chip (RS_ITEMS_SMALL keys per
// thread: 196 blocks of 4096 keys for 800 K Gaussians are fewer than the CUs -- 0.078 -> 0.068 ms for a single view's depth sort,
// 0.0190 -> 0.0183 per view at 12 views per call; the tile sort is slower that way, 0.057 -> 0.066)
constexpr int RS_ITEMS_SMALL = 8;
constexpr int RS_TILE_SMALL = RS_THREADS * RS_ITEMS_SMALL;
inline int sort_hist_stride(int64_t n, int tile = RS_TILE) { return (int)(((n + tile - 1) / tile + 15) / 16 * 16); }

// ---- single-read histogram + look-back passes ("onesweep", sort.hip) ---------------------------------------------------------
// One kernel reads the keys once and counts the digits of EVERY pass (global digit totals do not depend on the order of the
// keys); a pass is then ONE scatter launch whose workgroups find their per-digit offsets by looking back at the counts the
// preceding workgroups of the same launch have published.  Bookkeeping of one pass (u32 words, cleared by the histogram kernel):
//   [LB_HDR]            word 0 = the ticket counter that numbers the workgroups in the order they START
//   [nblk][nd]          count + 1 of digit d in block b        (0 = not published yet)
//   [ngrp][nd]          count + 1 of digit d in the LB_GROUP blocks of group g, published by the group's last block
// nd = digits of the pass (rows are compact).  A block reads the sums of the groups before its own and the counts of the
// earlier blocks of its own group: at most ngrp + LB_GROUP - 1 rows, every one of them final when it is first non-zero -- no
// chain of dependent hops (a hop to another XCD's data costs about a microsecond on this part).
constexpr int LB_HDR = 64;
constexpr int LB_GROUP = 32;
constexpr int64_t LB_MAX_BLOCKS = 4096;   // larger problems run the three-launch passes (a block would read > 160 rows)
inline int64_t lb_blocks(int64_t n) { return (n + RS_TILE - 1) / RS_TILE; }
inline bool lb_fits(int64_t n) { return lb_blocks(n) <= LB_MAX_BLOCKS; }
inline size_t lb_pass_words(int64_t n)    // arena words of one pass' bookkeeping for up to n keys (non-decreasing in n: arena sizes must be)
{
    const int64_t nblk = lb_blocks(n) < LB_MAX_BLOCKS ? lb_blocks(n) : LB_MAX_BLOCKS;
    return (size_t)LB_HDR + (size_t)(nblk + (nblk + LB_GROUP - 1) / LB_GROUP) * RADIX;
}
constexpr int LB_TILE_PASSES = 2;         // tile sort: images of up to LB_MAX_TILES tiles (two passes, full-key histogram in LDS)
constexpr int LB_MAX_TILES = 32768;
constexpr int PRE_THREADS = 256;          // Gaussians per k_preprocess workgroup (one depth-key range record each)

constexpr int DUP_THREADS = 256;  // Gaussians per pair-emission workgroup (binning.hip)
#ifndef GSR_DUP_G
#define GSR_DUP_G 4
#endif
constexpr int DUP_G = GSR_DUP_G;                    // Gaussians per thread of the pair emission
constexpr int DUP_BLOCK = DUP_THREADS * DUP_G;     // Gaussians per emission workgroup (one look-back)
#ifndef GSR_DUP_COPIES
#define GSR_DUP_COPIES 8
#endif
constexpr int DUP_COPIES = GSR_DUP_COPIES;         // copies of the emission's status words (binning.hip: hot lines)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename T>
__host__ __device__ __forceinline__ T* at_view(T* p, size_t stride_bytes, uint32_t view)
{
    // pointer arithmetic, not integer arithmetic: the compiler keeps the kernel argument's global address space and emits
    // global_load / global_store (SGPR base + 32-bit lane offset) instead of flat accesses with 64-bit lane addresses
    using B = typename std::conditional<std::is_const<T>::value, const char, char>::type;
    return (T*)((B*)p + stride_bytes * view);
}

// Per-Gaussian packed splat record: what the render kernels gather per list entry.  Padded to one 64-B cache line, so a
// gather touches exactly one line (a 48-B record straddles two half of the time) and preprocess writes whole lines.
//   q0 = (x, y, conic.x, conic.y)   q1 = (conic.z, opacity, r, g)   q2 = (b, depth, 0, 0)
//   q3 = raw bits for the pair emission: (first tile x | y << 16, last tile + 1, pairs emitted, pairs of the reference's rectangle), or --
//        rectangles of at most 8 x 15 tiles after footprint clipping -- (first tile, row spans of rows 0-3, of rows 4-7, reference
//        pairs | SPANS_FLAG): one byte per tile row, first column in the low nibble, columns in the high one (tile_cull.hpp)
struct __attribute__((aligned(64))) Splat {
    float4 q0, q1, q2, q3;
};

// grad_rec: [P][GRAD_REC_WORDS] accumulation records of the render backward, one 64-B line per Gaussian:
//   0 mean2D.x  1 mean2D.y  2 conic.x  3 conic.y  4 conic.w  5..7 colour r g b  8 opacity  (9..15 unused)
// The nine float atomics a (quadrant, entry) issues land in ONE cache line instead of four arrays' worth
// (scripts/probe/atomic_probe.hip: 4x the atomic throughput).  The records of the V views follow the V per-view geometry
// arenas inside the caller's geometry allocation (Batch::grad_rec; only calls with need_backward carve and need them),
// zeroed by the forward's preprocess kernel, so the backward needs no fill pass.
constexpr int GRAD_REC_WORDS = 16;
inline size_t grad_rec_bytes(int P) { return ((size_t)(P > 0 ? P : 1) * GRAD_REC_WORDS * sizeof(float) + 255) / 256 * 256; }

// counters[] slots (geometry arena, per view)
constexpr int CNT_NUM_RENDERED = 0;  // true number of (tile, Gaussian) pairs, even when it exceeds the arena capacity
constexpr int CNT_TRAP = 1;          // prefiltered = 1 but a Gaussian was culled
constexpr int CNT_STALL = 2;         // the pair emission gave up waiting for a preceding workgroup's count (never expected;
                                     // cleared by k_preprocess, checked by the host: every spin in the library is bounded)
constexpr int CNT_NUM_REFERENCE = 4; // pairs of the reference's unclipped tile rectangles = the num_rendered the API reports
constexpr int LAND_NUM_REFERENCE = 3; // its slot in the host landing zone ([V][4]: CNT_NUM_RENDERED, CNT_TRAP, CNT_STALL, this)
constexpr int CNT_BWD_DIRTY = 3;     // a backward has accumulated into this view's gradient records since the forward cleared
                                     // them: the next backward on the same arenas clears them first (k_bwd_items)

// ---- arena views (device pointers carved out of the caller's opaque buffers) -------------------
struct GeomView {
    Splat* splat;             // [P]
    uint32_t* tiles_touched;  // [P]
    uint8_t* clamped;         // [P] bit k = colour channel k was clamped at 0 (CR/forward.cu:66-69)
    uint32_t* dkey[2];        // [P] depth-bit keys, ping-pong (dkey[0] is also preprocess' output)
    uint32_t* dval[2];        // [P] Gaussian ids, ping-pong; after 4 passes dval[0] = ids in depth order
    uint32_t* hist;           // [RADIX * nblk(P)] per-workgroup digit counts (depth sort)
    uint32_t* totals;         // [RADIX]
    uint32_t* blk_minmax;     // [2 * nblk(P)] smallest / largest depth key of a visible Gaussian per sort block (pass 0's histogram)
    uint32_t* pre_minmax;     // [2 * ceil(P / PRE_THREADS)] the same per k_preprocess workgroup (single-read histogram kernel)
    uint32_t* lb;             // [4 * lb_pass_words(P)] look-back bookkeeping of the four depth-sort passes
    uint32_t* sortctl;        // [4] SORTCTL_*: which key bits the depth sort really has to look at this frame (sort.hip)
    uint64_t* dup_status;     // [DUP_COPIES][ceil(P / DUP_BLOCK)] pair count + 1 of each emission workgroup, then the ticket counter
    uint32_t* ghist;          // [4 * RADIX] digit totals of the depth keys, all passes (accumulated by k_depth_hist)
    uint64_t* counters;       // [8]  (CNT_*; the trap word is cleared by the host only for prefiltered calls)
    char* zero_begin;         // dup_status, ghist: cleared by k_preprocess at the start of every frame
    size_t zero_bytes;
    size_t bytes;
};

// Backward work items: a tile's consumed list is cut into chunks of BWD_CHUNK entries that different waves walk
// concurrently; the forward pass leaves the per-pixel state (T, accumulated colour) at every chunk boundary it crosses.
// The chunk length depends on the views per submission (both halves of a frame see the same V, so both derive the same length):
// a single view's backward launch ends on its longest serial walks and wants them short -- 0.267 / 0.236 / 0.229 ms per view with
// chunks of 1024 / 512 / 256 entries, against 0.012 / 0.016 / 0.019 ms for the item list -- while already two views hide each
// other's walks and a batch pays for the extra items and boundary states instead (ms per view with 1024 / 512 entries: 2 views 0.228 /
// 0.227, 3: 0.220 / 0.217, 4: 0.217 / 0.233, 6: 0.215 / 0.211, 8: 0.215 / 0.215, 12: 0.207 / 0.215; scripts/debug/chunk_v_exp.sh).
constexpr int BWD_CHUNK_SHIFT_MIN = 9;     // the binning arena's boundary-state area is carved for this length
#ifndef GSR_CHUNK_V
#define GSR_CHUNK_V 2     // views per submission from which the slices are 1024 entries long
#endif
__host__ __device__ inline int bwd_chunk_shift(int V) { return V >= GSR_CHUNK_V ? 10 : 9; }
constexpr int BWD_MAX_CHUNKS = 32;   // per tile; the last one takes whatever is left
constexpr int BWD_QUEUE_WORD = 8;    // ImageView::bwd_count
constexpr int BWD_TILE_BITS = 27;    // item = tile | chunk << 27 (check_params limits images to 2^27 tiles)

// The binning arena is carved by CAPACITY (pairs), not by the frame's pair count: the count only exists on the device
// while the frame is being enqueued (no host round trip), and forward and backward must carve identically, so both derive
// the capacity from the arena's byte size (bin_capacity_from_bytes).
struct BinView {
    uint32_t* key[2];  // [cap] tile ids, ping-pong (stored as uint16_t when the image has <= 65536 tiles)
    uint32_t* val[2];  // [cap] Gaussian ids, ping-pong
    uint32_t* hist;    // [RADIX * nblk(cap)]
    uint32_t* totals;  // [RADIX]
    uint32_t* lb;      // [LB_TILE_PASSES * lb_pass_words(cap)] look-back bookkeeping of the tile sort's passes
    float4* ckpt;      // [((cap >> BWD_CHUNK_SHIFT_MIN) + 2) * 256] forward state (T, C.rgb) per pixel of a tile at list position
                       // range.x + k * chunk, slot (range.x >> chunk shift) + k  (unique: lists do not overlap)
    int64_t cap;
    size_t bytes;
};

struct ImageView {
    uint32_t* tile_need;  // [T] entries walked by the forward render  } cleared by k_preprocess at the start of
    uint32_t* bwd_count;  // [16] [0] number of backward items         } every frame
                          //      words BWD_QUEUE_WORD .. + 7: the render backward's work-unit counters of the view, one per XCD (zeroed by k_bwd_items)
    uint32_t* tile_count; // [T] pairs per tile                        } (accumulated by k_tile_hist: ranges = their prefix sums)
    uint32_t* tile_ghist; // [LB_TILE_PASSES * RADIX] digit totals of the tile keys, both passes
    uint2* ranges;        // [T] (first, one past last) list position per tile; (0, 0) for empty tiles
    float* final_T;       // [N]
    uint32_t* n_contrib;  // [N]
    uint32_t* tile_order; // [T] tiles by descending work estimate (forward render launch order, binning.hip k_tile_order)
    float* accum;         // [3N] colour accumulated by the forward render, without the background term (only written
                          //      for quadrants that crossed a BWD_CHUNK boundary: the only ones whose backward reads it)
    uint32_t* bwd_items;  // [BWD_MAX_CHUNKS * T] backward work items: tile | chunk << BWD_TILE_BITS, heaviest first
    char* zero_begin;
    size_t zero_bytes;
    size_t bytes;
};

template <typename T>
inline void carve(char*& cur, T*& ptr, size_t count)
{
    ptr = reinterpret_cast<T*>(cur);
    cur += align_up(count * sizeof(T), 256);
}

inline GeomView geom_view(void* base, int P)
{
    GeomView g;
    char* cur = reinterpret_cast<char*>(base);
    const size_t p = (size_t)(P > 0 ? P : 1);
    const size_t nblk = (size_t)sort_hist_stride((int64_t)p, RS_TILE_SMALL);
    carve(cur, g.splat, p);
    carve(cur, g.tiles_touched, p);
    carve(cur, g.clamped, p);
    carve(cur, g.dkey[0], p);
    carve(cur, g.dkey[1], p);
    carve(cur, g.dval[0], p);
    carve(cur, g.dval[1], p);
    carve(cur, g.hist, RADIX * nblk);
    carve(cur, g.totals, (size_t)RADIX);
    carve(cur, g.blk_minmax, 2 * nblk);
    carve(cur, g.pre_minmax, 2 * (size_t)div_up((int64_t)p, PRE_THREADS));
    carve(cur, g.lb, 4 * lb_pass_words((int64_t)p));
    carve(cur, g.sortctl, (size_t)4);
    g.zero_begin = cur;
    carve(cur, g.dup_status, (size_t)DUP_COPIES * (size_t)div_up((int64_t)p, DUP_BLOCK) + 1);   // + the ticket counter
    carve(cur, g.ghist, (size_t)4 * RADIX);
    g.zero_bytes = (size_t)(cur - g.zero_begin);
    carve(cur, g.counters, (size_t)8);
    g.bytes = (size_t)(cur - reinterpret_cast<char*>(base));
    return g;
}

inline BinView bin_view(void* base, int64_t cap)
{
    BinView b;
    char* cur = reinterpret_cast<char*>(base);
    const size_t r = (size_t)(cap > 0 ? cap : 1);
    const size_t nblk = (size_t)sort_hist_stride((int64_t)r);
    carve(cur, b.key[0], r);
    carve(cur, b.key[1], r);
    carve(cur, b.val[0], r);
    carve(cur, b.val[1], r);
    carve(cur, b.hist, RADIX * nblk);
    carve(cur, b.totals, (size_t)RADIX);
    carve(cur, b.lb, LB_TILE_PASSES * lb_pass_words((int64_t)r));
    carve(cur, b.ckpt, ((r >> BWD_CHUNK_SHIFT_MIN) + 2) * 256);
    b.cap = (int64_t)r;
    b.bytes = (size_t)(cur - reinterpret_cast<char*>(base));
    return b;
}

// largest capacity whose arena fits `bytes` (bin_view(cap).bytes is non-decreasing in cap); 0 if not even one pair fits
inline int64_t bin_capacity_from_bytes(size_t bytes)
{
    if (bin_view(nullptr, 1).bytes > bytes) return 0;
    int64_t lo = 1, hi = (int64_t)(bytes / 16) + 1;   // 16 B per pair is a lower bound of the per-pair cost
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo + 1) / 2;
        if (bin_view(nullptr, mid).bytes <= bytes) lo = mid; else hi = mid - 1;
    }
    return lo;
}

inline ImageView image_view(void* base, int W, int H)
{
    ImageView v;
    char* cur = reinterpret_cast<char*>(base);
    const size_t N = (size_t)W * (size_t)H;
    const size_t T = (size_t)((W + TILE_X - 1) / TILE_X) * (size_t)((H + TILE_Y - 1) / TILE_Y);
    v.zero_begin = cur;
    carve(cur, v.tile_need, T ? T : 1);
    carve(cur, v.bwd_count, (size_t)16);
    carve(cur, v.tile_count, T ? T : 1);
    carve(cur, v.tile_ghist, (size_t)LB_TILE_PASSES * RADIX);
    v.zero_bytes = (size_t)(cur - v.zero_begin);
    carve(cur, v.ranges, T ? T : 1);
    carve(cur, v.final_T, N ? N : 1);
    carve(cur, v.n_contrib, N ? N : 1);
    carve(cur, v.tile_order, T ? T : 1);
    carve(cur, v.accum, 3 * (N ? N : 1));
    carve(cur, v.bwd_items, (size_t)BWD_MAX_CHUNKS * (T ? T : 1));
    v.bytes = (size_t)(cur - reinterpret_cast<char*>(base));
    return v;
}

// Mirrors getHigherMsb, reference CR/rasterizer_impl.cu:35-50 (floor(log2 n) + 1 for n >= 1).
inline uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// ---- launch-side helpers defined in the .hip files ---------------------------------------------
struct Launch {
    hipStream_t stream;
    int debug;
};

// One batch of V views: the view-0 carving of each arena plus the byte stride to the next view's arena.
struct Batch {
    int V;
    int chunk_shift() const { return bwd_chunk_shift(V); }   // log2 of the backward work items' length (list entries)
    GeomView g;
    size_t g_stride;
    ImageView iv;
    size_t iv_stride;
    BinView b;          // carved by capacity (b.cap); b.key[0] == nullptr when no binning arena was given
    size_t b_stride;
    float* grad_rec;    // view 0's [P][GRAD_REC_WORDS] gradient records, behind the V geometry arenas; NULL without need_backward
    size_t gr_stride;
};

int check_launch(const Launch& L, const char* what);  // api.hip

// preprocess.hip
int launch_preprocess(const Launch& L, const gsr_params& p, const Batch& B, int* radii);
int launch_recolor(const Launch& L, const gsr_params& p, const Batch& B, size_t colors_view_stride);
int launch_mark_visible(const Launch& L, int P, const float* means3D, const float* view, uint8_t* present);
// sort.hip
// Stable LSD radix sort of `V` independent (u32 key, u32 value) problems laid out at `stride` bytes from each other.
// n_dev != NULL: the element count of view v is min(*at_view(n_dev, n_stride, v), cap) (device resident); else cap.
// key16: the key arrays hold uint16_t (tile ids of images up to 65536 tiles: 14 instead of 20 B per pair per pass)
struct SortJob {
    uint32_t* key[2];
    uint32_t* val[2];
    uint32_t* hist;
    uint32_t* totals;
    size_t stride;
    const uint64_t* n_dev;
    size_t n_stride;
    int64_t cap;
    int V;
    // Depth sort only (else NULL): pass 0 also finds the smallest / largest key that is not CULLED_KEY, and the passes over key
    // bits that no two such keys differ in leave at once (SORTCTL_*; the result's ping-pong buffer is then sortctl-dependent)
    uint32_t* blk_minmax = nullptr;   // [2 * nblk_pad] per view
    uint32_t* sortctl = nullptr;      // [4] per view
    bool small_blocks = false;        // u32 keys, 8-bit digits only: blocks of RS_TILE_SMALL keys (hist / blk_minmax carved for them)
};
// sortctl words: keys are compared as (key - SORTCTL_BASE) on bits [0, SORTCTL_BITS); SORTCTL_BASE has its low 8 bits clear, so
// pass 0 (which runs before the words exist) sees the same digit either way
constexpr int SORTCTL_BASE = 0;
constexpr int SORTCTL_BITS = 1;     // >= 8: passes whose shift is >= this many bits are skipped
constexpr uint32_t CULLED_KEY = 0xFFFFFFFFu;   // depth key of a Gaussian that emits no pairs: its place in the order is immaterial
// number of depth-sort passes a frame with SORTCTL_BITS = bits executes; ids in depth order end up in dval[passes & 1]
__host__ __device__ inline uint32_t depth_sort_passes(uint32_t bits) { return (bits + RADIX_BITS - 1) / RADIX_BITS; }
int launch_radix_sort_pairs(const Launch& L, const SortJob& job, bool iota_vals, int end_bit, int* result_buffer, bool key16 = false);
// The same sorts as single-read histogram + look-back passes (sort.hip; bit-identical results, fewer launches).
struct LbJob {
    uint32_t* lb;                 // view 0's look-back bookkeeping (in the arena of the keys: job.stride apart), pass after pass
    size_t pass_words;
    uint32_t* ghist;              // view 0's digit totals [passes][RADIX], cleared at the start of the frame
    size_t ghist_stride;
    uint64_t* counters;           // view 0's CNT_* words
    size_t cnt_stride;
    uint64_t* host_land;          // mapped host landing zone or NULL
    const uint32_t* pre_minmax = nullptr;   // depth sort: k_preprocess' per-workgroup key extremes
    uint32_t* tile_count = nullptr;         // tile sort: [T] pairs per tile, cleared at the start of the frame
};
int launch_depth_sort_lookback(const Launch& L, const SortJob& job, const LbJob& lj);
int launch_tile_sort_lookback(const Launch& L, const SortJob& job, const LbJob& lj, int T, int end_bit, int* result_buffer, bool key16);
// which way the sorts of a submission run: 0 three launches per pass, 1 look-back passes, 2 (default) look-back for submissions of up
// to lookback_max_views() views.  set < 0 only queries.  (api.hip; GSR_SORT_MODE / GSR_SORT_LB_VIEWS in the environment)
// How the kernels whose workgroups wait for lower-numbered workgroups of the same launch (pair emission, look-back scatter) number
// themselves.  1 (default): by a ticket drawn with an atomic when the workgroup STARTS -- a lower number has started earlier, so the
// waits below can only be for workgroups that are running or done, whatever order the hardware dispatches blockIdx in (HIP promises
// none; rocPRIM's look-back scans draw tickets for the same reason).  The price: every workgroup of a launch hits ONE address, and
// same-address atomics from eight XCDs complete at about one per 13-70 ns: 10 us of a single view's 52-us emission.
// 0 (GSR_TICKETS=0, opt-in): by blockIdx -- correct as long as the hardware hands the workgroups of a launch to each XCD in
// increasing order (it does today: workgroup i goes to XCD i mod 8, each XCD takes its share in order; the lowest-numbered unfinished
// workgroup is then always resident or next in line, waits for nothing unfinished, and by induction every wait ends), which is an
// observation about this part, not a contract.  Every wait is bounded either way (CNT_STALL: the frame fails, it does not hang).
int block_tickets(int set);
int sort_mode(int set);
int lookback_max_views(int set);
inline bool tile_keys16(int T) { return T <= 65536; }
// binning.hip
int launch_duplicate(const Launch& L, int P, const Batch& B, int gridx, bool key16, uint64_t* host_land);
int launch_tile_ranges(const Launch& L, const Batch& B, const uint32_t* sorted_keys, int T, bool key16);
int launch_tile_order(const Launch& L, const Batch& B, int T);
int launch_ranges_order(const Launch& L, const Batch& B, int T);   // both from iv.tile_count (look-back tile sort), one launch
int launch_bwd_items(const Launch& L, const Batch& B, int T, int P);
// render_fwd.hip / render_bwd.hip
// point_list: view 0's sorted ids (binning arena); with_ckpt: record the chunk-boundary state for the backward pass
// Extra channels composited by the forward render with the alphas of the colour pass (4 or 8 per call).
struct ExtraChannels {
    int nx;                   // 4 or 8
    const float* values;      // [P][nx], or [V][P][nx] with view_stride = P * nx
    const float* view_scale;  // [V][nx] or NULL
    const float* bg;          // [nx]
    float* out;               // [V][nx][H][W]
    size_t view_stride;       // floats between consecutive views' value arrays (0: one array shared by the views)
    const float* values_hi = nullptr;   // nx = 8, split layout: channels 4..7 as [V][P][4], channels 0..3 in `values` as [P][4] shared
    size_t hi_view_stride = 0;
};
int launch_render_forward(const Launch& L, const gsr_params& p, const Batch& B, const uint32_t* point_list, float* out_color,
                          bool with_ckpt, const ExtraChannels* X = nullptr);
int launch_render_backward(const Launch& L, const gsr_params& p, const Batch& B, const uint32_t* point_list, const float* dL_dpix);
int backward_subquadrant_moments(int set);   // render_bwd.hip: set >= 0 stores; 1 = moments about the sub-quadrant centres
int forward_half_views(int set);   // render_fwd.hip: set >= 0 stores; returns the views per submission up to which the half-quadrant forward runs
int selftest_mm(hipStream_t stream, float* d_scratch256);   // the matrix-core pixel contraction of the render backward
#ifdef GSR_STATS
int debug_bwd_stats(unsigned long long* out8, int reset);   // instrumentation build only
int debug_dup_times(unsigned long long* out8, int reset);
int debug_scatter_times(unsigned long long* out8, int reset);
int debug_fwd_times(unsigned long long* out8, int reset);
int debug_fwd_records(unsigned* out, int n);
int debug_bwd_times(unsigned long long* out8, int reset);
int debug_bwd_records(unsigned* out, int n);
#endif
// preprocess_bwd.hip
// reads grad_rec of every view; writes the user-facing gradients summed over the views of the batch
int launch_preprocess_backward(const Launch& L, const gsr_params& p, const Batch& B, const int* radii, float* dL_dmean2D,
                               float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                               float* dL_dscale, float* dL_drot);

// device helper: clear [p, p + bytes) (16-B aligned, multiple of 16) with the calling grid's threads
__device__ __forceinline__ void zero_region(char* p, size_t bytes, size_t gtid, size_t nthreads)
{
    uint4* q = reinterpret_cast<uint4*>(p);
    const size_t n = bytes / 16;
    for (size_t i = gtid; i < n; i += nthreads) q[i] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace gsr
