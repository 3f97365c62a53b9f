// common.hpp -- shared by every translation unit of libgsr_hip.so (gfx950 only).
//
// Floating-point policy: the whole library is compiled with -ffp-contract=off and performs the
// per-Gaussian / per-pixel arithmetic in the order the reference SOURCE writes it (one IEEE
// rounding per written operation; +,-,*,/ and sqrt are correctly rounded under hipcc defaults), so
// that integer outputs (radii, tile counts, sorted lists, ranges, n_contrib) are reproducible
// bit-for-bit against the CPU oracle and the reference build ("Numerics").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsr.h"

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

// ---- scan geometry ----------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Per-Gaussian packed splat record: what the render kernels gather per list entry.  Padded to one 64-B cache line, so a
// gather touches exactly one line (a 48-B record straddles two half of the time) and preprocess writes whole lines.
//   q0 = (x, y, conic.x, conic.y)   q1 = (conic.z, opacity, r, g)   q2 = (b, depth, 0, 0)   q3 = padding
struct __attribute__((aligned(64))) Splat {
    float4 q0, q1, q2, q3;
};

// ---- arena views (device pointers carved out of the caller's opaque buffers) -------------------
struct GeomView {
    Splat* splat;             // [P]
    uint32_t* tiles_touched;  // [P]
    uint2* rect;              // [P] x = minx | miny<<16, y = maxx | maxy<<16  (tile units)
    uint8_t* clamped;         // [P] bit k = colour channel k was clamped at 0 (CR/forward.cu:66-69)
    uint32_t* dkey[2];        // [P] depth-bit keys, ping-pong (dkey[0] is also preprocess' output)
    uint32_t* dval[2];        // [P] Gaussian ids, ping-pong; after 4 passes dval[0] = ids in depth order
    uint32_t* dup_offset;     // [P] exclusive prefix of tiles_touched in depth order
    uint32_t* hist;           // [RADIX * nblk(P)] per-workgroup digit counts
    uint32_t* totals;         // [RADIX]
    uint32_t* scan_tmp;       // [nscan(P) + 1]
    uint64_t* counters;       // [8]: 0 = num_rendered, 1 = trap flag
    size_t bytes;
};

// Backward work items: a tile's consumed list is cut into chunks of BWD_CHUNK entries that different waves walk
// concurrently; the forward pass leaves the per-pixel state (T, accumulated colour) at every chunk boundary it crosses.
constexpr int BWD_CHUNK = 1024;
constexpr int BWD_CHUNK_SHIFT = 10;
constexpr int BWD_MAX_CHUNKS = 16;   // per tile; the last one takes whatever is left

struct BinView {
    uint32_t* key[2];  // [R] tile ids, ping-pong (stored as uint16_t when the image has <= 65536 tiles)
    uint32_t* val[2];  // [R] Gaussian ids, ping-pong
    uint32_t* hist;    // [RADIX * nblk(R)]
    uint32_t* totals;  // [RADIX]
    float4* ckpt;      // [(R / BWD_CHUNK + 2) * 256] forward state (T, C.rgb) per pixel of a tile at list position
                       // range.x + k * BWD_CHUNK, slot (range.x >> BWD_CHUNK_SHIFT) + k  (unique: lists do not overlap)
    size_t bytes;
};

struct ImageView {
    uint2* ranges;        // [T]
    float* final_T;       // [N]
    uint32_t* n_contrib;  // [N]
    uint32_t* tile_order; // [T] tiles by descending list length (forward render launch order)
    uint32_t* tile_need;  // [T] entries walked by the forward render
    float* accum;         // [3N] colour accumulated by the forward render, without the background term
    uint32_t* bwd_items;  // [BWD_MAX_CHUNKS * T] backward work items: tile | chunk << 20, heaviest first
    uint32_t* bwd_count;  // [4] number of items
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
    const size_t nblk = (size_t)div_up((int64_t)p, RS_TILE);
    const size_t nscan = (size_t)div_up((int64_t)p, SCAN_TILE);
    carve(cur, g.splat, p);
    carve(cur, g.tiles_touched, p);
    carve(cur, g.rect, p);
    carve(cur, g.clamped, p);
    carve(cur, g.dkey[0], p);
    carve(cur, g.dkey[1], p);
    carve(cur, g.dval[0], p);
    carve(cur, g.dval[1], p);
    carve(cur, g.dup_offset, p);
    carve(cur, g.hist, RADIX * nblk);
    carve(cur, g.totals, (size_t)RADIX);
    carve(cur, g.scan_tmp, nscan + 1);
    carve(cur, g.counters, (size_t)8);
    g.bytes = (size_t)(cur - reinterpret_cast<char*>(base));
    return g;
}

inline BinView bin_view(void* base, int64_t R)
{
    BinView b;
    char* cur = reinterpret_cast<char*>(base);
    const size_t r = (size_t)(R > 0 ? R : 1);
    const size_t nblk = (size_t)div_up((int64_t)r, RS_TILE);
    carve(cur, b.key[0], r);
    carve(cur, b.key[1], r);
    carve(cur, b.val[0], r);
    carve(cur, b.val[1], r);
    carve(cur, b.hist, RADIX * nblk);
    carve(cur, b.totals, (size_t)RADIX);
    carve(cur, b.ckpt, (r / BWD_CHUNK + 2) * 256);
    b.bytes = (size_t)(cur - reinterpret_cast<char*>(base));
    return b;
}

inline ImageView image_view(void* base, int W, int H)
{
    ImageView v;
    char* cur = reinterpret_cast<char*>(base);
    const size_t N = (size_t)W * (size_t)H;
    const size_t T = (size_t)((W + TILE_X - 1) / TILE_X) * (size_t)((H + TILE_Y - 1) / TILE_Y);
    carve(cur, v.ranges, T ? T : 1);
    carve(cur, v.final_T, N ? N : 1);
    carve(cur, v.n_contrib, N ? N : 1);
    carve(cur, v.tile_order, T ? T : 1);
    carve(cur, v.tile_need, T ? T : 1);
    carve(cur, v.accum, 3 * (N ? N : 1));
    carve(cur, v.bwd_items, (size_t)BWD_MAX_CHUNKS * (T ? T : 1));
    carve(cur, v.bwd_count, (size_t)4);
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

int check_launch(const Launch& L, const char* what);  // api.hip

// preprocess.hip
int launch_preprocess(const Launch& L, const gsr_params& p, const GeomView& g, int* radii);
int launch_recolor(const Launch& L, const gsr_params& p, const GeomView& g);
int launch_mark_visible(const Launch& L, int P, const float* means3D, const float* view, uint8_t* present);
// sort.hip
// key16: the key arrays hold uint16_t (tile ids of images up to 65536 tiles: 14 instead of 20 B per pair per pass)
int launch_radix_sort_pairs(const Launch& L, int64_t n, uint32_t* key[2], uint32_t* val[2], bool iota_vals,
                            int end_bit, uint32_t* hist, uint32_t* totals, int* result_buffer, bool key16 = false);
inline bool tile_keys16(int T) { return T <= 65536; }
int launch_offsets_scan(const Launch& L, int P, const uint32_t* order, const uint32_t* tiles_touched,
                        uint32_t* dup_offset, uint32_t* scan_tmp, uint64_t* total_out);
// binning.hip
int launch_duplicate(const Launch& L, int P, const GeomView& g, const uint32_t* order, int gridx, uint32_t* keys,
                     uint32_t* vals, bool key16);
int launch_tile_ranges(const Launch& L, int64_t R, const uint32_t* sorted_keys, uint2* ranges, int T, bool key16);
int launch_tile_order(const Launch& L, const ImageView& iv, int T);
int launch_bwd_items(const Launch& L, const ImageView& iv, int T);
// render_fwd.hip / render_bwd.hip
// ckpt: chunk-boundary state for the backward pass (NULL: not recorded, e.g. inference / colour-only re-render)
int launch_render_forward(const Launch& L, const gsr_params& p, const GeomView& g, const uint32_t* point_list,
                          const ImageView& iv, float* out_color, float4* ckpt);
// grad_rec: [P][GRAD_REC_WORDS] zero-filled accumulation records, one 64-B line per Gaussian:
//   0 mean2D.x  1 mean2D.y  2 conic.x  3 conic.y  4 conic.w  5..7 colour r g b  8 opacity  (9..15 unused)
// The nine float atomics a (quadrant, entry) issues land in ONE cache line instead of four arrays' worth
// (scripts/probe/atomic_probe.hip: 4x the atomic throughput).
constexpr int GRAD_REC_WORDS = 16;
int launch_render_backward(const Launch& L, const gsr_params& p, const GeomView& g, const uint32_t* point_list,
                           const ImageView& iv, const float4* ckpt, const float* dL_dpix, float* grad_rec);
int selftest_reduce(hipStream_t stream, float* d_scratch128);
// preprocess_bwd.hip
// reads grad_rec; writes the user-facing dL_dmean2D [P,3], dL_dopacity [P], dL_dcolor [P,3] (copies of the record's
// fields) for every Gaussian besides the geometric gradients
int launch_preprocess_backward(const Launch& L, const gsr_params& p, const GeomView& g, const int* radii,
                               const float* grad_rec, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor,
                               float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot);

}  // namespace gsr
