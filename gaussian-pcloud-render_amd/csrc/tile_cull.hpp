// tile_cull.hpp -- exact-safe footprint test used by both render kernels when they stage a tile's list.
//
// A 16x16 tile's list (built from each Gaussian's 3-sigma bounding SQUARE in tile units, reference
// CR/auxiliary.h:46-56) holds many entries that cannot reach alpha >= 1/255 at any pixel of a given 8x8
// quadrant.  The reference evaluates them anyway and skips them pixel by pixel (CR/forward.cu:336-347); here
// the staging lane decides once per (entry, quadrant) and a wave only walks the entries that may matter.  The test
// rectangle is the bounding box of the pixels that can still be affected (all 64 at first; it shrinks as pixels of the
// quadrant terminate in the forward pass, and covers only the pixels that consumed that deep in the backward pass).
// Skipping is invisible in the results as long as it is CONSERVATIVE: an entry is dropped for a quadrant only
// when  max over the quadrant's pixel centres of power(d) + E  <  log(1/(255*opacity)),  where
// power(d) = -0.5 (A dx^2 + C dy^2) - B dx dy is concave for a positive-definite conic, its maximum over a
// rectangle is found exactly on the centre / the four edges, and E bounds the fp32 rounding of both this
// bound and the per-pixel evaluation (~170 ulp of the largest term, plus 1e-4 absolute).
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// May the Gaussian (mean (mx,my), conic (A,B,C), opacity o) reach alpha >= 1/255 at a pixel centre inside the
// rectangle [x0, x1] x [y0, y1] (pixel-centre coordinates, inclusive)?  false = provably not.
__device__ __forceinline__ bool may_touch_rect(float mx, float my, float A, float B, float C, float o, float x0, float y0,
                                               float x1, float y1)
{
    // NaN anywhere must mean "keep".  No explicit test is needed: every comparison below is written so that it is
    // false for NaN operands and false means keep (NaN opacity: not <= 0, thr = NaN; NaN conic: not provably concave;
    // NaN mean: E = NaN), and fmaxf drops a NaN candidate without making the bound larger than a real one.
    if (o <= 0.f) return false;  // alpha = o*exp(..) <= 0 < 1/255 at every pixel
    if (!(A > 0.f && C > 0.f && A * C - B * B > 0.f)) return true;  // not provably concave: keep
    const float thr = -__logf(255.0f * o);  // alpha >= 1/255  <=>  power >= thr
    // d = mean - pixel over the rectangle
    const float dxl = mx - x1, dxh = mx - x0, dyl = my - y1, dyh = my - y0;
    float m;
    if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) {
        m = 0.f;  // the mean lies inside the rectangle
    } else {
        m = -3.0e38f;
        // Along an edge the maximiser is -B e / C (or / A), clamped to the edge.  A 1-ulp reciprocal is enough: evaluating
        // the concave power a relative 1e-7 away from its maximiser lowers the value by ~1e-14 of its terms, far inside E.
        const float nB_over_C = -B * __builtin_amdgcn_rcpf(C), nB_over_A = -B * __builtin_amdgcn_rcpf(A);
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float ex = e ? dxh : dxl;
            const float yy = clampf(nB_over_C * ex, dyl, dyh);
            m = fmaxf(m, -0.5f * (A * ex * ex + C * yy * yy) - B * ex * yy);
            const float ey = e ? dyh : dyl;
            const float xx = clampf(nB_over_A * ey, dxl, dxh);
            m = fmaxf(m, -0.5f * (A * xx * xx + C * ey * ey) - B * xx * ey);
        }
    }
    const float ax = fmaxf(fabsf(dxl), fabsf(dxh)), ay = fmaxf(fabsf(dyl), fabsf(dyh));
    const float E = 1.0e-5f * (A * ax * ax + C * ay * ay + fabsf(B) * ax * ay) + 1.0e-4f + 1.0e-5f * fabsf(thr);
    return !(m + E < thr);
}

__device__ __forceinline__ bool may_touch_8x8(float mx, float my, float A, float B, float C, float o, float x0, float y0)
{
    return may_touch_rect(mx, my, A, B, C, o, x0, y0, x0 + 7.f, y0 + 7.f);
}

// ---- footprint clipping of a Gaussian's tile rectangle (pair emission) ---------------------------------------------
// The reference emits one (tile, Gaussian) pair for every tile of the bounding SQUARE of radius ceil(3 sqrt(lambda_max))
// (CR/auxiliary.h:46-56, CR/forward.cu:255-262), whatever the splat's shape and opacity.  A pixel only blends the splat when
// alpha = o exp(power) >= 1/255, i.e. inside the ellipse  d^T conic d <= 2 log(255 o);  its axis-aligned bounding box is
// |dx| <= sqrt(2 log(255 o) Sigma_xx), |dy| <= sqrt(2 log(255 o) Sigma_yy)  with Sigma the (dilated) 2D covariance the conic
// was inverted from.  Tiles of the square that lie outside that box hold no pixel the reference would blend (it evaluates
// them and skips them pixel by pixel, CR/forward.cu:336-347), so leaving them out of the lists changes no pixel, no
// transmittance and no gradient -- only the private lists get shorter (a third of the pairs on the benchmark cloud:
// anisotropic splats, opacities below one).  As for the quadrant test above, the clipping must be CONSERVATIVE against the
// fp32 arithmetic of the per-pixel evaluation:
//   * a blended pixel has power_fp32 >= -tau - 1e-6 (tau = log(255 o); rounding of o * exp(), of exp and of logf);
//   * power_fp32 differs from the exact value of the quadratic form of the fp32 conic by at most a few ulp of its largest
//     term; with |dx|, |dy| <= r + 16 inside the reference's rectangle that is far below
//     E = 1e-5 (A + C + |B|) (r + 16)^2 + 1e-4 + 1e-5 tau  (the bound of may_touch_rect);
//   * conic = adj(Sigma) / det_fp32, so the exact form is s d^T Sigma^-1 d with s = det_exact / det_fp32 >= 1 - eta,
//     eta = 4e-7 (Sxx Syy + Sxy^2) / det (cancellation in the fp32 determinant), the element roundings being inside E;
// hence |dx| <= sqrt(2 (tau + E) / (1 - eta) Sxx), widened by 1e-5 relative + 0.02 px for the square root and the products.
// Anything that is not provably an ellipse (NaN, det <= 0, eta too large) keeps the reference's rectangle.
// rect = (minx | miny << 16, maxx | maxy << 16) in tiles; returns the number of tiles left (0: the Gaussian emits nothing).
//
// Row spans.  Inside the clipped rectangle the ellipse still misses tiles -- the corners of a round footprint, most of the box of
// a thin diagonal one (a tenth of the pairs the box keeps on the benchmark cloud).  For a rectangle of at most SPAN_ROWS tile rows
// and 15 columns, `spans` receives one byte per tile row: first column (relative to minx) in the low nibble, number of columns
// in the high one.  A tile row is the strip of pixel rows y0..y0+15, i.e. dy = mean.y - y in [d0, d1]; for a given dy the ellipse
// d^T Sigma^-1 d <= k holds dx in m(dy) +- w(dy), m = (Sxy / Syy) dy, w^2 = (Sxx - Sxy^2 / Syy) (k - dy^2 / Syy), so the strip's
// pixels lie in [px - max(m + w), px + max(-m + w)]: each maximum of a concave function over [d0, d1], attained at the ellipse's
// leftmost / rightmost point if its dy lies in the strip, else at one of the strip's ends.  Same k (rounding allowances E, eta
// inside); the cancellation in w^2 is covered by 4e-7 hx^2 under the root, the rest by 0.03 px + 1e-4 hx on the interval.
// *spans_valid = false: rectangle too large for the encoding (or not clipped at all): every tile of rect is emitted.
constexpr uint32_t SPAN_ROWS = 8;
constexpr uint32_t SPANS_FLAG = 0x80000000u;   // in the fourth word of Splat::q3 (reference pair counts are < 2^27: api.hip check_params)
__device__ __forceinline__ uint32_t clip_rect_to_footprint(float px, float py, float sxx, float sxy, float syy, float det,
                                                            float A, float B, float C, float o, float r, uint2& rect,
                                                            uint32_t gridx, uint32_t gridy, uint2* spans = nullptr,
                                                            bool* spans_valid = nullptr)
{
    if (spans_valid) *spans_valid = false;
    uint32_t minx = rect.x & 0xFFFFu, miny = rect.x >> 16, maxx = rect.y & 0xFFFFu, maxy = rect.y >> 16;
    const uint32_t full = (maxx - minx) * (maxy - miny);
    if (o < 1.0f / 255.0f) return 0u;      // alpha = o exp(power <= 0) <= o < 1/255 at every pixel (entries with power > 0 are skipped)
    if (!(det > 0.f && sxx > 0.f && syy > 0.f && o <= 1.0e30f)) return full;   // (also NaN)
    const float tau = logf(255.0f * o);
    const float ext = r + 16.0f;
    const float E = 1.0e-5f * ((A + C + fabsf(B)) * ext * ext) + 1.0e-4f + 1.0e-5f * tau;
    const float eta = 4.0e-7f * (sxx * syy + sxy * sxy) / det;
    if (!(eta < 0.25f) || !(E < 1.0e30f)) return full;
    const float k = 2.0f * (tau + E) / (1.0f - eta);
    const float hx = sqrtf(k * sxx) * (1.0f + 1.0e-5f) + 0.02f, hy = sqrtf(k * syy) * (1.0f + 1.0e-5f) + 0.02f;
    if (!(hx >= 0.f && hy >= 0.f)) return full;
    // pixel centres are integers: columns ceil(px - hx) .. floor(px + hx), rows likewise; their tiles, inside the rectangle
    const float lim = 4.0e6f;
    const float x_lo = fminf(fmaxf(ceilf(px - hx), -lim), lim), x_hi = fminf(fmaxf(floorf(px + hx), -lim), lim);
    const float y_lo = fminf(fmaxf(ceilf(py - hy), -lim), lim), y_hi = fminf(fmaxf(floorf(py + hy), -lim), lim);
    if (!(x_lo == x_lo && x_hi == x_hi && y_lo == y_lo && y_hi == y_hi)) return full;   // NaN mean
    const int tx0 = (int)floorf(x_lo * (1.0f / 16.0f)), tx1 = (int)floorf(x_hi * (1.0f / 16.0f)) + 1;
    const int ty0 = (int)floorf(y_lo * (1.0f / 16.0f)), ty1 = (int)floorf(y_hi * (1.0f / 16.0f)) + 1;
    const uint32_t cx0 = tx0 < 0 ? 0u : ((uint32_t)tx0 > gridx ? gridx : (uint32_t)tx0);
    const uint32_t cx1 = tx1 < 0 ? 0u : ((uint32_t)tx1 > gridx ? gridx : (uint32_t)tx1);
    const uint32_t cy0 = ty0 < 0 ? 0u : ((uint32_t)ty0 > gridy ? gridy : (uint32_t)ty0);
    const uint32_t cy1 = ty1 < 0 ? 0u : ((uint32_t)ty1 > gridy ? gridy : (uint32_t)ty1);
    minx = minx > cx0 ? minx : cx0;
    maxx = maxx < cx1 ? maxx : cx1;
    miny = miny > cy0 ? miny : cy0;
    maxy = maxy < cy1 ? maxy : cy1;
    if (maxx <= minx || maxy <= miny) return 0u;
    rect = make_uint2(minx | (miny << 16), maxx | (maxy << 16));
    const uint32_t w = maxx - minx, h = maxy - miny;
    if (spans == nullptr || h > SPAN_ROWS || w > 15u) return w * h;
    // ---- row spans
    const float inv_syy = 1.0f / syy;
    const float slope = sxy * inv_syy;                       // m(dy) = slope * dy
    const float v = fmaxf(sxx - sxy * slope, 0.f);           // Sxx - Sxy^2 / Syy (conditional variance of dx)
    const float hx_raw = sqrtf(k * sxx), hy_raw = sqrtf(k * syy);
    const float root_pad = 4.0e-7f * (k * sxx);              // cancellation in v and in k - dy^2 / Syy, both <~ 1e-7 hx^2 in w^2
    const float pad = 0.03f + 1.0e-4f * hx_raw;
    const float dstar = hx_raw * sxy / sxx;                  // dy of the ellipse's leftmost point (dx = +hx); rightmost: -dstar
    if (!(hx_raw >= 0.f && hy_raw >= 0.f && v == v && dstar == dstar)) return w * h;
    uint32_t lo = 0u, hi = 0u, total = 0u;
    // The strips are taken half a pixel wider than their pixel rows, [16 j - 0.5, 16 j + 15.5], so that consecutive tile rows share
    // a boundary and its square root: h + 1 evaluations of (m, w) for h rows.  A boundary beyond the ellipse's own extent in y is
    // pulled back to it (w = 0 there: the ellipse's top / bottom point).
    const float hyc = hy_raw + 0.02f;
    float d_raw = py - ((float)(miny * 16u) - 0.5f);                                    // dy at the upper boundary of row 0
    float d_hi = fminf(fmaxf(d_raw, -hyc), hyc);
    float w_b = __builtin_amdgcn_sqrtf(v * fmaxf(k - d_hi * d_hi * inv_syy, 0.f) + root_pad);
    float er_hi = -slope * d_hi + w_b, el_hi = slope * d_hi + w_b;                      // (pixel x - px) and (px - pixel x) there
    const int iminx = (int)minx, imaxx = (int)maxx;
    for (uint32_t row = 0; row < h; row++) {
        d_raw -= 16.0f;                                                                 // the next boundary: 16 pixel rows further down
        const float d_lo = fminf(fmaxf(d_raw, -hyc), hyc);
        const float w_n = __builtin_amdgcn_sqrtf(v * fmaxf(k - d_lo * d_lo * inv_syy, 0.f) + root_pad);
        const float er_lo = -slope * d_lo + w_n, el_lo = slope * d_lo + w_n;
        float right = fmaxf(er_hi, er_lo), left = fmaxf(el_hi, el_lo);
        if (-dstar >= d_lo - 0.05f && -dstar <= d_hi + 0.05f) right = fmaxf(right, hx_raw);
        if (dstar >= d_lo - 0.05f && dstar <= d_hi + 0.05f) left = fmaxf(left, hx_raw);
        const float xl = fminf(fmaxf(ceilf(px - (left + pad)), -lim), lim), xr = fminf(fmaxf(floorf(px + (right + pad)), -lim), lim);
        if (!(xl == xl && xr == xr)) return w * h;           // NaN: no row clipping
        const int ta = (int)xl >> 4, tb = ((int)xr >> 4) + 1;        // (arithmetic shift = floor(x / 16), also below zero)
        const int ca = ta < iminx ? iminx : (ta > imaxx ? imaxx : ta), cb = tb < iminx ? iminx : (tb > imaxx ? imaxx : tb);
        const uint32_t count = cb > ca ? (uint32_t)(cb - ca) : 0u, first = cb > ca ? (uint32_t)(ca - iminx) : 0u;
        const uint32_t byte = first | (count << 4);
        if (row < 4u) lo |= byte << (8u * row); else hi |= byte << (8u * (row - 4u));
        total += count;
        d_hi = d_lo; er_hi = er_lo; el_hi = el_lo;
    }
    *spans = make_uint2(lo, hi);
    *spans_valid = true;
    return total;
}

// Bounding rectangle, in lane coordinates (x = lane & 7, y = lane >> 3), of the lanes set in a ballot of an 8x8
// quadrant wave.  Scalar code.  mask must not be 0.
__device__ __forceinline__ void live_box(uint64_t mask, int& xmin, int& ymin, int& xmax, int& ymax)
{
    ymin = (int)__builtin_ctzll(mask) >> 3;
    ymax = (63 - (int)__builtin_clzll(mask)) >> 3;
    uint32_t rows = (uint32_t)mask | (uint32_t)(mask >> 32);   // fold the 8 rows onto one byte
    rows |= rows >> 16;
    rows |= rows >> 8;
    rows &= 0xFFu;
    xmin = (int)__builtin_ctz(rows);
    xmax = 31 - (int)__builtin_clz(rows);
}

// 4-bit mask over the quadrants q = (qx | qy<<1) of the 16x16 tile whose first pixel is (tile_px, tile_py).
__device__ __forceinline__ uint32_t quadrant_mask(float mx, float my, float A, float B, float C, float o, float tile_px,
                                                  float tile_py)
{
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (may_touch_8x8(mx, my, A, B, C, o, tile_px + (float)((q & 1) * 8), tile_py + (float)((q >> 1) * 8))) mask |= 1u << q;
    return mask;
}

}  // namespace gsr
