"""numpy model of ONE quadrant's second-moment sum  sum_p q_p dx_p^2  (dx = bx - cx, cx = pixel offset from the quadrant centre),
two float32 formulations against float64:
  reference  every pixel's product q dx dx rounded on its own, then a sequential float sum (the atomics);
  moments    S1 = sum q, Sx = sum q cx, Sxx = sum q cx^2 by sequential float FMAs (what the matrix cores do), then
             bx (bx S1 - Sx) - bx Sx + Sxx in float (render_bwd.hip, the flush);
  moments64  the same with exact sums rounded to float once (a lower bound for any float32 moment representation).
Prints rms error of `moments` and `moments64` in units of the reference's rms error:
  * weights spread over the whole quadrant (a splat larger than the quadrant): ratio 1.0 at any distance bx;
  * weights concentrated in a disc of sigma << |bx| (a small splat off the quadrant centre): the moment forms lose (bx / sigma)^2.
usage: python scripts/debug/moment_shift_model.py"""
import numpy as np

rng = np.random.default_rng(0)
f, d = np.float32, np.float64


def fma(a, b, c):
    return f(d(a) * d(b) + d(c))


def trial(bx, sigma=None, n=400):
    cx = (np.arange(64) % 8 - 3.5).astype(f)
    cy = (np.arange(64) // 8 - 3.5).astype(f)
    e_ref, e_mom, e_m64 = [], [], []
    for _ in range(n):
        q = rng.standard_normal(64).astype(f)
        if sigma is not None:       # a small splat centred at (bx, 0) relative to the quadrant centre: weights fall off like its footprint
            w = np.exp(-0.5 * (((d(bx) - cx) ** 2 + cy.astype(d) ** 2) / sigma ** 2))
            q = (q * w).astype(f)
            q[w < 1e-3] = 0
        dx = (f(bx) - cx).astype(f)
        exact = np.sum(q.astype(d) * dx.astype(d) ** 2)
        t = ((q * dx).astype(f) * dx).astype(f)
        s = f(0)
        for v in t:
            s = f(s + v)
        e_ref.append(s - exact)
        S1 = Sx = Sxx = f(0)
        for p in range(64):
            S1, Sx, Sxx = fma(f(1), q[p], S1), fma(cx[p], q[p], Sx), fma(f(cx[p] * cx[p]), q[p], Sxx)
        Dx = fma(f(bx), S1, -Sx)
        e_mom.append(fma(f(bx), Dx, fma(-f(bx), Sx, Sxx)) - exact)
        S1d, Sxd, Sxxd = np.sum(q.astype(d)), np.sum(q.astype(d) * cx), np.sum(q.astype(d) * cx.astype(d) ** 2)
        e_m64.append(d(bx) ** 2 * f(S1d) - 2 * d(bx) * f(Sxd) + f(Sxxd) - exact)
    r = np.sqrt(np.mean(np.square(e_ref))) + 1e-300
    return np.sqrt(np.mean(np.square(e_mom))) / r, np.sqrt(np.mean(np.square(e_m64))) / r


print("weights over the whole quadrant            bx   moments/ref  moments64/ref")
for bx in (0.3, 6.0, 100.0, 3000.0):
    print("                                     %8.1f   %8.2f   %8.2f" % ((bx,) + trial(bx)))
print("small splat at distance bx, sigma          bx  sigma   moments/ref  moments64/ref   (bx/sigma)^2")
for bx, sg in ((3.0, 0.6), (3.0, 1.2), (5.0, 0.8), (5.0, 1.6), (1.0, 0.6)):
    a, b = trial(bx, sg)
    print("                                     %8.1f  %5.1f   %8.2f   %8.2f   %8.1f" % (bx, sg, a, b, (bx / sg) ** 2))
