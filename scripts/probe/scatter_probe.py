"""Is a COUNTING sort of the (tile, Gaussian) pairs viable on this part?  

A single-pass tile sort would write every pair's 4-byte Gaussian id straight to its final slot in the tile lists: R scattered
4-byte stores (a chunk of depth-consecutive Gaussians contributes ~10 consecutive slots to each tile it touches), where the radix
scatter writes digit runs that the LDS reorder made contiguous.  This probe takes the REAL permutation of the headline workload
(emission order -> sorted position, rebuilt with torch from the frame's geometry) and times nothing but that store pattern,
against a coalesced copy of the same bytes, to see what the scattered stores cost before anything is built around them.

usage: python scripts/probe/scatter_probe.py [--points N] [--width W --height H]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--workload", default="synth-THuman-800K")
    ap.add_argument("--chunk", type=int, default=1024)
    a = ap.parse_args()
    from pcrender import camera, synth
    from diff_gaussian_rasterization import _native as N
    dev = torch.device("cuda:0")
    cloud = synth.make_cloud(a.workload, seed=0, P=a.points)
    g = synth.make_gaussians(cloud, profile="training", seed=1)
    W, H = a.width, a.height
    v = camera.circle_views(12, fov_deg=45.0, width_px=W, height_px=H)[1]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    e = torch.empty(0)
    args = (torch.ones(3, device=dev), t(g["means3D"]), e, t(g["opacities"]), t(g["scales"]), t(g["rotations"]), 1.0, e,
            v["viewmatrix"].reshape(4, 4).to(dev), v["projmatrix"].reshape(4, 4).to(dev), v["tanfovx"], v["tanfovy"], H, W, t(g["shs"]),
            g["sh_degree"], v["campos"].to(dev), False, False)
    R, color, radii, geom, binning, img = N.rasterize_gaussians(*args, need_backward=False)
    P = g["means3D"].shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    m2 = N.query("MEANS2D", P, W, H, R, geom, binning, img)
    dep = N.query("DEPTHS", P, W, H, R, geom, binning, img)
    r = radii.to(torch.float32)
    vis = radii > 0
    # getRect (auxiliary.h:46-56)
    x0 = torch.clamp(((m2[:, 0] - r) / 16).to(torch.int64), 0, gx); x1 = torch.clamp(((m2[:, 0] + r + 15) / 16).to(torch.int64), 0, gx)
    y0 = torch.clamp(((m2[:, 1] - r) / 16).to(torch.int64), 0, gy); y1 = torch.clamp(((m2[:, 1] + r + 15) / 16).to(torch.int64), 0, gy)
    cnt = torch.where(vis, (x1 - x0) * (y1 - y0), torch.zeros_like(x0))
    assert int(cnt.sum()) == R, (int(cnt.sum()), R)
    order = torch.sort(dep.view(torch.int32).to(torch.int64) * (1 << 21) + torch.arange(P, device=dev), stable=True)[1]   # depth bits, then id
    order = order[vis[order]]
    c = cnt[order]
    gid = torch.repeat_interleave(order, c)                               # Gaussian of every pair, emission order
    start = torch.cumsum(c, 0) - c
    k = torch.arange(R, device=dev) - torch.repeat_interleave(start, c)   # index of the pair inside its rectangle
    w = (x1 - x0)[gid]
    tile = (y0[gid] + k // w) * gx + x0[gid] + k % w
    dest = torch.empty(R, dtype=torch.int64, device=dev)
    dest[torch.sort(tile, stable=True)[1]] = torch.arange(R, device=dev)   # emission index -> final slot
    want = N.query("POINT_LIST", P, W, H, R, geom, binning, img)
    out = torch.empty(R, dtype=torch.int32, device=dev)
    vals = gid.to(torch.int32)
    out[dest] = vals
    assert torch.equal(out, want), "rebuilt permutation does not reproduce the library's lists"
    dest32 = dest.to(torch.int32)
    # how scattered is it: slots a chunk of `chunk` depth-consecutive Gaussians contributes per tile it touches
    chunk_of_pair = torch.repeat_interleave(torch.arange(order.numel(), device=dev) // a.chunk, c)
    key = chunk_of_pair * (gx * gy) + tile
    runs = torch.unique(key).numel()
    print("R = %d pairs, %d Gaussians visible, %d tiles; chunk of %d Gaussians: %.1f consecutive slots per (chunk, tile) on average" % (
        R, order.numel(), gx * gy, a.chunk, R / runs))

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    dst64 = dest
    t_sc = timeit(lambda: out.index_copy_(0, dst64, vals))
    t_cp = timeit(lambda: out.copy_(vals))
    t_ga = timeit(lambda: torch.index_select(vals, 0, dst64, out=out))
    print("scattered 4-B stores in emission order (index_copy_):  %7.1f us   (%.0f GB/s of ids+index read, ids written)" % (t_sc, (R * 16) / t_sc / 1e3))
    print("gather instead (index_select with the inverse pattern): %7.1f us" % t_ga)
    print("coalesced copy of the same ids:                         %7.1f us" % t_cp)
    print("for scale: pair emission + two radix passes + ranges take ~120 us per view in a 12-view batch, ~200 us alone")


if __name__ == "__main__":
    main()
