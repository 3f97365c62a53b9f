"""debug: dump the forward of a named test scene (tests/util.py) to an .npz; compare two dumps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]
import numpy as np
if sys.argv[1] == "dump":
    import torch, util
    s = util.build_scene(sys.argv[2])
    p, _ = util.run_product(s, torch.device("cuda:0"))
    np.savez(sys.argv[3], **{k: v for k, v in p.items() if isinstance(v, np.ndarray)})
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        if a[k].shape != b[k].shape:
            print(k, "shape", a[k].shape, b[k].shape); continue
        d = a[k] != b[k]
        print(k, "differs in", int(d.sum()), "of", d.size)
    d = (a["out_color"] != b["out_color"]).any(axis=0) | (a["n_contrib"] != b["n_contrib"])
    ys, xs = np.nonzero(d)
    H, W = d.shape
    print("image", W, "x", H, "bad pixels", len(ys))
    tiles = {}
    for y, x in zip(ys, xs):
        key = (y // 16, x // 16, (y % 16) // 8, (x % 16) // 8)
        tiles[key] = tiles.get(key, 0) + 1
    r = a["ranges"].reshape(-1, 2)
    gx = (W + 15) // 16
    for (ty, tx, qy, qx), n in sorted(tiles.items()):
        t = ty * gx + tx
        print("tile (%d,%d) q(%d,%d): %d bad px, list length %d" % (tx, ty, qx, qy, n, r[t, 1] - r[t, 0]))
    print("all list lengths:", (r[:, 1] - r[:, 0]).tolist())
