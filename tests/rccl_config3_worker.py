"""Helper of test_gpu_bench.py: BASELINE configs[3] at full size on ONE rank with the `nccl` backend (= RCCL on ROCm): the 8
camera views of synth-THuman-800K at 1920x1080 rendered by the HIP rasterizer (one rasterize_views submission), sent through
pcrender.multiview's frame gather with FULL-SIZE gather buffers ([8,3,1080,1920] fp32 = 199 MB on the root), in both gather
modes, and checked bit for bit against per-view GaussianRasterizer calls (the reference caller's loop,
simple_raw_render.py:259-278)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-pcloud-render_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29578")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views
    from pcrender import camera, multiview, synth
    W, H, V = 1920, 1080, 8
    cloud = synth.make_cloud("synth-THuman-800K", seed=0)
    g = synth.make_gaussians(cloud, profile="training", seed=1)
    views = camera.circle_views(V, fov_deg=45.0, width_px=W, height_px=H)
    bg = torch.ones(3, device=dev)
    settings = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=v["tanfovx"], tanfovy=v["tanfovy"], bg=bg, scale_modifier=1.0,
        viewmatrix=v["viewmatrix"].to(dev), projmatrix=v["projmatrix"].to(dev), sh_degree=g["sh_degree"], campos=v["campos"].to(dev),
        prefiltered=False, debug=False) for v in views]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    m3, shs, op, sc, ro = t(g["means3D"]), t(g["shs"]), t(g["opacities"]), t(g["scales"]), t(g["rotations"])
    with torch.no_grad():
        batch, _ = rasterize_views(m3, torch.zeros_like(m3), op, settings, shs=shs, scales=sc, rotations=ro)
        assert batch.shape == (V, 3, H, W)
        for mode in ("collective", "p2p"):
            out = multiview.render_views(lambda v: batch[v], V, dst=0, mode=mode)     # this rank owns all 8 views at world 1
            assert out.is_cuda and out.shape == (V, 3, H, W) and out.numel() * 4 == 8 * 3 * 1080 * 1920 * 4
            assert torch.equal(out, batch), mode
        for v in (0, 3, 7):
            one, _ = GaussianRasterizer(settings[v])(means3D=m3, means2D=torch.zeros_like(m3), opacities=op, shs=shs, scales=sc, rotations=ro)
            assert torch.equal(one, out[v]), v
    gsum = [torch.ones((800000, 3), device=dev)]
    multiview.reduce_gradients(gsum)
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_CONFIG3_OK")


if __name__ == "__main__":
    main()
