"""Helper of test_gpu_bench.py: ONE rank on cuda:0 with the `nccl` backend (= RCCL on ROCm).  Runs the multi-GPU code path of
pcrender.multiview for real -- sharded render_views + frame gather, gradient all-reduce, barrier -- with device tensors
rendered by the HIP rasterizer, and checks the collectives returned what a single rank must get."""
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
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29577")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    import util
    from pcrender import multiview
    s = util.build_scene("capsule_circle")
    frames = {}

    def render_one(v):
        s2 = util.build_scene("capsule_circle")
        s2.bg[:] = 0.25 * v
        img = torch.from_numpy(util.run_product(s2, dev, light=True)[0]["out_color"]).to(dev)
        frames[v] = img
        return img

    out = multiview.render_views(render_one, 3, dst=0)
    assert out.is_cuda and out.shape == (3, 3, s.H, s.W)
    for v in range(3):
        assert torch.equal(out[v], frames[v])
    g = [torch.full((5, 3), 2.0, device=dev), None, torch.arange(4, dtype=torch.float32, device=dev)]
    multiview.reduce_gradients(g)
    assert torch.equal(g[0], torch.full((5, 3), 2.0, device=dev)) and torch.equal(g[2].cpu(), torch.arange(4, dtype=torch.float32))
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK")


if __name__ == "__main__":
    main()
