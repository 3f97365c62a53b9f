"""Process plumbing of bench.py: the self-launch of an N-rank run under torch.distributed.run (one process per GPU over RCCL; the
round driver normally does that itself), and the child processes behind two side figures of the line -- the drop-in rate from
fresh processes and the world-size-1 anchor of the RCCL frame gather.  A side figure never takes the headline down: every
failure comes back as an `error` / `errors` entry."""
import json
import os
import socket
import subprocess
import sys

import numpy as np


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args, script, n_dev):
    """Re-run `script` with the same command line as args.gpus ranks on 127.0.0.1; returns the exit code."""
    if args.dist_backend == "nccl" and args.device_index < 0 and n_dev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; RCCL needs one device per rank "
                         "(use --dist-backend gloo --device-index 0 to exercise the control flow on one GPU)" % (args.gpus, n_dev))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), script] + sys.argv[1:]
    return subprocess.call(cmd)


def _workload_flags(args, W, H):
    return ["--config", str(args.config), "--workload", args.workload, "--width", str(W), "--height", str(H), "--profile", args.profile,
            "--no-cpu-baseline"] + (["--points", str(args.points)] if args.points else []) + (["--forward-only"] if args.forward_only else [])


def drop_in_fresh_processes(args, script, W, H):
    """The drop-in figure from FRESH processes (each: import, build the cloud, one second of warm-up, three blocks of 48 frames): it
    must not depend on what the main process did before, nor on how a process's streams happened to be set up."""
    cmd = [sys.executable, script, "--drop-in-probe"] + _workload_flags(args, W, H)
    vals, errs = [], []
    for _ in range(args.drop_in_processes):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            ls = [l for l in r.stdout.splitlines() if l.startswith("{")]
            vals.append(json.loads(ls[-1])["drop_in_frames_per_s"]) if ls else errs.append((r.stdout + r.stderr)[-200:])
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))
    return {"n": len(vals), "frames_per_s": vals, "min": min(vals) if vals else None,
            "median": float(np.median(vals)) if vals else None, "max": max(vals) if vals else None,
            "spread": round((max(vals) - min(vals)) / float(np.median(vals)), 4) if vals else None, "errors": errs or None}


def gather_anchor(args, script, W, H, dev_index):
    """1-rank anchor for the first multi-GPU run: a child process under RANK=0 WORLD_SIZE=1 (an RCCL process group of one) times the
    same blocks with and without the gather of every submission's full-size frames on its side stream (bench.py --gather-probe)."""
    try:
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, script, "--gather-probe", "--gpus", "1", "--steps", "48", "--device-index", str(dev_index),
               "--gather-mode", args.gather_mode] + _workload_flags(args, W, H)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        ls = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(ls[-1]) if ls else {"error": (r.stdout + r.stderr)[-300:]}
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)}
