"""Build libgsr_hip.so (the C-ABI HIP library of include/gsr.h) for gfx950, in-tree.

    python gaussian-pcloud-render_amd/build.py [--force] [--save-temps]

hipcc cross-compiles without a GPU.  The .so lands next to the Python mirror package
(diff_gaussian_rasterization/libgsr_hip.so) so it travels with the tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "diff_gaussian_rasterization")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libgsr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

UNITS = ["api", "preprocess", "sort", "binning", "render_fwd", "render_bwd", "preprocess_bwd"]

# -ffp-contract=off : one rounding per written operation (integer outputs reproducible, DESIGN.md "Numerics")
# -munsafe-fp-atomics: float atomicAdd -> global_atomic_add_f32 / ds_add_f32 instead of a CAS loop
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", *os.environ.get("GSR_EXTRA_FLAGS", "").split()]


def _deps_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return max(m, os.path.getmtime(os.path.abspath(__file__)))


def build(force=False, save_temps=False, verbose=True):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)

    def cc(u):
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(OBJ_DIR, u + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if u == "api":
            cmd.insert(-4, "-fvisibility=default")
        if save_temps:
            cmd += ["-save-temps=obj"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ_DIR)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (u, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(UNITS))) as ex:
        objs = list(ex.map(cc, UNITS))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-Wl,-Bsymbolic", "-o", LIB] + objs)
    if verbose:
        print("[build] %s" % LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
