#!/usr/bin/env bash
# build_ref.sh -- build the REFERENCE rasterizer core for gfx950 into oracle/_ref/.
#
# TEST INFRASTRUCTURE ONLY.  The reference's three CUDA translation units
#   /root/reference/diff-gaussian-rasterization/cuda_rasterizer/{rasterizer_impl,forward,backward}.cu
# (+ their headers and the vendored GLM) are compiled from where they lie with the ROCm tools
# that ship in this image:
#   1. /opt/rocm/bin/hipify-perl (AMD's CUDA->HIP source translator) renames the CUDA runtime /
#      CUB / cooperative-groups spellings; its output goes to oracle/_ref/src/ (git-ignored).
#   2. five mechanical line edits hipify-perl leaves undone (none touches a kernel's arithmetic):
#        - drop the now-empty `#include ""` it emits for device_launch_parameters.h
#        - drop `#include <cooperative_groups/reduce.h>`       (included, never used)
#        - drop `#include <cub/device/device_radix_sort.cuh>`  (hipcub.hpp already provides it)
#        - `<< <` / `>> >`  ->  `<<<` / `>>>`                  (nvcc tolerates the spaces, clang does not)
#        - `-D__trap=__builtin_trap` on the command line       (CUDA's __trap(); only reached when
#                                                               prefiltered=True flags a culled point)
#   3. hipcc --offload-arch=gfx950 links them with our host driver oracle/ref_driver.cpp.
# Nothing is copied into the tracked tree; no stand-in headers or libraries are written.
#
# Two variants are built:
#   libgsr_ref_strict.so  -ffp-contract=off : the reference SOURCE semantics (one rounding per
#                                             written operation) - what the CPU oracle and the
#                                             product kernels are bit-compared against.
#   libgsr_ref_fast.so    hipcc default (-ffp-contract=fast): stands in for nvcc's default
#                                             -fmad=true; used to report how many integer /
#                                             threshold decisions move under FMA contraction.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${REF_ROOT:-/root/reference}/diff-gaussian-rasterization"
OUT="$HERE/_ref"
SRC="$OUT/src"
if [ ! -d "$REF/cuda_rasterizer" ]; then
    echo "[build_ref] $REF not present (GPU box?) - using prebuilt oracle/_ref if any"
    exit 0
fi
if [ -f "$OUT/libgsr_ref_strict.so" ] && [ -f "$OUT/libgsr_ref_fast.so" ] \
   && [ "$OUT/libgsr_ref_strict.so" -nt "$HERE/ref_driver.cpp" ] \
   && [ "$OUT/libgsr_ref_strict.so" -nt "$HERE/build_ref.sh" ] && [ -z "${FORCE:-}" ]; then
    echo "[build_ref] oracle/_ref up to date"
    exit 0
fi
mkdir -p "$SRC"
for f in rasterizer_impl.cu forward.cu backward.cu rasterizer_impl.h forward.h backward.h auxiliary.h config.h rasterizer.h; do
    /opt/rocm/bin/hipify-perl "$REF/cuda_rasterizer/$f" 2>/dev/null \
      | sed -e '/#include ""/d' \
            -e '/cooperative_groups\/reduce.h/d' \
            -e '/cub\/device\/device_radix_sort.cuh/d' \
            -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' > "$SRC/$f"
done
COMMON=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -D__trap=__builtin_trap -w
        -I"$SRC" -I"$HERE" -I"$REF/third_party/glm")
build_variant() {  # name, extra flags...
    local name="$1"; shift
    local objs=()
    for u in rasterizer_impl forward backward; do
        /opt/rocm/bin/hipcc "${COMMON[@]}" "$@" -c "$SRC/$u.cu" -o "$OUT/${u}_$name.o" &
        objs+=("$OUT/${u}_$name.o")
    done
    /opt/rocm/bin/hipcc "${COMMON[@]}" "$@" -fvisibility=default -c "$HERE/ref_driver.cpp" -o "$OUT/ref_driver_$name.o" &
    objs+=("$OUT/ref_driver_$name.o")
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -Wl,-Bsymbolic -o "$OUT/libgsr_ref_$name.so" "${objs[@]}"
    rm -f "${objs[@]}"
}
build_variant strict -ffp-contract=off
build_variant fast
ls -la "$OUT"/*.so
