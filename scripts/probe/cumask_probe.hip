// cumask_probe: can a VALU-bound kernel and an HBM-bound kernel be co-scheduled on disjoint CU sets with
// hipExtStreamCreateWithCUMask, and does the pair then finish in less than the sum of the two?
//   hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_valu(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-5f, c = 1.0f, d = 0.5f;
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fmaf(a, 0.999f, b); b = __builtin_fmaf(b, 0.998f, c); c = __builtin_fmaf(c, 0.997f, d); d = __builtin_fmaf(d, 0.996f, a);
    }
    if (a + b + c + d == 12345.f) out[0] = a;
}

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

static float run(hipStream_t sv, hipStream_t sc, bool do_v, bool do_c, float* out, int iters, int vgrid, const float4* in, float4* o4, size_t n, int cgrid,
                 float* tv, float* tc)
{
    hipEvent_t a, b, v0, v1, c0, c1;
    OK(hipEventCreate(&a)); OK(hipEventCreate(&b)); OK(hipEventCreate(&v0)); OK(hipEventCreate(&v1)); OK(hipEventCreate(&c0)); OK(hipEventCreate(&c1));
    OK(hipDeviceSynchronize());
    OK(hipEventRecord(a, 0));
    OK(hipStreamWaitEvent(sv, a, 0)); OK(hipStreamWaitEvent(sc, a, 0));
    const int reps = 10;
    OK(hipEventRecord(v0, sv)); OK(hipEventRecord(c0, sc));
    for (int r = 0; r < reps; ++r) {
        if (do_v) k_valu<<<vgrid, 256, 0, sv>>>(out, iters);
        if (do_c) k_copy<<<cgrid, 256, 0, sc>>>(in, o4, n);
    }
    OK(hipEventRecord(v1, sv)); OK(hipEventRecord(c1, sc));
    OK(hipStreamWaitEvent(0, v1, 0)); OK(hipStreamWaitEvent(0, c1, 0));
    OK(hipEventRecord(b, 0));
    OK(hipDeviceSynchronize());
    float ms, m1, m2;
    OK(hipEventElapsedTime(&ms, a, b)); OK(hipEventElapsedTime(&m1, v0, v1)); OK(hipEventElapsedTime(&m2, c0, c1));
    *tv = m1 / reps; *tc = m2 / reps;
    return ms / reps;
}

int main()
{
    hipDeviceProp_t p; OK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s: %d CUs\n", p.name, ncu);
    const size_t n = (size_t)1 << 26;            // 64 M float4 = 1 GiB in, 1 GiB out
    float4 *in, *o4; float* out;
    OK(hipMalloc(&in, n * 16)); OK(hipMalloc(&o4, n * 16)); OK(hipMalloc(&out, 64));
    OK(hipMemset(in, 1, n * 16));
    hipStream_t s1, s2;
    OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int iters = 6000, vgrid = 256 * 16;
    float tv, tc, t;
    run(s1, s2, true, true, out, iters, vgrid, in, o4, n, 4096, &tv, &tc);   // warm
    t = run(s1, s2, true, false, out, iters, vgrid, in, o4, n, 4096, &tv, &tc); printf("valu alone, all CUs:            %.3f ms\n", t);
    t = run(s1, s2, false, true, out, iters, vgrid, in, o4, n, 4096, &tv, &tc); printf("copy alone, all CUs:            %.3f ms  (%.2f TB/s)\n", t, 2.0 * n * 16 / t * 1e-9);
    t = run(s1, s2, true, true, out, iters, vgrid, in, o4, n, 4096, &tv, &tc);  printf("both, two plain streams:        %.3f ms  (valu stream %.3f, copy stream %.3f)\n", t, tv, tc);
    for (int pattern = 0; pattern < 3; ++pattern)
    for (int ccu : {32, 64, 128}) {
        std::vector<uint32_t> mv((ncu + 31) / 32, 0), mc((ncu + 31) / 32, 0);
        // pattern 0: the first ccu bits; pattern 1: the first ccu/8 bits of every group of 32; pattern 2: every (256/ccu)-th bit
        for (int i = 0; i < ncu; ++i) {
            const bool c = pattern == 0 ? (i < ccu) : pattern == 1 ? ((i % 32) < ccu / 8) : (i % (ncu / ccu) == 0);
            (c ? mc : mv)[i / 32] |= 1u << (i % 32);
        }
        hipStream_t sv, sc;
        if (hipExtStreamCreateWithCUMask(&sv, (uint32_t)mv.size(), mv.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&sc, (uint32_t)mc.size(), mc.data()) != hipSuccess) {
            printf("hipExtStreamCreateWithCUMask failed\n"); return 1;
        }
        float a1 = run(sv, sc, true, false, out, iters, vgrid, in, o4, n, ccu * 16, &tv, &tc);
        float a2 = run(sv, sc, false, true, out, iters, vgrid, in, o4, n, ccu * 16, &tv, &tc);
        t = run(sv, sc, true, true, out, iters, vgrid, in, o4, n, ccu * 16, &tv, &tc);
        printf("pattern %d copy on %3d CUs / valu on %3d: valu alone %.3f, copy alone %.3f (%.2f TB/s), both %.3f ms (valu %.3f, copy %.3f)\n", pattern, ccu, ncu - ccu, a1, a2,
               2.0 * n * 16 / a2 * 1e-9, t, tv, tc);
        OK(hipStreamDestroy(sv)); OK(hipStreamDestroy(sc));
    }
    return 0;
}
