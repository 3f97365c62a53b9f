// Does v_mfma_f32_16x16x4_f32 run beside plain fp32 VALU work on a SIMD (from the same wave / from other waves), or do the
// two add up?  Three kernels with the same loop count: VALU only (NV independent v_fma_f32 per trip), MFMA only (NM dependent
// MFMAs per trip), both.  1024 workgroups of 256 threads = 4 waves per SIMD on every SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_overlap_probe mfma_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int NM, bool BF16>
__global__ __launch_bounds__(256) void k(float* out, int trips, float seed)
{
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + threadIdx.x + i;
    f32x4 acc = {0, 0, 0, 0};
    float a = seed + threadIdx.x, b = seed * 2.f;
    typedef short bf8 __attribute__((ext_vector_type(8)));
    bf8 ab = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x};
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int m = 0; m < (NM > NV / 8 ? NM : NV / 8); m++) {
            if (m < NM) {
                if (BF16) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
            if (m < NV / 8) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            }
        }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int NM, bool BF16>
float run(float* d, int trips, int wg, int threads)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NM, BF16>), dim3(wg), dim3(threads), 0, 0, d, 10, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NM, BF16>), dim3(wg), dim3(threads), 0, 0, d, trips, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    const int trips = 20000;
    for (int pass = 0; pass < 2; pass++) {
        const int wg = 1024, th = pass == 0 ? 256 : 64;   // 4 waves per SIMD | 1 wave per SIMD
        printf("%d waves per SIMD, %d trips\n", th / 64, trips);
        printf("  VALU only   (128 fma / trip)          %.3f ms\n", run<128, 0, false>(d, trips, wg, th));
        printf("  f32 MFMA only (8 dependent / trip)    %.3f ms\n", run<0, 8, false>(d, trips, wg, th));
        printf("  both (128 fma + 8 f32 MFMA)           %.3f ms\n", run<128, 8, false>(d, trips, wg, th));
        printf("  bf16 MFMA only (8 x 16x16x32 / trip)  %.3f ms\n", run<0, 8, true>(d, trips, wg, th));
        printf("  both (128 fma + 8 bf16 MFMA)          %.3f ms\n", run<128, 8, true>(d, trips, wg, th));
        printf("  VALU only   (64 fma / trip)           %.3f ms\n", run<64, 0, false>(d, trips, wg, th));
        printf("  both (64 fma + 8 f32 MFMA)            %.3f ms\n", run<64, 8, false>(d, trips, wg, th));
    }
    return 0;
}
