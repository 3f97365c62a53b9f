// microbenchmark: issue cost of v_readlane_b32 vs a plain VALU op vs s_load (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ __launch_bounds__(64) void k_readlane(float* out, int iters, int sel)
{
    float v = threadIdx.x * 1.0f, acc = 0;
    for (int i = 0; i < iters; i++) {
        asm volatile(REP16("v_readlane_b32 s20, %0, %1\n v_readlane_b32 s21, %0, %1\n v_readlane_b32 s22, %0, %1\n v_readlane_b32 s23, %0, %1\n")
                     : : "v"(v), "s"(sel) : "s20", "s21", "s22", "s23");
    }
    out[blockIdx.x * 64 + threadIdx.x] = v + acc;
}
__global__ __launch_bounds__(64) void k_valu(float* out, int iters, int sel)
{
    float v = threadIdx.x * 1.0f;
    float a = 1, b = 2, c = 3, d = 4;
    for (int i = 0; i < iters; i++) {
        asm volatile(REP16("v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(v));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
// mixed: 9 readlanes + 20 v_add per "entry" vs 1 readlane + 20 v_add
__global__ __launch_bounds__(64) void k_mix9(float* out, int iters, int sel)
{
    float v = threadIdx.x * 1.0f;
    float a = 1, b = 2, c = 3, d = 4;
    for (int i = 0; i < iters; i++) {
        asm volatile(REP16("v_readlane_b32 s20, %4, %5\n v_readlane_b32 s21, %4, %5\n v_readlane_b32 s22, %4, %5\n v_readlane_b32 s23, %4, %5\n v_readlane_b32 s24, %4, %5\n"
                           "v_readlane_b32 s25, %4, %5\n v_readlane_b32 s26, %4, %5\n v_readlane_b32 s27, %4, %5\n v_readlane_b32 s28, %4, %5\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(v), "s"(sel) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(64) void k_mix1(float* out, int iters, int sel)
{
    float v = threadIdx.x * 1.0f;
    float a = 1, b = 2, c = 3, d = 4;
    for (int i = 0; i < iters; i++) {
        asm volatile(REP16("v_readlane_b32 s20, %4, %5\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n"
                           "v_add_f32 %0, %4, %0\n v_add_f32 %1, %4, %1\n v_add_f32 %2, %4, %2\n v_add_f32 %3, %4, %3\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(v), "s"(sel) : "s20");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
// scalar-load broadcast: 1 readlane (id) + s_load_dwordx8 + s_load_dword per entry + 20 v_add using the loaded SGPRs
__global__ __launch_bounds__(64) void k_sload(float* out, const float* __restrict__ table, int iters, int nrec)
{
    float a = 1, b = 2, c = 3, d = 4;
    uint32_t id = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) % (uint32_t)nrec;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t sid = __builtin_amdgcn_readlane(id, (j * 4 + i) & 63);
            const float* p = table + (size_t)sid * 12;
            float s0, s1, s2, s3, s4, s5, s6, s7, s8;
            asm volatile("s_load_dwordx8 s[20:27], %9, 0x0\n s_load_dword s28, %9, 0x20\n s_waitcnt lgkmcnt(0)\n"
                         "s_mov_b32 %0, s20\n s_mov_b32 %1, s21\n s_mov_b32 %2, s22\n s_mov_b32 %3, s23\n s_mov_b32 %4, s24\n"
                         "s_mov_b32 %5, s25\n s_mov_b32 %6, s26\n s_mov_b32 %7, s27\n s_mov_b32 %8, s28\n"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7), "=s"(s8)
                         : "s"(p) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "memory");
            a += s0; b += s1; c += s2; d += s3; a += s4; b += s5; c += s6; d += s7; a += s8;
            a += s0; b += s1; c += s2; d += s3; a += s4; b += s5; c += s6; d += s7; a += s8; b += s0; c += s1;
        }
        id = id * 1664525u % (uint32_t)nrec;
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
template <typename F> static float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    const int blocks = 256 * 32, iters = 2000;   // 8 waves per SIMD resident
    float *out, *table; hipMalloc(&out, blocks * 64 * 4);
    const int nrec = 800000; hipMalloc(&table, (size_t)nrec * 48); hipMemset(table, 0, (size_t)nrec * 48);
    const double instr = (double)blocks * iters * 64;   // wave-instrs of the 64-instr body
    float t;
    t = timeit([&] { hipLaunchKernelGGL(k_readlane, dim3(blocks), dim3(64), 0, 0, out, iters, 5); });
    printf("readlane x64/iter : %.3f ms  -> %.1f G wave-instr/s\n", t, instr / t * 1e-6);
    t = timeit([&] { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(64), 0, 0, out, iters, 5); });
    printf("v_add    x64/iter : %.3f ms  -> %.1f G wave-instr/s\n", t, instr / t * 1e-6);
    t = timeit([&] { hipLaunchKernelGGL(k_mix9, dim3(blocks), dim3(64), 0, 0, out, iters / 4, 5); });
    printf("9 readlane + 20 add per entry, 16 entries/iter: %.3f ms -> %.2f ns/entry/wave-slot\n", t, t * 1e6 / ((double)iters / 4 * 16));
    t = timeit([&] { hipLaunchKernelGGL(k_mix1, dim3(blocks), dim3(64), 0, 0, out, iters / 4, 5); });
    printf("1 readlane + 20 add per entry, 16 entries/iter: %.3f ms -> %.2f ns/entry/wave-slot\n", t, t * 1e6 / ((double)iters / 4 * 16));
    t = timeit([&] { hipLaunchKernelGGL(k_sload, dim3(blocks), dim3(64), 0, 0, out, table, iters / 4, nrec); });
    printf("1 readlane + s_load x9 (no pipelining) + 20 add per entry: %.3f ms -> %.2f ns/entry/wave-slot\n", t, t * 1e6 / ((double)iters / 4 * 16));
    return 0;
}
