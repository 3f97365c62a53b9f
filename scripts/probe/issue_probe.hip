// lone-wave VALU issue cadence on gfx950: dependent chain vs independent instructions vs packed vs transcendental
// (one wave per SIMD; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(NAME, BODY)                                                              \
    __global__ __launch_bounds__(64) void NAME(float* out, int iters)                   \
    {                                                                                   \
        float a = threadIdx.x, b = 2, c = 3, d = 4, e = 5, f = 6, g = 7, h = 8;         \
        typedef float f2 __attribute__((ext_vector_type(2)));                           \
        f2 p = {a, b}, q = {c, d}, r = {e, f}, s = {g, h};                              \
        long long t0 = clock64();                                                       \
        for (int i = 0; i < iters; i++) { asm volatile(REP16(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q), "+v"(r), "+v"(s)); } \
        long long t1 = clock64();                                                       \
        out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + p.x + q.x + r.x + s.y;     \
        if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;  \
    }
KERNEL(k_dep, "v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n")
KERNEL(k_ind4, "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n")
KERNEL(k_pk_dep, "v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %4, %4, %4\n")
KERNEL(k_pk_ind4, "v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %5, %5, %5\n v_pk_mul_f32 %6, %6, %6\n v_pk_mul_f32 %7, %7, %7\n")
KERNEL(k_exp_dep, "v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n")
KERNEL(k_exp_ind4, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
KERNEL(k_cnd_dep, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n")
template <typename K> static void run(const char* name, K k, int blocks, float* out)
{
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, ((long long*)out) + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("%-10s blocks=%5d: %.3f ms, %.2f clock64 ticks per instruction (wave 0), %.2f ns per instruction per wave\n", name, blocks, ms,
           (double)cyc / (iters * 64.0), ms * 1e6 / (iters * 64.0));
}
int main()
{
    float* out; hipMalloc(&out, (1 << 23) + 64);
    for (int blocks : {1024, 4096}) {   // 1 and 4 waves per SIMD
        run("dep", k_dep, blocks, out); run("ind4", k_ind4, blocks, out);
        run("pk_dep", k_pk_dep, blocks, out); run("pk_ind4", k_pk_ind4, blocks, out);
        run("exp_dep", k_exp_dep, blocks, out); run("exp_ind4", k_exp_ind4, blocks, out);
        run("cnd_dep", k_cnd_dep, blocks, out);
    }
    return 0;
}
