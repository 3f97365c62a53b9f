// per-instruction VALU cost on gfx950 for the instruction kinds the render kernels use (4 waves per SIMD; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(NAME, BODY)                                                              \
    __global__ __launch_bounds__(64) void NAME(float* out, int iters)                   \
    {                                                                                   \
        float a = threadIdx.x, b = 2, c = 3, d = 4;                                     \
        int ia = threadIdx.x, ib = 5;                                                   \
        for (int i = 0; i < iters; i++) { asm volatile(REP16(BODY) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib) : : "vcc", "s20", "s21", "s22", "s23"); } \
        out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + ia + ib;                   \
    }
KERNEL(k_add, "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0\n")
KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[22:23]\n v_cndmask_b32_e64 %3, %3, %0, s[22:23]\n")
KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n")
KERNEL(k_cmp_e64, "v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cmp_lt_f32_e64 s[20:21], %2, %3\n v_cmp_lt_f32_e64 s[22:23], %3, %0\n")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %3\n v_cndmask_b32 %0, %0, %2, vcc\n")
KERNEL(k_and, "v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %4\n v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %4\n")
KERNEL(k_ashr, "v_ashrrev_i32 %4, 31, %4\n v_ashrrev_i32 %5, 31, %5\n v_ashrrev_i32 %4, 31, %4\n v_ashrrev_i32 %5, 31, %5\n")
KERNEL(k_min, "v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %0\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4\n")
KERNEL(k_rndne, "v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n")
KERNEL(k_cvt, "v_cvt_i32_f32 %4, %0\n v_cvt_i32_f32 %5, %1\n v_cvt_i32_f32 %4, %2\n v_cvt_i32_f32 %5, %3\n")
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n")
KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n")
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_sand, "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[22:23], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[22:23], s[20:21], s[22:23]\n")
template <typename K> static void run(const char* name, K k, float* out)
{
    const int iters = 1000, blocks = 4096;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s %.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / (iters * 64.0 * 4));
}
int main()
{
    float* out; hipMalloc(&out, 1 << 23);
    run("add", k_add, out); run("fma", k_fma, out); run("min", k_min, out); run("and", k_and, out); run("ashr", k_ashr, out);
    run("cnd_vcc", k_cnd_vcc, out); run("cnd_e64", k_cnd_e64, out); run("cmp_vcc", k_cmp_vcc, out); run("cmp_e64", k_cmp_e64, out);
    run("cmp+cnd", k_cmp_cnd, out); run("rcp", k_rcp, out); run("ldexp", k_ldexp, out); run("rndne", k_rndne, out); run("cvt", k_cvt, out);
    run("mov_dpp", k_mov_dpp, out); run("swap32", k_swap32, out); run("swap16", k_swap16, out); run("add_dpp", k_add_dpp, out); run("s_and", k_sand, out);
    return 0;
}
