// cost of the backward pass's per-group gradient atomics: four separate arrays vs one 64-B record per Gaussian
// (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_sep(float* m2, float* con, float* op, float* col, const uint32_t* ids, int iters, int P)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t i = lane >> 1, c = i & 7u, k = i >> 3;
    float* base = nullptr; uint32_t stride = 0; bool on = false; uint32_t kk = k;
    if ((lane & 1u) == 0) { on = true; if (c < 2) { base = m2 + c; stride = 3; } else if (c < 5) { base = con + (c == 4 ? 3 : c - 2); stride = 4; } else { base = col + (c - 5); stride = 3; } }
    else if ((lane & 15u) == 1) { on = true; base = op; stride = 1; kk = lane >> 4; }
    for (int it = 0; it < iters; it++) {
        const uint32_t id = ids[((blockIdx.x * iters + it) * 4 + kk) % (P * 4)] ;
        if (on) atomicAdd(base + (size_t)id * stride, 1.0f);
    }
}
__global__ __launch_bounds__(64) void k_rec(float* rec, const uint32_t* ids, int iters, int P)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t i = lane >> 1, c = i & 7u, k = i >> 3;
    uint32_t off = 0; bool on = false; uint32_t kk = k;
    if ((lane & 1u) == 0) { on = true; off = c; }
    else if ((lane & 15u) == 1) { on = true; off = 8; kk = lane >> 4; }
    for (int it = 0; it < iters; it++) {
        const uint32_t id = ids[((blockIdx.x * iters + it) * 4 + kk) % (P * 4)];
        if (on) atomicAdd(rec + (size_t)id * 16 + off, 1.0f);
    }
}
int main()
{
    const int P = 800000, blocks = 8192, iters = 64;   // 8192*64 groups = 524K groups = 2.1 M entries
    float *m2, *con, *op, *col, *rec; uint32_t* ids;
    hipMalloc(&m2, P * 12); hipMalloc(&con, P * 16); hipMalloc(&op, P * 4); hipMalloc(&col, P * 12); hipMalloc(&rec, (size_t)P * 64);
    hipMalloc(&ids, P * 16);
    uint32_t* h = new uint32_t[P * 4];
    uint32_t x = 12345;
    for (int i = 0; i < P * 4; i++) { x = x * 1664525u + 1013904223u; h[i] = (x >> 8) % P; }
    hipMemcpy(ids, h, P * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_sep, dim3(blocks), dim3(64), 0, 0, m2, con, op, col, ids, iters, P); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("4 arrays : %.3f ms for %d groups of 36 atomics\n", ms, blocks * iters);
        hipEventRecord(e0); hipLaunchKernelGGL(k_rec, dim3(blocks), dim3(64), 0, 0, rec, ids, iters, P); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("64-B rec : %.3f ms\n", ms);
    }
    return 0;
}
