// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of this library,
// against known byte counts (MI355X_MICROARCH.md calibrates only the wide coalesced read: FETCH_SIZE = 1/2 of the bytes).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/fetch_calib scripts/probe/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o c -- scripts/probe/fetch_calib      (and again with WRITE_SIZE)
//   python scripts/probe/fetch_calib_report.py out
// Every kernel touches a fresh region of a 4 GiB buffer (far beyond the 256 MiB Infinity Cache), prints what it asked for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct __attribute__((aligned(64))) Rec { float4 q0, q1, q2, q3; };

__global__ void k_stream16(const float4* __restrict__ p, size_t n, float* out)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void k_stream4(const float* __restrict__ p, size_t n, float* out)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678f) out[0] = s;
}
// one 16-B read per lane out of a random 64-B record (the pair emission's gather)
__global__ void k_gather16(const Rec* __restrict__ r, const uint32_t* __restrict__ idx, size_t n, float* out)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = r[idx[i]].q3; s += v.x + v.w; }
    if (s == 12345.678f) out[0] = s;
}
// 36 B of a random 64-B record per lane: q0, q1 and one word of q2 (the render kernels' gather)
__global__ void k_gather36(const Rec* __restrict__ r, const uint32_t* __restrict__ idx, size_t n, float* out)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Rec* p = r + idx[i];
        const float4 a = p->q0, b = p->q1;
        s += a.x + a.w + b.x + b.w + p->q2.x;
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void k_write16(float4* __restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// a whole 64-B record per lane, consecutive lanes consecutive records (preprocess' Splat store)
__global__ void k_write_rec(Rec* __restrict__ r, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = make_float4(1.f, 2.f, 3.f, (float)i);
        r[i].q0 = v; r[i].q1 = v; r[i].q2 = v; r[i].q3 = v;
    }
}
// scattered 4-B atomics into random 64-B records, nine words of a record per lane (the render backward's gradient records)
__global__ void k_atomic9(float* __restrict__ rec, const uint32_t* __restrict__ idx, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float* p = rec + (size_t)idx[i / 9] * 16 + (i % 9);
        atomicAdd(p, 1.0f);
    }
}

int main()
{
    const size_t BYTES = 4ull << 30;
    char* buf = nullptr;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMemset(buf, 0, BYTES));
    float* out = nullptr;
    CK(hipMalloc(&out, 256));
    const size_t NREC = 16u << 20;                      // 16 Mi records of 64 B = 1 GiB region for the gathers
    std::vector<uint32_t> h(NREC);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < NREC; i++) h[i] = (uint32_t)i;
    for (size_t i = NREC - 1; i > 0; i--) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const size_t j = x % (i + 1); const uint32_t t = h[i]; h[i] = h[j]; h[j] = t; }
    uint32_t* idx = nullptr;
    CK(hipMalloc(&idx, NREC * 4));
    CK(hipMemcpy(idx, h.data(), NREC * 4, hipMemcpyHostToDevice));
    const dim3 grid(256 * 16), block(256);
    const size_t GB = 1ull << 30;
    // name, bytes the lanes ask for, bytes of the 64-B lines they touch
    printf("EXPECT k_stream16 %zu %zu\n", GB, GB);
    hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, (const float4*)buf, GB / 16, out);
    printf("EXPECT k_stream4 %zu %zu\n", GB / 4, GB / 4);
    hipLaunchKernelGGL(k_stream4, grid, block, 0, 0, (const float*)(buf + GB), GB / 16, out);
    printf("EXPECT k_gather16 %zu %zu\n", NREC * 16 + NREC * 4, NREC * 64 + NREC * 4);
    hipLaunchKernelGGL(k_gather16, grid, block, 0, 0, (const Rec*)(buf + 2 * GB), idx, NREC, out);
    printf("EXPECT k_gather36 %zu %zu\n", NREC * 36 + NREC * 4, NREC * 64 + NREC * 4);
    hipLaunchKernelGGL(k_gather36, grid, block, 0, 0, (const Rec*)(buf + 3 * GB), idx, NREC, out);
    CK(hipDeviceSynchronize());
    printf("EXPECT k_write16 %zu %zu\n", GB, GB);
    hipLaunchKernelGGL(k_write16, grid, block, 0, 0, (float4*)buf, GB / 16);
    printf("EXPECT k_write_rec %zu %zu\n", GB, GB);
    hipLaunchKernelGGL(k_write_rec, grid, block, 0, 0, (Rec*)(buf + GB), NREC);
    printf("EXPECT k_atomic9 %zu %zu\n", (NREC / 4) * 9 * 4, (NREC / 4) * 64);
    hipLaunchKernelGGL(k_atomic9, grid, block, 0, 0, (float*)(buf + 2 * GB), idx, (NREC / 4) * 9);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
