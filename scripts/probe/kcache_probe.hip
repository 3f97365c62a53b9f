// is the scalar cache invalidated between back-to-back kernels of one stream?  (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_write(float* tab, int n, float v) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) tab[i] = v + i; }
__global__ void k_sread(const float* tab, int n, float v, int* bad)
{
    int nb = 0;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        float x;
        asm volatile("s_load_dword %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=&s"(x) : "s"(tab + i) : "memory");
        if (x != v + i) nb++;
    }
    if (threadIdx.x == 0 && nb) atomicAdd(bad, nb);
}
int main()
{
    const int n = 1 << 16;
    float* d; int* bad; hipMalloc(&d, n * 4); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 20; rep++) {
        hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, s, d, n, (float)rep * 1000.f);
        hipLaunchKernelGGL(k_sread, dim3(1024), dim3(64), 0, s, d, n, (float)rep * 1000.f, bad);
    }
    hipStreamSynchronize(s);
    int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("stale scalar reads across 20 write/read kernel pairs: %d\n", hb);
    return 0;
}
