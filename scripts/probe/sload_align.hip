// does s_load_dwordx8 return the right data at 16-B (not 32-B) aligned addresses?  (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* tab, int* bad, int n, int mode)
{
    int nb = 0;
    for (int i = 0; i < n; i++) {
        const float* p = tab + (size_t)i * 12;
        f32x8 q; float b;
        if (mode == 0)
            asm volatile("s_load_dwordx8 %0, %2, 0x0\n s_load_dword %1, %2, 0x20\n s_waitcnt lgkmcnt(0)" : "=&s"(q), "=&s"(b) : "s"(p) : "memory");
        else {
            f32x4 a, c;
            asm volatile("s_load_dwordx4 %0, %3, 0x0\n s_load_dwordx4 %1, %3, 0x10\n s_load_dword %2, %3, 0x20\n s_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(c), "=&s"(b) : "s"(p) : "memory");
            q = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        }
        for (int k = 0; k < 8; k++) if (q[k] != (float)(i * 12 + k)) nb++;
        if (b != (float)(i * 12 + 8)) nb++;
    }
    if (threadIdx.x == 0) *bad = nb;
}
int main()
{
    const int n = 1000;
    float* h = new float[n * 12];
    for (int i = 0; i < n * 12; i++) h[i] = (float)i;
    float* d; int* bad; hipMalloc(&d, n * 48); hipMalloc(&bad, 4);
    hipMemcpy(d, h, n * 48, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, bad, n, mode);
        int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("mode %d (%s): mismatches %d of %d\n", mode, mode ? "x4+x4+x1" : "x8+x1", hb, n * 9);
    }
    return 0;
}
