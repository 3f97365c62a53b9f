// cumask_where: which (XCC, SE, CU) do the workgroups of a stream created with hipExtStreamCreateWithCUMask land on?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include <map>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_where(uint32_t* out, int spin)
{
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = __builtin_fmaf(a, 0.999f, 1.0f);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc & 15; }
    if (a == 123.f) out[0] = 0;
}

int main()
{
    const int ncu = 256, nb = 8192;
    uint32_t* d; OK(hipMalloc(&d, nb * 8));
    std::vector<uint32_t> h(nb * 2);
    struct Pat { const char* name; int kind, arg; };
    Pat pats[] = {{"first 32 bits", 0, 32}, {"first 64 bits", 0, 64}, {"bits 0-7 of every 32", 1, 8}, {"every 4th bit", 2, 4}, {"every 8th bit", 2, 8}, {"bits 64..255", 3, 64}, {"all but every 4th", 4, 4}};
    for (auto& pt : pats) {
        std::vector<uint32_t> m(ncu / 32, 0);
        int nbits = 0;
        for (int i = 0; i < ncu; ++i) {
            bool on = pt.kind == 0 ? i < pt.arg : pt.kind == 1 ? (i % 32) < pt.arg : pt.kind == 2 ? (i % pt.arg == 0) : pt.kind == 3 ? i >= pt.arg : (i % pt.arg != 0);
            if (on) { m[i / 32] |= 1u << (i % 32); ++nbits; }
        }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data());
        if (e != hipSuccess) { printf("%-22s: create failed %s\n", pt.name, hipGetErrorString(e)); continue; }
        k_where<<<nb, 256, 0, s>>>(d, 20000);
        OK(hipStreamSynchronize(s));
        OK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
        std::map<int, std::set<int>> per_xcc;
        for (int b = 0; b < nb; ++b) {
            const uint32_t hw = h[2 * b], x = h[2 * b + 1];
            const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[x].insert(se * 32 + sh * 16 + cu);
        }
        printf("%-22s (%3d bits): ", pt.name, nbits);
        int tot = 0;
        for (auto& kv : per_xcc) { printf("xcc%d:%zu ", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
        printf(" total %d CUs\n", tot);
        if (pt.kind == 0 && pt.arg == 32) { printf("   xcc0 CUs (se*32+sh*16+cu):"); for (int c : per_xcc.begin()->second) printf(" %d", c); printf("\n"); }
        OK(hipStreamDestroy(s));
    }
    return 0;
}
