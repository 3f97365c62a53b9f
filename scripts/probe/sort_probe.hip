// yardstick: this library's pair sort vs rocprim::radix_sort_pairs on tile-sort shaped input (not part of the product)
#include <cstring>
#include "../../gaussian-pcloud-render_amd/csrc/sort.hip"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
namespace gsr { int check_launch(const Launch&, const char*) { return 0; } }
int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 11800000;
    const int bits = argc > 2 ? atoi(argv[2]) : 13;
    std::vector<uint32_t> hk(n), hv(n);
    std::mt19937 rng(1);
    for (int64_t i = 0; i < n; i++) { hk[i] = rng() % 8160; hv[i] = (uint32_t)i; }
    uint32_t *k[2], *v[2], *hist, *tot;
    for (int i = 0; i < 2; i++) { hipMalloc(&k[i], n * 4); hipMalloc(&v[i], n * 4); }
    const int nblk = (int)((n + gsr::RS_TILE - 1) / gsr::RS_TILE);
    hipMalloc(&hist, (size_t)nblk * 256 * 4); hipMalloc(&tot, 1024);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    gsr::Launch L{s, 0};
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(k[0], hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v[0], hv.data(), n * 4, hipMemcpyHostToDevice);
        int res = 0;
        hipEventRecord(e0, s);
        gsr::launch_radix_sort_pairs(L, n, k, v, false, bits, hist, tot, &res);
        hipEventRecord(e1, s); hipStreamSynchronize(s); hipEventElapsedTime(&ms, e0, e1);
        printf("gsr   sort n=%lld bits=%d: %.3f ms (%.1f GB/s algorithmic @20B/pair/pass)\n", (long long)n, bits, ms,
               20.0 * n * ((bits + 7) / 8) / ms * 1e-6);
    }
    size_t tb = 0;
    rocprim::radix_sort_pairs(nullptr, tb, k[0], k[1], v[0], v[1], (size_t)n, 0, bits, s);
    void* tmp; hipMalloc(&tmp, tb);
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(k[0], hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(v[0], hv.data(), n * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0, s);
        rocprim::radix_sort_pairs(tmp, tb, k[0], k[1], v[0], v[1], (size_t)n, 0, bits, s);
        hipEventRecord(e1, s); hipStreamSynchronize(s); hipEventElapsedTime(&ms, e0, e1);
        printf("rocprim sort n=%lld bits=%d: %.3f ms (tmp %zu B)\n", (long long)n, bits, ms, tb);
    }
    // depth-sort shaped: 32-bit keys, n = 800K
    return 0;
}
