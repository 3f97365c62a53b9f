// yardstick: this library's pair sort -- three launches per pass, and the opt-in look-back passes -- vs rocprim::radix_sort_pairs on
// tile-sort shaped input (not part of the product; hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w scripts/probe/sort_probe.hip)
// usage: sort_probe [pairs = 7300000] [key bits = 13] [tickets = 1]
#include <cstring>
#include "../../gaussian-pcloud-render_amd/csrc/sort.hip"
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
static int g_tickets = 1;
namespace gsr {
int check_launch(const Launch&, const char*) { return hipGetLastError() == hipSuccess ? 0 : -2; }
int block_tickets(int) { return g_tickets; }
}
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s:%d\n", __FILE__, __LINE__); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 7300000;
    const int bits = argc > 2 ? atoi(argv[2]) : 13;
    g_tickets = argc > 3 ? atoi(argv[3]) : 1;
    const int T = 1 << bits;
    std::vector<uint32_t> hk(n), hv(n);
    std::mt19937 rng(1);
    for (int64_t i = 0; i < n; i++) { hk[i] = rng() % (uint32_t)(T - 32); hv[i] = (uint32_t)i; }
    uint32_t *k[2], *v[2], *hist, *tot, *lbw, *ghist, *tcount;
    uint64_t* cnt;
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&k[i], n * 4)); CK(hipMalloc(&v[i], n * 4)); }
    const int nblk = gsr::sort_hist_stride(n);
    const size_t pw = gsr::lb_pass_words(n);
    CK(hipMalloc(&hist, (size_t)nblk * 256 * 4)); CK(hipMalloc(&tot, 1024));
    CK(hipMalloc(&lbw, 2 * pw * 4)); CK(hipMalloc(&ghist, 2 * 256 * 4)); CK(hipMalloc(&tcount, (size_t)T * 4)); CK(hipMalloc(&cnt, 64));
    const uint64_t hc[8] = {(uint64_t)n, 0, 0, 0, 0, 0, 0, 0};
    CK(hipMemcpy(cnt, hc, sizeof(hc), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    gsr::Launch L{s, 0};
    float ms;
    std::vector<uint32_t> ref(n), got(n);
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemcpy(k[0], hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v[0], hv.data(), n * 4, hipMemcpyHostToDevice));
        int res = 0;
        const gsr::SortJob job{{k[0], k[1]}, {v[0], v[1]}, hist, tot, 0, cnt, 0, n, 1};
        CK(hipEventRecord(e0, s));
        if (gsr::launch_radix_sort_pairs(L, job, false, bits, &res, false)) return 1;
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("three launches per pass  n=%lld bits=%d: %.3f ms\n", (long long)n, bits, ms);
        CK(hipMemcpy(ref.data(), v[res], n * 4, hipMemcpyDeviceToHost));
    }
    if (gsr::lb_fits(n) && bits <= 15) {
        for (int rep = 0; rep < 4; rep++) {
            CK(hipMemcpy(k[0], hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v[0], hv.data(), n * 4, hipMemcpyHostToDevice));
            CK(hipMemset(ghist, 0, 2 * 256 * 4)); CK(hipMemset(tcount, 0, (size_t)T * 4));
            int res = 0;
            const gsr::SortJob job{{k[0], k[1]}, {v[0], v[1]}, nullptr, nullptr, 0, cnt, 0, n, 1};
            gsr::LbJob lj{lbw, pw, ghist, 0, cnt, 0, nullptr};
            lj.tile_count = tcount;
            CK(hipEventRecord(e0, s));
            if (gsr::launch_tile_sort_lookback(L, job, lj, T, bits, &res, false)) return 1;
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(got.data(), v[res], n * 4, hipMemcpyDeviceToHost));
            printf("look-back passes (%s) n=%lld bits=%d: %.3f ms, order %s\n", g_tickets ? "tickets" : "blockIdx", (long long)n, bits, ms,
                   got == ref ? "identical" : "DIFFERENT");
        }
    }
    size_t tb = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tb, k[0], k[1], v[0], v[1], (size_t)n, 0, bits, s));
    void* tmp; CK(hipMalloc(&tmp, tb));
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemcpy(k[0], hk.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v[0], hv.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, s));
        CK(rocprim::radix_sort_pairs(tmp, tb, k[0], k[1], v[0], v[1], (size_t)n, 0, bits, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(got.data(), v[1], n * 4, hipMemcpyDeviceToHost));
        printf("rocprim::radix_sort_pairs n=%lld bits=%d: %.3f ms (tmp %zu B), order %s\n", (long long)n, bits, ms, tb, got == ref ? "identical" : "DIFFERENT");
    }
    return 0;
}
