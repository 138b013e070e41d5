// Phase timing of the radix pass kernel: where does a workgroup's life go?  Includes the library source with
// GSR_RADIX_TRACE, sorts random keys under each workgroup shape and prints per-phase statistics.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../autovfx_amd/csrc radix_trace.hip -o radix_trace
#define GSR_RADIX_TRACE 1
#include "../../autovfx_amd/csrc/gsr_radix.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 3000000u;
    const int bits = argc > 2 ? atoi(argv[2]) : 32;
    std::vector<uint32_t> h(n);
    uint32_t x = 12345u;
    for (auto& v : h) {
        x = x * 1664525u + 1013904223u;
        if (bits == 32) { float d = 0.2f + (x >> 8) * (60.0f / 16777216.0f); memcpy(&v, &d, 4); }  // depth-like keys
        else v = (x >> 7) % 8160u;
    }
    uint32_t *ka, *kb, *va, *vb, *scratch;
    unsigned long long* trace;
    CK(hipMalloc(&ka, n * 4)); CK(hipMalloc(&kb, n * 4)); CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&vb, n * 4));
    CK(hipMalloc(&scratch, gsr::radix_scratch_words(n) * 4));
    const size_t trace_words = (size_t)(n / 2048 + 2) * 8;
    CK(hipMalloc(&trace, trace_words * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best_total = 1e9f;
    uint32_t *ks = nullptr, *vs = nullptr;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemcpy(ka, h.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemset(trace, 0, trace_words * 8));
        unsigned long long* tp = rep == 5 ? trace : nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(gsr::g_radix_trace), &tp, sizeof tp));
        CK(hipEventRecord(e0));
        CK(gsr::radix_sort_pairs(scratch, n, bits, ka, kb, va, vb, true, true, &ks, &vs, 0));
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep < 5) best_total = std::min(best_total, ms);
    }
    // the traced run was a full sort: the trace holds the LAST pass (each pass overwrites the slots)
    std::vector<unsigned long long> t(trace_words);
    CK(hipMemcpy(t.data(), trace, trace_words * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t_end = 0; size_t blocks = 0;
    double ph[4] = {0, 0, 0, 0};
    for (size_t b = 0; b * 8 + 4 < trace_words; ++b) {
        const unsigned long long* r = &t[b * 8];
        if (!r[0] || !r[4]) continue;
        ++blocks; t0 = std::min(t0, r[0]); t_end = std::max(t_end, r[4]);
        for (int k = 0; k < 4; ++k) ph[k] += (double)(r[k + 1] - r[k]) * 0.01;  // us
    }
    printf("   starts(us) of workgroup 0/25/50/75/100%%:");
    for (int q = 0; q <= 4; ++q) { size_t b = (blocks - 1) * q / 4; printf(" %.1f", (double)(t[b * 8] - t0) * 0.01); }
    printf("\n");
    printf("n=%u bits=%d: sort %.1f us (%d passes) | last scatter: %zu wgs, span %.1f us, mean phase us: load+rank %.1f scans %.1f park %.1f write %.1f\n",
           n, bits, best_total * 1e3, (bits + 7) / 8, blocks, (double)(t_end - t0) * 0.01, ph[0] / blocks, ph[1] / blocks,
           ph[2] / blocks, ph[3] / blocks);
    // check against the host
    std::vector<uint32_t> gk(n), gv(n);
    CK(hipMemcpy(gk.data(), ks, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gv.data(), vs, n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;
    const uint32_t keep = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return (h[a] & keep) < (h[b] & keep); });
    size_t bad = 0;
    for (uint32_t i = 0; i < n; ++i) bad += gv[i] != idx[i] || gk[i] != h[idx[i]];
    printf("host check: %zu mismatches\n", bad);
    return bad != 0;
}
