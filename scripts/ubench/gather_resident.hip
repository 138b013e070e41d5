// Where do bin_gather_kernel's random 16-byte gathers come from?  The counters bill every gather as a 128-byte request
// from L2 to the fabric (profiles/r02_pmc_calibration.csv), but the fabric side has the 256 MiB Infinity Cache, and the
// table bin_gather reads (16 B x P = 48 MB at 3 M Gaussians) was written by the kernel before it.  This program times the
// same access pattern -- n random 16-byte records, 4 per lane, out of a table that a kernel has just written -- for table
// sizes from 12 MB to 1 GiB: a table that fits the Infinity Cache is gathered at a multiple of the rate of one that has to
// come from HBM.  usage: hipcc --offload-arch=gfx950 -O3 gather_resident.hip -o gather_resident && ./gather_resident
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void __launch_bounds__(256) produce(uint4* __restrict__ t, uint32_t records) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < records) t[i] = make_uint4(i, i ^ 0x55u, i * 3u, ~i);
}
__global__ void __launch_bounds__(256) gather4(const uint4* __restrict__ t, uint32_t records, uint32_t n, uint32_t seed, uint32_t* __restrict__ out) {
    const uint32_t k0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (k0 + j < n) { const uint4 r = t[hash32((k0 + j) ^ seed) % records]; acc += r.x + r.y + r.z + r.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint32_t n = 2300000;   // gathers per launch: C3's visible splats
    uint32_t* out = nullptr;
    CK(hipMalloc((void**)&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("table_MB,records,gathers,us_per_launch,G_gathers_per_s,GBps_of_128B_lines\n");
    for (size_t mb : {12, 48, 96, 192, 384, 1024}) {
        const uint32_t records = (uint32_t)(mb * 1024 * 1024 / 16);
        uint4* t = nullptr;
        CK(hipMalloc((void**)&t, (size_t)records * 16));
        float best = 1e9f;
        for (int rep = 0; rep < 8; ++rep) {
            hipLaunchKernelGGL(produce, dim3((records + 255) / 256), dim3(256), 0, 0, t, records);   // the table is fresh, as bins[] is
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(gather4, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, t, records, n, 977u * rep, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%zu,%u,%u,%.1f,%.1f,%.0f\n", mb, records, n, best * 1e3, n / (best * 1e-3) / 1e9, n * 128.0 / (best * 1e-3) / 1e9);
        CK(hipFree(t));
    }
    return 0;
}
