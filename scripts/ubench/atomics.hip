// Micro-benchmark: throughput of global atomicAdd on a few thousand counters (tile cursors), with and
// without using the returned value, plus scattered 8-byte stores into per-tile segments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// tile pattern: each "gaussian" g touches a 2x3 block of tiles around a random centre (like a real splat)
__device__ __forceinline__ uint32_t tile_of(uint32_t i, uint32_t gx, uint32_t gy) {
    const uint32_t g = i / 6, k = i % 6;
    const uint32_t h = hash(g);
    const uint32_t cx = h % (gx - 2), cy = (h >> 16) % (gy - 1);
    return (cy + k / 3) * gx + cx + k % 3;
}

__global__ void count_noret(uint32_t n, uint32_t gx, uint32_t gy, uint32_t* cnt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cnt[tile_of(i, gx, gy)], 1u);
}
__global__ void count_ret(uint32_t n, uint32_t gx, uint32_t gy, uint32_t* cnt, const uint32_t* off, uint64_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t t = tile_of(i, gx, gy);
        const uint32_t slot = off[t] + atomicAdd(&cnt[t], 1u);
        out[slot] = ((uint64_t)hash(i) << 32) | i;
    }
}

int main() {
    const uint32_t gx = 120, gy = 68, T = gx * gy;
    for (uint32_t n : {3500000u * 1u, 8500000u, 13500000u}) {
        uint32_t *cnt, *off; uint64_t* out;
        CK(hipMalloc(&cnt, T * 4)); CK(hipMalloc(&off, T * 4)); CK(hipMalloc(&out, (size_t)n * 8));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        float ms1 = 0, ms2 = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(cnt, 0, T * 4));
            CK(hipEventRecord(a)); count_noret<<<(n + 255) / 256, 256>>>(n, gx, gy, cnt); CK(hipEventRecord(b));
            CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms1, a, b));
        }
        std::vector<uint32_t> h(T), o(T);
        CK(hipMemcpy(h.data(), cnt, T * 4, hipMemcpyDeviceToHost));
        uint32_t run = 0; for (uint32_t t = 0; t < T; ++t) { o[t] = run; run += h[t]; }
        CK(hipMemcpy(off, o.data(), T * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(cnt, 0, T * 4));
            CK(hipEventRecord(a)); count_ret<<<(n + 255) / 256, 256>>>(n, gx, gy, cnt, off, out); CK(hipEventRecord(b));
            CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms2, a, b));
        }
        printf("n=%u pairs, T=%u: count(no return) %.3f ms = %.1f G atomics/s ; slot+scatter(returning, 8B store) %.3f ms = %.1f G/s ; total=%u\n",
               n, T, ms1, n / ms1 * 1e-6, ms2, n / ms2 * 1e-6, run);
        hipFree(cnt); hipFree(off); hipFree(out);
    }
    return 0;
}
