// Micro-benchmark: how fast does gfx950 start waves?  A kernel that does one store per lane, launched with
// different workgroup sizes, static LDS sizes and VGPR footprints.  Prints waves per microsecond.
//   hipcc --offload-arch=gfx950 -O3 wave_launch.hip -o wave_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int kLds, int kVgprs>
__global__ void touch(uint32_t* out) {
    __shared__ uint32_t pad[kLds / 4 + 1];
    if (kLds > 0 && threadIdx.x == 0) pad[0] = blockIdx.x;
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (kVgprs > 32) {  // keep kVgprs registers live across a barrier
        uint32_t r[kVgprs > 0 ? kVgprs : 1];
#pragma unroll
        for (int i = 0; i < kVgprs; ++i) r[i] = v * (i + 1);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kVgprs; ++i) asm volatile("" : "+v"(r[i]));
#pragma unroll
        for (int i = 1; i < kVgprs; ++i) r[0] ^= r[i];
        v = r[0];
    }
    if (kLds > 0) { __syncthreads(); v += pad[0]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

template <int kLds, int kVgprs>
int run(const char* name, uint32_t* out, size_t lanes, int threads) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((touch<kLds, kVgprs>), dim3((unsigned)(lanes / threads)), dim3(threads), 0, 0, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-28s wg=%4d waves=%7zu  %8.1f us  %7.0f waves/us  (%.0f GB/s of stores)\n", name, threads, lanes / 64, best * 1e3,
           lanes / 64 / (best * 1e3), lanes * 4 / (best * 1e-3) / 1e9);
    return 0;
}

int main() {
    const size_t lanes = 3u << 20;  // ~3 M lanes = 49152 waves
    uint32_t* out;
    CK(hipMalloc(&out, lanes * 4));
    for (int threads : {64, 256, 1024}) {
        run<0, 0>("no lds, few vgprs", out, lanes, threads);
        run<0, 96>("no lds, ~96 vgprs", out, lanes, threads);
        run<16384, 0>("16 KB lds", out, lanes, threads);
        run<40960, 0>("40 KB lds", out, lanes, threads);
    }
    for (size_t l : {(size_t)1 << 18, (size_t)1 << 20, (size_t)1 << 22, (size_t)1 << 24}) run<0, 0>("size sweep", out = out, l <= lanes ? l : lanes, 256);
    return 0;
}
