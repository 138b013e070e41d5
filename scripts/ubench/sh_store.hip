// Micro-benchmark: writing one 192-byte SH-gradient record per lane.  (a) lane-strided 12-byte stores (what a
// one-lane-per-Gaussian kernel does naturally), (b) wave-cooperative coalesced 16-byte stores staged through LDS.
//   hipcc --offload-arch=gfx950 -O3 sh_store.hip -o sh_store
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct __attribute__((aligned(4))) F3 { float x, y, z; };

__global__ void __launch_bounds__(256) strided12(float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    F3* p = reinterpret_cast<F3*>(out + 48 * (size_t)i);
    const float v = (float)i;
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = F3{v + k, v * k, v - k};
}

__global__ void __launch_bounds__(256) staged(float* __restrict__ out, int n) {
    __shared__ float s[4][48 * 65];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = blockIdx.x * 256 + wave * 64;
    if (g0 >= n) return;
    float* mine = s[wave];
    const float v = (float)(g0 + lane);
#pragma unroll
    for (int k = 0; k < 16; ++k) {  // [coefficient word][lane], pitch 65: conflict-free both ways
        mine[(3 * k + 0) * 65 + lane] = v + k; mine[(3 * k + 1) * 65 + lane] = v * k; mine[(3 * k + 2) * 65 + lane] = v - k;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float4* dst = reinterpret_cast<float4*>(out + 48 * (size_t)g0);
    const int chunks = min(64, n - g0) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int c = k * 64 + lane;
        if (c < chunks) {
            const int g = c / 12, f = (c - g * 12) * 4;
            dst[c] = make_float4(mine[(f + 0) * 65 + g], mine[(f + 1) * 65 + g], mine[(f + 2) * 65 + g], mine[(f + 3) * 65 + g]);
        }
    }
}

int main() {
    const int n = 3000000;
    float* out; CK(hipMalloc(&out, (size_t)n * 48 * 4 + 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[2] = {"lane-strided 12 B stores", "LDS transpose + coalesced 16 B"};
    for (int v = 0; v < 2; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(strided12, dim3((n + 255) / 256), dim3(256), 0, 0, out, n);
            if (v == 1) hipLaunchKernelGGL(staged, dim3((n + 255) / 256), dim3(256), 0, 0, out, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-32s %7.1f us  %6.0f GB/s\n", names[v], best * 1e3, (double)n * 192 / (best * 1e-3) / 1e9);
    }
    return 0;
}
