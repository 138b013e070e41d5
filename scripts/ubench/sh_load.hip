// Micro-benchmark: reading one 192-byte SH record per lane.  (a) lane-strided 12-byte loads (what a
// one-lane-per-Gaussian kernel does naturally), (b) lane-strided 16-byte loads, (c) wave-cooperative coalesced
// 16-byte loads staged through LDS in a [coefficient][lane] layout with a 65-word pitch.
//   hipcc --offload-arch=gfx950 -O3 sh_load.hip -o sh_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct __attribute__((aligned(4))) F3 { float x, y, z; };

__global__ void __launch_bounds__(256) strided12(const float* __restrict__ sh, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const F3* p = reinterpret_cast<const F3*>(sh + 48 * (size_t)i);
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const F3 v = p[k]; a += v.x * (k + 1); b += v.y * (k + 2); c += v.z * (k + 3); }
    out[i] = a + b + c;
}

__global__ void __launch_bounds__(256) strided16(const float* __restrict__ sh, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* p = reinterpret_cast<const float4*>(sh + 48 * (size_t)i);
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) { const float4 v = p[k]; a += v.x * (k + 1) + v.y * (k + 2) + v.z * (k + 3) + v.w * (k + 4); }
    out[i] = a;
}

__global__ void __launch_bounds__(256) staged(const float* __restrict__ sh, float* __restrict__ out, int n) {
    __shared__ float s[4][48 * 65];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = blockIdx.x * 256 + wave * 64;  // first Gaussian of this wave
    if (g0 >= n) return;
    float* mine = s[wave];
    const float4* src = reinterpret_cast<const float4*>(sh + 48 * (size_t)g0);
    const int chunks = min(64, n - g0) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int c = k * 64 + lane;  // 16-byte chunk of the wave's 12 KB
        if (c < chunks) {
            const float4 v = src[c];
            const int g = c / 12, f = (c - g * 12) * 4;
            mine[(f + 0) * 65 + g] = v.x; mine[(f + 1) * 65 + g] = v.y; mine[(f + 2) * 65 + g] = v.z; mine[(f + 3) * 65 + g] = v.w;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);  // single wave owns its slice: no workgroup barrier needed
    __builtin_amdgcn_wave_barrier();
    const int i = g0 + lane;
    if (i >= n) return;
    float a = 0.f;
#pragma unroll
    for (int f = 0; f < 48; ++f) a += mine[f * 65 + lane] * (f + 1);
    out[i] = a;
}

int main() {
    const int n = 3000000;
    float *sh, *out;
    CK(hipMalloc(&sh, (size_t)n * 48 * 4 + 256)); CK(hipMalloc(&out, (size_t)n * 4));
    CK(hipMemset(sh, 0, (size_t)n * 48 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"lane-strided 12 B loads", "lane-strided 16 B loads", "coalesced + LDS transpose"};
    for (int v = 0; v < 3; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(strided12, dim3((n + 255) / 256), dim3(256), 0, 0, sh, out, n);
            if (v == 1) hipLaunchKernelGGL(strided16, dim3((n + 255) / 256), dim3(256), 0, 0, sh, out, n);
            if (v == 2) hipLaunchKernelGGL(staged, dim3((n + 255) / 256), dim3(256), 0, 0, sh, out, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-28s %7.1f us  %6.0f GB/s\n", names[v], best * 1e3, (double)n * 196 / (best * 1e-3) / 1e9);
    }
    return 0;
}
