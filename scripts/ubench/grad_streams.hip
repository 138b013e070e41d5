// Micro-benchmark: the MEMORY PATTERN of preprocess_backward_kernel without its arithmetic -- what the access pattern alone costs at C3.
// Per Gaussian (one lane): read radius (4 B); for a visible one (77 %) read its 64-byte line of sums; for one with a gradient (20 % of the
// visible) read 250 B of parameters lane-strided (xyz 12, scale 12, rotation 16, opacity 4, SH 192: the forward's layout); write
// dL_dmean2D 12, dL_dopacity 4, dL_dmean3D 12, dL_dscale 12, dL_drot 16 lane-strided (contiguous over a wave) and dL_dsh 192 through the
// compacted LDS stage as whole 16-byte chunks (zeros for Gaussians without a gradient).  Variants: (a) exactly that; (b) without the
// parameter reads; (c) outputs only (pure fill of the six gradient arrays); (d) a float4 copy of the same number of bytes as (a).
//   hipcc --offload-arch=gfx950 -O3 grad_streams.hip -o grad_streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct __attribute__((aligned(4))) F3 { float x, y, z; };
struct __attribute__((aligned(16))) F4 { float x, y, z, w; };
constexpr int kSlots = 32, kPitch = 49;

struct Args {
    int P;
    const int* radii; const float4* accum; const float* xyz; const float* scales; const float* rots; const float* opac; const float* shs;
    float* d2d; float* dop; float* d3d; float* dscale; float* drot; float* dsh;
};

template <int kMode>   // 0: full pattern, 1: no parameter reads, 2: outputs only
__global__ void __launch_bounds__(256) pattern(Args g) {
    __shared__ float s_stage[4][kSlots * kPitch];
    const int idx = blockIdx.x * 256 + threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bool live = false;
    float acc = 0.f;
    if (idx < g.P) {
        if (kMode < 2) {
            const bool visible = g.radii[idx] > 0;
            if (visible) {
                const float4* line = g.accum + 4 * (size_t)idx;
                const float4 a = line[0], b = line[1], c = line[2], d = line[3];
                acc = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
                live = acc != 0.f;
            }
        } else {
            live = (idx % 5) == 0;
        }
    }
    const unsigned long long mask = __ballot(live);
    const int slot = __popcll(mask & ((1ull << lane) - 1ull));
    float* stage = slot < kSlots ? s_stage[wave] + slot * kPitch : nullptr;
    if (idx < g.P) {
        float v = acc;
        if (live && kMode == 0) {
            const F3 m = *reinterpret_cast<const F3*>(g.xyz + 3 * (size_t)idx);
            const F3 s = *reinterpret_cast<const F3*>(g.scales + 3 * (size_t)idx);
            const F4 q = *reinterpret_cast<const F4*>(g.rots + 4 * (size_t)idx);
            v += m.x + m.y + m.z + s.x + s.y + s.z + q.x + q.y + q.z + q.w + g.opac[idx];
            const float* sh = g.shs + 48 * (size_t)idx;
#pragma unroll
            for (int k = 0; k < 16; ++k) { const F3 c = *reinterpret_cast<const F3*>(sh + 3 * k); v += c.x * (k + 1) + c.y - c.z; }
        }
        if (live && stage != nullptr) {
#pragma unroll
            for (int f = 0; f < 48; ++f) stage[f] = v + f;
        }
        *reinterpret_cast<F3*>(g.d2d + 3 * (size_t)idx) = F3{v, v, v};
        g.dop[idx] = v;
        *reinterpret_cast<F3*>(g.d3d + 3 * (size_t)idx) = F3{v, -v, v};
        *reinterpret_cast<F3*>(g.dscale + 3 * (size_t)idx) = F3{v, v, -v};
        *reinterpret_cast<F4*>(g.drot + 4 * (size_t)idx) = F4{v, v, v, -v};
    }
    const int g0 = blockIdx.x * 256 + wave * 64;
    if (g0 >= g.P) return;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const float* mine = s_stage[wave];
    float4* dst = reinterpret_cast<float4*>(g.dsh + 48 * (size_t)g0);
    const int chunks = min(64, g.P - g0) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int c = k * 64 + lane;
        if (c < chunks) {
            const int gi = c / 12, f = (c - gi * 12) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((mask >> gi) & 1ull) {
                const int sl = __popcll(mask & ((1ull << gi) - 1ull));
                if (sl < kSlots) { const float* r = mine + sl * kPitch + f; v = make_float4(r[0], r[1], r[2], r[3]); }
            }
            dst[c] = v;
        }
    }
}

__global__ void __launch_bounds__(256) copy16(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

int main() {
    const int P = 3000000;
    std::vector<int> radii(P);
    std::vector<float> accum((size_t)P * 16, 0.f);
    srand(1);
    size_t visible = 0, live = 0;
    for (int i = 0; i < P; ++i) {
        const bool v = rand() % 100 < 77, l = v && rand() % 100 < 20;
        radii[i] = v ? 3 : 0;
        if (l) for (int k = 0; k < 16; ++k) accum[(size_t)i * 16 + k] = 1.f + k;
        visible += v; live += l;
    }
    Args g; g.P = P;
    int* d_radii; float *d_accum, *xyz, *sc, *rot, *op, *sh;
    CK(hipMalloc(&d_radii, P * 4)); CK(hipMalloc(&d_accum, (size_t)P * 64));
    CK(hipMemcpy(d_radii, radii.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_accum, accum.data(), (size_t)P * 64, hipMemcpyHostToDevice));
    CK(hipMalloc(&xyz, (size_t)P * 12)); CK(hipMalloc(&sc, (size_t)P * 12)); CK(hipMalloc(&rot, (size_t)P * 16)); CK(hipMalloc(&op, (size_t)P * 4));
    CK(hipMalloc(&sh, (size_t)P * 192));
    CK(hipMemset(xyz, 0, (size_t)P * 12)); CK(hipMemset(sc, 0, (size_t)P * 12)); CK(hipMemset(rot, 0, (size_t)P * 16)); CK(hipMemset(op, 0, (size_t)P * 4));
    CK(hipMemset(sh, 0, (size_t)P * 192));
    g.radii = d_radii; g.accum = reinterpret_cast<const float4*>(d_accum); g.xyz = xyz; g.scales = sc; g.rots = rot; g.opac = op; g.shs = sh;
    CK(hipMalloc(&g.d2d, (size_t)P * 12)); CK(hipMalloc(&g.dop, (size_t)P * 4)); CK(hipMalloc(&g.d3d, (size_t)P * 12)); CK(hipMalloc(&g.dscale, (size_t)P * 12));
    CK(hipMalloc(&g.drot, (size_t)P * 16)); CK(hipMalloc(&g.dsh, (size_t)P * 192 + 256));
    const double out_bytes = (double)P * (12 + 4 + 12 + 12 + 16 + 192);
    const double in_a = (double)P * 4 + (double)visible * 64 + (double)live * 236, in_b = (double)P * 4 + (double)visible * 64;
    const size_t copy_n = (size_t)((out_bytes + in_a) / 2 / 16);
    float4 *cin, *cout; CK(hipMalloc(&cin, copy_n * 16)); CK(hipMalloc(&cout, copy_n * 16)); CK(hipMemset(cin, 0, copy_n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"(a) the kernel's pattern", "(b) without parameter reads", "(c) the six outputs only", "(d) float4 copy, same bytes"};
    const double bytes[4] = {out_bytes + in_a, out_bytes + in_b, out_bytes, 2.0 * copy_n * 16};
    printf("P %d, visible %zu, with a gradient %zu\n", P, visible, live);
    for (int v = 0; v < 4; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 8; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(pattern<0>, dim3((P + 255) / 256), dim3(256), 0, 0, g);
            if (v == 1) hipLaunchKernelGGL(pattern<1>, dim3((P + 255) / 256), dim3(256), 0, 0, g);
            if (v == 2) hipLaunchKernelGGL(pattern<2>, dim3((P + 255) / 256), dim3(256), 0, 0, g);
            if (v == 3) hipLaunchKernelGGL(copy16, dim3((unsigned)((copy_n + 255) / 256)), dim3(256), 0, 0, cin, cout, copy_n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-32s %7.1f us  %6.0f MB  %6.0f GB/s\n", names[v], best * 1e3, bytes[v] / 1e6, bytes[v] / (best * 1e-3) / 1e9);
    }
    return 0;
}
