// Micro-benchmark: does a DEPENDENT chain built from v_pk_fma_f32 issue faster than the scalar chain it replaces?
// The blend's per-entry power term: 2 sub + 4 mul (x and y axes) as scalar ops, or 3 dependent v_pk_fma_f32 on (x, y)
// pairs; followed by the same 4 scalar ops.  8 waves per SIMD, every SIMD busy.
//   hipcc --offload-arch=gfx950 -O3 pk_chain.hip -o pk_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int kMode>
__global__ void __launch_bounds__(256) spin(float* out, int iters) {
    float ex = threadIdx.x * 0.001f, ey = ex + 0.5f, fx = 3.f, fy = 4.f, ca = -0.25f, cc = -0.125f, cb = 0.01f, acc = 0.f;
    v2f e = {ex, ey}, nf = {-fx, -fy}, c = {ca, cc}, one = {1.f, 1.f}, nz = {-0.f, -0.f};
    asm volatile("" : "+v"(one), "+v"(nz));
    for (int i = 0; i < iters; ++i) {
        float p;
        if (kMode == 0) {   // scalar: 10 VALU
            asm volatile("v_sub_f32 %1, %2, %4\n v_sub_f32 %0, %3, %5\n"      // dx (in %1), dy (in %0)
                         "v_mul_f32 %2, %1, %6\n v_mul_f32 %3, %0, %7\n"      // clobber e as temporaries
                         "v_mul_f32 %2, %1, %2\n v_mul_f32 %3, %0, %3\n"
                         "v_mul_f32 %1, %1, %8\n v_add_f32 %2, %2, %3\n v_mul_f32 %1, %0, %1\n v_sub_f32 %0, %2, %1"
                         : "=&v"(p), "=&v"(acc), "+v"(ex), "+v"(ey) : "v"(fx), "v"(fy), "v"(ca), "v"(cc), "v"(cb));
        } else if (kMode == 1) {   // 3 dependent pk_fma + 4 scalar
            v2f d, m;
            asm volatile("v_pk_fma_f32 %0, %2, %4, %3\n v_pk_fma_f32 %1, %5, %0, %6\n v_pk_fma_f32 %1, %1, %0, %6"
                         : "=&v"(d), "=&v"(m) : "v"(e), "v"(nf), "v"(one), "v"(c), "v"(nz));
            asm volatile("v_mul_f32 %0, %5, %2\n v_add_f32 %1, %3, %4\n v_mul_f32 %0, %6, %0\n v_sub_f32 %0, %1, %0"
                         : "=&v"(p), "=&v"(acc) : "v"(d.x), "v"(m.x), "v"(m.y), "v"(cb), "v"(d.y));
            e.x = p * 1e-30f + e.x;
        } else if (kMode == 2) {   // 3 independent pk_fma only
            v2f a = e, b = nf, d = c;
            asm volatile("v_pk_fma_f32 %0, %0, %3, %4\n v_pk_fma_f32 %1, %1, %3, %4\n v_pk_fma_f32 %2, %2, %3, %4"
                         : "+v"(a), "+v"(b), "+v"(d) : "v"(one), "v"(nz));
            e = a; nf = b; c = d; p = a.x;
        } else {   // 6 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n"
                         "v_fma_f32 %3, %3, %6, %7\n v_fma_f32 %4, %4, %6, %7\n v_fma_f32 %5, %5, %6, %7"
                         : "+v"(ex), "+v"(ey), "+v"(fx), "+v"(fy), "+v"(ca), "+v"(cc) : "v"(one.x), "v"(nz.x));
            p = ex;
        }
        acc += p;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + ex + ey + e.x + e.y + nf.x + c.x + fx + fy + ca + cc;
}

template <int kMode>
int run(const char* name, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8, iters = 20000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(spin<kMode>, dim3(blocks), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double wave_iters = (double)blocks * 4 * iters;
    printf("%-44s %8.2f ms  %.1f cycles per iteration per SIMD at 2.4 GHz\n", name, best, 2.4e9 * 1024 / (wave_iters / (best * 1e-3)));
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("scalar chain (10 VALU + loop add)", out);
    run<1>("3 dependent pk_fma + 4 scalar (+2 loop ops)", out);
    run<2>("3 independent pk_fma (+ loop add)", out);
    run<3>("6 independent v_fma (+ loop add)", out);
    return 0;
}
