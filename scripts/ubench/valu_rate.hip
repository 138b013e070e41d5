// Micro-benchmark: issue rate of scalar fp32 vs packed fp32 vector instructions on gfx950 (per wave64, per SIMD).
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int kMode>
__global__ void __launch_bounds__(256) spin(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0000001f;
    for (int i = 0; i < iters; ++i) {
        if (kMode == 0) {  // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (kMode == 1) {  // 4 independent v_pk_mul_f32 (same 8 multiplies)
            asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n"
                         "v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((v2f){m, m}));
        } else if (kMode == 2) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (kMode == 3) {  // 4 v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((v2f){m, m}));
        } else if (kMode == 4) {  // 8 v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else {  // 8 v_cmp_lt_f32 into vcc
            asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                         "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int kMode>
int run(const char* name, float* out, int per_iter) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 8, iters = 20000;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(spin<kMode>, dim3(blocks), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double wave_instrs = (double)blocks * 4 * iters * per_iter;
    printf("%-22s %8.2f ms  %7.1f G wave-instr/s  = %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, best,
           wave_instrs / (best * 1e-3) / 1e9, 2.4e9 * 1024 / (wave_instrs / (best * 1e-3)));
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("v_mul_f32", out, 8);
    run<1>("v_pk_mul_f32", out, 4);
    run<2>("v_fma_f32", out, 8);
    run<3>("v_pk_fma_f32", out, 4);
    run<4>("v_exp_f32", out, 8);
    run<5>("v_cmp_lt_f32", out, 8);
    return 0;
}
