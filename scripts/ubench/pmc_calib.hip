// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access patterns this
// library's kernels use (the guide calibrates only wide coalesced reads: MI355X_MICROARCH.md, "HBM"):
//   read_stream16 / read_stream4   coalesced 16-byte / 4-byte loads per lane           (projection inputs, sort passes)
//   read_strided192                one 192-byte record per lane, 12-byte loads          (SH coefficients)
//   gather32 / gather16 / gather12 one random 32 / 16 / 12-byte record per lane         (blend: raster record, colour;
//                                                                                        binning: splat record)
//   write_stream16 / write_stream4 coalesced stores; write_strided32: one 32-byte record per lane (raster records)
//   scatter4                       random 4-byte stores                                 (radix scatter before the LDS park)
// Every table is 1 GiB (four times the 256 MiB Infinity Cache) and touched once, so what is asked for has to come from
// HBM.  Usage (two passes, one counter each):
//   hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -- ./pmc_calib
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -- ./pmc_calib
// The program prints, per kernel, the bytes asked for and the bytes of the 64-byte lines they touch; scripts/pmc_calib_reduce.py
// divides the counters by them.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr size_t kTableBytes = size_t(1) << 30;
constexpr int kLanes = 1 << 24;  // 16 M lanes per kernel

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) read_stream16(const float4* __restrict__ t, float* __restrict__ out, int per_lane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a = 0.f;
    for (int k = 0; k < per_lane; ++k) { const float4 v = t[i + (size_t)k * kLanes]; a += v.x + v.y + v.z + v.w; }
    if (a == 12345.f) out[i] = a;
}
__global__ void __launch_bounds__(256) read_stream4(const float* __restrict__ t, float* __restrict__ out, int per_lane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a = 0.f;
    for (int k = 0; k < per_lane; ++k) a += t[i + (size_t)k * kLanes];
    if (a == 12345.f) out[i] = a;
}
struct __attribute__((aligned(4))) F3 { float x, y, z; };
__global__ void __launch_bounds__(256) read_strided192(const float* __restrict__ t, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // 5.59 M records fit 1 GiB
    const F3* p = reinterpret_cast<const F3*>(t + 48 * i);
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const F3 v = p[k]; a += v.x + v.y + v.z; }
    if (a == 12345.f) out[i] = a;
}
template <int kBytes>
__global__ void __launch_bounds__(256) gather(const char* __restrict__ t, float* __restrict__ out, uint32_t records) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t r = hash32(i) % records;   // (a permutation is not needed: 16 M draws from 33-89 M records, few repeats)
    const char* p = t + (size_t)r * kBytes;
    float a = 0.f;
    if (kBytes == 32) { const float4 u = *reinterpret_cast<const float4*>(p), v = *reinterpret_cast<const float4*>(p + 16); a = u.x + u.w + v.x + v.w; }
    if (kBytes == 16) { const float4 u = *reinterpret_cast<const float4*>(p); a = u.x + u.y + u.z + u.w; }
    if (kBytes == 12) { const F3 u = *reinterpret_cast<const F3*>(p); a = u.x + u.y + u.z; }
    if (a == 12345.f) out[i] = a;
}
__global__ void __launch_bounds__(256) write_stream16(float4* __restrict__ t, int per_lane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < per_lane; ++k) t[i + (size_t)k * kLanes] = make_float4(1.f, 2.f, 3.f, (float)k);
}
__global__ void __launch_bounds__(256) write_stream4(float* __restrict__ t, int per_lane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < per_lane; ++k) t[i + (size_t)k * kLanes] = (float)k;
}
__global__ void __launch_bounds__(256) write_strided32(float4* __restrict__ t) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    t[2 * i] = make_float4(1.f, 2.f, 3.f, 4.f);
    t[2 * i + 1] = make_float4(5.f, 6.f, 7.f, 8.f);
}
__global__ void __launch_bounds__(256) scatter4(uint32_t* __restrict__ t, uint32_t words) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    t[hash32(i) % words] = i;
}

int main() {
    char* table = nullptr;
    float* out = nullptr;
    CK(hipMalloc((void**)&table, kTableBytes));
    CK(hipMalloc((void**)&out, sizeof(float) * kLanes));
    CK(hipMemset(table, 0, kTableBytes));
    const int blocks = kLanes / 256;
    auto flush = [&]() { return hipDeviceSynchronize(); };
    printf("kernel,asked_bytes,line64_bytes\n");
    // reads
    hipLaunchKernelGGL(read_stream16, dim3(blocks), dim3(256), 0, 0, (const float4*)table, out, 4); CK(flush());
    printf("read_stream16,%zu,%zu\n", (size_t)kLanes * 16 * 4, (size_t)kLanes * 16 * 4);
    hipLaunchKernelGGL(read_stream4, dim3(blocks), dim3(256), 0, 0, (const float*)table, out, 16); CK(flush());
    printf("read_stream4,%zu,%zu\n", (size_t)kLanes * 4 * 16, (size_t)kLanes * 4 * 16);
    const int recs192 = 5 * 1024 * 1024;  // 960 MiB
    hipLaunchKernelGGL(read_strided192, dim3(recs192 / 256), dim3(256), 0, 0, (const float*)table, out); CK(flush());
    printf("read_strided192,%zu,%zu\n", (size_t)recs192 * 192, (size_t)recs192 * 192);
    hipLaunchKernelGGL(gather<32>, dim3(blocks), dim3(256), 0, 0, table, out, (uint32_t)(kTableBytes / 32)); CK(flush());
    printf("gather<32>,%zu,%zu\n", (size_t)kLanes * 32, (size_t)kLanes * 64);
    hipLaunchKernelGGL(gather<16>, dim3(blocks), dim3(256), 0, 0, table, out, (uint32_t)(kTableBytes / 16)); CK(flush());
    printf("gather<16>,%zu,%zu\n", (size_t)kLanes * 16, (size_t)kLanes * 64);
    hipLaunchKernelGGL(gather<12>, dim3(blocks), dim3(256), 0, 0, table, out, (uint32_t)(kTableBytes / 12)); CK(flush());
    printf("gather<12>,%zu,%zu\n", (size_t)kLanes * 12, (size_t)kLanes * (64 + 64 * 8 / 64));  // 1 in 8 straddles two lines (offsets 56, 60 of 16)
    // writes
    hipLaunchKernelGGL(write_stream16, dim3(blocks), dim3(256), 0, 0, (float4*)table, 4); CK(flush());
    printf("write_stream16,%zu,%zu\n", (size_t)kLanes * 16 * 4, (size_t)kLanes * 16 * 4);
    hipLaunchKernelGGL(write_stream4, dim3(blocks), dim3(256), 0, 0, (float*)table, 16); CK(flush());
    printf("write_stream4,%zu,%zu\n", (size_t)kLanes * 4 * 16, (size_t)kLanes * 4 * 16);
    hipLaunchKernelGGL(write_strided32, dim3(blocks), dim3(256), 0, 0, (float4*)table); CK(flush());
    printf("write_strided32,%zu,%zu\n", (size_t)kLanes * 32, (size_t)kLanes * 32);
    hipLaunchKernelGGL(scatter4, dim3(blocks), dim3(256), 0, 0, (uint32_t*)table, (uint32_t)(kTableBytes / 4)); CK(flush());
    printf("scatter4,%zu,%zu\n", (size_t)kLanes * 4, (size_t)kLanes * 64);
    CK(hipFree(table));
    CK(hipFree(out));
    return 0;
}
