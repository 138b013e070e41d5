// gsr_blend.hip -- front-to-back alpha compositing of the per-tile lists, for gfx950.
//
//   blend_quadrant_kernel <- renderCUDA   DGR/cuda_rasterizer/forward.cu:261-378
//   (DGR = sugar/gaussian_splatting/submodules/diff-gaussian-rasterization, under /root/reference)
//
// One wave64 per 8x8 QUADRANT of a 16x16 tile (4 single-wave workgroups per tile), one pixel per lane; no workgroup
// barriers.  Per (pixel, entry) the arithmetic is the reference's (forward.cu:331-364), op for op; the shape of the
// work is not:
//   * a quadrant stops as soon as ITS 64 pixels are done, and a tile whose list is long or never saturates does not
//     pin one 256-thread block for the whole launch;
//   * while staging a batch of 64 list entries each lane runs the conservative reach test of its entry against the
//     quadrant (gsr_device.h: splat_reaches_rect); the wave then walks only the set bits of the ballot, so entries
//     that cannot touch the quadrant cost no LDS read and no per-pixel work at all;
//   * entries are parked in LDS as one 48-byte record (one address register, three broadcast reads);
//   * which pixels have stopped is a wave-uniform 64-bit mask in scalar registers, combined only in uniform control
//     flow, so "any pixel live?" / "all done?" cost no vector instructions and the accumulate block is gated through
//     the exec mask (inverse_ballot);
//   * `power < skip_below` (skip_below = -ln(255 o) - 1e-4) skips expf for pairs that cannot reach alpha >= 1/255:
//     below it o * exp(power) is < 1/255 by a margin ~100x the combined rounding error of logf / expf / the products,
//     so the reference's `alpha < 1/255` test would have skipped the pair too;
//   * optionally a second per-Gaussian feature triple is composited in the same walk (kExtra).
//
// Depth slabs (inference calls, gsr_api.hip): the list of a tile arrives in up to kMaxSlabs segments, front to back,
// and a launch walks a range of them.  Between two launches the pixel state lives in the output images themselves --
// accumulated colour / depth / second feature set without the background term, the transmittance T in the alpha plane
// with its sign set once the pixel has stopped, the last contributor in n_contrib -- and a quadrant whose 64 pixels
// have all stopped writes its final values at once, marks itself in `quad_done` (later launches leave after one
// scalar load) and, when it completes its tile, the tile in `done_rows`, which is what lets the next slab drop that
// tile's pairs before they are expanded.  A pixel's sequence of operations is exactly that of one walk over the
// concatenated list: same bits.
#include "gsr_device.h"

// Profiling aid (python -m autovfx_amd.build --trace, scripts/kernel_trace.py --blend): every single-wave workgroup stamps the
// 100 MHz wall clock at its start and end.  Compiled out of the normal library.
#ifdef GSR_KERNEL_TRACE
__device__ unsigned long long* g_blend_trace = nullptr;
extern "C" __attribute__((visibility("default"))) int gsr_debug_set_blend_trace(void* device_words) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_blend_trace), &device_words, sizeof device_words);
}
#define GSR_LTRACE(id, slot) do { if (threadIdx.x == 0 && g_blend_trace) g_blend_trace[(size_t)(id) * 2 + (slot)] = wall_clock64(); } while (0)
#else
#define GSR_LTRACE(id, slot) do { } while (0)
#endif

namespace gsr {
namespace {

// C += feature * alpha * T (forward.cu:357-360) with the last product fused into the addition, fma(feature * alpha,
// T, C): what nvcc, which contracts by default, makes of that line on the reference's own hardware, and one
// full-rate fused instruction in place of a multiply and an add.  This library is otherwise built without
// contraction; this is the one place it is written out, because it is on the per-pair-per-pixel path and touches
// only the float images (transmittance, the stopping rule and every integer output do not depend on it).
//
// GSR_UNFUSED_BLEND (python -m autovfx_amd.build --unfused -> lib/libgsr_hip_unfused.so, a TEST build): the same line as the
// reference's sources say it when compiled without contraction -- multiply, multiply, add.  The reference's own kernels
// compiled for gfx950 with -ffp-contract=off (oracle/_ref/libgsr_ref_hip.so) then produce the SAME BITS in every image, which
// turns the image tolerance of the parity tests into an equality once per suite run (tests/test_parity_gpu.py:
// test_unfused_blend_build_equals_the_reference_kernels_bit_for_bit).
__device__ __forceinline__ float composite(float C, float feature, float alpha, float T) {
#ifdef GSR_UNFUSED_BLEND
    return C + (feature * alpha) * T;
#else
    return __builtin_fmaf(feature * alpha, T, C);
#endif
}
// Two channels at once: v_pk_mul_f32 + v_pk_fma_f32, written with a vector type so that the pairing does not depend
// on what the SLP vectorizer decides.  Same roundings as two calls of composite().
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f composite2(v2f C, v2f feature, float alpha, float T) {
#ifdef GSR_UNFUSED_BLEND
    return C + (feature * alpha) * (v2f){T, T};
#else
    return __builtin_elementwise_fma(feature * alpha, (v2f){T, T}, C);
#endif
}

__global__ void exp_selftest_kernel(uint32_t first_bits, uint32_t count, unsigned long long* mismatches) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float x = __uint_as_float(first_bits + i);
    const float a = expf(x), b = exp_nonpositive(x);
    if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) atomicAdd(mismatches, 1ull);
}

struct BlendArgs {
    int W, H, grid_x, num_tiles;
    BlendOrder order;
    BlendSegments segs;
    int seg_begin, seg_end;
    int fresh, final;
    const SplatRaster* raster;
    const float* features;
    const float* extra_features;
    const float* background;
    float* out_color;
    float* out_depth;
    float* out_alpha;
    float* out_extra;
    uint32_t* n_contrib;
    uint32_t* quad_done;
    uint32_t* done_rows;
    int row_words;
};

// This quadrant is finished for good: later launches skip it; the quadrant that completes a tile marks the tile.
__device__ __forceinline__ void mark_quadrant_done(const BlendArgs& a, int item) {
    if (threadIdx.x != 0) return;
    const uint32_t bit = 1u << (item & 31);
    const uint32_t old = atomicOr(&a.quad_done[item >> 5], bit);
    const int shift = (item & 31) & ~3;  // the four quadrants of a tile share one aligned nibble
    if ((((old | bit) >> shift) & 0xFu) == 0xFu) {
        const int tile = item >> 2, tx = tile % a.grid_x, ty = tile / a.grid_x;
        atomicOr(&a.done_rows[ty * a.row_words + (tx >> 5)], 1u << (tx & 31));
    }
}

// kExtra: a second per-Gaussian feature triple (extra_features[P,3] -> out_extra[3,H,W]) is composited in the
// same walk with the same alpha and transmittance -- what the reference's render() obtains from a second full
// rasterizer pass for its normal map (gaussian_renderer/__init__.py:176-184): identical arithmetic per channel,
// one list walk instead of two.
template <bool kExtra>
__global__ void __launch_bounds__(64, 8) blend_quadrant_kernel(BlendArgs a) {
    __shared__ BlendEntry s_entry[64];
    __shared__ float4 s_extra[kExtra ? 64 : 1];

    constexpr int kQ = kTile / 2;
    const int W = a.W, H = a.H;
    // the 4 quadrants of a tile share an XCD; large images: longest lists first inside the XCD's band (BlendOrder)
    const int item = a.order.counts == nullptr ? xcd_band_tile(blockIdx.x, 4 * a.num_tiles) : ordered_item(a.order, blockIdx.x);
    if (item < 0) return;   // (the ordered grid is padded to the largest band)
    const int tile = item >> 2, quad = item & 3;
    const int lane = threadIdx.x;
    const bool fresh = a.fresh != 0, final = a.final != 0;
    GSR_LTRACE(a.seg_begin * 40960 + blockIdx.x, 0);
    if (!fresh && ((a.quad_done[item >> 5] >> (item & 31)) & 1u)) return;  // finished by an earlier launch
    const int qx0 = (tile % a.grid_x) * kTile + kQ * (quad & 1);
    const int qy0 = (tile / a.grid_x) * kTile + kQ * (quad >> 1);
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const float fx = (float)px, fy = (float)py;
    const bool inside = px < W && py < H;
    const size_t plane = (size_t)W * (size_t)H;
    const size_t pid = (size_t)W * (size_t)py + (size_t)px;
    // Which pixels have stopped is a wave-uniform 64-bit mask in scalar registers: tests on it ("any pixel
    // live?", "all done?") cost no vector instructions, and it gates the per-pixel block through the
    // execution mask directly.
    unsigned long long done_mask = __ballot(!inside);
    if (done_mask == ~0ull) {  // quadrant entirely outside the image: nothing to write
        if (!final) mark_quadrant_done(a, item);
        return;
    }

    float T = 1.f, Eb = 0.f;
    v2f Crg = {0.f, 0.f}, Cbz = {0.f, 0.f}, Erg = {0.f, 0.f};  // red|green, blue|depth, second set red|green
    uint32_t last = 0u;
    if (!fresh) {  // resume the state an earlier launch parked in the output images
        bool stopped = false;
        if (inside) {
            const float ts = a.out_alpha[pid];
            stopped = (__float_as_uint(ts) >> 31) != 0u;
            T = fabsf(ts);
            Crg = (v2f){a.out_color[pid], a.out_color[plane + pid]};
            Cbz = (v2f){a.out_color[2 * plane + pid], a.out_depth[pid]};
            last = a.n_contrib[pid];
            if (kExtra) {
                Erg = (v2f){a.out_extra[pid], a.out_extra[plane + pid]};
                Eb = a.out_extra[2 * plane + pid];
            }
        }
        done_mask |= __ballot(stopped);
    }

    uint32_t seg_base = 0u;  // list positions of earlier segments (n_contrib counts through the concatenation)
    for (int seg = 0; seg < a.seg_begin; ++seg) {  // ... also of the segments earlier launches walked
        const uint2 earlier = a.segs.ranges[seg][tile];
        seg_base += earlier.y - earlier.x;
    }
    for (int seg = a.seg_begin; seg < a.seg_end && done_mask != ~0ull; ++seg) {
        const uint2 range = a.segs.ranges[seg][tile];
        const uint32_t count = range.y - range.x;
        const uint32_t* __restrict__ point_list = a.segs.point_list[seg] + range.x;

        float2 g_xy = make_float2(0.f, 0.f);
        float4 g_co = make_float4(0.f, 0.f, 0.f, 0.f);
        F3 g_rgb = {0.f, 0.f, 0.f}, g_ext = {0.f, 0.f, 0.f};
        float g_z = 0.f, g_skip = 0.f;
        auto gather = [&](uint32_t first) {
            const uint32_t e = first + (uint32_t)lane;
            if (e < count) {
                const uint32_t id = point_list[e];
                const float4* rec = reinterpret_cast<const float4*>(a.raster + id);  // 32 bytes, one cache line
                const float4 r0 = rec[0], r1 = rec[1];
                g_xy = make_float2(r0.x, r0.y);
                g_co = make_float4(r0.z, r0.w, r1.x, r1.y);
                g_z = r1.z;
                g_skip = r1.w;
                g_rgb = ld3(a.features + 3 * (size_t)id);
                if (kExtra) g_ext = ld3(a.extra_features + 3 * (size_t)id);
            }
        };
        if (count > 0) gather(0);

        for (uint32_t first = 0; first < count; first += 64) {
            const bool mine = first + (uint32_t)lane < count;
            unsigned long long todo = __ballot(mine && splat_reaches_rect(g_co, g_skip, g_xy, qx0, qy0, kQ, kQ));
            if (todo != 0ull) {
                __syncthreads();  // single-wave workgroup: orders this wave's LDS reads / writes only
                float4* rec = reinterpret_cast<float4*>(&s_entry[lane]);
                // The conic's diagonal is parked already multiplied by -0.5: scaling by a power of two commutes with every
                // rounding of -0.5 * (cxx*dx*dx + cyy*dy*dy), so the walk below gets the same bits with one multiply less
                // per (entry, pixel).
                rec[0] = make_float4(g_xy.x, g_xy.y, -0.5f * g_co.x, g_co.y);
                rec[1] = make_float4(-0.5f * g_co.z, g_skip, g_co.w, 0.f);
                rec[2] = make_float4(g_rgb.x, g_rgb.y, g_rgb.z, g_z);
                if (kExtra) s_extra[lane] = make_float4(g_ext.x, g_ext.y, g_ext.z, 0.f);
                __syncthreads();
            }
            if (first + 64 < count) gather(first + 64);

            while (todo != 0ull) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                // The entry's LDS offset is wave-uniform; parked in ONE vector register (opaque to the compiler, which
                // would otherwise re-create it from the scalar before each of the three reads of the record).
                uint32_t entry_offset;
                asm("v_mov_b32 %0, %1" : "=v"(entry_offset) : "s"(j * (int)sizeof(BlendEntry)));
                const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_entry) + entry_offset);
                const float4 ra = rec[0], rb = rec[1];
                struct { float x, y, mh_cxx, cxy; } ea = {ra.x, ra.y, ra.z, ra.w};   // mh_ = times minus one half
                struct { float mh_cyy, skip_below, opacity; } eb = {rb.x, rb.y, rb.z};
                // One list entry against this lane's pixel: forward.cu:331-364, same bits as
                // -0.5f * (cxx * dx * dx + cyy * dy * dy) - cxy * dx * dy.
                const float dx = ea.x - fx, dy = ea.y - fy;
                const float power = (ea.mh_cxx * dx * dx + eb.mh_cyy * dy * dy) - ea.cxy * dx * dy;
                // (one ballot per comparison: the ballot of a conjunction goes through a vector register and back)
                const unsigned long long live = __ballot(!(power > 0.0f)) & __ballot(!(power < eb.skip_below)) & ~done_mask;
                if (live == 0ull) continue;
                // From here every lane computes (a vector instruction costs the same with 1 or 64 lanes enabled);
                // the outcome of a lane that is not live is masked out below.  All masks stay wave-uniform scalars
                // because they are only combined in uniform control flow.
                const float alpha = fminf(0.99f, eb.opacity * exp_nonpositive(power));
                const unsigned long long blends = live & __ballot(!(alpha < 1.0f / 255.0f));
                if (blends == 0ull) continue;
                const float test_T = T * (1.f - alpha);
                const unsigned long long stops = blends & __ballot(test_T < 0.0001f);
                done_mask |= stops;
                const unsigned long long adds = blends & ~stops;
                if (adds != 0ull && __builtin_amdgcn_inverse_ballot_w64(adds)) {
                    const float4 c = rec[2];  // r g b z
                    Crg = composite2(Crg, (v2f){c.x, c.y}, alpha, T);
                    Cbz = composite2(Cbz, (v2f){c.z, c.w}, alpha, T);
                    if (kExtra) {
                        const float4 e = s_extra[j];
                        Erg = composite2(Erg, (v2f){e.x, e.y}, alpha, T);
                        Eb = composite(Eb, e.z, alpha, T);
                    }
                    T = test_T;
                    last = seg_base + first + (uint32_t)j + 1u;
                }
                if (done_mask == ~0ull) break;
            }
            if (done_mask == ~0ull) break;
        }
        seg_base += count;
    }

    const bool all_done = done_mask == ~0ull;
    if (final || all_done) {
        if (inside) {
            a.out_alpha[pid] = 1.f - T;
            if (a.n_contrib != nullptr) a.n_contrib[pid] = last;
            a.out_color[pid] = Crg.x + T * a.background[0];
            a.out_color[plane + pid] = Crg.y + T * a.background[1];
            a.out_color[2 * plane + pid] = Cbz.x + T * a.background[2];
            a.out_depth[pid] = Cbz.y;
            if (kExtra) {
                a.out_extra[pid] = Erg.x + T * a.background[0];
                a.out_extra[plane + pid] = Erg.y + T * a.background[1];
                a.out_extra[2 * plane + pid] = Eb + T * a.background[2];
            }
        }
        if (!final) mark_quadrant_done(a, item);
    } else if (inside) {  // park the state for the next slab's launch (T > 0 always: the sign is free)
        const bool stopped = (done_mask >> lane) & 1ull;
        a.out_alpha[pid] = stopped ? -T : T;
        a.n_contrib[pid] = last;
        a.out_color[pid] = Crg.x;
        a.out_color[plane + pid] = Crg.y;
        a.out_color[2 * plane + pid] = Cbz.x;
        a.out_depth[pid] = Cbz.y;
        if (kExtra) {
            a.out_extra[pid] = Erg.x;
            a.out_extra[plane + pid] = Erg.y;
            a.out_extra[2 * plane + pid] = Eb;
        }
    }
    GSR_LTRACE(a.seg_begin * 40960 + blockIdx.x, 1);
}

} // namespace

hipError_t launch_blend(const Camera& cam, const BlendSegments& segs, int seg_begin, int seg_end, bool fresh, bool final,
                        const SplatRaster* raster, const float* features, const float* background, float* out_color,
                        float* out_depth, float* out_alpha, uint32_t* n_contrib, uint32_t* quad_done, uint32_t* done_rows,
                        int row_words, hipStream_t stream, const float* extra_features, float* out_extra, const BlendOrder* order) {
    BlendArgs a;
    a.W = cam.width; a.H = cam.height; a.grid_x = cam.grid_x; a.num_tiles = cam.grid_x * cam.grid_y;
    a.segs = segs; a.seg_begin = seg_begin; a.seg_end = seg_end; a.fresh = fresh ? 1 : 0; a.final = final ? 1 : 0;
    a.raster = raster; a.features = features; a.extra_features = extra_features; a.background = background;
    a.out_color = out_color; a.out_depth = out_depth; a.out_alpha = out_alpha; a.out_extra = out_extra;
    a.n_contrib = n_contrib; a.quad_done = quad_done; a.done_rows = done_rows; a.row_words = row_words;
    int blocks = 4 * a.num_tiles;
    a.order = BlendOrder{nullptr, nullptr, 0, 0, 0};
    if (order != nullptr && order->counts != nullptr) {
        a.order = *order;
        blocks = 8 * 4 * order->cap;   // every XCD gets workgroups for the largest band; the surplus leaves at once
    }
    if (extra_features != nullptr)
        hipLaunchKernelGGL(blend_quadrant_kernel<true>, dim3(blocks), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL(blend_quadrant_kernel<false>, dim3(blocks), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_exp_selftest(uint32_t first_bits, uint32_t count, unsigned long long* mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(exp_selftest_kernel, dim3(div_up((int)count, 256)), dim3(256), 0, stream, first_bits, count, mismatches);
    return hipGetLastError();
}

} // namespace gsr
