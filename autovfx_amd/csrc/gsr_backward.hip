// gsr_backward.hip -- the backward pass of the rasterizer.
//
//   render_backward_kernel      <- BACKWARD::renderCUDA   DGR/cuda_rasterizer/backward.cu:415-599
//   preprocess_backward_kernel  <- BACKWARD::preprocess   backward.cu:144-413 (+ SH :20-138, cov3D :278-342)
//
// Its own translation unit: the backward pass shares only the inline helpers of gsr_device.h with the forward
// kernels.  Like the rest of the library it is built with -fno-slp-vectorize (autovfx_amd/build.py): these two kernels
// are long stretches of scalar fp32 algebra, which the SLP vectorizer turns into half-rate packed multiplies / adds plus
// register shuffles -- 117 instead of 91 VGPRs in preprocess_backward and a training-style iteration at C3 10 % slower
// (2.90 vs 2.62 ms on MI355X).
#include "gsr_device.h"

// Census of render_backward_kernel's inner loop (python -m autovfx_amd.build --trace, scripts/backward_census.py): how many
// (quadrant, list entry) iterations reach each stage and how many of the 64 lanes contribute.  Compiled out of the normal
// library.  Words: 0 waves that walk a list, 1 staged batches, 2 entries past the quadrant reach test, 3 with a live pixel,
// 4 with a contributing pixel, 5 sum of contributing lanes, 6..11 entries by contributing lanes (1-2, 3-4, 5-8, 9-16, 17-32,
// 33-64), 12 sum of live lanes.
#ifdef GSR_KERNEL_TRACE
__device__ unsigned long long* g_backward_census = nullptr;
extern "C" __attribute__((visibility("default"))) int gsr_debug_set_backward_census(void* device_words) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_backward_census), &device_words, sizeof device_words);
}
#define GSR_BCENSUS(stmt) do { stmt; } while (0)
#else
#define GSR_BCENSUS(stmt) do { } while (0)
#endif

namespace gsr {
namespace {

// ================================================================================================
// BACKWARD PASS
//   render_backward_kernel     <- BACKWARD::renderCUDA       DGR/cuda_rasterizer/backward.cu:415-599
//   preprocess_backward_kernel <- computeCov2DCUDA           backward.cu:144-276
//                                 + preprocessCUDA (bwd)     backward.cu:346-413
//                                 + computeColorFromSH (bwd) backward.cu:20-138
//                                 + computeCov3D (bwd)       backward.cu:278-342
// Per-pixel and per-Gaussian arithmetic follow the reference's operation order.  What differs is
// how per-Gaussian sums are formed: the reference issues one atomicAdd per (pixel, Gaussian, value);
// here the 64 pixels of a quadrant are summed inside the wave first and one lane adds the result,
// so the number of global atomics drops 64x and their order (hence the last bits of the sums) is
// this library's, not the reference's -- parity for gradients is stated with a tolerance.
// ================================================================================================
// v + (v moved by a DPP control word): cross-lane add without LDS traffic.
template <int kCtrl>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, 0xF, 0xF, false);
    return v + __int_as_float(moved);
}

// Reduce-scatter of ten per-lane values over the wave with gfx950's half-swapping permutes: v_permlane32_swap
// exchanges the upper 32 lanes of one register with the lower 32 of another, so ONE swap + ONE add folds two
// values by a factor of two at once (lanes 0-31 then hold partial sums of the first, 32-63 of the second);
// v_permlane16_swap does the same between odd and even rows of 16.  After the two levels three registers hold
// the ten values, one per row, and four row-rotating DPP adds finish each: 28 instructions instead of the 60 of
// ten separate butterfly sums.
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fold32(float a, float b) {  // -> [sum pairs of a | sum pairs of b]
    const v2u_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float fold16(float a, float b) {  // rows -> [a r0+r1 | b r0+r1 | a r2+r3 | b r2+r3]
    const v2u_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float row_sum_all_lanes(float v) {  // every lane of a 16-lane row gets the row's sum
    v = dpp_add<0x128>(v);  // row_ror:8
    v = dpp_add<0x124>(v);  // row_ror:4
    v = dpp_add<0x122>(v);  // row_ror:2
    v = dpp_add<0x121>(v);  // row_ror:1
    return v;
}
constexpr int kAccumStride = 16;  // floats per Gaussian in the accumulation scratch: one 64-byte line
// slots: 0 r, 1 g, 2 b, 3 depth, 4 S_u, 5 S_v, 6 S_{u dx}, 7 S_{u dy}, 8 S_{v dy} (u = dL_dG G dx, v = dL_dG G dy: the factored
// sums behind dL_dmean2D and dL_dconic, finished per Gaussian by preprocess_backward_lane), 9 opacity; 10 - 12 the colour sums
// of a pass over the call's SECOND feature set (gsr_backward_raw with dL_dpix_normal: its geometry sums add to 4 - 9), 13 unused

// kDepthAlpha = false: the caller has no gradient for the depth and alpha images (dL_dpixel_depths / dL_dpixel_alphas are not
// read): their terms of dL_dalpha, the two accumulators behind them and the depth sum drop out of the loop -- the usual
// training loss (train.py:84-134, scene_representation.py:495-520) only looks at the colour image.
//
// kDet (GSR_OPT_BACKWARD_DETERMINISTIC): no atomics on floats.  The ten sums of a (quadrant, list entry) go, as plain stores, into
// the record of that (sorted list position, quadrant) in `det_partial` -- 10 floats, record index 4 * position + quadrant --
// and the quadrant's bit of the position is set in `det_bits` (4 bits per position, an integer atomicOr: its result does not
// depend on the order it is served in).  det_reduce_kernel then adds every Gaussian's records in ascending (position, quadrant)
// order: the same bits on every run, on every box.
template <bool kDepthAlpha, bool kDet>
__global__ void __launch_bounds__(64, 4) render_backward_kernel(
    int W, int H, int grid_x, int num_tiles, BlendSegments segs, int num_segs, const float* __restrict__ background,
    const SplatRaster* __restrict__ raster, const float* __restrict__ colors, const float* __restrict__ accum_alphas,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
    const float* __restrict__ dL_dpixel_depths, const float* __restrict__ dL_dpixel_alphas,
    float* __restrict__ accum /*[P,16], zero on entry*/, int colour_slot /*0; 10: a pass over the second feature set*/,
    float* __restrict__ det_partial, uint32_t* __restrict__ det_bits) {
    __shared__ BlendEntry s_entry[64];  // the forward's 48-byte record; the Gaussian id rides in its pad word

    constexpr int kQ = kTile / 2;
    const int item = xcd_band_tile(blockIdx.x, 4 * num_tiles);
    const int tile = item >> 2, quad = item & 3;
    const int lane = threadIdx.x;
    const int qx0 = (tile % grid_x) * kTile + kQ * (quad & 1);
    const int qy0 = (tile / grid_x) * kTile + kQ * (quad >> 1);
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const float fx = (float)px, fy = (float)py;
    const bool inside = px < W && py < H;
    const size_t pid = (size_t)W * (size_t)py + (size_t)px;
    const size_t plane = (size_t)W * (size_t)H;

    // The tile's list arrives in num_segs front-to-back segments (one per depth slab of the forward call; a full call has one).
    // A pixel's positions -- n_contrib among them -- count through their concatenation, as the forward blend counted them.
    uint32_t count = 0u;
    for (int k = 0; k < num_segs; ++k) {
        const uint2 r = segs.ranges[k][tile];
        count += r.y - r.x;
    }

    // forward results for this pixel (backward.cu:459-479)
    // which slot of a Gaussian's accumulation line this lane adds to after the reduce-scatter (see below)
    const int red_k = lane & 15, red_row = lane >> 4;
    const int my_slot = red_k == 0 ? colour_slot + ((red_row & 1) << 1 | (red_row >> 1))
                      : red_k == 1 ? 4 + ((red_row & 1) << 1 | (red_row >> 1))
                      : (red_k == 2 && (red_row & 1) == 0) ? 8 + (red_row >> 1) : -1;

    const float T_final = inside ? (1.f - accum_alphas[pid]) : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pid] : 0u;
    float dLr = 0.f, dLg = 0.f, dLb = 0.f, dLd = 0.f, dLa = 0.f;
    if (inside) {
        dLr = dL_dpixels[pid];
        dLg = dL_dpixels[plane + pid];
        dLb = dL_dpixels[2 * plane + pid];
        if (kDepthAlpha) {
            dLd = dL_dpixel_depths[pid];
            dLa = dL_dpixel_alphas[pid];
        }
    }
    // entries at list positions >= every pixel's last contributor are skipped by every pixel
    uint32_t walk = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) walk = max(walk, (uint32_t)__shfl_xor((int)walk, d));
    walk = min(walk, count);
    if (walk == 0) return;

    const unsigned long long inside_mask = __ballot(inside);
#ifdef GSR_KERNEL_TRACE
    unsigned long long cz[13] = {1ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_a = 0.f;  // accum_rec / accum_red / accum_rea, updated eagerly
    const float bg_dot = (background[0] * dLr + background[1] * dLg) + background[2] * dLb;  // left to right

    // back to front: the segments last to first, inside a segment batches of 64 positions, highest first
    uint32_t seg_end = count;   // concatenated position just behind the segment being walked
    for (int k = num_segs - 1; k >= 0; --k) {
    const uint2 range = segs.ranges[k][tile];
    const uint32_t* __restrict__ point_list = segs.point_list[k];
    const uint32_t seg_base = seg_end - (range.y - range.x);   // concatenated position of the segment's first entry
    seg_end = seg_base;
    if (walk <= seg_base) continue;                            // nobody's last contributor reaches into this segment
    for (uint32_t top = walk - seg_base < range.y - range.x ? walk - seg_base : range.y - range.x; top > 0; top = top > 64 ? top - 64 : 0) {
        const uint32_t first = top > 64 ? top - 64 : 0;  // batch covers the segment's positions [first, top)
        const uint32_t e = first + (uint32_t)lane;
        float2 g_xy = make_float2(0.f, 0.f);
        float4 g_co = make_float4(0.f, 0.f, 0.f, 0.f);
        F3 g_rgb = {0.f, 0.f, 0.f};
        float g_z = 0.f, g_skip = 0.f;
        uint32_t g_id = 0;
        const bool mine = e < top;
        if (mine) {
            g_id = point_list[range.x + e];
            const float4* rec = reinterpret_cast<const float4*>(raster + g_id);
            const float4 r0 = rec[0], r1 = rec[1];
            g_xy = make_float2(r0.x, r0.y);
            g_co = make_float4(r0.z, r0.w, r1.x, r1.y);
            g_z = r1.z;
            g_skip = r1.w;
            g_rgb = ld3(colors + 3 * (size_t)g_id);
        }
        unsigned long long todo = __ballot(mine && splat_reaches_rect(g_co, g_skip, g_xy, qx0, qy0, kQ, kQ));
        if (todo == 0ull) continue;
        GSR_BCENSUS(cz[1] += 1ull; cz[2] += (unsigned long long)__popcll(todo));
        __syncthreads();
        {
            float4* rec = reinterpret_cast<float4*>(&s_entry[lane]);
            rec[0] = make_float4(g_xy.x, g_xy.y, g_co.x, g_co.y);
            rec[1] = make_float4(g_co.z, g_skip, g_co.w, __uint_as_float(g_id));
            rec[2] = make_float4(g_rgb.x, g_rgb.y, g_rgb.z, g_z);
        }
        __syncthreads();

        while (todo != 0ull) {
            const int j = 63 - __builtin_clzll(todo);  // highest position first
            todo &= ~(1ull << j);
            const uint32_t pos = seg_base + first + (uint32_t)j;   // concatenated position: what n_contrib counts
            uint32_t entry_offset;  // as in the forward blend: one vector register for the record's three reads
            asm("v_mov_b32 %0, %1" : "=v"(entry_offset) : "s"(j * (int)sizeof(BlendEntry)));
            const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_entry) + entry_offset);
            const float4 ra = rec[0], rb = rec[1];
            struct { float x, y, cxx, cxy; } a = {ra.x, ra.y, ra.z, ra.w};
            struct { float cyy, skip_below, opacity; } b = {rb.x, rb.y, rb.z};
            const float dx = a.x - fx, dy = a.y - fy;
            const float power = -0.5f * (a.cxx * dx * dx + b.cyy * dy * dy) - a.cxy * dx * dy;
            // pixel predicates as wave-uniform masks, as in the forward blend (one ballot per comparison)
            const unsigned long long live = __ballot(pos < last_contributor) & __ballot(!(power > 0.0f)) &
                                            __ballot(!(power < b.skip_below)) & inside_mask;
            if (live == 0ull) continue;
            GSR_BCENSUS(cz[3] += 1ull; cz[12] += (unsigned long long)__popcll(live));
            const float G = exp_nonpositive(power);  // == expf on the contributing lanes' domain (power <= 0)
            const float alpha = fminf(0.99f, b.opacity * G);
            const unsigned long long contrib = live & __ballot(!(alpha < 1.0f / 255.0f));
            if (contrib == 0ull) continue;
            GSR_BCENSUS(const int n_ = __popcll(contrib); cz[4] += 1ull; cz[5] += (unsigned long long)n_;
                        cz[n_ <= 2 ? 6 : n_ <= 4 ? 7 : n_ <= 8 ? 8 : n_ <= 16 ? 9 : n_ <= 32 ? 10 : 11] += 1ull);
            const float4 cd = rec[2];  // r g b depth
            struct { float r, g, b, opacity; } c = {cd.x, cd.y, cd.z, b.opacity};
            const float z = cd.w;
            // The gradient block runs on ALL 64 lanes with alpha = G = 0 on the lanes that do not contribute to this entry:
            // with the accumulators updated eagerly (acc <- alpha c + (1 - alpha) acc right after use: the value the reference's
            // lazy `last_alpha * last_color + (1 - last_alpha) * accum_rec` (backward.cu:516-522) takes at the next contributing
            // entry, same operands, same roundings) every state update is the identity for alpha = 0 -- T * rcp(1) = T,
            // 0 c + 1 acc = acc -- and all ten partial sums come out as exact zeros.  No exec mask, no zero-fill of the sums,
            // no last_* registers (round 3: 10 + 5 moves per entry of a 68-instruction block).
            const bool contributes = __builtin_amdgcn_inverse_ballot_w64(contrib);
            const float alpha_m = contributes ? alpha : 0.f, G_m = contributes ? G : 0.f;
            float g_cr, g_cg, g_cb, g_dep, g_sx, g_sy, g_kxx, g_kxy, g_kyy, g_op;
            {
                // Products feeding sums are fused in this block (as nvcc does for the reference): it only forms
                // gradients, which are stated with a tolerance; alpha, T and the contributor tests above are not in it.
#pragma clang fp contract(fast)
                // One reciprocal serves the two divisions by (1 - alpha) (backward.cu:506,548).  Gradients are sums
                // over atomics whose order is not the reference's anyway; the tolerance of the parity tests covers
                // the 1-ulp difference between x * rcp(y) and x / y.
                const float one_m = 1.f - alpha_m;
                const float inv_1ma = __builtin_amdgcn_rcpf(one_m);
                T = T * inv_1ma;
                const float dchannel_dcolor = alpha_m * T;
                float dL_dalpha = (c.r - acc_r) * dLr;
                dL_dalpha += (c.g - acc_g) * dLg;
                dL_dalpha += (c.b - acc_b) * dLb;
                if (kDepthAlpha) {
                    dL_dalpha += (z - acc_d) * dLd;
                    dL_dalpha += (1.f - acc_a) * dLa;
                }
                g_cr = dchannel_dcolor * dLr;
                g_cg = dchannel_dcolor * dLg;
                g_cb = dchannel_dcolor * dLb;
                g_dep = kDepthAlpha ? dchannel_dcolor * dLd : 0.f;
                acc_r = alpha_m * c.r + one_m * acc_r;
                acc_g = alpha_m * c.g + one_m * acc_g;
                acc_b = alpha_m * c.b + one_m * acc_b;
                if (kDepthAlpha) {
                    acc_d = alpha_m * z + one_m * acc_d;
                    acc_a = alpha_m + one_m * acc_a;
                }
                dL_dalpha *= T;
                dL_dalpha += (-T_final * inv_1ma) * bg_dot;
                const float dL_dG = c.opacity * dL_dalpha;
                // The conic is the same for every pixel of the entry, so the sums over pixels of the mean2D and conic gradients
                // (backward.cu:563-581) factor: with u = dL_dG G dx, v = dL_dG G dy,
                //   dL_dmean2D = -(W/2) (cxx S_u + cxy S_v), -(H/2) (cyy S_v + cxy S_u);  dL_dconic = -1/2 (S_{u dx}, S_{u dy}, S_{v dy}).
                // The five sums are formed here (5 multiplies instead of 17 instructions per lane); the per-Gaussian pass, which
                // recomputes the forward's conic anyway, finishes them.
                const float u = dL_dG * (G_m * dx), v = dL_dG * (G_m * dy);
                g_sx = u;
                g_sy = v;
                g_kxx = u * dx;
                g_kxy = u * dy;
                g_kyy = v * dy;
                g_op = G_m * dL_dalpha;
            }
            // r|g, b|depth, su|sv, kxx|kxy, kyy|opacity -> rows [r b g depth], [su kxx sv kxy], [kyy - opacity -]
            float x0 = row_sum_all_lanes(fold16(fold32(g_cr, g_cg), fold32(g_cb, g_dep)));
            float x1 = row_sum_all_lanes(fold16(fold32(g_sx, g_sy), fold32(g_kxx, g_kxy)));
            asm volatile("" : "+v"(x0), "+v"(x1));
            float x2 = row_sum_all_lanes(fold16(fold32(g_kyy, g_op), 0.f));
            // keep the last row-rotate add out here, where it is one DPP instruction per value (sunk into the branch
            // below it becomes a zero-fill, a DPP move and an add)
            asm volatile("" : "+v"(x2));
            if (my_slot >= 0) {  // ten lanes, one 64-byte line: a single atomic instruction per (quadrant, entry)
                const float v = red_k == 0 ? x0 : red_k == 1 ? x1 : x2;
                if (kDet) {
                    const size_t s = (size_t)range.x + (pos - seg_base);   // position in the sorted list (deterministic calls have ONE segment)
                    det_partial[(4 * s + (size_t)quad) * 10 + (size_t)(my_slot >= 10 ? my_slot - 10 : my_slot)] = v;
                    if (lane == 0) atomicOr(det_bits + (s >> 3), 1u << (4u * (uint32_t)(s & 7) + (uint32_t)quad));   // (lane 0 has a slot)
                } else {
                    atomicAdd(accum + (size_t)kAccumStride * __float_as_uint(rb.w) + my_slot, v);
                }
            }
        }
    }
    }
#ifdef GSR_KERNEL_TRACE
    if (lane == 0 && g_backward_census != nullptr)
        for (int i = 0; i < 13; ++i)
            if (cz[i] != 0ull) atomicAdd(g_backward_census + i, cz[i]);
#endif
}

// ---- GSR_OPT_BACKWARD_DETERMINISTIC: the per-Gaussian sums in a fixed order ----
// sorted_ids: the call's point list sorted (stable) by Gaussian id; a Gaussian's entries are positions [first, end) of it.
__global__ void __launch_bounds__(256) det_segments_kernel(const uint32_t* __restrict__ n_device, const uint32_t* __restrict__ sorted_ids,
                                                          uint32_t* __restrict__ seg_first, uint32_t* __restrict__ seg_end /*both zero on entry*/) {
    const uint32_t n = *n_device;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = sorted_ids[i];
    if (i == 0u || sorted_ids[i - 1u] != g) seg_first[g] = i;
    if (i + 1u == n || sorted_ids[i + 1u] != g) seg_end[g] = i + 1u;
}

// One record set: the (position, quadrant) records of one per-pixel pass and their presence bits.
struct DetRecords { const uint32_t* bits; const float* partial; };
__device__ __forceinline__ void det_add_position(const DetRecords& r, uint32_t s, float acc[10]) {
    const uint32_t b = (r.bits[s >> 3] >> (4u * (s & 7u))) & 15u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (b >> q & 1u) {
            const float* rec = r.partial + (4 * (size_t)s + (size_t)q) * 10;
#pragma unroll
            for (int k = 0; k < 10; ++k) acc[k] += rec[k];
        }
}

// One lane = one Gaussian: its records are added in ascending (list position, quadrant) order -- the list positions of a Gaussian
// ascend with the tile, so the order is (tile, quadrant).  A Gaussian with more than 64 entries (a splat of hundreds of tiles) is
// summed by the whole wave instead: lane l takes its entries l, l + 64, ... in that order and the 64 partial sums meet in a
// fixed butterfly.  Which of the two forms a Gaussian takes depends on its entry count only: same inputs, same bits.
// accum gets ALL sixteen slots of every Gaussian (zeros where nothing contributed): no memset in front.
__global__ void __launch_bounds__(64) det_reduce_kernel(int P, const uint32_t* __restrict__ seg_first, const uint32_t* __restrict__ seg_end,
                                                       const uint32_t* __restrict__ sorted_pos, const uint32_t* __restrict__ bits1,
                                                       const float* __restrict__ partial1, const uint32_t* __restrict__ bits2,
                                                       const float* __restrict__ partial2, float* __restrict__ accum) {
    const int lane = threadIdx.x;
    const int g = blockIdx.x * 64 + lane;
    const DetRecords r1 = {bits1, partial1}, r2 = {bits2, partial2};
    uint32_t first = 0u, end = 0u;
    if (g < P) { first = seg_first[g]; end = seg_end[g]; }
    float a1[10], a2[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) a1[k] = a2[k] = 0.f;
    const bool long_one = end - first > 64u;
    if (!long_one)
        for (uint32_t i = first; i < end; ++i) {
            const uint32_t s = sorted_pos[i];
            det_add_position(r1, s, a1);
            if (partial2 != nullptr) det_add_position(r2, s, a2);
        }
    unsigned long long todo = __ballot(long_one);
    while (todo != 0ull) {
        const int owner = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t f = (uint32_t)__shfl((int)first, owner), e = (uint32_t)__shfl((int)end, owner);
        float b1[10], b2[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) b1[k] = b2[k] = 0.f;
        for (uint32_t i = f + (uint32_t)lane; i < e; i += 64u) {
            const uint32_t s = sorted_pos[i];
            det_add_position(r1, s, b1);
            if (partial2 != nullptr) det_add_position(r2, s, b2);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                b1[k] += __shfl_xor(b1[k], d);
                b2[k] += __shfl_xor(b2[k], d);
            }
            if (lane == owner) { a1[k] = b1[k]; a2[k] = b2[k]; }
        }
    }
    if (g >= P) return;
    float4* line = reinterpret_cast<float4*>(accum + (size_t)kAccumStride * g);
    // slots 0 - 3 colour + depth of the first pass; 4 - 9 geometry of both; 10 - 12 colour of the second; the rest zero
    line[0] = make_float4(a1[0], a1[1], a1[2], a1[3]);
    line[1] = make_float4(a1[4] + a2[4], a1[5] + a2[5], a1[6] + a2[6], a1[7] + a2[7]);
    line[2] = make_float4(a1[8] + a2[8], a1[9] + a2[9], a2[0], a2[1]);
    line[3] = make_float4(a2[2], 0.f, 0.f, 0.f);
}

// auxiliary.h:103-114
__device__ __forceinline__ F3 dnormvdv3(F3 v, F3 dv) {
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    F3 o;
    o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return o;
}

// backward.cu:20-138, split in two: d(colour)/d(sh_k) is a scalar per coefficient shared by the
// three channels (written as one 12-byte store per coefficient), d(colour)/d(dir) is per channel.
__device__ __forceinline__ void sh_coefficient_grads(int deg, float x, float y, float z, float k[16]) {
    k[0] = kSH0;
    if (deg > 0) {
        k[1] = -kSH1 * y;
        k[2] = kSH1 * z;
        k[3] = -kSH1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            k[4] = kSH2_0 * xy;
            k[5] = kSH2_1 * yz;
            k[6] = kSH2_2 * (2.f * zz - xx - yy);
            k[7] = kSH2_3 * xz;
            k[8] = kSH2_4 * (xx - yy);
            if (deg > 2) {
                k[9] = kSH3_0 * y * (3.f * xx - yy);
                k[10] = kSH3_1 * xy * z;
                k[11] = kSH3_2 * y * (4.f * zz - xx - yy);
                k[12] = kSH3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                k[13] = kSH3_4 * x * (4.f * zz - xx - yy);
                k[14] = kSH3_5 * z * (xx - yy);
                k[15] = kSH3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

__device__ __forceinline__ F3 sh_dir_grads_channel(int deg, float x, float y, float z, const float* __restrict__ sh, int c) {
#define SHC(k) sh[3 * (k) + c]
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (deg > 0) {
        dx = -kSH1 * SHC(3);
        dy = -kSH1 * SHC(1);
        dz = kSH1 * SHC(2);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dx += kSH2_0 * y * SHC(4) + kSH2_2 * 2.f * -x * SHC(6) + kSH2_3 * z * SHC(7) + kSH2_4 * 2.f * x * SHC(8);
            dy += kSH2_0 * x * SHC(4) + kSH2_1 * z * SHC(5) + kSH2_2 * 2.f * -y * SHC(6) + kSH2_4 * 2.f * -y * SHC(8);
            dz += kSH2_1 * y * SHC(5) + kSH2_2 * 2.f * 2.f * z * SHC(6) + kSH2_3 * x * SHC(7);
            if (deg > 2) {
                dx += (kSH3_0 * SHC(9) * 3.f * 2.f * xy + kSH3_1 * SHC(10) * yz + kSH3_2 * SHC(11) * -2.f * xy +
                       kSH3_3 * SHC(12) * -3.f * 2.f * xz + kSH3_4 * SHC(13) * (-3.f * xx + 4.f * zz - yy) +
                       kSH3_5 * SHC(14) * 2.f * xz + kSH3_6 * SHC(15) * 3.f * (xx - yy));
                dy += (kSH3_0 * SHC(9) * 3.f * (xx - yy) + kSH3_1 * SHC(10) * xz +
                       kSH3_2 * SHC(11) * (-3.f * yy + 4.f * zz - xx) + kSH3_3 * SHC(12) * -3.f * 2.f * yz +
                       kSH3_4 * SHC(13) * -2.f * xy + kSH3_5 * SHC(14) * -2.f * yz + kSH3_6 * SHC(15) * -3.f * 2.f * xy);
                dz += (kSH3_1 * SHC(10) * xy + kSH3_2 * SHC(11) * 4.f * 2.f * yz +
                       kSH3_3 * SHC(12) * 3.f * (2.f * zz - xx - yy) + kSH3_4 * SHC(13) * 4.f * 2.f * xz +
                       kSH3_5 * SHC(14) * (xx - yy));
            }
        }
    }
#undef SHC
    return F3{dx, dy, dz};
}

struct BackwardArgs {
    int P, sh_degree, M;
    const float* means3D;
    const int* radii;
    const float* shs;            // nullable
    const float* scales;         // nullable
    const float* rotations;      // nullable
    const float* cov3D_precomp;  // nullable
    float scale_modifier;
    const float* accum;          // [P,16] sums of render_backward_kernel (slots: see kAccumStride)
    float* dL_dmean2D;           // [P,3]  written here from accum
    float* dL_dconic;            // [P,4]
    float* dL_dopacity;          // [P]
    float* dL_dcolor;            // [P,3]
    float* dL_ddepth;            // [P]
    float* dL_dmean3D;           // [P,3]
    float* dL_dcov3D;            // [P,6]
    float* dL_dsh;               // [P,M,3] nullable
    float* dL_dscale;            // [P,3]
    float* dL_drot;              // [P,4]
    // gsr_backward_raw: scales / rotations / shs are the model's RAW tensors (log scales, unnormalised quaternions, _features_dc)
    // and the gradients leave with the activations' chain rule applied: dL_dscale -> d/d(log scale), dL_drot -> d/d(raw
    // quaternion), dL_dopacity -> d/d(logit), dL_dsh -> d/d(_features_dc) [P,1,3], dL_dsh_rest -> d/d(_features_rest) [P,M-1,3]
    int raw;
    const float* shs_rest;       // _features_rest [P,M-1,3] (raw, M > 1)
    const float* opacity_logits; // _opacity [P] (raw)
    float* dL_dsh_rest;          // [P,M-1,3] (raw, M > 1)
    int normal_grads;            // raw: accum slots 10 - 12 hold dL/d(view normal colour): chained through get_normal to the quaternion
    const float* cam_pos_normals;
};

// d/d(x) of x / max(||x||, 1e-12) against an upstream gradient g (what autograd makes of F.normalize: the clamp passes no
// gradient): (g - n (n . g)) / ||x|| with n the normalised vector; g / 1e-12 below the clamp.
__device__ __forceinline__ F4 normalize4_backward(F4 x, F4 n, F4 g) {
    const float len = sqrtf((x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w));
    if (!(len > 1e-12f)) return F4{g.x / 1e-12f, g.y / 1e-12f, g.z / 1e-12f, g.w / 1e-12f};
    const float d = n.x * g.x + n.y * g.y + n.z * g.z + n.w * g.w;
    return F4{(g.x - n.x * d) / len, (g.y - n.y * d) / len, (g.z - n.z * d) / len, (g.w - n.w * d) / len};
}

// Backward of gsr_device.h: view_normal_rgb(p, cam, min_axis(s, q)) with respect to the (normalised) quaternion q: through
// * 0.5 + 0.5, the unit normalisation, the flip (a constant sign), the selected column of build_rotation(q) and that
// function's own renormalisation of q (general_utils.py:78-101,135-157; gaussian_model.py get_normal).  The position and the
// scales only enter through a sign and an argmin: no gradient, as in autograd.
__device__ __forceinline__ F4 view_normal_backward(F3 p, F3 cam, F3 s, F4 q, F3 g_rgb) {
    const float n2 = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float w = q.x / n2, x = q.y / n2, y = q.z / n2, z = q.w / n2;
    int c;
    if (s.x < s.y) c = s.z < s.x ? 2 : 0;
    else if (s.y < s.x) c = s.z < s.y ? 2 : 1;
    else c = s.x < s.z ? 1 : 2;
    F3 a, dw, dx, dy, dz;   // the column and its derivatives with respect to w, x, y, z
    if (c == 0) {
        a = F3{1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y)};
        dw = F3{0.f, 2.f * z, -2.f * y}; dx = F3{0.f, 2.f * y, 2.f * z}; dy = F3{-4.f * y, 2.f * x, -2.f * w}; dz = F3{-4.f * z, 2.f * w, 2.f * x};
    } else if (c == 1) {
        a = F3{2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x)};
        dw = F3{-2.f * z, 0.f, 2.f * x}; dx = F3{2.f * y, -4.f * x, 2.f * w}; dy = F3{2.f * x, 0.f, 2.f * z}; dz = F3{-2.f * w, -4.f * z, 2.f * y};
    } else {
        a = F3{2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y)};
        dw = F3{2.f * y, -2.f * x, 0.f}; dx = F3{2.f * z, -2.f * w, -4.f * x}; dy = F3{2.f * w, 2.f * z, -4.f * y}; dz = F3{2.f * x, 2.f * y, 0.f};
    }
    const F3 d = {p.x - cam.x, p.y - cam.y, p.z - cam.z};
    const float len = torch_norm3(d);
    const float dot = torch_sum3(a.x * -(d.x / len), a.y * -(d.y / len), a.z * -(d.z / len));
    const float sg = dot >= 0.f ? 1.f : -1.f;
    const F3 m = {a.x * sg, a.y * sg, a.z * sg};
    const float ml = torch_norm3(m);
    const F3 n = {m.x / ml, m.y / ml, m.z / ml};
    const F3 gn = {0.5f * g_rgb.x, 0.5f * g_rgb.y, 0.5f * g_rgb.z};
    const float nd = n.x * gn.x + n.y * gn.y + n.z * gn.z;
    const F3 ga = {sg * (gn.x - n.x * nd) / ml, sg * (gn.y - n.y * nd) / ml, sg * (gn.z - n.z * nd) / ml};   // d/d(axis)
    const F4 gq = {ga.x * dw.x + ga.y * dw.y + ga.z * dw.z, ga.x * dx.x + ga.y * dx.y + ga.z * dx.z,
                   ga.x * dy.x + ga.y * dy.y + ga.z * dy.z, ga.x * dz.x + ga.y * dz.y + ga.z * dz.z};        // d/d(w, x, y, z)
    const float qd = w * gq.x + x * gq.y + y * gq.z + z * gq.w;
    return F4{(gq.x - w * qd) / n2, (gq.y - x * qd) / n2, (gq.z - y * qd) / n2, (gq.w - z * qd) / n2};    // through q / ||q||
}

// The LDS stage of dL_dsh: one slot of 48 floats per Gaussian of the workgroup THAT HAS A GRADIENT, handed out in Gaussian order,
// kShStageSlots per workgroup of 256 Gaussians (51 contribute on average at C3).  Pitch 49: the lanes of a wave write float f of 64
// different slots to 64 different banks.  A workgroup with more contributing Gaussians than slots (dense training views) stores the
// records of the surplus ones straight to HBM.
constexpr int kShStagePitch = 49;
constexpr int kShStageSlots = 96;

// One lane = one Gaussian, in two phases (round 6).  Phase 1, every Gaussian on its own lane: is it idle -- not rendered, or rendered
// without a contribution?  Then all its gradients are zeros, written here (dL_dsh excepted when the workgroup stages it: `stage` non-null
// says so, and the copy-out writes those zeros without reading anything from LDS) and 0 is returned; otherwise 3: "has a gradient",
// nothing computed yet.  Phase 2, the Gaussians WITH a gradient packed onto the first lanes of the workgroup (four of five have none
// at C3, scattered over the waves: run in place, nearly every wave went through the whole per-Gaussian chain at 20 % lane
// efficiency): the chain itself; `stage` is this Gaussian's slot of the LDS stage for dL_dsh, or null (store it straight to HBM).
// Returns where the dL_dsh record went: 1 its slot, 2 HBM (0: there is none -- colours were given, not SH).
template <bool kRaw, int kPhase>
__device__ __forceinline__ int preprocess_backward_lane(const BackwardArgs& g, const Camera& cam, int idx, float* stage) {
    // the sums of this Gaussian, one 64-byte line (ten slots; 10 - 12: a second feature set's colour sums; the rest stay zero)
    const float4* line = reinterpret_cast<const float4*>(g.accum + (size_t)kAccumStride * idx);
    bool idle = kPhase == 2 ? false : !(g.radii[idx] > 0);
    const bool staged = stage != nullptr;      // phase 1: the workgroup's dL_dsh records leave through the LDS stage
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2q = s0, s3q = s0;
    if (kPhase == 2) {
        s0 = line[0]; s1 = line[1]; s2q = line[2]; s3q = line[3];     // (read by phase 1 a moment ago: an L1 / L2 hit)
    } else if (!idle) {
        s0 = line[0]; s1 = line[1]; s2q = line[2]; s3q = line[3];
        // Rendered, but no pixel composited it (behind an opaque front, or alpha < 1/255 everywhere: 80 % of C3's rendered
        // Gaussians, 99 % of the trained-scene stand-in's): every sum is an exact zero and every gradient below is a product
        // with one of them, so the 250 bytes of parameters and coefficients need not be read to write zeros.  (A NaN sum
        // compares unequal and takes the full path; what differs from the full path is 0 x Inf for a parameter that
        // overflowed, which would have stored NaN.)
        idle = s0.x == 0.f && s0.y == 0.f && s0.z == 0.f && s0.w == 0.f && s1.x == 0.f && s1.y == 0.f && s1.z == 0.f && s1.w == 0.f &&
               s2q.x == 0.f && s2q.y == 0.f && s2q.z == 0.f && s2q.w == 0.f && s3q.x == 0.f && s3q.y == 0.f && s3q.z == 0.f && s3q.w == 0.f;
    }
    if (idle) {
        // Not rendered, or rendered without a contribution: every gradient of this Gaussian is zero.  The kernel defines ALL
        // output elements, so the caller does not have to zero-fill a gigabyte of gradient tensors first (dL_dsh alone is
        // 576 MB at 3 M).
        const F3 z3 = {0.f, 0.f, 0.f};
        *reinterpret_cast<F3*>(g.dL_dmean2D + 3 * (size_t)idx) = z3;
        if (g.dL_dconic != nullptr) *reinterpret_cast<F4*>(g.dL_dconic + 4 * (size_t)idx) = F4{0.f, 0.f, 0.f, 0.f};
        g.dL_dopacity[idx] = 0.f;
        if (g.dL_dcolor != nullptr) *reinterpret_cast<F3*>(g.dL_dcolor + 3 * (size_t)idx) = z3;
        if (g.dL_ddepth != nullptr) g.dL_ddepth[idx] = 0.f;
        *reinterpret_cast<F3*>(g.dL_dmean3D + 3 * (size_t)idx) = z3;
        if (g.dL_dcov3D != nullptr) {
            *reinterpret_cast<F3*>(g.dL_dcov3D + 6 * (size_t)idx) = z3;
            *reinterpret_cast<F3*>(g.dL_dcov3D + 6 * (size_t)idx + 3) = z3;
        }
        if (staged) {
            // (nothing: see the return value.  Four Gaussians in five take this path at C3; their 48 LDS writes each and the 48 reads that
            // fetched the zeros back were most of this kernel's LDS traffic)
        } else if (g.dL_dsh != nullptr) {
            if (kRaw) {
                *reinterpret_cast<F3*>(g.dL_dsh + 3 * (size_t)idx) = z3;
                for (int k = 1; k < g.M; ++k) *reinterpret_cast<F3*>(g.dL_dsh_rest + 3 * ((size_t)(g.M - 1) * idx + k - 1)) = z3;
            } else {
                for (int k = 0; k < g.M; ++k) *reinterpret_cast<F3*>(g.dL_dsh + 3 * ((size_t)g.M * idx + k)) = z3;
            }
        }
        *reinterpret_cast<F3*>(g.dL_dscale + 3 * (size_t)idx) = z3;
        *reinterpret_cast<F4*>(g.dL_drot + 4 * (size_t)idx) = F4{0.f, 0.f, 0.f, 0.f};
        return 0;
    }
    if (kPhase == 1) return 3;
    const float* __restrict__ view = cam.viewmatrix;
    const float* __restrict__ proj = cam.projmatrix;
    const F3 mean = ld3(g.means3D + 3 * (size_t)idx);

    // the ten sums, spread into the reference's gradient arrays
    const float2 s2 = make_float2(s2q.x, s2q.y);
    const float dLc_r = s0.x, dLc_g = s0.y, dLc_b = s0.z, gdep = s0.w;
    const float S_u = s1.x, S_v = s1.y;
    const float dLcx = -0.5f * s1.z, dLcy = -0.5f * s1.w, dLcz = -0.5f * s2.x;   // backward.cu:577-581, the -1/2 taken out of the sums
    // (intermediates the caller does not want -- dL_dconic / dL_ddepth always, dL_dcolor with SH colours, dL_dcov3D with scales and
    // rotations -- may be NULL: 56 bytes per Gaussian less to write)
    if (g.dL_dcolor != nullptr) *reinterpret_cast<F3*>(g.dL_dcolor + 3 * (size_t)idx) = F3{dLc_r, dLc_g, dLc_b};
    if (g.dL_ddepth != nullptr) g.dL_ddepth[idx] = gdep;
    if (g.dL_dconic != nullptr) *reinterpret_cast<F4*>(g.dL_dconic + 4 * (size_t)idx) = F4{dLcx, dLcy, 0.f, dLcz};   // 2x2 with one unused slot
    if (kRaw) {   // through sigmoid (gaussian_model.py:125-126): dL/d(logit) = dL/d(opacity) o (1 - o)
        const float o = torch_sigmoid(g.opacity_logits[idx]);
        g.dL_dopacity[idx] = s2.y * ((1.f - o) * o);
    } else {
        g.dL_dopacity[idx] = s2.y;
    }

    // 3D covariance as the forward computed it
    float c3[6];
    Mat3 R = {}, S = {};
    float sx = 0.f, sy = 0.f, sz = 0.f, qr = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
    F4 q_raw = {0.f, 0.f, 0.f, 0.f};
    F3 s_act = {0.f, 0.f, 0.f};
    if (g.cov3D_precomp != nullptr) {
        const float* c = g.cov3D_precomp + 6 * (size_t)idx;
#pragma unroll
        for (int k = 0; k < 6; ++k) c3[k] = c[k];
    } else {
        F3 s = ld3(g.scales + 3 * (size_t)idx);
        F4 q = *reinterpret_cast<const F4*>(g.rotations + 4 * (size_t)idx);
        if (kRaw) {   // the forward's activations (gsr_device.h: torch_*)
            q_raw = q;
            s = F3{expf(s.x), expf(s.y), expf(s.z)};
            q = torch_normalize4(q);
            s_act = s;
        }
        qr = q.x; qx = q.y; qy = q.z; qz = q.w;
        sx = g.scale_modifier * s.x; sy = g.scale_modifier * s.y; sz = g.scale_modifier * s.z;
        S.m[0][0] = sx; S.m[1][1] = sy; S.m[2][2] = sz;
        const float r = qr, x = qx, y = qy, z = qz;
        R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[1][0] = 2.f * (x * y - r * z);       R.m[2][0] = 2.f * (x * z + r * y);
        R.m[0][1] = 2.f * (x * y + r * z);       R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[2][1] = 2.f * (y * z - r * x);
        R.m[0][2] = 2.f * (x * z - r * y);       R.m[1][2] = 2.f * (y * z + r * x);       R.m[2][2] = 1.f - 2.f * (x * x + y * y);
        const Mat3 Mm = mul3(S, R);
        const Mat3 Sg = mul3(transpose3(Mm), Mm);
        c3[0] = Sg.m[0][0]; c3[1] = Sg.m[1][0]; c3[2] = Sg.m[2][0];
        c3[3] = Sg.m[1][1]; c3[4] = Sg.m[2][1]; c3[5] = Sg.m[2][2];
    }

    // ---- computeCov2DCUDA (backward.cu:144-276) ----
    float tx = view[0] * mean.x + view[4] * mean.y + view[8] * mean.z + view[12];
    float ty = view[1] * mean.x + view[5] * mean.y + view[9] * mean.z + view[13];
    const float tz_ = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14];
    const float h_x = cam.focal_x, h_y = cam.focal_y;
    const float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
    const float txtz = tx / tz_, tytz = ty / tz_;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz_;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz_;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    Mat3 J = {{{h_x / tz_, 0.f, 0.f}, {0.f, h_y / tz_, 0.f}, {-(h_x * tx) / (tz_ * tz_), -(h_y * ty) / (tz_ * tz_), 0.f}}};
    Mat3 Wm;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Wm.m[a][b] = view[4 * a + b];
    const Mat3 T = mul3(Wm, J);
    Mat3 V = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
    const Mat3 cov = mul3(mul3(transpose3(T), transpose3(V)), T);
#define TG(c, r) T.m[r][c]
#define VG(c, r) V.m[r][c]
#define WG(c, r) Wm.m[r][c]
    const float a = cov.m[0][0] + 0.3f, b = cov.m[1][0], c_ = cov.m[1][1] + 0.3f;
    const float denom = a * c_ - b * b;
    // dL_dmean2D from the factored sums and the forward's conic (forward.cu:219-223, recomputed with its arithmetic):
    // dG/d(delx) = -G dx cxx - G dy cxy, d(delx)/dx = W / 2 (backward.cu:563-571)
    float g2x, g2y;
    {
        const float det_inv = 1.f / denom;
        const float cxx = c_ * det_inv, cxy = -b * det_inv, cyy = a * det_inv;
        g2x = -(float)(0.5 * cam.width) * (cxx * S_u + cxy * S_v);
        g2y = -(float)(0.5 * cam.height) * (cyy * S_v + cxy * S_u);
    }
    *reinterpret_cast<F3*>(g.dL_dmean2D + 3 * (size_t)idx) = F3{g2x, g2y, 0.f};            // z is never used (backward.cu)
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcov[6];
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c_ * c_ * dLcx + 2 * b * c_ * dLcy + (denom - a * c_) * dLcz);
        dL_dc = denom2inv * (-a * a * dLcz + 2 * a * b * dLcy + (denom - a * c_) * dLcx);
        dL_db = denom2inv * 2 * (b * c_ * dLcx - (denom + 2 * b * b) * dLcy + a * b * dLcz);
        dcov[0] = (TG(0, 0) * TG(0, 0) * dL_da + TG(0, 0) * TG(1, 0) * dL_db + TG(1, 0) * TG(1, 0) * dL_dc);
        dcov[3] = (TG(0, 1) * TG(0, 1) * dL_da + TG(0, 1) * TG(1, 1) * dL_db + TG(1, 1) * TG(1, 1) * dL_dc);
        dcov[5] = (TG(0, 2) * TG(0, 2) * dL_da + TG(0, 2) * TG(1, 2) * dL_db + TG(1, 2) * TG(1, 2) * dL_dc);
        dcov[1] = 2 * TG(0, 0) * TG(0, 1) * dL_da + (TG(0, 0) * TG(1, 1) + TG(0, 1) * TG(1, 0)) * dL_db + 2 * TG(1, 0) * TG(1, 1) * dL_dc;
        dcov[2] = 2 * TG(0, 0) * TG(0, 2) * dL_da + (TG(0, 0) * TG(1, 2) + TG(0, 2) * TG(1, 0)) * dL_db + 2 * TG(1, 0) * TG(1, 2) * dL_dc;
        dcov[4] = 2 * TG(0, 2) * TG(0, 1) * dL_da + (TG(0, 1) * TG(1, 2) + TG(0, 2) * TG(1, 1)) * dL_db + 2 * TG(1, 1) * TG(1, 2) * dL_dc;
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) dcov[i] = 0;
    }
    if (g.dL_dcov3D != nullptr) {
#pragma unroll
        for (int i = 0; i < 6; ++i) g.dL_dcov3D[6 * (size_t)idx + i] = dcov[i];
    }
    const float dL_dT00 = 2 * (TG(0, 0) * VG(0, 0) + TG(0, 1) * VG(0, 1) + TG(0, 2) * VG(0, 2)) * dL_da + (TG(1, 0) * VG(0, 0) + TG(1, 1) * VG(0, 1) + TG(1, 2) * VG(0, 2)) * dL_db;
    const float dL_dT01 = 2 * (TG(0, 0) * VG(1, 0) + TG(0, 1) * VG(1, 1) + TG(0, 2) * VG(1, 2)) * dL_da + (TG(1, 0) * VG(1, 0) + TG(1, 1) * VG(1, 1) + TG(1, 2) * VG(1, 2)) * dL_db;
    const float dL_dT02 = 2 * (TG(0, 0) * VG(2, 0) + TG(0, 1) * VG(2, 1) + TG(0, 2) * VG(2, 2)) * dL_da + (TG(1, 0) * VG(2, 0) + TG(1, 1) * VG(2, 1) + TG(1, 2) * VG(2, 2)) * dL_db;
    const float dL_dT10 = 2 * (TG(1, 0) * VG(0, 0) + TG(1, 1) * VG(0, 1) + TG(1, 2) * VG(0, 2)) * dL_dc + (TG(0, 0) * VG(0, 0) + TG(0, 1) * VG(0, 1) + TG(0, 2) * VG(0, 2)) * dL_db;
    const float dL_dT11 = 2 * (TG(1, 0) * VG(1, 0) + TG(1, 1) * VG(1, 1) + TG(1, 2) * VG(1, 2)) * dL_dc + (TG(0, 0) * VG(1, 0) + TG(0, 1) * VG(1, 1) + TG(0, 2) * VG(1, 2)) * dL_db;
    const float dL_dT12 = 2 * (TG(1, 0) * VG(2, 0) + TG(1, 1) * VG(2, 1) + TG(1, 2) * VG(2, 2)) * dL_dc + (TG(0, 0) * VG(2, 0) + TG(0, 1) * VG(2, 1) + TG(0, 2) * VG(2, 2)) * dL_db;
    const float dL_dJ00 = WG(0, 0) * dL_dT00 + WG(0, 1) * dL_dT01 + WG(0, 2) * dL_dT02;
    const float dL_dJ02 = WG(2, 0) * dL_dT00 + WG(2, 1) * dL_dT01 + WG(2, 2) * dL_dT02;
    const float dL_dJ11 = WG(1, 0) * dL_dT10 + WG(1, 1) * dL_dT11 + WG(1, 2) * dL_dT12;
    const float dL_dJ12 = WG(2, 0) * dL_dT10 + WG(2, 1) * dL_dT11 + WG(2, 2) * dL_dT12;
#undef TG
#undef VG
#undef WG
    const float tzi = 1.f / tz_, tz2 = tzi * tzi, tz3 = tz2 * tzi;
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * tx) * tz3 * dL_dJ02 + (2 * h_y * ty) * tz3 * dL_dJ12;
    float dmx = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;   // transformVec4x3Transpose
    float dmy = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
    float dmz = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;

    // ---- preprocessCUDA backward (backward.cu:346-413) ----
    const float mhw = proj[3] * mean.x + proj[7] * mean.y + proj[11] * mean.z + proj[15];
    const float m_w = 1.0f / (mhw + 0.0000001f);
    const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
    dmx += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    dmy += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    dmz += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    const float mul3_ = view[2] * mean.x + view[6] * mean.y + view[10] * mean.z + view[14];
    dmx += (view[2] - view[3] * mul3_) * gdep;
    dmy += (view[6] - view[7] * mul3_) * gdep;
    dmz += (view[10] - view[11] * mul3_) * gdep;

    if (g.shs != nullptr) {
        int deg = g.sh_degree < 3 ? g.sh_degree : 3;  // bands above 3 do not exist here (backward.cu:20-138): M = 25 gets 9 zero bands
        if (deg > 2 && g.M < 16) deg = 2;
        if (deg > 1 && g.M < 9) deg = 1;
        if (deg > 0 && g.M < 4) deg = 0;
        const float* sh = kRaw ? g.shs + 3 * (size_t)idx : g.shs + 3 * (size_t)g.M * idx;   // coefficient 0
        const float* shr = (kRaw && g.M > 1) ? g.shs_rest + 3 * (size_t)(g.M - 1) * idx - 3 : sh;   // coefficient k >= 1 at shr + 3 k
        float* dsh = kRaw ? g.dL_dsh + 3 * (size_t)idx : g.dL_dsh + 3 * (size_t)g.M * idx;
        float* dshr = (kRaw && g.M > 1) ? g.dL_dsh_rest + 3 * (size_t)(g.M - 1) * idx - 3 : dsh;
        const F3 cp = ld3(cam.cam_pos);
        const F3 o = F3{mean.x - cp.x, mean.y - cp.y, mean.z - cp.z};
        const float len = sqrtf(o.x * o.x + o.y * o.y + o.z * o.z);
        const float x = o.x / len, y = o.y / len, z = o.z / len;
        // the forward's clamp decision, recomputed with the forward's own arithmetic
        F3 pre = sh_unclamped(deg, x, y, z, sh, shr);
        const float dL0 = dLc_r * (pre.x < 0 ? 0.f : 1.f), dL1 = dLc_g * (pre.y < 0 ? 0.f : 1.f),
                    dL2 = dLc_b * (pre.z < 0 ? 0.f : 1.f);
        float kk[16];
        sh_coefficient_grads(deg, x, y, z, kk);
        const int ncoef = (deg + 1) * (deg + 1);
        if (stage != nullptr) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {  // staged records are M = 16 wide; bands above the degree are zero
                const float m = k < ncoef ? kk[k] : 0.f;
                stage[3 * k + 0] = m * dL0;
                stage[3 * k + 1] = m * dL1;
                stage[3 * k + 2] = m * dL2;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < ncoef) *reinterpret_cast<F3*>((k == 0 ? dsh : dshr) + 3 * k) = F3{kk[k] * dL0, kk[k] * dL1, kk[k] * dL2};
            for (int k = ncoef; k < g.M; ++k) *reinterpret_cast<F3*>((k == 0 ? dsh : dshr) + 3 * k) = F3{0.f, 0.f, 0.f};  // bands above the degree
        }
        const F3 d0 = sh_dir_grads_channel(deg, x, y, z, shr, 0);   // (reads coefficients k >= 1 only)
        const F3 d1 = sh_dir_grads_channel(deg, x, y, z, shr, 1);
        const F3 d2 = sh_dir_grads_channel(deg, x, y, z, shr, 2);
        const F3 ddir = F3{d0.x * dL0 + d1.x * dL1 + d2.x * dL2, d0.y * dL0 + d1.y * dL1 + d2.y * dL2,
                           d0.z * dL0 + d1.z * dL1 + d2.z * dL2};
        const F3 dm = dnormvdv3(o, ddir);
        dmx += dm.x; dmy += dm.y; dmz += dm.z;
    }
    g.dL_dmean3D[3 * (size_t)idx + 0] = dmx;
    g.dL_dmean3D[3 * (size_t)idx + 1] = dmy;
    g.dL_dmean3D[3 * (size_t)idx + 2] = dmz;

    if (g.scales != nullptr) {
        // computeCov3D backward (backward.cu:278-342)
        const Mat3 Mm = mul3(S, R);
        Mat3 dSig = {{{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                      {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}}};
        Mat3 M2;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M2.m[i][j] = 2.0f * Mm.m[i][j];
        const Mat3 dM = mul3(M2, dSig);
        const Mat3 Rt = transpose3(R);
        Mat3 dMt = transpose3(dM);
#define COL(Mx, c, r) Mx.m[r][c]
        const float sv[3] = {sx, sy, sz};
        const float act[3] = {s_act.x, s_act.y, s_act.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ds = COL(Rt, c, 0) * COL(dMt, c, 0) + COL(Rt, c, 1) * COL(dMt, c, 1) + COL(Rt, c, 2) * COL(dMt, c, 2);
            g.dL_dscale[3 * (size_t)idx + c] = kRaw ? ds * act[c] : ds;   // raw: through exp (gaussian_model.py:96-97)
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) COL(dMt, c, rr) *= sv[c];
#define D(c, rr) COL(dMt, c, rr)
        const float r = qr, x = qx, y = qy, z = qz;
        float dq[4];
        dq[0] = 2 * z * (D(0, 1) - D(1, 0)) + 2 * y * (D(2, 0) - D(0, 2)) + 2 * x * (D(1, 2) - D(2, 1));
        dq[1] = 2 * y * (D(1, 0) + D(0, 1)) + 2 * z * (D(2, 0) + D(0, 2)) + 2 * r * (D(1, 2) - D(2, 1)) - 4 * x * (D(2, 2) + D(1, 1));
        dq[2] = 2 * x * (D(1, 0) + D(0, 1)) + 2 * r * (D(2, 0) - D(0, 2)) + 2 * z * (D(1, 2) + D(2, 1)) - 4 * y * (D(2, 2) + D(0, 0));
        dq[3] = 2 * r * (D(0, 1) - D(1, 0)) + 2 * x * (D(2, 0) + D(0, 2)) + 2 * y * (D(1, 2) + D(2, 1)) - 4 * z * (D(1, 1) + D(0, 0));
        F4 gq = {dq[0], dq[1], dq[2], dq[3]};
        if (kRaw) {
            const F4 qn = {r, x, y, z};
            if (g.normal_grads) {   // the normal map's share (accum slots 10 - 12 = dL/d(view normal colour))
                const F4 gn = view_normal_backward(mean, ld3(g.cam_pos_normals), s_act, qn, F3{s2q.z, s2q.w, s3q.x});   // slots 10 - 12
                gq = F4{gq.x + gn.x, gq.y + gn.y, gq.z + gn.z, gq.w + gn.w};
            }
            gq = normalize4_backward(q_raw, qn, gq);   // through F.normalize (gaussian_model.py:100-101)
        }
        *reinterpret_cast<F4*>(g.dL_drot + 4 * (size_t)idx) = gq;
#undef D
#undef COL
    } else {
        *reinterpret_cast<F3*>(g.dL_dscale + 3 * (size_t)idx) = F3{0.f, 0.f, 0.f};
        *reinterpret_cast<F4*>(g.dL_drot + 4 * (size_t)idx) = F4{0.f, 0.f, 0.f, 0.f};
    }
    return g.shs == nullptr ? 0 : stage != nullptr ? 1 : 2;
}

// dL_dsh is 192 bytes per Gaussian: written lane by lane it goes out as 12-byte pieces 192 bytes apart (measured
// 2.8 TB/s, scripts/ubench/sh_store.hip); staged through LDS ([float][lane], pitch 65: conflict-free both ways) the
// wave writes its 12 KB as contiguous 16-byte stores (5.9 TB/s).  Taken when M == 16 and the tensor is 16-byte aligned.
template <bool kRaw>
__global__ void __launch_bounds__(256) preprocess_backward_kernel(BackwardArgs g, Camera cam) {
    __shared__ float s_stage[kShStageSlots * kShStagePitch];
    __shared__ uint16_t s_list[256];     // the workgroup's Gaussians with a gradient, in order
    __shared__ uint16_t s_rank[256];     // per Gaussian: its place in that list, 0xFFFF: no gradient
    __shared__ uint32_t s_count[4];
    const int base = blockIdx.x * 256, tid = threadIdx.x;
    const int idx = base + tid;
    const int wave = tid >> 6, lane = tid & 63;
    // (raw: the record leaves as _features_dc's 12 bytes and _features_rest's 180: both dense arrays, 16-byte aligned starts)
    const bool staged = g.dL_dsh != nullptr && g.shs != nullptr && g.M == 16 &&
                        (reinterpret_cast<uintptr_t>(kRaw ? g.dL_dsh_rest : g.dL_dsh) & 15u) == 0;  // uniform
    // phase 1: who has a gradient?  (the others' zeros are written here)
    const bool has = idx < g.P && preprocess_backward_lane<kRaw, 1>(g, cam, idx, staged ? s_stage : nullptr) == 3;
    const unsigned long long mask = __ballot(has);
    if (lane == 0) s_count[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = 0u;
    for (int w = 0; w < wave; ++w) before += s_count[w];
    const uint32_t n_live = s_count[0] + s_count[1] + s_count[2] + s_count[3];
    const uint32_t rank = before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    s_rank[tid] = has ? (uint16_t)rank : (uint16_t)0xFFFFu;
    if (has) s_list[rank] = (uint16_t)tid;
    __syncthreads();
    // phase 2: the per-Gaussian chain, on the first n_live lanes (usually the first wave alone)
    if ((uint32_t)tid < n_live) {
        float* slot = (staged && tid < kShStageSlots) ? s_stage + tid * kShStagePitch : nullptr;
        (void)preprocess_backward_lane<kRaw, 2>(g, cam, base + (int)s_list[tid], slot);
    }
    if (!staged) return;
    __syncthreads();
    // phase 3: dL_dsh of the workgroup's 256 Gaussians as whole 16-byte chunks: a staged record from its slot, zeros for a Gaussian
    // without a gradient (no trip through LDS), nothing where the record was stored directly (no slot left)
    const int count = min(256, g.P - base);
    if (!kRaw) {
        float4* dst = reinterpret_cast<float4*>(g.dL_dsh + 48 * (size_t)base);
        const int chunks = count * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int c = k * 256 + tid;  // 16-byte chunk of the workgroup's 48 KB
            if (c < chunks) {
                const int gi = c / 12, f = (c - gi * 12) * 4;
                const uint32_t r = s_rank[gi];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r != 0xFFFFu) {
                    if (r >= (uint32_t)kShStageSlots) continue;
                    const float* rec = s_stage + r * kShStagePitch + f;
                    v = make_float4(rec[0], rec[1], rec[2], rec[3]);
                }
                dst[c] = v;
            }
        }
    } else {
        // coefficient 0 -> _features_dc's gradient: 12 bytes per Gaussian, neighbouring lanes neighbouring addresses
        if (tid < count) {
            const uint32_t r = s_rank[tid];
            if (r == 0xFFFFu) *reinterpret_cast<F3*>(g.dL_dsh + 3 * (size_t)(base + tid)) = F3{0.f, 0.f, 0.f};
            else if (r < (uint32_t)kShStageSlots) {
                const float* rec = s_stage + r * kShStagePitch;
                *reinterpret_cast<F3*>(g.dL_dsh + 3 * (size_t)(base + tid)) = F3{rec[0], rec[1], rec[2]};
            }
        }
        // coefficients 1..15 -> _features_rest's: the workgroup's 256 x 45 floats are one contiguous run (a multiple of 16 bytes),
        // written as 16-byte chunks; float F of the run belongs to Gaussian F / 45, record float 3 + F % 45
        float4* dst = reinterpret_cast<float4*>(g.dL_dsh_rest + 45 * (size_t)base);
        const int floats = count * 45;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int c = k * 256 + tid;
            const int F = 4 * c;
            if (F >= floats) continue;
            float v[4];
            bool whole = F + 3 < floats;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int gi = min((F + t) / 45, 255), rr = (F + t) - 45 * ((F + t) / 45);
                const uint32_t r = s_rank[gi];
                v[t] = (r != 0xFFFFu && r < (uint32_t)kShStageSlots) ? s_stage[r * kShStagePitch + 3 + rr] : 0.f;
                if (r != 0xFFFFu && r >= (uint32_t)kShStageSlots) whole = false;
            }
            if (whole) {
                dst[c] = make_float4(v[0], v[1], v[2], v[3]);
            } else {   // the last chunk of a ragged last workgroup, or a chunk that touches a Gaussian that stored its own record
                for (int t = 0; t < 4 && F + t < floats; ++t) {
                    const uint32_t r = s_rank[(F + t) / 45];
                    if (r == 0xFFFFu || r < (uint32_t)kShStageSlots) g.dL_dsh_rest[45 * (size_t)base + F + t] = v[t];
                }
            }
        }
    }
}

} // namespace

hipError_t launch_render_backward(const Camera& cam, const BlendSegments& segs, int num_segs,
                                  const float* background, const SplatRaster* raster, const float* colors,
                                  const float* accum_alphas,
                                  const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dpix_depth,
                                  const float* dL_dpix_alpha, float* accum, hipStream_t stream, int colour_slot,
                                  float* det_partial, uint32_t* det_bits) {
    const int T = cam.grid_x * cam.grid_y;
    const bool full = dL_dpix_depth != nullptr && dL_dpix_alpha != nullptr, det = det_partial != nullptr;
#define GSR_RB_LAUNCH(A, B)                                                                                                       \
    hipLaunchKernelGGL((render_backward_kernel<A, B>), dim3(4 * T), dim3(64), 0, stream, cam.width, cam.height, cam.grid_x, T, segs, \
                       num_segs, background, raster, colors, accum_alphas, n_contrib, dL_dpix, full ? dL_dpix_depth : nullptr,      \
                       full ? dL_dpix_alpha : nullptr, accum, colour_slot, det_partial, det_bits)
    if (full && det) GSR_RB_LAUNCH(true, true);
    else if (full) GSR_RB_LAUNCH(true, false);
    else if (det) GSR_RB_LAUNCH(false, true);
    else GSR_RB_LAUNCH(false, false);
#undef GSR_RB_LAUNCH
    return hipGetLastError();
}

hipError_t launch_det_segments(const uint32_t* n_device, const uint32_t* sorted_ids, uint32_t bound, uint32_t* seg_first, uint32_t* seg_end,
                               hipStream_t stream) {
    if (bound == 0u) return hipSuccess;
    hipLaunchKernelGGL(det_segments_kernel, dim3((bound + 255u) / 256u), dim3(256), 0, stream, n_device, sorted_ids, seg_first, seg_end);
    return hipGetLastError();
}

hipError_t launch_det_reduce(int P, const uint32_t* seg_first, const uint32_t* seg_end, const uint32_t* sorted_pos, const uint32_t* bits1,
                             const float* partial1, const uint32_t* bits2, const float* partial2, float* accum, hipStream_t stream) {
    hipLaunchKernelGGL(det_reduce_kernel, dim3(div_up(P, 64)), dim3(64), 0, stream, P, seg_first, seg_end, sorted_pos, bits1, partial1, bits2,
                       partial2, accum);
    return hipGetLastError();
}

hipError_t launch_preprocess_backward(const BackwardInputs& b, const Camera& cam, hipStream_t stream) {
    BackwardArgs g;
    g.P = b.P; g.sh_degree = b.sh_degree; g.M = b.M;
    g.means3D = b.means3D; g.radii = b.radii; g.shs = b.shs; g.scales = b.scales; g.rotations = b.rotations;
    g.cov3D_precomp = b.cov3D_precomp; g.scale_modifier = b.scale_modifier;
    g.accum = b.accum; g.dL_dmean2D = b.dL_dmean2D; g.dL_dconic = b.dL_dconic; g.dL_dopacity = b.dL_dopacity;
    g.dL_dcolor = b.dL_dcolor; g.dL_ddepth = b.dL_ddepth;
    g.dL_dmean3D = b.dL_dmean3D; g.dL_dcov3D = b.dL_dcov3D; g.dL_dsh = b.dL_dsh; g.dL_dscale = b.dL_dscale;
    g.dL_drot = b.dL_drot;
    g.raw = b.raw; g.shs_rest = b.shs_rest; g.opacity_logits = b.opacity_logits; g.dL_dsh_rest = b.dL_dsh_rest;
    g.normal_grads = b.normal_grads; g.cam_pos_normals = cam.cam_pos;
    if (b.raw) hipLaunchKernelGGL(preprocess_backward_kernel<true>, dim3(div_up(b.P, 256)), dim3(256), 0, stream, g, cam);
    else hipLaunchKernelGGL(preprocess_backward_kernel<false>, dim3(div_up(b.P, 256)), dim3(256), 0, stream, g, cam);
    return hipGetLastError();
}

} // namespace gsr
