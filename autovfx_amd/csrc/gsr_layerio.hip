// gsr_layerio.hip -- the compositor's INPUT files on the GPU (blender/blend_all.py::blend_frames reads,
// for every frame, six RGBA PNG layers and four OpenEXR depth passes at Blender's resolution -- blend_all.py:185-205, PIL and OpenCV
// on one host thread).  A PNG or an EXR block is a zlib stream over a PREDICTED image; inflating the stream is a host library's
// job (byte-serial, one stream per file), undoing the predictor is not:
//
//   png_unfilter_kernel : the inflated IDAT stream (per row: one filter-type byte, then the row's bytes; PNG specification section 9,
//       filter types None / Sub / Up / Average / Paeth) -> RGBA8 pixels.  Every byte depends on its left, upper and upper-left
//       neighbours, so the image is a wavefront: lane l of a wave owns row l of a band of 64 rows and runs l pixels behind the lane
//       above it, whose output of the previous step -- the pixel right above -- arrives through one DPP move (wave_shr:1); the
//       four waves of a workgroup (one per SIMD) work on four consecutive bands, each 20 four-pixel groups behind the one above,
//       whose last row it reads from LDS (a workgroup barrier every four groups).  All four bytes of a pixel go through the five predictors at once, two bytes per
//       32-bit register (16-bit fields, packed min / max / shifts); a row's filter type selects by masks, so rows of different
//       types share the instruction stream.  ~60 VALU instructions per pixel step: a 1920x1080 layer in ~1.3 ms on ONE compute
//       unit -- the layers of a frame, and the frames of the decode threads, run beside each other.
//   exr_unpack_channel_kernel : the inflated scanline blocks of an OpenEXR file (ZIP / ZIPS / RLE: per block the byte-wise running
//       sum and the even / odd byte interleave, OpenEXR "ImfZip.cpp" as the file format documents it) -> the bytes of ONE channel's
//       plane.  One workgroup per block: per-thread run totals, a workgroup scan, a second walk that stores the wanted bytes.
//
//   inflate_zlib_kernel : the zlib streams of an OpenEXR ZIP / ZIPS part themselves (one per scanline block: 68 for a 1080p pass) ->
//       the predictor-coded blocks exr_unpack_channel_kernel takes.  One single-wave workgroup per stream, the window in LDS:
//       gsr_inflate_core.h (the same source runs on one host lane under the CPU tests).
//
// Host-side mirror: autovfx_amd/layer_io.py (chunk / header parsing, zlib, the fall-back to Pillow / autovfx_amd.exr for files these
// kernels do not cover).
#include "gsr_internal.h"
#include "gsr_inflate_core.h"

#include <algorithm>
#include <mutex>

namespace gsr {
namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

constexpr int kUnfWaves = 4;        // one per SIMD: the kernel is bound by VALU issue
constexpr int kUnfSpan = 4;         // four-pixel groups between two workgroup barriers
constexpr int kUnfLag = 5;          // spans a band runs behind the band above: after 4 spans + 3 groups more, lane 63 up there has finished pixel 16 s + 16
constexpr int kUnfPad = 64;         // pixels in front of a scratch row (lane l starts l pixels to the left of the image)
constexpr int kUnfTail = 96;        // and behind it (lane 0 runs 63 pixels past the end while lane 63 finishes; whole spans)
constexpr int kUnfMaxWidth = 4096;  // four LDS edge rows of 2 (W + kUnfTail) words: 134 KB
constexpr int kUnfMaxJobs = 8;      // images per launch

struct UnfJob {
    const uint8_t* stream;          // inflated IDAT data
    uint32_t* words;                // scratch: H rows of W + kUnfPad + kUnfTail words
    uint32_t* out;                  // [H][W] RGBA8
    int W, H, C;
};
struct UnfBatch {
    UnfJob job[kUnfMaxJobs];
};

__device__ __forceinline__ u16x2 as_u16(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// Keeps a value the compiler would otherwise see through (a sign mask it turns back into compares and selects, a 0 / ~0 mask it turns
// into a branch): costs nothing, the value stays in its register.
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}

// The Paeth predictor (PNG specification 9.4) on two bytes held in 16-bit fields: p = a + b - c; the neighbour closest to p, ties in
// the order a, b, c.  |p - a| = |b - c|, |p - b| = |a - c|, |p - c| = |a + b - 2c|.  18 instructions for the two bytes.
__device__ __forceinline__ uint32_t paeth2(uint32_t a_, uint32_t b_, uint32_t c_) {
    const u16x2 a = as_u16(a_), b = as_u16(b_), c = as_u16(c_);
    const u16x2 pa = __builtin_elementwise_max(b, c) - __builtin_elementwise_min(b, c);
    const u16x2 pb = __builtin_elementwise_max(a, c) - __builtin_elementwise_min(a, c);
    const u16x2 s = a + b, c2 = c + c;
    const u16x2 pc = __builtin_elementwise_max(s, c2) - __builtin_elementwise_min(s, c2);
    const u16x2 m = __builtin_elementwise_min(pb, pc);
    const uint32_t not_a = opaque(__builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, (u16x2)(m - pa)) >> (i16x2)(15)));    // ones where pa > min(pb, pc)
    const uint32_t not_b = opaque(__builtin_bit_cast(uint32_t, __builtin_bit_cast(i16x2, (u16x2)(pc - pb)) >> (i16x2)(15)));   // ones where pb > pc
    const uint32_t bc = (c_ & not_b) | (b_ & ~not_b);
    return (bc & not_a) | (a_ & ~not_a);
}

__device__ __forceinline__ uint32_t average2(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (u16x2)((as_u16(a) + as_u16(b)) >> (u16x2)(1)));
}

template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_keep(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, kCtrl, 0xF, 0xF, false);
}

// Row bytes -> one aligned 32-bit word per pixel (alpha byte 0 for RGB), in a scratch image whose rows have kUnfPad pixels in front
// and kUnfTail behind: the wavefront kernel's 16-byte loads are then in bounds for every lane at every step.
__global__ void __launch_bounds__(256) png_rows_to_words_kernel(UnfBatch batch) {
    const UnfJob& j = batch.job[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= j.W || y >= j.H) return;
    const uint8_t* p = j.stream + (size_t)y * ((size_t)j.W * j.C + 1) + 1 + (size_t)x * j.C;
    uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    if (j.C == 4) v |= (uint32_t)p[3] << 24;
    j.words[(size_t)y * (j.W + kUnfPad + kUnfTail) + kUnfPad + x] = v;
}

__global__ void __launch_bounds__(kUnfWaves * 64) png_unfilter_kernel(UnfBatch batch) {
    // [kUnfWaves][2][edge_pitch]: the last row of each wave's band as it is produced, even bytes / odd bytes in the form the step uses
    extern __shared__ uint32_t s_edge[];
    const UnfJob& job = batch.job[blockIdx.x];
    const int W = job.W, H = job.H, C = job.C;
    const uint8_t* __restrict__ stream = job.stream;
    const int pitch = W + kUnfPad + kUnfTail;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int edge_pitch = ((W + kUnfTail + 3) & ~3);
    const int groups = (W + 62) / 4 + 1;                 // lane 63 reaches pixel W - 1 in group (W + 62) / 4
    const int spans = (groups + kUnfSpan - 1) / kUnfSpan;
    const int round_spans = spans + kUnfLag * (kUnfWaves - 1);
    const int bands = (H + 63) >> 6;
    const int rounds = (bands + kUnfWaves - 1) / kUnfWaves;
    const size_t stride = (size_t)W * C + 1;
    const uint32_t alpha = (C == 3) ? 0xFF000000u : 0u;
    uint32_t* edge_out = s_edge + wave * 2 * edge_pitch;
    const uint32_t* edge_in = s_edge + ((wave + kUnfWaves - 1) % kUnfWaves) * 2 * edge_pitch;
    constexpr uint32_t kLow = 0x00FF00FFu;

    for (int round = 0; round < rounds; ++round) {
        const int band = round * kUnfWaves + wave;
        const bool band_on = band < bands;               // (wave-uniform)
        const int row = band * 64 + lane;
        const bool row_on = band_on && row < H;
        const int type = row_on ? (int)stream[(size_t)row * stride] : 0;
        const uint32_t m_sub = opaque(type == 1 ? ~0u : 0u), m_up = opaque(type == 2 ? ~0u : 0u), m_avg = opaque(type == 3 ? ~0u : 0u),
                       m_paeth = opaque(type == 4 ? ~0u : 0u);
        const bool edge_on = band > 0;                   // band 0 has zeros above it
        const uint32_t* row_words = job.words + (size_t)(row_on ? row : 0) * pitch + kUnfPad - lane;   // group g: pixels 4 g - lane ... + 3
        uint32_t* row_out = job.out + (size_t)(row_on ? row : 0) * W;
        uint32_t left_e = 0u, left_o = 0u;               // this row's previous pixel: even bytes (R, B) / odd bytes (G, A) in 16-bit fields
        uint32_t corner_e = 0u, corner_o = 0u;           // the row above at the previous step: this step's upper-left
        uint4 cur[kUnfSpan];
#pragma unroll
        for (int k = 0; k < kUnfSpan; ++k) {
            cur[k] = make_uint4(0, 0, 0, 0);
            if (band_on) __builtin_memcpy(&cur[k], row_words + 4 * k, 16);
        }
        for (int t = 0; t < round_spans; ++t) {
            const int s = t - kUnfLag * wave;
            if (band_on && s >= 0 && s < spans) {
                // the next span's pixels (in flight for a whole span) and, from LDS, the row above this band for the whole span
                uint4 nxt[kUnfSpan], ee[kUnfSpan], eo[kUnfSpan];
#pragma unroll
                for (int k = 0; k < kUnfSpan; ++k) {
                    nxt[k] = make_uint4(0, 0, 0, 0);
                    if (s + 1 < spans) __builtin_memcpy(&nxt[k], row_words + 4 * (kUnfSpan * (s + 1) + k), 16);   // (at most W + 62 + 16 + 3 words in: inside kUnfTail)
                    ee[k] = eo[k] = make_uint4(0, 0, 0, 0);
                    if (edge_on) {                       // the same address in every lane: a broadcast; only lane 0 uses it
                        ee[k] = *reinterpret_cast<const uint4*>(edge_in + 4 * (kUnfSpan * s + k));
                        eo[k] = *reinterpret_cast<const uint4*>(edge_in + edge_pitch + 4 * (kUnfSpan * s + k));
                    }
                }
#pragma unroll
                for (int k = 0; k < kUnfSpan; ++k) {
                    const uint32_t f[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
                    const uint32_t above_e[4] = {ee[k].x, ee[k].y, ee[k].z, ee[k].w}, above_o[4] = {eo[k].x, eo[k].y, eo[k].z, eo[k].w};
                    uint32_t px[4];
                    const int x0 = 4 * (kUnfSpan * s + k) - lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int x = x0 + i;
                        // the pixel above: what the lane above produced in the previous step; lane 0 keeps the band above's last row
                        const uint32_t b_e = dpp_keep<0x138>(above_e[i], left_e);
                        const uint32_t b_o = dpp_keep<0x138>(above_o[i], left_o);
                        const uint32_t keep = opaque((unsigned)x < (unsigned)W ? kLow : 0u);     // outside the image everything is zero
                        const uint32_t pred_e = (left_e & m_sub) | (b_e & m_up) | (average2(left_e, b_e) & m_avg) | (paeth2(left_e, b_e, corner_e) & m_paeth);
                        const uint32_t pred_o = (left_o & m_sub) | (b_o & m_up) | (average2(left_o, b_o) & m_avg) | (paeth2(left_o, b_o, corner_o) & m_paeth);
                        corner_e = b_e; corner_o = b_o;
                        left_e = ((f[i] & kLow) + pred_e) & keep;
                        left_o = (((f[i] >> 8) & kLow) + pred_o) & keep;
                        px[i] = left_e | (left_o << 8);
                        if (lane == 63 && (unsigned)x < (unsigned)W) {
                            edge_out[x] = left_e;
                            edge_out[edge_pitch + x] = left_o;
                        }
                    }
                    if (row_on) {
                        if (x0 >= 0 && x0 + 3 < W) {
                            const uint4 v = make_uint4(px[0] | alpha, px[1] | alpha, px[2] | alpha, px[3] | alpha);
                            __builtin_memcpy(row_out + x0, &v, 16);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if ((unsigned)(x0 + i) < (unsigned)W) row_out[x0 + i] = px[i] | alpha;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < kUnfSpan; ++k) cur[k] = nxt[k];
            }
            // hand the edge pixels over: LDS operations only (the image loads and stores stay in flight across the barrier)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
}

// ---- OpenEXR blocks ------------------------------------------------------------------------------------------------------------
constexpr int kExrThreads = 256;
constexpr int kExrTile = kExrThreads * 16;     // bytes per step of a workgroup: a 16-byte piece per thread

struct ExrPlan {
    const uint8_t* blocks;   // the inflated blocks one after another, in increasing y
    uint8_t* plane;          // [H][c_bytes]
    int H, bytes_per_line, lines_per_block, c_at, c_bytes;
};

// One workgroup per scanline block.  The block is [even bytes of all its lines | odd bytes], each byte the running sum (mod 256) of
// the stored bytes up to it, + 128 per odd position (the format's bias).  The workgroup walks the block in 4 KB tiles, 16 consecutive
// bytes per thread (coalesced 16-byte loads): per-thread byte totals, a workgroup scan, the tile's carry; then every thread runs
// the sum through its 16 bytes and stores those that belong to the wanted channel -- byte k of the first half is byte 2 k of the
// block's lines, byte k of the second half byte 2 k + 1.
__global__ void __launch_bounds__(kExrThreads) exr_unpack_channel_kernel(ExrPlan p) {
    __shared__ uint32_t s_wave[kExrThreads / 64];
    const int y0 = blockIdx.x * p.lines_per_block;
    const int lines = min(p.lines_per_block, p.H - y0);
    const uint32_t n = (uint32_t)lines * (uint32_t)p.bytes_per_line;       // (even: every pixel type has an even size)
    const uint32_t half = n >> 1, bpl = (uint32_t)p.bytes_per_line;
    const uint8_t* t = p.blocks + (size_t)y0 * p.bytes_per_line;
    const bool aligned = (reinterpret_cast<uintptr_t>(t) & 15u) == 0;
    const uint32_t c_at = (uint32_t)p.c_at, c_end = (uint32_t)(p.c_at + p.c_bytes);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;                                                     // the running sum in front of the tile (mod 256 at use)
    for (uint32_t tile = 0; tile < n; tile += kExrTile) {
        const uint32_t at = tile + threadIdx.x * 16u;
        uint32_t q[4] = {0u, 0u, 0u, 0u};
        if (at + 16u <= n && aligned) {
            const uint4 v = *reinterpret_cast<const uint4*>(t + at);
            q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
        } else if (at < n) {
            for (uint32_t j = 0; j < 16u && at + j < n; ++j) q[j >> 2] |= (uint32_t)t[at + j] << (8u * (j & 3u));
        }
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) sum = __builtin_amdgcn_sad_u8(q[k], 0u, sum);
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        __syncthreads();                                                    // (s_wave of the previous tile has been read)
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = carry + incl - sum, total = 0;
#pragma unroll
        for (int w = 0; w < kExrThreads / 64; ++w) {
            if (w < wave) before += s_wave[w];
            total += s_wave[w];
        }
        carry += total;
        if (at >= n) continue;
        // the thread's bytes: where byte `at` of the block lands in the block's lines
        uint32_t running = before;
        uint32_t h = at >= half ? 1u : 0u;
        uint32_t off = 2u * (at - h * half) + h;
        uint32_t line = off / bpl, within = off - line * bpl;
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j) {
            const uint32_t i = at + j;
            if (i == half && j != 0u) {                                     // the second half starts inside this piece (a short last block)
                h = 1u;
                line = 0u;
                within = 1u;
            }
            running += (q[j >> 2] >> (8u * (j & 3u))) & 255u;
            if (i < n && within >= c_at && within < c_end)
                p.plane[(size_t)(y0 + (int)line) * p.c_bytes + (within - c_at)] = (uint8_t)(running + ((i & 1u) ? 128u : 0u));
            within += 2u;
            if (within >= bpl) {
                within -= bpl;
                ++line;
            }
        }
    }
}

// ---- zlib streams: one single-wave workgroup per stream (gsr_inflate_core.h) ----------------------------------------------------
__global__ void __launch_bounds__(64) inflate_zlib_kernel(const uint8_t* __restrict__ streams, uint8_t* __restrict__ out, const InflateJob* __restrict__ jobs,
                                                          int* __restrict__ status, int* __restrict__ any_error) {
    __shared__ inflate::Shared sh;
    const InflateJob j = jobs[blockIdx.x];
    const int rc = inflate::inflate_zlib<64>(streams + j.src_at, j.src_bytes, out + j.dst_at, j.dst_bytes, sh);
    if (threadIdx.x == 0) {
        status[blockIdx.x] = rc;
        if (rc != 0 && any_error) atomicOr(any_error, 1);
    }
}

} // namespace

hipError_t launch_inflate_zlib_blocks(const uint8_t* streams, uint8_t* out, const InflateJob* jobs, int count, int* status, int* any_error, hipStream_t stream) {
    if (count > 0) hipLaunchKernelGGL(inflate_zlib_kernel, dim3(count), dim3(64), 0, stream, streams, out, jobs, status, any_error);
    return hipGetLastError();
}

size_t png_unfilter_scratch_bytes(int W, int H) {
    if (W <= 0 || H <= 0 || W > kUnfMaxWidth) return 0;
    return (size_t)H * (size_t)(W + kUnfPad + kUnfTail) * 4;
}

hipError_t launch_png_unfilter_batch(int n, const PngUnfilterJob* jobs, hipStream_t stream) {
    static std::once_flag once[16];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipError_t attr = hipSuccess;
    std::call_once(once[dev & 15], [&] {
        attr = hipFuncSetAttribute(reinterpret_cast<const void*>(png_unfilter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    if (attr != hipSuccess) return attr;
    for (int at = 0; at < n; at += kUnfMaxJobs) {
        UnfBatch batch = {};
        const int m = std::min(kUnfMaxJobs, n - at);
        int max_w = 0, max_h = 0;
        for (int i = 0; i < m; ++i) {
            const PngUnfilterJob& j = jobs[at + i];
            batch.job[i] = {j.scanlines, reinterpret_cast<uint32_t*>(j.scratch), reinterpret_cast<uint32_t*>(j.out_rgba), j.width, j.height, j.channels};
            max_w = std::max(max_w, j.width);
            max_h = std::max(max_h, j.height);
        }
        hipLaunchKernelGGL(png_rows_to_words_kernel, dim3((max_w + 255) / 256, max_h, m), dim3(256), 0, stream, batch);
        const size_t lds = (size_t)kUnfWaves * 2 * ((max_w + kUnfTail + 3) & ~3) * 4;
        hipLaunchKernelGGL(png_unfilter_kernel, dim3(m), dim3(kUnfWaves * 64), lds, stream, batch);
    }
    return hipGetLastError();
}

hipError_t launch_exr_unpack_channel(const uint8_t* blocks, int H, int bytes_per_line, int lines_per_block, int c_at, int c_bytes, uint8_t* plane,
                                     hipStream_t stream) {
    ExrPlan p = {blocks, plane, H, bytes_per_line, lines_per_block, c_at, c_bytes};
    const int n_blocks = (H + lines_per_block - 1) / lines_per_block;
    hipLaunchKernelGGL(exr_unpack_channel_kernel, dim3(n_blocks), dim3(kExrThreads), 0, stream, p);
    return hipGetLastError();
}

} // namespace gsr
