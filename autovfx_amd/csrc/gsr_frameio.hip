// gsr_frameio.hip -- the frame loop's outputs and the compositor's inputs on the GPU.
//
//   png_encode_kernel / png_crc_kernel / png_finish_kernel : an 8-bit RGB / RGBA image that lives on the GPU -> the bytes of its PNG FILE, also on
//       the GPU, so that one device-to-host copy and one write() put a frame on disk.  Replaces, for the reference's per-frame
//       files (scene_representation.py:425-438: torchvision.utils.save_image, cv2.imwrite x 2), the host-side zlib pass that made
//       the unchanged trajectory job I/O-bound 28x (round 4: 6.8 ms per 960x540 frame through a 32-thread pool, 0.24 ms to
//       render and composite it).
//   resize kernels                         : PIL's Image.resize(BILINEAR) on RGBA8 and Image.resize(NEAREST) on fp32 depth,
//       bit for bit (blender/blend_all.py:21-28 downsample_image, called at :217-234).
//
// PNG as written here: signature, IHDR, ONE IDAT chunk, IEND.  The IDAT payload is a zlib stream of STORED deflate blocks
// (RFC 1951 section 3.2.4: BTYPE = 00, up to 65535 bytes each, no compression): filter-0 scanlines copied through.  Every PNG
// reader inflates it to the same pixels as the reference's compressed files; the file is as large as the raw image
// (960x540 RGBA: 2.07 MB).  The two checksums a reader verifies are computed here:
//   * Adler-32 of the scanline stream (RFC 1950): s1 = 1 + sum b_r, s2 = N + sum (N - r) b_r, both mod 65521 -- two integer sums,
//     accumulated in 64 bits with integer atomics (order-free) and reduced at the end;
//   * CRC-32 of "IDAT" + payload (PNG section 5.5): the CRC without its pre / post conditioning is LINEAR over GF(2), so the
//     message splits into 16-byte pieces whose raw CRCs are shifted to their place -- multiplied by x^(8 * bytes behind the
//     piece) mod P, zlib's crc32_combine arithmetic (multmodp / x2nmodp, crc32.c) -- and XORed together in any order; the
//     conditioning is one more term, 0xFFFFFFFF x^(8 N) ^ 0xFFFFFFFF.
#include "gsr_internal.h"

#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

namespace gsr {
namespace {

constexpr uint32_t kCrcPoly = 0xEDB88320u;   // reflected CRC-32 (PNG, zlib)
constexpr uint32_t kStored = 65535u;         // bytes of a stored deflate block

// a(x) * b(x) mod P(x), reflected bit order (zlib crc32.c: multmodp)
__host__ __device__ inline uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0u;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0u) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

struct PngTables {
    uint32_t x2n[32];     // x^(2^k) mod P
    uint32_t byte[256];   // CRC table, one byte at a time
};

// x^(n * 2^k) mod P (crc32.c: x2nmodp)
__host__ __device__ inline uint32_t crc_x2nmodp(const uint32_t* x2n, unsigned long long n, unsigned k) {
    uint32_t p = 1u << 31;
    while (n) {
        if (n & 1ull) p = crc_multmodp(x2n[k & 31u], p);
        n >>= 1;
        ++k;
    }
    return p;
}

struct PngLayout {
    int W, H, C, planar;
    uint32_t row_len;            // 1 + W * C
    unsigned long long N;        // bytes of the scanline stream = H * row_len
    unsigned long long blocks;   // stored blocks
    unsigned long long data_at;  // file offset of the IDAT payload (= 41)
    unsigned long long data_len; // 2 + 5 * blocks + N + 4
    unsigned long long file_len;
    uint32_t crc_init_term;      // 0xFFFFFFFF x^(8 * (4 + data_len)) mod P: the CRC's pre-conditioning as one more linear term (host-computed)
    uint32_t crc_tail_shift;     // x^(8 * ((data_at + data_len) mod 64)) mod P
    uint8_t head[48];            // the first data_at bytes of the file: signature, IHDR chunk, IDAT length + type
    uint8_t tail[16];            // the 12 bytes after the IDAT CRC: IEND chunk
};

// One lane = 16 consecutive, 16-byte aligned bytes of the FILE: where each byte comes from is worked out once per lane (two 64-bit
// divisions) and stepped from byte to byte; the CRC is a kernel of its own (below).  Adler-32's two sums are taken here, where the
// scanline bytes pass through registers anyway.
__global__ void __launch_bounds__(256) png_encode_kernel(PngLayout L, const uint8_t* __restrict__ pixels, uint8_t* __restrict__ out,
                                                        unsigned long long* __restrict__ adler_partials /*[workgroups][2], every entry written*/) {
    __shared__ unsigned long long s_sum[4][2];
    const unsigned long long first = ((unsigned long long)blockIdx.x * 256ull + threadIdx.x) * 16ull;
    unsigned long long a1 = 0ull, a2 = 0ull;
    if (first < L.file_len) {
        const unsigned long long adler_at = L.data_at + L.data_len - 4ull;   // Adler-32, then the chunk's CRC, then IEND
        const unsigned long long stream_at = L.data_at + 2ull;                // first stored block's header
        // position of this lane's first byte inside the deflate stream, if it is there: (block, offset in the block incl. its 5-byte header),
        // and for a scanline byte (row, column in the row incl. its filter byte, pixel x, channel)
        unsigned long long block = 0ull, r = 0ull, row = 0ull;
        uint32_t off = 0u, col = 0u, x = 0u, ch = 0u;
        bool placed = false;
        // Phase 1: where each of the 16 bytes comes from -- a constant, or an address in the image -- without touching the image;
        // phase 2: the (up to 16) image bytes, all loads in flight together; phase 3: assemble, and the Adler sums.
        uint32_t konst[16];
        long long src[16];           // >= 0: index into pixels; -1: konst[k]
        unsigned long long weight[16];   // N - r for a scanline byte (Adler's second sum), 0 otherwise
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned long long f = first + (unsigned long long)k;
            uint32_t b = 0u;
            long long from = -1;
            unsigned long long wgt = 0ull;
            if (f >= L.file_len) {
            } else if (f < L.data_at) {
                b = L.head[f];
            } else if (f >= adler_at) {
                const unsigned long long t = f - adler_at;
                b = t < 8ull ? 0u : L.tail[t - 8ull];       // the two checksums are filled in by png_finish_kernel
            } else if (f < stream_at) {
                b = f == L.data_at ? 0x78u : 0x01u;         // zlib header: deflate, 32 K window, no dictionary, fastest (FCHECK makes it % 31 == 0)
            } else {
                if (!placed) {
                    const unsigned long long e = f - stream_at;
                    block = e / (kStored + 5ull);
                    off = (uint32_t)(e - block * (kStored + 5ull));
                    placed = true;
                    r = block * kStored + (off >= 5u ? off - 5u : 0u);   // this scanline byte, or the one that follows the header
                    row = r / L.row_len;
                    col = (uint32_t)(r - row * L.row_len);
                    if (col != 0u) { x = (col - 1u) / (uint32_t)L.C; ch = (col - 1u) - x * (uint32_t)L.C; }
                }
                if (off < 5u) {
                    const unsigned long long left = L.N - block * kStored;
                    const uint32_t len = left < kStored ? (uint32_t)left : kStored;
                    b = off == 0u ? (block + 1ull == L.blocks ? 1u : 0u)
                      : off == 1u ? (len & 0xFFu) : off == 2u ? (len >> 8) : off == 3u ? (~len & 0xFFu) : ((~len >> 8) & 0xFFu);
                } else {
                    wgt = L.N - r;
                    if (col != 0u) {                        // (col 0: the row's filter type, 0)
                        from = L.planar ? (long long)(((size_t)ch * L.H + row) * L.W + x) : (long long)((row * L.W + x) * L.C + ch);
                        if (++ch == (uint32_t)L.C) { ch = 0u; ++x; }
                    }
                    ++r;
                    if (++col == L.row_len) { col = 0u; x = 0u; ch = 0u; ++row; }
                }
                if (++off == kStored + 5u) { off = 0u; ++block; }
            }
            konst[k] = b; src[k] = from; weight[k] = wgt;
        }
        uint32_t bytes[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) bytes[k] = src[k] >= 0 ? (uint32_t)pixels[src[k]] : konst[k];
        uint32_t words[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            words[k >> 2] |= bytes[k] << (8 * (k & 3));
            if (weight[k] != 0ull) { a1 += bytes[k]; a2 += weight[k] * bytes[k]; }
        }
        if (first + 16ull <= L.file_len) {
            *reinterpret_cast<uint4*>(out + first) = make_uint4(words[0], words[1], words[2], words[3]);
        } else {
            for (unsigned long long f = first; f < L.file_len; ++f) out[f] = (uint8_t)(words[(f - first) >> 2] >> (8 * ((f - first) & 3)));
        }
    }
    // wave sums -> workgroup sums -> ONE plain store per workgroup; png_finish_kernel adds the workgroups up.  (Same-address atomics
    // are served one per ~12 ns on this GPU: two per wave made this kernel 50 us for a 2 MB image, a memset in front of it included.)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a1 += __shfl_xor(a1, d);
        a2 += __shfl_xor(a2, d);
    }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6][0] = a1; s_sum[threadIdx.x >> 6][1] = a2 % 65521ull; }
    __syncthreads();
    if (threadIdx.x == 0) {
        adler_partials[2 * (size_t)blockIdx.x + 0] = s_sum[0][0] + s_sum[1][0] + s_sum[2][0] + s_sum[3][0];
        adler_partials[2 * (size_t)blockIdx.x + 1] = (s_sum[0][1] + s_sum[1][1] + s_sum[2][1] + s_sum[3][1]) % 65521ull;
    }
}

// x^(8 * 64 * m) mod P for m = i, 256 i, 65536 i (i = 0 .. 255): a CRC piece's shift by 64 m bytes is a product of three table entries.
__device__ uint32_t g_crc_shift64[3][256];

// CRC-32 of the IDAT chunk's type + payload, read back from the file image png_encode_kernel just wrote (L2-resident).  One lane =
// kCrcChunk consecutive file bytes, loaded up front: their raw CRC four bytes at a time (slicing-by-4: four table lookups per word
// instead of a chain of four), then ONE shift to the lane's place in the message -- x^(8 * bytes behind it) mod P -- and an XOR into
// the total.
constexpr uint32_t kCrcChunk = 64u;
__global__ void __launch_bounds__(256) png_crc_kernel(PngLayout L, PngTables T, const uint8_t* __restrict__ file,
                                                     uint32_t* __restrict__ crc_partials /*[workgroups], every entry written*/) {
    __shared__ uint32_t s_t[4][256];
    __shared__ uint32_t s_shift[3][256];
    __shared__ uint32_t s_crc[4];
    {
        const int i = threadIdx.x;
        const uint32_t t0 = T.byte[i];
        const uint32_t t1 = (t0 >> 8) ^ T.byte[t0 & 0xFFu];
        const uint32_t t2 = (t1 >> 8) ^ T.byte[t1 & 0xFFu];
        s_t[0][i] = t0; s_t[1][i] = t1; s_t[2][i] = t2; s_t[3][i] = (t2 >> 8) ^ T.byte[t2 & 0xFFu];
        s_shift[0][i] = g_crc_shift64[0][i]; s_shift[1][i] = g_crc_shift64[1][i]; s_shift[2][i] = g_crc_shift64[2][i];
    }
    __syncthreads();
    const unsigned long long crc_from = L.data_at - 4ull, crc_end = L.data_at + L.data_len;   // "IDAT" ... Adler-32 (still zeros) inclusive
    const unsigned long long lo0 = ((unsigned long long)blockIdx.x * 256ull + threadIdx.x) * kCrcChunk;
    uint32_t crc = 0u;
    if (lo0 < crc_end && lo0 + kCrcChunk > crc_from) {
        // the lane's 64 bytes in four 16-byte loads, all issued before the first is used (the file image ends with 32 bytes of
        // scratch behind a 16-byte boundary: reading the whole chunk is always inside the buffer)
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)   // (a 16-byte piece that starts inside the message ends inside the file)
            v[k] = lo0 + 16u * k < crc_end ? *reinterpret_cast<const uint4*>(file + lo0 + 16u * k) : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t w[16] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
        const unsigned long long lo = lo0 < crc_from ? crc_from : lo0, hi = lo0 + kCrcChunk < crc_end ? lo0 + kCrcChunk : crc_end;
        if (lo == lo0 && hi == lo0 + kCrcChunk) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t x = crc ^ w[k];
                crc = s_t[3][x & 0xFFu] ^ s_t[2][(x >> 8) & 0xFFu] ^ s_t[1][(x >> 16) & 0xFFu] ^ s_t[0][x >> 24];
            }
        } else {   // the first and the last chunk of the message: byte by byte
            for (unsigned long long f = lo; f < hi; ++f) {
                const uint32_t b = (w[(f - lo0) >> 2] >> (8 * ((f - lo0) & 3ull))) & 0xFFu;
                crc = s_t[0][(crc ^ b) & 0xFFu] ^ (crc >> 8);
            }
        }
        if (crc != 0u && hi < crc_end) {
            // shift to its place: crc_end - hi = 64 m + (crc_end mod 64) bytes lie behind this piece (hi is a multiple of 64 here);
            // x^(8 * 64 m) from the three tables, x^(8 * (crc_end mod 64)) host-computed -- four modular products instead of the
            // twenty of an exponentiation (38 -> 12 us for a 2 MB image)
            const unsigned long long m = (crc_end - hi) >> 6;
            crc = crc_multmodp(L.crc_tail_shift, crc);
            if (m & 0xFFull) crc = crc_multmodp(s_shift[0][m & 0xFFull], crc);
            if ((m >> 8) & 0xFFull) crc = crc_multmodp(s_shift[1][(m >> 8) & 0xFFull], crc);
            if (m >> 16) crc = crc_multmodp(s_shift[2][(m >> 16) & 0xFFull], crc);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) crc ^= (uint32_t)__shfl_xor((int)crc, d);
    if ((threadIdx.x & 63) == 0) s_crc[threadIdx.x >> 6] = crc;
    __syncthreads();
    if (threadIdx.x == 0) crc_partials[blockIdx.x] = s_crc[0] ^ s_crc[1] ^ s_crc[2] ^ s_crc[3];   // (XOR: any order, same result)
}

// One wave: adds up the workgroups' Adler sums and XORs their CRC terms, writes the two checksums into the file image.
__global__ void __launch_bounds__(64) png_finish_kernel(PngLayout L, uint8_t* __restrict__ out, const unsigned long long* __restrict__ adler_partials,
                                                       uint32_t adler_groups, const uint32_t* __restrict__ crc_partials, uint32_t crc_groups) {
    unsigned long long a1 = 0ull, a2 = 0ull;
    uint32_t crc_sum = 0u;
    for (uint32_t i = threadIdx.x; i < adler_groups; i += 64u) { a1 += adler_partials[2 * (size_t)i]; a2 += adler_partials[2 * (size_t)i + 1]; }
    for (uint32_t i = threadIdx.x; i < crc_groups; i += 64u) crc_sum ^= crc_partials[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a1 += __shfl_xor(a1, d);
        a2 += __shfl_xor(a2, d);
        crc_sum ^= (uint32_t)__shfl_xor((int)crc_sum, d);
    }
    if (threadIdx.x != 0) return;
    const uint32_t s1 = (uint32_t)((1ull + a1) % 65521ull);
    const uint32_t s2 = (uint32_t)((L.N % 65521ull + a2) % 65521ull);
    const uint32_t adler = (s2 << 16) | s1;
    const unsigned long long adler_at = L.data_at + L.data_len - 4ull;
    uint32_t crc_a = 0u;   // raw CRC of the four Adler bytes: the last bytes of the message, nothing behind them
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = (adler >> (24 - 8 * k)) & 0xFFu;   // big-endian
        out[adler_at + k] = (uint8_t)b;
        crc_a ^= b;
        for (int i = 0; i < 8; ++i) crc_a = (crc_a & 1u) ? (crc_a >> 1) ^ kCrcPoly : crc_a >> 1;
    }
    const uint32_t crc = (crc_sum ^ crc_a ^ L.crc_init_term) ^ 0xFFFFFFFFu;
    for (int k = 0; k < 4; ++k) out[adler_at + 4 + k] = (uint8_t)(crc >> (24 - 8 * k));
}

const PngTables& png_tables() {
    static PngTables t = [] {
        PngTables v;
        uint32_t p = 1u << 30;   // x^1
        v.x2n[0] = p;
        for (int n = 1; n < 32; ++n) v.x2n[n] = p = crc_multmodp(p, p);
        for (uint32_t i = 0; i < 256u; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
            v.byte[i] = c;
        }
        return v;
    }();
    return t;
}

uint32_t host_crc(const uint8_t* p, size_t n) {
    const PngTables& t = png_tables();
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = t.byte[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

bool png_layout(int W, int H, int C, int planar, PngLayout* L) {
    if (W <= 0 || H <= 0 || (C != 3 && C != 4)) return false;
    L->W = W; L->H = H; L->C = C; L->planar = planar ? 1 : 0;
    L->row_len = 1u + (uint32_t)W * (uint32_t)C;
    L->N = (unsigned long long)H * L->row_len;
    L->blocks = (L->N + kStored - 1ull) / kStored;
    L->data_at = 41ull;
    L->data_len = 2ull + 5ull * L->blocks + L->N + 4ull;
    if (L->data_len > 0x7FFFFFFFull) return false;   // one IDAT chunk: a 31-bit length (an image of 2 GB)
    L->file_len = L->data_at + L->data_len + 4ull + 12ull;
    L->crc_init_term = crc_multmodp(crc_x2nmodp(png_tables().x2n, 4ull + L->data_len, 3u), 0xFFFFFFFFu);
    L->crc_tail_shift = crc_x2nmodp(png_tables().x2n, (L->data_at + L->data_len) & 63ull, 3u);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    uint8_t* h = L->head;
    for (int i = 0; i < 48; ++i) h[i] = 0;
    for (int i = 0; i < 8; ++i) h[i] = sig[i];
    put_be32(h + 8, 13u);
    h[12] = 'I'; h[13] = 'H'; h[14] = 'D'; h[15] = 'R';
    put_be32(h + 16, (uint32_t)W);
    put_be32(h + 20, (uint32_t)H);
    h[24] = 8; h[25] = C == 4 ? 6 : 2; h[26] = 0; h[27] = 0; h[28] = 0;   // 8 bits, truecolour (+ alpha), deflate, adaptive filtering, no interlace
    put_be32(h + 29, host_crc(h + 12, 17));
    put_be32(h + 33, (uint32_t)L->data_len);
    h[37] = 'I'; h[38] = 'D'; h[39] = 'A'; h[40] = 'T';
    uint8_t* t = L->tail;
    for (int i = 0; i < 16; ++i) t[i] = 0;
    put_be32(t, 0u);
    t[4] = 'I'; t[5] = 'E'; t[6] = 'N'; t[7] = 'D';
    put_be32(t + 8, host_crc(t + 4, 4));
    return true;
}

// ================================================================================================
// Compressed PNGs (round 6): the same file with an IDAT of DYNAMIC-HUFFMAN deflate blocks instead of stored ones.
//
// What the reference writes are compressed files (scene_representation.py:427-438: save_image -> PIL -> zlib level 6; cv2.imwrite ->
// libpng).  What compresses a rendered frame is (a) a predictive scanline filter and (b) an entropy code fitted to the residuals;
// LZ77 string matching buys almost nothing on such data (zlib's own Z_RLE strategy exists for that reason).  So:
//   * every scanline is Paeth-filtered (PNG filter type 4: a pure function of the raw pixel and its left / upper / upper-left
//     neighbours -- embarrassingly parallel);
//   * tokens are literals and run-length matches only (distance 1, length 3 .. 64: "the byte before, n more times" -- flat
//     backgrounds), found independently in every 64-byte piece of the filtered stream, one lane per piece;
//   * ONE Huffman code per image, built on the GPU from the image's token histogram (length-limited to 15 bits, canonical), and
//     sent in every block's header with a fixed 4-bit code for the code lengths (RFC 1951 3.2.7; 153 bytes per block);
//   * the stream is cut into blocks of 16 KB of filtered bytes, one workgroup each.  A non-final block ends with an empty stored
//     block (zlib's Z_SYNC_FLUSH marker, 00 00 FF FF after padding to a byte): every block's bits start on a byte boundary, so a
//     block's place in the file is a prefix sum of BYTE counts -- known from the per-block histograms and the code lengths before
//     a single bit is written; a block that would not shrink is sent stored.
// Kernels, each launched ONCE for all images of a batch (a frame's three PNGs: blockIdx.y = image):
//   png_filter_kernel   the filtered stream to scratch, 16 bytes per lane, Adler-32 partial sums            (HBM-bound, a few us)
//   png_hist_kernel     one workgroup per block: tokens, histogram
//   png_table_kernel    one workgroup per image: sort by count, Huffman's two-queue merge (the one serial step: 285 merges by one lane
//                       with the queue heads in registers), depths / lengths / canonical codes in parallel, block sizes and places
//   png_deflate_kernel  one workgroup per block: tokens -> bits in LDS (phase-aligned with the file), one coalesced copy out
//   png_crc_dynamic_kernel / png_finish_dynamic_kernel   as for stored files, the lengths read from the device
// First version (same round) re-filtered inside the histogram and deflate kernels and built the code on one lane out of LDS: 58 + 182 + 61 us
// per image, three images one after the other (profiles/r06_png_deflate.md).
// ================================================================================================
constexpr uint32_t kDefBlock = 16384u;        // filtered-stream bytes per deflate block (= per workgroup)
constexpr uint32_t kDefPiece = 64u;           // ... per lane
constexpr uint32_t kDefPitch = 68u;           // LDS pitch of a piece: 17 words, conflict-free word accesses across lanes
constexpr uint32_t kDefSyms = 288u;           // literal / length alphabet (286 used) padded
constexpr uint32_t kDefHeaderBits = 3u + 14u + 19u * 3u + 287u * 4u;   // BFINAL + BTYPE, HLIT / HDIST / HCLEN, code-length code, 286 + 1 lengths = 1222
constexpr uint32_t kDefHeaderBytes = (kDefHeaderBits + 7u) / 8u;        // 153
constexpr uint32_t kFilterBytes = 4096u;      // stream bytes per workgroup of the filter kernel (16 per lane)
constexpr int kMaxBatch = 3;

struct PngDynamic {                // what the table kernel leaves for the kernels behind it (device memory)
    unsigned long long data_len;   // IDAT payload: 2 + deflate bytes + 4
    unsigned long long file_len;
    uint32_t crc_init_term, crc_tail_shift;
    uint32_t table[kDefSyms];      // bit-reversed code | length << 16
    uint32_t header_words[(kDefHeaderBytes + 3u) / 4u + 1u];   // a dynamic block's first 1222 bits (BFINAL = 0)
};

struct DeflateScratch {            // offsets from the start of the image's scratch; `out` holds the file alone
    size_t dyn_at, hist_at, adler_at, off_at, crc_at, stream_at, total, out_room;
    uint32_t blocks, crc_groups, filter_groups;
    unsigned long long file_max;
};

DeflateScratch deflate_scratch(const PngLayout& L) {
    DeflateScratch d;
    d.blocks = (uint32_t)((L.N + kDefBlock - 1ull) / kDefBlock);
    d.filter_groups = (uint32_t)((L.N + kFilterBytes - 1ull) / kFilterBytes);
    d.file_max = L.data_at + 2ull + 5ull * d.blocks + L.N + 4ull + 4ull + 12ull;
    d.crc_groups = (uint32_t)((((d.file_max + kCrcChunk - 1ull) / kCrcChunk) + 255ull) / 256ull);
    d.out_room = ((size_t)d.file_max + 64 + 15) & ~size_t(15);    // (the CRC kernel reads whole 64-byte chunks)
    size_t at = 0;
    d.dyn_at = at;    at += (sizeof(PngDynamic) + 15) & ~size_t(15);
    d.hist_at = at;   at += (size_t)d.blocks * kDefSyms * 4;
    d.adler_at = at;  at += (size_t)d.filter_groups * 16;
    d.off_at = at;    at += (((size_t)d.blocks + 1) * 4 + 15) & ~size_t(15);
    d.crc_at = at;    at += ((size_t)d.crc_groups * 4 + 15) & ~size_t(15);
    d.stream_at = at; at += (size_t)d.blocks * kDefBlock + 16;   // the filtered scanline stream, whole blocks (the tail is never interpreted)
    d.total = at;
    return d;
}

struct PngJob {
    PngLayout L;
    const uint8_t* pixels;
    uint8_t* out;                       // the file image
    uint8_t* stream;                    // filtered scanline stream (scratch)
    PngDynamic* dyn;
    uint32_t* hist;                     // [blocks][288]
    unsigned long long* adler;          // [filter_groups][2]
    uint32_t* block_off;                // [blocks]: every block's size in the file; bit 31: stored
    uint32_t* crc_partials;             // [crc_groups]
    unsigned long long* out_len;        // nullable
    uint32_t blocks, filter_groups, crc_groups;
};
struct PngBatch { int n; PngJob job[kMaxBatch]; };

// The filtered scanline stream's byte r: the row's filter type (4) in column 0, else raw - paeth(left, up, upper left) mod 256.
__device__ __forceinline__ uint32_t paeth_stream_byte(const PngLayout& L, const uint8_t* __restrict__ pixels, uint32_t row, uint32_t col) {
    if (col == 0u) return 4u;
    const uint32_t c = col - 1u;
    const uint32_t x = L.C == 4 ? c >> 2 : c / 3u, ch = c - x * (uint32_t)L.C;
    const size_t at = L.planar ? ((size_t)ch * L.H + row) * L.W + x : ((size_t)row * L.W + x) * L.C + ch;
    const size_t dx = L.planar ? 1 : (size_t)L.C, dy = L.planar ? (size_t)L.W : (size_t)L.W * L.C;
    const int raw = pixels[at];
    const int a = x > 0u ? pixels[at - dx] : 0, b = row > 0u ? pixels[at - dy] : 0, cc = (x > 0u && row > 0u) ? pixels[at - dy - dx] : 0;
    const int pr = a + b - cc;
    const int pa = abs(pr - a), pb = abs(pr - b), pc = abs(pr - cc);
    const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : cc);
    return (uint32_t)(raw - pred) & 255u;
}

// The filtered stream to scratch, 4 KB per workgroup: in every step the lanes of a wave take CONSECUTIVE bytes, so a wave's loads of the
// raw pixels and their three neighbours are a few whole cache lines and its stores one (a lane that walked 16 consecutive bytes of its
// own made every load a 64-line gather: 24 us for a frame's three images instead of 4); Adler-32's two sums per workgroup.
__global__ void __launch_bounds__(256) png_filter_kernel(PngBatch B) {
    if ((int)blockIdx.y >= B.n) return;
    const PngJob& J = B.job[blockIdx.y];
    if (blockIdx.x >= J.filter_groups) return;
    const PngLayout& L = J.L;
    __shared__ unsigned long long s_sum[4][2];
    const unsigned long long base = (unsigned long long)blockIdx.x * kFilterBytes;
    unsigned long long a1 = 0ull, a2 = 0ull;
#pragma unroll 4
    for (uint32_t k = 0; k < kFilterBytes / 256u; ++k) {
        const unsigned long long r = base + k * 256u + threadIdx.x;
        if (r < L.N) {
            const uint32_t row = (uint32_t)r / L.row_len, col = (uint32_t)r - row * L.row_len;     // (N < 2^31)
            const uint32_t v = paeth_stream_byte(L, J.pixels, row, col);
            J.stream[r] = (uint8_t)v;
            a1 += v;
            a2 += (L.N - r) * v;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a1 += __shfl_xor(a1, d); a2 += __shfl_xor(a2, d); }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6][0] = a1; s_sum[threadIdx.x >> 6][1] = a2 % 65521ull; }
    __syncthreads();
    if (threadIdx.x == 0) {
        J.adler[2 * (size_t)blockIdx.x + 0] = s_sum[0][0] + s_sum[1][0] + s_sum[2][0] + s_sum[3][0];
        J.adler[2 * (size_t)blockIdx.x + 1] = (s_sum[0][1] + s_sum[1][1] + s_sum[2][1] + s_sum[3][1]) % 65521ull;
    }
}

// A workgroup's 16 KB block of the filtered stream into LDS, consecutive lanes loading consecutive 16-byte pieces (piece p of the block
// lands at p / 4 * kDefPitch + p % 4 * 16: a lane's 64-byte piece has the pitch that keeps the later per-lane word reads conflict-free).
// Call __syncthreads() behind it.  piece_prev(): the byte in front of lane t's piece (256: none -- the first byte of the image).
__device__ __forceinline__ void load_block(const uint8_t* __restrict__ stream, uint32_t block, uint8_t* s_stream) {
    const uint4* src = reinterpret_cast<const uint4*>(stream + (size_t)block * kDefBlock);
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t p = k * 256u + threadIdx.x;
        const uint4 v = src[p];
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_stream + (p >> 2) * kDefPitch + (p & 3u) * 16u);
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}
__device__ __forceinline__ uint32_t piece_prev(const uint8_t* __restrict__ stream, uint32_t block, uint32_t t, const uint8_t* s_stream) {
    if (t) return s_stream[(t - 1u) * kDefPitch + 63u];
    return block == 0u ? 256u : (uint32_t)stream[(size_t)block * kDefBlock - 1u];
}

// length 3 .. 64 -> (symbol - 257, extra bits, extra value)  (RFC 1951 3.2.5)
__device__ __forceinline__ void length_code(uint32_t n, uint32_t* k, uint32_t* ebits, uint32_t* evalue) {
    if (n < 11u) { *k = n - 3u; *ebits = 0u; *evalue = 0u; }
    else if (n < 19u) { *k = 8u + ((n - 11u) >> 1); *ebits = 1u; *evalue = (n - 11u) & 1u; }
    else if (n < 35u) { *k = 12u + ((n - 19u) >> 2); *ebits = 2u; *evalue = (n - 19u) & 3u; }
    else { *k = 16u + ((n - 35u) >> 3); *ebits = 3u; *evalue = (n - 35u) & 7u; }
}
__host__ __device__ inline uint32_t length_symbol_extra_bits(uint32_t sym) {   // extra bits + the 1-bit distance code of a match symbol
    if (sym < 257u) return 0u;
    const uint32_t k = sym - 257u;
    return (k < 8u ? 0u : k < 12u ? 1u : k < 16u ? 2u : k < 20u ? 3u : k < 24u ? 4u : k < 28u ? 5u : 0u) + 1u;
}

// One lane's piece of the block as tokens: f.literal(byte) / f.match(length).  `same` bit i: byte i equals the byte before it.
template <class F>
__device__ __forceinline__ void walk_piece(const uint8_t* piece, uint32_t prev, uint32_t count, F& f) {
    unsigned long long same = 0ull;
    uint32_t p = prev;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(piece);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t v = w[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t b = (v >> (8 * j)) & 255u;
            if (b == p) same |= 1ull << (4 * q + j);
            p = b;
        }
    }
    if (count < 64u) same &= (1ull << count) - 1ull;
    uint32_t i = 0u;
    while (i < count) {
        if ((same >> i) & 1ull) {
            const unsigned long long rest = ~(same >> i);
            uint32_t n = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
            if (n > count - i) n = count - i;
            if (n >= 3u) { f.match(n); i += n; continue; }
        }
        f.literal(piece[i]);
        ++i;
    }
}

struct HistTokens {
    uint32_t* hist;
    __device__ __forceinline__ void literal(uint32_t b) { atomicAdd(&hist[b], 1u); }
    __device__ __forceinline__ void match(uint32_t n) { uint32_t k, e, v; length_code(n, &k, &e, &v); atomicAdd(&hist[257u + k], 1u); }
};

__device__ __forceinline__ uint32_t block_length(const PngLayout& L, uint32_t block) {
    const unsigned long long left = L.N - (unsigned long long)block * kDefBlock;
    return left < kDefBlock ? (uint32_t)left : kDefBlock;
}

__global__ void __launch_bounds__(256) png_hist_kernel(PngBatch B) {
    if ((int)blockIdx.y >= B.n) return;
    const PngJob& J = B.job[blockIdx.y];
    if (blockIdx.x >= J.blocks) return;
    __shared__ __attribute__((aligned(16))) uint8_t s_stream[256 * kDefPitch];
    __shared__ uint32_t s_hist[4][kDefSyms];
    const uint32_t block = blockIdx.x, t = threadIdx.x;
    const uint32_t blen = block_length(J.L, block);
    for (uint32_t i = t; i < 4u * kDefSyms; i += 256u) (&s_hist[0][0])[i] = 0u;
    load_block(J.stream, block, s_stream);
    __syncthreads();
    const uint32_t prev = piece_prev(J.stream, block, t, s_stream);
    const uint32_t first = t * kDefPiece;
    if (first < blen) {
        HistTokens f = {s_hist[t >> 6]};
        walk_piece(s_stream + t * kDefPitch, prev, blen - first < kDefPiece ? blen - first : kDefPiece, f);
    }
    __syncthreads();
    for (uint32_t i = t; i < kDefSyms; i += 256u)
        J.hist[(size_t)block * kDefSyms + i] = s_hist[0][i] + s_hist[1][i] + s_hist[2][i] + s_hist[3][i] + (i == 256u ? 1u : 0u);   // + the block's end-of-block
}

__device__ __forceinline__ uint32_t bit_reverse(uint32_t v, uint32_t n) { return __brev(v) >> (32u - n); }

// ONE workgroup per image: the Huffman code of the image's tokens and the dynamic blocks' common header.
__global__ void __launch_bounds__(256) png_table_kernel(PngBatch B) {
    if ((int)blockIdx.x >= B.n) return;
    const PngJob& J = B.job[blockIdx.x];
    const uint32_t blocks = J.blocks;
    __shared__ unsigned long long s_key[288];     // count << 16 | symbol per symbol (~0: unused)
    __shared__ unsigned long long s_sorted[288];  // ... of the used symbols, ascending
    __shared__ uint32_t s_weight[kDefSyms];       // internal nodes of the Huffman tree, in creation order
    __shared__ uint16_t s_parent_leaf[kDefSyms], s_parent_node[kDefSyms];
    __shared__ uint8_t s_len[kDefSyms];           // by symbol
    __shared__ uint8_t s_depth[kDefSyms];         // by sorted position
    __shared__ uint32_t s_code[kDefSyms];
    __shared__ uint32_t s_count_of[17], s_next_code[17];
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_header[(kDefHeaderBytes + 3u) / 4u + 1u];
    const uint32_t t = threadIdx.x;
    if (t == 0) s_n = 0u;
    for (uint32_t s = t; s < 288u; s += 256u) {
        unsigned long long key = ~0ull;
        if (s < 286u) {
            uint32_t c = 0u;
#pragma unroll 16
            for (uint32_t b = 0; b < blocks; ++b) c += J.hist[(size_t)b * kDefSyms + s];
            if (c) key = ((unsigned long long)c << 16) | s;
        }
        s_key[s] = key;
    }
    for (uint32_t s = t; s < kDefSyms; s += 256u) { s_len[s] = 0; s_code[s] = 0u; s_depth[s] = 0; }
    for (uint32_t s = t; s < sizeof(s_header) / 4u; s += 256u) s_header[s] = 0u;
    if (t < 17u) s_count_of[t] = 0u;
    __syncthreads();
    // sort by rank: a used symbol's place is the number of smaller keys (the keys are distinct: the symbol is their low half); 286 reads of
    // LDS per lane and two barriers instead of the 45 compare-exchange steps of a bitonic network
    for (uint32_t q = t; q < 286u; q += 256u) {
        const unsigned long long key = s_key[q];
        if (key != ~0ull) {
            uint32_t r = 0u;
            for (uint32_t o = 0; o < 286u; ++o) r += s_key[o] < key ? 1u : 0u;
            s_sorted[r] = key;
            atomicAdd(&s_n, 1u);
        }
    }
    __syncthreads();
    if (t == 0) {
        const uint32_t n = s_n;
        // Huffman's algorithm on sorted leaves with two queues (leaves by count, internal nodes in creation order).  The heads of both
        // queues live in registers and the next head is fetched while the current one is used: the merge never waits on LDS twice in a row.
        if (n >= 2u) {
            const uint32_t kInf = 0xFFFFFFFFu;
            uint32_t li = 0u, ni = 0u;
            uint32_t leaf_w = (uint32_t)(s_sorted[0] >> 16), leaf_next = n > 1u ? (uint32_t)(s_sorted[1] >> 16) : kInf;
            uint32_t node_w = kInf, node_next = kInf;          // weights of nodes ni, ni + 1 (kInf: not made yet)
            for (uint32_t k = 0; k + 1u < n; ++k) {
                uint32_t w = 0u;
#pragma unroll
                for (int pick = 0; pick < 2; ++pick) {
                    if (leaf_w <= node_w) {                     // (a tie takes the leaf: shallower trees)
                        w += leaf_w;
                        s_parent_leaf[li] = (uint16_t)k;
                        ++li;
                        leaf_w = leaf_next;
                        leaf_next = li + 1u < n ? (uint32_t)(s_sorted[li + 1u] >> 16) : kInf;
                    } else {
                        w += node_w;
                        s_parent_node[ni] = (uint16_t)k;
                        ++ni;
                        node_w = node_next;
                        node_next = ni + 1u < k ? s_weight[ni + 1u] : kInf;     // nodes < k exist; k itself is forwarded below
                    }
                }
                s_weight[k] = w;
                if (ni == k) node_w = w;                        // the queue was empty: the new node is its head
                else if (ni + 1u == k) node_next = w;           // ... or the one behind the head
            }
        }
    }
    __syncthreads();
    const uint32_t n = s_n;
    // every leaf walks up to the root (node n - 2): its depth, clamped to 15
    for (uint32_t j = t; j < n; j += 256u) {
        uint32_t d = 1u;
        if (n >= 2u) {
            uint32_t p = s_parent_leaf[j];
            while (p != n - 2u) { p = s_parent_node[p]; ++d; }
        }
        if (d > 15u) d = 15u;
        s_depth[j] = (uint8_t)d;
        atomicAdd(&s_count_of[d], 1u);
    }
    __syncthreads();
    if (t == 0) {
        // length limit (zlib trees.c gen_bitlen): while the code is over-subscribed, split a shorter code and drop one 15-bit code
        for (;;) {
            unsigned long long kraft = 0ull;
            for (uint32_t d = 1; d <= 15u; ++d) kraft += (unsigned long long)s_count_of[d] << (15u - d);
            if (kraft <= (1ull << 15) || n < 2u) break;
            uint32_t bits = 14u;
            while (s_count_of[bits] == 0u) --bits;
            --s_count_of[bits]; s_count_of[bits + 1u] += 2u; --s_count_of[15];
        }
        uint32_t code = 0u;
        s_count_of[0] = 0u;
        for (uint32_t d = 1; d <= 15u; ++d) { code = (code + s_count_of[d - 1u]) << 1; s_next_code[d] = code; }
        // the fixed part of a dynamic block's header: BFINAL 0, BTYPE 10, HLIT 29, HDIST 0, HCLEN 15, the code-length code: lengths 0 .. 15 as 4-bit codes
        uint32_t bit = 0u;
        auto put = [&](uint32_t v, uint32_t nb) { s_header[bit >> 5] |= v << (bit & 31u); if ((bit & 31u) + nb > 32u) s_header[(bit >> 5) + 1u] |= v >> (32u - (bit & 31u)); bit += nb; };
        put(0u, 1u); put(2u, 2u); put(29u, 5u); put(0u, 5u); put(15u, 4u);
        const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int q = 0; q < 19; ++q) put(order[q] >= 16 ? 0u : 4u, 3u);
    }
    __syncthreads();
    // the rarest symbols get the longest codes: sorted position j (ascending count) has the length d with
    // sum_{e > d} count_of[e] <= j < sum_{e >= d} count_of[e]
    for (uint32_t j = t; j < n; j += 256u) {
        uint32_t before = 0u, d = 15u;
        for (; d >= 1u; --d) { if (j < before + s_count_of[d]) break; before += s_count_of[d]; }
        s_len[(uint32_t)(s_sorted[j] & 0xFFFFull)] = (uint8_t)d;
    }
    __syncthreads();
    // canonical codes (RFC 1951 3.2.2): a symbol's code is its length's first code plus the number of smaller symbols of that length
    for (uint32_t s = t; s < 286u; s += 256u) {
        const uint32_t len = s_len[s];
        if (len) {
            uint32_t rank = 0u;
            for (uint32_t q = 0; q < s; ++q) rank += s_len[q] == len ? 1u : 0u;
            s_code[s] = bit_reverse(s_next_code[len] + rank, len);
        }
    }
    // the 286 literal / length code lengths and the one distance code length (1), each as its 4-bit code, most significant bit first
    for (uint32_t s = t; s < 287u; s += 256u) {
        const uint32_t v = bit_reverse(s < 286u ? s_len[s] : 1u, 4u), bit = 74u + 4u * s;
        atomicOr(&s_header[bit >> 5], v << (bit & 31u));
        if ((bit & 31u) > 28u) atomicOr(&s_header[(bit >> 5) + 1u], v >> (32u - (bit & 31u)));
    }
    __syncthreads();
    for (uint32_t s = t; s < kDefSyms; s += 256u) J.dyn->table[s] = s_code[s] | ((uint32_t)s_len[s] << 16);
    for (uint32_t s = t; s < sizeof(s_header) / 4u; s += 256u) J.dyn->header_words[s] = s_header[s];
}

// A wave per block: the block's size in bytes -- its 286 token counts times their code lengths; the cheaper of dynamic and stored (bit 31).
__global__ void __launch_bounds__(256) png_size_kernel(PngBatch B) {
    if ((int)blockIdx.y >= B.n) return;
    const PngJob& J = B.job[blockIdx.y];
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (b >= J.blocks) return;
    uint32_t bits = 0u;
#pragma unroll
    for (uint32_t q = 0; q < 5u; ++q) {
        const uint32_t sym = q * 64u + lane;
        if (sym < 286u) bits += J.hist[(size_t)b * kDefSyms + sym] * ((J.dyn->table[sym] >> 16) + length_symbol_extra_bits(sym));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) bits += __shfl_xor(bits, d);
    if (lane == 0u) {
        const uint32_t blen = block_length(J.L, b);
        const unsigned long long all = (unsigned long long)kDefHeaderBits + bits;
        const unsigned long long dyn_bytes = b + 1u == J.blocks ? (all + 7ull) >> 3 : ((all + 3ull + 7ull) >> 3) + 4ull;
        J.block_off[b] = dyn_bytes <= 5ull + blen ? (uint32_t)dyn_bytes : ((5u + blen) | (1u << 31));
    }
}

struct SizeTokens {
    const uint32_t* table;
    uint32_t bits;
    __device__ __forceinline__ void literal(uint32_t b) { bits += table[b] >> 16; }
    __device__ __forceinline__ void match(uint32_t n) { uint32_t k, e, v; length_code(n, &k, &e, &v); bits += (table[257u + k] >> 16) + e + 1u; }
};
struct EmitTokens {
    const uint32_t* table;
    uint32_t* out;                 // LDS words, zeroed
    unsigned long long acc;
    uint32_t nacc, word;
    __device__ __forceinline__ void put(uint32_t v, uint32_t nb) {
        acc |= (unsigned long long)v << nacc;
        nacc += nb;
        if (nacc >= 32u) { atomicOr(&out[word++], (uint32_t)acc); acc >>= 32; nacc -= 32u; }
    }
    __device__ __forceinline__ void literal(uint32_t b) { const uint32_t e = table[b]; put(e & 0xFFFFu, e >> 16); }
    __device__ __forceinline__ void match(uint32_t n) {
        uint32_t k, eb, ev;
        length_code(n, &k, &eb, &ev);
        const uint32_t e = table[257u + k];
        put(e & 0xFFFFu, e >> 16);
        put(ev, eb + 1u);          // the extra bits, then the one distance code: a single 0 bit
    }
    __device__ __forceinline__ void flush() { if (nacc) atomicOr(&out[word], (uint32_t)acc); }
};

__global__ void __launch_bounds__(256) png_deflate_kernel(PngBatch B, PngTables T) {
    if ((int)blockIdx.y >= B.n) return;
    const PngJob& J = B.job[blockIdx.y];
    if (blockIdx.x >= J.blocks) return;
    const PngLayout& L = J.L;
    uint8_t* __restrict__ out = J.out;
    __shared__ __attribute__((aligned(16))) uint8_t s_stream[256 * kDefPitch];
    __shared__ uint32_t s_out[(kDefBlock + 5u + 3u) / 4u + 3u];
    __shared__ uint32_t s_table[kDefSyms];
    __shared__ uint32_t s_wave[4];
    const uint32_t block = blockIdx.x, t = threadIdx.x, blocks = J.blocks;
    const uint32_t blen = block_length(L, block);
    const bool last = block + 1u == blocks;
    // the block's place: the sizes of the blocks in front of it (png_size_kernel), added up here.  The FIRST block's workgroup (nothing in
    // front of it, and the first to start) adds up all of them instead: the file's length, for the head of the file and the checksum kernels.
    uint32_t before_bytes = 0u;
    const uint32_t upto = block == 0u ? blocks : block;
    for (uint32_t b = t; b < upto; b += 256u) before_bytes += J.block_off[b] & 0x7FFFFFFFu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before_bytes += __shfl_xor(before_bytes, d);
    if ((t & 63u) == 0u) s_wave[t >> 6] = before_bytes;
    __syncthreads();
    const uint32_t summed = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    const uint32_t off = block == 0u ? 0u : summed;
    const uint32_t mine = J.block_off[block];
    const bool stored = (mine >> 31) != 0u;
    const uint32_t bytes = mine & 0x7FFFFFFFu;
    const unsigned long long g0 = L.data_at + 2ull + off;                            // the block's first byte in the file
    if (block == 0u && t == 64u) {   // (a lane of the second wave: the first has the end-of-block duty below)
        const uint32_t total = summed;
        const unsigned long long data_len = 2ull + total + 4ull, file_len = L.data_at + data_len + 4ull + 12ull;
        J.dyn->data_len = data_len;
        J.dyn->file_len = file_len;
        // (x^(8 n) for n < 64: six modular products at most.  The pre-conditioning term, x^(8 (4 + data_len)), is twenty of them in a row on one
        // lane -- 35 us that every other wave of this workgroup waited for at the next barrier -- and is formed by the finishing kernel
        // instead, as a product tree over a wave.)
        J.dyn->crc_tail_shift = crc_x2nmodp(T.x2n, (L.data_at + data_len) & 63ull, 3u);
        if (J.out_len) *J.out_len = file_len;
        for (uint32_t i = 0; i < (uint32_t)L.data_at; ++i) out[i] = L.head[i];
        out[33] = (uint8_t)(data_len >> 24); out[34] = (uint8_t)(data_len >> 16); out[35] = (uint8_t)(data_len >> 8); out[36] = (uint8_t)data_len;
        out[L.data_at] = 0x78u; out[L.data_at + 1ull] = 0x01u;
    }
    if (stored) {
        const uint8_t* src = J.stream + (size_t)block * kDefBlock;
        if (t < 5u) out[g0 + t] = (uint8_t)(t == 0u ? (last ? 1u : 0u) : t == 1u ? (blen & 255u) : t == 2u ? (blen >> 8) : t == 3u ? (~blen & 255u) : ((~blen >> 8) & 255u));
        for (uint32_t i = t; i < blen; i += 256u) out[g0 + 5ull + i] = src[i];
        return;
    }
    for (uint32_t i = t; i < kDefSyms; i += 256u) s_table[i] = J.dyn->table[i];
    for (uint32_t i = t; i < sizeof(s_out) / 4u; i += 256u) s_out[i] = 0u;
    load_block(J.stream, block, s_stream);
    __syncthreads();
    const uint32_t prev = piece_prev(J.stream, block, t, s_stream);
    // LDS image of the block, phased like the file: LDS byte (g0 & 3) is file byte g0, so whole LDS words are whole file words
    const uint32_t phase = (uint32_t)(g0 & 3ull);
    const uint32_t first = t * kDefPiece;
    const uint32_t count = first < blen ? (blen - first < kDefPiece ? blen - first : kDefPiece) : 0u;
    SizeTokens sz = {s_table, 0u};
    if (count) walk_piece(s_stream + t * kDefPitch, prev, count, sz);
    // exclusive prefix sum of the lanes' bit counts
    uint32_t incl = sz.bits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if ((t & 63u) >= (uint32_t)d) incl += v; }
    if ((t & 63u) == 63u) s_wave[t >> 6] = incl;
    // the block header's bytes, meanwhile (byte granular: the block starts on a byte)
    if (t < kDefHeaderBytes) {
        uint32_t v = reinterpret_cast<const uint8_t*>(J.dyn->header_words)[t];
        if (t == 0u && last) v |= 1u;     // BFINAL
        atomicOr(&s_out[(phase + t) >> 2], v << (8u * ((phase + t) & 3u)));
    }
    __syncthreads();
    uint32_t before = 0u;
    for (uint32_t w = 0; w < (t >> 6); ++w) before += s_wave[w];
    const uint32_t total_bits = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    const uint32_t my_bit = phase * 8u + kDefHeaderBits + before + incl - sz.bits;
    if (count) {
        EmitTokens em = {s_table, s_out, 0ull, my_bit & 31u, my_bit >> 5};
        walk_piece(s_stream + t * kDefPitch, prev, count, em);
        em.flush();
    }
    if (t == 0) {   // end of block; for a non-final block an empty stored block: 3 zero bits, padding to a byte, 00 00 FF FF
        const uint32_t e = s_table[256];
        const uint32_t at = phase * 8u + kDefHeaderBits + total_bits;
        EmitTokens em = {s_table, s_out, 0ull, at & 31u, at >> 5};
        em.put(e & 0xFFFFu, e >> 16);
        em.flush();
        if (!last) {
            const uint32_t nlen = ((at + (e >> 16) + 3u + 7u) & ~7u) + 16u;     // LEN (16 zero bits) starts on the next byte boundary behind the 3 header bits; NLEN = FFFF
            atomicOr(&s_out[nlen >> 5], 0xFFFFu << (nlen & 31u));
            if ((nlen & 31u) > 16u) atomicOr(&s_out[(nlen >> 5) + 1u], 0xFFFFu >> (32u - (nlen & 31u)));
        }
    }
    __syncthreads();
    // copy out: file bytes [g0, g0 + bytes) are LDS bytes [phase, phase + bytes); whole words where the file word is wholly this block's
    const uint8_t* ob = reinterpret_cast<const uint8_t*>(s_out);
    const uint32_t lo = phase, hi = phase + bytes;                       // LDS byte range
    const uint32_t wlo = (lo + 3u) >> 2, whi = hi >> 2;                  // whole words [wlo, whi)
    uint8_t* fbase = out + (g0 - phase);                                 // file address of LDS byte 0 (4-byte aligned)
    for (uint32_t w = wlo + t; w < whi; w += 256u) reinterpret_cast<uint32_t*>(fbase)[w] = s_out[w];
    if (t < 4u) {
        const uint32_t i = lo + t;
        if (i < (wlo << 2) && i < hi) fbase[i] = ob[i];
        const uint32_t j = (whi << 2) + t;
        if (j >= lo && j < hi && whi >= wlo) fbase[j] = ob[j];
    }
}

// CRC-32 and the last bytes of a deflate-compressed file: as png_crc_kernel / png_finish_kernel, with the lengths read from the device
__global__ void __launch_bounds__(256) png_crc_dynamic_kernel(PngBatch B, PngTables T) {
    if ((int)blockIdx.y >= B.n) return;
    const PngJob& J = B.job[blockIdx.y];
    if (blockIdx.x >= J.crc_groups) return;
    const PngLayout& L = J.L;
    const uint8_t* __restrict__ file = J.out;
    __shared__ uint32_t s_t[4][256];
    __shared__ uint32_t s_shift[3][256];
    __shared__ uint32_t s_crc[4];
    {
        const int i = threadIdx.x;
        const uint32_t t0 = T.byte[i];
        const uint32_t t1 = (t0 >> 8) ^ T.byte[t0 & 0xFFu];
        const uint32_t t2 = (t1 >> 8) ^ T.byte[t1 & 0xFFu];
        s_t[0][i] = t0; s_t[1][i] = t1; s_t[2][i] = t2; s_t[3][i] = (t2 >> 8) ^ T.byte[t2 & 0xFFu];
        s_shift[0][i] = g_crc_shift64[0][i]; s_shift[1][i] = g_crc_shift64[1][i]; s_shift[2][i] = g_crc_shift64[2][i];
    }
    __syncthreads();
    const unsigned long long data_len = J.dyn->data_len;
    const uint32_t tail_shift = J.dyn->crc_tail_shift;
    // "IDAT" ... the last deflate byte; the four Adler bytes that end the message are folded in by the finish kernel, which writes them
    const unsigned long long crc_from = L.data_at - 4ull, crc_end = L.data_at + data_len, body_end = crc_end - 4ull;
    const unsigned long long lo0 = ((unsigned long long)blockIdx.x * 256ull + threadIdx.x) * kCrcChunk;
    uint32_t crc = 0u;
    if (lo0 < body_end && lo0 + kCrcChunk > crc_from) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lo0 + 16u * k < body_end ? *reinterpret_cast<const uint4*>(file + lo0 + 16u * k) : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t w[16] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
        const unsigned long long lo = lo0 < crc_from ? crc_from : lo0, hi = lo0 + kCrcChunk < body_end ? lo0 + kCrcChunk : body_end;
        if (lo == lo0 && hi == lo0 + kCrcChunk) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t x = crc ^ w[k];
                crc = s_t[3][x & 0xFFu] ^ s_t[2][(x >> 8) & 0xFFu] ^ s_t[1][(x >> 16) & 0xFFu] ^ s_t[0][x >> 24];
            }
        } else {
            for (unsigned long long f = lo; f < hi; ++f) {
                const uint32_t b = (w[(f - lo0) >> 2] >> (8 * ((f - lo0) & 3ull))) & 0xFFu;
                crc = s_t[0][(crc ^ b) & 0xFFu] ^ (crc >> 8);
            }
        }
        if (crc != 0u) {   // crc_end - hi bytes lie behind this piece: a multiple of 64 plus (crc_end mod 64) when hi is a chunk end, anything for the last piece
            const unsigned long long behind = crc_end - hi;
            if ((hi & 63ull) == 0ull) {
                const unsigned long long m = behind >> 6;
                crc = crc_multmodp(tail_shift, crc);
                if (m & 0xFFull) crc = crc_multmodp(s_shift[0][m & 0xFFull], crc);
                if ((m >> 8) & 0xFFull) crc = crc_multmodp(s_shift[1][(m >> 8) & 0xFFull], crc);
                if (m >> 16) crc = crc_multmodp(s_shift[2][(m >> 16) & 0xFFull], crc);
            } else {
                crc = crc_multmodp(crc_x2nmodp(T.x2n, behind, 3u), crc);
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) crc ^= (uint32_t)__shfl_xor((int)crc, d);
    if ((threadIdx.x & 63) == 0) s_crc[threadIdx.x >> 6] = crc;
    __syncthreads();
    if (threadIdx.x == 0) J.crc_partials[blockIdx.x] = s_crc[0] ^ s_crc[1] ^ s_crc[2] ^ s_crc[3];
}

__global__ void __launch_bounds__(64) png_finish_dynamic_kernel(PngBatch B, PngTables T) {
    if ((int)blockIdx.x >= B.n) return;
    const PngJob& J = B.job[blockIdx.x];
    const PngLayout& L = J.L;
    uint8_t* __restrict__ out = J.out;
    // the CRC's pre-conditioning as one more linear term: 0xFFFFFFFF x^(8 n) mod P, n = 4 + data_len = 64 m + r.  x^(8 64 m) is a product of
    // three entries of the shift tables the CRC kernel uses, x^(8 r) six squarings' worth at most: nine modular products on one lane
    // (twenty in a row cost 35 us where this first lived; a product tree over the wave was no faster: every lane runs the worst case)
    uint32_t init_term = 0u;
    if (threadIdx.x == 0) {
        const unsigned long long n = 4ull + J.dyn->data_len, m = n >> 6;
        uint32_t f = crc_x2nmodp(T.x2n, n & 63ull, 3u);
        if (m & 0xFFull) f = crc_multmodp(g_crc_shift64[0][m & 0xFFull], f);
        if ((m >> 8) & 0xFFull) f = crc_multmodp(g_crc_shift64[1][(m >> 8) & 0xFFull], f);
        if ((m >> 16) & 0xFFull) f = crc_multmodp(g_crc_shift64[2][(m >> 16) & 0xFFull], f);
        init_term = crc_multmodp(f, 0xFFFFFFFFu);
    }
    unsigned long long a1 = 0ull, a2 = 0ull;
    uint32_t crc_sum = 0u;
    for (uint32_t i = threadIdx.x; i < J.filter_groups; i += 64u) { a1 += J.adler[2 * (size_t)i]; a2 += J.adler[2 * (size_t)i + 1]; }
    for (uint32_t i = threadIdx.x; i < J.crc_groups; i += 64u) crc_sum ^= J.crc_partials[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a1 += __shfl_xor(a1, d);
        a2 += __shfl_xor(a2, d);
        crc_sum ^= (uint32_t)__shfl_xor((int)crc_sum, d);
    }
    if (threadIdx.x != 0) return;
    const uint32_t s1 = (uint32_t)((1ull + a1) % 65521ull);
    const uint32_t s2 = (uint32_t)((L.N % 65521ull + a2) % 65521ull);
    const uint32_t adler = (s2 << 16) | s1;
    const unsigned long long adler_at = L.data_at + J.dyn->data_len - 4ull;
    uint32_t crc_a = 0u;
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = (adler >> (24 - 8 * k)) & 0xFFu;
        out[adler_at + k] = (uint8_t)b;
        crc_a ^= b;
        for (int i = 0; i < 8; ++i) crc_a = (crc_a & 1u) ? (crc_a >> 1) ^ kCrcPoly : crc_a >> 1;
    }
    const uint32_t crc = (crc_sum ^ crc_a ^ init_term) ^ 0xFFFFFFFFu;
    for (int k = 0; k < 4; ++k) out[adler_at + 4 + k] = (uint8_t)(crc >> (24 - 8 * k));
    for (int k = 0; k < 12; ++k) out[adler_at + 8 + k] = L.tail[k];
}

// ================================================================================================
// PIL's Image.resize, bit for bit (blender/blend_all.py:21-28: downsample_image = Image.fromarray(a).resize(new_size, BILINEAR) for
// the RGBA8 layers, resize(new_size, NEAREST) for the float depth maps; called on every Blender layer of every frame, :217-234).
//
// BILINEAR on an RGBA image is three steps in Pillow (src/PIL/Image.py resize; src/libImaging/Convert.c, Resample.c):
//   1. RGBA -> RGBa: colour channels premultiplied, MULDIV255(c, a) = (t = c a + 128, ((t >> 8) + t) >> 8);
//   2. a separable resample in 8-bit fixed point, horizontal pass first, each pass rounding to 8 bits: the triangle filter's
//      support is stretched by the down-scale factor (an area-weighted average, not a 2 x 2 lookup), the weights of an output
//      pixel are normalised to sum 1 in double and converted to integers with 22 fraction bits ((int)(0.5 + w 2^22)), a pixel is
//      clip8((2^21 + sum k_i p_i) >> 22);
//   3. RGBa -> RGBA: c = min(255, 255 c / a) (integer division) unless a is 0 or 255.
// An image that already has the target size is copied (no premultiply round trip); a pass whose size does not change is skipped.
// The weight tables depend only on (input size, output size): computed on the host in the doubles Pillow uses, cached per device.
//
// NEAREST on a mode "F" image is an affine scale with the source coordinate ACCUMULATED in double (Geometry.c
// ImagingScaleAffine: xo = a / 2, then xo += a per output pixel, index = (int)xo): the index tables are built on the host the same way.
// ================================================================================================
constexpr int kResampleBits = 32 - 8 - 2;

struct ResampleTable {
    int dev, in, out, kind;   // kind 0: bilinear weights, 1: nearest indices
    int ksize;
    int* bounds;              // [out][2] (first input index, count); nearest: [out] indices
    int* coef;                // [out][ksize]
};
std::mutex g_table_mutex;
std::vector<ResampleTable> g_tables;

double triangle(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// Resample.c: precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0) over the whole input (box = image)
void bilinear_weights(int in_size, int out_size, std::vector<int>* bounds, std::vector<int>* coef, int* ksize_out) {
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds->assign((size_t)out_size * 2, 0);
    coef->assign((size_t)out_size * ksize, 0);
    std::vector<double> k((size_t)ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        const double ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = triangle((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            (*coef)[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << kResampleBits)) : (int)(0.5 + k[x] * (1 << kResampleBits));
        }
        (*bounds)[2 * (size_t)xx] = xmin;
        (*bounds)[2 * (size_t)xx + 1] = xmax;
    }
    *ksize_out = ksize;
}

// Geometry.c ImagingScaleAffine, nearest: the source index of every output pixel (-1: outside)
void nearest_indices(int in_size, int out_size, std::vector<int>* index) {
    const double a = (double)in_size / out_size;
    double xo = a * 0.5;
    index->assign((size_t)out_size, -1);
    for (int x = 0; x < out_size; ++x) {
        const int xin = xo < 0.0 ? -1 : (int)xo;
        (*index)[x] = (xin >= 0 && xin < in_size) ? xin : -1;
        xo += a;
    }
}

hipError_t resample_table(int in_size, int out_size, int kind, ResampleTable* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_table_mutex);
    for (const ResampleTable& t : g_tables)
        if (t.dev == dev && t.in == in_size && t.out == out_size && t.kind == kind) { *out = t; return hipSuccess; }
    std::vector<int> bounds, coef;
    ResampleTable t = {dev, in_size, out_size, kind, 0, nullptr, nullptr};
    if (kind == 0) bilinear_weights(in_size, out_size, &bounds, &coef, &t.ksize);
    else nearest_indices(in_size, out_size, &bounds);
    if ((e = hipMalloc((void**)&t.bounds, bounds.size() * sizeof(int))) != hipSuccess) return e;
    if ((e = hipMemcpy(t.bounds, bounds.data(), bounds.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if (!coef.empty()) {
        if ((e = hipMalloc((void**)&t.coef, coef.size() * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMemcpy(t.coef, coef.data(), coef.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
    g_tables.push_back(t);   // (a handful of sizes per process; never freed)
    *out = t;
    return hipSuccess;
}

__device__ __forceinline__ uint32_t muldiv255(uint32_t c, uint32_t a) {
    const uint32_t t = c * a + 128u;
    return ((t >> 8) + t) >> 8;
}
__device__ __forceinline__ int clip8(int v) {
    v >>= kResampleBits;   // arithmetic shift: floor, as the reference's lookup table is indexed
    return v < 0 ? 0 : v > 255 ? 255 : v;
}

// One lane = one output pixel of one pass.  kHorizontal: out[row][xx] from in[row][xmin .. xmin + n); else out[yy][x] from
// in[ymin .. ymin + n)[x].  kPremultiply: the input is straight RGBA (the first pass of a call); kUnpremultiply: the output is
// converted back (the last pass).
template <bool kHorizontal, bool kPremultiply, bool kUnpremultiply>
__global__ void __launch_bounds__(256) resample_rgba8_kernel(const uchar4* __restrict__ in, int in_w, uchar4* __restrict__ out, int out_w, int out_h,
                                                            const int* __restrict__ bounds, const int* __restrict__ coef, int ksize) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= out_w || y >= out_h) return;
    const int o = kHorizontal ? x : y;
    const int first = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = coef + (size_t)o * ksize;
    int s0 = 1 << (kResampleBits - 1), s1 = s0, s2 = s0, s3 = s0;
    for (int i = 0; i < n; ++i) {
        const uchar4 p = kHorizontal ? in[(size_t)y * in_w + first + i] : in[(size_t)(first + i) * in_w + x];
        uint32_t r = p.x, g = p.y, b = p.z;
        const uint32_t a = p.w;
        if (kPremultiply) { r = muldiv255(r, a); g = muldiv255(g, a); b = muldiv255(b, a); }
        const int w = k[i];
        s0 += (int)r * w; s1 += (int)g * w; s2 += (int)b * w; s3 += (int)a * w;
    }
    int r = clip8(s0), g = clip8(s1), b = clip8(s2);
    const int a = clip8(s3);
    if (kUnpremultiply && a != 255 && a != 0) {
        r = min(255, 255 * r / a); g = min(255, 255 * g / a); b = min(255, 255 * b / a);
    }
    out[(size_t)y * out_w + x] = make_uchar4((unsigned char)r, (unsigned char)g, (unsigned char)b, (unsigned char)a);
}

__global__ void __launch_bounds__(256) nearest_f32_kernel(const float* __restrict__ in, int in_w, float* __restrict__ out, int out_w, int out_h,
                                                         const int* __restrict__ xi, const int* __restrict__ yi) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= out_w || y >= out_h) return;
    const int sx = xi[x], sy = yi[y];
    if (sx >= 0 && sy >= 0) out[(size_t)y * out_w + x] = in[(size_t)sy * in_w + sx];   // (never outside for a whole-image resize)
}

} // namespace

hipError_t launch_resize_rgba8_bilinear(const uint8_t* src, int src_w, int src_h, uint8_t* dst, int dst_w, int dst_h, uint8_t* tmp,
                                        hipStream_t stream) {
    const uchar4* in = reinterpret_cast<const uchar4*>(src);
    uchar4* out = reinterpret_cast<uchar4*>(dst);
    if (src_w == dst_w && src_h == dst_h)   // Image.resize returns a copy: no premultiply round trip
        return hipMemcpyAsync(dst, src, (size_t)src_w * src_h * 4, hipMemcpyDeviceToDevice, stream);
    const bool horizontal = src_w != dst_w, vertical = src_h != dst_h;
    ResampleTable tx = {}, ty = {};
    hipError_t e;
    if (horizontal && (e = resample_table(src_w, dst_w, 0, &tx)) != hipSuccess) return e;
    if (vertical && (e = resample_table(src_h, dst_h, 0, &ty)) != hipSuccess) return e;
    if (horizontal && vertical) {
        uchar4* mid = reinterpret_cast<uchar4*>(tmp);   // [src_h, dst_w]
        hipLaunchKernelGGL((resample_rgba8_kernel<true, true, false>), dim3((dst_w + 255) / 256, src_h), dim3(256), 0, stream, in, src_w, mid, dst_w,
                           src_h, tx.bounds, tx.coef, tx.ksize);
        hipLaunchKernelGGL((resample_rgba8_kernel<false, false, true>), dim3((dst_w + 255) / 256, dst_h), dim3(256), 0, stream, mid, dst_w, out, dst_w,
                           dst_h, ty.bounds, ty.coef, ty.ksize);
    } else if (horizontal) {
        hipLaunchKernelGGL((resample_rgba8_kernel<true, true, true>), dim3((dst_w + 255) / 256, src_h), dim3(256), 0, stream, in, src_w, out, dst_w,
                           src_h, tx.bounds, tx.coef, tx.ksize);
    } else {
        hipLaunchKernelGGL((resample_rgba8_kernel<false, true, true>), dim3((dst_w + 255) / 256, dst_h), dim3(256), 0, stream, in, src_w, out, dst_w,
                           dst_h, ty.bounds, ty.coef, ty.ksize);
    }
    return hipGetLastError();
}

hipError_t launch_resize_f32_nearest(const float* src, int src_w, int src_h, float* dst, int dst_w, int dst_h, hipStream_t stream) {
    if (src_w == dst_w && src_h == dst_h) return hipMemcpyAsync(dst, src, (size_t)src_w * src_h * 4, hipMemcpyDeviceToDevice, stream);
    ResampleTable tx = {}, ty = {};
    hipError_t e;
    if ((e = resample_table(src_w, dst_w, 1, &tx)) != hipSuccess) return e;
    if ((e = resample_table(src_h, dst_h, 1, &ty)) != hipSuccess) return e;
    hipLaunchKernelGGL(nearest_f32_kernel, dim3((dst_w + 255) / 256, dst_h), dim3(256), 0, stream, src, src_w, dst, dst_w, dst_h, tx.bounds, ty.bounds);
    return hipGetLastError();
}

namespace {
// The two 8-bit images the reference makes with numpy / OpenCV before cv2.imwrite (scene_representation.py:429-438), one lane per
// pixel: the turbo-coloured depth preview -- depth2img(depth, scale): uint8(clip(depth / scale, 0, 1) * 255) through the colour
// table -- and the normal map, uint8((n + 1) / 2 * 255): the same fp32 operations in the same order, truncation.
__global__ void __launch_bounds__(256) frame_previews_kernel(const float* __restrict__ depth, const float* __restrict__ normal /*[H,W,3]*/,
                                                            float depth_scale, const uint8_t* __restrict__ lut /*[256,3]*/, size_t n_pixels,
                                                            uint8_t* __restrict__ depth_rgb, uint8_t* __restrict__ normal_rgb,
                                                            float* __restrict__ depth_copy /*the .npy plane: the depth map as it is*/) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    const float depth_i = depth[i];
    depth_copy[i] = depth_i;
    const float d = fminf(fmaxf(depth_i / depth_scale, 0.0f), 1.0f) * 255.0f;
    const uint32_t idx = (uint32_t)(int)d & 255u;
    depth_rgb[3 * i + 0] = lut[3 * idx + 0];
    depth_rgb[3 * i + 1] = lut[3 * idx + 1];
    depth_rgb[3 * i + 2] = lut[3 * idx + 2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (normal[3 * i + c] + 1.0f) / 2.0f * 255.0f;
        normal_rgb[3 * i + c] = (uint8_t)(int)fminf(fmaxf(v, 0.0f), 255.0f);
    }
}
} // namespace

hipError_t launch_png_encode_deflate_batch(int n, const uint8_t* const* pixels, const int* Ws, const int* Hs, const int* Cs, const int* planars,
                                           uint8_t* const* outs, uint8_t* const* scratches, unsigned long long* const* out_lens, hipStream_t stream);

// One frame's four files (gsr.h: gsr_frame_files): quantise, colour, encode, copy -- ten launches queued by ONE host call.
hipError_t launch_frame_files(const float* color, const float* alpha, const float* depth, const float* normal, float depth_scale,
                              const uint8_t* turbo_lut, int W, int H, uint8_t* png_rgba, uint8_t* png_depth, uint8_t* png_normal,
                              float* npy_plane, uint8_t* work, uint8_t* png_scratch, unsigned long long* png_lengths /*both null: stored PNGs; else
                              scratch for the three compressed files and [3] lengths, device*/, hipStream_t stream) {
    const size_t n = (size_t)W * H;
    uint8_t* rgba8 = work;                 // planar [4,H,W]
    uint8_t* depth_rgb = work + 4 * n;     // [H,W,3]
    uint8_t* normal_rgb = work + 7 * n;    // [H,W,3]
    hipError_t e = launch_pack_rgba8(color, alpha, rgba8, n, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(frame_previews_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, depth, normal, depth_scale, turbo_lut, n,
                       depth_rgb, normal_rgb, npy_plane);
    const size_t sizes[3] = {png_file_bytes(W, H, 4), png_file_bytes(W, H, 3), png_file_bytes(W, H, 3)};
    uint8_t* outs[3] = {png_rgba, png_depth, png_normal};
    const uint8_t* srcs[3] = {rgba8, depth_rgb, normal_rgb};
    for (int k = 0; k < 3; ++k)
        if (sizes[k] == 0) return hipErrorInvalidValue;
    if (png_lengths) {
        const int Ws[3] = {W, W, W}, Hs[3] = {H, H, H}, Cs[3] = {4, 3, 3}, planars[3] = {1, 0, 0};
        unsigned long long* lens[3] = {png_lengths, png_lengths + 1, png_lengths + 2};
        const size_t s4 = png_deflate_scratch_bytes(W, H, 4), s3 = png_deflate_scratch_bytes(W, H, 3);
        uint8_t* scratches[3] = {png_scratch, png_scratch + s4, png_scratch + s4 + s3};
        return launch_png_encode_deflate_batch(3, srcs, Ws, Hs, Cs, planars, outs, scratches, lens, stream);
    }
    for (int k = 0; k < 3; ++k)
        if ((e = launch_png_encode(srcs[k], W, H, k == 0 ? 4 : 3, k == 0 ? 1 : 0, outs[k], stream)) != hipSuccess) return e;
    return hipSuccess;
}

// ---- deflate-compressed files: sizes and the launch sequence ----
size_t png_deflate_max_bytes(int W, int H, int C) {
    PngLayout L;
    return png_layout(W, H, C, 0, &L) ? (size_t)deflate_scratch(L).file_max : 0;
}
size_t png_deflate_room_bytes(int W, int H, int C) {
    PngLayout L;
    return png_layout(W, H, C, 0, &L) ? deflate_scratch(L).out_room : 0;
}
size_t png_deflate_scratch_bytes(int W, int H, int C) {
    PngLayout L;
    return png_layout(W, H, C, 0, &L) ? deflate_scratch(L).total : 0;
}

size_t png_file_bytes(int W, int H, int C) {
    PngLayout L;
    return png_layout(W, H, C, 0, &L) ? (size_t)L.file_len : 0;
}

namespace {
struct PngScratch { size_t adler_at, crc_at, total; uint32_t enc_groups, crc_groups; };   // offsets from the start of `out`
PngScratch png_scratch(const PngLayout& L) {
    PngScratch p;
    p.enc_groups = (uint32_t)((((L.file_len + 15ull) / 16ull) + 255ull) / 256ull);
    p.crc_groups = (uint32_t)((((L.file_len + kCrcChunk - 1ull) / kCrcChunk) + 255ull) / 256ull);
    p.adler_at = ((size_t)L.file_len + 15) & ~size_t(15);
    p.crc_at = p.adler_at + (size_t)p.enc_groups * 16;
    p.total = p.crc_at + (((size_t)p.crc_groups * 4 + 15) & ~size_t(15));
    return p;
}
} // namespace

// Bytes `out` of launch_png_encode must hold: the file, then (from the next 16-byte boundary) the workgroups' partial checksums.
size_t png_room_bytes(int W, int H, int C) {
    PngLayout L;
    return png_layout(W, H, C, 0, &L) ? png_scratch(L).total : 0;
}

namespace {
std::mutex g_shift_mutex;
bool g_shift_ready[64] = {};
hipError_t ensure_shift_tables() {   // once per device: 3 KB of constants for png_crc_kernel
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_shift_mutex);
    if (g_shift_ready[dev]) return hipSuccess;
    static uint32_t host[3][256];
    const uint32_t* x2n = png_tables().x2n;
    for (int t = 0; t < 3; ++t) {
        const uint32_t step = crc_x2nmodp(x2n, 64ull << (8 * t), 3u);   // x^(8 * 64 * 256^t)
        uint32_t p = 1u << 31;                                           // x^0
        for (int i = 0; i < 256; ++i) { host[t][i] = p; p = crc_multmodp(step, p); }
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_crc_shift64), host, sizeof host);
    if (e == hipSuccess) g_shift_ready[dev] = true;
    return e;
}
} // namespace

// out: png_room_bytes(...) bytes, 16-byte aligned (the file image, then the kernels' partial checksums).  Three launches, no memset.
hipError_t launch_png_encode(const uint8_t* pixels, int W, int H, int C, int planar, uint8_t* out, hipStream_t stream) {
    PngLayout L;
    if (!png_layout(W, H, C, planar, &L)) return hipErrorInvalidValue;
    const hipError_t ready = ensure_shift_tables();
    if (ready != hipSuccess) return ready;
    const PngScratch p = png_scratch(L);
    unsigned long long* adler_partials = reinterpret_cast<unsigned long long*>(out + p.adler_at);
    uint32_t* crc_partials = reinterpret_cast<uint32_t*>(out + p.crc_at);
    hipLaunchKernelGGL(png_encode_kernel, dim3(p.enc_groups), dim3(256), 0, stream, L, pixels, out, adler_partials);
    hipLaunchKernelGGL(png_crc_kernel, dim3(p.crc_groups), dim3(256), 0, stream, L, png_tables(), out, crc_partials);
    hipLaunchKernelGGL(png_finish_kernel, dim3(1), dim3(64), 0, stream, L, out, adler_partials, p.enc_groups, crc_partials, p.crc_groups);
    return hipGetLastError();
}

// A batch of up to three images (a frame's three PNGs), every kernel launched once for all of them.  outs[i]: png_deflate_room_bytes(...)
// bytes (the file), scratches[i]: png_deflate_scratch_bytes(...) bytes, both 16-byte aligned; the files' lengths go to out_lens[i] (device
// memory; entries may be null).  Six launches per batch, no memset, no global atomics.
hipError_t launch_png_encode_deflate_batch(int n, const uint8_t* const* pixels, const int* Ws, const int* Hs, const int* Cs, const int* planars,
                                           uint8_t* const* outs, uint8_t* const* scratches, unsigned long long* const* out_lens, hipStream_t stream) {
    if (n < 1 || n > kMaxBatch) return hipErrorInvalidValue;
    const hipError_t ready = ensure_shift_tables();
    if (ready != hipSuccess) return ready;
    PngBatch B;
    B.n = n;
    uint32_t max_filter = 0u, max_blocks = 0u, max_crc = 0u;
    for (int i = 0; i < n; ++i) {
        PngJob& J = B.job[i];
        if (!png_layout(Ws[i], Hs[i], Cs[i], planars[i], &J.L)) return hipErrorInvalidValue;
        const DeflateScratch d = deflate_scratch(J.L);
        uint8_t* sc = scratches[i];
        J.pixels = pixels[i];
        J.out = outs[i];
        J.stream = sc + d.stream_at;
        J.dyn = reinterpret_cast<PngDynamic*>(sc + d.dyn_at);
        J.hist = reinterpret_cast<uint32_t*>(sc + d.hist_at);
        J.adler = reinterpret_cast<unsigned long long*>(sc + d.adler_at);
        J.block_off = reinterpret_cast<uint32_t*>(sc + d.off_at);
        J.crc_partials = reinterpret_cast<uint32_t*>(sc + d.crc_at);
        J.out_len = out_lens ? out_lens[i] : nullptr;
        J.blocks = d.blocks; J.filter_groups = d.filter_groups; J.crc_groups = d.crc_groups;
        max_filter = std::max(max_filter, d.filter_groups); max_blocks = std::max(max_blocks, d.blocks); max_crc = std::max(max_crc, d.crc_groups);
    }
    hipLaunchKernelGGL(png_filter_kernel, dim3(max_filter, n), dim3(256), 0, stream, B);
    hipLaunchKernelGGL(png_hist_kernel, dim3(max_blocks, n), dim3(256), 0, stream, B);
    hipLaunchKernelGGL(png_table_kernel, dim3(n), dim3(256), 0, stream, B);
    hipLaunchKernelGGL(png_size_kernel, dim3((max_blocks + 3u) / 4u, n), dim3(256), 0, stream, B);
    hipLaunchKernelGGL(png_deflate_kernel, dim3(max_blocks, n), dim3(256), 0, stream, B, png_tables());
    hipLaunchKernelGGL(png_crc_dynamic_kernel, dim3(max_crc, n), dim3(256), 0, stream, B, png_tables());
    hipLaunchKernelGGL(png_finish_dynamic_kernel, dim3(n), dim3(64), 0, stream, B, png_tables());
    return hipGetLastError();
}

hipError_t launch_png_encode_deflate(const uint8_t* pixels, int W, int H, int C, int planar, uint8_t* out, uint8_t* scratch, unsigned long long* out_len,
                                     hipStream_t stream) {
    return launch_png_encode_deflate_batch(1, &pixels, &W, &H, &C, &planar, &out, &scratch, &out_len, stream);
}

} // namespace gsr
