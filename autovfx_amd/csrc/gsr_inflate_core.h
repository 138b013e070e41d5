// gsr_inflate_core.h -- a zlib (RFC 1950) / deflate (RFC 1951) decoder for ONE stream run by ONE wave, written once for host and
// device: on the GPU the 64 lanes of a single-wave workgroup run it (kLanes = 64, lane = threadIdx.x), on the host one "lane" runs
// the same code (kLanes = 1) -- that instantiation is what the CPU tests hold against zlib, stream for stream, error for error.
//
// Why: an OpenEXR ZIP part is one zlib stream per block of 16 scanlines -- a 1080p depth pass is 68 independent streams, a frame of
// blender/blend_all.py::blend_frames' inputs 272 -- and inflating them was the largest share of the host's time per frame
// (gsr_layerfiles.hip: 10 ms per pass).  One stream is byte-serial; hundreds are not.
//
// How a wave decodes a serial format: everything that decides WHAT comes next -- the bit reader, the Huffman look-ups, block headers
// -- is computed uniformly (every lane holds the same values; LDS reads of one address are broadcasts; only lane 0 writes); the lanes
// divide what is data-parallel: filling the look-up tables, copying a match (lane i copies byte i; an overlapping match is a
// repetition of its first `distance` bytes, so byte i comes from offset i mod distance -- no lane waits for another), moving finished
// output from the LDS window to memory in 16-byte pieces, and the Adler-32 sums.  The 32 KB window lives in LDS as a ring; output
// leaves in 4 KB segments.  Codes up to 10 (literal / length) and 8 (distance) bits resolve with one table read; longer ones walk
// the canonical code length by length (puff.c's method: zlib/contrib/puff is also the model for what counts as an invalid stream).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define GSR_HD __host__ __device__
#else
#define GSR_HD
#endif

namespace gsr {
namespace inflate {

constexpr int kWindow = 32768;   // deflate's maximum distance; the LDS ring
constexpr int kLitBits = 10;     // literal / length codes this short resolve in one look-up
constexpr int kDistBits = 8;
constexpr int kFlush = 4096;     // bytes per segment leaving the ring
constexpr int kMaxSymbols = 320; // 288 literal / length + 32 distance code lengths of a dynamic header

enum Status : int {
    kOk = 0,
    kBadHeader = 1,        // not a zlib stream (method, window, preset dictionary, header check), or misaligned arguments
    kBadBlockType = 2,
    kBadStored = 3,        // stored block: LEN != ~NLEN
    kBadLengths = 4,       // dynamic header: counts out of range, a repeat without a previous length, too many lengths, no end-of-block code,
                           // an over-subscribed or incomplete code
    kBadCode = 5,          // a bit pattern that is no code word, or a length / distance symbol that does not exist
    kBadDistance = 6,      // a match reaching in front of the output
    kOutputOverflow = 7,   // more output than the caller expects
    kInputOverrun = 8,     // the stream continues past its end
    kBadChecksum = 9,      // Adler-32
    kShortOutput = 10      // less output than the caller expects
};

struct Code {                      // one canonical Huffman code: the slow path's view (puff.c: struct huffman)
    uint16_t count[16];            // code words per length
    uint16_t sorted[kMaxSymbols];  // symbols by (length, symbol)
};

struct Shared {                    // per stream; LDS on the device (38 KB)
    uint8_t ring[kWindow];
    uint16_t lit_lut[1 << kLitBits];     // (symbol << 4) | length, 0 = not a short code
    uint16_t dist_lut[1 << kDistBits];
    uint16_t code_of[kMaxSymbols];       // builder scratch: each symbol's code word, bit-reversed as it appears in the stream
    uint16_t next_code[16], offs[16];    // builder scratch: per length, the next code word / slot in `sorted`
    uint8_t lens[kMaxSymbols];
    Code lit, dist;
};

// ---- what differs between the two instantiations ------------------------------------------------------------------------------
template <int kLanes>
struct Exec;

template <>
struct Exec<1> {                                   // host: one lane does every lane's share
    static GSR_HD int lane() { return 0; }
    static GSR_HD void sync() {}
    static GSR_HD uint32_t uni(uint32_t v) { return v; }
    static GSR_HD uint64_t sum(uint64_t v) { return v; }
    struct Input {};                               // the stream's words: read where they lie
    static GSR_HD void input_open(Input&, const uint32_t*, uint32_t) {}
    static GSR_HD uint32_t input_word(Input&, const uint32_t* words, uint32_t n_words, uint32_t i) { return i < n_words ? words[i] : 0u; }
};

#if defined(__HIPCC__)
template <>
struct Exec<64> {                                  // device: a single-wave workgroup
    static __device__ int lane() { return (int)threadIdx.x; }
    // The workgroup IS one wave: its LDS operations execute in program order, so a lane sees what another lane wrote before without
    // waiting for anything; all that is needed is that the compiler keeps the order.  (__syncthreads() would also wait for the
    // output stores and the prefetched input word in flight -- twice per match.)
    static __device__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    static __device__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }   // the value is the same in every lane: say so
    static __device__ uint64_t sum(uint64_t v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, d, 64);
        return v;
    }
    // The stream's words, 64 at a time: lane k holds word base + k of the current piece and of the next (one coalesced 256-byte load
    // each, the second in flight while the first is being used); the reader takes a word with v_readlane -- no memory access, and so
    // no memory latency, on the decoder's critical path.  (A plain load where the word is needed cost ~1 us every 32 bits.)
    struct Input {
        uint32_t cur, nxt, base;
    };
    static __device__ uint32_t piece(const uint32_t* words, uint32_t n_words, uint32_t base) {
        const uint32_t i = base + threadIdx.x;
        return i < n_words ? words[i] : 0u;
    }
    static __device__ void input_open(Input& in, const uint32_t* words, uint32_t n_words) {
        in.base = 0;
        in.cur = piece(words, n_words, 0);
        in.nxt = piece(words, n_words, 64);
    }
    static __device__ uint32_t input_word(Input& in, const uint32_t* words, uint32_t n_words, uint32_t i) {      // i: uniform, never behind in.base
        if (i >= in.base + 128u) {                 // a jump (behind a long stored block)
            in.base = i & ~63u;
            in.cur = piece(words, n_words, in.base);
            in.nxt = piece(words, n_words, in.base + 64u);
        } else if (i >= in.base + 64u) {
            in.base += 64u;
            in.cur = in.nxt;
            in.nxt = piece(words, n_words, in.base + 64u);
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)in.cur, (int)(i - in.base));
    }
};
#endif

// ---- the bit reader: 32-bit words of a 4-byte aligned stream, least significant bit first -----------------------------------------
// (kept lean: one wave alone on its SIMD issues an instruction every five to ten cycles, so the ~100 scalar instructions a symbol cost
// at first were the decoder's time: the bits taken so far are derived, not counted; peeks are 32-bit)
template <int kLanes>
struct Bits {
    const uint32_t* words;
    uint32_t n_words, next;  // word next - 1 is in `ahead`, everything before it has entered `buf`
    uint64_t buf;
    int cnt;                 // valid bits in buf
    uint32_t ahead;          // the next word
    typename Exec<kLanes>::Input in;
};

template <int kLanes>
GSR_HD inline void bits_open(Bits<kLanes>& b, const uint8_t* src, size_t src_len) {
    b.words = reinterpret_cast<const uint32_t*>(src);
    b.n_words = (uint32_t)((src_len + 3) / 4);
    b.buf = 0;
    b.cnt = 0;
    b.next = 1;
    Exec<kLanes>::input_open(b.in, b.words, b.n_words);
    b.ahead = Exec<kLanes>::input_word(b.in, b.words, b.n_words, 0u);
}
// at least 32 valid bits afterwards -- the longest run of takes between two refills is a distance code and its extra bits, 28 --
// (zeros behind the end of the stream: bits_consumed tells)
template <int kLanes>
GSR_HD inline void bits_refill(Bits<kLanes>& b) {
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)b.ahead << b.cnt;
        b.cnt += 32;
        b.ahead = Exec<kLanes>::input_word(b.in, b.words, b.n_words, b.next);
        ++b.next;
    }
}
template <int kLanes>
GSR_HD inline uint64_t bits_consumed(const Bits<kLanes>& b) { return (uint64_t)(b.next - 1u) * 32u - (uint64_t)b.cnt; }
template <int kLanes>
GSR_HD inline uint32_t bits_peek(const Bits<kLanes>& b, int n) { return (uint32_t)b.buf & ((1u << n) - 1u); }      // n <= 28, within the 32 refilled bits
template <int kLanes>
GSR_HD inline void bits_drop(Bits<kLanes>& b, int n) {
    b.buf >>= n;
    b.cnt -= n;
}
template <int kLanes>
GSR_HD inline uint32_t bits_take(Bits<kLanes>& b, int n) {
    const uint32_t v = bits_peek(b, n);
    bits_drop(b, n);
    return v;
}
// the reader repositioned to byte `at` of the stream (behind a stored block)
template <int kLanes>
GSR_HD inline void bits_seek(Bits<kLanes>& b, uint32_t at) {
    b.next = at / 4u;
    b.buf = 0;
    b.cnt = 0;
    b.ahead = Exec<kLanes>::input_word(b.in, b.words, b.n_words, b.next);
    ++b.next;
    bits_refill(b);
    const int skip = (int)(at & 3u) * 8;
    b.buf >>= skip;
    b.cnt -= skip;
}

GSR_HD inline uint32_t reverse_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// Build one code from sh.lens[at ... at + n): counts, the sorted symbols, the look-up table.  Returns puff's verdict: 0 complete,
// > 0 incomplete (bits of code space left), < 0 over-subscribed.
template <int kLanes>
GSR_HD inline int build_code(Shared& sh, int at, int n, Code& code, uint16_t* lut, int lut_bits) {
    typedef Exec<kLanes> X;
    const int lane = X::lane();
    X::sync();
    // code words per length: lane l counts length l (every lane reads the same lens[s]: a broadcast)
    for (int l = lane; l < 16; l += kLanes) {
        uint32_t c = 0;
        for (int s = 0; s < n; ++s) c += (uint32_t)(sh.lens[at + s] == l);
        code.count[l] = (uint16_t)(l ? c : 0u);
    }
    X::sync();
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= (int)X::uni(code.count[l]);
        if (left < 0) return left;
    }
    const int entries = 1 << lut_bits;
    for (int k = lane; k < entries; k += kLanes) lut[k] = 0;
    // (serial, one lane) canonical code words in symbol order: the first word and the first slot of every length, then one pass
    if (lane == 0) {
        sh.next_code[0] = 0; sh.offs[0] = 0; sh.next_code[1] = 0; sh.offs[1] = 0;
        for (int l = 2; l <= 15; ++l) {
            sh.next_code[l] = (uint16_t)((sh.next_code[l - 1] + code.count[l - 1]) << 1);
            sh.offs[l] = (uint16_t)(sh.offs[l - 1] + code.count[l - 1]);
        }
        for (int s = 0; s < n; ++s) {
            const int l = sh.lens[at + s];
            if (l == 0) continue;
            const uint32_t c = sh.next_code[l]++, slot = sh.offs[l]++;
            code.sorted[slot] = (uint16_t)s;
            sh.code_of[s] = (uint16_t)reverse_bits(c, l);
        }
    }
    X::sync();
    // (parallel) every short code word owns the table entries whose low bits are the word
    for (int s = lane; s < n; s += kLanes) {
        const int l = sh.lens[at + s];
        if (l == 0 || l > lut_bits) continue;
        const uint16_t entry = (uint16_t)((s << 4) | l);
        for (int k = sh.code_of[s]; k < entries; k += 1 << l) lut[k] = entry;
    }
    X::sync();
    return left;
}

// One symbol.  < 0: no code word matches.
template <int kLanes>
GSR_HD inline int decode_symbol(const Code& code, const uint16_t* lut, int lut_bits, Bits<kLanes>& b) {
    typedef Exec<kLanes> X;
    const uint32_t e = X::uni(lut[bits_peek(b, lut_bits)]);
    if (e) {
        bits_drop(b, (int)(e & 15u));
        return (int)(e >> 4);
    }
    int word = 0, first = 0, index = 0;
    const uint32_t bits = bits_peek(b, 15);
    for (int len = 1; len <= 15; ++len) {          // puff.c: decode()
        word |= (int)((bits >> (len - 1)) & 1u);
        const int count = (int)X::uni(code.count[len]);
        if (word - count < first) {
            bits_drop(b, len);
            return (int)X::uni(code.sorted[index + (word - first)]);
        }
        index += count;
        first += count;
        first <<= 1;
        word <<= 1;
    }
    return -1;
}

// A match: bytes [pos, pos + len) of the output = the bytes dist in front of them, len <= 258.  Byte i of the match is byte
// (i mod dist) of the dist bytes in front of it -- a match that overlaps itself repeats its first dist bytes -- so no byte waits for
// another: every lane reads ALL its bytes (at most five: one LDS round trip, not five), then writes them.
template <int kLanes>
GSR_HD inline void copy_match(Shared& sh, uint32_t pos, uint32_t dist, uint32_t len) {
    typedef Exec<kLanes> X;
    constexpr uint32_t kMask = (uint32_t)(kWindow - 1);
    const uint32_t start = pos - dist;
    X::sync();
    if (kLanes == 1) {
        uint32_t from = 0;
        for (uint32_t i = 0; i < len; ++i) {
            sh.ring[(pos + i) & kMask] = sh.ring[(start + from) & kMask];
            if (++from == dist) from = 0;
        }
    } else {
        const uint32_t lane = (uint32_t)X::lane();
        // lane mod dist and kLanes mod dist without an integer division (two of them cost more than the rest of a short match):
        // a float quotient, corrected by one where it rounds the wrong way (the operands are below 2^16: exact enough)
        uint32_t from = lane, step = (uint32_t)kLanes;
        if (dist < len && dist <= (uint32_t)kLanes) {
            const float inv = 1.0f / (float)dist;
            from = lane - (uint32_t)((float)lane * inv) * dist;
            from = (int32_t)from < 0 ? from + dist : from >= dist ? from - dist : from;      // (a quotient one too many / one too few)
            step = (uint32_t)kLanes - (uint32_t)((float)kLanes * inv) * dist;
            step = (int32_t)step < 0 ? step + dist : step >= dist ? step - dist : step;
        }
        if (len <= (uint32_t)kLanes) {               // the common case: one piece
            if (lane < len) sh.ring[(pos + lane) & kMask] = sh.ring[(start + from) & kMask];
        } else {
            constexpr int kPieces = (258 + kLanes - 1) / kLanes;
            uint8_t bytes[kPieces];
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                bytes[k] = lane + (uint32_t)(k * kLanes) < len ? sh.ring[(start + from) & kMask] : (uint8_t)0;
                from += step;
                if (dist < len && from >= dist) from -= dist;
            }
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                const uint32_t i = lane + (uint32_t)(k * kLanes);
                if (i < len) sh.ring[(pos + i) & kMask] = bytes[k];
            }
        }
    }
}

struct Adler {
    uint32_t s1, s2;
};

// Bytes [from, from + m) of the output leave the ring for dst; m <= kFlush.  The Adler-32 sums advance.
template <int kLanes>
GSR_HD inline void flush(Shared& sh, uint8_t* dst, uint32_t from, uint32_t m, Adler& ad) {
    typedef Exec<kLanes> X;
    const int lane = X::lane();
    X::sync();
    uint64_t a = 0, w = 0;
    const uint32_t base = from & (uint32_t)(kWindow - 1);          // (segments start on multiples of kFlush: they do not wrap)
    if (m == (uint32_t)kFlush && ((reinterpret_cast<uintptr_t>(dst) + from) & 15u) == 0) {
        for (uint32_t i = (uint32_t)lane * 16u; i < m; i += (uint32_t)kLanes * 16u) {
            uint32_t q[4];
            for (int k = 0; k < 4; ++k) q[k] = *reinterpret_cast<const uint32_t*>(&sh.ring[base + i + 4 * k]);
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<uint32_t*>(dst + from + i + 4 * k) = q[k];
                for (int j = 0; j < 4; ++j) {
                    const uint32_t byte = (q[k] >> (8 * j)) & 255u;
                    a += byte;
                    w += (uint64_t)(m - (i + 4 * k + j)) * byte;
                }
            }
        }
    } else {
        for (uint32_t i = (uint32_t)lane; i < m; i += (uint32_t)kLanes) {
            const uint32_t byte = sh.ring[base + i];
            dst[from + i] = (uint8_t)byte;
            a += byte;
            w += (uint64_t)(m - i) * byte;
        }
    }
    a = X::sum(a);
    w = X::sum(w);
    ad.s2 = (uint32_t)((ad.s2 + (uint64_t)m * ad.s1 + w) % 65521u);
    ad.s1 = (uint32_t)((ad.s1 + a) % 65521u);
    X::sync();
}

// RFC 1951 3.2.5 as arithmetic (a table in memory costs a load on the critical path): lengths 3 ... 10 one by one, then four codes per
// extra bit, 258 with its own code; distances 1 ... 4 one by one, then two codes per extra bit.
GSR_HD inline int length_extra_bits(int ls) { return ls < 8 || ls == 28 ? 0 : (ls - 4) >> 2; }
GSR_HD inline uint32_t length_base(int ls) { return ls < 8 ? 3u + (uint32_t)ls : ls == 28 ? 258u : 3u + ((4u + ((uint32_t)ls & 3u)) << length_extra_bits(ls)); }
GSR_HD inline int distance_extra_bits(int ds) { return ds < 4 ? 0 : (ds - 2) >> 1; }
GSR_HD inline uint32_t distance_base(int ds) { return ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1u)) << distance_extra_bits(ds)); }

struct Output {              // where the decoder stands in its output
    uint8_t* dst;
    uint32_t out_len, pos, flushed;
    Adler ad;
};

template <int kLanes>
GSR_HD inline void flush_full_segments(Shared& sh, Output& o) {
    while (o.pos - o.flushed >= (uint32_t)kFlush) {
        flush<kLanes>(sh, o.dst, o.flushed, (uint32_t)kFlush, o.ad);
        o.flushed += (uint32_t)kFlush;
    }
}

// One symbol of a block, every step in turn: 0 go on, 1 end of block, < 0 minus a Status.
template <int kLanes>
GSR_HD inline int symbol_step(Shared& sh, Bits<kLanes>& b, Output& o) {
    bits_refill(b);
    int sym = decode_symbol<kLanes>(sh.lit, sh.lit_lut, kLitBits, b);
    if (sym < 0) return -kBadCode;
    if (sym < 256) {
        if (o.pos >= o.out_len) return -kOutputOverflow;
        if (Exec<kLanes>::lane() == 0) sh.ring[o.pos & (uint32_t)(kWindow - 1)] = (uint8_t)sym;
        ++o.pos;
    } else if (sym == 256) {
        return 1;
    } else {
        sym -= 257;
        if (sym >= 29) return -kBadCode;
        const int l_extra = length_extra_bits(sym);
        const uint32_t len = length_base(sym) + bits_take(b, l_extra);
        bits_refill(b);
        const int dsym = decode_symbol<kLanes>(sh.dist, sh.dist_lut, kDistBits, b);
        if (dsym < 0 || dsym >= 30) return -kBadCode;
        const int d_extra = distance_extra_bits(dsym);
        const uint32_t dist = distance_base(dsym) + bits_take(b, d_extra);
        if (dist > o.pos) return -kBadDistance;
        if (len > o.out_len - o.pos) return -kOutputOverflow;
        copy_match<kLanes>(sh, o.pos, dist, len);
        o.pos += len;
    }
    flush_full_segments<kLanes>(sh, o);
    return 0;
}

// The symbols of a block up to its end-of-block code.  0, or minus a Status.
template <int kLanes>
GSR_HD inline int decode_block(Shared& sh, Bits<kLanes>& b, Output& o) {
    for (;;) {
        const int r = symbol_step<kLanes>(sh, b, o);
        if (r != 0) return r < 0 ? r : 0;
    }
}

// The zlib stream [src, src + src_len) -> dst[0 ... dst_len).  src: 4-byte aligned, readable up to the next multiple of 4.
template <int kLanes>
GSR_HD inline int inflate_zlib(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len, Shared& sh) {
    typedef Exec<kLanes> X;
    const int lane = X::lane();
    const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

    if ((reinterpret_cast<uintptr_t>(src) & 3u) != 0 || src_len < 6 || dst_len > 0xFFFFFFFFull - 512ull) return kBadHeader;
    const uint64_t total_bits = (uint64_t)src_len * 8u;
    Bits<kLanes> b;
    bits_open(b, src, src_len);
    bits_refill(b);
    const uint32_t cmf = bits_take(b, 8), flg = bits_take(b, 8);
    if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || (flg & 32u) != 0u || ((cmf << 8) | flg) % 31u != 0u) return kBadHeader;

    Output o = {dst, (uint32_t)dst_len, 0u, 0u, {1u, 0u}};
    uint32_t& pos = o.pos;
    uint32_t& flushed = o.flushed;
    Adler& ad = o.ad;
    const uint32_t out_len = o.out_len;
    for (;;) {
        bits_refill(b);
        const uint32_t last = bits_take(b, 1), type = bits_take(b, 2);
        if (type == 3u) return kBadBlockType;
        if (type == 0u) {
            // ---- stored: to the next byte boundary, LEN, ~LEN, LEN bytes
            bits_drop(b, (int)((8u - (uint32_t)(bits_consumed(b) & 7u)) & 7u));
            bits_refill(b);
            const uint32_t len = bits_take(b, 16);
            bits_refill(b);
            const uint32_t nlen = bits_take(b, 16);
            if ((len ^ nlen) != 0xFFFFu) return kBadStored;
            if (bits_consumed(b) + (uint64_t)len * 8u > total_bits) return kInputOverrun;
            if (len > out_len - pos) return kOutputOverflow;
            const uint32_t from = (uint32_t)(bits_consumed(b) >> 3);  // byte offset in the stream
            uint32_t done = 0;
            while (done < len) {                                       // (in pieces, so that the ring's unflushed part stays small)
                const uint32_t piece = len - done < (uint32_t)kFlush ? len - done : (uint32_t)kFlush;
                X::sync();
                for (uint32_t i = (uint32_t)lane; i < piece; i += (uint32_t)kLanes) sh.ring[(pos + i) & (uint32_t)(kWindow - 1)] = src[from + done + i];
                pos += piece;
                done += piece;
                while (pos - flushed >= (uint32_t)kFlush) {
                    flush<kLanes>(sh, dst, flushed, (uint32_t)kFlush, ad);
                    flushed += (uint32_t)kFlush;
                }
            }
            bits_seek(b, from + len);      // the reader starts over behind the copied bytes
        } else {
            int n_lit = 288, n_dist = 30;
            if (type == 1u) {
                // ---- fixed code (RFC 1951 3.2.6)
                X::sync();
                for (int s = lane; s < 288 + 30; s += kLanes) sh.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                X::sync();
            } else {
                // ---- dynamic code: the code length code, then the two codes' lengths
                bits_refill(b);
                n_lit = (int)bits_take(b, 5) + 257;
                n_dist = (int)bits_take(b, 5) + 1;
                const int n_cl = (int)bits_take(b, 4) + 4;
                if (n_lit > 286 || n_dist > 30) return kBadLengths;
                X::sync();
                for (int s = lane; s < 19; s += kLanes) sh.lens[s] = 0;
                X::sync();
                for (int k = 0; k < n_cl; ++k) {
                    bits_refill(b);
                    const uint32_t l = bits_take(b, 3);
                    if (lane == 0) sh.lens[cl_order[k]] = (uint8_t)l;
                }
                X::sync();
                if (build_code<kLanes>(sh, 0, 19, sh.dist, sh.dist_lut, 7) != 0) return kBadLengths;       // (must be complete: puff.c)
                // the lengths are decoded with the code built from lens[0 ... 19): keep them in registers until that code is done with
                int have = 0;
                uint32_t prev = 0;
                // (lens[] is being read by nobody now: the code length code lives in sh.dist / dist_lut)
                while (have < n_lit + n_dist) {
                    bits_refill(b);
                    if (bits_consumed(b) > total_bits) return kInputOverrun;
                    const int sym = decode_symbol<kLanes>(sh.dist, sh.dist_lut, 7, b);
                    if (sym < 0) return kBadCode;
                    if (sym < 16) {
                        if (lane == 0) sh.lens[have] = (uint8_t)sym;
                        prev = (uint32_t)sym;
                        ++have;
                    } else {
                        uint32_t value = 0;
                        int repeat;
                        if (sym == 16) {
                            if (have == 0) return kBadLengths;
                            value = prev;
                            repeat = 3 + (int)bits_take(b, 2);
                        } else if (sym == 17) {
                            repeat = 3 + (int)bits_take(b, 3);
                        } else {
                            repeat = 11 + (int)bits_take(b, 7);
                        }
                        if (have + repeat > n_lit + n_dist) return kBadLengths;
                        if (lane == 0)
                            for (int r = 0; r < repeat; ++r) sh.lens[have + r] = (uint8_t)value;
                        have += repeat;
                        prev = value;
                    }
                }
                X::sync();
                if (X::uni(sh.lens[256]) == 0u) return kBadLengths;       // no end-of-block code
            }
            // lens[0 ... n_lit) literal / length, lens[n_lit ... n_lit + n_dist) distance
            const int dist_at = type == 1u ? 288 : n_lit;
            int err = build_code<kLanes>(sh, 0, n_lit, sh.lit, sh.lit_lut, kLitBits);
            if (err < 0) return kBadLengths;
            if (err > 0 && type == 2u) {   // incomplete: allowed only for a single code word of length 1 (puff.c); the fixed code is what it is
                int nonzero = 0, ones = 0;
                for (int s = 0; s < n_lit; ++s) {
                    const uint32_t l = X::uni(sh.lens[s]);
                    nonzero += l != 0u;
                    ones += l == 1u;
                }
                if (!(nonzero == 1 && ones == 1)) return kBadLengths;
            }
            err = build_code<kLanes>(sh, dist_at, n_dist, sh.dist, sh.dist_lut, kDistBits);
            if (err < 0) return kBadLengths;
            if (err > 0 && type == 2u) {
                int nonzero = 0, ones = 0;
                for (int s = 0; s < n_dist; ++s) {
                    const uint32_t l = X::uni(sh.lens[dist_at + s]);
                    nonzero += l != 0u;
                    ones += l == 1u;
                }
                if (!((nonzero == 1 && ones == 1) || nonzero == 0)) return kBadLengths;
            }
            // ---- the symbols of the block.  (No end-of-input test per symbol: behind the stream's end the reader delivers zeros, and a
            // stream of zeros ends -- in an output overflow, or in an end-of-block symbol and then a stored block whose LEN / NLEN do
            // not match; the test behind the block catches the overrun.)
            const int r = decode_block<kLanes>(sh, b, o);
            if (r != 0) return -r;
        }
        if (bits_consumed(b) > total_bits) return kInputOverrun;
        if (last) break;
    }
    if (pos > flushed) flush<kLanes>(sh, dst, flushed, pos - flushed, ad);
    if (pos != out_len) return kShortOutput;
    // the Adler-32 of the output, big-endian, on the next byte boundary
    bits_drop(b, (int)((8u - (uint32_t)(bits_consumed(b) & 7u)) & 7u));
    uint32_t stored = 0;
    for (int k = 0; k < 4; ++k) {
        bits_refill(b);
        stored = (stored << 8) | bits_take(b, 8);
    }
    if (bits_consumed(b) > total_bits) return kInputOverrun;
    if (stored != ((ad.s2 << 16) | ad.s1)) return kBadChecksum;
    return kOk;
}

} // namespace inflate
} // namespace gsr
