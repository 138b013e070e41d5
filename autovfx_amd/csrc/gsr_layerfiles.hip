// gsr_layerfiles.hip -- the HOST side of the compositor's input files (no device code): container parsing and zlib's inflate for the
// PNG layers and OpenEXR depth passes blender/blend_all.py::blend_frames reads (blend_all.py:56-75: load_rgb, load_depth_exr), as one
// native call per file.  What comes out is what gsr_layerio.hip's kernels take: the inflated IDAT scanline stream of a PNG, the
// inflated scanline blocks of an OpenEXR part -- written straight into the caller's (page-locked) buffer.
//
// Why native: the same parsing in Python (autovfx_amd/layer_io.py: png_chunks / _exr_plan, kept as the readable restatement the tests
// compare this file with) held the interpreter lock for ~6 ms per frame -- 11 files, ~300 small calls -- and with every pool thread of
// blend_frames doing the same, the lock, not the cores, set the frame rate past 16 threads.
//
// Formats: PNG (ISO/IEC 15948) chunk layout + CRC, 8-bit truecolour with or without alpha, non-interlaced; OpenEXR 2 single-part
// scanline files ("OpenEXR File Layout"), compression RLE / ZIPS / ZIP, the wanted channel HALF or FLOAT, no sub-sampling.  Anything
// else is reported as "not covered" (the caller's host decoder reads it), never guessed at.
#include "gsr_internal.h"
#include "gsr_inflate_core.h"

#include <zlib.h>

#include <cstring>
#include <vector>

namespace gsr {
namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
inline int32_t le32(const uint8_t* p) {
    uint32_t v;
    std::memcpy(&v, p, 4);
    return (int32_t)v;   // (little-endian host: the library is built for x86-64 / gfx950 boxes only)
}
inline uint64_t le64(const uint8_t* p) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    return v;
}

const uint8_t kPngSignature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

// Walks the chunks of a PNG file: IHDR, the IDAT pieces (offset, length), stops at IEND.  False: not a file the unfilter kernel takes.
struct PngWalk {
    int width = 0, height = 0, channels = 0;
    std::vector<std::pair<size_t, size_t>> idat;
};

bool walk_png(const uint8_t* f, size_t n, PngWalk* out) {
    if (n < 8 || std::memcmp(f, kPngSignature, 8) != 0) return false;
    size_t at = 8;
    bool have_header = false;
    while (at + 12 <= n) {
        const size_t len = be32(f + at);
        const uint8_t* kind = f + at + 4;
        if (len > n || at + 12 + len > n) return false;
        const uint8_t* data = f + at + 8;
        const bool ihdr = std::memcmp(kind, "IHDR", 4) == 0, idat = std::memcmp(kind, "IDAT", 4) == 0;
        if (ihdr || idat) {
            const uint32_t crc = (uint32_t)crc32(crc32(0L, kind, 4), data, (uInt)len);
            if (crc != be32(data + len)) return false;
        }
        if (ihdr) {
            if (len != 13) return false;
            const uint32_t w = be32(data), h = be32(data + 4);
            const int depth = data[8], colour = data[9], compression = data[10], filtering = data[11], interlace = data[12];
            if (depth != 8 || (colour != 2 && colour != 6) || compression || filtering || interlace || w < 1 || h < 1 || w > (1u << 20) || h > (1u << 24))
                return false;
            out->width = (int)w;
            out->height = (int)h;
            out->channels = colour == 2 ? 3 : 4;
            have_header = true;
        } else if (idat) {
            out->idat.emplace_back(at + 8, len);
        } else if (std::memcmp(kind, "tRNS", 4) == 0 || std::memcmp(kind, "PLTE", 4) == 0) {
            return false;
        } else if (std::memcmp(kind, "IEND", 4) == 0) {
            break;
        }
        at += 12 + len;
    }
    if (!have_header || out->idat.empty()) return false;
    if (png_unfilter_scratch_bytes(out->width, out->height) == 0 || (size_t)out->height * (1 + 4 * (size_t)out->width) > ((size_t)1 << 30)) return false;
    return true;
}

// ---- OpenEXR ---------------------------------------------------------------------------------------------------------------------
struct ExrWalk {
    ExrFileLayout layout = {};
    int compression = -1;
    int n_blocks = 0;
    size_t offsets_at = 0;
    int ymin = 0;
};

// a NUL-terminated string inside [at, n) (attribute and channel names are at most 255 bytes): its length, or -1
inline long cstr_len(const uint8_t* f, size_t at, size_t n) {
    for (size_t i = at; i < n && i < at + 256; ++i)
        if (f[i] == 0) return (long)(i - at);
    return -1;
}

bool walk_exr(const uint8_t* f, size_t n, const char* want, ExrWalk* out) {
    if (n < 9 || le32(f) != 20000630) return false;
    const int version = le32(f + 4);
    if ((version & 0xFF) != 2 || (version & 0x200) || (version & 0x1800)) return false;      // tiled / deep / multi-part: not here
    size_t at = 8;
    bool have_channels = false, have_window = false;
    size_t chl_at = 0, chl_len = 0;
    int xmin = 0, ymin = 0, xmax = -1, ymax = -1;
    while (true) {
        if (at >= n) return false;
        if (f[at] == 0) { ++at; break; }
        const long name_len = cstr_len(f, at, n);
        if (name_len < 0) return false;
        const char* name = reinterpret_cast<const char*>(f + at);
        at += (size_t)name_len + 1;
        const long type_len = cstr_len(f, at, n);
        if (type_len < 0) return false;
        at += (size_t)type_len + 1;
        if (at + 4 > n) return false;
        const int size = le32(f + at);
        at += 4;
        if (size < 0 || at + (size_t)size > n) return false;
        if (std::strcmp(name, "channels") == 0) { chl_at = at; chl_len = (size_t)size; have_channels = true; }
        else if (std::strcmp(name, "compression") == 0) {
            if (size != 1) return false;
            out->compression = f[at];
        } else if (std::strcmp(name, "dataWindow") == 0) {
            if (size != 16) return false;
            xmin = le32(f + at); ymin = le32(f + at + 4); xmax = le32(f + at + 8); ymax = le32(f + at + 12);
            have_window = true;
        } else if (std::strcmp(name, "lineOrder") == 0 && size < 1) {
            return false;
        }
        at += (size_t)size;
    }
    if (!have_channels || !have_window || xmax < xmin || ymax < ymin) return false;
    const long long W = (long long)xmax - xmin + 1, H = (long long)ymax - ymin + 1;
    if (W > (1 << 20) || H > (1 << 24)) return false;
    int lines_per_block;
    switch (out->compression) {
        case 1: case 2: lines_per_block = 1; break;     // RLE, ZIPS
        case 3: lines_per_block = 16; break;            // ZIP
        default: return false;                          // NONE has nothing to undo; PIZ, PXR24, B44, DWA: a host decoder's
    }
    // the channel list: name \0, pixel type i32, pLinear u8 + 3 reserved, x / y sampling i32; channels are stored in this order, line by line
    struct Chan { const char* name; int type; };
    std::vector<Chan> chans;
    size_t c = chl_at;
    const size_t c_end = chl_at + chl_len;
    while (c < c_end && f[c] != 0) {
        const long len = cstr_len(f, c, c_end);
        if (len < 0 || c + (size_t)len + 1 + 16 > c_end) return false;
        const uint8_t* rest = f + c + len + 1;
        const int type = le32(rest), xs = le32(rest + 8), ys = le32(rest + 12);
        if (type < 0 || type > 2 || xs != 1 || ys != 1) return false;
        chans.push_back({reinterpret_cast<const char*>(f + c), type});
        c += (size_t)len + 1 + 16;
    }
    if (chans.empty()) return false;
    int pick = -1;
    if (want && *want) {
        for (size_t i = 0; i < chans.size(); ++i)
            if (std::strcmp(chans[i].name, want) == 0) pick = (int)i;
    } else {
        // what cv2.imread(path, ANYCOLOR | ANYDEPTH)[:, :, 0] is: the B channel of a colour file, else the one grey-ish channel there is
        for (const char* pref : {"B", "G", "R", "Y", "Z", "V"}) {
            for (size_t i = 0; i < chans.size() && pick < 0; ++i)
                if (std::strcmp(chans[i].name, pref) == 0) pick = (int)i;
            if (pick >= 0) break;
        }
        if (pick < 0) pick = 0;
    }
    if (pick < 0 || chans[pick].type == 0) return false;            // (UINT samples: a host reader's)
    long long per_pixel = 0, before = 0;
    for (size_t i = 0; i < chans.size(); ++i) {
        const int size = chans[i].type == 1 ? 2 : 4;
        if ((int)i < pick) before += size;
        per_pixel += size;
    }
    const long long bytes_per_line = per_pixel * W;
    if (bytes_per_line * lines_per_block > (1ll << 30) || bytes_per_line * H > (1ll << 32)) return false;
    ExrFileLayout& L = out->layout;
    L.width = (int)W;
    L.height = (int)H;
    L.bytes_per_line = (int)bytes_per_line;
    L.lines_per_block = lines_per_block;
    L.channel_at = (int)(before * W);
    L.channel_bytes = (int)((chans[pick].type == 1 ? 2 : 4) * W);
    L.channel_is_half = chans[pick].type == 1;
    L.blocks_bytes = (size_t)(bytes_per_line * H);
    L.compression = out->compression;
    L.n_blocks = (int)((H + lines_per_block - 1) / lines_per_block);
    std::memset(L.channel, 0, sizeof L.channel);
    std::strncpy(L.channel, chans[pick].name, sizeof L.channel - 1);
    out->n_blocks = (int)((H + lines_per_block - 1) / lines_per_block);
    out->offsets_at = at;
    out->ymin = ymin;
    if (at + 8 * (size_t)out->n_blocks > n) return false;
    // every block present, in place and actually compressed (one that did not shrink is stored without the predictor)
    for (int k = 0; k < out->n_blocks; ++k) {
        const uint64_t off = le64(f + at + 8 * (size_t)k);
        if (off > n || off + 8 > n) return false;
        const int y = le32(f + off), size = le32(f + off + 4);
        const long long lines = std::min<long long>(lines_per_block, H - (long long)k * lines_per_block);
        if (y != ymin + k * lines_per_block || size < 0 || (long long)size >= lines * bytes_per_line || off + 8 + (uint64_t)size > n) return false;
    }
    return true;
}

// OpenEXR's run-length code (ImfRle: a signed count n; n < 0: -n literal bytes follow; n >= 0: the next byte n + 1 times)
bool exr_rle_decode(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out) {
    size_t i = 0, o = 0;
    while (i < n_in) {
        const int count = (int8_t)in[i++];
        if (count < 0) {
            const size_t m = (size_t)(-count);
            if (i + m > n_in || o + m > n_out) return false;
            std::memcpy(out + o, in + i, m);
            i += m;
            o += m;
        } else {
            const size_t m = (size_t)count + 1;
            if (i >= n_in || o + m > n_out) return false;
            std::memset(out + o, in[i++], m);
            o += m;
        }
    }
    return o == n_out;
}

} // namespace

// The wave decoder's host instantiation (one lane): what the CPU tests compare with zlib.  src need not be aligned here.
int inflate_zlib_host(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len) {
    std::vector<uint32_t> aligned((src_len + 3) / 4 + 2, 0u);
    std::memcpy(aligned.data(), src, src_len);
    std::vector<uint8_t> shared(sizeof(inflate::Shared));
    return inflate::inflate_zlib<1>(reinterpret_cast<const uint8_t*>(aligned.data()), src_len, dst, dst_len,
                                    *reinterpret_cast<inflate::Shared*>(shared.data()));
}

int png_file_probe(const uint8_t* file, size_t n, PngFileLayout* out) {
    PngWalk w;
    if (!walk_png(file, n, &w)) return 1;
    out->width = w.width;
    out->height = w.height;
    out->channels = w.channels;
    out->scanline_bytes = (size_t)w.height * (1 + (size_t)w.width * w.channels);
    return 0;
}

int png_file_inflate(const uint8_t* file, size_t n, uint8_t* scanlines, size_t scanline_bytes) {
    PngWalk w;
    if (!walk_png(file, n, &w)) return 1;
    const size_t stride = 1 + (size_t)w.width * w.channels;
    if (scanline_bytes != (size_t)w.height * stride) return 1;
    z_stream z;
    std::memset(&z, 0, sizeof z);
    if (inflateInit(&z) != Z_OK) return 1;
    z.next_out = scanlines;
    z.avail_out = (uInt)scanline_bytes;      // (at most 1 GB: walk_png)
    uint8_t spill[64];                       // where output beyond the image would go: its presence fails the file below
    int rc = Z_OK;
    for (size_t k = 0; k < w.idat.size() && rc == Z_OK; ++k) {
        z.next_in = const_cast<Bytef*>(file + w.idat[k].first);
        z.avail_in = (uInt)w.idat[k].second;
        while (z.avail_in > 0 && rc == Z_OK) {
            if (z.avail_out == 0) {
                z.next_out = spill;
                z.avail_out = sizeof spill;
            }
            rc = ::inflate(&z, Z_NO_FLUSH);
            if (z.total_out > scanline_bytes) rc = Z_DATA_ERROR;
        }
    }
    const bool complete = rc == Z_STREAM_END && z.total_out == scanline_bytes;
    inflateEnd(&z);
    if (!complete) return 1;
    for (int y = 0; y < w.height; ++y)
        if (scanlines[(size_t)y * stride] > 4) return 1;               // filter types are 0 ... 4
    return 0;
}

int exr_file_probe(const uint8_t* file, size_t n, const char* want_channel, ExrFileLayout* out) {
    ExrWalk w;
    if (!walk_exr(file, n, want_channel, &w)) return 1;
    *out = w.layout;
    return 0;
}

int exr_file_inflate(const uint8_t* file, size_t n, const char* want_channel, uint8_t* blocks, size_t blocks_bytes) {
    ExrWalk w;
    if (!walk_exr(file, n, want_channel, &w)) return 1;
    const ExrFileLayout& L = w.layout;
    if (blocks_bytes != L.blocks_bytes) return 1;
    size_t at = 0;
    for (int k = 0; k < w.n_blocks; ++k) {
        const uint64_t off = le64(file + w.offsets_at + 8 * (size_t)k);
        const size_t size = (size_t)le32(file + off + 4);
        const size_t lines = (size_t)std::min<long long>(L.lines_per_block, (long long)L.height - (long long)k * L.lines_per_block);
        const size_t expected = lines * (size_t)L.bytes_per_line;
        if (w.compression == 1) {
            if (!exr_rle_decode(file + off + 8, size, blocks + at, expected)) return 1;
        } else {
            uLongf got = (uLongf)expected;
            if (uncompress(blocks + at, &got, file + off + 8, (uLong)size) != Z_OK || got != expected) return 1;
        }
        at += expected;
    }
    return 0;
}

int exr_file_pack(const uint8_t* file, size_t n, const char* want_channel, uint8_t* packed, size_t packed_room, InflateJob* jobs, size_t* packed_bytes) {
    ExrWalk w;
    if (!walk_exr(file, n, want_channel, &w) || w.compression == 1) return 1;      // (RLE is not a zlib stream: exr_file_inflate)
    const ExrFileLayout& L = w.layout;
    size_t at = 0, out_at = 0;
    for (int k = 0; k < w.n_blocks; ++k) {
        const uint64_t off = le64(file + w.offsets_at + 8 * (size_t)k);
        const size_t size = (size_t)le32(file + off + 4);
        const size_t lines = (size_t)std::min<long long>(L.lines_per_block, (long long)L.height - (long long)k * L.lines_per_block);
        const size_t padded = (size + 3) & ~(size_t)3;
        if (at + padded > packed_room) return 1;
        std::memcpy(packed + at, file + off + 8, size);
        std::memset(packed + at + size, 0, padded - size);
        jobs[k] = {(uint32_t)at, (uint32_t)size, (uint32_t)out_at, (uint32_t)(lines * (size_t)L.bytes_per_line)};
        at += padded;
        out_at += lines * (size_t)L.bytes_per_line;
    }
    *packed_bytes = at;
    return 0;
}

} // namespace gsr
