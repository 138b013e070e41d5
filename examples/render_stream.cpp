// render_stream.cpp -- several frames in flight from ONE host thread through the split call of the C ABI
// (gsr_forward_begin / gsr_forward_finish), plain C++ / HIP.
//
//   render_stream <scene.bin> <out.bin> <frames> [streams=3]
//
// Renders the scene of render_raw.cpp's file format `frames` times, frame i on stream i % streams: the first half of
// frame i + streams - 1 (projection, depth sort) is queued before the thread waits for the pair count of frame i, so
// the GPU never idles on the host round trip.  Every stream owns its outputs and its three scratch arenas.  Writes
// the LAST frame's outputs in render_raw's format (same scene => same bytes as render_raw) and prints frames/s.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/gsr.h"

#define HIP_OK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));                    \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

namespace {

struct Arena {
    char* base = nullptr;
    size_t capacity = 0;
};

// Grows with 25 % head room and never shrinks: after the first frames of a trajectory the callbacks stop allocating.
// Freeing the old block is safe because a slot's arenas are only re-requested after its previous frame was finished
// and its stream drained of that frame's work (see the loop below).
char* grow(size_t bytes, void* user) {
    Arena* a = static_cast<Arena*>(user);
    if (bytes > a->capacity) {
        if (a->base) (void)hipFree(a->base);
        a->base = nullptr;
        const size_t want = bytes + bytes / 4;
        if (hipMalloc(reinterpret_cast<void**>(&a->base), want) != hipSuccess) return nullptr;
        a->capacity = want;
    }
    return a->base;
}

struct Slot {  // everything one in-flight frame owns
    hipStream_t stream = nullptr;
    Arena geom, binning, image;
    float *color = nullptr, *depth = nullptr, *alpha = nullptr;
    int* radii = nullptr;
    void* call = nullptr;
};

template <typename T>
bool read_array(FILE* f, std::vector<T>& v, size_t n) {
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

template <typename T>
hipError_t upload(const std::vector<T>& h, T** d) {
    *d = nullptr;
    if (h.empty()) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

} // namespace

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s scene.bin out.bin frames [streams=3]\n", argv[0]);
        return 1;
    }
    const int frames = atoi(argv[3]);
    const int S = argc > 4 ? atoi(argv[4]) : 3;
    if (frames < 1 || S < 1 || S > 16) { fprintf(stderr, "bad frames / streams\n"); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t head[6];
    float params[3];
    if (fread(head, sizeof head, 1, f) != 1 || fread(params, sizeof params, 1, f) != 1) { fprintf(stderr, "short header\n"); return 1; }
    const int P = head[0], M = head[1], D = head[2], W = head[3], H = head[4], prefiltered = head[5];
    std::vector<float> bg, means, shs, opac, scales, rots, view, proj, campos;
    const size_t n = (size_t)P;
    if (!read_array(f, bg, 3) || !read_array(f, means, 3 * n) || !read_array(f, shs, 3 * n * M) || !read_array(f, opac, n) ||
        !read_array(f, scales, 3 * n) || !read_array(f, rots, 4 * n) || !read_array(f, view, 16) || !read_array(f, proj, 16) ||
        !read_array(f, campos, 3)) {
        fprintf(stderr, "%s: truncated\n", argv[1]);
        return 1;
    }
    fclose(f);

    float *d_bg, *d_means, *d_shs, *d_opac, *d_scales, *d_rots, *d_view, *d_proj, *d_campos;
    HIP_OK(upload(bg, &d_bg)); HIP_OK(upload(means, &d_means)); HIP_OK(upload(shs, &d_shs)); HIP_OK(upload(opac, &d_opac));
    HIP_OK(upload(scales, &d_scales)); HIP_OK(upload(rots, &d_rots)); HIP_OK(upload(view, &d_view));
    HIP_OK(upload(proj, &d_proj)); HIP_OK(upload(campos, &d_campos));

    const size_t px = (size_t)W * H;
    std::vector<Slot> slots(S);
    for (Slot& s : slots) {
        HIP_OK(hipStreamCreate(&s.stream));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&s.color), 3 * px * sizeof(float)));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&s.depth), px * sizeof(float)));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&s.alpha), px * sizeof(float)));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&s.radii), (n ? n : 1) * sizeof(int)));
        HIP_OK(hipMemset(s.color, 0, 3 * px * sizeof(float)));
        HIP_OK(hipMemset(s.depth, 0, px * sizeof(float)));
        HIP_OK(hipMemset(s.alpha, 0, px * sizeof(float)));
    }
    HIP_OK(hipDeviceSynchronize());

    auto begin = [&](Slot& s) {
        s.call = gsr_forward_begin(grow, &s.geom, grow, &s.binning, grow, &s.image, P, D, M, d_bg, W, H, d_means,
                                   M ? d_shs : nullptr, nullptr, d_opac, d_scales, params[2], d_rots, nullptr, d_view, d_proj,
                                   d_campos, params[0], params[1], prefiltered, s.color, s.depth, s.alpha, s.radii,
                                   /*extra_features=*/nullptr, /*out_extra=*/nullptr, GSR_FORWARD_INFERENCE, /*debug=*/0, s.stream);
        return s.call != nullptr;
    };
    int rendered = 0;
    auto finish = [&](Slot& s) {
        rendered = gsr_forward_finish(s.call);
        s.call = nullptr;
        return rendered >= 0;
    };

    // `warm` untimed frames size the arenas of every slot, then `frames` timed ones
    const int warm = 2 * S;
    std::chrono::steady_clock::time_point t0;
    for (int i = 0; i < warm + frames; ++i) {
        if (i == warm) {
            for (int k = 0; k < S; ++k)
                if (slots[(i + k) % S].call && !finish(slots[(i + k) % S])) { fprintf(stderr, "finish: %s\n", gsr_last_error()); return 3; }
            HIP_OK(hipDeviceSynchronize());
            t0 = std::chrono::steady_clock::now();
        }
        Slot& s = slots[i % S];
        if (s.call && !finish(s)) { fprintf(stderr, "finish: %s\n", gsr_last_error()); return 3; }   // the oldest frame
        if (!begin(s)) { fprintf(stderr, "begin: %s\n", gsr_last_error()); return 3; }
    }
    for (int k = 0; k < S; ++k) {   // drain in frame order
        Slot& s = slots[(warm + frames + k) % S];
        if (s.call && !finish(s)) { fprintf(stderr, "finish: %s\n", gsr_last_error()); return 3; }
    }
    HIP_OK(hipDeviceSynchronize());
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fprintf(stderr, "%d Gaussians, %dx%d, %d streams, one host thread: %d frames in %.2f ms = %.1f frames/s, num_rendered %d\n", P,
            W, H, S, frames, sec * 1e3, frames / sec, rendered);

    const Slot& last = slots[(warm + frames - 1) % S];
    std::vector<float> color(3 * px), depth(px), alpha(px);
    std::vector<int32_t> radii(n);
    HIP_OK(hipMemcpy(color.data(), last.color, color.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(depth.data(), last.depth, depth.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(alpha.data(), last.alpha, alpha.size() * sizeof(float), hipMemcpyDeviceToHost));
    if (n) HIP_OK(hipMemcpy(radii.data(), last.radii, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    const int32_t nr = rendered;
    fwrite(&nr, sizeof nr, 1, o);
    fwrite(color.data(), sizeof(float), color.size(), o);
    fwrite(depth.data(), sizeof(float), depth.size(), o);
    fwrite(alpha.data(), sizeof(float), alpha.size(), o);
    fwrite(radii.data(), sizeof(int32_t), radii.size(), o);
    fclose(o);
    return 0;
}
