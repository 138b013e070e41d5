// render_raw.cpp -- the C ABI of libgsr_hip.so used from plain C++ / HIP, no Python, no torch.
//
// What the reference's own binding does around CudaRasterizer::Rasterizer::forward
// (DGR/rasterize_points.cu:36-119): allocate outputs, hand the library three growable scratch buffers through
// callbacks, call it on a stream.  Here the buffers are hipMalloc'ed arenas that only ever grow.
//
//   render_raw <scene.bin> <out.bin> [repeat]
//
// scene.bin (little endian): int32 P, M, D, W, H, prefiltered; float32 tan_fovx, tan_fovy, scale_modifier;
//   then float32 arrays background[3], means3D[P*3], shs[P*M*3], opacities[P], scales[P*3], rotations[P*4],
//   viewmatrix[16], projmatrix[16], campos[3]   (tests/test_cabi_native.py writes it).
// out.bin: int32 num_rendered; float32 color[3*H*W], depth[H*W], alpha[H*W]; int32 radii[P].
#include <hip/hip_runtime.h>

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

struct Arena {  // resizeFunctional (rasterize_points.cu:27-33) without a tensor library: grow, never shrink
    char* base = nullptr;
    size_t capacity = 0;
};

char* grow(size_t bytes, void* user) {
    Arena* a = static_cast<Arena*>(user);
    if (bytes > a->capacity) {
        if (a->base) (void)hipFree(a->base);   // safe here: the example drives one stream and syncs between frames
        a->base = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&a->base), bytes) != hipSuccess) return nullptr;
        a->capacity = bytes;
    }
    return a->base;
}

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
    if (argc < 3) {
        fprintf(stderr, "usage: %s scene.bin out.bin [repeat]\n", argv[0]);
        return 1;
    }
    const int repeat = argc > 3 ? atoi(argv[3]) : 1;
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
    if (gsr_abi_version() != GSR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    float *d_bg, *d_means, *d_shs, *d_opac, *d_scales, *d_rots, *d_view, *d_proj, *d_campos;
    HIP_OK(upload(bg, &d_bg)); HIP_OK(upload(means, &d_means)); HIP_OK(upload(shs, &d_shs)); HIP_OK(upload(opac, &d_opac));
    HIP_OK(upload(scales, &d_scales)); HIP_OK(upload(rots, &d_rots)); HIP_OK(upload(view, &d_view));
    HIP_OK(upload(proj, &d_proj)); HIP_OK(upload(campos, &d_campos));

    const size_t px = (size_t)W * H;
    float *d_color, *d_depth, *d_alpha;
    int* d_radii = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_color), 3 * px * sizeof(float)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_depth), px * sizeof(float)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_alpha), px * sizeof(float)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_radii), (n ? n : 1) * sizeof(int)));
    // rasterize_points.cu:68-71 zero-fills its outputs; only P == 0 ever shows those zeros
    HIP_OK(hipMemset(d_color, 0, 3 * px * sizeof(float)));
    HIP_OK(hipMemset(d_depth, 0, px * sizeof(float)));
    HIP_OK(hipMemset(d_alpha, 0, px * sizeof(float)));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    Arena geom, binning, image;
    int rendered = 0;
    hipEvent_t t0, t1;
    HIP_OK(hipEventCreate(&t0)); HIP_OK(hipEventCreate(&t1));
    for (int r = 0; r < repeat; ++r) {
        if (r == 1) HIP_OK(hipEventRecord(t0, stream));   // the first call sizes the arenas
        rendered = gsr_forward(grow, &geom, grow, &binning, grow, &image, P, D, M, d_bg, W, H, d_means, M ? d_shs : nullptr,
                               nullptr, d_opac, d_scales, params[2], d_rots, nullptr, d_view, d_proj, d_campos, params[0],
                               params[1], prefiltered, d_color, d_depth, d_alpha, d_radii, /*debug=*/0, stream);
        if (rendered < 0) {
            fprintf(stderr, "gsr_forward failed (%d): %s\n", rendered, gsr_last_error());
            return 3;
        }
        HIP_OK(hipStreamSynchronize(stream));   // the arenas may be re-grown by the next call
    }
    if (repeat > 1) {
        HIP_OK(hipEventRecord(t1, stream));
        HIP_OK(hipEventSynchronize(t1));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, t0, t1));
        fprintf(stderr, "%d Gaussians, %dx%d: %.3f ms per frame over %d frames, num_rendered %d\n", P, W, H,
                ms / (repeat - 1), repeat - 1, rendered);
    }

    std::vector<float> color(3 * px), depth(px), alpha(px);
    std::vector<int32_t> radii(n);
    HIP_OK(hipMemcpy(color.data(), d_color, color.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(depth.data(), d_depth, depth.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(alpha.data(), d_alpha, alpha.size() * sizeof(float), hipMemcpyDeviceToHost));
    if (n) HIP_OK(hipMemcpy(radii.data(), d_radii, n * sizeof(int32_t), hipMemcpyDeviceToHost));
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
