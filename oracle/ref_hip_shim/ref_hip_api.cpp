// ref_hip_api.cpp -- C entry points over the REFERENCE's own host orchestration (CudaRasterizer::Rasterizer::forward,
// DGR/cuda_rasterizer/rasterizer_impl.cu:197-339) compiled for gfx950 by oracle/build_ref_hip.py.
// TEST / BENCH INFRASTRUCTURE ONLY.  All pointers are device pointers.  The three scratch arenas are device buffers
// that grow on demand and are kept between calls (what the torch caching allocator does for the reference's binding).
#include "rasterizer_impl.h"

#include <cstdio>

using namespace CudaRasterizer;

namespace {
struct Arena {
    char* p = nullptr;
    size_t cap = 0;
    char* get(size_t n) {
        if (n > cap) {
            if (p) (void)hipFree(p);
            cap = n + n / 4 + 4096;
            if (hipMalloc((void**)&p, cap) != hipSuccess) { p = nullptr; cap = 0; }
        }
        return p;
    }
};
Arena g_geom, g_binning, g_img;
std::function<char*(size_t)> fn(Arena& a) { return [&a](size_t n) { return a.get(n); }; }
} // namespace

extern "C" {

// Mirrors RasterizeGaussiansCUDA (DGR/rasterize_points.cu:36-119) minus torch.  Runs on the default stream like the
// reference; returns num_rendered.
int gsr_refhip_forward(int P, int D, int M, const float* background, int width, int height, const float* means3D,
                       const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, float* out_color,
                       float* out_depth, float* out_alpha, int* radii) {
    if (P == 0) return 0;
    return Rasterizer::forward(fn(g_geom), fn(g_binning), fn(g_img), P, D, M, background, width, height, means3D, shs,
                               colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                               projmatrix, cam_pos, tan_fovx, tan_fovy, false, out_color, out_depth, out_alpha, radii,
                               false);
}

// Backward of the last gsr_refhip_forward call (Rasterizer::backward, rasterizer_impl.cu:343-446) over the arenas that
// call left behind; the gradient buffers must arrive zero-filled, as the reference's binding provides them.
int gsr_refhip_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                        const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                        const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                        const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii,
                        const float* out_alpha, const float* dL_dpix, const float* dL_dpix_depth, const float* dL_dpix_alpha,
                        float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                        float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
    if (P == 0 || !g_geom.p || !g_binning.p || !g_img.p) return -1;
    Rasterizer::backward(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier,
                         rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii, g_geom.p,
                         g_binning.p, g_img.p, out_alpha, dL_dpix, dL_dpix_depth, dL_dpix_alpha, dL_dmean2D, dL_dconic,
                         dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, false);
    return 0;
}

// Device pointers into the last call's scratch (the reference's own fromChunk layout): n_contrib[H*W], ranges[T][2],
// point_list[num_rendered].
int gsr_refhip_last_lists(int P, int width, int height, int num_rendered, const unsigned** n_contrib,
                          const unsigned** ranges, const unsigned** point_list) {
    if (!g_img.p || !g_binning.p) return -1;
    char* ip = g_img.p;
    ImageState im = ImageState::fromChunk(ip, (size_t)width * height);
    char* bp = g_binning.p;
    BinningState b = BinningState::fromChunk(bp, (size_t)num_rendered);
    (void)P;
    *n_contrib = im.n_contrib;
    *ranges = reinterpret_cast<const unsigned*>(im.ranges);
    *point_list = b.point_list;
    return 0;
}

} // extern "C"
