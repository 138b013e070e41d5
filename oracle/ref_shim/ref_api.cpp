// ref_api.cpp -- C entry points over the REFERENCE's own host orchestration
// (CudaRasterizer::Rasterizer::forward / markVisible, DGR/cuda_rasterizer/rasterizer_impl.cu:141-153,
// 197-339), compiled for the host by oracle/build_ref.py.  TEST INFRASTRUCTURE ONLY.
//
// Scratch arenas are plain host vectors; intermediates are decoded with the reference's own
// GeometryState / BinningState / ImageState::fromChunk (rasterizer_impl.cu:155-193) so no layout is
// restated here.
#include "rasterizer_impl.h"

#include <cstring>
#include <vector>

using namespace CudaRasterizer;

namespace {
std::function<char*(size_t)> arena(std::vector<char>& v) {
    return [&v](size_t n) { v.assign(n + 256, 0); return v.data(); };
}
} // namespace

extern "C" {

// Mirrors RasterizeGaussiansCUDA (DGR/rasterize_points.cu:36-119) minus torch: outputs must arrive
// zero-filled, P == 0 touches nothing.  Optional intermediates may be NULL.
long long gsr_ref_forward(int P, int D, int M, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp,
                          const float* opacities, const float* scales, float scale_modifier,
                          const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                          float* out_color, float* out_depth, float* out_alpha, int* radii,
                          float* o_means2D, float* o_depths, float* o_conic_opacity, float* o_rgb,
                          unsigned* o_tiles, unsigned* o_offsets, unsigned* o_ncontrib, size_t cap,
                          unsigned long long* keys_out, unsigned* list_out, unsigned* ranges_out) {
    if (P == 0) return 0;
    std::vector<char> geom, binning, img;
    const int n = Rasterizer::forward(arena(geom), arena(binning), arena(img), P, D, M, background, width, height,
                                      means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                      cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, false,
                                      out_color, out_depth, out_alpha, radii, false);
    char* gp = geom.data();
    GeometryState g = GeometryState::fromChunk(gp, P);
    char* ip = img.data();
    ImageState im = ImageState::fromChunk(ip, (size_t)width * height);
    const size_t np = (size_t)P;
    // rows of culled Gaussians were never written by the kernels (arena is zero-filled here)
    if (o_means2D) memcpy(o_means2D, g.means2D, np * sizeof(float2));
    if (o_depths) memcpy(o_depths, g.depths, np * sizeof(float));
    if (o_conic_opacity) memcpy(o_conic_opacity, g.conic_opacity, np * sizeof(float4));
    if (o_rgb) memcpy(o_rgb, g.rgb, np * 3 * sizeof(float));
    if (o_tiles) memcpy(o_tiles, g.tiles_touched, np * sizeof(uint32_t));
    if (o_offsets) memcpy(o_offsets, g.point_offsets, np * sizeof(uint32_t));
    if (o_ncontrib) memcpy(o_ncontrib, im.n_contrib, (size_t)width * height * sizeof(uint32_t));
    const int T = ((width + 15) / 16) * ((height + 15) / 16);
    if (ranges_out) memcpy(ranges_out, im.ranges, (size_t)T * sizeof(uint2));
    if (n > 0 && (size_t)n <= cap) {
        char* bp = binning.data();
        BinningState b = BinningState::fromChunk(bp, n);
        if (keys_out) memcpy(keys_out, b.point_list_keys, (size_t)n * sizeof(uint64_t));
        if (list_out) memcpy(list_out, b.point_list, (size_t)n * sizeof(uint32_t));
    }
    return n;
}

void gsr_ref_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                          unsigned char* present) {
    if (P == 0) return;
    static_assert(sizeof(bool) == 1, "bool is one byte");
    Rasterizer::markVisible(P, const_cast<float*>(means3D), const_cast<float*>(viewmatrix),
                            const_cast<float*>(projmatrix), reinterpret_cast<bool*>(present));
}

} // extern "C"

// Forward then backward through the reference's own host code (rasterizer_impl.cu:197-339, 343-446),
// keeping its three scratch arenas alive in between, the way RasterizeGaussiansCUDA /
// RasterizeGaussiansBackwardCUDA (rasterize_points.cu:36-119, 121-209) do through ctx.saved_tensors.
extern "C" long long gsr_ref_forward_backward(
    int P, int D, int M, const float* background, int width, int height, const float* means3D, const float* shs,
    const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
    const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
    const float* cam_pos, float tan_fovx, float tan_fovy, const float* dL_dout_color, const float* dL_dout_depth,
    const float* dL_dout_alpha, float* out_color, float* out_depth, float* out_alpha, int* radii, float* dL_dmeans2D,
    float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
    float* dL_drots, float* dL_dconic, float* dL_ddepths) {
    if (P == 0) return 0;
    std::vector<char> geom, binning, img;
    const int R = Rasterizer::forward(arena(geom), arena(binning), arena(img), P, D, M, background, width, height,
                                      means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                      cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, false,
                                      out_color, out_depth, out_alpha, radii, false);
    Rasterizer::backward(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier,
                         rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, radii,
                         geom.data(), binning.data(), img.data(), out_alpha, dL_dout_color, dL_dout_depth,
                         dL_dout_alpha, dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors, dL_ddepths, dL_dmeans3D,
                         dL_dcov3D, dL_dsh, dL_dscales, dL_drots, false);
    return R;
}
