// train_step_raw.cpp -- the raw-parameter pair of the C ABI (gsr_forward_raw / gsr_backward_raw) from plain C++ / HIP:
// one differentiable render of a model's RAW tensors and its gradients, no Python, no torch.
//
// What a training loop does per iteration around the rasterizer (scene_representation.py:495-520 -> render() ->
// loss.backward()): here the six tensors go in as GaussianModel stores them (scene/gaussian_model.py:49-56: _xyz, _scaling
// (log), _rotation (unnormalised), _opacity (logit), _features_dc, _features_rest), the activations and their chain rule happen
// inside the kernels, and the gradients come back with respect to those same six tensors.
//
//   train_step_raw <model.bin> <out.bin>
//
// model.bin (little endian): int32 P, M, D, W, H; float32 tan_fovx, tan_fovy; then float32 arrays background[3], xyz[P*3],
//   log_scales[P*3], rotations[P*4], opacity_logits[P], features_dc[P*3], features_rest[P*(M-1)*3], viewmatrix[16],
//   projmatrix[16], campos[3], dL_dcolor[3*H*W], dL_dnormal[3*H*W]        (tests/test_cabi_native.py writes it).
// out.bin: int32 num_rendered; float32 color[3*H*W], depth[H*W], alpha[H*W], normal[3*H*W]; int32 radii[P]; float32
//   dL_dxyz[P*3], dL_dlog_scales[P*3], dL_drotations[P*4], dL_dopacity_logits[P], dL_dfeatures_dc[P*3],
//   dL_dfeatures_rest[P*(M-1)*3], dL_dmean2D[P*3].
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

char* grow(size_t bytes, void* user) {
    Arena* a = static_cast<Arena*>(user);
    if (bytes > a->capacity) {
        if (a->base) (void)hipFree(a->base);
        a->base = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&a->base), bytes) != hipSuccess) return nullptr;
        a->capacity = bytes;
    }
    return a->base;
}

struct DeviceArray {
    float* d = nullptr;
    size_t n = 0;
};

bool load(FILE* f, DeviceArray& a, size_t n) {   // file -> device
    a.n = n;
    if (n == 0) return true;
    std::vector<float> h(n);
    if (fread(h.data(), sizeof(float), n, f) != n) return false;
    if (hipMalloc(reinterpret_cast<void**>(&a.d), n * sizeof(float)) != hipSuccess) return false;
    return hipMemcpy(a.d, h.data(), n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
}

bool fresh(DeviceArray& a, size_t n) {   // an output: 0xFF bytes, so that an element the library fails to write shows up as NaN
    a.n = n;
    if (n == 0) return true;
    if (hipMalloc(reinterpret_cast<void**>(&a.d), n * sizeof(float)) != hipSuccess) return false;
    return hipMemset(a.d, 0xFF, n * sizeof(float)) == hipSuccess;
}

bool dump(FILE* o, const DeviceArray& a) {
    if (a.n == 0) return true;
    std::vector<float> h(a.n);
    if (hipMemcpy(h.data(), a.d, a.n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return false;
    return fwrite(h.data(), sizeof(float), a.n, o) == a.n;
}

} // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s model.bin out.bin\n", argv[0]);
        return 1;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t head[5];
    float fov[2];
    if (fread(head, sizeof head, 1, f) != 1 || fread(fov, sizeof fov, 1, f) != 1) { fprintf(stderr, "short header\n"); return 1; }
    const int P = head[0], M = head[1], D = head[2], W = head[3], H = head[4];
    const size_t n = (size_t)P, px = (size_t)W * H;
    if (gsr_abi_version() != GSR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    // an optional third argument "deterministic": gradient sums in a fixed order (GSR_OPT_BACKWARD_DETERMINISTIC) -- two runs, or this
    // program and any other caller of the library on the same tensors, then produce the same bits
    if (argc > 3 && strcmp(argv[3], "deterministic") == 0 && gsr_set_option(GSR_OPT_BACKWARD_DETERMINISTIC, 1) != GSR_OK) {
        fprintf(stderr, "gsr_set_option: %s\n", gsr_last_error());
        return 1;
    }
    DeviceArray bg, xyz, ls, rot, op, dc, rest, view, proj, campos, g_color, g_normal;
    if (!load(f, bg, 3) || !load(f, xyz, 3 * n) || !load(f, ls, 3 * n) || !load(f, rot, 4 * n) || !load(f, op, n) || !load(f, dc, 3 * n) ||
        !load(f, rest, 3 * n * (size_t)(M - 1)) || !load(f, view, 16) || !load(f, proj, 16) || !load(f, campos, 3) ||
        !load(f, g_color, 3 * px) || !load(f, g_normal, 3 * px)) {
        fprintf(stderr, "%s: truncated or out of memory\n", argv[1]);
        return 1;
    }
    fclose(f);

    DeviceArray color, depth, alpha, normal, g_xyz, g_ls, g_rot, g_op, g_dc, g_rest, g_2d, accum;
    int* d_radii = nullptr;
    if (!fresh(color, 3 * px) || !fresh(depth, px) || !fresh(alpha, px) || !fresh(normal, 3 * px) || !fresh(g_xyz, 3 * n) ||
        !fresh(g_ls, 3 * n) || !fresh(g_rot, 4 * n) || !fresh(g_op, n) || !fresh(g_dc, 3 * n) || !fresh(g_rest, 3 * n * (size_t)(M - 1)) ||
        !fresh(g_2d, 3 * n) || !fresh(accum, 16 * n)) {
        fprintf(stderr, "out of device memory\n");
        return 2;
    }
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_radii), (n ? n : 1) * sizeof(int)));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    Arena geom, binning, image;
    const gsr_raw_params raw = {xyz.d, ls.d, rot.d, op.d, dc.d, rest.d};
    // forward: a FULL call (flags = 0) -- its scratch is what the backward differentiates
    const int rendered = gsr_forward_raw(grow, &geom, grow, &binning, grow, &image, P, D, M, bg.d, W, H, &raw, 1.0f, view.d, proj.d,
                                         campos.d, fov[0], fov[1], /*prefiltered=*/0, color.d, depth.d, alpha.d, d_radii, normal.d,
                                         /*flags=*/0u, /*debug=*/0, stream);
    if (rendered < 0) {
        fprintf(stderr, "gsr_forward_raw failed (%d): %s\n", rendered, gsr_last_error());
        return 3;
    }
    // backward: the loss reads the colour and the normal image, not depth or alpha (NULL: their terms are skipped)
    const int rc = gsr_backward_raw(P, D, M, rendered, bg.d, W, H, &raw, 1.0f, view.d, proj.d, campos.d, fov[0], fov[1], d_radii,
                                    geom.base, binning.base, image.base, alpha.d, g_color.d, nullptr, nullptr, g_normal.d, g_2d.d,
                                    g_xyz.d, g_ls.d, g_rot.d, g_op.d, g_dc.d, M > 1 ? g_rest.d : nullptr, accum.d, /*debug=*/0, stream);
    if (rc != 0) {
        fprintf(stderr, "gsr_backward_raw failed (%d): %s\n", rc, gsr_last_error());
        return 3;
    }
    HIP_OK(hipStreamSynchronize(stream));

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    const int32_t nr = rendered;
    fwrite(&nr, sizeof nr, 1, o);
    std::vector<int32_t> radii(n);
    if (n) HIP_OK(hipMemcpy(radii.data(), d_radii, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    bool ok = dump(o, color) && dump(o, depth) && dump(o, alpha) && dump(o, normal);
    ok = ok && fwrite(radii.data(), sizeof(int32_t), radii.size(), o) == radii.size();
    ok = ok && dump(o, g_xyz) && dump(o, g_ls) && dump(o, g_rot) && dump(o, g_op) && dump(o, g_dc) && dump(o, g_rest) && dump(o, g_2d);
    fclose(o);
    return ok ? 0 : 4;
}
