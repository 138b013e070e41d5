/*
 * gsr.h -- C ABI of the MI355X-native 3D-Gaussian-splatting forward rasterizer (libgsr_hip.so).
 *
 * This is the drop-in boundary for the reference's device library.  Each entry point replaces
 * one member of the reference interface (paths under /root/reference, DGR = sugar/
 * gaussian_splatting/submodules/diff-gaussian-rasterization):
 *
 *   gsr_forward        <- CudaRasterizer::Rasterizer::forward   DGR/cuda_rasterizer/rasterizer.h:31-55
 *                         (impl DGR/cuda_rasterizer/rasterizer_impl.cu:197-339), as called by
 *                         RasterizeGaussiansCUDA               DGR/rasterize_points.cu:36-119
 *   gsr_mark_visible   <- CudaRasterizer::Rasterizer::markVisible DGR/cuda_rasterizer/rasterizer.h:24-29
 *                         (impl rasterizer_impl.cu:141-153), as called by markVisible
 *                         DGR/rasterize_points.cu:211-230
 *   gsr_backward       <- CudaRasterizer::Rasterizer::backward  DGR/cuda_rasterizer/rasterizer.h:57-90
 *                         (impl rasterizer_impl.cu:343-446), as called by RasterizeGaussiansBackwardCUDA
 *                         DGR/rasterize_points.cu:121-209
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch / C++ types cross this boundary;
 *   - every data pointer is a DEVICE pointer (HIP, gfx950) unless the name says host;
 *   - the std::function<char*(size_t)> scratch callbacks of the reference become a C function
 *     pointer plus a user cookie; each is called exactly once per gsr_forward, like the reference;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it.  gsr_forward blocks
 *     the host once (to learn num_rendered before sizing the binning scratch), exactly where the
 *     reference does (rasterizer_impl.cu:282);
 *   - errors never propagate as C++ exceptions: functions return a negative gsr_status and
 *     gsr_last_error() describes the failure (thread-local string).
 *
 * Tensor layouts are the reference's: means3D[P,3], shs[P,M,3],
 * colors_precomp[P,3], opacities[P], scales[P,3], rotations[P,4] (w,x,y,z), cov3D_precomp[P,6],
 * viewmatrix / projmatrix = 16 floats of the transposed matrices, out_color[3,H,W] planar,
 * out_depth[H,W], out_alpha[H,W], radii[P] int32.  "nullable" pointers select the alternative
 * input exactly as the reference's nullptr tests do (forward.cu:205,241).
 */
#ifndef GSR_H_INCLUDED
#define GSR_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 14 (round 6): + the compositor's input files -- gsr_png_unfilter(_batch), gsr_exr_unpack_channel, gsr_inflate_zlib_blocks, the host-side
 * gsr_png_file_* / gsr_exr_file_* readers, gsr_upload.  13: + compressed frame files (gsr_png_encode_deflate, gsr_frame_files_deflate).
 * Additions only: a binding written against 12 works unchanged apart from the version it checks. */
#define GSR_ABI_VERSION 14

#if defined(__GNUC__)
#define GSR_API __attribute__((visibility("default")))
#else
#define GSR_API
#endif

typedef enum gsr_status {
    GSR_OK = 0,
    GSR_ERR_INVALID_ARG = -1,  /* bad sizes / missing required pointer / both alternatives given   */
    GSR_ERR_ALLOC = -2,        /* a scratch callback returned NULL                                  */
    GSR_ERR_HIP = -3,          /* a HIP runtime call or kernel failed (message has the HIP string) */
    GSR_ERR_UNSUPPORTED = -4,  /* entry point declared but not built yet                            */
    GSR_ERR_PREFILTERED = -5,  /* prefiltered=1 but a Gaussian failed the near-plane test (debug)   */
    GSR_ERR_INTERNAL = -6      /* debug=1 only: a per-tile list came out of the sorts in the wrong order  */
} gsr_status;

/* Scratch provider: must return a device pointer to at least `nbytes` bytes that stays valid until
 * the work enqueued by the current call has finished (rasterize_points.cu:27-33). */
typedef char* (*gsr_alloc_fn)(size_t nbytes, void* user);

/*
 * Forward rasterization.  Returns num_rendered (>= 0: number of (tile, Gaussian) pairs, the
 * reference's return value) or a negative gsr_status.
 *
 *   P  number of Gaussians;  D  active SH degree;  M  SH coefficients per Gaussian (0 if no shs)
 *   exactly one of {shs, colors_precomp} and exactly one of {scales+rotations, cov3D_precomp}
 *   radii may be NULL (then an internal array is used), every other output is required.
 *   P == 0 is legal and returns 0 without touching any output (rasterize_points.cu:83).
 */
GSR_API int gsr_forward(gsr_alloc_fn geom_alloc, void* geom_user,
                gsr_alloc_fn binning_alloc, void* binning_user,
                gsr_alloc_fn image_alloc, void* image_user,
                int P, int D, int M,
                const float* background, int width, int height,
                const float* means3D, const float* shs /*nullable*/,
                const float* colors_precomp /*nullable*/, const float* opacities,
                const float* scales /*nullable*/, float scale_modifier,
                const float* rotations /*nullable*/, const float* cov3D_precomp /*nullable*/,
                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered,
                float* out_color, float* out_depth, float* out_alpha, int* radii /*nullable*/,
                int debug, void* stream);

/* Per-call flags of gsr_forward_extra / gsr_forward_begin (gsr_forward itself always makes a full call: 0).
 *
 * GSR_FORWARD_INFERENCE: the caller will not run gsr_backward on this call's scratch (the reference's binding cannot
 * know that; a Python binding can: nothing requires a gradient).  The images, radii and the returned num_rendered are
 * bit-identical with or without the flag; what changes is how the lists are built:
 *   - the depth-sorted splats are expanded, sorted and blended in front-to-back depth SLABS, and a slab drops every
 *     (tile, Gaussian) pair whose tile has all its 256 pixels finished by the slabs in front of it -- the reference's
 *     block-wide early exit (forward.cu:312-314) applied before the pair is written instead of after it was sorted;
 *   - SH colours are evaluated only for the splats that reach a list (the reference evaluates every visible one).
 * The scratch then holds one list per slab (gsr_blend walks them; gsr_backward rejects them with a clear error). */
#define GSR_FORWARD_INFERENCE 1u
#define GSR_MAX_SLABS 8

/* gsr_forward with a SECOND per-Gaussian feature triple composited in the same walk of the per-tile lists:
 * out_extra[3,H,W] = sum_i extra_features[i] * alpha_i * T_i + T_final * background, exactly what a second
 * gsr_forward call with colors_precomp = extra_features would put into its out_color (same alpha, same
 * transmittance, same operation order), for the cost of three more multiply-adds per contributing pair instead of
 * a second pass.  The reference's render() makes that second pass for its normal map
 * (sugar/gaussian_splatting/gaussian_renderer/__init__.py:176-184). */
GSR_API int gsr_forward_extra(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                              gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                              int width, int height, const float* means3D, const float* shs /*nullable*/,
                              const float* colors_precomp /*nullable*/, const float* opacities,
                              const float* scales /*nullable*/, float scale_modifier, const float* rotations /*nullable*/,
                              const float* cov3D_precomp /*nullable*/, const float* viewmatrix, const float* projmatrix,
                              const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color,
                              float* out_depth, float* out_alpha, int* radii /*nullable*/,
                              const float* extra_features /*[P,3], nullable*/, float* out_extra /*[3,H,W], nullable*/,
                              unsigned flags, int debug, void* stream);

/* gsr_forward_extra in two halves, for callers that keep several frames in flight from ONE host thread.
 *
 * gsr_forward has one host round trip in the middle (the pair count sizes the binning arena: rasterizer_impl.cu:282),
 * so a caller that wants frame i+1's projection and depth sort queued while frame i is still on the GPU needs either
 * a thread per stream or this split:
 *
 *   gsr_forward_begin   validates, calls the geometry and image callbacks, queues projection + depth sort and the
 *                       copy of the frame counters on `stream`, and returns an opaque call handle without waiting
 *                       (NULL on error, gsr_last_error() says why).  extra_features / out_extra may both be NULL.
 *   gsr_forward_finish  waits for that call's counters only, calls the binning callback, queues the remaining
 *                       stages, frees the handle (also on error) and returns what gsr_forward returns.
 *   gsr_forward_ready   1 when the call's counters have reached the host (gsr_forward_finish will not wait), 0 when not
 *                       yet, negative on error: lets a driver finish frames as they become ready instead of in lock step.
 *   gsr_forward_cancel  frees a handle whose finish will not be called (outputs are then undefined).
 *
 * begin immediately followed by finish IS gsr_forward_extra: same launches, same results.  Every pointer passed to
 * begin must stay valid until finish has been called and `stream` has drained; finish must be called on the
 * thread that called begin (the gsr_last_* accessors then describe that call). */
GSR_API void* gsr_forward_begin(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                                gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                                int width, int height, const float* means3D, const float* shs /*nullable*/,
                                const float* colors_precomp /*nullable*/, const float* opacities,
                                const float* scales /*nullable*/, float scale_modifier,
                                const float* rotations /*nullable*/, const float* cov3D_precomp /*nullable*/,
                                const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_alpha,
                                int* radii /*nullable*/, const float* extra_features /*nullable*/,
                                float* out_extra /*nullable*/, unsigned flags, int debug, void* stream);
GSR_API int gsr_forward_finish(void* call);
GSR_API int gsr_forward_ready(void* call);
GSR_API void gsr_forward_cancel(void* call);

/*
 * The forward pass straight from a model's RAW parameter tensors (second half).
 *
 * The reference's render() (sugar/gaussian_splatting/gaussian_renderer/__init__.py:83-218) re-activates the whole model
 * in PyTorch on every frame before it calls the rasterizer -- exp(_scaling), sigmoid(_opacity), normalize(_rotation),
 * cat(_features_dc, _features_rest) (scene/gaussian_model.py:95-128), the view direction (:118-119) and
 * get_normal(dir) * 0.5 + 0.5 via argsort / gather / build_rotation (utils/general_utils.py:78-101,135-157) -- about
 * 2.3 GB of framework traffic per frame at 3 M Gaussians, more than the rasterizer itself moves.  gsr_forward_raw takes
 * the six tensors as the model stores them and applies those operations inside the projection / colour kernels:
 *
 *   xyz            [P,3]      GaussianModel._xyz
 *   log_scales     [P,3]      ._scaling         scale    = exp(x)                          (gaussian_model.py:96-97)
 *   rotations      [P,4]      ._rotation        q        = x / max(||x||, 1e-12)           (:100-101, F.normalize)
 *   opacity_logits [P] / [P,1] ._opacity        opacity  = 1 / (1 + exp(-x))               (:125-126)
 *   features_dc    [P,1,3]    ._features_dc     SH coefficient 0
 *   features_rest  [P,M-1,3]  ._features_rest   SH coefficients 1 .. M-1 (NULL allowed iff M == 1); no concatenation
 *
 * every operation rounded as, and in the order, PyTorch-ROCm evaluates it on this GPU (identified on the device:
 * scripts/experiments/torch_op_probe.py; e.g. the norm of a quaternion is (x0^2 + x1^2) + (x2^2 + x3^2)), so radii, lists
 * and images are bit-identical to activating in PyTorch and calling gsr_forward_extra (tests/test_raw_gpu.py).
 *
 * out_normal (nullable) [3,H,W]: the reference's second rasterizer pass (:169-184) -- the per-Gaussian view normal
 * get_normal(dir) * 0.5 + 0.5 composited with the same alpha and transmittance -- worked out per Gaussian in the projection
 * kernel (into GSR_GEOM_VIEW_NORMALS of the geometry arena) and composited in the same walk of the lists as the colour.
 * Ties between a Gaussian's scales pick the axis torch.argsort's bitonic network would (gsr_device.h: min_axis).
 *
 * flags / debug / stream / scratch callbacks / return value: as gsr_forward_extra.  A full (non-inference) raw call leaves
 * scratch that gsr_backward accepts together with the activated tensors.  gsr_forward_raw_begin is to gsr_forward_raw what
 * gsr_forward_begin is to gsr_forward_extra; its handle goes to gsr_forward_finish / _ready / _cancel.
 */
typedef struct gsr_raw_params {
    const float* xyz;
    const float* log_scales;
    const float* rotations;
    const float* opacity_logits;
    const float* features_dc;
    const float* features_rest;
} gsr_raw_params;
GSR_API int gsr_forward_raw(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                            gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                            int width, int height, const gsr_raw_params* raw, float scale_modifier,
                            const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                            float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_alpha,
                            int* radii /*nullable*/, float* out_normal /*[3,H,W], nullable*/, unsigned flags, int debug,
                            void* stream);
GSR_API void* gsr_forward_raw_begin(gsr_alloc_fn geom_alloc, void* geom_user, gsr_alloc_fn binning_alloc, void* binning_user,
                                    gsr_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background,
                                    int width, int height, const gsr_raw_params* raw, float scale_modifier,
                                    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                    float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_alpha,
                                    int* radii /*nullable*/, float* out_normal /*nullable*/, unsigned flags, int debug,
                                    void* stream);

/* present[i] = (view-space z of means3D[i]) > 0.2 ; present is a device array of P bytes (bool). */
GSR_API int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/*
 * The blend stage on its own (renderCUDA, DGR/cuda_rasterizer/forward.cu:261-378): composite `features`
 * [P,3] over the per-tile lists of an earlier forward call.  gaussian_renderer.render() rasterizes every
 * frame twice over identical geometry -- SH colours, then normals as colors_precomp
 * (gaussian_renderer/__init__.py:151-159,176-184) -- and the reference recomputes projection, keys and the
 * sort for the second pass; with this entry point the second pass costs one kernel.
 *   geom/binning/image_buffer   the three scratch arenas of that call, untouched since (the pointers its scratch
 *                     callbacks returned); they are self-describing, any number of depth slabs is handled
 * The outputs are those of a forward call with colors_precomp = features over the same geometry, bit for bit.
 */
GSR_API int gsr_blend(const char* geom_buffer, const char* binning_buffer, const char* image_buffer, int width, int height,
                      const float* features, const float* background, float* out_color, float* out_depth,
                      float* out_alpha, void* stream);

/*
 * Compositor: the per-pixel layer arithmetic of blender/blend_all.py::blend_frames
 * (:236-300, :341-343) for ONE frame.  All colour layers are interleaved RGBA8 [H,W,4] and all depth maps fp32
 * [H,W], already at the output resolution (the reference resizes Blender's layers with PIL first, :217-234):
 *   bg_c   the 3DGS background frame (images/<n>.png)
 *   o_c, o_d      Blender object pass;  s_c, s_d  shadow-catcher pass;  o_s_c  object + shadow-catcher pass
 *   o_gs_c, o_gs_d  (nullable pair) 3DGS objects re-rendered by Blender
 *   s_f_c, s_f_d    (nullable pair) smoke / fire, s_f_d already filled with its 0.001-th percentile where
 *                   the layer has alpha (:207-215);  s_f_c_pre (nullable) premultiplied fire
 * out is RGBA8 [H,W,4], bit-identical to the reference's frame.
 */
GSR_API int gsr_composite(int width, int height, const uint8_t* bg_c, const uint8_t* o_c, const float* o_d,
                          const uint8_t* s_c, const float* s_d, const uint8_t* o_s_c,
                          const uint8_t* o_gs_c /*nullable*/, const float* o_gs_d /*nullable*/,
                          const uint8_t* s_f_c /*nullable*/, const float* s_f_d /*nullable*/,
                          const uint8_t* s_f_c_pre /*nullable*/, uint8_t* out, void* stream);

/* Frame hand-off used by the trajectory driver: planar fp32 color[3,H,W] + alpha[H,W] -> planar
 * uint8 rgba8[4,H,W], quantised as torchvision.utils.save_image does for the RGBA PNGs the reference
 * writes (scene_representation.py:427): clamp(x * 255 + 0.5, 0, 255), truncated. */
GSR_API int gsr_pack_rgba8(const float* color, const float* alpha, uint8_t* rgba8, int width, int height,
                           void* stream);

/* Frame files at render rate (the reference writes four files per frame, scene_representation.py:425-438: an RGBA PNG through
 * torchvision.utils.save_image, the depth map as .npy, and two PNGs through cv2.imwrite).  gsr_png_encode turns an 8-bit image
 * that lives on the GPU -- interleaved [H,W,C] or, planar != 0, [C,H,W] as gsr_pack_rgba8 leaves it; C = 3 or 4 -- into the
 * bytes of its PNG FILE, on the GPU: 8-bit truecolour (with alpha), one IDAT chunk holding a zlib stream of stored deflate
 * blocks (filter-0 scanlines, no compression), Adler-32 and CRC-32 computed in the kernel.  Any PNG reader decodes it to the
 * same pixels as the reference's compressed file; the caller copies gsr_png_size(...) bytes to the host and writes them out.
 * `out` must hold gsr_png_room(width, height, channels) bytes -- the file, then from the next 16-byte boundary on the kernels'
 * per-workgroup partial checksums (16 bytes per 4 KB of file, 4 per 16 KB) -- and be 16-byte aligned.  Both return 0 for sizes
 * that cannot be encoded.  Three launches, no memset, no atomics. */
GSR_API size_t gsr_png_size(int width, int height, int channels);
GSR_API size_t gsr_png_room(int width, int height, int channels);
/* ... and one frame's four files in one call (ten launches on `stream`): color [3,H,W] + alpha [H,W] -> the RGBA PNG (save_image's
 * rounding, gsr_pack_rgba8); depth [H,W] -> the turbo-coloured preview PNG (depth2img(depth, depth_scale): uint8(clip(d / scale,
 * 0, 1) * 255) through turbo_lut, 256 RGB triples on the device) and, copied behind the caller's .npy header, the fp32 plane;
 * normal [H,W,3] -> uint8((n + 1) / 2 * 255), truncated.  png_* as `out` of gsr_png_encode (gsr_png_room(w, h, 4 / 3 / 3) bytes,
 * 16-byte aligned); npy_plane: W * H floats; work: 10 * W * H bytes of device scratch. */
GSR_API int gsr_frame_files(const float* color, const float* alpha, const float* depth, const float* normal, float depth_scale,
                            const uint8_t* turbo_lut, int width, int height, uint8_t* png_rgba, uint8_t* png_depth_preview,
                            uint8_t* png_normal, float* npy_plane, uint8_t* work, void* stream);
GSR_API int gsr_png_encode(const uint8_t* pixels, int width, int height, int channels, int planar, uint8_t* out, void* stream);
/* The same files COMPRESSED, as the reference's are (torchvision.utils.save_image -> PIL -> zlib; cv2.imwrite -> libpng:
 * scene_representation.py:427,433,438) -- ABI 13.  The IDAT holds dynamic-Huffman deflate blocks of Paeth-filtered scanlines
 * (PNG filter type 4) with run-length matches (distance 1): one Huffman code per image, built on the GPU from the image's token
 * histogram; blocks of 16 KB of filtered bytes, each ending on a byte boundary (an empty stored block, zlib's sync-flush marker),
 * a block that would not shrink is sent stored.  Any PNG reader decodes the reference's pixels; the file is 1.0 - 1.2 x the size
 * PIL writes at its default level for rendered frames.  The file's length depends on the image: it is written to *out_len
 * (DEVICE memory, 8-byte aligned; at most gsr_png_deflate_max_size) by the kernels -- copy it back with the file.  `out` must hold
 * gsr_png_deflate_room(...) bytes (the file's bound and 64 bytes the checksum kernel may read behind it), `scratch`
 * gsr_png_deflate_scratch(...) bytes (filtered scanlines, per-block histograms / offsets, checksum partials); both 16-byte aligned.
 * Six launches, no memset, no global atomics.  gsr_frame_files_deflate is gsr_frame_files with the three PNGs compressed -- each
 * kernel launched once for all three --, png_scratch of gsr_png_deflate_scratch(w, h, 4) + 2 * gsr_png_deflate_scratch(w, h, 3)
 * bytes and the files' lengths in png_lengths[3] (device; order: RGBA, depth preview, normal). */
GSR_API size_t gsr_png_deflate_max_size(int width, int height, int channels);
GSR_API size_t gsr_png_deflate_room(int width, int height, int channels);
GSR_API size_t gsr_png_deflate_scratch(int width, int height, int channels);
GSR_API int gsr_png_encode_deflate(const uint8_t* pixels, int width, int height, int channels, int planar, uint8_t* out,
                                   uint8_t* scratch, uint64_t* out_len, void* stream);
GSR_API int gsr_frame_files_deflate(const float* color, const float* alpha, const float* depth, const float* normal, float depth_scale,
                                    const uint8_t* turbo_lut, int width, int height, uint8_t* png_rgba, uint8_t* png_depth_preview,
                                    uint8_t* png_normal, float* npy_plane, uint8_t* work, uint8_t* png_scratch, uint64_t* png_lengths,
                                    void* stream);

/* The compositor's input side (blender/blend_all.py:21-28,217-234: every Blender layer of every frame is brought to the size of
 * the rendered frame with PIL -- Image.resize(new_size, BILINEAR) for the RGBA8 layers, Image.resize(new_size, NEAREST) for the
 * fp32 depth maps).  Both on the GPU, BIT-IDENTICAL to Pillow (12.x): the RGBA resize premultiplies, resamples in two 8-bit
 * fixed-point passes with the triangle filter stretched by the scale factor, and un-premultiplies, exactly as Image.resize does;
 * the nearest resize picks Pillow's source indices (accumulated in double).  src / dst: [H,W,4] u8 or [H,W] f32, contiguous.
 * tmp: src_h * dst_w * 4 bytes of device memory (used when width AND height change; may be NULL otherwise). */
GSR_API int gsr_resize_rgba8_bilinear(const uint8_t* src, int src_width, int src_height, uint8_t* dst, int dst_width, int dst_height,
                                      uint8_t* tmp, void* stream);
GSR_API int gsr_resize_f32_nearest(const float* src, int src_width, int src_height, float* dst, int dst_width, int dst_height,
                                   void* stream);

/* ABI 14.  The compositor's input FILES (blender/blend_all.py:56-75,185-205: load_rgb -- Image.open(path).convert("RGBA") -- for six PNG layers
 * and load_depth_exr -- cv2.imread(path, ANYCOLOR | ANYDEPTH)[:, :, 0] -- for four OpenEXR depth passes per frame, at Blender's
 * resolution).  Inflating a file's zlib stream(s) stays with the caller (zlib: byte-serial, one stream per PNG / per EXR block);
 * what is left -- undoing the image predictor -- runs here, on what the caller uploads (gsr_upload: hipMemcpyAsync from any host
 * memory, so that a binding needs no second runtime handle):
 *  gsr_png_unfilter: `scanlines` = the inflated IDAT data of an 8-bit RGB (channels 3) or RGBA (4), non-interlaced PNG: height rows of
 *   1 filter-type byte (0 ... 4; the caller checks them) + width * channels bytes, device memory.  out_rgba: [height, width, 4] u8,
 *   alpha 255 for RGB -- the pixels Image.open(...).convert("RGBA") returns.  scratch: gsr_png_unfilter_scratch(w, h) bytes (0: the
 *   width is over 4096 -- use a host decoder); both 16-byte aligned.  Two launches; one workgroup walks the image as a wavefront.
 *  gsr_exr_unpack_channel: `blocks` = the inflated, still predictor-coded scanline blocks (ZIP / ZIPS / RLE) of one part, one after
 *   another in increasing y, each lines * bytes_per_line bytes; plane: [height][channel_bytes] = the bytes of the channel stored at
 *   [channel_at, channel_at + channel_bytes) of every line (half / float / uint samples in the file's byte order). */
GSR_API size_t gsr_png_unfilter_scratch(int width, int height);
GSR_API int gsr_png_unfilter(const uint8_t* scanlines, int width, int height, int channels, uint8_t* out_rgba, uint8_t* scratch,
                             void* stream);
/* Several images in one pair of launches (the layers of a frame: a workgroup each, side by side). */
typedef struct GsrPngUnfilterJob {
    const uint8_t* scanlines;
    int width, height, channels;
    uint8_t* out_rgba;
    uint8_t* scratch;
} GsrPngUnfilterJob;
GSR_API int gsr_png_unfilter_batch(int count, const GsrPngUnfilterJob* jobs, void* stream);
GSR_API int gsr_exr_unpack_channel(const uint8_t* blocks, int height, int bytes_per_line, int lines_per_block, int channel_at,
                                   int channel_bytes, uint8_t* plane, void* stream);
GSR_API int gsr_upload(void* device_dst, const void* host_src, size_t bytes, void* stream);

/* The HOST side of the same files, one native call per file (no GPU involved): container parsing and zlib's inflate, writing what the
 * two kernels above take straight into the caller's buffer (page-locked, if gsr_upload is to be asynchronous).  `file`: the whole
 * file in memory.  Return GSR_OK for a file the kernels cover, GSR_NOT_COVERED (1) for any other flavour or a damaged file -- decode
 * those with the host library the reference uses (Pillow, OpenCV) --, a negative gsr_status for bad arguments.
 *  PNG: 8-bit truecolour with / without alpha, not interlaced, no tRNS, at most 4096 wide; IHDR / IDAT CRCs verified, the stream must
 *   inflate to exactly height * (1 + width * channels) bytes with filter types 0 ... 4.
 *  OpenEXR: version 2 single-part scanline files, RLE / ZIPS / ZIP, no sub-sampling, every block actually compressed; `channel`:
 *   the wanted channel, NULL for what cv2.imread(path, ANYCOLOR | ANYDEPTH)[:, :, 0] is (B, else G, R, Y, Z, V, else the first);
 *   it must hold HALF or FLOAT samples.  blocks: info.blocks_bytes = height * bytes_per_line bytes, the blocks in increasing y. */
#define GSR_NOT_COVERED 1
typedef struct GsrPngFileInfo {
    int width, height, channels;
    size_t scanline_bytes;
} GsrPngFileInfo;
typedef struct GsrExrFileInfo {
    int width, height, bytes_per_line, lines_per_block, channel_at, channel_bytes, channel_is_half;
    int compression, n_blocks; /* the compression attribute (1 RLE, 2 ZIPS, 3 ZIP); scanline blocks in the part */
    size_t blocks_bytes;
    char channel[32];
} GsrExrFileInfo;
GSR_API int gsr_png_file_probe(const uint8_t* file, size_t file_bytes, GsrPngFileInfo* info);
GSR_API int gsr_png_file_inflate(const uint8_t* file, size_t file_bytes, uint8_t* scanlines, size_t scanline_bytes);
GSR_API int gsr_exr_file_probe(const uint8_t* file, size_t file_bytes, const char* channel, GsrExrFileInfo* info);
GSR_API int gsr_exr_file_inflate(const uint8_t* file, size_t file_bytes, const char* channel, uint8_t* blocks, size_t blocks_bytes);
/* ZIP / ZIPS OpenEXR parts are one zlib stream per scanline block (68 for a 1080p ZIP pass): independent streams, so the GPU can
 * inflate them -- one single-wave workgroup per stream, the 32 KB window in LDS, matches copied and tables filled by the 64 lanes --
 * instead of the host (gsr_exr_file_inflate).
 *  gsr_exr_file_pack (host): the blocks' compressed streams copied one behind the other, each at a 4-byte aligned offset, into
 *   `packed` (room: file_bytes + 4 * info.n_blocks + 4), and jobs[info.n_blocks] = where each stream is and where its output goes.
 *   GSR_NOT_COVERED for RLE files and whatever gsr_exr_file_probe does not cover.
 *  gsr_inflate_zlib_blocks (GPU): `streams`, `out`, `jobs`, `status[count]` and `any_error` (may be NULL) are DEVICE pointers;
 *   stream i = streams[src_at, src_at + src_bytes) -> out[dst_at, dst_at + dst_bytes), which it must fill exactly; status[i] = 0 or
 *   why not (gsr_inflate_core.h: bad header / code / distance / checksum / sizes ...), *any_error |= 1 on any failure.  A stream that
 *   zlib's uncompress() accepts for that output size decodes to the same bytes; anything else is refused (Adler-32 verified). */
typedef struct GsrInflateJob {
    uint32_t src_at, src_bytes, dst_at, dst_bytes;
} GsrInflateJob;
GSR_API int gsr_exr_file_pack(const uint8_t* file, size_t file_bytes, const char* channel, uint8_t* packed, size_t packed_room,
                              GsrInflateJob* jobs, size_t* packed_bytes);
GSR_API int gsr_inflate_zlib_blocks(const uint8_t* streams, uint8_t* out, const GsrInflateJob* jobs, int count, int* status,
                                    int* any_error, void* stream);
/* The GPU's zlib decoder (gsr_inflate_zlib_blocks above) run by ONE host lane: the same source, for tests against zlib.  Returns 0 or
 * the decoder's status (1 bad header ... 10 short output: gsr_inflate_core.h).  out_bytes: the exact size the stream inflates to. */
GSR_API int gsr_selftest_inflate_host(const uint8_t* zlib_stream, size_t stream_bytes, uint8_t* out, size_t out_bytes);

/* The sort stage on its own (what gsr_forward runs twice per call; replaces the reference's
 * cub::DeviceRadixSort::SortPairs, rasterizer_impl.cu:304-309): stable ascending sort of n (u32 key, u32
 * payload) pairs on the low `bits` key bits.  *_alt are ping-pong partners of the same length; on return
 * *sorted_in_alt says which buffers hold the result.  iota_payload != 0: the payload is 0..n-1 and `vals`
 * is not read.  scratch: gsr_radix_scratch_bytes(n, bits) bytes of device memory, any content.
 * Enqueues on `stream`; n < 2^30. */
GSR_API size_t gsr_radix_scratch_bytes(uint32_t n, int bits);
GSR_API int gsr_radix_sort_pairs(uint32_t n, int bits, uint32_t* keys, uint32_t* keys_alt, uint32_t* vals,
                                 uint32_t* vals_alt, int iota_payload, void* scratch, size_t scratch_bytes,
                                 int* sorted_in_alt, void* stream);

/* The elementwise work of the reference's per-frame render() around its two rasterizer passes
 * (sugar/gaussian_splatting/gaussian_renderer/__init__.py:118-146,169-208), as two kernels instead of ~40 framework
 * launches.  Used by autovfx_amd/renderer.py when autograd is off; same formulas and operation order as the Python.
 *   gsr_view_normals: colors[P,3] = unit(flip_towards_camera(axis[P,3], means3D - cam_pos)) * 0.5 + 0.5
 *                     (axis = rotation column of the smallest scale, utils/general_utils.py:135-157)
 *   gsr_normal_maps : normal[H,W,3] = unit((normal_rgb[3,H,W] - 0.5) * 2);  pseudo_normal[H,W,3] = unit normal of the
 *                     un-projected depth map by central differences (:22-38,:41-80), zero on the border.
 *                     c2w: 16 device floats, the row-major 4x4 matrix the Python names c2w (rows 0..2 are read). */
GSR_API int gsr_view_normals(int P, const float* means3D, const float* axis, const float* cam_pos, float* colors,
                             void* stream);
GSR_API int gsr_normal_maps(int width, int height, const float* normal_rgb, const float* depth, const float* c2w,
                            float fx, float fy, float cx, float cy, float* normal, float* pseudo_normal, void* stream);

/*
 * Dynamic scenes (BASELINE configs[4]; SURVEY.md: scene_representation.py:357-372): for every frame of an edited scene the
 * reference deep-copies the whole scene, re-loads each inserted object's PLY, applies the frame's rigid-body transform in
 * PyTorch (gaussians_utils.py:85-118 transform_gaussians) and concatenates everything (:71-82 merge_two_gaussians, ~0.7 GB
 * of copies at 3 M Gaussians) before it calls render().  Here the scene lives in ONE resident buffer set sized for the base
 * scene plus all objects; the base part is written once, and per frame only the objects are placed:
 *
 * gsr_place_object transforms n Gaussians of one object from their RAW parameters (as the PLY holds them) and writes them
 * ACTIVATED (what the rasterizer consumes) at the given output pointers, i.e. at the object's offset in the scene buffers:
 *   xyz        : ((x - c0) * s + c0 - c0) R^T + c0 + (c - c0), every operation a separate fp32 rounding in exactly that
 *                order (gaussians_utils.py:94-108; the 3x3 product as (x0 r0 + x1 r1) + x2 r2)
 *   rotation   : unit( standardize( q_R (x) q_raw ) ) with q_R = matrix_to_quaternion(R) supplied by the caller
 *                (rotation_utils.py:24-85,113-135; F.normalize as gaussian_model.py:100-101)
 *   scale      : exp(log_scale + log_s)          (gaussians_utils.py:97 then gaussian_model.py:96-97)
 *   opacity, SH: copied (already activated / concatenated by the caller once per object; nullable = leave as is)
 *   min_axis   : (nullable) the rotation-matrix column of the smallest activated scale, what render() turns into the
 *                per-Gaussian normal (general_utils.py:78-101 build_rotation, :135-141 get_minimum_axis), from the
 *                normalised quaternion, in that code's operation order; on a tie between scales the column torch.argsort
 *                leaves in front on this GPU (its unstable 32-wide bitonic network: see gsr_forward_raw)
 * placement: 21 host floats -- center c[3], rotation R[9] row-major, scale s, initial_center c0[3], q_R[4] (w,x,y,z),
 * log_s (= (float)log((double)s)).  One streaming kernel: 40 B in + 40 B out per Gaussian (+ 196 B each way with SH).
 */
GSR_API int gsr_place_object(int n, const float* xyz, const float* rotation_raw, const float* log_scale,
                             const float* opacity /*nullable*/, const float* shs /*nullable*/, int M,
                             const float* placement /*host, 21 floats*/, float* out_means3D, float* out_scales,
                             float* out_rotations, float* out_opacities /*nullable*/, float* out_shs /*nullable*/,
                             float* out_min_axis /*[n,3], nullable*/, void* stream);

/* gsr_place_object for a SUBSET of an object's Gaussians: output j is built from input subset[j] (m ascending indices into the
 * object's arrays, device memory) -- the melting branch of the reference's frame loop (scene_representation.py:373-421), which
 * merges `orig_gaussians._xyz[mask]`, `._rotation[mask]`, ... of every melting mesh into the scene for one frame.  That branch
 * applies NO rigid transform: with placement == NULL positions are copied bit for bit, the raw quaternion is only normalised,
 * scale = exp(log_scale) (passing an identity placement instead would apply the identity's roundings, which the reference
 * does not).  With a placement the subset is transformed like gsr_place_object transforms the whole object. */
GSR_API int gsr_place_object_subset(int m, const uint32_t* subset, const float* xyz, const float* rotation_raw,
                                    const float* log_scale, const float* opacity /*nullable*/, const float* shs /*nullable*/, int M,
                                    const float* placement /*host, 21 floats; NULL = untransformed*/, float* out_means3D,
                                    float* out_scales, float* out_rotations, float* out_opacities /*nullable*/,
                                    float* out_shs /*nullable*/, float* out_min_axis /*[m,3], nullable*/, void* stream);

/* Self-test of the blend kernel's exp(): adds to *device_mismatches (a zeroed device u64) the number of floats
 * with bit patterns first_bits .. first_bits + count - 1 whose exp differs from the device library's expf.
 * The blend evaluates exp only for arguments <= 0; tests sweep every float of [-103, 0]. */
GSR_API int gsr_selftest_exp(uint32_t first_bits, uint32_t count, unsigned long long* device_mismatches, void* stream);
/* Self-test of the hardware behaviour the radix sort's ranking relies on: when several lanes of one wave64 instruction
 * add to the same LDS counter with a returning atomic, the lanes are served in ascending lane order (and instructions of
 * a wave in program order).  `workgroups` x 4 waves x `rounds` instructions with lane -> counter maps of every density;
 * *device_mismatches (caller-zeroed) counts the lanes whose returned value is not (the counter before the instruction) +
 * (lower lanes on the same counter).  tests/test_radix_gpu.py requires 0. */
GSR_API int gsr_selftest_lds_atomic_order(uint32_t workgroups, uint32_t rounds, uint32_t seed, unsigned long long* device_mismatches,
                                          void* stream);

/*
 * Backward rasterization: gradients of a gsr_forward call.  Mirrors Rasterizer::backward
 * (rasterizer.h:57-90, rasterizer_impl.cu:343-446) as called by RasterizeGaussiansBackwardCUDA
 * (DGR/rasterize_points.cu:121-209).
 *   R                 the num_rendered that gsr_forward returned for this call
 *   geom/binning/image_buffer   the three scratch arenas of that call, untouched since (the pointers the
 *                     scratch callbacks returned); they are self-describing and are validated
 *   accum_alphas      the forward's out_alpha;  dL_dpix[3,H,W], dL_dpix_depth[H,W], dL_dpix_alpha[H,W]
 *                     (the last two may both be NULL = all zeros: a loss that only looks at the colour image -- train.py:84-134,
 *                     scene_representation.py:495-520 -- has no gradient for them, and the per-pixel pass then skips their terms;
 *                     same gradients as with zero-filled arrays, ~10 % fewer instructions)
 *   outputs (every element is written: unlike the reference, whose binding zero-fills them first, :158-168,
 *   they may arrive with any content; dL_dconic, dL_dcolor, dL_ddepth and dL_dcov3D may be NULL when the caller has no use
 *   for them -- the first and third are intermediates the reference's binding never returns): dL_dmean2D[P,3],
 *   dL_dconic[P,4] (xx, xy, unused, yy), dL_dopacity[P], dL_dcolor[P,3], dL_ddepth[P], dL_dmean3D[P,3],
 *   dL_dcov3D[P,6], dL_dsh[P,M,3] (NULL allowed without shs), dL_dscale[P,3], dL_drot[P,4]
 *   accum_scratch     16*P floats of device memory, any content: the per-pixel pass sums each Gaussian's ten
 *                     partial gradients into one 64-byte line (one atomic instruction per list entry and
 *                     8x8 pixel block instead of ten); the per-Gaussian pass spreads them into the arrays above
 * Per-Gaussian sums are formed with one wave-level reduction and one atomic per 8x8 pixel block instead
 * of one atomic per pixel, so their last bits are order-dependent like the reference's.
 * Returns GSR_OK or a negative gsr_status.  Blocks the host once (reads the three arena headers).
 */
GSR_API int gsr_backward(int P, int D, int M, int R, const float* background, int width, int height,
                         const float* means3D, const float* shs /*nullable*/,
                         const float* colors_precomp /*nullable*/, const float* scales /*nullable*/,
                         float scale_modifier, const float* rotations /*nullable*/,
                         const float* cov3D_precomp /*nullable*/, const float* viewmatrix,
                         const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                         const int* radii /*nullable*/, const char* geom_buffer, const char* binning_buffer,
                         const char* image_buffer, const float* accum_alphas, const float* dL_dpix,
                         const float* dL_dpix_depth /*nullable*/, const float* dL_dpix_alpha /*nullable*/, float* dL_dmean2D,
                         float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepth,
                         float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh /*nullable*/, float* dL_dscale,
                         float* dL_drot, float* accum_scratch /* 16*P floats, any content */, int debug, void* stream);

/*
 * Gradients of a gsr_forward_raw call (a FULL call: flags without GSR_FORWARD_INFERENCE) with respect to the model's RAW
 * parameter tensors: gsr_backward with the activations' chain rule applied inside the per-Gaussian pass, i.e. what autograd
 * produces for the reference's render() (gaussian_renderer/__init__.py:83-218 over scene/gaussian_model.py:95-128) --
 *   dL_dlog_scales      = dL/d(scale) * scale                       (through exp)
 *   dL_drotations       = (g - q (q . g)) / ||raw||                 (through F.normalize; g / 1e-12 below its clamp)
 *   dL_dopacity_logits  = dL/d(opacity) * o (1 - o)                 (through sigmoid)
 *   dL_dfeatures_dc [P,1,3], dL_dfeatures_rest [P,M-1,3]            (the two halves of dL_dsh: torch.cat's backward)
 *   dL_dxyz [P,3], dL_dmean2D [P,3]                                 (as gsr_backward's dL_dmean3D / dL_dmean2D)
 * -- without the ~1 GB of activated tensors and their gradients ever existing.  dL_dpix_normal (nullable) [3,H,W] is the
 * gradient of the normal image the forward composited (out_normal): one more per-pixel pass over the same lists, its
 * per-Gaussian colour gradient chained through get_normal / build_rotation to the quaternion (positions and scales enter the
 * normal only through a sign and an argmin: no gradient, as in autograd).  dL_dpix_depth / dL_dpix_alpha: nullable as in
 * gsr_backward.  radii / scratch buffers / accum_scratch / return value: as gsr_backward.  Every output element is written.
 */
GSR_API int gsr_backward_raw(int P, int D, int M, int R, const float* background, int width, int height,
                             const gsr_raw_params* raw, float scale_modifier, const float* viewmatrix, const float* projmatrix,
                             const float* cam_pos, float tan_fovx, float tan_fovy, const int* radii /*nullable*/,
                             const char* geom_buffer, const char* binning_buffer, const char* image_buffer,
                             const float* accum_alphas, const float* dL_dpix, const float* dL_dpix_depth /*nullable*/,
                             const float* dL_dpix_alpha /*nullable*/, const float* dL_dpix_normal /*nullable*/,
                             float* dL_dmean2D, float* dL_dxyz, float* dL_dlog_scales, float* dL_drotations,
                             float* dL_dopacity_logits, float* dL_dfeatures_dc, float* dL_dfeatures_rest /*NULL iff M == 1*/,
                             float* accum_scratch /* 16*P floats, any content */, int debug, void* stream);

/* ---- introspection (used by the parity tests and bench.py; not part of the reference API) ---- */

/* Sub-arrays of the geometry / binning / image scratch, as byte offsets from the pointer the
 * callback returned.  Offsets depend only on the sizes given. */
typedef enum gsr_geom_slot {
    GSR_GEOM_RASTER = 0,        /* f32[8P]  per splat: pixel x, y, conic xx, xy, yy (inverse 2D covariance),
                                   opacity, view-space z, -ln(255 opacity) - 1e-4 -- valid where radii > 0             */
    GSR_GEOM_RGB,               /* f32[3P]  SH-evaluated colour (unused with colors_precomp)       */
    GSR_GEOM_SPLAT_BINS,        /* u32[4P]  per splat: first tile x | y << 16 and size w | h << 16 of its tight rectangle
                                   (0 = emits nothing), then the 64-bit mask of its live tiles when w * h <= 64 (for a
                                   larger splat, whose live tiles are found later: the tile rows of the larger splats before
                                   it among the 256 Gaussians it shares id >> 8 with, then all ones); GSR_OPT_TILE_CULL */
    GSR_GEOM_INTERNAL_RADII,    /* i32[P]   used when the caller passes radii == NULL              */
    GSR_GEOM_DEPTH_ORDER,       /* u32[P]   Gaussian ids, ascending (depth bits, id): the visible ones; GSR_OPT_DEPTH_DROP */
    GSR_GEOM_POINT_OFFSETS,     /* u32[P]   inclusive scan of the pair counts in DEPTH_ORDER order     */
    GSR_GEOM_LISTED,            /* u8[P]    inference calls with deferred colours that were cut into depth slabs: s + 1 when
                                   slab s (the last one that did) put the Gaussian into a list, 0 when none did; such a call
                                   evaluates SH colours for exactly these.  Offset 0 = the call has no such array. */
    GSR_GEOM_VIEW_NORMALS,      /* f32[3P]  gsr_forward_raw with out_normal: get_normal(dir) * 0.5 + 0.5 of every splat with a
                                   rectangle (radii > 0 and a tile).  Offset 0 = the call has no such array. */
    GSR_GEOM_NUM_SLOTS
} gsr_geom_slot;

typedef enum gsr_binning_slot {
    GSR_BIN_POINT_LIST = 0,     /* u32[live pairs] Gaussian ids sorted by (tile, depth bits, id); see GSR_OPT_TILE_CULL.
                                   An inference call has one list per depth slab: this is the first slab's. */
    GSR_BIN_TILE_KEYS,          /* u32[live pairs] tile id of each entry of the LAST slab's list (full calls: of POINT_LIST) */
    GSR_BIN_NUM_SLOTS
} gsr_binning_slot;

typedef enum gsr_image_slot {
    GSR_IMG_RANGES = 0,         /* u32[2T] [begin,end) into POINT_LIST per 16x16 tile; (0,0) if empty (inference calls:
                                   one such table per depth slab, back to back) */
    GSR_IMG_N_CONTRIB,          /* u32[W*H] 1-based list position of the last blended entry          */
    GSR_IMG_NUM_SLOTS
} gsr_image_slot;

/* Valid after a gsr_forward call on this thread: where that call put each slot. */
GSR_API int gsr_last_geom_offsets(size_t offsets[GSR_GEOM_NUM_SLOTS]);
GSR_API int gsr_last_binning_offsets(size_t offsets[GSR_BIN_NUM_SLOTS]);
GSR_API int gsr_last_image_offsets(size_t offsets[GSR_IMG_NUM_SLOTS]);
/* counts[0] = num_rendered as the reference defines it (sum of rectangle areas, the return value of
 * gsr_forward); counts[1] = live pairs = pairs that were expanded and sorted (the length of POINT_LIST for a full
 * call).  Past the sizing read-back a call keeps its pair counts on the device; this accessor waits for the call's
 * stream to reach the copy it queued at its end. */
GSR_API int gsr_last_pair_counts(uint32_t counts[2]);
/* Pairs in the list of each depth slab of that call; returns the number of slabs (1 for a full call). */
GSR_API int gsr_last_slab_pairs(uint32_t pairs[GSR_MAX_SLABS]);

/* ---- options (process-wide; defaults in brackets) ---- */
typedef enum gsr_option {
    /* [1] Exact-image tile culling: a (tile, Gaussian) pair of the reference's rectangle for which
     * no pixel of the tile can reach alpha >= 1/255 is never expanded, sorted or blended.
     * color / depth / alpha / radii and the returned num_rendered are bit-identical with the option
     * on or off; POINT_LIST then holds only the live pairs (a subsequence of the reference's list,
     * same order), and SPLAT_BINS / POINT_OFFSETS / RANGES / N_CONTRIB count live pairs.  Splats of any size are
     * culled: a bit mask over the tight rectangle up to 64 tiles, one run of live columns per tile row above.
     * 0 reproduces the reference's lists exactly. */
    GSR_OPT_TILE_CULL = 0,
    /* [2] Most depth slabs an inference call (GSR_FORWARD_INFERENCE) may use (0 = as many as the slab sizes call for, up
     * to GSR_MAX_SLABS; 1 switches the occlusion culling between slabs off).  Same images.  Two is the default because
     * behind a first slab that finished most tiles almost nothing is left (measured: 95 - 100 % of the remaining pairs
     * are dropped), so a third slab only adds its dozen of launches. */
    GSR_OPT_SLABS = 1,
    /* [400] Pairs per tile, on average, that the first depth slab holds; every further slab is three times as large
     * as the one before.  A tuning knob: the first slab should finish most tiles of a scene with an opaque front. */
    GSR_OPT_SLAB_FIRST = 2,
    /* [1] Inference calls evaluate SH colours only for the splats that reach a list (0: for every visible one, in the
     * projection kernel, as full calls do).  Same bits wherever a colour is used. */
    GSR_OPT_DEFER_COLOUR = 3,
    /* [3000000] An inference call is only cut into depth slabs when more pairs than this lie behind the first slab: a
     * further slab costs a fixed dozen of small launches, what it saves grows with the pairs it can drop.  (At
     * 960x540 with 1 M Gaussians and ~2 M live pairs slabs lose 13 %; with 6 - 18 M live pairs they gain 5 - 60 %.) */
    GSR_OPT_SLAB_MIN_REST = 4,
    /* [2] How the radix sort ranks the keys of a wave (gsr_radix.hip).  All forms produce the same stable sort, and none of
     * them relies on anything the ISA does not promise.  0: ballots.  1: one returning LDS add per key, VERIFIED -- the values
     * returned are the stable rank if lanes of one instruction that hit the same LDS counter are served in ascending lane
     * order, which gfx950 is observed to do and no manual promises; so every 4096-key tile checks the order its ranks produce
     * (one compare per key) before it writes anything, and a tile that fails is ranked again with ballots in place and counted
     * in GSR_OPT_RADIX_RANK_FALLBACKS.  2 (default): as 1 on a device that passed gsr_selftest_lds_atomic_order's kernel
     * (~0.4 M instructions of every conflict density, run by the first sort on each device after the request), plain ballots
     * on a device that failed it (there every tile would pay for both); the test runs on a stream of its own (the caller's is
     * neither drained nor used; nothing is allocated or freed) and the host waits a quarter of a millisecond for it, once per
     * device, under a process-wide lock; a test that cannot run at all is given up after four sorts (ballots from then on); while the caller's stream is capturing a graph it does not run and that sort uses ballots (a test that could not
     * run is retried by the next sort; gsr_get_option(GSR_OPT_RADIX_RANK_ACTIVE) at start-up gets it out of the way).  3: as 1 with an inversion injected into every wave -- a test hook for the check and the repair.
     * Verified LDS adds are ~20 % faster per pass than ballots: ~3 % of a single-stream 3 M-Gaussian frame. */
    GSR_OPT_RADIX_RANK = 5,
    /* read-only: what sorts queued on the CURRENT device use -- 1 verified LDS adds (2: with the injected inversion), 0 ballots
     * (runs the self-test if this device has not been tested yet and the request is 2).  gsr_set_option rejects it. */
    GSR_OPT_RADIX_RANK_ACTIVE = 6,
    /* [1] The depth sort drops the Gaussians that produce no pair (culled, or every tile dead) in its first pass instead of
     * carrying them to the end of the order through all four: the later passes sort the visible ones only.  The first
     * `visible` entries of GSR_GEOM_DEPTH_ORDER are the same either way; with 1 the entries behind them are undefined
     * (0: the culled Gaussians, in index order, as rounds 1 - 2 left them). */
    GSR_OPT_DEPTH_DROP = 7,
    /* [1] Images of more than 256 tiles: every XCD blends four strips of tiles spread over the image (instead of one contiguous
     * eighth: the picture's middle is busier than its edges) and walks them longest tile list first (7 classes of list
     * length, then the tiles with nothing to blend), so that the waves still running when a launch with more workgroups than
     * wave slots runs dry are short ones.  Placement only: same results. */
    GSR_OPT_BLEND_ORDER = 8,
    /* read-only: 4096-key tiles, on the CURRENT device since the library was loaded, whose LDS-add ranks failed the order check
     * and were ranked again with ballots (saturates at INT_MAX; -1 if the device word could not be read).  0 on every MI355X
     * this has run on; non-zero only costs time.  Reads a device word through the null stream: synchronise the streams that
     * sorted first.  gsr_set_option rejects it. */
    GSR_OPT_RADIX_RANK_FALLBACKS = 9,
    /* [0] gsr_backward / gsr_backward_raw form every per-Gaussian sum in a FIXED order: same inputs, same bits, on every run and
     * every box.  By default the sums of a Gaussian's (tile quadrant, list entry) contributions meet in HBM through float atomics,
     * in the order the waves arrive (the reference's own backward does the same per pixel, backward.cu:553-596), so the last
     * bits of a gradient -- and, for an ill-conditioned Gaussian, much more than the last bits -- differ between two runs.
     * With 1 the per-pixel passes store one 40-byte record per contributing (sorted list position, quadrant), the point list
     * is sorted by Gaussian id with the library's own (stable) radix sort, and one lane per Gaussian adds its records in
     * ascending (tile, quadrant) order.  Costs 160 bytes of pool memory per live pair for the duration of the call (taken
     * with hipMallocAsync on the call's stream: not capturable into a graph) and ~25 % of a training iteration at C3. */
    GSR_OPT_BACKWARD_DETERMINISTIC = 10,
    /* [1] A hint for BINDINGS (the library itself only stores it): a forward call whose result WILL be differentiated may still be
     * made as an inference call (GSR_FORWARD_INFERENCE: depth slabs with occlusion culling between them, SH colours only for
     * listed splats) -- gsr_backward / gsr_backward_raw accept such a call's scratch and walk the slabs' list segments back to
     * front (a pair a later slab dropped belongs to a tile whose pixels had all stopped: no gradient flows through it).  Same
     * images, same gradients up to the order of the atomic sums; the training forward pays for the pairs and colours it uses
     * instead of all of them.  This repository's Python binding follows the hint unless GSR_OPT_BACKWARD_DETERMINISTIC is set
     * (that mode sorts ONE list per tile).  0: grad-mode forward calls are full calls, as before round 5. */
    GSR_OPT_GRAD_SLABS = 11,
    GSR_OPT_NUM
} gsr_option;
GSR_API int gsr_set_option(int option, int value);
/* How an inference call with `live_pairs` pairs (after tile culling) on a `width` x `height` image would be cut into
 * depth slabs under the current options: cuts[i] = inclusive pair offset at which slab i ends (slabs - 1 entries are
 * written); returns the number of slabs (1 = not cut).  Host logic only: no device is touched. */
GSR_API int gsr_plan_slabs(uint32_t live_pairs, int width, int height, uint32_t cuts[GSR_MAX_SLABS]);
GSR_API int gsr_get_option(int option);

/* Per-stage device timing of gsr_forward via hipEvents on `stream` (off by default). */
typedef enum gsr_stage {
    GSR_STAGE_PREPROCESS = 0,   /* per-Gaussian projection, covariance, SH                         */
    GSR_STAGE_DEPTH_SORT,       /* radix sort of P depth keys                                       */
    GSR_STAGE_SCAN,             /* the host's wait for the pair count (the GPU is idle here only if the depth sort is done) */
    GSR_STAGE_DUPLICATE,        /* gather + scan of the pair counts, (tile, id) pair expansion (all depth slabs)  */
    GSR_STAGE_TILE_SORT,        /* stable radix sort of pairs by tile id                            */
    GSR_STAGE_RANGES,           /* per-tile [begin,end)                                             */
    GSR_STAGE_BLEND,            /* per-tile front-to-back compositing                               */
    GSR_STAGE_COLOUR,           /* SH colours of the listed splats (inference calls; else part of PREPROCESS)      */
    GSR_STAGE_NUM
} gsr_stage;
/* Enabling (or re-enabling) resets the per-thread record of timed calls. */
GSR_API void gsr_set_stage_timing(int enable);
/* Mean milliseconds per stage over the gsr_forward calls made on this thread since timing was
 * enabled (at most the last 256; synchronises on their events).  Returns the number of calls
 * averaged, or a negative gsr_status.  Early exits (P == 0, errors) are not recorded. */
GSR_API int gsr_get_stage_times(float ms[GSR_STAGE_NUM]);
/* First-kernel-to-last-kernel device time of each of the most recent timed gsr_forward calls of this thread, newest
 * first; returns how many were written (at most `capacity` and at most the ring of 256). */
GSR_API int gsr_get_call_times(float* ms, int capacity);
/* While stage timing is enabled gsr_backward records HIP events on its stream as well: ms[0] = mean milliseconds of the
 * per-pixel pass (render_backward_kernel, with the clearing of its accumulation scratch), ms[1] = of the per-Gaussian pass
 * (preprocess_backward_kernel), over the gsr_backward calls of the whole process since timing was enabled (at most the
 * last 64; autograd runs backward on its own thread, so this record is process-wide).  Returns the number of calls averaged. */
GSR_API int gsr_get_backward_times(float ms[2]);

GSR_API const char* gsr_last_error(void);
GSR_API int gsr_abi_version(void);
/* Name of the compiled code-object target ("gfx950"). */
GSR_API const char* gsr_target_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_INCLUDED */
