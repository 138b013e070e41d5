/*
 * gsr_oracle.c -- CPU restatement of the reference 3D-Gaussian-splatting forward rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under autovfx_amd/ or diff_gaussian_rasterization/ may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py use it, and only as the checker / the reported CPU baseline.
 *
 * What it restates (paths relative to /root/reference, DGR = sugar/gaussian_splatting/
 * submodules/diff-gaussian-rasterization):
 *   DGR/cuda_rasterizer/auxiliary.h:22-39   SH basis constants
 *   DGR/cuda_rasterizer/auxiliary.h:41-44   ndc -> pixel (evaluated in double)
 *   DGR/cuda_rasterizer/auxiliary.h:46-56   tile rectangle of a splat
 *   DGR/cuda_rasterizer/auxiliary.h:58-77   4x3 / 4x4 point transforms
 *   DGR/cuda_rasterizer/auxiliary.h:139-164 near-plane cull
 *   DGR/cuda_rasterizer/forward.cu:20-71    SH -> RGB
 *   DGR/cuda_rasterizer/forward.cu:74-113   EWA 2D covariance
 *   DGR/cuda_rasterizer/forward.cu:118-152  3D covariance from scale + quaternion
 *   DGR/cuda_rasterizer/forward.cu:155-256  per-Gaussian preprocess
 *   DGR/cuda_rasterizer/forward.cu:261-378  per-tile front-to-back blend
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:35-50    key width
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:70-111   (tile,depth) key duplication
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:116-138  per-tile ranges
 *   DGR/cuda_rasterizer/rasterizer_impl.cu:197-339  orchestration
 *   DGR/rasterize_points.cu:36-119          zero-filled outputs, P==0 short-circuit
 * CUB's InclusiveSum / stable SortPairs are not vendored in the reference; their published
 * semantics (inclusive prefix sum; stable ascending sort on a bit range) are restated here.
 *
 * Arithmetic is fp32 in the reference's operation order; build with -ffp-contract=off so the
 * compiler does not fuse a*b+c (see oracle/Makefile).  GLM is column-major; every matrix below
 * is written in ordinary (row, col) math notation, and m3_mul() sums k = 0,1,2 left to right,
 * which is what glm's mat3 * mat3 does (third_party/glm/glm/detail/type_mat3x3.inl:486-518).
 *
 * Parity pin: the reference ships no golden vectors for this path.  This
 * file is pinned instead against the reference's own sources compiled for the host
 * (oracle/build_ref.py -> oracle/_ref/) and against analytic known-answer cases in tests/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16 /* DGR/cuda_rasterizer/config.h:16-17 */

static const float kC0 = 0.28209479177387814f;
static const float kC1 = 0.4886025119029199f;
static const float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                             -1.0925484305920792f, 0.5462742152960396f};
static const float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                             0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                             -0.5900435899266435f};

/* float -> int the way the GPU converts (round toward zero, saturating, NaN -> 0). */
static int f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

typedef struct { float m[3][3]; } m3; /* m[row][col] */

static m3 m3_mul(const m3* a, const m3* b) {
    m3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = a->m[i][0] * b->m[0][j] + a->m[i][1] * b->m[1][j] + a->m[i][2] * b->m[2][j];
    return r;
}
static m3 m3_t(const m3* a) {
    m3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a->m[j][i];
    return r;
}

/* auxiliary.h:58-77 -- m is the transposed matrix stored row-major (m[4*col+row]). */
static void xform43(const float* m, const float p[3], float o[3]) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform44(const float* m, const float p[3], float o[4]) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* auxiliary.h:41-44: literals are double, so the whole expression is double, then narrowed. */
static float ndc_to_pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:46-56.  Returns the half-open tile rectangle [x0,x1) x [y0,y1). */
static void tile_rect(float px, float py, int radius, int gx, int gy, int* x0, int* y0, int* x1, int* y1) {
    *x0 = imin(gx, imax(0, f2i((px - radius) / TILE)));
    *y0 = imin(gy, imax(0, f2i((py - radius) / TILE)));
    *x1 = imin(gx, imax(0, f2i((px + radius + TILE - 1) / TILE)));
    *y1 = imin(gy, imax(0, f2i((py + radius + TILE - 1) / TILE)));
}

/* forward.cu:118-152 */
static void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float out[6]) {
    m3 S = {{{0}}}, R, M, Mt, Sg;
    S.m[0][0] = mod * s[0];
    S.m[1][1] = mod * s[1];
    S.m[2][2] = mod * s[2];
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    /* glm::mat3(a,b,c, d,e,f, g,h,i) fills COLUMNS: (a,b,c) is column 0. */
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[1][0] = 2.f * (x * y - r * z);       R.m[2][0] = 2.f * (x * z + r * y);
    R.m[0][1] = 2.f * (x * y + r * z);       R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[2][1] = 2.f * (y * z - r * x);
    R.m[0][2] = 2.f * (x * z - r * y);       R.m[1][2] = 2.f * (y * z + r * x);       R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    M = m3_mul(&S, &R);
    Mt = m3_t(&M);
    Sg = m3_mul(&Mt, &M);
    /* Sigma[c][r] in glm = Sg.m[r][c] here. */
    out[0] = Sg.m[0][0]; out[1] = Sg.m[1][0]; out[2] = Sg.m[2][0];
    out[3] = Sg.m[1][1]; out[4] = Sg.m[2][1]; out[5] = Sg.m[2][2];
}

/* forward.cu:74-113 */
static void cov2d(const float p[3], float fx, float fy, float tanx, float tany, const float c3[6],
                  const float* view, float out[3]) {
    float t[3];
    xform43(view, p, t);
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    m3 J = {{{0}}}, W, T, V, Tt, Vt, A, C;
    J.m[0][0] = fx / t[2]; J.m[2][0] = -(fx * t[0]) / (t[2] * t[2]);
    J.m[1][1] = fy / t[2]; J.m[2][1] = -(fy * t[1]) / (t[2] * t[2]);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) W.m[r][c] = view[4 * r + c];
    T = m3_mul(&W, &J);
    V.m[0][0] = c3[0]; V.m[1][0] = c3[1]; V.m[2][0] = c3[2];
    V.m[0][1] = c3[1]; V.m[1][1] = c3[3]; V.m[2][1] = c3[4];
    V.m[0][2] = c3[2]; V.m[1][2] = c3[4]; V.m[2][2] = c3[5];
    Tt = m3_t(&T);
    Vt = m3_t(&V);
    A = m3_mul(&Tt, &Vt);
    C = m3_mul(&A, &T);
    out[0] = C.m[0][0] + 0.3f; /* cov[0][0] */
    out[1] = C.m[1][0];        /* cov[0][1] in glm = (row 1, col 0) */
    out[2] = C.m[1][1] + 0.3f; /* cov[1][1] */
}

/* forward.cu:20-71.  sh points at this Gaussian's M x 3 block. */
static void sh_to_rgb(int deg, const float p[3], const float cam[3], const float* sh, float rgb[3],
                      uint8_t clamped[3]) {
    float d[3] = {p[0] - cam[0], p[1] - cam[1], p[2] - cam[2]};
    const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] = d[0] / len; d[1] = d[1] / len; d[2] = d[2] / len;
    const float x = d[0], y = d[1], z = d[2];
    for (int c = 0; c < 3; ++c) {
#define SH(k) sh[3 * (k) + c]
        float v = kC0 * SH(0);
        if (deg > 0) {
            v = v - kC1 * y * SH(1) + kC1 * z * SH(2) - kC1 * x * SH(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                v = v + kC2[0] * xy * SH(4) + kC2[1] * yz * SH(5) + kC2[2] * (2.0f * zz - xx - yy) * SH(6) +
                    kC2[3] * xz * SH(7) + kC2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    v = v + kC3[0] * y * (3.0f * xx - yy) * SH(9) + kC3[1] * xy * z * SH(10) +
                        kC3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                        kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        kC3[4] * x * (4.0f * zz - xx - yy) * SH(13) + kC3[5] * z * (xx - yy) * SH(14) +
                        kC3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        v += 0.5f;
        clamped[c] = (uint8_t)(v < 0);
        rgb[c] = fmaxf(v, 0.0f);
    }
}

/* rasterizer_impl.cu:35-50: bits needed above the 32 depth bits. */
uint32_t gsro_key_bits(uint32_t n) {
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* rasterizer_impl.cu:54-66,141-153 */
void gsro_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present) {
    (void)proj;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        float pv[3];
        xform43(view, means3D + 3 * (size_t)i, pv);
        present[i] = (uint8_t)(pv[2] > 0.2f);
    }
}

/*
 * Per-Gaussian preprocess (forward.cu:155-256).  All outputs have P rows; rows of culled
 * Gaussians are left as they were passed in, except radii/tiles_touched which are zeroed, as in
 * the reference.  cov3D_out (6P) and clamped (3P) may be NULL.
 */
void gsro_preprocess(int P, int deg, int M, const float* means3D, const float* scales, float mod,
                     const float* rots, const float* opac, const float* shs, const float* cov3D_pre,
                     const float* colors_pre, const float* view, const float* proj, const float* cam,
                     int W, int H, float tanx, float tany, int* radii, float* means2D, float* depths,
                     float* cov3D_out, float* rgb, float* conic_opacity, uint32_t* tiles_touched,
                     uint8_t* clamped) {
    const float fy = H / (2.0f * tany), fx = W / (2.0f * tanx); /* rasterizer_impl.cu:223-224 */
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        const float* p = means3D + 3 * (size_t)i;
        float pv[3], ph[4];
        xform44(proj, p, ph);
        const float pw = 1.0f / (ph[3] + 0.0000001f);
        const float ndc[2] = {ph[0] * pw, ph[1] * pw};
        xform43(view, p, pv);
        if (pv[2] <= 0.2f) continue; /* auxiliary.h:154 (prefiltered trap not restated) */

        float c3tmp[6];
        const float* c3;
        if (cov3D_pre) {
            c3 = cov3D_pre + 6 * (size_t)i;
        } else {
            cov3d_from_scale_rot(scales + 3 * (size_t)i, mod, rots + 4 * (size_t)i, c3tmp);
            if (cov3D_out) memcpy(cov3D_out + 6 * (size_t)i, c3tmp, sizeof c3tmp);
            c3 = c3tmp;
        }
        float cv[3];
        cov2d(p, fx, fy, tanx, tany, c3, view, cv);
        const float det = cv[0] * cv[2] - cv[1] * cv[1];
        if (det == 0.0f) continue;
        const float det_inv = 1.f / det;
        const float conic[3] = {cv[2] * det_inv, -cv[1] * det_inv, cv[0] * det_inv};
        const float mid = 0.5f * (cv[0] + cv[2]);
        const float lam1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lam2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float rad = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
        const float px = ndc_to_pix(ndc[0], W), py = ndc_to_pix(ndc[1], H);
        int x0, y0, x1, y1;
        tile_rect(px, py, f2i(rad), gx, gy, &x0, &y0, &x1, &y1);
        const uint32_t area = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
        if (area == 0) continue;
        if (!colors_pre) {
            uint8_t cl[3];
            sh_to_rgb(deg, p, cam, shs + 3 * (size_t)M * i, rgb + 3 * (size_t)i, cl);
            if (clamped) memcpy(clamped + 3 * (size_t)i, cl, 3);
        }
        depths[i] = pv[2];
        radii[i] = f2i(rad);
        means2D[2 * (size_t)i] = px;
        means2D[2 * (size_t)i + 1] = py;
        conic_opacity[4 * (size_t)i + 0] = conic[0];
        conic_opacity[4 * (size_t)i + 1] = conic[1];
        conic_opacity[4 * (size_t)i + 2] = conic[2];
        conic_opacity[4 * (size_t)i + 3] = opac[i];
        tiles_touched[i] = area;
    }
}

/* cub::DeviceScan::InclusiveSum (rasterizer_impl.cu:278).  Returns the total. */
uint32_t gsro_inclusive_sum(int P, const uint32_t* in, uint32_t* out) {
    uint32_t acc = 0;
    for (int i = 0; i < P; ++i) { acc += in[i]; out[i] = acc; }
    return acc;
}

/* rasterizer_impl.cu:70-111: emit (tile << 32 | depth bits, gaussian id) in index order. */
void gsro_duplicate(int P, int W, int H, const float* means2D, const float* depths, const uint32_t* offsets,
                    const int* radii, uint64_t* keys, uint32_t* vals) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int i = 0; i < P; ++i) {
        if (radii[i] <= 0) continue;
        uint32_t off = i == 0 ? 0 : offsets[i - 1];
        int x0, y0, x1, y1;
        tile_rect(means2D[2 * (size_t)i], means2D[2 * (size_t)i + 1], radii[i], gx, gy, &x0, &y0, &x1, &y1);
        uint32_t dbits;
        memcpy(&dbits, depths + i, 4);
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                uint64_t k = (uint64_t)(uint32_t)(y * gx + x);
                k <<= 32;
                k |= dbits;
                keys[off] = k;
                vals[off] = (uint32_t)i;
                ++off;
            }
    }
}

/* cub::DeviceRadixSort::SortPairs on bits [0,end_bit): stable LSD radix, 11-bit digits. */
void gsro_sort_pairs(size_t n, uint64_t* keys, uint32_t* vals, unsigned end_bit) {
    if (n == 0) return;
    uint64_t* k2 = (uint64_t*)malloc(n * sizeof(uint64_t));
    uint32_t* v2 = (uint32_t*)malloc(n * sizeof(uint32_t));
    size_t* cnt = (size_t*)malloc(2048 * sizeof(size_t));
    uint64_t *ka = keys, *kb = k2;
    uint32_t *va = vals, *vb = v2;
    for (unsigned sh = 0; sh < end_bit; sh += 11) {
        unsigned bits = end_bit - sh < 11 ? end_bit - sh : 11;
        const uint64_t mask = (1ull << bits) - 1;
        memset(cnt, 0, 2048 * sizeof(size_t));
        for (size_t i = 0; i < n; ++i) cnt[(ka[i] >> sh) & mask]++;
        size_t run = 0;
        for (unsigned b = 0; b < 2048; ++b) { size_t c = cnt[b]; cnt[b] = run; run += c; }
        for (size_t i = 0; i < n; ++i) {
            size_t d = cnt[(ka[i] >> sh) & mask]++;
            kb[d] = ka[i];
            vb[d] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    if (ka != keys) { memcpy(keys, ka, n * sizeof(uint64_t)); memcpy(vals, va, n * sizeof(uint32_t)); }
    free(k2); free(v2); free(cnt);
}

/* rasterizer_impl.cu:116-138 (+ the memset at :311).  ranges is uint2[T] flattened. */
void gsro_tile_ranges(size_t n, const uint64_t* keys, int T, uint32_t* ranges) {
    memset(ranges, 0, (size_t)T * 2 * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) {
        const uint32_t cur = (uint32_t)(keys[i] >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == n - 1) ranges[2 * cur + 1] = (uint32_t)n;
    }
}

/*
 * forward.cu:261-378: every pixel walks its tile's list front to back.
 * feat is [P,3]; out_color is planar [3,H,W].
 */
void gsro_blend(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                const float* feat, const float* depths, const float* conic_opacity, const float* bg,
                float* out_color, float* out_depth, float* out_alpha, uint32_t* n_contrib) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ++ty)
        for (int tx = 0; tx < gx; ++tx) {
            const uint32_t lo = ranges[2 * (ty * gx + tx)], hi = ranges[2 * (ty * gx + tx) + 1];
            for (int ly = 0; ly < TILE; ++ly)
                for (int lx = 0; lx < TILE; ++lx) {
                    const int px = tx * TILE + lx, py = ty * TILE + ly;
                    if (px >= W || py >= H) continue;
                    const float pxf = (float)px, pyf = (float)py;
                    float T = 1.0f, C[3] = {0, 0, 0}, Dacc = 0;
                    uint32_t visited = 0, last = 0;
                    for (uint32_t e = lo; e < hi; ++e) {
                        const uint32_t g = point_list[e];
                        ++visited;
                        const float dx = means2D[2 * (size_t)g] - pxf, dy = means2D[2 * (size_t)g + 1] - pyf;
                        const float* co = conic_opacity + 4 * (size_t)g;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float alpha = fminf(0.99f, co[3] * expf(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        const float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break; /* pixel saturated; this entry is not blended */
                        for (int ch = 0; ch < 3; ++ch) C[ch] += feat[3 * (size_t)g + ch] * alpha * T;
                        Dacc += depths[g] * alpha * T;
                        T = test_T;
                        last = visited;
                    }
                    const size_t pid = (size_t)W * py + px;
                    out_alpha[pid] = 1 - T;
                    if (n_contrib) n_contrib[pid] = last;
                    for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
                    out_depth[pid] = Dacc;
                }
        }
}

/*
 * Whole forward pass (rasterizer_impl.cu:197-339 + rasterize_points.cu:36-119).
 * Outputs must be zero-filled by the caller (the binding does torch::full(0)); with P == 0 nothing
 * is touched.  Optional intermediates (may be NULL): means2D[2P], depths[P], conic_opacity[4P],
 * rgb[3P], tiles_touched[P], offsets[P].  If keys_out/list_out/ranges_out are non-NULL they must
 * hold cap pairs / cap ids / 2*T words; when num_rendered > cap they are not written.
 * Returns num_rendered.
 */
int64_t gsro_forward(int P, int deg, int M, const float* bg, int W, int H, const float* means3D,
                     const float* shs, const float* colors_pre, const float* opac, const float* scales,
                     float mod, const float* rots, const float* cov3D_pre, const float* view,
                     const float* proj, const float* cam, float tanx, float tany, float* out_color,
                     float* out_depth, float* out_alpha, int* radii, float* o_means2D, float* o_depths,
                     float* o_conic_opacity, float* o_rgb, uint32_t* o_tiles, uint32_t* o_offsets,
                     uint32_t* o_ncontrib, size_t cap, uint64_t* keys_out, uint32_t* list_out,
                     uint32_t* ranges_out) {
    if (P == 0) return 0;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
    const size_t n = (size_t)P;
    float* means2D = (float*)calloc(2 * n, 4);
    float* depths = (float*)calloc(n, 4);
    float* conop = (float*)calloc(4 * n, 4);
    float* rgb = (float*)calloc(3 * n, 4);
    uint32_t* tiles = (uint32_t*)calloc(n, 4);
    uint32_t* offs = (uint32_t*)calloc(n, 4);
    int* rad_local = radii ? NULL : (int*)calloc(n, 4);
    int* rad = radii ? radii : rad_local;

    gsro_preprocess(P, deg, M, means3D, scales, mod, rots, opac, shs, cov3D_pre, colors_pre, view, proj, cam,
                    W, H, tanx, tany, rad, means2D, depths, NULL, rgb, conop, tiles, NULL);
    const uint32_t D = gsro_inclusive_sum(P, tiles, offs);
    uint64_t* keys = (uint64_t*)malloc((D ? D : 1) * sizeof(uint64_t));
    uint32_t* vals = (uint32_t*)malloc((D ? D : 1) * sizeof(uint32_t));
    uint32_t* ranges = (uint32_t*)malloc((size_t)T * 2 * sizeof(uint32_t));
    gsro_duplicate(P, W, H, means2D, depths, offs, rad, keys, vals);
    gsro_sort_pairs(D, keys, vals, 32 + gsro_key_bits((uint32_t)T));
    gsro_tile_ranges(D, keys, T, ranges);
    gsro_blend(W, H, ranges, vals, means2D, colors_pre ? colors_pre : rgb, depths, conop, bg, out_color,
               out_depth, out_alpha, o_ncontrib);

    if (o_means2D) memcpy(o_means2D, means2D, 2 * n * 4);
    if (o_depths) memcpy(o_depths, depths, n * 4);
    if (o_conic_opacity) memcpy(o_conic_opacity, conop, 4 * n * 4);
    if (o_rgb) memcpy(o_rgb, rgb, 3 * n * 4);
    if (o_tiles) memcpy(o_tiles, tiles, n * 4);
    if (o_offsets) memcpy(o_offsets, offs, n * 4);
    if (D <= cap) {
        if (keys_out) memcpy(keys_out, keys, (size_t)D * 8);
        if (list_out) memcpy(list_out, vals, (size_t)D * 4);
    }
    if (ranges_out) memcpy(ranges_out, ranges, (size_t)T * 8);
    free(means2D); free(depths); free(conop); free(rgb); free(tiles); free(offs); free(rad_local);
    free(keys); free(vals); free(ranges);
    return (int64_t)D;
}
