/*
 * ORACLE — test infrastructure only.  Nothing under ocrfdet_amd/ may import, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * CPU restatement of the tile-based Gaussian rasteriser forward that OcRFDet calls through
 * `diff_gaussian_rasterization` (mmdet3d/models/necks/MVSGaussian/lib/gaussian_renderer/__init__.py:14,39-70).
 * Paths below are relative to
 *   mmdet3d/models/necks/MVSGaussian/lib/submodules/diff-gaussian-rasterization/
 * (the vendored stock Inria rasteriser @59f5f77, the algorithmic base of the w-depth fork).
 * The restatement keeps the reference's *structure* on purpose (per-Gaussian preprocess ->
 * inclusive scan -> key duplication -> stable sort by (tile | depth bits) -> tile ranges ->
 * per-pixel front-to-back blend) so that it is independent of the HIP path, which bins and
 * sorts differently.
 *
 * PARITY STATUS
 *   colour / final_T / radii / n_contrib : follow cuda_rasterizer/forward.cu, rasterizer_impl.cu,
 *       auxiliary.h line by line (cited below).  The reference sources are CUDA + glm + cub and
 *       cannot be built in this image (no nvcc, no glm, no cub) and the reference holds no
 *       golden vectors for the rasteriser, so this part is pinned only by closed-form renders
 *       (tests/test_oracle_rasterize.py) and an independent numpy restatement: PARITY UNPINNED
 *       against a reference binary.
 *   depth channel : the w-depth fork's source is absent from /root/reference
 *       (diff-gaussian-rasterization-w-depth/ holds a README only).  Implemented from
 *       diff-gaussian-rasterization-w-depth/README.md:5-11: "median depth" = view-space z of the
 *       Gaussian whose blend makes transmittance cross 0.5, default 15.0; a mean-depth mode is
 *       kept as the option the README mentions.  PARITY UNPINNED.
 *
 * Floating point: built with -ffp-contract=off; every fused multiply-add is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16 /* cuda_rasterizer/config.h:15-17 */
#define BLOCK_Y 16
#define NUM_CHANNELS 3

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:41-44 — note the double-precision literals in the reference. */
static inline float ndc2Pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

/* auxiliary.h:46-56 */
static inline void getRect(float px, float py, int max_radius, int gx, int gy, int *rmin, int *rmax)
{
    rmin[0] = imin(gx, imax(0, (int)((px - (float)max_radius) / (float)BLOCK_X)));
    rmin[1] = imin(gy, imax(0, (int)((py - (float)max_radius) / (float)BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, (int)((px + (float)max_radius + (float)(BLOCK_X - 1)) / (float)BLOCK_X)));
    rmax[1] = imin(gy, imax(0, (int)((py + (float)max_radius + (float)(BLOCK_Y - 1)) / (float)BLOCK_Y)));
}

/* auxiliary.h:58-77: row-vector convention, matrix stored transposed. */
static inline void transformPoint4x3(const float *p, const float *m, float *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void transformPoint4x4(const float *p, const float *m, float *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* forward.cu:118-152.  Quaternion (r,x,y,z) is NOT normalised (:127).  glm::mat3 literals are
 * column-major, so with R the usual rotation matrix of q:  Sigma = R diag(s^2) R^T. */
static void computeCov3D(const float *scale, float mod, const float *rot, float *cov3D)
{
    const float sx = mod * scale[0], sy = mod * scale[1], sz = mod * scale[2];
    const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    const float R[3][3] = {
        {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
        {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
        {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    float M[3][3]; /* M[k][i] = s_k * R[i][k]   (glm: M = S * R) */
    for (int i = 0; i < 3; i++) {
        M[0][i] = sx * R[i][0];
        M[1][i] = sy * R[i][1];
        M[2][i] = sz * R[i][2];
    }
    /* Sigma = M^T M */
    float S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            S[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
    cov3D[0] = S[0][0]; cov3D[1] = S[0][1]; cov3D[2] = S[0][2];
    cov3D[3] = S[1][1]; cov3D[4] = S[1][2]; cov3D[5] = S[2][2];
}

/* forward.cu:74-113.  cov2D = (J Rv) Sigma (J Rv)^T with Rv[i][j] = viewmatrix[4*j+i]. */
static void computeCov2D(const float *mean, float focal_x, float focal_y, float tan_fovx,
                         float tan_fovy, const float *cov3D, const float *vm, float *cov)
{
    float t[3];
    transformPoint4x3(mean, vm, t);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
    t[1] = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
    const float j00 = focal_x / t[2], j02 = -(focal_x * t[0]) / (t[2] * t[2]);
    const float j11 = focal_y / t[2], j12 = -(focal_y * t[1]) / (t[2] * t[2]);
    /* A = J * Rv  (rows 0,1 only; row 2 of J is zero) */
    float A[2][3];
    for (int c = 0; c < 3; c++) {
        const float r0 = vm[4 * c + 0], r1 = vm[4 * c + 1], r2 = vm[4 * c + 2]; /* Rv[0..2][c] */
        A[0][c] = j00 * r0 + j02 * r2;
        A[1][c] = j11 * r1 + j12 * r2;
    }
    const float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    float B[2][3]; /* B = A * Sigma */
    for (int i = 0; i < 2; i++)
        for (int c = 0; c < 3; c++)
            B[i][c] = A[i][0] * V[0][c] + A[i][1] * V[1][c] + A[i][2] * V[2][c];
    cov[0] = (B[0][0] * A[0][0] + B[0][1] * A[0][1] + B[0][2] * A[0][2]) + 0.3f; /* :110 */
    cov[1] = B[0][0] * A[1][0] + B[0][1] * A[1][1] + B[0][2] * A[1][2];
    cov[2] = (B[1][0] * A[1][0] + B[1][1] * A[1][1] + B[1][2] * A[1][2]) + 0.3f; /* :111 */
}

typedef struct { uint64_t key; uint32_t val; uint32_t pos; } kv_t;
static int kv_cmp(const void *a, const void *b)
{
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);   /* cub radix sort is stable */
}

/* rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

/*
 * Whole forward (rasterizer_impl.cu:198-336 + forward.cu:155-256,261-374).
 *  depth_mode 0 = median depth (w-depth default), 1 = mean depth (the README's alternative).
 * Outputs (all caller-allocated):
 *  out_color (3,H,W), out_depth (H,W), out_final_T (H,W), out_n_contrib (H,W) uint32,
 *  radii (P) int32, and the per-Gaussian state for stage-wise parity checks:
 *  means2D (P,2), depths (P), conic_opacity (P,4), tiles_touched (P) uint32 (values are only
 *  defined where radii>0, exactly as in the reference, but are zero-initialised here).
 * Returns num_rendered (R), or -1 on allocation failure.
 */
long oracle_rasterize_forward(int P, const float *background, int W, int H, const float *means3D,
                              const float *colors_precomp, const float *opacities,
                              const float *scales, float scale_modifier, const float *rotations,
                              const float *viewmatrix, const float *projmatrix, float tan_fovx,
                              float tan_fovy, int depth_mode, float *out_color, float *out_depth,
                              float *out_final_T, uint32_t *out_n_contrib, int *radii,
                              float *means2D, float *depths, float *conic_opacity,
                              uint32_t *tiles_touched)
{
    const float focal_y = (float)H / (2.0f * tan_fovy); /* rasterizer_impl.cu:222-223 */
    const float focal_x = (float)W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const long npix = (long)W * H;

    /* rasterize_points.cu:68-69 — outputs are zero-filled first. */
    memset(out_color, 0, sizeof(float) * 3 * npix);
    memset(out_depth, 0, sizeof(float) * npix);
    memset(out_final_T, 0, sizeof(float) * npix);
    memset(out_n_contrib, 0, sizeof(uint32_t) * npix);
    if (P == 0) return 0;
    memset(means2D, 0, sizeof(float) * 2 * (size_t)P);
    memset(depths, 0, sizeof(float) * (size_t)P);
    memset(conic_opacity, 0, sizeof(float) * 4 * (size_t)P);

    /* ---- preprocessCUDA, forward.cu:155-256 ---- */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;
        const float *p_orig = means3D + 3 * (size_t)idx;
        float p_view[3];
        transformPoint4x3(p_orig, viewmatrix, p_view);
        if (p_view[2] <= 0.2f) continue; /* auxiliary.h:154 */
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        const float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        const float p_proj[2] = {p_hom[0] * p_w, p_hom[1] * p_w};
        float cov3D[6];
        computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, cov3D);
        float cov[3];
        computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, cov);
        const float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        const float det_inv = 1.f / det;
        const float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        const float mid = 0.5f * (cov[0] + cov[2]);
        const float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
        const float pix[2] = {ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H)};
        int rmin[2], rmax[2];
        getRect(pix[0], pix[1], (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        depths[idx] = p_view[2];
        radii[idx] = (int)my_radius;
        means2D[2 * (size_t)idx] = pix[0];
        means2D[2 * (size_t)idx + 1] = pix[1];
        conic_opacity[4 * (size_t)idx + 0] = conic[0];
        conic_opacity[4 * (size_t)idx + 1] = conic[1];
        conic_opacity[4 * (size_t)idx + 2] = conic[2];
        conic_opacity[4 * (size_t)idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }

    /* ---- InclusiveSum, rasterizer_impl.cu:277 ---- */
    uint32_t *offsets = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)P);
    if (!offsets) return -1;
    uint64_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; offsets[i] = (uint32_t)acc; }
    const long R = (long)acc;

    /* ---- duplicateWithKeys, rasterizer_impl.cu:70-111 ---- */
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)(R > 0 ? R : 1));
    if (!kv) { free(offsets); return -1; }
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
            int rmin[2], rmax[2];
            getRect(means2D[2 * (size_t)idx], means2D[2 * (size_t)idx + 1], radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    kv[off].key = key; kv[off].val = (uint32_t)idx; kv[off].pos = off;
                    off++;
                }
        }
    }
    free(offsets);

    /* ---- SortPairs on bits [0, 32+msb), rasterizer_impl.cu:300-308 (stable) ---- */
    const int bit = (int)getHigherMsb((uint32_t)(gx * gy));
    const uint64_t mask = (32 + bit >= 64) ? ~0ull : ((1ull << (32 + bit)) - 1ull);
    for (long i = 0; i < R; i++) kv[i].key &= mask; /* radix sort only looks at these bits */
    qsort(kv, (size_t)R, sizeof(kv_t), kv_cmp);

    /* ---- identifyTileRanges, rasterizer_impl.cu:116-138 (ranges zero-initialised :310) ---- */
    const int ntiles = gx * gy;
    uint32_t *ranges = (uint32_t *)calloc((size_t)ntiles * 2, sizeof(uint32_t));
    if (!ranges) { free(kv); return -1; }
    for (long i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(kv[i].key >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(kv[i - 1].key >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }

    /* ---- renderCUDA, forward.cu:261-374 (+ w-depth README:5-11 for the depth channel) ---- */
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < ntiles; tile++) {
        const int ty = tile / gx, tx = tile % gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px >= W || py >= H) continue;
                const long pix_id = (long)W * py + px;
                const float pixf[2] = {(float)px, (float)py};
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float C[NUM_CHANNELS] = {0, 0, 0};
                float D = depth_mode == 0 ? 15.0f : 0.0f;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t id = kv[k].val;
                    const float dx = means2D[2 * (size_t)id] - pixf[0];
                    const float dy = means2D[2 * (size_t)id + 1] - pixf[1];
                    const float *co = conic_opacity + 4 * (size_t)id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = fminf_(0.99f, co[3] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break; /* done = true */
                    for (int ch = 0; ch < NUM_CHANNELS; ch++)
                        C[ch] += colors_precomp[(size_t)id * NUM_CHANNELS + ch] * alpha * T;
                    if (depth_mode == 0) {
                        if (T > 0.5f && test_T < 0.5f) D = depths[id];
                    } else {
                        D += depths[id] * alpha * T;
                    }
                    T = test_T;
                    last_contributor = contributor;
                }
                out_final_T[pix_id] = T;
                out_n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < NUM_CHANNELS; ch++)
                    out_color[(size_t)ch * npix + pix_id] = C[ch] + T * background[ch];
                out_depth[pix_id] = D;
            }
    }
    free(ranges);
    free(kv);
    return R;
}
