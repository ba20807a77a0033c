/*
 * oracle/zephyr_oracle.c -- CPU restatement of the Zephyr hypothesis-scoring hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's shared object.  The product path
 * (ossid_code_amd/) never imports, links or calls anything under oracle/.
 *
 * PARITY UNPINNED: the algorithm restated here lives in the third-party package
 * `zephyr` (github.com/r-pad/zephyr, default branch, no pinned version:
 * /root/reference/readme.md:36-51; env.yml has no entry) and, for the scorer, in
 * `pointnet2_ops` (erikwijmans/Pointnet2_PyTorch v3.0.0) which zephyr wraps.  Neither
 * source tree is under /root/reference and the reference holds no tests, golden vectors or
 * fixtures for this path.  What IS anchored on the reference is the call-site
 * contract:
 *   - dict packing, blur(5x5, sigma 0) + /255        utils/zephyr_utils.py:13-26
 *   - getPointNetData(data, return_uv_original=True)  utils/zephyr_utils.py:31
 *   - model({"point_x": ...}) -> one score per hypo   utils/zephyr_utils.py:34
 *   - in-place filtering of transforms / pp_err       utils/zephyr_utils.py:39-43
 *   - integer uv[..., 0]=x(col), uv[..., 1]=y(row), out-of-bounds test, `uv[invalid]=0`
 *                                                     utils/zephyr_utils.py:58-65
 *   - dataset="HSVD_diff_uv_norm", no_valid_proj, no_valid_depth, inconst_ratio_th
 *     (100 LM-O / 10 YCB-V, a percentage)             scripts/online_learning.py:174,184,191-196
 *   - PointNet2SSG(dim_point, args, num_class=1)      scripts/online_learning.py:212-227
 * Everything else (channel order, normalisation, filter margin, accumulation order) is this
 * build's own specification, written down in SPEC.md; this file is its executable form and
 * the HIP path must agree with it bit for bit.
 *
 * Arithmetic contract shared with the HIP kernels (SPEC.md section 2):
 *   - every float op is IEEE-754 binary32, round-to-nearest-even, NO fused contraction
 *     (build with -ffp-contract=off) except where fmaf() is written explicitly;
 *   - dense layers accumulate as one fmaf chain per output, started from the folded bias,
 *     walking the input channels in CANONICAL order: 8-blocks ascending, and inside a block
 *     the offsets 0,4,1,5,2,6,3,7 (this is the k order of v_mfma_f32_32x32x2_f32 when the
 *     previous layer's accumulator tile is fed straight back as the B operand).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define OZR_OK 0
#define OZR_EINVAL (-22)

int ozr_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Z0: cv2.GaussianBlur(img, (5,5), 0) on u8 RGB  (utils/zephyr_utils.py:13).
 * OpenCV semantics restated: ksize 5 with sigma 0 selects the fixed binomial kernel
 * [1,4,6,4,1]/16; borders are BORDER_REFLECT_101; u8 images go through exact fixed-point
 * arithmetic and are rounded half-up once, after both passes: out = (S + 128) >> 8 with
 * S = sum_ij k_i k_j v_ij, sum k_i k_j = 256.
 * ------------------------------------------------------------------------------------------ */
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

int ozr_blur5_u8(const uint8_t* img, int H, int W, int C, uint8_t* out) {
    static const int k[5] = {1, 4, 6, 4, 1};
    if (H <= 0 || W <= 0 || C <= 0) return OZR_EINVAL;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < C; ++c) {
                int S = 0;
                for (int dy = -2; dy <= 2; ++dy) {
                    int yy = reflect101(y + dy, H);
                    int rowsum = 0;
                    for (int dx = -2; dx <= 2; ++dx) {
                        int xx = reflect101(x + dx, W);
                        rowsum += k[dx + 2] * (int)img[((size_t)yy * W + xx) * C + c];
                    }
                    S += k[dy + 2] * rowsum;
                }
                out[((size_t)y * W + x) * C + c] = (uint8_t)((S + 128) >> 8);
            }
    return OZR_OK;
}

/* u8 -> float in [0,1]: the reference divides in float64 and getPointNetData casts to
 * float32 (zephyr_utils.py:14); (float)((double)v/255.0) == (float)v/255.0f for all 256
 * values (checked in tests/test_oracle.py), so the f32 division is used everywhere. */
int ozr_u8_to_unit(const uint8_t* in, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) out[i] = (float)in[i] / 255.0f;
    return OZR_OK;
}

/* Interleave colour and depth into the staged frame layout rgbd[H][W][4] = (r,g,b,depth). */
int ozr_pack_rgbd(const float* rgb, const float* depth, int H, int W, float* rgbd) {
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        rgbd[4 * i + 0] = rgb[3 * i + 0];
        rgbd[4 * i + 1] = rgb[3 * i + 1];
        rgbd[4 * i + 2] = rgb[3 * i + 2];
        rgbd[4 * i + 3] = depth[i];
    }
    return OZR_OK;
}

/* matplotlib.colors.rgb_to_hsv restated for one pixel (the "HSV" token of
 * dataset="HSVD_diff_uv_norm", online_learning.py:192). */
static inline void rgb2hsv(float r, float g, float b, float* h, float* s, float* v) {
    float mx = fmaxf(r, fmaxf(g, b));
    float mn = fminf(r, fminf(g, b));
    float delta = mx - mn;
    float ss = 0.0f, hh = 0.0f;
    if (mx > 0.0f) ss = delta / mx;
    if (delta > 0.0f) {
        if (r == mx)
            hh = (g - b) / delta;
        else if (g == mx)
            hh = 2.0f + (b - r) / delta;
        else
            hh = 4.0f + (r - g) / delta;
        hh = hh / 6.0f;
        if (hh < 0.0f) hh = hh + 1.0f; /* python (h/6) % 1.0 on (-1/6, 5/6] */
    }
    *h = hh;
    *s = ss;
    *v = mx;
}

int ozr_rgb_to_hsv(const float* rgb, size_t n, float* hsv) {
    for (size_t i = 0; i < n; ++i)
        rgb2hsv(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], &hsv[3 * i], &hsv[3 * i + 1], &hsv[3 * i + 2]);
    return OZR_OK;
}

/* ------------------------------------------------------------------------------------------
 * Z1: projectPointsUv(pose_hypos, model_points, meta_data) -> int uv[N,M,2]
 *     (call site utils/zephyr_utils.py:58; K2meta utils/__init__.py:148-156).
 * p' = R p + t with the sum order ((r0*x + r1*y) + r2*z) + t; u = (x'/z')*fx + cx, truncated
 * toward zero like torch .long().  z' <= 1e-6 or a non-finite / out-of-int-range coordinate
 * gives the marker (-1,-1), which every caller treats as out of bounds.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float x, y, z;
} v3;

static inline v3 rot(const float* T, float x, float y, float z) {
    v3 o;
    o.x = (T[0] * x + T[1] * y) + T[2] * z;
    o.y = (T[4] * x + T[5] * y) + T[6] * z;
    o.z = (T[8] * x + T[9] * y) + T[10] * z;
    return o;
}

static inline void project1(const float* T, const float* p, float fx, float fy, float cx, float cy,
                            v3* cam, float* uf, float* vf, int* u, int* v) {
    v3 c = rot(T, p[0], p[1], p[2]);
    c.x = c.x + T[3];
    c.y = c.y + T[7];
    c.z = c.z + T[11];
    *cam = c;
    int ok = c.z > 1e-6f;
    float a = 0.0f, b = 0.0f;
    if (ok) {
        a = (c.x / c.z) * fx + cx;
        b = (c.y / c.z) * fy + cy;
        ok = isfinite(a) && isfinite(b) && fabsf(a) < 1.0e9f && fabsf(b) < 1.0e9f;
    }
    *uf = a;
    *vf = b;
    if (ok) {
        *u = (int)a; /* C cast truncates toward zero == torch .long() */
        *v = (int)b;
    } else {
        *u = -1;
        *v = -1;
    }
}

int ozr_project_uv(const float* T, const float* pts, int N, int M, float fx, float fy, float cx,
                   float cy, int32_t* uv) {
    if (N < 0 || M < 0) return OZR_EINVAL;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m) {
            v3 cam;
            float uf, vf;
            int u, v;
            project1(T + 16 * (size_t)n, pts + 3 * (size_t)m, fx, fy, cx, cy, &cam, &uf, &vf, &u, &v);
            uv[((size_t)n * M + m) * 2 + 0] = u;
            uv[((size_t)n * M + m) * 2 + 1] = v;
        }
    return OZR_OK;
}

/* ------------------------------------------------------------------------------------------
 * Z2: ScoreDataset.getPointNetData (call site utils/zephyr_utils.py:31), per hypothesis.
 * point_x[n][m][8] = (x, y, 0, dH, dS, dV, dD, cosN)   -- SPEC.md section 3.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float r, g, b, d;
} px4;

static inline px4 fetch(const float* rgbd, int W, int u, int v) {
    const float* p = rgbd + 4 * ((size_t)v * W + u);
    px4 o = {p[0], p[1], p[2], p[3]};
    return o;
}

static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* observation at a projected point. interp 0: the pixel (u,v); interp 1: bilinear over the four
 * pixel centres around (uf,vf) (taps clamped to the frame); depth falls back to the nearest
 * pixel when any of its four taps is invalid (0). */
static inline px4 observe(const float* rgbd, int H, int W, int u, int v, float uf, float vf, int interp) {
    px4 c = fetch(rgbd, W, u, v);
    if (!interp) return c;
    float xf = uf - 0.5f, yf = vf - 0.5f;
    float x0f = floorf(xf), y0f = floorf(yf);
    float wx = xf - x0f, wy = yf - y0f;
    int x0 = clampi((int)x0f, 0, W - 1), x1 = clampi((int)x0f + 1, 0, W - 1);
    int y0 = clampi((int)y0f, 0, H - 1), y1 = clampi((int)y0f + 1, 0, H - 1);
    px4 a = fetch(rgbd, W, x0, y0), b = fetch(rgbd, W, x1, y0);
    px4 cc = fetch(rgbd, W, x0, y1), d = fetch(rgbd, W, x1, y1);
    float w00 = (1.0f - wx) * (1.0f - wy), w10 = wx * (1.0f - wy);
    float w01 = (1.0f - wx) * wy, w11 = wx * wy;
    px4 o;
    o.r = ((a.r * w00 + b.r * w10) + cc.r * w01) + d.r * w11;
    o.g = ((a.g * w00 + b.g * w10) + cc.g * w01) + d.g * w11;
    o.b = ((a.b * w00 + b.b * w10) + cc.b * w01) + d.b * w11;
    if (a.d > 0.0f && b.d > 0.0f && cc.d > 0.0f && d.d > 0.0f)
        o.d = ((a.d * w00 + b.d * w10) + cc.d * w01) + d.d * w11;
    else
        o.d = c.d;
    return o;
}

/* model table row, 12 floats: p(3) n(3) hsv(3) pad(3) */
int ozr_prep_model(const float* pts, const float* nrm, const float* rgb, int M, float* tab) {
    for (int m = 0; m < M; ++m) {
        float* t = tab + 12 * (size_t)m;
        t[0] = pts[3 * m];
        t[1] = pts[3 * m + 1];
        t[2] = pts[3 * m + 2];
        t[3] = nrm[3 * m];
        t[4] = nrm[3 * m + 1];
        t[5] = nrm[3 * m + 2];
        rgb2hsv(rgb[3 * m], rgb[3 * m + 1], rgb[3 * m + 2], &t[6], &t[7], &t[8]);
        t[9] = t[10] = t[11] = 0.0f;
    }
    return OZR_OK;
}

/* free-space-violation count of one hypothesis (the "inconst" filter behind
 * inconst_ratio_th, online_learning.py:174,184,196; zephyr_utils.py:42-43): a model point
 * that projects inside the frame onto a valid depth pixel and lies more than `margin`
 * metres in FRONT of the observed surface. Always evaluated on the nearest pixel. */
int ozr_inconst_count(const float* rgbd, int H, int W, const float* T, int N, const float* tab, int M,
                      float fx, float fy, float cx, float cy, float margin, int32_t* count) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        int cnt = 0;
        for (int m = 0; m < M; ++m) {
            v3 cam;
            float uf, vf;
            int u, v;
            project1(T + 16 * (size_t)n, tab + 12 * (size_t)m, fx, fy, cx, cy, &cam, &uf, &vf, &u, &v);
            int inb = (u >= 0) && (u < W) && (v >= 0) && (v < H);
            if (!inb) continue;
            float d = rgbd[4 * ((size_t)v * W + u) + 3];
            if (d > 0.0f && (d - cam.z) > margin) ++cnt;
        }
        count[n] = cnt;
    }
    return OZR_OK;
}

/* Featurize hypotheses sel[0..Nsel) (sel == NULL: 0..Nsel-1) into point_x[Nsel][M][8] and,
 * if uv_out != NULL, uv_original[Nsel][M][2]. */
int ozr_featurize(const float* rgbd, int H, int W, const float* T, const int32_t* sel, int Nsel,
                  const float* tab, int M, float fx, float fy, float cx, float cy, int interp,
                  float* point_x, int32_t* uv_out) {
    if (Nsel < 0 || M <= 0) return OZR_EINVAL;
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < Nsel; ++i) {
        const float* Tn = T + 16 * (size_t)(sel ? sel[i] : i);
        float* out = point_x + (size_t)i * M * 8;
        int* ucl = (int*)malloc(sizeof(int) * 2 * (size_t)M);
        long su = 0, sv = 0;
        for (int m = 0; m < M; ++m) {
            const float* t = tab + 12 * (size_t)m;
            v3 cam;
            float uf, vf;
            int u, v;
            project1(Tn, t, fx, fy, cx, cy, &cam, &uf, &vf, &u, &v);
            if (uv_out) {
                uv_out[((size_t)i * M + m) * 2 + 0] = u;
                uv_out[((size_t)i * M + m) * 2 + 1] = v;
            }
            int inb = (u >= 0) && (u < W) && (v >= 0) && (v < H);
            if (!inb) { /* uv[invalid_proj] = 0  (zephyr_utils.py:63) -> pixel (0,0), nearest */
                u = 0;
                v = 0;
            }
            ucl[2 * m] = u;
            ucl[2 * m + 1] = v;
            su += u;
            sv += v;
            px4 o = observe(rgbd, H, W, u, v, uf, vf, interp && inb);
            float oh, os, ov;
            rgb2hsv(o.r, o.g, o.b, &oh, &os, &ov);
            float dh = fabsf(oh - t[6]);
            dh = fminf(dh, 1.0f - dh);
            float* f = out + 8 * (size_t)m;
            f[2] = 0.0f;
            f[3] = dh;
            f[4] = os - t[7];
            f[5] = ov - t[8];
            f[6] = (o.d > 0.0f) ? (o.d - cam.z) : 0.0f;
            v3 nr = rot(Tn, t[3], t[4], t[5]);
            float dot = (nr.x * cam.x + nr.y * cam.y) + nr.z * cam.z;
            float len = sqrtf((cam.x * cam.x + cam.y * cam.y) + cam.z * cam.z);
            f[7] = (len > 0.0f) ? dot / len : 0.0f;
        }
        /* uv normalisation: centre on the mean pixel, scale by the largest |offset| */
        float mu = (float)su / (float)M, mv = (float)sv / (float)M;
        float ext = 0.0f;
        for (int m = 0; m < M; ++m) {
            ext = fmaxf(ext, fabsf((float)ucl[2 * m] - mu));
            ext = fmaxf(ext, fabsf((float)ucl[2 * m + 1] - mv));
        }
        if (!(ext > 0.0f)) ext = 1.0f;
        for (int m = 0; m < M; ++m) {
            out[8 * (size_t)m + 0] = ((float)ucl[2 * m] - mu) / ext;
            out[8 * (size_t)m + 1] = ((float)ucl[2 * m + 1] - mv) / ext;
        }
        free(ucl);
    }
    return OZR_OK;
}

/* ------------------------------------------------------------------------------------------
 * Z3: PointNet2SSG (online_learning.py:212-227), restating pointnet2_ops v3.0.0:
 * furthest_point_sample, ball_query, QueryAndGroup/GroupAll, shared MLP (1x1 conv + BN + ReLU,
 * BN folded: SPEC.md 4.2), max-pool, FC head.
 * ------------------------------------------------------------------------------------------ */

/* furthest_point_sample: starts at index 0; a point with |p|^2 <= 1e-3 is never a candidate
 * and its running distance is not updated (pointnet2_ops sampling_gpu.cu); ties -> lowest index. */
static void fps(const float* xyz, int stride, int n, int npoint, int32_t* idx, float* tmp) {
    for (int k = 0; k < n; ++k) tmp[k] = 1e10f;
    int old = 0;
    idx[0] = 0;
    for (int j = 1; j < npoint; ++j) {
        float x1 = xyz[(size_t)old * stride], y1 = xyz[(size_t)old * stride + 1], z1 = xyz[(size_t)old * stride + 2];
        float best = -1.0f;
        int besti = 0;
        for (int k = 0; k < n; ++k) {
            float x2 = xyz[(size_t)k * stride], y2 = xyz[(size_t)k * stride + 1], z2 = xyz[(size_t)k * stride + 2];
            float mag = (x2 * x2 + y2 * y2) + z2 * z2;
            if (mag <= 1e-3f) continue;
            float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
            float d = (dx * dx + dy * dy) + dz * dz;
            float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            if (d2 > best) {
                best = d2;
                besti = k;
            }
        }
        old = besti;
        idx[j] = old;
    }
}

/* ball_query: first nsample indices (ascending) with d2 < r^2; short lists padded with the
 * first hit (pointnet2_ops ball_query_gpu.cu). */
static void ball_query(const float* xyz, int stride, int n, const float* cen, int npoint, float radius,
                       int nsample, int32_t* idx) {
    float r2 = radius * radius;
    for (int j = 0; j < npoint; ++j) {
        float cx = cen[3 * j], cy = cen[3 * j + 1], cz = cen[3 * j + 2];
        int cnt = 0;
        int32_t* o = idx + (size_t)j * nsample;
        for (int k = 0; k < n && cnt < nsample; ++k) {
            float dx = cx - xyz[(size_t)k * stride], dy = cy - xyz[(size_t)k * stride + 1],
                  dz = cz - xyz[(size_t)k * stride + 2];
            float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < r2) {
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) o[l] = k;
                o[cnt++] = k;
            }
        }
        if (cnt == 0) /* cannot happen when centres are members of the set; keep defined */
            for (int l = 0; l < nsample; ++l) o[l] = 0;
    }
}

#define SBLK 64
static const int CANON[8] = {0, 4, 1, 5, 2, 6, 3, 7};

/* y[o] = relu?( b[o] + sum_k W[o][k] x[k] ), canonical fmaf chain; W is [cout][kpad], kpad % 8 == 0 */
static inline void dense(const float* W, const float* b, int kpad, int cout, const float* x, float* y, int relu) {
    for (int o = 0; o < cout; ++o) {
        const float* w = W + (size_t)o * kpad;
        float acc = b[o];
        for (int kb = 0; kb < kpad; kb += 8)
            for (int i = 0; i < 8; ++i) acc = fmaf(w[kb + CANON[i]], x[kb + CANON[i]], acc);
        y[o] = relu ? fmaxf(acc, 0.0f) : acc;
    }
}

/* EXPERIMENT ONLY (tools/six_product_emulation.py): product mode 2 emulates on the CPU the arithmetic a
 * six-product split-bf16 form of SA1 / SA2 would have on the matrix cores -- x = p0 + p1 + p2 with bf16 pieces (24
 * significant bits), the six products with i + j <= 2, every v_mfma_f32_32x32x16_bf16 modelled as "the exact dot product
 * of one 16-deep slice, added to the f32 accumulator with one rounding", smallest terms first -- to measure whether the
 * hypothesis rank order would survive it. Mode 0 (default) is SPEC.md's fmaf chain, the only arithmetic the product and
 * every parity test use. */
static int g_product_mode = 0;
int ozr_set_product_mode(int mode) {
    g_product_mode = mode;
    return OZR_OK;
}
static inline float bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    float y;
    memcpy(&y, &u, 4);
    return y;
}
static inline void split3(float x, float* p0, float* p1, float* p2) {
    *p0 = bf16_rne(x);
    float r = x - *p0;
    *p1 = bf16_rne(r);
    r -= *p1;
    *p2 = bf16_rne(r);
}
static const int SIX_I[6] = {2, 0, 1, 1, 0, 0}, SIX_J[6] = {0, 2, 1, 0, 1, 0}; /* smallest terms first */

/* mode-2 form of dense_block: acc[s] continues from acc0[s] (NULL: the bias) over input channels [k0, k1) of W's rows */
static void dense_block_six(const float* W, const float* b, int kpad, int cout, const float* xs, float* ys, int S, int relu,
                            const float* acc0, int k0, int k1, int finish) {
    const int K = k1 - k0, K16 = (K + 15) / 16 * 16;
    float* xp = (float*)calloc((size_t)3 * K16 * S, sizeof(float));
    float* wp = (float*)calloc((size_t)3 * K16, sizeof(float));
    for (int k = 0; k < K; ++k)
        for (int s = 0; s < S; ++s)
            split3(xs[(size_t)(k0 + k) * S + s], xp + ((size_t)0 * K16 + k) * S + s, xp + ((size_t)1 * K16 + k) * S + s,
                   xp + ((size_t)2 * K16 + k) * S + s);
    for (int o = 0; o < cout; ++o) {
        const float* w = W + (size_t)o * kpad + k0;
        for (int k = 0; k < K; ++k) split3(w[k], wp + k, wp + K16 + k, wp + 2 * K16 + k);
        float acc[SBLK];
        for (int s = 0; s < S; ++s) acc[s] = acc0 ? acc0[(size_t)o * S + s] : b[o];
        for (int sl = 0; sl < K16; sl += 16)
            for (int t = 0; t < 6; ++t) {
                const float* wi = wp + (size_t)SIX_I[t] * K16 + sl;
                const float* xj = xp + ((size_t)SIX_J[t] * K16 + sl) * S;
                double d[SBLK];
                for (int s = 0; s < S; ++s) d[s] = 0.0;
                for (int k = 0; k < 16; ++k) {
                    const double wk = (double)wi[k];
                    const float* xk = xj + (size_t)k * S;
                    for (int s = 0; s < S; ++s) d[s] += wk * (double)xk[s];
                }
                for (int s = 0; s < S; ++s) acc[s] = (float)((double)acc[s] + d[s]);
            }
        float* yo = ys + (size_t)o * S;
        for (int s = 0; s < S; ++s) yo[s] = (relu && finish) ? fmaxf(acc[s], 0.0f) : acc[s];
    }
    free(xp);
    free(wp);
}

/* same layer over a block of S samples held channel-major xs[k][S]; the inner loop runs over
 * samples so the compiler can vectorise while every sample keeps its own exact fmaf chain. */
#define SBLK_IS_DEFINED_ABOVE 1
static void dense_block(const float* W, const float* b, int kpad, int cout, const float* xs, float* ys, int S,
                        int relu) {
    if (g_product_mode == 2 && kpad <= 136) {          /* SA1 / SA2 layers only (kpad 8, 64, 128, 136) */
        dense_block_six(W, b, kpad, cout, xs, ys, S, relu, NULL, 0, kpad, 1);
        return;
    }
    for (int o = 0; o < cout; ++o) {
        const float* w = W + (size_t)o * kpad;
        float acc[SBLK];
        for (int s = 0; s < S; ++s) acc[s] = b[o];
        for (int kb = 0; kb < kpad; kb += 8)
            for (int i = 0; i < 8; ++i) {
                int k = kb + CANON[i];
                float wk = w[k];
                const float* xk = xs + (size_t)k * S;
                for (int s = 0; s < S; ++s) acc[s] = fmaf(wk, xk[s], acc[s]);
            }
        float* yo = ys + (size_t)o * S;
        if (relu)
            for (int s = 0; s < S; ++s) yo[s] = fmaxf(acc[s], 0.0f);
        else
            for (int s = 0; s < S; ++s) yo[s] = acc[s];
    }
}

/* Folded-weight bundle, all row-major [cout][kpad] with input channels in the order given in
 * SPEC.md 4.3:
 *   0: SA1 L1 [64][8]    in = (dx,dy,dz,f0..f4)           1: [64][64]   2: [128][64]
 *   3: SA2 L1 [128][136] in = (g0..g127,dx,dy,dz,0*5)      4: [128][128] 5: [256][128]
 *   6: SA3 L1 [256][264] in = (g0..g255,x,y,z,0*5)         7: [512][256] 8: [1024][512]
 *   9: FC1 [512][1024]  10: FC2 [256][512]  11: FC3 [1][256] (no ReLU)
 */
typedef struct {
    const float* W[12];
    const float* b[12];
    int npoint1, nsample1, npoint2, nsample2;
    float radius1, radius2;
} ozr_pn2;

static const int L_K[12] = {8, 64, 64, 136, 128, 128, 264, 256, 512, 1024, 512, 256};
static const int L_C[12] = {64, 64, 128, 128, 128, 256, 256, 512, 1024, 512, 256, 1};

/* Score B hypotheses. Optional debug outputs (may be NULL): fps1[B][np1], ball1[B][np1][ns1],
 * feat1[B][np1][128], fps2[B][np2], ball2[B][np2][ns2], feat2[B][np2][256], feat3[B][1024]. */
int ozr_pn2_score(const float* point_x, int B, int M, const ozr_pn2* P, float* scores, int32_t* dbg_fps1,
                  int32_t* dbg_ball1, float* dbg_feat1, int32_t* dbg_fps2, int32_t* dbg_ball2,
                  float* dbg_feat2, float* dbg_feat3) {
    const int np1 = P->npoint1, ns1 = P->nsample1, np2 = P->npoint2, ns2 = P->nsample2;
    if (B < 0 || M < np1 || np1 < np2 || ns1 != 64 || ns2 != 64 || np2 % 32 != 0) return OZR_EINVAL;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float* px = point_x + (size_t)b * M * 8;
        int32_t* fps1 = (int32_t*)malloc(sizeof(int32_t) * np1);
        int32_t* ball1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)np1 * ns1);
        float* tmp = (float*)malloc(sizeof(float) * (size_t)(M > np1 ? M : np1));
        float* xyz1 = (float*)malloc(sizeof(float) * 3 * np1);
        float* feat1 = (float*)malloc(sizeof(float) * 128 * (size_t)np1);
        float* p2 = (float*)malloc(sizeof(float) * 128 * (size_t)np1);
        int32_t* fps2 = (int32_t*)malloc(sizeof(int32_t) * np2);
        int32_t* ball2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)np2 * ns2);
        float* xyz2 = (float*)malloc(sizeof(float) * 3 * np2);
        float* feat2 = (float*)malloc(sizeof(float) * 256 * (size_t)np2);
        float* bufa = (float*)malloc(sizeof(float) * 1024 * SBLK);
        float* bufb = (float*)malloc(sizeof(float) * 1024 * SBLK);
        if (!fps1 || !ball1 || !tmp || !xyz1 || !feat1 || !p2 || !fps2 || !ball2 || !xyz2 || !feat2 || !bufa ||
            !bufb) {
            err = 1;
        } else {
            /* ---- SA1: FPS(np1) over xyz = point_x[..., 0:3], ball(r1, 64), MLP 8->64->64->128, max */
            fps(px, 8, M, np1, fps1, tmp);
            for (int j = 0; j < np1; ++j)
                for (int c = 0; c < 3; ++c) xyz1[3 * j + c] = px[(size_t)fps1[j] * 8 + c];
            ball_query(px, 8, M, xyz1, np1, P->radius1, ns1, ball1);
            for (int j = 0; j < np1; ++j) {
                for (int s = 0; s < ns1; ++s) {
                    const float* r = px + (size_t)ball1[(size_t)j * ns1 + s] * 8;
                    bufa[0 * ns1 + s] = r[0] - xyz1[3 * j + 0];
                    bufa[1 * ns1 + s] = r[1] - xyz1[3 * j + 1];
                    bufa[2 * ns1 + s] = r[2] - xyz1[3 * j + 2];
                    for (int c = 3; c < 8; ++c) bufa[c * ns1 + s] = r[c];
                }
                dense_block(P->W[0], P->b[0], 8, 64, bufa, bufb, ns1, 1);
                dense_block(P->W[1], P->b[1], 64, 64, bufb, bufa, ns1, 1);
                dense_block(P->W[2], P->b[2], 64, 128, bufa, bufb, ns1, 1);
                for (int o = 0; o < 128; ++o) {
                    float m = bufb[(size_t)o * ns1];
                    for (int s = 1; s < ns1; ++s) m = fmaxf(m, bufb[(size_t)o * ns1 + s]);
                    feat1[(size_t)j * 128 + o] = m;
                }
            }
            /* ---- SA2: FPS(np2) over xyz1, ball(r2, 64), MLP 131->128->128->256, max.
             * L1 walks (g0..g127, dx,dy,dz): the g part does not depend on the centre, so it is
             * evaluated once per point (p2 = chain from the bias over g) and every sample
             * continues that same chain with its three offsets -- bit-identical to running the
             * whole 136-long chain per sample. */
            fps(xyz1, 3, np1, np2, fps2, tmp);
            for (int j = 0; j < np2; ++j)
                for (int c = 0; c < 3; ++c) xyz2[3 * j + c] = xyz1[3 * fps2[j] + c];
            ball_query(xyz1, 3, np1, xyz2, np2, P->radius2, ns2, ball2);
            if (g_product_mode == 2) {                 /* experiment: the per-point part of SA2 L1 on six products */
                for (int i0 = 0; i0 < np1; i0 += SBLK) {
                    const int S = (np1 - i0) < SBLK ? (np1 - i0) : SBLK;
                    for (int s = 0; s < S; ++s)
                        for (int c = 0; c < 128; ++c) bufa[(size_t)c * S + s] = feat1[(size_t)(i0 + s) * 128 + c];
                    dense_block_six(P->W[3], P->b[3], 136, 128, bufa, bufb, S, 0, NULL, 0, 128, 0);
                    for (int s = 0; s < S; ++s)
                        for (int o = 0; o < 128; ++o) p2[(size_t)(i0 + s) * 128 + o] = bufb[(size_t)o * S + s];
                }
            } else
            for (int i = 0; i < np1; ++i) {
                const float* g = feat1 + (size_t)i * 128;
                for (int o = 0; o < 128; ++o) {
                    const float* w = P->W[3] + (size_t)o * 136;
                    float acc = P->b[3][o];
                    for (int kb = 0; kb < 128; kb += 8)
                        for (int q = 0; q < 8; ++q) acc = fmaf(w[kb + CANON[q]], g[kb + CANON[q]], acc);
                    p2[(size_t)i * 128 + o] = acc;
                }
            }
            for (int j = 0; j < np2; ++j) {
                if (g_product_mode == 2) {             /* experiment: continue every sample's chain with its offset slice */
                    float* xo = bufb;                  /* [8][ns2] offsets (dx,dy,dz,0..), acc0 [128][ns2] behind it */
                    float* a0 = bufb + 8 * ns2;
                    for (int s = 0; s < ns2; ++s) {
                        int i = ball2[(size_t)j * ns2 + s];
                        for (int c = 0; c < 3; ++c) xo[(size_t)c * ns2 + s] = xyz1[3 * i + c] - xyz2[3 * j + c];
                        for (int c = 3; c < 8; ++c) xo[(size_t)c * ns2 + s] = 0.0f;
                        for (int o = 0; o < 128; ++o) a0[(size_t)o * ns2 + s] = p2[(size_t)i * 128 + o];
                    }
                    /* W rows are [136]: the offset block sits at columns 128..135; xs is indexed from k0 */
                    dense_block_six(P->W[3], P->b[3], 136, 128, xo - (size_t)128 * ns2, bufa, ns2, 1, a0, 128, 136, 1);
                } else
                for (int s = 0; s < ns2; ++s) {
                    int i = ball2[(size_t)j * ns2 + s];
                    float dx = xyz1[3 * i] - xyz2[3 * j], dy = xyz1[3 * i + 1] - xyz2[3 * j + 1],
                          dz = xyz1[3 * i + 2] - xyz2[3 * j + 2];
                    for (int o = 0; o < 128; ++o) {
                        const float* w = P->W[3] + (size_t)o * 136 + 128;
                        float acc = p2[(size_t)i * 128 + o];
                        acc = fmaf(w[0], dx, acc); /* block (dx,dy,dz,0,0,0,0,0) in order 0,4,1,5,2,6,3,7 */
                        acc = fmaf(w[4], 0.0f, acc);
                        acc = fmaf(w[1], dy, acc);
                        acc = fmaf(w[5], 0.0f, acc);
                        acc = fmaf(w[2], dz, acc);
                        acc = fmaf(w[6], 0.0f, acc);
                        acc = fmaf(w[3], 0.0f, acc);
                        acc = fmaf(w[7], 0.0f, acc);
                        bufa[(size_t)o * ns2 + s] = fmaxf(acc, 0.0f);
                    }
                }
                dense_block(P->W[4], P->b[4], 128, 128, bufa, bufb, ns2, 1);
                dense_block(P->W[5], P->b[5], 128, 256, bufb, bufa, ns2, 1);
                for (int o = 0; o < 256; ++o) {
                    float m = bufa[(size_t)o * ns2];
                    for (int s = 1; s < ns2; ++s) m = fmaxf(m, bufa[(size_t)o * ns2 + s]);
                    feat2[(size_t)j * 256 + o] = m;
                }
            }
            /* ---- SA3: GroupAll over the np2 points, in = (g0..g255, x,y,z, 0*5) -> 256->512->1024, max */
            float f3[1024];
            for (int o = 0; o < 1024; ++o) f3[o] = -INFINITY;
            for (int s0 = 0; s0 < np2; s0 += SBLK) {
                int S = (np2 - s0) < SBLK ? (np2 - s0) : SBLK;
                for (int s = 0; s < S; ++s) {
                    int i = s0 + s;
                    for (int c = 0; c < 256; ++c) bufa[(size_t)c * S + s] = feat2[(size_t)i * 256 + c];
                    for (int c = 0; c < 3; ++c) bufa[(size_t)(256 + c) * S + s] = xyz2[3 * i + c];
                    for (int c = 259; c < 264; ++c) bufa[(size_t)c * S + s] = 0.0f;
                }
                dense_block(P->W[6], P->b[6], 264, 256, bufa, bufb, S, 1);
                dense_block(P->W[7], P->b[7], 256, 512, bufb, bufa, S, 1);
                dense_block(P->W[8], P->b[8], 512, 1024, bufa, bufb, S, 1);
                for (int o = 0; o < 1024; ++o)
                    for (int s = 0; s < S; ++s) f3[o] = fmaxf(f3[o], bufb[(size_t)o * S + s]);
            }
            /* ---- FC head 1024 -> 512 -> 256 -> 1 */
            float h1[512], h2[256], out;
            dense(P->W[9], P->b[9], 1024, 512, f3, h1, 1);
            dense(P->W[10], P->b[10], 512, 256, h1, h2, 1);
            dense(P->W[11], P->b[11], 256, 1, h2, &out, 0);
            scores[b] = out;

            if (dbg_fps1) memcpy(dbg_fps1 + (size_t)b * np1, fps1, sizeof(int32_t) * np1);
            if (dbg_ball1) memcpy(dbg_ball1 + (size_t)b * np1 * ns1, ball1, sizeof(int32_t) * (size_t)np1 * ns1);
            if (dbg_feat1) memcpy(dbg_feat1 + (size_t)b * np1 * 128, feat1, sizeof(float) * 128 * (size_t)np1);
            if (dbg_fps2) memcpy(dbg_fps2 + (size_t)b * np2, fps2, sizeof(int32_t) * np2);
            if (dbg_ball2) memcpy(dbg_ball2 + (size_t)b * np2 * ns2, ball2, sizeof(int32_t) * (size_t)np2 * ns2);
            if (dbg_feat2) memcpy(dbg_feat2 + (size_t)b * np2 * 256, feat2, sizeof(float) * 256 * (size_t)np2);
            if (dbg_feat3) memcpy(dbg_feat3 + (size_t)b * 1024, f3, sizeof(float) * 1024);
        }
        free(fps1);
        free(ball1);
        free(tmp);
        free(xyz1);
        free(feat1);
        free(p2);
        free(fps2);
        free(ball2);
        free(xyz2);
        free(feat2);
        free(bufa);
        free(bufb);
    }
    (void)L_K;
    (void)L_C;
    return err ? -12 : OZR_OK;
}

/* stand-alone stage entry points for stage-wise parity tests */
int ozr_fps(const float* xyz, int stride, int B, int n, int npoint, int32_t* idx) {
    if (n <= 0 || npoint <= 0) return OZR_EINVAL;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        float* tmp = (float*)malloc(sizeof(float) * n);
        fps(xyz + (size_t)b * n * stride, stride, n, npoint, idx + (size_t)b * npoint, tmp);
        free(tmp);
    }
    return OZR_OK;
}

int ozr_ball_query(const float* xyz, int stride, int B, int n, const float* cen, int npoint, float radius,
                   int nsample, int32_t* idx) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        ball_query(xyz + (size_t)b * n * stride, stride, n, cen + (size_t)b * npoint * 3, npoint, radius, nsample,
                   idx + (size_t)b * npoint * nsample);
    return OZR_OK;
}
